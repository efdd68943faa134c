#!/usr/bin/env python3
"""Role micro-benchmarks of k_nfm_fwd (pss_ubench.h; needs the variant build: python tools/build_variant.py ubench -DPSS_UBENCH).
Every wavefront of a k_nfm_fwd-shaped launch runs one role on on-chip data — no barriers, no HBM:
    mode 0 FIR worker, 1 IIR wavefront, 2 discriminator, 3 the kernel's role mix (wave 0 IIR, waves 1-3 FIR + discriminator).
Prints ms per launch for 40 chunks of 24 samples (= one 1024-sample frame) and the implied clocks per chunk and wavefront."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

lib = C.CDLL(os.environ.get("PSS_LIBRARY", os.path.join(ROOT, "pyspecsdr_amd", "libpss_ubench.so")))
h = C.c_void_p()
assert lib.pss_create(0, C.byref(h)) == 0
lib.pss_ubench_role.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_float)]
iq = torch.randn((65536, 1024, 2), device="cuda", dtype=torch.float32) * 0.3
y = torch.empty((1024 * 16 * 64,), device="cuda", dtype=torch.float64)
torch.cuda.synchronize()
names = {0: "FIR workers only", 1: "IIR wavefronts only", 2: "discriminator only", 3: "kernel's role mix", 4: "FIR, one code path"}
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4]
best = {}
for rnd in range(4):            # rounds interleave the modes (clock / thermal drift shows up as round-to-round spread); the minimum is reported
    for mode in modes:
        ms = C.c_float()
        r = lib.pss_ubench_role(h, mode, iq.data_ptr(), y.data_ptr(), 1024, 2.4e6, chunks, 40, C.byref(ms))
        assert r == 0, r
        best.setdefault(mode, []).append(ms.value)
for mode in modes:
    v = best[mode]
    clk = min(v) * 1e-3 * 2.3e9 / chunks / 4          # per chunk and wavefront at 4 wavefronts per SIMD, 2.3 GHz
    print(f"mode {mode} {names[mode]:22s} min {min(v):.4f} ms (rounds {' '.join(f'{x:.4f}' for x in v)}) per launch of {chunks} chunks -> {clk:7.0f} clk per chunk and wavefront")
