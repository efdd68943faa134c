#!/usr/bin/env python3
"""Second golden set: the reference's NFM path under OTHER NumPy CPU dispatches.

NumPy picks its float32 loops (arctan2, complex multiply, abs) by CPU feature at import time, so the reference's own
bits depend on the host.  The main goldens (tools/make_goldens.py) are stamped AVX512_SKX — what an x86 server runs —
and the oracle / kernels replay exactly that dispatch.  This script re-runs the reference in child interpreters with
NPY_DISABLE_CPU_FEATURES set and stores what THEY produce for the same inputs, so that the host dependence is a
recorded fixture (tests/test_oracle_golden.py::test_reference_is_not_bit_stable_across_cpu_dispatch) instead of prose.

    python tools/make_goldens_dispatch.py          # build container only (imports /root/reference); writes tests/golden/nfm_dispatch.npz
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "nfm_dispatch.npz")
AVX512 = ("AVX512F AVX512CD AVX512VL AVX512BW AVX512DQ AVX512VNNI AVX512IFMA AVX512VBMI AVX512VBMI2 AVX512BITALG AVX512FP16 "
          "AVX512VPOPCNTDQ AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR")
VARIANTS = {"avx2_fma3": AVX512, "baseline": AVX512 + " AVX2 FMA3"}

CHILD = r'''
import json, sys
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import numpy as np
import signal_processing as sp
g = np.load(sys.argv[1])
out = {}
for tag in ("a", "b", "c", "e", "f"):
    iq, fs = g["iq_" + tag], float(g["fs_" + tag])
    aud = np.stack([sp.demodulate_signal(f, fs, "NFM") for f in iq])
    out["pcm_" + tag] = np.int16(aud * 32767)
    x = iq[0]
    out["disc_" + tag] = (np.angle(x[1:] * np.conj(x[:-1])) * (fs / (2 * np.pi))).astype(np.float32)
from numpy._core._multiarray_umath import __cpu_features__ as F
out["features"] = np.array(json.dumps(sorted(k for k, v in F.items() if v)))
np.savez(sys.argv[2], **out)
'''


def main():
    src = os.path.join(ROOT, "tests", "golden", "nfm.npz")
    d = {}
    for name, disable in VARIANTS.items():
        tmp = f"/tmp/nfm_dispatch_{name}.npz"
        env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=disable, PYTHONDONTWRITEBYTECODE="1")
        subprocess.run([sys.executable, "-c", CHILD, src, tmp], check=True, env=env)
        z = np.load(tmp)
        for k in z.files:
            d[f"{name}_{k}"] = z[k]
        print(name, "features:", str(z["features"])[:120])
    d["variants"] = np.array(list(VARIANTS))
    d["note"] = np.array("reference NFM outputs for the inputs of nfm.npz under NPY_DISABLE_CPU_FEATURES (see tools/make_goldens_dispatch.py)")
    np.savez_compressed(OUT, **d)
    # summary against the main (AVX512_SKX) goldens
    g = np.load(src)
    for name in VARIANTS:
        tot = diff = dd = dt = 0
        mx = 0
        for tag in ("a", "b", "c", "e", "f"):
            a, b = g["pcm_" + tag], d[f"{name}_pcm_{tag}"]
            tot += a.size // 2
            diff += int((a[..., 0] != b[..., 0]).sum())
            mx = max(mx, int(np.abs(a.astype(int) - b.astype(int)).max()))
            x, y = g["disc_" + tag], d[f"{name}_disc_{tag}"]
            dd += int((x.view(np.uint32) != y.view(np.uint32)).sum())
            dt += x.size
        print(f"{name}: int16 samples differing from the AVX512_SKX goldens: {diff} of {tot} (max |delta| {mx} LSB); "
              f"float32 discriminator values differing: {dd} of {dt} ({100.0 * dd / dt:.1f} %)")


if __name__ == "__main__":
    main()
