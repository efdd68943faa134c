#!/usr/bin/env python3
"""A few plain launches of one spectrum shape, for rocprofv3 counter passes (tools/prof_xl.sh) and quick timing:
    python tools/run_spec.py <n_fft> <n_frames> [scan] [reps]
prints the mean launch time (HIP events) of the kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench_configs as BC  # noqa: E402
from pyspecsdr_amd.engine import Engine  # noqa: E402

n, nf = int(sys.argv[1]), int(sys.argv[2])
scan = "scan" in sys.argv[3:]
reps = int(sys.argv[-1]) if sys.argv[-1].isdigit() and len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
e = Engine(0, order="none")
for a in sys.argv[3:]:
    if "=" in a:
        k, v = a.split("=")
        e.set_option(k, int(v))
iq = BC.synth("scan" if scan else "fm", nf, n, 2.4e6, dev, 7)
db = torch.empty((nf, n), dtype=torch.float32, device=dev)
pk = torch.empty((nf,), dtype=torch.float32, device=dev)
bw = torch.empty((nf,), dtype=torch.float64, device=dev)
cnt = torch.empty((nf,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
fn = (lambda: e.scan(iq, nf, n, 2.4e6, db, pk, bw, cnt)) if scan else (lambda: e.spectrum_db(iq, nf, n, db))
fn(); e.sync()
e.enable_timing(True)
for _ in range(reps):
    fn()
e.sync()
v = e.kernel_times()["k_spectrum"]
bytes_ = nf * n * 12
print(f"k_spectrum n={n} nf={nf} {'scan' if scan else 'db'}: mean {sum(v) / len(v):.4f} ms  min {min(v):.4f} ms  "
      f"{bytes_ / (min(v) * 1e-3) / 1e12:.2f} TB/s (best)  checksum {float(db.double().sum()):.6e}")
