import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from pyspecsdr_amd.engine import Engine
dev = torch.device("cuda", 0)
e = Engine(0)
nf = 65536
for n in (1024, 1040, 1056, 1088, 1024):
    iq = (torch.randn((nf, n, 2), device=dev) * 0.3).contiguous()
    n_out = e.demod_out_len(0, n, bench.FS)
    pcm = torch.empty((nf, n_out, 2), dtype=torch.int16, device=dev)
    for _ in range(3): e.demod(0, iq, nf, n, bench.FS, pcm, None)
    e.sync(); e.enable_timing(True)
    for _ in range(20): e.demod(0, iq, nf, n, bench.FS, pcm, None)
    e.sync(); kt = e.kernel_times(); e.enable_timing(False)
    f = sum(kt["k_nfm_fwd"]) / len(kt["k_nfm_fwd"]); b = sum(kt["k_nfm_bwd"]) / len(kt["k_nfm_bwd"])
    print(f"n {n}: fwd {f:.4f} ms = {f / n * 1024:.4f} per 1024 samples; bwd {b:.4f}")
    del iq, pcm
