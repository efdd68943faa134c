#!/usr/bin/env python3
"""Compare the oracle's np.log10 float32 model (oracle/pss_oracle.c pss_o_log10f_np — the same arithmetic as the device's
log10f_np in pss_device.h) with NumPy on EVERY positive finite float32.  Needs a NumPy whose float32 log10 dispatches to
SVML (x86-64 with AVX512_SKX; the build container's does).  ~40 s.
    python tools/check_log10f_model.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib as O

t0, bad, tot, CH = time.time(), 0, 0, 1 << 24
for c in range(0, 0x7f800000, CH):
    x = np.arange(max(c, 1), min(c + CH, 0x7f800000), dtype=np.uint32).view(np.float32)
    ne = O.log10f(x).view(np.uint32) != np.log10(x).view(np.uint32)
    if ne.any() and bad < 10:
        i = np.nonzero(ne)[0][:3]
        print(hex(c), int(ne.sum()), x[i])
    bad += int(ne.sum())
    tot += len(x)
print(f"{tot} values, {bad} differing, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
