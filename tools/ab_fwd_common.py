"""engine_for(spec): an Engine on a chosen build of the library, several builds side by side in one process (tools/ab_fwd.py, ab_spec.py).
spec = 'product' | path | NAME (pyspecsdr_amd/libpss_NAME.so), optionally ':opt=val,opt=val' (pss_set_option)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspecsdr_amd import _lib as L
from pyspecsdr_amd.engine import Engine


def engine_for(spec):
    path, _, opts = spec.partition(":")
    if path in ("", "product"):
        path = os.path.join(ROOT, "pyspecsdr_amd", "libpss.so")
    elif not os.path.exists(path):
        path = os.path.join(ROOT, "pyspecsdr_amd", f"libpss_{path}.so")
    L._lib, L.LIB_PATH = None, path          # a fresh CDLL per build (distinct files: distinct handles)
    full = dict(L._SIGS)
    probe = C.CDLL(path)
    for name in list(L._SIGS):               # an older build of the ABI (a baseline from another commit) lacks the newest entry points
        if not hasattr(probe, name):
            del L._SIGS[name]
    try:
        e = Engine(0)
    finally:
        L._SIGS.clear()
        L._SIGS.update(full)
    for kv in filter(None, opts.split(",")):
        k, _, v = kv.partition("=")
        e.set_option(k, int(v))
    return e


