#!/usr/bin/env python3
"""Build a second copy of libpss.so with extra compiler flags (kernel experiments), next to the product library:

    python tools/build_variant.py nohead -DPSS_EXP_NOHEAD        ->  pyspecsdr_amd/libpss_nohead.so
    PSS_LIBRARY=pyspecsdr_amd/libpss_nohead.so python tools/bench_alone.py

The product build (pyspecsdr_amd/build.py) is not touched; variant objects live in pyspecsdr_amd/_build/<name>/."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyspecsdr_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    bdir = os.path.join(B.BUILD, name)
    os.makedirs(bdir, exist_ok=True)
    cc = B.hipcc()
    objs = []
    procs = []
    for src, extra in B.UNITS:
        o = os.path.join(bdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([cc] + B.COMMON + extra + flags + ["-c", os.path.join(B.CSRC, src), "-o", o]))
    if any(p.wait() for p in procs):
        sys.exit("compile failed")
    out = os.path.join(B.HERE, f"libpss_{name}.so")
    subprocess.run([cc, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs, check=True)
    print(out)


if __name__ == "__main__":
    main()
