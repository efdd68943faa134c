#!/usr/bin/env python3
"""Scan the kernels' ISA (pyspecsdr_amd/_build/asm/*.s, written by tools/kernel_resources.py) for the gfx950 store-data hazard the
compiler does not cover: a buffer store of more than 64 bits whose scalar-offset operand is an SGPR, followed within two instructions
by a VALU write of one of its data registers.  LLVM's hazard recogniser skips MUBUF stores with a register soffset; on MI355X the data
is sampled late all the same (k_hilbert_xl returned 16-48 wrong samples on a cold launch, round 3).  Exit status 1 if a site is found.

    python tools/check_store_hazard.py [file.s ...]"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAIT_STATES = 2


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    hits = []
    kernel = "?"
    window = []   # (instructions since the store, data registers, text)
    for ln, raw in enumerate(open(path), 1):
        s = raw.split(";")[0].strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel, window = m.group(1), []
            continue
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        op, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(",")]
        if op == "s_nop":
            n = int(ops[0], 0) + 1
            window = [(d + n, r, t) for d, r, t in window if d + n < WAIT_STATES]
            continue
        if op.startswith("v_") and ops:
            w = regs(ops[0])
            for d, r, t in window:
                if w & r:
                    hits.append((path, ln, kernel, t, s))
        window = [(d + 1, r, t) for d, r, t in window if d + 1 < WAIT_STATES]
        if re.match(r"buffer_store_(dwordx[34]|b96|b128)", op) and len(ops) >= 4:
            soff = ops[3].split()[0]
            if re.fullmatch(r"s\d+|m0|ttmp\d+", soff):
                window.append((0, regs(ops[0]), s))
    return hits


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "pyspecsdr_amd", "_build", "asm", "*.s")))
    if not files:
        sys.exit("no ISA files: run tools/kernel_resources.py first")
    hits = [h for f in files for h in scan(f)]
    for path, ln, kernel, st, wr in hits:
        print(f"{os.path.basename(path)}:{ln}: {kernel[:60]}: `{wr}` overwrites data of `{st}`")
    print(f"{len(files)} file(s), {len(hits)} unprotected store-data hazard site(s)")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
