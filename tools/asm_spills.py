#!/usr/bin/env python3
"""Where a kernel spills: scratch loads / stores of one kernel of pyspecsdr_amd/_build/asm/*.s (tools/kernel_resources.py writes them),
listed with the number of s_barrier instructions that precede each (= the phase of the kernel it sits in), plus the instruction mix.

    python tools/asm_spills.py pss_fft k_ssb_hilbert_xlILi2E
"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit, pat = sys.argv[1], sys.argv[2]
txt = open(os.path.join(ROOT, "pyspecsdr_amd", "_build", "asm", unit + ".s")).read()
m = re.search(r"^(_Z\w*%s\w*):" % re.escape(pat), txt, re.M)
if not m:
    sys.exit("no such kernel")
start = m.end()
end = txt.index(".Lfunc_end", start)
body = txt[start:end].split("\n")
print(m.group(1), len(body), "lines")
bar, per = 0, collections.Counter()
for i, l in enumerate(body):
    if "s_barrier" in l:
        bar += 1
    if "scratch_" in l:
        per[(bar, l.split()[0])] += 1
for (b, op), c in sorted(per.items()):
    print(f"after barrier {b:3d}: {op:22s} x {c}")
print("barriers", bar)
c = collections.Counter(l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";")))
print(c.most_common(30))
