#!/usr/bin/env python3
"""Table of pss_fft_r16.h db_of: for the 97 centres c_i = 0.75 + i / 128 the double nearest 1 / c_i and -log10 of THAT double
(evaluated in x87 extended precision, then rounded), so that log10 z = log10(z * invc) + logc holds without the table's rounding.
    python tools/make_db_table.py > table.txt"""
import numpy as np

ld = np.longdouble
for i in range(97):
    c = 0.75 + i / 128.0
    invc = np.float64(1.0) if i == 32 else np.float64(1.0) / np.float64(c)
    logc = ld(0) if i == 32 else -np.log10(ld(invc))
    print("    {%s, %s}," % (float(invc).hex(), float(np.float64(logc)).hex()))
print("LOG10_2", float(np.float64(np.log10(ld(2)))).hex(), "1/LN10", float(np.float64(ld(1) / np.log(ld(10)))).hex())
