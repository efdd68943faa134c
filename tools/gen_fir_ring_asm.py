#!/usr/bin/env python3
"""Generator of pyspecsdr_amd/csrc/pss_fir_ring_asm.h: the 65-tap FIR of the fused NFM forward kernel (k_nfm_fwd, FIR workers) for
FOUR consecutive outputs of one frame per lane, as one hand-scheduled gfx950 instruction stream.

    python tools/gen_fir_ring_asm.py            (rewrites the header; the header is committed, this script documents how it was made)

What the stream computes is fir_batch / fir_ring of pss_nfm_fused.h bit for bit — OpenBLAS's ddot tree (scipy lfilter -> np.convolve ->
cblas_ddot, signal_processing.py:108): four phases of 16 taps, per phase and output four accumulator lanes l with
    lo = fma(xb[o+l], yb[l], fma(xa[o+l], ya[l], +0)),  hi = fma(xb[o+l+4], yb[l+4], fma(xa[o+l+4], ya[l+4], +0)),  s[o][l] (+)= lo + hi
then dot = (s0 + s2) + (s1 + s3) and out = fma(tap64, x[o+64], dot).  Why by hand: the compiler's version waits for every phase's
scalar tap loads and LDS window reads right behind issuing them; here phase k + 1's operands are requested BEFORE phase k is computed
(two SGPR tap sets at a distance of one phase; the window floats at half a phase, into the buffers the phase has already converted), every window float is converted to float64 once
(the three columns two consecutive phases share are carried over converted), and the register budget is fixed: 94 VGPRs, 64 SGPRs.

Register map inside the statement (all clobbers; operands are the compiler's):
    v[16:47]    s[o][l]                      16 accumulators
    v[48:59]    raw window floats, run a     3 buffers of 4 (ds_read_b128): a phase's three groups; the two that are converted by the
    v[60:71]    raw window floats, run b     middle of the phase receive the next phase's two new groups
    v[72:91]    converted columns            10 doubles: the live stretch of 5 columns per run
    v[92:107]   chain temporaries            8 doubles (4 chains x lo / hi)
    v[108:109]  LDS addresses
    s[36:67]    taps of the even phases (ya[0..7] = s[36:51], yb[0..7] = s[52:67]);  s[68:99]  taps of the odd phases
    vcc         ring-position arithmetic
"""
import os

WCOLS = 96
ACC0, RAWA0, RAWB0, CONV0, TMP0, ADDR0 = 16, 48, 60, 72, 92, 108
LAST_V = 109
TAPS = {0: 36, 1: 68}          # SGPR base of tap set (phase parity)


def acc(o, l):
    r = ACC0 + 2 * (4 * o + l)
    return f"v[{r}:{r + 1}]"


def pair(r):
    return f"v[{r}:{r + 1}]"


def tap(k, which, i):
    """SGPR pair of ya[i] (which = 0) / yb[i] (which = 1) of phase k."""
    r = TAPS[k & 1] + 16 * which + 2 * i
    return f"s[{r}:{r + 1}]"


class Gen:
    def __init__(self):
        self.lines = []
        # raw buffers per run: group index (in units of 4 columns from the CURRENT phase's column 0) -> buffer number
        self.free_buf = {"a": [0, 1, 2], "b": [0, 1, 2]}
        self.grp = {"a": {}, "b": {}}
        self.conv = {"a": {}, "b": {}}       # column (of the current phase) -> converted VGPR pair base
        self.free_conv = list(range(CONV0, CONV0 + 20, 2))

    def emit(self, s):
        self.lines.append(s)

    # ---- window reads --------------------------------------------------------------------------------------------------
    def raw_reg(self, run, g, comp):
        base = (RAWA0 if run == "a" else RAWB0) + 4 * self.grp[run][g]
        return base + comp

    def fetch_group(self, run, g, colconst, addr):
        """ds_read_b128 of the group whose first column is ring position wrap(q + colconst) into a free buffer; group key g."""
        b = self.free_buf[run].pop(0)
        self.grp[run][g] = b
        base = (RAWA0 if run == "a" else RAWB0) + 4 * b
        # vcc_lo = q + colconst; wrapped at WCOLS (q < WCOLS and colconst < WCOLS: one subtraction)
        self.emit(f"s_add_i32 vcc_lo, %2, {colconst}")
        self.emit(f"s_sub_i32 vcc_hi, vcc_lo, {WCOLS}")
        self.emit(f"s_cmp_ge_i32 vcc_lo, {WCOLS}")
        self.emit("s_cselect_b32 vcc_lo, vcc_hi, vcc_lo")
        self.emit(f"v_lshl_add_u32 v{addr}, vcc_lo, 2, %0")
        self.emit(f"ds_read_b128 v[{base}:{base + 3}], v{addr}")

    def fetch_taps(self, k):
        r = TAPS[k & 1]
        self.emit(f"s_load_dwordx16 s[{r}:{r + 15}], %3, {8 * 8 * k}")              # ya[0..7] = rev[8k ..]
        self.emit(f"s_load_dwordx16 s[{r + 16}:{r + 31}], %3, {8 * (32 + 8 * k)}")  # yb[0..7] = rev[32 + 8k ..]

    # ---- conversions ---------------------------------------------------------------------------------------------------
    def need(self, run, col):
        if col in self.conv[run]:
            return self.conv[run][col]
        p = self.free_conv.pop(0)
        self.conv[run][col] = p
        self.emit(f"v_cvt_f64_f32_e32 {pair(p)}, v{self.raw_reg(run, col // 4, col % 4)}")
        return p

    def drop(self, run, col):
        p = self.conv[run].pop(col)
        self.free_conv.append(p)

    # ---- one phase -----------------------------------------------------------------------------------------------------
    def compute(self, k, mid=None):
        """The phase's 16 chains, ordered by window column c = o + l.  `mid`: called once the groups 0 and 1 of both runs are fully
        converted (behind the conversions of c = 3): their buffers are free, the next phase's window floats are requested into them."""
        first = k == 0
        for c in range(7):
            chains = [(o, c - o) for o in range(4) if 0 <= c - o < 4]
            xa0, xb0 = self.need("a", c), self.need("b", c)
            xa4, xb4 = self.need("a", c + 4), self.need("b", c + 4)
            if c == 3:
                for run in ("a", "b"):
                    for g in (0, 1):
                        self.free_buf[run].append(self.grp[run].pop(g))
                if mid:
                    mid()
            t = {ch: (TMP0 + 4 * i, TMP0 + 4 * i + 2) for i, ch in enumerate(chains)}
            for (o, l) in chains:
                self.emit(f"v_fma_f64 {pair(t[(o, l)][0])}, {pair(xa0)}, {tap(k, 0, l)}, 0")
            for (o, l) in chains:
                self.emit(f"v_fma_f64 {pair(t[(o, l)][1])}, {pair(xa4)}, {tap(k, 0, l + 4)}, 0")
            for (o, l) in chains:
                self.emit(f"v_fmac_f64_e32 {pair(t[(o, l)][0])}, {tap(k, 1, l)}, {pair(xb0)}")
            for (o, l) in chains:
                self.emit(f"v_fmac_f64_e32 {pair(t[(o, l)][1])}, {tap(k, 1, l + 4)}, {pair(xb4)}")
            for (o, l) in chains:
                dst = acc(o, l) if first else pair(t[(o, l)][0])
                self.emit(f"v_add_f64 {dst}, {pair(t[(o, l)][0])}, {pair(t[(o, l)][1])}")
            if not first:
                for (o, l) in chains:
                    self.emit(f"v_add_f64 {acc(o, l)}, {acc(o, l)}, {pair(t[(o, l)][0])}")
            self.drop("a", c)
            self.drop("b", c)
        self.drop("a", 7)
        self.drop("b", 7)
        # columns 8, 9, 10 stay converted: the next phase's columns 0, 1, 2

    def shift_phase(self):
        """Rename for the next phase: column c -> c - 8, group g -> g - 2."""
        for run in ("a", "b"):
            self.conv[run] = {c - 8: p for c, p in self.conv[run].items()}
            self.grp[run] = {g - 2: b for g, b in self.grp[run].items()}

    def generate(self):
        e = self.emit
        A0, A1 = ADDR0, ADDR0 + 1
        # phase 0's operands
        self.fetch_taps(0)
        for g in range(3):
            self.fetch_group("a", g, 4 * g, A0 if g % 2 == 0 else A1)
            self.fetch_group("b", g, 32 + 4 * g, A1 if g % 2 == 0 else A0)
        for k in range(4):
            e("s_waitcnt lgkmcnt(0)")
            # requests of the NEXT phase run beside this phase's arithmetic: the taps (scalar loads, longest latency) at once, the window
            # floats (LDS) from the middle of the phase on, into the buffers of the groups that are converted by then
            if k < 3:
                self.fetch_taps(k + 1)

                def mid(k=k):
                    for g in (3, 4):        # groups 1, 2 of phase k + 1 = groups 3, 4 counted from this phase's column 0
                        self.fetch_group("a", g, 8 * k + 4 * g, A0)
                        self.fetch_group("b", g, 8 * k + 32 + 4 * g, A1)
            else:
                # tail: tap 64 (into the free even set) and the window columns o + 64, o = 0..3 (run a's buffer)
                e(f"s_load_dwordx2 s[{TAPS[0]}:{TAPS[0] + 1}], %3, {8 * 64}")

                def mid():
                    self.fetch_group("a", 10, 64, A0)     # key 10: never collides with the phase's own 0..4
            self.compute(k, mid)
            if k < 3:
                self.shift_phase()
        e("s_waitcnt lgkmcnt(0)")
        tail_b = self.grp["a"][10]
        for o in range(4):
            xt = TMP0 + 8 + 2 * o
            e(f"v_cvt_f64_f32_e32 {pair(xt)}, v{RAWA0 + 4 * tail_b + o}")
        for o in range(4):
            e(f"v_add_f64 {pair(TMP0)}, {acc(o, 0)}, {acc(o, 2)}")
            e(f"v_add_f64 {pair(TMP0 + 2)}, {acc(o, 1)}, {acc(o, 3)}")
            e(f"v_add_f64 {pair(TMP0)}, {pair(TMP0)}, {pair(TMP0 + 2)}")
            e(f"v_fma_f64 {pair(TMP0 + 4)}, s[{TAPS[0]}:{TAPS[0] + 1}], {pair(TMP0 + 8 + 2 * o)}, {pair(TMP0)}")
            e(f"ds_write_b64 %1, {pair(TMP0 + 4)} offset:{512 * o}")
        return self.lines


def main():
    lines = Gen().generate()
    n = {"valu": sum(l.startswith("v_") for l in lines), "f64": sum(l.startswith(("v_fma", "v_fmac", "v_add_f64", "v_cvt")) for l in lines),
         "lds": sum(l.startswith("ds_") for l in lines), "smem": sum(l.startswith("s_load") for l in lines)}
    clob = [f"v{i}" for i in range(16, LAST_V + 1)] + [f"s{i}" for i in range(36, 100)] + ["vcc", "scc", "memory"]
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyspecsdr_amd", "csrc", "pss_fir_ring_asm.h")
    with open(out, "w") as f:
        f.write("// pss_fir_ring_asm.h — GENERATED by tools/gen_fir_ring_asm.py (edit the generator, not this file).\n")
        f.write("// Four consecutive outputs of the 65-tap FIR of k_nfm_fwd's workers as one hand-scheduled gfx950 instruction stream:\n")
        f.write("// the ddot tree of fir_batch bit for bit, phase k + 1's taps (scalar loads) and window floats (ds_read_b128) requested before\n")
        f.write("// phase k is computed, every window float converted once.  Register map and rationale: the generator's docstring.\n")
        f.write(f"// {len(lines)} instructions: {n['f64']} float64-class VALU, {n['valu'] - n['f64']} other VALU, {n['lds']} LDS, {n['smem']} scalar loads.\n")
        f.write("#pragma once\n\nnamespace fused {\n\n")
        f.write("// row_addr: LDS byte address of this lane's window row; u_addr: LDS byte address of this lane's slot of the thread's first\n")
        f.write("// output in the u chunk buffer (the four outputs go to u_addr + 512 o); q: ring position (columns, a multiple of 4, < 96) of\n")
        f.write("// the window column that meets tap 64 of the first output; rev: the reversed taps in device memory (rev[j] = taps[64 - j], 65 doubles).\n")
        f.write("// The outputs are LDS writes still in flight when the statement ends (lgkmcnt): the kernel's next lds_barrier() waits for them.\n")
        f.write("__device__ __forceinline__ void fir_ring_asm(unsigned row_addr, unsigned u_addr, int q, const double *rev)\n{\n")
        f.write("    asm volatile(\n")
        for l in lines:
            f.write(f'        "{l}\\n\\t"\n')
        f.write('        :\n        : "v"(row_addr), "v"(u_addr), "s"(q), "s"(rev)\n        : ')
        f.write(", ".join(f'"{c}"' for c in clob))
        f.write(");\n}\n\n}  // namespace fused\n")
    print(out, n, len(lines), "instructions")


if __name__ == "__main__":
    main()
