/*
 * derive_rcp14.c — characterise the AVX-512 VRCP14 instruction on the build host and emit the compact
 * table used by the SVML-atan2f restatement (oracle/pss_oracle.c and pyspecsdr_amd/csrc/pss_device.h).
 *
 *   gcc -O1 -mavx512f tools/derive_rcp14.c -o /tmp/derive_rcp14 && /tmp/derive_rcp14
 *
 * Findings (Xeon "Sapphire Rapids", exhaustive over all 2^23 mantissas, exponents -100..100 sampled):
 *   - the result scales exactly with the exponent and is 2^-e for a zero mantissa;
 *   - otherwise it depends only on the top 16 mantissa bits i = m >> 7, has a 16-bit fraction, and equals
 *       v17 = (A[i >> 10] - B[i >> 10] * (i & 1023)) >> 9,  result = v17 * 2^-17 * 2^-e
 *     for the 64-entry tables printed below (A is unique, B is the first slope that reproduces the
 *     whole segment).  The program re-verifies the model against the instruction for every mantissa.
 */
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float rcp14_hw(float x) { __m128 v = _mm_set_ss(x); v = _mm_rcp14_ss(v, v); return _mm_cvtss_f32(v); }

int main(void)
{
    static uint32_t v17[65536];
    for (uint32_t i = 0; i < 65536; i++) {
        uint32_t xb = 0x3f800000u | (i << 7) | 1u, rb;
        float x, r;
        memcpy(&x, &xb, 4);
        r = rcp14_hw(x);
        memcpy(&rb, &r, 4);
        v17[i] = 65536u + ((rb & 0x7fffffu) >> 7);
        if ((rb >> 23) == 127) v17[i] = 131072u; /* cannot happen for m != 0 */
    }
    uint32_t A[64], B[64];
    for (int s = 0; s < 64; s++) {
        int found = 0;
        for (uint32_t b = 200; b < 1100 && !found; b++) {
            int64_t lo = -1, hi = (int64_t)1 << 40;
            for (int l = 0; l < 1024; l++) {
                int64_t v = v17[s * 1024 + l];
                int64_t l0 = v * 512 + (int64_t)b * l, h0 = (v + 1) * 512 + (int64_t)b * l;
                if (l0 > lo) lo = l0;
                if (h0 < hi) hi = h0;
            }
            if (lo < hi) { A[s] = (uint32_t)lo; B[s] = b; found = 1; }
        }
        if (!found) { printf("segment %d: no linear model\n", s); return 1; }
    }
    long bad = 0;
    for (uint32_t m = 0; m < (1u << 23); m++) {
        uint32_t xb = 0x3f800000u | m, rb, model;
        float x, r;
        memcpy(&x, &xb, 4);
        r = rcp14_hw(x);
        memcpy(&rb, &r, 4);
        if (m == 0) model = 0x3f800000u;
        else {
            uint32_t i = m >> 7, v = (A[i >> 10] - B[i >> 10] * (i & 1023u)) >> 9;
            model = (126u << 23) | ((v & 0xffffu) << 7);
        }
        bad += (model != rb);
    }
    printf("model mismatches over 2^23 mantissas: %ld\n", bad);
    printf("static const uint32_t RCP14_A[64] = {");
    for (int s = 0; s < 64; s++) printf("%s%uu", s ? ", " : "", A[s]);
    printf("};\nstatic const uint16_t RCP14_B[64] = {");
    for (int s = 0; s < 64; s++) printf("%s%u", s ? ", " : "", B[s]);
    printf("};\n");
    return bad != 0;
}
