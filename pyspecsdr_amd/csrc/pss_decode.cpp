// pss_decode.cpp — the per-message halves of the reference's decoders (xqtr/PySpecSDR decoders.py), host side, plain C++ behind the
// C ABI of include/pss.h.  The sample-rate halves run on the GPU (pss_morse_edges, pss_afsk_bits, pss_row_normalise); what is left
// works on a few dozen to a few thousand integers per message — branchy, sequential, tiny — and runs where its caller is.
//
// Written against the DATA CONTRACT of those functions (tests/golden/decoders.npz: m_text_*, m_timing_*, a_packets_*, ax_out; fuzzed
// against the reference in the build container, tools/fuzz_decoders_vs_reference.py):
//
//   Morse (decode_morse, decoders.py:167-231)   rise / fall sample indices of the keyed envelope -> pulse lengths and gaps in seconds ->
//       two pulse classes (dot / dash) -> symbols, letter gaps (> 3 dots), word gaps (> 7 dots) -> text through the ITU table.
//       The reference finds the two classes with scipy.cluster.vq.kmeans(durations, 2): Lloyd's iteration from RANDOM starting points
//       (NumPy's global generator), best of 20 starts by mean |x - centroid|.  Its answer is therefore a partition that Lloyd's
//       iteration leaves unchanged, and for keyed signals there is exactly one such partition (all five keyed goldens: every seed gives
//       the same centroids); here ALL stable two-class partitions of the sorted pulse lengths are enumerated and the one with the smallest
//       mean distance is taken, the class means summed in observation order as scipy's update step does — the same doubles.  Where the
//       reference itself depends on its random draw (pure noise: sub-millisecond glitches, for which kmeans' 1e-5 s stopping threshold
//       ends the iteration before it has converged) there is nothing to be identical to; the result here is deterministic.
//   AX.25 (decode_ax25_frame + decode_aprs_payload, decoders.py:6-88)   first flag 01111110, bits up to the next flag with the
//       zero after five ones dropped, bytes LSB first, addresses as 7-bit characters shifted down by one, "SRC>DEST:info".
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "pss_ctx.h"

namespace {

struct MorseEntry { const char *sym; const char *txt; };
// International Morse code (ITU-R M.1677) as far as the application's table goes, plus the SOS prosign keyed as one symbol
const MorseEntry MORSE[] = {
    {".-", "A"}, {"-...", "B"}, {"-.-.", "C"}, {"-..", "D"}, {".", "E"}, {"..-.", "F"}, {"--.", "G"}, {"....", "H"}, {"..", "I"},
    {".---", "J"}, {"-.-", "K"}, {".-..", "L"}, {"--", "M"}, {"-.", "N"}, {"---", "O"}, {".--.", "P"}, {"--.-", "Q"}, {".-.", "R"},
    {"...", "S"}, {"-", "T"}, {"..-", "U"}, {"...-", "V"}, {".--", "W"}, {"-..-", "X"}, {"-.--", "Y"}, {"--..", "Z"},
    {".----", "1"}, {"..---", "2"}, {"...--", "3"}, {"....-", "4"}, {".....", "5"}, {"-....", "6"}, {"--...", "7"}, {"---..", "8"},
    {"----.", "9"}, {"-----", "0"}, {"--..--", ","}, {".-.-.-", "."}, {"..--..", "?"}, {"-..-.", "/"}, {"-....-", "-"}, {"-.--.", "("},
    {"-.--.-", ")"}, {".-...", "&"}, {"---...", ":"}, {"-.-.-.", ";"}, {"-...-", "="}, {".-.-.", "+"}, {".-..-.", "\""}, {"...-..-", "$"},
    {".--.-.", "@"}, {"..--.-", "_"}, {"...---...", "SOS"}};

const char *morse_lookup(const std::string &s)
{
    for (const auto &e : MORSE)
        if (s == e.sym) return e.txt;
    return nullptr;
}

// np.add.reduce over contiguous float64: 8192-element chunks added in order, inside a chunk NumPy's pairwise tree (8 accumulators, blocks
// of 128, halves rounded down to multiples of 8) — what np.mean(gaps) sums with
double pairwise_chunk(const double *a, long n)
{
    if (n < 8) {
        double r = 0.0;
        for (long i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_chunk(a, n2) + pairwise_chunk(a + n2, n - n2);
}
double np_sum(const double *a, long n)
{
    const long B = 8192;
    if (n <= B) return pairwise_chunk(a, n);
    double acc = pairwise_chunk(a, B);
    for (long st = B; st < n; st += B) acc += pairwise_chunk(a + st, (n - st) < B ? (n - st) : B);
    return acc;
}

// The two pulse classes: every split of the sorted lengths that one Lloyd step maps to itself, best mean |x - c| first.
// -> number of classes found (1: all pulses equal), centroids in c[0] <= c[1].
int two_classes(const std::vector<double> &d, double c[2])
{
    const long n = (long)d.size();
    std::vector<long> order(n);
    for (long i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](long a, long b) { return d[a] < d[b]; });
    // rank of every observation in the sorted order: observation i belongs to the upper class of split k iff rank[i] >= k
    std::vector<long> rank(n);
    for (long k = 0; k < n; k++) rank[order[k]] = k;
    bool found = false;
    double best = 0.0;
    for (long k = 1; k < n; k++) {
        if (d[order[k]] == d[order[k - 1]]) continue;       // equal lengths cannot be told apart by a distance
        double s0 = 0.0, s1 = 0.0;                          // class sums in observation order (scipy's update_cluster_means)
        for (long i = 0; i < n; i++) {
            if (rank[i] >= k) s1 += d[i];
            else s0 += d[i];
        }
        const double c0 = s0 / (double)k, c1 = s1 / (double)(n - k);
        // stable: the largest member of the lower class is no farther from c0 than from c1 (a tie goes to the first centroid in scipy's
        // argmin), the smallest member of the upper class strictly nearer to c1
        const double lo_max = d[order[k - 1]], hi_min = d[order[k]];
        if (!(std::fabs(lo_max - c0) <= std::fabs(lo_max - c1)) || !(std::fabs(hi_min - c1) < std::fabs(hi_min - c0))) continue;
        double dist = 0.0;
        for (long i = 0; i < n; i++) dist += std::fabs(d[i] - (rank[i] >= k ? c1 : c0));
        dist /= (double)n;
        if (!found || dist < best) { found = true; best = dist; c[0] = c0; c[1] = c1; }
    }
    if (found) return 2;
    // no split: every pulse has the same length (or a single class is the only fixed point) — one centroid, the mean
    double s = 0.0;
    for (long i = 0; i < n; i++) s += d[i];
    c[0] = c[1] = s / (double)n;
    return 1;
}

// the characters str.strip() removes, as far as 7-bit characters go
bool py_space7(unsigned ch) { return ch == 0x20 || (ch >= 0x09 && ch <= 0x0d) || (ch >= 0x1c && ch <= 0x1f); }

}  // namespace

extern "C" int pss_h_morse_decode(const int32_t *rise, long n_rise, const int32_t *fall, long n_fall, double fs, char *text, long text_cap,
                                  double *timing3)
{
    if (n_rise < 0 || n_fall < 0 || (n_rise > 0 && !rise) || (n_fall > 0 && !fall) || !text || text_cap < 1 || !timing3 || !(fs > 0.0)) return PSS_E_ARG;
    text[0] = 0;
    timing3[0] = timing3[1] = timing3[2] = 0.0;
    if (n_rise == 0 || n_fall == 0) return 0;
    // a fall before the first rise belongs to a pulse that began before the buffer; a last rise without a fall to one that ends after it
    if (fall[0] < rise[0]) { fall++; n_fall--; }
    if (n_rise > n_fall) n_rise--;
    if (n_rise != n_fall) return PSS_E_ARG;      // (edges that do not alternate: the reference's array subtraction raises)
    const long np_ = n_rise;
    if (np_ == 0) return 0;
    std::vector<double> dur(np_), gap(np_ > 1 ? np_ - 1 : 0);
    for (long i = 0; i < np_; i++) dur[i] = (double)((long long)fall[i] - (long long)rise[i]) / fs;
    for (long i = 0; i + 1 < np_; i++) gap[i] = (double)((long long)rise[i + 1] - (long long)fall[i]) / fs;
    double dot, dash;
    if (np_ > 1) {
        double c[2];
        two_classes(dur, c);
        dot = c[0];
        dash = c[1];
    } else {
        dot = dur[0];
        dash = dot * 3;
    }
    std::string out, letter;
    auto flush = [&]() {
        const char *t = morse_lookup(letter);
        out += t ? t : "?";
        letter.clear();
    };
    const double mid = (dot + dash) / 2;
    for (long i = 0; i < np_; i++) {
        letter += dur[i] < mid ? '.' : '-';
        if (i < (long)gap.size() && gap[i] > dot * 3) {
            flush();
            if (gap[i] > dot * 7) out += ' ';
        }
    }
    if (!letter.empty()) flush();
    timing3[0] = dot;
    timing3[1] = dash;
    timing3[2] = gap.empty() ? 0.0 : np_sum(gap.data(), (long)gap.size()) / (double)gap.size();
    if ((long)out.size() + 1 > text_cap) return PSS_E_ARG;
    memcpy(text, out.c_str(), out.size() + 1);
    return (int)out.size();
}

// -> 1 and the packet "SRC>DEST:info" in out (bytes = the characters' code points 0..255, *out_len of them, NOT NUL-terminated: info may
// hold NULs), 0 if the stream holds no decodable frame (the reference returns None), negative on bad arguments / a buffer too small.
extern "C" int pss_h_ax25_frame(const uint8_t *bits, long n_bits, char *out, long out_cap, long *out_len)
{
    if (n_bits < 0 || (n_bits > 0 && !bits) || !out || !out_len) return PSS_E_ARG;
    *out_len = 0;
    auto is_flag = [&](const uint8_t *p) {
        return p[0] == 0 && p[1] == 1 && p[2] == 1 && p[3] == 1 && p[4] == 1 && p[5] == 1 && p[6] == 1 && p[7] == 0;
    };
    long start = -1;
    for (long i = 0; i + 7 < n_bits; i++)
        if (is_flag(bits + i)) { start = i + 8; break; }
    if (start < 0) return 0;
    std::vector<uint8_t> fb;
    int ones = 0;
    long i = start;
    while (i < n_bits - 7) {
        const uint8_t b = bits[i] ? 1 : 0;
        fb.push_back(b);
        ones = b ? ones + 1 : 0;
        if (ones == 5 && i + 1 < n_bits && bits[i + 1] == 0) {   // the transmitter's stuffed zero: dropped (and no flag test on this turn)
            i += 2;
            ones = 0;
            continue;
        }
        i += 1;
        if (fb.size() >= 8 && is_flag(fb.data() + fb.size() - 8)) {
            fb.resize(fb.size() - 8);
            break;
        }
    }
    std::vector<unsigned> by;
    for (size_t k = 0; k + 8 <= fb.size(); k += 8) {
        unsigned v = 0;
        for (int j = 0; j < 8; j++) v |= (unsigned)fb[k + j] << j;
        by.push_back(v);
    }
    if (by.size() < 14) return 0;
    auto addr = [&](size_t a, size_t b) {
        std::string s;
        for (size_t k = a; k < b; k++) s += (char)((by[k] >> 1) & 0x7F);
        size_t lo = 0, hi = s.size();
        while (lo < hi && py_space7((unsigned char)s[lo])) lo++;
        while (hi > lo && py_space7((unsigned char)s[hi - 1])) hi--;
        return s.substr(lo, hi - lo);
    };
    std::string pk = addr(7, 13) + ">" + addr(0, 6) + ":";
    for (size_t k = 15; k < by.size(); k++) pk += (char)by[k];
    if ((long)pk.size() > out_cap) return PSS_E_ARG;
    memcpy(out, pk.data(), pk.size());
    *out_len = (long)pk.size();
    return 1;
}

// decode_morse (decoders.py:136-231) on one host buffer: envelope / threshold / edges on the GPU (pss_h_morse_edges), the timing above.
extern "C" int pss_h_decode_morse(pss_ctx *ctx, const float *h_iq, int n, double fs, double threshold_db, char *text, long text_cap,
                                  double *timing3)
{
    if (!ctx) return PSS_E_ARG;
    if (!h_iq || n < 1 || !text || text_cap < 1 || !timing3) return pss_fail(ctx, PSS_E_ARG, "pss_h_decode_morse: bad argument");
    const int cap = n / 2 + 1;
    std::vector<int32_t> rise(cap), fall(cap);
    int nr = 0, nf = 0;
    int r = pss_h_morse_edges(ctx, h_iq, n, threshold_db, cap, rise.data(), fall.data(), &nr, &nf);
    if (r) return r;
    r = pss_h_morse_decode(rise.data(), nr, fall.data(), nf, fs, text, text_cap, timing3);
    if (r < 0) return pss_fail(ctx, r, "pss_h_decode_morse: text buffer too small or edges that do not alternate");
    return r;
}

// decode_aprs (decoders.py:115-133) on one host buffer of REAL audio (np.real of a complex buffer is the caller's): normalisation and the
// AFSK bit slicer on the GPU (pss_h_afsk_bits), the AX.25 framing above.  -> 1 + packet, 0 = no packet (the reference returns []).
extern "C" int pss_h_decode_aprs(pss_ctx *ctx, const double *h_audio, int n, double fs, const double *sos1200, const double *sos2200, int nsec,
                                 char *out, long out_cap, long *out_len)
{
    if (!ctx) return PSS_E_ARG;
    if (!h_audio || n < 1 || !out || !out_len) return pss_fail(ctx, PSS_E_ARG, "pss_h_decode_aprs: bad argument");
    *out_len = 0;
    const int nb = pss_afsk_n_bits(n, fs);
    if (nb <= 0) return 0;
    std::vector<uint8_t> bits(nb);
    int r = pss_h_afsk_bits(ctx, h_audio, n, fs, 1, sos1200, sos2200, nsec, bits.data());
    if (r) return r;
    r = pss_h_ax25_frame(bits.data(), nb, out, out_cap, out_len);
    if (r < 0) return pss_fail(ctx, r, "pss_h_decode_aprs: packet buffer too small");
    return r;
}
