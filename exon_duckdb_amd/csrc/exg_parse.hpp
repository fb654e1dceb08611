// exg_parse.hpp — number parsing shared by the VCF tokeniser (POS, QUAL) and the typed INFO / FORMAT
// columns.  `Src` is any byte source with `uint32_t b(int i) const`.
#pragma once
#include "exg_common.hpp"
#include "exg_float_el.hpp"

namespace exg {

// ---- exact decimal -> float32 ---------------------------------------------------------------------
// f32::from_str is correctly rounded for every literal.  Here: literals of at most 7 significant digits without an
// exponent take Clinger's fast path in f32 (one correctly rounded division); everything else goes through the
// Eisel-Lemire algorithm on the first 19 significant digits (exg_float_el.hpp: exact for any 64-bit significand and any
// exponent, subnormals and overflow to infinity included).  A literal with MORE than 19 significant digits is decided by
// converting both ends of the interval its first 19 digits pin down — [w, w + 1] x 10^q; they round to the same float
// unless a rounding boundary falls strictly inside an interval of relative width 10^-19 (about one literal in 10^11;
// a boundary AT one of the two ends — "halfway, then zeros, then a 1" — is decided by the side the literal lies on): only then
// the kernel returns status 2 and its caller hands the literal to the exact big-integer parser (exg_float_slow.hpp), which
// runs in one-block fix-up code behind the scan (k_vcf_finalize) / the typed-column kernels (k_f32_slow).

// status: 0 ok, 1 syntax error, 2 more than 19 significant digits AND a rounding boundary within 10^-19 of them
template <class Src>
__device__ int parse_f32(const Src &src, int s, int e, float *out) {
    int i = s;
    if (i >= e) return 1;
    bool neg = false;
    uint32_t c = src.b(i);
    if (c == '+' || c == '-') {
        neg = c == '-';
        i++;
    }
    int n = e - i;
    if (n == 3 || n == 8) {
        uint32_t l0 = src.b(i) | 0x20, l1 = src.b(i + 1) | 0x20, l2 = src.b(i + 2) | 0x20;
        if (n == 3 && l0 == 'n' && l1 == 'a' && l2 == 'n') {
            *out = __uint_as_float(0x7FC00000u);
            return 0;
        }
        bool inf = l0 == 'i' && l1 == 'n' && l2 == 'f';
        if (inf && n == 8) {
            const char *rest = "inity";
            for (int k = 0; k < 5; k++) inf = inf && (src.b(i + 3 + k) | 0x20) == (uint32_t)rest[k];
        }
        if (inf) {
            *out = __uint_as_float(neg ? 0xFF800000u : 0x7F800000u);
            return 0;
        }
    }
    // The usual literal — digits with at most one '.', no exponent, at most 7 significant digits ("50.0", "0.0312") —
    // in 32-bit arithmetic: the mantissa (< 2^24) and 10^k (k <= 10) are both exact floats, so ONE correctly rounded
    // float division is the correctly rounded result (Clinger's fast path; HIP compiles `/` on floats correctly
    // rounded).  Anything else (exponent, more digits, a syntax error) falls through to the general code below.
    if (n <= 12) {
        // (the bytes come out of ONE 12-byte read — src.u96: an unaligned LDS read in the fused kernels — and the loop runs to
        // the longest literal among the wave's active lanes, with every lane predicated: a loop over byte loads is a chain
        // of LDS round trips under a divergent exit, 0.8 ms of the 4.8 ms of a 5 GB scan)
        uint32_t w0, w1, w2;
        src.u96(i, &w0, &w1, &w2);
        uint32_t m32 = 0;
        int sig32 = 0, frac = 0, nd32 = 0;
        bool dot = false, plain = true;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            if (!__ballot(j < n)) break;
            const uint32_t w = j < 4 ? w0 : j < 8 ? w1 : w2;
            const uint32_t ch = (w >> (8 * (j & 3))) & 0xFFu;
            const bool on = j < n && plain;
            const bool is_dot = ch == '.';
            const uint32_t d = ch - '0';
            const bool is_digit = d <= 9u;
            plain = plain && (j >= n || is_digit || (is_dot && !dot));
            const bool take = on && is_digit;
            nd32 += take ? 1 : 0;
            const bool signif = take && (m32 != 0 || d != 0);
            m32 = signif ? m32 * 10u + d : m32;
            sig32 += signif ? 1 : 0;
            frac += (take && dot) ? 1 : 0;
            dot = dot || (on && is_dot);
        }
        if (plain && nd32 > 0 && sig32 <= 7 && frac <= 10) {
            static constexpr float kPow10f[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
            float p = 1.0f;
#pragma unroll
            for (int k = 1; k <= 10; k++) p = frac == k ? kPow10f[k] : p;  // a select chain, not a memory table
            const float f = m32 ? (float)m32 / p : 0.0f;
            *out = neg ? -f : f;
            return 0;
        }
    }
    unsigned long long m = 0;
    int nd = 0, sig = 0, e10 = 0;
    bool inexact = false, seen_dot = false;
    for (; i < e; i++) {
        c = src.b(i);
        if (c == '.') {
            if (seen_dot) return 1;
            seen_dot = true;
            continue;
        }
        if (c < '0' || c > '9') break;
        nd++;
        if (sig < 19) {
            if (m || c != '0') {
                m = m * 10 + (c - '0');
                sig++;
            }
            if (seen_dot) e10--;
        } else {
            if (c != '0') inexact = true;
            if (!seen_dot) e10++;
        }
    }
    if (nd == 0) return 1;
    if (i < e) {
        c = src.b(i);
        if (c != 'e' && c != 'E') return 1;
        i++;
        bool eneg = false;
        if (i < e && (src.b(i) == '+' || src.b(i) == '-')) eneg = src.b(i++) == '-';
        if (i >= e) return 1;
        int ev = 0;
        for (; i < e; i++) {
            c = src.b(i);
            if (c < '0' || c > '9') return 1;
            if (ev < 100000) ev = ev * 10 + (int)(c - '0');
        }
        e10 += eneg ? -ev : ev;
    }
    if (m == 0) {
        *out = neg ? -0.0f : 0.0f;
        return 0;
    }
    bool tie_lo = false, tie_hi = false;
    uint32_t bits = el_f32_bits(m, e10, &tie_lo);
    if (inexact) {
        // the literal lies strictly inside (m, m + 1) x 10^e10
        const uint32_t up = el_f32_bits(m + 1, e10, &tie_hi);
        if (up != bits) {
            if (tie_lo) bits = up;       // just above an exact halfway point ("16777217.000...01"): rounds up
            else if (!tie_hi) return 2;  // (just below one: rounds down = bits) else the boundary is inside the interval
        }
    }
    *out = __uint_as_float(bits | (neg ? 0x80000000u : 0u));
    return 0;
}

// usize::from_str: optional '+', one or more digits, no overflow (63 bits kept)
template <class Src>
__device__ bool parse_pos(const Src &src, int s, int e, long long *out) {
    int i = s;
    if (e - s <= 10 && e > s) {
        // the usual field — at most nine digits behind an optional '+' — out of ONE 12-byte read (src.u96: an unaligned LDS
        // read in the fused kernels), then nine predicated multiply-adds in registers: no loop, no byte loads (a loop over
        // the bytes is a chain of LDS round trips: 0.5 ms of the 4.8 ms of a 5 GB scan)
        uint32_t w0, w1, w2;
        src.u96(s, &w0, &w1, &w2);
        int n = e - s;
        if ((w0 & 0xFFu) == '+') {
            w0 = __builtin_amdgcn_alignbyte(w1, w0, 1);
            w1 = __builtin_amdgcn_alignbyte(w2, w1, 1);
            w2 >>= 8;
            n--;
        }
        if (n >= 1 && n <= 9) {
            uint32_t v32 = 0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const uint32_t w = j < 4 ? w0 : j < 8 ? w1 : w2;
                const uint32_t d = ((w >> (8 * (j & 3))) & 0xFFu) - '0';
                const bool on = j < n;
                bad = bad || (on && d > 9u);
                v32 = on ? v32 * 10u + d : v32;
            }
            if (bad) return false;
            *out = (long long)v32;
            return true;
        }
        if (n <= 0) return false;
    }
    if (i < e && src.b(i) == '+') i++;
    if (i >= e) return false;
    if (e - i <= 9) {  // cannot overflow 32 bits: one multiply-add per digit (the 64-bit form checks a quotient per digit)
        uint32_t v32 = 0;
        for (; i < e; i++) {
            const uint32_t d = src.b(i) - '0';
            if (d > 9u) return false;
            v32 = v32 * 10u + d;
        }
        *out = (long long)v32;
        return true;
    }
    unsigned long long v = 0;
    for (; i < e; i++) {
        uint32_t c = src.b(i);
        if (c < '0' || c > '9') return false;
        if (v > (0x7FFFFFFFFFFFFFFFull - (c - '0')) / 10) return false;
        v = v * 10 + (c - '0');
    }
    *out = (long long)v;
    return true;
}

// i32::from_str: optional sign, one or more digits, no overflow
template <class Src>
__device__ bool parse_i32(const Src &src, int s, int e, int *out) {
    int i = s;
    bool neg = false;
    if (i < e && (src.b(i) == '+' || src.b(i) == '-')) neg = src.b(i++) == '-';
    if (i >= e) return false;
    if (e - i <= 9) {  // cannot overflow: 32-bit multiply-adds
        uint32_t v32 = 0;
        for (; i < e; i++) {
            const uint32_t d = src.b(i) - '0';
            if (d > 9u) return false;
            v32 = v32 * 10u + d;
        }
        *out = neg ? -(int)v32 : (int)v32;
        return true;
    }
    long long v = 0;
    for (; i < e; i++) {
        uint32_t c = src.b(i);
        if (c < '0' || c > '9') return false;
        v = v * 10 + (long long)(c - '0');
        if (v > 2147483648ll) return false;
    }
    if (neg) v = -v;
    if (v > 2147483647ll) return false;
    *out = (int)v;
    return true;
}

}  // namespace exg
