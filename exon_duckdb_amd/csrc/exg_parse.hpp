// exg_parse.hpp — number parsing shared by the VCF tokeniser (POS, QUAL) and the typed INFO / FORMAT
// columns.  `Src` is any byte source with `uint32_t b(int i) const`.
#pragma once
#include "exg_common.hpp"

namespace exg {

// ---- exact decimal -> float32 ---------------------------------------------------------------------
// f32::from_str is correctly rounded.  Device domain: <= 15 significant digits (mantissa < 2^53),
// |decimal exponent| <= 22, result 0 or a normal float.  One correctly rounded f64 operation
// (Clinger) + an FMA-exact residual decides the single case a double-rounded conversion could get
// wrong (the f64 result sitting exactly on a float rounding boundary).  Literals outside the
// domain are reported (EXG_PE_VCF_BAD_QUAL + EXG_RF_QUAL_RANGE), never mis-rounded.
static __device__ __constant__ double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                             1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

__device__ __forceinline__ float round_exact(double q, double resid) {
    // true value = q + (something with the sign of resid, magnitude < 1/2 ulp(q))
    float f = (float)q;  // round to nearest even of q itself
    unsigned long long b = (unsigned long long)__double_as_longlong(q);
    if ((b & 0x1FFFFFFFull) == 0x10000000ull && resid != 0.0) {
        float t = (float)__longlong_as_double((long long)(b & ~0x1FFFFFFFull));  // q truncated to 24 bits
        f = resid > 0.0 ? __uint_as_float(__float_as_uint(t) + 1u) : t;
    }
    return f;
}

// status: 0 ok, 1 syntax error, 2 outside the exact device domain
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
        uint32_t m32 = 0;
        int sig32 = 0, frac = 0, nd32 = 0;
        bool dot = false, plain = true;
        for (int j = i; j < e; j++) {
            const uint32_t ch = src.b(j);
            if (ch == '.') {
                if (dot) plain = false;
                dot = true;
                continue;
            }
            const uint32_t d = ch - '0';
            if (d > 9u) {
                plain = false;
                break;
            }
            nd32++;
            if (m32 || d) {
                m32 = m32 * 10u + d;
                sig32++;
            }
            if (dot) frac++;
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
    if (inexact || m >= (1ull << 53) || e10 < -22 || e10 > 22) return 2;
    double dm = (double)m, q, resid;
    if (e10 < 0) {
        double p = kPow10[-e10];
        q = dm / p;
        resid = __fma_rn(-q, p, dm);  // m - q p, exact
    } else {
        double p = kPow10[e10];
        q = dm * p;
        resid = __fma_rn(dm, p, -q);  // m p - q, exact
    }
    if (!(q >= 1.1754943508222875e-38 && q <= 3.4028234663852886e38)) return 2;
    float f = round_exact(q, resid);
    *out = neg ? -f : f;
    return 0;
}

// usize::from_str: optional '+', one or more digits, no overflow (63 bits kept)
template <class Src>
__device__ bool parse_pos(const Src &src, int s, int e, long long *out) {
    int i = s;
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
