// exg_float_slow.hpp — f32::from_str for the literals exg_parse.hpp cannot decide with 19 digits: more than 19
// significant digits AND a float rounding boundary strictly inside the interval those digits pin down (about one literal in
// 10^11; the ones people write on purpose: "halfway between two floats, then twenty zeros, then a 1").  Exact, by big-integer
// comparison of the literal with that boundary — what Rust's dec2flt slow path (and glibc's strtof) decide.  It runs in
// single-block fix-up code (k_vcf_finalize, k_f32_slow), never inside the scan kernels: its arrays would cost every scan
// wave registers or scratch.  Compiles for the host too (tests/float_el_check.cpp checks it against strtof).
#pragma once
#include <stdint.h>

#include "exg_float_el.hpp"

namespace exg {

// unsigned big integer, little-endian 32-bit limbs (enough for 128 digits x 10^45 x 2^150)
struct BigU {
    static constexpr int kLimbs = 40;
    uint32_t w[kLimbs];
    int n;  // limbs in use
    EXG_HD void set(uint32_t v) {
        w[0] = v;
        n = v ? 1 : 0;
    }
    EXG_HD bool mul_add(uint32_t m, uint32_t a) {  // this = this * m + a; false on overflow
        uint64_t carry = a;
        for (int i = 0; i < n; i++) {
            const uint64_t t = (uint64_t)w[i] * m + carry;
            w[i] = (uint32_t)t;
            carry = t >> 32;
        }
        if (carry) {
            if (n >= kLimbs) return false;
            w[n++] = (uint32_t)carry;
        }
        return true;
    }
    EXG_HD bool shl(int bits) {
        while (bits > 0) {
            const int s = bits > 31 ? 31 : bits;
            if (!mul_add(1u << s, 0)) return false;
            bits -= s;
        }
        return true;
    }
    EXG_HD bool mul_pow10(int e) {
        while (e > 0) {
            const int s = e > 9 ? 9 : e;
            uint32_t m = 1;
            for (int i = 0; i < s; i++) m *= 10u;
            if (!mul_add(m, 0)) return false;
            e -= s;
        }
        return true;
    }
};
EXG_HD inline int big_cmp(const BigU &a, const BigU &b) {
    if (a.n != b.n) return a.n < b.n ? -1 : 1;
    for (int i = a.n - 1; i >= 0; i--)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
}

// Correctly rounded float32 of the decimal literal p[0, n) (the f32::from_str grammar: [+-] digits [. digits] [e [+-] digits],
// or inf / infinity / nan).  Returns 0 and *bits, or 1 on a syntax error.  Any number of digits.
EXG_HD inline int f32_parse_exact(const uint8_t *p, int n, uint32_t *bits) {
    int i = 0;
    if (n <= 0) return 1;
    uint32_t sign = 0;
    if (p[0] == '+' || p[0] == '-') {
        sign = p[0] == '-' ? 0x80000000u : 0u;
        i = 1;
    }
    const int rest = n - i;
    if (rest == 3 || rest == 8) {
        const uint32_t a = p[i] | 0x20, b = p[i + 1] | 0x20, c = p[i + 2] | 0x20;
        if (rest == 3 && a == 'n' && b == 'a' && c == 'n') {
            *bits = 0x7FC00000u;
            return 0;
        }
        bool inf = a == 'i' && b == 'n' && c == 'f';
        if (inf && rest == 8) {
            const char *tail = "inity";
            for (int k = 0; k < 5; k++) inf = inf && (p[i + 3 + k] | 0x20) == (uint32_t)tail[k];
        }
        if (inf) {
            *bits = sign | 0x7F800000u;
            return 0;
        }
    }
    // significant digits: the first 19 as a machine word (Eisel-Lemire), the first kKeep as a big integer, the rest sticky
    constexpr int kKeep = 128;
    uint64_t m19 = 0;
    BigU D;
    D.set(0);
    int sig = 0, nd = 0, e10 = 0, e10_big = 0;
    bool sticky19 = false, sticky_big = false, seen_dot = false;
    for (; i < n; i++) {
        const uint32_t c = p[i];
        if (c == '.') {
            if (seen_dot) return 1;
            seen_dot = true;
            continue;
        }
        if (c < '0' || c > '9') break;
        nd++;
        const uint32_t d = c - '0';
        if (sig == 0 && d == 0) {  // leading zero
            if (seen_dot) e10--, e10_big--;
            continue;
        }
        if (sig < 19) {
            m19 = m19 * 10 + d;
            if (seen_dot) e10--;
        } else {
            if (d) sticky19 = true;
            if (!seen_dot) e10++;
        }
        if (sig < kKeep) {
            D.mul_add(10, d);
            if (seen_dot) e10_big--;
        } else {
            if (d) sticky_big = true;
            if (!seen_dot) e10_big++;
        }
        sig++;
    }
    if (nd == 0) return 1;
    if (i < n) {
        if (p[i] != 'e' && p[i] != 'E') return 1;
        i++;
        bool eneg = false;
        if (i < n && (p[i] == '+' || p[i] == '-')) eneg = p[i++] == '-';
        if (i >= n) return 1;
        int ev = 0;
        for (; i < n; i++) {
            const uint32_t c = p[i];
            if (c < '0' || c > '9') return 1;
            if (ev < 100000) ev = ev * 10 + (int)(c - '0');
        }
        e10 += eneg ? -ev : ev;
        e10_big += eneg ? -ev : ev;
    }
    if (sig == 0) {
        *bits = sign;
        return 0;
    }
    bool tie_lo = false, tie_hi = false;
    uint32_t lo = el_f32_bits(m19, e10, &tie_lo);
    if (sticky19) {
        const uint32_t up = el_f32_bits(m19 + 1, e10, &tie_hi);
        if (up != lo) {
            if (tie_lo) {
                lo = up;
            } else if (!tie_hi) {
                // the boundary between the adjacent floats lo and up lies strictly inside (m19, m19 + 1) x 10^e10:
                // compare the literal D x 10^e10_big with B = (2 M + 1) x 2^(E - 1), lo = M x 2^E
                const uint32_t ef = lo >> 23, frac = lo & 0x7FFFFFu;
                const uint32_t M = ef ? frac | 0x800000u : frac;
                const int E = (ef ? (int)ef : 1) - 150;
                BigU A = D, B;
                B.set(2 * M + 1);
                const int t = E - 1;
                bool ok = true;
                if (e10_big > 0) ok = ok && A.mul_pow10(e10_big);
                if (e10_big < 0) ok = ok && B.mul_pow10(-e10_big);
                if (t > 0) ok = ok && B.shl(t);
                if (t < 0) ok = ok && A.shl(-t);
                if (!ok) return 1;  // (cannot happen within the float range: the limbs cover it)
                const int c = big_cmp(A, B);
                if (c > 0 || (c == 0 && (sticky_big || (M & 1)))) lo = up;  // above the boundary, or a tie to even
            }
        }
    }
    *bits = sign | lo;
    return 0;
}

}  // namespace exg
