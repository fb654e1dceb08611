// exg_float_el.hpp — decimal -> float32, correctly rounded, for a decimal significand w (< 2^64, up to 19 digits) and a
// decimal exponent q: the Eisel-Lemire algorithm (Lemire, "Number parsing at a gigabyte per second", SP&E 2021; the
// fallback-free form of Mushtak & Lemire, "Fast number parsing without fallback", 2023): one or two 64 x 64 -> 128 bit
// multiplications by a 128-bit approximation of 5^q decide the 24-bit mantissa, ties included.  This is what makes the
// device's f32::from_str exact beyond Clinger's fast path (<= 15 digits, |q| <= 22) without a big-integer slow path in
// the scan kernels.  Compiles for the device (table in constant memory) and for the host (tests/test_float_el.py checks it
// against strtof on millions of literals).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define EXG_HD __host__ __device__
#else
#define EXG_HD
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define EXG_POW5_STORAGE static __device__ __constant__ const
#else
#define EXG_POW5_STORAGE static const
#endif
#include "exg_pow5_table.hpp"

namespace exg {

struct U128 {
    uint64_t lo, hi;
};
EXG_HD inline U128 mul_64x64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return U128{a * b, __umul64hi(a, b)};
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    return U128{(uint64_t)p, (uint64_t)(p >> 64)};
#endif
}
EXG_HD inline int clz_u64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)v);
#else
    return __builtin_clzll(v);
#endif
}

// bits of the float nearest to w x 10^q (round half to even); w == 0 gives +0.  *tie (optional): w x 10^q lies exactly
// halfway between two floats
EXG_HD inline uint32_t el_f32_bits(uint64_t w, int q, bool *tie = nullptr) {
    if (tie) *tie = false;
    constexpr int kMantBits = 23, kMinExp = -127, kInfPower = 0xFF;
    if (w == 0 || q < kPow5Lo + 1) return 0u;  // w 10^q < 2^64 10^-65 < half the smallest subnormal
    if (q > kPow5Hi) return 0x7F800000u;
    const int lz = clz_u64(w);
    w <<= lz;
    // w x 5^q to kMantBits + 3 bits: the high word of the table entry first, the low word only when the bits below the
    // precision are all ones (the product could still carry)
    const int idx = 2 * (q - kPow5Lo);
    U128 p = mul_64x64(w, kPow5[idx]);
    const uint64_t precision_mask = 0xFFFFFFFFFFFFFFFFull >> (kMantBits + 3);
    if ((p.hi & precision_mask) == precision_mask) {
        const U128 p2 = mul_64x64(w, kPow5[idx + 1]);
        p.lo += p2.hi;
        if (p2.hi > p.lo) p.hi++;
    }
    const int upperbit = (int)(p.hi >> 63);
    const int shift = upperbit + 64 - kMantBits - 3;
    uint64_t mant = p.hi >> shift;
    int power2 = (((152170 + 65536) * q) >> 16) + 63 + upperbit - lz - kMinExp;
    if (power2 <= 0) {  // subnormal
        if (-power2 + 1 >= 64) return 0u;
        mant >>= -power2 + 1;
        mant += mant & 1;
        mant >>= 1;
        power2 = mant < (1ull << kMantBits) ? 0 : 1;
        return (uint32_t)((mant & ((1ull << kMantBits) - 1)) | ((uint64_t)power2 << kMantBits));
    }
    // exactly between two floats (only possible for small |q|): round to even
    if (p.lo <= 1 && q >= -17 && q <= 10 && (mant & 1) == 1 && (mant << shift) == p.hi) {
        if (tie) *tie = true;
        if ((mant & 3) == 1) mant &= ~1ull;
    }
    mant += mant & 1;
    mant >>= 1;
    if (mant >= (2ull << kMantBits)) {
        mant = 1ull << kMantBits;
        power2++;
    }
    mant &= ~(1ull << kMantBits);
    if (power2 >= kInfPower) return 0x7F800000u;
    return (uint32_t)(mant | ((uint64_t)power2 << kMantBits));
}

}  // namespace exg
