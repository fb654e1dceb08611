// exg_common.hpp — shared host/device helpers of libexon_gpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/exon_gpu.h"

namespace exg {

// thread-local error text returned by exg_last_error_message()
void set_error(const char *fmt, ...);

#define EXG_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::exg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                             __LINE__);                                                       \
            return EXG_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

static_assert(sizeof(exg_string_t) == 16, "duckdb::string_t is 16 bytes");
static_assert(sizeof(exg_scan_result) == 64, "exg_scan_result is 64 bytes");

// ---- device helpers -----------------------------------------------------------
#if defined(__HIPCC__)

// SWAR byte match: 0x80 in every byte of w equal to the byte replicated in pat4. Exact (no
// borrow false positives): y keeps bit 7 clear only where all of x's low 7 bits are 0.
__device__ __forceinline__ uint32_t match4(uint32_t w, uint32_t pat4) {
    uint32_t x = w ^ pat4;
    uint32_t y = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(y | x | 0x7F7F7F7Fu);
}

// bits 7,15,23,31 -> 4-bit nibble.  Round 6: by V_DOT4_U32_U8 — the four bytes (0x80 or 0) times the weights 1, 2, 4, 8 is 128 x the
// nibble: one full-rate instruction and a shift where the multiply that gathered them (m * 0x00204081 >> 28) was a quarter-rate
// V_MUL_LO_U32; a chunk's 16-bit mask is four dots (the second of a pair adds into the first) instead of four multiplies, four
// shifts and three ORs.  The FASTA scan is VALU-issue bound (profiles/r06_c_pmc_sq_k_fa_fused.csv) and builds three such masks a row.
// -DEXG_NO_DOT4: the multiply form (A/B).
#ifndef EXG_NO_DOT4
__device__ __forceinline__ uint32_t nib4(uint32_t m) { return __builtin_amdgcn_udot4(m, 0x08040201u, 0u, false) >> 7; }
__device__ __forceinline__ uint32_t match16(uint4 v, uint32_t pat4) {
    const uint32_t lo = __builtin_amdgcn_udot4(match4(v.y, pat4), 0x80402010u, __builtin_amdgcn_udot4(match4(v.x, pat4), 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(match4(v.w, pat4), 0x80402010u, __builtin_amdgcn_udot4(match4(v.z, pat4), 0x08040201u, 0u, false), false);
    return ((hi << 8) + lo) >> 7;  // (lo, hi: 128 x their byte of the mask)
}
#else
__device__ __forceinline__ uint32_t nib4(uint32_t m) { return (m * 0x00204081u) >> 28; }

// 16-bit match mask of one 16-byte chunk
__device__ __forceinline__ uint32_t match16(uint4 v, uint32_t pat4) {
    return nib4(match4(v.x, pat4)) | (nib4(match4(v.y, pat4)) << 4) | (nib4(match4(v.z, pat4)) << 8) |
           (nib4(match4(v.w, pat4)) << 12);
}
#endif

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// 16 bytes of the input stream / of an output column.  The input is read once (twice by the FASTA passes, far apart) and
// the columns are written once: non-temporal accesses.  A/B inside one box (tools/ab_bench.sh, ab_vcf.sh, ab_fasta.sh;
// -DEXG_NO_NT builds the plain form): FASTQ 2.34 -> 2.27 ms per 10 GB, VCF 5.07 -> 4.86 ms per 5.25 GB, FASTA
// 1.10 -> 1.08 ms per GB; loads alone or stores alone gave a third of that or nothing.
#ifndef EXG_NO_NT
#define EXG_NT_LOAD 1
#define EXG_NT_STORE 1
#endif
typedef uint32_t exg_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream16(const uint8_t *p) {
#ifdef EXG_NT_LOAD
    const exg_v4u v = __builtin_nontemporal_load(reinterpret_cast<const exg_v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4 *>(p);
#endif
}
__device__ __forceinline__ void st_stream16(uint4 *p, uint4 v) {
#ifdef EXG_NT_STORE
    exg_v4u w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<exg_v4u *>(p));
#else
    *p = v;
#endif
}

// inclusive wave64 prefix sum
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

// inclusive wave64 prefix sum on the VALU (DPP row shifts + row broadcasts: no LDS traffic, unlike __shfl_up).
// Not a free win: in the fused FASTQ kernel, which is VALU-bound with an idle LDS pipeline, it measured 4 % SLOWER
// than the ds_bpermute form above (A/B on one box); it pays where the LDS pipeline is the busy one (FASTA tiles).
__device__ __forceinline__ uint32_t wave_incl_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// Build a duckdb::string_t for the field d_in[s, s+len) reading bytes from global memory.
__device__ __forceinline__ uint4 make_string_global(const uint8_t *d_in, uint64_t s, uint64_t len,
                                                     uint64_t payload_base) {
    uint4 r;
    r.x = (uint32_t)len;
    r.y = r.z = r.w = 0;
    if (len <= EXG_INLINE_LENGTH) {
        uint32_t w[3] = {0, 0, 0};
        for (uint32_t i = 0; i < (uint32_t)len; i++) w[i >> 2] |= (uint32_t)d_in[s + i] << (8 * (i & 3));
        r.y = w[0];
        r.z = w[1];
        r.w = w[2];
    } else {
        r.y = (uint32_t)d_in[s] | ((uint32_t)d_in[s + 1] << 8) | ((uint32_t)d_in[s + 2] << 16) |
              ((uint32_t)d_in[s + 3] << 24);
        uint64_t p = payload_base + s;
        r.z = (uint32_t)p;
        r.w = (uint32_t)(p >> 32);
    }
    return r;
}

// core::str::from_utf8 acceptance over d_in[s, e)
__device__ inline bool utf8_valid_global(const uint8_t *__restrict__ p, uint64_t s, uint64_t e) {
    uint64_t i = s;
    while (i < e) {
        uint32_t b = p[i];
        if (b < 0x80) {
            i++;
            continue;
        }
        if (b >= 0xC2 && b <= 0xDF) {
            if (i + 1 >= e || (p[i + 1] & 0xC0) != 0x80) return false;
            i += 2;
        } else if (b >= 0xE0 && b <= 0xEF) {
            if (i + 2 >= e) return false;
            uint32_t c1 = p[i + 1], c2 = p[i + 2];
            uint32_t lo = b == 0xE0 ? 0xA0 : 0x80, hi = b == 0xED ? 0x9F : 0xBF;
            if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80) return false;
            i += 3;
        } else if (b >= 0xF0 && b <= 0xF4) {
            if (i + 3 >= e) return false;
            uint32_t c1 = p[i + 1], c2 = p[i + 2], c3 = p[i + 3];
            uint32_t lo = b == 0xF0 ? 0x90 : 0x80, hi = b == 0xF4 ? 0x8F : 0xBF;
            if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80 || (c3 & 0xC0) != 0x80) return false;
            i += 4;
        } else {
            return false;
        }
    }
    return true;
}


#endif  // __HIPCC__

// error word packing: (record index << 8) | code, atomicMin picks the first failing record.
static constexpr uint64_t kNoError = ~0ull;

}  // namespace exg
