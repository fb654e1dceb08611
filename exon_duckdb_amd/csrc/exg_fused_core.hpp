// exg_fused_core.hpp — the single-pass skeleton shared by the fused record-scan kernels.
//
// (see exg_fastq_fused.hip for the design rationale and the measurements behind it)
//   * 256-thread workgroup per super-tile of F::kHalves x 16 KiB held in registers, processed one 16 KiB half at a
//     time through one LDS buffer (+ the 1 KiB window that precedes the half);
//   * newline count published per super-tile, exclusive prefix from the central scanner wave
//     (block 0), awaited only after half 0 has been staged;
//   * per half: bytes -> LDS, contiguous 64 B per thread re-classified from LDS (conflict-free
//     read order), one wave scan -> u16 newline list; format policy F emits the records.
// A format policy F provides:
//   typename F::Dev                       by-value kernel argument (d_in, n_bytes, lead, flags,
//                                         first_line_index, payload_base, capacity + outputs)
//   F::eof_extra_lines(line_index_total)  extra virtual (empty) lines at EOF besides the unterminated one
//   F::emit_half(...)                     records ending in the staged half
//   F::analytic_prefix(offset)            dev-only ablation hook
//   F::kNlCap                             newline positions kept in LDS at a time: a half with more lines (short reads, short
//                                         VCF lines) is emitted in passes of kNlCap lines each
//   F::write_far(...)                     (inside emit_half) the ONE record per half that begins before the window: a FarRec
//                                         for the k_*_far kernel behind this one (exg_fastq_ws.hpp) — any record length stays
//                                         on this single pass over the input
//   F::kTabMapLean / kTabMapFull          also keep a '\t' bitmap of every half (VCF's lean scan: short lines, every byte lies within 64 bytes of a
//                                         line start; not its any-shape scan: on wide lines only the first nine tabs of a line are ever looked at)
//   F::kHalves                            16 KiB halves per workgroup (bytes waiting in registers: 16 VGPRs each)
//   F::kMinWavesPerSimd                   occupancy the register allocator must respect (FASTQ lean scan: 6 = six workgroups per CU at
//                                         80 VGPRs and 24.7 KiB of LDS each; VCF: 5; the any-shape instances one less)
#pragma once
#include "exg_fastq_ws.hpp"

#ifndef EXG_FIRST_POLL_SLEEP
#define EXG_FIRST_POLL_SLEEP 0
#endif
#ifndef EXG_POLL_SLEEP
#define EXG_POLL_SLEEP 2
#endif

namespace exg {


static constexpr int kTile = kFusedTileBytes;  // 16384: one half, the unit of LDS staging and of tile_qend
static constexpr int kMaxHalves = 4;  // a format picks F::kHalves 16 KiB halves per workgroup (its bytes wait in registers)
static constexpr int kWin = kFusedWindow;       // 1024
static constexpr int kThreads = 256;
static constexpr int kRows = kTile / (kThreads * 16);  // 4 chunk rows (4 KiB each) per half
static constexpr int kLdsBytes = kWin + kTile + 96;
static constexpr uint32_t kNoneE = 0xFFFFu;

static constexpr uint32_t kFlagA = 1u << 31;        // tileA[t]: the count is published
static constexpr uint32_t kNonAsciiA = 1u << 30;    // ... and the super-tile (or the 1 KiB window in front of it) holds a byte >= 0x80
static constexpr uint32_t kCountA = kNonAsciiA - 1; // ... the count itself (<= 49 152)
static constexpr unsigned long long kFlag = 1ull << 63;  // descriptor word is published
static constexpr unsigned long long kVal = (1ull << 48) - 1;

// NL = newline positions kept per half (a half with more lines goes to the general path)
template <int NL, int H, bool TABS = false>
struct FusedLdsT {
    static constexpr int kNlCap = NL;
    static constexpr bool kHasTabs = TABS;
    // TABS (VCF): '\t' mask of every 16-byte chunk next to the '\n' one; a line's first eight tabs are then eight
    // bit pops out of one funnel-shifted 64-bit word instead of a SWAR search over the line's bytes.  Rows carry
    // 8 bytes of slack so that the word after a line's last one can always be read.
    static constexpr int kTabRow = TABS ? kTile / 16 + 4 : 4;
    __attribute__((aligned(8))) uint16_t tabmap[TABS ? H : 1][kTabRow];
    uint8_t bytes[kLdsBytes];        // [0,kWin) window, then the half; e = p + kWin
    uint16_t nlist[4 + NL + 4];  // e-offsets of newlines: [0..3] the 4 before the half (oldest first)
    uint16_t bitmap[H][kTile / 16];  // '\n' mask of every 16-byte chunk, written by the first pass
    uint32_t wtot[4];   // per-wave newline counts of the staged half
    uint32_t wcnt[4];   // per-wave packed (half 0 | half 1 << 16) newline counts
    unsigned long long prefix;            // '\n' in the buffer before this super-tile
    uint32_t hi_or[4];
    // the 4 newlines before the staged half / pass (oldest first) as CODES (FarRec::pos): >= 0 offset inside the super-tile,
    // -1 - j: the j-th newest newline in front of the super-tile; nlist[0..3] holds those that lie in the LDS window
    int32_t prev32[4];
    int32_t carry32[4];  // ... before the NEXT half (written when a half's list is complete)
    uint16_t carry[4];   // (lean scan) the 4 newlines before the next half, relative to it: those inside its window
};

template <class L>
__device__ __forceinline__ uint32_t ldw(const L &s, uint32_t e_aligned) {
    return *reinterpret_cast<const uint32_t *>(s.bytes + e_aligned);
}
template <class L>
__device__ __forceinline__ uint32_t ldb(const L &s, int e) { return s.bytes[e]; }
// 4 bytes at extended offset e (any alignment: LDS accesses need none on gfx950)
typedef uint32_t lds_u32u __attribute__((aligned(1)));
typedef uint32_t lds_v3u __attribute__((ext_vector_type(3), aligned(1)));
template <class L>
__device__ __forceinline__ uint32_t ldu32(const L &s, int e) {
    return *reinterpret_cast<const lds_u32u *>(s.bytes + e);
}

// duckdb::string_t of the field [e, e+len); ptr_of_e0 = payload pointer of extended offset 0
template <class L>
__device__ __forceinline__ uint4 make_string_lds(const L &s, int e, uint32_t len, uint64_t ptr_of_e0) {
    uint4 r;
    r.x = len;
    if (len <= EXG_INLINE_LENGTH) {
        const lds_v3u w = *reinterpret_cast<const lds_v3u *>(s.bytes + e);  // one 12-byte read
        uint32_t m0 = len >= 4 ? 0xFFFFFFFFu : ((1u << (8 * len)) - 1u);
        uint32_t l1 = len > 4 ? len - 4 : 0, l2 = len > 8 ? len - 8 : 0;
        uint32_t m1 = l1 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * l1)) - 1u);
        uint32_t m2 = l2 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * l2)) - 1u);
        r.y = w.x & m0;
        r.z = w.y & m1;
        r.w = w.z & m2;
    } else {
        uint64_t ptr = ptr_of_e0 + (uint64_t)e;
        r.y = ldu32(s, e);
        r.z = (uint32_t)ptr;
        r.w = (uint32_t)(ptr >> 32);
    }
    return r;
}

// core::str::from_utf8 acceptance over the LDS bytes [b, e) (extended offsets)
template <class L>
__device__ inline bool utf8_valid_lds(const L &s, int b, int e) {
    int i = b;
    while (i < e) {
        const uint32_t c0 = ldb(s, i);
        if (c0 < 0x80) {
            i++;
            continue;
        }
        if (c0 >= 0xC2 && c0 <= 0xDF) {
            if (i + 1 >= e || (ldb(s, i + 1) & 0xC0) != 0x80) return false;
            i += 2;
        } else if (c0 >= 0xE0 && c0 <= 0xEF) {
            if (i + 2 >= e) return false;
            const uint32_t c1 = ldb(s, i + 1), c2 = ldb(s, i + 2);
            const uint32_t lo = c0 == 0xE0 ? 0xA0 : 0x80, hi = c0 == 0xED ? 0x9F : 0xBF;
            if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80) return false;
            i += 3;
        } else if (c0 >= 0xF0 && c0 <= 0xF4) {
            if (i + 3 >= e) return false;
            const uint32_t c1 = ldb(s, i + 1), c2 = ldb(s, i + 2), c3 = ldb(s, i + 3);
            const uint32_t lo = c0 == 0xF0 ? 0x90 : 0x80, hi = c0 == 0xF4 ? 0x8F : 0xBF;
            if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80 || (c3 & 0xC0) != 0x80) return false;
            i += 4;
        } else {
            return false;
        }
    }
    return true;
}
// k_*_far: does a super-tile that holds a byte of [lo, hi] — or the window in front of one — hold a byte >= 0x80?
__device__ inline bool tiles_non_ascii(const unsigned int *__restrict__ tileA, uint64_t super_bytes, int64_t lo, int64_t hi) {
    if (lo < 0) lo = 0;
    if (hi < lo) return false;
    for (uint64_t t = (uint64_t)lo / super_bytes, t1 = (uint64_t)hi / super_bytes; t <= t1; t++)
        if (tileA[t] & kNonAsciiA) return true;
    return false;
}

// '\n' count of bytes [b, e), by one wave, straight from global memory (helping path)
__device__ unsigned long long help_count_bytes(const uint8_t *__restrict__ d_in, uint64_t n_bytes, uint64_t b,
                                               uint64_t e, uint32_t lane) {
    unsigned long long cnt = 0;
    if (e > n_bytes) e = n_bytes;
    for (uint64_t off = b + (uint64_t)lane * 16; off < e; off += 1024) {
        uint4 q = *reinterpret_cast<const uint4 *>(d_in + off);
        uint32_t mm = match16(q, 0x0A0A0A0Au);
        if (off + 16 > e) mm &= (1u << (uint32_t)(e - off)) - 1u;
        cnt += __popc(mm);
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    return __shfl(cnt, 0, 64);
}

// x, but the compiler may not compute anything that depends on it ahead of this point: the three halves of a super-tile are
// unrolled, and addresses like &s.carry32[lane] — three VALU instructions — are otherwise computed once, kept across the
// halves and, at 80 registers, SPILLED: a scratch reload in front of an LDS access makes the wave wait (vmcnt counts loads and
// stores alike on gfx9) for every column store it has in flight, which cost the 10 GB FASTQ launch 6 % (A/B in one box).
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ uint32_t opaque_s(uint32_t x) {  // the same for a wave-uniform value (a scalar register)
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ uint64_t opaque_s64(uint64_t x) {
    asm volatile("" : "+s"(x));
    return x;
}
// (a format says whether its lean scan wants these barriers — F::kBarriers: FASTQ does, at 80 registers; the VCF scan, at 96,
// ran 3 % slower with them: A/B in one box)
template <bool kOn> __device__ __forceinline__ uint32_t opaque_if(uint32_t x) { return kOn ? opaque(x) : x; }
template <bool kOn> __device__ __forceinline__ uint32_t opaque_s_if(uint32_t x) { return kOn ? opaque_s(x) : x; }
template <bool kOn> __device__ __forceinline__ uint64_t opaque_s64_if(uint64_t x) { return kOn ? opaque_s64(x) : x; }
__device__ __forceinline__ unsigned long long rfl64(unsigned long long x) {  // wave-uniform value -> SGPRs
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long ld_desc(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_desc(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- ordered prefix: central scanner --------------------------------------------------------------
static constexpr int kScanBatches = 8;  // 1024 descriptors per scanner probe (32-bit count words keep this in registers)

template <int kSuper>
__device__ void scanner_wave(const uint8_t *__restrict__ d_in, uint64_t n_bytes,
                             const unsigned int *__restrict__ tileA, unsigned long long *__restrict__ tileP,
                             uint32_t n_super, uint32_t lane) {
    __builtin_amdgcn_s_setprio(3);
    uint64_t next = 0;
    unsigned long long running = 0;
    unsigned long long t_last = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (next < n_super) {
        unsigned int d[kScanBatches];
#pragma unroll
        for (int k = 0; k < kScanBatches; k++) {
            uint64_t idx = next + (uint64_t)k * 64 + lane;
            d[k] = idx < n_super ? __hip_atomic_load(&tileA[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
        bool progressed = false;
#pragma unroll
        for (int k = 0; k < kScanBatches; k++) {
            unsigned long long rdy = __ballot((d[k] & kFlagA) != 0);
            int r = rdy == ~0ull ? 64 : __ffsll((long long)~rdy) - 1;  // leading run of published counts
            if (r > 0) {
                uint32_t c = (int)lane < r ? (d[k] & kCountA) : 0u;
                uint32_t inc = wave_incl_sum(c);
                if ((int)lane < r) st_desc(&tileP[next + lane], kFlag | (running + inc - c));
                running += __shfl(inc, 63, 64);
                next += (uint64_t)r;
                progressed = true;
            }
            if (r < 64) break;
        }
        if (progressed) {
            t_last = __builtin_amdgcn_s_memrealtime();
        } else if (__builtin_amdgcn_s_memrealtime() - t_last > 4000) {
            // ~40 us without the next count: that block may not have been dispatched; count its bytes
            // ourselves so that progress never depends on the dispatch order.
            unsigned long long c = help_count_bytes(d_in, n_bytes, next * kSuper, (next + 1) * kSuper, lane);
            if (lane == 0) st_desc(&tileP[next], kFlag | running);
            running += c;
            next++;
            t_last = __builtin_amdgcn_s_memrealtime();
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// Workgroup side (wave 0): wait for the exclusive prefix of super-tile st (its count is published).
template <int kSuper>
__device__ unsigned long long wait_prefix(const uint8_t *__restrict__ d_in, uint64_t n_bytes,
                                          unsigned int *__restrict__ tileA,
                                          unsigned long long *__restrict__ tileP, uint32_t st, uint32_t lane) {
    if (st == 0) return 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_sleep(EXG_FIRST_POLL_SLEEP);
    for (;;) {
        unsigned long long x = lane == 0 ? ld_desc(&tileP[st]) : 0ull;
        x = (unsigned long long)__shfl((long long)x, 0, 64);
        if (x & kFlag) return x & kVal;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000) break;  // 2 ms: the scanner is not running
        __builtin_amdgcn_s_sleep(EXG_POLL_SLEEP);
    }
    // Last resort (never seen with in-order dispatch): sum every predecessor ourselves.
    unsigned long long sum = 0;
    for (uint64_t b = 0; b < st; b += 64) {
        uint64_t idx = b + lane;
        unsigned long long x = kFlag;
        if (idx < st) {
            unsigned int a32 = __hip_atomic_load(&tileA[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x = (a32 & kFlagA) ? (kFlag | (a32 & kCountA)) : 0ull;
        }
        unsigned long long miss = __ballot((x & kFlag) == 0);
        while (miss) {
            int l = __ffsll((long long)miss) - 1;
            miss &= miss - 1;
            unsigned long long c = help_count_bytes(d_in, n_bytes, (b + l) * kSuper, (b + l + 1) * kSuper, lane);
            if ((int)lane == l) x = kFlag | c;
        }
        unsigned long long v = idx < st ? (x & kVal) : 0;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        sum += __shfl(v, 0, 64);
    }
    return sum;
}

// ---- k_*_far: the newline positions of a FarRec as offsets into the buffer -------------------------------------------
// pos[k] >= 0: inside super-tile st; -1 - j: the j-th newest newline in front of it — found by walking back over the tiles:
// tileA[t] says how many newlines tile t holds, tileL[4 t ..] its last four (slot 3 the newest; a tile with c < 4 newlines has
// them in slots 4 - c .. 3).  With EXG_F_BOF a line begins at d_input[0]: a newline at -1.  false: a position lies in front
// of the buffer (the record's head is not in it).  Runs behind k_fused: every word is final, plain reads.
template <int N>
__device__ inline bool far_positions(const FarRec &f, uint32_t st, uint64_t super_bytes, const unsigned int *__restrict__ tileA,
                                     const int32_t *__restrict__ tileL, bool bof, int64_t *p) {
    int need = 0;
#pragma unroll
    for (int k = 0; k < N; k++)
        if (f.pos[k] < 0 && -f.pos[k] > need) need = -f.pos[k];
    int64_t back0 = 0, back1 = 0, back2 = 0, back3 = 0;  // the j-th newest newline in front of the super-tile
    int found = 0;
    for (uint32_t t = st; found < need && t > 0;) {
        t--;
        uint32_t c = tileA[t] & kCountA;
        if (!c) continue;
        if (c > 4) c = 4;
        const int4 l = *reinterpret_cast<const int4 *>(tileL + (uint64_t)t * 4);
        const int64_t base = (int64_t)((uint64_t)t * super_bytes);
        const int32_t e[4] = {l.w, l.z, l.y, l.x};  // newest first
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if ((uint32_t)q < c && found < need) {
                const int64_t v = base + e[q];
                if (found == 0) back0 = v;
                else if (found == 1) back1 = v;
                else if (found == 2) back2 = v;
                else back3 = v;
                found++;
            }
        }
    }
    if (found < need && bof) {
        const int64_t v = -1;
        if (found == 0) back0 = v;
        else if (found == 1) back1 = v;
        else if (found == 2) back2 = v;
        else back3 = v;
        found++;
    }
    if (found < need) return false;
    const int64_t super_off = (int64_t)((uint64_t)st * super_bytes);
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int32_t c = f.pos[k];
        p[k] = c >= 0 ? super_off + c : c == -1 ? back0 : c == -2 ? back1 : c == -3 ? back2 : back3;
    }
    return true;
}

struct TileCtx {  // what emission needs besides the LDS contents (all workgroup-uniform)
    uint64_t tile_off;          // offset of the half in d_input
    unsigned long long P;       // '\n' in the buffer before the half
    uint32_t n_lines;           // newline entries of the half (incl. virtual EOF lines)
    int lim_e;                  // extended offset of the end of input inside this half (kWin + min(lim, kTile))
    bool is_eof_tile;           // the input ends in this half and EXG_F_EOF
    bool first_of_buffer;       // half 0 of super-tile 0: what precedes is before d_input[0]
    bool non_ascii;             // the super-tile or its window holds a byte >= 0x80: the any-shape scan validates UTF-8
    uint32_t pass_base;         // lines of the half emitted by earlier passes (0: first pass; P and n_lines are the pass's)
    int half;                   // index of the half inside its super-tile
};

// ---- the scan, in three instances -------------------------------------------------------------------------------------
// kLean        the scan of ordinary inputs (150 bp reads, 50-byte VCF lines): what is measured as the headline.  A record
//              that begins in front of its half's window, a half with more lines than the list holds, a super-tile whose
//              last four newlines it cannot name: it MARKS the super-tile (tile_redo) and goes on.  It runs at the edge of
//              its 80 registers: a register spilled in its hot path makes the wave wait for all its column stores at every
//              reload (vmcnt counts loads and stores alike on gfx9) — with the any-shape code inside it the 10 GB FASTQ
//              launch took 2.35 - 2.87 ms against 2.25 (A/B in one box, six forms) although that code never ran.  So the
//              any-shape code is an instance of its own:
// kFullRedo    behind the lean scan on the stream, a fixed grid striding over the MARKED super-tiles with the prefix the
//              scanner published: all their halves again, any shape (below);
// kFullPrimary the any-shape scan as the launch's only scan (EXG_ALGO_FUSED_FULL): what a reader switches to when a
//              batch came back with marks (long reads, 36 bp reads with short names, multi-sample VCF lines) — such an
//              input pays the lean scan once.
// Any shape: the half's lines are emitted kNlCap at a time (the 4 newlines in front of a pass are the last 4 of the pass
// before); the record that begins in front of the window — at most one per half — is written as a FarRec for k_*_far; the 4
// newlines in front of a half are carried as CODES (FarRec::pos), not as window offsets.
// (kFullIndex: kFullPrimary with the format's "index" emission — VcfFormat: the half notes where its lines end, a kernel behind
// the scan parses the rows; an instantiation of its own so that the any-shape scan proper keeps its registers and schedule)
enum { kLean = 0, kFullPrimary = 1, kFullRedo = 2, kFullIndex = 3 };
static constexpr unsigned int kRedoFar = 1u, kRedoDense = 2u, kRedoLast4 = 4u, kRedoUtf8 = 8u;  // tile_redo[st]: why (diagnostics; any bit = redo)

// tile_redo (u32 per super-tile) lies between tileP and tile_qend: tileA | tileP | tile_redo are zeroed by one memset
template <bool kBarriers = true>
__device__ __forceinline__ unsigned int *tile_redo_of(unsigned long long *tile_qend, uint64_t n_bytes) {
    return reinterpret_cast<unsigned int *>(tile_qend - fused_n_tiles(opaque_s64_if<kBarriers>(n_bytes)));  // (computed where it is used: see opaque)
}
template <bool kBarriers = true>
__device__ __forceinline__ int32_t *tile_last4_of(unsigned long long *tile_qend, uint64_t n_bytes) {
    return reinterpret_cast<int32_t *>(tile_qend + fused_n_tiles(opaque_s64_if<kBarriers>(n_bytes)));
}
template <bool kBarriers = true>
__device__ __forceinline__ FarRec *far_rec_of(unsigned long long *tile_qend, uint64_t n_bytes) {
    return reinterpret_cast<FarRec *>(tile_qend + 3 * fused_n_tiles(opaque_s64_if<kBarriers>(n_bytes)));
}

template <class F, int kMode>
__global__ __launch_bounds__(kThreads, kMode == kLean ? F::kMinWavesPerSimd : (kMode == kFullPrimary || kMode == kFullIndex) ? F::kMinWavesPerSimdFull : F::kMinWavesPerSimdRedo) void k_fused(
    typename F::Dev a, unsigned int *__restrict__ tileA, unsigned long long *__restrict__ tileP, unsigned long long *__restrict__ tile_qend,
    ScanWsHeader *hdr, uint32_t n_super) {
    // (the any-shape scan that runs ALONE may cut its super-tiles differently from the lean scan and its redo run, which share
    // tile_redo: every per-half / per-super-tile array of the workspace is sized by 16 KiB tiles)
    constexpr int kHalves = (kMode == kFullPrimary || kMode == kFullIndex) ? F::kHalvesFull : F::kHalves;
    using FusedLds = FusedLdsT<F::kNlCap, kHalves, (kMode == kLean ? F::kTabMapLean : F::kTabMapFull)>;
    constexpr int kNlCap = F::kNlCap;
    constexpr int kSuper = kTile * kHalves;
    constexpr bool kFull = kMode != kLean;
    constexpr bool kB = F::kBarriers;
    __shared__ __attribute__((aligned(16))) FusedLds s;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    // DEV ONLY (tools/dev_probe.py): flags bits 8..11 select ablations on the synthetic FASTQ-150 file
    //   1: analytic prefix instead of the scanner   2: 1 + no output stores
    //   3: 1 + no emission at all                    4: scanner, no emission
#ifdef EXG_DEV_PROBE
    const uint32_t dev_mode = (a.flags >> 8) & 15u;  // ablations of tools/dev_probe.py (no prefix wait / no stores / no emission)
#else
    constexpr uint32_t dev_mode = 0;  // the product build carries no work-skipping mode (EXG_CXXFLAGS=-DEXG_DEV_PROBE builds them in)
#endif
    uint32_t st;
    if constexpr (kMode == kFullRedo) {
        if (!hdr->any_redo) return;
        st = blockIdx.x;
    } else {
        if (blockIdx.x == 0) {  // the scanner: one wave, no tile
            if (wave == 0 && !(dev_mode >= 1 && dev_mode <= 3)) scanner_wave<kHalves * kTile>(a.d_in, a.n_bytes, tileA, tileP, n_super, lane);
            return;
        }
        st = blockIdx.x - 1;
    }
    uint32_t n_redone = 0;  // (redo run) super-tiles this workgroup redid
    for (; kMode != kFullRedo || st < n_super; st += gridDim.x) {  // (one super-tile per workgroup but in the redo run)
    if constexpr (kMode == kFullRedo) {
        if (!tile_redo_of<kB>(tile_qend, a.n_bytes)[st]) continue;  // (workgroup-uniform)
        __syncthreads();                                         // the tile before is done with the LDS
        n_redone++;
    }
    const uint64_t super_off = (uint64_t)st * kSuper;
    const uint8_t *__restrict__ d_in = a.d_in;
    const uint64_t n_pad = (a.n_bytes + 15) & ~15ull;
    const int64_t lim64 = (int64_t)a.n_bytes - (int64_t)super_off;
    const int lim_s = lim64 < kSuper ? (int)lim64 : kSuper;  // super-tile-relative end of input (> 0)
    const bool last_super = st + 1 == n_super;

    // ---- loads: 8 strided 16 B chunks per thread, all in flight at once (+ window by wave 3) ------
    uint4 v[kHalves * kRows];
    if (lim_s == kSuper) {
        const uint8_t *mine = d_in + super_off + (uint64_t)tid * 16;
#pragma unroll
        for (int j = 0; j < kHalves * kRows; j++) v[j] = ld_stream16(mine + j * (kThreads * 16));
    } else {
#pragma unroll
        for (int j = 0; j < kHalves * kRows; j++) {
            uint64_t off = super_off + (uint64_t)(j * kThreads + tid) * 16;
            v[j] = off < n_pad ? ld_stream16(d_in + off) : make_uint4(0, 0, 0, 0);
        }
    }
    uint4 wv = make_uint4(0, 0, 0, 0);
    const int64_t woff = (int64_t)super_off - kWin + (int64_t)lane * 16;  // wave 3 only
    if (wave == 3 && woff >= 0) wv = *reinterpret_cast<const uint4 *>(d_in + woff);

    // ---- classify ONCE, in registers: 16-bit '\n' mask per chunk -> LDS bitmap; count for the publish ----
    // (the kernel is instruction-issue bound: classifying again when a half is staged cost 20 % more VALU)
    uint32_t hi = 0, cnt = 0;
#pragma unroll
    for (int j = 0; j < kHalves * kRows; j++) {
        uint32_t mj = match16(v[j], 0x0A0A0A0Au);
        hi |= v[j].x | v[j].y | v[j].z | v[j].w;
        if (lim_s != kSuper) {  // the input ends inside this super-tile: mask the bytes past the end
            int rem = lim_s - (int)(j * kThreads + tid) * 16;
            if (rem < 16) mj &= rem <= 0 ? 0u : ((1u << rem) - 1u);
        }
        s.bitmap[j / kRows][(j % kRows) * kThreads + tid] = (uint16_t)mj;
        if constexpr (FusedLds::kHasTabs) s.tabmap[j / kRows][(j % kRows) * kThreads + tid] = (uint16_t)match16(v[j], 0x09090909u);
        cnt += __popc(mj);
    }
    hi &= 0x80808080u;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (wave == 3) hi |= (wv.x | wv.y | wv.z | wv.w) & 0x80808080u;
    uint32_t any_hi = __any(hi != 0);
    if (lane == 0) {
        s.hi_or[wave] = any_hi;
        s.wcnt[wave] = cnt;
    }
    __syncthreads();  // #1
    const uint32_t n_nl_super = __builtin_amdgcn_readfirstlane(s.wcnt[0] + s.wcnt[1] + s.wcnt[2] + s.wcnt[3]);
    unsigned long long nl_before_half = 0;  // '\n' in the halves of this super-tile already processed
    const bool non_ascii = (s.hi_or[0] | s.hi_or[1] | s.hi_or[2] | s.hi_or[3]) != 0;

    // ---- publish the super-tile count; its prefix is awaited after half 0 has been staged ---------------
    const bool analytic = dev_mode >= 1 && dev_mode <= 3;
    if constexpr (kMode != kFullRedo) {
        if (tid == 0 && !analytic)
            __hip_atomic_store(&tileA[st], kFlagA | n_nl_super | (non_ascii ? kNonAsciiA : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long halo_nl = rfl64(hdr->halo_nl);
    // Bytes >= 0x80 need UTF-8 validation of every field (the reference builds Arrow Utf8 columns).  That is rare and not the
    // lean scan's business: it marks the super-tile; the any-shape scan validates the fields of the records that end here
    // (emit_half, from LDS; a record that begins in front of the window: k_*_far, which finds the flag in tileA)
    if constexpr (kMode != kFullRedo) {
        if (non_ascii && tid == 0) {
            atomicOr(&hdr->flags, EXG_RF_NON_ASCII);
            if constexpr (kMode == kLean) {
                tile_redo_of<kB>(tile_qend, a.n_bytes)[opaque_s_if<kB>(st)] = kRedoUtf8;
                hdr->any_redo = 1u;
            }
        }
    }

#pragma unroll
    for (int h = 0; h < kHalves; h++) {
        const int lim_h = lim_s - h * kTile;  // half-relative end of input
        if (h > 0 && lim_h <= 0) {
            // no input in this half: nothing ends here
            if (tid == 0)
                for (int hh = h; hh < kHalves; hh++) tile_qend[(uint64_t)st * kHalves + hh] = 0;
            break;
        }
        // ---- stage the half: window, bytes, newline list ------------------------------------------
        if (h == 0) {
            if (wave == 3) {
                *reinterpret_cast<uint4 *>(s.bytes + lane * 16) = wv;
                if (lane < 4) s.nlist[lane] = (uint16_t)kNoneE;
                if constexpr (kFull) {
                    if (lane < 4) s.prev32[lane] = (int32_t)lane - 4;  // slot 3 = the newest newline in front of the super-tile
                }
                uint32_t wm = woff >= 0 ? match16(wv, 0x0A0A0A0Au) : 0u;
                uint32_t wc = __popc(wm);
                uint32_t wincl = wave_incl_sum(wc);
                uint32_t W = __shfl(wincl, 63, 64);
                uint32_t r = wincl - wc;  // rank, oldest first; goes to slot 4 - (W - r) when >= 0
                while (wm) {
                    uint32_t b = __ffs(wm) - 1;
                    wm &= wm - 1;
                    int slot = 4 - (int)(W - r);
                    if (slot >= 0) s.nlist[slot] = (uint16_t)(lane * 16 + b);
                    r++;
                }
                // a line starts at d_input[0] when EXG_F_BOF: model it as a newline at offset -1 (W == 0 here)
                if (st == 0 && (a.flags & EXG_F_BOF) && lane == 0) s.nlist[3] = (uint16_t)(kWin - 1);
            }
        } else {
            // window = last 1 KiB of the previous half (still in LDS); its last 4 newlines were saved
            uint4 t = make_uint4(0, 0, 0, 0);
            if (wave == 3) t = *reinterpret_cast<const uint4 *>(s.bytes + kTile + lane * 16);
            __syncthreads();
            if (wave == 3) {
                *reinterpret_cast<uint4 *>(s.bytes + lane * 16) = t;
                if constexpr (kFull) {
                    if (lane < 4) {
                        const int32_t code = s.carry32[lane], rel = code - (h * kTile - kWin);  // e-offset in this half's buffer
                        s.prev32[lane] = code;
                        s.nlist[lane] = (code >= 0 && rel >= 0) ? (uint16_t)rel : (uint16_t)kNoneE;
                    }
                } else {
                    if (lane < 4) s.nlist[lane] = s.carry[lane];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kRows; j++)
            *reinterpret_cast<uint4 *>(s.bytes + kWin + (j * kThreads + tid) * 16) = v[h * kRows + j];
        uint32_t n_nl_h = 0;  // '\n' in this half
        {
            // Each thread now owns 64 CONTIGUOUS bytes of the half (4 chunks): their masks are one 8-byte
            // read of the bitmap, so newline ranks follow from one 32-bit wave scan.
            unsigned long long mask = *reinterpret_cast<const unsigned long long *>(&s.bitmap[h][tid * 4]);
            uint32_t c = (uint32_t)__popcll(mask);
            uint32_t inc = wave_incl_sum(c);
            if (lane == 63) s.wtot[wave] = inc;
            __syncthreads();
            uint32_t r = inc - c + (wave > 0 ? s.wtot[0] : 0) + (wave > 1 ? s.wtot[1] : 0) + (wave > 2 ? s.wtot[2] : 0);
            n_nl_h = __builtin_amdgcn_readfirstlane(s.wtot[0] + s.wtot[1] + s.wtot[2] + s.wtot[3]);
            const uint32_t e0 = kWin + tid * 64;
            while (mask) {
                uint32_t b = (uint32_t)__ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if (r < (uint32_t)kNlCap) s.nlist[4 + r] = (uint16_t)(e0 + b);
                r++;
            }
        }
        if (h == 0) {
            // the half is staged while the scanner turns the published count into our prefix
            if (wave == 0) {
                unsigned long long pre;
                if constexpr (kMode == kFullRedo) {
                    pre = st ? ld_desc(&tileP[st]) & kVal : 0ull;  // (published: the lean scan has completed)
                } else if (analytic) {
                    pre = F::analytic_prefix(super_off);
                } else {
                    pre = wait_prefix<kSuper>(d_in, a.n_bytes, tileA, tileP, st, lane);
                }
                if (lane == 0) s.prefix = pre;
            }
        }
        __syncthreads();  // staged (and, for h == 0, the prefix has arrived)

        TileCtx c;
        c.tile_off = super_off + (uint64_t)h * kTile;
        c.P = rfl64(s.prefix) + nl_before_half;
        nl_before_half += n_nl_h;
        c.first_of_buffer = st == 0 && h == 0;
        c.pass_base = 0;
        c.half = h;
        c.non_ascii = non_ascii;
        const bool ends_here = last_super && lim_h <= kTile;  // the input ends inside (or at the end of) this half
        c.is_eof_tile = ends_here && (a.flags & EXG_F_EOF);
        c.lim_e = (lim_h < kTile ? lim_h : kTile) + kWin;
        uint32_t n_lines = n_nl_h;
        if (c.is_eof_tile) {
            // noodles EOF rules: an unterminated last line is a line; a record with its '+' line but no
            // quality line gets an empty one (read_line returns 0 bytes at EOF without error).
            const unsigned long long P0 = a.first_line_index - halo_nl;
            bool unterminated = a.n_bytes > 0 && ldb(s, c.lim_e - 1) != '\n';
            uint32_t n0 = n_lines;
            if (unterminated) n_lines++;
            n_lines += F::eof_extra_lines(P0 + c.P + n_lines);
            if (tid == 0)
                for (uint32_t q = n0; q < n_lines && q < (uint32_t)kNlCap; q++) s.nlist[4 + q] = (uint16_t)c.lim_e;
            __syncthreads();
        }
        c.n_lines = n_lines;
        if (ends_here && tid == 0) {
            hdr->total_nl = c.P + n_nl_h;
            hdr->total_lines = c.P + n_lines;
        }
        const uint64_t half_index = (uint64_t)st * kHalves + h;
        if constexpr (!kFull) {
            // ---- the lean scan ------------------------------------------------------------------------------
            if (n_lines > (uint32_t)kNlCap) {  // more lines than the list holds: the any-shape run redoes this super-tile
                if (tid == 0) {
                    tile_redo_of<kB>(tile_qend, a.n_bytes)[opaque_s_if<kB>(st)] = kRedoDense;
                    hdr->any_redo = 1u;
                }
                return;
            }
            if (h + 1 < kHalves && tid < 4) {
                // the 4 newlines before the next half, relative to it (entries 4+n-4 .. 4+n-1 of this list)
                uint32_t e = s.nlist[n_lines + tid];
                s.carry[tid] = (e != kNoneE && e >= (uint32_t)kTile) ? (uint16_t)(e - kTile) : (uint16_t)kNoneE;
            }
            if (h + 1 == kHalves || lim_s <= (h + 1) * kTile) {  // (no input behind this half in the super-tile)
                // the super-tile's last four newlines (tileL: what k_*_far's look-back reads): slot 3 the newest.  The ones
                // this tile holds (n_nl_super of them) must be nameable from the last half's list; if not — sparse
                // newlines: long lines — the any-shape run redoes the super-tile and names them.
                if (wave == 0) {
                    const uint32_t k = opaque_if<kB>(lane), idx = n_lines + k;  // lanes 0 .. 3: entries n .. n + 3 counted from nlist[0]
                    const uint32_t st_here = opaque_s_if<kB>(st);               // (nothing of this block may be computed ahead: see opaque)
                    int32_t code = -1;
                    if (k < 4) {
                        const uint32_t e = s.nlist[idx];
                        code = e != kNoneE ? h * kTile + (int32_t)e - kWin : -1;
                        tile_last4_of<kB>(tile_qend, a.n_bytes)[(uint64_t)st_here * 4 + k] = code;
                    }
                    const uint32_t need = n_nl_super < 4u ? n_nl_super : 4u;  // slots 4 - need .. 3
                    if (__ballot(k < 4 && k >= 4 - need && code < 0) != 0 && k == 0) {
                        tile_redo_of<kB>(tile_qend, a.n_bytes)[st_here] = kRedoLast4;
                        hdr->any_redo = 1u;
                    }
                }
            }
            F::template emit_half<kLean>(s, a, hdr, c, halo_nl, dev_mode, lane, wave, tile_qend, half_index);
        } else {
            // ---- any shape ----------------------------------------------------------------------------------
            // positions of the newlines of ranks [base, base + kNlCap) -> the list (a later pass of a dense half)
            auto fill_list = [&](uint32_t base) {
                unsigned long long mask = *reinterpret_cast<const unsigned long long *>(&s.bitmap[h][tid * 4]);
                uint32_t cc = (uint32_t)__popcll(mask);
                uint32_t inc = wave_incl_sum(cc);
                uint32_t r = inc - cc + (wave > 0 ? s.wtot[0] : 0) + (wave > 1 ? s.wtot[1] : 0) + (wave > 2 ? s.wtot[2] : 0) - base;
                const uint32_t e0 = kWin + tid * 64;  // (r mod 2^32: ranks below `base` compare as huge)
                while (mask) {
                    uint32_t b = (uint32_t)__ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    if (r < (uint32_t)kNlCap) s.nlist[4 + r] = (uint16_t)(e0 + b);
                    r++;
                }
                if (tid == 0)  // virtual EOF lines (at most two) that fall into the pass
                    for (uint32_t q = n_nl_h > base ? n_nl_h : base; q < n_lines && q < base + (uint32_t)kNlCap; q++)
                        s.nlist[4 + q - base] = (uint16_t)c.lim_e;
            };
            const unsigned long long P_half = c.P;
#pragma unroll 1
            for (uint32_t base = 0;;) {
                const uint32_t m = n_lines - base < (uint32_t)kNlCap ? n_lines - base : (uint32_t)kNlCap;
                c.P = P_half + base;
                c.pass_base = base;
                c.n_lines = m;
                if (base + m >= n_lines && tid < 4) {
                    // the 4 newlines before the next half (entries m .. m + 3 of the last pass's list, counted from nlist[0]) as
                    // CODES; the last half's are the last 4 of the super-tile: tileL
                    const uint32_t idx = m + tid;
                    const int32_t code = idx >= 4 ? h * kTile + (int32_t)s.nlist[idx] - kWin : s.prev32[idx];
                    s.carry32[tid] = code;
                    if (h + 1 == kHalves || lim_s <= (h + 1) * kTile) tile_last4_of<kB>(tile_qend, a.n_bytes)[(uint64_t)st * 4 + tid] = code;
                }
                F::template emit_half<(kMode == kFullIndex ? kFullIndex : kFullPrimary)>(s, a, hdr, c, halo_nl, dev_mode, lane, wave, tile_qend, half_index);
                base += m;
                if (base >= n_lines) break;
                uint16_t keep = 0;
                if (tid < 4) keep = s.nlist[m + tid];  // (m == kNlCap) the last 4 entries of this pass
                __syncthreads();                       // everyone is done reading the list
                if (tid < 4) {
                    s.nlist[tid] = keep;
                    s.prev32[tid] = h * kTile + (int32_t)keep - kWin;
                }
                fill_list(base);
                __syncthreads();
            }
        }
        if (h + 1 < kHalves) __syncthreads();  // everyone is done reading this half
    }
    if constexpr (kMode != kFullRedo) break;
    }
    if constexpr (kMode == kFullRedo) {
        if (tid == 0 && n_redone) atomicAdd(&hdr->n_redo, n_redone);  // (what the caller's choice of scan for the next batch looks at)
    }
}


}  // namespace exg
