// exg_vcf.hip — VCF record scan (read_vcf_file_records / read_vcf): one row per data line, the 8
// fixed tab-separated columns CHROM POS ID REF ALT QUAL FILTER INFO (+ the FORMAT/sample remainder)
// as duckdb::string_t slices of the input, POS parsed to int64, QUAL to float32.
//
// Replaces the tokenising the reference gets from noodles-vcf 0.34.0 through exon 0.2.6
// (rust/src/arrow_reader.rs:116-153; registered at exon/src/exon_extension.cpp:55).  The
// reference's LIST / STRUCT typing of id/alt/filter/info/formats is the next row of the scope
// table and is not done here.
//
// Two implementations behind exg_vcf_scan:
//   * fused (exg_fused_core.hpp skeleton): thread = line, fields cut out of LDS;
//     a line that begins in front of the half's 1 KiB window (multi-sample VCF) is left to k_vcf_far behind the
//     kernel — at most one per half, read from global memory (exg_fastq_ws.hpp FarRec) —, a half with more lines than
//     its list holds is emitted in passes: the single pass over the input holds for any line length;
//   * general: line index (exg_lines.hip) + thread = line reading global memory — the differential partner, and the
//     path of inputs with non-ASCII bytes.
// Header lines (leading '#') are found by the host and passed as `lead`: lines that end before
// `lead` are not rows.
#include <stdlib.h>

#include "exg_fused_core.hpp"
#include "exg_lines.hpp"

#include "exg_parse.hpp"
#include "exg_float_slow.hpp"

namespace exg {

struct VcfDev {
    const uint8_t *d_in;
    uint64_t n_bytes;
    uint64_t lead;
    uint64_t first_line_index;  // unused (kept for the core's EOF arithmetic): 0
    uint64_t payload_base;
    uint32_t flags;
    uint32_t pad;
    exg_string_t *d_fields[9];
    int64_t *d_pos;
    float *d_qual;
    uint64_t *d_qual_valid;
    uint64_t *d_formats_valid;
    uint64_t capacity;
    SlowLiteral *d_slow;  // workspace list of the QUAL literals left to the exact parser
    uint64_t slow_cap;
    // EXG_ALGO_FUSED_INDEX: the any-shape scan writes where line j of the buffer ends to d_nl_pos[j] instead of parsing it
    // (nl_cap entries; NULL: it parses), k_vcf_lines parses behind it
    uint64_t *d_nl_pos;
    uint64_t nl_cap;
};

struct VcfRowInfo {
    bool qual_valid, rest_valid;
    uint32_t code;
    int slow_s, slow_len;  // slow_len > 0: the QUAL literal [slow_s, slow_s + slow_len) is left to the exact parser
};

// One data line [s, e) (CR already stripped), stored straight to row `out` (store == false: validate
// only).  Src: b(i) byte, u32(i) 4 bytes at any alignment, str(i, len) -> string_t.
// Everything is statically indexed (runtime-indexed arrays would live in scratch memory).
template <class Src>
__device__ __forceinline__ VcfRowInfo vcf_line(const Src &src, int s, int e, const VcfDev &a, unsigned long long out,
                                               bool store) {
    VcfRowInfo r;
    r.code = 0;
    r.qual_valid = false;
    r.rest_valid = false;
    r.slow_s = r.slow_len = 0;
    // positions of the first 8 tabs (e when there are fewer)
    int t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = e;
    int found = 0;
    bool have_tabs = false;
    {
        // fused path: the tab bits of the 64 bytes from s on come out of the half's bitmap; eight pops
        unsigned long long tb;
        if (src.tab_bits(s, &tb)) {
            const int len = e - s;
            if (len < 64) tb &= len <= 0 ? 0ull : ((1ull << len) - 1ull);
            if (len <= 64 || __popcll(tb) >= 8) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (tb) {
                        t[k] = s + __ffsll((long long)tb) - 1;
                        tb &= tb - 1;
                        found++;
                    }
                }
                have_tabs = true;
            }
        }
    }
    // otherwise (no map: the any-shape scan; a line starting in the window, or long with few tabs in front; general path): 64 bytes at a time
    for (int base = s; !have_tabs && base < e && found < 8; base += 64) {
        unsigned long long bits = src.tabs64(base);
        int rem = e - base;
        if (rem < 64) bits &= (1ull << rem) - 1ull;
        while (bits && found < 8) {
            int pos = base + __ffsll((long long)bits) - 1;
            bits &= bits - 1;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k == found) t[k] = pos;
            found++;
        }
    }
    // field k = [fs_k, t[k]); fs_0 = s, fs_k = t[k-1] + 1; it exists iff fs_k <= e
    if (found < 7) {
        r.code = EXG_PE_VCF_MISSING_FIELD;
        return r;
    }
    long long pos_v = 0;
    if (!parse_pos(src, t[0] + 1, t[1], &pos_v)) {
        r.code = EXG_PE_VCF_BAD_POS;
        return r;
    }
    float qual_v = 0.f;
    if (!(t[5] - (t[4] + 1) == 1 && src.b(t[4] + 1) == '.')) {
        int st = parse_f32(src, t[4] + 1, t[5], &qual_v);
        if (st == 2) {  // > 19 digits astride a rounding boundary: value (and sign check) by the finalize kernel
            r.slow_s = t[4] + 1;
            r.slow_len = t[5] - (t[4] + 1);
            qual_v = 0.f;
            st = 0;
        }
        if (st) {
            r.code = EXG_PE_VCF_BAD_QUAL;
            return r;
        }
        // noodles-vcf 0.34 record::QualityScore: TryFrom<f32> refuses n < 0.0 (so -1, -inf; not -0, not NaN)
        if (qual_v < 0.0f) {
            r.code = EXG_PE_VCF_BAD_QUAL;
            return r;
        }
        r.qual_valid = true;
    }
    r.rest_valid = found == 8;
    if (store) {
#ifdef EXG_DEV_PROBE
        if (((a.flags >> 8) & 15u) != 13) {
#endif
        if (a.d_pos) a.d_pos[out] = pos_v;
        if (a.d_qual) a.d_qual[out] = qual_v;
#ifdef EXG_DEV_PROBE
        }
#endif
        int fs = s;
#ifdef EXG_DEV_PROBE
        const uint32_t dm = (a.flags >> 8) & 15u;  // ablations: 7 = stores of (length, 0, 0, 0) — no string construction; 8 = construction, one store
        if (dm == 7 || dm == 8) {
            uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (dm == 7) {
                    if (a.d_fields[k]) st_stream16(reinterpret_cast<uint4 *>(a.d_fields[k]) + out, make_uint4((uint32_t)(t[k] - fs), 0, 0, 0));
                } else {
                    const uint4 q = src.str(fs, (uint32_t)(t[k] - fs));
                    acc.x ^= q.x, acc.y ^= q.y, acc.z ^= q.z, acc.w ^= q.w;
                }
                fs = t[k] + 1;
            }
            if (dm == 8 && a.d_fields[0]) st_stream16(reinterpret_cast<uint4 *>(a.d_fields[0]) + out, acc);
            return r;
        }
#endif
#pragma unroll
        for (int k = 0; k < 8; k++) {
#ifdef EXG_DEV_PROBE
            if (dm == 10) {
                fs = t[k] + 1;
                continue;  // (ablation: only the remainder's string_t)
            }
            if (dm == 12) {
                if (a.d_fields[k]) reinterpret_cast<uint4 *>(a.d_fields[k])[out] = src.str(fs, (uint32_t)(t[k] - fs));  // (plain stores)
                fs = t[k] + 1;
                continue;
            }
#endif
            if (a.d_fields[k]) st_stream16(reinterpret_cast<uint4 *>(a.d_fields[k]) + out, src.str(fs, (uint32_t)(t[k] - fs)));
            fs = t[k] + 1;
        }
#ifdef EXG_DEV_PROBE
        if (dm == 9) return r;  // (ablation: the eight fixed fields, not the FORMAT + samples remainder)
#endif
        if (a.d_fields[8]) {
            uint4 z = {0, 0, 0, 0};
            st_stream16(reinterpret_cast<uint4 *>(a.d_fields[8]) + out, r.rest_valid ? src.str(fs, (uint32_t)(e - fs)) : z);
        }
    }
    return r;
}

// validity bits of 64 consecutive rows starting at out_base (may be negative for unowned lanes)
__device__ __forceinline__ void store_validity64(uint64_t *words, unsigned long long bits, long long out_base,
                                                 uint32_t lane) {
    if (!words || !bits) return;
    if (out_base < 0) {
        bits >>= (unsigned long long)(-out_base);
        out_base = 0;
    }
    if (lane == 0 && bits) {
        uint32_t sh = (uint32_t)(out_base & 63);
        unsigned long long lo = bits << sh, hi = sh ? bits >> (64 - sh) : 0;
        if (lo) atomicOr((unsigned long long *)&words[out_base >> 6], lo);
        if (hi) atomicOr((unsigned long long *)&words[(out_base >> 6) + 1], hi);
    }
}

__device__ __forceinline__ void vcf_report(ScanWsHeader *hdr, uint32_t code, unsigned long long out, uint64_t line_off) {
    atomicMin(&hdr->err_word, (out << 8) | (code & 0x7Fu));
    atomicMin(&hdr->err_off, (unsigned long long)line_off);
}
// a QUAL literal for the exact parser (k_vcf_finalize).  The list holds one literal per 32 bytes of input, at most 4096 per
// launch (such a literal has more than 19 digits and sits astride a float rounding boundary: one in 10^11 by chance); more
// than that is an error (EXG_RF_QUAL_RANGE)
__device__ __forceinline__ void vcf_slow_qual(ScanWsHeader *hdr, const VcfDev &a, uint64_t lit_off, uint32_t len, unsigned long long out,
                                              uint64_t line_off) {
    const unsigned int k = atomicAdd(&hdr->n_slow, 1u);
    if (k < a.slow_cap && out <= 0xFFFFFFFFull) {
        a.d_slow[k].off = lit_off;
        a.d_slow[k].len = len;
        a.d_slow[k].row = (unsigned int)out;
    } else {
        atomicOr(&hdr->flags, EXG_RF_QUAL_RANGE);
        vcf_report(hdr, EXG_PE_VCF_BAD_QUAL, out, line_off);
    }
}

// ---- fused ------------------------------------------------------------------------------------------
template <class L>
struct LdsSrc {
    const L &s;
    uint64_t ptr_of_e0;
    const uint16_t *tabs;  // '\t' bitmap of the staged half (bit p = byte kWin + p)
    __device__ __forceinline__ uint32_t b(int e) const { return ldb(s, e); }
    __device__ __forceinline__ uint32_t u32(int e) const { return ldu32(s, e); }  // reads stay inside the LDS slack
    __device__ __forceinline__ void u96(int e, uint32_t *w0, uint32_t *w1, uint32_t *w2) const {  // 12 bytes, one read
        const lds_v3u w = *reinterpret_cast<const lds_v3u *>(s.bytes + e);
        *w0 = w.x, *w1 = w.y, *w2 = w.z;
    }
    // '\t' mask of the 64 bytes from extended offset e on, classified here: four 16-byte reads at any alignment
    __device__ __forceinline__ unsigned long long tabs64(int e) const {
        typedef uint32_t lds_v4u __attribute__((ext_vector_type(4), aligned(1)));
        unsigned long long bits = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const lds_v4u w = *reinterpret_cast<const lds_v4u *>(s.bytes + e + 16 * q);
            bits |= (unsigned long long)match16(make_uint4(w.x, w.y, w.z, w.w), 0x09090909u) << (16 * q);
        }
        return bits;
    }
    // tab bits of the 64 bytes starting at extended offset e out of the half's map (only for lines that start inside the half)
    __device__ __forceinline__ bool tab_bits(int e, unsigned long long *out) const {
        if (!L::kHasTabs) return false;
        const int p = e - kWin;
        if (p < 0) return false;
        const unsigned long long *w = reinterpret_cast<const unsigned long long *>(tabs) + (p >> 6);
        const uint32_t sh = (uint32_t)p & 63u;
        const unsigned long long lo = w[0], hi = w[1];
        *out = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
        return true;
    }
    __device__ __forceinline__ uint4 str(int e, uint32_t len) const { return make_string_lds(s, e, len, ptr_of_e0); }
};

#ifndef EXG_VCF_HALVES
#define EXG_VCF_HALVES 2
#endif
static constexpr int kVcfHalves = EXG_VCF_HALVES;
struct VcfFormat {
    using Dev = VcfDev;
    static constexpr int kNlCap = 1024;  // short data lines are common
    static constexpr int kHalves = kVcfHalves;
#ifndef EXG_VCF_HALVES_FULL
#define EXG_VCF_HALVES_FULL 2
#endif
    static constexpr int kHalvesFull = EXG_VCF_HALVES_FULL;  // the any-shape scan alone (EXG_ALGO_FUSED_FULL)
    // The lean scan (short lines: 49 bytes in BASELINE's config 3) classifies '\t' with '\n' in its single pass over the registers:
    // every byte lies within 64 bytes of a line start, so every mask is used.  The any-shape scan — what a reader runs on
    // cohort VCFs, lines of 400 bytes to 10 kB whose first nine tabs are all anybody looks at — does not: a line's tabs come
    // from its first 64-byte words in LDS, read when the line is cut (LdsSrc::tabs64): 99 % of the tab classification of such
    // a file, the 2 KiB map per half and its stores are gone (round 5: 2.04 -> TB/s on 100-sample lines)
    static constexpr bool kTabMapLean = true, kTabMapFull = false;
    static constexpr bool kBarriers = false;  // (exg_fused_core.hpp opaque: its lean scan ran 3 % slower with them)
#ifndef EXG_VCF_WAVES
#define EXG_VCF_WAVES 5
#endif
    static constexpr int kMinWavesPerSimd = EXG_VCF_WAVES;
#ifndef EXG_VCF_WAVES_FULL
#define EXG_VCF_WAVES_FULL 5
#endif
    static constexpr int kMinWavesPerSimdRedo = 4;  // the redo launch: few tiles, latency of one tile matters, not occupancy (128 VGPRs, 28 B of scratch)
    static constexpr int kMinWavesPerSimdFull = EXG_VCF_WAVES_FULL;  // the any-shape scan alone: 96 VGPRs + 32 B of scratch, and still 12 % faster on wide
    // (multi-sample) lines than at 4 waves without scratch — those lines write little, so the scan is bound by bytes in flight per CU
    __device__ static __forceinline__ uint32_t eof_extra_lines(unsigned long long) { return 0; }
    __device__ static __forceinline__ unsigned long long analytic_prefix(uint64_t) { return 0; }

    template <int kMode, class L>
    __device__ static __forceinline__ void emit_half(const L &s, const VcfDev &a, ScanWsHeader *hdr,
                                                     const TileCtx &c, unsigned long long halo_nl, uint32_t dev_mode,
                                                     uint32_t lane, uint32_t wave,
                                                     unsigned long long *__restrict__ tile_qend, uint64_t tile_index) {
        // (-DEXG_VCF_ROTATE: which wave takes which 64 lines rotates with the half — thread = line, so a half of wide lines has
        // ONE busy wave and a half of short ones 1.3 passes of 256; tried in case the busy waves of a CU's workgroups shared a SIMD)
#ifdef EXG_VCF_ROTATE
        const uint32_t rot_w = (uint32_t)tile_index & 3u;
#else
        const uint32_t rot_w = 0;  // (measured, A/B in one box: rotating is 2-4 % SLOWER on wide and on short lines alike — kept as a switch)
#endif
        const uint32_t t_rot = (threadIdx.x - rot_w * 64u) & (uint32_t)(kThreads - 1);  // this thread's line of a pass of 256
        unsigned long long qend_word = 0;  // (thread t_rot == 0: what it stored into tile_qend)
        if (t_rot == 0 && (c.pass_base == 0 || c.n_lines)) {  // offset just past the last line that ends in this half (0: none)
            long long e = 0;
            if (c.n_lines) {
                e = (long long)c.tile_off + (int)s.nlist[4 + c.n_lines - 1] - kWin + 1;
                if ((unsigned long long)e > a.n_bytes) e = (long long)a.n_bytes;
            }
            if (c.pass_base) e |= (long long)(tile_qend[tile_index] & kFarBit);  // (a later pass keeps the first pass's mark)
            qend_word = (unsigned long long)e;
            tile_qend[tile_index] = qend_word;
        }
        if (dev_mode == 3 || dev_mode == 4) return;
        if constexpr (kMode == kFullIndex) {
            // EXG_ALGO_FUSED_INDEX (round 5).  A half of a cohort VCF holds 1-40 lines: thread = line leaves ONE wave running each
            // line's ~1 500 dependent instructions while its sisters wait — the scan is bound by that latency per workgroup (2.2-2.3
            // TB/s; without the rows 3.1-3.9).  Here the half only says where its lines end — an 8-byte store a line, halo lines
            // included (the line behind them begins at theirs), and the line that began in front of the window is a line like any
            // other: no FarRec — and k_vcf_lines parses all rows of the buffer behind the scan, a thread per line, every line's
            // chain beside thousands of others.
            for (uint32_t jb = 0; jb < c.n_lines; jb += kThreads) {
                const uint32_t j = jb + t_rot;
                if (j >= c.n_lines) break;
                const unsigned long long row = c.P + j;
                if (row < a.nl_cap) a.d_nl_pos[row] = (unsigned long long)((long long)c.tile_off + (int)s.nlist[4 + j] - kWin);
                else hdr->overflow = 1u;  // (the result says EXG_RF_FALLBACK: the caller's general path takes the batch)
            }
            return;
        }
        const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
        const LdsSrc<L> src{s, a.payload_base + c.tile_off - kWin, s.tabmap[L::kHasTabs ? c.half : 0]};
        for (uint32_t jb = 0; jb < c.n_lines; jb += kThreads) {
            const uint32_t j = jb + t_rot;
            const long long out = (long long)(c.P + j) - (long long)halo_nl;
            bool act = j < c.n_lines;
            int e1 = 0;
            if (act) {
                e1 = s.nlist[4 + j];
                act = (uint64_t)((int64_t)c.tile_off + e1 - kWin) >= a.lead && out >= 0;
                if (act && !no_store && (unsigned long long)out >= a.capacity) {
                    atomicOr(&hdr->flags, EXG_RF_CAPACITY);
                    act = false;
                }
            }
            bool qv = false, rv = false;
            if (act) {
                const uint32_t q0 = s.nlist[3 + j];  // newline before the line
                if (q0 == kNoneE) {
                    // the line begins in front of the LDS window (only the first line of a half's first pass can: the thread with t_rot == 0)
                    if constexpr (kMode == kLean) {
                        // lean scan: the any-shape run redoes this super-tile
                        tile_redo_of<false>(tile_qend, a.n_bytes)[(uint32_t)(tile_index / kVcfHalves)] = kRedoFar;
                        hdr->any_redo = 1u;
                    } else {
                        // any-shape scan: its row is k_vcf_far's
                        FarRec f;
                        f.pos[0] = s.prev32[3];
                        f.pos[1] = c.half * kTile + e1 - kWin;
                        f.pos[2] = f.pos[3] = f.pos[4] = 0;
                        f.flags = c.is_eof_tile ? 1u : 0u;
                        f.out = out;
                        far_rec_of<false>(tile_qend, a.n_bytes)[tile_index] = f;
                        hdr->any_far = 1u;
                        tile_qend[tile_index] = qend_word | kFarBit;  // (this thread stored the word above: no load — a load behind
                                                                       // the store cost a memory round trip per half of a cohort VCF)
                    }
                } else {
                    int s0 = (int)q0 + 1;
                    if (e1 > s0 && !(c.is_eof_tile && e1 == c.lim_e) && ldb(s, e1 - 1) == '\r') e1--;
#ifdef EXG_DEV_PROBE
                    if (dev_mode == 6) continue;  // (ablation: the loop, the list reads and the validity ballots only)
#endif
                    VcfRowInfo r = vcf_line(src, s0, e1, a, (unsigned long long)out, !no_store && dev_mode != 2 && dev_mode != 5);
                    if constexpr (kMode != kLean) {  // noodles builds str fields: the line must be UTF-8
                        if (!r.code && c.non_ascii && !utf8_valid_lds(s, s0, e1)) r.code = EXG_PE_INVALID_UTF8;
                    }
                    if (r.code) vcf_report(hdr, r.code, (unsigned long long)out, c.tile_off + s0 - kWin);
                    else if (r.slow_len) vcf_slow_qual(hdr, a, c.tile_off + r.slow_s - kWin, (uint32_t)r.slow_len, (unsigned long long)out, c.tile_off + s0 - kWin);
                    qv = r.qual_valid;
                    rv = r.rest_valid;
                }
            }
            if (!no_store) {
                long long out_base = (long long)(c.P + jb + ((wave - rot_w) & 3u) * 64) - (long long)halo_nl;
                unsigned long long qb = __ballot(qv), rb = __ballot(rv);
                store_validity64(a.d_qual_valid, qb, out_base, lane);
                store_validity64(a.d_formats_valid, rb, out_base, lane);
            }
        }
    }
};

// ---- general path: thread = line, bytes from global memory -----------------------------------------------
struct GlobalSrc {
    const uint8_t *p;  // d_in (16-byte aligned)
    uint64_t base;     // offset added to the (int) positions
    uint64_t payload_base;
    uint64_t limit;  // n_bytes rounded up to 16: reads past it are not allowed
    // Round 5: aligned dword / 16-byte loads + a byte shift instead of a load per byte (u32 was four byte loads with a bound check
    // each, tabs64 sixty-four: the rows k_vcf_far and k_vcf_lines parse out of global memory cost ~2 us of dependent loads each —
    // a cohort VCF has one such row per half).  An aligned block that begins below `limit` lies inside the buffer; the last
    // bytes of the buffer take the former byte path.
    __device__ __forceinline__ uint32_t b(int i) const { return p[base + (uint64_t)(int64_t)i]; }
    __device__ __forceinline__ uint32_t u32_slow(uint64_t o) const {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (o + k < limit) w |= (uint32_t)p[o + k] << (8 * k);
        return w;
    }
    __device__ __forceinline__ uint32_t u32(int i) const {
        const uint64_t o = base + (uint64_t)(int64_t)i, al = o & ~3ull;
        if (al + 8 > limit) return u32_slow(o);
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p + al);
        return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(o & 3));
    }
    __device__ __forceinline__ void u96(int i, uint32_t *w0, uint32_t *w1, uint32_t *w2) const {
        const uint64_t o = base + (uint64_t)(int64_t)i, al = o & ~3ull;
        if (al + 16 > limit) {
            *w0 = u32_slow(o), *w1 = u32_slow(o + 4), *w2 = u32_slow(o + 8);
            return;
        }
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p + al);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], sh = (uint32_t)(o & 3);
        *w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
        *w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        *w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
    }
    __device__ __forceinline__ uint4 str(int i, uint32_t len) const {
        const uint64_t o = base + (uint64_t)(int64_t)i;
        if (((o & ~3ull) + 16) > limit) return make_string_global(p, o, len, payload_base);
        uint32_t w0, w1, w2;
        u96(i, &w0, &w1, &w2);
        uint4 r;
        r.x = len;
        if (len <= EXG_INLINE_LENGTH) {  // the bytes behind the field are not the string's: zeros
            const uint32_t n0 = len < 4u ? len : 4u, n1 = len < 4u ? 0u : len - 4u < 4u ? len - 4u : 4u, n2 = len < 8u ? 0u : len - 8u;
            r.y = n0 == 4 ? w0 : w0 & ((1u << (8 * n0)) - 1u);
            r.z = n1 == 4 ? w1 : w1 & ((1u << (8 * n1)) - 1u);
            r.w = n2 == 4 ? w2 : w2 & ((1u << (8 * n2)) - 1u);
        } else {
            const uint64_t ptr = payload_base + o;
            r.y = w0;
            r.z = (uint32_t)ptr;
            r.w = (uint32_t)(ptr >> 32);
        }
        return r;
    }
    __device__ __forceinline__ bool tab_bits(int, unsigned long long *) const { return false; }
    __device__ __forceinline__ unsigned long long tabs64(int base_i) const {
        const uint64_t o = base + (uint64_t)(int64_t)base_i, al = o & ~15ull;
        if (al + 80 > limit) {
            unsigned long long bits = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) bits |= (unsigned long long)nib4(match4(u32_slow(o + 4 * q), 0x09090909u)) << (4 * q);
            return bits;
        }
        const uint4 *q = reinterpret_cast<const uint4 *>(p + al);
        const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4];  // (five loads in flight)
        const unsigned long long lo = (unsigned long long)match16(v0, 0x09090909u) | ((unsigned long long)match16(v1, 0x09090909u) << 16) |
                                      ((unsigned long long)match16(v2, 0x09090909u) << 32) | ((unsigned long long)match16(v3, 0x09090909u) << 48);
        const unsigned long long hi = match16(v4, 0x09090909u);
        const uint32_t sh = (uint32_t)(o & 15);
        return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    }
};

__global__ __launch_bounds__(256) void k_vcf_lines(VcfDev a, const uint64_t *__restrict__ nl_pos, ScanWsHeader *hdr,
                                                   const unsigned int *gate) {
    if (gate && *gate == 0) return;
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    const uint64_t halo = hdr->halo_nl;
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
    const uint64_t n_iter = (T + 63) / 64;
    const uint64_t wave_id = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t it = wave_id; it < n_iter; it += n_waves) {
        const uint64_t j = it * 64 + lane_id();
        bool act = j < T && j >= halo;
        const uint64_t out = j - halo;
        if (act && !no_store && out >= a.capacity) {
            atomicOr(&hdr->flags, EXG_RF_CAPACITY);
            act = false;
        }
        bool qv = false, rv = false;
        if (act) {
            uint64_t e1 = nl_pos[j];
            bool resolved = j > 0 || (a.flags & EXG_F_BOF);
            uint64_t s0 = j > 0 ? nl_pos[j - 1] + 1 : 0;
            if (!resolved) {
                atomicAdd(&hdr->n_unresolved, 1ull);
                atomicOr(&hdr->flags, EXG_RF_HEAD_UNRESOLVED);
            } else if (e1 - s0 > 0x7FFFFFF0ull) {
                vcf_report(hdr, EXG_PE_FIELD_TOO_LONG, out, s0);
            } else {
                if (s0 > e1) s0 = e1;
                bool virt = e1 >= a.n_bytes;
                if (!virt && e1 > s0 && a.d_in[e1 - 1] == '\r') e1--;
                const GlobalSrc src{a.d_in, s0, a.payload_base, (a.n_bytes + 15) & ~15ull};
                VcfRowInfo r = vcf_line(src, 0, (int)(e1 - s0), a, out, !no_store);
                if (!r.code && (hdr->flags & EXG_RF_NON_ASCII)) {
                    // noodles builds str fields: the line must be UTF-8
                    if (!utf8_valid_global(a.d_in, s0, e1)) r.code = EXG_PE_INVALID_UTF8;
                }
                if (r.code) vcf_report(hdr, r.code, out, s0);
                else if (r.slow_len) vcf_slow_qual(hdr, a, s0 + (uint64_t)r.slow_s, (uint32_t)r.slow_len, out, s0);
                qv = r.qual_valid;
                rv = r.rest_valid;
            }
        }
        if (!no_store) {
            long long out_base = (long long)(it * 64) - (long long)halo;
            unsigned long long qb = __ballot(qv), rb = __ballot(rv);
            store_validity64(a.d_qual_valid, qb, out_base, lane_id());
            store_validity64(a.d_formats_valid, rb, out_base, lane_id());
        }
    }
}

// ---- EXG_ALGO_FUSED_INDEX: the rows behind the scan, a thread per line, the line's HEAD through LDS -------------------------
// k_vcf_lines walks its line in global memory: every byte the number parsers and the tab search look at is a load of its own
// (an L1 / L2 round trip each, one behind the other).  Here a thread first copies the first kStageBytes of its line — nine
// 16-byte loads, all in flight at once — into a row of its own in LDS and parses from there; what lies behind the staged bytes
// (the FORMAT-and-samples remainder needs only its first bytes; an INFO longer than the stage) is read from global memory as
// before.  Rows are kStageStride dwords apart (odd: the lanes of a wavefront hit different banks at the same offset).
static constexpr int kStageBytes = 128, kStageBlocks = kStageBytes / 16 + 1, kStageStride = kStageBlocks * 4 + 1;
struct StagedSrc {
    const uint32_t *w;  // this thread's row: byte 0 = the input byte at (line start & ~15)
    uint32_t lead;      // line start & 15
    uint32_t have;      // bytes of the line (from its start) that are staged
    GlobalSrc g;
    __device__ __forceinline__ uint32_t dw(uint32_t k) const { return w[k]; }
    __device__ __forceinline__ uint32_t b(int i) const {
        if ((uint32_t)i >= have) return g.b(i);
        const uint32_t p = lead + (uint32_t)i;
        return (w[p >> 2] >> (8 * (p & 3))) & 255u;
    }
    __device__ __forceinline__ uint32_t u32(int i) const {
        if ((uint32_t)i + 4 > have || i < 0) return g.u32(i);
        const uint32_t p = lead + (uint32_t)i, k = p >> 2;
        return __builtin_amdgcn_alignbyte(w[k + 1], w[k], p & 3);
    }
    __device__ __forceinline__ void u96(int i, uint32_t *w0, uint32_t *w1, uint32_t *w2) const {
        if ((uint32_t)i + 12 > have || i < 0) {
            g.u96(i, w0, w1, w2);
            return;
        }
        const uint32_t p = lead + (uint32_t)i, k = p >> 2, sh = p & 3;
        const uint32_t d0 = w[k], d1 = w[k + 1], d2 = w[k + 2], d3 = w[k + 3];
        *w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
        *w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        *w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
    }
    __device__ __forceinline__ uint4 str(int i, uint32_t len) const {
        if ((uint32_t)i + 12 > have || i < 0) return g.str(i, len);
        uint32_t w0, w1, w2;
        u96(i, &w0, &w1, &w2);
        uint4 r;
        r.x = len;
        if (len <= EXG_INLINE_LENGTH) {
            const uint32_t n0 = len < 4u ? len : 4u, n1 = len < 4u ? 0u : len - 4u < 4u ? len - 4u : 4u, n2 = len < 8u ? 0u : len - 8u;
            r.y = n0 == 4 ? w0 : w0 & ((1u << (8 * n0)) - 1u);
            r.z = n1 == 4 ? w1 : w1 & ((1u << (8 * n1)) - 1u);
            r.w = n2 == 4 ? w2 : w2 & ((1u << (8 * n2)) - 1u);
        } else {
            const uint64_t ptr = g.payload_base + g.base + (uint64_t)(int64_t)i;
            r.y = w0;
            r.z = (uint32_t)ptr;
            r.w = (uint32_t)(ptr >> 32);
        }
        return r;
    }
    __device__ __forceinline__ bool tab_bits(int, unsigned long long *) const { return false; }
    __device__ __forceinline__ unsigned long long tabs64(int base_i) const {
        if ((uint32_t)base_i + 68 > have || base_i < 0) return g.tabs64(base_i);
        const uint32_t p = lead + (uint32_t)base_i, k = p >> 2, sh = p & 3;
        unsigned long long lo = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) lo |= (unsigned long long)nib4(match4(w[k + q], 0x09090909u)) << (4 * q);
        const unsigned long long hi = nib4(match4(w[k + 16], 0x09090909u));
        return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    }
};

__global__ __launch_bounds__(256) void k_vcf_rows(VcfDev a, const uint64_t *__restrict__ nl_pos, ScanWsHeader *hdr) {
    __shared__ uint32_t stage[256 * kStageStride];
    uint32_t *const row = stage + threadIdx.x * kStageStride;
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    const uint64_t halo = hdr->halo_nl;
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
    const uint64_t limit = (a.n_bytes + 15) & ~15ull;
    const uint64_t n_iter = (T + 63) / 64;
    const uint64_t wave_id = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t it = wave_id; it < n_iter; it += n_waves) {
        const uint64_t j = it * 64 + lane_id();
        bool act = j < T && j >= halo;
        const uint64_t out = j - halo;
        if (act && !no_store && out >= a.capacity) {
            atomicOr(&hdr->flags, EXG_RF_CAPACITY);
            act = false;
        }
        bool qv = false, rv = false;
        if (act) {
            uint64_t e1 = nl_pos[j];
            bool resolved = j > 0 || (a.flags & EXG_F_BOF);
            uint64_t s0 = j > 0 ? nl_pos[j - 1] + 1 : 0;
            if (!resolved) {
                atomicAdd(&hdr->n_unresolved, 1ull);
                atomicOr(&hdr->flags, EXG_RF_HEAD_UNRESOLVED);
            } else if (e1 - s0 > 0x7FFFFFF0ull) {
                vcf_report(hdr, EXG_PE_FIELD_TOO_LONG, out, s0);
            } else {
                if (s0 > e1) s0 = e1;
                // the line's head: the 16-byte blocks from the one that holds its first byte (a block that begins below `limit`
                // lies inside the buffer), all loads first, then the row
                const uint64_t al = s0 & ~15ull;
                uint4 blk[kStageBlocks];
                uint32_t n_blk = 0;
#pragma unroll
                for (int q = 0; q < kStageBlocks; q++) {
                    const bool in = al + 16ull * q < limit;
                    blk[q] = in ? *reinterpret_cast<const uint4 *>(a.d_in + al + 16ull * q) : make_uint4(0, 0, 0, 0);
                    n_blk += in;
                }
#pragma unroll
                for (int q = 0; q < kStageBlocks; q++) {
                    row[4 * q + 0] = blk[q].x;
                    row[4 * q + 1] = blk[q].y;
                    row[4 * q + 2] = blk[q].z;
                    row[4 * q + 3] = blk[q].w;
                }
                row[4 * kStageBlocks] = 0;
                const bool virt = e1 >= a.n_bytes;
                if (!virt && e1 > s0 && a.d_in[e1 - 1] == '\r') e1--;
                StagedSrc src;
                src.w = row;
                src.lead = (uint32_t)(s0 & 15);
                src.have = 16u * n_blk > src.lead ? 16u * n_blk - src.lead : 0u;
                src.g = GlobalSrc{a.d_in, s0, a.payload_base, limit};
                VcfRowInfo r = vcf_line(src, 0, (int)(e1 - s0), a, out, !no_store);
                if (!r.code && (hdr->flags & EXG_RF_NON_ASCII)) {
                    // noodles builds str fields: the line must be UTF-8
                    if (!utf8_valid_global(a.d_in, s0, e1)) r.code = EXG_PE_INVALID_UTF8;
                }
                if (r.code) vcf_report(hdr, r.code, out, s0);
                else if (r.slow_len) vcf_slow_qual(hdr, a, s0 + (uint64_t)r.slow_s, (uint32_t)r.slow_len, out, s0);
                qv = r.qual_valid;
                rv = r.rest_valid;
            }
        }
        if (!no_store) {
            long long out_base = (long long)(it * 64) - (long long)halo;
            unsigned long long qb = __ballot(qv), rb = __ballot(rv);
            store_validity64(a.d_qual_valid, qb, out_base, lane_id());
            store_validity64(a.d_formats_valid, rb, out_base, lane_id());
        }
    }
}

// The rows k_fused<VcfFormat> left out: one line per marked half (it begins in front of the half's window), read from global
// memory like the general path reads its lines.  Runs behind k_fused on the stream.
template <uint32_t kHalves>
__global__ __launch_bounds__(256) void k_vcf_far(VcfDev a, const unsigned int *__restrict__ tileA, const int32_t *__restrict__ tileL,
                                                 const unsigned long long *__restrict__ tile_qend, const FarRec *__restrict__ far_rec,
                                                 ScanWsHeader *hdr, uint32_t n_halves) {
    if (!hdr->any_far) return;
    constexpr uint64_t kSuper = (uint64_t)kHalves * kTile;
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n_halves; x += (uint64_t)gridDim.x * blockDim.x) {
        if (!(tile_qend[x] & kFarBit)) continue;
        const FarRec f = far_rec[x];
        int64_t p[2];
        const unsigned long long out = (unsigned long long)f.out;
        if (!far_positions<2>(f, (uint32_t)(x / kHalves), kSuper, tileA, tileL, (a.flags & EXG_F_BOF) != 0, p)) {
            atomicAdd(&hdr->n_unresolved, 1ull);  // the line begins in front of d_input[0]: the caller widens the halo
            atomicOr(&hdr->flags, EXG_RF_HEAD_UNRESOLVED);
            continue;
        }
        uint64_t s0 = (uint64_t)(p[0] + 1), e1 = (uint64_t)p[1];
        if (e1 - s0 > 0x7FFFFFF0ull) {
            vcf_report(hdr, EXG_PE_FIELD_TOO_LONG, out, s0);
            continue;
        }
        if (s0 > e1) s0 = e1;
        const bool virt = e1 >= a.n_bytes;
        if (!virt && e1 > s0 && a.d_in[e1 - 1] == '\r') e1--;
        const GlobalSrc src{a.d_in, s0, a.payload_base, (a.n_bytes + 15) & ~15ull};
        VcfRowInfo r = vcf_line(src, 0, (int)(e1 - s0), a, out, !no_store);
        if (!r.code && tiles_non_ascii(tileA, kSuper, (int64_t)s0, (int64_t)e1) && !utf8_valid_global(a.d_in, s0, e1)) r.code = EXG_PE_INVALID_UTF8;
        if (r.code) vcf_report(hdr, r.code, out, s0);
        else if (r.slow_len) vcf_slow_qual(hdr, a, s0 + (uint64_t)r.slow_s, (uint32_t)r.slow_len, out, s0);
        if (!no_store) {
            if (r.qual_valid && a.d_qual_valid) atomicOr((unsigned long long *)&a.d_qual_valid[out >> 6], 1ull << (out & 63));
            if (r.rest_valid && a.d_formats_valid) atomicOr((unsigned long long *)&a.d_formats_valid[out >> 6], 1ull << (out & 63));
        }
    }
}

// Result block.  fused != 0: positions come from tile_qend; else from nl_pos.
__global__ __launch_bounds__(256) void k_vcf_finalize(VcfDev a, ScanWsHeader *hdr,
                                                      const unsigned long long *__restrict__ tile_qend, uint32_t n_tiles,
                                                      const uint64_t *__restrict__ nl_pos, int fused, exg_scan_result *res,
                                                      const unsigned int *gate) {
    if (gate && *gate == 0) return;
    __shared__ unsigned long long s_qend;
    __shared__ int s_found;
    if (threadIdx.x == 0) {
        s_qend = 0;
        s_found = 0;
    }
    // QUAL literals left to the exact parser: one thread each (big-integer arithmetic in this kernel only)
    {
        const unsigned int n_slow = hdr->n_slow < a.slow_cap ? hdr->n_slow : (unsigned int)a.slow_cap;
        for (unsigned int k = threadIdx.x; k < n_slow && !(fused && hdr->overflow); k += blockDim.x) {
            const SlowLiteral lit = a.d_slow[k];
            uint32_t bits = 0;
            const int rc = f32_parse_exact(a.d_in + lit.off, (int)lit.len, &bits);
            const float v = __uint_as_float(bits);
            if (rc || v < 0.0f) {
                unsigned long long line = lit.off;  // the record's offset = the start of its line
                while (line > 0 && a.d_in[line - 1] != '\n') line--;
                vcf_report(hdr, EXG_PE_VCF_BAD_QUAL, lit.row, line);
            }
            else if (a.d_qual && !(a.flags & EXG_F_NO_STORE) && lit.row < a.capacity)
                a.d_qual[lit.row] = v;
        }
    }
    __syncthreads();
    if (fused && !hdr->overflow) {
        for (int64_t base = (int64_t)n_tiles - 1; base >= 0; base -= 256) {
            int64_t t = base - threadIdx.x;
            unsigned long long q = t >= 0 ? tile_qend[t] & ~kFarBit : 0;
            if (q) atomicMax(&s_qend, q);
            if (q) s_found = 1;
            __syncthreads();
            if (s_found) break;
        }
    }
    __syncthreads();
    if (threadIdx.x) return;
    if (fused && hdr->overflow) {
        exg_scan_result r = {};
        r.flags = EXG_RF_FALLBACK;
        r.error_offset = ~0ull;
        r.error_record = ~0ull;
        *res = r;
        return;
    }
    const uint64_t T = hdr->total_lines, halo = hdr->halo_nl;
    uint64_t n_owned = T > halo ? T - halo : 0;
    uint64_t last_end = s_qend;
    if (!fused) {
        uint64_t Tc = T < hdr->lines_cap ? T : hdr->lines_cap;
        last_end = Tc ? nl_pos[Tc - 1] + 1 : 0;
        if (last_end > a.n_bytes) last_end = a.n_bytes;
    }
    exg_scan_result r;
    r.n_lines = n_owned;
    r.flags = hdr->flags | (gate ? EXG_RF_FALLBACK : 0u) | (fused && hdr->any_redo ? EXG_RF_REDO : 0u);
    if (!fused && T > hdr->lines_cap) r.flags |= EXG_RF_INDEX_OVERFLOW;
    r.payload_bytes = 0;
    r.redo_tiles = fused ? hdr->n_redo : 0;
    r.error_code = 0;
    r.error_offset = ~0ull;
    r.error_record = ~0ull;
    uint64_t n_rec = (n_owned < a.capacity || (a.flags & EXG_F_NO_STORE)) ? n_owned : a.capacity;
    uint64_t consumed = last_end > a.lead ? last_end : a.lead;
    unsigned long long err = hdr->err_word;
    if (err != kNoError) {
        uint64_t rec = err >> 8;
        r.error_code = (uint32_t)(err & 0xFF);
        r.error_record = rec;
        r.error_offset = hdr->err_off;
        if (rec < n_rec) {
            n_rec = rec;
            consumed = hdr->err_off > a.lead ? hdr->err_off : a.lead;
        }
    }
    r.n_records = n_rec;
    r.consumed_bytes = n_rec ? consumed : a.lead;
    *res = r;
}

__global__ void k_init_hdr(ScanWsHeader *hdr, uint64_t lines_cap, uint32_t mode);
__global__ void k_clear_words_gated(uint64_t *w, uint64_t n, const unsigned int *gate);

static int run_vcf_general(const VcfDev &dev, uint8_t *ws, const FastqWsLayout &l, exg_scan_result *d_result,
                           hipStream_t stream, bool after_fused) {
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    const uint64_t *nl_pos = reinterpret_cast<const uint64_t *>(ws + l.off_nl_pos);
    const unsigned int *gate = after_fused ? &hdr->overflow : nullptr;
    hipLaunchKernelGGL(k_init_hdr, dim3(1), dim3(1), 0, stream, hdr, l.lines_cap, after_fused ? 1u : 0u);
    if (after_fused && !(dev.flags & EXG_F_NO_STORE)) {
        uint64_t words = (dev.capacity + 63) / 64;
        uint32_t g = (uint32_t)((words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024);
        if (words && dev.d_qual_valid)
            hipLaunchKernelGGL(k_clear_words_gated, dim3(g), dim3(256), 0, stream, dev.d_qual_valid, words, gate);
        if (words && dev.d_formats_valid)
            hipLaunchKernelGGL(k_clear_words_gated, dim3(g), dim3(256), 0, stream, dev.d_formats_valid, words, gate);
    }
    int rc = launch_line_index(dev.d_in, dev.n_bytes, dev.lead, ws, l, (dev.flags & EXG_F_EOF) ? 1 : 0, 0, stream, gate);
    if (rc) return rc;
    uint64_t est = dev.n_bytes / 32 + 256;
    uint32_t grid = (uint32_t)((est + 255) / 256 < 2048 ? (est + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_vcf_lines, dim3(grid), dim3(256), 0, stream, dev, nl_pos, hdr, gate);
    hipLaunchKernelGGL(k_vcf_finalize, dim3(1), dim3(256), 0, stream, dev, hdr, (const unsigned long long *)nullptr, 0u,
                       nl_pos, 0, d_result, gate);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

static int run_vcf_fused(const VcfDev &dev_in, uint8_t *ws, const FastqWsLayout &l, exg_scan_result *d_result,
                         hipStream_t stream, bool full, bool index = false) {
    VcfDev dev = dev_in;
    if (index) {
        dev.d_nl_pos = reinterpret_cast<uint64_t *>(ws + l.off_nl_pos);
        dev.nl_cap = l.lines_cap;
    }
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    const uint32_t kHalvesHost = full ? VcfFormat::kHalvesFull : VcfFormat::kHalves;
    const uint64_t kSuperBytes = (uint64_t)kHalvesHost * kTile;
    uint64_t n_super64 = (dev.n_bytes + kSuperBytes - 1) / kSuperBytes;
    if (n_super64 == 0) n_super64 = 1;
    if (n_super64 > 0x7FFFFFF0ull) {
        set_error("exg_vcf_scan: buffer too large for one launch");
        return EXG_E_INVALID_ARG;
    }
    uint32_t n_super = (uint32_t)n_super64;
    // descriptor block (exg_fastq_ws.hpp): u64 tileA[n] (u32 counts), u64 tileP[n], u64 tile_redo[n] (u32 marks), u64 tile_qend[n],
    // int32 tileL[n][4], FarRec[n]
    const uint64_t n = l.n_tiles_fused;
    unsigned int *tileA = reinterpret_cast<unsigned int *>(ws + l.off_tile_desc);
    unsigned long long *tileP = reinterpret_cast<unsigned long long *>(ws + l.off_tile_desc) + n;
    unsigned long long *tile_qend = tileP + 2 * n;
    int32_t *tileL = reinterpret_cast<int32_t *>(ws + l.off_tile_last4);
    FarRec *far_rec = reinterpret_cast<FarRec *>(ws + l.off_far);
    hipLaunchKernelGGL(k_init_hdr, dim3(1), dim3(1), 0, stream, hdr, l.lines_cap, 0u);
    EXG_HIP_CHECK(hipMemsetAsync(tileA, 0, (size_t)n * 24, stream));
    if (dev.lead) {
        int rc = exg_count_newlines(dev.d_in, 0, dev.lead, (uint64_t *)&hdr->halo_nl, stream);
        if (rc) return rc;
    }
    if (index) {
        hipLaunchKernelGGL((k_fused<VcfFormat, kFullIndex>), dim3(n_super + 1), dim3(kThreads), 0, stream, dev, tileA, tileP, tile_qend, hdr, n_super);
    } else if (full) {
        hipLaunchKernelGGL((k_fused<VcfFormat, kFullPrimary>), dim3(n_super + 1), dim3(kThreads), 0, stream, dev, tileA, tileP, tile_qend, hdr, n_super);
    } else {
        hipLaunchKernelGGL((k_fused<VcfFormat, kLean>), dim3(n_super + 1), dim3(kThreads), 0, stream, dev, tileA, tileP, tile_qend, hdr, n_super);
        // the super-tiles the lean scan marked, any shape (returns at once when it marked none)
        hipLaunchKernelGGL((k_fused<VcfFormat, kFullRedo>), dim3(n_super < 1024 ? n_super : 1024), dim3(kThreads), 0, stream, dev, tileA, tileP,
                           tile_qend, hdr, n_super);
    }
    if (index) {
        // the rows, a thread per line (the any-shape scan left hdr->total_lines, halo_nl and the non-ASCII flag like the line index does)
        VcfDev rows = dev;
        rows.d_nl_pos = nullptr;
        const uint64_t est = dev.n_bytes / 256 + 256;
        const uint32_t grid = (uint32_t)((est + 255) / 256 < 4096 ? (est + 255) / 256 : 4096);
        hipLaunchKernelGGL(k_vcf_rows, dim3(grid), dim3(256), 0, stream, rows, (const uint64_t *)dev.d_nl_pos, hdr);
    } else {   // the rows of lines that begin in front of their half's window (returns at once when there is none)
        const uint32_t n_halves = n_super * kHalvesHost;
        const uint32_t grid = (n_halves + 255) / 256 < 4096 ? (n_halves + 255) / 256 : 4096;
        if (full)
            hipLaunchKernelGGL(k_vcf_far<VcfFormat::kHalvesFull>, dim3(grid), dim3(256), 0, stream, dev, tileA, tileL, tile_qend, far_rec, hdr, n_halves);
        else
            hipLaunchKernelGGL(k_vcf_far<VcfFormat::kHalves>, dim3(grid), dim3(256), 0, stream, dev, tileA, tileL, tile_qend, far_rec, hdr, n_halves);
    }
    hipLaunchKernelGGL(k_vcf_finalize, dim3(1), dim3(256), 0, stream, dev, hdr, tile_qend, n_super * kHalvesHost,
                       (const uint64_t *)nullptr, 1, d_result, (const unsigned int *)nullptr);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

}  // namespace exg

using namespace exg;

extern "C" int exg_vcf_scan(const exg_vcf_scan_args *a) {
    if (!a || !a->d_result || !a->d_workspace || ((uintptr_t)a->d_workspace & 255) || (a->n_bytes && !a->d_input) ||
        ((uintptr_t)a->d_input & 15) || a->lead > a->n_bytes) {
        set_error("exg_vcf_scan: bad arguments (null pointer, unaligned input or workspace, or lead > n_bytes)");
        return EXG_E_INVALID_ARG;
    }
    if (a->flags & ~EXG_F_ALL) {
        set_error("exg_vcf_scan: unknown flag bits 0x%x", a->flags & ~EXG_F_ALL);
        return EXG_E_INVALID_ARG;
    }
    FastqWsLayout l = fastq_ws_layout(a->n_bytes, a->workspace_bytes);
    if (a->workspace_bytes < fastq_ws_layout(a->n_bytes, 0).off_nl_pos + 64) {
        set_error("exg_vcf_scan: workspace too small");
        return EXG_E_INVALID_ARG;
    }
    VcfDev dev;
    dev.d_in = (const uint8_t *)a->d_input;
    dev.n_bytes = a->n_bytes;
    dev.lead = a->lead;
    dev.first_line_index = 0;
    dev.payload_base = a->payload_base;
    dev.flags = a->flags;
#ifdef EXG_DEV_PROBE
    if (const char *e = getenv("EXG_VCF_DEV_MODE")) dev.flags |= ((uint32_t)atoi(e) & 15u) << 8;  // development builds only: the core's ablations (no emission / no stores)
#endif
    dev.pad = 0;
    for (int k = 0; k < 9; k++) dev.d_fields[k] = a->d_fields[k];
    dev.d_pos = a->d_pos;
    dev.d_qual = a->d_qual;
    dev.d_qual_valid = a->d_qual_validity;
    dev.d_formats_valid = a->d_formats_validity;
    dev.capacity = a->capacity_records;
    hipStream_t stream = (hipStream_t)a->stream;
    uint8_t *ws = (uint8_t *)a->d_workspace;
    dev.d_slow = reinterpret_cast<SlowLiteral *>(ws + l.off_slow);
    dev.slow_cap = l.slow_cap;
    dev.d_nl_pos = nullptr;
    dev.nl_cap = 0;
    if (a->capacity_records && !(a->flags & EXG_F_NO_STORE)) {
        size_t vb = (size_t)((a->capacity_records + 63) / 64) * 8;
        if (dev.d_qual_valid) EXG_HIP_CHECK(hipMemsetAsync(dev.d_qual_valid, 0, vb, stream));
        if (dev.d_formats_valid) EXG_HIP_CHECK(hipMemsetAsync(dev.d_formats_valid, 0, vb, stream));
    }
    switch (a->algo) {
        case EXG_ALGO_MULTIPASS:
            return run_vcf_general(dev, ws, l, a->d_result, stream, false);
        case EXG_ALGO_FUSED:
            return run_vcf_fused(dev, ws, l, a->d_result, stream, false);
        case EXG_ALGO_FUSED_FULL:
            return run_vcf_fused(dev, ws, l, a->d_result, stream, true);
        case EXG_ALGO_FUSED_INDEX:
            return run_vcf_fused(dev, ws, l, a->d_result, stream, true, true);
        case EXG_ALGO_AUTO: {
            int rc = run_vcf_fused(dev, ws, l, a->d_result, stream, false);
            if (rc) return rc;
            return run_vcf_general(dev, ws, l, a->d_result, stream, true);
        }
        default:
            set_error("exg_vcf_scan: unknown algo %u", a->algo);
            return EXG_E_INVALID_ARG;
    }
}
