// exg_fastq_fused.hip — FASTQ record scan in ONE pass over the input (EXG_ALGO_FUSED).
//
// Replaces, for FASTQ, everything the reference does per batch on one CPU core: noodles-fastq
// 0.8.0 `read_record` (memchr per line, '@' / '+' checks, first-space split), exon 0.2.6
// FASTQArrayBuilder (one memcpy + offset push per field; empty description => NULL), both
// reached through rust/src/arrow_reader.rs:116-153, and DuckDB's ArrowToDuckDB string_t
// construction called at exon/src/exon/arrow_table_function/module.cpp:289.
//
// Design (MI355X / gfx950, wave64, HBM-bound byte work, no MFMA):
//   * one 256-thread workgroup per 16 KiB tile, tile = blockIdx.x.  No ticket counter: a single
//     hot atomic word caps the chip at ~90 tiles/us (measured: 3 such words held the first version
//     at 0.3 TB/s).  Forward progress does not depend on dispatch order either: a look-back that
//     waits longer than ~40 us for a predecessor's descriptor counts that tile's newlines itself
//     ("helping"), so a block never blocks on an undispatched block;
//   * coalesced 16 B/lane global loads -> LDS (each input byte leaves HBM once); the 1 KiB that
//     precedes the tile is staged too, so the record straddling the tile's left edge is resolved
//     from LDS (a record larger than that window raises `overflow` and the general multipass
//     kernels redo the buffer);
//   * LDS rows are padded 64 -> 80 bytes so that every thread can read ITS contiguous 64 bytes
//     with conflict-free ds_read_b128 (lane stride 20 banks covers all 64 banks in any 16 lanes);
//   * SWAR '\n' match -> 64-bit mask per thread -> popcount -> wave shuffle scan + 4 wave totals;
//   * tile newline count published in a single 64-bit descriptor {status:2, count:62};
//     wave-parallel decoupled look-back (64 predecessors per probe, relaxed agent-scope atomics —
//     the descriptor word IS the payload, so no fence is needed) gives the global line index,
//     hence the exact 4-line phase: '@' is also a quality character, so phase is never guessed;
//   * one thread per record that ENDS in the tile: 5 newline positions -> 4 field slices ->
//     '@' / '+' validation, CR strip, first-space split, four 16-byte string_t built from LDS
//     dwords with v_alignbyte, written as coalesced 16 B/lane stores per column; description
//     validity by one wave ballot (<= 2 atomic ORs per wave).
#include "exg_fastq.hpp"

namespace exg {

static constexpr int kTile = kFusedTileBytes;  // 16384
static constexpr int kWin = kFusedWindow;      // 1024
static constexpr int kThreads = 256;
static constexpr int kPT = kTile / kThreads;   // 64 bytes per thread
static constexpr int kRow = kPT + 16;          // padded row
static constexpr int kExt = kWin + kTile;      // extended tile: window + tile
static constexpr int kLdsBytes = kExt / kPT * kRow + 64;
static constexpr int kNlCap = 1024;            // newline positions kept per tile
static constexpr int kNone = -0x40000000;

static constexpr unsigned long long kStatusA = 1ull << 62;  // tile aggregate
static constexpr unsigned long long kStatusP = 2ull << 62;  // inclusive prefix
static constexpr unsigned long long kValueMask = (1ull << 62) - 1;

// extended offset e = p + kWin (p = tile-relative byte position, may be negative) -> LDS byte address
__device__ __forceinline__ uint32_t lds_phys(uint32_t e) { return e + ((e >> 6) << 4); }

struct FusedLds {
    uint8_t bytes[kLdsBytes];
    int nlist[4 + kNlCap + 4];  // [0..3] = 4 newlines before the tile (oldest first), then the tile's
    uint32_t wave_tot[4];
    uint32_t wwave_tot;
    unsigned long long prefix;  // '\n' in the buffer before this tile
    uint32_t tile;
    uint32_t hi_or[5];
};

__device__ __forceinline__ uint32_t ldw(const FusedLds &s, uint32_t e_aligned) {
    return *reinterpret_cast<const uint32_t *>(s.bytes + lds_phys(e_aligned));
}
__device__ __forceinline__ uint32_t ldb(const FusedLds &s, int p) { return s.bytes[lds_phys((uint32_t)(p + kWin))]; }
// 4 bytes at tile-relative position p (any alignment)
__device__ __forceinline__ uint32_t ldu32(const FusedLds &s, int p) {
    uint32_t e = (uint32_t)(p + kWin);
    uint32_t a = e & ~3u;
    uint32_t lo = ldw(s, a), hi = ldw(s, a + 4);
    return __builtin_amdgcn_alignbyte(hi, lo, e & 3u);
}

// duckdb::string_t of the field [p, p+len) of this tile
__device__ __forceinline__ uint4 make_string_lds(const FusedLds &s, int p, uint32_t len, uint64_t ptr_of_p0,
                                                  bool valid) {
    uint4 r = {0, 0, 0, 0};
    if (!valid) return r;
    r.x = len;
    uint32_t w0 = ldu32(s, p);
    if (len <= EXG_INLINE_LENGTH) {
        uint32_t w1 = ldu32(s, p + 4), w2 = ldu32(s, p + 8);
        // zero the bytes at and after len
        uint32_t m0 = len >= 4 ? 0xFFFFFFFFu : ((1u << (8 * len)) - 1u);
        uint32_t l1 = len > 4 ? len - 4 : 0, l2 = len > 8 ? len - 8 : 0;
        uint32_t m1 = l1 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * l1)) - 1u);
        uint32_t m2 = l2 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * l2)) - 1u);
        r.y = w0 & m0;
        r.z = w1 & m1;
        r.w = w2 & m2;
    } else {
        uint64_t ptr = ptr_of_p0 + (uint64_t)(int64_t)p;
        r.y = w0;
        r.z = (uint32_t)ptr;
        r.w = (uint32_t)(ptr >> 32);
    }
    return r;
}

__device__ bool utf8_valid_lds(const FusedLds &s, int b, int e) {
    int i = b;
    while (i < e) {
        uint32_t c = ldb(s, i);
        if (c < 0x80) {
            i++;
            continue;
        }
        if (c >= 0xC2 && c <= 0xDF) {
            if (i + 1 >= e || (ldb(s, i + 1) & 0xC0) != 0x80) return false;
            i += 2;
        } else if (c >= 0xE0 && c <= 0xEF) {
            if (i + 2 >= e) return false;
            uint32_t c1 = ldb(s, i + 1), c2 = ldb(s, i + 2);
            uint32_t lo = c == 0xE0 ? 0xA0 : 0x80, hi = c == 0xED ? 0x9F : 0xBF;
            if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80) return false;
            i += 3;
        } else if (c >= 0xF0 && c <= 0xF4) {
            if (i + 3 >= e) return false;
            uint32_t c1 = ldb(s, i + 1), c2 = ldb(s, i + 2), c3 = ldb(s, i + 3);
            uint32_t lo = c == 0xF0 ? 0x90 : 0x80, hi = c == 0xF4 ? 0x8F : 0xBF;
            if (c1 < lo || c1 > hi || (c2 & 0xC0) != 0x80 || (c3 & 0xC0) != 0x80) return false;
            i += 4;
        } else {
            return false;
        }
    }
    return true;
}

// '\n' count of tile t, by one wave, straight from global memory (look-back helping path)
__device__ unsigned long long help_count_tile(const uint8_t *__restrict__ d_in, uint64_t n_bytes, uint32_t t,
                                              uint32_t lane) {
    uint64_t tile_off = (uint64_t)t * kTile;
    uint32_t cnt = 0;
    for (int j = 0; j < kTile / 1024; j++) {
        uint64_t off = tile_off + (uint64_t)(j * 64 + lane) * 16;
        if (off < n_bytes) {
            uint4 q = *reinterpret_cast<const uint4 *>(d_in + off);
            uint32_t mm = match16(q, 0x0A0A0A0Au);
            if (off + 16 > n_bytes) mm &= (1u << (uint32_t)(n_bytes - off)) - 1u;
            cnt += __popc(mm);
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    return (unsigned long long)__shfl(cnt, 0, 64);
}

__global__ __launch_bounds__(kThreads) void k_fastq_fused(FastqDev a, unsigned long long *__restrict__ desc,
                                                          unsigned long long *__restrict__ tile_qend,
                                                          ScanWsHeader *hdr, uint32_t n_tiles) {
    __shared__ __attribute__((aligned(16))) FusedLds s;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;

    const uint32_t tile = blockIdx.x;
    const uint64_t tile_off = (uint64_t)tile * kTile;
    const uint8_t *__restrict__ d_in = a.d_in;
    const uint64_t n_pad = (a.n_bytes + 15) & ~15ull;

    // ---- stage window + tile in LDS (coalesced 16 B per lane) -------------------------------
    uint4 v[kPT / 16];
#pragma unroll
    for (int j = 0; j < kPT / 16; j++) {
        uint64_t off = tile_off + (uint64_t)(j * kThreads + tid) * 16;
        v[j] = off < n_pad ? *reinterpret_cast<const uint4 *>(d_in + off) : make_uint4(0, 0, 0, 0);
    }
    uint4 wv = make_uint4(0, 0, 0, 0);
    const int64_t woff = (int64_t)tile_off - kWin + (int64_t)tid * 16;  // wave 0 only
    if (tid < 64 && woff >= 0) wv = *reinterpret_cast<const uint4 *>(d_in + woff);
#pragma unroll
    for (int j = 0; j < kPT / 16; j++)
        *reinterpret_cast<uint4 *>(s.bytes + lds_phys((uint32_t)(kWin + (j * kThreads + tid) * 16))) = v[j];
    if (tid < 64) *reinterpret_cast<uint4 *>(s.bytes + lds_phys(tid * 16)) = wv;

    // ---- window: the 4 newlines that precede the tile (wave 0) -----------------------------
    // A line starts at d_input[0] when EXG_F_BOF: model it as a newline at offset -1.
    const bool win_at_start = tile_off <= (uint64_t)kWin;  // the window reaches d_input[0]
    uint32_t hi = 0;
    if (wave == 0) {
        if (lane < 4) s.nlist[lane] = kNone;
        uint32_t wm = woff >= 0 ? match16(wv, 0x0A0A0A0Au) : 0u;
        hi |= (wv.x | wv.y | wv.z | wv.w) & 0x80808080u;
        uint32_t wc = __popc(wm);
        uint32_t incl = wave_incl_sum(wc);
        uint32_t W = __shfl(incl, 63, 64);
        bool bof = win_at_start && (a.flags & EXG_F_BOF);
        // rank r newline (0-based, oldest first) goes to slot 4 - (W - r) if that is >= 0
        uint32_t r = incl - wc;
        while (wm) {
            uint32_t b = __ffs(wm) - 1;
            wm &= wm - 1;
            int slot = 4 - (int)(W - r);
            if (slot >= 0) s.nlist[slot] = (int)(lane * 16 + b) - kWin;
            r++;
        }
        if (bof && W < 4 && lane == 0) s.nlist[4 - (int)W - 1] = -(int)tile_off - 1;
    }
    __syncthreads();

    // ---- classify: each thread owns 64 contiguous bytes ------------------------------------
    unsigned long long m = 0;
    {
        const uint8_t *row = s.bytes + lds_phys((uint32_t)(kWin + tid * kPT));
#pragma unroll
        for (int c = 0; c < kPT / 16; c++) {
            uint4 q = *reinterpret_cast<const uint4 *>(row + c * 16);
            m |= (unsigned long long)match16(q, 0x0A0A0A0Au) << (16 * c);
            hi |= (q.x | q.y | q.z | q.w) & 0x80808080u;
        }
    }
    // bytes at or beyond n_bytes are not part of the input
    const int64_t lim64 = (int64_t)a.n_bytes - (int64_t)tile_off;
    const int lim = lim64 < kTile ? (int)lim64 : kTile;  // tile-relative end of input
    {
        int rem = lim - (int)tid * kPT;
        if (rem < 64) m &= rem <= 0 ? 0ull : ((1ull << rem) - 1ull);
    }
    uint32_t cnt = (uint32_t)__popcll(m);
    uint32_t incl = wave_incl_sum(cnt);
    if (lane == 63) s.wave_tot[wave] = incl;
    uint32_t any_hi = __any(hi != 0);
    if (lane == 0) s.hi_or[wave] = any_hi;
    __syncthreads();
    uint32_t wave_off = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) wave_off += k < wave ? s.wave_tot[k] : 0;
    uint32_t n_nl = s.wave_tot[0] + s.wave_tot[1] + s.wave_tot[2] + s.wave_tot[3];
    const bool tile_non_ascii = (s.hi_or[0] | s.hi_or[1] | s.hi_or[2] | s.hi_or[3]) != 0;

    // ---- publish the aggregate, then look back (wave 0) while the others write the list -----
    if (wave == 0) {
        unsigned long long excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&desc[0], kStatusP | n_nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&desc[tile], kStatusA | n_nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t base = (int64_t)tile - 1;
            for (;;) {
                int64_t idx = base - lane;
                unsigned long long d;
                unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
                for (;;) {
                    d = idx >= 0 ? __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStatusP;
                    unsigned long long missing = __ballot((d >> 62) == 0);
                    if (!missing) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 4000) {
                        // ~40 us without a descriptor: that block may not have been dispatched.
                        // Count its tile ourselves (any dispatch order makes progress).
                        while (missing) {
                            int l = __ffsll((long long)missing) - 1;
                            missing &= missing - 1;
                            uint32_t ht = (uint32_t)(base - l);
                            unsigned long long c = help_count_tile(d_in, a.n_bytes, ht, lane);
                            if ((int)lane == l) d = kStatusA | c;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                unsigned long long pm = __ballot((d >> 62) == 2);
                unsigned long long val = d & kValueMask;
                if (pm) {
                    int first = __ffsll((long long)pm) - 1;  // nearest predecessor holding an inclusive prefix
                    if ((int)lane > first) val = 0;
                }
                for (int o = 32; o > 0; o >>= 1) val += __shfl_down(val, o, 64);
                excl += __shfl(val, 0, 64);
                if (pm) break;
                base -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&desc[tile], kStatusP | (excl + n_nl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s.prefix = excl;
    }
    {
        uint32_t r = wave_off + incl - cnt;
        unsigned long long mm = m;
        while (mm) {
            int b = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            if (r < (uint32_t)kNlCap) s.nlist[4 + r] = (int)(tid * kPT) + b;
            r++;
        }
    }
    __syncthreads();

    // ---- per-record emission ----------------------------------------------------------------
    const unsigned long long halo_nl = hdr->halo_nl;
    const unsigned long long P0 = a.first_line_index - halo_nl;  // line index of d_input[0]
    const unsigned long long P = s.prefix;
    const bool last_tile = tile + 1 == n_tiles;
    uint32_t n_lines = n_nl;
    if (last_tile && (a.flags & EXG_F_EOF)) {
        // noodles EOF rules: an unterminated last line is a line; a record with its '+' line but no
        // quality line gets an empty one (read_line returns 0 bytes at EOF without error).
        bool unterminated = a.n_bytes > 0 && lim > 0 && ldb(s, lim - 1) != '\n';
        if (unterminated) {
            if (n_lines < (uint32_t)kNlCap && tid == 0) s.nlist[4 + n_lines] = lim;
            n_lines++;
        }
        if (((P0 + P + n_lines) & 3) == 3) {
            if (n_lines < (uint32_t)kNlCap && tid == 0) s.nlist[4 + n_lines] = lim;
            n_lines++;
        }
        __syncthreads();
        if (tid == 0) {
            hdr->total_nl = P + n_nl;
            hdr->total_lines = P + n_lines;
        }
    } else if (last_tile && tid == 0) {
        hdr->total_nl = P + n_nl;
        hdr->total_lines = P + n_lines;
    }
    if (n_lines > (uint32_t)kNlCap) {
        if (tid == 0) atomicOr(&hdr->overflow, 1u);
        return;
    }
    if (tile_non_ascii && tid == 0) atomicOr(&hdr->flags, EXG_RF_NON_ASCII);

    const uint32_t i0 = (uint32_t)((3 - P0) & 3);                    // first quality line of the buffer
    const uint32_t i_first = (uint32_t)((3 - (P0 + P)) & 3);         // first quality line of the tile
    const unsigned long long q_before = P > i0 ? (P - i0 + 3) / 4 : 0;  // quality lines before the tile
    const unsigned long long n_hc = halo_nl > i0 ? (halo_nl - i0 + 3) / 4 : 0;
    const uint32_t n_rec = n_lines > i_first ? (n_lines - i_first + 3) / 4 : 0;
    const uint64_t ptr_of_p0 = a.payload_base + tile_off;

    if (tid == 0) {  // offset just past the last quality line that ends in this tile (0: none)
        long long e = 0;
        if (n_rec) {
            e = (long long)tile_off + s.nlist[4 + i_first + 4 * (n_rec - 1)] + 1;
            if ((unsigned long long)e > a.n_bytes) e = (long long)a.n_bytes;
        }
        tile_qend[tile] = (unsigned long long)e;
    }
    for (uint32_t jb = 0; jb < n_rec; jb += kThreads) {  // one pass unless > 256 records end here
        uint32_t j = jb + tid;
        bool desc_valid = false;
        bool act = j < n_rec;
        long long out = (long long)(q_before + j) - (long long)n_hc;
        if (act) {
            int i = (int)(i_first + 4 * j);
            int p4 = s.nlist[4 + i];
            bool owned = (uint64_t)((int64_t)tile_off + p4) >= a.lead && out >= 0;
            if (owned && (unsigned long long)out >= a.capacity) {
                atomicOr(&hdr->flags, EXG_RF_CAPACITY);
                owned = false;
            }
            act = owned;
            if (owned) {
                int p0 = s.nlist[i], p1 = s.nlist[i + 1], p2 = s.nlist[i + 2], p3 = s.nlist[i + 3];
                if (p0 == kNone) {
                    // the record starts before the window
                    uint4 z = {0, 0, 0, 0};
                    if (win_at_start) {
                        atomicAdd(&hdr->n_unresolved, 1ull);
                        atomicOr(&hdr->flags, EXG_RF_HEAD_UNRESOLVED);
                        reinterpret_cast<uint4 *>(a.d_name)[out] = z;
                        reinterpret_cast<uint4 *>(a.d_desc)[out] = z;
                        reinterpret_cast<uint4 *>(a.d_seq)[out] = z;
                        reinterpret_cast<uint4 *>(a.d_qual)[out] = z;
                    } else {
                        atomicOr(&hdr->overflow, 1u);
                    }
                } else {
                    // line k = [pk + 1, pk+1); virtual EOF terminators sit at `lim`
                    int s0 = p0 + 1, e0 = p1, s1 = p1 + 1, e1 = p2, s2 = p2 + 1, e2 = p3, s3 = p3 + 1, e3 = p4;
                    if (s1 > e1) s1 = e1;
                    if (s2 > e2) s2 = e2;
                    if (s3 > e3) s3 = e3;
                    const bool is_eof_tile = last_tile && (a.flags & EXG_F_EOF);
                    bool name_ok = s0 < e0 && ldb(s, s0) == '@';
                    bool plus_ok = s2 < e2 && ldb(s, s2) == '+';
                    // a CR is stripped only in front of a real '\n' (virtual EOF terminators sit at lim)
                    if (e0 > s0 && !(is_eof_tile && e0 == lim) && ldb(s, e0 - 1) == '\r') e0--;
                    if (e1 > s1 && !(is_eof_tile && e1 == lim) && ldb(s, e1 - 1) == '\r') e1--;
                    if (e3 > s3 && !(is_eof_tile && e3 == lim) && ldb(s, e3 - 1) == '\r') e3--;
                    // first ' ' of the name line
                    int ns = s0 + 1 < e0 ? s0 + 1 : e0;
                    int sp = e0;
                    {
                        uint32_t eb = (uint32_t)(ns + kWin), ee = (uint32_t)(e0 + kWin);
                        for (uint32_t aa = eb & ~3u; aa < ee; aa += 4) {
                            uint32_t mm = match4(ldw(s, aa), 0x20202020u);
                            if (aa < eb) mm &= 0xFFFFFFFFu << (8 * (eb - aa));
                            if (mm) {
                                uint32_t pos = aa + ((__ffs(mm) - 1) >> 3);
                                if (pos < ee) sp = (int)pos - kWin;
                                break;
                            }
                        }
                    }
                    int ds = sp < e0 ? sp + 1 : e0;
                    uint32_t code = 0;
                    if (!name_ok)
                        code = EXG_PE_FASTQ_NAME_PREFIX;
                    else if (!plus_ok)
                        code = EXG_PE_FASTQ_PLUS_PREFIX;
                    else if (tile_non_ascii &&
                             !(utf8_valid_lds(s, ns, sp) && utf8_valid_lds(s, ds, e0) && utf8_valid_lds(s, s1, e1) &&
                               utf8_valid_lds(s, s3, e3)))
                        code = EXG_PE_INVALID_UTF8;
                    if (code) {
                        atomicMin(&hdr->err_word, ((unsigned long long)out << 8) | code);
                        atomicMin(&hdr->err_off, (unsigned long long)((int64_t)tile_off + s0));
                    }
                    desc_valid = e0 > ds;
                    reinterpret_cast<uint4 *>(a.d_name)[out] = make_string_lds(s, ns, (uint32_t)(sp - ns), ptr_of_p0, true);
                    reinterpret_cast<uint4 *>(a.d_desc)[out] =
                        make_string_lds(s, ds, (uint32_t)(e0 - ds), ptr_of_p0, desc_valid);
                    reinterpret_cast<uint4 *>(a.d_seq)[out] = make_string_lds(s, s1, (uint32_t)(e1 - s1), ptr_of_p0, true);
                    reinterpret_cast<uint4 *>(a.d_qual)[out] = make_string_lds(s, s3, (uint32_t)(e3 - s3), ptr_of_p0, true);
                }
            }
        }
        // description validity bits of this wave's 64 consecutive records
        unsigned long long b = __ballot(desc_valid);
        if (b) {
            long long out_base = (long long)(q_before + jb + wave * 64) - (long long)n_hc;
            if (out_base < 0) {
                b >>= (unsigned long long)(-out_base);
                out_base = 0;
            }
            if (lane == 0 && b) {
                uint32_t sh = (uint32_t)(out_base & 63);
                unsigned long long lo = b << sh, hi2 = sh ? b >> (64 - sh) : 0;
                if (lo) atomicOr((unsigned long long *)&a.d_desc_valid[out_base >> 6], lo);
                if (hi2) atomicOr((unsigned long long *)&a.d_desc_valid[(out_base >> 6) + 1], hi2);
            }
        }
    }
}

// Runs after k_fastq_fused on the same stream: folds the header into the 64-byte result.
__global__ __launch_bounds__(256) void k_fastq_finalize_fused(FastqDev a, ScanWsHeader *hdr,
                                                                const unsigned long long *__restrict__ tile_qend,
                                                                uint32_t n_tiles, exg_scan_result *res) {
    // last tile (searching backwards) in which a quality line ends
    __shared__ unsigned long long s_qend;
    __shared__ int s_found;
    if (threadIdx.x == 0) {
        s_qend = 0;
        s_found = 0;
    }
    __syncthreads();
    if (!hdr->overflow) {
        for (int64_t base = (int64_t)n_tiles - 1; base >= 0; base -= 256) {
            int64_t t = base - threadIdx.x;
            unsigned long long q = t >= 0 ? tile_qend[t] : 0;
            if (q) atomicMax(&s_qend, q);
            if (q) s_found = 1;
            __syncthreads();
            if (s_found) break;
        }
    }
    __syncthreads();
    const unsigned long long last_qend = s_qend;
    if (threadIdx.x) return;
    if (hdr->overflow) {  // the general kernels that follow on the stream overwrite this
        exg_scan_result r = {};
        r.flags = EXG_RF_FALLBACK;
        r.error_offset = ~0ull;
        r.error_record = ~0ull;
        *res = r;
        return;
    }
    uint64_t halo_nl = hdr->halo_nl, T = hdr->total_lines;
    uint64_t p0 = a.first_line_index - halo_nl;
    uint64_t i0 = (3 - p0) & 3;
    uint64_t n_cand = T > i0 ? (T - i0 + 3) / 4 : 0;
    uint64_t n_hc = halo_nl > i0 ? (halo_nl - i0 + 3) / 4 : 0;
    uint64_t n_owned = n_cand - (n_hc < n_cand ? n_hc : n_cand);
    unsigned long long err = hdr->err_word, err_off = hdr->err_off;
    uint64_t consumed = last_qend > a.lead ? last_qend : a.lead;
    if ((a.flags & EXG_F_EOF) && ((p0 + T) & 3) != 0 && T > halo_nl) {
        // the truncated record starts where the last complete one ended; the reader checks its '@'
        // before it can run out of lines
        uint64_t start = last_qend;
        uint32_t code = (start < a.n_bytes && a.d_in[start] == '@') ? EXG_PE_UNEXPECTED_EOF : EXG_PE_FASTQ_NAME_PREFIX;
        unsigned long long w = ((unsigned long long)n_owned << 8) | code;
        if (w < err) {
            err = w;
            err_off = start;
        }
    }
    exg_scan_result r;
    r.n_lines = T - halo_nl;
    r.flags = hdr->flags;
    r.payload_bytes = 0;
    r.reserved = 0;
    r.error_code = 0;
    r.error_offset = ~0ull;
    r.error_record = ~0ull;
    uint64_t n_rec = n_owned < a.capacity ? n_owned : a.capacity;
    if (err != kNoError) {
        uint64_t rec = err >> 8;
        r.error_code = (uint32_t)(err & 0xFF);
        r.error_record = rec;
        r.error_offset = err_off;
        if (rec < n_rec) {
            n_rec = rec;
            consumed = err_off > a.lead ? err_off : a.lead;
        }
    }
    r.n_records = n_rec;
    r.consumed_bytes = n_rec ? consumed : a.lead;
    *res = r;
}

int run_fastq_fused(const exg_fastq_scan_args *args, const FastqDev &dev, uint8_t *ws, const FastqWsLayout &l,
                    hipStream_t stream) {
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    unsigned long long *desc = reinterpret_cast<unsigned long long *>(ws + l.off_tile_desc);
    uint64_t n_tiles64 = (dev.n_bytes + kTile - 1) / kTile;
    if (n_tiles64 == 0) n_tiles64 = 1;
    if (n_tiles64 > 0x7FFFFFFFull) {
        set_error("exg_fastq_scan: buffer too large for one launch (%llu tiles)", (unsigned long long)n_tiles64);
        return EXG_E_INVALID_ARG;
    }
    uint32_t n_tiles = (uint32_t)n_tiles64;
    hipLaunchKernelGGL(k_init_hdr, dim3(1), dim3(1), 0, stream, hdr, l.lines_cap, 0u);
    EXG_HIP_CHECK(hipMemsetAsync(desc, 0, (size_t)n_tiles * 8, stream));
    if (dev.lead) {
        int rc = exg_count_newlines(dev.d_in, 0, dev.lead, (uint64_t *)&hdr->halo_nl, stream);
        if (rc) return rc;
    }
    unsigned long long *tile_qend = desc + l.n_tiles_fused;
    hipLaunchKernelGGL(k_fastq_fused, dim3(n_tiles), dim3(kThreads), 0, stream, dev, desc, tile_qend, hdr, n_tiles);
    hipLaunchKernelGGL(k_fastq_finalize_fused, dim3(1), dim3(256), 0, stream, dev, hdr, tile_qend, n_tiles,
                       args->d_result);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

}  // namespace exg
