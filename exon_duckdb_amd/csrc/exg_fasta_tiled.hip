// exg_fasta_tiled.hip — FASTA record scan in ONE pass over 32 KiB super-tiles.
//
// Same semantics as exg_fasta.hip (noodles-fasta 0.27.0 read_definition / read_sequence as driven by
// exon 0.2.6; reached from rust/src/arrow_reader.rs:116-153): a line whose first byte is '>' defines a
// record (id to the first ASCII whitespace, the rest trimmed = description); the sequence is every
// following line with LF (and a CR before it) removed, concatenated in a compacted payload buffer.
//
// What a run of bytes needs from the bytes in front of it is tiny — records before it, sequence bytes before it
// and one bit (does it begin inside a definition line) — so no line index is built:
//   k_fa_fused         a workgroup per super-tile, a wave per 8 KiB of it (eight 1 KiB rows kept in registers): newline /
//                      '>' / CR classification, the line state carried from row to row by ballots, the super-tile's aggregate
//                      published; one scanner wave (block 0) turns the aggregates into exclusive prefixes in order; the
//                      waves then compact their sequence bytes through LDS into the payload and record, for every '>' at a
//                      line start, the record's payload offset and the offset of its definition line
//   k_fa_tile_defs     id / description string_t of every record (thread = record, the line parsed from an LDS copy)
//   k_fa_tile_strings  sequence string_t (length = next record's payload offset - own)
// Traffic: 1 read + 1 write of the file (the first tiled form read it twice: a count pass, a scan, an emit pass; the
// multipass form: ~5 reads + 3 writes of it plus 40 B per line of index).  Algorithmic bytes: file read once + sequence
// bytes written once.
#include <stdlib.h>

#include "exg_fasta.hpp"

namespace exg {

namespace {

static constexpr uint32_t kThreads = 256;  // four waves

struct TileArrays {
    uint64_t *rec_start;       // [n_rec + 1] payload offset of every record's sequence
    uint64_t *rec_def_off;     // [n_rec] input offset of every record's '>'
    uint64_t rec_cap;          // records the two arrays can hold (more is reported as EXG_RF_INDEX_OVERFLOW)
    uint64_t *totals;          // [0] records, [1] sequence bytes of the whole input (written by the scanner)
    // what a super-tile publishes and what the scanner answers
    unsigned long long *f_agg;   // [n_super] packed aggregate, bit 63 = published
    unsigned int *f_nl;          // [n_super] its newline count, bit 31 = published
    unsigned long long *f_rec;   // [n_super] records in front of it, bit 63 = answered
    unsigned long long *f_pay;   // [n_super] sequence bytes in front of it, bit 62 = it begins inside a definition line, bit 63 = answered
};

// ---- the 16 bytes of one lane (a wave instruction = 1 KiB, coalesced) -----------------------------------------------
// Everything hangs off the NEWLINES of the chunk (one in four chunks has any): the line after a newline is a
// definition line iff the byte after it is '>', a CR counts only right in front of a newline.  Only the
// newline mask is a full SWAR match; the rest looks at single bytes next to the (rare) newline bits.
struct Chunk {
    uint32_t nl;           // newline bits
    uint32_t def_after;    // newline bits whose next line is a definition line
    uint32_t pay_known;    // sequence bytes after the first newline (their line's kind is known)
    uint32_t pay_head;     // sequence bytes before the first newline (kind comes from the left)
    bool has_nl, last_is_def, hi;
};

__device__ __forceinline__ uint32_t below(uint32_t b) { return b >= 32 ? ~0u : (1u << b) - 1u; }
__device__ __forceinline__ unsigned long long mask_below(uint32_t b) { return b >= 64 ? ~0ull : (1ull << b) - 1ull; }

__device__ __forceinline__ uint32_t byte_of(const uint4 v, uint32_t b) {  // b < 16
    const uint32_t w = b < 8 ? (b < 4 ? v.x : v.y) : (b < 12 ? v.z : v.w);
    return (w >> (8 * (b & 3))) & 0xFFu;
}

// the byte behind every lane's chunk (input offset o + 16; 0 behind the end of the input): the first byte of the lane
// above — one DPP wave shift — and, for lane 63, of the row after this one (`next_row_first`, when the caller has it in
// registers) or a single-lane load
__device__ __forceinline__ uint32_t byte_after_chunk(const uint4 v, bool have_next_row, uint32_t next_row_first,
                                                     const uint8_t *__restrict__ d_in, uint64_t o, uint64_t n_bytes) {
    uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v.x & 0xFFu), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    if ((threadIdx.x & 63) == 63) nb = have_next_row ? next_row_first : (o + 16 < n_bytes ? (uint32_t)d_in[o + 16] : 0u);
    return nb;
}

// `full`: the whole row lies inside the input (wave uniform): no per-lane bounds
template <class Off>  // offsets as the caller has them: absolute (64-bit) or relative to a wave-uniform base (32-bit)
__device__ __forceinline__ Chunk classify16(const uint4 v, uint32_t next_byte, bool full, Off o, Off n_bytes) {
    Chunk c;
    const uint32_t valid = full ? 0xFFFFu : (o >= n_bytes ? 0u : below((uint32_t)(n_bytes - o < 16 ? n_bytes - o : 16)));
    const uint32_t nl = match16(v, 0x0A0A0A0Au) & valid;
    c.hi = ((v.x | v.y | v.z | v.w) & 0x80808080u) != 0;  // bytes past the end are zero
    c.nl = nl;
    c.has_nl = nl != 0;
    uint32_t def_after = 0, strip = 0, def_region = 0;
#ifdef EXG_FA_CLASSIFY_LOOP
    // (the first form: a loop over the chunk's newline bits — one iteration for most chunks, but a wave runs as many as its
    // worst lane needs, and the short last line of a record in front of a definition line puts two or three newlines into one
    // chunk about once per row: ~150 of the row's ~270 wave instructions.  Kept as the A/B partner.)
    uint32_t m = nl;
    while (m) {
        const uint32_t b = (uint32_t)__ffs((int)m) - 1;
        m &= m - 1;
        // CR only in front of a real LF (a CR at byte 15 is looked at below, by its own chunk)
        if (b > 0 && byte_of(v, b - 1) == '\r') strip |= 1u << (b - 1);
        // the line that starts after this newline
        const uint32_t next = b < 15 ? ((valid >> (b + 1)) & 1u ? byte_of(v, b + 1) : 0u) : next_byte;
        if (next == '>') {
            def_after |= 1u << b;
            const uint32_t e = m ? (uint32_t)__ffs((int)m) - 1 : 16u;  // up to the next newline
            def_region |= below(e) & ~below(b + 1);
        }
    }
    if ((v.w >> 24) == '\r' && (valid >> 15) && next_byte == '\n') strip |= 1u << 15;
#else
    // Mask arithmetic on the 16-bit byte masks, the same for every lane whatever its newlines: '>' right behind a newline
    // opens a definition line, CR counts only right in front of a newline (bit 16 = the byte behind the chunk), and a
    // definition line's bytes — from behind its newline up to the next one — are one borrow chain: subtracting the start
    // bits from the stop bits sets every bit in between (each start has no other stop bit between itself and its own).
    const uint32_t gt = (match16(v, 0x3E3E3E3Eu) & valid) | (next_byte == '>' ? 0x10000u : 0u);
    const uint32_t cr = match16(v, 0x0D0D0D0Du) & valid;
    def_after = nl & (gt >> 1);
    strip = cr & ((nl | (next_byte == '\n' ? 0x10000u : 0u)) >> 1);
    const uint32_t stops = nl | 0x10000u, starts = (def_after << 1) & 0xFFFFu;
    def_region = (stops - starts) & ~stops & 0xFFFFu;
#endif
    c.def_after = def_after;
    c.last_is_def = c.has_nl && ((def_after >> (31 - __clz((int)nl))) & 1u);
    const uint32_t first = c.has_nl ? (uint32_t)__ffs((int)nl) - 1 : 16u;
    const uint32_t seq = valid & ~nl & ~strip;
    c.pay_known = seq & ~def_region & ~below(first);
    c.pay_head = seq & below(first);
    return c;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A wavefront owns a tile and walks it in sixteen 1 KiB rows (64 lanes x 16 bytes, four rows loaded at a time),
// carrying the line state from row to row: no workgroup barrier anywhere.  `Carry` = is there a newline to the
// left inside the tile (or the start of the input) and, if so, is the line after the nearest one a definition.
struct Carry {
    bool resolved, in_def;
};
// state entering this lane's chunk, then the carry after the row
__device__ __forceinline__ Carry lane_state(const Chunk &c, Carry &carry) {
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long bs = __ballot(c.has_nl), bd = __ballot(c.last_is_def);
    const unsigned long long left = bs & mask_below(lane);
    Carry st = carry;
    if (left) {
        st.resolved = true;
        st.in_def = (bd >> (63 - __clzll((long long)left))) & 1ull;
    }
    if (bs) {
        carry.resolved = true;
        carry.in_def = (bd >> (63 - __clzll((long long)bs))) & 1ull;
    }
    return st;
}

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return __shfl(v, 0, 64);
}

// ---- what a run of bytes contributes, and how two adjacent runs combine -----------------------------------------------------
// (definitions, sequence bytes after the run's first newline, sequence bytes in front of it — they count only if the run
// begins outside a definition line —, newlines, has-a-newline, what follows the last one): an ordered monoid
struct TileSum {
    unsigned long long defs, after, head, nls;  // head: sequence bytes in front of the first newline (their line's kind comes from the left)
    uint32_t has, tail;                         // a newline inside; what follows the last one is a definition line
};
// a then b
__device__ __forceinline__ TileSum compose(const TileSum &a, const TileSum &b) {
    TileSum r;
    r.defs = a.defs + b.defs;
    r.nls = a.nls + b.nls;
    if (a.has) {
        r.after = a.after + b.after + (a.tail ? 0ull : b.head);
        r.head = a.head;
        r.has = 1;
        r.tail = b.has ? b.tail : a.tail;
    } else {
        r.after = b.after;
        r.head = a.head + b.head;
        r.has = b.has;
        r.tail = b.tail;
    }
    return r;
}

// ---- pass 2 --------------------------------------------------------------------------------------------------------------
// one definition line: id / description string_t of record r (the line may run past the tile: read from HBM)
// bytes of the input through one cached, aligned 8-byte word: a walk along a line costs one dependent load per 8
// bytes instead of one per byte (the buffer is readable to round_up(n_bytes, 16))
struct ByteReader {
    const uint8_t *p;
    uint64_t base = ~0ull;
    unsigned long long w = 0;
    __device__ __forceinline__ uint32_t get(uint64_t i) {
        const uint64_t al = i & ~7ull;
        if (al != base) {
            w = *reinterpret_cast<const unsigned long long *>(p + al);
            base = al;
        }
        return (uint32_t)(w >> (8 * (i & 7))) & 0xFFu;
    }
};

// `in` = where the line's bytes are read from, indexed by INPUT offset: the input itself, or (k_fa_tile_defs) a copy of the
// line's first bytes in LDS, biased by the line's offset — the walk, the trims and the two string_t read the line a dozen
// times, byte by byte, and each read from HBM is a dependent round trip.  `limit` = end of what `in` holds (the input's end
// when it holds the whole line).
__device__ void emit_definition(const FastaDev &a, ScanWsHeader *hdr, uint64_t r, uint64_t s, const uint8_t *in, uint64_t raw_end) {
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
    // the end of the id (first ASCII whitespace)
    uint64_t id_e = ~0ull;
    for (uint64_t i = s + 1; i < raw_end; i++)
        if (is_ascii_ws(in[i])) {
            id_e = i;
            break;
        }
    uint64_t e = raw_end;
    if (raw_end < a.n_bytes && e > s && in[e - 1] == '\r') e--;  // CR only in front of a real LF
    uint32_t code = 0;
    if ((hdr->flags & EXG_RF_NON_ASCII) && !utf8_valid_global(in, s, e)) code = EXG_PE_INVALID_UTF8;
    const uint64_t id_s = s + 1;
    if (id_e == ~0ull || id_e > e) id_e = e;
    if (id_e < id_s) id_e = id_s;
    if (!code && id_e == id_s) code = EXG_PE_FASTA_MISSING_NAME;
    const bool has_desc = id_e < e;
    uint64_t d_s = has_desc ? id_e + 1 : e, d_e = e;
    if (has_desc) {
        int l;
        while (d_s < d_e && (l = ws_len_fwd(in, d_s, d_e)) > 0) d_s += (uint64_t)l;
        while (d_e > d_s && (l = ws_len_bwd(in, d_s, d_e)) > 0) d_e -= (uint64_t)l;
    }
    if (!code && (id_e - id_s > 0xFFFFFFFFull || d_e - d_s > 0xFFFFFFFFull)) code = EXG_PE_FIELD_TOO_LONG;
    if (code) {
        atomicMin(&hdr->err_word, (r << 8) | code);
        atomicMin(&hdr->err_off, (unsigned long long)s);
    }
    if (!no_store) {
        const uint4 z = {0, 0, 0, 0};
        reinterpret_cast<uint4 *>(a.d_id)[r] = make_string_global(in, id_s, id_e - id_s, a.payload_base);
        reinterpret_cast<uint4 *>(a.d_desc)[r] = has_desc ? make_string_global(in, d_s, d_e - d_s, a.payload_base) : z;
        if (has_desc) atomicOr((unsigned long long *)&a.d_desc_valid[r >> 6], 1ull << (r & 63));
    }
}

// The kept bytes of one lane's chunk -> their place in the compacted group in LDS (any byte offset).  A chunk is all
// sequence (one 16-byte store), or sequence with ONE run of bytes taken out — a newline, CR LF, the head of a definition line
// up to the end of the chunk, its tail from the start — which is closed in registers (a 128-bit shift and a byte select) and
// stored as 8 + 4 + 2 + 1 bytes; anything else (two newlines in 16 bytes) goes byte by byte.
typedef uint32_t fa_v4u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint64_t fa_u64u __attribute__((aligned(1)));
typedef uint32_t fa_u32u __attribute__((aligned(1)));
typedef uint16_t fa_u16u __attribute__((aligned(1)));

__device__ __forceinline__ void lds_store_kept(uint8_t *dst, const uint4 v, uint32_t pay) {
    if (pay == 0xFFFFu) {
        fa_v4u x = {v.x, v.y, v.z, v.w};
        *reinterpret_cast<fa_v4u *>(dst) = x;
        return;
    }
    const uint32_t z = ~pay & 0xFFFFu;             // bytes taken out (not zero here)
    const uint32_t b = (uint32_t)__ffs((int)z) - 1;  // the first of them
    const uint32_t zz = z >> b;
    if ((zz & (zz + 1u)) == 0) {
        const uint32_t s = (uint32_t)__popc(z);  // one run [b, b + s), s < 16: close it
        const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
        uint64_t slo, shi;  // the chunk shifted down by s bytes
        if (s >= 8) {
            slo = hi >> (8 * (s - 8));
            shi = 0;
        } else {
            slo = (lo >> (8 * s)) | (hi << (64 - 8 * s));
            shi = hi >> (8 * s);
        }
        const uint64_t mlo = b >= 8 ? ~0ull : ((1ull << (8 * b)) - 1ull), mhi = b > 8 ? ((1ull << (8 * (b - 8))) - 1ull) : 0ull;
        uint64_t r = (lo & mlo) | (slo & ~mlo);
        const uint64_t rhi = (hi & mhi) | (shi & ~mhi);
        const uint32_t k = 16 - s;
        if (k & 8) {
            *reinterpret_cast<fa_u64u *>(dst) = r;
            dst += 8;
            r = rhi;
        }
        if (k & 4) {
            *reinterpret_cast<fa_u32u *>(dst) = (uint32_t)r;
            dst += 4;
            r >>= 32;
        }
        if (k & 2) {
            *reinterpret_cast<fa_u16u *>(dst) = (uint16_t)r;
            dst += 2;
            r >>= 16;
        }
        if (k & 1) *dst = (uint8_t)r;
        return;
    }
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 16; q++)
        if (pay & (1u << q)) *dst++ = (uint8_t)(words[q >> 2] >> (8 * (q & 3)));
}

// ---- the scan ---------------------------------------------------------------------------------------------------------------------
// (The first tiled form was a count pass, a one-workgroup scan and an emit pass: every row classified twice, the input read
// twice, 0.80 ms per GB.)  k_fa_fused does it in one visit: a workgroup owns a 32 KiB super-tile, a wave 8 KiB of it (eight 1 KiB rows, loaded once
// and kept in registers); the wave walks its rows, carrying the line state, and keeps, per row and lane, the two possible
// sequence-byte masks (the wave begins outside / inside a definition line) and the definition mask; the workgroup's
// aggregate — the monoid above — is published, ONE scanner wave (block 0) turns the published aggregates into
// exclusive prefixes in order (the line state of 64 super-tiles by two ballots, the sums by DPP scans), and the waves emit
// from their registers as soon as their prefix has arrived.  Blocks are dispatched in order and publish before they
// wait, so the scanner never waits for a block behind a waiting one; should an aggregate still be missing after 40 us,
// the scanner computes it from the input itself.
static constexpr uint32_t kSuper = 32768, kWaveSpan = kSuper / (kThreads / 64), kWaveRows = kWaveSpan / 1024;
static constexpr unsigned long long kFPub = 1ull << 63;
static constexpr int kFHas = 62, kFTail = 61, kFAfter = 15, kFHead = 31;  // [0,15) definitions, [15,31) after, [31,47) head
static constexpr unsigned int kFPubNl = 1u << 31;

__device__ __forceinline__ unsigned long long fa_ld(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fa_st(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the aggregate of input bytes [base, base + span) by one wave, straight from the input (scanner's helping path)
__device__ TileSum fa_span_sum(const FastaDev &a, uint64_t base, uint32_t span, uint32_t lane) {
    const bool def0 = base == 0 && a.n_bytes > 0 && a.d_in[0] == '>';
    Carry carry = {base == 0, def0};
    uint32_t accA = 0, accB = (def0 && lane == 0) ? 1u : 0u;
    for (uint32_t row = 0; row * 1024 < span; row++) {
        const uint64_t rbase = base + (uint64_t)row * 1024;
        if (rbase >= a.n_bytes) break;
        const uint64_t o = rbase + (uint64_t)lane * 16;
        const uint4 v = o < a.n_bytes ? *reinterpret_cast<const uint4 *>(a.d_in + o) : make_uint4(0, 0, 0, 0);
        const uint32_t nb = byte_after_chunk(v, false, 0u, a.d_in, o, a.n_bytes);
        const Chunk c = classify16(v, nb, rbase + 1024 + 16 <= a.n_bytes, o, a.n_bytes);
        const Carry st = lane_state(c, carry);
        const uint32_t known = (uint32_t)__popc(c.pay_known), head = (uint32_t)__popc(c.pay_head);
        accA += known + ((st.resolved && !st.in_def) ? head : 0u) + ((st.resolved ? 0u : head) << 16);
        accB += (uint32_t)__popc(c.def_after) + ((uint32_t)__popc(c.nl) << 16);
    }
    const unsigned long long tot = wave_sum64((unsigned long long)accA | ((unsigned long long)accB << 32));
    TileSum e;
    e.after = tot & 0xFFFFull, e.head = (tot >> 16) & 0xFFFFull, e.defs = (tot >> 32) & 0xFFFFull, e.nls = tot >> 48;
    e.has = carry.resolved, e.tail = carry.in_def;
    return e;
}

// block 0, wave 0: exclusive prefixes of the published aggregates, in order
__device__ void fa_scanner(const FastaDev &a, const TileArrays &t, ScanWsHeader *hdr, uint32_t n_super, uint32_t help_ticks, uint32_t lane) {
    __builtin_amdgcn_s_setprio(3);
#ifndef EXG_FA_SCAN_BATCHES
#define EXG_FA_SCAN_BATCHES 4
#endif
    constexpr int kBatches = EXG_FA_SCAN_BATCHES;
    uint64_t next = 0;
    unsigned long long rec = 0, pay = 0, nls = 0;
    bool in_def = false;  // the input begins outside a definition line
    unsigned long long t_last = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    // one batch: lanes [0, r) hold consecutive aggregates
    auto absorb = [&](unsigned long long d, unsigned int nl, uint32_t r) {
        const bool mine = lane < r;
        const bool has = mine && ((d >> kFHas) & 1ull), tail = mine && ((d >> kFTail) & 1ull);
        const uint32_t defs = mine ? (uint32_t)(d & 0x7FFFull) : 0u, after = mine ? (uint32_t)((d >> kFAfter) & 0xFFFFull) : 0u,
                       head = mine ? (uint32_t)((d >> kFHead) & 0xFFFFull) : 0u, nlc = mine ? (nl & ~kFPubNl) : 0u;
        const unsigned long long bs = __ballot(has), bd = __ballot(tail);
        const unsigned long long left = bs & mask_below(lane);
        const bool st = left ? ((bd >> (63 - __clzll((long long)left))) & 1ull) != 0 : in_def;
        const uint32_t mine_pay = after + (st ? 0u : head);
        const uint32_t inc_d = wave_incl_sum_dpp(defs), inc_p = wave_incl_sum_dpp(mine_pay), inc_n = wave_incl_sum_dpp(nlc);
        if (mine) {
            fa_st(&t.f_rec[next + lane], kFPub | (rec + inc_d - defs));
            fa_st(&t.f_pay[next + lane], kFPub | ((unsigned long long)st << 62) | (pay + inc_p - mine_pay));
        }
        rec += (uint32_t)__builtin_amdgcn_readlane((int)inc_d, 63);
        pay += (uint32_t)__builtin_amdgcn_readlane((int)inc_p, 63);
        nls += (uint32_t)__builtin_amdgcn_readlane((int)inc_n, 63);
        if (bs) in_def = ((bd >> (63 - __clzll((long long)bs))) & 1ull) != 0;
        next += r;
    };
    while (next < n_super) {
        unsigned long long d[kBatches];
        unsigned int c[kBatches];
#pragma unroll
        for (int k = 0; k < kBatches; k++) {
            const uint64_t idx = next + (uint64_t)k * 64 + lane;
            d[k] = idx < n_super ? fa_ld(&t.f_agg[idx]) : 0ull;
            c[k] = idx < n_super ? __hip_atomic_load(&t.f_nl[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
        bool progressed = false;
#pragma unroll
        for (int k = 0; k < kBatches; k++) {
            const unsigned long long rdy = __ballot((d[k] & kFPub) != 0 && (c[k] & kFPubNl) != 0);
            const uint32_t r = rdy == ~0ull ? 64u : (uint32_t)__ffsll((long long)~rdy) - 1;  // leading run of published ones
            if (r > 0) {
                absorb(d[k], c[k], r);
                progressed = true;
            }
            if (r < 64) break;
        }
        if (progressed) {
            t_last = __builtin_amdgcn_s_memrealtime();
        } else if (__builtin_amdgcn_s_memrealtime() - t_last >= help_ticks) {
            // ~40 us (help_ticks of the 100 MHz clock) without the next aggregate: its block may not have been dispatched yet; progress never depends on that
            const TileSum e = fa_span_sum(a, next * kSuper, kSuper, lane);
            const unsigned long long dd = kFPub | ((unsigned long long)e.has << kFHas) | ((unsigned long long)e.tail << kFTail) | e.defs |
                                          (e.after << kFAfter) | (e.head << kFHead);
            absorb(dd, kFPubNl | (unsigned int)e.nls, 1);
            t_last = __builtin_amdgcn_s_memrealtime();
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (lane == 0) {
        t.totals[0] = rec;
        t.totals[1] = pay;
        if (rec <= t.rec_cap) t.rec_start[rec] = pay;  // sentinel: end of the last record's sequence
        hdr->total_nl = nls;
        hdr->total_lines = nls + ((a.n_bytes && a.d_in[a.n_bytes - 1] != '\n') ? 1 : 0);
    }
}

// six waves per SIMD (80 registers, 40 B of spills): 0.628 -> 0.607 ms per GB for the whole scan; seven: as five; eight (64 registers,
// 104 B of spills): 0.75.  What was tried beside it and changed nothing: the scanner wave taking 1 / 8 / 16 batches of 64 aggregates
// per poll instead of 4, and building the CR mask only in rows that hold a CR (23 of a row's ~280 instructions): the kernel is bound
// neither by the scanner nor by its instruction count.
#ifndef EXG_FA_WAVES
#define EXG_FA_WAVES 6
#endif
__global__ __launch_bounds__(kThreads, EXG_FA_WAVES) void k_fa_fused(FastaDev a, TileArrays t, ScanWsHeader *hdr, uint32_t n_super, uint32_t help_ticks) {
    __shared__ __attribute__((aligned(16))) uint32_t s_out_all[kThreads / 64][4096 / 4 + 8];
    __shared__ TileSum s_agg[kThreads / 64];
    __shared__ unsigned long long s_pre_rec, s_pre_pay;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (a scalar: what hangs off it is wave uniform)
    if (blockIdx.x == 0) {  // the scanner: one wave, no tile
        if (wave == 0) fa_scanner(a, t, hdr, n_super, help_ticks, lane);
        return;
    }
    const uint32_t st_i = blockIdx.x - 1;
    const uint64_t wbase = (uint64_t)st_i * kSuper + (uint64_t)wave * kWaveSpan;  // this wave's 8 KiB (wave uniform)
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
    // input bytes from wbase on, clamped: every per-lane offset and bound below is 32-bit against a wave-uniform base
    const uint32_t avail = wbase >= a.n_bytes ? 0u : (a.n_bytes - wbase > 0x40000000ull ? 0x40000000u : (uint32_t)(a.n_bytes - wbase));
    const uint8_t *wp = a.d_in + wbase;
    const uint32_t lo = lane * 16;
    // ---- the wave's rows, once
    uint4 v[kWaveRows];
#pragma unroll
    for (int j = 0; j < (int)kWaveRows; j++) {
        const uint32_t off = (uint32_t)j * 1024 + lo;
        v[j] = off < avail ? ld_stream16(wp + off) : make_uint4(0, 0, 0, 0);
    }
    // ---- the walk: line state from row to row; what the emission needs of it stays in registers
    const bool first = st_i == 0 && wave == 0;
    const bool def0 = first && a.n_bytes > 0 && a.d_in[0] == '>';
    Carry carry = {first, def0};
    uint32_t accA = 0, accB = (def0 && lane == 0) ? 1u : 0u;
    uint32_t pm[kWaveRows], dm[kWaveRows];  // sequence bytes if the wave begins outside | inside (<< 16) a definition line; definitions
    bool any_hi = false;
#pragma unroll
    for (int j = 0; j < (int)kWaveRows; j++) {
        const uint32_t rrel = (uint32_t)j * 1024;
        pm[j] = 0, dm[j] = 0;
        if (rrel < avail) {  // (wave uniform)
            const uint32_t orel = rrel + lo;
            const bool have_next = j + 1 < (int)kWaveRows;
            // the byte behind every lane's chunk: the first byte of the lane above (one DPP wave shift); for lane 63 the first
            // byte of the next row (in registers) or, behind the wave's last row, one byte of the input
            const uint32_t next_row_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[have_next ? j + 1 : j].x) & 0xFFu;  // (all lanes)
            uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v[j].x & 0xFFu), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
            if (lane == 63) nb = have_next ? next_row_first : (orel + 16 < avail ? (uint32_t)wp[orel + 16] : 0u);
            const Chunk c = classify16(v[j], nb, rrel + 1024 + 16 <= avail, orel, avail);
            const Carry stt = lane_state(c, carry);
            const uint32_t known = (uint32_t)__popc(c.pay_known), head = (uint32_t)__popc(c.pay_head);
            accA += known + ((stt.resolved && !stt.in_def) ? head : 0u) + ((stt.resolved ? 0u : head) << 16);
            accB += (uint32_t)__popc(c.def_after) + ((uint32_t)__popc(c.nl) << 16);
            any_hi = any_hi || c.hi;
            const uint32_t out_of_def = c.pay_known | ((stt.resolved ? !stt.in_def : true) ? c.pay_head : 0u);
            const uint32_t in_a_def = c.pay_known | ((stt.resolved ? !stt.in_def : false) ? c.pay_head : 0u);
            pm[j] = out_of_def | (in_a_def << 16);
            dm[j] = c.def_after;
        }
    }
    {
        const unsigned long long tot = wave_sum64((unsigned long long)accA | ((unsigned long long)accB << 32));
        const bool hi = __ballot(any_hi) != 0;
        if (lane == 0) {
            TileSum e;
            e.after = tot & 0xFFFFull, e.head = (tot >> 16) & 0xFFFFull, e.defs = (tot >> 32) & 0xFFFFull, e.nls = tot >> 48;
            e.has = carry.resolved, e.tail = carry.in_def;
            s_agg[wave] = e;
            if (hi) atomicOr(&hdr->flags, EXG_RF_NON_ASCII);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        TileSum e = s_agg[0];
        for (uint32_t k = 1; k < kThreads / 64; k++) e = compose(e, s_agg[k]);
        __hip_atomic_store(&t.f_nl[st_i], kFPubNl | (unsigned int)e.nls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fa_st(&t.f_agg[st_i], kFPub | ((unsigned long long)e.has << kFHas) | ((unsigned long long)e.tail << kFTail) | e.defs |
                                  (e.after << kFAfter) | (e.head << kFHead));
        // ... and wait for the answer
        unsigned long long r, p;
        for (;;) {
            r = fa_ld(&t.f_rec[st_i]);
            p = fa_ld(&t.f_pay[st_i]);
            if ((r & kFPub) && (p & kFPub)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        s_pre_rec = r & ~kFPub;
        s_pre_pay = p & ~kFPub;
    }
    __syncthreads();
    // ---- this wave's own prefix: the super-tile's, then the waves in front of it
    uint64_t rec_at = s_pre_rec, pay_at = s_pre_pay & ~(1ull << 62);
    bool wave_in_def = (s_pre_pay >> 62) & 1ull;
    for (uint32_t k = 0; k < wave; k++) {
        const TileSum e = s_agg[k];
        rec_at += e.defs;
        pay_at += e.after + (wave_in_def ? 0ull : e.head);
        if (e.has) wave_in_def = e.tail != 0;
    }
    if (def0) {  // the definition at offset 0: no newline announces it
        if (lane == 0 && rec_at < t.rec_cap) {
            t.rec_start[rec_at] = 0;
            t.rec_def_off[rec_at] = 0;
        }
        rec_at++;
    }
    // ---- emission from the registers
    uint32_t *s_out = s_out_all[wave];
    uint8_t *out8 = reinterpret_cast<uint8_t *>(s_out);
    uint32_t g_out = 0;
#pragma unroll
    for (int j = 0; j < (int)kWaveRows; j++) {
        const uint32_t rrel = (uint32_t)j * 1024;
        if (rrel < avail) {
            const uint32_t orel = rrel + lo;
            const uint32_t pay = wave_in_def ? pm[j] >> 16 : pm[j] & 0xFFFFu;
            const uint32_t cnt = (uint32_t)__popc(pay) | ((uint32_t)__popc(dm[j]) << 16);
            const uint32_t incl = wave_incl_sum_dpp(cnt);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t excl = incl - cnt;
            const uint32_t my_off = g_out + (excl & 0xFFFFu);
            uint32_t d = dm[j];
            uint64_t r = rec_at + (excl >> 16);
            while (d) {
                const uint32_t b = (uint32_t)__ffs((int)d) - 1;
                if (r < t.rec_cap && orel + b + 1 < avail) {
                    t.rec_start[r] = pay_at + my_off + (uint32_t)__popc(pay & below(b));
                    t.rec_def_off[r] = wbase + orel + b + 1;
                }
                r++;
                d &= d - 1;
            }
            if (!no_store && pay) lds_store_kept(out8 + my_off, v[j], pay);
            g_out += tot & 0xFFFFu;
            rec_at += tot >> 16;
            const bool last_row = j + 1 == (int)kWaveRows || rrel + 1024 >= avail;
            if (!no_store && g_out && ((j & 3) == 3 || last_row)) {  // (uniform per wave)
                wave_sync();
                uint8_t *dst0 = a.d_payload + pay_at;
                const uint32_t head = (16u - (uint32_t)((uintptr_t)dst0 & 15)) & 15u;
                uint32_t written = 0;
                if (g_out >= head) {
                    const uint32_t n_full = (g_out - head) / 16;
                    if (lane < head) dst0[lane] = out8[lane];
                    for (uint32_t q = lane; q < n_full; q += 64) {
                        const fa_v4u lv = *reinterpret_cast<const fa_v4u *>(out8 + head + q * 16);
                        uint4 ov;
                        ov.x = lv.x, ov.y = lv.y, ov.z = lv.z, ov.w = lv.w;
                        st_stream16(reinterpret_cast<uint4 *>(dst0 + head + q * 16), ov);
                    }
                    written = head + n_full * 16;
                }
                const uint32_t rem = g_out - written;
                if (last_row) {
                    if (lane < rem) dst0[written + lane] = out8[written + lane];
                    written = g_out;
                } else if (written) {
                    const uint8_t x = lane < rem ? out8[written + lane] : (uint8_t)0;
                    wave_sync();
                    if (lane < rem) out8[lane] = x;
                }
                wave_sync();
                pay_at += written;
                g_out -= written;
            }
            if (no_store) {
                pay_at += g_out;
                g_out = 0;
            }
        }
    }
}

// id / description of every record, a thread per record.  The thread copies the head of the line into LDS — 64 bytes (four
// independent 16-byte loads, one round trip), the next 64 only if the newline is not among them — and parses the copy; only a
// longer definition line is walked in HBM (a chain of dependent loads, ~15 us: inside the scan it would stall a whole
// wavefront for the one lane that owns a '>').
static constexpr uint32_t kDefStage = 128, kDefStride = kDefStage / 4 + 1;  // dwords per thread: the odd stride spreads the banks
// A buffer without EXG_F_EOF is a batch of a longer input: its last record is still open (its sequence may go on behind the
// buffer), so it is nobody's row yet — the scan reports where its definition line begins (consumed_bytes) and the next batch
// starts there.
__device__ __forceinline__ uint64_t fa_closed_records(const FastaDev &a, const TileArrays &t) {
    uint64_t n_rec = t.totals[0];
    if (!(a.flags & EXG_F_EOF) && n_rec) n_rec--;
    return n_rec;
}

__global__ __launch_bounds__(256) void k_fa_tile_defs(FastaDev a, TileArrays t, ScanWsHeader *hdr) {
    __shared__ uint32_t s_line[256 * kDefStride];
    uint64_t n_rec = fa_closed_records(a, t);
    if (n_rec > t.rec_cap) n_rec = t.rec_cap;
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
    uint32_t *slot = s_line + threadIdx.x * kDefStride;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += (uint64_t)gridDim.x * blockDim.x) {
        if (!no_store && r >= a.capacity) {
            atomicOr(&hdr->flags, EXG_RF_CAPACITY);
            continue;
        }
        const uint64_t s = t.rec_def_off[r];
        const uint32_t avail = a.n_bytes - s < kDefStage ? (uint32_t)(a.n_bytes - s) : kDefStage;
        typedef uint32_t v4u_u __attribute__((ext_vector_type(4), aligned(1)));
        const uint8_t *line = reinterpret_cast<const uint8_t *>(slot);
        uint32_t len = 0;
        // 64 bytes first (most definition lines end there: half the lines fetched), the other 64 only when they do not
        for (uint32_t half = 0; half < 2; half++) {
            const uint32_t k0 = half * (kDefStage / 32);
            if (16 * k0 >= avail) break;
            v4u_u q[kDefStage / 32];
#pragma unroll
            for (uint32_t k = 0; k < kDefStage / 32; k++) {
                const uint32_t kk = k0 + k;
                q[k] = (v4u_u){0, 0, 0, 0};
                if (16 * kk + 16 <= avail) {
                    q[k] = *reinterpret_cast<const v4u_u *>(a.d_in + s + 16 * kk);
                } else if (16 * kk < avail) {  // the input's last bytes
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (uint32_t j = 16 * kk; j < avail; j++) w[(j & 15) >> 2] |= (uint32_t)a.d_in[s + j] << (8 * (j & 3));
                    q[k] = (v4u_u){w[0], w[1], w[2], w[3]};
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < kDefStage / 32; k++) {
                slot[4 * (k0 + k)] = q[k].x;
                slot[4 * (k0 + k) + 1] = q[k].y;
                slot[4 * (k0 + k) + 2] = q[k].z;
                slot[4 * (k0 + k) + 3] = q[k].w;
            }
            const uint32_t upto = avail < (half + 1) * (kDefStage / 2) ? avail : (half + 1) * (kDefStage / 2);
            while (len < upto && line[len] != '\n') len++;
            if (len < upto) break;  // the newline is in
        }
        if (len < avail || s + avail == a.n_bytes) {
            emit_definition(a, hdr, r, s, line - s, s + len);  // the whole line is in the copy
        } else {
            uint64_t raw_end = s + avail;
            ByteReader rd;
            rd.p = a.d_in;
            while (raw_end < a.n_bytes && rd.get(raw_end) != '\n') raw_end++;
            emit_definition(a, hdr, r, s, a.d_in, raw_end);
        }
    }
}

__global__ __launch_bounds__(256) void k_fa_tile_strings(FastaDev a, TileArrays t, ScanWsHeader *hdr) {
    if (a.flags & EXG_F_NO_STORE) return;
    uint64_t n_rec = fa_closed_records(a, t);
    if (n_rec > t.rec_cap) n_rec = t.rec_cap;
    const uint64_t n = n_rec < a.capacity ? n_rec : a.capacity;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s = t.rec_start[r], len = t.rec_start[r + 1] - s;
        if (len > 0xFFFFFFFFull) {
            atomicMin(&hdr->err_word, (r << 8) | EXG_PE_FIELD_TOO_LONG);
            len = 0;
        }
        // sequences are validated after their definition (exon FASTAArrayBuilder::append order)
        if ((hdr->flags & EXG_RF_NON_ASCII) && reinterpret_cast<const uint32_t *>(a.d_id)[r * 4] != 0 &&
            !utf8_valid_global(a.d_payload, s, s + len)) {
            atomicMin(&hdr->err_word, (r << 8) | EXG_PE_INVALID_UTF8);
            atomicMin(&hdr->err_off, (unsigned long long)t.rec_def_off[r]);
        }
        reinterpret_cast<uint4 *>(a.d_seq)[r] = make_string_global(a.d_payload, s, len, a.seq_payload_base);
    }
}

__global__ void k_fa_tile_finalize(FastaDev a, TileArrays t, ScanWsHeader *hdr, exg_scan_result *res) {
    if (threadIdx.x || blockIdx.x) return;
    const uint64_t n_all = t.totals[0], n_owned = fa_closed_records(a, t);
    const bool open_tail = !(a.flags & EXG_F_EOF);
    exg_scan_result r;
    r.n_lines = hdr->total_lines;
    r.flags = hdr->flags;
    if (n_all > t.rec_cap) r.flags |= EXG_RF_INDEX_OVERFLOW;
    // (the open record's sequence bytes were compacted like everyone's: they are simply not handed out)
    r.payload_bytes = open_tail ? (n_all && n_all <= t.rec_cap ? t.rec_start[n_all - 1] : 0) : t.totals[1];
    r.redo_tiles = 0;
    r.error_code = 0;
    r.error_offset = ~0ull;
    r.error_record = ~0ull;
    // the reader wants a definition first
    if (a.n_bytes && a.d_in[0] != '>') {
        uint64_t e = 0;
        while (e < a.n_bytes && a.d_in[e] != '\n') e++;
        if (e < a.n_bytes && e > 0 && a.d_in[e - 1] == '\r') e--;
        atomicMin(&hdr->err_word, (0ull << 8) | (e == 0 ? EXG_PE_FASTA_EMPTY_DEF : EXG_PE_FASTA_MISSING_PREFIX));
    }
    uint64_t n_rec = (n_owned < a.capacity || (a.flags & EXG_F_NO_STORE)) ? n_owned : a.capacity;
    uint64_t consumed = open_tail ? (n_all && n_all <= t.rec_cap ? t.rec_def_off[n_all - 1] : 0) : a.n_bytes;
    const unsigned long long err = hdr->err_word;
    if (err != kNoError) {
        const uint64_t rec = err >> 8;
        r.error_code = (uint32_t)(err & 0xFF);
        r.error_record = rec;
        r.error_offset = hdr->err_off != ~0ull ? hdr->err_off : 0;
        if (r.error_code == EXG_PE_FASTA_EMPTY_DEF || r.error_code == EXG_PE_FASTA_MISSING_PREFIX) r.error_offset = 0;
        if (rec < n_rec) n_rec = rec;
        consumed = r.error_offset;
    }
    r.n_records = n_rec;
    r.consumed_bytes = consumed;
    *res = r;
}

}  // namespace

__global__ void k_init_hdr(ScanWsHeader *hdr, uint64_t lines_cap, uint32_t mode);

int run_fasta_tiled(const FastaDev &dev, uint8_t *ws, const FastqWsLayout &l, exg_scan_result *d_result, hipStream_t stream) {
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    TileArrays t;
    // 28 B per 32 KiB super-tile live in the fused kernels' descriptor region (24 B per 16 KiB); the per-record arrays in
    // the line arrays
    unsigned long long *region = reinterpret_cast<unsigned long long *>(ws + l.off_tile_desc);
    uint64_t *base = reinterpret_cast<uint64_t *>(ws + l.off_nl_pos);
    t.rec_start = base;
    t.rec_def_off = base + (l.lines_cap + 2);
    t.rec_cap = l.lines_cap;
    // records <= lines: the per-record arrays hold lines_cap + 2 entries each (a file with more definition
    // lines than that is reported like a line-index overflow)
    const uint64_t n_super = (dev.n_bytes + kSuper - 1) / kSuper;
    if (n_super > 0x7FFFFFF0ull) {
        set_error("exg_fasta_scan: buffer too large for one launch (%llu super-tiles)", (unsigned long long)n_super);
        return EXG_E_INVALID_ARG;
    }
    t.f_agg = region;
    t.f_rec = region + n_super;
    t.f_pay = region + 2 * n_super;
    t.totals = reinterpret_cast<uint64_t *>(region + 3 * n_super);
    t.f_nl = reinterpret_cast<unsigned int *>(region + 3 * n_super + 2);
    hipLaunchKernelGGL(k_init_hdr, dim3(1), dim3(1), 0, stream, hdr, l.lines_cap, 0u);
    EXG_HIP_CHECK(hipMemsetAsync(region, 0, (size_t)(3 * n_super + 2) * 8 + (size_t)n_super * 4, stream));
    // (EXG_FASTA_HELP_TICKS = 0 makes the scanner compute every aggregate it does not find at once: the tests' way into that path)
    static const uint32_t help_ticks = getenv("EXG_FASTA_HELP_TICKS") ? (uint32_t)strtoul(getenv("EXG_FASTA_HELP_TICKS"), nullptr, 10) : 4000u;
    hipLaunchKernelGGL(k_fa_fused, dim3((uint32_t)n_super + 1), dim3(kThreads), 0, stream, dev, t, hdr, (uint32_t)n_super, help_ticks);
    const uint64_t est_rec = dev.n_bytes / 64 + 256;
    const uint32_t sgrid = (uint32_t)((est_rec + 255) / 256 < 4096 ? (est_rec + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_fa_tile_defs, dim3(sgrid), dim3(256), 0, stream, dev, t, hdr);
    hipLaunchKernelGGL(k_fa_tile_strings, dim3(sgrid), dim3(256), 0, stream, dev, t, hdr);
    hipLaunchKernelGGL(k_fa_tile_finalize, dim3(1), dim3(1), 0, stream, dev, t, hdr, d_result);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

}  // namespace exg
