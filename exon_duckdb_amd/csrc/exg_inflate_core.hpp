// exg_inflate_core.hpp — the DEFLATE decoder of one wavefront (shared by exg_inflate.hip: one wave per gzip member,
// and exg_inflate_stream.hip: one big member decoded in chunks).  See exg_inflate.hip for the kernel shape.
#pragma once
#include <type_traits>

#include "exg_common.hpp"

namespace exg {

struct InflateMember {
    unsigned long long comp_off;   // offset of the DEFLATE stream in d_comp
    unsigned long long comp_size;  // bytes available from comp_off (deflate data + trailer)
    unsigned long long out_off;    // where the member's output starts in d_out
    unsigned long long out_cap;    // bytes it may produce (ISIZE when known)
};

struct InflateStatus {
    unsigned int code;             // 0 ok; 1 bad block type / stored len; 2 bad code lengths; 3 bad symbol or distance;
                                   // 4 output overflow; 5 input exhausted; 6 not text (InflateJob::text_probe)
    unsigned int pad;
    unsigned long long produced;   // bytes written
    unsigned long long consumed;   // compressed bytes consumed (from comp_off, byte aligned after the final block)
};

static constexpr int kWinBytes = 32768;
// LDS decides how many members a CU decodes at once, and this decode is a serial chain per member: what it needs is
// co-resident waves.  Round 3: 10 / 9-bit tables, a 512-byte input ring and six waves per SIMD (6.4 KiB per member) gave 63 GB/s
// of FASTQ; 9 / 9 bits, a 256-byte input ring (chunks of 128 B, two bytes per lane) fit 5112 B = four LDS granules: EIGHT
// waves per SIMD, 32 members per CU, 8192 in flight -> 70 GB/s (VCF 75 -> 83, FASTA 67 -> 73), although a 9-bit literal table
// sends more codes to the serial decoder (the same tables at six waves: 59 GB/s) and 64 VGPRs spill 30 registers
// (tools/ab_lib.sh, one box; -DEXG_INFLATE_LIT_BITS=10 -DEXG_INFLATE_DIST_BITS=9 -DEXG_INFLATE_IN_RING=512 -DEXG_INFLATE_WAVES=6
// builds the former shape).
#ifndef EXG_INFLATE_IN_RING
#define EXG_INFLATE_IN_RING 256
#endif
static constexpr int kInRing = EXG_INFLATE_IN_RING;  // two chunks of the compressed input (512: chunks of 256 B, a dword per lane; 256: 128 B, two bytes per lane)
static_assert(kInRing == 512 || kInRing == 256, "two chunks of 64 dwords or of 64 half-words");
static constexpr int kInChunk = kInRing / 2, kInChunkBitsLog2 = kInRing == 512 ? 11 : 10;  // 256 B = 2048 bits
static constexpr int kInWord = kInChunk / 64;        // bytes of a chunk every lane loads and stages
static constexpr int kInMirror = 40;                 // the ring's first bytes again behind it: one window read (36 B) never wraps
#ifndef EXG_INFLATE_LIT_BITS
#define EXG_INFLATE_LIT_BITS 9
#endif
#ifndef EXG_INFLATE_DIST_BITS
#define EXG_INFLATE_DIST_BITS 9
#endif
static constexpr int kLitBits = EXG_INFLATE_LIT_BITS, kDistBits = EXG_INFLATE_DIST_BITS;
#ifndef EXG_INFLATE_FAST_MATCHES
#define EXG_INFLATE_FAST_MATCHES 8
#endif
static constexpr int kFastSlots = EXG_INFLATE_FAST_MATCHES ? EXG_INFLATE_FAST_MATCHES : 1;  // 0: every match takes the in-order loop (the A/B partner)
#ifndef EXG_INFLATE_FAST_LEN
#define EXG_INFLATE_FAST_LEN 16
#endif
static constexpr uint32_t kFastLen = EXG_INFLATE_FAST_LEN;  // 8: a byte per lane; 16: the lanes of a longer match copy a second byte
static_assert(kFastLen == 8 || kFastLen == 16, "eight lanes per match, one or two bytes each");

// SYM = false: the window holds bytes (a gzip member decoded from its first bit).
// SYM = true:  the window holds 16-bit symbols — a byte, or 0x8000 | i for "byte i of the 32 KiB in front of where
//              this decode started", which are not known yet (exg_inflate_stream: one big member decoded in chunks).
// RING = 32 Ki: the whole window is an LDS ring (39 KiB / 72 KiB per wave: 4 / 2 waves per CU).
// RING < 32 Ki: the ring holds the newest RING elements only; a match that reaches further back reads the output the
//               wave flushed to HBM earlier (RING = 2 Ki: 6.4 KiB / 8.5 KiB per wave => 6 / 4-5 waves per SIMD, which
//               is what hides the latency of this serial decode).  A decode without output (d_out = NULL) copies nothing.
template <bool SYM, uint32_t RING = kWinBytes>
struct InflateLdsT {
    using Elem = typename std::conditional<SYM, uint16_t, uint8_t>::type;
    static_assert(RING >= 2048 && RING <= kWinBytes && (RING & (RING - 1)) == 0, "ring: power of two, >= 1 KiB flush + a match");
    static constexpr uint32_t kRing = RING;
    static constexpr bool kGlobalWindow = RING < kWinBytes;
    Elem win[kRing + 16 / sizeof(Elem)];  // + a spare element: where the lanes without a literal store theirs
    uint8_t in[kInRing + kInMirror];  // + the first bytes of the ring again: a window read never wraps
    // Primary tables, 0 = code longer than the table (or unused).  The entries carry what the token needs, so a
    // decode is peek -> lit_lut -> dist_lut, three dependent LDS levels instead of five:
    //   lit_lut  literal: bits 0-3 code length, 4-11 byte, 12 end-of-block, 13 invalid symbol
    //            length : bit 15, bits 0-3 code length, 4-6 extra bit count, 7-14 base length - 3
    //   dist_lut bits 0-3 code length, 4-8 distance symbol, 9 invalid symbol (base and extra bit count are arithmetic:
    //            dist_base_extra)
    //            (the code-length alphabet of a dynamic header borrows it as a plain u16 table)
    uint16_t lit_lut[1 << kLitBits];
    uint16_t dist_lut[1 << kDistBits];
    // lit_sorted / dist_sorted: symbols ordered by (length, symbol) — canonical decode of the codes longer than the tables.
    // lens: code lengths while a block header is read — [0,288) literal/length, [288,320) distance; [32,348) scratch while a
    // dynamic header is read.  They share their storage with lit_sorted (LDS decides how many members a CU decodes at once):
    // build_table reads all the lengths before it writes, and the distance table is built first
    union {
        uint16_t lit_sorted[288];
        uint8_t lens[384];
    };
    uint8_t dist_sorted[32];  // (distance symbols and the code-length alphabet fit a byte: what pays for fast_slot)
    uint16_t lit_count[16], dist_count[16];
    // descriptors of a step's independent short matches (copied eight at a time, eight lanes each: see `fast` below)
    uint32_t fast_slot[kFastSlots + 1];  // (+ one the lanes without a descriptor write to)
};

static __device__ __constant__ unsigned short kLenBase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                                      31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static __device__ __constant__ unsigned char kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static __device__ __constant__ unsigned short kDistBase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                                       33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                                       1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static __device__ __constant__ unsigned char kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static __device__ __constant__ unsigned char kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ uint32_t sgpr(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
// per lane: bit `lane` of a wave-uniform mask ? if1 : if0 — one v_cndmask with the mask as its scalar operand
// (readfirstlane: free where the compiler knows the mask to be uniform; where it does not — a decode inside a loop whose
// exit it takes for divergent — it is what puts the mask into scalar registers)
__device__ __forceinline__ unsigned long long uniform64(unsigned long long m) {
    return ((unsigned long long)sgpr((uint32_t)(m >> 32)) << 32) | sgpr((uint32_t)m);  // (the builtin returns int: sgpr() is unsigned)
}
__device__ __forceinline__ uint32_t mask_sel(unsigned long long m, uint32_t if0, uint32_t if1) {
    m = uniform64(m);
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t mask_sel0(unsigned long long m, uint32_t if1) {
    m = uniform64(m);
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(if1), "s"(m));
    return r;
}

// ---- bit input: absolute bit position + a 2 KiB LDS ring of the compressed bytes --------------------------
struct BitIn {
    unsigned long long bitpos;  // next bit, relative to g0 (wave uniform)
    uint32_t loaded;            // input chunks [0, loaded) (256 B each) have been staged
    uint32_t pre;               // this lane's dword of chunk `loaded`: loaded from HBM when the chunk in front of it was
                                // staged, so the load's latency passes while that chunk is decoded
    uint32_t limit;             // bytes available relative to g0
    const uint8_t *g0;          // 16-byte aligned global address of chunk 0
};

// Bit positions inside a decode are relative to a 16-byte aligned address close to where it STARTS (not to the
// start of the stream): byte offsets stay 32-bit however far into a multi-GB member a chunk begins.
struct BitBase {
    const uint8_t *g0;   // aligned address the local bit positions count from
    long long rel_bits;  // local position 0 = this bit of the stream (relative to comp_off; >= -120)
    uint32_t limit;      // bytes readable from g0 (clamped: one decode reads < 4 GiB of input)
};
__device__ __forceinline__ BitBase bit_base(const uint8_t *d_comp, unsigned long long comp_off, unsigned long long comp_size,
                                            unsigned long long first_bit) {
    const unsigned long long a0 = (comp_off + (first_bit >> 3)) & ~15ull;
    BitBase b;
    b.g0 = d_comp + a0;
    b.rel_bits = ((long long)a0 - (long long)comp_off) * 8;
    const unsigned long long avail = comp_off + comp_size - a0;
    b.limit = avail > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)avail;
    return b;
}

// this lane's word of chunk c of the compressed input (coalesced, 4 or 2 B per lane)
__device__ __forceinline__ uint32_t chunk_word(const BitIn &br, uint32_t c, uint32_t lane) {
    const uint32_t off = c * kInChunk + lane * kInWord;
    if (off >= ((br.limit + 15) & ~15u)) return 0u;
    if constexpr (kInWord == 4) return *reinterpret_cast<const uint32_t *>(br.g0 + off);
    return *reinterpret_cast<const uint16_t *>(br.g0 + off);
}
template <class L>
__device__ __forceinline__ void stage_word(L &s, uint32_t at, uint32_t w) {
    if constexpr (kInWord == 4) *reinterpret_cast<uint32_t *>(s.in + at) = w;
    else *reinterpret_cast<uint16_t *>(s.in + at) = (uint16_t)w;
}
// staging starts at the chunk that holds bit br.bitpos
__device__ __forceinline__ void start_input(BitIn &br, uint32_t lane) {
    br.loaded = (uint32_t)(br.bitpos >> kInChunkBitsLog2);
    br.pre = chunk_word(br, br.loaded, lane);
}

// keep the chunk that holds the current byte and the next one staged (reads go up to ~40 bytes ahead)
template <class L>
__device__ __forceinline__ void ensure(L &s, BitIn &br, uint32_t lane) {
    uint32_t c = (uint32_t)(br.bitpos >> kInChunkBitsLog2);
    while (c + 1 >= br.loaded) {
        stage_word(s, (br.loaded & 1) * kInChunk + lane * kInWord, br.pre);
        if (!(br.loaded & 1) && lane < kInMirror / kInWord) stage_word(s, kInRing + lane * kInWord, br.pre);
        br.loaded++;
        br.pre = chunk_word(br, br.loaded, lane);
    }
}

// 64 bits starting at absolute bit `o` (per lane), >= 57 of them valid
template <class L>
__device__ __forceinline__ unsigned long long peek_at(const L &s, unsigned long long o) {
    uint32_t byte = (uint32_t)(o >> 3);
    uint32_t a = byte & ~3u;
    uint32_t w0 = *reinterpret_cast<const uint32_t *>(s.in + (a & (kInRing - 1)));
    uint32_t w1 = *reinterpret_cast<const uint32_t *>(s.in + ((a + 4) & (kInRing - 1)));
    uint32_t w2 = *reinterpret_cast<const uint32_t *>(s.in + ((a + 8) & (kInRing - 1)));
    uint32_t sh = 8 * (byte & 3u) + (uint32_t)(o & 7);  // 0..31
    unsigned long long lo = ((unsigned long long)w1 << 32) | w0;
    unsigned long long v = lo >> sh;
    if (sh) v |= (unsigned long long)w2 << (64 - sh);
    return v;
}
// NW x 64 bits starting at bit `o` (per lane; the low 32 bits of the absolute position), as 64-bit windows (lo, hi): 2 NW + 1 dwords, 2 NW funnel shifts
template <int NW, class L>
__device__ __forceinline__ void peekn_at(const L &s, uint32_t o, uint32_t (&lo)[NW], uint32_t (&hi)[NW]) {
    static_assert(4 * (2 * NW + 1) <= kInMirror + 4, "the mirror covers one window read");
    const uint32_t a = (o >> 3) & (kInRing - 4);  // (the ring's size in bits divides 2^32)
    uint32_t w[2 * NW + 1];
#pragma unroll
    for (int i = 0; i < 2 * NW + 1; i++) w[i] = *reinterpret_cast<const uint32_t *>(s.in + a + 4 * i);
    const uint32_t sh = o & 31u;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        lo[k] = __builtin_amdgcn_alignbit(w[2 * k + 1], w[2 * k], sh);
        hi[k] = __builtin_amdgcn_alignbit(w[2 * k + 2], w[2 * k + 1], sh);
    }
}
template <class L>
__device__ __forceinline__ unsigned long long peek(L &s, BitIn &br, uint32_t lane) {
    ensure(s, br, lane);
    unsigned long long v = peek_at(s, br.bitpos);
    uint32_t lo = sgpr((uint32_t)v), hi = sgpr((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <class L>
__device__ __forceinline__ uint32_t getbits(L &s, BitIn &br, uint32_t n, uint32_t lane) {
    unsigned long long v = peek(s, br, lane);
    br.bitpos += n;
    return (uint32_t)v & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
}

// Build one decode table from code lengths lens[0..n): LUT (primary `bits`), sorted symbols, counts.
// Returns false when the lengths are over-subscribed or incomplete (except the single-code cases zlib allows).
// kFill: what the slots of codes longer than the table (and of unused codes) hold
struct EncPlain {  // (symbol << 4) | length
    static constexpr uint16_t kFill = 0;
    __device__ __forceinline__ uint16_t operator()(uint32_t sym, uint32_t l) const { return (uint16_t)((sym << 4) | l); }
};
// literal / length slots without an entry: both stop flags and a length of ONE bit — every speculative decode hops at least one
// bit without a max() of its own (the chain walk adds the hop to its position: a hop of 0 would never end)
static constexpr uint32_t kLitLong = 0x3001u;
__device__ __forceinline__ bool lit_is_long(uint32_t e) { return (e & 0xB000u) == 0x3000u; }
struct EncLit {
    static constexpr uint16_t kFill = (uint16_t)kLitLong;
    __device__ __forceinline__ uint16_t operator()(uint32_t sym, uint32_t l) const {
        if (sym < 256) return (uint16_t)((sym << 4) | l);
        if (sym == 256) return (uint16_t)((1u << 12) | l);
        if (sym > 285) return (uint16_t)((1u << 13) | l);
        const uint32_t si = sym - 257;
        return (uint16_t)(0x8000u | ((uint32_t)(kLenBase[si] - 3) << 7) | ((uint32_t)kLenExtra[si] << 4) | l);
    }
};
// (distance slots without an entry stay 0: a fill with flag bits, so that "ends the step" is one compare against 0x1000 without
// the subtract in front of it, measured 0 .. -1.7 %: kept out)
__device__ __forceinline__ bool dist_is_long(uint32_t de) { return de == 0; }
struct EncDist {
    static constexpr uint16_t kFill = 0;
    __device__ __forceinline__ uint16_t operator()(uint32_t sym, uint32_t l) const {
        if (sym > 29) return (uint16_t)((1u << 9) | l);
        return (uint16_t)((sym << 4) | l);
    }
};
// RFC 1951 distance symbol -> base distance and extra bit count without a table: symbols come in pairs that
// double the range (kDistBase / kDistExtra hold the same numbers for the serial path)
__device__ __forceinline__ void dist_base_extra(uint32_t sym, uint32_t *base, uint32_t *extra) {
    const uint32_t dx = sym < 2 ? 0u : (sym >> 1) - 1u;
    *extra = dx;
    *base = sym < 2 ? sym + 1u : ((2u | (sym & 1u)) << dx) + 1u;
}

template <class Lut, class Enc, class Sorted>
__device__ inline bool build_table(const uint8_t *lens, uint32_t n, Lut *lut, uint32_t bits, Sorted *sorted, uint16_t *count,
                            uint32_t lane, Enc enc) {
    for (uint32_t e = lane; e < (1u << bits); e += 64) lut[e] = Enc::kFill;
    // all the lengths (n <= 320) are read before anything is written: `sorted` may lie over `lens`
    uint32_t lv[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t sym = k * 64 + lane;
        lv[k] = sym < n ? lens[sym] : 0;
    }
    // counts per length
    uint32_t cnt[16];
#pragma unroll
    for (int L = 0; L < 16; L++) cnt[L] = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if ((uint32_t)k * 64 >= n) break;
#pragma unroll
        for (int L = 1; L < 16; L++) cnt[L] += __popcll(__ballot(lv[k] == (uint32_t)L));
    }
    // canonical first codes and offsets into `sorted`
    uint32_t first[16], offs[16];
    uint32_t code = 0, off = 0;
    int left = 1;
    bool bad = false;
#pragma unroll
    for (int L = 1; L < 16; L++) {
        left = (left << 1) - (int)cnt[L];
        if (left < 0) bad = true;
        code = (code + cnt[L - 1]) << 1;
        first[L] = code;
        offs[L] = off;
        off += cnt[L];
    }
    first[0] = 0;
    offs[0] = 0;
#pragma unroll
    for (int L = 0; L < 16; L++)
        if (lane == (uint32_t)L) count[L] = (uint16_t)(L ? cnt[L] : 0);
    if (bad) return false;
    // incomplete codes are only legal with a single code of length 1 (zlib / puff behaviour)
    if (left > 0 && !(off == 1 && cnt[1] == 1) && off != 0) return false;
    // per symbol: canonical code = first[len] + rank among equal lengths, in symbol order
    uint32_t run[16];
#pragma unroll
    for (int L = 0; L < 16; L++) run[L] = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if ((uint32_t)k * 64 >= n) break;
        const uint32_t sym = k * 64 + lane;
        const uint32_t l = lv[k];
        uint32_t rank = 0, f = 0, o = 0;
#pragma unroll
        for (int L = 1; L < 16; L++) {
            unsigned long long m = __ballot(l == (uint32_t)L);
            if (l == (uint32_t)L) {
                rank = run[L] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                f = first[L];
                o = offs[L];
            }
            run[L] += (uint32_t)__popcll(m);
        }
        if (l) {
            sorted[o + rank] = (Sorted)sym;
            if (l <= bits) {
                uint32_t c = f + rank;
                uint32_t r = __brev(c) >> (32 - l);  // codes are sent MSB first
                const Lut entry = enc(sym, l);
                for (uint32_t e = r; e < (1u << bits); e += 1u << l) lut[e] = entry;
            }
        }
    }
    return true;
}

// canonical decode, bit by bit, of the code at the front of `bits` (codes longer than the primary table; rare)
template <class Sorted>
__device__ __forceinline__ uint32_t decode_slow(unsigned long long bits, const Sorted *sorted, const uint16_t *count,
                                                uint32_t *len_out) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t len = 1; len <= 15; len++) {
        code |= (uint32_t)bits & 1u;
        bits >>= 1;
        uint32_t c = count[len];
        if (code < first + c) {
            *len_out = len;
            return sorted[index + (code - first)];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    *len_out = 0;
    return 0xFFFFFFFFu;
}

// decode one symbol serially (block headers); returns 0xFFFFFFFF on an invalid code
template <class L, class Sorted>
__device__ __forceinline__ uint32_t decode_sym(L &s, BitIn &br, const uint16_t *lut, uint32_t bits,
                                               const Sorted *sorted, const uint16_t *count, uint32_t lane) {
    unsigned long long v = peek(s, br, lane);
    uint32_t e = sgpr(lut[(uint32_t)v & ((1u << bits) - 1u)]);
    if (e) {
        br.bitpos += e & 15;
        return e >> 4;
    }
    uint32_t l = 0;
    uint32_t sym = sgpr(decode_slow(v, sorted, count, &l));
    br.bitpos += sgpr(l);
    return sym;
}

// one symbol decoded bit by bit by every lane uniformly (tokens the primary tables cannot resolve)
template <class L, class Sorted>
__device__ __forceinline__ uint32_t decode_serial(L &s, BitIn &br, const Sorted *sorted, const uint16_t *count,
                                                  uint32_t lane) {
    unsigned long long v = peek(s, br, lane);
    uint32_t l = 0;
    uint32_t sym = sgpr(decode_slow(v, sorted, count, &l));
    br.bitpos += sgpr(l);
    return sym;
}

// flush every completed 1024-element segment of the window ring to HBM (bytes: 16 B per lane; symbols: 32 B)
template <class L>
__device__ __forceinline__ void flush_segments(L &s, typename L::Elem *out, unsigned long long out_off, uint32_t &flushed,
                                               uint32_t pos, uint32_t lane) {
    using Elem = typename L::Elem;
    const bool any = flushed + 1024 <= pos;
    while (flushed + 1024 <= pos) {
        const Elem *src = s.win + ((flushed & (L::kRing - 1)) + lane * 16);
        Elem *dst = out + out_off + flushed + lane * 16;
        // the output offset is arbitrary: fall back to element stores when the destination is not 16-byte aligned
        if (!out) {
            // probing decode (block finder): nothing is kept
        } else if ((((uintptr_t)dst) & 15) == 0) {
#pragma unroll
            for (uint32_t k = 0; k < sizeof(Elem); k++)
                reinterpret_cast<uint4 *>(dst)[k] = reinterpret_cast<const uint4 *>(src)[k];
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) dst[k] = src[k];
        }
        flushed += 1024;
    }
    // later matches of this wave read these bytes back through the CU's own L1 (write-through, coherent inside a
    // workgroup): the stores only have to be complete first
    if (L::kGlobalWindow && any && out) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// LZ77 copy of `len` elements from `dist` back to position `pos`, all lanes; the source index is folded into
// [pos - dist, pos) so overlapping copies are exact.  `hi` = one past the highest position written to the ring so far
// (>= pos + len): a ring slot still holds position p iff p + RING >= hi; anything older was flushed to HBM.
template <class L>
__device__ __forceinline__ void copy_match(L &s, typename L::Elem *out, unsigned long long out_off, uint32_t pos, uint32_t len,
                                           uint32_t dist, uint32_t hi, uint32_t lane) {
    using Elem = typename L::Elem;
    constexpr bool SYM = sizeof(Elem) == 2;
    if (L::kGlobalWindow && !out) return;  // probing decode: the output never steers the decode
    // the common shape — one pass, no overlap, the whole source on one side of the ring — decided on the scalar unit
    if (dist >= len && len <= 64 && pos >= dist) {
        const uint32_t src0 = pos - dist;
        if (!L::kGlobalWindow || src0 + L::kRing >= hi) {
            if (lane < len) {
                const Elem x = s.win[(src0 + lane) & (L::kRing - 1)];
                s.win[(pos + lane) & (L::kRing - 1)] = x;
            }
            return;
        }
        if (src0 + len - 1 + L::kRing < hi) {
            if (lane < len) {
                const Elem x = __hip_atomic_load(out + out_off + src0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                s.win[(pos + lane) & (L::kRing - 1)] = x;
            }
            return;
        }
    }
    for (uint32_t i = lane; i < len; i += 64) {
        const long long src = (long long)pos - (long long)dist + (long long)(dist >= len ? i : i % dist);
        Elem x;
        if (SYM && src < 0) {  // a byte of the 32 KiB in front of this decode: named, resolved later
            x = (Elem)(0x8000u | (uint32_t)(32768 + src));
        } else if (!L::kGlobalWindow || (uint32_t)src + L::kRing >= hi) {
            x = s.win[(uint32_t)src & (L::kRing - 1)];
        } else {
            // older than the ring: flushed at least 1 KiB ago (src < hi - kRing < flushed)
            x = __hip_atomic_load(out + out_off + (uint32_t)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        s.win[(pos + i) & (L::kRing - 1)] = x;
    }
}

// a speculative step emits at most kStepOut elements + one match: with the flush lag of < 1 KiB the ring (>= 2 Ki)
// never wraps onto elements that are not flushed yet
static constexpr uint32_t kStepOut = 512;

// token kinds of the speculative decode
static constexpr uint32_t kLit = 0, kMatch = 1, kEob = 2, kSlow = 3, kBad = 4;

// One decode job of a wavefront.  Plain members: start_bit = stop_bit = 0, the window in front is empty.
struct InflateJob {
    unsigned long long comp_off;   // byte offset of the DEFLATE stream (or of the member's stream for a chunk)
    unsigned long long comp_size;  // bytes readable from comp_off
    unsigned long long out_off;    // element offset of the output in d_out
    unsigned long long out_cap;    // elements it may produce
    unsigned long long start_bit;  // first bit to decode, relative to comp_off (a block header)
    unsigned long long stop_bit;   // 0: decode to the final block; else stop at the first block boundary >= it
    unsigned int text_probe;       // != 0: a literal that is a control character (not \t \n \r) ends the decode with code 6
    unsigned int pad;
};
struct InflateJobStatus {
    unsigned int code;             // as InflateStatus
    unsigned int final_block;      // the stream's final block was decoded
    unsigned long long produced;   // elements written
    unsigned long long end_bit;    // bit after the last decoded block, relative to comp_off
};

// EMIT = 2, 4: EMIT x 64 bit offsets per speculative step, two or four windows per lane (4: +2.5 % on FASTQ and VCF
//           over 2; input that deflates 600-fold — where a step's output limit drops most of what it decoded — runs
//           at 256 instead of 380 GB/s of output, far above every stage around it);
// EMIT = 1: one window, the tokens of a step placed by a prefix sum over their output lengths (all literals in one
//           store, then the matches in order); EMIT = 0: literal runs and matches one after the other (the first
//           form).  Both kept as A/B partners (-DEXG_INFLATE_EMIT=...).
#ifndef EXG_INFLATE_EMIT
#define EXG_INFLATE_EMIT 4
#endif
template <bool SYM, uint32_t RING = kWinBytes, int EMIT = EXG_INFLATE_EMIT>
__device__ __forceinline__ void inflate_job(InflateLdsT<SYM, RING> &s, const uint8_t *__restrict__ d_comp,
                                            typename InflateLdsT<SYM, RING>::Elem *d_out, const InflateJob mb,
                                            InflateJobStatus *st_out) {
    using Elem = typename InflateLdsT<SYM, RING>::Elem;
    constexpr uint32_t kRingMask = RING - 1;
    const uint32_t lane = threadIdx.x;
    {
        BitIn br;
        const BitBase bb = bit_base(d_comp, mb.comp_off, mb.comp_size, mb.start_bit);
        br.g0 = bb.g0;
        br.limit = bb.limit;
        br.bitpos = (unsigned long long)((long long)mb.start_bit - bb.rel_bits);
        start_input(br, lane);
        __syncthreads();
        ensure(s, br, lane);

        uint32_t pos = 0, flushed = 0, err = 0;
        // positions are 32-bit: one job produces less than 4 GiB, and the bound is compared on the scalar unit
        const uint32_t cap = mb.out_cap > 0xFFFF0000ull ? 0xFFFF0000u : (uint32_t)mb.out_cap;
        bool last = false;
        const unsigned long long stop_at = mb.stop_bit ? (unsigned long long)((long long)mb.stop_bit - bb.rel_bits) : 0;
        while (!last && !err && !(stop_at && br.bitpos >= stop_at)) {
            if ((uint32_t)(br.bitpos >> 3) >= br.limit) {  // ran off the end of the input
                err = 5;
                break;
            }
            uint32_t hdr3 = getbits(s, br, 3, lane);
            last = (hdr3 & 1) != 0;
            uint32_t type = hdr3 >> 1;
            if (type == 0) {
                // stored: skip to a byte boundary, LEN / NLEN, raw bytes
                br.bitpos = (br.bitpos + 7) & ~7ull;
                uint32_t ln = getbits(s, br, 32, lane);
                uint32_t len = ln & 0xFFFFu, nlen = ln >> 16;
                if ((len ^ 0xFFFFu) != nlen) {
                    err = 1;
                    break;
                }
                if (pos + len > cap) {
                    err = 4;
                    break;
                }
                // 64 bytes per step, lane = byte
                for (uint32_t i = 0; i < len; i += 64) {
                    ensure(s, br, lane);
                    uint32_t n = len - i < 64 ? len - i : 64;
                    if (lane < n) s.win[(pos + i + lane) & kRingMask] = (Elem)s.in[((uint32_t)(br.bitpos >> 3) + lane) & (kInRing - 1)];
                    br.bitpos += 8ull * n;
                    flush_segments(s, d_out, mb.out_off, flushed, pos + i + n, lane);
                }
                pos += len;
                continue;
            }
            if (type == 3) {
                err = 1;
                break;
            }
            uint32_t nlit, ndist;
            if (type == 1) {
                for (uint32_t i = lane; i < 288; i += 64) s.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                if (lane < 32) s.lens[288 + lane] = 5;  // 32 codes make the fixed distance code complete; 30, 31 are invalid
                nlit = 288;
                ndist = 32;
            } else {
                uint32_t h14 = getbits(s, br, 14, lane);
                nlit = (h14 & 31) + 257;
                ndist = ((h14 >> 5) & 31) + 1;
                uint32_t ncode = (h14 >> 10) + 4;
                if (nlit > 286 || ndist > 30) {
                    err = 2;
                    break;
                }
                if (lane < 19) s.lens[lane] = 0;
                {
                    // 19 x 3 bits = 57 bits: one peek
                    unsigned long long v = peek(s, br, lane);
                    if (lane < ncode) s.lens[kClOrder[lane]] = (uint8_t)((v >> (3 * lane)) & 7);
                    br.bitpos += 3ull * ncode;
                }
                // the code-length code reuses the distance table storage (7-bit codes, 19 symbols)
                uint16_t *cl_lut = s.dist_lut;  // borrowed until the real tables are built
                if (!build_table(s.lens, 19, cl_lut, 7, s.dist_sorted, s.dist_count, lane, EncPlain())) {
                    err = 2;
                    break;
                }
                uint32_t idx = 0, prev = 0;
                while (idx < nlit + ndist) {
                    uint32_t sym = decode_sym(s, br, cl_lut, 7, s.dist_sorted, s.dist_count, lane);
                    if (sym == 0xFFFFFFFFu) {
                        err = 2;
                        break;
                    }
                    uint32_t rep = 1, val = sym;
                    if (sym == 16) {
                        if (idx == 0) {
                            err = 2;
                            break;
                        }
                        val = prev;
                        rep = 3 + getbits(s, br, 2, lane);
                    } else if (sym == 17) {
                        val = 0;
                        rep = 3 + getbits(s, br, 3, lane);
                    } else if (sym == 18) {
                        val = 0;
                        rep = 11 + getbits(s, br, 7, lane);
                    }
                    if (idx + rep > nlit + ndist) {
                        err = 2;
                        break;
                    }
                    for (uint32_t k = lane; k < rep; k += 64) s.lens[32 + idx + k] = (uint8_t)val;
                    idx += rep;
                    prev = val;
                }
                if (err) break;
                // move to their final places: literal/length lengths at [0, nlit), distance at [288, 288+ndist)
                uint8_t lv[5];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    uint32_t i = k * 64 + lane;
                    lv[k] = i < nlit + ndist ? s.lens[32 + i] : 0;
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    uint32_t i = k * 64 + lane;
                    if (i < nlit)
                        s.lens[i] = lv[k];
                    else if (i < nlit + ndist)
                        s.lens[288 + (i - nlit)] = lv[k];
                }
            }
            if (!build_table(s.lens + 288, ndist, s.dist_lut, kDistBits, s.dist_sorted, s.dist_count, lane, EncDist()) ||
                !build_table(s.lens, nlit, s.lit_lut, kLitBits, s.lit_sorted, s.lit_count, lane, EncLit())) {
                err = 2;
                break;
            }
            // ---- tokens of the block: 64 speculative decodes per step, then the true chain ----------------
            bool eob = false;
            while (!eob && !err) {
                if ((uint32_t)(br.bitpos >> 3) > br.limit + 8) {  // decoding the zero padding behind the input
                    err = 5;
                    break;
                }
                ensure(s, br, lane);
                // the token that would start at the first bit of v
                auto spec = [&](unsigned long long v, uint32_t &kind, uint32_t &tl, uint32_t &val) {
                    val = 0;
                    const uint32_t e = s.lit_lut[(uint32_t)v & ((1u << kLitBits) - 1u)];
                    const uint32_t l1 = e & 15;
                    if (lit_is_long(e)) {  // longer than the primary table: decoded serially IF it is a real token start
                        kind = kSlow;
                        tl = 0;
                    } else if (!(e & 0x8000u)) {
                        kind = (e & (1u << 13)) ? kBad : (e & (1u << 12)) ? kEob : kLit;
                        tl = kind == kBad ? 1 : l1;
                        val = (e >> 4) & 0xFFu;
                    } else {
                        const uint32_t lx = (e >> 4) & 7u;
                        const uint32_t len = ((e >> 7) & 0xFFu) + 3u + ((uint32_t)(v >> l1) & ((1u << lx) - 1u));
                        const uint32_t t = l1 + lx;
                        const unsigned long long v2 = v >> t;
                        const uint32_t de = s.dist_lut[(uint32_t)v2 & ((1u << kDistBits) - 1u)];
                        const uint32_t l2 = de & 15;
                        uint32_t dbase, dx;
                        dist_base_extra((de >> 4) & 31u, &dbase, &dx);
                        if (dist_is_long(de)) {
                            kind = kSlow;
                            tl = 0;
                        } else if (de & (1u << 9)) {
                            kind = kBad;
                            tl = 1;
                        } else {
                            const uint32_t dist = dbase + ((uint32_t)(v2 >> l2) & ((1u << dx) - 1u));
                            kind = kMatch;
                            tl = t + l2 + dx;  // <= 15 + 5 + 15 + 13 = 48 bits
                            val = len | (dist << 16);
                        }
                    }
                };
                uint32_t advance;
#ifdef EXG_INFLATE_PAD_S
                uint32_t advance_pad = 0;
#endif
                bool slow_token = false;
                if constexpr (EMIT >= 2) {
                    // ---- NW x 64 bit offsets per step: lane l decodes the tokens that would start at bits l, 64 + l, ...  What
                    // a step costs besides its tokens (staging check, window reads, the flush test and most of all the dependent
                    // LDS round trips window -> table -> table -> ring) is paid once per NW x 8 input bytes.
                    // The windows go through the tables side by side and without branches (every lane looks a distance up, a
                    // literal's lookup is dropped by a select): their LDS round trips overlap.  All of it is 32-bit work (a
                    // token's code and extra bits lie in the low 48 bits of its window), and what is wave-uniform — which
                    // lanes are token starts, live, kept — stays in scalar masks (mask_sel) instead of per-lane flags:
                    // the vector unit is the busiest one in this kernel
                    constexpr int NW = EMIT;
                    uint32_t lo[NW], hi[NW], e[NW], de[NW], tt[NW], v2[NW];
                    peekn_at<NW>(s, (uint32_t)br.bitpos + lane, lo, hi);
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        e[k] = s.lit_lut[lo[k] & ((1u << kLitBits) - 1u)];
                        asm("" : "+v"(e[k]));  // (a 32-bit value from here on: knowing it came from 16 bits, the compiler tests bit 15 as a
                                               // 16-bit sign and then masks every use to 16 bits again — four instructions per step)
                    }
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        tt[k] = (e[k] & 15u) + ((e[k] >> 4) & 7u);              // code + extra bits of a length: <= 20
                        v2[k] = __builtin_amdgcn_alignbit(hi[k], lo[k], tt[k]);  // bits [t, t + 32): distance code + extra bits <= 28
                        de[k] = s.dist_lut[v2[k] & ((1u << kDistBits) - 1u)];
                    }
                    uint32_t tl[NW], dxs[NW], olen1[NW];
                    unsigned long long len_mask[NW], stop_mask[NW];
                    const uint32_t one = 1;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        const uint32_t l1 = e[k] & 15u, lx = (e[k] >> 4) & 7u, l2 = de[k] & 15u;
                        // distance symbol 2 h + b: h = 0 -> base b + 1, no extra bits; else base ((2 | b) << (h - 1)) + 1
                        // (the distance itself is decoded where a match is copied: most windows have none)
                        const uint32_t dx = __builtin_elementwise_sub_sat((de[k] >> 5) & 15u, 1u);
                        dxs[k] = dx;
                        const uint32_t len = ((e[k] >> 7) & 0xFFu) + __builtin_amdgcn_ubfe(lo[k], l1, lx) + 3u;
                        // (one compare; the selects take its mask as their scalar operand — left to the compiler the
                        // predicate is evaluated three times and the length / distance arithmetic is put behind a branch)
                        len_mask[k] = __ballot(e[k] > 0x7FFFu);  // (a 32-bit compare: the 16-bit sign test makes the compiler mask every entry to 16 bits again)
                        olen1[k] = mask_sel(len_mask[k], one, len);
                        // every lane hops at least one bit (kLitLong carries a length of one): the walk runs on through a token it
                        // cannot use, what lies behind the first such token is dropped below
                        tl[k] = mask_sel(len_mask[k], l1, tt[k] + l2 + dx);  // 1 (kLitLong: l1 = 1) .. 15 + 5 + 15 + 13 = 48 bits
                        // tokens that end the step: end-of-block, invalid symbols (bits 12 / 13 of a literal entry, bit 9 of a
                        // distance entry) and codes longer than the tables (kLitLong / a distance entry of 0).  Valid entries are 1 .. 0xFFF / 1 .. 0x1FF
                        const uint32_t y = mask_sel(len_mask[k], e[k], de[k] << 3);
                        stop_mask[k] = __ballot(y - 1u >= 0xFFFu);
                    }
#ifdef EXG_INFLATE_PAD_V  // development probe: what a step costs when the vector unit has this many more instructions to issue
#pragma unroll
                    for (int i = 0; i < EXG_INFLATE_PAD_V; i++) asm volatile("v_or_b32 %0, %0, %0" : "+v"(tl[0]));
#endif
#ifdef EXG_INFLATE_PAD_S  // ... and the scalar unit
                    {
                        uint32_t pad_s = sgpr(advance_pad);
#pragma unroll
                        for (int i = 0; i < EXG_INFLATE_PAD_S; i++) asm volatile("s_or_b32 %0, %0, %0" : "+s"(pad_s) : : "scc");
                        advance_pad = pad_s;
                    }
#endif
                    // The real chain from offset 0, window after window.  The walk is the scalar unit's main load (~15 tokens
                    // per window), so its loop is written out: the position is kept as cur - 64 (mod 2^32; bit set and lane
                    // select use the low six bits), so that the add's carry is the exit test — three scalar instructions and
                    // the readlane per token
                    unsigned long long marks[NW];
                    uint32_t cur = 0;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        marks[k] = 0;
                        cur -= 64;
                        if ((int32_t)cur < 0) {
                            uint32_t t;
                            asm volatile(
                                "1:\n\t"
                                "s_bitset1_b64 %[m], %[c]\n\t"
                                "v_readlane_b32 %[t], %[hop], %[c]\n\t"
                                "s_add_u32 %[c], %[c], %[t]\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                "s_bitset1_b64 %[m], %[c]\n\t"
                                "v_readlane_b32 %[t], %[hop], %[c]\n\t"
                                "s_add_u32 %[c], %[c], %[t]\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                "s_bitset1_b64 %[m], %[c]\n\t"
                                "v_readlane_b32 %[t], %[hop], %[c]\n\t"
                                "s_add_u32 %[c], %[c], %[t]\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                "s_bitset1_b64 %[m], %[c]\n\t"
                                "v_readlane_b32 %[t], %[hop], %[c]\n\t"
                                "s_add_u32 %[c], %[c], %[t]\n\t"
                                "s_cbranch_scc0 1b\n"
                                "2:"
                                : [m] "+s"(marks[k]), [c] "+s"(cur), [t] "=&s"(t)
                                : [hop] "v"(tl[k])
                                : "scc");
                        }
                    }
                    advance = cur + 64u * NW;
                    // ---- placement by prefix sum over all windows.  The common step has no token that ends it and emits less
                    // than its limit: then every token start is live and kept, and the masks are the walk's marks as they are
                    uint32_t stop_pos = 64u * NW;
                    unsigned long long live[NW], any_stop = 0;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        live[k] = marks[k];
                        any_stop |= marks[k] & stop_mask[k];
                    }
                    if (any_stop) {
#pragma unroll
                        for (int k = NW - 1; k >= 0; k--) {
                            const unsigned long long m_stop = marks[k] & stop_mask[k];
                            if (m_stop) stop_pos = 64u * k + (uint32_t)__ffsll((long long)m_stop) - 1;
                        }
#pragma unroll
                        for (int k = 0; k < NW; k++) {
                            // literals and matches in front of the first token that ends the step
                            const uint32_t n = max(stop_pos, 64u * k) - 64u * k;
                            live[k] &= n >= 64 ? ~0ull : (1ull << n) - 1ull;
                        }
                    }
                    uint32_t excl[NW], olen[NW], incl[NW], carry = 0, total = 0;
                    unsigned long long keep[NW];
#pragma unroll
                    for (int k = 0; k < NW; k++) olen[k] = mask_sel0(live[k], olen1[k]);
#ifndef EXG_INFLATE_SCAN_EACH
                    // two windows per scan: a window's sum is at most 64 x 258 < 2^16, so two of them ride in one register through
                    // the six DPP steps (eight vector instructions less per step; the scans run side by side: a DPP step waits
                    // for the one before)
                    if constexpr (NW % 2 == 0) {
                        uint32_t pk[NW / 2];
#pragma unroll
                        for (int k = 0; k < NW / 2; k++) pk[k] = olen[2 * k] | (olen[2 * k + 1] << 16);
#pragma unroll
                        for (int k = 0; k < NW / 2; k++) pk[k] = wave_incl_sum_dpp(pk[k]);
#pragma unroll
                        for (int k = 0; k < NW / 2; k++) {
                            incl[2 * k] = pk[k] & 0xFFFFu;
                            incl[2 * k + 1] = pk[k] >> 16;
                        }
                    } else
#endif
                    {
#pragma unroll
                        for (int k = 0; k < NW; k++) incl[k] = wave_incl_sum_dpp(olen[k]);
                    }
                    // (excl = where a token's output goes: the output position itself, pos + the tokens in front — the window's carry
                    // starts at pos, so that the stores and the matches need no add of their own)
                    carry = pos;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        excl[k] = incl[k] - olen[k] + carry;
                        carry += __builtin_amdgcn_readlane(incl[k], 63);
                        keep[k] = live[k];
                    }
                    carry -= pos;
                    total = carry;
                    uint32_t drop_pos = 64u * NW;  // the first token pushed to the next step
                    if (carry > kStepOut) {
                        total = 0;
#pragma unroll
                        for (int k = NW - 1; k >= 0; k--) {
                            keep[k] = live[k] & __ballot(excl[k] - pos < kStepOut);
                            const unsigned long long m_drop = live[k] & ~keep[k];
                            if (m_drop) drop_pos = 64u * k + (uint32_t)__ffsll((long long)m_drop) - 1;
                        }
#pragma unroll
                        for (int k = 0; k < NW; k++)
                            if (keep[k]) {
                                const uint32_t l = 63 - __clzll((long long)keep[k]);
                                total = __builtin_amdgcn_readlane(excl[k], l) - pos + __builtin_amdgcn_readlane(olen[k], l);
                            }
                    }
                    if (pos + total > cap) {
                        err = 4;
                        break;
                    }
                    if (mb.text_probe) {
                        bool ctl = false;
#pragma unroll
                        for (int k = 0; k < NW; k++)
                        {
                            const uint32_t c = (e[k] >> 4) & 0xFFu;
                            ctl |= (((keep[k] & ~len_mask[k]) >> lane) & 1ull) && (c < 9u || (c > 13u && c < 32u) || c == 127u);
                        }
                        if (__ballot(ctl)) {
                            err = 6;
                            break;
                        }
                    }
                    const uint32_t hi_pos = pos + total;
                    // all literals in one store per window (the other lanes write to a spare element behind the ring)
#pragma unroll
                    for (int k = 0; k < NW; k++)
                        s.win[mask_sel(keep[k] & ~len_mask[k], RING, excl[k] & kRingMask)] = (Elem)((e[k] >> 4) & 0xFFu);
                    // ---- matches.  Most are short and reach far back (FASTQ / VCF / FASTA at zlib level 6: nine in ten are <= 8 bytes,
                    // seven in ten come from behind the 2 KiB ring): each one handled in order costs ~17 scalar instructions and,
                    // for a far source, an L1 / L2 round trip the wave sits out.  So the INDEPENDENT short ones go first, eight per
                    // pass with eight lanes each: a match of <= 16 bytes whose source ends in front of the step's first match (F)
                    // reads only bytes that are final once the literals are stored — earlier steps' output and this step's
                    // literals — whatever the order.  Their descriptors (rank = popcount over the windows) go through fast_slot;
                    // one ring read or one HBM read per pass serves all eight (two for the lanes of a match longer than eight).  The rest — long, overlapping, straddling the
                    // ring's edge, fed by a match of this step, or beyond the eight slots — follow in order as before, and may
                    // read what the fast ones wrote (they come later in the wave's LDS order).
                    const bool copies = !(InflateLdsT<SYM, RING>::kGlobalWindow && !d_out);  // a probing decode copies nothing
                    unsigned long long m_match[NW], any_match = 0;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        m_match[k] = keep[k] & len_mask[k];
                        any_match |= m_match[k];
                    }
                    if (any_match) {
                        static_assert((kFastSlots & (kFastSlots - 1)) == 0 && kFastSlots <= 8, "eight lanes per slot");
                        uint32_t n_fast = 0, first_dest = 0;
                        bool have_first = false;
                        // the pending fast matches, eight lanes each (lane = 8 j + i: byte i of slot j)
                        auto copy_fast = [&]() {
                            const uint32_t j = lane >> 3, i = lane & 7u;
                            const uint32_t d = s.fast_slot[j & (uint32_t)(kFastSlots - 1)];
                            const uint32_t flen = ((d >> 10) & 15u) + 3u, fdist = ((d >> 14) & 0x7FFFu) + 1u;
                            const bool mine = j < n_fast && i < flen, mine2 = kFastLen > 8 && j < n_fast && i + 8u < flen;
                            const uint32_t fdest = pos + (d & 1023u) + i, fsrc = fdest - fdist;
                            const bool two = kFastLen > 8 && __ballot(mine2) != 0;  // (wave uniform: most passes have no match longer than eight)
                            if (mine) {
                                const bool far = InflateLdsT<SYM, RING>::kGlobalWindow && (d >> 29);
                                Elem x, y = 0;
                                if (far)
                                    x = __hip_atomic_load(d_out + mb.out_off + fsrc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                else
                                    x = s.win[fsrc & kRingMask];
                                if (two && mine2) {
                                    if (far)
                                        y = __hip_atomic_load(d_out + mb.out_off + fsrc + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    else
                                        y = s.win[(fsrc + 8u) & kRingMask];
                                }
                                s.win[fdest & kRingMask] = x;
                                if (two && mine2) s.win[(fdest + 8u) & kRingMask] = y;
                            }
                            n_fast = 0;
                        };
                        // the others of one window, in order (what can be decided per match is decided for all of them at once:
                        // the scalar loop reads a descriptor)
                        auto copy_in_order = [&](unsigned long long m, uint32_t len_l, uint32_t dist_k, uint32_t dest_k) {
                            const uint32_t src0_l = dest_k - dist_k;
                            uint32_t cls = 2;  // 0: one pass out of the ring, 1: one pass out of HBM, 2: the general copy
                            if (dist_k >= len_l && len_l <= 64u && dest_k >= dist_k) {
                                if (!InflateLdsT<SYM, RING>::kGlobalWindow || src0_l + RING >= hi_pos)
                                    cls = 0;
                                else if (src0_l + len_l - 1 + RING < hi_pos)
                                    cls = 1;
                            }
                            const uint32_t desc_l = cls | (len_l << 2);
                            while (m) {
                                const uint32_t l = (uint32_t)__ffsll((long long)m) - 1;
                                m &= m - 1;
                                const uint32_t desc = __builtin_amdgcn_readlane(desc_l, l);
                                const uint32_t dest = __builtin_amdgcn_readlane(dest_k, l);
                                const uint32_t mlen = desc >> 2;
                                if ((desc & 3u) == 0) {
                                    const uint32_t src0 = __builtin_amdgcn_readlane(src0_l, l);
                                    if (lane < mlen) {
                                        const Elem x = s.win[(src0 + lane) & kRingMask];
                                        s.win[(dest + lane) & kRingMask] = x;
                                    }
                                } else if ((desc & 3u) == 1) {
                                    const uint32_t src0 = __builtin_amdgcn_readlane(src0_l, l);
                                    if (lane < mlen) {
                                        const Elem x = __hip_atomic_load(d_out + mb.out_off + src0 + lane, __ATOMIC_RELAXED,
                                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                                        s.win[(dest + lane) & kRingMask] = x;
                                    }
                                } else {
                                    copy_match(s, d_out, mb.out_off, dest, mlen, __builtin_amdgcn_readlane(dist_k, l), hi_pos, lane);
                                }
                            }
                        };
#pragma unroll
                        for (int k = 0; k < NW; k++) {
                            if (m_match[k] && !err) {
                                const uint32_t h = (de[k] >> 5) & 15u, bb = (de[k] >> 4) & 1u;
                                const uint32_t dist_k = (((min(h, 1u) << 1) | bb) << dxs[k]) + __builtin_amdgcn_ubfe(v2[k], de[k] & 15u, dxs[k]) + 1u;
                                const uint32_t dest_k = excl[k], len_l = olen1[k];
                                if (!SYM && (m_match[k] & __ballot(dist_k > dest_k))) {
                                    err = 3;
                                } else if (copies) {
                                    unsigned long long m_slow = m_match[k];
                                    if (EXG_INFLATE_FAST_MATCHES) {
                                        if (!have_first) {
                                            first_dest = __builtin_amdgcn_readlane(dest_k, (uint32_t)__ffsll((long long)m_match[k]) - 1);
                                            have_first = true;
                                        }
                                        const uint32_t src0_l = dest_k - dist_k;
                                        // the ring holds [hi_pos - RING, hi_pos): a source is read from it, or from HBM when all of it is
                                        // older; one that straddles the edge takes the general copy.  edge = ring's first position - src0
                                        const uint32_t edge = hi_pos - RING - src0_l;  // <= 0 (as int): in the ring; >= len: flushed
                                        const bool far_src = InflateLdsT<SYM, RING>::kGlobalWindow && (int32_t)edge >= (int32_t)len_l;
                                        const bool straddles = InflateLdsT<SYM, RING>::kGlobalWindow && edge - 1u < len_l - 1u;
                                        const bool ok = len_l <= kFastLen && (!SYM || dist_k <= dest_k) && src0_l + len_l <= first_dest && !straddles;
                                        unsigned long long m_fast = m_match[k] & __ballot(ok);
                                        m_slow &= ~m_fast;
                                        // excl < 1024 (a step emits <= 512 elements + one match), len - 3 in 4 bits, dist - 1 in 15
                                        const uint32_t packed = (excl[k] - pos) | ((len_l - 3u) << 10) | ((dist_k - 1u) << 14) | (far_src ? 1u << 29 : 0u);
                                        while (m_fast) {
                                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_fast >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_fast, n_fast));
                                            const unsigned long long take = m_fast & __ballot(rank < (uint32_t)kFastSlots);
                                            s.fast_slot[mask_sel(take, (uint32_t)kFastSlots, rank)] = packed;
                                            n_fast += (uint32_t)__popcll(take);
                                            m_fast &= ~take;
                                            if (m_fast) copy_fast();  // the slots are full: these go, the others follow
                                        }
                                    }
                                    if (m_slow) {
                                        if (n_fast) copy_fast();  // (one of them may feed a match of this window)
                                        copy_in_order(m_slow, len_l, dist_k, dest_k);
                                    }
                                }
                            }
                        }
                        if (n_fast && !err) copy_fast();
                    }
                    if (err) break;
                    pos = hi_pos;
                    if (flushed + 1024 <= pos) flush_segments(s, d_out, mb.out_off, flushed, pos, lane);
                    if (drop_pos < 64u * NW) {
                        advance = drop_pos;
                    } else if (stop_pos < 64u * NW) {
                        // what ended the step: the entries of that token
                        const uint32_t sl = stop_pos & 63u, sk = stop_pos >> 6;
                        uint32_t e_s = 0, de_s = 0;
#pragma unroll
                        for (int k = 0; k < NW; k++)
                            if (sk == (uint32_t)k) {
                                e_s = __builtin_amdgcn_readlane(e[k], sl);
                                de_s = __builtin_amdgcn_readlane(de[k], sl);
                            }
                        if (e_s & 0x8000u ? dist_is_long(de_s) : lit_is_long(e_s)) {
                            advance = stop_pos;  // a code longer than the tables: decoded by every lane uniformly below
                            slow_token = true;
                        } else if (!(e_s & 0x8000u) && (e_s & 0x3000u) == 0x1000u) {
                            advance = stop_pos + (e_s & 15u);
                            eob = true;
                        } else {
                            err = 3;
                            break;
                        }
                    }
                } else {
                // lane l decodes the token that would start at bit bitpos + l
                uint32_t kind, tl, val;
                spec(peek_at(s, br.bitpos + lane), kind, tl, val);
                // the real chain from offset 0: one readlane per token marks the token starts ...
                unsigned long long marks = 0;
                uint32_t cur = 0;
                if constexpr (EMIT == 1) {
                    // (a token for the serial decoder hops out of the window: the loop body is six scalar instructions,
                    // and the scalar unit — one per CU, shared by 20 wavefronts — is the busiest one in this kernel)
                    const uint32_t hop = kind == kSlow ? 64u : tl;
                    do {
                        marks |= 1ull << cur;
                        cur += __builtin_amdgcn_readlane(hop, cur);
                    } while (cur < 64);
                } else {
                    while (cur < 64) {
                        uint32_t t = __builtin_amdgcn_readlane(tl, cur);
                        if (t == 0) {  // kSlow: the window ends in front of it
                            slow_token = true;
                            break;
                        }
                        marks |= 1ull << cur;
                        cur += t;
                    }
                }
                advance = cur;
                if constexpr (EMIT == 1) {
                    // ---- placement by prefix sum ---------------------------------------------------------------
                    const bool real = (marks >> lane) & 1ull;
                    const unsigned long long m_stop = __ballot(real && (kind == kEob || kind == kBad || kind == kSlow));
                    const uint32_t stop_lane = m_stop ? (uint32_t)__ffsll((long long)m_stop) - 1 : 64u;
                    const bool live = real && lane < stop_lane;  // literals and matches in front of an end-of-block
                    const uint32_t olen = live ? (kind == kLit ? 1u : (val & 0xFFFFu)) : 0u;
                    const uint32_t incl = wave_incl_sum_dpp(olen);
                    const uint32_t excl = incl - olen;
                    const bool keep = live && excl < kStepOut;
                    const unsigned long long m_keep = __ballot(keep);
                    const unsigned long long m_drop = __ballot(live) & ~m_keep;  // pushed to the next step
                    const uint32_t total = m_keep ? __builtin_amdgcn_readlane(incl, 63 - __clzll((long long)m_keep)) : 0u;
                    if (pos + total > cap) {
                        err = 4;
                        break;
                    }
                    if (mb.text_probe) {
                        const bool ctl = keep && kind == kLit && (val < 9u || (val > 13u && val < 32u) || val == 127u);
                        if (__ballot(ctl)) {
                            err = 6;
                            break;
                        }
                    }
                    const uint32_t hi = pos + total;
                    if (keep && kind == kLit) s.win[(pos + excl) & kRingMask] = (Elem)val;
                    // matches, in order.  What can be decided per token is decided on the VALU for all tokens at once
                    // (class of the copy, source, bounds): the scalar loop only reads a descriptor per match.  No flush
                    // inside a step: the ring holds [flushed, hi) (< 1 KiB of lag + kStepOut + one match), and what a
                    // copy reads from HBM (older than hi - RING) was flushed before the step began.
                    const bool is_m = keep && kind == kMatch;
                    const uint32_t dest_l = pos + excl, len_l = val & 0xFFFFu, dist_l = val >> 16;
                    if (!SYM && __ballot(is_m && dist_l > dest_l)) {
                        err = 3;
                        break;
                    }
                    if (!(InflateLdsT<SYM, RING>::kGlobalWindow && !d_out)) {  // a probing decode copies nothing
                        const uint32_t src0_l = dest_l - dist_l;
                        uint32_t cls = 2;  // 0: one pass out of the ring, 1: one pass out of HBM, 2: the general copy
                        if (dist_l >= len_l && len_l <= 64u && dest_l >= dist_l) {
                            if (!InflateLdsT<SYM, RING>::kGlobalWindow || src0_l + RING >= hi)
                                cls = 0;
                            else if (src0_l + len_l - 1 + RING < hi)
                                cls = 1;
                        }
                        const uint32_t desc_l = cls | (len_l << 2);
                        unsigned long long m_match = __ballot(is_m);
                        while (m_match) {
                            const uint32_t l = (uint32_t)__ffsll((long long)m_match) - 1;
                            m_match &= m_match - 1;
                            const uint32_t desc = __builtin_amdgcn_readlane(desc_l, l);
                            const uint32_t dest = __builtin_amdgcn_readlane(dest_l, l);
                            const uint32_t len = desc >> 2;
                            if ((desc & 3u) == 0) {
                                const uint32_t src0 = __builtin_amdgcn_readlane(src0_l, l);
                                if (lane < len) {
                                    const Elem x = s.win[(src0 + lane) & kRingMask];
                                    s.win[(dest + lane) & kRingMask] = x;
                                }
                            } else if ((desc & 3u) == 1) {
                                const uint32_t src0 = __builtin_amdgcn_readlane(src0_l, l);
                                if (lane < len) {
                                    const Elem x = __hip_atomic_load(d_out + mb.out_off + src0 + lane, __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                                    s.win[(dest + lane) & kRingMask] = x;
                                }
                            } else {
                                copy_match(s, d_out, mb.out_off, dest, len, __builtin_amdgcn_readlane(dist_l, l), hi, lane);
                            }
                        }
                    }
                    if (err) break;
                    pos = hi;
                    if (flushed + 1024 <= pos) flush_segments(s, d_out, mb.out_off, flushed, pos, lane);
                    if (m_drop) {
                        advance = (uint32_t)__ffsll((long long)m_drop) - 1;
                    } else if (stop_lane < 64) {
                        const uint32_t k = __builtin_amdgcn_readlane(kind, stop_lane);
                        if (k == kEob) {
                            advance = stop_lane + __builtin_amdgcn_readlane(tl, stop_lane);
                            eob = true;
                        } else if (k == kSlow) {
                            advance = stop_lane;  // decoded by every lane uniformly below
                            slow_token = true;
                        } else {
                            err = 3;
                            break;
                        }
                    }
                } else {
                // ... then runs of literals go out in ONE step (rank = popcount of the marks below the lane);
                // only matches / end-of-block / bad codes are handled one at a time, in order
                const unsigned long long m_lit = __ballot(kind == kLit) & marks;
                unsigned long long m_other = marks & ~m_lit;
                uint32_t from = 0;
                for (;;) {
                    const uint32_t upto = m_other ? (uint32_t)__ffsll((long long)m_other) - 1 : 64u;
                    unsigned long long seg = m_lit;
                    if (upto < 64) seg &= (1ull << upto) - 1ull;
                    if (from >= 64)
                        seg = 0;
                    else if (from)
                        seg &= ~((1ull << from) - 1ull);
                    const uint32_t n = (uint32_t)__popcll(seg);
                    if (n) {
                        if (mb.text_probe) {
                            const bool ctl = ((seg >> lane) & 1ull) && (val < 9u || (val > 13u && val < 32u) || val == 127u);
                            if (__ballot(ctl)) {
                                err = 6;
                                break;
                            }
                        }
                        if (pos + n > cap) {
                            err = 4;
                            break;
                        }
                        uint32_t rank = (uint32_t)__popcll(seg & ((1ull << lane) - 1ull));
                        if ((seg >> lane) & 1ull) s.win[(pos + rank) & kRingMask] = (Elem)val;
                        uint32_t np = pos + n;
                        if ((np >> 10) != (pos >> 10)) flush_segments(s, d_out, mb.out_off, flushed, np, lane);
                        pos = np;
                    }
                    if (upto == 64) break;
                    const uint32_t k = __builtin_amdgcn_readlane(kind, upto);
                    const uint32_t x = __builtin_amdgcn_readlane(val, upto);
                    if (k == kMatch) {
                        uint32_t len = x & 0xFFFFu, dist = x >> 16;
                        if ((!SYM && dist > pos) || pos + len > cap) {
                            err = (!SYM && dist > pos) ? 3 : 4;
                            break;
                        }
                        copy_match(s, d_out, mb.out_off, pos, len, dist, pos + len, lane);
                        uint32_t np = pos + len;
                        if ((np >> 10) != (pos >> 10)) flush_segments(s, d_out, mb.out_off, flushed, np, lane);
                        pos = np;
                    } else if (k == kEob) {
                        advance = upto + __builtin_amdgcn_readlane(tl, upto);
                        eob = true;
                        break;
                    } else {
                        err = 3;
                        break;
                    }
                    m_other &= m_other - 1;
                    from = upto + 1;
                }
                }
                }
                br.bitpos += advance;
                if (slow_token && !eob && !err) {
                    // one token with a code longer than the primary tables, decoded by every lane uniformly
                    uint32_t sym = decode_serial(s, br, s.lit_sorted, s.lit_count, lane);
                    if (sym < 256) {
                        if (pos >= cap) {
                            err = 4;
                        } else {
                            if (lane == 0) s.win[pos & kRingMask] = (Elem)sym;
                            pos++;
                            if ((pos & 1023) == 0) flush_segments(s, d_out, mb.out_off, flushed, pos, lane);
                        }
                    } else if (sym == 256) {
                        eob = true;
                    } else if (sym > 285) {
                        err = 3;
                    } else {
                        sym -= 257;
                        uint32_t len = kLenBase[sym] + getbits(s, br, kLenExtra[sym], lane);
                        uint32_t ds = decode_serial(s, br, s.dist_sorted, s.dist_count, lane);
                        if (ds > 29) {
                            err = 3;
                        } else {
                            uint32_t dist = kDistBase[ds] + getbits(s, br, kDistExtra[ds], lane);
                            if ((!SYM && dist > pos) || pos + len > cap) {
                                err = (!SYM && dist > pos) ? 3 : 4;
                            } else {
                                copy_match(s, d_out, mb.out_off, pos, len, dist, pos + len, lane);
                                uint32_t np = pos + len;
                                if ((np >> 10) != (pos >> 10)) flush_segments(s, d_out, mb.out_off, flushed, np, lane);
                                pos = np;
                            }
                        }
                    }
                }
            }
        }
        // tail: the bytes after the last full segment
        if (!err && d_out) {
            for (uint32_t i = flushed + lane; i < pos; i += 64) d_out[mb.out_off + i] = s.win[i & kRingMask];
        }
        if (lane == 0) {
            InflateJobStatus st;
            st.code = err;
            st.final_block = last ? 1u : 0u;
            st.produced = pos;
            st.end_bit = (unsigned long long)((long long)br.bitpos + bb.rel_bits);
            *st_out = st;
        }
        __syncthreads();
    }
}


}  // namespace exg
