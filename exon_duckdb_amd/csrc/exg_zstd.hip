// exg_zstd.hip — Zstandard (RFC 8878) frames decoded on the device (gfx950).  See exg_zstd.hpp for what it replaces and
// why the stages are cut where they are.  Stages, all launched back to back on one stream:
//
//   k_zst_literals   one wavefront per EIGHT compressed blocks (since round 4: a wavefront per block had one or four working
//                    lanes per instruction on every SIMD — the stage was issue-bound across the chip): the literals
//                    section.  Raw / RLE literals are wave-wide copies, block after block; Huffman literals: the tree
//                    description (direct 4-bit weights, or FSE-coded weights) becomes a single-lookup table of 2^log
//                    entries in LDS, built by the block's first lane; the block's four backward bitstreams are four lanes
//                    (a block has four independent bitstreams — all the parallelism the format offers inside a block),
//                    each reading its stream from a 128-byte LDS ring of its own.
//   k_zst_sequences  one wavefront per EIGHT compressed blocks, a lane per block: the three FSE tables (predefined / RLE /
//                    described / repeated from an earlier block, whose description is simply parsed again) go to LDS as
//                    16-bit entries (2.5 KiB per block); the block's lane walks the backward bitstream — every field read
//                    straight from the lane's LDS ring — and writes (literal length, match length, offset) per sequence.
//                    Offsets that are repeat codes cannot be resolved without the previous blocks' history — they are
//                    emitted symbolically ("incoming slot j minus k"), and the block's effect on the history likewise.
//   k_zst_scan       one wavefront over all blocks in file order: output offsets (prefix sum) and the repeat-offset
//                    history at every block's start (composition of the blocks' symbolic updates).
//   k_zst_exec       one wavefront per chunk (a run of blocks of one frame): literal runs and LZ77 copies into a ring of
//                    the newest elements in LDS; completed 1 KiB segments leave for HBM with 16 B / lane stores, and a
//                    match that reaches behind the ring reads what the wave itself flushed earlier (through the CU's own
//                    L1: a workgroup-scope release after the flush is all the ordering it takes — the k_inflate design).
//   k_zst_xxh64      the frame's content checksum (XXH64, a serial recurrence: four lanes carry its four accumulators).
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <cmath>

#include "exg_common.hpp"
#include "exg_reader.hpp"
#include "exg_xxh64.hpp"
#include "exg_zstd.hpp"

namespace exg {
namespace zst {

// ---- bit access ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ld64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);  // one global_load_dwordx2: unaligned access is on for amdhsa
    return v;
}
__device__ __forceinline__ void st64(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

// 8 bytes of a stream starting at byte idx (may be negative: bytes in front of the stream read as zero, which is what
// an over-read backward bitstream delivers, RFC 8878 4.1)
__device__ __forceinline__ uint64_t ld64_guard(const uint8_t *base, long long idx) {
    if (idx >= 0) return ld64(base + idx);
    if (idx <= -8) return 0;
    uint64_t v = 0;
    for (int j = (int)-idx; j < 8; j++) v |= (uint64_t)base[idx + j] << (8 * j);
    return v;
}

// backward bitstream: the last byte holds the end mark (its highest set bit); bits are taken from there downwards
struct BitsRev {
    const uint8_t *base;
    long long idx;  // stream byte the container starts at
    uint64_t c;     // 8 stream bytes, little endian: the next bit to read is bit 63 - used
    uint32_t used;
    __device__ __forceinline__ bool init(const uint8_t *p, long long len) {
        base = p;
        if (len <= 0) return false;
        if (p[len - 1] == 0) return false;
        idx = len - 8;
        c = ld64_guard(p, idx);
        used = (uint32_t)__clzll((long long)c) + 1;
        return true;
    }
    __device__ __forceinline__ void refill() {
        idx -= used >> 3;
        used &= 7;
        c = ld64_guard(base, idx);
    }
    __device__ __forceinline__ void need(uint32_t n) {
        if (used + n > 64) refill();
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return n ? (uint32_t)((c << used) >> (64 - n)) : 0u; }
    __device__ __forceinline__ uint32_t read(uint32_t n) {  // need(n) first
        const uint32_t v = peek(n);
        used += n;
        return v;
    }
    __device__ __forceinline__ long long left() const { return idx * 8 + 64 - (long long)used; }
};

__device__ __forceinline__ uint32_t fwd_peek(const uint8_t *p, uint32_t bit, uint32_t n) {
    return (uint32_t)(ld64(p + (bit >> 3)) >> (bit & 7)) & ((1u << n) - 1);
}
__device__ __forceinline__ int hb32(uint32_t v) { return 31 - __clz((int)v); }

// ---- FSE ----------------------------------------------------------------------------------------------------------
// table entry: symbol | nbBits << 8 | newStateBase << 16
// RFC 8878 4.1.1: normalised counts from a forward bitstream.  Returns the bytes consumed or -1.  One lane.
__device__ __forceinline__ int fse_read_norm(const uint8_t *p, int len, int max_log, int max_sym, int16_t *norm, int *n_sym, int *log_out) {
    if (len < 1) return -1;
    // forward bits through a register container: one global load per ~57 bits, not one per field
    uint64_t c = ld64(p);
    uint32_t cbit = 0;  // c = the 64 bits from bit cbit (a multiple of 8) on
    auto peek = [&](uint32_t bit, uint32_t n) -> uint32_t {
        if (bit + n > cbit + 64) {
            cbit = bit & ~7u;
            c = ld64(p + (cbit >> 3));
        }
        return (uint32_t)(c >> (bit - cbit)) & ((1u << n) - 1);
    };
    uint32_t bit = 0;
    const int log = 5 + (int)peek(bit, 4);
    bit += 4;
    if (log > max_log) return -1;
    int remaining = 1 << log, s = 0;
    while (remaining > 0 && s <= max_sym) {
        const int nb = hb32((uint32_t)remaining + 1) + 1;
        if ((int)((bit + nb + 7) >> 3) > len + 4) return -1;
        uint32_t val = peek(bit, nb);
        const uint32_t lower = (1u << (nb - 1)) - 1, thresh = (1u << nb) - 1 - ((uint32_t)remaining + 1);
        if ((val & lower) < thresh) {
            bit += nb - 1;
            val &= lower;
        } else {
            bit += nb;
            if (val > lower) val -= thresh;
        }
        const int proba = (int)val - 1;
        remaining -= proba < 0 ? -proba : proba;
        norm[s++] = (int16_t)proba;
        if (proba == 0) {
            for (;;) {
                const int rep = (int)peek(bit, 2);
                bit += 2;
                for (int i = 0; i < rep && s <= max_sym; i++) norm[s++] = 0;
                if (rep != 3) break;
                if ((int)(bit >> 3) > len + 4) return -1;
            }
        }
    }
    if (remaining != 0 || s > max_sym + 1) return -1;
    const int bytes = (int)((bit + 7) >> 3);
    if (bytes > len) return -1;
    *n_sym = s;
    *log_out = log;
    return bytes;
}
// decoding table from normalised counts (one lane; `next` is scratch of 64 entries)
__device__ __forceinline__ int fse_build(uint32_t *tab, const int16_t *norm, int n_sym, int log, uint16_t *next) {
    const int size = 1 << log;
    int high = size;
    for (int s = 0; s < n_sym; s++)
        if (norm[s] == -1) {
            tab[--high] = (uint32_t)s;
            next[s] = 1;
        }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < n_sym; s++) {
        const int c = norm[s];
        if (c <= 0) continue;
        next[s] = (uint16_t)c;
        for (int i = 0; i < c; i++) {
            tab[pos] = (uint32_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos >= high);
        }
    }
    if (pos != 0) return -1;
    for (int i = 0; i < size; i++) {
        const uint32_t s = tab[i];
        const uint32_t x = next[s]++;
        const uint32_t nb = (uint32_t)(log - hb32(x));
        tab[i] = s | (nb << 8) | (((x << nb) - (uint32_t)size) << 16);
    }
    return 0;
}

// ---- literals -------------------------------------------------------------------------------------------------------
// A block's Huffman literals are four independent backward bitstreams: four lanes.  Like the sequences (below), a wavefront
// per block spends the chip's issue slots on four-lane instructions (5.8 ms per 1 GiB round, and the sequences kernel
// beside it starved of issue slots), so a wavefront takes kLitG blocks, four lanes each, with each block's single-lookup
// table (4 KiB) in LDS; a lane reads its stream through a 128-byte ring of its own in LDS (see SeqLds), which lies over
// the scratch the table was built with.
static constexpr int kLitG = 8;  // blocks per wavefront
struct LitBlk {
    uint16_t tab[2048];  // symbol | nbBits << 8, indexed by the next `log` bits
    union {
        struct {
            uint32_t fse[64];  // weights' FSE table (accuracy log <= 6)
            int16_t norm[64];
            uint16_t next[64];
            uint8_t w[256];
            uint32_t rank_start[16];
        };
        uint32_t ring[4][2 * 64 / 4 + 1];  // a stream's bytes: byte x at ring[q] byte x & 127 (+ 1: banks)
    };
    int result;  // tree description bytes, or -1
    int log;
};
struct LitLds {
    LitBlk blk[kLitG];
};

// Huffman tree description at p (at most `avail` bytes) -> s.tab, s.result, s.log.  One lane.  RFC 8878 4.2.1.
__device__ __forceinline__ void huf_build(LitBlk &s, const uint8_t *p, int avail) {
    int res = -1, n = 0;
    do {
        if (avail < 1) break;
        const int hdr = p[0];
        if (hdr >= 128) {  // direct: 4 bits per weight
            n = hdr - 127;
            const int bytes = (n + 1) / 2;
            if (1 + bytes > avail) break;
            for (int i = 0; i < n; i++) s.w[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
            res = 1 + bytes;
        } else {  // FSE-coded weights, two interleaved states
            if (hdr == 0 || 1 + hdr > avail) break;
            int ns = 0, log = 0;
            const int used = fse_read_norm(p + 1, hdr, 6, 12, s.norm, &ns, &log);
            if (used < 0) break;
            if (fse_build(s.fse, s.norm, ns, log, s.next)) break;
            BitsRev b;
            if (!b.init(p + 1 + used, hdr - used)) break;
            b.need(2 * log);
            uint32_t s1 = b.read(log), s2 = b.read(log);
            bool bad = false;
            for (;;) {
                if (n >= 254) { bad = true; break; }
                uint32_t e = s.fse[s1];
                s.w[n++] = (uint8_t)e;
                b.need((e >> 8) & 255);
                s1 = (e >> 16) + b.read((e >> 8) & 255);
                if (b.left() < 0) {
                    s.w[n++] = (uint8_t)s.fse[s2];
                    break;
                }
                if (n >= 254) { bad = true; break; }
                e = s.fse[s2];
                s.w[n++] = (uint8_t)e;
                b.need((e >> 8) & 255);
                s2 = (e >> 16) + b.read((e >> 8) & 255);
                if (b.left() < 0) {
                    s.w[n++] = (uint8_t)s.fse[s1];
                    break;
                }
            }
            if (bad) break;
            res = 1 + hdr;
        }
        // weights -> code lengths; the last weight is implied (the total must become a power of two)
        uint32_t total = 0;
        for (int k = 0; k < 13; k++) s.rank_start[k] = 0;  // (counts first, start positions below)
        bool ok = true;
        for (int i = 0; i < n; i++) {
            const uint32_t w = s.w[i];
            if (w > 11) { ok = false; break; }
            total += w ? 1u << (w - 1) : 0u;
            s.rank_start[w]++;
        }
        if (!ok || total == 0) { res = -1; break; }
        const int log = hb32(total) + 1;
        const uint32_t rest = (1u << log) - total;
        if (log > 11 || (rest & (rest - 1))) { res = -1; break; }
        const uint32_t lastw = (uint32_t)hb32(rest) + 1;
        s.w[n++] = (uint8_t)lastw;
        s.rank_start[lastw]++;
        if (s.rank_start[1] < 2 || (s.rank_start[1] & 1)) { res = -1; break; }  // libzstd (HUF_readStats): an even number >= 2 of the longest codes
        uint32_t pos = 0;
        for (int k = 1; k <= log; k++) {
            const uint32_t cnt = s.rank_start[k];
            s.rank_start[k] = pos;
            pos += cnt << (k - 1);
        }
        for (int i = 0; i < n; i++) {
            const uint32_t w = s.w[i];
            if (!w) continue;
            const uint32_t len = 1u << (w - 1), at = s.rank_start[w];
            const uint16_t e = (uint16_t)((uint32_t)i | ((uint32_t)(log + 1 - (int)w) << 8));
            for (uint32_t j = 0; j < len; j++) s.tab[at + j] = e;
            s.rank_start[w] = at + len;
        }
        s.log = log;
    } while (0);
    s.result = res;
}

// 64 bytes of the stream [sp, sp + len) for a lane's ring (seq_half_load's twin is below; this one is shared)
struct RingHalf {
    uint4 q[4];
};
__device__ __forceinline__ RingHalf ring_half_load(const uint8_t *sp, int len, int h) {
    RingHalf r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = h * 64 + 16 * k;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (h >= 0 && x < len) __builtin_memcpy(&v, sp + x, 16);  // (zeros in front of the stream; no read goes 16 bytes past its end)
        r.q[k] = v;
    }
    return r;
}
__device__ __forceinline__ void ring_half_store(uint32_t *ring, int h, const RingHalf &v) {
    uint32_t *dst = ring + (h & 1) * 16;  // (4-byte aligned only: the ring's stride is odd)
#pragma unroll
    for (int k = 0; k < 4; k++) dst[4 * k] = v.q[k].x, dst[4 * k + 1] = v.q[k].y, dst[4 * k + 2] = v.q[k].z, dst[4 * k + 3] = v.q[k].w;
}

// one Huffman stream of `nout` symbols -> out, read through `ring`.  false: the stream does not end where its symbols do.
__device__ __forceinline__ bool huf_stream(const uint16_t *tab, uint32_t log, uint32_t *ring, const uint8_t *p, int len, uint8_t *out, uint32_t nout) {
    if (len <= 0) return false;
    int w_h = ((len - 1) >> 6) - 1;  // the lower of the two resident halves: stream bytes [64 w_h, 64 w_h + 128)
    ring_half_store(ring, w_h + 1, ring_half_load(p, len, w_h + 1));
    ring_half_store(ring, w_h, ring_half_load(p, len, w_h));
    RingHalf below = ring_half_load(p, len, w_h - 1);  // in flight while the lane decodes
    const uint32_t top = (ring[((uint32_t)(len - 1) >> 2) & 31u] >> (8u * ((uint32_t)(len - 1) & 3u))) & 255u;
    if (top == 0) return false;  // no end mark in the last byte
    const int P = 8 * (len - 1) + hb32(top);  // the stream's bits [0, P) are to be read, from the top
    // c: the next bits, left-aligned, `avail` of them valid; wi: the 32-bit word of the stream that goes in next
    uint64_t c = 0;
    int avail = 0, wi = -1;
    if (P > 0) {
        wi = (P - 1) >> 5;
        const int cnt = P - 32 * wi;  // 1 .. 32
        const uint32_t word = ring[(uint32_t)wi & 31u] & (cnt == 32 ? ~0u : (1u << cnt) - 1);
        c = (uint64_t)word << (64 - cnt);
        avail = cnt;
        wi--;
    }
    const uint32_t peek_shift = 32 - log;
    auto sym = [&]() -> uint32_t {
        // a word goes in when 32 bits or fewer are left (after it avail >= 33, a symbol takes at most 11) — with a mask,
        // not a branch: the lanes of a wavefront refill at different symbols
        const uint32_t rf = avail <= 32 ? 1u : 0u;
        const uint64_t wv = (uint64_t)ring[(uint32_t)wi & 31u] << ((32 - avail) & 63);
        c |= wv & (0ull - (uint64_t)rf);
        avail += (int)(rf << 5);
        wi -= (int)rf;
        const uint32_t e = tab[(uint32_t)(c >> 32) >> peek_shift];
        c <<= e >> 8;
        avail -= (int)(e >> 8);
        return e & 255u;
    };
    auto ring_ok = [&]() {  // eight symbols take at most 88 bits: three words
        if (4 * (wi - 3) < w_h * 64) {
            w_h--;  // half w_h + 2 is behind the reader: the half below takes its place
            ring_half_store(ring, w_h, below);
            below = ring_half_load(p, len, w_h - 1);
        }
    };
    uint32_t i = 0;
    // head: bytes until the output address is 8-byte aligned, then 8 symbols per store
    ring_ok();
    while (i < nout && (((uintptr_t)(out + i)) & 7)) out[i++] = (uint8_t)sym();
    while (i + 8 <= nout) {
        ring_ok();
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) acc |= (uint64_t)sym() << (8 * k);
        *reinterpret_cast<uint64_t *>(out + i) = acc;
        i += 8;
    }
    ring_ok();
    while (i < nout) out[i++] = (uint8_t)sym();
    return 32 * (wi + 1) + avail == 0;
}

__device__ __forceinline__ void block_fail(Block *blocks, uint32_t b, uint32_t code) { atomicCAS(&blocks[b].status, 0u, code); }

// (b_begin: the blocks in front of it are only there as the sources of repeated tables — a round of a longer stream)
__global__ __launch_bounds__(64) void k_zst_literals(const uint8_t *__restrict__ comp, Block *blocks, uint32_t b_begin, uint32_t nb, uint8_t *lit) {
    __shared__ LitLds s;
    const uint32_t lane = threadIdx.x, g = (lane >> 2) & (kLitG - 1), q = lane & 3;
    for (uint32_t b0 = b_begin + blockIdx.x * kLitG; b0 < nb; b0 += gridDim.x * kLitG) {  // (wave-uniform)
        // ---- Huffman literals: lanes 4 g .. 4 g + 3 take block b0 + g ----
        const uint32_t b = b0 + g;
        const Block *const Bp = blocks + (b < nb ? b : b0);
        const bool huf = lane < 4 * kLitG && b < nb && Bp->type == 2 && Bp->lit_type >= 2;
        LitBlk &L = s.blk[g];
        if (huf && q == 0) {
            const Block *const S = blocks + Bp->huf_src;
            huf_build(L, comp + S->src_off + S->lit_hdr, (int)S->lit_csize);
        }
        __syncthreads();
        if (huf) {
            const int tree = L.result;
            const uint32_t log = (uint32_t)L.log;
            bool ok = tree >= 0;
            if (ok) {
                const uint32_t regen = Bp->lit_regen, lit_type = Bp->lit_type;
                const uint8_t *st = comp + Bp->src_off + Bp->lit_hdr + (lit_type == 2 ? tree : 0);
                const long long left = (long long)Bp->lit_csize - (lit_type == 2 ? tree : 0);
                uint8_t *out = lit + Bp->lit_off;
                if (Bp->lit_streams == 1) {
                    if (q == 0) ok = left >= 1 && huf_stream(L.tab, log, L.ring[0], st, (int)left, out, regen);
                } else {
                    // jump table: three 16-bit stream sizes; the fourth is the rest (4.2.2)
                    bool geo = left >= 10;  // libzstd: jump table + at least one byte per stream
                    uint32_t s1 = 0, s2 = 0, s3 = 0;
                    long long s4 = 0;
                    const uint32_t per = (regen + 3) / 4;
                    if (geo) {
                        s1 = st[0] | ((uint32_t)st[1] << 8);
                        s2 = st[2] | ((uint32_t)st[3] << 8);
                        s3 = st[4] | ((uint32_t)st[5] << 8);
                        s4 = left - 6 - (long long)s1 - s2 - s3;
                        geo = s4 >= 1 && s1 >= 1 && s2 >= 1 && s3 >= 1 && 3 * per <= regen;
                    }
                    if (!geo) {
                        ok = false;
                    } else {
                        const uint8_t *a = st + 6;
                        const uint32_t start = q == 0 ? 0 : q == 1 ? s1 : q == 2 ? s1 + s2 : s1 + s2 + s3;
                        const long long len = q == 0 ? s1 : q == 1 ? s2 : q == 2 ? s3 : s4;
                        const uint32_t nout = q < 3 ? per : regen - 3 * per;
                        ok = huf_stream(L.tab, log, L.ring[q], a + start, (int)len, out + q * per, nout);
                    }
                }
            }
            if (!ok) block_fail(blocks, b, kErrHuffman);
        }
        __syncthreads();  // the tables and rings are the next blocks' from here
        // ---- raw and RLE literals: the whole wavefront, block after block ----
        for (uint32_t k = 0; k < (uint32_t)kLitG && b0 + k < nb; k++) {
            const Block *const Rp = blocks + b0 + k;
            if (Rp->type != 2 || Rp->lit_type >= 2) continue;
            const uint8_t *src = comp + Rp->src_off + Rp->lit_hdr;
            uint8_t *out = lit + Rp->lit_off;
            const uint32_t regen = Rp->lit_regen;
            if (Rp->lit_type == 0) {
                for (uint32_t i = lane; i < regen; i += 64) out[i] = src[i];
            } else {
                const uint8_t v = src[0];
                for (uint32_t i = lane; i < regen; i += 64) out[i] = v;
            }
        }
    }
}

// ---- sequences --------------------------------------------------------------------------------------------------------
__constant__ int16_t kLLDef[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ int16_t kMLDef[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ int16_t kOFDef[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
// base | extra bits << 24 (RFC 8878 3.1.1.3.2.1.1)
__constant__ uint32_t kLLCode[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                     16 | (1u << 24), 18 | (1u << 24), 20 | (1u << 24), 22 | (1u << 24), 24 | (2u << 24), 28 | (2u << 24),
                                     32 | (3u << 24), 40 | (3u << 24), 48 | (4u << 24), 64 | (6u << 24), 128 | (7u << 24), 256 | (8u << 24),
                                     512 | (9u << 24), 1024 | (10u << 24), 2048 | (11u << 24), 4096 | (12u << 24), 8192 | (13u << 24),
                                     16384 | (14u << 24), 32768 | (15u << 24), 65536 | (16u << 24)};
__constant__ uint32_t kMLCode[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                                     35 | (1u << 24), 37 | (1u << 24), 39 | (1u << 24), 41 | (1u << 24), 43 | (2u << 24), 47 | (2u << 24),
                                     51 | (3u << 24), 59 | (3u << 24), 67 | (4u << 24), 83 | (4u << 24), 99 | (5u << 24), 131 | (7u << 24),
                                     259 | (8u << 24), 515 | (9u << 24), 1027 | (10u << 24), 2051 | (11u << 24), 4099 | (12u << 24),
                                     8195 | (13u << 24), 16387 | (14u << 24), 32771 | (15u << 24), 65539 | (16u << 24)};

// One lane decodes one block's sequences: the backward bitstream is a serial chain of dependent table lookups (RFC 8878
// 4.1 offers nothing else inside a block).  A wavefront with ONE such lane still pays a full issue slot per instruction, and
// with a wavefront per block every SIMD of the chip spent its cycles issuing single-lane instructions: 17.8 ms per 1 GiB
// round (8 192 blocks of ~9 500 sequences), whatever the lane waited for.  So:
//  * a wavefront takes kSeqG blocks, a lane each — one instruction stream serves them all;
//  * a block's three tables take 2.5 KiB of LDS, not 5: an entry is the symbol and x, the state's rank among the symbol's
//    states plus the symbol's count (16 bits); the bits to read and the next state's base follow from x in four
//    instructions (nb = log - floor(log2 x), base = (x << nb) - 2^log).  Six wavefronts per CU: every block of a 1 GiB
//    round is in flight at once, and the kernel's time is one block's chain;
//  * the chain never waits for HBM and has no container to refill: a lane reads each bit field straight from a 128-byte ring
//    of its own in LDS (two words, v_alignbit, v_bfe) at a position that follows from the code lookups — the six fields of a
//    sequence are independent reads.  The ring is refilled 64 bytes at a time from registers that were loaded one refill
//    earlier (the s_waitcnt in front of it also waits for the lane's stores — loads and stores share vmcnt on gfx9 — so it
//    is paid once per ~20 sequences).
static constexpr int kSeqG = 8;      // blocks per wavefront
static constexpr int kSeqHalf = 64;  // bytes of a ring half: half h = stream bytes [64 h, 64 h + 64), h < 0: zeros
struct SeqLds {
    uint16_t tab[kSeqG][1280];  // per block: LL [0, 512), OF [512, 768), ML [768, 1280); entry = symbol | x << 6
    alignas(256) uint32_t ring[kSeqG][64];  // stream byte x of block g lives at ring[g] byte x & 127 (halves h, h + 1 resident); word 32 mirrors
                                            // word 0 (a field's two words are one ds_read2_b32), and the 256-byte stride makes the address an OR
    int16_t norm[kSeqG][64];
    uint16_t next[kSeqG][64];
    uint32_t ll_code[36], ml_code[53];
};

// n (< 32) bits of the stream from bit `lowpos` up (bit k of the stream = bit k & 7 of byte k >> 3)
__device__ __forceinline__ uint32_t seq_field(const uint32_t *ring, int lowpos, uint32_t n) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(ring) + (((uint32_t)lowpos >> 3) & 0x7cu));
    const uint32_t lo = w[0], hi = w[1];
    return __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(hi, lo, (uint32_t)lowpos & 31u), 0u, n);
}
// a half into the sequences' ring (its word 32 mirrors word 0)
__device__ __forceinline__ void seq_ring_store(uint32_t *ring, int h, const RingHalf &v) {
    uint4 *dst = reinterpret_cast<uint4 *>(ring + (h & 1) * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) dst[k] = v.q[k];
    if (!(h & 1)) ring[32] = v.q[0].x;
}

// fse_build with 16-bit entries: symbol | x << 6, x = the symbol's count + the rank of the state among the symbol's states
__device__ __forceinline__ int fse_build_x(uint16_t *tab, const int16_t *norm, int n_sym, int log, uint16_t *next) {
    const int size = 1 << log;
    int high = size;
    for (int s = 0; s < n_sym; s++)
        if (norm[s] == -1) {
            tab[--high] = (uint16_t)s;
            next[s] = 1;
        }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < n_sym; s++) {
        const int c = norm[s];
        if (c <= 0) continue;
        next[s] = (uint16_t)c;
        for (int i = 0; i < c; i++) {
            tab[pos] = (uint16_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos >= high);
        }
    }
    if (pos != 0) return -1;
    for (int i = 0; i < size; i++) {
        const uint32_t s = tab[i];
        const uint32_t x = next[s]++;
        tab[i] = (uint16_t)(s | (x << 6));
    }
    return 0;
}

// table t (0 LL, 1 OF, 2 ML) as the sequences section at comp[src_off + seq_hdr ...] defines it (its mode there is not
// Repeat) -> tab, *log_out.  *after: the offset (from the block's start) just behind table t's description.  One lane.
// (The position is an offset that every branch advances BEFORE it looks at `want`: with a pointer post-incremented behind the
// `if (want)` block, hipcc 7.2 dropped the increment on the path of the table that is wanted — an RLE-mode table, one byte —
// and the bitstream began one byte early: "Data corruption detected (sequences)" on valid frames, tools/zstd_soak.py's find;
// tests/golden/zstd_soak/.)
__device__ __forceinline__ int seq_table_from(uint16_t *tab, int16_t *norm, uint16_t *next, int *log_out, int t, const uint8_t *comp, uint64_t src_off,
                                              uint32_t src_size, uint32_t seq_hdr, uint32_t *after) {
    const uint8_t *p = comp + src_off;
    const int modes = p[seq_hdr];
    uint32_t o = seq_hdr + 1;  // offset in the block
    for (int u = 0; u <= t; u++) {
        const bool want = u == t;
        const int m = (modes >> (6 - 2 * u)) & 3;
        const int max_log = u == 1 ? 8 : 9, max_sym = u == 0 ? 35 : u == 1 ? 31 : 52;
        if (m == 1) {
            if (o >= src_size) return -1;
            const uint32_t sym = p[o];
            o += 1;
            if (want) {
                if (sym > (uint32_t)max_sym) return -1;
                tab[0] = (uint16_t)(sym | (1u << 6));  // x = 1 with log 0: no bits, base 0
                *log_out = 0;
            }
        } else if (m == 2) {
            int ns = 0, log = 0;
            const int used = fse_read_norm(p + o, (int)(src_size - o), max_log, max_sym, norm, &ns, &log);
            if (used < 0) return -1;
            o += (uint32_t)used;
            if (want) {
                if (fse_build_x(tab, norm, ns, log, next)) return -1;
                *log_out = log;
            }
        } else if (m == 0) {
            if (want) {
                const int16_t *def = t == 0 ? kLLDef : t == 1 ? kOFDef : kMLDef;
                const int n = t == 0 ? 36 : t == 1 ? 29 : 53, log = t == 1 ? 5 : 6;
                for (int i = 0; i < n; i++) norm[i] = def[i];
                if (fse_build_x(tab, norm, n, log, next)) return -1;
                *log_out = log;
            }
        } else if (want) {
            return -1;  // a Repeat entry is never a source
        }
    }
    *after = o;
    return 0;
}

__global__ __launch_bounds__(64) void k_zst_sequences(const uint8_t *__restrict__ comp, Block *blocks, uint32_t b_begin, uint32_t nb, uint32_t *d_ll,
                                                      uint32_t *d_ml, uint32_t *d_off) {
    __shared__ SeqLds s;
    const uint32_t lane = threadIdx.x;
    if (lane < 36) s.ll_code[lane] = kLLCode[lane];
    if (lane < 53) s.ml_code[lane] = kMLCode[lane];
    __syncthreads();
    if (lane >= kSeqG) return;  // (the wavefront's other lanes have nothing to do: no barrier follows)
    __builtin_amdgcn_s_setprio(3);  // the longest chain of the round: its instructions go first where k_zst_literals' wavefronts share the SIMD
    uint16_t *const tab_ll = s.tab[lane], *const tab_of = s.tab[lane] + 512, *const tab_ml = s.tab[lane] + 768;
    uint32_t *const ring = s.ring[lane];
    for (uint32_t b = b_begin + blockIdx.x * kSeqG + lane; b < nb; b += gridDim.x * kSeqG) {
        const Block *const Bp = blocks + b;
        if (Bp->type != 2) continue;
        const uint32_t nseq = Bp->nseq, lit_regen = Bp->lit_regen;
        if (nseq == 0) {
            blocks[b].out_size = lit_regen;
            blocks[b].rep_out[0] = kRepSym;
            blocks[b].rep_out[1] = kRepSym | (1u << 29);
            blocks[b].rep_out[2] = kRepSym | (2u << 29);
            continue;
        }
        const uint64_t src_off = Bp->src_off, seq_off = Bp->seq_off;
        const uint32_t src_size = Bp->src_size, seq_hdr = Bp->seq_hdr;
        uint32_t err = 0;
        do {
            // the three tables: this block's own descriptions, or the block they are repeated from
            uint32_t q_off = seq_hdr + 1;
            int log3[3] = {0, 0, 0};
            bool bad_tab = false;
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const uint32_t sb = Bp->tbl_src[t];
                const Block *const S = blocks + sb;
                uint32_t after = 0;
                if (seq_table_from(t == 0 ? tab_ll : t == 1 ? tab_of : tab_ml, s.norm[lane], s.next[lane], &log3[t], t, comp, S->src_off, S->src_size,
                                   S->seq_hdr, &after))
                    bad_tab = true;
                else if (sb == b)
                    q_off = after;
                if (bad_tab) break;
            }
            if (bad_tab) { err = kErrFse; break; }
            if (q_off >= src_size) { err = kErrSequences; break; }
            const uint8_t *sp = comp + src_off + q_off;
            const int len = (int)(src_size - q_off);
            int w_h = ((len - 1) >> 6) - 1;  // the lower of the two resident halves
            seq_ring_store(ring, w_h + 1, ring_half_load(sp, len, w_h + 1));
            seq_ring_store(ring, w_h, ring_half_load(sp, len, w_h));
            RingHalf below = ring_half_load(sp, len, w_h - 1);  // in flight while the lane decodes
            const uint32_t top = (ring[((uint32_t)(len - 1) >> 2) & 31u] >> (8u * ((uint32_t)(len - 1) & 3u))) & 255u;
            if (top == 0) { err = kErrSequences; break; }  // no end mark in the last byte
            int P = 8 * (len - 1) + hb32(top);              // unread bits: the stream's bits [0, P)
            const uint32_t llog = (uint32_t)log3[0], olog = (uint32_t)log3[1], mlog = (uint32_t)log3[2];
            P -= (int)llog;
            uint32_t sl = seq_field(ring, P, llog);
            P -= (int)olog;
            uint32_t so = seq_field(ring, P, olog);
            P -= (int)mlog;
            uint32_t sm = seq_field(ring, P, mlog);
            if (P < 0) { err = kErrSequences; break; }
            uint32_t r0 = kRepSym, r1 = kRepSym | (1u << 29), r2 = kRepSym | (2u << 29);
            uint32_t sum_ll = 0, sum_ml = 0;
            uint32_t *o_ll = d_ll + seq_off, *o_ml = d_ml + seq_off, *o_off = d_off + seq_off;
            // (the loop is one lane's serial chain and the wavefront issues it in order: every instruction counts — flags are
            // collected with bit operations and the offset is chosen with selects, so that no lane's case costs the others a branch)
            uint32_t badw = 0, bad_offw = 0;
            const uint32_t size_l = 1u << llog, size_o = 1u << olog, size_m = 1u << mlog;
            for (uint32_t i = 0; i < nseq && !(badw | bad_offw); i++) {
                if ((P >> 3) - 16 < w_h * kSeqHalf) {  // (a sequence takes at most 89 bits and its reads begin 3 bytes below: it stays inside the ring)
                    w_h--;  // half w_h + 2 is behind the reader: the half below takes its place
                    seq_ring_store(ring, w_h, below);
                    below = ring_half_load(sp, len, w_h - 1);
                }
                const uint32_t el = tab_ll[sl], eo = tab_of[so], em = tab_ml[sm];
                const uint32_t lc = el & 63, oc = eo & 63, mc = em & 63;
                const uint32_t bad_sym = (uint32_t)(lc > 35) | (uint32_t)(oc > 31) | (uint32_t)(mc > 52);
                const uint32_t mcode = s.ml_code[mc], lcode = s.ll_code[lc];
                const uint32_t xl = el >> 6, xo = eo >> 6, xm = em >> 6;
                const uint32_t more = i + 1 < nseq ? ~0u : 0u;  // (the last sequence reads no next states)
                const uint32_t nl = (llog - 31 + (uint32_t)__builtin_clz(xl)) & more, nm = (mlog - 31 + (uint32_t)__builtin_clz(xm)) & more,
                               no = (olog - 31 + (uint32_t)__builtin_clz(xo)) & more;
                const uint32_t n1 = mcode >> 24, n2 = lcode >> 24;
                const int p0 = P - (int)oc, p1 = p0 - (int)n1, p2 = p1 - (int)n2, p3 = p2 - (int)nl, p4 = p3 - (int)nm, p5 = p4 - (int)no;
                const uint32_t f0 = seq_field(ring, p0, oc), f1 = seq_field(ring, p1, n1), f2 = seq_field(ring, p2, n2);
                const uint32_t f3 = seq_field(ring, p3, nl), f4 = seq_field(ring, p4, nm), f5 = seq_field(ring, p5, no);
                P = p5;
                const uint32_t ov = (1u << oc) + f0;
                const uint32_t ml = (mcode & 0xFFFFFFu) + f1;
                const uint32_t ll = (lcode & 0xFFFFFFu) + f2;
                sl = (xl << nl) + f3 - size_l;
                sm = (xm << nm) + f4 - size_m;
                so = (xo << no) + f5 - size_o;
                // the offset: a new one (ov > 3), or one of the three last (RFC 8878 3.1.1.5), symbolic while it names the history
                // in front of the block
                const uint32_t sel = ov > 3 ? 4u : ov - 1 + (uint32_t)(ll == 0);  // 0: r0, 1: r1, 2: r2, 3: r0 - 1, 4: new
                const uint32_t r0_sym = r0 >> 31;
                const uint32_t dec = r0 + (r0_sym << 1) - 1;  // "r0 minus one": a symbolic offset counts what was taken off
                uint32_t code = ov - 3;
                code = sel == 3 ? dec : code;
                code = sel == 2 ? r2 : code;
                code = sel == 1 ? r1 : code;
                code = sel == 0 ? r0 : code;
                bad_offw |= ~bad_sym & (((uint32_t)(sel == 4) & (uint32_t)(code >= (1u << 29))) | ((uint32_t)(sel == 3) & (r0_sym ^ 1u) & (uint32_t)(r0 <= 1)));
                r2 = sel >= 2 ? r1 : r2;
                r1 = sel >= 1 ? r0 : r1;
                r0 = code;
                o_ll[i] = ll;
                o_ml[i] = ml;
                o_off[i] = code;
                sum_ll += ll;
                sum_ml += ml;
                badw |= bad_sym | ((uint32_t)((sum_ll > sum_ml ? sum_ll : sum_ml) > kBlockMax) & ~bad_offw);
            }
            const bool bad = (badw & 1u) != 0, bad_off = (bad_offw & 1u) != 0;
            if (bad_off) { err = kErrOffset; break; }
            if (bad || P != 0) { err = kErrSequences; break; }
            if (sum_ll > lit_regen) { err = kErrSequences; break; }
            if (lit_regen + sum_ml > kBlockMax) { err = kErrSize; break; }
            blocks[b].out_size = lit_regen + sum_ml;
            blocks[b].rep_out[0] = r0;
            blocks[b].rep_out[1] = r1;
            blocks[b].rep_out[2] = r2;
        } while (0);
        if (err) block_fail(blocks, b, err);
    }
}

// ---- scan: output offsets and the repeat-offset history at every block's start -----------------------------------
__device__ __forceinline__ uint32_t rep_apply(uint32_t code, uint32_t r0, uint32_t r1, uint32_t r2, bool *bad) {
    if (!(code & kRepSym)) return code;
    const uint32_t slot = (code >> 29) & 3, k = code & 0x1FFFFFFFu;
    const uint32_t base = slot == 0 ? r0 : slot == 1 ? r1 : r2;
    if (base <= k) {
        *bad = true;
        return 1;
    }
    return base - k;
}

// state: in = {output offset of the first block, the repeat offsets in front of it (a frame that began in an earlier round)},
//        out = {end of the last block's output, the repeat offsets behind it}
struct ScanState {
    unsigned long long run;
    unsigned int rep[3], pad;
};
__global__ __launch_bounds__(64) void k_zst_scan(Block *blocks, uint32_t b_begin, uint32_t nb, ScanState *state) {
    const uint32_t lane = threadIdx.x;
    uint64_t run = state->run;
    uint32_t r0 = state->rep[0], r1 = state->rep[1], r2 = state->rep[2];  // wave-uniform
    for (uint32_t g = b_begin; g < nb; g += 64) {
        const uint32_t i = g + lane;
        const bool valid = i < nb;
        uint32_t out_size = 0, first = 0, has = 0, c0 = 0, c1 = 0, c2 = 0;
        if (valid) {
            const Block *B = blocks + i;
            out_size = B->status ? 0u : B->out_size;
            first = B->first_of_frame;
            has = B->type == 2 && B->nseq > 0 && B->status == 0;
            c0 = B->rep_out[0], c1 = B->rep_out[1], c2 = B->rep_out[2];
        }
        const uint32_t incl = wave_incl_sum(out_size);
        const uint32_t cnt = nb - g < 64 ? nb - g : 64;
        uint32_t in0 = 0, in1 = 0, in2 = 0;
        bool my_bad = false;
        for (uint32_t j = 0; j < cnt; j++) {
            if (__builtin_amdgcn_readlane(first, j)) r0 = 1, r1 = 4, r2 = 8;
            if (lane == j) in0 = r0, in1 = r1, in2 = r2;
            if (__builtin_amdgcn_readlane(has, j)) {
                bool bad = false;
                const uint32_t n0 = rep_apply(__builtin_amdgcn_readlane(c0, j), r0, r1, r2, &bad);
                const uint32_t n1 = rep_apply(__builtin_amdgcn_readlane(c1, j), r0, r1, r2, &bad);
                const uint32_t n2 = rep_apply(__builtin_amdgcn_readlane(c2, j), r0, r1, r2, &bad);
                r0 = n0, r1 = n1, r2 = n2;
                if (bad && lane == j) my_bad = true;
            }
        }
        if (valid) {
            Block *B = blocks + i;
            B->out_off = run + incl - out_size;
            B->rep_in[0] = in0, B->rep_in[1] = in1, B->rep_in[2] = in2;
            if (my_bad) atomicCAS(&B->status, 0u, (uint32_t)kErrOffset);
        }
        run += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) {
        state->run = run;
        state->rep[0] = r0, state->rep[1] = r1, state->rep[2] = r2;
    }
}

// ---- execution of the sequences -------------------------------------------------------------------------------------
// LDS decides how many chunks a CU executes at once, and a chunk is a serial chain: with a ring of 2 Ki symbols, 4 KiB of
// staged literals and 1 Ki staged match symbols (16.4 KiB) nine wavefronts fit a CU; half of each (EXG_ZST_EXEC_SMALL, the
// default: 8.2 KiB) lets sixteen in — the registers' limit.  -DEXG_ZST_EXEC_SMALL=0 builds the former shape (A/B).
#ifndef EXG_ZST_EXEC_SMALL
#define EXG_ZST_EXEC_SMALL 1
#endif
#ifndef EXG_ZST_EXEC_BY_SEQUENCE
#define EXG_ZST_EXEC_BY_SEQUENCE 0  // 1: the group loop of rounds 2-3 (a literal run + a match per turn), for A/B
#endif
static constexpr uint32_t kFlushLog2 = EXG_ZST_EXEC_SMALL ? 9 : 10;
static constexpr uint32_t kFlush = 1u << kFlushLog2;  // elements of a segment of the absolute grid that leaves for HBM at once
static constexpr uint32_t kPiece = kFlush;            // elements emitted between two flush checks (ring >= 2 * kPiece)
template <bool SYM>
struct ExecCfg {
    using Elem = uint8_t;
    static constexpr uint32_t kRing = 4 * kFlush;
};
template <>
struct ExecCfg<true> {
    using Elem = uint32_t;
    static constexpr uint32_t kRing = 2 * kFlush;
};

// Positions are 32-bit and relative to the chunk's first element (a chunk is a few MiB); `bias` = that element's index
// mod 1024, so that ring slots and the kFlush-element flush grid follow the ABSOLUTE index (full segments are 16-byte
// aligned in HBM whatever the chunk's offset).  Almost every copy is short (a literal run or a match of a few elements):
// those take one masked wave instruction and a boundary test — the general loops are for the long ones.
template <bool SYM>
struct Exec {
    using Elem = typename ExecCfg<SYM>::Elem;
    static constexpr uint32_t kRing = ExecCfg<SYM>::kRing, kMask = kRing - 1;
    static constexpr uint32_t kStage = kFlush;  // elements of far-match sources staged per 64-sequence group
    Elem *ring;        // LDS
    Elem *out0;        // the chunk's first element in HBM (bytes: the output; symbols: d_sym)
    uint32_t bias;     // (index of the chunk's first element) & 1023
    uint32_t pos;      // elements emitted so far
    uint32_t flushed;  // elements [0, flushed) are in HBM
    uint32_t lane;

    __device__ __forceinline__ uint32_t slot(uint32_t p) const { return (bias + p) & kMask; }
    // completed kFlush-element segments of the absolute grid -> HBM
    __device__ __forceinline__ void flush(bool all) {
        bool any = false;
        for (;;) {
            uint32_t next = ((bias + flushed) | (kFlush - 1u)) + 1 - bias;
            if (next > pos) {
                if (!all || flushed == pos) break;
                next = pos;
            }
            const uint32_t n = next - flushed;
            if (n == kFlush) {
                constexpr uint32_t kPerLane = kFlush / 64;  // 16 or 8 elements: 16 / 8 bytes, or 64 / 32 bytes of symbols
                const Elem *src = ring + slot(flushed) + lane * kPerLane;
                Elem *dst = out0 + flushed + lane * kPerLane;
                if constexpr (kPerLane * sizeof(Elem) >= 16) {
#pragma unroll
                    for (uint32_t k = 0; k < kPerLane * sizeof(Elem) / 16; k++) reinterpret_cast<uint4 *>(dst)[k] = reinterpret_cast<const uint4 *>(src)[k];
                } else {
                    *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(src);
                }
            } else {
                for (uint32_t e = lane; e < n; e += 64) out0[flushed + e] = ring[slot(flushed + e)];
            }
            flushed = next;
            any = true;
        }
        // later matches read these elements back through the CU's own L1 (write-through, coherent inside a workgroup):
        // the stores only have to be complete first
        if (any) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    __device__ __forceinline__ void advance(uint32_t n) {
        pos += n;
        if (((bias + pos) ^ (bias + flushed)) >> kFlushLog2) flush(false);  // a segment boundary was crossed
    }
    template <class Src>
    __device__ __forceinline__ void put_from(const Src *src, uint32_t n) {  // src: LDS or HBM, elements or bytes
        if (n <= 64) {
            if (lane < n) ring[slot(pos + lane)] = (Elem)src[lane];
            advance(n);
            return;
        }
        for (uint32_t done = 0; done < n;) {
            const uint32_t piece = n - done < kPiece ? n - done : kPiece;
            for (uint32_t i = lane; i < piece; i += 64) ring[slot(pos + i)] = (Elem)src[done + i];
            done += piece;
            advance(piece);
        }
    }
    __device__ __forceinline__ void put_bytes(const uint8_t *src, uint32_t n) { put_from(src, n); }
    __device__ __forceinline__ void put_lds(const uint8_t *src, uint32_t n) { put_from(src, n); }
    __device__ __forceinline__ void put_staged(const Elem *src, uint32_t n) { put_from(src, n); }
    __device__ __forceinline__ void put_fill(uint8_t v, uint32_t n) {
        for (uint32_t done = 0; done < n;) {
            const uint32_t piece = n - done < kPiece ? n - done : kPiece;
            for (uint32_t i = lane; i < piece; i += 64) ring[slot(pos + i)] = (Elem)v;
            done += piece;
            advance(piece);
        }
    }
    // one element of a match's source: chunk-relative position s (negative: in front of the chunk)
    __device__ __forceinline__ Elem fetch(int32_t s, uint32_t hi) const {
        if (SYM && s < 0) return (Elem)(kSymRef | (uint32_t)(-s));  // "the byte d in front of the chunk's first"
        if (!SYM && s < 0) return out0[(long long)s];  // a byte chunk that continues a frame: the window earlier rounds left in front
        if ((uint32_t)s + kRing >= hi) return ring[slot((uint32_t)s)];
        return __hip_atomic_load(out0 + (uint32_t)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // flushed >= 1 KiB ago
    }
    // LZ77 copy: `len` elements from `off` back; the source index is folded into [match start - off, match start), so an
    // overlapping copy is exact whatever the order the lanes work in
    __device__ __forceinline__ void put_match(uint32_t off, uint32_t len) {
        const int32_t s0 = (int32_t)pos - (int32_t)off;  // source of the match's first element
        if (len <= 64 && off >= len) {  // the usual shape: short, no overlap
            if (lane < len) ring[slot(pos + lane)] = fetch(s0 + (int32_t)lane, pos + len);
            advance(len);
            return;
        }
        for (uint32_t done = 0; done < len;) {
            const uint32_t piece = len - done < kPiece ? len - done : kPiece;
            const uint32_t hi = pos + piece;
            for (uint32_t i = lane; i < piece; i += 64) {
                const uint32_t k = done + i;
                ring[slot(pos + i)] = fetch(s0 + (int32_t)(off >= len ? k : k % off), hi);
            }
            done += piece;
            advance(piece);
        }
    }
};

static constexpr uint32_t kLitStage = 4 * kFlush;  // bytes of a 64-sequence group's literals staged in LDS

// one chunk (a run of blocks of one frame) by one wavefront
template <bool SYM>
__device__ void exec_chunk(const Chunk &C, typename ExecCfg<SYM>::Elem *ring, uint32_t *lit_stage, typename ExecCfg<SYM>::Elem *match_stage,
                           const uint8_t *__restrict__ comp,
                           const Block *blocks, const uint8_t *__restrict__ lit, const uint32_t *__restrict__ d_ll,
                           const uint32_t *__restrict__ d_ml, const uint32_t *__restrict__ d_off, typename ExecCfg<SYM>::Elem *out,
                           uint32_t *status_out, uint32_t lane) {
    Exec<SYM> ex;
    ex.ring = ring;
    ex.out0 = out + C.elem_off;
    ex.bias = (uint32_t)(C.elem_off & 1023);
    ex.pos = ex.flushed = 0;
    ex.lane = lane;
    const uint64_t frame_pos0 = C.out_off - C.frame_out_off;  // bytes of the frame in front of the chunk
    const uint8_t *stage8 = reinterpret_cast<const uint8_t *>(lit_stage);
    uint32_t err = 0;
    for (uint32_t bi = 0; bi < C.n_blocks && !err; bi++) {
        const uint32_t b = C.first_block + bi;
        const Block B = blocks[b];
        if (B.type == 0) {
            ex.put_bytes(comp + B.src_off, B.src_size);
        } else if (B.type == 1) {
            ex.put_fill(comp[B.src_off], B.src_size);
        } else {
            const uint8_t *L = lit + B.lit_off;
            uint32_t lit_pos = 0;
            for (uint32_t g = 0; g < B.nseq && !err; g += 64) {
                const uint32_t i = g + lane;
                const bool valid = i < B.nseq;
                uint32_t ll = 0, ml = 0, off = 1;
                bool bad = false;
                if (valid) {
                    ll = d_ll[B.seq_off + i];
                    ml = d_ml[B.seq_off + i];
                    const uint32_t code = d_off[B.seq_off + i];
                    if (code & kRepSym) {
                        const uint32_t slot = (code >> 29) & 3, k = code & 0x1FFFFFFFu;
                        const uint32_t base = B.rep_in[slot];
                        bad = base <= k;
                        off = bad ? 1u : base - k;
                    } else {
                        off = code;
                        bad = code == 0;
                    }
                }
                if (__any(bad)) {
                    err = kErrOffset;
                    break;
                }
                const uint32_t ll_incl = wave_incl_sum(ll);
                const uint32_t cnt = B.nseq - g < 64 ? B.nseq - g : 64;
                const uint32_t group_lits = __builtin_amdgcn_readlane(ll_incl, 63);
                // The literals of the group's sequences are one contiguous run of the literal buffer: staged in LDS by
                // the whole wave at once (dword loads from the 4-byte boundary below), so that the per-sequence copies
                // below do not each wait for HBM — that wait, once per sequence, was most of a chunk's time.
                const uint32_t mis = (uint32_t)((uintptr_t)(L + lit_pos) & 3);
                const bool staged = group_lits + mis <= kLitStage;
                if (staged && group_lits) {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(L + lit_pos - mis);
                    const uint32_t words = (group_lits + mis + 3) / 4;
                    for (uint32_t w = lane; w < words; w += 64) lit_stage[w] = src[w];
                }
                // Far matches: a match whose whole source already lies in HBM (flushed before this group began) would cost
                // one exposed memory round trip per sequence — with short matches found far back (DNA) that was ~1.4 us of
                // every sequence.  Every lane fetches ITS sequence's source into LDS (independent loads, all in flight
                // together); the sequence loop below then copies LDS -> LDS.  Where a sequence's output begins is a prefix
                // sum away (out_incl), so its source position is known before anything of the group is executed.
                const uint32_t out_incl = wave_incl_sum(ll + ml);
                const uint32_t m_pos = ex.pos + (out_incl - ml);  // where my match begins (chunk-relative)
                bool far = false;
                if (valid && ml && off <= m_pos)                      // the source lies inside this chunk ...
                    far = off >= ml && m_pos - off + ml <= ex.flushed;  // ... and all of it is in HBM already
                const uint32_t far_incl = wave_incl_sum(far ? ml : 0u);
                uint32_t far_off = far_incl - (far ? ml : 0u);
                if (far && far_incl > Exec<SYM>::kStage) far = false;     // (what does not fit is read the slow way)
                if (far) {
                    const typename ExecCfg<SYM>::Elem *src = ex.out0 + (m_pos - off);
                    for (uint32_t k = 0; k < ml; k++) match_stage[far_off + k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#if EXG_ZST_EXEC_BY_SEQUENCE
                const unsigned long long far_mask = __ballot(far);
                for (uint32_t j = 0; j < cnt; j++) {
                    const uint32_t L_j = __builtin_amdgcn_readlane(ll, j), M_j = __builtin_amdgcn_readlane(ml, j);
                    const uint32_t O_j = __builtin_amdgcn_readlane(off, j);
                    const uint32_t lrel = __builtin_amdgcn_readlane(ll_incl, j) - L_j;
                    if (L_j) {
                        if (staged)
                            ex.put_lds(stage8 + mis + lrel, L_j);
                        else
                            ex.put_bytes(L + lit_pos + lrel, L_j);
                    }
                    // libzstd: a match may reach back to the first byte of the frame's content, not beyond
                    if (O_j > ex.pos && (uint64_t)O_j > frame_pos0 + ex.pos) {
                        err = kErrOffset;
                        break;
                    }
                    if ((far_mask >> j) & 1)
                        ex.put_staged(match_stage + __builtin_amdgcn_readlane(far_off, j), M_j);
                    else
                        ex.put_match(O_j, M_j);
                }
#else
                // The group's output, 64 ELEMENTS at a time (not a sequence at a time: FASTQ at level 3 has ~14 elements per
                // sequence, and a literal run + a match per turn left most lanes idle through ~90 instructions — the kernel was
                // bound by the chip's issue slots).  Where a sequence's output begins is out_incl away; an element finds its
                // sequence by counting the sequence starts at or in front of it, takes what it needs of that lane (ds_bpermute),
                // and reads its literal or its match's source.  A source inside the same 64 elements (a lower lane: sources lie
                // in front) waits for that lane's turn; an overlapping match folds its source in front of itself first.
                (void)cnt;
                const uint32_t out_excl = out_incl - (ll + ml);  // where my sequence's output begins, group-relative
                if (__any(valid && off > m_pos && (uint64_t)off > frame_pos0 + m_pos)) {  // libzstd: a match may reach back to the first
                    err = kErrOffset;                                                       // byte of the frame's content, not beyond
                    break;
                }
                const uint32_t T = __builtin_amdgcn_readlane(out_incl, 63);
                const uint32_t G0 = ex.pos;
                const uint32_t far_key = far ? far_off : ~0u;
                const uint32_t lrel = ll_incl - ll;
                uint32_t started = 0;  // sequences that began in front of the pass (wave-uniform)
                volatile uint8_t *mark = reinterpret_cast<volatile uint8_t *>(lit_stage) + kLitStage + 16;  // (64 bytes behind the literal stage: k_zst_exec)
                for (uint32_t done = 0; done < T; done += 64) {
                    const uint32_t n = T - done < 64 ? T - done : 64;
                    // which elements of the pass begin a sequence
                    mark[lane] = 0;
                    __builtin_amdgcn_wave_barrier();
                    if (valid && out_excl - done < 64u) mark[out_excl - done] = 1;
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t is_start = mark[lane];
                    const uint32_t starts_incl = wave_incl_sum(is_start);
                    const uint32_t j = started + starts_incl - 1;  // my sequence (element 0 of a group begins sequence 0)
                    started += __builtin_amdgcn_readlane(starts_incl, 63);
                    const bool act = lane < n;
                    const uint32_t jb = (j & 63u) * 4;
                    const uint32_t j_excl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)jb, (int)out_excl);
                    const uint32_t j_ll = (uint32_t)__builtin_amdgcn_ds_bpermute((int)jb, (int)ll);
                    const uint32_t j_off = (uint32_t)__builtin_amdgcn_ds_bpermute((int)jb, (int)off);
                    const uint32_t j_lrel = (uint32_t)__builtin_amdgcn_ds_bpermute((int)jb, (int)lrel);
                    const uint32_t j_far = (uint32_t)__builtin_amdgcn_ds_bpermute((int)jb, (int)far_key);
                    const uint32_t rel = done + lane - j_excl;  // my element's index in its sequence's output
                    const uint32_t hi = G0 + done + n;
                    typename ExecCfg<SYM>::Elem v = 0;
                    bool dep = false;
                    int32_t src = 0;
                    if (act) {
                        if (rel < j_ll) {
                            v = staged ? (typename ExecCfg<SYM>::Elem)stage8[mis + j_lrel + rel] : (typename ExecCfg<SYM>::Elem)L[lit_pos + j_lrel + rel];
                        } else {
                            const uint32_t k = rel - j_ll;
                            if (j_far != ~0u) {
                                v = match_stage[j_far + k];
                            } else {
                                const uint32_t mstart = G0 + j_excl + j_ll;
                                src = (int32_t)mstart - (int32_t)j_off + (int32_t)(k < j_off ? k : k % j_off);
                                dep = src >= (int32_t)(G0 + done);
                                if (!dep) v = ex.fetch(src, hi);
                            }
                        }
                        if (!dep) ex.ring[ex.slot(G0 + done + lane)] = v;
                    }
                    // sources inside the pass: a lane's turn comes when its source's lane has had its own
                    unsigned long long ready = __ballot(!act || !dep);
                    while (ready != ~0ull) {
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t src_lane = (uint32_t)(src - (int32_t)(G0 + done));
                        const bool now = act && dep && ((ready >> (src_lane & 63u)) & 1ull);
                        if (now) {
                            ex.ring[ex.slot(G0 + done + lane)] = ex.ring[ex.slot((uint32_t)src)];
                            dep = false;
                        }
                        ready |= __ballot(now);
                    }
                    ex.advance(n);
                }
#endif
                lit_pos += group_lits;
            }
            if (!err && lit_pos < B.lit_regen) ex.put_bytes(L + lit_pos, B.lit_regen - lit_pos);
        }
    }
    ex.flush(true);
    if (lane == 0) *status_out = err;
}

// byte chunks (the first of every frame) and symbol chunks in ONE launch: a lone byte chunk — one wavefront, ~5 ms for a
// 128 KiB block — used to run ahead of the symbol chunks in a launch of its own
__global__ __launch_bounds__(64) void k_zst_exec(const uint8_t *__restrict__ comp, const Block *blocks, const Chunk *chunks, uint32_t n_chunks,
                                                 const uint8_t *__restrict__ lit, const uint32_t *__restrict__ d_ll,
                                                 const uint32_t *__restrict__ d_ml, const uint32_t *__restrict__ d_off, uint8_t *out_bytes,
                                                 uint32_t *out_syms, uint32_t *chunk_status) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[ExecCfg<true>::kRing];  // 8 KiB: 2 Ki symbols, or 4 Ki bytes
    __shared__ __attribute__((aligned(16))) uint32_t lit_stage[kLitStage / 4 + 4 + 16];  // + 16 bytes of slack + 64 bytes of sequence-start marks
    __shared__ __attribute__((aligned(16))) uint32_t match_stage[Exec<true>::kStage];  // 4 KiB: 1 Ki symbols (bytes use a quarter)
    static_assert(sizeof(ring) >= ExecCfg<false>::kRing, "the byte ring fits");
    const uint32_t lane = threadIdx.x;
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const Chunk C = chunks[c];
        if (C.symbolic)
            exec_chunk<true>(C, ring, lit_stage, match_stage, comp, blocks, lit, d_ll, d_ml, d_off, out_syms, chunk_status + c, lane);
        else
            exec_chunk<false>(C, reinterpret_cast<uint8_t *>(ring), lit_stage, reinterpret_cast<uint8_t *>(match_stage), comp, blocks, lit, d_ll, d_ml, d_off, out_bytes, chunk_status + c, lane);
        __syncthreads();
    }
}

// ---- XXH64 of a frame's content (RFC 8878 3.1.1: Content_Checksum = its low 32 bits, seed 0) ----------------------
static constexpr uint64_t XP1 = 11400714785074694791ull, XP2 = 14029467366897019727ull, XP3 = 1609587929392839161ull,
                          XP4 = 9650029242287828579ull, XP5 = 2870177450012600261ull;
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t in) { return rotl64(acc + in * XP2, 31) * XP1; }
__device__ __forceinline__ uint64_t xmerge(uint64_t h, uint64_t v) { return (h ^ xround(0, v)) * XP1 + XP4; }

__global__ __launch_bounds__(64) void k_zst_xxh64(const uint8_t *__restrict__ out, const Frame *frames, uint32_t nf, uint64_t max_bytes,
                                                  uint32_t *frame_status) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t f = blockIdx.x; f < nf; f += gridDim.x) {
        const Frame F = frames[f];
        if (!F.has_checksum || F.out_size > max_bytes) continue;
        const uint8_t *p = out + F.out_off;
        const uint64_t n = F.out_size, stripes = n / 32;
        // the four accumulators are four independent serial chains: lanes 0-3
        uint64_t acc = lane == 0 ? XP1 + XP2 : lane == 1 ? XP2 : lane == 2 ? 0 : 0 - XP1;
        if (lane < 4) {
            const uint8_t *q = p + 8 * lane;
            uint64_t s = 0;
            for (; s + 8 <= stripes; s += 8) {
                uint64_t x[8];
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = ld64(q + 32 * (s + k));
#pragma unroll
                for (int k = 0; k < 8; k++) acc = xround(acc, x[k]);
            }
            for (; s < stripes; s++) acc = xround(acc, ld64(q + 32 * s));
        }
        const uint64_t v1 = __shfl((unsigned long long)acc, 0, 64), v2 = __shfl((unsigned long long)acc, 1, 64),
                       v3 = __shfl((unsigned long long)acc, 2, 64), v4 = __shfl((unsigned long long)acc, 3, 64);
        if (lane == 0) {
            uint64_t h;
            if (n >= 32) {
                h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
                h = xmerge(h, v1), h = xmerge(h, v2), h = xmerge(h, v3), h = xmerge(h, v4);
            } else {
                h = XP5;
            }
            h += n;
            const uint8_t *t = p + stripes * 32, *end = p + n;
            while (t + 8 <= end) {
                h ^= xround(0, ld64(t));
                h = rotl64(h, 27) * XP1 + XP4;
                t += 8;
            }
            if (t + 4 <= end) {
                uint32_t w;
                __builtin_memcpy(&w, t, 4);
                h ^= (uint64_t)w * XP1;
                h = rotl64(h, 23) * XP2 + XP3;
                t += 4;
            }
            while (t < end) {
                h ^= (*t++) * XP5;
                h = rotl64(h, 11) * XP1;
            }
            h ^= h >> 33, h *= XP2, h ^= h >> 29, h *= XP3, h ^= h >> 32;
            frame_status[f] = (uint32_t)h == F.checksum ? 0u : (uint32_t)kErrChecksum;
        }
    }
}

// ---- symbol chunks: resolve "the byte d in front of the chunk" --------------------------------------------------------------
// A launch resolves a GROUP of consecutive chunks (about 512 KiB of output), a thread per four symbols.  A reference names a
// position in front of its chunk; that position holds a final byte (the frame's first chunk, or anything in front of the
// group) or another symbol — of an EARLIER chunk of the group, whose own references point further back still — so a thread
// follows the chain through the (read-only) symbol buffer until it meets a byte: at most one hop per chunk of the group.
// The groups run in order.  (Chains are long in real files — a read name copies its prefix from the record before it, which
// copied it from the one before — so following them across the whole frame costs far more than the order: 1.07 s per
// 512 MB.  Per 512 MB of FASTQ in 128 KiB chunks: a launch per chunk 13.6 ms (3 900 dependent launches), groups of four
// 10.9 ms, of sixteen 13.7 ms.)
struct ResolveArgs {
    const uint32_t *sym;
    const Chunk *chunks;        // all chunks
    const uint32_t *sym_chunks; // the round's symbolic chunks (indices into chunks), ascending elem_off: the group's first
    uint32_t n_sym_chunks;      // chunks of the group this launch looks at
    uint32_t n_ctx;             // k_zst_resolve_inner: the first n_ctx of them are only read, the rest is rewritten
    uint8_t *out;
    uint64_t final_below;       // everything in front of the group's first chunk is final (or left for a later pass)
    uint64_t elem0, n_elems;    // the symbol slots [elem0, elem0 + n_elems) this launch works on (chunks padded to 4)
};

static constexpr uint32_t kGroupMax = 256;     // chunks per resolve launch
static constexpr uint32_t kGroupMaxBytes = 32; // ... of round 4's groups by bytes (EXG_ZSTD_RESOLVE_INNER=0)

static constexpr uint32_t kResolveRows = 8;  // rows of 1 024 symbols per workgroup
static constexpr uint64_t kRowElems = 1024ull * kResolveRows;

// the last j in [0, n) with v[j] <= x (v ascending, v[0] <= x)
__device__ __forceinline__ uint32_t last_at_or_below(const uint64_t *v, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v[mid] <= x) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ROWS rows of 1 024 symbols per workgroup: the group's tables are loaded once for all of them, and all rows' symbols are asked
// for before the first is looked at (a workgroup per row was bound by the rate at which workgroups start: 0.6-1 TB/s).
// ROWS = 1 for round 4's small groups, whose launches are a chain of latencies.
template <uint32_t ROWS>
__device__ __forceinline__ void resolve_group(const ResolveArgs &a, uint32_t block_x) {
    // the group's chunks: where they start in the output and in the symbol buffer, what lies in front of them as bytes
    __shared__ uint64_t s_out[kGroupMax], s_elem[kGroupMax], s_size[kGroupMax];
    __shared__ uint64_t s_below_frame[2];
    const uint32_t n = a.n_sym_chunks;
    if (threadIdx.x < n) {
        const Chunk &c = a.chunks[a.sym_chunks[threadIdx.x]];
        s_out[threadIdx.x] = c.out_off;
        s_elem[threadIdx.x] = c.elem_off;
        s_size[threadIdx.x] = c.size;
        if (threadIdx.x == 0) {  // (a group lies in ONE frame: where its bytes end and where it begins are the group's)
            s_below_frame[0] = c.byte_end > a.final_below ? c.byte_end : a.final_below;
            s_below_frame[1] = c.frame_out_off;
        }
    }
    __syncthreads();
    const uint64_t below = s_below_frame[0], frame0 = s_below_frame[1];
    uint4 rows[ROWS];  // (elem_off is a multiple of 4, the slots are padded to it)
#pragma unroll
    for (uint32_t row = 0; row < ROWS; row++) {
        const uint64_t e = a.elem0 + ((uint64_t)block_x * ROWS + row) * 1024 + (uint64_t)threadIdx.x * 4;
        rows[row] = e < a.elem0 + a.n_elems ? *reinterpret_cast<const uint4 *>(a.sym + e) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t row = 0; row < ROWS; row++) {
        const uint64_t e = a.elem0 + ((uint64_t)block_x * ROWS + row) * 1024 + (uint64_t)threadIdx.x * 4;
        if (e >= a.elem0 + a.n_elems) break;
        const uint32_t k = last_at_or_below(s_elem, n, e);
        const uint64_t i = e - s_elem[k], size = s_size[k];
        if (i >= size) continue;  // padding between two chunks
        const uint4 xv = rows[row];
        uint32_t x[4] = {xv.x, xv.y, xv.z, xv.w}, at[4] = {k, k, k, k};
        const uint32_t nv = size - i < 4 ? (uint32_t)(size - i) : 4u;
        // the four chains advance together (their loads overlap; neighbours usually take the same hops).  Behind the inner
        // launches a chain is ONE hop: no reference leads into the group itself.
        while ((x[0] | x[1] | x[2] | x[3]) & kSymRef) {
            bool more = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!(x[j] & kSymRef)) continue;
                if ((uint32_t)j >= nv) {
                    x[j] = 0;
                    continue;
                }
                const uint64_t cs = s_out[at[j]], d = x[j] & ~kSymRef;
                if (d == 0 || d > cs - frame0) {  // (only after a failed decode: the chunk's status says so)
                    x[j] = 0;
                    continue;
                }
                const uint64_t q = cs - d;
                if (q < below) {
                    x[j] = a.out[q];
                    continue;
                }
                // q lies in an earlier chunk of the group: the last one that starts at or before it
                const uint32_t c = last_at_or_below(s_out, at[j], q);
                x[j] = a.sym[s_elem[c] + (q - s_out[c])];
                at[j] = c;
                more = more || (x[j] & kSymRef);
            }
            if (!more) break;
        }
        uint8_t *dst = a.out + s_out[k] + i;
        if (nv == 4) {
            const uint32_t w = (x[0] & 255u) | ((x[1] & 255u) << 8) | ((x[2] & 255u) << 16) | (x[3] << 24);
            __builtin_memcpy(dst, &w, 4);
        } else {
#pragma unroll
            for (int j = 0; j < 3; j++)
                if ((uint32_t)j < nv) dst[j] = (uint8_t)x[j];
        }
    }
}

template <uint32_t ROWS>
__global__ __launch_bounds__(256) void k_zst_resolve(const ResolveArgs a) {
    resolve_group<ROWS>(a, blockIdx.x);
}

// Groups that do not depend on each other in ONE launch: references never leave their frame, so the k-th groups of all the
// frames of a round resolve side by side (blockIdx.y: the group): a round of 8 MiB frames is a handful of launches.
template <uint32_t ROWS>
__global__ __launch_bounds__(256) void k_zst_resolve_many(const ResolveArgs *__restrict__ args) {
    const ResolveArgs a = args[blockIdx.y];
    if ((uint64_t)blockIdx.x * 1024 * ROWS >= a.n_elems) return;  // (the grid is as wide as the batch's largest group)
    resolve_group<ROWS>(a, blockIdx.x);
}

// Round 5, the chain of launches of ONE big frame cut from ~2 000 to ~130 per GiB.  The in-order launches take GROUPS of G1 (x G2)
// chunks (64 of 256 KiB = 16 MiB for a 1 GiB round: 64 launches), and BEFORE they run, G1 - 1 launches (and, with a second
// level — built, measured slower, off by default — G2 - 1 more) take the groups' INTERNAL references out, every group of the
// round at once (blockIdx.y):
//   level 1, launches j = 1 .. G1-1: chunk j of every run of G1 chunks — a reference that lands in an earlier chunk of its
//     own run reads that chunk's symbol, which launch < j has left as a byte or as a reference in front of the run, and becomes
//     that byte or that reference re-based on its own chunk: ONE hop, because the chunk it reads has been through the same;
//   level 2, launches v = 1 .. G2-1: run v of every group — what still refers into the group (in front of its own run) reads
//     a symbol of an earlier run, which launch < v has left as a byte or as a reference in front of the GROUP: one hop again.
// What refers in front of the pass's scope is left alone.  Afterwards no chain has a link inside a group, and an in-order
// launch resolves a whole group with one hop per reference, into bytes that are final.  (The fully parallel levels of this
// round's first attempt let every symbol FOLLOW its chain through the group: j hops in chunk j, 2x slower than the chain of
// launches; here the order inside the groups is kept and all groups walk it together.)  In place: a launch writes only the
// chunks it was given ([n_ctx, n_sym_chunks) of the scope) and reads only the ones in front of them.
__global__ __launch_bounds__(256) void k_zst_resolve_inner(const ResolveArgs *__restrict__ args, uint32_t *__restrict__ sym_rw) {
    const ResolveArgs a = args[blockIdx.y];
    if ((uint64_t)blockIdx.x * kRowElems >= a.n_elems) return;
    __shared__ uint64_t s_out[kGroupMax], s_elem[kGroupMax], s_size[kGroupMax];
    __shared__ uint64_t s_below_frame[2];
    const uint32_t n = a.n_sym_chunks, n_ctx = a.n_ctx;  // chunks [0, n_ctx): read; [n_ctx, n): rewritten
    if (threadIdx.x < n) {
        const Chunk &c = a.chunks[a.sym_chunks[threadIdx.x]];
        s_out[threadIdx.x] = c.out_off;
        s_elem[threadIdx.x] = c.elem_off;
        s_size[threadIdx.x] = c.size;
        if (threadIdx.x == 0) {
            s_below_frame[0] = c.byte_end > a.final_below ? c.byte_end : a.final_below;
            s_below_frame[1] = c.frame_out_off;
        }
    }
    __syncthreads();
    const uint64_t below = s_below_frame[0], frame0 = s_below_frame[1];
    uint4 rows[kResolveRows];
#pragma unroll
    for (uint32_t row = 0; row < kResolveRows; row++) {
        const uint64_t e = a.elem0 + ((uint64_t)blockIdx.x * kResolveRows + row) * 1024 + (uint64_t)threadIdx.x * 4;
        rows[row] = e < a.elem0 + a.n_elems ? *reinterpret_cast<const uint4 *>(a.sym + e) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t row = 0; row < kResolveRows; row++) {
        const uint64_t e = a.elem0 + ((uint64_t)blockIdx.x * kResolveRows + row) * 1024 + (uint64_t)threadIdx.x * 4;
        if (e >= a.elem0 + a.n_elems) break;
        const uint4 xv = rows[row];
        uint32_t x[4] = {xv.x, xv.y, xv.z, xv.w};
        if (!((x[0] | x[1] | x[2] | x[3]) & kSymRef)) continue;
        const uint32_t k = n_ctx + last_at_or_below(s_elem + n_ctx, n - n_ctx, e);
        const uint64_t i = e - s_elem[k], size = s_size[k];
        if (i >= size) continue;  // padding between two chunks
        const uint32_t nv = size - i < 4 ? (uint32_t)(size - i) : 4u;
        const uint64_t cs = s_out[k], reach = cs - frame0, ctx_end = s_out[n_ctx];
        bool changed = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((uint32_t)j >= nv || !(x[j] & kSymRef)) continue;
            const uint64_t d = x[j] & ~kSymRef;
            if (d == 0 || d > reach) continue;  // (only after a failed decode: the in-order launch makes it a zero)
            const uint64_t q = cs - d;
            // in front of the scope (or in the frame's bytes): a later pass reads it; q >= ctx_end cannot be behind the passes in
            // front of this one (it would lead into the chunks being rewritten) — left alone, the in-order launch follows it
            if (q < below || q >= ctx_end) continue;
            const uint32_t c = last_at_or_below(s_out, n_ctx, q);
            const uint32_t y = a.sym[s_elem[c] + (q - s_out[c])];
            if (y & kSymRef) {
                const uint64_t d2 = y & ~kSymRef, co = s_out[c];
                // (chunk c's reference leads in front of the scope; an invalid one — a failed decode — becomes a zero here as there)
                x[j] = (d2 == 0 || d2 > co - frame0) ? 0u : (kSymRef | (uint32_t)(cs - (co - d2)));
            } else {
                x[j] = y;
            }
            changed = true;
        }
        if (changed) *reinterpret_cast<uint4 *>(sym_rw + e) = make_uint4(x[0], x[1], x[2], x[3]);
    }
}

static double now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static double sync_ms(hipStream_t st) {
    (void)hipStreamSynchronize(st);
    return now_ms();
}

static const char *status_text(uint32_t code) {
    switch (code) {
        case kErrChecksum: return "Restored data doesn't match checksum";
        case kErrSize: return "Data corruption detected (size)";
        case kErrOffset: return "Data corruption detected (offset)";
        case kErrHuffman: return "Data corruption detected (Huffman literals)";
        case kErrFse: return "Data corruption detected (FSE table)";
        case kErrSequences: return "Data corruption detected (sequences)";
        default: return "Data corruption detected";
    }
}

}  // namespace zst
}  // namespace exg

// ---- C-ABI ----------------------------------------------------------------------------------------------------------
// All frames of a zstd stream (concatenated frames and skippable frames included, as ZSTD_decompressStream reads them).
// h_comp: the compressed bytes on the host (only headers are read: the frame / block walk); d_comp: the same bytes on the
// device, readable to n + 16.  On success *d_out is a hipMalloc'd buffer the caller hipFree()s (*produced bytes + 64
// zeroed).  Synchronises the stream.  Errors: EXG_E_PARSE with libzstd's wording in exg_last_error_message().
namespace exg {
namespace zst {

int host_verify(const void *d_out, const std::vector<PendingCheck> &pending, int device, std::string *err) {
    if (pending.empty()) return EXG_OK;
    exg_rd::DeviceGuard guard(device);
    hipStream_t st = nullptr;
    if (exg_rd::stream_pool()->take(device, &st) != hipSuccess) {
        *err = "zstd checksum verification: no HIP stream";
        return EXG_E_HIP;
    }
    constexpr size_t kPiece = 32u << 20;
    auto pool = exg_rd::global_pool();
    char *buf[2] = {nullptr, nullptr};
    size_t cap[2] = {kPiece, kPiece};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = EXG_OK;
    for (int k = 0; k < 2 && !rc; k++) {
        buf[k] = pool->take(&cap[k]);
        if (!buf[k] || hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) {
            *err = "zstd checksum verification: out of pinned host memory";
            rc = EXG_E_HIP;
        }
    }
    for (size_t f = 0; f < pending.size() && !rc; f++) {
        const PendingCheck &P = pending[f];
        const uint8_t *src = (const uint8_t *)d_out + P.out_off;
        Xxh64 h;
        const uint64_t n_pieces = (P.size + kPiece - 1) / kPiece;
        auto issue = [&](uint64_t i) {
            const uint64_t off = i * kPiece, len = std::min<uint64_t>(kPiece, P.size - off);
            hipError_t e = hipMemcpyAsync(buf[i & 1], src + off, len, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipEventRecord(ev[i & 1], st);
            return e;
        };
        hipError_t e = n_pieces ? issue(0) : hipSuccess;
        for (uint64_t i = 0; i < n_pieces && e == hipSuccess; i++) {
            if (i + 1 < n_pieces) e = issue(i + 1);  // the next piece travels while this one is hashed
            if (e == hipSuccess) e = hipEventSynchronize(ev[i & 1]);
            if (e == hipSuccess) h.update((const uint8_t *)buf[i & 1], (size_t)std::min<uint64_t>(kPiece, P.size - i * kPiece));
        }
        if (e != hipSuccess) {
            *err = std::string("zstd checksum verification: ") + hipGetErrorString(e);
            rc = EXG_E_HIP;
        } else if ((uint32_t)h.digest() != P.expect) {
            *err = std::string(status_text(kErrChecksum)) + " (zstd frame " + std::to_string(P.frame) + ")";
            rc = EXG_E_PARSE;
        }
    }
    (void)hipStreamSynchronize(st);
    for (int k = 0; k < 2; k++) {
        if (ev[k]) (void)hipEventDestroy(ev[k]);
        if (buf[k]) pool->give(buf[k], cap[k]);
    }
    exg_rd::stream_pool()->give(device, st);
    return rc;
}

// ---- a round in three phases ----------------------------------------------------------------------------------------------------
// (Written so that a reader could run round n + 1's entropy stages beside round n's execution on a second stream.  Measured
// through the reader — two slots, two streams, two compressed windows — it gave nothing: 10.3 against 10.2 GB/s on a 4 GB
// frame.  Every one of these kernels alone fills the CUs' LDS with its one-lane wavefronts (k_zst_exec: 16.5 KiB per wave),
// so the other stream's kernels queue up behind instead of running beside.  The reader calls decode_round; the phases stay
// as the function's structure.)
// begin:   the entropy stages + the scan (waits for them: the host needs the block sizes), the host's chunk plan.  Fills
//          R.blocks' results, R.rep_out, R.frames[].out_off / out_size and R.produced — all a caller needs to prepare the
//          round behind this one (its history size, its repeat offsets).
// enqueue: execution, resolve, on-device checksums — launches only, on the round's stream.  R.d_history (when R.history != 0)
//          must hold the window by now.
// wait:    waits for them, reads the statuses; R.d_buf is the caller's from here on.
// Temporaries come from (and go back to) the process-wide device pool: hipMalloc / hipFree of the multi-GB symbol and
// sequence buffers cost more than the decode (the same 4 GB decode: 0.20 s with warm buffers, 0.4 - 1.2 s without).  The
// stream is waited for before a block goes back: an error return may leave kernels in flight.
namespace {
struct DevTmp {
    int dev;
    hipStream_t st;
    void *p = nullptr;
    size_t sz = 0;
    DevTmp(int d, hipStream_t s) : dev(d), st(s) {}
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() {
        if (!p) return;
        (void)hipStreamSynchronize(st);
        exg_rd::dev_pool()->give(dev, p, sz);
    }
    hipError_t alloc(size_t bytes) {
        sz = bytes ? bytes : 16;
        p = exg_rd::dev_pool()->take(dev, sz);
        return p ? hipSuccess : hipErrorOutOfMemory;
    }
};
}  // namespace

struct RoundCtx {
    int dev;
    hipStream_t st;
    DevTmp d_blocks, d_lit, d_ll, d_ml, d_off, d_meta, d_frames, d_chunks, d_status, d_out, d_sym, d_lookup, d_rargs;
    ScanState state;
    uint64_t H = 0, total = 0, target = 0;
    uint32_t nb = 0, nx = 0, nf = 0, nc = 0;
    std::vector<Chunk> chunks;
    std::vector<Frame> dframes;
    // the statuses come back into pinned memory: a copy into pageable memory would hold the host until the stream is done
    char *h_status = nullptr;
    size_t h_status_cap = 0;
    size_t n_rounds = 0;
    double t_begin = 0, t_entropy = 0;
    // between the two halves of the enqueue phase
    struct SymRound {
        uint32_t c0, c1;
        uint32_t list0, n_list;  // its symbolic chunks in sym_list
        uint64_t elems;
    };
    std::vector<SymRound> rounds;
    std::vector<uint32_t> sym_list;
    std::vector<uint64_t> csize;
    uint8_t *out_bytes = nullptr;
    uint32_t *d_sym_list = nullptr;
    bool exec_enqueued = false;
    char *h_rargs = nullptr;  // the resolve launches' arguments (pinned), when groups of several frames share launches
    size_t h_rargs_cap = 0;
    hipStream_t hp = nullptr;  // the resolve launches' own high-priority stream (between two events on st; decode_round_enqueue_resolve)
    hipEvent_t hp_ev = nullptr;
    RoundCtx(int d, hipStream_t s)
        : dev(d), st(s), d_blocks(d, s), d_lit(d, s), d_ll(d, s), d_ml(d, s), d_off(d, s), d_meta(d, s), d_frames(d, s), d_chunks(d, s), d_status(d, s),
          d_out(d, s), d_sym(d, s), d_lookup(d, s), d_rargs(d, s) {}
    ~RoundCtx() {
        if (hp) exg_rd::stream_pool()->give(dev, hp, /*high=*/true);  // (synchronises it)
        if (hp_ev) (void)hipEventDestroy(hp_ev);
        if (h_status || h_rargs) (void)hipStreamSynchronize(st);
        if (h_status) exg_rd::global_pool()->give(h_status, h_status_cap);
        if (h_rargs) exg_rd::global_pool()->give(h_rargs, h_rargs_cap);
    }
};
void decode_round_abandon(RoundCtx *c) { delete c; }

int decode_round_begin(Round &R, void *stream_v, RoundCtx **out_ctx) {
    hipStream_t st = (hipStream_t)stream_v;
    const uint8_t *d_comp = (const uint8_t *)R.d_comp;
    *out_ctx = nullptr;
    R.d_buf = nullptr;
    R.alloc = 0;
    R.produced = 0;
    if (!d_comp || (R.front_reserve & 15) || R.n_extra > R.blocks.size()) {
        set_error("zstd decode: bad arguments");
        return EXG_E_INVALID_ARG;
    }
    static const bool trace = getenv("EXG_TRACE") != nullptr;
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    std::unique_ptr<RoundCtx> ctx(new RoundCtx(cur_dev, st));
    RoundCtx &C = *ctx;
    C.t_begin = trace ? now_ms() : 0;
    std::vector<Block> &blocks = R.blocks;
    const uint32_t nb = (uint32_t)blocks.size(), nx = R.n_extra, nf = (uint32_t)R.frames.size();
    C.nb = nb, C.nx = nx, C.nf = nf;
    // where every block's literals and sequences go (the sources in front hold none of their own)
    uint64_t lit_bytes = 0, n_seq = 0;
    for (uint32_t b = nx; b < nb; b++) {
        blocks[b].lit_off = lit_bytes;
        blocks[b].seq_off = n_seq;
        if (blocks[b].type == 2) lit_bytes += (blocks[b].lit_regen + 15) & ~15u, n_seq += blocks[b].nseq;
    }
    EXG_HIP_CHECK(C.d_meta.alloc(64));
    const uint64_t H = R.history;
    C.H = H;
    ScanState &state = C.state;
    state.run = H;  // buffer coordinates: [0, H) = what earlier rounds left of the frame that goes on, then this round's bytes
    state.rep[0] = R.rep_in[0], state.rep[1] = R.rep_in[1], state.rep[2] = R.rep_in[2], state.pad = 0;
    if (nb > nx) {
        EXG_HIP_CHECK(C.d_blocks.alloc((size_t)nb * sizeof(Block)));
        EXG_HIP_CHECK(C.d_lit.alloc(lit_bytes + 64));
        EXG_HIP_CHECK(C.d_ll.alloc(n_seq * 4 + 16));
        EXG_HIP_CHECK(C.d_ml.alloc(n_seq * 4 + 16));
        EXG_HIP_CHECK(C.d_off.alloc(n_seq * 4 + 16));
        EXG_HIP_CHECK(hipMemcpyAsync(C.d_blocks.p, blocks.data(), (size_t)nb * sizeof(Block), hipMemcpyHostToDevice, st));
        EXG_HIP_CHECK(hipMemcpyAsync(C.d_meta.p, &state, sizeof state, hipMemcpyHostToDevice, st));
        const uint32_t grid = nb - nx < 16384 ? nb - nx : 16384;
        // the two entropy stages are independent of each other (both run one or four LANES per wavefront: the chip is far
        // from full with either): literals on a second stream beside the sequences
        const int dev = cur_dev;
        hipStream_t st2 = nullptr;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        const bool side = exg_rd::stream_pool()->take(dev, &st2) == hipSuccess && hipEventCreateWithFlags(&ev0, hipEventDisableTiming) == hipSuccess &&
                          hipEventCreateWithFlags(&ev1, hipEventDisableTiming) == hipSuccess;
        if (side) {
            (void)hipEventRecord(ev0, st);
            (void)hipStreamWaitEvent(st2, ev0, 0);
        }
        hipLaunchKernelGGL(k_zst_literals, dim3((grid + kLitG - 1) / kLitG), dim3(64), 0, side ? st2 : st, d_comp, (Block *)C.d_blocks.p, nx, nb, (uint8_t *)C.d_lit.p);
        hipLaunchKernelGGL(k_zst_sequences, dim3((grid + kSeqG - 1) / kSeqG), dim3(64), 0, st, d_comp, (Block *)C.d_blocks.p, nx, nb, (uint32_t *)C.d_ll.p, (uint32_t *)C.d_ml.p,
                           (uint32_t *)C.d_off.p);
        if (side) {
            (void)hipEventRecord(ev1, st2);
            (void)hipStreamWaitEvent(st, ev1, 0);
        }
        struct SideGuard {
            int dev;
            hipStream_t s;
            hipEvent_t a, b;
            ~SideGuard() {
                if (a) (void)hipEventDestroy(a);
                if (b) (void)hipEventDestroy(b);
                if (s) exg_rd::stream_pool()->give(dev, s);  // (synchronises it)
            }
        } side_guard{dev, st2, ev0, ev1};
        hipLaunchKernelGGL(k_zst_scan, dim3(1), dim3(64), 0, st, (Block *)C.d_blocks.p, nx, nb, (ScanState *)C.d_meta.p);
        EXG_HIP_CHECK(hipGetLastError());
        // the block table and the scan's state come back by a kernel's stores into pinned memory (exg_crc32.hip: post_to_host) — a
        // copy is a packet on an SDMA engine, where it may queue behind a 1 GiB segment on its way to the host (HostMirror)
        static_assert(sizeof(Block) % 4 == 0 && sizeof(ScanState) % 4 == 0, "posted word by word");
        struct Back {
            char *p = nullptr;
            size_t cap = 0;
            ~Back() {
                if (p) exg_rd::global_pool()->give(p, cap);
            }
        } back;
        const size_t blocks_bytes = ((size_t)nb * sizeof(Block) + 15) & ~(size_t)15;
        back.cap = blocks_bytes + sizeof state + 64;
        back.p = (char *)exg_rd::global_pool()->take(&back.cap);
        if (!back.p) {
            set_error("zstd decode: out of pinned host memory");
            return EXG_E_NOMEM;
        }
        int prc = post_to_host(back.p, C.d_blocks.p, (size_t)nb * sizeof(Block), st);
        if (!prc) prc = post_to_host(back.p + blocks_bytes, C.d_meta.p, sizeof state, st);
        const hipError_t she = hipStreamSynchronize(st);  // (before `back` goes to the pool, whatever happened)
        if (prc) return prc;
        EXG_HIP_CHECK(she);
        memcpy(blocks.data(), back.p, (size_t)nb * sizeof(Block));
        memcpy(&state, back.p + blocks_bytes, sizeof state);
        for (uint32_t b = nx; b < nb; b++)
            if (blocks[b].status) {
                set_error("%s (zstd block %llu at byte %llu)", status_text(blocks[b].status), (unsigned long long)(R.first_block_id + (b - nx)),
                          (unsigned long long)(R.comp_base + blocks[b].src_off));
                return EXG_E_PARSE;
            }
    }
    const uint64_t total = state.run - H;
    C.total = total;
    R.rep_out[0] = state.rep[0], R.rep_out[1] = state.rep[1], R.rep_out[2] = state.rep[2];
    // frames: where their content lies.  Chunks: a frame is cut into runs of whole blocks of about `target` bytes, one
    // wavefront each.  The first chunk of a frame writes bytes; every other one cannot know what lies in front of it while it
    // runs (or, the first chunk of a frame that goes on from the round in front: not yet), so it writes 32-bit symbols —
    // a byte, or "the byte d in front of this chunk" — that are resolved chunk by chunk, in order, once everything in front
    // is final (k_zst_resolve).
    static const uint64_t target_env = getenv("EXG_ZSTD_CHUNK_BYTES") ? strtoull(getenv("EXG_ZSTD_CHUNK_BYTES"), nullptr, 10) : 0;
    const uint64_t target = target_env ? target_env : std::min<uint64_t>(2u << 20, std::max<uint64_t>(128u << 10, total / 4096));
    C.target = target;
    std::vector<Chunk> &chunks = C.chunks;
    std::vector<Frame> &dframes = C.dframes;
    dframes.resize(nf);
    uint64_t sym_elems = 0;
    for (uint32_t f = 0; f < nf; f++) {
        RoundFrame &F = R.frames[f];
        F.out_off = F.n_blocks ? blocks[F.first_block].out_off : state.run;
        uint64_t sz = 0;
        for (uint32_t b = 0; b < F.n_blocks; b++) sz += blocks[F.first_block + b].out_size;
        F.out_size = sz;
        // (a match may reach back to the first byte of the frame's content — as far as it is still there: a frame that
        // began in an earlier round has its window, not its beginning, in front of this round's bytes)
        const uint64_t frame0 = F.begins ? F.out_off : F.out_off - std::min<uint64_t>(F.history, F.out_off);
        memset(&dframes[f], 0, sizeof(Frame));
        dframes[f].out_off = F.out_off;
        dframes[f].out_size = sz;
        dframes[f].checksum = F.checksum;
        dframes[f].has_checksum = F.has_checksum && F.begins && F.ends;  // (a frame that spans rounds is hashed by the caller)
        for (uint32_t b = 0; b < F.n_blocks;) {
            Chunk c;
            memset(&c, 0, sizeof c);
            c.first_block = F.first_block + b;
            c.out_off = blocks[c.first_block].out_off;
            c.frame_out_off = frame0;
            uint64_t got = 0;
            while (b < F.n_blocks && (got < target || c.n_blocks == 0)) got += blocks[F.first_block + b].out_size, b++, c.n_blocks++;
            // (the part of a frame that an earlier round began has the window in front of it: its first chunk writes symbols
            // too, so that no chunk's execution waits for the round in front — only the resolve does)
            c.symbolic = c.first_block != F.first_block || !F.begins;
            c.elem_off = c.symbolic ? sym_elems : c.out_off;
            if (c.symbolic) sym_elems += got;
            chunks.push_back(c);
        }
    }
    C.t_entropy = trace ? now_ms() : 0;
    R.produced = total;
    *out_ctx = ctx.release();
    return EXG_OK;
}

// enqueue, first half: the execution of the chunks (of the first symbol round: there is one unless the symbol buffer is
// capped).  Needs nothing of the round in front — R.d_history is not read.
int decode_round_enqueue_exec(Round &R, RoundCtx *ctx_p) {
    std::unique_ptr<RoundCtx> ctx(ctx_p);
    RoundCtx &C = *ctx;
    hipStream_t st = C.st;
    const uint8_t *d_comp = (const uint8_t *)R.d_comp;
    std::vector<Block> &blocks = R.blocks;
    std::vector<Chunk> &chunks = C.chunks;
    const uint64_t H = C.H, total = C.total;
    const uint32_t nf = C.nf;
    // (allocated at the device pool's size class: the caller hands the buffer back to that pool)
    EXG_HIP_CHECK(C.d_out.alloc(R.front_reserve + H + total + 64));
    uint8_t *const out_bytes = (uint8_t *)C.d_out.p + R.front_reserve;  // buffer coordinate 0
    C.out_bytes = out_bytes;
    EXG_HIP_CHECK(hipMemsetAsync(out_bytes + H + total, 0, 64, st));
    const uint32_t nc = (uint32_t)chunks.size();
    C.nc = nc;
    if (nc) {
        // symbol chunks are executed and resolved in rounds that fit the symbol buffer (4 bytes per output byte)
        static const uint64_t sym_cap_env = getenv("EXG_ZSTD_SYM_BYTES") ? strtoull(getenv("EXG_ZSTD_SYM_BYTES"), nullptr, 10) : (32ull << 30);
        const uint64_t sym_cap = std::max<uint64_t>(sym_cap_env / 4, 1);
        EXG_HIP_CHECK(C.d_chunks.alloc((size_t)nc * sizeof(Chunk)));
        EXG_HIP_CHECK(C.d_frames.alloc((size_t)nf * sizeof(Frame)));
        EXG_HIP_CHECK(C.d_status.alloc(((size_t)nc + nf) * 4));
        EXG_HIP_CHECK(hipMemsetAsync(C.d_status.p, 0, ((size_t)nc + nf) * 4, st));
        // chunk sizes (bytes of output) and the rounds
        std::vector<uint64_t> &csize = C.csize;
        csize.resize(nc);
        for (uint32_t c = 0; c < nc; c++) {
            uint64_t sz = 0;
            for (uint32_t b = 0; b < chunks[c].n_blocks; b++) sz += blocks[chunks[c].first_block + b].out_size;
            csize[c] = sz;
            chunks[c].size = sz;
        }
        {   // a frame's first chunk holds bytes: its end is where the frame's symbols begin (a frame that goes on from the round
            // in front has none: an earlier end is a smaller claim, and what lies in front of a resolve group is final anyway)
            uint64_t end = 0;
            for (uint32_t c = 0; c < nc; c++) {
                if (!chunks[c].symbolic) end = chunks[c].out_off + csize[c];
                chunks[c].byte_end = end;
            }
        }
        typedef RoundCtx::SymRound SymRound;
        std::vector<SymRound> &rounds = C.rounds;
        std::vector<uint32_t> &sym_list = C.sym_list;
        uint64_t sym_need = 0;
        {
            SymRound Q{0, 0, 0, 0, 0};
            for (uint32_t c = 0; c < nc; c++) {
                if (chunks[c].symbolic && csize[c]) {
                    if (Q.elems && Q.elems + csize[c] + 4 > sym_cap) {
                        Q.c1 = c;
                        rounds.push_back(Q);
                        Q = SymRound{c, 0, (uint32_t)sym_list.size(), 0, 0};
                    }
                    chunks[c].elem_off = Q.elems;
                    Q.elems += (csize[c] + 3) & ~3ull;
                    Q.n_list++;
                    sym_list.push_back(c);
                    sym_need = std::max(sym_need, Q.elems);
                }
            }
            Q.c1 = nc;
            rounds.push_back(Q);
        }
        C.n_rounds = rounds.size();
        if (sym_need) EXG_HIP_CHECK(C.d_sym.alloc(sym_need * 4 + 64));
        // the tables below travel from pinned memory (pageable sources would be staged: the calls would wait)
        const size_t list_bytes = (sym_list.size() + 1) * 4, chunk_bytes = (size_t)nc * sizeof(Chunk), frame_bytes = (size_t)nf * sizeof(Frame),
                     status_bytes = ((size_t)nc + nf) * 4;
        C.h_status_cap = status_bytes + list_bytes + chunk_bytes + frame_bytes + 64;
        C.h_status = exg_rd::global_pool()->take(&C.h_status_cap);
        if (!C.h_status) {
            set_error("zstd decode: out of pinned host memory");
            return EXG_E_NOMEM;
        }
        char *h_list = C.h_status + status_bytes, *h_chunks = h_list + list_bytes, *h_frames = h_chunks + chunk_bytes;
        if (!sym_list.empty()) memcpy(h_list, sym_list.data(), sym_list.size() * 4);
        memcpy(h_chunks, chunks.data(), chunk_bytes);
        if (frame_bytes) memcpy(h_frames, C.dframes.data(), frame_bytes);
        EXG_HIP_CHECK(C.d_lookup.alloc(list_bytes));
        C.d_sym_list = (uint32_t *)C.d_lookup.p;
        if (!sym_list.empty()) EXG_HIP_CHECK(hipMemcpyAsync(C.d_sym_list, h_list, sym_list.size() * 4, hipMemcpyHostToDevice, st));
        EXG_HIP_CHECK(hipMemcpyAsync(C.d_chunks.p, h_chunks, chunk_bytes, hipMemcpyHostToDevice, st));
        EXG_HIP_CHECK(hipMemcpyAsync(C.d_frames.p, h_frames, frame_bytes, hipMemcpyHostToDevice, st));
        const SymRound &Q = rounds[0];
        const uint32_t cnt = Q.c1 - Q.c0, grid = cnt < 16384 ? cnt : 16384;
        if (cnt)
            hipLaunchKernelGGL(k_zst_exec, dim3(grid), dim3(64), 0, st, d_comp, (const Block *)C.d_blocks.p, (const Chunk *)C.d_chunks.p + Q.c0, cnt,
                               (const uint8_t *)C.d_lit.p, (const uint32_t *)C.d_ll.p, (const uint32_t *)C.d_ml.p, (const uint32_t *)C.d_off.p,
                               out_bytes, (uint32_t *)C.d_sym.p, (uint32_t *)C.d_status.p + Q.c0);
        EXG_HIP_CHECK(hipGetLastError());
    }
    C.exec_enqueued = true;
    ctx.release();
    return EXG_OK;
}

// enqueue, second half: the window in front (R.d_history must hold it by now), the resolve launches, the checksums
int decode_round_enqueue_resolve(Round &R, RoundCtx *ctx_p) {
    std::unique_ptr<RoundCtx> ctx(ctx_p);
    RoundCtx &C = *ctx;
    hipStream_t st = C.st;
    // Round 6: the resolve launches — a chain of ~130 short dependent launches per round — go out on a HIGH-PRIORITY stream of their
    // own, between two events on the round's stream: the round behind runs its entropy stages and its execution (a chip full of
    // wavefronts that live for milliseconds) beside them, and every launch of the chain waited for slots among those.  A 4 GB frame
    // into DataChunks 142-146 -> 133-137 ms (COUNT(*) unchanged: 91 ms).  EXG_ZSTD_RESOLVE_PRIORITY=0: on the round's stream (A/B)
    static const bool resolve_priority = !getenv("EXG_ZSTD_RESOLVE_PRIORITY") || atoi(getenv("EXG_ZSTD_RESOLVE_PRIORITY")) != 0;
    struct Rejoin {  // whatever this half enqueues on the high-priority stream, C.st goes on behind it
        RoundCtx &C;
        bool on = false;
        ~Rejoin() {
            if (on && (hipEventRecord(C.hp_ev, C.hp) != hipSuccess || hipStreamWaitEvent(C.st, C.hp_ev, 0) != hipSuccess)) (void)hipStreamSynchronize(C.hp);
        }
    } rejoin{C};
    if (resolve_priority && C.nc) {
        if (exg_rd::stream_pool()->take(C.dev, &C.hp, /*high=*/true) == hipSuccess && hipEventCreateWithFlags(&C.hp_ev, hipEventDisableTiming) == hipSuccess &&
            hipEventRecord(C.hp_ev, C.st) == hipSuccess && hipStreamWaitEvent(C.hp, C.hp_ev, 0) == hipSuccess) {
            st = C.hp;
            rejoin.on = true;
        } else {
            (void)hipGetLastError();
        }
    }
    const uint8_t *d_comp = (const uint8_t *)R.d_comp;
    std::vector<Chunk> &chunks = C.chunks;
    const uint64_t H = C.H;
    const uint32_t nf = C.nf, nc = C.nc;
    uint8_t *const out_bytes = C.out_bytes;
    if (H) EXG_HIP_CHECK(hipMemcpyAsync(out_bytes, R.d_history, H, hipMemcpyDeviceToDevice, st));
    if (nc) {
        typedef RoundCtx::SymRound SymRound;
        const std::vector<uint32_t> &sym_list = C.sym_list;
        const std::vector<uint64_t> &csize = C.csize;
        const uint32_t *d_sym_list = C.d_sym_list;
        const size_t status_bytes = ((size_t)nc + nf) * 4;
        bool first = true;
        for (const SymRound &Q : C.rounds) {
            const uint32_t cnt = Q.c1 - Q.c0, grid = cnt < 16384 ? cnt : 16384;
            if (!cnt) continue;
            if (!first)  // (the first symbol round's execution went out with the first half)
                hipLaunchKernelGGL(k_zst_exec, dim3(grid), dim3(64), 0, st, d_comp, (const Block *)C.d_blocks.p, (const Chunk *)C.d_chunks.p + Q.c0, cnt,
                                   (const uint8_t *)C.d_lit.p, (const uint32_t *)C.d_ll.p, (const uint32_t *)C.d_ml.p, (const uint32_t *)C.d_off.p,
                                   out_bytes, (uint32_t *)C.d_sym.p, (uint32_t *)C.d_status.p + Q.c0);
            first = false;
            static const uint64_t group_bytes = getenv("EXG_ZSTD_RESOLVE_BYTES") ? strtoull(getenv("EXG_ZSTD_RESOLVE_BYTES"), nullptr, 10) : (512ull << 10);
            // EXG_ZSTD_RESOLVE_INNER=0: round 4's groups of ~512 KiB whose internal chains every thread follows (A/B);
            // EXG_ZSTD_GROUP_CHUNKS=G1[,G2]: chunks per run and runs per group instead of the choice below
            static const bool inner_on = !getenv("EXG_ZSTD_RESOLVE_INNER") || atoi(getenv("EXG_ZSTD_RESOLVE_INNER")) != 0;
            static const char *group_env = getenv("EXG_ZSTD_GROUP_CHUNKS");
            uint32_t G1 = 0, G2 = 1;  // G1 = 0: groups by bytes, no inner launches
            if (inner_on) {
                uint32_t longest = 0;
                for (uint32_t k0 = 0; k0 < Q.n_list;) {
                    const uint64_t frame0 = chunks[sym_list[Q.list0 + k0]].frame_out_off;
                    uint32_t k1 = k0;
                    while (k1 < Q.n_list && chunks[sym_list[Q.list0 + k1]].frame_out_off == frame0) k1++;
                    longest = std::max(longest, k1 - k0);
                    k0 = k1;
                }
                // dependent launches of the longest frame: (G1 - 1) + (G2 - 1) + longest / (G1 G2).  ONE level of inner launches with
                // G1 ~ sqrt(longest) (4 096 chunks: 63 + 64 launches); a second level (G1 = G2 = 16: 46 launches) reads every symbol
                // once more and measured SLOWER on a 4 GB frame (113-122 ms against 102-108; the launches are no longer what the
                // round waits for) — EXG_ZSTD_GROUP_CHUNKS=16,16 builds it
                G1 = std::min<uint32_t>(kGroupMax, std::max<uint32_t>(2, (uint32_t)std::ceil(std::sqrt((double)longest))));
                if (group_env) {
                    G1 = (uint32_t)std::max(2, atoi(group_env));
                    G2 = strchr(group_env, ',') ? (uint32_t)std::max(1, atoi(strchr(group_env, ',') + 1)) : 1;
                    G1 = std::min(G1, kGroupMax);
                    G2 = std::min(G2, kGroupMax / G1);
                }
            }
            // groups of consecutive chunks of ONE frame (a reference never leaves its frame); batch u = the u-th group of every
            // frame of the round: its groups are independent of each other, the batches run in order.  inner1[j - 1]: chunk j of
            // every run that has one; inner2[v - 1]: run v of every group that has one (k_zst_resolve_inner)
            std::vector<std::vector<ResolveArgs>> batches, inner1, inner2;
            for (uint32_t k0 = 0; k0 < Q.n_list;) {
                const uint64_t frame0 = chunks[sym_list[Q.list0 + k0]].frame_out_off;
                for (uint32_t u = 0; k0 < Q.n_list && chunks[sym_list[Q.list0 + k0]].frame_out_off == frame0; u++) {
                    uint32_t k1 = k0;
                    uint64_t bytes = 0;
                    while (k1 < Q.n_list && (G1 ? k1 - k0 < G1 * G2 : (k1 - k0 < kGroupMaxBytes && (bytes < group_bytes || k1 == k0))) &&
                           chunks[sym_list[Q.list0 + k1]].frame_out_off == frame0)
                        bytes += csize[sym_list[Q.list0 + k1]], k1++;
                    auto slots_end = [&](uint32_t k) {  // the end of list entry k's symbol slots
                        const Chunk &c = chunks[sym_list[Q.list0 + k]];
                        return c.elem_off + ((c.size + 3) & ~3ull);
                    };
                    const Chunk &first_c = chunks[sym_list[Q.list0 + k0]];
                    ResolveArgs ra;
                    ra.sym = (const uint32_t *)C.d_sym.p;
                    ra.chunks = (const Chunk *)C.d_chunks.p;
                    ra.sym_chunks = d_sym_list + Q.list0 + k0;
                    ra.n_sym_chunks = k1 - k0;
                    ra.n_ctx = 0;
                    ra.out = out_bytes;
                    ra.final_below = first_c.out_off;
                    ra.elem0 = first_c.elem_off;
                    ra.n_elems = slots_end(k1 - 1) - first_c.elem_off;
                    if (batches.size() <= u) batches.resize(u + 1);
                    batches[u].push_back(ra);
                    for (uint32_t v = 0, r0 = k0; G1 && r0 < k1; v++, r0 += G1) {  // the group's runs
                        const uint32_t r1 = std::min(k1, r0 + G1);
                        const Chunk &run_c = chunks[sym_list[Q.list0 + r0]];
                        for (uint32_t m = 1; m < r1 - r0; m++) {
                            ResolveArgs ia = ra;
                            ia.sym_chunks = d_sym_list + Q.list0 + r0;
                            ia.n_sym_chunks = m + 1;
                            ia.n_ctx = m;
                            ia.final_below = run_c.out_off;
                            ia.elem0 = chunks[sym_list[Q.list0 + r0 + m]].elem_off;
                            ia.n_elems = slots_end(r0 + m) - ia.elem0;
                            if (inner1.size() < m) inner1.resize(m);
                            inner1[m - 1].push_back(ia);
                        }
                        if (v) {
                            ResolveArgs ia = ra;
                            ia.n_sym_chunks = r1 - k0;
                            ia.n_ctx = r0 - k0;
                            ia.elem0 = run_c.elem_off;
                            ia.n_elems = slots_end(r1 - 1) - ia.elem0;
                            if (inner2.size() < v) inner2.resize(v);
                            inner2[v - 1].push_back(ia);
                        }
                    }
                    k0 = k1;
                }
            }
            size_t n_shared = 0;  // arguments that travel through memory (the inner launches'; batches of more than one group)
            for (const auto &b : inner1) n_shared += b.size();
            for (const auto &b : inner2) n_shared += b.size();
            for (const auto &b : batches) n_shared += b.size() > 1 ? b.size() : 0;
            const ResolveArgs *d_args = nullptr;
            if (n_shared) {
                if (C.h_rargs) {  // (a second symbol round: the first one's launches must have read theirs)
                    EXG_HIP_CHECK(hipStreamSynchronize(st));
                    exg_rd::global_pool()->give(C.h_rargs, C.h_rargs_cap);
                    C.h_rargs = nullptr;
                }
                C.h_rargs_cap = n_shared * sizeof(ResolveArgs);
                C.h_rargs = exg_rd::global_pool()->take(&C.h_rargs_cap);
                if (!C.h_rargs) {
                    set_error("zstd decode: out of pinned host memory");
                    return EXG_E_NOMEM;
                }
                EXG_HIP_CHECK(C.d_rargs.alloc(n_shared * sizeof(ResolveArgs)));
                size_t at = 0;
                for (const auto *set : {&inner1, &inner2})
                    for (const auto &b : *set) memcpy(C.h_rargs + at * sizeof(ResolveArgs), b.data(), b.size() * sizeof(ResolveArgs)), at += b.size();
                for (const auto &b : batches)
                    if (b.size() > 1) memcpy(C.h_rargs + at * sizeof(ResolveArgs), b.data(), b.size() * sizeof(ResolveArgs)), at += b.size();
                EXG_HIP_CHECK(hipMemcpyAsync(C.d_rargs.p, C.h_rargs, n_shared * sizeof(ResolveArgs), hipMemcpyHostToDevice, st));
                d_args = (const ResolveArgs *)C.d_rargs.p;
            }
            size_t at = 0;
            // (grid.y <= 65535: more groups than that in one batch are cut into several launches)
            for (const auto *set : {&inner1, &inner2})
                for (const auto &b : *set) {
                    uint64_t widest = 0;
                    for (const ResolveArgs &ra : b) widest = std::max<uint64_t>(widest, ra.n_elems);
                    for (size_t g0 = 0; g0 < b.size(); g0 += 32768) {
                        const size_t ng = std::min<size_t>(32768, b.size() - g0);
                        hipLaunchKernelGGL(k_zst_resolve_inner, dim3((uint32_t)((widest + kRowElems - 1) / kRowElems), (uint32_t)ng), dim3(256), 0, st,
                                           d_args + at + g0, (uint32_t *)C.d_sym.p);
                    }
                    at += b.size();
                }
            for (const auto &b : batches) {
                if (b.size() == 1) {
                    if (G1) hipLaunchKernelGGL(k_zst_resolve<kResolveRows>, dim3((uint32_t)((b[0].n_elems + kRowElems - 1) / kRowElems)), dim3(256), 0, st, b[0]);
                    else hipLaunchKernelGGL(k_zst_resolve<1>, dim3((uint32_t)((b[0].n_elems + 1023) / 1024)), dim3(256), 0, st, b[0]);
                    continue;
                }
                uint64_t widest = 0;
                for (const ResolveArgs &ra : b) widest = std::max<uint64_t>(widest, ra.n_elems);
                for (size_t g0 = 0; g0 < b.size(); g0 += 32768) {
                    const size_t ng = std::min<size_t>(32768, b.size() - g0);
                    if (G1)
                        hipLaunchKernelGGL(k_zst_resolve_many<kResolveRows>, dim3((uint32_t)((widest + kRowElems - 1) / kRowElems), (uint32_t)ng), dim3(256), 0, st,
                                           d_args + at + g0);
                    else
                        hipLaunchKernelGGL(k_zst_resolve_many<1>, dim3((uint32_t)((widest + 1023) / 1024), (uint32_t)ng), dim3(256), 0, st, d_args + at + g0);
                }
                at += b.size();
            }
        }
        hipLaunchKernelGGL(k_zst_xxh64, dim3(nf < 16384 ? nf : 16384), dim3(64), 0, st, (const uint8_t *)out_bytes, (const Frame *)C.d_frames.p, nf,
                           R.verify_max, (uint32_t *)C.d_status.p + nc);
        EXG_HIP_CHECK(hipGetLastError());
        {
            const int prc = post_to_host(C.h_status, C.d_status.p, status_bytes, st);
            if (prc) return prc;
        }
    }
    ctx.release();
    return EXG_OK;
}

// (on an error the context is gone: its blocks went back to the pool behind a synchronised stream)
int decode_round_enqueue(Round &R, RoundCtx *ctx) {
    int rc = decode_round_enqueue_exec(R, ctx);
    if (!rc) rc = decode_round_enqueue_resolve(R, ctx);
    return rc;
}

int decode_round_wait(Round &R, RoundCtx *ctx_p) {
    std::unique_ptr<RoundCtx> ctx(ctx_p);
    RoundCtx &C = *ctx;
    static const bool trace = getenv("EXG_TRACE") != nullptr;
    EXG_HIP_CHECK(hipStreamSynchronize(C.st));
    const uint32_t nc = C.nc, nf = C.nf, nx = C.nx;
    if (nc) {
        const uint32_t *status = (const uint32_t *)C.h_status;
        if (trace)
            fprintf(stderr, "[exg] zstd: %u blocks, %u chunks (target %llu KiB, %zu round(s)), entropy+scan %.1f ms, execution + resolve + xxh64 done %.1f ms "
                            "after the round began, %.1f MB out (+ %.1f MB of window in front)\n",
                    C.nb - nx, nc, (unsigned long long)(C.target >> 10), C.n_rounds, C.t_entropy - C.t_begin, now_ms() - C.t_begin, C.total / 1e6, C.H / 1e6);
        for (uint32_t c = 0; c < nc; c++)
            if (status[c]) {
                set_error("%s (zstd blocks %llu..%llu)", status_text(status[c]), (unsigned long long)(R.first_block_id + C.chunks[c].first_block - nx),
                          (unsigned long long)(R.first_block_id + C.chunks[c].first_block - nx + C.chunks[c].n_blocks - 1));
                return EXG_E_PARSE;
            }
        for (uint32_t f = 0; f < nf; f++) {
            if (status[nc + f]) {
                set_error("%s (zstd frame %u)", status_text(status[nc + f]), R.frames[f].frame_id);
                return EXG_E_PARSE;
            }
            R.frames[f].verified = C.dframes[f].has_checksum && R.frames[f].out_size <= R.verify_max;
        }
    }
    R.d_buf = C.d_out.p;  // the caller's from here on (exg_rd::dev_pool()->give(dev, p, alloc))
    R.alloc = C.d_out.sz;
    C.d_out.p = nullptr;
    R.produced = C.total;
    return EXG_OK;
}

int decode_round(Round &R, void *stream_v) {
    RoundCtx *ctx = nullptr;
    int rc = decode_round_begin(R, stream_v, &ctx);
    if (!rc) rc = decode_round_enqueue(R, ctx);
    if (!rc) rc = decode_round_wait(R, ctx);
    return rc;
}

uint64_t default_verify_max() {
    static const uint64_t v = getenv("EXG_ZSTD_VERIFY_MAX") ? strtoull(getenv("EXG_ZSTD_VERIFY_MAX"), nullptr, 10) : (64ull << 20);  // ~0.55 GB/s per frame: 64 MiB = 0.12 s
    return v;
}

// the whole stream as one round
int decode(const uint8_t *h_comp, const void *d_comp_v, uint64_t n, void **d_out_p, uint64_t *produced, void *stream_v, std::vector<PendingCheck> *pending,
           Index *prebuilt, uint64_t front_reserve) {
    if (!h_comp || !d_comp_v || !d_out_p || !produced) {
        set_error("exg_zstd_decode: null argument");
        return EXG_E_INVALID_ARG;
    }
    if (front_reserve & 15) {
        set_error("exg_zstd_decode: the room in front of the output must be a multiple of 16");
        return EXG_E_INVALID_ARG;
    }
    *d_out_p = nullptr;
    *produced = 0;
    Index own;
    Index &idx = prebuilt ? *prebuilt : own;  // (a reader walks the headers while the compressed bytes travel)
    if (!prebuilt && !build_index(h_comp, n, idx)) {
        set_error("%s", idx.error.c_str());
        return EXG_E_PARSE;
    }
    Round R;
    R.blocks = idx.blocks;
    R.d_comp = d_comp_v;
    R.front_reserve = front_reserve;
    R.verify_max = default_verify_max();
    for (uint32_t f = 0; f < idx.frames.size(); f++) {
        const Frame &F = idx.frames[f];
        RoundFrame rf;
        rf.first_block = F.first_block;
        rf.n_blocks = F.n_blocks;
        rf.frame_id = f;
        rf.begins = rf.ends = true;
        rf.has_checksum = F.has_checksum;
        rf.checksum = F.checksum;
        R.frames.push_back(rf);
    }
    const int rc = decode_round(R, stream_v);
    if (rc) return rc;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (uint32_t f = 0; f < idx.frames.size(); f++) {
        const Frame &F = idx.frames[f];
        const RoundFrame &rf = R.frames[f];
        if (F.content_size != ~0ull && F.content_size != rf.out_size) {
            (void)hipStreamSynchronize((hipStream_t)stream_v);
            exg_rd::dev_pool()->give(dev, R.d_buf, R.alloc);
            set_error("Data corruption detected (zstd frame %u regenerates %llu bytes, its header says %llu)", f, (unsigned long long)rf.out_size,
                      (unsigned long long)F.content_size);
            return EXG_E_PARSE;
        }
        if (pending && F.has_checksum && !rf.verified) pending->push_back(PendingCheck{rf.out_off, rf.out_size, F.checksum, f});
    }
    *d_out_p = R.d_buf;
    *produced = R.produced;
    return EXG_OK;
}

}  // namespace zst
}  // namespace exg

extern "C" int exg_zstd_decode(const uint8_t *h_comp, const void *d_comp, uint64_t n, void **d_out, uint64_t *produced, void *stream) {
    std::vector<exg::zst::PendingCheck> pending;
    int rc = exg::zst::decode(h_comp, d_comp, n, d_out, produced, stream, &pending, nullptr, 0);
    if (rc || pending.empty()) return rc;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::string err;
    rc = exg::zst::host_verify(*d_out, pending, dev, &err);
    if (rc) {
        exg_rd::dev_pool()->give(dev, *d_out, (size_t)*produced + 64);
        *d_out = nullptr;
        *produced = 0;
        exg::set_error("%s", err.c_str());
    }
    return rc;
}
