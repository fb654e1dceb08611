// exg_inflate_stream.hip — ONE big DEFLATE stream (what gzip / pigz write: a single gzip member) inflated by many
// wavefronts at once.  exg_inflate.hip parallelises over members, so such a file would run on one wavefront
// (10 MB/s, measured); here the stream is cut into chunks of compressed bytes that are decoded concurrently.
//
// Replaces the same reference code as exg_inflate.hip (DataFusion 28 `FileCompressionType::convert_stream` ->
// async-compression -> flate2, rust/src/arrow_reader.rs:60-91) for this input shape.  The method is the one of
// pugz / rapidgzip, restated for wavefronts:
//   1. k_find_blocks   a wave per chunk boundary looks for the first deflate block that starts at or after it:
//                      every lane tests one bit offset with a cheap filter (BTYPE = dynamic, HLIT / HDIST in range,
//                      the code-length code complete: Kraft sum exactly 1), survivors are validated by really
//                      decoding from there (header, tables, the first few hundred symbols);
//   2. k_inflate_chunks the ordinary decoder (exg_inflate_core.hpp) from each found block start up to the next one,
//                      with a window of 16-bit symbols: a byte, or "byte i of the 32 KiB in front of this chunk"
//                      (not known yet);
//   3. the host checks that every chunk ended exactly where the next one started (a false block start — rare —
//      makes the previous chunk run on through it) ;
//   4. k_windows       chunk by chunk (sequential, 32 Ki symbols each): the last 32 KiB of output before every chunk;
//   5. k_resolve       all symbols -> bytes in parallel (markers looked up in their chunk's window).
// Everything stays in HBM; the host only sees bit positions and counts.
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <iterator>
#include <map>
#include <memory>
#include <vector>

#include "exg_inflate_core.hpp"
#include "exg_reader.hpp"

namespace exg {

namespace {

static constexpr uint32_t kProbeSymbols = 320;  // a candidate block start must decode this far without an error
// Text streams (FASTA / FASTQ / VCF are): a false block start decodes garbage — literals spread over all 256 byte
// values — so a control character among its literals gives it away within a few dozen symbols, while nothing else
// does before its end-of-block code (a valid header makes random bits decode "validly", with the very symbol
// statistics the code was built for: ~one real block's worth of work per false candidate, 14-68 ms of finder time
// depending on which boundaries met one).  A candidate that decodes this many symbols of clean text is accepted.
static constexpr uint32_t kTextProbeSymbols = 4096;

// ---- 1. block finder ---------------------------------------------------------------------------------------------------
struct FindJob {
    unsigned long long comp_off, comp_size;  // the stream
    unsigned long long from_bit, to_bit;     // search [from_bit, to_bit) relative to comp_off
    unsigned int text, pad;                  // text != 0: the stream inflates to text (see kTextProbeSymbols)
};

// cheap per-lane filter on the 64 + 64 bits that start at a candidate offset: could a dynamic block begin here?
__device__ __forceinline__ bool header_filter(unsigned long long v, unsigned long long v17) {
    if ((v & 7ull) != 4ull) return false;  // BFINAL = 0 (a final block mid-stream is never needed as a chunk start:
                                           // the chunk in front simply runs to the end), BTYPE = 10 (dynamic Huffman)
    const uint32_t hlit = (uint32_t)(v >> 3) & 31u, hdist = (uint32_t)(v >> 8) & 31u, hclen = (uint32_t)(v >> 13) & 15u;
    if (hlit > 29 || hdist > 29) return false;
    // code-length code: hclen + 4 lengths of 3 bits from bit 17; it must be complete (zlib writes complete codes)
    const uint32_t n = hclen + 4;
    uint32_t kraft = 0, used = 0;
#pragma unroll
    for (uint32_t i = 0; i < 19; i++) {
        const uint32_t l = i < n ? (uint32_t)(v17 >> (3 * i)) & 7u : 0u;
        if (l) kraft += 128u >> l, used++;
    }
    return kraft == 128u && used >= 2;
}

__global__ __launch_bounds__(64) void k_find_blocks(const uint8_t *__restrict__ d_comp, const FindJob *__restrict__ jobs,
                                                    unsigned long long *found, uint16_t *d_scratch, uint32_t n_jobs) {
    // probing decodes keep nothing: the smallest ring (11 KiB of LDS per wave instead of 72 KiB)
    __shared__ __attribute__((aligned(16))) InflateLdsT<true, 2048> s;
    __shared__ InflateJobStatus s_st;
    const uint32_t lane = threadIdx.x;
    for (uint32_t j = blockIdx.x; j < n_jobs; j += gridDim.x) {
        const FindJob fj = jobs[j];
        unsigned long long result = ~0ull;
        // The input is staged as the search moves on — a chunk of the ring serves sixteen steps of 64 offsets, the chunk after
        // it is on its way — and again from scratch only behind a probing decode, which uses the ring for itself.  (Staging
        // anew for every 64 offsets put an HBM round trip into every step: 10 ms of a round's 35.)
        BitIn br;
        const BitBase bb = bit_base(d_comp, fj.comp_off, fj.comp_size, fj.from_bit);
        br.g0 = bb.g0;
        br.limit = bb.limit;
        bool staged = false;
        for (unsigned long long base = fj.from_bit; base < fj.to_bit && result == ~0ull; base += 64) {
            // every lane looks at its own offset
            br.bitpos = (unsigned long long)((long long)base - bb.rel_bits);
            if (!staged) {
                start_input(br, lane);
                __syncthreads();
                staged = true;
            }
            // (the chunk that holds `base` and the one behind it: the filter reads 64 + 17 + 64 + 64 bits from there, 27 bytes)
            ensure(s, br, lane);
            const unsigned long long o = (unsigned long long)((long long)base - bb.rel_bits) + lane;
            const bool in_range = base + lane < fj.to_bit && ((o + 17 + 64) >> 3) < br.limit;
            const bool pass = in_range && header_filter(peek_at(s, o), peek_at(s, o + 17));
            unsigned long long cand = __ballot(pass);
            if (cand) staged = false;  // (the decodes below stage their own input)
            while (cand && result == ~0ull) {
                const uint32_t l = (uint32_t)__ffsll((long long)cand) - 1;
                cand &= cand - 1;
                // Real decode from there.  A valid header alone is not enough (about one boundary in ten of a FASTQ
                // stream had a false one in front of the real block start): the WHOLE block must decode, and what
                // follows it must again be a block that decodes (its header + the first kProbeSymbols symbols).
                InflateJob jb;
                jb.comp_off = fj.comp_off;
                jb.text_probe = fj.text;
                jb.pad = 0;
                if (fj.text) {
                    jb.comp_size = fj.comp_size;
                    jb.out_off = 0;
                    jb.out_cap = kTextProbeSymbols;
                    jb.start_bit = base + l;
                    jb.stop_bit = 0;
                    inflate_job<true, 2048>(s, d_comp, (uint16_t *)nullptr, jb, &s_st);
                    const uint32_t code = s_st.code;
                    const bool fin = s_st.final_block != 0;
                    __syncthreads();
                    if (code == 4 || (code == 0 && fin)) result = base + l;  // clean text this far (or to the end of the stream)
                    continue;
                }
                // a block longer than 128 KiB of input is not accepted as a chunk start (zlib closes a block after 16 Ki
                // symbols: 20-60 KB of FASTQ): garbage behind a false header can run for millions of symbols before it
                // meets an end-of-block code, and one such candidate holds the whole finder up (seen: 4.5 s without a
                // bound, 230 ms with 512 KiB).  A real block beyond the bound only costs a chunk start.
                jb.comp_size = std::min<unsigned long long>(fj.comp_size, ((base + l) >> 3) + (128u << 10));
                jb.out_off = 0;
                jb.out_cap = 1ull << 22;  // nothing is stored: the bound only stops a runaway decode
                jb.start_bit = base + l;
                jb.stop_bit = base + l + 1;
                inflate_job<true, 2048>(s, d_comp, (uint16_t *)nullptr, jb, &s_st);
                uint32_t code = s_st.code;
                const bool final1 = s_st.final_block != 0;
                const unsigned long long end1 = s_st.end_bit;
                __syncthreads();
                if (final1) code = 1;
                if (code == 0) {
                    jb.comp_size = fj.comp_size;
                    jb.out_cap = kProbeSymbols;
                    jb.start_bit = end1;
                    jb.stop_bit = end1 + 1;
                    inflate_job<true, 2048>(s, d_comp, (uint16_t *)nullptr, jb, &s_st);
                    code = s_st.code == 4 ? 0u : s_st.code;
                    __syncthreads();
                }
                if (code == 0) result = base + l;
            }
        }
        if (lane == 0) found[j] = result;
        __syncthreads();
    }
}

// ---- 2. chunk decode -----------------------------------------------------------------------------------------------------
template <uint32_t RING>
__global__ __launch_bounds__(64) void k_inflate_chunks(const uint8_t *__restrict__ d_comp, uint16_t *d_sym,
                                                       const InflateJob *__restrict__ jobs, InflateJobStatus *status,
                                                       uint32_t n_jobs) {
    __shared__ __attribute__((aligned(16))) InflateLdsT<true, RING> s;
    for (uint32_t j = blockIdx.x; j < n_jobs; j += gridDim.x) inflate_job<true, RING>(s, d_comp, d_sym, jobs[j], &status[j]);
}

// ---- 4. windows ----------------------------------------------------------------------------------------------------------
struct ChunkOut {
    const uint16_t *sym;         // the chunk's symbols
    unsigned long long n;        // symbols (= bytes) it produced
    unsigned long long out_off;  // where its bytes go in the output
};
// windows[c] = the 32 KiB of output in front of chunk c (windows[0] is empty: zeros).  The window behind a chunk is
// a gather from the window in front of it (tail symbol = a byte, or "byte j of the window in front"; a chunk shorter
// than 32 KiB also passes the end of the old window on), and gathers compose.  Three steps instead of one workgroup
// walking all chunks (which cost ~15 us per chunk: 25 ms of a 1 GB file):
//   k_win_group  one workgroup per group of G chunks composes, chunk by chunk, maps[c] = window c as a gather from the
//                window in front of the group's first chunk (16-bit symbols: byte, or 0x8000 | index);
//   k_win_chain  one workgroup resolves the windows of the group starts, group by group (n / G steps);
//   k_win_apply  every other window = its map applied to its group's start window, all in parallel.
// tail symbol i of chunk `ch` relative to the window in front of it
__device__ __forceinline__ uint16_t tail_symbol(const ChunkOut &ch, uint32_t i) {
    const long long k = (long long)ch.n + (long long)i - 32768;
    return k >= 0 ? ch.sym[k] : (uint16_t)(0x8000u | (uint32_t)(32768 + k));
}
__global__ __launch_bounds__(1024) void k_win_group(const ChunkOut *__restrict__ chunks, uint16_t *maps, uint32_t n_chunks,
                                                    uint32_t group) {
    const uint32_t c0 = blockIdx.x * group;
    for (uint32_t c = c0; c < c0 + group && c + 1 < n_chunks; c++) {
        const ChunkOut ch = chunks[c];
        const uint16_t *mprev = maps + (size_t)c * 32768;  // window c relative to window c0 (identity for c == c0)
        uint16_t *mnext = maps + (size_t)(c + 1) * 32768;
        for (uint32_t i = threadIdx.x; i < 32768; i += 1024) {
            uint16_t sy = tail_symbol(ch, i);
            if ((sy & 0x8000u) && c != c0) sy = mprev[sy & 0x7FFFu];
            mnext[i] = sy;
        }
        __threadfence_block();
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_win_chain(const uint16_t *__restrict__ maps, uint8_t *windows, uint32_t n_chunks,
                                                    uint32_t group) {
    for (uint32_t c0 = 0; c0 + group < n_chunks; c0 += group) {
        const uint8_t *wprev = windows + (size_t)c0 * 32768;
        uint8_t *wnext = windows + (size_t)(c0 + group) * 32768;
        const uint16_t *m = maps + (size_t)(c0 + group) * 32768;
        for (uint32_t i = threadIdx.x; i < 32768; i += 1024) {
            const uint16_t sy = m[i];
            wnext[i] = sy & 0x8000u ? wprev[sy & 0x7FFFu] : (uint8_t)sy;
        }
        __threadfence_block();
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_win_apply(const uint16_t *__restrict__ maps, uint8_t *windows, uint32_t n_chunks,
                                                   uint32_t group) {
    const uint32_t c = blockIdx.y;
    if (c % group == 0) return;  // group starts are already bytes
    const uint8_t *w0 = windows + (size_t)(c - c % group) * 32768;
    const uint16_t *m = maps + (size_t)c * 32768;
    uint8_t *w = windows + (size_t)c * 32768;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < 32768; i += gridDim.x * 256) {
        const uint16_t sy = m[i];
        w[i] = sy & 0x8000u ? w0[sy & 0x7FFFu] : (uint8_t)sy;
    }
}

// ---- 5. resolve ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve(const ChunkOut *__restrict__ chunks, const uint8_t *__restrict__ windows,
                                                 uint8_t *__restrict__ d_out, uint32_t n_chunks, uint32_t *bad, uint32_t have_window0) {
    // blockIdx.y = chunk; 16 symbols per thread -> one 16-byte store when the destination is aligned
    const uint32_t c = blockIdx.y;
    const ChunkOut ch = chunks[c];
    const uint8_t *w = windows + (size_t)c * 32768;
    for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; g * 16 < ch.n; g += (unsigned long long)gridDim.x * 256) {
        const unsigned long long i0 = g * 16;
        uint8_t b[16];
        const uint32_t cnt = ch.n - i0 < 16 ? (uint32_t)(ch.n - i0) : 16u;
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            uint16_t sy = k < cnt ? ch.sym[i0 + k] : (uint16_t)0;
            b[k] = sy & 0x8000u ? w[sy & 0x7FFFu] : (uint8_t)sy;
            if ((sy & 0x8000u) && c == 0 && !have_window0) atomicOr(bad, 1u);  // nothing precedes a member's first chunk
        }
        uint8_t *dst = d_out + ch.out_off + i0;
        if (cnt == 16 && (((uintptr_t)dst) & 15) == 0) {
            uint4 v;
            v.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
            v.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
            v.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((uint32_t)b[11] << 24);
            v.w = b[12] | (b[13] << 8) | (b[14] << 16) | ((uint32_t)b[15] << 24);
            *reinterpret_cast<uint4 *>(dst) = v;
        } else {
            for (uint32_t k = 0; k < cnt; k++) dst[k] = b[k];
        }
    }
}

// The 32 KiB of output a later round of the same member will have in front of it: the tail of this round's bytes, topped up
// from the window this round had in front when it produced less than that (old_win NULL: zeros).
__global__ __launch_bounds__(256) void k_win_tail(const uint8_t *__restrict__ out, unsigned long long produced, const uint8_t *old_win,
                                                  uint8_t *new_win) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < 32768; i += gridDim.x * 256) {
        const long long k = (long long)produced + (long long)i - 32768;
        new_win[i] = k >= 0 ? out[k] : (old_win ? old_win[32768 + k] : (uint8_t)0);
    }
}

// Scratch of one call, from the library's device pool: a 5 GB stream needs a 29 GB symbol buffer, and hipMalloc of that
// was measured at 0.24 - 1.3 s (per call; the pool pays it once).  The call's stream is waited for before a block goes back
// (idle already on the normal path: the results of every stage are read on the host)
thread_local hipStream_t t_call_stream = nullptr;  // the stream of the exg_inflate_stream call this thread is in
struct DevBuf {
    void *p = nullptr;
    size_t sz = 0;
    int dev = 0;
    hipStream_t stream = t_call_stream;
    DevBuf() { (void)hipGetDevice(&dev); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void release() {
        if (!p) return;
        (void)hipStreamSynchronize(stream);
        exg_rd::dev_pool()->give(dev, p, sz);
        p = nullptr;
    }
    ~DevBuf() { release(); }
    hipError_t alloc(size_t n) {
        release();
        sz = n ? n : 16;
        p = exg_rd::dev_pool()->take(dev, sz);
        return p ? hipSuccess : hipErrorOutOfMemory;
    }
};

}  // namespace
}  // namespace exg

using namespace exg;

static double st_now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
#define ST_TRACE(...)                                   \
    do {                                                \
        if (getenv("EXG_TRACE")) fprintf(stderr, __VA_ARGS__); \
    } while (0)

#define ST_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::exg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return EXG_E_HIP;                                                                     \
        }                                                                                         \
    } while (0)

// d_comp: the file's compressed bytes on the device (16-byte aligned); the DEFLATE stream of the member starts at
// comp_off and at most comp_size bytes may be read.  On success *d_out_p is a hipMalloc'd buffer (caller frees) holding
// *produced inflated bytes (+ 64 bytes of zeroed slack) and *consumed the compressed bytes used (byte aligned after the
// final block).  Synchronises `stream` several times (it returns sizes to the host).
//
// One ROUND of a member (exg_inflate_round, include/exon_gpu.h) is the same with three differences: the decode starts at
// `start_bit` (where the round before ended) with the 32 KiB that round left as the window in front of the first chunk
// (`d_window`, updated in place for the next round); when `partial` is set more compressed bytes follow behind comp_size,
// so the piece behind the last block start that was found is NOT decoded — the round ends at a block boundary the chain
// reached at or behind that start, and nothing of the stream's end (trailer, ISIZE) is looked at; and `front_reserve` bytes
// stay free in front of the output (the reader prepends the previous segment's unconsumed tail there).
struct RoundArgs {
    uint64_t start_bit = 0;
    bool partial = false;
    void *d_window = nullptr;   // 32768 bytes: in = the output in front of start_bit (have_window), out = in front of end_bit
    bool have_window = false;
    uint64_t front_reserve = 0;
    double ratio_hint = 0;      // inflated / compressed seen so far (0: unknown)
    // results
    uint64_t out_alloc = 0;     // what *d_out was taken from the pool with (give it back with this size)
    uint64_t end_bit = 0;
    bool final_block = false;
    bool need_more = false;     // partial: no block start in the window (or the chain ran out of input): retry with more bytes
};
static int inflate_stream_impl(const void *d_comp_v, uint64_t comp_off, uint64_t comp_size, uint64_t chunk_bytes, void **d_out_p,
                               uint64_t *produced, RoundArgs &ra, void *stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    t_call_stream = stream;
    const uint8_t *d_comp = (const uint8_t *)d_comp_v;
    if (!d_comp || !d_out_p || !produced || ((uintptr_t)d_comp & 15)) {
        set_error("exg_inflate_stream: bad arguments");
        return EXG_E_INVALID_ARG;
    }
    *d_out_p = nullptr;
    *produced = 0;
    ra.need_more = ra.final_block = false;
    ra.end_bit = ra.start_bit;
    if (chunk_bytes < 32768) chunk_bytes = 32768;
    const double t_begin = st_now();
    const uint64_t total_bits = comp_size * 8;
    const uint64_t bit0 = ra.start_bit;
    // ---- 1. block starts near the chunk boundaries (counted from the byte the round starts in)
    const uint64_t byte0 = bit0 / 8;
    const uint32_t n_bound = (uint32_t)std::min<uint64_t>((comp_size - std::min(comp_size, byte0) + chunk_bytes - 1) / chunk_bytes, 1u << 20);
    std::vector<uint64_t> starts;  // bit offsets of the chunk starts (sorted, unique); starts[0] = where the round starts
    starts.push_back(bit0);
    // first block start in each [from, to) (bit ranges), ~0 where none was found
    bool text_mode = !getenv("EXG_STREAM_NO_TEXT_PROBE");
    auto find_starts = [&](const std::vector<std::pair<uint64_t, uint64_t>> &ranges, std::vector<uint64_t> *found) -> int {
        std::vector<FindJob> fj(ranges.size());
        for (size_t k = 0; k < ranges.size(); k++) {
            fj[k].comp_off = comp_off;
            fj[k].comp_size = comp_size;
            fj[k].from_bit = ranges[k].first;
            fj[k].to_bit = ranges[k].second;
            fj[k].text = text_mode ? 1u : 0u;
            fj[k].pad = 0;
        }
        DevBuf d_fj, d_found, d_scratch;
        ST_HIP(d_fj.alloc(fj.size() * sizeof(FindJob)));
        ST_HIP(d_found.alloc(fj.size() * 8));
        ST_HIP(d_scratch.alloc(16));
        ST_HIP(hipMemcpyAsync(d_fj.p, fj.data(), fj.size() * sizeof(FindJob), hipMemcpyHostToDevice, stream));
        const uint32_t grid = (uint32_t)std::min<size_t>(fj.size(), 8192);
        hipLaunchKernelGGL(k_find_blocks, dim3(grid), dim3(64), 0, stream, d_comp, (const FindJob *)d_fj.p,
                           (unsigned long long *)d_found.p, (uint16_t *)d_scratch.p, (uint32_t)fj.size());
        ST_HIP(hipGetLastError());
        found->resize(fj.size());
        ST_HIP(hipMemcpyAsync(found->data(), d_found.p, found->size() * 8, hipMemcpyDeviceToHost, stream));
        ST_HIP(hipStreamSynchronize(stream));
        return EXG_OK;
    };
    if (n_bound > 1) {
        std::vector<std::pair<uint64_t, uint64_t>> ranges;
        for (uint32_t k = 1; k < n_bound; k++)
            ranges.emplace_back((byte0 + (uint64_t)k * chunk_bytes) * 8, std::min<uint64_t>((byte0 + (uint64_t)(k + 1) * chunk_bytes) * 8, total_bits));
        std::vector<uint64_t> found;
        int rc = find_starts(ranges, &found);
        if (rc) return rc;
        if (text_mode) {
            // not text after all (fewer than half of the boundaries found a start): validate by whole blocks instead
            size_t hits = 0;
            for (uint64_t f : found) hits += f != ~0ull;
            if (hits * 2 < found.size()) {
                ST_TRACE("[exg] inflate stream: only %zu of %zu boundaries have a text block start: binary search mode\n", hits, found.size());
                text_mode = false;
                rc = find_starts(ranges, &found);
                if (rc) return rc;
            }
        }
        for (uint64_t f : found)
            if (f != ~0ull && f > starts.back()) starts.push_back(f);
        ST_TRACE("[exg] inflate stream: %u boundaries searched, %zu block starts found, %.1f ms\n", n_bound - 1, starts.size() - 1,
                 (st_now() - t_begin) * 1e3);
    }
    // a partial round ends at (or, behind a false start, just behind) the last block start found: what follows it is the
    // next round's, decoded when its bytes are there
    const uint64_t round_end = ra.partial ? starts.back() : 0;
    if (ra.partial && starts.size() < 2) {
        ra.need_more = true;
        return EXG_OK;
    }
    // ---- 2 + 3. decode every candidate chunk, then walk the chain of chunks from bit 0: a chunk counts only if it
    // starts exactly where the previous one ended.  A false block start is simply never reached (the chunk in front
    // of it runs on to the next real boundary); if that boundary is not a candidate either, the piece from there to
    // the next candidate is decoded in another round.
    // Symbol buffer: cap_factor0 symbols per compressed byte and piece.  8 covers most text, but 16 B per compressed
    // byte is 9 GB for a 0.5 GB file (and hipMalloc of that was seen to take 0.5 s): when the stream runs to the end
    // of the buffer, the gzip trailer's ISIZE (mod 2^32) gives the real ratio; pieces that still overflow are refitted.
    uint64_t cap_factor0 = 8;
    if (ra.ratio_hint > 0 && ra.ratio_hint < 64.0) cap_factor0 = std::min<uint64_t>(8, std::max<uint64_t>(3, (uint64_t)(ra.ratio_hint * 1.5 + 1.0)));
    if (!ra.partial && bit0 == 0 && comp_size >= 8) {
        uint32_t isize = 0;
        ST_HIP(hipMemcpyAsync(&isize, d_comp + comp_off + comp_size - 4, 4, hipMemcpyDeviceToHost, stream));
        ST_HIP(hipStreamSynchronize(stream));
        uint64_t est = isize;
        while (est < comp_size) est += 1ull << 32;  // deflate does not expand text
        const double ratio = (double)est / (double)comp_size;
        if (ratio < 64.0) cap_factor0 = std::min<uint64_t>(8, std::max<uint64_t>(3, (uint64_t)(ratio * 1.5 + 1.0)));
        ST_TRACE("[exg] inflate stream: ISIZE hints at ratio %.2f: %llu symbols per compressed byte\n", ratio, (unsigned long long)cap_factor0);
    }
    struct Piece {
        InflateJobStatus st;
        const uint16_t *sym;
        uint64_t stop_bit;
    };
    std::map<uint64_t, Piece> decoded;
    std::vector<std::unique_ptr<DevBuf>> sym_bufs;
    // exact_cap != 0: one job whose output size is known; measure_only: decode without storing (its size is the answer)
    auto run_jobs = [&](const std::vector<std::pair<uint64_t, uint64_t>> &spans, uint64_t cap_factor,
                        const std::vector<uint64_t> *exact_caps, bool measure_only,
                        std::vector<InflateJobStatus> *measured) -> int {
        const uint32_t n = (uint32_t)spans.size();
        std::vector<InflateJob> jobs(n);
        std::vector<uint64_t> off(n + 1, 0);
        for (uint32_t k = 0; k < n; k++) {
            const uint64_t end_bit = spans[k].second ? spans[k].second : total_bits;
            jobs[k].comp_off = comp_off;
            jobs[k].comp_size = comp_size;
            jobs[k].out_off = off[k];
            jobs[k].out_cap = measure_only ? (comp_size - spans[k].first / 8) * 1032 + 65536 : exact_caps ? (*exact_caps)[k] : (end_bit - spans[k].first + 7) / 8 * cap_factor + 65536;
            jobs[k].start_bit = spans[k].first;
            jobs[k].stop_bit = spans[k].second;
            if (measure_only) continue;
            off[k + 1] = off[k] + ((jobs[k].out_cap + 15) & ~15ull);
        }
        sym_bufs.emplace_back(new DevBuf());
        DevBuf &d_sym = *sym_bufs.back();
        DevBuf d_jobs, d_st;
        const double t_alloc = st_now();
        if (!measure_only) ST_HIP(d_sym.alloc(off[n] * 2 + 64));
        ST_HIP(d_jobs.alloc(n * sizeof(InflateJob)));
        ST_HIP(d_st.alloc(n * sizeof(InflateJobStatus)));
        ST_TRACE("[exg] inflate stream: symbol buffer %.2f GB allocated in %.1f ms\n", off[n] * 2 / 1e9, (st_now() - t_alloc) * 1e3);
        ST_HIP(hipMemcpyAsync(d_jobs.p, jobs.data(), n * sizeof(InflateJob), hipMemcpyHostToDevice, stream));
        static const int ring = [] {
            const char *e = getenv("EXG_STREAM_RING");  // A/B switch: 32768 = the whole symbol window in LDS (72 KiB)
            return e ? atoi(e) : 2048;
        }();
        if (ring == 32768)
            hipLaunchKernelGGL(k_inflate_chunks<32768>, dim3(std::min<uint32_t>(n, 8192)), dim3(64), 0, stream, d_comp,
                               (uint16_t *)d_sym.p, (const InflateJob *)d_jobs.p, (InflateJobStatus *)d_st.p, n);
        else if (ring == 4096)
            hipLaunchKernelGGL(k_inflate_chunks<4096>, dim3(std::min<uint32_t>(n, 8192)), dim3(64), 0, stream, d_comp,
                               (uint16_t *)d_sym.p, (const InflateJob *)d_jobs.p, (InflateJobStatus *)d_st.p, n);
        else if (ring == 8192)
            hipLaunchKernelGGL(k_inflate_chunks<8192>, dim3(std::min<uint32_t>(n, 8192)), dim3(64), 0, stream, d_comp,
                               (uint16_t *)d_sym.p, (const InflateJob *)d_jobs.p, (InflateJobStatus *)d_st.p, n);
        else
            hipLaunchKernelGGL(k_inflate_chunks<2048>, dim3(std::min<uint32_t>(n, 8192)), dim3(64), 0, stream, d_comp,
                               (uint16_t *)d_sym.p, (const InflateJob *)d_jobs.p, (InflateJobStatus *)d_st.p, n);
        ST_HIP(hipGetLastError());
        std::vector<InflateJobStatus> st(n);
        ST_HIP(hipMemcpyAsync(st.data(), d_st.p, n * sizeof(InflateJobStatus), hipMemcpyDeviceToHost, stream));
        ST_HIP(hipStreamSynchronize(stream));
        if (measure_only) {
            *measured = st;
            return EXG_OK;
        }
        for (uint32_t k = 0; k < n; k++) decoded[spans[k].first] = Piece{st[k], (const uint16_t *)d_sym.p + off[k], spans[k].second};
        ST_TRACE("[exg] inflate stream: decoded %u piece(s), at %.1f ms\n", n, (st_now() - t_begin) * 1e3);
        return EXG_OK;
    };
    {
        std::vector<std::pair<uint64_t, uint64_t>> spans;
        for (size_t k = 0; k + (ra.partial ? 1 : 0) < starts.size(); k++) spans.emplace_back(starts[k], k + 1 < starts.size() ? starts[k + 1] : 0);
        int rc = run_jobs(spans, cap_factor0, nullptr, false, nullptr);
        if (rc) return rc;
    }
    std::vector<ChunkOut> co;
    uint64_t total = 0, end_bit = 0;
    {
        // Every round walks the chain from bit 0.  Where it stands at a bit no piece starts from (the candidate behind
        // it was a false block start, so the piece in front ran on to the next REAL boundary e), the walk jumps to the
        // next candidate to collect the other breaks of this round too; then every gap [e, next candidate) is cut at
        // block starts found inside it and all the pieces are decoded at once — a gap costs a round of short decodes,
        // not a chunk-long decode by one wavefront (measured: 72 ms per gap on a 1 GB file before).
        const uint64_t sub_bits = 8ull * std::max<uint64_t>(32768, std::min<uint64_t>(chunk_bytes / 8, 65536));
        const int max_rounds = (int)starts.size() + 8;  // every false block start can break the chain once
        uint64_t last_first_break = 0;
        for (int rounds = 0;; rounds++) {
            co.clear();
            total = 0;
            std::vector<std::pair<uint64_t, uint64_t>> gaps;  // [e, next candidate or 0 = end of stream)
            uint64_t refit = ~0ull;                             // a piece whose output did not fit
            bool done = false;
            for (uint64_t e = bit0;;) {
                if (ra.partial && e >= round_end) {  // a block boundary the chain reached: the round ends here
                    done = gaps.empty();
                    if (done) end_bit = e;
                    break;
                }
                auto it = decoded.find(e);
                if (it == decoded.end()) {
                    auto nxt = std::upper_bound(starts.begin(), starts.end(), e);
                    gaps.emplace_back(e, nxt == starts.end() ? 0 : *nxt);
                    if (nxt == starts.end()) break;
                    e = *nxt;
                    continue;
                }
                const Piece &pc = it->second;
                if (pc.st.code == 4) {
                    if (gaps.empty()) refit = e;
                    break;
                }
                if (pc.st.code) {
                    if (!gaps.empty()) break;  // behind a break the walk is only a guess: settle the gaps first
                    if (ra.partial && (pc.st.end_bit + 7) / 8 + 64 >= comp_size) {  // ran into the end of the bytes that are here
                        ra.need_more = true;
                        return EXG_OK;
                    }
                    set_error("corrupt deflate stream (code %u at bit %llu)", pc.st.code, (unsigned long long)e);
                    return EXG_E_PARSE;
                }
                if (gaps.empty()) {
                    co.push_back(ChunkOut{pc.sym, pc.st.produced, total});
                    total += pc.st.produced;
                    end_bit = pc.st.end_bit;
                }
                if (pc.st.final_block) {
                    done = gaps.empty();
                    if (done) ra.final_block = true;
                    break;
                }
                if (pc.st.end_bit <= e) {
                    if (!gaps.empty()) break;
                    set_error("exg_inflate_stream: no progress at bit %llu", (unsigned long long)e);
                    return EXG_E_PARSE;
                }
                e = pc.st.end_bit;
            }
            if (done) break;
            const uint64_t first_break = gaps.empty() ? refit : gaps[0].first;
            if (rounds >= max_rounds || (rounds && first_break <= last_first_break && refit == ~0ull)) {
                set_error("corrupt deflate stream (the chunk chain does not settle after bit %llu)", (unsigned long long)first_break);
                return EXG_E_PARSE;
            }
            last_first_break = first_break;
            if (refit != ~0ull) {
                // Outputs that did not fit (every such piece at once) are first MEASURED (decoded without storing), then
                // decoded into buffers of those sizes: growing a buffer blindly would re-decode again and again — on a
                // corrupt stream for minutes.
                std::vector<std::pair<uint64_t, uint64_t>> spans;
                for (auto &kv : decoded)
                    if (kv.second.st.code == 4) spans.emplace_back(kv.first, kv.second.stop_bit);
                ST_TRACE("[exg] inflate stream: chain stands at bit %llu (output did not fit; %zu such piece(s))\n",
                         (unsigned long long)refit, spans.size());
                std::vector<InflateJobStatus> m;
                int rc = run_jobs(spans, 0, nullptr, true, &m);
                if (rc) return rc;
                std::vector<uint64_t> caps(spans.size());
                for (size_t k = 0; k < spans.size(); k++) {
                    if (m[k].code && spans[k].first == refit) {
                        set_error("corrupt deflate stream (code %u after bit %llu)", m[k].code, (unsigned long long)refit);
                        return EXG_E_PARSE;
                    }
                    caps[k] = m[k].produced + 64;
                }
                rc = run_jobs(spans, 0, &caps, false, nullptr);
                if (rc) return rc;
                if (decoded[refit].st.code == 4) {
                    set_error("exg_inflate_stream: a chunk produced more than its measured size");
                    return EXG_E_PARSE;
                }
                continue;
            }
            // cut the gaps
            std::vector<std::pair<uint64_t, uint64_t>> ranges;
            for (auto &g : gaps) {
                const uint64_t g_end = g.second ? g.second : total_bits;
                for (uint64_t b = g.first + sub_bits; b + sub_bits / 2 < g_end; b += sub_bits)
                    ranges.emplace_back(b, std::min<uint64_t>(b + sub_bits, g_end));
            }
            std::vector<uint64_t> fresh;
            for (auto &g : gaps) fresh.push_back(g.first);
            if (!ranges.empty()) {
                std::vector<uint64_t> found;
                int rc = find_starts(ranges, &found);
                if (rc) return rc;
                for (uint64_t f : found)
                    if (f != ~0ull) fresh.push_back(f);
            }
            std::sort(fresh.begin(), fresh.end());
            fresh.erase(std::unique(fresh.begin(), fresh.end()), fresh.end());
            {
                std::vector<uint64_t> merged;
                std::set_union(starts.begin(), starts.end(), fresh.begin(), fresh.end(), std::back_inserter(merged));
                starts.swap(merged);
            }
            std::vector<std::pair<uint64_t, uint64_t>> spans;
            for (uint64_t f : fresh) {
                if (decoded.count(f)) continue;
                auto nxt = std::upper_bound(starts.begin(), starts.end(), f);
                spans.emplace_back(f, nxt == starts.end() ? 0 : *nxt);
            }
            ST_TRACE("[exg] inflate stream: %zu break(s) in the chain, first at bit %llu: %zu piece(s) to decode\n", gaps.size(),
                     (unsigned long long)gaps[0].first, spans.size());
            if (spans.empty()) {
                set_error("exg_inflate_stream: the chain breaks at bit %llu although a piece starts there", (unsigned long long)gaps[0].first);
                return EXG_E_PARSE;
            }
            int rc = run_jobs(spans, cap_factor0, nullptr, false, nullptr);
            if (rc) return rc;
        }
    }
    // ---- 4 + 5. windows, then bytes
    const uint32_t n = (uint32_t)co.size();
    DevBuf d_co, d_win, d_bad;
    void *d_out = nullptr;
    ST_HIP(d_co.alloc(n * sizeof(ChunkOut)));
    ST_HIP(d_win.alloc((size_t)n * 32768));
    ST_HIP(d_bad.alloc(4));
    // (allocated at the device pool's size class: the reader hands the buffer to that pool when the file is done)
    int out_dev = 0;
    ST_HIP(hipGetDevice(&out_dev));
    // whatever way the rest is left, the output goes back to the pool unless it is handed to the caller
    struct OutGuard {
        int dev;
        hipStream_t st;
        void *p = nullptr;
        size_t sz = 0;
        ~OutGuard() {
            if (!p) return;
            (void)hipStreamSynchronize(st);
            exg_rd::dev_pool()->give(dev, p, sz);
        }
    } out_guard{out_dev, stream};
    const uint64_t reserve = ra.front_reserve;
    out_guard.sz = (size_t)(reserve + total + 64);
    out_guard.p = exg_rd::dev_pool()->take(out_dev, out_guard.sz);
    if (!out_guard.p) ST_HIP(hipErrorOutOfMemory);
    for (ChunkOut &c : co) c.out_off += reserve;
    d_out = out_guard.p;
    ST_HIP(hipMemcpyAsync(d_co.p, co.data(), n * sizeof(ChunkOut), hipMemcpyHostToDevice, stream));
    if (ra.have_window && ra.d_window)
        ST_HIP(hipMemcpyAsync(d_win.p, ra.d_window, 32768, hipMemcpyDeviceToDevice, stream));
    else
        ST_HIP(hipMemsetAsync(d_win.p, 0, 32768, stream));
    ST_HIP(hipMemsetAsync(d_bad.p, 0, 4, stream));
    ST_HIP(hipMemsetAsync((char *)d_out + reserve + total, 0, 64, stream));
    {
        uint32_t group = 1;
        while (group * group < n) group++;
        DevBuf d_maps;
        ST_HIP(d_maps.alloc((size_t)n * 32768 * 2));
        hipLaunchKernelGGL(k_win_group, dim3((n + group - 1) / group), dim3(1024), 0, stream, (const ChunkOut *)d_co.p,
                           (uint16_t *)d_maps.p, n, group);
        hipLaunchKernelGGL(k_win_chain, dim3(1), dim3(1024), 0, stream, (const uint16_t *)d_maps.p, (uint8_t *)d_win.p, n, group);
        hipLaunchKernelGGL(k_win_apply, dim3(8, n), dim3(256), 0, stream, (const uint16_t *)d_maps.p, (uint8_t *)d_win.p, n, group);
        ST_HIP(hipGetLastError());
        ST_HIP(hipStreamSynchronize(stream));  // d_maps is freed when this scope ends
    }
    if (n)
        hipLaunchKernelGGL(k_resolve, dim3(256, n), dim3(256), 0, stream, (const ChunkOut *)d_co.p, (const uint8_t *)d_win.p,
                           (uint8_t *)d_out, n, (uint32_t *)d_bad.p, ra.have_window ? 1u : 0u);
    if (ra.d_window)  // what the next round of this member has in front of it
        hipLaunchKernelGGL(k_win_tail, dim3(32), dim3(256), 0, stream, (const uint8_t *)d_out + reserve, (unsigned long long)total,
                           ra.have_window ? (const uint8_t *)d_win.p : (const uint8_t *)nullptr, (uint8_t *)ra.d_window);
    uint32_t bad = 0;
    hipError_t he = hipGetLastError();
    if (he == hipSuccess) he = hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, stream);
    if (he == hipSuccess) he = hipStreamSynchronize(stream);
    if (he != hipSuccess || bad) {
        if (bad)
            set_error("corrupt deflate stream (distance before the start of the output)");
        else
            set_error("inflate stream kernels failed: %s", hipGetErrorString(he));
        return bad ? EXG_E_PARSE : EXG_E_HIP;
    }
    ST_TRACE("[exg] inflate stream: %u chunks in the chain, %llu bytes, %.1f ms in all\n", n, (unsigned long long)total,
             (st_now() - t_begin) * 1e3);
    *d_out_p = d_out;
    ra.out_alloc = out_guard.sz;
    out_guard.p = nullptr;  // the caller's from here on
    *produced = total;
    ra.end_bit = end_bit;
    return EXG_OK;
}

// d_comp: the file's compressed bytes on the device (16-byte aligned); the DEFLATE stream of the member starts at
// comp_off and at most comp_size bytes may be read.  On success *d_out_p is a block of the library's device pool holding
// *produced inflated bytes (+ 64 bytes of zeroed slack) — the caller gives it back with exg_free_device(p) (or hipFree: pool
// blocks are whole allocations) — and *consumed the compressed bytes used (byte aligned after the final block).
// Synchronises `stream` several times (it returns sizes to the host).  On any error nothing is left allocated.
extern "C" int exg_inflate_stream(const void *d_comp_v, uint64_t comp_off, uint64_t comp_size, uint64_t chunk_bytes,
                                  void **d_out_p, uint64_t *produced, uint64_t *consumed, void *stream_v) {
    if (!consumed) {
        set_error("exg_inflate_stream: bad arguments");
        return EXG_E_INVALID_ARG;
    }
    *consumed = 0;
    RoundArgs ra;
    const int rc = inflate_stream_impl(d_comp_v, comp_off, comp_size, chunk_bytes, d_out_p, produced, ra, stream_v);
    if (rc) return rc;
    *consumed = (ra.end_bit + 7) / 8;
    return EXG_OK;
}

extern "C" int exg_inflate_round(exg_inflate_round_args *a) {
    if (!a) {
        set_error("exg_inflate_round: null argument");
        return EXG_E_INVALID_ARG;
    }
    RoundArgs ra;
    ra.start_bit = a->start_bit;
    ra.partial = a->partial != 0;
    ra.d_window = a->d_window;
    ra.have_window = a->have_window != 0;
    ra.front_reserve = a->front_reserve;
    ra.ratio_hint = a->ratio_hint;
    a->d_out = nullptr;
    a->produced = a->out_alloc = 0;
    a->need_more = a->final_block = 0;
    a->end_bit = a->start_bit;
    const int rc = inflate_stream_impl(a->d_comp, a->comp_off, a->comp_size, a->chunk_bytes, &a->d_out, &a->produced, ra, a->stream);
    if (rc) return rc;
    a->out_alloc = ra.out_alloc;
    a->end_bit = ra.end_bit;
    a->final_block = ra.final_block;
    a->need_more = ra.need_more;
    return EXG_OK;
}
