// exg_fasta.hip — FASTA record scan (read_fasta): id, description (NULL when absent), sequence.
//
// Semantics restated (not the code): noodles-fasta 0.27.0 Reader::read_definition / read_sequence and
// record::Definition::from_str as driven by exon 0.2.6 datasources::fasta (reached from
// rust/src/arrow_reader.rs:116-153; registered at exon/src/exon_extension.cpp:50):
//   definition line = any line whose first byte is '>' (the first line must be one); after '>' the
//   id runs to the first ASCII whitespace, the rest, trimmed, is the description; the sequence is
//   every following line up to the next definition line, LF (and a CR before it) removed and the
//   lines CONCATENATED — so it is materialised in a compacted payload buffer, it is not a slice of
//   the input.
//
// Round-1 implementation: the general multipass shape (line index -> per-line classification ->
// device-wide scans of record count and payload bytes -> definitions -> payload copy -> sequence
// string_t).  Whole-file buffers only (EXG_F_BOF | EXG_F_EOF): a FASTA record can span the whole
// input, so byte-range shards would split sequences.  The single-pass form is the
// next step for this format.
#include "exg_fasta.hpp"
#include "exg_lines.hpp"

namespace exg {

struct FastaArrays {
    const uint64_t *nl_pos;
    uint64_t *rec_pre;    // [i] = definition lines before line i   (T + 1 entries after the scan)
    uint64_t *pay_pre;    // [i] = sequence bytes before line i      (T + 1 entries)
    uint64_t *rec_start;  // [r] = payload offset of record r's sequence (n_rec + 1 entries)
    uint64_t *blk_first;  // [b] = line that holds payload byte b * 16384 (k_fa_copy's entry points)
};

__device__ __forceinline__ void line_bounds(const FastaDev &a, const uint64_t *nl_pos, uint64_t i, uint64_t *s,
                                            uint64_t *e) {
    uint64_t raw_end = nl_pos[i];
    uint64_t start = i ? nl_pos[i - 1] + 1 : 0;
    if (start > raw_end) start = raw_end;
    uint64_t end = raw_end;
    if (raw_end < a.n_bytes && end > start && a.d_in[end - 1] == '\r') end--;  // CR only in front of a real LF
    *s = start;
    *e = end;
}

// Per line i: is it a definition line, how many sequence bytes does it contribute — both follow from the
// line index and the flags k_emit_nl recorded next to it (the byte after each '\n', the byte before it), so
// no kernel has to go back to the input for them.
struct LineClass {
    const uint64_t *nl_pos;
    const uint8_t *flags;
    const uint8_t *d_in;
    uint64_t n_bytes;
    __device__ __forceinline__ void get(uint64_t i, uint64_t *def, uint64_t *len) const {
        const uint64_t raw_end = nl_pos[i];
        const uint64_t s = i ? nl_pos[i - 1] + 1 : 0;
        const bool d = i ? (flags[i - 1] & 1) != 0 : (raw_end > 0 && d_in[0] == '>');
        uint64_t e = raw_end;
        if (raw_end < n_bytes && e > s && (flags[i] & 2)) e--;  // CR only in front of a real LF
        *def = d ? 1 : 0;
        *len = d || e < s ? 0 : e - s;
    }
};

// ---- in-place exclusive scan of both per-line quantities (length lives on the device) -------------------
static constexpr uint32_t kScanChunk = 4096;  // 1024 threads x 4

__device__ __forceinline__ void block_scan2(unsigned long long sa, unsigned long long sb, unsigned long long *oa,
                                            unsigned long long *ob, unsigned long long *ta, unsigned long long *tb) {
    __shared__ unsigned long long s_a[16], s_b[16];
    unsigned long long ia = sa, ib = sb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long xa = __shfl_up(ia, d, 64), xb = __shfl_up(ib, d, 64);
        if ((int)(threadIdx.x & 63) >= d) ia += xa, ib += xb;
    }
    if ((threadIdx.x & 63) == 63) s_a[threadIdx.x >> 6] = ia, s_b[threadIdx.x >> 6] = ib;
    __syncthreads();
    unsigned long long fa = 0, fb = 0, za = 0, zb = 0;
    for (uint32_t k = 0; k < 16; k++) {
        if (k < (threadIdx.x >> 6)) fa += s_a[k], fb += s_b[k];
        za += s_a[k], zb += s_b[k];
    }
    *oa = fa + ia - sa;
    *ob = fb + ib - sb;
    *ta = za;
    *tb = zb;
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_fa_scan_local(LineClass lc, FastaArrays w, const ScanWsHeader *hdr,
                                                        uint64_t *block_sums) {
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
    if (base > T) return;  // entries 0..T
    uint64_t d[4], l[4], sd = 0, sl = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        d[k] = l[k] = 0;
        if (idx < T) lc.get(idx, &d[k], &l[k]);
        sd += d[k];
        sl += l[k];
    }
    unsigned long long od, ol, td, tl;
    block_scan2(sd, sl, &od, &ol, &td, &tl);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        if (idx <= T) w.rec_pre[idx] = od, w.pay_pre[idx] = ol;
        od += d[k];
        ol += l[k];
    }
    if (threadIdx.x == 0) block_sums[2 * blockIdx.x] = td, block_sums[2 * blockIdx.x + 1] = tl;
}

__global__ __launch_bounds__(1024) void k_fa_scan_blocks(uint64_t *block_sums, const ScanWsHeader *hdr) {
    __shared__ unsigned long long s_ra, s_rb;
    const uint64_t n = (hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap) + 1;
    const uint64_t nb = (n + kScanChunk - 1) / kScanChunk;
    if (threadIdx.x == 0) s_ra = s_rb = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nb; base += 1024) {
        uint64_t idx = base + threadIdx.x;
        unsigned long long ca = idx < nb ? block_sums[2 * idx] : 0, cb = idx < nb ? block_sums[2 * idx + 1] : 0;
        unsigned long long oa, ob, ta, tb;
        block_scan2(ca, cb, &oa, &ob, &ta, &tb);
        if (idx < nb) block_sums[2 * idx] = s_ra + oa, block_sums[2 * idx + 1] = s_rb + ob;
        __syncthreads();
        if (threadIdx.x == 0) s_ra += ta, s_rb += tb;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_fa_scan_add(FastaArrays w, const ScanWsHeader *hdr, const uint64_t *block_sums) {
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    const uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
    if (base > T || blockIdx.x == 0) return;
    const uint64_t ad = block_sums[2 * blockIdx.x], al = block_sums[2 * blockIdx.x + 1];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        if (idx <= T) w.rec_pre[idx] += ad, w.pay_pre[idx] += al;
    }
}

// ---- definitions ----------------------------------------------------------------------------------------
// pass 1, thread = line: remember which line defines record r (kept in rec_start[r] until pass 2)
__global__ __launch_bounds__(256) void k_fa_def_lines(FastaDev a, FastaArrays w, ScanWsHeader *hdr) {
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = w.rec_pre[i];
        if (w.rec_pre[i + 1] != r) w.rec_start[r] = i;
    }
}

// pass 2, thread = record (every lane has a definition to parse: the per-line form left 96 % of the lanes
// idle while the few definition lanes walked their bytes): id / description, then rec_start[r] becomes the
// payload offset of the record's sequence
__global__ __launch_bounds__(256) void k_fa_defs(FastaDev a, FastaArrays w, ScanWsHeader *hdr) {
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    const uint64_t n_rec = w.rec_pre[T];
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = w.rec_start[r];
        w.rec_start[r] = w.pay_pre[i];  // its sequence starts where the payload stands
        if (!no_store && r >= a.capacity) {
            atomicOr(&hdr->flags, EXG_RF_CAPACITY);
            continue;
        }
        uint64_t s, e;
        line_bounds(a, w.nl_pos, i, &s, &e);
        uint32_t code = 0;
        if ((hdr->flags & EXG_RF_NON_ASCII) && !utf8_valid_global(a.d_in, s, e)) code = EXG_PE_INVALID_UTF8;
        uint64_t id_s = s + 1, id_e = id_s;
        while (id_e < e && !is_ascii_ws(a.d_in[id_e])) id_e++;
        if (!code && id_e == id_s) code = EXG_PE_FASTA_MISSING_NAME;
        bool has_desc = id_e < e;
        uint64_t d_s = has_desc ? id_e + 1 : e, d_e = e;
        if (has_desc) {
            int l;
            while (d_s < d_e && (l = ws_len_fwd(a.d_in, d_s, d_e)) > 0) d_s += (uint64_t)l;
            while (d_e > d_s && (l = ws_len_bwd(a.d_in, d_s, d_e)) > 0) d_e -= (uint64_t)l;
        }
        if (!code && (id_e - id_s > 0xFFFFFFFFull || d_e - d_s > 0xFFFFFFFFull)) code = EXG_PE_FIELD_TOO_LONG;
        if (code) {
            atomicMin(&hdr->err_word, (r << 8) | code);
            atomicMin(&hdr->err_off, (unsigned long long)s);
        }
        if (!no_store) {
            uint4 z = {0, 0, 0, 0};
            reinterpret_cast<uint4 *>(a.d_id)[r] = make_string_global(a.d_in, id_s, id_e - id_s, a.payload_base);
            reinterpret_cast<uint4 *>(a.d_desc)[r] =
                has_desc ? make_string_global(a.d_in, d_s, d_e - d_s, a.payload_base) : z;
            if (has_desc) atomicOr((unsigned long long *)&a.d_desc_valid[r >> 6], 1ull << (r & 63));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.rec_start[n_rec] = w.pay_pre[T];  // sentinel: total payload
        if (T > 0 && w.rec_pre[1] == 0) {   // the reader wants a definition first
            uint64_t s, e;
            line_bounds(a, w.nl_pos, 0, &s, &e);
            atomicMin(&hdr->err_word, (0ull << 8) | (s == e ? EXG_PE_FASTA_EMPTY_DEF : EXG_PE_FASTA_MISSING_PREFIX));
        }
    }
}

// ---- payload: gather every sequence byte into its compacted position ------------------------------------
// Output-centric: a workgroup owns 16 KiB of the payload, a thread 64 bytes of it (four aligned 16-byte
// stores).  The lines that feed the workgroup's range are found by two binary searches in the payload
// prefix, their (payload offset, start) pairs are staged in LDS, and every thread locates its first
// line there.  (The first version scattered input bytes with byte stores: 3.4 ms per GB, 2/3 of the scan.)
// Fast path (the common shape: lines of >= 16 bytes on average): the input bytes that feed the
// workgroup's 16 KiB are one contiguous range of <= kStageBytes — they are staged in LDS with coalesced
// 16-byte loads together with the (payload offset, start) pairs of their lines, and a thread assembles its
// 64 bytes from LDS dwords (two aligned reads + v_alignbyte per unaligned dword).  Anything else (very
// short lines, long runs of definition lines) takes the general path: binary searches in the global
// prefix and byte loads.  (First version, scattering input bytes with byte stores: 3.4 ms per GB; byte
// gathers from global memory: 2.25 ms; staged: see DESIGN.md.)
static constexpr uint32_t kFastLines = 1024;
static constexpr uint32_t kStageBytes = 20480;

__device__ __forceinline__ uint64_t line_of_payload(const uint64_t *pay_pre, uint64_t lo, uint64_t hi, uint64_t o) {
    // largest i in [lo, hi) with pay_pre[i] <= o  (pay_pre is non-decreasing; hi is exclusive, pay_pre[lo] <= o)
    while (hi - lo > 1) {
        uint64_t mid = (lo + hi) >> 1;
        if (pay_pre[mid] <= o)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// entry points of the copy: the line holding the first byte of every 16 KiB payload block (a thread per
// line looks at the block boundaries inside its own payload range; usually none, sometimes one)
__global__ __launch_bounds__(256) void k_fa_block_index(FastaDev a, FastaArrays w, ScanWsHeader *hdr) {
    if (a.flags & EXG_F_NO_STORE) return;
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p0 = w.pay_pre[i], p1 = w.pay_pre[i + 1];
        for (uint64_t b = (p0 + 16383) >> 14; (b << 14) < p1; b++) w.blk_first[b] = i;
    }
}

__global__ __launch_bounds__(256) void k_fa_copy(FastaDev a, FastaArrays w, ScanWsHeader *hdr) {
    if (a.flags & EXG_F_NO_STORE) return;
    __shared__ int32_t s_pay[kFastLines + 1];  // payload offset of the line, relative to o_begin
    __shared__ int32_t s_start[kFastLines];    // first byte of the line, relative to in_base
    __shared__ __attribute__((aligned(16))) uint32_t s_in[kStageBytes / 4 + 8];
    __shared__ uint64_t s_l1;
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    const uint64_t total = w.pay_pre[T];
    for (uint64_t blk = blockIdx.x; blk * 16384 < total; blk += gridDim.x) {
        const uint64_t o_begin = blk * 16384, o_end = o_begin + 16384 < total ? o_begin + 16384 : total;
        // l0 = the line that holds byte o_begin; l1 = the line that holds byte o_end (one more line than needed
        // when o_end is a line start: harmless) or, in the last block, the last line with sequence bytes
        const uint64_t l0 = w.blk_first[blk];
        if (o_end == total && threadIdx.x == 0) s_l1 = line_of_payload(w.pay_pre, l0, T + 1, o_end - 1);
        __syncthreads();
        const uint64_t l1 = o_end == total ? s_l1 : w.blk_first[blk + 1];
        const uint64_t st0 = l0 ? w.nl_pos[l0 - 1] + 1 : 0, st1 = l1 ? w.nl_pos[l1 - 1] + 1 : 0;
        const uint64_t in_begin = st0 + (o_begin - w.pay_pre[l0]);
        const uint64_t in_end = st1 + (o_end - w.pay_pre[l1]);  // one past the last input byte used
        const uint64_t in_base = in_begin & ~15ull;
        const bool fast = l1 - l0 + 1 <= kFastLines && in_end - in_base <= kStageBytes;
        uint8_t *dst = a.d_payload + o_begin + (uint64_t)threadIdx.x * 64;  // 64-byte aligned (hipMalloc base)
        if (fast) {
            const uint32_t nl = (uint32_t)(l1 - l0 + 1);
            // line l0 is entered at payload byte o_begin (its own start may be gigabytes back: single-line
            // sequences), the last line's end is only ever compared against: both are clamped into 32 bits
            for (uint32_t k = threadIdx.x; k <= nl; k += 256) {
                const uint64_t v = w.pay_pre[l0 + k];
                s_pay[k] = k == 0 ? 0 : (int32_t)(v - o_begin < (1u << 20) ? v - o_begin : (1u << 20));
            }
            for (uint32_t k = threadIdx.x; k < nl; k += 256)
                s_start[k] = k == 0 ? (int32_t)(in_begin - in_base) : (int32_t)(w.nl_pos[l0 + k - 1] + 1 - in_base);
            const uint32_t n16 = (uint32_t)((in_end - in_base + 15) / 16);
            for (uint32_t q = threadIdx.x; q < n16; q += 256)
                reinterpret_cast<uint4 *>(s_in)[q] = *reinterpret_cast<const uint4 *>(a.d_in + in_base + (uint64_t)q * 16);
            __syncthreads();
            int32_t o = (int32_t)threadIdx.x * 64;
            const int32_t o_len = (int32_t)(o_end - o_begin);
            if (o < o_len) {
                const int32_t stop = o + 64 < o_len ? o + 64 : o_len;
                uint32_t lo = 0, hi = nl;  // largest k < nl with s_pay[k] <= o
                while (hi - lo > 1) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (s_pay[mid] <= o)
                        lo = mid;
                    else
                        hi = mid;
                }
                uint32_t k = lo;
                int32_t p0 = s_pay[k], p1 = s_pay[k + 1], st = s_start[k];
                uint32_t wreg[16];
                int32_t filled = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    uint32_t word = 0;
                    if (o + 4 <= stop && o + 4 <= p1) {  // the whole dword comes from the current line
                        const uint32_t src = (uint32_t)(st + (o - p0));
                        const uint32_t w0 = s_in[src >> 2], w1 = s_in[(src >> 2) + 1];
                        word = __builtin_amdgcn_alignbyte(w1, w0, src & 3);
                        o += 4;
                        filled += 4;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (o < stop) {
                                while (p1 <= o) {  // skip lines without sequence bytes
                                    k++;
                                    p0 = s_pay[k], p1 = s_pay[k + 1], st = s_start[k];
                                }
                                const uint32_t src = (uint32_t)(st + (o - p0));
                                word |= ((s_in[src >> 2] >> (8 * (src & 3))) & 0xFFu) << (8 * r);
                                o++;
                                filled++;
                            }
                        }
                    }
                    wreg[q] = word;
                }
                if (filled == 64) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        reinterpret_cast<uint4 *>(dst)[q] = make_uint4(wreg[4 * q], wreg[4 * q + 1], wreg[4 * q + 2], wreg[4 * q + 3]);
                } else {
                    for (int32_t b = 0; b < filled; b++) dst[b] = (uint8_t)(wreg[b >> 2] >> (8 * (b & 3)));
                }
            }
        } else {
            uint64_t o = o_begin + (uint64_t)threadIdx.x * 64;
            if (o < o_end) {
                const uint64_t stop = o + 64 < o_end ? o + 64 : o_end;
                uint64_t li = line_of_payload(w.pay_pre, l0, l1 + 1, o);  // line holding payload byte o
                uint32_t wreg[16];
                uint64_t p0 = w.pay_pre[li], p1 = w.pay_pre[li + 1], st = li ? w.nl_pos[li - 1] + 1 : 0;
                uint32_t filled = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    uint32_t word = 0;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (o < stop) {
                            while (p1 <= o) {
                                li++;
                                p0 = p1;
                                p1 = w.pay_pre[li + 1];
                                st = w.nl_pos[li - 1] + 1;
                            }
                            word |= (uint32_t)a.d_in[st + (o - p0)] << (8 * r);
                            o++;
                            filled++;
                        }
                    }
                    wreg[q] = word;
                }
                if (filled == 64) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        reinterpret_cast<uint4 *>(dst)[q] = make_uint4(wreg[4 * q], wreg[4 * q + 1], wreg[4 * q + 2], wreg[4 * q + 3]);
                } else {
                    for (uint32_t b = 0; b < filled; b++) dst[b] = (uint8_t)(wreg[b >> 2] >> (8 * (b & 3)));
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_fa_seq_strings(FastaDev a, FastaArrays w, ScanWsHeader *hdr) {
    if (a.flags & EXG_F_NO_STORE) return;
    const uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    const uint64_t n_rec = w.rec_pre[T];
    const uint64_t n = n_rec < a.capacity ? n_rec : a.capacity;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t s = w.rec_start[r], len = w.rec_start[r + 1] - s;
        if (len > 0xFFFFFFFFull) {
            atomicMin(&hdr->err_word, (r << 8) | EXG_PE_FIELD_TOO_LONG);
            len = 0;
        }
        // sequences are validated after their definition (exon FASTAArrayBuilder::append order)
        if ((hdr->flags & EXG_RF_NON_ASCII) && reinterpret_cast<const uint32_t *>(a.d_id)[r * 4] != 0 &&
            !utf8_valid_global(a.d_payload, s, s + len)) {
            atomicMin(&hdr->err_word, (r << 8) | EXG_PE_INVALID_UTF8);
            // offset of the record = start of its definition line: first line i with rec_pre[i + 1] > r
            uint64_t lo = 0, hi = T;
            while (lo < hi) {
                uint64_t mid = (lo + hi) >> 1;
                if (w.rec_pre[mid + 1] > r)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            atomicMin(&hdr->err_off, (unsigned long long)(lo ? w.nl_pos[lo - 1] + 1 : 0));
        }
        reinterpret_cast<uint4 *>(a.d_seq)[r] = make_string_global(a.d_payload, s, len, a.seq_payload_base);
    }
}

__global__ void k_fa_finalize(FastaDev a, FastaArrays w, ScanWsHeader *hdr, exg_scan_result *res) {
    if (threadIdx.x || blockIdx.x) return;
    const uint64_t T = hdr->total_lines, Tc = T < hdr->lines_cap ? T : hdr->lines_cap;
    const uint64_t n_owned = w.rec_pre[Tc];
    exg_scan_result r;
    r.n_lines = T;
    r.flags = hdr->flags;
    if (T > hdr->lines_cap) r.flags |= EXG_RF_INDEX_OVERFLOW;
    r.payload_bytes = w.pay_pre[Tc];
    r.redo_tiles = 0;
    r.error_code = 0;
    r.error_offset = ~0ull;
    r.error_record = ~0ull;
    uint64_t n_rec = (n_owned < a.capacity || (a.flags & EXG_F_NO_STORE)) ? n_owned : a.capacity;
    uint64_t consumed = a.n_bytes;
    unsigned long long err = hdr->err_word;
    if (err != kNoError) {
        uint64_t rec = err >> 8;
        r.error_code = (uint32_t)(err & 0xFF);
        r.error_record = rec;
        // offset of the failing record's definition line (errors on the sequence report it too)
        r.error_offset = hdr->err_off != ~0ull ? hdr->err_off : 0;
        if (r.error_code == EXG_PE_FASTA_EMPTY_DEF || r.error_code == EXG_PE_FASTA_MISSING_PREFIX) r.error_offset = 0;
        if (rec < n_rec) n_rec = rec;
        consumed = r.error_offset;
    }
    r.n_records = n_rec;
    r.consumed_bytes = consumed;
    *res = r;
}

__global__ void k_init_hdr(ScanWsHeader *hdr, uint64_t lines_cap, uint32_t mode);

}  // namespace exg

using namespace exg;

extern "C" int exg_fasta_scan(const exg_fasta_scan_args *a) {
    if (!a || !a->d_result || !a->d_workspace || ((uintptr_t)a->d_workspace & 255) || (a->n_bytes && !a->d_input) ||
        ((uintptr_t)a->d_input & 15)) {
        set_error("exg_fasta_scan: bad arguments (null pointer, unaligned input or workspace)");
        return EXG_E_INVALID_ARG;
    }
    if (a->flags & ~EXG_F_ALL) {
        set_error("exg_fasta_scan: unknown flag bits 0x%x", a->flags & ~EXG_F_ALL);
        return EXG_E_INVALID_ARG;
    }
    // A buffer begins with a record (lead = 0, EXG_F_BOF).  Without EXG_F_EOF it is a batch of a longer input: the last record
    // in it is still open and is left to the next batch (consumed_bytes = where its definition line begins; 0 records when the
    // buffer holds only that one: the caller widens the batch — a FASTA record can be as long as the input).  The one-pass
    // form implements that; the multipass form (the tests' differential partner) scans whole inputs only.
    if (a->lead != 0 || !(a->flags & EXG_F_BOF) || (!(a->flags & EXG_F_EOF) && a->algo == EXG_ALGO_MULTIPASS)) {
        set_error("exg_fasta_scan: a buffer begins with a record (lead = 0, EXG_F_BOF); EXG_ALGO_MULTIPASS scans whole inputs only (EXG_F_EOF)");
        return EXG_E_UNSUPPORTED;
    }
    const bool no_store = a->flags & EXG_F_NO_STORE;
    if (!no_store && a->capacity_records &&
        (!a->d_id || !a->d_description || !a->d_sequence || !a->d_description_validity || !a->d_seq_payload)) {
        set_error("exg_fasta_scan: null output");
        return EXG_E_INVALID_ARG;
    }
    FastqWsLayout l = fastq_ws_layout(a->n_bytes, a->workspace_bytes, 4);
    if (a->workspace_bytes < fastq_ws_layout(a->n_bytes, 0, 4).off_nl_pos + 64 * 4 || l.lines_cap < 2) {
        set_error("exg_fasta_scan: workspace too small");
        return EXG_E_INVALID_ARG;
    }
    FastaDev dev;
    dev.d_in = (const uint8_t *)a->d_input;
    dev.n_bytes = a->n_bytes;
    dev.payload_base = a->payload_base;
    dev.seq_payload_base = a->seq_payload_base;
    dev.flags = a->flags;
    dev.pad = 0;
    dev.d_id = a->d_id;
    dev.d_desc = a->d_description;
    dev.d_seq = a->d_sequence;
    dev.d_desc_valid = a->d_description_validity;
    dev.d_payload = a->d_seq_payload;
    dev.capacity = a->capacity_records;
    hipStream_t stream = (hipStream_t)a->stream;
    uint8_t *ws = (uint8_t *)a->d_workspace;
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    FastaArrays w;
    uint64_t *base = reinterpret_cast<uint64_t *>(ws + l.off_nl_pos);
    w.nl_pos = base;
    w.rec_pre = base + (l.lines_cap + 2);
    w.pay_pre = base + 2 * (l.lines_cap + 2);
    w.rec_start = base + 3 * (l.lines_cap + 2);
    w.blk_first = reinterpret_cast<uint64_t *>(ws + l.off_tile_desc);  // n / 16384 + 2 entries fit (24 B per 16 KiB tile)
    uint64_t *block_sums = reinterpret_cast<uint64_t *>(ws + l.off_block_sums);
    if (a->capacity_records && !no_store)
        EXG_HIP_CHECK(hipMemsetAsync(a->d_description_validity, 0, (size_t)((a->capacity_records + 63) / 64) * 8, stream));
    if (a->algo != EXG_ALGO_MULTIPASS) return run_fasta_tiled(dev, ws, l, a->d_result, stream);
    hipLaunchKernelGGL(k_init_hdr, dim3(1), dim3(1), 0, stream, hdr, l.lines_cap, 0u);
    uint8_t *line_flags = reinterpret_cast<uint8_t *>(w.rec_start);  // dead before k_fa_def_lines fills rec_start
    int rc = launch_line_index(dev.d_in, dev.n_bytes, 0, ws, l, 1, 0, stream, nullptr, line_flags);
    if (rc) return rc;
    uint64_t est_lines = dev.n_bytes / 16 + 256;
    uint32_t grid = (uint32_t)((est_lines + 255) / 256 < 4096 ? (est_lines + 255) / 256 : 4096);
    // scans: the grid covers the index capacity, blocks past the real line count return at once
    uint64_t cover = dev.n_bytes + 2 < l.lines_cap + 1 ? dev.n_bytes + 2 : l.lines_cap + 1;  // lines <= bytes + 1
    uint32_t sgrid = (uint32_t)((cover + kScanChunk - 1) / kScanChunk);
    if (sgrid == 0) sgrid = 1;
    LineClass lc{w.nl_pos, line_flags, dev.d_in, dev.n_bytes};
    hipLaunchKernelGGL(k_fa_scan_local, dim3(sgrid), dim3(1024), 0, stream, lc, w, hdr, block_sums);
    hipLaunchKernelGGL(k_fa_scan_blocks, dim3(1), dim3(1024), 0, stream, block_sums, hdr);
    hipLaunchKernelGGL(k_fa_scan_add, dim3(sgrid), dim3(1024), 0, stream, w, hdr, block_sums);
    hipLaunchKernelGGL(k_fa_def_lines, dim3(grid), dim3(256), 0, stream, dev, w, hdr);
    hipLaunchKernelGGL(k_fa_defs, dim3(grid), dim3(256), 0, stream, dev, w, hdr);
    uint64_t n_blk = (dev.n_bytes + 16383) / 16384;  // payload <= input
    uint32_t cgrid = (uint32_t)(n_blk < 16384 ? n_blk : 16384);
    if (cgrid == 0) cgrid = 1;
    hipLaunchKernelGGL(k_fa_block_index, dim3(grid), dim3(256), 0, stream, dev, w, hdr);
    hipLaunchKernelGGL(k_fa_copy, dim3(cgrid), dim3(256), 0, stream, dev, w, hdr);
    hipLaunchKernelGGL(k_fa_seq_strings, dim3(grid), dim3(256), 0, stream, dev, w, hdr);
    hipLaunchKernelGGL(k_fa_finalize, dim3(1), dim3(1), 0, stream, dev, w, hdr, a->d_result);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}
