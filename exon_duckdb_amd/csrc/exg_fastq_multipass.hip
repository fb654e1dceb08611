// exg_fastq_multipass.hip — general FASTQ path (EXG_ALGO_MULTIPASS): line index + one thread per
// record reading its fields from global memory.  Any line length, any record size; ~3x the input
// bytes of HBM traffic.  It is the fallback of the fused kernel and its differential partner.
//
// Semantics restated (not the code): noodles-fastq 0.8.0 Reader::read_record + exon 0.2.6
// FASTQArrayBuilder::append, reached from rust/src/arrow_reader.rs:116-153, and the Arrow ->
// string_t conversion DuckDB performs for exon/src/exon/arrow_table_function/module.cpp:289.
#include "exg_fastq.hpp"
#include "exg_lines.hpp"

namespace exg {

// Lines of candidate record j (its quality line is line iq = i0 + 4 j of the buffer).
__device__ __forceinline__ FastqGeom fastq_geometry(const uint8_t *__restrict__ d_in, uint64_t n_bytes,
                                                     const uint64_t *__restrict__ nl_pos, int64_t iq,
                                                     bool line_starts_at_0) {
    int64_t p[5];
    if (iq - 4 >= 0)
        p[0] = (int64_t)nl_pos[iq - 4];
    else if (iq - 4 == -1 && line_starts_at_0)
        p[0] = -1;
    else {
        FastqGeom g = {};
        g.resolved = false;
        return g;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t li = iq - 3 + k;
        p[k + 1] = li >= 0 ? (int64_t)nl_pos[li] : 0;
    }
    return fastq_geometry_at(d_in, n_bytes, p);
}

struct FastqCounts {
    uint64_t i0, n_cand, n_hc;
};

__device__ __forceinline__ FastqCounts fastq_counts(const ScanWsHeader *hdr, uint64_t first_line_index) {
    FastqCounts c;
    uint64_t p0 = first_line_index - hdr->halo_nl;
    uint64_t T = hdr->total_lines < hdr->lines_cap ? hdr->total_lines : hdr->lines_cap;
    c.i0 = (3 - p0) & 3;
    c.n_cand = T > c.i0 ? (T - c.i0 + 3) / 4 : 0;
    c.n_hc = hdr->halo_nl > c.i0 ? (hdr->halo_nl - c.i0 + 3) / 4 : 0;
    return c;
}

__global__ __launch_bounds__(256) void k_fastq_records(FastqDev a, const uint64_t *__restrict__ nl_pos,
                                                       ScanWsHeader *hdr, const unsigned int *gate) {
    if (gate && *gate == 0) return;
    const uint8_t *d_in = a.d_in;
    FastqCounts c = fastq_counts(hdr, a.first_line_index);
    uint64_t n_iter = (c.n_cand + 63) / 64;  // wave-granular so ballots stay uniform
    uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t it = wave; it < n_iter; it += n_waves) {
        uint64_t j = it * 64 + lane_id();
        bool owned = j < c.n_cand && j >= c.n_hc;
        int64_t out = (int64_t)j - (int64_t)c.n_hc;
        const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
        bool in_cap = owned && (no_store || (uint64_t)out < a.capacity);
        if (owned && !in_cap) atomicOr(&hdr->flags, EXG_RF_CAPACITY);
        bool desc_valid = false;
        if (in_cap) {
            FastqGeom g = fastq_geometry(d_in, a.n_bytes, nl_pos, (int64_t)(c.i0 + 4 * j), a.flags & EXG_F_BOF);
            if (!g.resolved) {
                atomicAdd(&hdr->n_unresolved, 1ull);
                atomicOr(&hdr->flags, EXG_RF_HEAD_UNRESOLVED);
                uint4 z = {0, 0, 0, 0};
                if (!no_store) {
                    reinterpret_cast<uint4 *>(a.d_name)[out] = z;
                    reinterpret_cast<uint4 *>(a.d_desc)[out] = z;
                    reinterpret_cast<uint4 *>(a.d_seq)[out] = z;
                    reinterpret_cast<uint4 *>(a.d_qual)[out] = z;
                }
            } else {
                uint32_t code = 0;
                if (!g.name_ok)
                    code = EXG_PE_FASTQ_NAME_PREFIX;
                else if (!g.plus_ok)
                    code = EXG_PE_FASTQ_PLUS_PREFIX;
                uint64_t lens[4] = {g.name_e - g.s[0], g.e[0] - g.desc_s, g.e[1] - g.s[1], g.e[3] - g.s[3]};
                if (!code && (lens[0] > 0xFFFFFFFFull || lens[1] > 0xFFFFFFFFull || lens[2] > 0xFFFFFFFFull ||
                              lens[3] > 0xFFFFFFFFull))
                    code = EXG_PE_FIELD_TOO_LONG;
                if (code) atomicMin(&hdr->err_word, ((unsigned long long)out << 8) | code);
                desc_valid = lens[1] != 0 && !no_store;
                uint4 z = {0, 0, 0, 0};
                if (!no_store) {
                    reinterpret_cast<uint4 *>(a.d_name)[out] = make_string_global(d_in, g.s[0], lens[0], a.payload_base);
                    reinterpret_cast<uint4 *>(a.d_desc)[out] =
                        desc_valid ? make_string_global(d_in, g.desc_s, lens[1], a.payload_base) : z;
                    reinterpret_cast<uint4 *>(a.d_seq)[out] = make_string_global(d_in, g.s[1], lens[2], a.payload_base);
                    reinterpret_cast<uint4 *>(a.d_qual)[out] = make_string_global(d_in, g.s[3], lens[3], a.payload_base);
                }
            }
        }
        // description validity: one ballot per wave, at most two word updates
        unsigned long long b = __ballot(desc_valid);
        if (b) {
            int64_t out_base = (int64_t)(it * 64) - (int64_t)c.n_hc;
            if (out_base < 0) {
                b >>= (uint64_t)(-out_base);
                out_base = 0;
            }
            if (lane_id() == 0 && b) {
                uint32_t sh = (uint32_t)(out_base & 63);
                unsigned long long lo = b << sh, hi = sh ? b >> (64 - sh) : 0;
                if (lo) atomicOr((unsigned long long *)&a.d_desc_valid[out_base >> 6], lo);
                if (hi) atomicOr((unsigned long long *)&a.d_desc_valid[(out_base >> 6) + 1], hi);
            }
        }
    }
}

// Slow kernel, only does work when pass 1 saw a byte >= 0x80.
__global__ __launch_bounds__(256) void k_fastq_utf8(FastqDev a, const uint64_t *__restrict__ nl_pos,
                                                    ScanWsHeader *hdr, const unsigned int *gate) {
    if (gate && *gate == 0) return;
    if (!(hdr->flags & EXG_RF_NON_ASCII)) return;
    FastqCounts c = fastq_counts(hdr, a.first_line_index);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + c.n_hc; j < c.n_cand;
         j += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t out = j - c.n_hc;
        if (out >= a.capacity && !(a.flags & EXG_F_NO_STORE)) break;
        FastqGeom g = fastq_geometry(a.d_in, a.n_bytes, nl_pos, (int64_t)(c.i0 + 4 * j), a.flags & EXG_F_BOF);
        if (!g.resolved) continue;
        bool ok = utf8_valid_global(a.d_in, g.s[0], g.name_e) && utf8_valid_global(a.d_in, g.desc_s, g.e[0]) &&
                  utf8_valid_global(a.d_in, g.s[1], g.e[1]) && utf8_valid_global(a.d_in, g.s[3], g.e[3]);
        if (!ok) atomicMin(&hdr->err_word, ((unsigned long long)out << 8) | EXG_PE_INVALID_UTF8);
    }
}

__global__ void k_fastq_finalize_mp(FastqDev a, const uint64_t *__restrict__ nl_pos, ScanWsHeader *hdr,
                                    exg_scan_result *res, const unsigned int *gate) {
    if (threadIdx.x || blockIdx.x) return;
    if (gate && *gate == 0) return;
    FastqCounts c = fastq_counts(hdr, a.first_line_index);
    uint64_t n_owned = c.n_cand - (c.n_hc < c.n_cand ? c.n_hc : c.n_cand);
    uint64_t T = hdr->total_lines;
    uint32_t flags = hdr->flags;
    if (gate) flags |= EXG_RF_FALLBACK;
    if (T > hdr->lines_cap) flags |= EXG_RF_INDEX_OVERFLOW;
    // EOF inside a record: 1 or 2 lines of a last record (noodles: UnexpectedEof at the '+' read)
    unsigned long long err = hdr->err_word;
    uint64_t p0 = a.first_line_index - hdr->halo_nl;
    if ((a.flags & EXG_F_EOF) && ((p0 + T) & 3) != 0 && T > hdr->halo_nl) {
        // the reader checks '@' before it can run out of lines
        int64_t name_li = (int64_t)(c.i0 + 4 * c.n_cand) - 3;
        uint64_t start = name_li - 1 >= 0 ? nl_pos[name_li - 1] + 1 : 0;
        uint32_t code = (start < a.n_bytes && a.d_in[start] == '@') ? EXG_PE_UNEXPECTED_EOF : EXG_PE_FASTQ_NAME_PREFIX;
        unsigned long long w = ((unsigned long long)n_owned << 8) | code;
        if (w < err) err = w;
    }
    exg_scan_result r;
    r.n_lines = T - hdr->halo_nl;
    r.flags = flags;
    r.payload_bytes = 0;
    r.redo_tiles = 0;
    r.error_code = 0;
    r.error_offset = ~0ull;
    r.error_record = ~0ull;
    uint64_t n_rec = (n_owned < a.capacity || (a.flags & EXG_F_NO_STORE)) ? n_owned : a.capacity;
    if (err != kNoError) {
        uint64_t rec = err >> 8;
        r.error_code = (uint32_t)(err & 0xFF);
        r.error_record = rec;
        if (rec < n_rec) n_rec = rec;
        // offset of the failing record's first byte: line index (local) of its name line
        int64_t name_li = (int64_t)(c.i0 + 4 * (rec + c.n_hc)) - 3;
        if (name_li - 1 >= 0 && (uint64_t)(name_li - 1) < hdr->lines_cap)
            r.error_offset = nl_pos[name_li - 1] + 1;
        else
            r.error_offset = 0;
    }
    r.n_records = n_rec;
    // consumed: just past the quality line of the last emitted record
    if (n_rec > 0) {
        uint64_t iq = c.i0 + 4 * (n_rec - 1 + c.n_hc);
        uint64_t e = iq < hdr->lines_cap ? nl_pos[iq] : a.n_bytes;
        r.consumed_bytes = e + 1 < a.n_bytes ? e + 1 : a.n_bytes;
    } else {
        r.consumed_bytes = a.lead;
    }
    *res = r;
}

// mode 0: full reset.  mode 1: reset for the general path after the fused kernel, only when the
// fused kernel raised `overflow` (which is kept, it gates the kernels that follow).
__global__ void k_init_hdr(ScanWsHeader *hdr, uint64_t lines_cap, uint32_t mode) {
    if (threadIdx.x || blockIdx.x) return;
    unsigned int overflow = hdr->overflow;
    if (mode == 1 && overflow == 0) return;
    ScanWsHeader h;
    h.n_slow = h.slow_pad = 0;
    h.any_far = h.any_redo = 0;
    h.n_redo = h.redo_pad = 0;
    for (unsigned int i = 0; i < 18; i++) h.reserved[i] = 0;
    h.last_qend = 0;
    h.total_nl = h.total_lines = h.halo_nl = h.n_unresolved = 0;
    h.err_word = kNoError;
    h.err_off = ~0ull;
    h.consumed = 0;
    h.flags = 0;
    h.ticket = 0;
    h.epoch = 0;
    h.overflow = mode == 1 ? overflow : 0;
    h.lines_cap = lines_cap;
    *hdr = h;
}

__global__ void k_clear_words_gated(uint64_t *w, uint64_t n, const unsigned int *gate) {
    if (gate && *gate == 0) return;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        w[i] = 0;
}

int run_fastq_multipass(const exg_fastq_scan_args *args, const FastqDev &dev, uint8_t *ws, const FastqWsLayout &l,
                        hipStream_t stream, bool after_fused) {
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    const uint64_t *nl_pos = reinterpret_cast<const uint64_t *>(ws + l.off_nl_pos);
    const unsigned int *gate = after_fused ? &hdr->overflow : nullptr;
    hipLaunchKernelGGL(k_init_hdr, dim3(1), dim3(1), 0, stream, hdr, l.lines_cap, after_fused ? 1u : 0u);
    if (after_fused) {
        // the fused kernel may have set validity bits before it gave up: clear them, on the device,
        // only in that case (no host round trip on the stream)
        uint64_t words = (dev.capacity + 63) / 64;
        uint32_t g = (uint32_t)((words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024);
        if (words && !(dev.flags & EXG_F_NO_STORE)) hipLaunchKernelGGL(k_clear_words_gated, dim3(g), dim3(256), 0, stream, dev.d_desc_valid, words, gate);
    }
    int rc = launch_line_index(dev.d_in, dev.n_bytes, dev.lead, ws, l, (dev.flags & EXG_F_EOF) ? 2 : 0,
                               dev.first_line_index, stream, gate);
    if (rc) return rc;
    uint64_t est = dev.n_bytes / 64 + 256;  // grid-stride covers the rest
    uint32_t grid = (uint32_t)((est + 255) / 256 < 2048 ? (est + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_fastq_records, dim3(grid), dim3(256), 0, stream, dev, nl_pos, hdr, gate);
    hipLaunchKernelGGL(k_fastq_utf8, dim3(grid), dim3(256), 0, stream, dev, nl_pos, hdr, gate);
    hipLaunchKernelGGL(k_fastq_finalize_mp, dim3(1), dim3(1), 0, stream, dev, nl_pos, hdr, args->d_result, gate);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

}  // namespace exg
