// exg_lines.hip — general line index (multipass): count '\n' per tile, scan the tile counts,
// emit every newline offset.  Used by the general (any line length) paths of all three formats
// and as the differential partner of the fused single-pass kernels.
//
// Reference behaviour being replaced: the `memchr(b'\n')` line splitting inside the noodles
// readers that exon drives per record (external crates; call sites rust/src/arrow_reader.rs:116-153).
#include "exg_fastq_ws.hpp"
#include "exg_lines.hpp"

namespace exg {

// Pass 1: per 16 KiB tile newline count; also counts newlines inside the halo [0, lead) and ORs
// a non-ASCII flag.  Reads each input byte once, coalesced 16 B per lane.
__global__ __launch_bounds__(kMpThreads) void k_count_nl(const uint8_t *__restrict__ d_in, uint64_t n_bytes,
                                                         uint64_t lead, uint32_t *__restrict__ tile_counts,
                                                         ScanWsHeader *hdr, const unsigned int *gate) {
    __shared__ uint32_t s_cnt[4], s_halo[4], s_hi[4];
    if (gate && *gate == 0) return;
    for (uint64_t tile = blockIdx.x; tile * kMpTileBytes < n_bytes || (tile == 0 && n_bytes == 0);
         tile += gridDim.x) {
        uint32_t cnt = 0, halo = 0, hi = 0;
        uint64_t tile_off = tile * (uint64_t)kMpTileBytes;
#pragma unroll
        for (uint32_t j = 0; j < kMpIters; j++) {
            uint64_t off = tile_off + (uint64_t)j * kMpIterBytes + (uint64_t)threadIdx.x * 16;
            if (off < n_bytes) {
                uint4 v = *reinterpret_cast<const uint4 *>(d_in + off);
                uint32_t m = match16(v, 0x0A0A0A0Au);
                uint32_t h = nib4(v.x & 0x80808080u) | (nib4(v.y & 0x80808080u) << 4) |
                             (nib4(v.z & 0x80808080u) << 8) | (nib4(v.w & 0x80808080u) << 12);
                if (off + 16 > n_bytes) {
                    uint32_t keep = (1u << (uint32_t)(n_bytes - off)) - 1u;
                    m &= keep;
                    h &= keep;
                }
                cnt += __popc(m);
                hi |= h;
                if (off < lead) {
                    uint32_t hm = m;
                    if (off + 16 > lead) hm &= (1u << (uint32_t)(lead - off)) - 1u;
                    halo += __popc(hm);
                }
            }
        }
        // block reduce
        for (int d = 32; d > 0; d >>= 1) {
            cnt += __shfl_down(cnt, d, 64);
            halo += __shfl_down(halo, d, 64);
            hi |= __shfl_down(hi, d, 64);
        }
        uint32_t w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            s_cnt[w] = cnt;
            s_halo[w] = halo;
            s_hi[w] = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            uint32_t h = s_halo[0] + s_halo[1] + s_halo[2] + s_halo[3];
            tile_counts[tile] = c;
            if (h) atomicAdd(&hdr->halo_nl, (unsigned long long)h);
            if (s_hi[0] | s_hi[1] | s_hi[2] | s_hi[3]) atomicOr(&hdr->flags, EXG_RF_NON_ASCII);
        }
        __syncthreads();
        if (n_bytes == 0) break;
    }
}

// Pass 2: single-block exclusive scan of the tile counts; appends the virtual line ends that the
// readers' EOF rules imply (format specific, selected by eof_mode).
//   eof_mode 0: none (not at EOF)
//   eof_mode 1: an unterminated last line counts as a line (all formats)
//   eof_mode 2: FASTQ — additionally, a record that has its '+' line but no quality line gets an
//               empty quality line (noodles read_line returns 0 bytes at EOF without error)
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint8_t *__restrict__ d_in, uint64_t n_bytes,
                                                     uint64_t n_tiles, const uint32_t *__restrict__ tile_counts,
                                                     uint64_t *__restrict__ tile_offsets, uint64_t *nl_pos,
                                                     ScanWsHeader *hdr, int eof_mode, uint64_t first_line_index,
                                                     const unsigned int *gate) {
    __shared__ unsigned long long s_wave[16];
    if (gate && *gate == 0) return;
    __shared__ unsigned long long s_running;
    if (threadIdx.x == 0) s_running = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_tiles; base += 1024) {
        uint64_t t = base + threadIdx.x;
        unsigned long long c = t < n_tiles ? tile_counts[t] : 0;
        unsigned long long incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long o = __shfl_up(incl, d, 64);
            if ((int)(threadIdx.x & 63) >= d) incl += o;
        }
        uint32_t w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 63) s_wave[w] = incl;
        __syncthreads();
        unsigned long long wave_off = 0;
        for (uint32_t k = 0; k < w; k++) wave_off += s_wave[k];
        unsigned long long run = s_running;
        if (t < n_tiles) tile_offsets[t] = run + wave_off + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_running = run + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsigned long long total = s_running;
        hdr->total_nl = total;
        unsigned long long lines = total;
        if (eof_mode >= 1 && n_bytes > 0 && d_in[n_bytes - 1] != '\n') {
            if (lines < hdr->lines_cap) nl_pos[lines] = n_bytes;
            lines++;
        }
        if (eof_mode == 2) {
            unsigned long long p0 = first_line_index - hdr->halo_nl;
            if (((p0 + lines) & 3) == 3) {
                if (lines < hdr->lines_cap) nl_pos[lines] = n_bytes;
                lines++;
            }
        }
        hdr->total_lines = lines;
    }
}

// Pass 3: write the offset of every '\n' in order.  Re-reads the input once.
// line_flags (optional, FASTA): per newline, bit 0 = the byte after it is '>' (the NEXT line is a definition
// line), bit 1 = the byte before it is '\r' — both are in the registers of the thread that finds it, except
// at the edges of its 16-byte chunk.
__global__ __launch_bounds__(kMpThreads) void k_emit_nl(const uint8_t *__restrict__ d_in, uint64_t n_bytes,
                                                        const uint64_t *__restrict__ tile_offsets,
                                                        uint64_t *__restrict__ nl_pos, uint64_t lines_cap,
                                                        const unsigned int *gate, uint8_t *__restrict__ line_flags) {
    __shared__ uint32_t s_wave[4];
    if (gate && *gate == 0) return;
    for (uint64_t tile = blockIdx.x; tile * kMpTileBytes < n_bytes; tile += gridDim.x) {
        uint64_t tile_off = tile * (uint64_t)kMpTileBytes;
        uint64_t rank_base = tile_offsets[tile];
        for (uint32_t j = 0; j < kMpIters; j++) {
            uint64_t off = tile_off + (uint64_t)j * kMpIterBytes + (uint64_t)threadIdx.x * 16;
            uint32_t m = 0, gt = 0, cr = 0;
            if (off < n_bytes) {
                uint4 v = *reinterpret_cast<const uint4 *>(d_in + off);
                m = match16(v, 0x0A0A0A0Au);
                if (off + 16 > n_bytes) m &= (1u << (uint32_t)(n_bytes - off)) - 1u;
                if (line_flags && m) {
                    gt = match16(v, 0x3E3E3E3Eu) >> 1;  // bit b: byte b + 1 is '>'
                    cr = match16(v, 0x0D0D0D0Du) << 1;  // bit b: byte b - 1 is '\r'
                    if ((m & 0x8000u) && off + 16 < n_bytes && d_in[off + 16] == '>') gt |= 0x8000u;
                    if ((m & 1u) && off > 0 && d_in[off - 1] == '\r') cr |= 1u;
                }
            }
            uint32_t c = __popc(m);
            uint32_t incl = wave_incl_sum(c);
            uint32_t w = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 63) s_wave[w] = incl;
            __syncthreads();
            uint32_t wave_off = 0, iter_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            for (uint32_t k = 0; k < w; k++) wave_off += s_wave[k];
            uint64_t r = rank_base + wave_off + incl - c;
            while (m) {
                uint32_t b = __ffs(m) - 1;
                m &= m - 1;
                if (r < lines_cap) {
                    nl_pos[r] = off + b;
                    if (line_flags) line_flags[r] = (uint8_t)(((gt >> b) & 1u) | (((cr >> b) & 1u) << 1));
                }
                r++;
            }
            rank_base += iter_total;
            __syncthreads();
        }
    }
}

int launch_line_index(const uint8_t *d_in, uint64_t n_bytes, uint64_t lead, uint8_t *ws, const FastqWsLayout &l,
                      int eof_mode, uint64_t first_line_index, hipStream_t stream, const unsigned int *gate,
                      uint8_t *line_flags) {
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    uint32_t *tile_counts = reinterpret_cast<uint32_t *>(ws + l.off_tile_counts);
    uint64_t *tile_offsets = reinterpret_cast<uint64_t *>(ws + l.off_tile_offsets);
    uint64_t *nl_pos = reinterpret_cast<uint64_t *>(ws + l.off_nl_pos);
    uint64_t n_tiles = (n_bytes + kMpTileBytes - 1) / kMpTileBytes;
    uint32_t grid = (uint32_t)(n_tiles < 8192 ? (n_tiles ? n_tiles : 1) : 8192);
    hipLaunchKernelGGL(k_count_nl, dim3(grid), dim3(kMpThreads), 0, stream, d_in, n_bytes, lead, tile_counts, hdr, gate);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, stream, d_in, n_bytes, n_tiles, tile_counts,
                       tile_offsets, nl_pos, hdr, eof_mode, first_line_index, gate);
    if (n_tiles)
        hipLaunchKernelGGL(k_emit_nl, dim3(grid), dim3(kMpThreads), 0, stream, d_in, n_bytes, tile_offsets, nl_pos,
                           l.lines_cap, gate, line_flags);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

// ---- exg_count_newlines ------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_count_range(const uint8_t *__restrict__ d_in, uint64_t begin, uint64_t end,
                                                     unsigned long long *d_count) {
    // 16-byte aligned chunks covering [begin, end); lanes mask bytes outside the range
    uint64_t a0 = begin & ~15ull;
    uint64_t n_chunks = (end > a0) ? (end - a0 + 15) / 16 : 0;
    unsigned long long cnt = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the body of the range: four 16-byte loads in flight per lane (one load per trip reads at 5.5 TB/s, four at the
    // box's streaming rate), no masks — only the first and the last chunk of the range can reach outside it
    const uint64_t c_lo = begin > a0 ? 1 : 0, c_hi = (a0 + n_chunks * 16 > end) ? n_chunks - 1 : n_chunks;
    for (; c + 3 * stride < c_hi && c >= c_lo; c += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = ld_stream16(d_in + a0 + (c + k * stride) * 16);
#pragma unroll
        for (int k = 0; k < 4; k++) cnt += __popc(match16(v[k], 0x0A0A0A0Au));
    }
    for (; c < n_chunks; c += stride) {
        uint64_t off = a0 + c * 16;
        uint4 v = *reinterpret_cast<const uint4 *>(d_in + off);
        uint32_t m = match16(v, 0x0A0A0A0Au);
        if (off < begin) m &= ~((1u << (uint32_t)(begin - off)) - 1u);
        if (off + 16 > end) m &= (1u << (uint32_t)(end - off)) - 1u;
        cnt += __popc(m);
    }
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(d_count, cnt);
}

}  // namespace exg

extern "C" int exg_count_newlines(const void *d_input, uint64_t begin, uint64_t end, uint64_t *d_count,
                                  void *stream) {
    if (!d_count || (end > begin && !d_input) || ((uintptr_t)d_input & 15)) {
        exg::set_error("exg_count_newlines: bad arguments");
        return EXG_E_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    EXG_HIP_CHECK(hipMemsetAsync(d_count, 0, 8, s));
    if (end > begin) {
        uint64_t n_chunks = (end - (begin & ~15ull) + 15) / 16;
        uint64_t blocks = (n_chunks + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(exg::k_count_range, dim3((uint32_t)blocks), dim3(256), 0, s,
                           (const uint8_t *)d_input, begin, end, (unsigned long long *)d_count);
        EXG_HIP_CHECK(hipGetLastError());
    }
    return EXG_OK;
}

// ---- exg_fastq_guess_phase -------------------------------------------------------------------------
// Shard-local 4-line phase of FASTQ (multi-GPU byte-range shards start anywhere): the phase p of the
// first line that starts at or after `lead` must put '@' on every line k with (p + k) % 4 == 0 and
// '+' on every line with (p + k) % 4 == 2, over up to 32 lines.  Well-formed FASTQ leaves exactly one
// candidate ('@' can also open a quality line, which is why one line is not enough); the caller still
// VERIFIES it against the exchanged newline counts, so a wrong guess can cost a re-scan, never a result.
namespace exg {
__global__ void k_fastq_guess_phase(const uint8_t *__restrict__ d_in, uint64_t n_bytes, uint64_t lead, uint32_t *d_phase) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t pos = 0;
    if (lead > 0) {
        pos = lead - 1;
        while (pos < n_bytes && d_in[pos] != '\n') pos++;
        pos++;  // first line start at or after lead
    }
    uint32_t ok = 0xF;  // candidate phases still standing
    int lines = 0;
    const uint64_t limit = pos + 65536 < n_bytes ? pos + 65536 : n_bytes;
    while (lines < 32 && pos < limit) {
        uint32_t c = d_in[pos];
        for (uint32_t p = 0; p < 4; p++) {
            uint32_t ph = (p + (uint32_t)lines) & 3;
            if ((ph == 0 && c != '@') || (ph == 2 && c != '+')) ok &= ~(1u << p);
        }
        lines++;
        while (pos < limit && d_in[pos] != '\n') pos++;
        pos++;
    }
    uint32_t res = 0xFFFFFFFFu;
    if (lines >= 8 && ok && (ok & (ok - 1)) == 0) res = (uint32_t)__ffs(ok) - 1;
    *d_phase = res;
}
}  // namespace exg

extern "C" int exg_fastq_guess_phase(const void *d_input, uint64_t n_bytes, uint64_t lead, uint32_t *d_phase,
                                     void *stream) {
    if (!d_phase || (n_bytes && !d_input) || lead > n_bytes) {
        exg::set_error("exg_fastq_guess_phase: bad arguments");
        return EXG_E_INVALID_ARG;
    }
    hipLaunchKernelGGL(exg::k_fastq_guess_phase, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint8_t *)d_input,
                       n_bytes, lead, d_phase);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

// ---- exg_fasta_find_record --------------------------------------------------------------------------------------
// The first FASTA record start — a '>' at the beginning of a line — at an offset in [begin, end) of d_bytes: what a shard of
// a compressed FASTA needs at both of its ends (a record belongs to the shard in whose bytes its '>' line begins; in text
// files the host finds it with memchr over the mapping).  The byte in front of `begin` is looked at (begin = 0: at_bof says
// whether d_bytes[0] is the file's first byte, i.e. a line start).
namespace exg {
__global__ __launch_bounds__(256) void k_fasta_find_record(const uint8_t *__restrict__ d, uint64_t begin, uint64_t end, uint32_t at_bof,
                                                           unsigned long long *d_pos) {
    // a thread per 16 bytes (plain byte loads: any alignment); the lowest hit wins
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t p0 = begin + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; p0 < end; p0 += stride) {
        if (p0 >= *(volatile unsigned long long *)d_pos) return;  // someone in front of this thread has found one already
        const uint64_t p1 = p0 + 16 < end ? p0 + 16 : end;
        uint8_t prev = p0 ? d[p0 - 1] : (uint8_t)(at_bof ? '\n' : 0);
        for (uint64_t p = p0; p < p1; p++) {
            const uint8_t c = d[p];
            if (c == '>' && prev == '\n') {
                atomicMin(d_pos, (unsigned long long)p);
                return;
            }
            prev = c;
        }
    }
}
}  // namespace exg

extern "C" int exg_fasta_find_record(const void *d_bytes, uint64_t begin, uint64_t end, int at_bof, uint64_t *d_pos, void *stream) {
    if (!d_bytes || !d_pos || begin > end) {
        exg::set_error("exg_fasta_find_record: bad arguments");
        return EXG_E_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    EXG_HIP_CHECK(hipMemsetAsync(d_pos, 0xFF, 8, s));
    if (end > begin) {
        const uint64_t threads = (end - begin + 15) / 16;
        const uint64_t blocks = std::min<uint64_t>((threads + 255) / 256, 4096);
        hipLaunchKernelGGL(exg::k_fasta_find_record, dim3((uint32_t)blocks), dim3(256), 0, s, (const uint8_t *)d_bytes, begin, end,
                           at_bof ? 1u : 0u, (unsigned long long *)d_pos);
        EXG_HIP_CHECK(hipGetLastError());
    }
    return EXG_OK;
}
