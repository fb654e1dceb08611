// exg_crc32.hip — CRC-32 (IEEE 802.3, the gzip trailer's checksum, RFC 1952 2.3.1) of inflated bytes that are already in
// HBM.  The reference's decoders verify it (flate2 1.0.26 GzDecoder / MultiGzDecoder, noodles-bgzf: "corrupt gzip stream
// does not have a matching checksum", rust/Cargo.lock:1328-1329, behind rust/src/arrow_reader.rs:60-91); so does this.
//
// One wavefront per segment (a BGZF member's <= 64 KiB of output; longer outputs are cut into segments and the host
// combines the segments' checksums — arithmetic on 32-bit values, exg_crc32_combine).  The 64 lanes take 64 consecutive
// slices of L = ceil(len / 64) bytes, aligned to the END of the piece so that the bytes behind lane i's slice are exactly
// (63 - i) L: every lane runs the table-driven byte recurrence over its slice from state 0 (the lane that holds the piece's
// first byte from the running state), and six combine levels add the lanes up — a state followed by n bytes of anything is
// that state times x^(8n) mod P, one carry-less multiply-reduce (32 shift-xor steps), the multiplier squared per level.
#include <algorithm>
#include <stdlib.h>

#include "exg_common.hpp"

namespace exg {

static constexpr uint32_t kCrcPoly = 0xEDB88320u;

// a(x) b(x) mod P in the reflected representation (bit 31 = x^0), as in zlib's crc32.c multmodp
__host__ __device__ inline uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = b & 1 ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
// x^(n 2^k) mod P
__host__ __device__ inline uint32_t crc_x2nmodp(uint64_t n, uint32_t k) {
    uint32_t p = 1u << 31;  // x^0
    uint32_t sq = 1u << 30; // x^1
    for (uint32_t i = 0; i < k; i++) sq = crc_multmodp(sq, sq);  // x^(2^k)
    while (n) {
        if (n & 1) p = crc_multmodp(sq, p);
        n >>= 1;
        if (n) sq = crc_multmodp(sq, sq);
    }
    return p;
}

struct CrcSeg {
    uint64_t off;  // first byte in d_data
    uint64_t len;
};

// slicing-by-4 (Kounavis & Berry): tab[k][b] = the state after byte b followed by k zero bytes, so a dword costs four
// INDEPENDENT lookups instead of a chain of four; 16 bytes per load
__device__ __forceinline__ uint32_t crc_dword(const uint32_t (*tab)[256], uint32_t c, uint32_t w) {
    c ^= w;
    return tab[3][c & 0xFF] ^ tab[2][(c >> 8) & 0xFF] ^ tab[1][(c >> 16) & 0xFF] ^ tab[0][c >> 24];
}
__device__ __forceinline__ uint32_t crc_bytes(const uint32_t (*tab)[256], const uint8_t *p, uint64_t n, uint32_t c) {
    while (n && ((uintptr_t)p & 15)) {  // head to a 16-byte boundary
        c = tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
        n--;
    }
    for (; n >= 16; n -= 16, p += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        c = crc_dword(tab, c, v.x);
        c = crc_dword(tab, c, v.y);
        c = crc_dword(tab, c, v.z);
        c = crc_dword(tab, c, v.w);
    }
    while (n--) c = tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c;
}

// MODE 0: segs[] given.  MODE 1: segment i = the output of inflate member i (members[i].out_off, status[i].produced).
template <int MODE>
__global__ __launch_bounds__(64) void k_crc32(const uint8_t *__restrict__ data, const CrcSeg *__restrict__ segs, const exg_inflate_member *members,
                                              const exg_inflate_status *status, uint32_t n_segs, uint32_t *crc_out) {
    __shared__ uint32_t tab[4][256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t t = lane; t < 256; t += 64) {
        uint32_t c = t;
        for (int k = 0; k < 8; k++) c = c & 1 ? (c >> 1) ^ kCrcPoly : c >> 1;
        tab[0][t] = c;
    }
    __syncthreads();
    for (int k = 1; k < 4; k++) {
        for (uint32_t t = lane; t < 256; t += 64) {
            const uint32_t c = tab[k - 1][t];
            tab[k][t] = tab[0][c & 0xFF] ^ (c >> 8);
        }
        __syncthreads();
    }
    for (uint32_t sidx = blockIdx.x; sidx < n_segs; sidx += gridDim.x) {
        uint64_t off, len;
        if (MODE == 0) {
            off = segs[sidx].off, len = segs[sidx].len;
        } else {
            off = members[sidx].out_off, len = status[sidx].code ? 0 : status[sidx].produced;
        }
        uint32_t state = 0xFFFFFFFFu;  // wave-uniform running state
        for (uint64_t done = 0; done < len; done += 65536) {
            const uint32_t ln = (uint32_t)(len - done < 65536 ? len - done : 65536);
            const uint32_t L = (ln + 63) / 64;
            const long long lo = (long long)ln - (long long)(64 - lane) * L, hi = (long long)ln - (long long)(63 - lane) * L;
            uint32_t c = 0;
            if (hi > 0) {
                const long long a = lo > 0 ? lo : 0;
                c = crc_bytes(tab, data + off + done + a, (uint64_t)(hi - a), lo <= 0 ? state : 0u);
            }
            // lanes -> one state: level d folds lane i + d into lane i with the multiplier x^(8 d L)
            uint32_t M = crc_x2nmodp(L, 3);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t other = __shfl_down(c, d, 64);
                c = crc_multmodp(M, c) ^ other;  // (only lanes that are multiples of 2d keep a meaningful value)
                M = crc_multmodp(M, M);
            }
            state = __builtin_amdgcn_readfirstlane(c);
        }
        if (lane == 0) crc_out[sidx] = state ^ 0xFFFFFFFFu;
    }
}

}  // namespace exg

// CRC-32 of n_segs byte ranges of d_data (device): d_segs[i] = {offset, length}; d_crc[i] receives the finished checksum
// (init and final xor 0xFFFFFFFF applied).  Asynchronous.
extern "C" int exg_crc32_segments(const void *d_data, const exg_crc_segment *d_segs, uint32_t n_segs, uint32_t *d_crc, void *stream) {
    static_assert(sizeof(exg_crc_segment) == sizeof(exg::CrcSeg), "layout");
    if (!n_segs) return EXG_OK;
    if (!d_data || !d_segs || !d_crc) {
        exg::set_error("exg_crc32_segments: null argument");
        return EXG_E_INVALID_ARG;
    }
    hipLaunchKernelGGL((exg::k_crc32<0>), dim3(n_segs < 16384 ? n_segs : 16384), dim3(64), 0, (hipStream_t)stream, (const uint8_t *)d_data,
                       (const exg::CrcSeg *)d_segs, nullptr, nullptr, n_segs, d_crc);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}
// ... of the outputs of exg_inflate_members (d_members[i].out_off, d_status[i].produced bytes), in the same order
extern "C" int exg_crc32_members(const void *d_out, const exg_inflate_member *d_members, const exg_inflate_status *d_status, uint32_t n_members,
                                 uint32_t *d_crc, void *stream) {
    if (!n_members) return EXG_OK;
    if (!d_out || !d_members || !d_status || !d_crc) {
        exg::set_error("exg_crc32_members: null argument");
        return EXG_E_INVALID_ARG;
    }
    hipLaunchKernelGGL((exg::k_crc32<1>), dim3(n_members < 16384 ? n_members : 16384), dim3(64), 0, (hipStream_t)stream, (const uint8_t *)d_out,
                       nullptr, d_members, d_status, n_members, d_crc);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}
// A decoder's small results (member statuses + checksums, ~100 KB a window) written to PINNED HOST memory by a kernel, not by a
// copy: a hipMemcpyAsync device-to-host is a packet on an SDMA engine, and the engine this stream's copies go to may be the
// one that is sending 280 MB segment mirrors to the host (exg_rd_source.hpp: HostMirror) — the 100 KB then waited behind one
// or two of them, 10 ms a window on every third BGZF lane (round 5: config 4's all-columns drain at 0.60 s instead of 0.45 s).
// The stores leave over PCIe as they are made; they are visible to the host when an event recorded behind the kernel has
// completed (hipHostMalloc memory is coherent by default).  bytes: a multiple of 4; both pointers 4-byte aligned.
namespace exg {
__global__ void k_post_to_host(uint32_t *__restrict__ h_dst, const uint32_t *__restrict__ d_src, uint32_t n_words) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) h_dst[i] = d_src[i];
}
int post_to_host(void *h_dst, const void *d_src, uint64_t bytes, void *stream) {
    if (!bytes) return EXG_OK;
    if (!h_dst || !d_src || (bytes & 3) || (((uintptr_t)h_dst | (uintptr_t)d_src) & 3) || bytes > (1ull << 32)) {
        set_error("post_to_host: null, unaligned or oversized argument");
        return EXG_E_INVALID_ARG;
    }
    // (A/B: the copy this replaced.  Also under HIP_HOST_COHERENT=0: pinned memory is then not coherent, and the kernel's stores
    // are only guaranteed visible at an event with hipEventReleaseToSystem — which the callers' events are not)
    static const bool by_copy = getenv("EXG_POST_BY_COPY") != nullptr || (getenv("HIP_HOST_COHERENT") && atoi(getenv("HIP_HOST_COHERENT")) == 0);
    if (by_copy) {
        EXG_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        return EXG_OK;
    }
    const uint32_t n_words = (uint32_t)(bytes / 4), blocks = (n_words + 255) / 256;
    hipLaunchKernelGGL(k_post_to_host, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, (hipStream_t)stream, (uint32_t *)h_dst,
                       (const uint32_t *)d_src, n_words);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}
// Bulk bytes to pinned host memory by a kernel's 16-byte stores (no copy engine: what a concurrent upload's slices cannot be queued
// behind).  Both pointers 16-byte aligned; `bytes` is rounded up to 16 (the caller's blocks have the room).  64 workgroups: PCIe is the
// bound (50-55 GB/s against hipMemcpyAsync's 57 with 32 to 2 048 of them), and kernels that run beside the copy slow down less the fewer
// wavefronts sit on stores the link has not taken yet (the Arrow emitter's column kernels: FASTQ at that boundary 110 -> 107 ms)
typedef uint32_t stream_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_to_host(stream_v4 *__restrict__ h_dst, const stream_v4 *__restrict__ d_src, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&d_src[i]), &h_dst[i]);
}
int stream_to_host(void *h_dst, const void *d_src, uint64_t bytes, void *stream) {
    if (!bytes) return EXG_OK;
    if (!h_dst || !d_src || (((uintptr_t)h_dst | (uintptr_t)d_src) & 15)) {
        set_error("stream_to_host: null or unaligned argument");
        return EXG_E_INVALID_ARG;
    }
    static const bool by_copy = getenv("EXG_POST_BY_COPY") != nullptr || (getenv("HIP_HOST_COHERENT") && atoi(getenv("HIP_HOST_COHERENT")) == 0);
    if (by_copy) {
        EXG_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        return EXG_OK;
    }
    static const uint32_t blocks = getenv("EXG_STREAM_TO_HOST_BLOCKS") ? (uint32_t)std::max(1, atoi(getenv("EXG_STREAM_TO_HOST_BLOCKS"))) : 64u;
    const uint64_t n16 = (bytes + 15) / 16;
    hipLaunchKernelGGL(k_stream_to_host, dim3((uint32_t)std::min<uint64_t>(blocks, (n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (stream_v4 *)h_dst,
                       (const stream_v4 *)d_src, n16);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}
}  // namespace exg
// Host: the checksum of A followed by B from crc(A), crc(B) and len(B) (zlib's crc32_combine)
extern "C" uint32_t exg_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return exg::crc_multmodp(exg::crc_x2nmodp(len_b, 3), crc_a) ^ crc_b;
}
