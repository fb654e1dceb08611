// exg_synth.hip — deterministic synthetic FASTQ generated in HBM (bench / tests input only).
// Byte-for-byte the generator SURVEY.md §8 D2 specifies; tests compare it with the oracle's.
#include "exg_common.hpp"

namespace exg {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t synth_word(uint64_t seed, uint64_t k, uint64_t j) {
    return splitmix64((seed ^ (k * 0x9E3779B97F4A7C15ull)) + j);
}

__device__ uint32_t synth_fastq_byte(uint64_t seed, uint64_t off) {
    uint64_t k = off / 332, w = off % 332;
    if (w < 28) {
        if (w < 4) return (uint32_t) "@SYN"[w];
        if (w < 16) {
            uint64_t v = k % 1000000000000ull;
            for (uint64_t i = 15; i > w; i--) v /= 10;
            return (uint32_t)('0' + v % 10);
        }
        if (w == 16) return ' ';
        if (w == 17) return (uint32_t)('0' + k % 4);
        if (w < 27) return (uint32_t) ":N:0:ACGT"[w - 18];
        return '\n';
    }
    if (w < 178) {
        uint64_t i = w - 28;
        return (uint32_t) "ACGT"[(synth_word(seed, k, i / 32) >> (2 * (i % 32))) & 3];
    }
    if (w == 178) return '\n';
    if (w == 179) return '+';
    if (w == 180) return '\n';
    if (w < 331) {
        uint64_t i = w - 181;
        uint64_t b = (synth_word(seed, k, 8 + i / 8) >> (8 * (i % 8))) & 0xFF;
        return (uint32_t)('!' + ((b * 41) >> 8));
    }
    return '\n';
}

// one thread per 16 output bytes
__global__ __launch_bounds__(256) void k_synth_fastq(uint8_t *__restrict__ out, uint64_t file_offset, uint64_t n_bytes,
                                                     uint64_t seed) {
    uint64_t n_chunks = (n_bytes + 15) / 16;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks;
         c += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w[4] = {0, 0, 0, 0};
        uint64_t base = c * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint64_t o = base + i;
            uint32_t b = o < n_bytes ? synth_fastq_byte(seed, file_offset + o) : 0u;
            w[i >> 2] |= b << (8 * (i & 3));
        }
        *reinterpret_cast<uint4 *>(out + base) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

}  // namespace exg

extern "C" int exg_synth_fastq(void *d_out, uint64_t file_offset, uint64_t n_bytes, uint64_t seed, void *stream) {
    if (!d_out || ((uintptr_t)d_out & 15)) {
        exg::set_error("exg_synth_fastq: output must be a 16-byte aligned device pointer");
        return EXG_E_INVALID_ARG;
    }
    if (!n_bytes) return EXG_OK;
    uint64_t n_chunks = (n_bytes + 15) / 16;
    uint64_t blocks = (n_chunks + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(exg::k_synth_fastq, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, (uint8_t *)d_out,
                       file_offset, n_bytes, seed);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}
