// exg_scan.hpp — device-wide exclusive prefix sum of f(0..n) into u64 (three launches: block-local scan,
// scan of the block totals, add).  Used by the Arrow emitters (offsets of strings, lists, row maps);
// n is a host value, f a small functor passed by value.  out has n + 1 entries, out[n] = total.
#pragma once
#include "exg_common.hpp"

namespace exg {

static constexpr uint32_t kXScanChunk = 4096;  // 1024 threads x 4

inline uint64_t xscan_blocks(uint64_t n) { return (n + kXScanChunk - 1) / kXScanChunk; }
// u64 entries the caller provides for the block totals
inline uint64_t xscan_tmp_entries(uint64_t n) { return xscan_blocks(n) + 2; }

template <class F>
__global__ __launch_bounds__(1024) void k_xscan_local(F f, uint64_t n, uint64_t *out, uint64_t *bsum) {
    __shared__ unsigned long long s_w[16];
    const uint64_t base = (uint64_t)blockIdx.x * kXScanChunk;
    uint64_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        v[k] = idx < n ? (uint64_t)f(idx) : 0;
        sum += v[k];
    }
    unsigned long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long o = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long off = 0, tot = 0;
    for (uint32_t k = 0; k < 16; k++) {
        if (k < (threadIdx.x >> 6)) off += s_w[k];
        tot += s_w[k];
    }
    uint64_t run = off + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        if (idx < n) out[idx] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// exclusive scan of bsum[0..nb) in place, bsum[nb] = grand total
static __global__ __launch_bounds__(1024) void k_xscan_blocks(uint64_t *bsum, uint64_t nb) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nb; base += 1024) {
        uint64_t idx = base + threadIdx.x;
        unsigned long long c = idx < nb ? bsum[idx] : 0, incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long o = __shfl_up(incl, d, 64);
            if ((int)(threadIdx.x & 63) >= d) incl += o;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long off = 0;
        for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) off += s_w[k];
        unsigned long long run = s_run;
        if (idx < nb) bsum[idx] = run + off + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = run + off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nb] = s_run;
}

static __global__ __launch_bounds__(1024) void k_xscan_add(uint64_t *out, uint64_t n, const uint64_t *bsum, uint64_t nb) {
    const uint64_t base = (uint64_t)blockIdx.x * kXScanChunk;
    const uint64_t add = bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        if (idx < n) out[idx] += add;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = bsum[nb];
}

template <class F>
inline void launch_xscan(F f, uint64_t n, uint64_t *d_out, uint64_t *d_tmp, hipStream_t stream) {
    const uint64_t nb = xscan_blocks(n);
    if (nb) hipLaunchKernelGGL(k_xscan_local<F>, dim3((uint32_t)nb), dim3(1024), 0, stream, f, n, d_out, d_tmp);
    hipLaunchKernelGGL(k_xscan_blocks, dim3(1), dim3(1024), 0, stream, d_tmp, nb);
    hipLaunchKernelGGL(k_xscan_add, dim3((uint32_t)(nb ? nb : 1)), dim3(1024), 0, stream, d_out, n, d_tmp, nb);
}

}  // namespace exg
