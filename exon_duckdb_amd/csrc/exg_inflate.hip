// exg_inflate.hip — DEFLATE (RFC 1951) on the device, one wavefront per gzip member.
//
// Replaces the decompression the reference gets from DataFusion 28 `FileCompressionType::convert_stream`
// -> async-compression 0.4.0 -> flate2 1.0.26 (and noodles-bgzf for .vcf.gz), selected at
// rust/src/arrow_reader.rs:60-91.  The host parses the gzip member framing (RFC 1952; BGZF members
// carry their compressed size in the 'BC' extra subfield, so they are found without decoding and are
// inflated in parallel); the inflated bytes stay in HBM and feed the scan kernels directly.
//
// Kernel shape: block = 1 wave (64 lanes); the decoder body is exg_inflate_core.hpp.  Block headers
// are decoded by all lanes uniformly (bit position and output position live in SGPRs); inside a Huffman block the lanes
// decode speculatively the tokens that would start at the next 4 x 64 bit offsets, one scalar walk marks the real ones,
// a prefix sum places them:
//   * compressed bytes are staged through a 512-byte LDS ring (two 256-byte chunks, the next chunk's load in flight);
//   * Huffman tables (10-bit / 9-bit primary LUTs in LDS + canonical count / sorted-symbol arrays for the rare longer
//     codes) are built cooperatively: canonical codes by wave ballots, LUT fill per symbol;
//   * the newest 2 Ki bytes of the window are an LDS ring, older bytes are read back from the output the wave itself
//     flushed to HBM (every completed 1 KiB, 16 B/lane stores); matches are copied by all lanes (source index folded
//     modulo the distance, so overlapping copies are exact).
// LDS: 5112 B per member (four granules), 64 VGPRs => 8 waves per SIMD, 32 members per CU, 8192 in flight.
#include <stdlib.h>

#include "exg_inflate_core.hpp"

namespace exg {
template <uint32_t RING, int EMIT>
#ifndef EXG_INFLATE_WAVES
#define EXG_INFLATE_WAVES 8
#endif
__global__ __launch_bounds__(64, RING <= 2048 ? EXG_INFLATE_WAVES : 1) void k_inflate(const uint8_t *__restrict__ d_comp, uint8_t *d_out,
                                                const InflateMember *__restrict__ members, InflateStatus *status,
                                                uint32_t n_members) {
    __shared__ __attribute__((aligned(16))) InflateLdsT<false, RING> s;
    InflateJobStatus s_st;  // written and read by lane 0 only (registers: LDS decides how many members a CU decodes at once)
    for (uint32_t m = blockIdx.x; m < n_members; m += gridDim.x) {
        const InflateMember mb = members[m];
        InflateJob jb;
        jb.comp_off = mb.comp_off;
        jb.comp_size = mb.comp_size;
        jb.out_off = mb.out_off;
        jb.out_cap = mb.out_cap;
        jb.start_bit = 0;
        jb.stop_bit = 0;
        jb.text_probe = 0;
        jb.pad = 0;
        inflate_job<false, RING, EMIT>(s, d_comp, d_out, jb, &s_st);
        if (threadIdx.x == 0) {
            InflateStatus st;
            st.code = s_st.code;
            st.pad = 0;
            st.produced = s_st.produced;
            // consumed: bytes up to the byte boundary after the final block, relative to comp_off
            st.consumed = (s_st.end_bit + 7) >> 3;
            status[m] = st;
        }
        __syncthreads();
    }
}

}  // namespace exg

// ---- C-ABI ----------------------------------------------------------------------------------------------
// members / status are device arrays of exg_inflate_member / exg_inflate_status (same layout as above).
extern "C" int exg_inflate_members(const void *d_comp, void *d_out, const exg_inflate_member *d_members,
                                   exg_inflate_status *d_status, uint32_t n_members, void *stream) {
    static_assert(sizeof(exg_inflate_member) == sizeof(exg::InflateMember), "layout");
    static_assert(sizeof(exg_inflate_status) == sizeof(exg::InflateStatus), "layout");
    if (!n_members) return EXG_OK;
    if (!d_comp || !d_out || !d_members || !d_status || ((uintptr_t)d_comp & 15)) {
        exg::set_error("exg_inflate_members: bad arguments (null or unaligned compressed buffer)");
        return EXG_E_INVALID_ARG;
    }
    uint32_t grid = n_members < 8192 ? n_members : 8192;
#define EXG_LAUNCH_INFLATE(R, E, OUT)                                                                                  \
    hipLaunchKernelGGL((exg::k_inflate<R, E>), dim3(grid), dim3(64), dyn_lds, (hipStream_t)stream, (const uint8_t *)d_comp,  \
                       (uint8_t *)(OUT), (const exg::InflateMember *)d_members, (exg::InflateStatus *)d_status, n_members)
    unsigned dyn_lds = 0;
#ifdef EXG_DEV_PROBE
    dyn_lds = getenv("EXG_INFLATE_DYNLDS") ? atoi(getenv("EXG_INFLATE_DYNLDS")) : 0;  // (occupancy experiments)
    // development builds only (tools/ab_inflate.sh): the ring size, the first form of the emit step, a decode that keeps nothing
    static const int ring = getenv("EXG_INFLATE_RING") ? atoi(getenv("EXG_INFLATE_RING")) : 2048;
    static const int emit = getenv("EXG_INFLATE_EMIT") ? atoi(getenv("EXG_INFLATE_EMIT")) : EXG_INFLATE_EMIT;
    void *out = getenv("EXG_INFLATE_NOOUT") ? nullptr : d_out;
    if (emit == 0) {
        EXG_LAUNCH_INFLATE(2048, 0, out);
    } else if (emit == 1) {
        EXG_LAUNCH_INFLATE(2048, 1, out);
    } else if (emit == 2) {
        EXG_LAUNCH_INFLATE(2048, 2, out);
    } else {
        switch (ring) {
            case 4096: EXG_LAUNCH_INFLATE(4096, 4, out); break;
            case 32768: EXG_LAUNCH_INFLATE(32768, 4, out); break;
            default: EXG_LAUNCH_INFLATE(2048, 4, out); break;
        }
    }
#else
    EXG_LAUNCH_INFLATE(2048, EXG_INFLATE_EMIT, d_out);
#endif
#undef EXG_LAUNCH_INFLATE
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}
