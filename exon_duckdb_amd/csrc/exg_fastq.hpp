// exg_fastq.hpp — FASTQ scan: device argument block + the two implementations.
#pragma once
#include "exg_fastq_ws.hpp"

namespace exg {

// By-value kernel argument (what the kernels need from exg_fastq_scan_args).
struct FastqDev {
    const uint8_t *d_in;
    uint64_t n_bytes;
    uint64_t lead;
    uint64_t first_line_index;
    uint64_t payload_base;
    uint32_t flags;
    uint32_t pad;
    exg_string_t *d_name, *d_desc, *d_seq, *d_qual;
    uint64_t *d_desc_valid;
    uint64_t capacity;
};

#if defined(__HIPCC__)
// A record's lines from global memory (the general path's kernels and k_fastq_far).
struct FastqGeom {
    uint64_t s[4], e[4];  // field lines, CR stripped: name line, sequence, plus line, quality
    uint64_t raw_e[4];
    uint64_t name_e, desc_s;  // split of the name line at the first ' '
    bool name_ok, plus_ok;
    bool resolved;
};
// p[0 .. 4]: the five newlines that delimit the record (p[0] = -1: its name line begins at d_input[0]; p[k] >= n_bytes: a
// virtual line at the end of the input, empty)
__device__ __forceinline__ FastqGeom fastq_geometry_at(const uint8_t *__restrict__ d_in, uint64_t n_bytes, const int64_t *p) {
    FastqGeom g;
    g.resolved = true;
    uint64_t start = (uint64_t)(p[0] + 1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t raw_end = (uint64_t)p[k + 1];
        if (start > raw_end) start = raw_end;  // virtual (EOF) lines are empty
        uint64_t end = raw_end;
        bool virt = raw_end >= n_bytes;
        if (!virt && end > start && d_in[end - 1] == '\r') end--;
        g.s[k] = start;
        g.e[k] = end;
        g.raw_e[k] = raw_end;
        start = raw_end + 1;
    }
    g.name_ok = g.s[0] < g.raw_e[0] && d_in[g.s[0]] == '@';
    g.plus_ok = g.s[2] < g.raw_e[2] && d_in[g.s[2]] == '+';
    // name = [s0+1, first ' '), description = (first ' ', e0)
    uint64_t q0 = g.s[0] + 1;
    if (q0 > g.e[0]) q0 = g.e[0];
    uint64_t q = q0;
    while (q < g.e[0] && d_in[q] != ' ') q++;
    g.s[0] = q0;
    g.name_e = q;
    g.desc_s = q < g.e[0] ? q + 1 : g.e[0];
    return g;
}
#endif

// General path: line index + per-record field extraction from global memory (4 passes).
int run_fastq_multipass(const exg_fastq_scan_args *args, const FastqDev &dev, uint8_t *ws, const FastqWsLayout &l,
                        hipStream_t stream, bool after_fused);

// Fast path: single pass.  full = false: the lean scan, then the any-shape scan over the super-tiles it marked; true: the
// any-shape scan alone.
int run_fastq_fused(const exg_fastq_scan_args *args, const FastqDev &dev, uint8_t *ws, const FastqWsLayout &l,
                    hipStream_t stream, bool full);

__global__ void k_init_hdr(ScanWsHeader *hdr, uint64_t lines_cap, uint32_t mode);

}  // namespace exg
