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

// General path: line index + per-record field extraction from global memory (4 passes).
int run_fastq_multipass(const exg_fastq_scan_args *args, const FastqDev &dev, uint8_t *ws, const FastqWsLayout &l,
                        hipStream_t stream, bool after_fused);

// Fast path: single pass, decoupled look-back, LDS-staged tiles.
int run_fastq_fused(const exg_fastq_scan_args *args, const FastqDev &dev, uint8_t *ws, const FastqWsLayout &l,
                    hipStream_t stream);

__global__ void k_init_hdr(ScanWsHeader *hdr, uint64_t lines_cap, uint32_t mode);

}  // namespace exg
