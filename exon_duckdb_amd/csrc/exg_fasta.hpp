// exg_fasta.hpp — what the two FASTA implementations (exg_fasta.hip: multipass over a line index;
// exg_fasta_tiled.hip: one pass over 32 KiB super-tiles) share.
#pragma once
#include "exg_fastq_ws.hpp"

namespace exg {

struct FastaDev {
    const uint8_t *d_in;
    uint64_t n_bytes;
    uint64_t payload_base;
    uint64_t seq_payload_base;
    uint32_t flags;
    uint32_t pad;
    exg_string_t *d_id, *d_desc, *d_seq;
    uint64_t *d_desc_valid;
    uint8_t *d_payload;
    uint64_t capacity;
};


#if defined(__HIPCC__)
__device__ __forceinline__ bool is_ascii_ws(uint32_t b) { return b == ' ' || b == '\t' || b == '\n' || b == '\f' || b == '\r'; }

// length of a Unicode White_Space scalar starting at p[i] (str::trim), 0 if none
__device__ inline int ws_len_fwd(const uint8_t *p, uint64_t i, uint64_t e) {
    if (i >= e) return 0;
    uint32_t b = p[i];
    if ((b >= 0x09 && b <= 0x0D) || b == 0x20) return 1;
    if (i + 1 < e && b == 0xC2 && (p[i + 1] == 0x85 || p[i + 1] == 0xA0)) return 2;
    if (i + 2 < e) {
        uint32_t c1 = p[i + 1], c2 = p[i + 2];
        if (b == 0xE1 && c1 == 0x9A && c2 == 0x80) return 3;
        if (b == 0xE2 && c1 == 0x80 && ((c2 >= 0x80 && c2 <= 0x8A) || c2 == 0xA8 || c2 == 0xA9 || c2 == 0xAF)) return 3;
        if (b == 0xE2 && c1 == 0x81 && c2 == 0x9F) return 3;
        if (b == 0xE3 && c1 == 0x80 && c2 == 0x80) return 3;
    }
    return 0;
}
__device__ inline int ws_len_bwd(const uint8_t *p, uint64_t s, uint64_t e) {
    for (int l = 1; l <= 3; l++)
        if (e - s >= (uint64_t)l && ws_len_fwd(p, e - l, e) == l) return l;
    return 0;
}

#endif

// one pass over 32 KiB super-tiles (exg_fasta_tiled.hip); the multipass form stays as its differential partner
int run_fasta_tiled(const FastaDev &dev, uint8_t *ws, const FastqWsLayout &l, exg_scan_result *d_result, hipStream_t stream);

}  // namespace exg
