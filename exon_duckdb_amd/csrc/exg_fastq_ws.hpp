// exg_fastq_ws.hpp — device workspace layout shared by the FASTQ kernels.
#pragma once
#include "exg_common.hpp"

namespace exg {

// Multipass tiles: 256 threads x 16 B x 4 iterations.
static constexpr uint32_t kMpThreads = 256;
static constexpr uint32_t kMpIterBytes = kMpThreads * 16;  // 4 KiB
static constexpr uint32_t kMpIters = 4;
static constexpr uint32_t kMpTileBytes = kMpIterBytes * kMpIters;  // 16 KiB

// Fused kernel tile (see exg_fastq_fused.hip)
static constexpr uint32_t kFusedTileBytes = 16384;
static constexpr uint32_t kFusedWindow = 1024;  // bytes before the tile staged in LDS for straddling records

struct alignas(256) ScanWsHeader {
    unsigned long long total_nl;     // real '\n' in [0, n_bytes)
    unsigned long long total_lines;  // + virtual lines appended at EOF
    unsigned long long halo_nl;      // '\n' at offsets < lead
    unsigned long long err_word;     // (output record index << 8) | EXG_PE_*, atomicMin; ~0 = none
    unsigned long long n_unresolved; // owned records whose first line starts before d_input[0]
    unsigned int flags;              // EXG_RF_*
    unsigned int ticket;             // fused kernel: dynamic tile counter
    unsigned int epoch;              // fused kernel: tag of the current launch in the tile descriptors
    unsigned int overflow;           // fused kernel: a record did not fit the LDS window
    unsigned long long lines_cap;    // capacity of nl_pos
    unsigned long long err_off;      // atomicMin: start offset of a failing record; ~0 = none
    unsigned long long consumed;     // atomicMax: offset just past the last owned quality line
    unsigned long long last_qend;    // atomicMax: offset just past the last quality line, owned or not
    // VCF: QUAL literals the scan kernel could not decide with 19 digits (exg_parse.hpp, status 2) are listed in the
    // workspace (FastqWsLayout::off_slow) and decided exactly by the finalize kernel (exg_float_slow.hpp).
    unsigned int n_slow, slow_pad;
    unsigned long long reserved[20];
};
struct SlowLiteral {
    unsigned long long off;  // input offset of the literal
    unsigned int len;
    unsigned int row;        // output row whose QUAL it is
};
static_assert(sizeof(ScanWsHeader) == 256, "header is 256 bytes");

struct FastqWsLayout {
    uint64_t n_tiles_mp;
    uint64_t n_tiles_fused;
    uint64_t off_tile_counts;   // u32[n_tiles_mp]
    uint64_t off_tile_offsets;  // u64[n_tiles_mp]
    uint64_t off_tile_desc;     // u64[n_tiles_fused] look-back descriptors + u64[n_tiles_fused] tile_qend
    uint64_t off_block_sums;    // 2 x u64[lines / 4096 + 2] (FASTA device-wide scans: records, payload)
    uint64_t off_slow;          // SlowLiteral[slow_cap] (VCF: QUAL literals for the exact parser)
    uint64_t slow_cap;          // one per 32 bytes of input, at most 4096: a literal of that class has more than 19 digits
    uint64_t off_nl_pos;        // u64[lines_cap]
    uint64_t lines_cap;
    uint64_t total_bytes;
};

static inline uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// n_line_arrays: FASTA keeps 4 u64 arrays per line (offsets, record prefix, payload prefix, record starts).
// nl_pos capacity: the general kernels index every line.  Worst case is one line per byte; the
// default provisions one line per 8 bytes (FASTQ/VCF lines are far longer) but never less than
// min(n, 1 Mi) + 8 so small inputs are always safe.  Overflow is reported, never silent.
static inline FastqWsLayout fastq_ws_layout(uint64_t n_bytes, uint64_t ws_bytes_or_0, uint64_t n_line_arrays = 1) {
    FastqWsLayout l;
    l.n_tiles_mp = (n_bytes + kMpTileBytes - 1) / kMpTileBytes + 1;
    l.n_tiles_fused = (n_bytes + kFusedTileBytes - 1) / kFusedTileBytes + 2;
    uint64_t at = sizeof(ScanWsHeader);
    l.off_tile_counts = at;
    at = round_up(at + l.n_tiles_mp * 4, 256);
    l.off_tile_offsets = at;
    at = round_up(at + l.n_tiles_mp * 8, 256);
    l.off_tile_desc = at;
    at = round_up(at + l.n_tiles_fused * 24, 256);  // tileA, tileP, tile_qend
    uint64_t want = n_bytes / 8;
    uint64_t small = n_bytes < (1ull << 20) ? n_bytes : (1ull << 20);
    if (want < small) want = small;
    want += 8;
    l.off_block_sums = at;
    at = round_up(at + ((n_bytes + 16) / 4096 + 4) * 16, 256);  // 2 sums per block; lines <= bytes + 1, whatever the workspace size
    l.slow_cap = n_bytes / 32 + 16 < 4096 ? n_bytes / 32 + 16 : 4096;
    l.off_slow = at;
    at = round_up(at + l.slow_cap * sizeof(SlowLiteral), 256);
    l.off_nl_pos = at;
    if (ws_bytes_or_0) {
        uint64_t avail = ws_bytes_or_0 > at ? (ws_bytes_or_0 - at) / (8 * n_line_arrays) : 0;
        l.lines_cap = avail > 2 ? avail - 2 : 0;  // each per-line array holds lines_cap + 2 entries
    } else {
        l.lines_cap = want;
    }
    l.total_bytes = at + (l.lines_cap + 2) * 8 * n_line_arrays;
    return l;
}

}  // namespace exg
