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
    unsigned int any_far;            // fused kernels: some half left a record to k_*_far (plain store of 1)
    unsigned int any_redo;           // lean scan: some super-tile is marked (tile_redo) for the any-shape run behind it
    unsigned int n_redo, redo_pad;   // super-tiles the any-shape run redid (one atomicAdd per workgroup of that run)
    unsigned long long reserved[18];
};
struct SlowLiteral {
    unsigned long long off;  // input offset of the literal
    unsigned int len;
    unsigned int row;        // output row whose QUAL it is
};
static_assert(sizeof(ScanWsHeader) == 256, "header is 256 bytes");

// A record (FASTQ) / line (VCF) that ends in a 16 KiB half but begins before the 1 KiB window in front of it — a long read, a
// multi-sample VCF line — has its delimiting newlines outside LDS.  The fused kernel does not give the launch up for it: it
// writes what it knows (one FarRec per half: at most ONE such record ends in a half, the first) and k_*_far, a small kernel
// behind it on the stream, emits these rows from global memory.  pos[k]: the record's delimiting newlines, oldest first
// (FASTQ: 5, VCF: 2); >= 0: offset inside the half's super-tile; -1 - j: the j-th newest newline in front of the super-tile,
// found by looking back over the tiles' counts (tileA) and last-four lists (tileL).
struct FarRec {
    int32_t pos[5];
    uint32_t flags;      // bit 0: the half holds the end of the input with EXG_F_EOF (virtual EOF lines sit at n_bytes)
    long long out;       // output row
};
static_assert(sizeof(FarRec) == 32, "FarRec is 32 bytes");
static constexpr unsigned long long kFarBit = 1ull << 63;    // in tile_qend[half]: far_rec[half] is to be emitted

struct FastqWsLayout {
    uint64_t n_tiles_mp;
    uint64_t n_tiles_fused;
    uint64_t off_tile_counts;   // u32[n_tiles_mp]
    uint64_t off_tile_offsets;  // u64[n_tiles_mp]
    // one block of 80 n_tiles_fused bytes: u64 tileA[n] (u32 counts) | u64 tileP[n] | u64 tile_redo[n] (u32 marks; these three
    // are zeroed per launch) | u64 tile_qend[n] | int32 tileL[n][4]: the last four newlines of every super-tile (codes like
    // FarRec::pos) | FarRec[n].  n is a multiple of 4 and a function of n_bytes alone (fused_n_tiles), so that the kernels
    // find tile_redo, tileL and FarRec[] from tile_qend without another argument or load (two more kernel arguments cost the
    // FASTQ scan scalar registers it does not have: spills)
    uint64_t off_tile_desc;
    uint64_t off_tile_last4;
    uint64_t off_far;
    uint64_t off_block_sums;    // 2 x u64[lines / 4096 + 2] (FASTA device-wide scans: records, payload)
    uint64_t off_slow;          // SlowLiteral[slow_cap] (VCF: QUAL literals for the exact parser)
    uint64_t slow_cap;          // one per 32 bytes of input, at most 4096: a literal of that class has more than 19 digits
    uint64_t off_nl_pos;        // u64[lines_cap]
    uint64_t lines_cap;
    uint64_t total_bytes;
};

static inline uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
// entries of the fused kernels' per-tile arrays for a buffer of n_bytes (host and device agree on it)
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint64_t fused_n_tiles(uint64_t n_bytes) { return ((n_bytes + kFusedTileBytes - 1) / kFusedTileBytes + 2 + 3) & ~3ull; }

// n_line_arrays: FASTA keeps 4 u64 arrays per line (offsets, record prefix, payload prefix, record starts).
// nl_pos capacity: the general kernels index every line.  Worst case is one line per byte; the
// default provisions one line per 8 bytes (FASTQ/VCF lines are far longer) but never less than
// min(n, 1 Mi) + 8 so small inputs are always safe.  Overflow is reported, never silent.
static inline FastqWsLayout fastq_ws_layout(uint64_t n_bytes, uint64_t ws_bytes_or_0, uint64_t n_line_arrays = 1) {
    FastqWsLayout l;
    l.n_tiles_mp = (n_bytes + kMpTileBytes - 1) / kMpTileBytes + 1;
    l.n_tiles_fused = fused_n_tiles(n_bytes);
    uint64_t at = sizeof(ScanWsHeader);
    l.off_tile_counts = at;
    at = round_up(at + l.n_tiles_mp * 4, 256);
    l.off_tile_offsets = at;
    at = round_up(at + l.n_tiles_mp * 8, 256);
    l.off_tile_desc = at;
    l.off_tile_last4 = at + l.n_tiles_fused * 32;
    l.off_far = at + l.n_tiles_fused * 48;
    at = round_up(at + l.n_tiles_fused * 80, 256);
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
