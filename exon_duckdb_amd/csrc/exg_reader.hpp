// exg_reader.hpp — internals of the reader level shared by exg_reader.cpp (DuckDB-shaped chunks) and
// exg_arrow_stream.cpp (the reference's new_reader: Arrow record batches).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "exg_common.hpp"

namespace exg_rd {

struct PinnedBlock {  // host memory: pinned (hipHostMalloc) or a read-only file mapping
    void *p = nullptr;
    size_t n = 0;
    size_t mapped = 0;  // != 0: p is an mmap of that many bytes
    ~PinnedBlock();
};

struct Batch {  // host vectors of one device batch, shared by its chunks
    std::shared_ptr<PinnedBlock> file;
    int n_cols = 0;
    PinnedBlock cols[9];
    uint32_t elem[9] = {16, 16, 16, 16, 16, 16, 16, 16, 16};  // bytes per row
    PinnedBlock validity[9];                                     // empty => all rows valid
    PinnedBlock payload;                                         // FASTA: compacted sequences
    uint64_t n_rows = 0;
};

struct ChunkKeep {
    std::shared_ptr<Batch> batch;
};

enum Compression { kNone, kGzip, kZstd, kBzip2, kXz };

// what the scan of one device batch left in HBM (valid until the next batch is scanned)
struct ScanCtx {
    const void *d_input;    // device address of the scanned bytes
    const uint8_t *h;       // host address the string_t pointers are relative to (payload_base)
    uint64_t n_records;
    exg_scan_result res;
    const uint8_t *h_seq_payload;  // FASTA: payload_base of the sequence column (device bytes: d_payload)
};

}  // namespace exg_rd

struct exg_reader {
    int format = 0;
    exg_rd::Compression compression = exg_rd::kNone;
    std::vector<std::string> files;
    size_t file_idx = 0;
    uint64_t batch_rows = EXG_VECTOR_SIZE;
    uint64_t device_batch_bytes = 256ull << 20;
    int device = 0;
    std::string error;
    hipStream_t stream = nullptr;

    // current file
    std::shared_ptr<exg_rd::PinnedBlock> file;
    uint64_t file_pos = 0;  // first byte not yet consumed by a complete record
    bool file_done = true;

    // device buffers (sized for device_batch_bytes)
    void *d_in = nullptr, *d_ws = nullptr, *d_res = nullptr;
    void *d_valid[2] = {nullptr, nullptr};
    void *d_cols[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *d_pos = nullptr, *d_qual = nullptr, *d_payload = nullptr;
    uint64_t d_in_cap = 0, ws_bytes = 0, cap_records = 0;
    uint64_t vcf_header_bytes = 0;
    struct FdCloser {
        int fd;
        ~FdCloser();
    };
    std::unique_ptr<FdCloser> fd_keep;  // current file (pread source of the bounce buffer)
    exg_rd::PinnedBlock staging;     // pinned bounce buffer for H2D (the file itself is only mapped)
    void *d_file = nullptr;  // gzip input: the inflated bytes live here and are scanned in place
    uint64_t d_file_bytes = 0;

    // current batch
    std::shared_ptr<exg_rd::Batch> batch;
    uint64_t batch_row = 0;
    uint32_t pending_error = 0;  // parse error to raise once the rows before it have been handed out
    uint64_t pending_error_offset = 0;

    // Arrow mode (new_reader): the columns stay on the device and `arrow_emit` turns them into Arrow buffers
    int (*arrow_emit)(exg_reader *, const exg_rd::ScanCtx &) = nullptr;
    std::shared_ptr<void> arrow_state;

    void free_device();
    ~exg_reader();
};

namespace exg_rd {
int fail(exg_reader *r, int code, const std::string &msg);
int open_next_file(exg_reader *r);
// Scan the next device batch of the current file (see exg_reader.cpp)
int next_batch(exg_reader *r, bool count_only, uint64_t *n_records_out);
}  // namespace exg_rd
