// exg_reader.hpp — internals of the reader level shared by exg_reader.cpp (DuckDB-shaped chunks) and
// exg_arrow_stream.cpp (the reference's new_reader: Arrow record batches).
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>
#include <mutex>
#include <string>
#include <vector>

#include "exg_block_pool.hpp"
#include "exg_common.hpp"

// exg_crc32.hip: bytes (a multiple of 4, both ends 4-byte aligned) of device memory written to pinned host memory by a kernel on
// `stream` — a decoder's small results, which a copy would queue behind the big copies of its SDMA engine
namespace exg {
int post_to_host(void *h_dst, const void *d_src, uint64_t bytes, void *stream);
int stream_to_host(void *h_dst, const void *d_src, uint64_t bytes, void *stream);  // bulk bytes, 16-byte aligned, by a kernel's stores (exg_crc32.hip)
}

namespace exg_rd {
class DecodedSource;
class FanOut;

// HIP's current device is per host thread: every entry point that touches a reader runs with the reader's device
// current (the consumer may call from any thread — DuckDB binds on one thread and scans on others — and a process may
// hold readers on several devices), and leaves the caller's device as it found it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) switched = hipSetDevice(dev) == hipSuccess;
        if (prev != dev && !switched) (void)hipGetLastError();  // (a sticky error would surface at the next launch check)
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

struct PinnedBlock {  // host memory: pinned (hipHostMalloc) or a read-only file mapping
    void *p = nullptr;
    size_t n = 0;
    size_t mapped = 0;  // != 0: p is an mmap of that many bytes
    bool pooled = false;  // p came from the BlockPool (block size `cap`) and goes back to it
    size_t cap = 0;
    int guard = -1;  // a mapping: its slot in exg_map_guard.hpp (a file truncated while it is read is EXG_E_IO, not a SIGBUS)
    ~PinnedBlock();
};

// The pool of pinned host blocks (exg_block_pool.hpp: recycled, keyed by the NUMA node of the device that was current when a
// block was made), with HIP behind its hooks.  One pool per process: streams come and go (the reference opens one at bind
// and one per scan).
int numa_node_of_device(int device);  // exg_rd_io.cpp: sysfs numa_node of the device's PCI function (0 when unknown)
inline std::shared_ptr<BlockPool> global_pool() {
    // never destroyed: hipHostFree after the HIP runtime has shut down is not safe
    static auto *pool = [] {
        BlockPool::Hooks h;
        h.alloc = [](size_t n) -> void * {
            void *p = nullptr;
            if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            return p;
        };
        h.release = [](void *p) { (void)hipHostFree(p); };
        h.current_node = [] {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess) return 0;
            return numa_node_of_device(dev);
        };
        h.current_device = [] {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess) return 0;
            return dev;
        };
        const char *e = getenv("EXG_PINNED_POOL_MB");
        return new std::shared_ptr<BlockPool>(std::make_shared<BlockPool>(h, e ? ((size_t)std::max(0, atoi(e)) << 20) : 0));
    }();
    return *pool;
}

// Device buffers of a reader (input slots, workspace, column vectors, the Arrow emitter's arena) are the same
// sizes scan after scan; mapping and unmapping gigabytes of HBM per open costs tens of ms and was seen to
// stall later GPU work for far longer.  They go round through this pool instead (exact size match).
// HIP streams are expensive to create (a hardware queue each: ~1 ms measured per reader open) and a query on a small
// file needs two: they are recycled process-wide, per device; a stream goes back synchronised.
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> free_streams;  // first: device, or kHigh + device for a high-priority stream
    static constexpr int kHigh = 1 << 16;
    // high: the stream of a reader's SCAN.  A compressed input keeps the chip full of decoder wavefronts that live for
    // milliseconds (a wavefront per gzip member); the scan of the batch in front of them is 0.1 ms of work that the consumer —
    // and everything behind it: the column copies, the chunks — waits for: its workgroups go first when slots free up.
    hipError_t take(int dev, hipStream_t *out, bool high = false) {
        const int key = (high ? kHigh : 0) + dev;
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_streams.size(); i++)
                if (free_streams[i].first == key) {
                    *out = free_streams[i].second;
                    free_streams.erase(free_streams.begin() + (long)i);
                    return hipSuccess;
                }
        }
        DeviceGuard g(dev);
        if (high) {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least &&
                hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest) == hipSuccess)
                return hipSuccess;
            (void)hipGetLastError();
        }
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    }
    void give(int dev, hipStream_t s, bool high = false) {
        if (!s) return;
        (void)hipStreamSynchronize(s);
        std::lock_guard<std::mutex> g(mu);
        if (free_streams.size() < 32) {
            free_streams.emplace_back((high ? kHigh : 0) + dev, s);
            return;
        }
        (void)hipStreamDestroy(s);
    }
    void trim() {
        std::lock_guard<std::mutex> g(mu);
        for (auto &e : free_streams) (void)hipStreamDestroy(e.second);
        free_streams.clear();
    }
    // A stream for BIG device-to-host copies that run beside uploads (round 6: the columns' stream, the Arrow emitter's, a decoded
    // stream's mirrors).  Which copy engine a stream's D2H copies run on is a property of the stream (tools/sdma_pairs_probe.py: with
    // some streams as the D2H side ANY upload beside them queues on the same engine — 55 GB/s in sum instead of 96 — and it stays so
    // for the life of the process: the run-to-run halves of the D2H-bound legs, 0.91 or 0.78 of the link).  `calibrate`: a new
    // stream is tried out against an upload (32 MiB each way, ~3 ms) and a stream that shares the uploads' engine is set aside for
    // other work; streams that passed are kept apart and reused (exg_rd_io.cpp).  EXG_NO_D2H_CALIBRATION: any stream (A/B).
    static constexpr int kD2H = 1 << 17;
    std::vector<std::pair<hipStream_t, double>> d2h_tried;  // when a kept stream last passed its try-out (a long-lived process: tried again when a second has passed)
    hipError_t take_d2h(int dev, hipStream_t *out, bool calibrate);
    void give_d2h(int dev, hipStream_t s) {
        if (!s) return;
        (void)hipStreamSynchronize(s);
        std::lock_guard<std::mutex> g(mu);
        if (free_streams.size() < 32) {
            free_streams.emplace_back(kD2H + dev, s);
            return;
        }
        (void)hipStreamDestroy(s);
    }
};
inline StreamPool *stream_pool() {
    static StreamPool *pool = new StreamPool();  // never destroyed, like the other pools
    return pool;
}

// Device bytes a reader holds — its own buffers, the segments and scratch of its decoder — current and high-water mark
// (exg_reader_stats; what EXG_DEVICE_MEM_CAP_MB is checked against in the tests).  The pool charges the meter of the
// calling thread: the reader's entry points and its worker threads run inside a MeterScope.
struct MemMeter {
    std::atomic<uint64_t> cur{0}, peak{0};
    uint64_t cap = 0;  // EXG_DEVICE_MEM_CAP_MB of the reader that owns the meter (0: none): who allocates with slack looks here
    void add(uint64_t n) {
        const uint64_t v = cur.fetch_add(n) + n;
        uint64_t p = peak.load();
        while (v > p && !peak.compare_exchange_weak(p, v)) {
        }
    }
    void sub(uint64_t n) {
        uint64_t c = cur.load();
        while (!cur.compare_exchange_weak(c, c > n ? c - n : 0)) {
        }
    }
};
inline MemMeter *&tl_meter() {
    static thread_local MemMeter *m = nullptr;
    return m;
}
struct MeterScope {
    MemMeter *prev;
    explicit MeterScope(MemMeter *m) : prev(tl_meter()) { tl_meter() = m; }
    ~MeterScope() { tl_meter() = prev; }
    MeterScope(const MeterScope &) = delete;
    MeterScope &operator=(const MeterScope &) = delete;
};

struct DevPool {
    struct Blk {
        int dev;
        void *p;
        size_t sz;
    };
    std::mutex mu;
    std::vector<Blk> free_blocks;
    size_t pooled_bytes = 0;
    // of 288 GB; trimmed when a hipMalloc fails, by exg_trim_pools(), and bounded by EXG_POOL_MAX_GB (default 64)
    static size_t max_pooled() {
        static const size_t v = [] {
            const char *e = getenv("EXG_POOL_MAX_GB");
            return (size_t)(e ? strtoull(e, nullptr, 10) : 64) << 30;
        }();
        return v;
    }
    // Requests are rounded up to a size class — eight per octave (at most 12.5 % over) — so that the buffers of one file
    // serve the next file of about the same size; an exact-size pool kept one block per distinct byte count alive.
    static size_t size_class(size_t sz) {
        if (sz <= (1u << 20)) return (sz + 4095) & ~(size_t)4095;
        size_t step = (size_t)1 << (63 - __builtin_clzll((unsigned long long)sz));  // largest power of two <= sz
        step >>= 3;
        return (sz + step - 1) & ~(step - 1);
    }
    void *take(int dev, size_t sz) {
        sz = size_class(sz);
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_blocks.size(); i++)
                if (free_blocks[i].dev == dev && free_blocks[i].sz == sz) {
                    void *p = free_blocks[i].p;
                    pooled_bytes -= sz;
                    free_blocks.erase(free_blocks.begin() + (long)i);
                    if (MemMeter *m = tl_meter()) m->add(sz);
                    return p;
                }
        }
        DeviceGuard g(dev);
        void *p = nullptr;
        if (hipMalloc(&p, sz) != hipSuccess) {
            (void)hipGetLastError();
            trim(0);  // give the cached blocks back and try once more
            if (hipMalloc(&p, sz) != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
        }
        if (MemMeter *m = tl_meter()) m->add(sz);
        return p;
    }
    void give(int dev, void *p, size_t sz) {  // sz: what take() was asked for (or any size of the same class)
        if (!p) return;
        sz = size_class(sz);
        if (MemMeter *m = tl_meter()) m->sub(sz);
        {
            std::lock_guard<std::mutex> g(mu);
            if (pooled_bytes + sz <= max_pooled()) {  // (small blocks too: a hipMalloc / hipFree pair per open or per file is 0.1-1 ms)
                free_blocks.push_back(Blk{dev, p, sz});
                pooled_bytes += sz;
                return;
            }
        }
        (void)hipFree(p);
    }
    void trim(size_t keep_bytes) {
        std::lock_guard<std::mutex> g(mu);
        while (!free_blocks.empty() && pooled_bytes > keep_bytes) {
            (void)hipFree(free_blocks.back().p);
            pooled_bytes -= free_blocks.back().sz;
            free_blocks.pop_back();
        }
    }
};
inline DevPool *dev_pool() {
    static DevPool *pool = new DevPool();  // intentionally never destroyed: the HIP runtime may be gone by then
    return pool;
}

struct HostArena {  // pinned blocks that live as long as the batch they back
    std::shared_ptr<BlockPool> pool = global_pool();
    std::vector<std::pair<char *, size_t>> blocks;
    size_t used = 0, total = 0;
    // first block sized for the whole batch (hint = what the previous batch of this scan needed)
    void reserve(size_t hint) {
        if (!blocks.empty() || hint == 0) return;
        size_t sz = hint;
        char *p = pool->take(&sz);
        if (p) blocks.emplace_back(p, sz), used = 0;
    }
    void *alloc(size_t n) {
        n = (n + 63) & ~(size_t)63;
        if (n == 0) n = 64;
        if (blocks.empty() || used + n > blocks.back().second) {
            size_t sz = n;
            char *p = pool->take(&sz);
            if (!p) return nullptr;
            blocks.emplace_back(p, sz);
            used = 0;
        }
        void *p = blocks.back().first + used;
        used += n;
        total += n;
        return p;
    }
    ~HostArena() {
        for (auto &b : blocks) pool->give(b.first, b.second);
    }
};

// One node of a nested column (LIST / STRUCT and what is inside) of a device batch, on the host.  A node has an index
// space — rows for a top-level column, the elements of the enclosing list otherwise — and a DataChunk is a slice of it:
// rows [cB, cB + B) for row space, [base[c], base[c + 1]) for the children of a LIST, where base = the list node's
// child_base (n_chunks + 1 entries, made on the device together with the chunk-relative list entries).
struct NVec {
    int type = 0;                        // EXG_TYPE_*; 0: not a nested column
    const void *data = nullptr;          // batch-wide array in the batch's pinned arena
    const uint64_t *validity = nullptr;  // bit per element, or NULL
    uint32_t elem = 0;                   // bytes per element
    uint64_t length = 0;
    const uint64_t *child_base = nullptr;  // LIST: where every chunk's elements begin in children[0]
    // round 6: no element of this node is valid / non-empty in the whole batch — data and validity are ONE block of zeros that every
    // chunk's slice points at (it never crossed PCIe): a LIST's entries are all {0, 0}, a scalar's values 0 and its bits NULL
    bool zero = false;
    std::vector<NVec> children;
};

struct Batch {  // host vectors of one device batch, shared by its chunks
    std::shared_ptr<PinnedBlock> file;
    HostArena host;
    int n_cols = 0;
    void *cols[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t elem[9] = {16, 16, 16, 16, 16, 16, 16, 16, 16};  // bytes per row
    void *validity[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // NULL => all valid
    void *payload = nullptr;  // FASTA: compacted sequences
    uint64_t n_rows = 0;
    uint64_t seq = 0;          // which device batch of its reader this is (exg_chunk.batch_no)
    std::vector<NVec> nested;  // per column; type == 0 for the flat ones
    // read_vcf (round 6): the batch is handed to the reader while its vectors are still on their way — recorded behind the last
    // copy; whoever hands out the batch's first chunk waits for it (wait_landed).  NULL: everything had landed when it was made.
    hipEvent_t landed = nullptr;
    int wait_landed() {
        if (!landed) return 0;
        const hipError_t e = hipEventSynchronize(landed);
        (void)hipEventDestroy(landed);
        landed = nullptr;
        return e == hipSuccess ? 0 : -1;
    }
    ~Batch() {
        if (landed) {
            (void)hipEventSynchronize(landed);  // (the pinned blocks go back to the pool: nothing may still be writing them)
            (void)hipEventDestroy(landed);
        }
    }
};

struct ChunkKeep {
    std::shared_ptr<Batch> batch;
    // the chunk's exg_vector trees and the validity words of slices that do not begin on a word boundary
    std::vector<std::unique_ptr<exg_vector[]>> nodes;
    std::vector<std::unique_ptr<uint64_t[]>> words;
    exg_vector top[16];
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
    uint64_t want_cols = ~0ull;  // exg_open_args.columns: the columns whose vectors are copied back (all are parsed)
    bool want(int c) const { return (want_cols >> c) & 1ull; }
    bool expect_chunks = false;  // EXG_OPEN_CHUNKS: chunks will be pulled (a decoded source mirrors its segments to the host from the first one on)
    int device = 0;
    std::string error;
    hipStream_t stream = nullptr;

    // current file
    std::shared_ptr<exg_rd::PinnedBlock> file;
    uint64_t file_pos = 0;  // first byte not yet consumed by a complete record
    bool file_done = true;

    // device buffers (sized for device_batch_bytes).  Two input slots: while slot A is scanned, the bytes the
    // next batch is expected to need are already travelling into slot B on `up_stream` (see next_batch).
    void *d_in = nullptr, *d_ws = nullptr, *d_res = nullptr;  // d_in = the slot the current batch sits in
    void *d_in_slot[2] = {nullptr, nullptr};
    hipStream_t up_stream = nullptr;
    // read_vcf: the flat columns' copies back run on a stream of their own beside the nested columns' kernels (next_batch)
    hipStream_t col_stream = nullptr;
    hipEvent_t col_ev = nullptr;
    // read_vcf: the next scan may overwrite the columns only when the flat columns' copies of the batch before have left them
    hipEvent_t flat_ev = nullptr;
    bool flat_pending = false;
    bool lazy_landing = false;  // this batch's vectors are not waited for inside next_batch (Batch::landed)
    // per input slot: the thread of the upload that is filling it (pread + H2D enqueue), its result, the event behind its copies
    std::thread up_thread_of[2];
    int up_rc_of[2] = {0, 0};
    hipEvent_t up_done_of[2] = {nullptr, nullptr};
    int join_prefetch();  // the upload of the batch this call is about to use (pf)
    struct Prefetch {
        bool valid = false;
        uint64_t file_start = 0, len = 0;  // file bytes [file_start, file_start + len) are (being) uploaded
        int slot = 0;
    } pf;
    // a second upload in flight: the batch AFTER the one that is about to be scanned starts travelling as soon as that one's
    // own upload is known to cover it (the slot of the batch before is free by then), so that the link never waits for a scan
    Prefetch pf2;
    void drop_prefetch2();  // join + let it land + forget
    // The first batches of a text file are small and double (round 6): 32 MiB, 64, 128, ... up to device_batch_bytes.  The first
    // upload has nothing to hide behind — a full 256 MiB batch kept the scan, and the link back to the host, idle for ~6 ms at
    // the head of every query; behind a 32 MiB batch the first DataChunk leaves after ~1.5 ms and every later upload travels
    // beside the vectors of the batch in front.  0: not ramping (the batches are device_batch_bytes).
    uint64_t ramp_bytes = 0;
    uint64_t next_ramp() {  // the size of the next upload, and one step up the ramp
        if (!ramp_bytes) return device_batch_bytes;
        const uint64_t n = std::min<uint64_t>(ramp_bytes, device_batch_bytes);
        ramp_bytes = n >= device_batch_bytes ? 0 : n * 2;
        return n;
    }
    int cur_slot = 0;
    size_t host_hint = 0;  // pinned bytes the previous batch's host vectors needed
    std::vector<std::pair<void **, size_t>> dev_allocs;  // pooled device buffers of this reader (slot, bytes)
    int dev_alloc(void **slot, size_t bytes);
    uint32_t shard_index = 0, shard_count = 1;  // byte-range shards of every file (exg_open_args)
    uint64_t range_hi = 0;    // this reader's bytes of the current file end here (file size without shards)
    bool shard_first = false; // the next batch is the first of a shard that begins inside the file: halo + phase
    // bytes in front of a shard's cut that travel with its first batch (the beginning of the record that ends behind the
    // cut).  1 MiB (EXG_SHARD_HALO) to begin with; when the scan reports that the record begins further back
    // (EXG_RF_HEAD_UNRESOLVED) the halo is grown eightfold and the batch scanned again, up to the first byte of the data:
    // a record of any length across a cut is found, like the unsharded scan finds it
    uint64_t halo_want = 0;
    // a shard of a compressed FASTQ whose 4-line phase cannot be told from the bytes around the cut: the newlines in front of
    // its own bytes are COUNTED, by a decoder of their own over the members / frames in front (own_c_begin: where the shard's
    // own begin in the file), segment by segment — the prefix is never resident
    uint64_t own_c_begin = 0;
    bool exact_nl_known = false;
    uint64_t exact_nl = 0;
    bool fa_shard = false;     // a shard of a compressed FASTA: file_pos is its first record, fa_end (once known) the first that is not its own
    uint64_t fa_end = ~0ull;
    bool range_eof = true;    // range_hi is the end of the file's data (a later shard follows otherwise)
    bool range_preset = false;  // BGZF shard: inflate_file chose the members, file_pos / range_hi refer to ITS inflated bytes
    uint64_t preset_pos = 0;    // ... first owned inflated byte (what is in front of it is the halo)
    uint64_t data_base = 0;           // first data byte in the coordinates of file_pos (behind the VCF header; 0 in a BGZF shard's own buffer)
    bool data0_is_line_start = true;  // byte 0 of the data (behind the header) begins a line (not so for a BGZF shard's halo)
    void *d_phase = nullptr;  // device u32 for exg_fastq_guess_phase
    uint64_t gz_header_prefix = 0;  // gzip + VCF: bytes of the inflated file's start held in file->p (header parse)
    bool worst_case_rows = false;
    bool ws_full = false;  // under EXG_DEVICE_MEM_CAP_MB: a batch overflowed the budgeted line index, the workspace is at full size
    // Which scan a batch starts with is sticky (an input keeps its shape): EXG_ALGO_FUSED (the lean scan + the any-shape run
    // over what it marked) until a batch comes back with EXG_RF_REDO on more than an eighth of its super-tiles — long reads,
    // reads below ~45 bp, multi-sample VCF lines, bytes >= 0x80 throughout —, then EXG_ALGO_FUSED_FULL, the any-shape scan
    // alone, for the rest of the input (the lean scan would mark every tile and be a pass wasted per batch).
    uint32_t fused_algo = EXG_ALGO_FUSED;
    // exg_open_args.filters: postfix program + constants in device memory, a row map and one column of scratch
    bool has_filter = false;
    uint64_t filter_cols = 0;  // the columns the predicate reads (bit c: schema column c)
    void *d_filter_prog = nullptr, *d_filter_consts = nullptr;  // pooled
    size_t filter_prog_bytes = 0, filter_consts_bytes = 0;
    void *d_row_map = nullptr, *d_gather = nullptr, *d_filter_tmp = nullptr;  // output vectors sized for the densest possible input (after an overflow)
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
    exg_rd::PinnedBlock staging[2];  // pinned bounce buffers for H2D, one per slot (the file itself is only mapped)
    // the decoder behind the current file has ended without an error (a checksum that is verified behind the last segment
    // is reported when the file's last batch has been handed out, where a streaming decoder reports it too)
    int finish_source();
    // gzip / zstd input: the decoded bytes arrive as a bounded stream of segments in HBM and are scanned in place
    // (exg_rd_source.hpp); positions (file_pos, range_hi, ...) are offsets in the DECODED stream
    std::unique_ptr<exg_rd::DecodedSource> src;
    // shard_count = 0 on a box with several devices: the input is read as stripes by readers of their own on all devices and
    // handed out here in file order (exg_rd_fanout.hpp); this reader then only slices the batches into chunks
    std::unique_ptr<exg_rd::FanOut> fan;
    exg_rd::MemMeter meter;      // device bytes held on behalf of this reader
    uint64_t mem_cap = 0;        // EXG_DEVICE_MEM_CAP_MB: what the batch / segment sizes are derived from (0: defaults)
    std::atomic<uint64_t> n_segments{0};  // decoded segments consumed so far (read by exg_reader_stats_of from any thread)
    std::atomic<uint64_t> n_batches{0};   // device batches scanned so far
    std::atomic<uint64_t> nested_ns{0}, host_vector_bytes{0};  // exg_reader_stats (ABI 9)

    // current batch
    std::shared_ptr<exg_rd::Batch> batch;
    uint64_t batch_row = 0;
    uint64_t batch_seq = 0;  // device batches handed out so far (all files)
    uint32_t pending_error = 0;  // parse error to raise once the rows before it have been handed out
    uint64_t pending_error_offset = 0;

    // Arrow mode (new_reader): the columns stay on the device and `arrow_emit` turns them into Arrow buffers
    int (*arrow_emit)(exg_reader *, const exg_rd::ScanCtx &) = nullptr;
    std::shared_ptr<void> arrow_state;
    // chunk mode, VCF: the INFO / FORMAT keys of the header, the schema trees, the emitter's device arena
    std::shared_ptr<void> nested_state;

    void free_device();
    exg_reader();  // (out of line: `src` is a pointer to a type this header only declares)
    // The chunk boundary runs one device batch AHEAD of its consumer (round 6): `cur` is the batch whose chunks are being handed
    // out; when its first chunk leaves, a thread of the reader's own starts making the batch behind it (advance_batch -> `batch`) —
    // upload wait, scan, nested columns, the vectors' way back — while the consumer (DuckDB's operators) works through cur's
    // chunks.  cur is whole on the host by then and the device buffers are the next scan's.  One thread at a time; every other
    // entry point joins it first (join_ahead).
    std::shared_ptr<exg_rd::Batch> cur;
    uint64_t cur_row = 0;
    std::thread ahead;
    int ahead_rc = 0;
    bool ahead_end = false, ahead_done = false;
    void join_ahead() {
        if (ahead.joinable()) ahead.join();
    }
    ~exg_reader();
};

namespace exg_rd {
int fail(exg_reader *r, int code, const std::string &msg);
int open_next_file(exg_reader *r);
// Scan the next device batch of the current file (see exg_reader.cpp)
int next_batch(exg_reader *r, bool count_only, uint64_t *n_records_out);
// exg_arrow_stream.cpp — the nested VCF columns at the chunk boundary (DuckDB vector layouts built on the device):
// header keys + schema trees (needs the first file open), and the columns of one scanned batch
int nested_prepare(exg_reader *r);
void nested_schema(exg_reader *r, exg_schema *out);
int nested_emit(exg_reader *r, const ScanCtx &ctx, Batch *b, const uint32_t *d_row_map, uint64_t *n_rows);
}  // namespace exg_rd
