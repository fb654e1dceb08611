// exg_rd_internal.hpp — what the translation units of the reader level share (not part of any ABI):
//   exg_rd_io.cpp      pinned blocks, NUMA pinning, page cache -> pinned -> HBM uploads, the reader's device buffers
//   exg_rd_bgzf.cpp    host only: the BGZF / gzip member walk (no HIP call; under ASan in tests/host_asan_driver.cpp)
//   exg_rd_plan.cpp    host only: compression inference, shard planning, replacement_scan
//   exg_rd_source.cpp  compressed inputs as a bounded stream of decoded segments in HBM (DecodedSource)
//   exg_rd_gzip.cpp    the gzip / BGZF producer of such a stream;  exg_rd_zstd.cpp  the zstd producer
//   exg_rd_batch.cpp   open_next_file, next_batch: one device batch -> host vectors
//   exg_reader.cpp     the C entry points (exg_open ... exg_close) and the chunk slicing
#pragma once
#include <stdio.h>
#include <time.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "exg_reader.hpp"

namespace exg_rd {

#define RD_HIP(r, expr)                                                                            \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return ::exg_rd::fail(r, EXG_E_HIP, std::string(#expr " failed: ") + hipGetErrorString(_e)); \
    } while (0)

inline double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
inline bool trace_on() {
    static int on = getenv("EXG_TRACE") ? 1 : 0;
    return on;
}
#define TRACE(label, t0)                                                            \
    do {                                                                            \
        if (::exg_rd::trace_on()) fprintf(stderr, "[exg] %-22s %.1f ms\n", label, (::exg_rd::now_s() - (t0)) * 1e3); \
    } while (0)

// EXG_TRACE=2: absolute timestamps (ms since the process's first such line) of the producer's and the consumer's steps, one
// line each — what a pipeline's bubbles are read from
inline void trace_at(const char *what, uint64_t idx) {
    static const int lvl = getenv("EXG_TRACE") ? atoi(getenv("EXG_TRACE")) : 0;
    if (lvl < 2) return;
    static const double t0 = now_s();
    fprintf(stderr, "[exg@] %9.2f %s %llu\n", (now_s() - t0) * 1e3, what, (unsigned long long)idx);
}

// roctx ranges around the reader's stages (upload, inflate / decode, scan, columns back) so that a rocprofiler timeline
// (rocprofv3 --marker-trace) of a query is readable (the reference has no tracing of its own).  The marker
// library is looked up at run time (librocprofiler-sdk-roctx.so, ROCm's): the product does not link against a profiler; without
// it — or without EXG_ROCTX=1 — a range costs one predictable branch.
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
const RoctxApi &roctx_api();  // exg_rd_io.cpp
struct TraceRange {
    bool on;
    explicit TraceRange(const char *name) : on(roctx_api().push != nullptr) {
        if (on) (void)roctx_api().push(name);
    }
    ~TraceRange() {
        if (on) (void)roctx_api().pop();
    }
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
};

// a device buffer from the pool for the length of a scope. The stream that used it is waited for before the block goes
// back (idle already on the normal path; an error return may leave work in flight, and the pool is process-wide)
struct PoolBuf {
    int dev;
    hipStream_t stream;
    void *p = nullptr;
    size_t sz = 0;
    PoolBuf(int d, hipStream_t s) : dev(d), stream(s) {}
    PoolBuf(const PoolBuf &) = delete;
    PoolBuf &operator=(const PoolBuf &) = delete;
    void *take(size_t bytes) {
        release();
        sz = bytes ? bytes : 16;
        return p = dev_pool()->take(dev, sz);
    }
    void *detach() {  // the caller owns the block from here on (it goes back with dev_pool()->give(dev, p, sz))
        void *q = p;
        p = nullptr;
        return q;
    }
    void release() {
        if (!p) return;
        (void)hipStreamSynchronize(stream);
        dev_pool()->give(dev, p, sz);
        p = nullptr;
    }
    ~PoolBuf() { release(); }
};

// bytes in front of a shard that travel with its first batch (the beginning of the record that ends behind the cut);
// grown by the reader when the record turns out to begin further back
static constexpr uint64_t kShardHalo = 1u << 20;
// The next batch starts where this one's last complete record ends - known only after the scan - so the
// prefetch starts this many bytes before the end of the current batch; a batch whose unconsumed tail is
// longer (one giant record) falls back to the synchronous upload.
static constexpr uint64_t kPrefetchSlack = 1u << 20;
static constexpr uint64_t kRampFirstBytes = 32u << 20;  // the first device batch of a text file (exg_reader.hpp: ramp_bytes)
static constexpr size_t kUploadWindow = 256u << 20;

// ---- exg_rd_io.cpp
void pin_to_device_node(int device);
int list_files(exg_reader *r, const std::string &path);
// a consumer that follows an upload window by window (events recorded on the upload's stream)
struct UploadProgress {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<hipEvent_t> done;  // one per window, created by the consumer
    size_t recorded = 0;
    bool finished = false;
    int rc = 0;
    // window w has been enqueued (true) / the upload ended without it (false)
    bool wait_for(size_t w) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return recorded > w || finished; });
        return recorded > w;
    }
};
// file bytes [file_off, file_off + n) -> d_dst on `st` (NULL: r->stream): windows of 256 MiB through two pooled pinned
// blocks, each window read by parallel pread and sent slice by slice
int upload_file(exg_reader *r, void *d_dst, uint64_t n, uint64_t file_off = 0, hipStream_t st = nullptr, UploadProgress *prog = nullptr);
// the same without a reader (a decoder thread: errors go to *err, not to the reader's message)
int upload_fd(int device, int fd, void *d_dst, uint64_t n, uint64_t file_off, hipStream_t st, UploadProgress *prog, std::string *err);
// file bytes [off, off + n) -> the slot's pinned bounce buffer (parallel pread) -> d_in_slot[slot], on `st`
int upload_range(exg_reader *r, uint64_t off, uint64_t n, int slot, hipStream_t st);
// the same on a host thread of its own (pread + the H2D enqueue block their caller for as long as the bytes take to leave)
void start_upload(exg_reader *r, exg_reader::Prefetch *which, uint64_t start, uint64_t len, int slot);
// fd bytes [off, off + n) -> host memory `dst` by up to 8 threads pinned to the device's NUMA node; when d_dst != NULL every
// slice is sent on to d_dst + (its offset) on `st` as soon as it has been read.  false: short read (*hip_failed: a copy failed)
bool pread_parallel(int device, int fd, uint64_t off, size_t n, char *dst, char *d_dst, hipStream_t st, bool *hip_failed);

// ---- exg_rd_bgzf.cpp (host only)
// The few bytes the BGZF walk looks at — a member's header, the trailer right in front of the next header — read with pread
// into a small window, NOT through the file's mapping: a fault on the mapping maps sixteen pages (fault-around), two faults
// per 18 KB member map the whole file, and unmapping a 5 GB file that had been mapped that way cost 80-120 ms (page-table
// teardown, TLB shootdowns on a 256-thread host) behind a 300 ms decode; the walk itself was page-fault bound (35-55 ms
// per 5 GB on eight threads).  fd < 0: the bytes are in memory at `map`.
struct Peek {
    const uint8_t *map;
    int fd;
    uint64_t n;
    uint8_t buf[512];
    uint64_t b0 = ~0ull, b1 = 0;  // buf holds file bytes [b0, b1)
    Peek(const uint8_t *m, int f, uint64_t size) : map(m), fd(f), n(size) {}
    const uint8_t *at(uint64_t off, size_t len);  // file bytes [off, off + len), len <= 256; nullptr past the end of the file
};
uint64_t bgzf_member_at(Peek &f, uint64_t pos, exg_inflate_member *m, uint32_t *crc = nullptr);
uint64_t bgzf_find(const uint8_t *d, int fd, uint64_t n, uint64_t from);
bool bgzf_parallel_index(const uint8_t *d, int fd, uint64_t n, exg_inflate_member *members, uint64_t cap, uint64_t *k_out, uint64_t *total_out,
                         std::vector<uint32_t> *crc_out, uint64_t upto = ~0ull);

// ---- exg_rd_plan.cpp (host only)
bool parse_compression(const std::string &s, Compression *out);
Compression compression_of(const exg_open_args *args);


struct Stripe;
int list_path(const std::string &path, std::vector<std::string> *files, std::string *err);
int plan_stripes(const std::vector<std::string> &files, Compression compression, const exg_open_args *args, std::vector<Stripe> *out, unsigned *n_workers);

// ---- exg_rd_batch.cpp
int n_string_cols(int format);
// the next device batch with rows -> r->batch (false + *end: every file is exhausted); what exg_next_chunk and a fan-out
// worker both do between two batches: pending parse errors, the end of a file, the next file
int advance_batch(exg_reader *r, bool *end);
// the mapping of the current file lost pages to a truncation (exg_map_guard.hpp): EXG_E_IO
int truncated_while_read(exg_reader *r);

}  // namespace exg_rd
