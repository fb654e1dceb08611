// exg_rd_io.cpp — reader level, I/O: pinned host blocks, the reader's pooled device buffers and upload threads, and the
// page cache -> pinned bounce buffer -> HBM copies (parallel pread, each slice followed at once by its own H2D).
// Part of the replacement of `new_reader`'s file opening / BufReader (rust/src/arrow_reader.rs:104-118).
#include <dirent.h>
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <sched.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "exg_rd_fanout.hpp"
#include "exg_map_guard.hpp"
#include "exg_rd_source.hpp"

namespace exg_rd {

const RoctxApi &roctx_api() {
    static const RoctxApi api = [] {
        RoctxApi a;
        if (!getenv("EXG_ROCTX")) return a;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        a.pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
        return a;
    }();
    return api;
}

PinnedBlock::~PinnedBlock() {
    if (!p) return;
    if (mapped) {
        MapGuard::remove(guard);
        munmap(p, mapped);
    }
    else if (pooled)
        global_pool()->give((char *)p, cap);
    else
        (void)hipHostFree(p);
}

// The reader's own I/O threads (pread into the pinned bounce buffers, H2D enqueue) run on the CPUs of the NUMA node the
// GPU hangs off (sysfs local_cpulist of its PCI function): the bounce buffers are local to the DMA engine, and a file
// that is read cold lands in that node's page cache.  Best effort — a cpuset that forbids it is not an error.
void pin_to_device_node(int device) {
    struct Mask {
        bool ok = false;
        cpu_set_t set;
    };
    static Mask masks[64];
    static std::once_flag once[64];
    if (device < 0 || device >= 64 || getenv("EXG_NO_NUMA_PIN")) return;
    std::call_once(once[device], [device] {
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) return;
        for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
        const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist";
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return;
        char line[4096] = {0};
        const bool got = fgets(line, sizeof line, f) != nullptr;
        fclose(f);
        if (!got) return;
        Mask &m = masks[device];
        CPU_ZERO(&m.set);
        int n_cpus = 0;
        for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
            int a = 0, b = 0;
            const int k = sscanf(tok, "%d-%d", &a, &b);
            if (k == 1) b = a;
            if (k < 1) continue;
            for (int c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET(c, &m.set), n_cpus++;
        }
        m.ok = n_cpus > 0;
    });
    if (masks[device].ok) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &masks[device].set);
}

// NUMA node of the device's PCI function (what a pinned block made with that device current is placed on): the pool's key
int numa_node_of_device(int device) {
    static int nodes[64];
    static std::once_flag once[64];
    if (device < 0 || device >= 64) return 0;
    std::call_once(once[device], [device] {
        nodes[device] = 0;
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) return;
        for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
        const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
        if (FILE *f = fopen(path.c_str(), "r")) {
            int v = -1;
            if (fscanf(f, "%d", &v) == 1 && v >= 0) nodes[device] = v;
            fclose(f);
        }
    });
    return nodes[device];
}

// do a D2H copy on `c` and an upload (4 MiB slices on another stream) overlap?  16 MiB each way out of ONE pinned and ONE device block:
// ~0.35 ms when they do, ~0.6 ms when they queue on one engine; the best of three timed rounds behind a warm-up (a stream's first copy
// sets its queue up).  With the blocks and the streams it makes, a try-out costs the process ~20 ms, once.
static bool d2h_overlaps_uploads(int dev, hipStream_t c) {
    constexpr size_t kHalf = 16u << 20, kSlice = 4u << 20;
    hipStream_t up = nullptr;
    if (stream_pool()->take(dev, &up) != hipSuccess) return true;
    size_t cap = 2 * kHalf;
    char *h = global_pool()->take(&cap);
    char *d = (char *)dev_pool()->take(dev, 2 * kHalf);
    bool ok = true;
    if (h && d) {
        double best = 1e9;
        for (int round = 0; round < 4 && ok; round++) {
            (void)hipStreamSynchronize(up);
            (void)hipStreamSynchronize(c);
            const double t0 = now_s();
            ok = hipMemcpyAsync(h + kHalf, d + kHalf, kHalf, hipMemcpyDeviceToHost, c) == hipSuccess;
            for (size_t o = 0; ok && o < kHalf; o += kSlice) ok = hipMemcpyAsync(d + o, h + o, kSlice, hipMemcpyHostToDevice, up) == hipSuccess;
            ok = ok && hipStreamSynchronize(up) == hipSuccess && hipStreamSynchronize(c) == hipSuccess;
            if (round) best = std::min(best, now_s() - t0);
        }
        if (!ok) (void)hipGetLastError();
        if (getenv("EXG_TRACE")) fprintf(stderr, "[exg] D2H stream %p beside an upload: 2 x 16 MiB in %.2f ms (%s)\n", (void *)c, best * 1e3, best < 0.47e-3 ? "they overlap" : "one engine");
        ok = !ok || best < 0.47e-3;  // (a failed probe decides nothing)
    }
    if (h) global_pool()->give(h, cap);
    if (d) dev_pool()->give(dev, d, 2 * kHalf);
    stream_pool()->give(dev, up);
    return ok;
}

hipError_t StreamPool::take_d2h(int dev, hipStream_t *out, bool calibrate) {
    static const bool no_calibration = getenv("EXG_NO_D2H_CALIBRATION") != nullptr;
    if (no_calibration) calibrate = false;  // (streams given back through give_d2h are still found below)
    DeviceGuard g(dev);
    auto passed_at = [&](hipStream_t st, double t) {  // (mu held)
        for (auto &e : d2h_tried)
            if (e.first == st) {
                e.second = t;
                return;
            }
        d2h_tried.emplace_back(st, t);
    };
    for (;;) {
        hipStream_t kept = nullptr;
        double tried = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_streams.size(); i++)
                if (free_streams[i].first == kD2H + dev) {  // (one that passed before)
                    kept = free_streams[i].second;
                    free_streams.erase(free_streams.begin() + (long)i);
                    break;
                }
            for (const auto &e : d2h_tried)
                if (e.first == kept) tried = e.second;
        }
        if (!kept) break;
        // (which engine a stream's copies take is not promised to stay: in a long-lived process a kept stream is tried again — ~2 ms
        // with warm pools — when a big input asks for it more than a second after it last passed)
        if (!calibrate || now_s() - tried < 1.0 || d2h_overlaps_uploads(dev, kept)) {
            if (calibrate && now_s() - tried >= 1.0) {
                std::lock_guard<std::mutex> lk(mu);
                passed_at(kept, now_s());
            }
            *out = kept;
            return hipSuccess;
        }
        give(dev, kept);  // (no longer apart: fine for kernels and uploads)
    }
    if (!calibrate) return take(dev, out);
    const double t_cal = now_s();
    struct CalTrace {
        double t0;
        ~CalTrace() {
            if (getenv("EXG_TRACE")) fprintf(stderr, "[exg] D2H stream try-out: %.1f ms in all\n", (now_s() - t0) * 1e3);
        }
    } cal_trace{t_cal};
    hipStream_t cand = nullptr;
    for (int k = 0; k < 4; k++) {
        const hipError_t e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        if (k == 3 || d2h_overlaps_uploads(dev, cand)) break;
        give(dev, cand);  // (fine for kernels and uploads)
        cand = nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        passed_at(cand, now_s());
    }
    *out = cand;
    return hipSuccess;
}

int fail(exg_reader *r, int code, const std::string &msg) {
    r->error = msg;
    exg::set_error("%s", msg.c_str());
    return code;
}

int list_files(exg_reader *r, const std::string &path) {
    std::string err;
    const int rc = list_path(path, &r->files, &err);  // (exg_rd_plan.cpp: host only)
    return rc ? fail(r, rc, err) : EXG_OK;
}

}  // namespace exg_rd

exg_reader::FdCloser::~FdCloser() {
    if (fd >= 0) close(fd);
}
int exg_reader::join_prefetch() {
    int rc = 0;
    for (int k = 0; k < 2; k++) {
        if (pf2.valid && pf2.slot == k) continue;  // (the batch after the coming one: not this call's)
        if (up_thread_of[k].joinable()) up_thread_of[k].join();
        if (up_rc_of[k] && !rc) rc = up_rc_of[k];
        up_rc_of[k] = 0;
    }
    if (rc) pf.valid = false;
    return rc;
}
void exg_reader::drop_prefetch2() {
    for (int k = 0; k < 2; k++)
        if (pf2.valid && pf2.slot == k && up_thread_of[k].joinable()) {
            up_thread_of[k].join();
            up_rc_of[k] = 0;
        }
    if (pf2.valid && up_stream) (void)hipStreamSynchronize(up_stream);
    pf2.valid = false;
}
void exg_reader::free_device() {
    (void)join_prefetch();
    drop_prefetch2();
    if (up_stream) (void)hipStreamSynchronize(up_stream);
    if (col_stream) (void)hipStreamSynchronize(col_stream);  // (copies out of the columns that are about to be freed)
    flat_pending = false;
    pf.valid = false;
    d_in = nullptr;
    for (auto &a : dev_allocs) {
        exg_rd::dev_pool()->give(device, *a.first, a.second);
        *a.first = nullptr;
    }
    dev_allocs.clear();
}
int exg_reader::dev_alloc(void **slot, size_t bytes) {
    bytes = (bytes + 4095) & ~(size_t)4095;
    *slot = exg_rd::dev_pool()->take(device, bytes);
    if (!*slot) return exg_rd::fail(this, EXG_E_HIP, "out of device memory (" + std::to_string(bytes >> 20) + " MiB)");
    dev_allocs.emplace_back(slot, bytes);
    return EXG_OK;
}
int exg_reader::finish_source() {
    if (!src) return EXG_OK;
    std::string e;
    const int rc = src->finish(&e);
    return rc ? exg_rd::fail(this, rc, e) : EXG_OK;
}
exg_reader::exg_reader() {}
exg_reader::~exg_reader() {
    join_ahead();
    exg_rd::DeviceGuard guard(device);
    exg_rd::MeterScope meter_scope(&meter);
    fan.reset();  // (its workers close their readers)
    src.reset();  // (its thread reads the file through fd_keep: before the descriptor closes)
    free_device();
    if (d_res) exg_rd::dev_pool()->give(device, d_res, 4096);
    if (d_phase) exg_rd::dev_pool()->give(device, d_phase, 4096);
    if (d_filter_prog) exg_rd::dev_pool()->give(device, d_filter_prog, filter_prog_bytes);
    if (d_filter_consts) exg_rd::dev_pool()->give(device, d_filter_consts, filter_consts_bytes);
    for (int k = 0; k < 2; k++)
        if (up_done_of[k]) (void)hipEventDestroy(up_done_of[k]);
    if (col_ev) (void)hipEventDestroy(col_ev);
    if (flat_ev) (void)hipEventDestroy(flat_ev);
    exg_rd::stream_pool()->give_d2h(device, col_stream);
    exg_rd::stream_pool()->give(device, up_stream);
    exg_rd::stream_pool()->give(device, stream, /*high=*/getenv("EXG_NO_SCAN_PRIORITY") == nullptr);
}

namespace exg_rd {

// CPUs this process may really use: the affinity mask and the cgroup CPU quota, not the machine's core count
static unsigned usable_cpus() {
    static const unsigned v = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[64] = {0};
            long period = 0;
            if (fscanf(f, "%63s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
                n = std::min<unsigned>(n, (unsigned)std::max(1l, atol(quota) / period));
            fclose(f);
        }
        return n;
    }();
    return v;
}
// Readers that are uploading at the same time share the host's cores and memory system (one per GPU inside a fan-out, or
// one process per GPU): eight pread threads each is right for one reader, and measured WORSE than two each when eight
// readers run on a box with sixteen usable cores (bench.py host_pipeline_scaling: 27 GB/s against 80).
static std::atomic<int> g_uploads_in_flight{0};
static size_t io_threads_now() {
    static const int forced = getenv("EXG_IO_THREADS") ? std::max(1, atoi(getenv("EXG_IO_THREADS"))) : 0;
    if (forced) return (size_t)forced;
    int peers = std::max(1, g_uploads_in_flight.load(std::memory_order_relaxed));
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) peers = std::max(peers, atoi(e));  // one process per GPU (torchrun)
    return (size_t)std::min(8u, std::max(2u, usable_cpus() / (unsigned)peers));
}

bool pread_parallel(int device, int fd, uint64_t off, size_t n, char *dst, char *d_dst, hipStream_t st, bool *hip_failed) {
    TraceRange range(d_dst ? "exg: pread + h2d" : "exg: pread");
    static const size_t slice = getenv("EXG_IO_SLICE_MB") ? ((size_t)std::max(1, atoi(getenv("EXG_IO_SLICE_MB"))) << 20) : (8u << 20);
    struct InFlight {
        InFlight() { g_uploads_in_flight.fetch_add(1, std::memory_order_relaxed); }
        ~InFlight() { g_uploads_in_flight.fetch_sub(1, std::memory_order_relaxed); }
    } in_flight;
    const size_t max_io_threads = io_threads_now();
    const size_t n_slices = (n + slice - 1) / slice;
    const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), std::min<size_t>(n_slices, max_io_threads));
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&](bool own_thread) {
        (void)hipSetDevice(device);
        if (own_thread) pin_to_device_node(device);
        for (size_t i = next.fetch_add(1); i < n_slices; i = next.fetch_add(1)) {
            const size_t o = i * slice, len = std::min<size_t>(slice, n - o);
            size_t got = 0;
            while (got < len) {
                const ssize_t k = pread(fd, dst + o + got, len - got, (off_t)(off + o + got));
                if (k <= 0) {
                    bad = 1;
                    return;
                }
                got += (size_t)k;
            }
            if (d_dst && hipMemcpyAsync(d_dst + o, dst + o, len, hipMemcpyHostToDevice, st) != hipSuccess) bad = 2;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, true);
    work(false);
    for (auto &t : th) t.join();
    if (hip_failed) *hip_failed = bad == 2;
    return bad == 0;
}

// The whole (compressed) file, or a range of it -> d_dst: windows of 256 MiB through two pooled pinned blocks, each window
// read by parallel pread and sent slice by slice (a hipMemcpyAsync straight from the page-cache mapping is a pageable
// copy: one staging thread inside the runtime, 10-33 GB/s depending on the box).
int upload_fd(int device, int fd, void *d_dst, uint64_t n, uint64_t file_off, hipStream_t st, UploadProgress *prog, std::string *err) {
    const size_t window = kUploadWindow;
    char *blk[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    hipEvent_t ev[2] = {nullptr, nullptr};
    struct Cleanup {
        hipStream_t st;
        char **blk;
        size_t *cap;
        hipEvent_t *ev;
        ~Cleanup() {
            (void)hipStreamSynchronize(st);  // the blocks are sources of copies in flight
            for (int k = 0; k < 2; k++) {
                if (blk[k]) global_pool()->give(blk[k], cap[k]);
                if (ev[k]) (void)hipEventDestroy(ev[k]);
            }
        }
    } cleanup{st, blk, cap, ev};
#define UP_HIP(expr)                                                                 \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            *err = std::string(#expr " failed: ") + hipGetErrorString(_e);           \
            return EXG_E_HIP;                                                        \
        }                                                                            \
    } while (0)
    for (int k = 0; k < 2 && (uint64_t)k * window < n; k++) {
        cap[k] = (size_t)std::min<uint64_t>(window, n - (uint64_t)k * window) + 64;
        blk[k] = global_pool()->take(&cap[k]);
        if (!blk[k]) {
            *err = "out of pinned host memory";
            return EXG_E_HIP;
        }
        UP_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    }
    uint64_t off = 0;
    for (uint64_t w = 0; off < n; w++) {
        const int b = (int)(w & 1);
        if (w >= 2) UP_HIP(hipEventSynchronize(ev[b]));  // the block's previous window has left
        const size_t len = (size_t)std::min<uint64_t>(window, n - off);
        bool hip_failed = false;
        if (!pread_parallel(device, fd, file_off + off, len, blk[b], (char *)d_dst + off, st, &hip_failed)) {
            *err = hip_failed ? "hipMemcpyAsync failed" : "short read";
            return hip_failed ? EXG_E_HIP : EXG_E_IO;
        }
        UP_HIP(hipEventRecord(ev[b], st));
        if (prog) {
            UP_HIP(hipEventRecord(prog->done[w], st));
            std::lock_guard<std::mutex> g(prog->mu);
            prog->recorded = (size_t)w + 1;
            prog->cv.notify_all();
        }
        off += len;
    }
#undef UP_HIP
    return EXG_OK;
}
int upload_file(exg_reader *r, void *d_dst, uint64_t n, uint64_t file_off, hipStream_t st, UploadProgress *prog) {
    std::string err;
    const int rc = upload_fd(r->device, r->fd_keep->fd, d_dst, n, file_off, st ? st : r->stream, prog, &err);
    return rc ? fail(r, rc, err) : EXG_OK;
}

// file bytes [off, off + n) -> the slot's pinned bounce buffer (parallel pread) -> d_in_slot[slot], on `st`.
// pread and H2D are pipelined slice by slice: a slice travels while the next ones are still being read.
int upload_range(exg_reader *r, uint64_t off, uint64_t n, int slot, hipStream_t st) {
    const uint64_t padded = (n + 15) / 16 * 16;
    PinnedBlock &stg = r->staging[slot];
    if (stg.n < padded) {
        RD_HIP(r, hipStreamSynchronize(r->stream));
        RD_HIP(r, hipStreamSynchronize(r->up_stream));
        if (stg.p) global_pool()->give((char *)stg.p, stg.cap), stg.p = nullptr, stg.n = 0;
        double t0 = now_s();
        size_t want = (size_t)std::max<uint64_t>(padded, std::min<uint64_t>(r->d_in_cap, r->file->n + 16)) + 64;
        stg.p = global_pool()->take(&want);
        if (!stg.p) return fail(r, EXG_E_HIP, "out of pinned host memory");
        stg.n = want;
        stg.cap = want;
        stg.pooled = true;
        TRACE("pinned staging", t0);
    }
    double t0 = now_s();
    char *dst = (char *)stg.p;
    char *d_dst = (char *)r->d_in_slot[slot];
    // the last slice's copy carries the zero padding up to the 16-byte boundary: read everything first when there is one
    const uint64_t body = n & ~15ull;  // bytes that travel slice by slice
    bool hip_failed = false;
    if (body && !pread_parallel(r->device, r->fd_keep->fd, off, (size_t)body, dst, d_dst, st, &hip_failed))
        return hip_failed ? fail(r, EXG_E_HIP, "hipMemcpyAsync failed") : fail(r, EXG_E_IO, "short read");
    if (padded > body) {
        memset(dst + body, 0, (size_t)(padded - body));
        size_t got = 0;
        while (body + got < n) {
            const ssize_t k = pread(r->fd_keep->fd, dst + body + got, (size_t)(n - body) - got, (off_t)(off + body + got));
            if (k <= 0) return fail(r, EXG_E_IO, "short read");
            got += (size_t)k;
        }
        RD_HIP(r, hipMemcpyAsync(d_dst + body, dst + body, (size_t)(padded - body), hipMemcpyHostToDevice, st));
    }
    TRACE("pread + h2d enqueue", t0);
    return EXG_OK;
}

// an upload of file bytes [start, start + len) into input slot `slot`, on a host thread of its own: pread + the H2D enqueue
// block their caller for as long as the bytes take to leave (5.5 ms per 256 MiB)
void start_upload(exg_reader *r, exg_reader::Prefetch *which, uint64_t start, uint64_t len, int slot) {
    r->up_rc_of[slot] = 0;
    r->up_thread_of[slot] = std::thread([r, start, len, slot] {
        (void)hipSetDevice(r->device);
        pin_to_device_node(r->device);
        trace_at("U upload begins, slot", (uint64_t)slot);
        int rc3 = upload_range(r, start, len, slot, r->up_stream);
        if (!rc3 && hipEventRecord(r->up_done_of[slot], r->up_stream) != hipSuccess) rc3 = EXG_E_HIP;
        trace_at("U upload enqueued, slot", (uint64_t)slot);
        static const bool lvl2 = getenv("EXG_TRACE") && atoi(getenv("EXG_TRACE")) >= 2;
        if (lvl2 && !rc3) {  // (when the bytes have landed: a timeline's question)
            (void)hipEventSynchronize(r->up_done_of[slot]);
            trace_at("U upload landed, slot", (uint64_t)slot);
        }
        r->up_rc_of[slot] = rc3;
    });
    which->valid = true;
    which->file_start = start;
    which->len = len;
    which->slot = slot;
}

}  // namespace exg_rd
