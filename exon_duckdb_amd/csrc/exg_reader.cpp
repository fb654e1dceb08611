// exg_reader.cpp — reader level of the C-ABI (include/exon_gpu.h, layer 2):
//   file on the host -> pinned host memory -> HBM -> scan kernels -> DataChunk-shaped host vectors.
//
// Replaces `new_reader` + the ArrowArrayStream it hands back (exon/include/rust.hpp:41-46,
// rust/src/arrow_reader.rs:38-166) and the per-batch pull in WTArrowTableFunction::Scan
// (exon/src/exon/arrow_table_function/module.cpp:257-294).  One reader per scan, one thread at a
// time, like the reference (MaxThreads() == 1).
//
// Data layout: the whole file is read into ONE pinned host block.  Device batches are record
// aligned (each starts on the first byte after the last complete record of the previous one, so
// EXG_F_BOF always holds); the kernels emit string_t whose pointers address the pinned block
// (payload_base = host address of the batch start), i.e. the DataChunk payload is zero-copy and only
// 64 B/record of string_t + validity cross PCIe on the way back.  Chunks are 2048-row slices of the
// batch's host vectors; buffers are reference counted until exg_release_chunk.
#include <dirent.h>
#include <pthread.h>
#include <sched.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "exg_filter.hpp"
#include "exg_reader.hpp"
#include "exg_zstd.hpp"

namespace exg_rd {

PinnedBlock::~PinnedBlock() {
    if (!p) return;
    if (mapped)
        munmap(p, mapped);
    else if (pooled)
        global_pool()->give((char *)p, cap);
    else
        (void)hipHostFree(p);
}

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
        sz = bytes ? bytes : 16;
        return p = exg_rd::dev_pool()->take(dev, sz);
    }
    ~PoolBuf() {
        if (!p) return;
        (void)hipStreamSynchronize(stream);
        exg_rd::dev_pool()->give(dev, p, sz);
    }
};

// DataFusion 28 FileCompressionType::from_str as used at rust/src/arrow_reader.rs:87-88
static bool parse_compression(const std::string &s, Compression *out) {
    std::string u;
    for (char ch : s) u.push_back((char)toupper((unsigned char)ch));
    if (u == "GZIP" || u == "GZ") return *out = kGzip, true;
    if (u == "ZSTD" || u == "ZST") return *out = kZstd, true;
    if (u == "BZIP2" || u == "BZ2") return *out = kBzip2, true;
    if (u == "XZ") return *out = kXz, true;
    if (u.empty()) return *out = kNone, true;
    return false;
}

}  // namespace exg_rd

// The reader's own I/O threads (pread into the pinned bounce buffers, H2D enqueue) run on the CPUs of the NUMA node the
// GPU hangs off (sysfs local_cpulist of its PCI function): the bounce buffers are local to the DMA engine, and a file
// that is read cold lands in that node's page cache.  Best effort — a cpuset that forbids it is not an error.
static void pin_to_device_node(int device) {
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
int exg_reader::join_zstd_check() {
    if (zst_check.joinable()) zst_check.join();
    if (!zst_check_rc) return EXG_OK;
    const int rc = zst_check_rc;
    zst_check_rc = 0;
    return exg_rd::fail(this, rc, zst_check_error);
}
exg_reader::~exg_reader() {
    exg_rd::DeviceGuard guard(device);
    if (zst_check.joinable()) zst_check.join();
    free_device();
    if (d_res) exg_rd::dev_pool()->give(device, d_res, 4096);
    if (d_phase) exg_rd::dev_pool()->give(device, d_phase, 4096);
    if (d_filter_prog) (void)hipFree(d_filter_prog);
    if (d_filter_consts) (void)hipFree(d_filter_consts);
    if (d_file) exg_rd::dev_pool()->give(device, d_file, d_file_cap);
    for (int k = 0; k < 2; k++)
        if (up_done_of[k]) (void)hipEventDestroy(up_done_of[k]);
    exg_rd::stream_pool()->give(device, up_stream);
    exg_rd::stream_pool()->give(device, stream);
}

namespace exg_rd {

int fail(exg_reader *r, int code, const std::string &msg) {
    r->error = msg;
    exg::set_error("%s", msg.c_str());
    return code;
}

#define RD_HIP(r, expr)                                                                            \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(r, EXG_E_HIP, std::string(#expr " failed: ") + hipGetErrorString(_e));     \
    } while (0)

int list_files(exg_reader *r, const std::string &path) {
    struct stat st;
    if (path.empty() || stat(path.c_str(), &st) != 0)
        return fail(r, EXG_E_IO, "could not register table: cannot open '" + path + "': " + strerror(errno));
    if (S_ISDIR(st.st_mode)) {
        // the reference lists a directory (test_fasta_scan.test:55-59, test_fastq_scan.test:65-68)
        DIR *d = opendir(path.c_str());
        if (!d) return fail(r, EXG_E_IO, "cannot list '" + path + "'");
        while (dirent *e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            std::string p = path + (path.back() == '/' ? "" : "/") + e->d_name;
            struct stat s2;
            if (stat(p.c_str(), &s2) == 0 && S_ISREG(s2.st_mode)) r->files.push_back(p);
        }
        closedir(d);
        std::sort(r->files.begin(), r->files.end());
    } else {
        r->files.push_back(path);
    }
    return EXG_OK;
}

static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
static bool trace_on() {
    static int on = getenv("EXG_TRACE") ? 1 : 0;
    return on;
}
#define TRACE(label, t0)                                                            \
    do {                                                                            \
        if (trace_on()) fprintf(stderr, "[exg] %-22s %.1f ms\n", label, (now_s() - (t0)) * 1e3); \
    } while (0)

static constexpr uint64_t kShardHaloBytes = 1u << 20;  // (= kShardHalo, which is defined with the batch logic below)

// gzip + VCF: the header is parsed on the host, so the leading '#' lines of the inflated bytes come back
// (blk->p then holds a prefix of the file, blk->n stays the inflated size; the DataChunk payload travels per batch)
static int gz_host_header(exg_reader *r, PinnedBlock &b, const void *d_file) {
    for (size_t want = 4u << 20;; want *= 8) {
        const size_t len = std::min<size_t>(want, b.n);
        if (b.p) global_pool()->give((char *)b.p, b.cap), b.p = nullptr;
        size_t cap = len + 64;
        b.p = global_pool()->take(&cap);
        if (!b.p) return fail(r, EXG_E_HIP, "out of pinned host memory for the VCF header");
        b.cap = cap;
        b.pooled = true;
        RD_HIP(r, hipMemcpyAsync(b.p, d_file, len, hipMemcpyDeviceToHost, r->stream));
        RD_HIP(r, hipStreamSynchronize(r->stream));
        // complete when a line that does not start with '#' begins inside the prefix (or the prefix is the file)
        const char *d = (const char *)b.p;
        size_t pos = 0;
        while (pos < len && d[pos] == '#') {
            const void *nl = memchr(d + pos, '\n', len - pos);
            pos = nl ? (size_t)((const char *)nl - d) + 1 : len;
        }
        if (pos < len || len == b.n) {
            r->gz_header_prefix = len;
            return EXG_OK;
        }
    }
}

// The whole (compressed) file -> d_dst on r->stream: windows of 256 MiB through two pooled pinned blocks, each window
// read by parallel pread and sent slice by slice (the mechanism of upload_range; a hipMemcpyAsync straight from the
// page-cache mapping is a pageable copy: one staging thread inside the runtime, 10-33 GB/s depending on the box).
static constexpr size_t kUploadWindow = 256u << 20;
// a consumer that follows the upload window by window (events recorded on the upload's stream)
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
static int upload_file(exg_reader *r, void *d_dst, uint64_t n, uint64_t file_off = 0, hipStream_t st = nullptr,
                       UploadProgress *prog = nullptr) {
    if (!st) st = r->stream;
    const size_t window = kUploadWindow, slice = 8u << 20;
    const int fd = r->fd_keep->fd;
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
    for (int k = 0; k < 2 && (uint64_t)k * window < n; k++) {
        cap[k] = (size_t)std::min<uint64_t>(window, n - (uint64_t)k * window) + 64;
        blk[k] = global_pool()->take(&cap[k]);
        if (!blk[k]) return fail(r, EXG_E_HIP, "out of pinned host memory");
        RD_HIP(r, hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    }
    uint64_t off = 0;
    for (uint64_t w = 0; off < n; w++) {
        const int b = (int)(w & 1);
        if (w >= 2) RD_HIP(r, hipEventSynchronize(ev[b]));  // the block's previous window has left
        const size_t len = (size_t)std::min<uint64_t>(window, n - off);
        const size_t n_slices = (len + slice - 1) / slice;
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        auto work = [&](bool own_thread) {
            (void)hipSetDevice(r->device);
            if (own_thread) pin_to_device_node(r->device);
            for (size_t i = next.fetch_add(1); i < n_slices; i = next.fetch_add(1)) {
                const size_t o = i * slice, sl = std::min<size_t>(slice, len - o);
                size_t got = 0;
                while (got < sl) {
                    ssize_t k = pread(fd, blk[b] + o + got, sl - got, (off_t)(file_off + off + o + got));
                    if (k <= 0) {
                        bad = 1;
                        return;
                    }
                    got += (size_t)k;
                }
                if (hipMemcpyAsync((char *)d_dst + off + o, blk[b] + o, sl, hipMemcpyHostToDevice, st) != hipSuccess) bad = 2;
            }
        };
        const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), std::min<size_t>(n_slices, 8));
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, true);
        work(false);
        for (auto &t : th) t.join();
        if (bad == 1) return fail(r, EXG_E_IO, "short read");
        if (bad == 2) return fail(r, EXG_E_HIP, "hipMemcpyAsync failed");
        RD_HIP(r, hipEventRecord(ev[b], st));
        if (prog) {
            RD_HIP(r, hipEventRecord(prog->done[w], st));
            std::lock_guard<std::mutex> g(prog->mu);
            prog->recorded = (size_t)w + 1;
            prog->cv.notify_all();
        }
        off += len;
    }
    return EXG_OK;
}

// ---- gzip trailers (RFC 1952 2.3.1): CRC-32 and ISIZE of every member, verified like flate2 / noodles-bgzf verify them ----
static uint32_t rd_le32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// Members [0, count) were inflated by exg_inflate_members(d_out, d_members, d_status) on r->stream: their checksums are
// computed on the device behind it and compared with the trailers in the compressed bytes on the host (`comp`; a
// member's comp_off is relative to comp + bias).  open_last: the last member ran to its own end (trailer behind the bytes
// it consumed).  Also returns the members' statuses (st).
static int check_members(exg_reader *r, const uint8_t *comp, uint64_t n_comp, uint64_t bias, const void *d_out, const exg_inflate_member *d_members,
                         const exg_inflate_status *d_status, const exg_inflate_member *h_members, uint64_t count, bool open_last,
                         std::vector<exg_inflate_status> &st, const std::string &path, const uint32_t *d_crc_ready = nullptr,
                         const uint32_t *h_crc_expect = nullptr) {
    if (!count) return EXG_OK;
    struct Pooled {
        int dev;
        void *p;
        size_t sz;
        ~Pooled() { if (p) exg_rd::dev_pool()->give(dev, p, sz); }
    };
    const size_t crc_bytes = ((count * 4 + 4095) & ~(size_t)4095) + (1u << 20);
    Pooled d_crc{r->device, d_crc_ready ? nullptr : exg_rd::dev_pool()->take(r->device, crc_bytes), crc_bytes};
    if (!d_crc_ready) {
        if (!d_crc.p) return fail(r, EXG_E_HIP, "out of device memory for the member checksums");
        int rc = exg_crc32_members(d_out, d_members, d_status, (uint32_t)count, (uint32_t *)d_crc.p, r->stream);
        if (rc) return fail(r, rc, exg_last_error_message());
    }
    st.resize(count);
    std::vector<uint32_t> crc(count);
    RD_HIP(r, hipMemcpyAsync(st.data(), d_status, count * sizeof(exg_inflate_status), hipMemcpyDeviceToHost, r->stream));
    RD_HIP(r, hipMemcpyAsync(crc.data(), d_crc_ready ? (const void *)d_crc_ready : d_crc.p, count * 4, hipMemcpyDeviceToHost, r->stream));
    RD_HIP(r, hipStreamSynchronize(r->stream));
    for (uint64_t i = 0; i < count; i++) {
        const bool open = open_last && i + 1 == count;
        if (st[i].code || (!open && st[i].produced != h_members[i].out_cap))
            return fail(r, EXG_E_PARSE, "corrupt deflate stream (member " + std::to_string(i) + ", code " + std::to_string(st[i].code) + ") in '" + path + "'");
        if (h_crc_expect && !open) {  // the index walk read the trailer already (ISIZE = out_cap, compared above)
            if (h_crc_expect[i] != crc[i])
                return fail(r, EXG_E_PARSE, "corrupt gzip stream does not have a matching checksum (member " + std::to_string(i) + " of '" + path + "')");
            continue;
        }
        const uint64_t trailer = bias + h_members[i].comp_off + (open ? st[i].consumed : h_members[i].comp_size - 8);
        if (trailer + 8 > n_comp) return fail(r, EXG_E_PARSE, "truncated gzip member (no trailer) in '" + path + "'");
        if (rd_le32(comp + trailer) != crc[i] || rd_le32(comp + trailer + 4) != (uint32_t)st[i].produced)
            return fail(r, EXG_E_PARSE, "corrupt gzip stream does not have a matching checksum (member " + std::to_string(i) + " of '" + path + "')");
    }
    return EXG_OK;
}

// one long output (exg_inflate_stream) against its trailer at comp[trailer]: 64 KiB segments on the device, combined here
static int check_stream(exg_reader *r, const uint8_t *comp, uint64_t n_comp, uint64_t trailer, const void *d_out, uint64_t produced, const std::string &path) {
    if (trailer + 8 > n_comp) return fail(r, EXG_E_PARSE, "truncated gzip member (no trailer) in '" + path + "'");
    const uint64_t seg = 65536, n_seg = (produced + seg - 1) / seg;
    uint32_t total = 0;  // crc32 of nothing
    if (n_seg) {
        std::vector<exg_crc_segment> segs(n_seg);
        for (uint64_t i = 0; i < n_seg; i++) segs[i] = exg_crc_segment{i * seg, std::min<uint64_t>(seg, produced - i * seg)};
        PoolBuf fr(r->device, r->stream);
        void *d = fr.take(n_seg * (sizeof(exg_crc_segment) + 4) + 64);
        if (!d) return fail(r, EXG_E_HIP, "out of device memory for the checksum segments");
        uint32_t *d_crc = (uint32_t *)((char *)d + n_seg * sizeof(exg_crc_segment));
        RD_HIP(r, hipMemcpyAsync(d, segs.data(), n_seg * sizeof(exg_crc_segment), hipMemcpyHostToDevice, r->stream));
        int rc = exg_crc32_segments(d_out, (const exg_crc_segment *)d, (uint32_t)n_seg, d_crc, r->stream);
        if (rc) return fail(r, rc, exg_last_error_message());
        std::vector<uint32_t> crc(n_seg);
        RD_HIP(r, hipMemcpyAsync(crc.data(), d_crc, n_seg * 4, hipMemcpyDeviceToHost, r->stream));
        RD_HIP(r, hipStreamSynchronize(r->stream));
        // fold: all segments but the last have the same length, so the multiplier x^(8 len) is the same one
        total = crc[0];
        for (uint64_t i = 1; i < n_seg; i++) total = exg_crc32_combine(total, crc[i], segs[i].len);
    }
    if (rd_le32(comp + trailer) != total || rd_le32(comp + trailer + 4) != (uint32_t)produced)
        return fail(r, EXG_E_PARSE, "corrupt gzip stream does not have a matching checksum ('" + path + "')");
    return EXG_OK;
}

// BGZF input read as shard `shard_index` of `shard_count`: the members are divided among the shards (the index costs a
// pointer chase, no decode), this reader uploads and inflates only its own members plus ~1 MiB of members in front of
// them — the halo that holds the beginning of the record that ends behind the cut — and scans them like a text shard.
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
    // file bytes [off, off + len), len <= 256; nullptr past the end of the file
    const uint8_t *at(uint64_t off, size_t len) {
        if (off + len > n) return nullptr;
        if (fd < 0) return map + off;
        if (off >= b0 && off + len <= b1) return buf + (off - b0);
        const size_t want = (size_t)std::min<uint64_t>(sizeof buf, n - off);
        size_t got = 0;
        while (got < want) {
            const ssize_t k = pread(fd, buf + got, want - got, (off_t)(off + got));
            if (k <= 0) break;
            got += (size_t)k;
        }
        if (got < len) return nullptr;
        b0 = off, b1 = off + got;
        return buf;
    }
};

// One BGZF member at `pos` (RFC 1952 header with FEXTRA and a 'BC' subfield, as bgzip / htslib write it):
// fills m (out_off = 0), *crc = the trailer's CRC-32, and returns the offset of the next member, or 0 when this is not such
// a header.
static uint64_t bgzf_member_at(Peek &f, uint64_t pos, exg_inflate_member *m, uint32_t *crc = nullptr) {
    const uint8_t *h = f.at(pos, 18);
    if (!h || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return 0;
    const uint64_t xlen = h[10] | ((uint64_t)h[11] << 8);
    if (xlen > 240 || !(h = f.at(pos, 12 + (size_t)xlen))) return 0;  // (bgzip writes 6; anything long is not BGZF to this walk)
    int64_t bsize = -1;
    for (uint64_t q = 12; q + 4 <= 12 + xlen;) {
        const uint64_t slen = h[q + 2] | ((uint64_t)h[q + 3] << 8);
        if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) bsize = h[q + 4] | ((int64_t)h[q + 5] << 8);
        q += 4 + slen;
    }
    if (bsize < 0) return 0;
    const uint64_t end = pos + (uint64_t)bsize + 1, p = pos + 12 + xlen;
    if (end > f.n || end < p + 8) return 0;
    // the trailer: the window read for it also holds the next member's header
    const uint8_t *t = f.at(end - 8, 8);
    if (!t) return 0;
    m->comp_off = p;
    m->comp_size = end - p;
    m->out_off = 0;
    m->out_cap = t[4] | ((uint64_t)t[5] << 8) | ((uint64_t)t[6] << 16) | ((uint64_t)t[7] << 24);
    if (crc) *crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    return m->out_cap <= 65536 ? end : 0;
}
// first member that starts at or after `from`: a header whose chain holds for four more members (or runs into the
// end of the file) — the signature alone also occurs inside compressed data
static uint64_t bgzf_find(const uint8_t *d, int fd, uint64_t n, uint64_t from) {
    Peek f(d, fd, n);
    uint8_t chunk[4096];
    for (uint64_t base = from; base + 18 <= n;) {
        // candidates: 0x1f bytes of the next 4 KiB
        const size_t len = (size_t)std::min<uint64_t>(sizeof chunk, n - base);
        const uint8_t *c = d + base;
        if (fd >= 0) {
            size_t got = 0;
            while (got < len) {
                const ssize_t k = pread(fd, chunk + got, len - got, (off_t)(base + got));
                if (k <= 0) return n;
                got += (size_t)k;
            }
            c = chunk;
        }
        for (size_t i = 0; i < len;) {
            const void *hit = memchr(c + i, 0x1f, len - i);
            if (!hit) break;
            const uint64_t pos = base + (uint64_t)((const uint8_t *)hit - c);
            if (pos + 18 > n) return n;
            exg_inflate_member m;
            uint64_t q = pos;
            int hops = 0;
            while (hops < 5 && q < n) {
                const uint64_t nx = bgzf_member_at(f, q, &m);
                if (!nx) break;
                q = nx;
                hops++;
            }
            if (hops == 5 || (hops > 0 && q == n)) return pos;
            i = (size_t)(pos - base) + 1;
        }
        base += len;
    }
    return n;
}

// Member index of a pure BGZF file by several host threads (the serial pointer chase through the page cache costs
// 110-150 ms per 10 GB: ~190 ns of cache misses per member): every thread finds a header near its cut, then walks to
// the next thread's start.  false: not (only) BGZF, or a walk did not land on its neighbour's start — the caller falls
// back to the serial RFC 1952 index.
// upto < n: only the members whose header begins in front of the first member that starts at or behind `upto` (the head of
// the file, for a decode that starts before the whole index is there).
static bool bgzf_parallel_index(const uint8_t *d, int fd, uint64_t n, exg_inflate_member *members, uint64_t cap, uint64_t *k_out,
                                uint64_t *total_out, std::vector<uint32_t> *crc_out, uint64_t upto = ~0ull) {
    exg_inflate_member probe;
    {
        Peek f(d, fd, n);
        if (!n || !bgzf_member_at(f, 0, &probe)) return false;
    }
    const uint64_t limit = upto >= n ? n : bgzf_find(d, fd, n, upto);
    // (a pread per member, ~5 us each here: the walk scales with its threads until the cores run out)
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(32u, hw), limit >> 24));  // 16 MiB per thread at least
    std::vector<uint64_t> starts(T + 1, limit);
    starts[0] = 0;
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++)
            th.emplace_back([&, t] { starts[t] = std::min(limit, bgzf_find(d, fd, n, (uint64_t)((unsigned __int128)limit * t / T))); });
        for (auto &x : th) x.join();
    }
    for (unsigned t = 1; t <= T; t++)
        if (starts[t] < starts[t - 1]) return false;
    std::vector<std::vector<exg_inflate_member>> parts(T);
    std::vector<std::vector<uint32_t>> crcs(T);
    std::vector<int> ok(T, 0);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t] {
                Peek f(d, fd, n);
                uint64_t pos = starts[t];
                auto &v = parts[t];
                auto &cv = crcs[t];
                v.reserve((size_t)((starts[t + 1] - starts[t]) / 8192 + 64));
                cv.reserve(v.capacity());
                while (pos < starts[t + 1]) {
                    exg_inflate_member m;
                    uint32_t crc = 0;
                    const uint64_t nx = bgzf_member_at(f, pos, &m, &crc);
                    if (!nx) return;
                    v.push_back(m);
                    cv.push_back(crc);
                    pos = nx;
                }
                ok[t] = pos == starts[t + 1];
            });
        for (auto &x : th) x.join();
    }
    uint64_t k = 0, out = 0;
    if (crc_out) crc_out->clear();
    for (unsigned t = 0; t < T; t++) {
        if (!ok[t] || k + parts[t].size() > cap) return false;
        for (auto &m : parts[t]) {
            m.out_off = out;
            out += m.out_cap;
            members[k++] = m;
        }
        if (crc_out) crc_out->insert(crc_out->end(), crcs[t].begin(), crcs[t].end());
    }
    *k_out = k;
    *total_out = out;
    return true;
}

// BGZF input read as shard `shard_index` of `shard_count`: a member belongs to the shard in whose 1/shard_count of the
// FILE's bytes its header begins.  The reader finds its members without indexing the file (a header search near the
// cut, then a walk through its own range: a pointer chase over the whole file costs 110 ms per 10 GB), uploads and
// inflates only them plus ~1 MiB of members in front — the halo that holds the beginning of the record that ends behind
// the cut — and scans them like a text shard.
static int inflate_file_shard(exg_reader *r, std::shared_ptr<PinnedBlock> &blk, const std::string &path) {
    const uint8_t *comp = (const uint8_t *)blk->p;
    const uint64_t n = blk->n;
    const int fd = r->fd_keep ? r->fd_keep->fd : -1;
    Peek peek(comp, fd, n);
    {
        exg_inflate_member probe;
        if (!bgzf_member_at(peek, 0, &probe))
            return fail(r, EXG_E_UNSUPPORTED, "shards of a gzip input need BGZF framing (every member carries its size): '" + path + "'");
    }
    const uint64_t lo = (uint64_t)((unsigned __int128)n * r->shard_index / r->shard_count);
    const uint64_t hi = r->shard_index + 1 == r->shard_count ? n : (uint64_t)((unsigned __int128)n * (r->shard_index + 1) / r->shard_count);
    // VCF: every rank needs the header (schema, and where the data begins): the leading members are inflated until the
    // '#' lines end; their text stays on the host as the file block's prefix, like in the unsharded gzip path
    auto out_blk = std::make_shared<PinnedBlock>();
    uint64_t header_members_end = 0;  // compressed offset behind the members that were needed for the header
    r->gz_header_prefix = 0;
    if (r->format == EXG_FMT_VCF) {
        for (uint64_t want = 16;; want *= 8) {
            std::vector<exg_inflate_member> hm;
            uint64_t q = 0, out = 0;
            while (q < n && hm.size() < want) {
                exg_inflate_member m;
                const uint64_t nx = bgzf_member_at(peek, q, &m);
                if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(q) + " of '" + path + "'");
                m.out_off = out;
                out += m.out_cap;
                hm.push_back(m);
                q = nx;
            }
            struct Pooled {
                int dev;
                void *p;
                size_t sz;
                ~Pooled() { if (p) exg_rd::dev_pool()->give(dev, p, sz); }
            };
            Pooled dc{r->device, exg_rd::dev_pool()->take(r->device, q + 64), (size_t)(q + 64)};
            Pooled dd{r->device, exg_rd::dev_pool()->take(r->device, out + 64), (size_t)(out + 64)};
            Pooled dm{r->device, exg_rd::dev_pool()->take(r->device, hm.size() * 56 + 64), hm.size() * 56 + 64};
            if (!dc.p || !dd.p || !dm.p) return fail(r, EXG_E_HIP, "out of device memory for the VCF header members");
            int rc0 = upload_file(r, dc.p, q, 0);
            if (rc0) return rc0;
            exg_inflate_status *d_st = (exg_inflate_status *)((char *)dm.p + hm.size() * sizeof(exg_inflate_member));
            RD_HIP(r, hipMemcpyAsync(dm.p, hm.data(), hm.size() * sizeof(exg_inflate_member), hipMemcpyHostToDevice, r->stream));
            rc0 = exg_inflate_members(dc.p, dd.p, (const exg_inflate_member *)dm.p, d_st, (uint32_t)hm.size(), r->stream);
            if (rc0) return fail(r, rc0, exg_last_error_message());
            {
                std::vector<exg_inflate_status> hst;
                if ((rc0 = check_members(r, comp, n, 0, dd.p, (const exg_inflate_member *)dm.p, d_st, hm.data(), hm.size(), false, hst, path))) return rc0;
            }
            if (out_blk->p) global_pool()->give((char *)out_blk->p, out_blk->cap), out_blk->p = nullptr;
            size_t cap = out + 64;
            out_blk->p = global_pool()->take(&cap);
            if (!out_blk->p) return fail(r, EXG_E_HIP, "out of pinned host memory for the VCF header");
            out_blk->cap = cap;
            out_blk->pooled = true;
            RD_HIP(r, hipMemcpyAsync(out_blk->p, dd.p, out, hipMemcpyDeviceToHost, r->stream));
            RD_HIP(r, hipStreamSynchronize(r->stream));
            const char *d = (const char *)out_blk->p;
            uint64_t hpos = 0;
            while (hpos < out && d[hpos] == '#') {
                const void *nl = memchr(d + hpos, '\n', (size_t)(out - hpos));
                hpos = nl ? (uint64_t)((const char *)nl - d) + 1 : out;
            }
            if (hpos < out || q >= n) {
                r->gz_header_prefix = out;
                header_members_end = q;
                break;
            }
        }
    }
    // candidates for the halo: members that begin in the ~1.5 MiB of file in front of the cut (BGZF does not expand)
    std::vector<exg_inflate_member> mem;
    std::vector<uint64_t> hdr;  // where each member's gzip header begins
    const uint64_t back = kShardHaloBytes + (kShardHaloBytes >> 1);
    uint64_t pos = lo == 0 ? 0 : bgzf_find(comp, fd, n, lo > back ? lo - back : 0);
    uint64_t m0 = 0;  // index in `mem` of the first own member
    bool seen_own = false;
    while (pos < hi && pos < n) {
        exg_inflate_member m;
        const uint64_t nx = bgzf_member_at(peek, pos, &m);
        if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(pos) + " of '" + path + "'");
        if (!seen_own && pos >= lo) {
            seen_own = true;
            m0 = mem.size();
        }
        mem.push_back(m);
        hdr.push_back(pos);
        pos = nx;
    }
    if (!seen_own) m0 = mem.size();
    uint64_t m1 = mem.size();
    // does any inflated byte follow this reader's members?  (the empty BGZF end marker — or a later shard that owns
    // nothing else — must not keep the shard with the file's last record from seeing the end of the file)
    bool bytes_follow = false;
    for (uint64_t q = pos; q < n && !bytes_follow;) {
        exg_inflate_member m;
        const uint64_t nx = bgzf_member_at(peek, q, &m);
        if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(q) + " of '" + path + "'");
        bytes_follow = m.out_cap != 0;
        q = nx;
    }
    // keep only ~1 MiB (inflated) of the members in front
    uint64_t h0 = m0, halo_bytes = 0;
    while (h0 > 0 && halo_bytes < kShardHaloBytes) halo_bytes += mem[--h0].out_cap;
    if (h0 < m1 && hdr[h0] != 0 && hdr[h0] < header_members_end) {
        // the halo would begin among the members that hold the VCF header: take everything from the start of the
        // file instead, so that the header's end is a known offset of this buffer
        std::vector<exg_inflate_member> front;
        std::vector<uint64_t> front_hdr;
        for (uint64_t q = 0; q < hdr[h0];) {
            exg_inflate_member m;
            const uint64_t nx = bgzf_member_at(peek, q, &m);
            if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(q) + " of '" + path + "'");
            front.push_back(m);
            front_hdr.push_back(q);
            q = nx;
        }
        mem.erase(mem.begin(), mem.begin() + (long)h0);
        hdr.erase(hdr.begin(), hdr.begin() + (long)h0);
        m0 -= h0;
        mem.insert(mem.begin(), front.begin(), front.end());
        hdr.insert(hdr.begin(), front_hdr.begin(), front_hdr.end());
        m0 += front.size();
        h0 = 0;
    }
    m1 = mem.size();
    const bool halo_from_file_start = h0 == 0 && !hdr.empty() && hdr[0] == 0;  // byte 0 of the inflated halo begins a line
    exg_inflate_member *members = mem.data();
    {
        uint64_t out = 0;
        for (uint64_t i = h0; i < m1; i++) {
            members[i].out_off = out;
            out += members[i].out_cap;
        }
    }
    int rc = 0;
    r->range_preset = true;
    r->preset_pos = 0;
    r->range_eof = !bytes_follow;
    r->data0_is_line_start = true;
    r->d_file = nullptr;
    r->d_file_bytes = 0;
    if (m1 == m0) {  // more shards than members: nothing here
        blk = out_blk;
        return EXG_OK;
    }
    const uint64_t c0 = hdr[h0];  // the gzip header of member h0
    const uint64_t c0a = c0 & ~15ull, c1 = members[m1 - 1].comp_off + members[m1 - 1].comp_size;
    const uint64_t out0 = members[h0].out_off, out_total = members[m1 - 1].out_off + members[m1 - 1].out_cap - out0;
    struct Pooled {
        int dev;
        void *p;
        size_t sz;
        ~Pooled() { if (p) exg_rd::dev_pool()->give(dev, p, sz); }
    };
    Pooled comp_buf{r->device, exg_rd::dev_pool()->take(r->device, c1 - c0a + 64), (size_t)(c1 - c0a + 64)};
    Pooled out_buf{r->device, exg_rd::dev_pool()->take(r->device, out_total + 64), (size_t)(out_total + 64)};
    if (!comp_buf.p || !out_buf.p) return fail(r, EXG_E_HIP, "out of device memory for the shard's members");
    if ((rc = upload_file(r, comp_buf.p, c1 - c0a, c0a))) return rc;
    const uint64_t cnt = m1 - h0;
    for (uint64_t i = h0; i < m1; i++) members[i].comp_off -= c0a, members[i].out_off -= out0;
    PoolBuf fm(r->device, r->stream), fs(r->device, r->stream);
    void *d_members = fm.take(cnt * sizeof(exg_inflate_member)), *d_status = fs.take(cnt * sizeof(exg_inflate_status));
    if (!d_members || !d_status) return fail(r, EXG_E_HIP, "out of device memory for the member table");
    RD_HIP(r, hipMemcpyAsync(d_members, members + h0, cnt * sizeof(exg_inflate_member), hipMemcpyHostToDevice, r->stream));
    rc = exg_inflate_members(comp_buf.p, out_buf.p, (const exg_inflate_member *)d_members, (exg_inflate_status *)d_status, (uint32_t)cnt,
                             r->stream);
    if (rc) return fail(r, rc, exg_last_error_message());
    std::vector<exg_inflate_status> st;
    RD_HIP(r, hipMemsetAsync((char *)out_buf.p + out_total, 0, 64, r->stream));
    if ((rc = check_members(r, comp, n, c0a, out_buf.p, (const exg_inflate_member *)d_members, (const exg_inflate_status *)d_status, members + h0, cnt,
                            false, st, path)))
        return rc;
    out_blk->n = out_total;
    blk = out_blk;
    r->d_file = out_buf.p;
    r->d_file_cap = out_buf.sz;
    out_buf.p = nullptr;  // owned by the reader now
    r->d_file_bytes = out_total;
    r->preset_pos = members[m0].out_off;  // (rebased) = inflated bytes of the halo members
    r->data0_is_line_start = halo_from_file_start;
    return EXG_OK;
}

// gzip input: H2D the compressed bytes, inflate every member on the device (exg_inflate.hip), keep the
// inflated bytes in HBM for the scan and bring one copy back for the DataChunk payload.
int inflate_file(exg_reader *r, std::shared_ptr<PinnedBlock> &blk, const std::string &path) {
    double t_all = now_s();
    struct TraceAll {
        double t0;
        ~TraceAll() { TRACE("gz: inflate_file total", t0); }
    } trace_all{t_all};
    const uint8_t *comp = (const uint8_t *)blk->p;
    const uint64_t n = blk->n;
    if (n == 0) return fail(r, EXG_E_PARSE, "empty gzip file '" + path + "'");
    if (r->shard_count > 1) return inflate_file_shard(r, blk, path);
    // the big device buffers (compressed bytes, inflated bytes) come from the device pool: a query that opens the
    // same file again finds them there (hipMalloc / hipFree of tens of GB were seen to cost up to 0.9 s per open)
    struct Pooled {
        int dev;
        void *p;
        size_t sz;
        ~Pooled() { if (p) exg_rd::dev_pool()->give(dev, p, sz); }
    };
    void *d_comp = exg_rd::dev_pool()->take(r->device, n + 64);
    if (!d_comp) return fail(r, EXG_E_HIP, "out of device memory for the compressed file");
    Pooled free_comp{r->device, d_comp, (size_t)(n + 64)};
    // worst case one member per 18 bytes; NOT value-initialised (a 0.5 GB file would zero 1 GB here: measured 240 ms)
    const uint64_t members_cap = std::max<uint64_t>(16, n / 18 + 4);
    std::unique_ptr<exg_inflate_member[]> members(new exg_inflate_member[members_cap]);
    // the member index of the first round (a pointer chase through the page cache: 110 ms per 10 GB of BGZF) is
    // made on a second host thread while the compressed bytes travel
    struct FirstIndex {
        uint64_t k = 0, total = 0;
        int open_ended = 0, rc = 0;
        std::string err;
        double ms = 0;
        std::vector<uint32_t> crc;  // BGZF walk: the members' trailer checksums (their ISIZE is out_cap)
    } first;
    std::thread index_thread([&] {
        double t0 = now_s();
        if (!bgzf_parallel_index(comp, r->fd_keep ? r->fd_keep->fd : -1, n, members.get(), members_cap, &first.k, &first.total, &first.crc)) {
            first.crc.clear();
            first.k = first.total = 0;
            first.rc = exg_gzip_index(comp, n, 0, members.get(), members_cap, &first.k, &first.total, &first.open_ended);
            if (first.rc) first.err = exg_last_error_message();  // the message is thread-local
        }
        first.ms = (now_s() - t0) * 1e3;
    });
    struct Joiner {
        std::thread *t;
        ~Joiner() { if (t->joinable()) t->join(); }
    } index_joiner{&index_thread};
    // the compressed bytes travel on a stream of their own, window by window, from a host thread of their own
    double t_h2d = now_s();
    hipStream_t up = nullptr;
    RD_HIP(r, exg_rd::stream_pool()->take(r->device, &up));
    UploadProgress prog;
    prog.done.resize((size_t)((n + kUploadWindow - 1) / kUploadWindow), nullptr);
    struct UpGuard {
        int dev;
        hipStream_t up;
        UploadProgress *prog;
        std::thread *th;
        ~UpGuard() {
            if (th->joinable()) th->join();
            (void)hipStreamSynchronize(up);
            for (hipEvent_t e : prog->done)
                if (e) (void)hipEventDestroy(e);
            exg_rd::stream_pool()->give(dev, up);
        }
    };
    std::thread up_thread;
    UpGuard up_guard{r->device, up, &prog, &up_thread};
    for (auto &e : prog.done) RD_HIP(r, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    up_thread = std::thread([&] {
        (void)hipSetDevice(r->device);
        pin_to_device_node(r->device);
        const int rc = upload_file(r, d_comp, n, 0, up, &prog);
        std::lock_guard<std::mutex> g(prog.mu);
        prog.rc = rc;
        prog.finished = true;
        prog.cv.notify_all();
    });
    uint64_t out_cap_total = 0, produced_total = 0;
    void *d_out = nullptr;
    uint64_t d_out_cap = 0;
    // whatever way this function is left, the output buffer goes back to the pool unless it became r->d_file
    struct OutGuard {
        int dev;
        void **p;
        uint64_t *cap;
        ~OutGuard() { if (*p) exg_rd::dev_pool()->give(dev, *p, (size_t)*cap); }
    } out_guard{r->device, &d_out, &d_out_cap};
    // A big file's index takes ~30 ms (a pread per member) and its first window is on the device after 5: the members of
    // that window are indexed on their own (1-2 ms), the output buffer is sized from their ratio (+ 25 %), and they are
    // inflated while the full index is still being made.  Should the file turn out larger than the estimate, what has been
    // inflated moves into a buffer of the right size (a device copy of one window's output).
    struct Head {
        std::vector<exg_inflate_member> m;
        uint64_t k = 0, total = 0, launched = 0;
        void *d_m = nullptr, *d_s = nullptr, *d_c = nullptr;
        PoolBuf bm, bs, bc;
        Head(int dev, hipStream_t s) : bm(dev, s), bs(dev, s), bc(dev, s) {}
    } head(r->device, r->stream);
    static const bool no_pipeline = getenv("EXG_NO_GZ_PIPELINE") != nullptr;
    if (!no_pipeline && n > 2 * (uint64_t)kUploadWindow) {
        // (the full index of a 5 GB file takes ~33 ms here, the upload moves a window in ~5: three windows keep the device busy
        // until the index is there)
        const uint64_t head_windows = std::min<uint64_t>(3, n / kUploadWindow - 1), head_bytes = head_windows * kUploadWindow;
        head.m.resize(head_bytes / 1024 + 64);  // (a member per KiB: anything denser is left to the full index)
        if (bgzf_parallel_index(comp, r->fd_keep ? r->fd_keep->fd : -1, n, head.m.data(), head.m.size(), &head.k, &head.total, nullptr, head_bytes) &&
            head.k) {
            const exg_inflate_member &last = head.m[head.k - 1];
            const uint64_t comp_bytes = last.comp_off + last.comp_size;
            // (EXG_GZ_HEAD_EST_PCT: the tests' way into the "estimate was short" path)
            static const double est_scale = getenv("EXG_GZ_HEAD_EST_PCT") ? atof(getenv("EXG_GZ_HEAD_EST_PCT")) / 100.0 : 1.25;
            const double est = (double)head.total / (double)comp_bytes * (double)n * est_scale + (est_scale >= 1.0 ? (double)(64u << 20) : 0.0);
            d_out_cap = std::max<uint64_t>((uint64_t)est, head.total) + 64;  // (what the head itself produces always fits)
            d_out = exg_rd::dev_pool()->take(r->device, d_out_cap);
            if (d_out && (head.d_m = head.bm.take(head.k * sizeof(exg_inflate_member))) && (head.d_s = head.bs.take(head.k * sizeof(exg_inflate_status))) &&
                (head.d_c = head.bc.take(head.k * 4 + 64))) {
                RD_HIP(r, hipMemcpyAsync(head.d_m, head.m.data(), head.k * sizeof(exg_inflate_member), hipMemcpyHostToDevice, r->stream));
                for (uint64_t w = 0; w < head_windows && prog.wait_for((size_t)w); w++) {
                    RD_HIP(r, hipStreamWaitEvent(r->stream, prog.done[w], 0));
                    const uint64_t ready = std::min<uint64_t>(n, (w + 1) * kUploadWindow);
                    uint64_t q = head.launched;
                    while (q < head.k && head.m[q].comp_off + head.m[q].comp_size <= ready) q++;
                    if (q > head.launched) {
                        const uint64_t q0 = head.launched;
                        int rc = exg_inflate_members(d_comp, d_out, (const exg_inflate_member *)head.d_m + q0, (exg_inflate_status *)head.d_s + q0,
                                                     (uint32_t)(q - q0), r->stream);
                        if (!rc)
                            rc = exg_crc32_members(d_out, (const exg_inflate_member *)head.d_m + q0, (const exg_inflate_status *)head.d_s + q0,
                                                   (uint32_t)(q - q0), (uint32_t *)head.d_c + q0, r->stream);
                        if (rc) return fail(r, rc, exg_last_error_message());
                        head.launched = q;
                    }
                }
            }
        }
    }
    index_thread.join();
    if (trace_on()) fprintf(stderr, "[exg] %-22s %.1f ms (beside the upload)\n", "gz: member index", first.ms);
    uint64_t start = 0;
    const bool pipeline = !first.rc && first.k && !first.open_ended && !no_pipeline;
    if (head.launched) {
        // does the head agree with the full index?  (it must: the same walk over the same bytes)
        bool same = pipeline && head.launched <= first.k;
        for (uint64_t i = 0; same && i < head.launched; i++)
            same = members[i].comp_off == head.m[i].comp_off && members[i].comp_size == head.m[i].comp_size && members[i].out_off == head.m[i].out_off &&
                   members[i].out_cap == head.m[i].out_cap;
        if (!same) {
            RD_HIP(r, hipStreamSynchronize(r->stream));
            head.launched = 0;
        }
    }
    if (!pipeline && d_out) {  // not (only) BGZF after all: the general path allocates for itself
        RD_HIP(r, hipStreamSynchronize(r->stream));
        exg_rd::dev_pool()->give(r->device, d_out, (size_t)d_out_cap);
        d_out = nullptr, d_out_cap = 0;
    }
    if (pipeline) {
        // BGZF: every member's place is known — the members of a window are inflated as soon as the window has arrived,
        // while the next windows are still on their way
        const uint64_t k = first.k;
        if (!d_out || first.total + 64 > d_out_cap) {
            const uint64_t cap2 = first.total + 64;
            void *p2 = exg_rd::dev_pool()->take(r->device, cap2);
            if (!p2) return fail(r, EXG_E_HIP, "out of device memory for the inflated file");
            if (d_out) {
                if (head.launched) {
                    const exg_inflate_member &lm = members[head.launched - 1];
                    RD_HIP(r, hipMemcpyAsync(p2, d_out, lm.out_off + lm.out_cap, hipMemcpyDeviceToDevice, r->stream));
                }
                RD_HIP(r, hipStreamSynchronize(r->stream));  // (the estimate was short: rare, and the old buffer leaves now)
                exg_rd::dev_pool()->give(r->device, d_out, (size_t)d_out_cap);
            }
            d_out = p2, d_out_cap = cap2;
        }
        PoolBuf fm(r->device, r->stream), fs(r->device, r->stream), fc(r->device, r->stream);
        void *d_members = fm.take(k * sizeof(exg_inflate_member)), *d_status = fs.take(k * sizeof(exg_inflate_status));
        void *d_crc_all = fc.take(k * 4 + 64);
        if (!d_members || !d_status || !d_crc_all) return fail(r, EXG_E_HIP, "out of device memory for the member table");
        RD_HIP(r, hipMemcpyAsync(d_members, members.get(), k * sizeof(exg_inflate_member), hipMemcpyHostToDevice, r->stream));
        uint64_t i0 = head.launched;
        if (i0) {  // the head's statuses and checksums take their places in the tables of the whole file
            RD_HIP(r, hipMemcpyAsync(d_status, head.d_s, i0 * sizeof(exg_inflate_status), hipMemcpyDeviceToDevice, r->stream));
            RD_HIP(r, hipMemcpyAsync(d_crc_all, head.d_c, i0 * 4, hipMemcpyDeviceToDevice, r->stream));
        }
        // A window holds ~3 900 members and the device holds 5 120 wavefronts: one launch does not fill it, and on ONE stream
        // the next window's launch waits for the stragglers of this one.  The windows therefore go round three streams
        // (the members are independent of each other); the reader's own stream waits for the other two at the end.
        struct Side {
            int dev;
            hipStream_t s[2] = {nullptr, nullptr};
            hipEvent_t tables = nullptr, done[2] = {nullptr, nullptr};
            ~Side() {
                for (int i = 0; i < 2; i++) {
                    if (done[i]) (void)hipEventDestroy(done[i]);
                    if (s[i]) {
                        (void)hipStreamSynchronize(s[i]);  // idle already unless this is an error return
                        exg_rd::stream_pool()->give(dev, s[i]);
                    }
                }
                if (tables) (void)hipEventDestroy(tables);
            }
        } side{r->device};
        bool fan = hipEventCreateWithFlags(&side.tables, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 2 && fan; i++)
            fan = exg_rd::stream_pool()->take(r->device, &side.s[i]) == hipSuccess && hipEventCreateWithFlags(&side.done[i], hipEventDisableTiming) == hipSuccess;
        if (fan) {
            RD_HIP(r, hipEventRecord(side.tables, r->stream));  // the member table (and the head's results) are on r->stream
            for (int i = 0; i < 2; i++) RD_HIP(r, hipStreamWaitEvent(side.s[i], side.tables, 0));
        }
        for (size_t w = 0; w < prog.done.size(); w++) {
            if (!prog.wait_for(w)) break;  // the upload failed: its error is reported below
            hipStream_t ws = fan && w % 3 ? side.s[w % 3 - 1] : r->stream;
            RD_HIP(r, hipStreamWaitEvent(ws, prog.done[w], 0));
            const uint64_t ready = std::min<uint64_t>(n, (uint64_t)(w + 1) * kUploadWindow);
            uint64_t i1 = i0;
            while (i1 < k && members[i1].comp_off + members[i1].comp_size <= ready) i1++;
            if (i1 > i0) {
                int rc = exg_inflate_members(d_comp, d_out, (const exg_inflate_member *)d_members + i0, (exg_inflate_status *)d_status + i0,
                                             (uint32_t)(i1 - i0), ws);
                // ... and their checksums right behind them, while the next windows still travel
                if (!rc)
                    rc = exg_crc32_members(d_out, (const exg_inflate_member *)d_members + i0, (const exg_inflate_status *)d_status + i0,
                                           (uint32_t)(i1 - i0), (uint32_t *)d_crc_all + i0, ws);
                if (rc) return fail(r, rc, exg_last_error_message());
            }
            i0 = i1;
        }
        if (fan)
            for (int i = 0; i < 2; i++) {
                RD_HIP(r, hipEventRecord(side.done[i], side.s[i]));
                RD_HIP(r, hipStreamWaitEvent(r->stream, side.done[i], 0));
            }
        up_thread.join();
        if (prog.rc) return prog.rc;
        if (i0 < k) return fail(r, EXG_E_PARSE, "truncated gzip member in '" + path + "'");
        std::vector<exg_inflate_status> st;
        int crc_rc = check_members(r, comp, n, 0, d_out, (const exg_inflate_member *)d_members, (const exg_inflate_status *)d_status, members.get(), k,
                                   false, st, path, (const uint32_t *)d_crc_all, first.crc.size() == k ? first.crc.data() : nullptr);
        TRACE("gz: h2d + inflate + crc32", t_h2d);
        if (crc_rc) return crc_rc;
        produced_total = out_cap_total = first.total;
        start = n;
    } else {
        up_thread.join();
        if (prog.rc) return prog.rc;
        RD_HIP(r, hipStreamSynchronize(up));
        TRACE("gz: h2d compressed", t_h2d);
    }
    while (start < n) {
        uint64_t k = 0, total = produced_total;
        int open_ended = 0, rc = 0;
        if (start == 0) {
            k = first.k, total = first.total, open_ended = first.open_ended, rc = first.rc;
            if (rc) return fail(r, rc, first.err + " in '" + path + "'");
        } else {
            double t_idx = now_s();
            rc = exg_gzip_index(comp, n, start, members.get(), members_cap, &k, &total, &open_ended);
            TRACE("gz: member index", t_idx);
            if (rc) return fail(r, rc, std::string(exg_last_error_message()) + " in '" + path + "'");
        }
        if (k == 0) break;
        // one big member of unknown size (what gzip / pigz write): per-member parallelism would put the whole file
        // on ONE wavefront — decode it in chunks instead (exg_inflate_stream.hip)
        static const uint64_t stream_min = getenv("EXG_STREAM_MIN_BYTES") ? strtoull(getenv("EXG_STREAM_MIN_BYTES"), nullptr, 10) : (128ull << 10);  // (one wavefront does ~13 MB/s: 4 MB took 0.3 s)
        const bool stream_ok = !getenv("EXG_NO_STREAM_INFLATE");
        uint64_t resume = 0;  // != 0: where the next round starts (the gzip header of a member left out of this one)
        if (k > 1 && open_ended && stream_ok && members[k - 1].comp_size >= stream_min) {
            // sized members followed by a big one of unknown size: these first, the big one in a round of its own
            k--;
            open_ended = 0;
            total = members[k].out_off;                                 // the sum up to the member left out
            resume = members[k - 1].comp_off + members[k - 1].comp_size;  // sized member: its end is the next header
        }
        if (k == 1 && open_ended && stream_ok && members[0].comp_size >= stream_min) {
            // at most one piece per decoding wavefront the chip holds (the symbol decoder: 4 per SIMD = 4096; a few more
            // pieces than slots would cost a second round for them alone: 4756 pieces took 43 ms, 3830 take 35), at least 32 KiB each
            // (a block is 20-60 KB of input): a small file's decode lasts as long as one piece
            uint64_t chunk = std::max<uint64_t>(32u << 10, (members[0].comp_size / 3900 + 16383) & ~16383ull);
            if (getenv("EXG_STREAM_CHUNK_BYTES")) chunk = strtoull(getenv("EXG_STREAM_CHUNK_BYTES"), nullptr, 10);
            uint64_t produced = 0, consumed = 0;
            void *d_big = nullptr;
            rc = exg_inflate_stream(d_comp, members[0].comp_off, members[0].comp_size, chunk, &d_big, &produced, &consumed, r->stream);
            if (rc) return fail(r, rc, std::string(exg_last_error_message()) + " in '" + path + "'");
            if ((rc = check_stream(r, comp, n, members[0].comp_off + consumed, d_big, produced, path))) {
                exg_rd::dev_pool()->give(r->device, d_big, produced + 64);
                return rc;
            }
            if (!d_out) {
                d_out = d_big;
                d_out_cap = produced + 64;
                produced_total = produced;
            } else {
                // a later member of a concatenation (`cat a.gz b.gz`): its bytes go behind what is there
                const uint64_t ncap = produced_total + produced + 64;
                void *nd = exg_rd::dev_pool()->take(r->device, ncap);
                hipError_t he = nd ? hipMemcpyAsync(nd, d_out, produced_total, hipMemcpyDeviceToDevice, r->stream) : hipErrorOutOfMemory;
                if (he == hipSuccess) he = hipMemcpyAsync((char *)nd + produced_total, d_big, produced, hipMemcpyDeviceToDevice, r->stream);
                if (he == hipSuccess) he = hipStreamSynchronize(r->stream);
                exg_rd::dev_pool()->give(r->device, d_big, produced + 64);
                if (he != hipSuccess) {
                    if (nd) exg_rd::dev_pool()->give(r->device, nd, ncap);
                    return fail(r, EXG_E_HIP, std::string("appending the inflated member failed: ") + hipGetErrorString(he));
                }
                exg_rd::dev_pool()->give(r->device, d_out, d_out_cap);
                d_out = nd;
                d_out_cap = ncap;
                produced_total += produced;
            }
            out_cap_total = produced_total;
            start = members[0].comp_off + consumed + 8;
            continue;
        }
        out_cap_total = total;
        if (out_cap_total + 64 > d_out_cap) {  // grow the output (members of earlier rounds are kept)
            uint64_t ncap = out_cap_total + 64;
            void *nd = exg_rd::dev_pool()->take(r->device, ncap);
            if (!nd) return fail(r, EXG_E_HIP, "out of device memory for the inflated file");
            if (d_out) {
                hipError_t he = hipMemcpyAsync(nd, d_out, produced_total, hipMemcpyDeviceToDevice, r->stream);
                if (he == hipSuccess) he = hipStreamSynchronize(r->stream);
                exg_rd::dev_pool()->give(r->device, d_out, d_out_cap);
                d_out = nullptr;
                if (he != hipSuccess) {
                    exg_rd::dev_pool()->give(r->device, nd, ncap);
                    return fail(r, EXG_E_HIP, std::string("copy of the inflated bytes failed: ") + hipGetErrorString(he));
                }
            }
            d_out = nd;
            d_out_cap = ncap;
        }
        PoolBuf fm(r->device, r->stream), fs(r->device, r->stream);
        void *d_members = fm.take(k * sizeof(exg_inflate_member)), *d_status = fs.take(k * sizeof(exg_inflate_status));
        if (!d_members || !d_status) return fail(r, EXG_E_HIP, "out of device memory for the member table");
        RD_HIP(r, hipMemcpyAsync(d_members, members.get(), k * sizeof(exg_inflate_member), hipMemcpyHostToDevice, r->stream));
        double t_inf = now_s();
        rc = exg_inflate_members(d_comp, d_out, (const exg_inflate_member *)d_members, (exg_inflate_status *)d_status,
                                 (uint32_t)k, r->stream);
        if (rc) return fail(r, rc, exg_last_error_message());
        std::vector<exg_inflate_status> st;
        rc = check_members(r, comp, n, 0, d_out, (const exg_inflate_member *)d_members, (const exg_inflate_status *)d_status, members.get(), k,
                           open_ended != 0, st, path);
        TRACE("gz: inflate members + crc32", t_inf);
        if (rc) return rc;
        if (open_ended) {
            // the last member ran to its own end: compact its output, continue after its 8-byte trailer
            const exg_inflate_member &m = members[k - 1];
            produced_total = m.out_off + st[k - 1].produced;
            start = m.comp_off + st[k - 1].consumed + 8;
        } else {
            produced_total = out_cap_total;
            start = resume ? resume : n;
        }
    }
    // The inflated bytes stay in HBM.  What the string_t payload pointers address is a host copy made batch by
    // batch (next_batch: one pooled pinned block per device batch, kept alive by its chunks) — COUNT(*) and the
    // Arrow stream never need one; the block handed back here only knows the inflated size.
    auto out_blk = std::make_shared<PinnedBlock>();
    out_blk->n = produced_total;
    if (d_out) RD_HIP(r, hipMemsetAsync((char *)d_out + produced_total, 0, 64, r->stream));
    RD_HIP(r, hipStreamSynchronize(r->stream));
    blk = out_blk;
    r->d_file = d_out;
    r->d_file_cap = d_out_cap;
    d_out = nullptr;  // owned by the reader from here on (gz_host_header below reads r->d_file)
    r->d_file_bytes = produced_total;
    r->gz_header_prefix = 0;
    if (r->format == EXG_FMT_VCF && produced_total) {
        int rc = gz_host_header(r, *blk, r->d_file);
        if (rc) return rc;
    }
    return EXG_OK;
}

// zstd input (.zst, compression='zstd'): H2D the compressed bytes, decode every frame on the device (exg_zstd.hip: the
// host only walks the frame / block headers of the mapped file), keep the bytes in HBM for the scan — the rest of the
// reader treats them exactly like an inflated gzip file (r->d_file).
int zstd_file(exg_reader *r, std::shared_ptr<PinnedBlock> &blk, const std::string &path) {
    double t_all = now_s();
    const uint64_t n = blk->n;
    struct Pooled {
        int dev;
        void *p;
        size_t sz;
        ~Pooled() { if (p) exg_rd::dev_pool()->give(dev, p, sz); }
    };
    Pooled comp{r->device, exg_rd::dev_pool()->take(r->device, n + 64), (size_t)(n + 64)};
    if (!comp.p) return fail(r, EXG_E_HIP, "out of device memory for the compressed file");
    // the host's walk over the frame / block headers runs beside the upload
    exg::zst::Index idx;
    bool idx_ok = false;
    std::thread idx_thread([&] { idx_ok = exg::zst::build_index((const uint8_t *)blk->p, n, idx); });
    int up_rc = n ? upload_file(r, comp.p, n, 0) : EXG_OK;
    idx_thread.join();
    if (up_rc) return up_rc;
    if (!idx_ok) return fail(r, EXG_E_PARSE, idx.error + " in '" + path + "'");
    RD_HIP(r, hipMemsetAsync((char *)comp.p + n, 0, 64, r->stream));
    void *d_out = nullptr;
    uint64_t produced = 0;
    std::vector<exg::zst::PendingCheck> pending;
    int rc = exg::zst::decode((const uint8_t *)blk->p, comp.p, n, &d_out, &produced, r->stream, &pending, &idx);
    if (rc) return fail(r, rc, std::string(exg_last_error_message()) + " in '" + path + "'");
    TRACE("zstd: h2d + decode", t_all);
    if (!pending.empty()) {
        r->zst_check = std::thread([r, pending, d_out, path]() {
            std::string err;
            const int vrc = exg::zst::host_verify(d_out, pending, r->device, &err);
            if (vrc) {
                r->zst_check_error = err + " in '" + path + "'";
                r->zst_check_rc = vrc;
            }
        });
    }
    auto out_blk = std::make_shared<PinnedBlock>();
    out_blk->n = produced;
    blk = out_blk;
    r->d_file = d_out;
    r->d_file_cap = produced + 64;
    r->d_file_bytes = produced;
    r->gz_header_prefix = 0;
    if (r->format == EXG_FMT_VCF && produced) {
        rc = gz_host_header(r, *blk, r->d_file);
        if (rc) return rc;
    }
    return EXG_OK;
}

int open_next_file(exg_reader *r) {
    if (int jrc = r->join_zstd_check()) return jrc;
    const std::string &p = r->files[r->file_idx++];
    double t_all = now_s();
    int fd = open(p.c_str(), O_RDONLY);
    if (fd < 0) return fail(r, EXG_E_IO, "cannot open '" + p + "': " + strerror(errno));
    struct stat st;
    fstat(fd, &st);
    // The file is mapped, not copied: DataChunk strings point straight into the page cache mapping
    // (kept alive by the chunks); bytes travel to the device through a pinned bounce buffer.
    auto blk = std::make_shared<PinnedBlock>();
    blk->n = (size_t)st.st_size;
    double t0 = now_s();
    if (blk->n) {
        void *m = mmap(nullptr, blk->n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            close(fd);
            return fail(r, EXG_E_IO, "cannot map '" + p + "': " + strerror(errno));
        }
        blk->p = m;
        blk->mapped = blk->n;
    } else {
        hipError_t he = hipHostMalloc(&blk->p, 64, hipHostMallocDefault);
        if (he != hipSuccess) {
            close(fd);
            return fail(r, EXG_E_HIP, std::string("hipHostMalloc failed: ") + hipGetErrorString(he));
        }
        memset(blk->p, 0, 64);
    }
    r->fd_keep.reset(new exg_reader::FdCloser{fd});
    TRACE("mmap(file)", t0);
    (void)t_all;
    if (r->d_file) exg_rd::dev_pool()->give(r->device, r->d_file, r->d_file_cap), r->d_file = nullptr;
    r->range_preset = false;
    r->range_eof = true;
    r->data0_is_line_start = true;
    if (r->compression == kGzip) {
        int rc = inflate_file(r, blk, p);  // replaces blk by the inflated bytes (host copy) and sets d_file
        if (rc) return rc;
    } else if (r->compression == kZstd) {
        int rc = zstd_file(r, blk, p);
        if (rc) return rc;
    }
    (void)r->join_prefetch();
    r->drop_prefetch2();
    if (r->pf.valid && r->up_stream) (void)hipStreamSynchronize(r->up_stream);  // a prefetch of the previous file
    r->pf.valid = false;
    r->file = blk;
    r->file_pos = 0;
    r->file_done = false;
    if (r->format == EXG_FMT_VCF) {
        // header = the leading '#' lines (noodles-vcf read_header); it must hold the #CHROM line
        const char *d = (const char *)blk->p;
        const size_t hn = r->compression != kNone ? (size_t)r->gz_header_prefix : blk->n;  // gzip / zstd: only the header prefix is on the host
        size_t pos = 0;
        bool chrom = false;
        while (pos < hn && d[pos] == '#') {
            if (hn - pos >= 6 && memcmp(d + pos, "#CHROM", 6) == 0) chrom = true;
            const void *nl = memchr(d + pos, '\n', hn - pos);
            pos = nl ? (size_t)((const char *)nl - d) + 1 : hn;
        }
        if (!chrom) return fail(r, EXG_E_PARSE, std::string(exg_parse_error_string(EXG_PE_VCF_NO_HEADER)) + " in '" + p + "'");
        r->vcf_header_bytes = pos;
        r->file_pos = pos;
    }
    // byte-range shard of this file: [lo, hi) of the bytes behind the header; records / lines belong to the shard they END in
    r->range_hi = blk->n;
    r->shard_first = false;
    r->data_base = r->file_pos;  // 0, or the end of the VCF header
    if (r->range_preset) {  // BGZF shard: the members were chosen in inflate_file_shard
        // its buffer begins with the file (header and all) or somewhere behind the header
        r->data_base = r->data0_is_line_start ? r->data_base : 0;
        r->file_pos = std::max<uint64_t>(r->preset_pos, r->data_base);
        r->shard_first = r->file_pos > r->data_base;
    } else if (r->shard_count > 1 && r->format == EXG_FMT_FASTA) {
        // FASTA: a record belongs to the shard in whose bytes its '>' line BEGINS, and a shard is the run of whole
        // records from its first such line to the next shard's — scanned like a file of its own (a record is never
        // cut, however long its sequence: the run simply reaches as far as it has to)
        const char *d = (const char *)blk->p;
        const uint64_t N = blk->n;
        auto first_record_at_or_after = [&](uint64_t pos) -> uint64_t {
            if (pos == 0) return 0;
            for (uint64_t q = pos - 1; q + 1 < N;) {  // a line start is the byte behind a newline
                const void *hit = memchr(d + q, '\n', (size_t)(N - q));
                if (!hit) return N;
                q = (uint64_t)((const char *)hit - d) + 1;
                if (q < N && d[q] == '>') return q;
            }
            return N;
        };
        const uint64_t lo = (uint64_t)((unsigned __int128)N * r->shard_index / r->shard_count);
        const uint64_t hi = r->shard_index + 1 == r->shard_count ? N : (uint64_t)((unsigned __int128)N * (r->shard_index + 1) / r->shard_count);
        r->file_pos = first_record_at_or_after(lo);
        r->range_hi = hi == N ? N : first_record_at_or_after(hi);
        if (r->range_hi < r->file_pos) r->range_hi = r->file_pos;
        r->range_eof = true;  // the run is a FASTA file of its own
    } else if (r->shard_count > 1) {
        const uint64_t base = r->file_pos, span = blk->n - base;
        const uint64_t lo = base + (uint64_t)((unsigned __int128)span * r->shard_index / r->shard_count);
        const uint64_t hi = r->shard_index + 1 == r->shard_count
                                ? blk->n
                                : base + (uint64_t)((unsigned __int128)span * (r->shard_index + 1) / r->shard_count);
        r->file_pos = lo;
        r->range_hi = hi;
        r->shard_first = lo > base;
        r->range_eof = hi == blk->n;
    }
    return EXG_OK;
}

// The next batch starts where this one's last complete record ends - known only after the scan - so the
// prefetch starts this many bytes before the end of the current batch; a batch whose unconsumed tail is
// longer (one giant record) falls back to the synchronous upload.
static constexpr uint64_t kPrefetchSlack = 1u << 20;
// bytes in front of a shard that travel with its first batch (the beginning of the record that ends behind the cut)
static constexpr uint64_t kShardHalo = 1u << 20;
static_assert(kShardHalo == kShardHaloBytes, "one halo size");

int n_string_cols(int format) { return format == EXG_FMT_FASTQ ? 4 : format == EXG_FMT_FASTA ? 3 : 9; }

int ensure_device(exg_reader *r, uint64_t need_bytes) {
    if (r->d_in && need_bytes <= r->d_in_cap) return EXG_OK;
    if (r->d_in) {
        RD_HIP(r, hipStreamSynchronize(r->stream));
        r->free_device();
    }
    if (r->format != EXG_FMT_FASTA && r->file)  // room for a prefetched batch (its slack included), small files stay small
        need_bytes = std::max<uint64_t>(need_bytes, std::min<uint64_t>(r->device_batch_bytes, r->file->n) + kPrefetchSlack + 64);
    uint64_t cap = std::max<uint64_t>(need_bytes, 1 << 16);
    r->d_in_cap = cap;
    // Rows the output vectors can hold.  Realistic density first (a FASTQ record under 32 bytes, a FASTA record
    // or a VCF line under 16 would be unusual) — the worst case (FASTQ "@\n\n+\n" = 5 bytes, FASTA ">a\n" minus
    // LF, a blank VCF line) would pin 16 B x 9 columns per input BYTE of device memory; a batch that does
    // overflow is reported by the kernels (EXG_RF_CAPACITY) and rescanned with worst-case vectors.
    const uint64_t div = r->worst_case_rows ? (r->format == EXG_FMT_FASTQ ? 5 : r->format == EXG_FMT_FASTA ? 2 : 1)
                                            : (r->format == EXG_FMT_FASTQ ? 32 : 16);
    r->cap_records = cap / div + 4096;
    r->ws_bytes = exg_scan_workspace_bytes(r->format, cap);
    // FASTA scans the whole file as one batch: one slot
    int arc = 0;
    for (int k = 0; k < (r->format == EXG_FMT_FASTA ? 1 : 2); k++)
        if ((arc = r->dev_alloc(&r->d_in_slot[k], cap + 64))) return arc;
    r->d_in = r->d_in_slot[0];
    r->cur_slot = 0;
    if (!r->up_stream) {
        RD_HIP(r, exg_rd::stream_pool()->take(r->device, &r->up_stream));
        for (int k = 0; k < 2; k++) RD_HIP(r, hipEventCreateWithFlags(&r->up_done_of[k], hipEventDisableTiming));
    }
    if ((arc = r->dev_alloc(&r->d_ws, r->ws_bytes))) return arc;
    for (int k = 0; k < 2; k++)
        if ((arc = r->dev_alloc(&r->d_valid[k], (r->cap_records + 63) / 64 * 8))) return arc;
    for (int c = 0; c < n_string_cols(r->format); c++)
        if ((arc = r->dev_alloc(&r->d_cols[c], r->cap_records * 16))) return arc;
    if (r->format == EXG_FMT_VCF) {
        if ((arc = r->dev_alloc(&r->d_pos, r->cap_records * 8))) return arc;
        if ((arc = r->dev_alloc(&r->d_qual, r->cap_records * 4))) return arc;
    }
    if (r->format == EXG_FMT_FASTA && (arc = r->dev_alloc(&r->d_payload, cap + 64))) return arc;
    if (r->has_filter) {
        if ((arc = r->dev_alloc(&r->d_row_map, r->cap_records * 4 + 64))) return arc;
        if ((arc = r->dev_alloc(&r->d_gather, r->cap_records * 16))) return arc;
        if ((arc = r->dev_alloc(&r->d_filter_tmp, (r->cap_records + 1 + exg::arrow::scan_tmp_entries(r->cap_records)) * 8))) return arc;
    }
    if (!r->d_res && !(r->d_res = exg_rd::dev_pool()->take(r->device, 4096))) return fail(r, EXG_E_HIP, "out of device memory");
    return EXG_OK;
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
    static const size_t slice = getenv("EXG_IO_SLICE_MB") ? ((size_t)std::max(1, atoi(getenv("EXG_IO_SLICE_MB"))) << 20) : (8u << 20);
    const size_t n_slices = (n + slice - 1) / slice;
    static const size_t max_io_threads = getenv("EXG_IO_THREADS") ? (size_t)std::max(1, atoi(getenv("EXG_IO_THREADS"))) : 8;
    unsigned nt = (unsigned)std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), std::min<size_t>(n_slices, max_io_threads));
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    const int fd = r->fd_keep->fd;
    char *dst = (char *)stg.p;
    char *d_dst = (char *)r->d_in_slot[slot];
    auto work = [&](bool own_thread) {
        (void)hipSetDevice(r->device);
        if (own_thread) pin_to_device_node(r->device);
        for (size_t i = next.fetch_add(1); i < n_slices; i = next.fetch_add(1)) {
            size_t o = i * slice, len = std::min<size_t>(slice, n - o), got = 0;
            while (got < len) {
                ssize_t k = pread(fd, dst + o + got, len - got, (off_t)(off + o + got));
                if (k <= 0) {
                    bad = 1;
                    return;
                }
                got += (size_t)k;
            }
            size_t len16 = len;
            if (i + 1 == n_slices) {
                memset(dst + n, 0, padded - n);
                len16 = padded - o;
            }
            if (hipMemcpyAsync(d_dst + o, dst + o, len16, hipMemcpyHostToDevice, st) != hipSuccess) bad = 2;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, true);
    work(false);
    for (auto &t : th) t.join();
    if (bad == 1) return fail(r, EXG_E_IO, "short read");
    if (bad == 2) return fail(r, EXG_E_HIP, "hipMemcpyAsync failed");
    TRACE("pread + h2d enqueue", t0);
    return EXG_OK;
}

// an upload of file bytes [start, start + len) into input slot `slot`, on a host thread of its own: pread + the H2D enqueue
// block their caller for as long as the bytes take to leave (5.5 ms per 256 MiB)
static void start_upload(exg_reader *r, exg_reader::Prefetch *which, uint64_t start, uint64_t len, int slot) {
    r->up_rc_of[slot] = 0;
    r->up_thread_of[slot] = std::thread([r, start, len, slot] {
        (void)hipSetDevice(r->device);
        pin_to_device_node(r->device);
        int rc3 = upload_range(r, start, len, slot, r->up_stream);
        if (!rc3 && hipEventRecord(r->up_done_of[slot], r->up_stream) != hipSuccess) rc3 = EXG_E_HIP;
        r->up_rc_of[slot] = rc3;
    });
    which->valid = true;
    which->file_start = start;
    which->len = len;
    which->slot = slot;
}

// Scan the next device batch of the current file.  On return r->batch holds its host vectors
// (n_rows may be 0 when the file is exhausted).  count_only: no column leaves the device.
int next_batch(exg_reader *r, bool count_only, uint64_t *n_records_out) {
    *n_records_out = 0;
    r->batch.reset();
    r->batch_row = 0;
    uint64_t want = r->device_batch_bytes;
    double t_batch = now_s();
    for (;;) {
        const uint64_t remaining = r->range_hi > r->file_pos ? r->range_hi - r->file_pos : 0;
        if (remaining == 0) {
            r->file_done = true;
            return EXG_OK;
        }
        if (r->format == EXG_FMT_FASTA) want = remaining;  // a FASTA record can span the whole file: one batch
        uint64_t n = std::min<uint64_t>(want, remaining);
        bool range_end = n == remaining;                    // the batch reaches the end of this reader's bytes ...
        bool eof = range_end && r->range_eof;               // ... which is the end of the file unless a later shard follows
        // first batch of a shard that begins inside the file: up to 1 MiB in front of it travels along (`lead`), so that
        // the record / line that ends behind the cut — it belongs to this shard — has its beginning in the buffer
        uint64_t shard_halo = 0;
        if (r->shard_first) {
            static const uint64_t halo_max = getenv("EXG_SHARD_HALO") ? strtoull(getenv("EXG_SHARD_HALO"), nullptr, 10) : kShardHalo;
            const uint64_t base = r->data_base;
            const uint64_t from = r->file_pos - std::min<uint64_t>(halo_max, r->file_pos - base);
            // (a buffer that already lives in HBM must be entered at a 16-byte boundary: a few bytes of the header's
            // last line may then come along in front — they end inside the halo and are nobody's rows)
            shard_halo = r->file_pos - (r->d_file ? (std::max<uint64_t>(base, from) & ~15ull) : std::max<uint64_t>(base, from & ~15ull));
        }
        int rc = ensure_device(r, n + shard_halo + 16);
        if (rc) return rc;
        // Input of the scan: the inflated bytes already in HBM (gzip), the prefetched slot, or a synchronous
        // H2D copy.  In the first two cases the batch start is only byte aligned: the buffer starts at the
        // 16-byte boundary below it and `lead` skips the tail of the previous record (whose last '\n' is
        // then inside the buffer).
        // The batch about to be scanned is on its way (or there); if its upload is the one this call will use, the batch
        // AFTER it starts travelling now, into the slot of the batch that was scanned last (free: its columns have left) —
        // issued after this call's scan, an upload began only when the link had already been idle for a scan + a D2H.
        static const bool no_prefetch = getenv("EXG_NO_PREFETCH") != nullptr;
        if (!no_prefetch && r->pf.valid && !r->pf2.valid && !r->d_file && r->format != EXG_FMT_FASTA && want == r->device_batch_bytes &&
            r->file_pos >= r->pf.file_start && r->file_pos < r->pf.file_start + r->pf.len && r->d_in_slot[r->pf.slot ^ 1] &&
            !r->up_thread_of[r->pf.slot ^ 1].joinable()) {
            const uint64_t end1 = r->pf.file_start + r->pf.len;  // where the coming batch's bytes end
            if (end1 < r->range_hi) {
                const uint64_t slack = std::min<uint64_t>(kPrefetchSlack, (end1 - r->file_pos) / 2);
                const uint64_t start = (end1 - slack) & ~15ull;
                const uint64_t len = std::min<uint64_t>(r->range_hi - start, r->device_batch_bytes + slack);
                if (len + 16 <= r->d_in_cap) start_upload(r, &r->pf2, start, len, r->pf.slot ^ 1);
            }
        }
        if ((rc = r->join_prefetch())) return rc;  // the upload thread of the previous call (its error is this call's)
        const uint8_t *h = (const uint8_t *)r->file->p + r->file_pos;
        const void *d_input = nullptr;
        uint64_t lead = 0;
        uint64_t batch_end = r->file_pos + n;  // file offset one past the bytes of this batch
        std::shared_ptr<PinnedBlock> gz_payload;  // gzip: this batch's inflated bytes on the host (string_t payload)
        if (r->d_file) {
            lead = r->shard_first ? shard_halo : (r->file_pos & 15);
            d_input = (const uint8_t *)r->d_file + (r->file_pos - lead);
            n += lead;
            if (!count_only && !r->arrow_emit) {
                gz_payload = std::make_shared<PinnedBlock>();
                size_t cap = n + 64;
                gz_payload->p = global_pool()->take(&cap);
                if (!gz_payload->p) return fail(r, EXG_E_HIP, "out of pinned host memory for the inflated bytes");
                gz_payload->cap = cap;
                gz_payload->pooled = true;
                gz_payload->n = n;
                h = (const uint8_t *)gz_payload->p;
            } else {
                // COUNT(*) / the Arrow stream: no host copy; h is only the base the device subtracts again
                h = (const uint8_t *)(uintptr_t)0x100000000000ull + (r->file_pos - lead);
            }
        } else if (r->pf.valid && want == r->device_batch_bytes && r->file_pos >= r->pf.file_start &&
                   r->file_pos < r->pf.file_start + r->pf.len) {
            const uint64_t off = r->file_pos - r->pf.file_start;
            lead = off & 15;
            r->cur_slot = r->pf.slot;
            r->d_in = r->d_in_slot[r->cur_slot];
            d_input = (const uint8_t *)r->d_in + (off - lead);
            batch_end = r->pf.file_start + r->pf.len;
            n = batch_end - r->file_pos + lead;
            range_end = batch_end == r->range_hi;
            eof = range_end && r->range_eof;
            h -= lead;
            r->pf.valid = false;
            RD_HIP(r, hipStreamWaitEvent(r->stream, r->up_done_of[r->cur_slot], 0));
        } else {
            if (r->pf.valid) RD_HIP(r, hipStreamSynchronize(r->up_stream));  // a prefetch that missed: let it land first
            r->pf.valid = false;
            r->d_in = r->d_in_slot[r->cur_slot];
            lead = shard_halo;
            int rc2;
            if (r->format == EXG_FMT_FASTA && n > (512ull << 20)) {
                // a whole genome in one batch: through two 256 MiB pinned windows, not one pinned block of its size
                rc2 = upload_file(r, r->d_in, n, r->file_pos);
                if (!rc2 && hipMemsetAsync((char *)r->d_in + n, 0, 16, r->stream) != hipSuccess)
                    rc2 = fail(r, EXG_E_HIP, "hipMemsetAsync failed");
            } else {
                rc2 = upload_range(r, r->file_pos - lead, n + lead, r->cur_slot, r->stream);
            }
            if (rc2) return rc2;
            d_input = r->d_in;
            n += lead;
            h -= lead;
        }
        uint64_t first_line_index = 0;
        if (lead && r->shard_first && r->format == EXG_FMT_FASTQ) {
            // the 4-line phase of the line that holds the shard's first byte, from the bytes around the cut ('@' opens a
            // record but also quality lines, so several records are looked at: exg_fastq_guess_phase)
            if (!r->d_phase && !(r->d_phase = exg_rd::dev_pool()->take(r->device, 4096))) return fail(r, EXG_E_HIP, "out of device memory");
            uint32_t guess = 0xFFFFFFFFu;
            rc = exg_fastq_guess_phase(d_input, n, lead, (uint32_t *)r->d_phase, r->stream);
            if (rc) return fail(r, rc, exg_last_error_message());
            RD_HIP(r, hipMemcpyAsync(&guess, r->d_phase, 4, hipMemcpyDeviceToHost, r->stream));
            RD_HIP(r, hipStreamSynchronize(r->stream));
            if (guess <= 3) {
                uint8_t prev = 0;
                if (r->d_file) {
                    RD_HIP(r, hipMemcpyAsync(&prev, (const uint8_t *)r->d_file + r->file_pos - 1, 1, hipMemcpyDeviceToHost, r->stream));
                    RD_HIP(r, hipStreamSynchronize(r->stream));
                } else {
                    prev = ((const uint8_t *)r->file->p)[r->file_pos - 1];
                }
                first_line_index = prev == '\n' ? guess : (guess + 3) % 4;
            } else if (r->d_file) {
                // BGZF shard: exact only when the halo begins with the file (the newlines in front are then all in HBM)
                if (!(r->data0_is_line_start && lead == r->file_pos))
                    return fail(r, EXG_E_PARSE, "cannot tell the FASTQ record phase at the shard boundary of '" + r->files[r->file_idx - 1] + "'");
                unsigned long long nl = 0;
                rc = exg_count_newlines(r->d_file, 0, r->file_pos, (uint64_t *)r->d_phase, r->stream);
                if (rc) return fail(r, rc, exg_last_error_message());
                RD_HIP(r, hipMemcpyAsync(&nl, r->d_phase, 8, hipMemcpyDeviceToHost, r->stream));
                RD_HIP(r, hipStreamSynchronize(r->stream));
                first_line_index = nl;
            } else {
                // too few lines around the cut to tell (a tiny file, a tiny shard) or several phases fit: count the
                // newlines in front of it — exact, and only as slow as a memchr over the page cache
                const char *d = (const char *)r->file->p;
                uint64_t nl = 0;
                for (const char *q = d, *end = d + r->file_pos; q < end;) {
                    const void *hit = memchr(q, '\n', (size_t)(end - q));
                    if (!hit) break;
                    nl++;
                    q = (const char *)hit + 1;
                }
                first_line_index = nl;
            }
        }
        exg_scan_result res;
        const bool no_store = count_only && !r->has_filter;  // a predicate needs the columns even for COUNT(*)
        // a line starts at d_input[0] when the batch is record aligned, or when a shard's halo reaches back to the
        // first byte behind the header
        const bool at_line_start = lead == 0 || (r->shard_first && shard_halo && lead == shard_halo && r->data0_is_line_start &&
                                                 r->file_pos - lead == r->data_base);
        const uint32_t fl = (at_line_start ? EXG_F_BOF : 0u) | (eof ? EXG_F_EOF : 0u) | (no_store ? EXG_F_NO_STORE : 0u);
        std::shared_ptr<Batch> b;
        bool fused_first = false;
        std::function<int()> rescan_general;
        if (r->format == EXG_FMT_FASTQ) {
            exg_fastq_scan_args a;
            memset(&a, 0, sizeof a);
            a.d_input = d_input;
            a.n_bytes = n;
            a.lead = lead;
            a.first_line_index = first_line_index;
            a.payload_base = (uint64_t)(uintptr_t)h;
            a.flags = fl;
            a.algo = EXG_ALGO_FUSED;
            a.d_name = (exg_string_t *)r->d_cols[0];
            a.d_description = (exg_string_t *)r->d_cols[1];
            a.d_sequence = (exg_string_t *)r->d_cols[2];
            a.d_quality = (exg_string_t *)r->d_cols[3];
            a.d_description_validity = (uint64_t *)r->d_valid[0];
            a.capacity_records = r->cap_records;
            a.d_workspace = r->d_ws;
            a.workspace_bytes = r->ws_bytes;
            a.d_result = (exg_scan_result *)r->d_res;
            a.stream = r->stream;
            rc = exg_fastq_scan(&a);
            fused_first = true;
            rescan_general = [a]() mutable {
                a.algo = EXG_ALGO_MULTIPASS;
                return exg_fastq_scan(&a);
            };
        } else if (r->format == EXG_FMT_VCF) {
            exg_vcf_scan_args a;
            memset(&a, 0, sizeof a);
            a.d_input = d_input;
            a.n_bytes = n;
            a.lead = lead;
            a.payload_base = (uint64_t)(uintptr_t)h;
            a.flags = fl;
            a.algo = EXG_ALGO_FUSED;
            for (int c = 0; c < 9; c++) a.d_fields[c] = (exg_string_t *)r->d_cols[c];
            a.d_pos = (int64_t *)r->d_pos;
            a.d_qual = (float *)r->d_qual;
            a.d_qual_validity = (uint64_t *)r->d_valid[0];
            a.d_formats_validity = (uint64_t *)r->d_valid[1];
            a.capacity_records = r->cap_records;
            a.d_workspace = r->d_ws;
            a.workspace_bytes = r->ws_bytes;
            a.d_result = (exg_scan_result *)r->d_res;
            a.stream = r->stream;
            rc = exg_vcf_scan(&a);
            fused_first = true;
            rescan_general = [a]() mutable {
                a.algo = EXG_ALGO_MULTIPASS;
                return exg_vcf_scan(&a);
            };
        } else {
            b = std::make_shared<Batch>();
            if (!count_only) {
                b->payload = b->host.alloc(n + 64);
                if (!b->payload) return fail(r, EXG_E_HIP, "out of pinned host memory");
            }
            exg_fasta_scan_args a;
            memset(&a, 0, sizeof a);
            a.d_input = d_input;
            a.n_bytes = n;
            a.payload_base = (uint64_t)(uintptr_t)h;
            a.seq_payload_base = (uint64_t)(uintptr_t)b->payload;
            a.flags = fl;
            a.d_id = (exg_string_t *)r->d_cols[0];
            a.d_description = (exg_string_t *)r->d_cols[1];
            a.d_sequence = (exg_string_t *)r->d_cols[2];
            a.d_description_validity = (uint64_t *)r->d_valid[0];
            a.d_seq_payload = (uint8_t *)r->d_payload;
            a.capacity_records = r->cap_records;
            a.d_workspace = r->d_ws;
            a.workspace_bytes = r->ws_bytes;
            a.d_result = (exg_scan_result *)r->d_res;
            a.stream = r->stream;
            rc = exg_fasta_scan(&a);
        }
        if (rc) return fail(r, rc, exg_last_error_message());
        double t_scan = now_s();
        rc = exg_fetch_result((const exg_scan_result *)r->d_res, r->stream, &res);
        if (rc) return fail(r, rc, exg_last_error_message());
        if (fused_first && (res.flags & EXG_RF_FALLBACK)) {
            // a record longer than the fused kernel's window, a byte >= 0x80, ...: the general path, on the same batch
            // (the reader launches it only now — EXG_ALGO_AUTO would enqueue its ten gated kernels behind every scan)
            rc = rescan_general();
            if (rc) return fail(r, rc, exg_last_error_message());
            rc = exg_fetch_result((const exg_scan_result *)r->d_res, r->stream, &res);
            if (rc) return fail(r, rc, exg_last_error_message());
            res.flags |= EXG_RF_FALLBACK;
        }
        TRACE("wait(h2d) + scan", t_scan);
        if (res.flags & EXG_RF_INDEX_OVERFLOW)
            return fail(r, EXG_E_CAPACITY, "line index overflow in the general path (pathological line density)");
        if ((res.flags & EXG_RF_CAPACITY) && !no_store) {
            if (r->worst_case_rows) return fail(r, EXG_E_CAPACITY, "more rows than bytes allow: internal error");
            RD_HIP(r, hipStreamSynchronize(r->stream));  // denser rows than provisioned: worst-case vectors, same batch again
            r->free_device();
            r->worst_case_rows = true;
            continue;
        }
        if (res.n_records == 0 && !res.error_code && !eof && !range_end) {
            want *= 2;  // not even one complete record in the batch: widen it
            continue;
        }
        if (res.error_code) {
            r->pending_error = res.error_code;
            r->pending_error_offset = r->file_pos - lead + res.error_offset;
        }
        double t_pf = now_s();
        // While the columns travel back (and the consumer works through the chunks): the bytes the next batch will need
        // move into the other slot — unless they left at the top of this call already (pf2), which is the steady state
        {
            const bool can = !range_end && !res.error_code && !r->d_file && r->format != EXG_FMT_FASTA && want == r->device_batch_bytes && !no_prefetch;
            const uint64_t slack = std::min<uint64_t>(kPrefetchSlack, (batch_end - r->file_pos) / 2);
            const uint64_t start = (batch_end - slack) & ~15ull;
            const uint64_t len = std::min<uint64_t>(r->range_hi - start, r->device_batch_bytes + slack);
            const int other = r->cur_slot ^ 1;
            if (r->pf2.valid) {
                if (can && r->pf2.file_start == start && r->pf2.len == len && r->pf2.slot == other) {
                    r->pf = r->pf2;
                    r->pf2.valid = false;
                } else {
                    r->drop_prefetch2();  // (the batch turned out otherwise: an error, a retry, the end of the range)
                }
            }
            if (can && !r->pf.valid && len + 16 <= r->d_in_cap && r->d_in_slot[other] && !r->up_thread_of[other].joinable())
                start_upload(r, &r->pf, start, len, other);
        }
        TRACE("prefetch issue", t_pf);
        uint64_t k = res.n_records;
        const uint32_t *row_map = nullptr;
        if (r->has_filter && k && !r->arrow_emit) {
            // rows where the predicate is TRUE -> row map; the columns are gathered through it on their way out
            namespace ea = exg::arrow;
            ea::FilterCols fc;
            memset(&fc, 0, sizeof fc);
            const int nsc = n_string_cols(r->format);
            for (int c = 0; c < nsc; c++) {
                fc.kind[c] = ea::kColStr;
                fc.data[c] = r->d_cols[c];
                fc.d_base[c] = (const uint8_t *)d_input;
                fc.payload_base[c] = (uint64_t)(uintptr_t)h;
            }
            if (r->format == EXG_FMT_VCF) {
                fc.kind[1] = ea::kColI64, fc.data[1] = r->d_pos;
                fc.kind[5] = ea::kColF32, fc.data[5] = r->d_qual, fc.validity[5] = (const uint64_t *)r->d_valid[0];
                fc.validity[8] = (const uint64_t *)r->d_valid[1];
            } else {
                fc.validity[1] = (const uint64_t *)r->d_valid[0];
                if (r->format == EXG_FMT_FASTA) {
                    fc.d_base[2] = (const uint8_t *)r->d_payload;
                    fc.payload_base[2] = (uint64_t)(uintptr_t)(b ? b->payload : nullptr);
                }
            }
            uint64_t *d_goff = (uint64_t *)r->d_filter_tmp, *d_tmp = d_goff + r->cap_records + 1;
            ea::FilterCols *d_fc = (ea::FilterCols *)r->d_gather;  // the scratch column is free until the gathers
            RD_HIP(r, hipMemcpyAsync(d_fc, &fc, sizeof fc, hipMemcpyHostToDevice, r->stream));
            ea::filter_rows((const ea::FilterProgram *)r->d_filter_prog, d_fc, (const uint8_t *)r->d_filter_consts, k, d_goff,
                            d_tmp, (uint32_t *)r->d_row_map, r->stream);
            uint64_t n_sel = 0;
            RD_HIP(r, hipMemcpyAsync(&n_sel, d_goff + k, 8, hipMemcpyDeviceToHost, r->stream));
            RD_HIP(r, hipStreamSynchronize(r->stream));
            k = n_sel;
            row_map = (const uint32_t *)r->d_row_map;
        }
        *n_records_out = k;
        if (r->arrow_emit && !count_only) {
            // new_reader: the columns stay in HBM and become Arrow buffers there (exg_arrow_stream.cpp)
            ScanCtx ctx;
            ctx.d_input = d_input;
            ctx.h = h;
            ctx.n_records = k;
            ctx.res = res;
            ctx.h_seq_payload = b ? (const uint8_t *)b->payload : nullptr;
            double t_emit = now_s();
            if (k && (rc = r->arrow_emit(r, ctx))) return rc;
            TRACE("arrow emit", t_emit);
        } else if (k && !count_only) {
            if (!b) b = std::make_shared<Batch>();
            b->host.reserve(r->host_hint);
            b->file = gz_payload ? gz_payload : r->file;
            if (gz_payload) RD_HIP(r, hipMemcpyAsync(gz_payload->p, d_input, n, hipMemcpyDeviceToHost, r->stream));
            b->n_rows = k;
            const int ns = n_string_cols(r->format);
            // schema order (exg_schema_of): VCF exposes parsed POS / QUAL in place of their raw text
            b->n_cols = ns;
            const size_t vw = (size_t)((k + 63) / 64) * 8;
            const bool nested_vcf = r->format == EXG_FMT_VCF;  // id, alt, filter, info, formats: built by nested_emit below
            for (int c = 0; c < ns; c++) {
                if (nested_vcf && (c == 2 || c == 4 || c >= 6)) {
                    b->elem[c] = 0;
                    continue;
                }
                const void *src = r->d_cols[c];
                uint32_t es = 16;
                if (r->format == EXG_FMT_VCF && c == 1) src = r->d_pos, es = 8;
                if (r->format == EXG_FMT_VCF && c == 5) src = r->d_qual, es = 4;
                b->elem[c] = es;
                if (!(b->cols[c] = b->host.alloc(k * es))) return fail(r, EXG_E_HIP, "out of pinned host memory");
                if (row_map) {
                    if (es == 16)
                        exg::arrow::gather_u128(src, row_map, k, r->d_gather, r->stream);
                    else if (es == 8)
                        exg::arrow::gather_u64((const uint64_t *)src, row_map, k, (uint64_t *)r->d_gather, r->stream);
                    else
                        exg::arrow::gather_u32((const uint32_t *)src, row_map, k, (uint32_t *)r->d_gather, r->stream);
                    src = r->d_gather;
                }
                RD_HIP(r, hipMemcpyAsync(b->cols[c], src, k * es, hipMemcpyDeviceToHost, r->stream));
            }
            auto copy_validity = [&](int col, const void *d) -> int {
                if (!(b->validity[col] = b->host.alloc(vw))) return fail(r, EXG_E_HIP, "out of pinned host memory");
                if (row_map) {
                    exg::arrow::gather_bits((const uint64_t *)d, row_map, k, (uint64_t *)r->d_gather, r->stream);
                    d = r->d_gather;
                }
                RD_HIP(r, hipMemcpyAsync(b->validity[col], d, vw, hipMemcpyDeviceToHost, r->stream));
                return EXG_OK;
            };
            if (r->format == EXG_FMT_VCF) {
                if ((rc = copy_validity(5, r->d_valid[0]))) return rc;
                if (!r->nested_state && (rc = nested_prepare(r))) return rc;
                ScanCtx ctx;
                ctx.d_input = d_input;
                ctx.h = h;
                ctx.n_records = k;
                ctx.res = res;
                ctx.h_seq_payload = nullptr;
                uint64_t deliver = k;
                if ((rc = nested_emit(r, ctx, b.get(), row_map, &deliver))) return rc;
                b->n_rows = deliver;
            } else {
                if ((rc = copy_validity(1, r->d_valid[0]))) return rc;
            }
            if (r->format == EXG_FMT_FASTA && res.payload_bytes)
                RD_HIP(r, hipMemcpyAsync(b->payload, r->d_payload, res.payload_bytes, hipMemcpyDeviceToHost, r->stream));
            RD_HIP(r, hipStreamSynchronize(r->stream));
            r->host_hint = b->host.total + b->host.total / 8 + (1u << 20);
            r->batch = b;
        }
        r->shard_first = false;
        if (res.error_code || eof) {
            r->file_done = true;
        } else {
            r->file_pos += res.consumed_bytes - lead;
            if (range_end) {
                // what is left belongs to the next shard, whose halo must reach back to where that record begins
                r->file_done = true;
                if (batch_end - r->file_pos > kShardHalo)
                    return fail(r, EXG_E_UNSUPPORTED, "a record longer than the 1 MiB shard halo crosses the shard boundary at byte " +
                                                          std::to_string(batch_end) + " of '" + r->files[r->file_idx - 1] + "'");
            }
        }
        TRACE("batch (h2d+scan+d2h)", t_batch);
        return EXG_OK;
    }
}

}  // namespace exg_rd

using namespace exg_rd;

static void flat_schema(const exg_reader *r, exg_schema *out);

// compression: NULL => extension sniffing (arrow_reader.rs:60-75); unknown string => uncompressed (:87-88)
static Compression compression_of(const exg_open_args *args) {
    Compression c = kNone;
    if (!args->compression) {
        const std::string path = args->path;
        size_t dot = path.rfind('.');
        std::string ext = dot == std::string::npos ? path : path.substr(dot + 1);
        c = ext == "gz" ? kGzip : ext == "zst" ? kZstd : kNone;
    } else if (!parse_compression(args->compression, &c)) {
        c = kNone;
    }
    return c;
}

// How many byte-range shards a scan of this input is worth and where they run (the table function's init_global asks:
// MaxThreads() = *n_shards, init_local i opens shard i on devices[i]).  One shard per visible device when the input can
// be sharded — text FASTQ / VCF / FASTA, or BGZF FASTQ / VCF (members carry their size) — and holds at least 256 MiB per
// shard; otherwise one.  EXON_GPU_SHARDS=n forces n shards (tests: several shards on one device).
extern "C" int exg_plan_shards(const exg_open_args *args, uint32_t *n_shards, int *devices, uint32_t devices_cap) {
    if (!args || !args->path || !args->file_format || !n_shards || !devices || !devices_cap) {
        exg::set_error("exg_plan_shards: null argument");
        return EXG_E_INVALID_ARG;
    }
    *n_shards = 1;
    devices[0] = args->device;
    const int n_dev = exg_device_count();
    if (n_dev < 1) return EXG_E_NO_DEVICE;
    std::string fmt = args->file_format;
    for (char &ch : fmt) ch = (char)tolower((unsigned char)ch);
    const Compression comp = compression_of(args);
    exg_reader tmp;
    if (list_files(&tmp, args->path) != EXG_OK) return EXG_OK;  // the open will report it
    uint64_t bytes = 0;
    bool shardable = comp == kNone || (comp == kGzip && fmt != "fasta");
    for (const std::string &f : tmp.files) {
        struct stat st;
        if (stat(f.c_str(), &st) != 0) continue;
        bytes += (uint64_t)st.st_size;
        if (comp == kGzip && shardable) {  // BGZF: FEXTRA with a 'BC' subfield in the first member
            uint8_t h[18] = {0};
            FILE *fp = fopen(f.c_str(), "rb");
            const size_t got = fp ? fread(h, 1, sizeof h, fp) : 0;
            if (fp) fclose(fp);
            shardable = got == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
        }
    }
    uint32_t want = 1;
    if (const char *e = getenv("EXON_GPU_SHARDS")) {
        want = (uint32_t)std::max(1, atoi(e));
    } else {
        static const uint64_t per_shard = 256ull << 20;
        want = (uint32_t)std::min<uint64_t>((uint64_t)n_dev, std::max<uint64_t>(1, bytes / per_shard));
    }
    if (!shardable) want = 1;
    want = std::min<uint32_t>(want, devices_cap);
    *n_shards = want;
    for (uint32_t i = 0; i < want; i++) devices[i] = want == 1 ? args->device : (int)(i % (uint32_t)n_dev);
    return EXG_OK;
}

// exon/include/rust.hpp:48, rust/src/arrow_reader.rs:173-197 — same symbol, same result struct: the file type named by
// the last extension, skipping one compression extension; NULL when it is not one of the path's formats
extern "C" ReplacementScanResult replacement_scan(const char *uri) {
    ReplacementScanResult res;
    res.file_type = nullptr;
    if (!uri) return res;
    std::string lower = uri;
    for (char &c : lower) c = (char)tolower((unsigned char)c);
    auto ext_of = [](const std::string &s, size_t end) {
        size_t dot = s.rfind('.', end == std::string::npos ? end : end - 1);
        return dot == std::string::npos ? std::make_pair(s.substr(0, end), (size_t)0)
                                        : std::make_pair(s.substr(dot + 1, (end == std::string::npos ? s.size() : end) - dot - 1), dot);
    };
    auto e1 = ext_of(lower, std::string::npos);
    std::string ext = e1.first;
    static const char *compressed[] = {"gz", "gzip", "zst", "zstd", "bz2", "bzip2", "xz"};
    if (std::find_if(std::begin(compressed), std::end(compressed), [&](const char *c) { return ext == c; }) != std::end(compressed) &&
        e1.second > 0)
        ext = ext_of(lower, e1.second).first;
    if (ext == "fasta" || ext == "fa" || ext == "fna") res.file_type = "FASTA";
    if (ext == "fastq" || ext == "fq") res.file_type = "FASTQ";
    if (ext == "vcf") res.file_type = "VCF";
    return res;
}

extern "C" int exg_open(const exg_open_args *args, exg_reader **out) {
    if (!args || !out || !args->path || !args->file_format) {
        exg::set_error("exg_open: null argument");
        return EXG_E_INVALID_ARG;
    }
    *out = nullptr;
    std::unique_ptr<exg_reader> r(new exg_reader());
    std::string fmt = args->file_format;
    for (char &ch : fmt) ch = (char)tolower((unsigned char)ch);
    if (fmt == "fasta")
        r->format = EXG_FMT_FASTA;
    else if (fmt == "fastq")
        r->format = EXG_FMT_FASTQ;
    else if (fmt == "vcf")
        r->format = EXG_FMT_VCF;
    else {
        // rust/src/arrow_reader.rs:93-102
        exg::set_error("could not parse file_format %s", args->file_format);
        return EXG_E_INVALID_ARG;
    }
    std::string path = args->path;
    r->compression = compression_of(args);
    if (args->batch_rows) r->batch_rows = args->batch_rows;
    if (r->batch_rows % 64) {
        exg::set_error("exg_open: batch_rows must be a multiple of 64 (validity words)");
        return EXG_E_INVALID_ARG;
    }
    if (args->device_batch_bytes) r->device_batch_bytes = (args->device_batch_bytes + 15) / 16 * 16;
    else if (const char *e = getenv("EXG_DEVICE_BATCH_BYTES"))  // tuning / test knob
        r->device_batch_bytes = std::max<uint64_t>(4096, (strtoull(e, nullptr, 10) + 15) / 16 * 16);
    r->device = args->device;
    r->shard_count = args->shard_count ? args->shard_count : 1;
    r->shard_index = args->shard_index;
    if (r->shard_index >= r->shard_count) {
        exg::set_error("exg_open: shard_index %u is not below shard_count %u", r->shard_index, r->shard_count);
        return EXG_E_INVALID_ARG;
    }
    if (r->shard_count > 1 && r->format == EXG_FMT_FASTA && r->compression == kGzip) {
        exg::set_error("a gzip FASTA is not sharded: the records' '>' lines are found in the text");
        return EXG_E_UNSUPPORTED;
    }
    int rc = list_files(r.get(), path);
    if (rc) return rc;
    if (r->compression != kNone && r->compression != kGzip && r->compression != kZstd) {
        exg::set_error("compression is not supported: gzip and zstd have device decoders, bzip2 / xz do not, and there is no CPU fallback");
        return EXG_E_UNSUPPORTED;
    }
    if (r->shard_count > 1 && r->compression == kZstd) {
        exg::set_error("a zstd input is not sharded (its frames are decoded by the whole device at once)");
        return EXG_E_UNSUPPORTED;
    }
    const int n_dev = exg_device_count();
    if (n_dev < 1) return EXG_E_NO_DEVICE;
    if (r->device < 0 || r->device >= n_dev) {
        exg::set_error("exg_open: device %d does not exist (%d visible)", r->device, n_dev);
        return EXG_E_INVALID_ARG;
    }
    DeviceGuard guard(r->device);
    hipError_t he = exg_rd::stream_pool()->take(r->device, &r->stream);
    if (he != hipSuccess) {
        exg::set_error("cannot initialise device %d: %s", r->device, hipGetErrorString(he));
        return EXG_E_HIP;
    }
    if (args->filters && *args->filters) {
        // `SELECT * FROM exon_table WHERE <filters>` (arrow_reader.rs:125-141), evaluated on the device
        exg_schema sch;
        flat_schema(r.get(), &sch);
        std::vector<FilterColumn> fcols;  // nested columns ('x') are refused by the parser, like in new_reader
        for (int c = 0; c < sch.n_columns; c++)
            fcols.push_back({sch.names[c], sch.types[c] == EXG_TYPE_BIGINT ? 'l' : sch.types[c] == EXG_TYPE_FLOAT ? 'f' : sch.types[c] == EXG_TYPE_VARCHAR ? 'u' : 'x'});
        const std::string text = args->filters;
        FilterParser fp(text, fcols);
        if (!fp.parse()) {
            exg::set_error("could not execute sql: %s", fp.err.c_str());
            return EXG_E_INVALID_ARG;
        }
        if (hipMalloc(&r->d_filter_prog, sizeof fp.prog) != hipSuccess ||
            hipMalloc(&r->d_filter_consts, fp.consts.size() + 16) != hipSuccess ||
            hipMemcpy(r->d_filter_prog, &fp.prog, sizeof fp.prog, hipMemcpyHostToDevice) != hipSuccess ||
            (!fp.consts.empty() &&
             hipMemcpy(r->d_filter_consts, fp.consts.data(), fp.consts.size(), hipMemcpyHostToDevice) != hipSuccess)) {
            exg::set_error("could not execute sql: device allocation failed");
            return EXG_E_HIP;
        }
        r->has_filter = true;
    }
    *out = r.release();
    return EXG_OK;
}

// names / types of the columns; no file is touched (the VCF trees come from nested_schema)
static void flat_schema(const exg_reader *r, exg_schema *out) {
    static const exg_type fastq_t[4] = {{EXG_TYPE_VARCHAR, 0, "name", 0, nullptr}, {EXG_TYPE_VARCHAR, 1, "description", 0, nullptr},
                                        {EXG_TYPE_VARCHAR, 0, "sequence", 0, nullptr}, {EXG_TYPE_VARCHAR, 0, "quality_scores", 0, nullptr}};
    static const exg_type fasta_t[3] = {{EXG_TYPE_VARCHAR, 0, "id", 0, nullptr}, {EXG_TYPE_VARCHAR, 1, "description", 0, nullptr},
                                        {EXG_TYPE_VARCHAR, 0, "sequence", 0, nullptr}};
    memset(out, 0, sizeof *out);
    if (r->format == EXG_FMT_FASTQ) {
        // order pinned by test_fastq_scan.test:35-41; names as exon 0.2.6 registers them
        out->n_columns = 4;
        for (int i = 0; i < 4; i++) out->names[i] = fastq_t[i].name, out->types[i] = EXG_TYPE_VARCHAR, out->nullable[i] = fastq_t[i].nullable, out->tree[i] = &fastq_t[i];
    } else if (r->format == EXG_FMT_FASTA) {
        // `id` pinned by test_fasta_scan.test:34-37, order + NULL description by test_fasta_copy.test:75-80
        out->n_columns = 3;
        for (int i = 0; i < 3; i++) out->names[i] = fasta_t[i].name, out->types[i] = EXG_TYPE_VARCHAR, out->nullable[i] = fasta_t[i].nullable, out->tree[i] = &fasta_t[i];
    } else {
        // test_vcf_record_scan.test:10-19: alt is a LIST, info a STRUCT (module.cpp:126-147 maps exon's Arrow schema)
        static const char *n[] = {"chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "formats"};
        static const int t[] = {EXG_TYPE_VARCHAR, EXG_TYPE_BIGINT, EXG_TYPE_LIST, EXG_TYPE_VARCHAR, EXG_TYPE_LIST, EXG_TYPE_FLOAT, EXG_TYPE_LIST, EXG_TYPE_STRUCT, EXG_TYPE_LIST};
        out->n_columns = 9;
        for (int i = 0; i < 9; i++) out->names[i] = n[i], out->types[i] = t[i], out->nullable[i] = !(i == 0 || i == 1 || i == 3);
    }
}

extern "C" int exg_schema_of(exg_reader *r, exg_schema *out) {
    if (!r || !out) return EXG_E_INVALID_ARG;
    flat_schema(r, out);
    if (r->format == EXG_FMT_VCF) {
        DeviceGuard guard(r->device);
        int rc = nested_prepare(r);  // the INFO / FORMAT keys of the first file's header are part of the schema
        if (rc) return rc;
        nested_schema(r, out);
    }
    return EXG_OK;
}

// elements [e0, e1) of a nested node as the vector of DataChunk `chunk`: data is a slice of the batch-wide array; validity
// too when the slice begins on a word boundary, else its bits are shifted into words of their own; the children of a LIST
// are the elements between the chunk's bases, those of a STRUCT share the parent's range
static void slice_vector(const NVec &v, uint64_t e0, uint64_t e1, uint64_t chunk, ChunkKeep *keep, exg_vector *out) {
    memset(out, 0, sizeof *out);
    out->length = e1 - e0;
    out->data = v.data ? (void *)((const char *)v.data + e0 * v.elem) : nullptr;
    if (v.validity) {
        if ((e0 & 63) == 0) {
            out->validity = (uint64_t *)v.validity + e0 / 64;
        } else {
            const uint64_t n = e1 - e0, words = (n + 63) / 64, sh = e0 & 63;
            keep->words.emplace_back(new uint64_t[words ? words : 1]);
            uint64_t *w = keep->words.back().get();
            const uint64_t last_word = (e1 + 63) / 64;  // words of the source that exist
            for (uint64_t i = 0; i < words; i++) {
                const uint64_t lo = v.validity[e0 / 64 + i] >> sh;
                const uint64_t hi = e0 / 64 + i + 1 < last_word ? v.validity[e0 / 64 + i + 1] << (64 - sh) : 0;
                w[i] = lo | hi;
            }
            out->validity = w;
        }
    }
    if (v.children.empty()) return;
    keep->nodes.emplace_back(new exg_vector[v.children.size()]);
    exg_vector *kids = keep->nodes.back().get();
    for (size_t i = 0; i < v.children.size(); i++) {
        if (v.type == EXG_TYPE_LIST)
            slice_vector(v.children[i], v.child_base[chunk], v.child_base[chunk + 1], chunk, keep, &kids[i]);
        else
            slice_vector(v.children[i], e0, e1, chunk, keep, &kids[i]);
    }
    out->n_children = (int)v.children.size();
    out->children = kids;
}

extern "C" int exg_next_chunk(exg_reader *r, exg_chunk *out) {
    if (!r || !out) return EXG_E_INVALID_ARG;
    DeviceGuard guard(r->device);
    memset(out, 0, sizeof *out);
    for (;;) {
        if (r->batch && r->batch_row < r->batch->n_rows) {
            uint64_t row0 = r->batch_row;
            uint64_t n = std::min<uint64_t>(r->batch_rows, r->batch->n_rows - row0);
            out->n_rows = n;
            out->n_columns = r->batch->n_cols;
            ChunkKeep *keep = new ChunkKeep();
            keep->batch = r->batch;
            const uint64_t chunk = row0 / r->batch_rows;
            for (int c = 0; c < r->batch->n_cols; c++) {
                exg_vector &v = keep->top[c];
                if ((size_t)c < r->batch->nested.size() && r->batch->nested[c].type) {
                    slice_vector(r->batch->nested[c], row0, row0 + n, chunk, keep, &v);
                } else {
                    memset(&v, 0, sizeof v);
                    v.data = (char *)r->batch->cols[c] + row0 * r->batch->elem[c];
                    v.validity = r->batch->validity[c] ? (uint64_t *)r->batch->validity[c] + row0 / 64 : nullptr;
                    v.length = n;
                }
                out->data[c] = v.data;
                out->validity[c] = v.validity;
                out->vectors[c] = &v;
            }
            out->keepalive = keep;
            r->batch_row += n;
            return EXG_OK;
        }
        if (r->pending_error) {
            // rows before the failing record have been handed out; now surface the error
            std::string msg = std::string(exg_parse_error_string(r->pending_error)) + " at byte " +
                              std::to_string(r->pending_error_offset) + " of " + r->files[r->file_idx - 1];
            r->pending_error = 0;
            r->batch.reset();
            return fail(r, EXG_E_PARSE, msg);
        }
        if (r->file_done) {
            if (int jrc = r->join_zstd_check()) return jrc;
            if (r->file_idx >= r->files.size()) {
                r->batch.reset();
                return EXG_OK;  // n_rows == 0: end of stream
            }
            int rc = open_next_file(r);
            if (rc) return rc;
        }
        uint64_t k;
        int rc = next_batch(r, false, &k);
        if (rc) return rc;
    }
}

extern "C" void exg_release_chunk(exg_reader *, exg_chunk *chunk) {
    if (chunk && chunk->keepalive) {
        delete (ChunkKeep *)chunk->keepalive;
        chunk->keepalive = nullptr;
    }
}

extern "C" int exg_count_only(exg_reader *r, uint64_t *n_rows) {
    if (!r || !n_rows) return EXG_E_INVALID_ARG;
    DeviceGuard guard(r->device);
    uint64_t total = 0;
    for (;;) {
        if (r->pending_error) {
            std::string msg = std::string(exg_parse_error_string(r->pending_error)) + " at byte " +
                              std::to_string(r->pending_error_offset) + " of " + r->files[r->file_idx - 1];
            r->pending_error = 0;
            return fail(r, EXG_E_PARSE, msg);
        }
        if (r->file_done) {
            if (int jrc = r->join_zstd_check()) return jrc;
            if (r->file_idx >= r->files.size()) break;
            int rc = open_next_file(r);
            if (rc) return rc;
        }
        uint64_t k;
        int rc = next_batch(r, true, &k);
        if (rc) return rc;
        total += k;
    }
    *n_rows = total;
    return EXG_OK;
}

// Pull and release every remaining chunk — what a consumer that only walks the DataChunks does (bench.py's end-to-end
// leg: file in the page cache -> host DataChunks, without an interpreter in the loop).
extern "C" int exg_drain_chunks(exg_reader *r, uint64_t *n_rows, uint64_t *n_chunks) {
    if (!r || !n_rows || !n_chunks) return EXG_E_INVALID_ARG;
    *n_rows = *n_chunks = 0;
    for (;;) {
        exg_chunk c;
        const int rc = exg_next_chunk(r, &c);
        if (rc) return rc;
        if (c.n_rows == 0) return EXG_OK;
        *n_rows += c.n_rows;
        *n_chunks += 1;
        exg_release_chunk(r, &c);
    }
}

// Give back what the process-wide pools hold (device buffers, pinned host blocks, streams of closed readers): for the
// extension's unload / idle path — a long-lived DuckDB process sharing the GPU should not sit on tens of GB it no longer
// uses.  Readers that are open keep what they hold.
extern "C" void exg_trim_pools(void) {
    exg_rd::dev_pool()->trim(0);
    exg_rd::global_pool()->trim();
    exg_rd::stream_pool()->trim();
}

extern "C" const char *exg_reader_error(exg_reader *r) { return r ? r->error.c_str() : ""; }

extern "C" void exg_close(exg_reader *r) { delete r; }
