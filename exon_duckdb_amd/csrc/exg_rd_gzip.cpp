// exg_rd_gzip.cpp — reader level, gzip inputs as a bounded stream of inflated segments (exg_rd_source.hpp).
//
// Replaces DataFusion 28 `FileCompressionType::GZIP.convert_stream` -> async-compression -> flate2 (and noodles-bgzf for
// BGZF) behind rust/src/arrow_reader.rs:60-91 — a streaming decoder whose memory does not depend on the file's size.
// The producer thread walks the file front to back:
//   * BGZF members (every member states its compressed and inflated size): a window of compressed bytes is read into a
//     pinned block by parallel pread, each slice followed at once by its H2D copy; the members that are complete in the
//     window are found by a pointer chase over the pinned copy (no decoding, ~0.2 us per member) and inflated by one
//     wavefront each (exg_inflate.hip) straight into a segment; their CRC-32s are computed behind the inflate on the
//     same stream and compared with the trailers, like flate2 / noodles-bgzf verify them.  Three windows are in flight
//     on three streams: a window's ~4 000 members do not fill the device's 6 144 wavefront slots, and on one stream the
//     next window's launch would wait for the stragglers of this one;
//   * a member of unknown size (what gzip / pigz write) is decoded in ROUNDS of a bounded number of compressed bytes by
//     exg_inflate_round (chunks with a symbolic window, rapidgzip's method), the 32 KiB window and the CRC-32 carried
//     from round to round; a small one by a single wavefront.
// Neither the compressed nor the inflated file is ever resident: a few windows and segments are, whatever the size.
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <deque>
#include <memory>
#include <thread>

#include "exg_rd_source.hpp"

namespace exg_rd {

static uint32_t rd_le32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

namespace {

#define GZ_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            *err = std::string(#expr " failed: ") + hipGetErrorString(_e);                         \
            return EXG_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

// pinned host memory / device memory that grows on demand and goes back to its pool at the end
struct PinBuf {
    char *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        release();
        size_t want = n;
        p = global_pool()->take(&want);
        cap = p ? want : 0;
        return p != nullptr;
    }
    void release() {
        if (p) global_pool()->give(p, cap);
        p = nullptr, cap = 0;
    }
    ~PinBuf() { release(); }
};
struct DevBlock {
    int dev;
    void *p = nullptr;
    size_t cap = 0;
    explicit DevBlock(int d) : dev(d) {}
    bool ensure(size_t n) {
        if (n <= cap) return true;
        release();
        p = dev_pool()->take(dev, n);
        cap = p ? n : 0;
        return p != nullptr;
    }
    void release() {
        if (p) dev_pool()->give(dev, p, cap);
        p = nullptr, cap = 0;
    }
    ~DevBlock() { release(); }
};

class GzipProducer : public SegmentProducer {
public:
    GzipProducer(exg_reader *r, int fd, uint64_t c_begin, uint64_t c_end, uint64_t target, const std::string &path, bool bgzf_only)
        : device_(r->device), fd_(fd), c_pos_(c_begin), c_end_(c_end), target_(std::max<uint64_t>(target, 64u << 10)), path_(path), bgzf_only_(bgzf_only),
          n_lanes_(r->mem_cap ? 2 : 3), capped_(r->mem_cap != 0) {
        if (const char *e = getenv("EXG_GZ_LANES")) n_lanes_ = (size_t)std::max(1, atoi(e));
        // A round of the chunked decoder (one big member) costs a fixed ~10 ms of dependent launches (block finder, chunk
        // decode, window composition, resolve: each as long as its slowest wavefront) and fills the chip only from ~4000
        // chunks on: rounds of one device batch (256 MiB) ran at 11 GB/s where rounds of 1 GiB run at 20+.  So without a
        // memory cap (and without a deliberately small device batch) a round decodes 1 GiB; the segment is scanned in
        // batches as usual.
        member_round_ = target_;
        if (!r->mem_cap && target_ >= (128ull << 20)) member_round_ = std::max<uint64_t>(target_, 1ull << 30);
        if (const char *e = getenv("EXG_STREAM_ROUND_OUT")) member_round_ = std::max<uint64_t>(64u << 10, strtoull(e, nullptr, 10));
    }
    ~GzipProducer() override {
        for (auto &l : lanes_) l->join_read();
        for (auto &l : lanes_) {
            if (l->st) stream_pool()->give(device_, l->st);  // (synchronises it: the blocks below are idle afterwards)
            if (l->ev) (void)hipEventDestroy(l->ev);
        }
    }
    int run(SegmentSink &sink, std::string *err) override;
    void set_reserve(uint64_t r) { reserve_ = (r + 15) & ~15ull; }
    void set_marks(const uint64_t m[2]) { mark_at_[0] = m[0], mark_at_[1] = m[1]; }

private:
    struct Lane {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        PinBuf pin, tab;  // the compressed window; the member table + what comes back (statuses, checksums)
        DevBlock d_comp, d_tab;
        Segment seg;
        uint64_t k = 0, first_member = 0;
        std::vector<uint32_t> crc_expect;
        // the window this lane will decode next, being read (pread + H2D enqueue) on a thread of its own while the producer
        // waits for the device: file bytes [ra_a0, ra_a0 + ra_len)
        std::thread ra;
        bool ra_valid = false, ra_ok = false, ra_hip_failed = false;
        uint64_t ra_a0 = 0, ra_len = 0;
        explicit Lane(int dev) : d_comp(dev), d_tab(dev) {}
        void join_read() {
            if (ra.joinable()) ra.join();
        }
        ~Lane() { join_read(); }
    };
    void start_read(Lane &l, uint64_t a0, uint64_t len);
    uint64_t window_bytes() const {  // compressed bytes that should inflate to about one segment (+ a member's worth)
        const uint64_t want = (uint64_t)((double)target_ / std::max(1.0, ratio_) * 1.05) + (128u << 10);
        return std::max<uint64_t>(want, 256u << 10);
    }
    Lane *lane(size_t i, std::string *err);
    int new_segment(SegmentSink &sink, uint64_t out_bytes, Segment *seg, std::string *err);
    int bgzf_run(SegmentSink &sink, std::string *err);
    int bgzf_issue(SegmentSink &sink, Lane &l, bool *not_bgzf, std::string *err);
    int bgzf_finish(Lane &l, std::string *err);
    int plain_member(SegmentSink &sink, std::string *err);
    int small_member(SegmentSink &sink, uint64_t stream_off, std::string *err);
    int big_member(SegmentSink &sink, uint64_t stream_off, std::string *err);
    int crc_of(const void *d, uint64_t n, hipStream_t st, uint32_t *crc, std::string *err);
    int read_trailer(uint64_t at, uint32_t *crc, uint32_t *isize, std::string *err);

    int device_, fd_;
    uint64_t c_pos_, c_end_, target_;
    uint64_t member_round_ = 0;  // decoded bytes a round of the chunked decoder aims at
    std::string path_;
    bool bgzf_only_;
    size_t n_lanes_;  // windows of BGZF members in flight (two under a memory cap)
    bool capped_;
    uint64_t reserve_ = 1u << 20;
    uint64_t d_pos_ = 0;       // decoded bytes produced so far
    uint64_t n_members_ = 0;   // members seen so far (error messages)
    double ratio_ = 3.0;       // inflated / compressed, running estimate (sizes the next window)
    bool pushed_last_ = false;
    uint64_t mark_at_[2] = {~0ull, ~0ull};  // member offsets whose decoded positions the reader wants to know (a shard's boundaries)
    bool marked_[2] = {false, false};
    void marks(SegmentSink &sink) {  // (c_pos_ / d_pos_: the next member to be decoded and where its bytes will lie)
        for (int i = 0; i < 2; i++)
            if (!marked_[i] && mark_at_[i] != ~0ull && c_pos_ >= mark_at_[i]) sink.set_mark(i, d_pos_), marked_[i] = true;
    }
    std::vector<std::unique_ptr<Lane>> lanes_;
};

GzipProducer::Lane *GzipProducer::lane(size_t i, std::string *err) {
    while (lanes_.size() <= i) {
        std::unique_ptr<Lane> l(new Lane(device_));
        if (stream_pool()->take(device_, &l->st) != hipSuccess || hipEventCreateWithFlags(&l->ev, hipEventDisableTiming) != hipSuccess) {
            *err = "cannot create a stream for the gzip decoder";
            if (l->st) stream_pool()->give(device_, l->st);
            return nullptr;
        }
        lanes_.push_back(std::move(l));
    }
    return lanes_[i].get();
}

// a pooled block for decoded bytes [d_pos_, d_pos_ + out_bytes) with `reserve_` bytes of room in front
int GzipProducer::new_segment(SegmentSink &sink, uint64_t out_bytes, Segment *seg, std::string *err) {
    const size_t cap = (size_t)(reserve_ + 16 + out_bytes + 64);
    seg->buf = sink.take(cap);
    if (!seg->buf) {
        *err = "out of device memory (" + std::to_string(cap >> 20) + " MiB) for a segment of inflated bytes of '" + path_ + "'";
        return EXG_E_HIP;
    }
    seg->cap = cap;
    seg->org = (int64_t)(d_pos_ & ~15ull) - (int64_t)reserve_;
    seg->lo = seg->start = d_pos_;
    seg->hi = d_pos_ + out_bytes;
    seg->last = false;
    return EXG_OK;
}

int GzipProducer::read_trailer(uint64_t at, uint32_t *crc, uint32_t *isize, std::string *err) {
    uint8_t t[8];
    size_t got = 0;
    while (at + 8 <= c_end_ && got < 8) {
        const ssize_t k = pread(fd_, t + got, 8 - got, (off_t)(at + got));
        if (k <= 0) break;
        got += (size_t)k;
    }
    if (got < 8) {
        *err = "truncated gzip member (no trailer) in '" + path_ + "'";
        return EXG_E_PARSE;
    }
    *crc = rd_le32(t);
    *isize = rd_le32(t + 4);
    return EXG_OK;
}

// CRC-32 of n device bytes: 64 KiB segments on the device (exg_crc32.hip), folded here
int GzipProducer::crc_of(const void *d, uint64_t n, hipStream_t st, uint32_t *crc, std::string *err) {
    *crc = 0;
    if (!n) return EXG_OK;
    const uint64_t seg = 65536, n_seg = (n + seg - 1) / seg;
    std::vector<exg_crc_segment> segs(n_seg);
    for (uint64_t i = 0; i < n_seg; i++) segs[i] = exg_crc_segment{i * seg, std::min<uint64_t>(seg, n - i * seg)};
    PoolBuf fr(device_, st);
    void *dt = fr.take(n_seg * (sizeof(exg_crc_segment) + 4) + 64);
    if (!dt) {
        *err = "out of device memory for the checksum segments";
        return EXG_E_HIP;
    }
    uint32_t *d_crc = (uint32_t *)((char *)dt + n_seg * sizeof(exg_crc_segment));
    GZ_HIP(hipMemcpyAsync(dt, segs.data(), n_seg * sizeof(exg_crc_segment), hipMemcpyHostToDevice, st));
    const int rc = exg_crc32_segments(d, (const exg_crc_segment *)dt, (uint32_t)n_seg, d_crc, st);
    if (rc) {
        *err = exg_last_error_message();
        return rc;
    }
    std::vector<uint32_t> c(n_seg);
    GZ_HIP(hipMemcpyAsync(c.data(), d_crc, n_seg * 4, hipMemcpyDeviceToHost, st));
    GZ_HIP(hipStreamSynchronize(st));
    uint32_t total = c[0];
    for (uint64_t i = 1; i < n_seg; i++) total = exg_crc32_combine(total, c[i], segs[i].len);
    *crc = total;
    return EXG_OK;
}

// ---- BGZF: windows of members, up to three in flight ----------------------------------------------------------------
int GzipProducer::bgzf_issue(SegmentSink &sink, Lane &l, bool *not_bgzf, std::string *err) {
    TraceRange range("exg: bgzf window (pread + h2d + inflate + crc32 enqueue)");
    trace_at("P issue-begin", n_members_);
    *not_bgzf = false;
    l.k = 0;
    marks(sink);
    // The window: read ahead by this lane's thread while the producer waited for the device (it begins a little in front of
    // where the window before was EXPECTED to end, so the member that really comes next lies inside it), or read now.
    l.join_read();
    uint64_t a0 = c_pos_ & ~15ull, len = 0;
    bool have = false;
    if (l.ra_valid) {
        l.ra_valid = false;
        have = l.ra_ok && c_pos_ >= l.ra_a0 && c_pos_ + (256u << 10) <= l.ra_a0 + l.ra_len;
        if (!have && l.ra_ok && c_pos_ >= l.ra_a0 && l.ra_a0 + l.ra_len >= c_end_) have = true;  // (the file's last window)
        if (have) a0 = l.ra_a0, len = l.ra_len;
        else if (hipStreamSynchronize(l.st) != hipSuccess) (void)hipGetLastError();  // (its copies must not land in the buffer read next)
    }
    if (!have) {
        len = std::min<uint64_t>(window_bytes() + (c_pos_ - a0), c_end_ - a0);
        if (!l.pin.ensure((size_t)len + 64) || !l.d_comp.ensure((size_t)len + 64)) {
            *err = "out of memory for a window of compressed bytes of '" + path_ + "'";
            return EXG_E_HIP;
        }
        bool hip_failed = false;
        if (!pread_parallel(device_, fd_, a0, (size_t)len, l.pin.p, (char *)l.d_comp.p, l.st, &hip_failed)) {
            *err = hip_failed ? "hipMemcpyAsync failed" : "short read of '" + path_ + "'";
            return hip_failed ? EXG_E_HIP : EXG_E_IO;
        }
    }
    // the members that are complete in the window, up to a segment's worth of output
    Peek pk((const uint8_t *)l.pin.p, -1, len);
    const uint64_t limit = target_ + target_ / 4;
    std::vector<exg_inflate_member> mem;
    l.crc_expect.clear();
    uint64_t rel = c_pos_ - a0, out = 0;
    while (rel < len) {
        exg_inflate_member m;
        uint32_t crc = 0;
        const uint64_t nx = bgzf_member_at(pk, rel, &m, &crc);
        if (!nx) break;
        if (!mem.empty() && out + m.out_cap > limit) break;
        if (!mem.empty() && (a0 + rel == mark_at_[0] || a0 + rel == mark_at_[1])) break;  // a segment begins at every mark
        m.out_off = out;
        out += m.out_cap;
        mem.push_back(m);
        l.crc_expect.push_back(crc);
        rel = nx;
        if (mem.size() >= 0x7FFFFFFFu) break;
    }
    if (mem.empty()) {
        // not a (whole) BGZF member at c_pos_: another kind of gzip member (the caller looks), or a truncated file
        const uint8_t *h = pk.at(rel, 18);
        const bool looks_bgzf = h && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
        if (looks_bgzf || !h) {
            *err = "truncated gzip member at byte " + std::to_string(c_pos_) + " in '" + path_ + "'";
            return EXG_E_PARSE;
        }
        *not_bgzf = true;
        return EXG_OK;
    }
    int rc = new_segment(sink, out, &l.seg, err);
    if (rc) return rc;
    const uint64_t k = mem.size();
    const uint64_t base = (uint64_t)((int64_t)d_pos_ - l.seg.org);  // offset of the segment's first byte in its block
    for (auto &m : mem) m.out_off += base;
    // device tables: members | statuses | checksums ; host (pinned): members | statuses | checksums
    const size_t tab_bytes = k * (sizeof(exg_inflate_member) + sizeof(exg_inflate_status) + 4) + 64;
    if (!l.tab.ensure(tab_bytes) || !l.d_tab.ensure(tab_bytes)) {
        sink.give(l.seg.buf, l.seg.cap);
        l.seg.buf = nullptr;
        *err = "out of memory for the member table";
        return EXG_E_HIP;
    }
    memcpy(l.tab.p, mem.data(), k * sizeof(exg_inflate_member));
    exg_inflate_member *d_m = (exg_inflate_member *)l.d_tab.p;
    exg_inflate_status *d_s = (exg_inflate_status *)((char *)l.d_tab.p + k * sizeof(exg_inflate_member));
    uint32_t *d_c = (uint32_t *)((char *)d_s + k * sizeof(exg_inflate_status));
    hipError_t he = hipMemcpyAsync(d_m, l.tab.p, k * sizeof(exg_inflate_member), hipMemcpyHostToDevice, l.st);
    if (he == hipSuccess) {
        rc = exg_inflate_members(l.d_comp.p, l.seg.buf, d_m, d_s, (uint32_t)k, l.st);
        if (!rc) rc = exg_crc32_members(l.seg.buf, d_m, d_s, (uint32_t)k, d_c, l.st);
        if (rc) *err = exg_last_error_message();
    }
    if (!rc && he == hipSuccess) he = hipMemsetAsync((char *)l.seg.buf + base + out, 0, 64, l.st);
    // (statuses + checksums come back by a kernel's stores, not by a copy: exg_crc32.hip: post_to_host — a copy would queue on
    // an SDMA engine behind the segments' host mirrors)
    if (!rc && he == hipSuccess && (rc = exg::post_to_host(l.tab.p + k * sizeof(exg_inflate_member), d_s, k * (sizeof(exg_inflate_status) + 4), l.st)))
        *err = exg_last_error_message();
    if (!rc && he == hipSuccess) he = hipEventRecord(l.ev, l.st);
    if (rc || he != hipSuccess) {
        (void)hipStreamSynchronize(l.st);
        sink.give(l.seg.buf, l.seg.cap);
        l.seg.buf = nullptr;
        if (!rc) {
            *err = std::string("enqueueing the inflate failed: ") + hipGetErrorString(he);
            rc = EXG_E_HIP;
        }
        return rc;
    }
    l.k = k;
    l.first_member = n_members_;
    n_members_ += k;
    const uint64_t used = a0 + rel - c_pos_;
    if (out) ratio_ = 0.5 * ratio_ + 0.5 * std::min(1000.0, (double)out / (double)std::max<uint64_t>(1, used));
    c_pos_ = a0 + rel;
    d_pos_ += out;
    l.seg.last = c_pos_ >= c_end_;
    trace_at(have ? "P issue-end(read ahead)" : "P issue-end(read now)", n_members_);
    return EXG_OK;
}

// file bytes [a0, a0 + len) -> the lane's pinned block -> its device block, on a thread of the lane's (the lane is idle)
void GzipProducer::start_read(Lane &l, uint64_t a0, uint64_t len) {
    l.join_read();
    l.ra_valid = false;
    if (!l.pin.ensure((size_t)len + 64) || !l.d_comp.ensure((size_t)len + 64)) return;  // (the window is read when it is its turn)
    l.ra_a0 = a0, l.ra_len = len;
    l.ra_ok = false;
    l.ra_valid = true;
    Lane *lp = &l;
    const int dev = device_, fd = fd_;
    MemMeter *meter = tl_meter();
    l.ra = std::thread([lp, dev, fd, a0, len, meter] {
        (void)hipSetDevice(dev);
        pin_to_device_node(dev);
        MeterScope scope(meter);
        lp->ra_ok = pread_parallel(dev, fd, a0, (size_t)len, lp->pin.p, (char *)lp->d_comp.p, lp->st, &lp->ra_hip_failed);
    });
}

int GzipProducer::bgzf_finish(Lane &l, std::string *err) {
    GZ_HIP(hipEventSynchronize(l.ev));
    const exg_inflate_member *m = (const exg_inflate_member *)l.tab.p;
    const exg_inflate_status *s = (const exg_inflate_status *)(l.tab.p + l.k * sizeof(exg_inflate_member));
    const uint32_t *c = (const uint32_t *)((const char *)s + l.k * sizeof(exg_inflate_status));
    for (uint64_t i = 0; i < l.k; i++) {
        if (s[i].code || s[i].produced != m[i].out_cap) {
            *err = "corrupt deflate stream (member " + std::to_string(l.first_member + i) + ", code " + std::to_string(s[i].code) + ") in '" + path_ + "'";
            return EXG_E_PARSE;
        }
        if (c[i] != l.crc_expect[i]) {
            *err = "corrupt gzip stream does not have a matching checksum (member " + std::to_string(l.first_member + i) + " of '" + path_ + "')";
            return EXG_E_PARSE;
        }
    }
    return EXG_OK;
}

int GzipProducer::bgzf_run(SegmentSink &sink, std::string *err) {
    const size_t n_lanes = n_lanes_;
    std::deque<size_t> inflight;
    size_t next_lane = 0;
    bool stop = false;
    int rc = EXG_OK;
    auto drain = [&] {
        for (size_t i : inflight) {
            Lane &l = *lanes_[i];
            (void)hipStreamSynchronize(l.st);
            sink.give(l.seg.buf, l.seg.cap);
            l.seg.buf = nullptr;
        }
        inflight.clear();
    };
    for (;;) {
        while (!stop && inflight.size() < n_lanes && c_pos_ < c_end_) {
            Lane *l = lane(next_lane, err);
            if (!l) {
                drain();
                return EXG_E_HIP;
            }
            bool not_bgzf = false;
            if ((rc = bgzf_issue(sink, *l, &not_bgzf, err))) {
                // what was issued before the failing window is handed out first: the consumer reaches the error behind it
                stop = true;
                break;
            }
            if (not_bgzf) {
                stop = true;
                break;
            }
            inflight.push_back(next_lane);
            next_lane = (next_lane + 1) % n_lanes;
        }
        // the window after the ones in flight is read while this thread waits for the device below: its lane is the one that
        // will be issued next, idle as soon as its last segment has been handed over
        auto read_ahead = [&] {
            static const bool off = getenv("EXG_GZ_NO_READAHEAD") != nullptr;
            if (off || stop || c_pos_ >= c_end_ || lanes_.size() <= next_lane) return;
            Lane &nl = *lanes_[next_lane];
            if (nl.ra_valid || nl.ra.joinable()) return;
            for (size_t i : inflight)
                if (i == next_lane) return;  // still decoding
            const uint64_t a0 = c_pos_ & ~15ull;  // (exact: every window in front of it has been walked)
            start_read(nl, a0, std::min<uint64_t>(window_bytes() + (c_pos_ - a0), c_end_ - a0));
        };
        read_ahead();
        if (inflight.empty()) break;
        Lane &l = *lanes_[inflight.front()];
        std::string e2;
        trace_at("P finish-wait", l.first_member);
        const int frc = bgzf_finish(l, &e2);
        trace_at("P finished", l.first_member);
        if (frc) {
            drain();
            *err = e2;
            return frc;
        }
        inflight.pop_front();
        pushed_last_ = l.seg.last;
        if (!sink.push(std::move(l.seg))) {
            drain();
            return EXG_OK;  // the consumer is gone
        }
        trace_at("P pushed", l.first_member);
        read_ahead();
        if (c_pos_ >= c_end_ && inflight.empty()) break;
    }
    // a window read ahead that is not going to be decoded here (another kind of member follows, or an error): let it land
    for (auto &lp : lanes_) {
        lp->join_read();
        if (lp->ra_valid) (void)hipStreamSynchronize(lp->st);
        lp->ra_valid = false;
    }
    return rc;
}

// ---- a member that does not state its size ----------------------------------------------------------------------------
int GzipProducer::plain_member(SegmentSink &sink, std::string *err) {
    // the RFC 1952 header (FEXTRA / FNAME / FCOMMENT / FHCRC are skipped by the index walk of exg_gzip.cpp)
    const size_t peek = (size_t)std::min<uint64_t>(c_end_ - c_pos_, 70000);
    std::vector<uint8_t> head(peek);
    size_t got = 0;
    while (got < peek) {
        const ssize_t k = pread(fd_, head.data() + got, peek - got, (off_t)(c_pos_ + got));
        if (k <= 0) break;
        got += (size_t)k;
    }
    exg_inflate_member m;
    uint64_t k = 0, total = 0;
    int open_ended = 0;
    const int rc = exg_gzip_index(head.data(), got, 0, &m, 1, &k, &total, &open_ended);
    if (rc == EXG_E_CAPACITY) {  // (a sized member first: BGZF in a window the walk above could not finish — a truncated file)
        *err = "truncated gzip member at byte " + std::to_string(c_pos_) + " in '" + path_ + "'";
        return EXG_E_PARSE;
    }
    if (rc || k == 0) {
        const std::string what = rc ? exg_last_error_message() : "invalid gzip header";
        *err = what + " (member at byte " + std::to_string(c_pos_) + ") in '" + path_ + "'";
        return EXG_E_PARSE;
    }
    n_members_++;
    const uint64_t stream_off = c_pos_ + m.comp_off;
    static const uint64_t stream_min = getenv("EXG_STREAM_MIN_BYTES") ? strtoull(getenv("EXG_STREAM_MIN_BYTES"), nullptr, 10) : (128ull << 10);  // (one wavefront does ~13 MB/s: 4 MB took 0.3 s)
    static const bool stream_ok = !getenv("EXG_NO_STREAM_INFLATE");
    if (stream_ok && c_end_ - stream_off >= stream_min) return big_member(sink, stream_off, err);
    return small_member(sink, stream_off, err);
}

// one wavefront: the member (and whatever follows it in the file) is at most a few hundred KiB of input
int GzipProducer::small_member(SegmentSink &sink, uint64_t stream_off, std::string *err) {
    Lane *lp = lane(0, err);
    if (!lp) return EXG_E_HIP;
    Lane &l = *lp;
    const uint64_t a0 = stream_off & ~15ull;
    // with the chunked decoder switched off a big member comes here too: bounded by what one block may hold
    const uint64_t len = c_end_ - a0;
    if (!l.pin.ensure((size_t)len + 64) || !l.d_comp.ensure((size_t)len + 64)) {
        *err = "out of memory for the compressed bytes of '" + path_ + "'";
        return EXG_E_HIP;
    }
    bool hip_failed = false;
    if (!pread_parallel(device_, fd_, a0, (size_t)len, l.pin.p, (char *)l.d_comp.p, l.st, &hip_failed)) {
        *err = hip_failed ? "hipMemcpyAsync failed" : "short read of '" + path_ + "'";
        return hip_failed ? EXG_E_HIP : EXG_E_IO;
    }
    const uint64_t comp_size = c_end_ - stream_off;
    // DEFLATE expands at most 1032:1; `cat a.vcf.gz b.vcf.gz` of highly compressible members must not be reported as
    // corrupt for outgrowing a guessed ratio.  A big member (only with the chunked decoder off) gets 8:1 + its ISIZE.
    uint64_t bound = comp_size < (128u << 10) ? comp_size * 1032 + 65536 : comp_size * 8 + 65536;
    if (comp_size >= (128u << 10) && len >= 4) bound = std::max<uint64_t>(bound, rd_le32((const uint8_t *)l.pin.p + len - 4));
    Segment seg;
    int rc = new_segment(sink, bound, &seg, err);
    if (rc) return rc;
    struct Table {
        exg_inflate_member m;
        exg_inflate_status s;
        uint32_t crc;
    };
    if (!l.tab.ensure(sizeof(Table) + 64) || !l.d_tab.ensure(sizeof(Table) + 64)) {
        sink.give(seg.buf, seg.cap);
        *err = "out of memory for the member table";
        return EXG_E_HIP;
    }
    Table *h = (Table *)l.tab.p, *d = (Table *)l.d_tab.p;
    const uint64_t base = (uint64_t)((int64_t)d_pos_ - seg.org);
    h->m = exg_inflate_member{stream_off - a0, comp_size, base, bound};
    hipError_t he = hipMemcpyAsync(&d->m, &h->m, sizeof h->m, hipMemcpyHostToDevice, l.st);
    if (he == hipSuccess) {
        rc = exg_inflate_members(l.d_comp.p, seg.buf, &d->m, &d->s, 1, l.st);
        if (!rc) rc = exg_crc32_members(seg.buf, &d->m, &d->s, 1, &d->crc, l.st);
        if (rc) *err = exg_last_error_message();
    }
    if (!rc && he == hipSuccess && (rc = exg::post_to_host(&h->s, &d->s, sizeof h->s + 4, l.st))) *err = exg_last_error_message();
    if (he == hipSuccess) he = hipStreamSynchronize(l.st);
    if (!rc && he != hipSuccess) {
        *err = std::string("inflate failed: ") + hipGetErrorString(he);
        rc = EXG_E_HIP;
    }
    if (!rc && h->s.code) {
        *err = "corrupt deflate stream (member " + std::to_string(n_members_ - 1) + ", code " + std::to_string(h->s.code) + ") in '" + path_ + "'";
        rc = EXG_E_PARSE;
    }
    uint32_t crc = 0, isize = 0;
    if (!rc) rc = read_trailer(stream_off + h->s.consumed, &crc, &isize, err);
    if (!rc && (crc != h->crc || isize != (uint32_t)h->s.produced)) {
        *err = "corrupt gzip stream does not have a matching checksum (member " + std::to_string(n_members_ - 1) + " of '" + path_ + "')";
        rc = EXG_E_PARSE;
    }
    if (!rc && hipMemsetAsync((char *)seg.buf + base + h->s.produced, 0, 64, l.st) != hipSuccess) rc = EXG_E_HIP;
    if (!rc && hipStreamSynchronize(l.st) != hipSuccess) rc = EXG_E_HIP;
    if (rc) {
        if (err->empty()) *err = "inflate failed";
        sink.give(seg.buf, seg.cap);
        return rc;
    }
    seg.hi = d_pos_ + h->s.produced;
    d_pos_ = seg.hi;
    c_pos_ = stream_off + h->s.consumed + 8;
    seg.last = c_pos_ >= c_end_;
    pushed_last_ = seg.last;
    (void)sink.push(std::move(seg));
    return EXG_OK;
}

// rounds of the chunked decoder: a bounded window of compressed bytes each, the LZ77 window and the checksum carried
int GzipProducer::big_member(SegmentSink &sink, uint64_t stream_off, std::string *err) {
    // two lanes take turns: while one window is decoded, the next one is read (pread + H2D) by the other lane's thread —
    // a round of 1 GiB reads ~0.5 GB of compressed bytes, 10 ms that used to stand in front of every round
    Lane *lanes2[2] = {lane(0, err), nullptr};
    if (!lanes2[0]) return EXG_E_HIP;
    const bool ahead = !capped_ && !getenv("EXG_NO_PREFETCH");
    if (ahead && !(lanes2[1] = lane(1, err))) return EXG_E_HIP;
    struct DropReads {  // (whatever way this function is left: no read may still be writing into a lane's buffers)
        Lane **l;
        ~DropReads() {
            for (int i = 0; i < 2; i++)
                if (l[i]) {
                    l[i]->join_read();
                    if (l[i]->ra_valid && hipStreamSynchronize(l[i]->st) != hipSuccess) (void)hipGetLastError();
                    l[i]->ra_valid = false;
                }
        }
    } drop_reads{lanes2};
    size_t li = 0;
    DevBlock d_window(device_);
    if (!d_window.ensure(32768)) {
        *err = "out of device memory";
        return EXG_E_HIP;
    }
    uint64_t bit = stream_off * 8;  // absolute bit position in the file of the next block header
    bool have_window = false;
    uint32_t crc_run = 0;
    uint64_t produced_total = 0;
    uint64_t grow = 1;  // window multiplier after a round that found no block start
    for (;;) {
        if (sink.cancelled()) return EXG_OK;
        Lane &l = *lanes2[li];
        uint64_t a0 = (bit / 8) & ~15ull;
        uint64_t want = (uint64_t)((double)member_round_ / std::max(1.0, ratio_)) + (64u << 10);
        want = std::max<uint64_t>(want, 256u << 10) * grow;
        if (getenv("EXG_STREAM_ROUND_BYTES")) want = strtoull(getenv("EXG_STREAM_ROUND_BYTES"), nullptr, 10) * grow;
        uint64_t len = std::min<uint64_t>(want, c_end_ - a0);
        // the window this lane read ahead, if it holds where the round before really ended (+ a good part of a round)
        l.join_read();
        bool have = false;
        if (l.ra_valid) {
            l.ra_valid = false;
            have = l.ra_ok && grow == 1 && a0 >= l.ra_a0 && (l.ra_a0 + l.ra_len >= c_end_ || a0 + len / 2 <= l.ra_a0 + l.ra_len);
            if (have) {
                a0 = l.ra_a0;  // (bytes in front of the block header are skipped by start_bit)
                len = l.ra_len;
            } else if (hipStreamSynchronize(l.st) != hipSuccess) {
                (void)hipGetLastError();  // (its copies must not land in the buffer read next)
            }
        }
        const bool partial = a0 + len < c_end_;
        if (!have) {
            if (!l.pin.ensure((size_t)len + 64) || !l.d_comp.ensure((size_t)len + 64)) {
                *err = "out of memory for a window of compressed bytes of '" + path_ + "'";
                return EXG_E_HIP;
            }
            bool hip_failed = false;
            if (!pread_parallel(device_, fd_, a0, (size_t)len, l.pin.p, (char *)l.d_comp.p, l.st, &hip_failed)) {
                *err = hip_failed ? "hipMemcpyAsync failed" : "short read of '" + path_ + "'";
                return hip_failed ? EXG_E_HIP : EXG_E_IO;
            }
        }
        if (ahead && partial) {
            // the round ends at a block boundary at or in front of the last block start it finds: the next window begins a few
            // chunks in front of this one's end (read twice: ~1 % of a round)
            const uint64_t chunk = std::max<uint64_t>(32u << 10, (len / 3900 + 16383) & ~16383ull);
            const uint64_t slack = std::max<uint64_t>(4u << 20, 8 * chunk);
            const uint64_t na0 = (a0 + len > slack ? a0 + len - slack : 0) & ~15ull;
            if (na0 > a0) start_read(*lanes2[li ^ 1], na0, std::min<uint64_t>(want + slack, c_end_ - na0));
        }
        GZ_HIP(hipMemsetAsync((char *)l.d_comp.p + len, 0, 64, l.st));
        exg_inflate_round_args a;
        memset(&a, 0, sizeof a);
        a.d_comp = l.d_comp.p;
        a.comp_off = 0;
        a.comp_size = len;
        a.start_bit = bit - a0 * 8;
        // at most one piece per decoding wavefront the chip holds (4 per SIMD = 4096; a few more pieces than slots would cost
        // a second round for them alone), at least 32 KiB each (a block is 20-60 KB of input)
        a.chunk_bytes = std::max<uint64_t>(32u << 10, (len / 3900 + 16383) & ~16383ull);
        if (getenv("EXG_STREAM_CHUNK_BYTES")) a.chunk_bytes = strtoull(getenv("EXG_STREAM_CHUNK_BYTES"), nullptr, 10);
        a.partial = partial;
        a.have_window = have_window;
        a.d_window = d_window.p;
        a.front_reserve = reserve_ + (d_pos_ & 15);
        a.ratio_hint = produced_total ? ratio_ : 0.0;
        a.stream = l.st;
        int rc;
        {
            TraceRange range("exg: gzip member round (chunked inflate)");
            rc = exg_inflate_round(&a);
        }
        if (rc) {
            *err = std::string(exg_last_error_message()) + " in '" + path_ + "'";
            return rc;
        }
        if (a.need_more) {
            if (!partial) {
                *err = "corrupt deflate stream (no block start in the rest of the member) in '" + path_ + "'";
                return EXG_E_PARSE;
            }
            grow *= 2;
            if (ahead) {  // (the window read ahead began where this one was expected to end: not what the bigger retry needs)
                Lane &o = *lanes2[li ^ 1];
                o.join_read();
                if (o.ra_valid && hipStreamSynchronize(o.st) != hipSuccess) (void)hipGetLastError();
                o.ra_valid = false;
            }
            continue;
        }
        grow = 1;
        Segment seg;
        seg.buf = a.d_out;
        seg.cap = (size_t)a.out_alloc;
        seg.org = (int64_t)(d_pos_ & ~15ull) - (int64_t)reserve_;
        seg.lo = seg.start = d_pos_;
        seg.hi = d_pos_ + a.produced;
        // the member's checksum runs over all rounds
        uint32_t c = 0;
        int crc_rc = crc_of(seg.at(d_pos_), a.produced, l.st, &c, err);
        if (crc_rc) {
            sink.give(seg.buf, seg.cap);
            return crc_rc;
        }
        crc_run = produced_total ? exg_crc32_combine(crc_run, c, a.produced) : c;
        produced_total += a.produced;
        const uint64_t used_bits = a.end_bit - a.start_bit;
        if (a.produced && used_bits) ratio_ = 0.5 * ratio_ + 0.5 * std::min(1000.0, (double)a.produced * 8.0 / (double)used_bits);
        d_pos_ = seg.hi;
        bit = a0 * 8 + a.end_bit;
        have_window = true;
        if (a.final_block) {
            const uint64_t trailer = (bit + 7) / 8;
            uint32_t crc = 0, isize = 0;
            int trc = read_trailer(trailer, &crc, &isize, err);
            if (!trc && (crc != crc_run || isize != (uint32_t)produced_total)) {
                *err = "corrupt gzip stream does not have a matching checksum ('" + path_ + "')";
                trc = EXG_E_PARSE;
            }
            if (trc) {
                // the rows of this round are handed out before the error, like a streaming decoder hands out what it decoded
                // before it fails at the trailer (flate2 behind a BufReader: rust/src/arrow_reader.rs:60-91); the zstd
                // producer does the same with a Content_Checksum
                pushed_last_ = false;
                (void)sink.push(std::move(seg));
                return trc;
            }
            c_pos_ = trailer + 8;
            seg.last = c_pos_ >= c_end_;
            pushed_last_ = seg.last;
            (void)sink.push(std::move(seg));
            return EXG_OK;
        }
        if (used_bits == 0 && a.produced == 0) {
            sink.give(seg.buf, seg.cap);
            *err = "corrupt deflate stream (the chunked decoder makes no progress) in '" + path_ + "'";
            return EXG_E_PARSE;
        }
        pushed_last_ = false;
        if (!sink.push(std::move(seg))) return EXG_OK;
        if (ahead) li ^= 1;
    }
}

int GzipProducer::run(SegmentSink &sink, std::string *err) {
    while (c_pos_ < c_end_ && !sink.cancelled()) {
        marks(sink);
        // what kind of member begins here?  (BGZF: FEXTRA with a 'BC' subfield that states the member's size)
        Peek pk(nullptr, fd_, c_end_);
        exg_inflate_member m;
        int rc;
        if (bgzf_member_at(pk, c_pos_, &m)) {
            rc = bgzf_run(sink, err);
        } else if (bgzf_only_) {
            *err = "not a BGZF member at byte " + std::to_string(c_pos_) + " of '" + path_ + "'";
            rc = EXG_E_PARSE;
        } else {
            const uint8_t *h = pk.at(c_pos_, 18);
            const bool looks_bgzf = h && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
            if (looks_bgzf) {  // a BGZF header whose member does not fit the file: the walk above says why
                *err = "truncated gzip member at byte " + std::to_string(c_pos_) + " in '" + path_ + "'";
                rc = EXG_E_PARSE;
            } else {
                rc = plain_member(sink, err);
            }
        }
        if (rc) return rc;
    }
    marks(sink);
    if (!pushed_last_ && !sink.cancelled()) {  // an empty range (a shard without members): the stream still ends
        Segment seg;
        int rc = new_segment(sink, 0, &seg, err);
        if (rc) return rc;
        Lane *l = lane(0, err);
        if (!l) {
            sink.give(seg.buf, seg.cap);
            return EXG_E_HIP;
        }
        hipError_t he = hipMemsetAsync((char *)seg.buf + ((int64_t)d_pos_ - seg.org), 0, 64, l->st);
        if (he == hipSuccess) he = hipStreamSynchronize(l->st);
        if (he != hipSuccess) {
            sink.give(seg.buf, seg.cap);
            *err = std::string("hipMemsetAsync failed: ") + hipGetErrorString(he);
            return EXG_E_HIP;
        }
        seg.last = true;
        (void)sink.push(std::move(seg));
    }
    return EXG_OK;
}

}  // namespace

std::unique_ptr<SegmentProducer> make_gzip_producer(exg_reader *r, int fd, uint64_t c_begin, uint64_t c_end, uint64_t target, const std::string &path,
                                                    bool bgzf_only, uint64_t reserve, const uint64_t mark_at[2]) {
    std::unique_ptr<GzipProducer> p(new GzipProducer(r, fd, c_begin, c_end, target, path, bgzf_only));
    p->set_reserve(reserve);
    if (mark_at) p->set_marks(mark_at);
    return std::unique_ptr<SegmentProducer>(p.release());
}

}  // namespace exg_rd
