// exg_rd_zstd.cpp — reader level, zstd inputs (.zst, compression='zstd') as a bounded stream of decoded segments
// (exg_rd_source.hpp).  Replaces DataFusion 28 `FileCompressionType::ZSTD.convert_stream` -> async-compression -> zstd
// 0.12.3 behind rust/src/arrow_reader.rs:73, :87-88 — a streaming decoder with a window, whose memory does not depend on the
// file's size.  Neither does this one's:
//   * the host walks the frame / block headers of the mapped file (exg_zstd_index.cpp: a pointer chase, no decoding);
//   * the producer thread decodes the blocks in ROUNDS of about one segment of output (exg::zst::decode_round): the
//     compressed bytes of the round's blocks travel to the device (parallel pread + H2D), all of them are entropy-decoded
//     at once, executed in chunks and resolved — whole frames, or a part of a frame;
//   * what a frame that goes on needs from earlier rounds travels along: the repeat offsets, up to Window_Size bytes of its
//     output (kept in a device buffer of the producer's and laid in front of the next round's bytes — where the scan's
//     carried tail lands too), and the compressed bytes of the blocks whose Huffman tree / FSE tables it repeats;
//   * Content_Checksums: a frame inside one round is hashed on the device (XXH64, up to EXG_ZSTD_VERIFY_MAX bytes); a frame
//     that is larger or spans rounds is hashed on a host thread from copies of its parts that come back over PCIe while the
//     next round is being decoded; a mismatch is reported behind the frame's rows, where a streaming decoder reports it.
#include <errno.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <unistd.h>

#include <algorithm>
#include <deque>
#include <thread>

#include "exg_rd_source.hpp"
#include "exg_xxh64.hpp"
#include "exg_zstd.hpp"

namespace exg_rd {

namespace zst = exg::zst;

namespace {

// The stage behind the decoder: XXH64 of the frames that span rounds (or are too large for the device's serial hash), then the
// segment goes to the consumer — on a thread of its own, so that round n + 1 is decoded while round n's bytes come back
// over PCIe (32 MiB pieces, two pinned buffers in turn, a stream of its own) and are hashed (~27 GB/s on a core of the GPU
// box against ~12 GB/s of decoding: the checksum of a frame of any size hides behind the decode.  The first form copied and
// hashed on the decoder's own thread with four pieces of slack: decode + hash in series, 8.3 instead of 12 GB/s.)
class FrameHasher {
public:
    struct Part {  // a run of a frame's bytes inside the segment
        const uint8_t *d_src = nullptr;
        uint64_t len = 0;
        uint32_t frame = 0, expect = 0;
        bool begins = false, ends = false;
    };
    struct Job {
        Segment seg;
        std::vector<Part> parts;
    };
    FrameHasher(int device, SegmentSink *sink, MemMeter *meter)
        : device_(device), sink_(sink), meter_(meter),
          // segments that may wait behind the one being hashed: one (two or three measured nothing on a 4 GB checksummed frame —
          // 250 / 264 against 247 / 284 ms, inside the boxes' noise — and each is a segment of device memory)
          depth_(getenv("EXG_ZSTD_HASH_QUEUE") && !(meter && meter->cap) ? std::max(1, atoi(getenv("EXG_ZSTD_HASH_QUEUE"))) : 1),
          thread_([this] { loop(); }) {}
    ~FrameHasher() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
            cv_.notify_all();
        }
        thread_.join();
        for (Job &j : queue_) sink_->give(j.seg.buf, j.seg.cap);  // (left behind by an early return of the decoder)
    }
    // hands a decoded segment over (blocks while one waits behind the one being hashed: memory stays bounded);
    // false: nothing more is wanted — a checksum did not match (error()), or the consumer closed the stream (gone())
    bool submit(Job &&job) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return queue_.size() < depth_ || failed_ || gone_; });
        if (failed_ || gone_) {
            lk.unlock();
            sink_->give(job.seg.buf, job.seg.cap);
            return false;
        }
        queue_.push_back(std::move(job));
        cv_.notify_all();
        return true;
    }
    // everything submitted has been hashed and pushed; false + error(): a mismatch or a failed copy
    bool drain() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return queue_.empty() && !busy_; });  // (also after a mismatch: its segment's rows go out first)
        return !failed_;
    }
    bool gone() {
        std::lock_guard<std::mutex> g(mu_);
        return gone_;
    }
    std::string error() {
        std::lock_guard<std::mutex> g(mu_);
        return error_;
    }

private:
    static constexpr size_t kPiece = 32u << 20;
    static constexpr size_t kHelpers = 4;  // threads that hash whole frames beside this one
    void fail(const std::string &what) {
        std::lock_guard<std::mutex> g(mu_);
        if (!failed_) failed_ = true, error_ = what;
        cv_.notify_all();
    }
    // one part: pieces copied into the two pinned buffers in turn, piece k + 1 on its way while piece k is hashed
    // (bad_frame: the lowest frame of the job whose checksum did not match — several threads hash a job's frames)
    bool hash_part(const Part &part, hipStream_t st, char *pin[2], hipEvent_t ev[2], exg::Xxh64 &h_, std::atomic<uint32_t> &bad_frame) {
        if (part.begins) h_ = exg::Xxh64();
        const uint64_t n_pieces = (part.len + kPiece - 1) / kPiece;
        auto issue = [&](uint64_t k) -> bool {
            const size_t len = (size_t)std::min<uint64_t>(kPiece, part.len - k * kPiece);
            return hipMemcpyAsync(pin[k & 1], part.d_src + k * kPiece, len, hipMemcpyDeviceToHost, st) == hipSuccess &&
                   hipEventRecord(ev[k & 1], st) == hipSuccess;
        };
        if (n_pieces && !issue(0)) return false;
        static const bool trace = getenv("EXG_TRACE") != nullptr;
        double t_wait = 0, t_hash = 0;
        for (uint64_t k = 0; k < n_pieces; k++) {
            const double t0 = trace ? now_s() : 0;
            if (hipEventSynchronize(ev[k & 1]) != hipSuccess) return false;
            if (k + 1 < n_pieces && !issue(k + 1)) return false;
            const double t1 = trace ? now_s() : 0;
            h_.update((const uint8_t *)pin[k & 1], (size_t)std::min<uint64_t>(kPiece, part.len - k * kPiece));
            if (trace) t_wait += t1 - t0, t_hash += now_s() - t1;
        }
        if (trace && part.len >= (64u << 20))
            fprintf(stderr, "[exg] zstd hasher: %.1f MB of frame %u: %.1f ms waiting for the copies, %.1f ms hashing (%.1f GB/s)\n", part.len / 1e6, part.frame,
                    t_wait * 1e3, t_hash * 1e3, part.len / t_hash / 1e9);
        if (part.ends && (uint32_t)h_.digest() != part.expect) {
            uint32_t seen = bad_frame.load();
            while (part.frame < seen && !bad_frame.compare_exchange_weak(seen, part.frame)) {
            }
        }
        return true;  // (a mismatch is not a failed copy: the segment still goes out — the error comes behind its rows)
    }
    // the same part out of the segment's host mirror (the consumer wants the bytes on the host anyway: they are brought back ONCE,
    // in pieces, and hashed as the pieces arrive — not copied for the hash and again for the strings)
    bool hash_part_mirror(const Part &part, const Segment &seg, exg::Xxh64 &h_, std::atomic<uint32_t> &bad_frame) {
        const HostMirror &m = *seg.mirror;
        if (part.begins) h_ = exg::Xxh64();
        const uint64_t x0 = m.from + (uint64_t)(part.d_src - seg.at(m.from));  // the part's first stream byte
        for (uint64_t x = x0; x < x0 + part.len;) {
            const uint64_t k = (x - m.from) / m.piece_bytes, piece_end = m.from + (k + 1) * m.piece_bytes;
            const uint64_t upto = std::min<uint64_t>(piece_end, x0 + part.len);
            if (k >= m.piece_ev.size() || hipEventSynchronize(m.piece_ev[k]) != hipSuccess) return false;
            h_.update(m.host_of(x, seg.org), (size_t)(upto - x));
            x = upto;
        }
        if (part.ends && (uint32_t)h_.digest() != part.expect) {
            uint32_t seen = bad_frame.load();
            while (part.frame < seen && !bad_frame.compare_exchange_weak(seen, part.frame)) {
            }
        }
        return true;
    }
    void loop() {
        (void)hipSetDevice(device_);
        pin_to_device_node(device_);
        MeterScope meter_scope(meter_);
        hipStream_t st = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        char *pin[2] = {nullptr, nullptr};
        size_t pin_cap[2] = {0, 0};
        bool ready = false;  // (stream, events and buffers are made when the first frame needs them)
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !queue_.empty() || stop_; });
                if (queue_.empty()) break;
                job = std::move(queue_.front());
                queue_.pop_front();
                if (stop_) {  // (the decoder left early: nothing more is hashed or handed out)
                    sink_->give(job.seg.buf, job.seg.cap);
                    continue;
                }
                busy_ = true;
                cv_.notify_all();
            }
            bool ok = true;
            // the consumer wants the decoded bytes on the host (string columns): the mirror is made here, in pieces, and the hash
            // reads it — otherwise the bytes come back through this stage's own two buffers, as before
            const bool from_mirror = !job.parts.empty() && sink_->mirror_now(job.seg, kPiece) && job.seg.mirror && job.seg.mirror->piece_bytes;
            if (!job.parts.empty() && !ready && !from_mirror) {
                ok = stream_pool()->take(device_, &st) == hipSuccess;
                for (int i = 0; i < 2 && ok; i++) {
                    ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
                    pin_cap[i] = kPiece + 64;
                    if (ok) ok = (pin[i] = global_pool()->take(&pin_cap[i])) != nullptr;
                }
                ready = ok;
                if (!ok) fail("out of pinned host memory (or streams) for the checksum of a zstd frame");
            }
            // A frame that lies inside the segment is nobody's business but its own: helper threads take such frames in turn
            // (a stream, two pinned buffers and a hash state each) while this thread folds the frame that spans segments — a file
            // of many checksummed frames (pzstd, the seekable format) is hashed at several cores' rate, not one's.
            std::vector<size_t> whole;
            for (size_t i = 0; i < job.parts.size(); i++)
                if (job.parts[i].begins && job.parts[i].ends) whole.push_back(i);
            std::vector<std::thread> helpers;
            std::atomic<size_t> next{0};
            std::atomic<bool> helpers_ok{true};
            std::atomic<uint32_t> bad_frame{~0u};
            if (ok && whole.size() >= 2) {
                static const size_t helpers_max = getenv("EXG_ZSTD_HASH_HELPERS") ? std::max(1, atoi(getenv("EXG_ZSTD_HASH_HELPERS"))) : kHelpers;
                const size_t nh = std::min<size_t>(helpers_max, whole.size());
                for (size_t t = 0; t < nh; t++)
                    helpers.emplace_back([&, this] {
                        (void)hipSetDevice(device_);
                        pin_to_device_node(device_);
                        MeterScope scope(meter_);
                        hipStream_t hst = nullptr;
                        hipEvent_t hev[2] = {nullptr, nullptr};
                        char *hpin[2] = {nullptr, nullptr};
                        size_t hcap[2] = {kPiece + 64, kPiece + 64};
                        // (a helper that cannot get its stream, events or pinned blocks is no helper: it claims nothing, and what the
                        // helpers leave is hashed by this stage's own thread below — only a copy that really fails fails the job)
                        exg::Xxh64 h;
                        if (from_mirror) {  // (nothing to set up: the pieces arrive in the segment's mirror)
                            for (size_t k; (k = next.fetch_add(1)) < whole.size();)
                                if (!hash_part_mirror(job.parts[whole[k]], job.seg, h, bad_frame)) {
                                    helpers_ok.store(false);
                                    break;
                                }
                            return;
                        }
                        bool up = stream_pool()->take(device_, &hst) == hipSuccess;
                        for (int i = 0; i < 2 && up; i++)
                            up = hipEventCreateWithFlags(&hev[i], hipEventDisableTiming) == hipSuccess && (hpin[i] = global_pool()->take(&hcap[i])) != nullptr;
                        for (size_t k; up && (k = next.fetch_add(1)) < whole.size();)
                            if (!hash_part(job.parts[whole[k]], hst, hpin, hev, h, bad_frame)) up = false, helpers_ok.store(false);
                        if (hst && hipStreamSynchronize(hst) != hipSuccess) (void)hipGetLastError();
                        for (int i = 0; i < 2; i++) {
                            if (hpin[i]) global_pool()->give(hpin[i], hcap[i]);
                            if (hev[i]) (void)hipEventDestroy(hev[i]);
                        }
                        if (hst) stream_pool()->give(device_, hst);
                    });
            }
            for (size_t i = 0; i < job.parts.size() && ok; i++) {
                if (!helpers.empty() && job.parts[i].begins && job.parts[i].ends) continue;
                ok = from_mirror ? hash_part_mirror(job.parts[i], job.seg, h_, bad_frame) : hash_part(job.parts[i], st, pin, ev, h_, bad_frame);
                if (!ok) fail("copying a zstd frame back for its checksum failed");
            }
            if (!helpers.empty()) {  // the whole frames the helpers have not claimed (all of them when no helper came up)
                exg::Xxh64 h;        // (h_ carries the frame that goes on into the next segment)
                for (size_t k; ok && helpers_ok.load() && (k = next.fetch_add(1)) < whole.size();) {
                    ok = from_mirror ? hash_part_mirror(job.parts[whole[k]], job.seg, h, bad_frame) : hash_part(job.parts[whole[k]], st, pin, ev, h, bad_frame);
                    if (!ok) fail("copying a zstd frame back for its checksum failed");
                }
            }
            for (auto &t : helpers) t.join();
            if (ok && !helpers_ok.load()) {
                ok = false;
                fail("copying a zstd frame back for its checksum failed");
            }
            if (ok && bad_frame.load() != ~0u) fail("Restored data doesn't match checksum (zstd frame " + std::to_string(bad_frame.load()) + ")");
            if (st && hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();  // (no copy may still read the segment)
            bool pushed = false;
            if (ok) pushed = sink_->push(std::move(job.seg));
            else sink_->give(job.seg.buf, job.seg.cap);
            std::lock_guard<std::mutex> g(mu_);
            if (ok && !pushed) gone_ = true;
            busy_ = false;
            cv_.notify_all();
        }
        for (int i = 0; i < 2; i++) {
            if (pin[i]) global_pool()->give(pin[i], pin_cap[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
        if (st) stream_pool()->give(device_, st);
    }
    int device_;
    SegmentSink *sink_;
    MemMeter *meter_;
    size_t depth_;
    exg::Xxh64 h_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Job> queue_;
    bool stop_ = false, failed_ = false, busy_ = false, gone_ = false;
    std::string error_;
    std::thread thread_;
};

class ZstdProducer : public SegmentProducer {
public:
    ZstdProducer(exg_reader *r, int fd, uint64_t n, uint64_t c_begin, uint64_t c_end, uint64_t target, const std::string &path, uint64_t reserve,
                 const uint64_t mark_at[2])
        : device_(r->device), fd_(fd), n_(n), c_begin_(c_begin), c_end_(c_end), target_(std::max<uint64_t>(target, 128u << 10)), path_(path),
          reserve_((reserve + 15) & ~15ull) {
        if (mark_at) mark_at_[0] = mark_at[0], mark_at_[1] = mark_at[1];
        // A round's kernels are dependent launches, each as long as its slowest block's chain (~10 ms whatever the size, until
        // the chip is full): without a memory cap a round decodes ~1 GiB instead of one device batch (exg_rd_gzip.cpp does
        // the same for the rounds of one big gzip member)
        if (!r->mem_cap && target_ >= (128ull << 20)) target_ = std::max<uint64_t>(target_, 1ull << 30);
        read_ahead_ = !r->mem_cap && !getenv("EXG_ZSTD_NO_READAHEAD");
        if (const char *e = getenv("EXG_STREAM_ROUND_OUT")) target_ = std::max<uint64_t>(128u << 10, strtoull(e, nullptr, 10));
    }
    int run(SegmentSink &sink, std::string *err) override;

private:
    int device_, fd_;
    uint64_t n_, c_begin_, c_end_, target_;
    uint64_t mark_at_[2] = {~0ull, ~0ull};  // frame offsets whose decoded positions the reader wants to know (a shard's boundaries)
    std::string path_;
    uint64_t reserve_;
    bool read_ahead_ = false;  // the next round's compressed bytes travel while this round is decoded (a second window: not under a cap)
};

#define ZS_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            *err = std::string(#expr " failed: ") + hipGetErrorString(_e);                         \
            return EXG_E_HIP;                                                                      \
        }                                                                                          \
    } while (0)

int ZstdProducer::run(SegmentSink &sink, std::string *err) {
    const double t_run0 = now_s();
    struct RunTrace {
        double t0;
        ~RunTrace() { if (getenv("EXG_TRACE")) fprintf(stderr, "[exg] zstd producer: gone %.1f ms after it began\n", (now_s() - t0) * 1e3); }
    } run_trace{t_run0};
    // Round 6: a consumer that pulls string columns gets rounds of 640 MiB instead of 1 GiB.  Behind the ramp the decode and the link
    // run at about the same rate (a 1 GiB round every ~23 ms, its bytes + vectors 22 ms of link), so the drain ends when the link
    // has caught up with everything the decoder had ready before it: smaller rounds start the link earlier and leave a smaller last
    // segment (4 GB: 160 + 320 + 5 x 640 + 135 MiB).  A 4 GB frame into DataChunks, three boxes, A/B inside each: 1 GiB rounds 133-138
    // ms, 768 MiB 126-133, 640 MiB 125-129, 512 MiB 141 (the decode itself slows down: COUNT(*) 99 ms against 91; 640 MiB: 93.5).
    // COUNT(*) keeps its 1 GiB rounds
    if (target_ == (1ull << 30) && sink.mirror_wanted() && !getenv("EXG_STREAM_ROUND_OUT")) target_ = 640ull << 20;
    hipStream_t st = nullptr;
    if (stream_pool()->take(device_, &st) != hipSuccess) {
        *err = "cannot create a stream for the zstd decoder";
        return EXG_E_HIP;
    }
    struct StreamBack {
        int dev;
        hipStream_t s;
        ~StreamBack() { stream_pool()->give(dev, s); }  // (synchronises it)
    } stream_back{device_, st};
    zst::Index idx;
    const double t_idx0 = now_s();
    // The walk over the frame / block headers reads the file with two small preads per block.  (It used to read a mapping of
    // the whole file: a page fault per 64 KiB — 80-100 ms per 2 GB on this thread, 20 ms with eight threads taking the faults
    // first — and 49 ms to unmap it again at the end, a quarter of what a 4 GB frame took end to end.)
    std::string damage;  // a malformed or truncated stream: the rows in front of the damage first, like a streaming decoder
    // Round 5: a whole big file begins its first round on a PREFIX of the index (the blocks of one round and two more: ~6 ms of a
    // 4 GB frame's 23) while a helper walks the whole file; the whole index takes the prefix's place behind the first round's
    // entropy stages (finish_index), before anything asks for a block behind the prefix.  EXG_ZSTD_NO_INDEX_OVERLAP: never (A/B).
    struct Walker {
        std::thread th;
        zst::Index full;
        bool ok = false, pending = false;
        ~Walker() {
            if (th.joinable()) th.join();
        }
    } walker;
    // the round's compressed bytes: [the blocks whose tables are repeated (at most four) | the round's own, from a 16-byte boundary]
    static constexpr uint64_t kSideSlot = (zst::kBlockMax + 64 + 15) & ~15ull, kSide = 4 * kSideSlot;
    // (three windows of compressed bytes in turn: while round n is decoded out of one — and round n - 1 still executes out of
    // the one before — a helper thread reads round n + 1's bytes into the third and sends them on a stream of its own: 9 ms of a
    // round otherwise spent in front of the decode)
    hipStream_t st_io = nullptr;
    if (read_ahead_ && stream_pool()->take(device_, &st_io) != hipSuccess) st_io = nullptr, read_ahead_ = false;
    struct StreamBack2 {
        int dev;
        hipStream_t s;
        ~StreamBack2() { if (s) stream_pool()->give(dev, s); }
    } stream_back2{device_, st_io};
    PoolBuf d_comp_a(device_, st), d_comp_b(device_, st), d_comp_c(device_, st), d_hist(device_, st);
    PoolBuf *d_comps[3] = {&d_comp_a, &d_comp_b, &d_comp_c};
    struct Pin {
        char *p = nullptr;
        size_t cap = 0;
        ~Pin() { if (p) global_pool()->give(p, cap); }
        bool ensure(size_t n) {
            if (n <= cap) return true;
            if (p) global_pool()->give(p, cap);
            size_t want = n;
            p = global_pool()->take(&want);
            cap = p ? want : 0;
            return p != nullptr;
        }
    } pins[3];
    size_t d_comp_caps[3] = {0, 0, 0}, d_hist_cap = 4096;
    struct Ahead {  // the window a helper thread is filling (or has filled) for the round that begins with block `b0`
        std::thread th;
        uint64_t b0 = ~0ull;
        uint64_t lo = 0, hi = 0;  // the file bytes it holds (a window planned without the whole index is a guess: the round checks)
        int slot = 0;
        bool ok = false, hip_failed = false;
        hipEvent_t ev = nullptr;
        hipStream_t io = nullptr;  // the stream the helper's H2D slices were enqueued on (pread_parallel only enqueues them)
        // Whatever way run() is left — the consumer closed early (a LIMIT query), a cancelled sink, an error — a window may
        // still be travelling: its pinned block and its device window are locals declared in front of this one, i.e. released
        // to the process-wide pools AFTER this destructor.  The copies have to be over by then, or another reader of the same
        // device takes a block that a DMA is still reading from / writing to (advisor, round 4).
        ~Ahead() {
            if (th.joinable()) th.join();
            if (io) (void)hipStreamSynchronize(io);
            if (ev) (void)hipEventDestroy(ev);
        }
    } ahead;
    ahead.io = st_io;
    if (read_ahead_ && hipEventCreateWithFlags(&ahead.ev, hipEventDisableTiming) != hipSuccess) read_ahead_ = false;
    auto ensure_window = [&](int slot, uint64_t comp_len) -> bool {
        if (kSide + comp_len + 64 > d_comp_caps[slot]) {
            d_comp_caps[slot] = (size_t)(kSide + comp_len + comp_len / 4 + 64);
            if (!d_comps[slot]->take(d_comp_caps[slot])) return false;
        }
        return pins[slot].ensure((size_t)(kSide + comp_len + 64));
    };
    // file bytes [lo, lo + len) -> window `slot` on the helper thread (pread + H2D on st_io), for the round that begins with block b0
    auto launch_ahead = [&](uint64_t b0_of, uint64_t lo, uint64_t len, int slot) {
        ahead.b0 = b0_of;
        ahead.lo = lo, ahead.hi = lo + len;
        ahead.slot = slot;
        ahead.ok = ahead.hip_failed = false;
        char *h_dst = pins[slot].p + kSide, *d_dst = (char *)d_comps[slot]->p + kSide;
        ahead.th = std::thread([this, &ahead, lo, len, h_dst, d_dst, st_io] {
            (void)hipSetDevice(device_);
            bool hf = false;
            bool ok = !len || pread_parallel(device_, fd_, lo, (size_t)len, h_dst, d_dst, st_io, &hf);
            if (ok && hipMemsetAsync(d_dst + len, 0, 64, st_io) != hipSuccess) ok = false, hf = true;
            if (ok && hipEventRecord(ahead.ev, st_io) != hipSuccess) ok = false, hf = true;
            ahead.hip_failed = hf;
            ahead.ok = ok;
        });
    };
    static constexpr uint64_t kFirstRound = ~1ull;  // launch_ahead's b0 while the first block is not known yet
    {
        const bool no_index_overlap = getenv("EXG_ZSTD_NO_INDEX_OVERLAP") != nullptr;  // (read per stream: the tests set them inside one process)
        const uint64_t overlap_min = getenv("EXG_ZSTD_INDEX_OVERLAP_MIN") ? strtoull(getenv("EXG_ZSTD_INDEX_OVERLAP_MIN"), nullptr, 10) : (256ull << 20);  // (tests: 0)
        const bool whole_file = c_begin_ == 0 && c_end_ >= n_ && mark_at_[0] == ~0ull && mark_at_[1] == ~0ull;
        bool have = false, ok = false;
        if (whole_file && read_ahead_ && !no_index_overlap && n_ >= overlap_min) {
            walker.th = std::thread([&walker, this] { walker.ok = zst::build_index_fd(fd_, n_, walker.full); });
            // ... and the first round's compressed bytes begin to travel before any block is known: the file's first 0.75 x a round's
            // output (a round of a stream that compresses 1.33-fold or better lies inside; the round checks, and reads again if not)
            // (a first round that will be a quarter — plan(): the consumer pulls strings, or the first frame carries a checksum, which
            // its descriptor byte says — wants a quarter of that: the round waits for the whole guess to land)
            uint8_t head[5] = {0, 0, 0, 0, 0};
            const bool head_ok = pread(fd_, head, 5, 0) == 5;
            const bool first_has_checksum = head_ok && head[0] == 0x28 && head[1] == 0xB5 && head[2] == 0x2F && head[3] == 0xFD && ((head[4] >> 2) & 1);
            const uint64_t div_env = getenv("EXG_ZSTD_FIRST_ROUND_DIV") ? std::max<uint64_t>(1, strtoull(getenv("EXG_ZSTD_FIRST_ROUND_DIV"), nullptr, 10)) : 0;
            const uint64_t div = div_env ? div_env : (first_has_checksum || sink.mirror_wanted()) ? 4 : 1;
            const uint64_t first_out = std::min<uint64_t>(target_, std::max<uint64_t>(target_ / div, 16u << 20));
            const uint64_t guess = std::min<uint64_t>(n_, first_out / 2 + first_out / 4);
            if (ensure_window(0, guess)) launch_ahead(kFirstRound, 0, guess, 0);
            bool stopped = false;
            if (zst::build_index_prefix_fd(fd_, n_, first_out + 2 * (uint64_t)zst::kBlockMax, idx, &stopped) && stopped) {
                walker.pending = true;
                have = ok = true;
            } else {  // (the stream is shorter than a round, or its prefix is damaged: the whole walk says what it is)
                walker.th.join();
                idx = std::move(walker.full);
                have = true;
                ok = walker.ok;
            }
        }
        if (!have) ok = zst::build_index_fd(fd_, n_, idx);
        if (!ok) {
            damage = idx.error + " in '" + path_ + "'";
            if (!zst::salvage_index(idx)) {
                *err = damage;
                return EXG_E_PARSE;
            }
        }
    }
    if (getenv("EXG_TRACE"))
        fprintf(stderr, "[exg] zstd producer: index of %.1f MB (%zu blocks%s) %.1f ms\n", n_ / 1e6, idx.blocks.size(), walker.pending ? ": a prefix, the rest on a helper" : "",
                (now_s() - t_idx0) * 1e3);
    // the frames whose first byte lies in [c_begin, c_end) (a shard decodes its own frames and a halo of frames in front)
    uint64_t b_first = 0, n_blocks = 0, b_mark[2] = {~0ull, ~0ull};
    bool marked[2] = {false, false};
    auto scan_frames = [&] {
        b_first = idx.blocks.size(), n_blocks = 0;
        for (const zst::Frame &F : idx.frames) {
            if (F.src_off >= c_begin_ && F.src_off < c_end_) {
                b_first = std::min<uint64_t>(b_first, F.first_block);
                n_blocks = std::max<uint64_t>(n_blocks, (uint64_t)F.first_block + F.n_blocks);
            }
            for (int i = 0; i < 2; i++)
                if (mark_at_[i] != ~0ull && F.src_off >= mark_at_[i] && b_mark[i] == ~0ull) b_mark[i] = F.first_block;
        }
        if (b_first > n_blocks) b_first = n_blocks;
    };
    scan_frames();
    if (ahead.b0 == kFirstRound) ahead.b0 = b_first;
    // the whole index in the prefix's place (block and frame numbers are the same: both walks begin at byte 0)
    auto finish_index = [&](uint64_t blocks_used) -> int {
        if (!walker.pending) return EXG_OK;
        walker.pending = false;
        walker.th.join();
        idx = std::move(walker.full);
        if (!walker.ok) {
            damage = idx.error + " in '" + path_ + "'";
            if (!zst::salvage_index(idx)) {
                *err = damage;
                return EXG_E_PARSE;
            }
        }
        if (idx.blocks.size() < blocks_used) {  // (the file changed between the two walks)
            *err = damage.empty() ? "the zstd stream changed while it was read: '" + path_ + "'" : damage;
            return EXG_E_PARSE;
        }
        scan_frames();
        if (getenv("EXG_TRACE")) fprintf(stderr, "[exg] zstd producer: whole index (%zu blocks) there %.1f ms after the start\n", idx.blocks.size(), (now_s() - t_idx0) * 1e3);
        return EXG_OK;
    };
    auto marks = [&](uint64_t b_next, uint64_t pos) {  // b_next: the next block to be decoded, pos: where its bytes will lie
        for (int i = 0; i < 2; i++)
            if (!marked[i] && mark_at_[i] != ~0ull && (b_next >= b_mark[i] || b_next >= n_blocks)) sink.set_mark(i, pos), marked[i] = true;
    };
    uint64_t ramp_second = ~0ull;  // the block the second round begins with (plan)
    uint64_t first_div_used = 1;   // what the first round's size was divided by (plan)
    // where a round that begins with block b ends, and which file bytes it needs
    auto plan = [&](uint64_t from, uint64_t *to, uint64_t *lo, uint64_t *hi) {
        // The first round is a quarter of the size (never more than a round, never below 16 MiB of one) when rounds overlap and
        // the first frame carries a Content_Checksum: the host's XXH64, which hashes slower than the device decodes, begins after
        // ~25 ms instead of ~60 (223-262 against 247-284 ms on a 4 GB frame; nothing without a checksum: 121-127 against 126 ms).
        static const uint64_t first_div_env = getenv("EXG_ZSTD_FIRST_ROUND_DIV") ? std::max<uint64_t>(1, strtoull(getenv("EXG_ZSTD_FIRST_ROUND_DIV"), nullptr, 10)) : 0;
        // (... and when the consumer pulls string columns: until the first segment is out nothing crosses the link, and behind it the
        // drain is the link's — a 4 GB frame into DataChunks 161-165 -> 153-155 ms)
        const uint64_t first_div =
            first_div_env ? first_div_env : ((!idx.frames.empty() && idx.frames[idx.blocks[from].frame].has_checksum) || sink.mirror_wanted() ? 4 : 1);
        // (and the second round half: a whole round behind the quarter left the hasher idle for ~14 ms of a 4 GB frame's time)
        const uint64_t div = !read_ahead_ ? 1 : from == b_first ? first_div : from == ramp_second && first_div > 1 ? first_div / 2 : 1;
        if (from == b_first) first_div_used = div;
        const uint64_t want_out = std::min<uint64_t>(target_, std::max<uint64_t>(target_ / std::max<uint64_t>(div, 1), 16u << 20));
        uint64_t b1 = from, est = 0;
        while (b1 < n_blocks && (b1 == from || est < want_out)) {
            if (b1 > from && (b1 == b_mark[0] || b1 == b_mark[1])) break;
            const zst::Block &B = idx.blocks[b1];
            est += B.type == 2 ? zst::kBlockMax : B.src_size;  // raw / RLE: src_size is the regenerated size
            b1++;
            if (b1 - from >= 0x7FFFFF00u) break;
        }
        if (from == b_first) ramp_second = b1;
        *to = b1;
        *lo = idx.blocks[from].src_off & ~15ull;
        const zst::Block &BL = idx.blocks[b1 - 1];
        *hi = std::min<uint64_t>(n_, BL.src_off + (BL.type == 1 ? 1 : BL.src_size));
    };
    int cur = 0;
    if (!d_hist.take(d_hist_cap)) {
        *err = "out of device memory";
        return EXG_E_HIP;
    }
    FrameHasher hasher(device_, &sink, tl_meter());
    // A frame inside one round is hashed on the device up to this size, one wavefront per frame at ~0.55 GB/s: the round waits
    // for the slowest of them.  1 MiB = 2 ms of a ~30 ms round; frames of 8 MiB cost 78 ms per 4 GB and frames of 64 MiB 610 ms
    // (5.4 instead of 30.9 GB/s) when this was the decoder's 64 MiB.  Larger frames go to the hasher threads behind the decoder.
    const uint64_t verify_max = getenv("EXG_ZSTD_VERIFY_MAX") ? zst::default_verify_max() : std::min<uint64_t>(zst::default_verify_max(), 1u << 20);
    uint64_t d_pos = 0;             // decoded bytes produced so far
    uint64_t b0 = b_first;          // next block
    uint32_t rep[3] = {1, 4, 8};    // repeat offsets behind block b0 - 1 (of the frame that goes on)
    uint64_t frame_done = 0;        // bytes of the frame that holds block b0 decoded so far (0: it begins with b0)
    uint64_t hist = 0, pad = 0;     // d_hist holds [pad | hist bytes]: the end of that frame's output so far
    bool pushed_last = false;
    // Two rounds overlap (not under a memory cap — the second window is what read_ahead_ stands for): while round n's ~1 900
    // dependent resolve launches run (launch latency: the chip is mostly idle), round n + 1's entropy stages, its scan and the
    // execution of its chunks run on the other stream.  Only round n + 1's resolve needs round n's last bytes (the window): it is
    // enqueued once round n is done, and round n's segment goes out then.  What the next round needs of a round — sizes, repeat offsets —
    // is known when its entropy stages are (decode_round_begin); what needs its bytes is deferred (InFlight::complete).
    hipStream_t st_b = nullptr;
    const bool overlap = read_ahead_ && !getenv("EXG_ZSTD_NO_OVERLAP") && stream_pool()->take(device_, &st_b) == hipSuccess;
    StreamBack2 stream_back3{device_, overlap ? st_b : nullptr};
    struct InFlight {
        zst::Round R;
        zst::RoundCtx *ctx = nullptr;  // between begin and wait
        hipStream_t st = nullptr;
        uint64_t d_pos = 0, hist = 0;  // when the round began
        bool last = false;
        // the window the round leaves (0: none — its last frame ends)
        uint64_t nh = 0, npad = 0, window = 0;
        bool active = false;
        // the ~1 900 resolve launches of a round take this long to ISSUE (~5 us each: 10 ms): a thread of their own issues them,
        // so that the round behind begins meanwhile (its entropy stages and execution then run beside these launches on the chip)
        std::thread launcher;
        int launch_rc = 0;
        std::string launch_err;
        ~InFlight() {
            if (launcher.joinable()) launcher.join();
            if (ctx) zst::decode_round_abandon(ctx);
        }
    } fl[2];
    int n_round = 0;
    InFlight *prev = nullptr;
    // round `F` is done on the device: its window into d_hist, its segment to the hasher / the consumer
    auto complete = [&](InFlight &F) -> int {
        F.active = false;
        if (F.launcher.joinable()) F.launcher.join();
        if (F.launch_rc) {  // (the launcher has disposed of the context)
            *err = F.launch_err + " in '" + path_ + "'";
            return F.launch_rc;
        }
        zst::RoundCtx *ctx = F.ctx;
        F.ctx = nullptr;
        int rc;
        {
            TraceRange range("exg: zstd round (wait)");
            rc = zst::decode_round_wait(F.R, ctx);
        }
        if (rc) {
            *err = std::string(exg_last_error_message()) + " in '" + path_ + "'";
            return rc;
        }
        zst::Round &R = F.R;
        const uint64_t H = R.history;  // pad + hist
        Segment seg;
        seg.buf = R.d_buf;
        seg.cap = R.alloc;
        seg.org = (int64_t)F.d_pos - (int64_t)H - (int64_t)reserve_;
        seg.lo = F.d_pos - F.hist;
        seg.start = F.d_pos;
        seg.hi = F.d_pos + R.produced;
        seg.last = F.last;
        const uint8_t *content = (const uint8_t *)R.d_buf + reserve_;  // buffer coordinate 0
        FrameHasher::Job job;
        for (const zst::RoundFrame &rf : R.frames) {
            const zst::Frame &Fr = idx.frames[rf.frame_id];
            if (Fr.has_checksum && !rf.verified) {
                // its bytes come back in pieces and are hashed beside the next round's decode (FrameHasher), before the segment goes out
                FrameHasher::Part part;
                part.d_src = content + rf.out_off;
                part.len = rf.out_size;
                part.frame = rf.frame_id;
                part.expect = Fr.checksum;
                part.begins = rf.begins;
                part.ends = rf.ends;
                job.parts.push_back(part);
            }
        }
        if (F.nh) {
            if (F.npad + F.nh > d_hist_cap) {
                // (the old window's bytes are in this round's buffer too: a new block loses nothing)
                d_hist_cap = (size_t)(F.npad + std::max<uint64_t>(F.nh, std::min<uint64_t>(F.window, zst::kWindowMax)) + 64);
                if (!d_hist.take(d_hist_cap)) {
                    sink.give(seg.buf, seg.cap);
                    *err = "out of device memory for the window of a zstd frame";
                    return EXG_E_HIP;
                }
            }
            // the last nh bytes of [history | produced] (nh <= history of this frame + what the round added to it)
            const uint8_t *src = content + H + R.produced - F.nh;
            hipError_t he = hipMemcpyAsync((char *)d_hist.p + F.npad, src, F.nh, hipMemcpyDeviceToDevice, F.st);
            if (he == hipSuccess) he = hipStreamSynchronize(F.st);
            if (he != hipSuccess) {
                sink.give(seg.buf, seg.cap);
                *err = std::string("keeping the window of a zstd frame failed: ") + hipGetErrorString(he);
                return EXG_E_HIP;
            }
        }
        pushed_last = seg.last;
        job.seg = std::move(seg);
        seg.buf = nullptr;
        if (!hasher.submit(std::move(job))) {
            if (hasher.gone()) return -1;  // the consumer closed the stream
            *err = hasher.error() + " in '" + path_ + "'";
            return EXG_E_PARSE;
        }
        return EXG_OK;
    };
    // (an error of round n + 1 is reported behind round n's rows: round n goes out first)
    auto flush_prev = [&]() -> int {
        if (!prev) return EXG_OK;
        InFlight *f = prev;
        prev = nullptr;
        return complete(*f);
    };
    while (b0 < n_blocks && !sink.cancelled()) {
        marks(b0, d_pos);
        InFlight &F = fl[n_round & 1];
        hipStream_t st_r = overlap && (n_round & 1) ? st_b : st;
        n_round++;
        // ---- the round's blocks: about one segment of output (a block regenerates at most 128 KiB); a round ends at a mark
        uint64_t b1 = b0, c_lo = 0, c_hi = 0;
        plan(b0, &b1, &c_lo, &c_hi);
        F.R = zst::Round();
        zst::Round &R = F.R;
        // blocks in front of the round whose tables its blocks repeat
        std::vector<uint64_t> extra_ids;
        auto local_of = [&](uint32_t g) -> uint32_t {
            if (g == zst::kNone) return zst::kNone;
            if (g >= b0) return (uint32_t)(g - b0) + (uint32_t)extra_ids.size();  // (patched below once the extras are known)
            for (size_t i = 0; i < extra_ids.size(); i++)
                if (extra_ids[i] == g) return (uint32_t)i;
            extra_ids.push_back(g);
            return (uint32_t)extra_ids.size() - 1;
        };
        for (uint64_t b = b0; b < b1; b++) {  // first pass: which sources
            const zst::Block &B = idx.blocks[b];
            if (B.type != 2) continue;
            if (B.huf_src != zst::kNone && B.huf_src < b0) (void)local_of(B.huf_src);
            for (int t = 0; t < 3; t++)
                if (B.nseq && B.tbl_src[t] != zst::kNone && B.tbl_src[t] < b0) (void)local_of(B.tbl_src[t]);
        }
        const uint32_t nx = (uint32_t)extra_ids.size();
        if (nx > 4) {
            *err = "internal: a zstd round repeats the tables of more than four earlier blocks";
            return EXG_E_INVALID_ARG;
        }
        const uint64_t comp_len = c_hi - c_lo;
        // this round's window: the one the helper thread has filled, or a read of its own
        bool have = false;
        const double t_join0 = now_s();
        if (ahead.th.joinable()) {
            ahead.th.join();
            if (ahead.b0 == b0 && ahead.ok && ahead.lo == c_lo && ahead.hi >= c_hi) {
                cur = ahead.slot;
                ZS_HIP(hipStreamWaitEvent(st_r, ahead.ev, 0));
                have = true;
            } else if (ahead.hip_failed) {
                *err = "hipMemcpyAsync failed";
                return EXG_E_HIP;
            } else {
                (void)hipStreamSynchronize(st_io);  // (a window nobody wants: let it land before its buffers are used again)
            }
            ahead.b0 = ~0ull;
        }
        if (!have) {
            // (a read of its own goes into the window the round in flight may be reading: that round first)
            if (int rc = flush_prev()) return rc < 0 ? EXG_OK : rc;
            if (!ensure_window(cur, comp_len)) {
                *err = "out of device / pinned memory for the compressed bytes of '" + path_ + "'";
                return EXG_E_HIP;
            }
        }
        PoolBuf &d_comp = *d_comps[cur];
        auto &pin = pins[cur];
        R.blocks.reserve(nx + (b1 - b0));
        for (uint32_t i = 0; i < nx; i++) {
            zst::Block E = idx.blocks[extra_ids[i]];
            const uint64_t sz = E.type == 2 ? E.src_size : 1;
            const size_t want = (size_t)std::min<uint64_t>(sz, kSideSlot);
            size_t got = 0;
            while (got < want) {
                const ssize_t k = pread(fd_, pin.p + i * kSideSlot + got, want - got, (off_t)(E.src_off + got));
                if (k <= 0) {
                    if (k < 0 && errno == EINTR) continue;
                    *err = "short read of '" + path_ + "'";
                    return EXG_E_IO;
                }
                got += (size_t)k;
            }
            E.src_off = i * kSideSlot;
            E.huf_src = E.tbl_src[0] = E.tbl_src[1] = E.tbl_src[2] = zst::kNone;  // (a source is only read, never decoded)
            R.blocks.push_back(E);
        }
        if (nx) ZS_HIP(hipMemcpyAsync(d_comp.p, pin.p, nx * kSideSlot, hipMemcpyHostToDevice, st_r));
        bool hip_failed = false;
        if (!have && comp_len && !pread_parallel(device_, fd_, c_lo, (size_t)comp_len, pin.p + kSide, (char *)d_comp.p + kSide, st_r, &hip_failed)) {
            *err = hip_failed ? "hipMemcpyAsync failed" : "short read of '" + path_ + "'";
            return hip_failed ? EXG_E_HIP : EXG_E_IO;
        }
        // the round behind this one: its bytes begin to travel now, into the next window (three in turn: the round in flight reads
        // the one before this round's)
        auto start_ahead = [&](bool guess) {
            if (!(read_ahead_ && b1 < n_blocks)) return;
            uint64_t nb1 = 0, nlo = 0, nhi = 0;
            if (guess) {
                // the index is still a prefix: the round behind this one begins with block b1 (in the prefix) and, with as many
                // blocks, is about as long as this one (twice, behind a quarter) — 5 % more of the file travel; a window that turns out short is read again
                nlo = idx.blocks[b1].src_off & ~15ull;
                // (behind a first round of a quarter comes one of a half)
                const uint64_t like = first_div_used > 1 ? 2 * comp_len : comp_len;
                nhi = std::min<uint64_t>(n_, nlo + like + like / 20 + (1u << 20));
            } else {
                plan(b1, &nb1, &nlo, &nhi);
            }
            const int other = (cur + 1) % 3;
            if (ensure_window(other, nhi - nlo)) launch_ahead(b1, nlo, nhi - nlo, other);
        };
        // (the first round of a file whose index is still a prefix: the round behind it is planned once the whole index is there,
        // behind this round's entropy stages — its bytes begin to travel now all the same, as a guess)
        const bool ahead_deferred = walker.pending;
        start_ahead(ahead_deferred);
        if (!have) ZS_HIP(hipMemsetAsync((char *)d_comp.p + kSide + comp_len, 0, 64, st_r));
        auto remap = [&](uint32_t g) -> uint32_t {
            if (g == zst::kNone) return zst::kNone;
            if (g >= b0) return (uint32_t)(g - b0) + nx;
            for (uint32_t i = 0; i < nx; i++)
                if (extra_ids[i] == g) return i;
            return zst::kNone;
        };
        for (uint64_t b = b0; b < b1; b++) {
            zst::Block B = idx.blocks[b];
            B.src_off = kSide + (B.src_off - c_lo);
            B.huf_src = remap(B.huf_src);
            for (int t = 0; t < 3; t++) B.tbl_src[t] = remap(B.tbl_src[t]);
            R.blocks.push_back(B);
        }
        R.n_extra = nx;
        // the frames (or parts of frames) in the round
        for (uint64_t b = b0; b < b1;) {
            const uint32_t f = idx.blocks[b].frame;
            const zst::Frame &Fr = idx.frames[f];
            const uint64_t f_end = (uint64_t)Fr.first_block + Fr.n_blocks, e = std::min<uint64_t>(b1, f_end);
            zst::RoundFrame rf;
            rf.first_block = (uint32_t)(b - b0) + nx;
            rf.n_blocks = (uint32_t)(e - b);
            rf.frame_id = f;
            rf.begins = b == Fr.first_block;
            rf.ends = e == f_end;
            rf.history = rf.begins ? 0 : hist;
            rf.has_checksum = Fr.has_checksum;
            rf.checksum = Fr.checksum;
            R.frames.push_back(rf);
            b = e;
        }
        R.rep_in[0] = rep[0], R.rep_in[1] = rep[1], R.rep_in[2] = rep[2];
        R.d_comp = d_comp.p;
        R.d_history = nullptr;  // (set when the round in front is done: d_hist may move)
        // the first history byte is stream byte d_pos - hist: `pad` (unused) bytes in front of it put it on the 16-byte grid the
        // decoder's stores and the scan's loads follow (buffer coordinate 0 is 16-byte aligned, and an address must be
        // congruent to its stream offset: pad = (d_pos - hist) & 15, also when nothing is kept)
        R.history = pad + hist;
        R.front_reserve = reserve_;
        R.verify_max = verify_max;
        R.first_block_id = b0;
        R.comp_base = c_lo - kSide;
        F.st = st_r;
        F.d_pos = d_pos;
        F.hist = hist;
        int rc;
        const double t_dec0 = now_s();
        {
            TraceRange range("exg: zstd round (entropy stages)");
            rc = zst::decode_round_begin(R, st_r, &F.ctx);
        }
        const double t_dec1 = now_s();
        if (rc) {
            const std::string msg = std::string(exg_last_error_message()) + " in '" + path_ + "'";
            if (int rp = flush_prev()) return rp < 0 ? EXG_OK : rp;
            *err = msg;
            return rc;
        }
        if (ahead_deferred) {
            if (int ri = finish_index(b1)) {
                zst::decode_round_abandon(F.ctx);
                F.ctx = nullptr;
                return ri;
            }
        }
        // ---- frames: sizes
        std::string size_error;
        for (const zst::RoundFrame &rf : R.frames) {
            const zst::Frame &Fr = idx.frames[rf.frame_id];
            const uint64_t before = rf.begins ? 0 : frame_done;
            if (rf.ends && Fr.content_size != ~0ull && Fr.content_size != before + rf.out_size) {
                size_error = "Data corruption detected (zstd frame " + std::to_string(rf.frame_id) + " regenerates " + std::to_string(before + rf.out_size) +
                             " bytes, its header says " + std::to_string(Fr.content_size) + ") in '" + path_ + "'";
                break;
            }
        }
        // this round's chunks execute beside the round in front's resolve launches (they need nothing of it)
        if (size_error.empty()) {
            TraceRange range("exg: zstd round (execution enqueued)");
            zst::RoundCtx *ctx = F.ctx;
            F.ctx = nullptr;
            rc = zst::decode_round_enqueue_exec(R, ctx);
            if (!rc) F.ctx = ctx;
            if (rc) {
                const std::string msg = std::string(exg_last_error_message()) + " in '" + path_ + "'";
                if (int rp = flush_prev()) return rp < 0 ? EXG_OK : rp;
                *err = msg;
                return rc;
            }
        }
        // the round in front: done by now or soon — its window is this round's history
        if (int rp = flush_prev()) return rp < 0 ? EXG_OK : rp;
        if (!size_error.empty()) {
            *err = size_error;
            return EXG_E_PARSE;
        }
        R.d_history = d_hist.p;
        F.launch_rc = 0;
        if (overlap) {
            MemMeter *const meter = tl_meter();
            F.launcher = std::thread([this, &F, meter] {
                (void)hipSetDevice(device_);
                MeterScope scope(meter);
                TraceRange range("exg: zstd round (resolve enqueued)");
                zst::RoundCtx *ctx = F.ctx;
                F.ctx = nullptr;
                F.launch_rc = zst::decode_round_enqueue_resolve(F.R, ctx);
                if (F.launch_rc) F.launch_err = exg_last_error_message();
                else F.ctx = ctx;
            });
        } else {
            TraceRange range("exg: zstd round (resolve enqueued)");
            zst::RoundCtx *ctx = F.ctx;
            F.ctx = nullptr;
            rc = zst::decode_round_enqueue_resolve(R, ctx);
            if (!rc) F.ctx = ctx;
            if (rc) {
                *err = std::string(exg_last_error_message()) + " in '" + path_ + "'";
                return rc;
            }
        }
        F.active = true;
        // ---- what the next round needs of this one
        const zst::RoundFrame &lastf = R.frames.back();
        const uint64_t last_before = lastf.begins ? 0 : frame_done;
        F.nh = F.npad = F.window = 0;
        if (lastf.ends) {
            frame_done = 0, hist = 0, pad = (d_pos + R.produced) & 15;
            rep[0] = 1, rep[1] = 4, rep[2] = 8;
        } else {
            frame_done = last_before + lastf.out_size;
            rep[0] = R.rep_out[0], rep[1] = R.rep_out[1], rep[2] = R.rep_out[2];
            const uint64_t window = idx.frames[lastf.frame_id].window;
            const uint64_t nh = std::min<uint64_t>(std::max<uint64_t>(window, 1), frame_done);
            const uint64_t end_pos = d_pos + R.produced;  // stream offset behind this round
            F.nh = nh, F.npad = (end_pos - nh) & 15, F.window = window;
            hist = nh, pad = F.npad;
        }
        d_pos += R.produced;
        b0 = b1;
        F.last = b0 >= n_blocks;
        if (getenv("EXG_TRACE"))
            fprintf(stderr, "[exg] zstd producer: round of %.1f MB compressed: window %.1f ms, entropy stages %.1f ms, round in front + enqueue %.1f ms\n",
                    comp_len / 1e6, (t_dec0 - t_join0) * 1e3, (t_dec1 - t_dec0) * 1e3, (now_s() - t_dec1) * 1e3);
        prev = &F;
        if (!overlap) {
            if (int rp = flush_prev()) return rp < 0 ? EXG_OK : rp;
        }
    }
    if (int rp = flush_prev()) return rp < 0 ? EXG_OK : rp;
    if (getenv("EXG_TRACE")) fprintf(stderr, "[exg] zstd producer: last segment out %.1f ms after it began\n", (now_s() - t_run0) * 1e3);
    // The checksums still being folded: a mismatch is this thread's result, which the reader looks at once the last
    // segment's rows have been handed out (DecodedSource::finish) — where a streaming decoder reports it too.
    if (!hasher.drain()) {
        *err = hasher.error() + " in '" + path_ + "'";
        return EXG_E_PARSE;
    }
    if (!damage.empty() && c_end_ >= n_ && !sink.cancelled()) {  // (the shard that reads to the end of the file reports it)
        *err = damage;
        return EXG_E_PARSE;
    }
    marks(n_blocks, d_pos);
    if (!pushed_last && !sink.cancelled()) {  // no block at all (an empty file, skippable frames only): the stream still ends
        Segment seg;
        seg.cap = (size_t)(reserve_ + 16 + 64);
        seg.buf = sink.take(seg.cap);
        if (!seg.buf) {
            *err = "out of device memory";
            return EXG_E_HIP;
        }
        seg.org = (int64_t)(d_pos & ~15ull) - (int64_t)reserve_;
        seg.lo = seg.start = seg.hi = d_pos;
        seg.last = true;
        hipError_t he = hipMemsetAsync((char *)seg.buf + reserve_, 0, 16 + 64, st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        if (he != hipSuccess) {
            sink.give(seg.buf, seg.cap);
            *err = std::string("hipMemsetAsync failed: ") + hipGetErrorString(he);
            return EXG_E_HIP;
        }
        (void)sink.push(std::move(seg));
    }
    return EXG_OK;
}

}  // namespace

std::unique_ptr<SegmentProducer> make_zstd_producer(exg_reader *r, int fd, uint64_t n, uint64_t c_begin, uint64_t c_end, uint64_t target,
                                                    const std::string &path, uint64_t reserve, const uint64_t mark_at[2]) {
    return std::unique_ptr<SegmentProducer>(new ZstdProducer(r, fd, n, c_begin, c_end, target, path, reserve, mark_at));
}

// Shard `shard_index` of `shard_count` of a zstd file: a FRAME belongs to the shard in whose 1/shard_count of the file's bytes
// it begins (a file of one frame — what the zstd CLI writes — is one shard's; pzstd / seekable-format files have many).
// out: the frames to decode are those that begin in [*c_begin, *c_end) — a halo of frames in front of the shard's own (about
// `halo_want` bytes of content, as far as the frame headers tell), then its own; own_lo / own_hi: where its own begin / end.
int plan_zstd_shard(exg_reader *r, int fd, uint64_t n, const std::string &path, uint64_t halo_want, uint64_t header_bytes, uint64_t *c_begin,
                    uint64_t *c_end, uint64_t *own_lo, uint64_t *own_hi, bool *bytes_follow) {
    zst::Index idx;
    // (damage: the shards plan with what lies in front of it; the one that reads to the end of the file reports it behind its rows)
    if (!zst::build_index_fd(fd, n, idx) && !zst::salvage_index(idx)) return fail(r, EXG_E_PARSE, idx.error + " in '" + path + "'");
    const uint64_t lo = (uint64_t)((unsigned __int128)n * r->shard_index / r->shard_count);
    const uint64_t hi = r->shard_index + 1 == r->shard_count ? n : (uint64_t)((unsigned __int128)n * (r->shard_index + 1) / r->shard_count);
    const size_t nf = idx.frames.size();
    size_t f_own = 0, f_hi = 0;
    while (f_own < nf && idx.frames[f_own].src_off < lo) f_own++;
    f_hi = f_own;
    while (f_hi < nf && (hi >= n || idx.frames[f_hi].src_off < hi)) f_hi++;
    auto off_of = [&](size_t f) { return f < nf ? idx.frames[f].src_off : n; };
    auto compressed_size = [&](size_t f) { return off_of(f + 1) - off_of(f); };
    size_t f_begin = f_own;
    uint64_t halo = 0;
    while (f_begin > 0 && halo < halo_want) {
        f_begin--;
        const zst::Frame &F = idx.frames[f_begin];
        halo += F.content_size != ~0ull ? F.content_size : compressed_size(f_begin);
    }
    if (header_bytes && f_begin > 0) {
        // the stream must begin behind the VCF header, or with the file: known only when the frames in front state their sizes
        uint64_t sum = 0;
        bool known = true;
        for (size_t f = 0; f < f_begin; f++) known = known && idx.frames[f].content_size != ~0ull, sum += known ? idx.frames[f].content_size : 0;
        if (!known || sum < header_bytes) f_begin = 0;
    }
    *c_begin = off_of(f_begin);
    *c_end = off_of(f_hi);
    *own_lo = off_of(f_own);
    *own_hi = off_of(f_hi);
    *bytes_follow = false;
    for (size_t f = f_hi; f < nf; f++)
        if (idx.frames[f].content_size != 0) *bytes_follow = true;  // (unknown counts as content)
    return EXG_OK;
}

}  // namespace exg_rd
