// exg_rd_source.cpp — the consumer side of a decoded stream (exg_rd_source.hpp): the queue between a producer thread and
// the scanning thread, and `acquire`, which makes the bytes a device batch needs contiguous in HBM by carrying the
// unconsumed tail of one segment into the room in front of the next.  Replaces the BufReader in front of exon's batch
// readers for compressed inputs (rust/src/arrow_reader.rs:60-91, 116-118).
#include "exg_rd_source.hpp"

#include <algorithm>

namespace exg_rd {

bool SegmentSink::cancelled() const {
    std::lock_guard<std::mutex> g(src->mu_);
    return src->closed_;
}
void *SegmentSink::take(size_t bytes) { return dev_pool()->take(src->device_, bytes ? bytes : 16); }
void SegmentSink::give(void *p, size_t bytes) {
    if (p) dev_pool()->give(src->device_, p, bytes ? bytes : 16);
}
void SegmentSink::set_mark(int id, uint64_t pos) {
    std::lock_guard<std::mutex> g(src->mu_);
    if (id < 0 || id > 1) return;
    src->mark_[id] = pos;
    src->mark_set_[id] = true;
    src->cv_.notify_all();
}
// the segment's own bytes [start, hi) begin to travel to the host (best effort: without pinned memory, a stream or an event the
// consumer copies them itself, as it does for every segment that has no mirror)
static void start_mirror(DecodedSource *src, int device, hipStream_t *d2h, Segment &s, bool capped, uint64_t piece_bytes = 0) {
    if (s.hi <= s.start || s.mirror) return;
    // (calibrated for big inputs only — the first segments of one are >= 64 MiB — and not under a memory cap: the try-out holds 64 MiB)
    if (!*d2h && stream_pool()->take_d2h(device, d2h, /*calibrate=*/s.hi - s.start >= (64ull << 20) && !capped) != hipSuccess) {
        (void)hipGetLastError();
        *d2h = nullptr;
        return;
    }
    auto m = std::make_shared<HostMirror>();
    m->blk = std::make_shared<PinnedBlock>();
    size_t cap = (size_t)((int64_t)s.hi - s.org) + 64;
    m->blk->p = global_pool()->take(&cap);
    if (!m->blk->p) return;
    m->blk->cap = cap;
    m->blk->pooled = true;
    m->blk->n = cap;
    m->from = s.start;
    m->hi = s.hi;
    if (hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        m->ev = nullptr;
        return;
    }
    char *dst = (char *)m->blk->p + ((int64_t)s.start - s.org);
    bool ok = true;
    if (piece_bytes) {
        m->piece_bytes = piece_bytes;
        for (uint64_t o = 0; ok && o < s.hi - s.start; o += piece_bytes) {
            const uint64_t len = std::min<uint64_t>(piece_bytes, s.hi - s.start - o);
            hipEvent_t e = nullptr;
            ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
            if (ok) m->piece_ev.push_back(e);
            ok = ok && hipMemcpyAsync(dst + o, s.at(s.start + o), len, hipMemcpyDeviceToHost, *d2h) == hipSuccess && hipEventRecord(e, *d2h) == hipSuccess;
        }
    } else {
        ok = hipMemcpyAsync(dst, s.at(s.start), s.hi - s.start, hipMemcpyDeviceToHost, *d2h) == hipSuccess;
    }
    if (!ok || hipEventRecord(m->ev, *d2h) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(*d2h);
        return;  // (m's destructor waits for whatever was enqueued)
    }
    s.mirror = std::move(m);
}

bool SegmentSink::mirror_wanted() const { return src->mirror_wanted_.load(std::memory_order_acquire); }
bool SegmentSink::mirror_now(Segment &s, uint64_t piece_bytes) {
    if (!mirror_wanted()) return false;
    std::lock_guard<std::mutex> g(src->mirror_mu_);  // (the d2h stream is made by whoever comes first: the producer's thread or a stage behind it)
    start_mirror(src, src->device_, &src->d2h_stream_, s, src->meter_ && src->meter_->cap, piece_bytes);
    return s.mirror != nullptr;
}

bool SegmentSink::push(Segment &&s) {
    if (src->mirror_wanted_.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> g(src->mirror_mu_);
        start_mirror(src, src->device_, &src->d2h_stream_, s, src->meter_ && src->meter_->cap);
    }
    std::unique_lock<std::mutex> lk(src->mu_);
    src->cv_.wait(lk, [&] { return src->queue_.size() < src->max_queued_ || src->closed_; });
    if (src->closed_) {
        lk.unlock();
        s.mirror.reset();  // (waits for its copy: the device block goes back to the pool)
        give(s.buf, s.cap);
        s.buf = nullptr;
        return false;
    }
    src->queue_.push_back(s);
    s.buf = nullptr;
    s.mirror.reset();
    src->cv_.notify_all();
    return true;
}

DecodedSource::DecodedSource(int device, hipStream_t consumer_stream, std::unique_ptr<SegmentProducer> producer, uint64_t reserve, size_t max_queued,
                             MemMeter *meter, bool mirror_from_start)
    : device_(device), stream_(consumer_stream), producer_(std::move(producer)), reserve_((reserve + 15) & ~15ull),
      max_queued_(max_queued ? max_queued : 1), meter_(meter) {
    mirror_wanted_.store(mirror_from_start, std::memory_order_release);  // (before the producer's thread exists)
    thread_ = std::thread([this] {
        (void)hipSetDevice(device_);
        pin_to_device_node(device_);
        MeterScope meter_scope(meter_);
        SegmentSink sink{this};
        std::string e;
        int rc = EXG_E_HIP;
        try {
            rc = producer_->run(sink, &e);
        } catch (const std::exception &ex) {  // (std::bad_alloc from a host vector: never let it leave the thread)
            e = std::string("decoder thread: ") + ex.what();
            rc = EXG_E_NOMEM;
        }
        std::lock_guard<std::mutex> g(mu_);
        done_ = true;
        rc_ = rc;
        err_ = e;
        cv_.notify_all();
    });
}

DecodedSource::~DecodedSource() {
    {
        std::lock_guard<std::mutex> g(mu_);
        closed_ = true;
        cv_.notify_all();
    }
    if (thread_.joinable()) thread_.join();
    MeterScope meter_scope(meter_);
    DeviceGuard guard(device_);
    producer_.reset();  // (its scratch goes back before the segments: it may still hold streams with work on them)
    (void)hipStreamSynchronize(stream_);  // a scan may still be reading the current segment
    if (have_cur_) free_segment(cur_);
    for (Segment &s : queue_) free_segment(s);
    queue_.clear();
    if (d2h_stream_) stream_pool()->give_d2h(device_, d2h_stream_);
}

bool DecodedSource::host_view(const uint8_t *d_at, const uint8_t **h_at, uint64_t *valid_from, std::shared_ptr<HostMirror> *keep) const {
    if (!have_cur_ || !cur_.mirror || cur_.mirror->hi != cur_.hi) return false;
    const int64_t off = d_at - (const uint8_t *)cur_.buf;
    if (off < 0 || (size_t)off >= cur_.mirror->blk->cap) return false;
    *h_at = (const uint8_t *)cur_.mirror->blk->p + off;
    *valid_from = cur_.mirror->from;
    *keep = cur_.mirror;
    return true;
}

void DecodedSource::free_segment(Segment &s) {
    if (!s.buf) return;
    s.mirror.reset();  // (the last reference here: a batch that points into the block keeps the block itself, not the mirror)
    dev_pool()->give(device_, s.buf, s.cap);
    s.buf = nullptr;
}

int DecodedSource::pop(Segment *out, std::string *err) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !queue_.empty() || done_; });
    if (!queue_.empty()) {
        *out = queue_.front();
        queue_.pop_front();
        cv_.notify_all();
        return EXG_OK;
    }
    out->buf = nullptr;
    if (rc_) {
        *err = err_;
        return rc_;
    }
    return EXG_OK;
}

bool DecodedSource::peek_mark(int id, uint64_t *pos) {
    std::lock_guard<std::mutex> g(mu_);
    if (mark_set_[id]) *pos = mark_[id];
    return mark_set_[id];
}

int DecodedSource::finish(std::string *err) {
    std::unique_lock<std::mutex> lk(mu_);
    // (a producer that is blocked on a full queue cannot end: only wait for one whose last segment has been taken)
    cv_.wait(lk, [&] { return done_ || !queue_.empty(); });
    if (done_ && rc_) {
        *err = err_;
        return rc_;
    }
    return EXG_OK;
}

int DecodedSource::acquire(uint64_t pos, uint64_t want, const uint8_t **d_pos, uint64_t *avail, bool *eof, std::string *err) {
    for (;;) {
        if (!have_cur_) {
            Segment s;
            const int rc = pop(&s, err);
            if (rc) return rc;
            if (!s.buf) {  // (a producer ends with a segment marked `last`; an empty stream of segments is an internal error)
                *err = "decoded stream ended without its last segment";
                return EXG_E_INVALID_ARG;
            }
            cur_ = s;
            have_cur_ = true;
        }
        if (pos < cur_.lo) {
            *err = "decoded stream: bytes at " + std::to_string(pos) + " are not resident (" + std::to_string(cur_.lo) + " .. " + std::to_string(cur_.hi) + ")";
            return EXG_E_INVALID_ARG;
        }
        if (cur_.last && pos > cur_.hi) {  // the stream ends in front of pos (a shard whose members hold only the file's header)
            *d_pos = cur_.at(cur_.hi);
            *avail = 0;
            *eof = true;
            return EXG_OK;
        }
        if (!cur_.last && pos > cur_.hi) {
            // the consumer begins behind this segment (a shard whose halo starts further in): nothing of it is needed
            (void)hipStreamSynchronize(stream_);
            free_segment(cur_);
            have_cur_ = false;
            n_consumed_++;
            continue;
        }
        if (cur_.last || pos + want <= cur_.hi) break;
        Segment nx;
        const int rc = pop(&nx, err);
        if (rc) {
            // the producer failed (a checksum behind the last block, a corrupt member further on): what was decoded in front of
            // the error is handed out first — fewer bytes than asked for, not the stream's end —, the error at the next call
            if (!error_deferred_ && cur_.hi > pos) {
                error_deferred_ = true;
                break;
            }
            return rc;
        }
        if (!nx.buf) {
            *err = "decoded stream ended without its last segment";
            return EXG_E_INVALID_ARG;
        }
        if (nx.start != cur_.hi || nx.lo > nx.start) {
            free_segment(nx);
            *err = "decoded stream: segments are not consecutive";
            return EXG_E_INVALID_ARG;
        }
        // the bytes the next batch needs from this segment — from the 16-byte boundary below pos — move in front of the next
        // one, unless that one brings them along (a window a decoder kept in front of its output)
        const uint64_t p0 = std::max<uint64_t>(pos & ~15ull, (cur_.lo & ~15ull));
        hipError_t he = hipSuccess;
        if (p0 < nx.lo) {
            const uint64_t carry = nx.lo - p0;
            if (carry <= nx.room_in_front()) {
                he = hipMemcpyAsync((void *)nx.at(p0), cur_.at(p0), carry, hipMemcpyDeviceToDevice, stream_);
                nx.lo = p0;
            } else if (nx.hi - cur_.hi <= cur_.room_behind()) {
                // ... and that block of their own has room behind its bytes (it was made with slack, below): the next
                // segment's bytes are appended, the tail stays where it is — a record of n segments costs n appends, not n
                // copies of everything in front (a multi-GB FASTA record)
                const uint64_t body = nx.hi - cur_.hi;  // (nx.start == cur_.hi; what nx keeps in front of it is here already)
                uint8_t *dst = const_cast<uint8_t *>(cur_.at(cur_.hi));
                if (body) he = hipMemcpyAsync(dst, nx.at(cur_.hi), body, hipMemcpyDeviceToDevice, stream_);
                if (he == hipSuccess) he = hipMemsetAsync(dst + body, 0, 64, stream_);
                if (he == hipSuccess) he = hipStreamSynchronize(stream_);
                cur_.hi = nx.hi;
                cur_.last = nx.last;
                cur_.mirror.reset();  // (a block of the consumer's own: it copies what it needs itself)
                free_segment(nx);
                n_consumed_++;
                if (he != hipSuccess) {
                    *err = std::string("carrying a record across decoded segments failed: ") + hipGetErrorString(he);
                    return EXG_E_HIP;
                }
                continue;
            } else {
                // a tail longer than the room a segment leaves in front of itself (one giant record, or a batch that was
                // widened): both move into a block of their own.  When that happens AGAIN for a tail that begins where the last
                // one began — a record that goes on growing — the block is made with as much room behind its bytes again
                // (geometric growth: the segments that follow are appended, above); a tail that moves on (a shard's halo while
                // its decoder works its way to the shard's first byte) gets a block of its size, and so does every tail under EXG_DEVICE_MEM_CAP_MB
                const uint64_t body = nx.hi - nx.lo;
                const bool growing = private_p0_ == p0 && !(meter_ && meter_->cap);  // (under a memory cap: no slack, by design)
                private_p0_ = p0;
                const size_t ncap = (size_t)(reserve_ + (growing ? 2 : 1) * (carry + body) + 64);
                void *nb = dev_pool()->take(device_, ncap);
                if (!nb) {
                    free_segment(nx);
                    *err = "out of device memory (" + std::to_string(ncap >> 20) + " MiB) for a record that spans decoded segments";
                    return EXG_E_HIP;
                }
                const int64_t norg = (int64_t)p0 - (int64_t)reserve_;
                uint8_t *base = (uint8_t *)nb + reserve_;
                he = hipMemcpyAsync(base, cur_.at(p0), carry, hipMemcpyDeviceToDevice, stream_);
                if (he == hipSuccess && body) he = hipMemcpyAsync(base + carry, nx.at(nx.lo), body, hipMemcpyDeviceToDevice, stream_);
                if (he == hipSuccess) he = hipMemsetAsync(base + carry + body, 0, 64, stream_);
                if (he == hipSuccess) he = hipStreamSynchronize(stream_);
                free_segment(nx);
                nx.mirror.reset();
                nx.buf = nb;
                nx.cap = ncap;
                nx.org = norg;
                nx.lo = p0;
            }
        }
        // the old block goes back to the (process-wide) pool: the copy out of it, and any scan still reading it, must be done
        if (he == hipSuccess) he = hipStreamSynchronize(stream_);
        free_segment(cur_);
        cur_ = nx;
        n_consumed_++;
        if (he != hipSuccess) {
            *err = std::string("carrying a record across decoded segments failed: ") + hipGetErrorString(he);
            return EXG_E_HIP;
        }
    }
    *d_pos = cur_.at(pos);
    *avail = cur_.hi - pos;
    *eof = cur_.last;
    return EXG_OK;
}

}  // namespace exg_rd
