// exg_rd_source.hpp — a compressed input as a BOUNDED stream of decoded bytes in HBM.
//
// The reference streams any size through a BufReader + DataFusion's convert_stream (rust/src/arrow_reader.rs:60-91,
// 116-153): its memory does not grow with the file.  Neither does this: a producer thread (gzip / BGZF:
// exg_rd_gzip.cpp, zstd: exg_rd_zstd.cpp) uploads a window of compressed bytes, decodes it into a SEGMENT — a pooled
// device block holding a run of decoded bytes — and hands the segments to the scanning thread through a short queue;
// the scan consumes a segment, carries the bytes of the record it cut (the tail behind the last complete record) into
// the free space in front of the next segment, and gives the old block back to the pool.  At any time a handful of
// segments exist, whatever the file's size; EXG_DEVICE_MEM_CAP_MB shrinks them (tests: inputs of >= 8x the cap).
#pragma once
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "exg_rd_internal.hpp"

namespace exg_rd {

// The decoded bytes of a segment on the HOST, on their way or there (round 5).  A consumer that hands out string columns needs
// the decoded bytes in host memory (its string_t point into them): copied behind the scan of every batch, that copy — 268 MB
// per 256 MiB segment, 5-6 ms of the link — was a step of the consumer's own, in series with its scan, its column copies and
// the caller's walk over the chunks.  Now the copy is issued when the producer hands the segment over, on a stream of the
// source's own: it runs while the segment waits in the queue and while the consumer is busy with the one in front, and the
// D2H link is kept busy back to back.  Layout: host byte x of the stream lives at blk->p + (x - org), the device block's own
// addressing; valid for [from, hi) once `ev` has completed (bytes in front of `from` — the tail the consumer carried over
// from the segment before — are copied by the consumer).
struct HostMirror {
    std::shared_ptr<PinnedBlock> blk;
    hipEvent_t ev = nullptr;
    uint64_t from = 0, hi = 0;
    // a mirror made in pieces (SegmentSink::mirror_now: the zstd checksum stage hashes the bytes as they arrive instead of
    // bringing them back a second time): piece k = stream bytes [from + k * piece_bytes, ...) is there when piece_ev[k] has completed
    std::vector<hipEvent_t> piece_ev;
    uint64_t piece_bytes = 0;
    const uint8_t *host_of(uint64_t x, int64_t org) const { return (const uint8_t *)blk->p + ((int64_t)x - org); }
    ~HostMirror() {
        if (ev) {
            (void)hipEventSynchronize(ev);  // (the block goes back to the pinned pool: no copy may still write it)
            (void)hipEventDestroy(ev);
        }
        for (hipEvent_t e : piece_ev)
            if (e) (void)hipEventDestroy(e);
    }
};

// Decoded stream bytes [lo, hi) in a pooled device block: byte x lives at buf + (x - org).  org is a multiple of 16 (and may
// be negative), so an address is congruent to its stream offset mod 16 — a batch that begins at stream offset p is entered
// at the 16-byte boundary below it with lead = p & 15, like a batch in an upload slot.  [lo, start) are bytes of earlier
// segments the producer kept in front (a zstd frame's window); the room in front of lo takes the consumer's carry.
struct Segment {
    void *buf = nullptr;
    size_t cap = 0;  // what the block was taken from the pool with
    int64_t org = 0;
    uint64_t lo = 0, start = 0, hi = 0;
    bool last = false;  // the stream ends at hi
    std::shared_ptr<HostMirror> mirror;  // the same bytes on their way to the host (SegmentSink::push, when the consumer wants them)
    const uint8_t *at(uint64_t x) const { return (const uint8_t *)buf + ((int64_t)x - org); }
    uint64_t room_in_front() const { return (uint64_t)((int64_t)lo - org); }
    uint64_t room_behind() const {  // bytes of the block behind hi, less the 64 zero bytes that follow the last byte
        const uint64_t used = (uint64_t)((int64_t)hi - org) + 64;
        return cap > used ? cap - used : 0;
    }
};

class DecodedSource;

// what a producer sees of its source
struct SegmentSink {
    DecodedSource *src;
    // hands a finished segment over (blocks while the queue is full); false: the consumer is gone — stop producing
    bool push(Segment &&s);
    bool cancelled() const;
    // a pooled device block (nullptr: out of device memory) / back to the pool
    void *take(size_t bytes);
    void give(void *p, size_t bytes);
    // a shard's decoder tells where, in decoded bytes, a boundary of its compressed input lies: mark 0 = the first byte of
    // the shard's own members / frames (what is in front is the halo), mark 1 = the first byte behind them (a FASTA shard
    // reads on to the next record start).  Set BEFORE the segment that begins there is pushed.
    void set_mark(int id, uint64_t pos);
    // the consumer hands out string columns: segments travel to the host (DecodedSource::want_host_mirror)
    bool mirror_wanted() const;
    // starts the segment's host mirror NOW, in pieces of piece_bytes with an event each — for a stage in front of push() that
    // wants the bytes on the host itself (the zstd checksum); push() then finds the mirror made.  false: no mirror (not wanted,
    // or no pinned memory / stream / event: the caller brings its bytes back itself)
    bool mirror_now(Segment &s, uint64_t piece_bytes);
};

struct SegmentProducer {
    virtual ~SegmentProducer() {}
    // Runs on the source's own thread with the device current: decode the stream from its beginning to its end, pushing
    // segments in order; the last one pushed has `last` set (it may be empty).  EXG_OK, or an error code + *err.
    virtual int run(SegmentSink &sink, std::string *err) = 0;
};

class DecodedSource {
public:
    DecodedSource(int device, hipStream_t consumer_stream, std::unique_ptr<SegmentProducer> producer, uint64_t reserve, size_t max_queued,
                  MemMeter *meter, bool mirror_from_start = false);
    ~DecodedSource();
    DecodedSource(const DecodedSource &) = delete;
    DecodedSource &operator=(const DecodedSource &) = delete;

    // Stream bytes from `pos` on, at least `want` of them unless the stream ends first, contiguous in HBM: *d_pos = the
    // device address of byte pos (congruent to pos mod 16; the 16-byte block it lies in is readable from its beginning, and
    // 64 bytes behind the last byte are zero), *avail = bytes from pos to the end of what is resident, *eof = the stream
    // ends there (*avail = 0 with *eof when it ends in front of pos).  Bytes in front of `pos` may be dropped: positions
    // never decrease from call to call.
    // EXG_OK or the producer's error (reported when the consumer reaches it: what was decoded before it is handed out).
    int acquire(uint64_t pos, uint64_t want, const uint8_t **d_pos, uint64_t *avail, bool *eof, std::string *err);
    // waits for the producer to end and returns its result (a checksum that is verified behind the last segment)
    int finish(std::string *err);
    // a mark of the producer, if it has been set (never a blocking wait: the producer cannot run further ahead of the
    // consumer than its queue allows) — while mark 1 is not set, every segment handed out so far ends at or in front of it
    bool peek_mark(int id, uint64_t *pos);
    uint64_t reserve() const { return reserve_; }
    uint64_t segments_consumed() const { return n_consumed_; }
    // The consumer hands out string columns of this stream: from now on the producer's segments travel to the host as soon as
    // they are handed over (HostMirror).  Segments pushed before the call have none: the consumer copies those itself.
    void want_host_mirror() { mirror_wanted_.store(true, std::memory_order_release); }
    // the host copy of the segment the last acquire() answered from, if it has one: *h_at = the host address of device address
    // d_at (same offset in the block), *valid_from = the stream offset from which the mirror holds (or will hold, once
    // (*keep)->ev has completed) the bytes; what the consumer needs in front of it, it copies itself
    bool host_view(const uint8_t *d_at, const uint8_t **h_at, uint64_t *valid_from, std::shared_ptr<HostMirror> *keep) const;

private:
    friend struct SegmentSink;
    int pop(Segment *out, std::string *err);  // next segment in order (waits); EXG_OK + out->buf == nullptr: no more
    void free_segment(Segment &s);

    int device_;
    hipStream_t stream_;
    std::unique_ptr<SegmentProducer> producer_;
    uint64_t reserve_;
    size_t max_queued_;
    MemMeter *meter_;
    std::thread thread_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Segment> queue_;
    bool done_ = false, closed_ = false;
    int rc_ = 0;
    std::string err_;
    uint64_t mark_[2] = {0, 0};
    bool mark_set_[2] = {false, false};
    Segment cur_;
    bool have_cur_ = false;
    uint64_t private_p0_ = ~0ull;  // where the tail began that last moved into a block of its own (acquire)
    bool error_deferred_ = false;  // the producer's error was met while bytes in front of it were still to be handed out
    uint64_t n_consumed_ = 0;
    std::atomic<bool> mirror_wanted_{false};
    hipStream_t d2h_stream_ = nullptr;  // the mirrors' copies (made by the first push / mirror_now that wants one)
    std::mutex mirror_mu_;
};

// exg_rd_gzip.cpp: file bytes [c_begin, c_end) of fd are gzip members (BGZF or not, any mixture); `target` = decoded bytes per
// segment.  bgzf_only: a member without the BGZF size field is an error (shards of a BGZF file).
// reserve: bytes of room every segment leaves in front of its first byte (DecodedSource's `reserve`)
// mark_at[i] (a member's offset, ~0: none): sink.set_mark(i, decoded offset of that member's first byte)
std::unique_ptr<SegmentProducer> make_gzip_producer(exg_reader *r, int fd, uint64_t c_begin, uint64_t c_end, uint64_t target, const std::string &path,
                                                    bool bgzf_only, uint64_t reserve, const uint64_t mark_at[2] = nullptr);
// exg_rd_zstd.cpp: the zstd frames of file bytes [0, n) of fd whose first byte lies in [c_begin, c_end) (the whole file: 0, n);
// mark_at[i]: a frame's offset (~0: none)
std::unique_ptr<SegmentProducer> make_zstd_producer(exg_reader *r, int fd, uint64_t n, uint64_t c_begin, uint64_t c_end, uint64_t target,
                                                    const std::string &path, uint64_t reserve, const uint64_t mark_at[2] = nullptr);

int plan_zstd_shard(exg_reader *r, int fd, uint64_t n, const std::string &path, uint64_t halo_want, uint64_t header_bytes, uint64_t *c_begin,
                    uint64_t *c_end, uint64_t *own_lo, uint64_t *own_hi, bool *bytes_follow);

}  // namespace exg_rd
