// exg_rd_fanout.hpp — ONE consumer-facing stream, N devices.
//
// The reference's FFI has no shard argument (`new_reader`, exon/include/rust.hpp:41-46) and its glue pins the scan to one
// thread (module.cpp:36, DuckDB's ArrowScanGlobalState): a consumer that pulls one stream — path (A) of INTEGRATION.md, or
// any MaxThreads() == 1 plan — would use one GPU.  With shard_count = 0 a reader therefore fans out by itself: the input is
// cut into STRIPES (byte-range shards of ~1 GiB, exg_open's own shard mechanism: a record belongs to the stripe its last
// line ends in), stripe s is read by a worker thread on device s mod N through a reader of its own, and the consumer takes
// the stripes' device batches in stripe order — i.e. in file order.  A WORKER holds at most `depth` batches the consumer has
// not taken yet, counted over all of its stripes (bounded per stripe, a worker whose stripes are shorter than `depth`
// batches would run through all of them while the consumer is still on the first: memory would grow with the file), so all N
// devices work on consecutive stripes while memory stays bounded; nothing is exchanged between them.
// COUNT(*) sums the stripes in any order.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "exg_common.hpp"

namespace exg_rd {

struct Stripe {
    std::string path;
    uint32_t shard_index = 0, shard_count = 1;
    int device = 0;
};
struct FanItem {
    std::shared_ptr<void> batch;  // a device batch's host buffers (exg_rd::Batch, or the Arrow stream's batch); null: no more
    uint64_t rows = 0;
};
struct FanSub {  // one stripe being read
    virtual ~FanSub() {}
    virtual int next(FanItem *out, std::string *err) = 0;  // out->batch == null: the stripe has ended
    virtual int count(uint64_t *rows, std::string *err) = 0;
    // device bytes held now / at most, device batches, decoded segments of this stripe's reader (0s when unknown)
    virtual void stats(uint64_t *now, uint64_t *peak, uint64_t *batches, uint64_t *segments) { *now = *peak = *batches = *segments = 0; }
};
using FanOpen = std::function<int(const Stripe &, std::unique_ptr<FanSub> *, std::string *)>;

class FanOut {
public:
    FanOut(std::vector<Stripe> stripes, unsigned n_workers, FanOpen open, size_t depth);
    ~FanOut();
    FanOut(const FanOut &) = delete;
    FanOut &operator=(const FanOut &) = delete;
    // the next batch in stripe order (out->batch == null at the end); an error is returned when the consumer reaches it
    int next(FanItem *out, std::string *err);
    // rows of all stripes (instead of next(): the stripes are counted, no batch is built)
    int count(uint64_t *rows, std::string *err);
    size_t n_stripes() const { return stripes_.size(); }
    // the most batches any worker has held at once (queued, not yet taken by the consumer): <= depth by construction
    size_t max_outstanding();
    // what the stripes' readers held / did, summed (a sub reports when it ends): exg_reader_stats_of of a fan-out reader
    struct Stats {
        uint64_t device_bytes_now = 0, device_bytes_peak = 0, device_batches = 0, decoded_segments = 0;
    };
    Stats stats();

private:
    struct Slot {
        std::deque<FanItem> q;
        bool done = false;
        int rc = 0;
        std::string err;
        uint64_t rows = 0;
    };
    void start(bool counting);
    void work(unsigned w);
    std::vector<Stripe> stripes_;
    unsigned n_workers_;
    FanOpen open_;
    size_t depth_;
    std::vector<Slot> slots_;
    std::vector<size_t> outstanding_;   // per worker: batches pushed and not yet taken
    std::vector<FanSub *> live_;        // per worker: the stripe reader it is on (stats)
    size_t max_outstanding_ = 0;
    Stats ended_;                       // sums over the stripes that have ended; peak: the largest sum seen of the live readers' peaks
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool started_ = false, counting_ = false, closed_ = false;
    size_t cur_ = 0;
};

}  // namespace exg_rd
