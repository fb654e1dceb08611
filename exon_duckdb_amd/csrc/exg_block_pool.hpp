// exg_block_pool.hpp — the process-wide pool of pinned host blocks (bounce buffers of the uploads, column vectors, Arrow
// buffers), keyed by the NUMA node the block's pages live on.
//
// Pinned blocks are expensive to create (hipHostMalloc is ~1 ms per 10 MiB), so they are recycled: a batch returns its blocks
// when the consumer releases its last chunk.  A block's pages are placed when it is made — near the device that is current
// on the allocating thread — and stay there; with readers on all eight devices of a node in one process (the fan-out of
// exg_rd_fanout.hpp, or a table function whose scan threads claim shards) a block first used by a reader on socket 0 must
// not be handed to a reader on socket 1: its DMA and its pread copies would cross the socket link for as long as the
// block lives.  So the free list is searched only among the blocks of the taker's node, and the pool's cap follows the
// number of devices IN USE — the distinct devices that were current on a thread that took a block, counted as they appear
// (a reader holds two upload slots and the column blocks of the batches in flight: ~1.5 GiB); a process that reads on one
// device of an 8-GPU node keeps 6 GiB, not 48.  A block whose node is unknown (sysfs says -1, or no hook) counts as node 0.
//
// No HIP in this header: allocation, release and "which node is this thread's device on" are hooks, so that the host-only
// sanitizer driver (tests/host_asan_driver.cpp) runs the bookkeeping over malloc / free.
#pragma once
#include <stddef.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

namespace exg_rd {

struct BlockPool {
    struct Hooks {
        void *(*alloc)(size_t) = nullptr;   // pinned allocation near the calling thread's current device; NULL on failure
        void (*release)(void *) = nullptr;
        int (*current_node)() = nullptr;    // NUMA node of the calling thread's current device (>= 0; 0 when unknown)
        int (*current_device)() = nullptr;  // the calling thread's current device (>= 0; 0 when unknown)
    };
    static constexpr size_t kPerDevice = 6ull << 30;  // pooled bytes kept per device that has taken a block (all nodes together)

    explicit BlockPool(Hooks h, size_t cap_override = 0) : hooks(h), cap_override_(cap_override) {}
    BlockPool(const BlockPool &) = delete;
    BlockPool &operator=(const BlockPool &) = delete;

    // sizes are multiples of 32 MiB; a free block is reused for a request it fits without wasting more than
    // half of it (batches of one scan are alike, so in the steady state the same few blocks go round)
    static size_t size_class(size_t n) { return (std::max<size_t>(n, 1) + (32u << 20) - 1) & ~(size_t)((32u << 20) - 1); }

    char *take(size_t *sz) {
        *sz = size_class(*sz);
        const int node = hooks.current_node ? std::max(0, hooks.current_node()) : 0;
        const int dev = hooks.current_device ? std::min(63, std::max(0, hooks.current_device())) : 0;
        {
            std::lock_guard<std::mutex> g(mu);
            devices_seen_ |= 1ull << dev;
            size_t best = free_blocks.size();
            for (size_t i = 0; i < free_blocks.size(); i++) {
                const Free &f = free_blocks[i];
                if (f.node == node && f.sz >= *sz && f.sz <= 2 * *sz + (64u << 20) && (best == free_blocks.size() || f.sz < free_blocks[best].sz)) best = i;
            }
            if (best != free_blocks.size()) {
                char *p = free_blocks[best].p;
                *sz = free_blocks[best].sz;
                pooled_bytes -= *sz;
                free_blocks.erase(free_blocks.begin() + (long)best);
                n_reused++;
                return p;
            }
        }
        char *p = (char *)hooks.alloc(*sz);
        if (!p) {
            // the pool may be what holds the memory: give the free blocks of every node back and try once more
            trim();
            p = (char *)hooks.alloc(*sz);
            if (!p) return nullptr;
        }
        std::lock_guard<std::mutex> g(mu);
        node_of[p] = node;
        n_created++;
        return p;
    }
    // A block that does not fit under the cap takes the place of the blocks that have lain free the LONGEST (round 5: the pool
    // kept whatever came first — a process that had read other shapes of input before kept their blocks and paid a
    // hipHostMalloc for every new size again and again: +11 ms per zstd frame in bench.py's full run).
    void give(char *p, size_t sz) {
        std::vector<char *> gone;
        bool keep = false;
        {
            std::lock_guard<std::mutex> g(mu);
            const auto it = node_of.find(p);
            const int node = it == node_of.end() ? 0 : it->second;
            if (sz <= cap()) {
                while (pooled_bytes + sz > cap() && !free_blocks.empty()) {  // (free_blocks: oldest first)
                    const Free f = free_blocks.front();
                    free_blocks.erase(free_blocks.begin());
                    pooled_bytes -= f.sz;
                    node_of.erase(f.p);
                    gone.push_back(f.p);
                }
                free_blocks.push_back(Free{p, sz, node});
                pooled_bytes += sz;
                keep = true;
            } else if (it != node_of.end()) {
                node_of.erase(it);
            }
        }
        for (char *q : gone) hooks.release(q);
        if (!keep) hooks.release(p);
    }
    void trim() {
        std::vector<Free> gone;
        {
            std::lock_guard<std::mutex> g(mu);
            gone.swap(free_blocks);
            for (const Free &f : gone) node_of.erase(f.p);
            pooled_bytes = 0;
        }
        for (const Free &f : gone) hooks.release(f.p);
    }
    size_t cap() const {  // (mu held, or a racy read for reporting)
        if (cap_override_) return cap_override_;
        return kPerDevice * (size_t)std::max(1, __builtin_popcountll(devices_seen_));
    }
    int devices_in_use() {
        std::lock_guard<std::mutex> g(mu);
        return __builtin_popcountll(devices_seen_);
    }
    // (reporting / tests)
    size_t pooled() {
        std::lock_guard<std::mutex> g(mu);
        return pooled_bytes;
    }
    size_t free_on_node(int node) {
        std::lock_guard<std::mutex> g(mu);
        size_t n = 0;
        for (const Free &f : free_blocks) n += f.node == node;
        return n;
    }
    ~BlockPool() {
        for (const Free &f : free_blocks) hooks.release(f.p);
    }

    Hooks hooks;
    size_t n_created = 0, n_reused = 0;

private:
    struct Free {
        char *p;
        size_t sz;
        int node;
    };
    std::mutex mu;
    std::vector<Free> free_blocks;
    std::unordered_map<char *, int> node_of;  // every live block made by this pool -> the node of its pages
    size_t pooled_bytes = 0;
    size_t cap_override_ = 0;
    unsigned long long devices_seen_ = 0;  // bit d: a thread with device d current has taken a block
};

}  // namespace exg_rd
