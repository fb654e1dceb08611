// exg_rd_fanout.cpp — see exg_rd_fanout.hpp: stripes of one input read by worker threads on N devices, handed to one
// consumer in file order.  (No HIP call here: the workers' readers make them.)
#include "exg_rd_fanout.hpp"

namespace exg_rd {

FanOut::FanOut(std::vector<Stripe> stripes, unsigned n_workers, FanOpen open, size_t depth)
    : stripes_(std::move(stripes)), n_workers_(n_workers ? n_workers : 1), open_(std::move(open)), depth_(depth ? depth : 1), slots_(stripes_.size()) {
    if (n_workers_ > stripes_.size()) n_workers_ = (unsigned)std::max<size_t>(1, stripes_.size());
    outstanding_.assign(n_workers_, 0);
    live_.assign(n_workers_, nullptr);
}

FanOut::~FanOut() {
    {
        std::lock_guard<std::mutex> g(mu_);
        closed_ = true;
        cv_.notify_all();
    }
    for (auto &t : threads_) t.join();
}

void FanOut::start(bool counting) {
    if (started_) return;
    started_ = true;
    counting_ = counting;
    for (unsigned w = 0; w < n_workers_; w++) threads_.emplace_back([this, w] { work(w); });
}

// worker w reads stripes w, w + W, w + 2W, ... one after the other
void FanOut::work(unsigned w) {
    for (size_t s = w; s < stripes_.size(); s += n_workers_) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (closed_) return;
        }
        Slot &slot = slots_[s];
        std::unique_ptr<FanSub> sub;
        std::string err;
        int rc = open_(stripes_[s], &sub, &err);
        {
            std::lock_guard<std::mutex> g(mu_);
            live_[w] = rc ? nullptr : sub.get();
        }
        while (!rc) {
            if (counting_) {
                uint64_t n = 0;
                rc = sub->count(&n, &err);
                std::lock_guard<std::mutex> g(mu_);
                slot.rows = n;
                break;
            }
            FanItem item;
            rc = sub->next(&item, &err);
            if (rc || !item.batch) break;
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return outstanding_[w] < depth_ || closed_; });
            if (closed_) {
                live_[w] = nullptr;
                return;  // (sub closes its reader on this thread)
            }
            slot.q.push_back(std::move(item));
            max_outstanding_ = std::max(max_outstanding_, ++outstanding_[w]);
            cv_.notify_all();
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            if (sub) {
                uint64_t now = 0, peak = 0, nb = 0, ns = 0;
                sub->stats(&now, &peak, &nb, &ns);
                ended_.device_batches += nb, ended_.decoded_segments += ns;
                ended_.device_bytes_peak = std::max(ended_.device_bytes_peak, peak);
            }
            live_[w] = nullptr;
        }
        sub.reset();
        std::lock_guard<std::mutex> g(mu_);
        slot.done = true;
        slot.rc = rc;
        slot.err = err;
        cv_.notify_all();
        if (rc) return;  // the stripes behind a failing one are never reached by the consumer
    }
}

int FanOut::next(FanItem *out, std::string *err) {
    out->batch.reset();
    out->rows = 0;
    std::unique_lock<std::mutex> lk(mu_);
    start(false);
    while (cur_ < slots_.size()) {
        Slot &slot = slots_[cur_];
        cv_.wait(lk, [&] { return !slot.q.empty() || slot.done; });
        if (!slot.q.empty()) {
            *out = std::move(slot.q.front());
            slot.q.pop_front();
            outstanding_[cur_ % n_workers_]--;
            cv_.notify_all();
            return EXG_OK;
        }
        if (slot.rc) {
            *err = slot.err;
            return slot.rc;
        }
        cur_++;
    }
    return EXG_OK;
}

size_t FanOut::max_outstanding() {
    std::lock_guard<std::mutex> g(mu_);
    return max_outstanding_;
}

// The stripes' readers charge meters of their own (one per reader, on their worker's thread): the front reader's meter sees
// none of it.  now = what the live stripe readers hold; peak = the largest such sum seen by a call, or any one ended
// stripe's peak, whichever is larger (a lower bound of the true peak, exact when the readers' peaks coincide).
FanOut::Stats FanOut::stats() {
    std::lock_guard<std::mutex> g(mu_);
    Stats s = ended_;
    uint64_t live_peak = 0;
    for (FanSub *sub : live_) {
        if (!sub) continue;
        uint64_t now = 0, peak = 0, nb = 0, ns = 0;
        sub->stats(&now, &peak, &nb, &ns);
        s.device_bytes_now += now, live_peak += peak, s.device_batches += nb, s.decoded_segments += ns;
    }
    ended_.device_bytes_peak = std::max(ended_.device_bytes_peak, live_peak);
    s.device_bytes_peak = ended_.device_bytes_peak;
    return s;
}

int FanOut::count(uint64_t *rows, std::string *err) {
    *rows = 0;
    std::unique_lock<std::mutex> lk(mu_);
    start(true);
    for (size_t s = 0; s < slots_.size(); s++) {
        Slot &slot = slots_[s];
        cv_.wait(lk, [&] { return slot.done; });
        if (slot.rc) {
            *err = slot.err;
            return slot.rc;
        }
        *rows += slot.rows;
    }
    return EXG_OK;
}

}  // namespace exg_rd
