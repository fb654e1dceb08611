// exg_rd_zstd.cpp — reader level, zstd inputs (.zst, compression='zstd') as a stream of decoded segments
// (exg_rd_source.hpp): compressed bytes -> HBM -> exg_zstd.hip.  Replaces DataFusion 28
// `FileCompressionType::ZSTD.convert_stream` -> async-compression -> zstd 0.12.3 behind rust/src/arrow_reader.rs:73, :87-88.
#include <string.h>
#include <sys/mman.h>

#include <thread>

#include "exg_rd_source.hpp"
#include "exg_zstd.hpp"

namespace exg_rd {

namespace {

class ZstdProducer : public SegmentProducer {
public:
    ZstdProducer(exg_reader *r, int fd, uint64_t n, uint64_t target, const std::string &path, uint64_t reserve)
        : device_(r->device), fd_(fd), n_(n), target_(target), path_(path), reserve_((reserve + 15) & ~15ull) {}
    int run(SegmentSink &sink, std::string *err) override;

private:
    int device_, fd_;
    uint64_t n_, target_;
    std::string path_;
    uint64_t reserve_;
};

int ZstdProducer::run(SegmentSink &sink, std::string *err) {
    // the host's walk over the frame / block headers reads the mapped file, beside the upload
    void *map = n_ ? mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0) : nullptr;
    if (map == MAP_FAILED) {
        *err = "cannot map '" + path_ + "'";
        return EXG_E_IO;
    }
    struct Unmap {
        void *p;
        size_t n;
        ~Unmap() { if (p) munmap(p, n); }
    } unmap{map, (size_t)n_};
    hipStream_t st = nullptr;
    if (stream_pool()->take(device_, &st) != hipSuccess) {
        *err = "cannot create a stream for the zstd decoder";
        return EXG_E_HIP;
    }
    struct StreamBack {
        int dev;
        hipStream_t s;
        ~StreamBack() { stream_pool()->give(dev, s); }
    } stream_back{device_, st};
    PoolBuf comp(device_, st);
    if (!comp.take(n_ + 64)) {
        *err = "out of device memory for the compressed file";
        return EXG_E_HIP;
    }
    exg::zst::Index idx;
    bool idx_ok = false;
    std::thread idx_thread([&] { idx_ok = exg::zst::build_index((const uint8_t *)map, n_, idx); });
    std::string up_err;
    int up_rc = n_ ? upload_fd(device_, fd_, comp.p, n_, 0, st, nullptr, &up_err) : EXG_OK;
    idx_thread.join();
    if (up_rc) {
        *err = up_err;
        return up_rc;
    }
    if (!idx_ok) {
        *err = idx.error + " in '" + path_ + "'";
        return EXG_E_PARSE;
    }
    if (hipMemsetAsync((char *)comp.p + n_, 0, 64, st) != hipSuccess) {
        *err = "hipMemsetAsync failed";
        return EXG_E_HIP;
    }
    void *d_out = nullptr;
    uint64_t produced = 0;
    std::vector<exg::zst::PendingCheck> pending;
    int rc = exg::zst::decode((const uint8_t *)map, comp.p, n_, &d_out, &produced, st, &pending, &idx, reserve_);
    if (rc) {
        *err = std::string(exg_last_error_message()) + " in '" + path_ + "'";
        return rc;
    }
    comp.release();
    Segment seg;
    seg.buf = d_out;
    seg.cap = (size_t)(reserve_ + produced + 64);
    seg.org = -(int64_t)reserve_;
    seg.lo = seg.start = 0;
    seg.hi = produced;
    seg.last = true;
    if (!sink.push(std::move(seg))) return EXG_OK;
    // frames too large for the device's serial XXH64 are hashed here, from a copy that travels back while the scan runs; the
    // reader looks at the result when the file's last batch has been handed out (a streaming decoder reports a checksum
    // mismatch at the end of the frame too) — the segment stays alive until then (DecodedSource::finish)
    if (!pending.empty()) {
        std::string verr;
        const int vrc = exg::zst::host_verify((const char *)d_out + reserve_, pending, device_, &verr);
        if (vrc) {
            *err = verr + " in '" + path_ + "'";
            return vrc;
        }
    }
    return EXG_OK;
}

}  // namespace

std::unique_ptr<SegmentProducer> make_zstd_producer(exg_reader *r, int fd, uint64_t n, uint64_t target, const std::string &path, uint64_t reserve) {
    return std::unique_ptr<SegmentProducer>(new ZstdProducer(r, fd, n, target, path, reserve));
}

}  // namespace exg_rd
