// exg_rd_zstd.cpp — reader level, zstd inputs (.zst, compression='zstd'): compressed bytes -> HBM -> exg_zstd.hip.
// Replaces DataFusion 28 `FileCompressionType::ZSTD.convert_stream` -> async-compression -> zstd 0.12.3 behind
// rust/src/arrow_reader.rs:73, :87-88.
#include <string.h>

#include <thread>

#include "exg_rd_internal.hpp"
#include "exg_zstd.hpp"

namespace exg_rd {

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

}  // namespace exg_rd
