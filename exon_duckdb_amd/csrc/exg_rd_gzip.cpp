// exg_rd_gzip.cpp — reader level, gzip inputs: compressed bytes -> HBM -> exg_inflate_* -> inflated bytes in HBM for the
// scans, every member's CRC-32 / ISIZE verified.  Replaces DataFusion 28 `FileCompressionType::GZIP.convert_stream` ->
// async-compression -> flate2 (and noodles-bgzf for BGZF) behind rust/src/arrow_reader.rs:60-91.
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>

#include "exg_rd_internal.hpp"

namespace exg_rd {

// gzip + VCF: the header is parsed on the host, so the leading '#' lines of the inflated bytes come back
// (blk->p then holds a prefix of the file, blk->n stays the inflated size; the DataChunk payload travels per batch)
int gz_host_header(exg_reader *r, PinnedBlock &b, const void *d_file) {
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
    const uint64_t back = kShardHalo + (kShardHalo >> 1);
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
    while (h0 > 0 && halo_bytes < kShardHalo) halo_bytes += mem[--h0].out_cap;
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

}  // namespace exg_rd
