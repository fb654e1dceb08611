// exg_rd_bgzf.cpp — host only (no HIP call): the walk over BGZF members (RFC 1952 headers with the 'BC' subfield bgzip /
// htslib write), which tells where every member's DEFLATE stream lies and how many bytes it inflates to without decoding
// anything.  Replaces the block framing of noodles-bgzf 0.22.0 (rust/Cargo.lock:2055-2056) behind
// rust/src/arrow_reader.rs:60-91.  Everything read here is a user's file: the TU is built with ASan / UBSan in
// tests/host_asan_driver.cpp.
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <thread>

#include "exg_rd_internal.hpp"

namespace exg_rd {

const uint8_t *Peek::at(uint64_t off, size_t len) {
    if (off + len > n) return nullptr;
    if (fd < 0) return map + off;
    if (off >= b0 && off + len <= b1) return buf + (off - b0);
    const size_t want = (size_t)std::min<uint64_t>(sizeof buf, n - off);
    size_t got = 0;
    while (got < want) {
        const ssize_t k = pread(fd, buf + got, want - got, (off_t)(off + got));
        if (k <= 0) break;
        got += (size_t)k;
    }
    if (got < len) return nullptr;
    b0 = off, b1 = off + got;
    return buf;
}

// One BGZF member at `pos` (RFC 1952 header with FEXTRA and a 'BC' subfield, as bgzip / htslib write it):
// fills m (out_off = 0), *crc = the trailer's CRC-32, and returns the offset of the next member, or 0 when this is not such
// a header.
uint64_t bgzf_member_at(Peek &f, uint64_t pos, exg_inflate_member *m, uint32_t *crc) {
    const uint8_t *h = f.at(pos, 18);
    if (!h || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return 0;
    const uint64_t xlen = h[10] | ((uint64_t)h[11] << 8);
    if (xlen > 240 || !(h = f.at(pos, 12 + (size_t)xlen))) return 0;  // (bgzip writes 6; anything long is not BGZF to this walk)
    int64_t bsize = -1;
    for (uint64_t q = 12; q + 4 <= 12 + xlen;) {
        const uint64_t slen = h[q + 2] | ((uint64_t)h[q + 3] << 8);
        if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) bsize = h[q + 4] | ((int64_t)h[q + 5] << 8);
        q += 4 + slen;
    }
    if (bsize < 0) return 0;
    const uint64_t end = pos + (uint64_t)bsize + 1, p = pos + 12 + xlen;
    if (end > f.n || end < p + 8) return 0;
    // the trailer: the window read for it also holds the next member's header
    const uint8_t *t = f.at(end - 8, 8);
    if (!t) return 0;
    m->comp_off = p;
    m->comp_size = end - p;
    m->out_off = 0;
    m->out_cap = t[4] | ((uint64_t)t[5] << 8) | ((uint64_t)t[6] << 16) | ((uint64_t)t[7] << 24);
    if (crc) *crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    return m->out_cap <= 65536 ? end : 0;
}
// first member that starts at or after `from`: a header whose chain holds for four more members (or runs into the
// end of the file) — the signature alone also occurs inside compressed data
uint64_t bgzf_find(const uint8_t *d, int fd, uint64_t n, uint64_t from) {
    Peek f(d, fd, n);
    uint8_t chunk[4096];
    for (uint64_t base = from; base + 18 <= n;) {
        // candidates: 0x1f bytes of the next 4 KiB
        const size_t len = (size_t)std::min<uint64_t>(sizeof chunk, n - base);
        const uint8_t *c = d + base;
        if (fd >= 0) {
            size_t got = 0;
            while (got < len) {
                const ssize_t k = pread(fd, chunk + got, len - got, (off_t)(base + got));
                if (k <= 0) return n;
                got += (size_t)k;
            }
            c = chunk;
        }
        for (size_t i = 0; i < len;) {
            const void *hit = memchr(c + i, 0x1f, len - i);
            if (!hit) break;
            const uint64_t pos = base + (uint64_t)((const uint8_t *)hit - c);
            if (pos + 18 > n) return n;
            exg_inflate_member m;
            uint64_t q = pos;
            int hops = 0;
            while (hops < 5 && q < n) {
                const uint64_t nx = bgzf_member_at(f, q, &m);
                if (!nx) break;
                q = nx;
                hops++;
            }
            if (hops == 5 || (hops > 0 && q == n)) return pos;
            i = (size_t)(pos - base) + 1;
        }
        base += len;
    }
    return n;
}

// Member index of a pure BGZF file by several host threads (the serial pointer chase through the page cache costs
// 110-150 ms per 10 GB: ~190 ns of cache misses per member): every thread finds a header near its cut, then walks to
// the next thread's start.  false: not (only) BGZF, or a walk did not land on its neighbour's start — the caller falls
// back to the serial RFC 1952 index.
// upto < n: only the members whose header begins in front of the first member that starts at or behind `upto` (the head of
// the file, for a decode that starts before the whole index is there).
bool bgzf_parallel_index(const uint8_t *d, int fd, uint64_t n, exg_inflate_member *members, uint64_t cap, uint64_t *k_out, uint64_t *total_out,
                         std::vector<uint32_t> *crc_out, uint64_t upto) {
    exg_inflate_member probe;
    {
        Peek f(d, fd, n);
        if (!n || !bgzf_member_at(f, 0, &probe)) return false;
    }
    const uint64_t limit = upto >= n ? n : bgzf_find(d, fd, n, upto);
    // (a pread per member, ~5 us each here: the walk scales with its threads until the cores run out)
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min(32u, hw), limit >> 24));  // 16 MiB per thread at least
    std::vector<uint64_t> starts(T + 1, limit);
    starts[0] = 0;
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++)
            th.emplace_back([&, t] { starts[t] = std::min(limit, bgzf_find(d, fd, n, (uint64_t)((unsigned __int128)limit * t / T))); });
        for (auto &x : th) x.join();
    }
    for (unsigned t = 1; t <= T; t++)
        if (starts[t] < starts[t - 1]) return false;
    std::vector<std::vector<exg_inflate_member>> parts(T);
    std::vector<std::vector<uint32_t>> crcs(T);
    std::vector<int> ok(T, 0);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t] {
                Peek f(d, fd, n);
                uint64_t pos = starts[t];
                auto &v = parts[t];
                auto &cv = crcs[t];
                v.reserve((size_t)((starts[t + 1] - starts[t]) / 8192 + 64));
                cv.reserve(v.capacity());
                while (pos < starts[t + 1]) {
                    exg_inflate_member m;
                    uint32_t crc = 0;
                    const uint64_t nx = bgzf_member_at(f, pos, &m, &crc);
                    if (!nx) return;
                    v.push_back(m);
                    cv.push_back(crc);
                    pos = nx;
                }
                ok[t] = pos == starts[t + 1];
            });
        for (auto &x : th) x.join();
    }
    uint64_t k = 0, out = 0;
    if (crc_out) crc_out->clear();
    for (unsigned t = 0; t < T; t++) {
        if (!ok[t] || k + parts[t].size() > cap) return false;
        for (auto &m : parts[t]) {
            m.out_off = out;
            out += m.out_cap;
            members[k++] = m;
        }
        if (crc_out) crc_out->insert(crc_out->end(), crcs[t].begin(), crcs[t].end());
    }
    *k_out = k;
    *total_out = out;
    return true;
}

}  // namespace exg_rd
