// exg_zstd_index.cpp — host walk over the frames and blocks of a Zstandard stream (RFC 8878 3.1).  Nothing is decoded
// here: the walk reads the frame header, the 3-byte block headers and, inside a compressed block, the few bytes that say
// how large its literals are, how many sequences it holds and which tables it defines or repeats — so that the device can
// decode all blocks at once (exg_zstd.hip).  The error texts are libzstd's (what the reference's zstd 0.12.3 would raise).
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "exg_zstd.hpp"

namespace exg {
namespace zst {

static inline uint32_t rd24(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
static inline uint32_t rd32(const uint8_t *p) { return rd24(p) | ((uint32_t)p[3] << 24); }

static bool fail(Index &idx, const char *what, uint64_t at) {
    idx.error = std::string(what) + " (zstd, byte " + std::to_string(at) + ")";
    return false;
}

// S.get(off, len): the stream's bytes [off, off + len) (the caller has checked that they exist), valid until the next get.
// S.failed(): a get could not bring its bytes (a file that shrank behind its fstat, an I/O error): the walk STOPS there — what a
// failed get returns are zeros, and a zero block header is a valid empty raw block that is not the last one: a walk that went on
// would push one Block per three bytes of the remaining file, a failing pread each (advisor, round 4: hundreds of millions of
// syscalls and tens of GB of host memory on a multi-GB file).  Nothing read behind the failure enters the index.
// stop_est != 0: a PREFIX of the index (build_index_prefix_fd) — the walk ends behind the block with which the blocks' estimated
// output (a compressed block: kBlockMax, the count a reader's round plan makes) reaches stop_est, inside a frame: that frame
// enters the index with the blocks seen so far (its header's fields are real, n_blocks is not, its checksum is 0) and
// *stopped is set.  A walk that meets the end of the stream first is the whole index.
template <class S>
static bool build_index_walk(S &src, uint64_t n, Index &idx, uint64_t stop_est = 0, bool *stopped = nullptr) {
    uint64_t est = 0;
    static const char *kSrcSize = "Src size is incorrect", *kCorrupt = "Data corruption detected", *kIo = "short read while walking the zstd block headers";
    uint64_t pos = 0;
    while (pos < n) {
        if (n - pos < 4) return fail(idx, kSrcSize, pos);
        const uint32_t magic = rd32(src.get(pos, 4));
        if (src.failed()) return fail(idx, kIo, pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame (3.1.2)
            if (n - pos < 8) return fail(idx, kSrcSize, pos);
            const uint64_t sz = rd32(src.get(pos + 4, 4));
            if (src.failed()) return fail(idx, kIo, pos);
            if (n - pos - 8 < sz) return fail(idx, kSrcSize, pos);
            pos += 8 + sz;
            continue;
        }
        if (magic != 0xFD2FB528u) return fail(idx, "Unknown frame descriptor", pos);
        const uint64_t frame_at = pos;
        pos += 4;
        if (pos >= n) return fail(idx, kSrcSize, frame_at);
        const uint8_t fhd = *src.get(pos++, 1);
        const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
        if (fhd & 8) return fail(idx, "Unsupported frame parameter", frame_at);  // reserved bit
        Frame fr;
        memset(&fr, 0, sizeof fr);
        fr.content_size = ~0ull;
        fr.src_off = frame_at;
        fr.has_checksum = (fhd >> 2) & 1;
        if (!single) {
            if (pos >= n) return fail(idx, kSrcSize, frame_at);
            const uint8_t wd = *src.get(pos++, 1);
            const int wlog = 10 + (wd >> 3);
            if (wlog > 31) return fail(idx, "Frame requires too much memory for decoding", frame_at);
            fr.window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wd & 7);
        }
        static const int did_bytes[4] = {0, 1, 2, 4};
        if (n - pos < (uint64_t)did_bytes[did_flag]) return fail(idx, kSrcSize, frame_at);
        uint32_t did = 0;
        if (did_bytes[did_flag]) {
            const uint8_t *dp = src.get(pos, (uint32_t)did_bytes[did_flag]);
            for (int i = 0; i < did_bytes[did_flag]; i++) did |= (uint32_t)dp[i] << (8 * i);
        }
        pos += did_bytes[did_flag];
        if (did) return fail(idx, "Dictionary mismatch", frame_at);  // the reference passes no dictionary
        const int fcs_bytes = fcs_flag == 0 ? single : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
        if (n - pos < (uint64_t)fcs_bytes) return fail(idx, kSrcSize, frame_at);
        if (fcs_bytes) {
            uint64_t v = 0;
            const uint8_t *fp = src.get(pos, (uint32_t)fcs_bytes);
            for (int i = 0; i < fcs_bytes; i++) v |= (uint64_t)fp[i] << (8 * i);
            if (fcs_bytes == 2) v += 256;
            fr.content_size = v;
            pos += fcs_bytes;
        }
        if (src.failed()) return fail(idx, kIo, frame_at);
        if (single) fr.window = fr.content_size;
        if (fr.window > kWindowMax) return fail(idx, "Frame requires too much memory for decoding", frame_at);
        fr.first_block = (uint32_t)idx.blocks.size();
        idx.open_frame = fr;
        idx.open_valid = true;  // (until the frame is complete: salvage_index)
        const uint32_t frame_id = (uint32_t)idx.frames.size();
        uint32_t huf_src = kNone, tbl_src[3] = {kNone, kNone, kNone};
        for (;;) {
            if (n - pos < 3) return fail(idx, kSrcSize, pos);
            const uint32_t bh = rd24(src.get(pos, 3));
            if (src.failed()) return fail(idx, kIo, pos);
            pos += 3;
            const int last = bh & 1, type = (bh >> 1) & 3;
            const uint32_t bsize = bh >> 3;
            if (type == 3 || bsize > kBlockMax) return fail(idx, kCorrupt, pos - 3);
            if (idx.blocks.size() >= 0xFFFFFFF0u) return fail(idx, "too many blocks", pos);
            Block b;
            memset(&b, 0, sizeof b);
            b.src_off = pos;
            b.src_size = bsize;
            b.type = (uint8_t)type;
            b.first_of_frame = idx.blocks.size() == fr.first_block;
            b.frame = frame_id;
            b.huf_src = kNone;
            b.tbl_src[0] = b.tbl_src[1] = b.tbl_src[2] = kNone;
            const uint32_t self = (uint32_t)idx.blocks.size();
            if (type == 1) {
                if (n - pos < 1) return fail(idx, kSrcSize, pos);
                b.out_size = bsize;
                idx.known_out += bsize;
                pos += 1;
            } else if (type == 0) {
                if (n - pos < bsize) return fail(idx, kSrcSize, pos);
                b.out_size = bsize;
                idx.known_out += bsize;
                pos += bsize;
            } else {
                if (n - pos < bsize) return fail(idx, kSrcSize, pos);
                if (bsize < 3) return fail(idx, kCorrupt, pos);  // libzstd: MIN_CBLOCK_SIZE
                uint8_t p[5] = {0, 0, 0, 0, 0};  // the literals section header: at most 5 bytes
                memcpy(p, src.get(pos, bsize < 5 ? bsize : 5u), bsize < 5 ? bsize : 5u);
                const int ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
                uint32_t regen, csize = 0, hsz, streams = 1;
                if (ltype < 2) {
                    if (!(sf & 1)) hsz = 1, regen = p[0] >> 3;
                    else if (sf == 1) hsz = 2, regen = (p[0] >> 4) | ((uint32_t)p[1] << 4);
                    else {
                        if (bsize < 3) return fail(idx, kCorrupt, pos);
                        hsz = 3, regen = (p[0] >> 4) | ((uint32_t)p[1] << 4) | ((uint32_t)p[2] << 12);
                    }
                } else {
                    if (bsize < 5) return fail(idx, kCorrupt, pos);  // libzstd: a compressed literals section needs 5 bytes of input
                    uint64_t v = 0;
                    for (int i = 0; i < 5; i++) v |= (uint64_t)p[i] << (8 * i);
                    if (sf == 0) hsz = 3, regen = (v >> 4) & 1023, csize = (v >> 14) & 1023;
                    else if (sf == 1) hsz = 3, streams = 4, regen = (v >> 4) & 1023, csize = (v >> 14) & 1023;
                    else if (sf == 2) hsz = 4, streams = 4, regen = (v >> 4) & 16383, csize = (uint32_t)(v >> 18) & 16383;
                    else hsz = 5, streams = 4, regen = (v >> 4) & 262143, csize = (uint32_t)(v >> 22) & 262143;
                }
                if (regen > kBlockMax) return fail(idx, kCorrupt, pos);
                const uint64_t lit_end = (uint64_t)hsz + (ltype == 0 ? regen : ltype == 1 ? 1 : csize);
                if (lit_end + 1 > bsize) return fail(idx, kCorrupt, pos);  // the sequence count follows
                if (ltype == 2) huf_src = self;
                if (ltype == 3 && huf_src == kNone) return fail(idx, "Dictionary mismatch", pos);  // libzstd: dictionary_corrupted
                b.lit_type = (uint8_t)ltype;
                b.lit_streams = (uint8_t)streams;
                b.lit_hdr = hsz;
                b.lit_regen = regen;
                b.lit_csize = csize;
                b.huf_src = ltype >= 2 ? huf_src : kNone;
                // the sequences section header: the count (1 - 3 bytes) and the modes byte; q / end: offsets in the block
                const uint32_t sh_have = (uint32_t)std::min<uint64_t>(4, bsize - lit_end);
                uint8_t sh[4] = {0, 0, 0, 0};
                memcpy(sh, src.get(pos + lit_end, sh_have), sh_have);
                uint64_t q = lit_end;
                const uint64_t end = bsize;
                uint32_t nseq = sh[q++ - lit_end];
                if (nseq >= 128) {
                    if (nseq == 255) {
                        if (q + 2 > end) return fail(idx, kCorrupt, pos);
                        nseq = sh[q - lit_end] + ((uint32_t)sh[q + 1 - lit_end] << 8) + 0x7F00;
                        q += 2;
                    } else {
                        if (q + 1 > end) return fail(idx, kCorrupt, pos);
                        nseq = ((nseq - 128) << 8) + sh[q - lit_end];
                        q += 1;
                    }
                }
                b.nseq = nseq;
                if (nseq == 0) {
                    if (q != end) return fail(idx, kCorrupt, pos);
                } else {
                    if (q >= end) return fail(idx, kCorrupt, pos);
                    const int modes = sh[q - lit_end];
                    if (modes & 3) return fail(idx, kCorrupt, pos);
                    b.seq_hdr = (uint32_t)q;
                    const int m[3] = {modes >> 6, (modes >> 4) & 3, (modes >> 2) & 3};
                    for (int t = 0; t < 3; t++) {
                        if (m[t] != 3) tbl_src[t] = self;
                        else if (tbl_src[t] == kNone) return fail(idx, kCorrupt, pos);  // Repeat_Mode with nothing to repeat
                        b.tbl_src[t] = tbl_src[t];
                    }
                }
                b.lit_off = idx.lit_bytes;
                b.seq_off = idx.n_seq;
                idx.lit_bytes += (regen + 15) & ~15u;
                idx.n_seq += nseq;
                pos += bsize;
            }
            if (src.failed()) return fail(idx, kIo, b.src_off);  // (its literals / sequences header did not come: not a block of the index)
            idx.blocks.push_back(b);
            if (last) break;
            est += type == 2 ? kBlockMax : bsize;
            if (stop_est && est >= stop_est) {
                fr.n_blocks = (uint32_t)idx.blocks.size() - fr.first_block;
                idx.frames.push_back(fr);
                idx.open_valid = false;
                if (stopped) *stopped = true;
                return true;
            }
        }
        fr.n_blocks = (uint32_t)idx.blocks.size() - fr.first_block;
        if (fr.has_checksum) {
            if (n - pos < 4) return fail(idx, kSrcSize, pos);
            fr.checksum = rd32(src.get(pos, 4));
            if (src.failed()) return fail(idx, kIo, pos);
            pos += 4;
        }
        idx.frames.push_back(fr);
        idx.open_valid = false;
    }
    return true;
}

// (a get that failed hands out zeros until the walk's next failed() check: whatever the zeros made the walk say, the error is the read's)
template <class S>
static bool build_index_from(S &src, uint64_t n, Index &idx) {
    const bool good = build_index_walk(src, n, idx);
    if (src.failed()) {
        idx.error = "short read while walking the zstd block headers";
        return false;
    }
    return good;
}

namespace {
struct MemSrc {
    const uint8_t *data;
    const uint8_t *get(uint64_t off, uint32_t) const { return data + off; }
    bool failed() const { return false; }
};
// the same walk over a file: two small reads per block (its header with the literals header behind it, its sequences
// header) and nothing mapped — a 2 GB mapping costs 20 ms of page faults on eight threads to walk and 49 ms to unmap
struct FdSrc {
    int fd;
    uint64_t n;
    uint8_t buf[64];
    uint64_t at = ~0ull;
    uint32_t have = 0;
    bool io_error = false;
    bool failed() const { return io_error; }
    const uint8_t *get(uint64_t off, uint32_t len) {
        if (at != ~0ull && off >= at && off + len <= at + have) return buf + (off - at);
        const uint32_t want = (uint32_t)std::min<uint64_t>(sizeof buf, n - off);
        uint32_t got = 0;
        while (got < want) {
            const ssize_t k = pread(fd, buf + got, want - got, (off_t)(off + got));
            if (k <= 0) {
                if (k < 0 && errno == EINTR) continue;
                break;
            }
            got += (uint32_t)k;
        }
        if (got < len) {  // (the file shrank, or an I/O error: the walk asks failed() behind every get and stops; zeros until then)
            io_error = true;
            memset(buf + got, 0, sizeof buf - got);
            got = len;
        }
        at = off;
        have = got;
        return buf;
    }
};
}  // namespace

bool build_index(const uint8_t *data, uint64_t n, Index &idx) {
    MemSrc src{data};
    return build_index_from(src, n, idx);
}

namespace {
// pread of exactly `len` bytes (false: fewer came)
bool pread_all(int fd, void *dst, size_t len, uint64_t off) {
    size_t got = 0;
    while (got < len) {
        const ssize_t k = pread(fd, (char *)dst + got, len - got, (off_t)(off + got));
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            return false;
        }
        got += (size_t)k;
    }
    return true;
}

// The walk over a big file is a chain of ~0.5 us preads, two per block: 30 ms per 2 GB in front of the first round.  Only the
// hops from block header to block header are a chain, though: a first pass makes them (one pread per block, which also
// brings the literals header), eight threads then fetch every block's sequences header, and the walk proper —
// build_index_from, the one implementation of what is accepted and which error is raised where — runs over what was fetched.
// The two passes in front only fetch: whatever they did not bring (a frame header, a file they could not follow) the walk
// reads itself, like FdSrc.
struct Snip {
    uint64_t off;
    uint8_t b[16];
};
struct PrefetchedSrc {
    FdSrc fallback;
    const std::vector<Snip> *hdr, *seq;  // ascending offsets
    size_t ih = 0, is = 0;
    bool failed() const { return fallback.io_error; }
    const uint8_t *get(uint64_t off, uint32_t len) {
        while (ih < hdr->size() && (*hdr)[ih].off + 16 <= off) ih++;
        if (ih < hdr->size() && (*hdr)[ih].off <= off && off + len <= (*hdr)[ih].off + 16) return (*hdr)[ih].b + (off - (*hdr)[ih].off);
        while (is < seq->size() && (*seq)[is].off + 16 <= off) is++;
        if (is < seq->size() && (*seq)[is].off <= off && off + len <= (*seq)[is].off + 16) return (*seq)[is].b + (off - (*seq)[is].off);
        return fallback.get(off, len);
    }
};
}  // namespace

bool build_index_fd(int fd, uint64_t n, Index &idx) {
    std::vector<Snip> hdr, seq;
    static const uint64_t prefetch_min = getenv("EXG_ZSTD_INDEX_PREFETCH_MIN") ? strtoull(getenv("EXG_ZSTD_INDEX_PREFETCH_MIN"), nullptr, 10) : (64ull << 20);  // (tests: 0)
    if (n >= prefetch_min) {
        // ---- pass 1: from block header to block header (frame headers are read and skipped; anything unexpected ends the pass)
        std::vector<uint64_t> seq_at;  // where each compressed block's sequences header lies (0: none to fetch)
        uint64_t pos = 0;
        bool follow = true;
        while (follow && pos + 4 <= n) {
            uint8_t fh[24] = {0};
            const uint32_t fh_have = (uint32_t)std::min<uint64_t>(sizeof fh, n - pos);
            if (!pread_all(fd, fh, fh_have, pos)) break;
            const uint32_t magic = rd32(fh);
            if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
                if (fh_have < 8) break;
                pos += 8 + (uint64_t)rd32(fh + 4);
                continue;
            }
            if (magic != 0xFD2FB528u || fh_have < 5) break;
            const uint8_t fhd = fh[4];
            const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
            static const int did_bytes[4] = {0, 1, 2, 4};
            const int fcs_bytes = fcs_flag == 0 ? single : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
            pos += 5 + (single ? 0 : 1) + did_bytes[did_flag] + fcs_bytes;
            for (;;) {  // the frame's blocks
                if (pos + 3 > n) { follow = false; break; }
                Snip h;
                h.off = pos;
                memset(h.b, 0, sizeof h.b);
                const uint32_t have = (uint32_t)std::min<uint64_t>(16, n - pos);
                if (!pread_all(fd, h.b, have, pos)) { follow = false; break; }
                if (have == 16) hdr.push_back(h);
                const uint32_t bh = rd24(h.b);
                const int last = bh & 1, type = (bh >> 1) & 3;
                const uint32_t bsize = bh >> 3;
                if (type == 3 || bsize > kBlockMax) { follow = false; break; }
                const uint64_t body = pos + 3;
                uint64_t sat = 0;
                if (type == 2 && bsize >= 5 && have >= 8) {  // where the sequences header lies: behind the literals section
                    const uint8_t *p = h.b + 3;
                    const int ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
                    uint64_t lit_end;
                    if (ltype < 2) {
                        const uint32_t hsz = !(sf & 1) ? 1 : sf == 1 ? 2 : 3;
                        const uint32_t regen = !(sf & 1) ? p[0] >> 3 : sf == 1 ? (p[0] >> 4) | ((uint32_t)p[1] << 4) : (p[0] >> 4) | ((uint32_t)p[1] << 4) | ((uint32_t)p[2] << 12);
                        lit_end = (uint64_t)hsz + (ltype == 0 ? regen : 1);
                    } else {
                        uint64_t v = 0;
                        for (int i = 0; i < 5; i++) v |= (uint64_t)p[i] << (8 * i);
                        const uint32_t hsz = sf < 2 ? 3 : sf == 2 ? 4 : 5;
                        const uint32_t csize = sf < 2 ? (uint32_t)(v >> 14) & 1023 : sf == 2 ? (uint32_t)(v >> 18) & 16383 : (uint32_t)(v >> 22) & 262143;
                        lit_end = (uint64_t)hsz + csize;
                    }
                    if (lit_end + 1 <= bsize && body + lit_end + 16 <= n) sat = body + lit_end;
                }
                seq_at.push_back(sat);
                pos = body + (type == 1 ? 1 : bsize);
                if (last) break;
            }
            if (follow && pos <= n) {
                if (fhd & 4) pos += 4;  // Content_Checksum
            }
        }
        // ---- pass 2: the sequences headers, eight threads
        size_t n_seq = 0;
        for (uint64_t a : seq_at) n_seq += a != 0;
        seq.resize(n_seq);
        {
            size_t k = 0;
            for (uint64_t a : seq_at)
                if (a) seq[k++].off = a;
        }
        const unsigned nt = 8;
        std::vector<std::thread> th;
        std::atomic<bool> ok{true};
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                for (size_t k = t; k < seq.size(); k += nt)
                    if (!pread_all(fd, seq[k].b, 16, seq[k].off)) ok.store(false);
            });
        for (auto &t : th) t.join();
        if (!ok.load()) seq.clear();  // (the walk reads them itself and meets the error where it lies)
    }
    PrefetchedSrc src;
    src.fallback.fd = fd;
    src.fallback.n = n;
    src.hdr = &hdr;
    src.seq = &seq;
    // (a failed read ends the walk where it happened — build_index_from — with the blocks in front of it in the index: the caller
    // salvages them, the rows in front of the damage come first)
    return build_index_from(src, n, idx);
}

// The first blocks of a stream, for a reader that begins its first round while a helper walks the whole file (round 5: the walk
// of a 4 GB frame's 30 000 blocks is 23 ms in front of everything else): two small reads per block, until the blocks' estimated
// output reaches stop_est.  true + *stopped: idx is a prefix (its last frame is open: see build_index_walk); true + !*stopped:
// the stream ended first — idx is the whole index; false: an error in the prefix (the caller lets the whole walk report it).
bool build_index_prefix_fd(int fd, uint64_t n, uint64_t stop_est, Index &idx, bool *stopped) {
    *stopped = false;
    FdSrc src;
    src.fd = fd;
    src.n = n;
    const bool good = build_index_walk(src, n, idx, stop_est ? stop_est : 1, stopped);
    return good && !src.failed();
}

bool salvage_index(Index &idx) {
    const size_t whole = idx.frames.empty() ? 0 : (size_t)idx.frames.back().first_block + idx.frames.back().n_blocks;
    if (idx.open_valid && idx.blocks.size() > idx.open_frame.first_block) {
        Frame f = idx.open_frame;
        f.n_blocks = (uint32_t)(idx.blocks.size() - f.first_block);
        f.has_checksum = 0;        // (never reached)
        f.content_size = ~0ull;    // (cannot be met)
        idx.frames.push_back(f);
    } else {
        idx.blocks.resize(whole);
    }
    idx.open_valid = false;
    return !idx.frames.empty();
}

}  // namespace zst
}  // namespace exg
