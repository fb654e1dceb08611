// exg_rd_batch.cpp — reader level: a file is opened (mapped, or handed to a decoder), then scanned one device batch at a
// time: bytes -> HBM (prefetched slot / decoded segment / synchronous upload) -> scan kernels -> the batch's host vectors.
// Replaces exon's BatchReader::read_batch loop behind the stream `new_reader` returns (rust/src/arrow_reader.rs:116-153)
// and the per-batch pull in WTArrowTableFunction::Scan (exon/src/exon/arrow_table_function/module.cpp:257-294).
//
// Data layout: device batches are record aligned (each starts on the first byte after the last complete record of the
// previous one); the kernels emit string_t whose pointers address host memory (the file's mapping, or the pinned block that
// receives a decoded batch), i.e. the DataChunk payload is zero-copy and only 64 B/record of string_t + validity cross
// PCIe on the way back.  Chunks are 2048-row slices of the batch's host vectors, reference counted until
// exg_release_chunk.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <functional>
#include <memory>
#include <thread>

#include "exg_fastq_ws.hpp"
#include "exg_filter.hpp"
#include "exg_map_guard.hpp"
#include "exg_rd_source.hpp"

namespace exg_rd {

// ---- compressed inputs: the file becomes a stream of decoded segments (exg_rd_source.hpp) --------------------------------

// room every segment leaves in front of its bytes for the tail the scan carries over (a longer tail moves into a block of its own)
static uint64_t source_reserve(const exg_reader *r) { return std::min<uint64_t>(std::max<uint64_t>(r->device_batch_bytes / 8, 64u << 10), 8u << 20); }

// The leading '#' lines of a decoded VCF come back to the host for the header parse (blk->p holds a prefix of the decoded
// stream; the DataChunk payload travels per batch): `src` is read from its first byte until a line that does not start with
// '#' begins inside the prefix, or the stream ends.
static int source_header_prefix(exg_reader *r, DecodedSource *src, PinnedBlock &b) {
    // (not more than a device batch at first: what is acquired here is the first batch's size, which sizes the scan's buffers)
    for (uint64_t want = std::min<uint64_t>(4u << 20, r->device_batch_bytes);; want *= 8) {
        const uint8_t *d_at = nullptr;
        uint64_t avail = 0;
        bool eof = false;
        std::string msg;
        int rc = src->acquire(0, want, &d_at, &avail, &eof, &msg);
        if (rc) return fail(r, rc, msg);
        const size_t len = (size_t)std::min<uint64_t>(want, avail);
        if (b.p) global_pool()->give((char *)b.p, b.cap), b.p = nullptr;
        size_t cap = len + 64;
        b.p = global_pool()->take(&cap);
        if (!b.p) return fail(r, EXG_E_HIP, "out of pinned host memory for the VCF header");
        b.cap = cap;
        b.pooled = true;
        if (len) RD_HIP(r, hipMemcpyAsync(b.p, d_at, len, hipMemcpyDeviceToHost, r->stream));
        RD_HIP(r, hipStreamSynchronize(r->stream));
        const char *d = (const char *)b.p;
        size_t pos = 0;
        while (pos < len && d[pos] == '#') {
            const void *nl = memchr(d + pos, '\n', len - pos);
            pos = nl ? (size_t)((const char *)nl - d) + 1 : len;
        }
        if (pos < len || (eof && len == avail)) {
            r->gz_header_prefix = len;
            return EXG_OK;
        }
    }
}

// The first FASTA record start ('>' at the beginning of a line) of a decoded stream at or behind `from` (~0: none before the
// stream ends).  Everything from `from` on stays resident: the record that begins there is the next batch.
static int source_find_record(exg_reader *r, uint64_t from, uint64_t *found) {
    *found = ~0ull;
    if (!r->d_phase && !(r->d_phase = dev_pool()->take(r->device, 4096))) return fail(r, EXG_E_HIP, "out of device memory");
    const uint64_t base = from ? from - 1 : 0;  // (the byte in front says whether `from` begins a line)
    for (uint64_t want = r->device_batch_bytes;; want *= 2) {
        const uint8_t *d_at = nullptr;
        uint64_t avail = 0;
        bool eof = false;
        std::string msg;
        int rc = r->src->acquire(base, want, &d_at, &avail, &eof, &msg);
        if (rc) return fail(r, rc, msg);
        if (base + avail > from) {
            unsigned long long pos = ~0ull;
            rc = exg_fasta_find_record(d_at, from - base, avail, base == 0 && r->data0_is_line_start, (uint64_t *)r->d_phase, r->stream);
            if (rc) return fail(r, rc, exg_last_error_message());
            RD_HIP(r, hipMemcpyAsync(&pos, r->d_phase, 8, hipMemcpyDeviceToHost, r->stream));
            RD_HIP(r, hipStreamSynchronize(r->stream));
            if (pos != ~0ull) {
                *found = base + pos;
                return EXG_OK;
            }
        }
        if (eof || avail < want) return EXG_OK;
    }
}

// Shard `shard_index` of `shard_count` of a BGZF file: a member belongs to the shard in whose 1/shard_count of the FILE's
// bytes its header begins.  The reader finds its members without indexing the file (a search for a header whose chain holds
// near each cut: the pointer chase over a whole file costs 110 ms per 10 GB and every rank would pay it) and streams
// [c_begin, c_end): ~1 MiB (inflated) of members in front of its own — the halo that holds the beginning of the record that
// ends behind the cut — then its own.  Positions of the reader are offsets in THAT stream.
static int plan_bgzf_shard(exg_reader *r, const std::string &path, uint64_t n, uint64_t *c_begin, uint64_t *c_end, uint64_t header_bytes) {
    const int fd = r->fd_keep->fd;
    Peek peek(nullptr, fd, n);
    exg_inflate_member probe;
    if (!bgzf_member_at(peek, 0, &probe))
        return fail(r, EXG_E_UNSUPPORTED, "shards of a gzip input need BGZF framing (every member carries its size): '" + path + "'");
    const uint64_t lo = (uint64_t)((unsigned __int128)n * r->shard_index / r->shard_count);
    const uint64_t hi = r->shard_index + 1 == r->shard_count ? n : (uint64_t)((unsigned __int128)n * (r->shard_index + 1) / r->shard_count);
    const uint64_t first_own = lo == 0 ? 0 : bgzf_find(nullptr, fd, n, lo);
    *c_end = hi >= n ? n : std::max<uint64_t>(first_own, bgzf_find(nullptr, fd, n, hi));
    // candidates for the halo: members that begin in the ~1.5 MiB of file in front of the cut (BGZF does not expand)
    std::vector<uint64_t> hdr, out;
    const uint64_t halo_want = r->halo_want;
    const uint64_t back = halo_want > n ? n : halo_want + (halo_want >> 1) + (128u << 10);
    for (uint64_t pos = first_own == 0 ? 0 : bgzf_find(nullptr, fd, n, lo > back ? lo - back : 0); pos < first_own;) {
        exg_inflate_member m;
        const uint64_t nx = bgzf_member_at(peek, pos, &m);
        if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(pos) + " of '" + path + "'");
        hdr.push_back(pos);
        out.push_back(m.out_cap);
        pos = nx;
    }
    size_t h0 = hdr.size();
    uint64_t halo_bytes = 0;
    while (h0 > 0 && halo_bytes < halo_want) halo_bytes += out[--h0];
    *c_begin = h0 < hdr.size() ? hdr[h0] : first_own;
    // VCF: where do the members that hold the header end?  A halo that would begin among them is taken from the start of
    // the file instead, so that the end of the header is a known offset of this reader's stream
    if (header_bytes && *c_begin != 0) {
        uint64_t q = 0, sum = 0;
        while (q < n && sum < header_bytes) {
            exg_inflate_member m;
            const uint64_t nx = bgzf_member_at(peek, q, &m);
            if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(q) + " of '" + path + "'");
            sum += m.out_cap;
            q = nx;
        }
        if (*c_begin < q) {  // q: compressed offset behind the header's members
            for (uint64_t pos = 0; pos < *c_begin;) {
                exg_inflate_member m;
                const uint64_t nx = bgzf_member_at(peek, pos, &m);
                if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(pos) + " of '" + path + "'");
                halo_bytes += m.out_cap;
                pos = nx;
            }
            *c_begin = 0;
        }
    }
    if (*c_begin > *c_end) *c_begin = *c_end;
    // does any inflated byte follow this reader's members?  (the empty BGZF end marker — or a later shard that owns nothing
    // else — must not keep the shard with the file's last record from seeing the end of the file)
    bool bytes_follow = false;
    for (uint64_t q = *c_end; q < n && !bytes_follow;) {
        exg_inflate_member m;
        const uint64_t nx = bgzf_member_at(peek, q, &m);
        if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(q) + " of '" + path + "'");
        bytes_follow = m.out_cap != 0;
        q = nx;
    }
    r->own_c_begin = first_own;
    r->range_preset = true;
    r->preset_pos = halo_bytes;  // inflated bytes of the members in front of its own
    r->range_eof = !bytes_follow;
    r->data0_is_line_start = *c_begin == 0;  // byte 0 of the stream begins a line only when the stream begins with the file
    return EXG_OK;
}

// How the bytes a decoded input's strings point into reach the host when chunks are handed out (never for COUNT(*) or the Arrow
// stream): kPayloadCompact — a projection that leaves payload-bearing columns out: the selected columns' out-of-line strings are
// closed up into a side buffer behind the scan; kPayloadMirror — the decoded segments themselves, sent ahead by the producer
// (HostMirror; a segment without one is copied behind its scan); kPayloadNone — no string column is selected.  FASTA (round 6):
// always the side buffer for id / description — its sequences are joined on the device and travel as that, so the decoded text is
// needed for the definition lines' strings alone, 2 % of it (the whole text went along until then: a bgzip FASTA into DataChunks
// sent every byte back twice, 92 ms for 1.96 GB).
enum PayloadRoute { kPayloadNone, kPayloadCompact, kPayloadMirror };
static PayloadRoute payload_route(const exg_reader *r) {
    if (r->format == EXG_FMT_FASTA) {
        static const bool fasta_whole = getenv("EXG_FASTA_WHOLE_TEXT") != nullptr;  // (A/B: the decoded text behind the scan)
        return (r->want_cols & 3ull) && !fasta_whole ? kPayloadCompact : kPayloadNone;
    }
    const uint64_t strs = r->format == EXG_FMT_VCF ? 0x1DDull : 0xFull, nested = r->format == EXG_FMT_VCF ? 0x1D4ull : 0ull;
    const uint64_t sel = r->want_cols & strs;
    if (!sel) return kPayloadNone;
    static const bool no_compact = getenv("EXG_NO_PAYLOAD_COMPACT") != nullptr;
    if (!no_compact && sel != strs && !(sel & nested)) return kPayloadCompact;
    static const bool no_mirror = getenv("EXG_NO_HOST_MIRROR") != nullptr;  // (A/B and tests: the copy behind the scan)
    return no_mirror ? kPayloadNone : kPayloadMirror;
}

static int open_source(exg_reader *r, std::shared_ptr<PinnedBlock> &blk, const std::string &path, uint64_t n) {
    const int fd = r->fd_keep->fd;
    const uint64_t reserve = source_reserve(r), target = r->device_batch_bytes;
    auto out_blk = std::make_shared<PinnedBlock>();  // (n = 0: the decoded size is not known; p: the VCF header prefix)
    r->gz_header_prefix = 0;
    const size_t queued = getenv("EXG_SOURCE_QUEUE") ? (size_t)std::max(1, atoi(getenv("EXG_SOURCE_QUEUE"))) : r->mem_cap ? 1 : 2;
    auto make = [&](uint64_t c_begin, uint64_t c_end, bool bgzf_only, const uint64_t *marks) {
        std::unique_ptr<SegmentProducer> prod = r->compression == kGzip ? make_gzip_producer(r, fd, c_begin, c_end, target, path, bgzf_only, reserve, marks)
                                                                        : make_zstd_producer(r, fd, n, c_begin, c_end, target, path, reserve, marks);
        // (EXG_OPEN_CHUNKS: the caller said it will pull chunks — the segments travel to the host from the first one on)
        const bool mirror0 = r->expect_chunks && !r->arrow_emit && payload_route(r) == kPayloadMirror;
        return std::unique_ptr<DecodedSource>(new DecodedSource(r->device, r->stream, std::move(prod), reserve, queued, &r->meter, mirror0));
    };
    if (r->compression == kGzip && n == 0) return fail(r, EXG_E_PARSE, "empty gzip file '" + path + "'");
    r->fa_shard = false;
    r->fa_end = ~0ull;
    if (r->shard_count > 1) {
        // VCF: every rank needs the header (schema, and where the data begins): read from the start of the file by a
        // source of its own, kept on the host like in the unsharded case
        uint64_t header_bytes = 0;
        if (r->format == EXG_FMT_VCF) {
            std::unique_ptr<DecodedSource> head = make(0, n, r->compression == kGzip, nullptr);
            int rc = source_header_prefix(r, head.get(), *out_blk);
            if (rc) return rc;
            const char *d = (const char *)out_blk->p;
            size_t pos = 0;
            while (pos < r->gz_header_prefix && d[pos] == '#') {
                const void *nl = memchr(d + pos, '\n', (size_t)r->gz_header_prefix - pos);
                pos = nl ? (size_t)((const char *)nl - d) + 1 : (size_t)r->gz_header_prefix;
            }
            header_bytes = pos;
        }
        const bool fasta = r->format == EXG_FMT_FASTA;
        uint64_t c_begin = 0, c_end = n, marks[2] = {~0ull, ~0ull};
        bool wait_own = false;  // where the shard's own bytes begin is told by the decoder (mark 0)
        if (r->compression == kGzip && !fasta) {
            int rc = plan_bgzf_shard(r, path, n, &c_begin, &c_end, header_bytes);
            if (rc) return rc;
        } else if (r->compression == kGzip) {
            // bgzip FASTA: a record belongs to the shard in whose members' bytes its '>' line begins.  The stream starts one
            // member in front of the shard's own (is their first byte a line start?) and runs on behind them until the next
            // record start is found
            Peek peek(nullptr, fd, n);
            exg_inflate_member probe;
            if (!bgzf_member_at(peek, 0, &probe))
                return fail(r, EXG_E_UNSUPPORTED, "shards of a gzip input need BGZF framing (every member carries its size): '" + path + "'");
            const uint64_t lo = (uint64_t)((unsigned __int128)n * r->shard_index / r->shard_count);
            const uint64_t hi = r->shard_index + 1 == r->shard_count ? n : (uint64_t)((unsigned __int128)n * (r->shard_index + 1) / r->shard_count);
            const uint64_t own_lo = lo == 0 ? 0 : bgzf_find(nullptr, fd, n, lo);
            const uint64_t own_hi = hi >= n ? n : std::max<uint64_t>(own_lo, bgzf_find(nullptr, fd, n, hi));
            c_begin = own_lo;
            for (uint64_t pos = own_lo == 0 ? 0 : bgzf_find(nullptr, fd, n, own_lo > (192u << 10) ? own_lo - (192u << 10) : 0); pos < own_lo;) {
                exg_inflate_member m;
                const uint64_t nx = bgzf_member_at(peek, pos, &m);
                if (!nx) return fail(r, EXG_E_PARSE, "not a BGZF member at byte " + std::to_string(pos) + " of '" + path + "'");
                c_begin = pos;  // (the last member in front of the shard's own)
                pos = nx;
            }
            marks[0] = own_lo;
            marks[1] = own_hi >= n ? ~0ull : own_hi;
            wait_own = true;
        } else {
            uint64_t own_lo = 0, own_hi = n;
            bool bytes_follow = false;
            int rc = plan_zstd_shard(r, fd, n, path, fasta ? 1 : r->halo_want, header_bytes, &c_begin, &c_end, &own_lo, &own_hi, &bytes_follow);
            if (rc) return rc;
            marks[0] = own_lo;
            r->own_c_begin = own_lo;
            if (fasta) {
                c_end = n;
                marks[1] = own_hi >= n ? ~0ull : own_hi;
            }
            r->range_eof = !bytes_follow;
            wait_own = true;
        }
        r->src = make(c_begin, c_end, true, marks);
        if (wait_own) {
            // The decoder tells where the shard's own bytes begin (mark 0) when it gets there; until then the halo's segments
            // are taken one by one — never a blocking wait for the mark: the decoder cannot run further ahead than its queue —
            // and only the newest `keep` bytes of them stay: a halo is made of whole frames / members, and with few large
            // frames in front of the shard everything from the stream's first byte would otherwise be resident (and copied
            // again for every segment more) before the shard's first row
            uint64_t own = 0;
            const uint64_t keep = std::max<uint64_t>(fasta ? 0 : r->halo_want, 256u << 10) + (64u << 10);
            for (uint64_t pos = 0, end = 0; !r->src->peek_mark(0, &own);) {
                if (end - pos > keep) pos = (end - keep) & ~15ull;  // (what lies in front of it is dropped by the acquire)
                const uint8_t *d_at = nullptr;
                uint64_t avail = 0;
                bool eof = false;
                std::string msg;
                int rc = r->src->acquire(pos, end - pos + 1, &d_at, &avail, &eof, &msg);  // one segment more
                if (rc) return fail(r, rc, msg);
                if (r->src->peek_mark(0, &own)) break;
                if (eof) {
                    own = ~0ull >> 1;  // (behind everything: the stream holds nothing of this shard's)
                    break;
                }
                end = pos + avail;
            }
            r->range_preset = true;
            r->preset_pos = own;
            r->data0_is_line_start = c_begin == 0;
        }
        if (fasta) {
            r->fa_shard = true;
            r->range_eof = true;  // (a shard of a FASTA is a FASTA file of its own)
        }
    } else {
        r->src = make(0, n, false, nullptr);
        if (r->format == EXG_FMT_VCF) {
            int rc = source_header_prefix(r, r->src.get(), *out_blk);
            if (rc) return rc;
        }
    }
    blk = out_blk;
    return EXG_OK;
}

// Newlines in the decoded bytes of the members / frames in front of a shard's own (file bytes [0, own_c_begin)), by a decoder
// of their own: a segment is counted on the device and dropped, nothing but it is resident (the phase of a shard of a
// compressed FASTQ when the bytes around the cut do not tell it: a memchr over the page cache in the text case).
static int count_newlines_in_front(exg_reader *r, uint64_t *out) {
    *out = 0;
    if (!r->own_c_begin) return EXG_OK;
    const std::string &path = r->files[r->file_idx - 1];
    const int fd = r->fd_keep->fd;
    struct stat st;
    if (fstat(fd, &st)) return fail(r, EXG_E_IO, "cannot stat '" + path + "'");
    const uint64_t reserve = source_reserve(r), target = r->device_batch_bytes;
    std::unique_ptr<SegmentProducer> prod = r->compression == kGzip
                                                ? make_gzip_producer(r, fd, 0, r->own_c_begin, target, path, true, reserve, nullptr)
                                                : make_zstd_producer(r, fd, (uint64_t)st.st_size, 0, r->own_c_begin, target, path, reserve, nullptr);
    DecodedSource head(r->device, r->stream, std::move(prod), reserve, 1, &r->meter);
    if (!r->d_phase && !(r->d_phase = exg_rd::dev_pool()->take(r->device, 4096))) return fail(r, EXG_E_HIP, "out of device memory");
    for (uint64_t pos = 0;;) {
        const uint8_t *d_at = nullptr;
        uint64_t avail = 0;
        bool eof = false;
        std::string msg;
        int rc = head.acquire(pos, 1, &d_at, &avail, &eof, &msg);
        if (rc) return fail(r, rc, msg);
        if (avail) {
            unsigned long long nl = 0;
            const uint64_t skew = pos & 15;  // (an address is congruent to its stream offset mod 16)
            rc = exg_count_newlines(d_at - skew, skew, skew + avail, (uint64_t *)r->d_phase, r->stream);
            if (rc) return fail(r, rc, exg_last_error_message());
            RD_HIP(r, hipMemcpyAsync(&nl, r->d_phase, 8, hipMemcpyDeviceToHost, r->stream));
            RD_HIP(r, hipStreamSynchronize(r->stream));
            *out += nl;
            pos += avail;
        }
        if (eof) break;
        if (!avail) return fail(r, EXG_E_HIP, "internal: a decoded source returned nothing before its end");
    }
    std::string e;
    const int frc = head.finish(&e);
    return frc ? fail(r, frc, e) : EXG_OK;
}

int open_next_file(exg_reader *r) {
    if (int jrc = r->finish_source()) return jrc;
    r->src.reset();  // (its thread reads the previous file's descriptor)
    const std::string &p = r->files[r->file_idx++];
    int fd = open(p.c_str(), O_RDONLY);
    if (fd < 0) return fail(r, EXG_E_IO, "cannot open '" + p + "': " + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) {
        const std::string why = strerror(errno);
        close(fd);
        return fail(r, EXG_E_IO, "cannot stat '" + p + "': " + why);
    }
    if (!S_ISREG(st.st_mode)) {
        close(fd);
        return fail(r, EXG_E_IO, "'" + p + "' is not a regular file");
    }
    auto blk = std::make_shared<PinnedBlock>();
    blk->n = (size_t)st.st_size;
    double t0 = now_s();
    if (r->compression != kNone) {
        // decoded by a producer thread that reads the file with pread: nothing is mapped
    } else if (blk->n) {
        // The file is mapped, not copied: DataChunk strings point straight into the page cache mapping
        // (kept alive by the chunks); bytes travel to the device through a pinned bounce buffer.
        void *m = mmap(nullptr, blk->n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            close(fd);
            return fail(r, EXG_E_IO, "cannot map '" + p + "': " + strerror(errno));
        }
        blk->p = m;
        blk->mapped = blk->n;
        // (another process may truncate the file while its rows are out: zeros + EXG_E_IO at the next call, not a SIGBUS that
        // ends the DuckDB process — exg_map_guard.hpp)
        blk->guard = MapGuard::add(m, blk->n);
        if (blk->guard < 0) {  // (the table holds 4096 mappings: chunks of thousands of files are still out)
            blk.reset();       // (unmaps)
            close(fd);
            return fail(r, EXG_E_NOMEM, "too many text files mapped at once (4096): release chunks or close readers before opening '" + p + "'");
        }
    } else {
        hipError_t he = hipHostMalloc(&blk->p, 64, hipHostMallocDefault);
        if (he != hipSuccess) {
            close(fd);
            return fail(r, EXG_E_HIP, std::string("hipHostMalloc failed: ") + hipGetErrorString(he));
        }
        memset(blk->p, 0, 64);
    }
    r->fd_keep.reset(new exg_reader::FdCloser{fd});
    TRACE("mmap(file)", t0);
    r->range_preset = false;
    r->range_eof = true;
    r->data0_is_line_start = true;
    r->own_c_begin = 0;
    r->exact_nl_known = false;
    (void)r->join_prefetch();
    r->drop_prefetch2();
    if (r->pf.valid && r->up_stream) (void)hipStreamSynchronize(r->up_stream);  // a prefetch of the previous file
    r->pf.valid = false;
    if (r->compression != kNone) {
        int rc = open_source(r, blk, p, (uint64_t)st.st_size);  // replaces blk (decoded size unknown; VCF: the header prefix on the host)
        if (rc) return rc;
    }
    r->file = blk;
    r->file_pos = 0;
    r->file_done = false;
    if (r->format == EXG_FMT_VCF) {
        // header = the leading '#' lines (noodles-vcf read_header); it must hold the #CHROM line
        const char *d = (const char *)blk->p;
        const size_t hn = r->compression != kNone ? (size_t)r->gz_header_prefix : blk->n;  // gzip / zstd: only the header prefix is on the host
        size_t pos = 0;
        bool chrom = false;
        while (pos < hn && d[pos] == '#') {
            if (hn - pos >= 6 && memcmp(d + pos, "#CHROM", 6) == 0) chrom = true;
            const void *nl = memchr(d + pos, '\n', hn - pos);
            pos = nl ? (size_t)((const char *)nl - d) + 1 : hn;
        }
        if (!chrom) return fail(r, EXG_E_PARSE, std::string(exg_parse_error_string(EXG_PE_VCF_NO_HEADER)) + " in '" + p + "'");
        r->vcf_header_bytes = pos;
        r->file_pos = pos;
    }
    // byte-range shard of this file: [lo, hi) of the bytes behind the header; records / lines belong to the shard they END in
    r->range_hi = r->src ? ~0ull : blk->n;  // (a decoded stream ends where its source says so)
    r->shard_first = false;
    r->data_base = r->file_pos;  // 0, or the end of the VCF header
    // Which scan first: from the first MiB behind the header, which is mapped anyway (a decoded stream has no bytes on the host: its
    // first batches' results decide, as before).  The batches' result flags correct the choice either way.
    if (!r->src && blk->p && blk->n > r->file_pos && !getenv("EXG_NO_ALGO_HINT"))
        r->fused_algo = (uint32_t)exg_scan_algo_hint(r->format, (const uint8_t *)blk->p + r->file_pos, blk->n - r->file_pos);
    static const bool no_ramp = getenv("EXG_NO_RAMP") != nullptr;
    r->ramp_bytes = (!r->src && r->format != EXG_FMT_FASTA && !no_ramp && r->device_batch_bytes > kRampFirstBytes) ? kRampFirstBytes : 0;
    if (r->fa_shard) {
        // a shard of a compressed FASTA: the run of whole records from the first '>' line that begins in its own bytes to the
        // first that begins behind them (found while scanning: next_batch), like a text shard's run
        r->shard_first = false;
        if (r->preset_pos == 0 && r->data0_is_line_start) {
            r->file_pos = 0;
        } else {
            uint64_t at = ~0ull;
            int rc = source_find_record(r, r->preset_pos, &at);
            if (rc) return rc;
            if (at == ~0ull) r->file_done = true;  // no record begins in this shard's bytes (or behind them)
            else r->file_pos = at;
        }
    } else if (r->range_preset) {  // BGZF / zstd shard: the members / frames were chosen in open_source
        // its stream begins with the file (header and all) or somewhere behind the header
        r->data_base = r->data0_is_line_start ? r->data_base : 0;
        r->file_pos = std::max<uint64_t>(r->preset_pos, r->data_base);
        r->shard_first = r->file_pos > r->data_base;
    } else if (r->shard_count > 1 && r->format == EXG_FMT_FASTA) {
        // FASTA: a record belongs to the shard in whose bytes its '>' line BEGINS, and a shard is the run of whole
        // records from its first such line to the next shard's — scanned like a file of its own (a record is never
        // cut, however long its sequence: the run simply reaches as far as it has to)
        const char *d = (const char *)blk->p;
        const uint64_t N = blk->n;
        auto first_record_at_or_after = [&](uint64_t pos) -> uint64_t {
            if (pos == 0) return 0;
            for (uint64_t q = pos - 1; q + 1 < N;) {  // a line start is the byte behind a newline
                const void *hit = memchr(d + q, '\n', (size_t)(N - q));
                if (!hit) return N;
                q = (uint64_t)((const char *)hit - d) + 1;
                if (q < N && d[q] == '>') return q;
            }
            return N;
        };
        const uint64_t lo = (uint64_t)((unsigned __int128)N * r->shard_index / r->shard_count);
        const uint64_t hi = r->shard_index + 1 == r->shard_count ? N : (uint64_t)((unsigned __int128)N * (r->shard_index + 1) / r->shard_count);
        r->file_pos = first_record_at_or_after(lo);
        r->range_hi = hi == N ? N : first_record_at_or_after(hi);
        if (r->range_hi < r->file_pos) r->range_hi = r->file_pos;
        r->range_eof = true;  // the run is a FASTA file of its own
    } else if (r->shard_count > 1) {
        const uint64_t base = r->file_pos, span = blk->n - base;
        const uint64_t lo = base + (uint64_t)((unsigned __int128)span * r->shard_index / r->shard_count);
        const uint64_t hi = r->shard_index + 1 == r->shard_count
                                ? blk->n
                                : base + (uint64_t)((unsigned __int128)span * (r->shard_index + 1) / r->shard_count);
        r->file_pos = lo;
        r->range_hi = hi;
        r->shard_first = lo > base;
        r->range_eof = hi == blk->n;
    }
    return EXG_OK;
}

int n_string_cols(int format) { return format == EXG_FMT_FASTQ ? 4 : format == EXG_FMT_FASTA ? 3 : 9; }

int next_batch(exg_reader *r, bool count_only, uint64_t *n_records_out);

int advance_batch(exg_reader *r, bool *end) {
    *end = false;
    for (;;) {
        if (r->batch && r->batch_row < r->batch->n_rows) return EXG_OK;
        if (r->pending_error) {
            // rows before the failing record have been handed out; now surface the error
            std::string msg = std::string(exg_parse_error_string(r->pending_error)) + " at byte " + std::to_string(r->pending_error_offset) + " of " +
                              r->files[r->file_idx - 1];
            r->pending_error = 0;
            r->batch.reset();
            return fail(r, EXG_E_PARSE, msg);
        }
        if (r->file_done) {
            if (int jrc = r->finish_source()) return jrc;
            if (r->file_idx >= r->files.size()) {
                r->batch.reset();
                *end = true;
                return EXG_OK;
            }
            int rc = open_next_file(r);
            if (rc) return rc;
        }
        uint64_t k;
        int rc = next_batch(r, false, &k);
        if (rc) return rc;
    }
}

int ensure_device(exg_reader *r, uint64_t need_bytes) {
    if (r->d_ws && need_bytes <= r->d_in_cap) return EXG_OK;
    if (r->d_ws) {
        RD_HIP(r, hipStreamSynchronize(r->stream));
        r->free_device();
    }
    if (r->src)  // segments come at about the target size + what the scan carries over: provision once
        need_bytes = std::max<uint64_t>(need_bytes, r->device_batch_bytes + r->device_batch_bytes / 4 + r->src->reserve() + 4096);
    else if (r->format != EXG_FMT_FASTA && r->file)  // room for a prefetched batch (its slack included), small files stay small
        need_bytes = std::max<uint64_t>(need_bytes, std::min<uint64_t>(r->device_batch_bytes, r->file->n) + kPrefetchSlack + 64);
    uint64_t cap = std::max<uint64_t>(need_bytes, 1 << 16);
    r->d_in_cap = cap;
    // Rows the output vectors can hold.  Realistic density first (a FASTQ record under 32 bytes, a FASTA record
    // or a VCF line under 16 would be unusual) — the worst case (FASTQ "@\n\n+\n" = 5 bytes, FASTA ">a\n" minus
    // LF, a blank VCF line) would pin 16 B x 9 columns per input BYTE of device memory; a batch that does
    // overflow is reported by the kernels (EXG_RF_CAPACITY) and rescanned with worst-case vectors.
    const uint64_t div = r->worst_case_rows ? (r->format == EXG_FMT_FASTQ ? 5 : r->format == EXG_FMT_FASTA ? 2 : 1)
                                            : (r->format == EXG_FMT_FASTQ ? 32 : 16);
    r->cap_records = cap / div + 4096;
    r->ws_bytes = exg_scan_workspace_bytes(r->format, cap);
    if (r->mem_cap && !r->ws_full) {
        // The general path's line index is provisioned for min(n, 1 Mi) lines whatever the batch size (8 MiB; FASTA keeps
        // four such arrays): under a memory budget, one line per 8 bytes + 64 Ki — a denser batch is reported
        // (EXG_RF_INDEX_OVERFLOW) and scanned again with the full workspace
        const uint64_t arrays = r->format == EXG_FMT_FASTA ? 4 : 1;
        const exg::FastqWsLayout l = exg::fastq_ws_layout(cap, 0, arrays);
        const uint64_t lines = cap / 8 + 65536 + 8;
        r->ws_bytes = std::min<uint64_t>(r->ws_bytes, l.off_nl_pos + (lines + 2) * 8 * arrays);
    }
    // two upload slots (a FASTA under a memory cap: one, its batches are then not prefetched); a decoded stream is scanned in its
    // segments: none (FASTA: one, for the 16-byte aligned copy its scan wants)
    int arc = 0;
    static const bool no_fasta_prefetch = getenv("EXG_NO_FASTA_PREFETCH") != nullptr;
    const int n_slots = r->format == EXG_FMT_FASTA ? (r->src || r->mem_cap || no_fasta_prefetch ? 1 : 2) : r->src ? 0 : 2;
    for (int k = 0; k < n_slots; k++)
        if ((arc = r->dev_alloc(&r->d_in_slot[k], cap + 64))) return arc;
    r->d_in = r->d_in_slot[0];
    r->cur_slot = 0;
    if (!r->up_stream) {
        RD_HIP(r, exg_rd::stream_pool()->take(r->device, &r->up_stream));
        for (int k = 0; k < 2; k++) RD_HIP(r, hipEventCreateWithFlags(&r->up_done_of[k], hipEventDisableTiming));
    }
    if ((arc = r->dev_alloc(&r->d_ws, r->ws_bytes))) return arc;
    for (int k = 0; k < 2; k++)
        if ((arc = r->dev_alloc(&r->d_valid[k], (r->cap_records + 63) / 64 * 8))) return arc;
    for (int c = 0; c < n_string_cols(r->format); c++)
        if ((arc = r->dev_alloc(&r->d_cols[c], r->cap_records * 16))) return arc;
    if (r->format == EXG_FMT_VCF) {
        if ((arc = r->dev_alloc(&r->d_pos, r->cap_records * 8))) return arc;
        if ((arc = r->dev_alloc(&r->d_qual, r->cap_records * 4))) return arc;
    }
    if (r->format == EXG_FMT_FASTA && (arc = r->dev_alloc(&r->d_payload, cap + 64))) return arc;
    if (r->has_filter) {
        if ((arc = r->dev_alloc(&r->d_row_map, r->cap_records * 4 + 64))) return arc;
        if ((arc = r->dev_alloc(&r->d_gather, r->cap_records * 16))) return arc;
        if ((arc = r->dev_alloc(&r->d_filter_tmp, (r->cap_records + 1 + exg::arrow::scan_tmp_entries(r->cap_records)) * 8))) return arc;
    }
    if (!r->d_res && !(r->d_res = exg_rd::dev_pool()->take(r->device, 4096))) return fail(r, EXG_E_HIP, "out of device memory");
    return EXG_OK;
}

// Scan the next device batch of the current file.  On return r->batch holds its host vectors
// (n_rows may be 0 when the file is exhausted).  count_only: no column leaves the device.
int truncated_while_read(exg_reader *r) {
    if (!r->file || r->file->guard < 0 || !MapGuard::hit(r->file->guard)) return EXG_OK;
    return fail(r, EXG_E_IO, "'" + r->files[r->file_idx - 1] + "' was truncated while it was being read");
}

int next_batch(exg_reader *r, bool count_only, uint64_t *n_records_out) {
    *n_records_out = 0;
    r->batch.reset();
    r->batch_row = 0;
    if (int trc = truncated_while_read(r)) return trc;
    if (r->file_done) return EXG_OK;  // (opening the file found nothing of this shard's in it)
    uint64_t want = r->device_batch_bytes;
    double t_batch = now_s();
    for (;;) {
        const uint64_t remaining = r->range_hi > r->file_pos ? r->range_hi - r->file_pos : 0;
        if (remaining == 0) {
            r->file_done = true;
            return EXG_OK;
        }
        uint64_t n = std::min<uint64_t>(want, remaining);
        // (the head of a text file: a small batch first — exg_reader.hpp ramp_bytes; the uploads behind it are sized where they are issued)
        if (r->ramp_bytes && !r->src && !r->pf.valid && want == r->device_batch_bytes) n = std::min<uint64_t>(n, r->next_ramp());
        bool range_end = n == remaining;                    // the batch reaches the end of this reader's bytes ...
        bool eof = range_end && r->range_eof;               // ... which is the end of the file unless a later shard follows
        // first batch of a shard that begins inside the file: up to 1 MiB in front of it travels along (`lead`), so that
        // the record / line that ends behind the cut — it belongs to this shard — has its beginning in the buffer
        uint64_t shard_halo = 0;
        if (r->shard_first) {
            const uint64_t halo_max = r->halo_want;
            const uint64_t base = r->data_base;
            const uint64_t from = r->file_pos - std::min<uint64_t>(halo_max, r->file_pos - base);
            // (a buffer that already lives in HBM must be entered at a 16-byte boundary: a few bytes of the header's
            // last line may then come along in front — they end inside the halo and are nobody's rows)
            shard_halo = r->file_pos - (r->src ? (std::max<uint64_t>(base, from) & ~15ull) : std::max<uint64_t>(base, from & ~15ull));
        }
        // a decoded stream: the bytes from file_pos on (and the halo in front) made contiguous in HBM — everything up to the end of
        // the segment that holds them is this batch
        const uint8_t *src_at = nullptr;  // device address of stream byte src_pos
        uint64_t src_pos = 0;
        bool src_eof = false;
        if (r->src) {
            src_pos = r->file_pos - shard_halo;
            uint64_t avail = 0;
            std::string msg;
            // What is asked of the stream is "the rest of the segment that holds file_pos", not a batch's worth of bytes: a
            // segment shorter than a device batch (the first windows of a file are small on purpose: latency) used to be MERGED
            // with the one behind it — a device copy of both, a pinned block of an odd size for the merged batch's payload
            // (hipHostMalloc: ~1 ms per 10 MiB) and no host mirror.  Only a batch that held no complete record asks for more.
            const uint64_t ask = (want > r->device_batch_bytes || r->format == EXG_FMT_FASTA) ? want : std::min<uint64_t>(want, 1u << 20);
            const double t_acq = now_s();
            int arc = r->src->acquire(src_pos, ask + shard_halo, &src_at, &avail, &src_eof, &msg);
            TRACE("acquire(decoded segment)", t_acq);
            trace_at("C acquired", r->n_batches);
            if (arc) return fail(r, arc, msg + (msg.find(r->files[r->file_idx - 1]) == std::string::npos ? " in '" + r->files[r->file_idx - 1] + "'" : ""));
            r->n_segments = r->src->segments_consumed() + 1;
            if (avail <= shard_halo && src_eof) {  // nothing behind file_pos: the stream has ended
                r->file_done = true;
                return EXG_OK;
            }
            n = avail - shard_halo;
            range_end = src_eof;
            eof = range_end && r->range_eof;
            if (r->fa_shard) {
                // where does the first record begin that is NOT this shard's?  (mark 1: the decoded offset behind its own
                // members / frames — while it is not set, nothing handed out so far lies behind it)
                uint64_t own_end = 0;
                if (r->fa_end == ~0ull && r->src->peek_mark(1, &own_end)) {
                    if (own_end <= r->file_pos) {
                        r->fa_end = r->file_pos;
                    } else if (own_end < src_pos + avail) {
                        unsigned long long pos = ~0ull;
                        if (!r->d_phase && !(r->d_phase = dev_pool()->take(r->device, 4096))) return fail(r, EXG_E_HIP, "out of device memory");
                        arc = exg_fasta_find_record(src_at, own_end - src_pos, avail, 0, (uint64_t *)r->d_phase, r->stream);
                        if (arc) return fail(r, arc, exg_last_error_message());
                        RD_HIP(r, hipMemcpyAsync(&pos, r->d_phase, 8, hipMemcpyDeviceToHost, r->stream));
                        RD_HIP(r, hipStreamSynchronize(r->stream));
                        if (pos != ~0ull) r->fa_end = src_pos + pos;
                    }
                }
                if (r->fa_end != ~0ull) {
                    if (r->fa_end <= r->file_pos) {
                        r->file_done = true;
                        return EXG_OK;
                    }
                    n = r->fa_end - r->file_pos;
                    range_end = eof = true;
                }
            }
        }
        int rc = ensure_device(r, n + shard_halo + 16);
        if (rc) return rc;
        // Input of the scan: the inflated bytes already in HBM (gzip), the prefetched slot, or a synchronous
        // H2D copy.  In the first two cases the batch start is only byte aligned: the buffer starts at the
        // 16-byte boundary below it and `lead` skips the tail of the previous record (whose last '\n' is
        // then inside the buffer).
        // The batch about to be scanned is on its way (or there); if its upload is the one this call will use, the batch
        // AFTER it starts travelling now, into the slot of the batch that was scanned last (free: its columns have left) —
        // issued after this call's scan, an upload began only when the link had already been idle for a scan + a D2H.
        static const bool no_prefetch = getenv("EXG_NO_PREFETCH") != nullptr;
        if (!no_prefetch && r->pf.valid && !r->pf2.valid && !r->src && r->format != EXG_FMT_FASTA && want == r->device_batch_bytes &&
            r->file_pos >= r->pf.file_start && r->file_pos < r->pf.file_start + r->pf.len && r->d_in_slot[r->pf.slot ^ 1] &&
            !r->up_thread_of[r->pf.slot ^ 1].joinable()) {
            const uint64_t end1 = r->pf.file_start + r->pf.len;  // where the coming batch's bytes end
            if (end1 < r->range_hi) {
                const uint64_t slack = std::min<uint64_t>(kPrefetchSlack, (end1 - r->file_pos) / 2);
                const uint64_t start = (end1 - slack) & ~15ull;
                const uint64_t len = std::min<uint64_t>(r->range_hi - start, r->next_ramp() + slack);
                if (len + 16 <= r->d_in_cap) start_upload(r, &r->pf2, start, len, r->pf.slot ^ 1);
            }
        }
        trace_at("N batch begins", r->n_batches);
        if ((rc = r->join_prefetch())) return rc;  // the upload thread of the previous call (its error is this call's)
        trace_at("N upload thread joined", r->n_batches);
        const uint8_t *h = (const uint8_t *)r->file->p + r->file_pos;
        const void *d_input = nullptr;
        uint64_t lead = 0;
        uint64_t batch_end = r->file_pos + n;  // file offset one past the bytes of this batch
        std::shared_ptr<PinnedBlock> gz_payload;  // gzip: this batch's inflated bytes on the host (string_t payload)
        std::shared_ptr<HostMirror> gz_mirror;    // ... when the producer sent them ahead (exg_rd_source.hpp); then only
        uint64_t gz_front = 0;                    // ... the first gz_front bytes of the batch (the carried tail) are copied here
        bool compact = false;                     // ... or: only the selected columns' out-of-line strings travel (a side buffer)
        if (r->src) {
            lead = shard_halo + (src_pos & 15);
            d_input = src_at - (src_pos & 15);
            n += lead;
            if (r->format == EXG_FMT_FASTA) {
                // the FASTA scan wants its batch on a 16-byte boundary with nothing in front: a device copy (the decoders
                // in front of it run at a few percent of what a copy does)
                if (shard_halo) return fail(r, EXG_E_INVALID_ARG, "internal: a FASTA batch has no halo");
                n -= lead;
                RD_HIP(r, hipMemcpyAsync(r->d_in_slot[0], src_at, n, hipMemcpyDeviceToDevice, r->stream));
                RD_HIP(r, hipMemsetAsync((char *)r->d_in_slot[0] + n, 0, 16, r->stream));
                r->d_in = r->d_in_slot[0];
                d_input = r->d_in;
                lead = 0;
            }
            // A projection that leaves payload-bearing columns out (SELECT name FROM read_fastq('x.fastq.gz'); chrom, pos, ref of a
            // bgzip VCF): the decoded bytes stay in HBM, the out-of-line strings of the selected columns are closed up into a side
            // buffer behind the scan and only that crosses PCIe (payload_*_from_col, repoint_strings).  Not when a nested VCF
            // column is selected: its element views are cut out of the line's text by the emitter.
            const PayloadRoute route = (!count_only && !r->arrow_emit) ? payload_route(r) : kPayloadNone;
            compact = route == kPayloadCompact;
            if (compact) {
                h = (const uint8_t *)(uintptr_t)0x100000000000ull + (r->file_pos - lead);  // (a base the side buffer's pointers replace)
            } else if (!count_only && !r->arrow_emit) {
                // The strings of this batch point into host memory that holds the decoded bytes.  From the first such batch on
                // the producer sends every segment to the host as it hands it over (HostMirror: the copy runs while the segment
                // waits in the queue and while this thread is busy with the batch in front): the batch then points into that
                // block, and only the bytes in front of the segment's own — the tail carried over from the segment before —
                // are copied here.  A segment without a mirror (pushed before the first call, a block of the consumer's own
                // making, FASTA) is copied behind the scan as before.
                const uint8_t *h_at = nullptr;
                uint64_t m_from = 0;
                if (route == kPayloadMirror) r->src->want_host_mirror();
                if (route == kPayloadMirror && r->src->host_view((const uint8_t *)d_input, &h_at, &m_from, &gz_mirror)) {
                    gz_payload = gz_mirror->blk;
                    h = h_at;
                    gz_front = m_from > src_pos - (src_pos & 15) ? std::min<uint64_t>(n, m_from - (src_pos - (src_pos & 15))) : 0;
                } else {
                    gz_mirror.reset();
                    gz_payload = std::make_shared<PinnedBlock>();
                    size_t cap = n + 64;
                    gz_payload->p = global_pool()->take(&cap);
                    if (!gz_payload->p) return fail(r, EXG_E_HIP, "out of pinned host memory for the inflated bytes");
                    gz_payload->cap = cap;
                    gz_payload->pooled = true;
                    gz_payload->n = n;
                    h = (const uint8_t *)gz_payload->p;
                }
            } else {
                // COUNT(*) / the Arrow stream: no host copy; h is only the base the device subtracts again
                h = (const uint8_t *)(uintptr_t)0x100000000000ull + (r->file_pos - lead);
            }
        } else if (r->pf.valid && want == r->device_batch_bytes && r->file_pos >= r->pf.file_start &&
                   r->file_pos < r->pf.file_start + r->pf.len) {
            const uint64_t off = r->file_pos - r->pf.file_start;
            lead = off & 15;
            r->cur_slot = r->pf.slot;
            r->d_in = r->d_in_slot[r->cur_slot];
            d_input = (const uint8_t *)r->d_in + (off - lead);
            batch_end = r->pf.file_start + r->pf.len;
            n = batch_end - r->file_pos + lead;
            range_end = batch_end == r->range_hi;
            eof = range_end && r->range_eof;
            h -= lead;
            r->pf.valid = false;
            RD_HIP(r, hipStreamWaitEvent(r->stream, r->up_done_of[r->cur_slot], 0));
        } else {
            if (r->pf.valid) RD_HIP(r, hipStreamSynchronize(r->up_stream));  // a prefetch that missed: let it land first
            r->pf.valid = false;
            r->d_in = r->d_in_slot[r->cur_slot];
            lead = shard_halo;
            int rc2;
            if (r->format == EXG_FMT_FASTA && n > (512ull << 20)) {
                // a whole genome in one batch: through two 256 MiB pinned windows, not one pinned block of its size
                rc2 = upload_file(r, r->d_in, n, r->file_pos);
                if (!rc2 && hipMemsetAsync((char *)r->d_in + n, 0, 16, r->stream) != hipSuccess)
                    rc2 = fail(r, EXG_E_HIP, "hipMemsetAsync failed");
            } else {
                rc2 = upload_range(r, r->file_pos - lead, n + lead, r->cur_slot, r->stream);
            }
            if (rc2) return rc2;
            d_input = r->d_in;
            n += lead;
            h -= lead;
        }
        uint64_t first_line_index = 0;
        if (lead && r->shard_first && r->format == EXG_FMT_FASTQ) {
            // the 4-line phase of the line that holds the shard's first byte, from the bytes around the cut ('@' opens a
            // record but also quality lines, so several records are looked at: exg_fastq_guess_phase)
            if (!r->d_phase && !(r->d_phase = exg_rd::dev_pool()->take(r->device, 4096))) return fail(r, EXG_E_HIP, "out of device memory");
            uint32_t guess = 0xFFFFFFFFu;
            rc = exg_fastq_guess_phase(d_input, n, lead, (uint32_t *)r->d_phase, r->stream);
            if (rc) return fail(r, rc, exg_last_error_message());
            RD_HIP(r, hipMemcpyAsync(&guess, r->d_phase, 4, hipMemcpyDeviceToHost, r->stream));
            RD_HIP(r, hipStreamSynchronize(r->stream));
            if (guess <= 3) {
                uint8_t prev = 0;
                if (r->src) {
                    RD_HIP(r, hipMemcpyAsync(&prev, (const uint8_t *)d_input + lead - 1, 1, hipMemcpyDeviceToHost, r->stream));
                    RD_HIP(r, hipStreamSynchronize(r->stream));
                } else {
                    prev = ((const uint8_t *)r->file->p)[r->file_pos - 1];
                }
                first_line_index = prev == '\n' ? guess : (guess + 3) % 4;
            } else if (r->src) {
                // a shard of a compressed input: exact when the halo begins with the file (the newlines in front are then all
                // in HBM); else (few lines in view — records far longer than the halo — or several phases fit) the newlines in
                // front of the shard's own bytes are counted by a decoder of their own, segment by segment
                unsigned long long nl = 0;
                if (r->data0_is_line_start && lead == r->file_pos) {
                    rc = exg_count_newlines(d_input, 0, lead, (uint64_t *)r->d_phase, r->stream);
                    if (rc) return fail(r, rc, exg_last_error_message());
                    RD_HIP(r, hipMemcpyAsync(&nl, r->d_phase, 8, hipMemcpyDeviceToHost, r->stream));
                    RD_HIP(r, hipStreamSynchronize(r->stream));
                } else {
                    if (!r->exact_nl_known) {
                        if ((rc = count_newlines_in_front(r, &r->exact_nl))) return rc;
                        r->exact_nl_known = true;
                    }
                    nl = r->exact_nl;
                }
                first_line_index = nl;
            } else {
                // too few lines around the cut to tell (a tiny file, a tiny shard) or several phases fit: count the
                // newlines in front of it — exact, and only as slow as a memchr over the page cache
                const char *d = (const char *)r->file->p;
                uint64_t nl = 0;
                for (const char *q = d, *end = d + r->file_pos; q < end;) {
                    const void *hit = memchr(q, '\n', (size_t)(end - q));
                    if (!hit) break;
                    nl++;
                    q = (const char *)hit + 1;
                }
                first_line_index = nl;
            }
        }
        exg_scan_result res;
        TraceRange scan_range(r->format == EXG_FMT_FASTQ ? "exg: scan fastq batch" : r->format == EXG_FMT_VCF ? "exg: scan vcf batch" : "exg: scan fasta batch");
        const bool no_store = count_only && !r->has_filter;  // a predicate needs the columns even for COUNT(*)
        // a line starts at d_input[0] when the batch is record aligned, or when a shard's halo reaches back to the
        // first byte behind the header
        const bool at_line_start = lead == 0 || (r->shard_first && shard_halo && lead == shard_halo && r->data0_is_line_start &&
                                                 r->file_pos - lead == r->data_base);
        const uint32_t fl = (at_line_start ? EXG_F_BOF : 0u) | (eof ? EXG_F_EOF : 0u) | (no_store ? EXG_F_NO_STORE : 0u);
        std::shared_ptr<Batch> b;
        bool fused_first = false;
        std::function<int()> rescan_general;
        if (r->format == EXG_FMT_FASTQ) {
            exg_fastq_scan_args a;
            memset(&a, 0, sizeof a);
            a.d_input = d_input;
            a.n_bytes = n;
            a.lead = lead;
            a.first_line_index = first_line_index;
            a.payload_base = (uint64_t)(uintptr_t)h;
            a.flags = fl;
            a.algo = r->fused_algo;
            a.d_name = (exg_string_t *)r->d_cols[0];
            a.d_description = (exg_string_t *)r->d_cols[1];
            a.d_sequence = (exg_string_t *)r->d_cols[2];
            a.d_quality = (exg_string_t *)r->d_cols[3];
            a.d_description_validity = (uint64_t *)r->d_valid[0];
            a.capacity_records = r->cap_records;
            a.d_workspace = r->d_ws;
            a.workspace_bytes = r->ws_bytes;
            a.d_result = (exg_scan_result *)r->d_res;
            a.stream = r->stream;
            rc = exg_fastq_scan(&a);
            fused_first = true;
            rescan_general = [a]() mutable {
                a.algo = EXG_ALGO_MULTIPASS;
                return exg_fastq_scan(&a);
            };
        } else if (r->format == EXG_FMT_VCF) {
            exg_vcf_scan_args a;
            memset(&a, 0, sizeof a);
            a.d_input = d_input;
            a.n_bytes = n;
            a.lead = lead;
            a.payload_base = (uint64_t)(uintptr_t)h;
            a.flags = fl;
            a.algo = r->fused_algo;
            for (int c = 0; c < 9; c++) a.d_fields[c] = (exg_string_t *)r->d_cols[c];
            if (!r->arrow_emit) {
                // The projection reaches the kernel (a NULL column is skipped: 16 B per row less to write): POS / QUAL leave as
                // numbers, their text is nobody's; CHROM / REF are written when they are selected or the predicate reads them.
                // The other five feed the nested columns, which are built — and validated: a malformed value is the same error
                // whether or not its column is selected — from their text
                a.d_fields[1] = a.d_fields[5] = nullptr;
                for (int c : {0, 3})
                    if (!r->want(c) && !((r->filter_cols >> c) & 1ull)) a.d_fields[c] = nullptr;
            }
            a.d_pos = (int64_t *)r->d_pos;
            a.d_qual = (float *)r->d_qual;
            a.d_qual_validity = (uint64_t *)r->d_valid[0];
            a.d_formats_validity = (uint64_t *)r->d_valid[1];
            a.capacity_records = r->cap_records;
            a.d_workspace = r->d_ws;
            a.workspace_bytes = r->ws_bytes;
            a.d_result = (exg_scan_result *)r->d_res;
            a.stream = r->stream;
            if (r->flat_pending) {  // (the batch before: its flat columns' copies read what this scan writes)
                RD_HIP(r, hipStreamWaitEvent(r->stream, r->flat_ev, 0));
                r->flat_pending = false;
            }
            rc = exg_vcf_scan(&a);
            fused_first = true;
            rescan_general = [a]() mutable {
                a.algo = EXG_ALGO_MULTIPASS;
                return exg_vcf_scan(&a);
            };
        } else {
            b = std::make_shared<Batch>();
            if (!count_only) {
                b->payload = b->host.alloc(n + 64);
                if (!b->payload) return fail(r, EXG_E_HIP, "out of pinned host memory");
            }
            exg_fasta_scan_args a;
            memset(&a, 0, sizeof a);
            a.d_input = d_input;
            a.n_bytes = n;
            a.payload_base = (uint64_t)(uintptr_t)h;
            a.seq_payload_base = (uint64_t)(uintptr_t)b->payload;
            a.flags = fl;
            a.d_id = (exg_string_t *)r->d_cols[0];
            a.d_description = (exg_string_t *)r->d_cols[1];
            a.d_sequence = (exg_string_t *)r->d_cols[2];
            a.d_description_validity = (uint64_t *)r->d_valid[0];
            a.d_seq_payload = (uint8_t *)r->d_payload;
            a.capacity_records = r->cap_records;
            a.d_workspace = r->d_ws;
            a.workspace_bytes = r->ws_bytes;
            a.d_result = (exg_scan_result *)r->d_res;
            a.stream = r->stream;
            rc = exg_fasta_scan(&a);
        }
        if (rc) return fail(r, rc, exg_last_error_message());
        r->n_batches++;
        double t_scan = now_s();
        rc = exg_fetch_result((const exg_scan_result *)r->d_res, r->stream, &res);
        if (rc) return fail(r, rc, exg_last_error_message());
        if (fused_first && (res.flags & EXG_RF_FALLBACK)) {
            // (no fused launch gives a batch up any more — long records, dense halves and bytes >= 0x80 are the any-shape
            // scan's —; should one ever say so, the general path takes the batch)
            rc = rescan_general();
            if (rc) return fail(r, rc, exg_last_error_message());
            rc = exg_fetch_result((const exg_scan_result *)r->d_res, r->stream, &res);
            if (rc) return fail(r, rc, exg_last_error_message());
            res.flags |= EXG_RF_FALLBACK;
        }
        if (res.flags & EXG_RF_REDO) {
            // sticky (exg_reader.hpp) — when the marks are the input's shape: more than an eighth of the batch's super-tiles.  The
            // odd long read in a short-read file is cheaper redone (its tiles only) than paid for by the any-shape scan's ~20 % on
            // every batch behind it
            const uint64_t tile_bytes = r->format == EXG_FMT_FASTQ ? 3u * 16384u : 2u * 16384u;
            if (res.redo_tiles * 8 > n / tile_bytes) r->fused_algo = EXG_ALGO_FUSED_FULL;
        }
        if (r->format == EXG_FMT_VCF && res.n_lines) {
            // the any-shape scan on WIDE lines (cohort VCFs) leaves the rows to a kernel of their own (EXG_ALGO_FUSED_INDEX: exg_vcf.hip):
            // measured over line widths (tools/vcf_index_crossover.py, TB/s indexed against rows inside): level at 483 B a line, 2.72
            // against 2.21 at 882 B, 3.13 against 2.15 at 1.7 kB, 3.82 against 2.28 at 10 kB — the switch at an average of 640 B;
            // sticky both ways with a gap between the thresholds (EXG_NO_VCF_INDEX: never — A/B)
            const bool no_index = getenv("EXG_NO_VCF_INDEX") != nullptr;  // (per batch: the tests switch it inside one process)
            const uint64_t per_line = (n - lead) / res.n_lines;
            if (r->fused_algo == EXG_ALGO_FUSED_FULL && per_line >= 640 && !no_index) r->fused_algo = EXG_ALGO_FUSED_INDEX;
            else if (r->fused_algo == EXG_ALGO_FUSED_INDEX && per_line < 448) r->fused_algo = EXG_ALGO_FUSED_FULL;
        }
        TRACE("wait(h2d) + scan", t_scan);
        trace_at("N scan result", r->n_batches);
        if (r->shard_first && (res.flags & EXG_RF_HEAD_UNRESOLVED) && r->file_pos - shard_halo > r->data_base) {
            // The record that ends behind the cut begins in front of the halo (a long read, a very wide VCF line): it belongs
            // to this shard, so this shard looks further back — eight times as far, up to the first byte of the data — and
            // scans the batch again.  (The shard in front leaves the record alone: it ends behind ITS range.)
            RD_HIP(r, hipStreamSynchronize(r->stream));
            r->halo_want = r->halo_want > (~0ull >> 4) ? ~0ull : r->halo_want * 8;
            if (r->src) {
                // a decoded stream begins with its halo: its members / frames are chosen again
                r->src.reset();
                r->file_idx--;
                if ((rc = open_next_file(r))) return rc;
            }
            continue;
        }
        if ((res.flags & EXG_RF_INDEX_OVERFLOW) && r->mem_cap && !r->ws_full) {
            RD_HIP(r, hipStreamSynchronize(r->stream));  // denser lines than the budgeted workspace indexes: the full one, same batch again
            r->free_device();
            r->ws_full = true;
            continue;
        }
        if (res.flags & EXG_RF_INDEX_OVERFLOW)
            return fail(r, EXG_E_CAPACITY, "line index overflow in the general path (pathological line density)");
        if ((res.flags & EXG_RF_CAPACITY) && !no_store) {
            if (r->worst_case_rows) return fail(r, EXG_E_CAPACITY, "more rows than bytes allow: internal error");
            RD_HIP(r, hipStreamSynchronize(r->stream));  // denser rows than provisioned: worst-case vectors, same batch again
            r->free_device();
            r->worst_case_rows = true;
            continue;
        }
        if (res.n_records == 0 && !res.error_code && !eof && !range_end) {
            want *= 2;  // not even one complete record in the batch: widen it
            continue;
        }
        if (res.error_code) {
            r->pending_error = res.error_code;
            r->pending_error_offset = r->file_pos - lead + res.error_offset;
        }
        double t_pf = now_s();
        // While the columns travel back (and the consumer works through the chunks): the bytes the next batch will need
        // move into the other slot — unless they left at the top of this call already (pf2), which is the steady state
        {
            // FASTA (round 6: its batches were uploaded, scanned and sent back one after the other — 22.7 GB/s end to end against
            // FASTQ's 50): the scan wants its batch at a 16-byte boundary with nothing in front, and where the next batch begins —
            // behind this one's last whole record — is known with the scan's result: its upload starts HERE, at exactly that file
            // offset into the other slot's first byte (no slack, no lead), beside this batch's sequences on their way back
            const bool fasta = r->format == EXG_FMT_FASTA;
            const bool can = !range_end && !res.error_code && !r->src && want == r->device_batch_bytes && !no_prefetch && (!fasta || res.n_records);
            const uint64_t slack = fasta ? 0 : std::min<uint64_t>(kPrefetchSlack, (batch_end - r->file_pos) / 2);
            const uint64_t start = fasta ? r->file_pos + (res.consumed_bytes - lead) : (batch_end - slack) & ~15ull;
            const int other = r->cur_slot ^ 1;
            if (r->pf2.valid) {
                // (its length was chosen where it was issued: a step of the ramp, or a full batch)
                if (can && r->pf2.file_start == start && r->pf2.len > 0 && r->pf2.slot == other) {
                    r->pf = r->pf2;
                    r->pf2.valid = false;
                } else {
                    r->drop_prefetch2();  // (the batch turned out otherwise: an error, a retry, the end of the range)
                }
            }
            if (can && !r->pf.valid && r->d_in_slot[other] && !r->up_thread_of[other].joinable()) {
                const uint64_t ramp_before = r->ramp_bytes;
                const uint64_t len = std::min<uint64_t>(r->range_hi - start, r->next_ramp() + slack);
                if (len + 16 <= r->d_in_cap) start_upload(r, &r->pf, start, len, other);
                else r->ramp_bytes = ramp_before;
            }
        }
        TRACE("prefetch issue", t_pf);
        uint64_t k = res.n_records;
        const uint32_t *row_map = nullptr;
        if (r->has_filter && k && !r->arrow_emit) {
            // rows where the predicate is TRUE -> row map; the columns are gathered through it on their way out
            namespace ea = exg::arrow;
            ea::FilterCols fc;
            memset(&fc, 0, sizeof fc);
            const int nsc = n_string_cols(r->format);
            for (int c = 0; c < nsc; c++) {
                fc.kind[c] = ea::kColStr;
                fc.data[c] = r->d_cols[c];
                fc.d_base[c] = (const uint8_t *)d_input;
                fc.payload_base[c] = (uint64_t)(uintptr_t)h;
            }
            if (r->format == EXG_FMT_VCF) {
                fc.kind[1] = ea::kColI64, fc.data[1] = r->d_pos;
                fc.kind[5] = ea::kColF32, fc.data[5] = r->d_qual, fc.validity[5] = (const uint64_t *)r->d_valid[0];
                fc.validity[8] = (const uint64_t *)r->d_valid[1];
            } else {
                fc.validity[1] = (const uint64_t *)r->d_valid[0];
                if (r->format == EXG_FMT_FASTA) {
                    fc.d_base[2] = (const uint8_t *)r->d_payload;
                    fc.payload_base[2] = (uint64_t)(uintptr_t)(b ? b->payload : nullptr);
                }
            }
            uint64_t *d_goff = (uint64_t *)r->d_filter_tmp, *d_tmp = d_goff + r->cap_records + 1;
            ea::FilterCols *d_fc = (ea::FilterCols *)r->d_gather;  // the scratch column is free until the gathers
            RD_HIP(r, hipMemcpyAsync(d_fc, &fc, sizeof fc, hipMemcpyHostToDevice, r->stream));
            ea::filter_rows((const ea::FilterProgram *)r->d_filter_prog, d_fc, (const uint8_t *)r->d_filter_consts, k, d_goff,
                            d_tmp, (uint32_t *)r->d_row_map, r->stream);
            uint64_t n_sel = 0;
            RD_HIP(r, hipMemcpyAsync(&n_sel, d_goff + k, 8, hipMemcpyDeviceToHost, r->stream));
            RD_HIP(r, hipStreamSynchronize(r->stream));
            k = n_sel;
            row_map = (const uint32_t *)r->d_row_map;
        }
        *n_records_out = k;
        if (r->arrow_emit && !count_only) {
            // new_reader: the columns stay in HBM and become Arrow buffers there (exg_arrow_stream.cpp)
            ScanCtx ctx;
            ctx.d_input = d_input;
            ctx.h = h;
            ctx.n_records = k;
            ctx.res = res;
            ctx.h_seq_payload = b ? (const uint8_t *)b->payload : nullptr;
            double t_emit = now_s();
            if (k && (rc = r->arrow_emit(r, ctx))) return rc;
            TRACE("arrow emit", t_emit);
        } else if (k && !count_only) {
            TraceRange d2h_range("exg: columns -> host");
            if (!b) b = std::make_shared<Batch>();
            b->host.reserve(r->host_hint);
            b->file = gz_payload ? gz_payload : r->file;
            // the projection (exg_open_args.columns): every column was parsed and validated above, only the wanted ones travel.
            // A decoded input's bytes are what its strings point into: they travel when any string column does
            const bool any_strings = r->format == EXG_FMT_VCF ? (r->want_cols & 0x1DDull) != 0 : (r->want_cols & ((1ull << n_string_cols(r->format)) - 1)) != 0;
            if (gz_payload && any_strings && !gz_mirror) RD_HIP(r, hipMemcpyAsync(gz_payload->p, d_input, n, hipMemcpyDeviceToHost, r->stream));
            if (gz_mirror && gz_front) RD_HIP(r, hipMemcpyAsync(const_cast<uint8_t *>(h), d_input, gz_front, hipMemcpyDeviceToHost, r->stream));
            b->n_rows = k;
            const int ns = n_string_cols(r->format);
            // compact: the selected string columns' out-of-line bytes, closed up per column into ONE side buffer
            struct SideCol {
                uint64_t *d_goff = nullptr;
                uint64_t total = 0, off = 0;
            } side[9];
            struct SideScratch {  // pooled device scratch of this batch's side buffer
                int dev;
                hipStream_t st;
                std::vector<std::pair<void *, size_t>> blocks;
                void *take(size_t n) {
                    void *p = dev_pool()->take(dev, n);
                    if (p) blocks.emplace_back(p, n);
                    return p;
                }
                ~SideScratch() {
                    if (!blocks.empty()) (void)hipStreamSynchronize(st);  // (an early return: kernels may still read them)
                    for (auto &bl : blocks) dev_pool()->give(dev, bl.first, bl.second);
                }
            } side_scratch{r->device, r->stream, {}};
            uint8_t *h_side = nullptr, *d_side = nullptr;
            if (compact) {
                namespace ea = exg::arrow;
                uint64_t *d_tmp = (uint64_t *)side_scratch.take((ea::scan_tmp_entries(k) + 2) * 8);
                if (!d_tmp) return fail(r, EXG_E_HIP, "out of device memory");
                for (int c = 0; c < ns; c++) {
                    if (!r->want(c) || (r->format == EXG_FMT_VCF && c != 0 && c != 3) || (r->format == EXG_FMT_FASTA && c == 2)) continue;
                    if (!(side[c].d_goff = (uint64_t *)side_scratch.take((k + 2) * 8))) return fail(r, EXG_E_HIP, "out of device memory");
                    const ea::StrCol sc{(const exg_string_t *)r->d_cols[c], (const uint8_t *)d_input, (uint64_t)(uintptr_t)h};
                    ea::payload_goff_from_col(sc, row_map, k, side[c].d_goff, d_tmp, r->stream);
                    RD_HIP(r, hipMemcpyAsync(&side[c].total, side[c].d_goff + k, 8, hipMemcpyDeviceToHost, r->stream));
                }
                RD_HIP(r, hipStreamSynchronize(r->stream));
                uint64_t side_total = 0;
                for (int c = 0; c < ns; c++) side[c].off = side_total, side_total += (side[c].total + 15) & ~15ull;
                if (side_total) {
                    if (!(h_side = (uint8_t *)b->host.alloc(side_total + 64))) return fail(r, EXG_E_HIP, "out of pinned host memory");
                    const uint32_t big_cap = (uint32_t)(side_total / 8192 + 1);
                    uint32_t *d_big = (uint32_t *)side_scratch.take(4 * ((size_t)big_cap + 1));
                    if (!(d_side = (uint8_t *)side_scratch.take(side_total + 64)) || !d_big) return fail(r, EXG_E_HIP, "out of device memory");
                    for (int c = 0; c < ns; c++) {
                        if (!side[c].d_goff || !side[c].total) continue;
                        const ea::StrCol sc{(const exg_string_t *)r->d_cols[c], (const uint8_t *)d_input, (uint64_t)(uintptr_t)h};
                        ea::payload_copy_from_col(sc, row_map, k, side[c].d_goff, d_side + side[c].off, d_big, big_cap, r->stream);
                    }
                    RD_HIP(r, hipMemcpyAsync(h_side, d_side, side_total, hipMemcpyDeviceToHost, r->stream));
                }
            }
            // schema order (exg_schema_of): VCF exposes parsed POS / QUAL in place of their raw text
            b->n_cols = ns;
            const size_t vw = (size_t)((k + 63) / 64) * 8;
            // read_vcf: behind the flat columns come the nested ones' kernels (nested_emit: dozens of small launches that build — or,
            // for columns the projection leaves out, only validate — id / alt / filter / info / formats).  In one stream the
            // 200 MB of flat vectors (3.7 ms of the link) stood in front of them; on a stream of their own the copies run beside
            // them (chrom, pos, ref of a 2 GB file: 70.6 -> 49.4 ms = 42 GB/s, COUNT(*) 47 ms: A/B in one box, EXG_VCF_ONE_STREAM)
            hipStream_t cs = r->stream;
            // (whatever way this block is left once copies are on the columns' stream: they have landed — or the batch carries the
            // event that says when — before its pinned blocks can go back to the pool)
            struct ColDrain {
                hipStream_t cs = nullptr;
                ~ColDrain() {
                    if (cs) (void)hipStreamSynchronize(cs);
                }
            } col_drain;
            // (FASTA, round 6: its joined sequences are as many bytes as the batch itself — on the scan's stream they shared a copy
            // engine with the next batch's upload, which landed 5 ms behind them: 9.7 ms per 256 MiB batch where the two directions
            // should overlap)
            static const bool fastq_cols_on_scan_stream = getenv("EXG_FASTQ_ONE_STREAM") != nullptr;  // (A/B)
            if ((r->format == EXG_FMT_VCF || r->format == EXG_FMT_FASTA || (!r->src && !fastq_cols_on_scan_stream)) && !compact && !getenv("EXG_VCF_ONE_STREAM")) {
                if (!r->col_stream) {
                    RD_HIP(r, stream_pool()->take_d2h(r->device, &r->col_stream, /*calibrate=*/r->file && r->file->n >= (512ull << 20) && !r->mem_cap));
                    RD_HIP(r, hipEventCreateWithFlags(&r->col_ev, hipEventDisableTiming));
                    RD_HIP(r, hipEventCreateWithFlags(&r->flat_ev, hipEventDisableTiming));
                }
                RD_HIP(r, hipEventRecord(r->col_ev, r->stream));          // (the scan, the predicate's row map)
                RD_HIP(r, hipStreamWaitEvent(r->col_stream, r->col_ev, 0));
                cs = r->col_stream;
                col_drain.cs = cs;
            }
            // read_vcf hands its batch on while the vectors are still travelling (Batch::landed): the batch behind it — upload wait,
            // scan, the nested columns' kernels — is made beside them, and the link back to the host does not idle between batches
            static const bool eager_landing = getenv("EXG_VCF_EAGER_LANDING") != nullptr;
            r->lazy_landing = r->format == EXG_FMT_VCF && cs == r->col_stream && cs != r->stream && !eager_landing;
            const bool nested_vcf = r->format == EXG_FMT_VCF;  // id, alt, filter, info, formats: built by nested_emit below
            for (int c = 0; c < ns; c++) {
                if ((nested_vcf && (c == 2 || c == 4 || c >= 6)) || !r->want(c)) {
                    b->elem[c] = 0;
                    b->cols[c] = nullptr;
                    continue;
                }
                const void *src = r->d_cols[c];
                uint32_t es = 16;
                if (r->format == EXG_FMT_VCF && c == 1) src = r->d_pos, es = 8;
                if (r->format == EXG_FMT_VCF && c == 5) src = r->d_qual, es = 4;
                b->elem[c] = es;
                if (!(b->cols[c] = b->host.alloc(k * es))) return fail(r, EXG_E_HIP, "out of pinned host memory");
                if (compact && es == 16 && side[c].d_goff) {
                    // (gathers through the row map itself; in place without one: the column is the scan's scratch from here on)
                    void *dst = row_map ? r->d_gather : r->d_cols[c];
                    exg::arrow::repoint_strings((const exg_string_t *)r->d_cols[c], row_map, k, side[c].d_goff, (uint64_t)(uintptr_t)(h_side + side[c].off),
                                                (exg_string_t *)dst, r->stream);
                    src = dst;
                } else if (row_map) {
                    if (es == 16)
                        exg::arrow::gather_u128(src, row_map, k, r->d_gather, cs);
                    else if (es == 8)
                        exg::arrow::gather_u64((const uint64_t *)src, row_map, k, (uint64_t *)r->d_gather, cs);
                    else
                        exg::arrow::gather_u32((const uint32_t *)src, row_map, k, (uint32_t *)r->d_gather, cs);
                    src = r->d_gather;
                }
                RD_HIP(r, hipMemcpyAsync(b->cols[c], src, k * es, hipMemcpyDeviceToHost, compact && es == 16 && side[c].d_goff ? r->stream : cs));
                r->host_vector_bytes += k * es;
            }
            auto copy_validity = [&](int col, const void *d) -> int {
                if (!(b->validity[col] = b->host.alloc(vw))) return fail(r, EXG_E_HIP, "out of pinned host memory");
                if (row_map) {
                    exg::arrow::gather_bits((const uint64_t *)d, row_map, k, (uint64_t *)r->d_gather, cs);
                    d = r->d_gather;
                }
                RD_HIP(r, hipMemcpyAsync(b->validity[col], d, vw, hipMemcpyDeviceToHost, cs));
                r->host_vector_bytes += vw;
                return EXG_OK;
            };
            if (r->format == EXG_FMT_VCF) {
                if (r->want(5) && (rc = copy_validity(5, r->d_valid[0]))) return rc;
                if (r->lazy_landing) {
                    RD_HIP(r, hipEventRecord(r->flat_ev, cs));  // (the scan's columns are free again behind this)
                    r->flat_pending = true;
                }
                if (!r->nested_state && (rc = nested_prepare(r))) return rc;
                ScanCtx ctx;
                ctx.d_input = d_input;
                ctx.h = h;
                ctx.n_records = k;
                ctx.res = res;
                ctx.h_seq_payload = nullptr;
                uint64_t deliver = k;
                const double t_ne = now_s();
                trace_at("N nested begins", r->n_batches);
                if ((rc = nested_emit(r, ctx, b.get(), row_map, &deliver))) return rc;
                trace_at("N nested landed", r->n_batches);
                TRACE("nested columns (kernels + their way back)", t_ne);
                b->n_rows = deliver;
                if (r->lazy_landing) {
                    RD_HIP(r, hipEventCreateWithFlags(&b->landed, hipEventDisableTiming));
                    RD_HIP(r, hipEventRecord(b->landed, cs));
                    col_drain.cs = nullptr;
                }
            } else {
                if (r->want(1) && (rc = copy_validity(1, r->d_valid[0]))) return rc;
            }
            if (r->format == EXG_FMT_FASTA && res.payload_bytes && r->want(2)) {
                // by a kernel's stores, not by a copy engine: the next batch's upload (slices that each take whichever engine is free
                // when they are enqueued) ended up behind this copy on ITS engine every other batch and landed 5 ms late
                static const bool by_engine = getenv("EXG_FASTA_D2H_ENGINE") != nullptr;  // (A/B)
                if (by_engine || r->src || (((uintptr_t)b->payload | (uintptr_t)r->d_payload) & 15)) {  // (a decoded stream: no upload beside it, the engine is faster)
                    RD_HIP(r, hipMemcpyAsync(b->payload, r->d_payload, res.payload_bytes, hipMemcpyDeviceToHost, cs));
                } else if (int prc = exg::stream_to_host(b->payload, r->d_payload, res.payload_bytes, cs)) {
                    return fail(r, prc, exg_last_error_message());
                }
                r->host_vector_bytes += res.payload_bytes;  // (the joined sequences: the strings' payload is made on the device)
            }
            const double t_cols = now_s();
            RD_HIP(r, hipStreamSynchronize(r->stream));
            if (r->col_stream && !b->landed) RD_HIP(r, hipStreamSynchronize(r->col_stream));
            col_drain.cs = nullptr;
            TRACE("wait(columns -> host)", t_cols);
            const double t_mir = now_s();
            if (gz_mirror) RD_HIP(r, hipEventSynchronize(gz_mirror->ev));  // the segment's own bytes have arrived
            if (gz_mirror) TRACE("wait(host mirror of the segment)", t_mir);
            trace_at("C batch-done", r->n_batches);
            r->host_hint = b->host.total + b->host.total / 8 + (1u << 20);
            b->seq = r->batch_seq++;
            r->batch = b;
        }
        r->shard_first = false;
        if (res.error_code || eof) {
            r->file_done = true;
        } else {
            r->file_pos += res.consumed_bytes - lead;
            if (range_end) r->file_done = true;  // what is left belongs to the next shard, whose halo reaches back to where that record begins
        }
        TRACE("batch (h2d+scan+d2h)", t_batch);
        if (trace_on())
            fprintf(stderr, "[exg] device bytes held: %.2f MiB now, %.2f MiB at the peak (batch of %llu bytes, %llu rows)\n", r->meter.cur.load() / 1048576.0,
                    r->meter.peak.load() / 1048576.0, (unsigned long long)n, (unsigned long long)k);
        return EXG_OK;
    }
}

}  // namespace exg_rd
