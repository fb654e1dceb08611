// Sanitizer driver of the host-only pieces of libexon_gpu (tests/test_host_asan.py builds it with
// -fsanitize=address,undefined and runs it): the gzip member index (exg_gzip.cpp), the BGZF member walk of the streaming
// reader (exg_rd_bgzf.cpp), the zstd frame / block walk (exg_zstd_index.cpp), the VCF header parser (exg_vcf_header.cpp)
// and the `filters` parser (exg_filter.hpp) on valid inputs, on truncations of them and on random mutations — every byte
// these parsers read comes from a user's file or query text.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>

#include <random>
#include <string>
#include <vector>

#include "exg_block_pool.hpp"
#include "exg_filter.hpp"
#include "exg_rd_fanout.hpp"
#include "exg_rd_internal.hpp"
#include "exg_vcf_header.hpp"
#include "exg_xxh64.hpp"
#include <sys/mman.h>
#include "exg_map_guard.hpp"
#include "exg_zstd.hpp"

namespace exg {
void set_error(const char *, ...) {}
}  // namespace exg

static int g_devices = 1;
extern "C" int exg_device_count(void) { return g_devices; }  // (the planner asks the library: here the test says)

static std::vector<uint8_t> read_file(const char *path) {
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    setenv("EXG_ZSTD_INDEX_PREFETCH_MIN", "0", 1);  // (exg_zstd_index.cpp: the fetching passes also on these small inputs)
    std::mt19937_64 rng(99);
    long runs = 0;
    // argv: files (gzip / zstd streams written by the test)
    for (int a = 1; a < argc; a++) {
        const std::vector<uint8_t> base = read_file(argv[a]);
        if (base.empty()) return 2;
        for (int trial = 0; trial < 400; trial++) {
            std::vector<uint8_t> d = base;
            if (trial % 4 == 1) d.resize(rng() % (d.size() + 1));
            if (trial % 4 >= 2)
                for (int k = 0; k < 1 + (int)(rng() % 4); k++) d[rng() % d.size()] ^= (uint8_t)(1u << (rng() % 8));
            // the parsers get exactly d.size() readable bytes (a heap block of that size: ASan sees any over-read)
            uint8_t *p = (uint8_t *)malloc(d.size() ? d.size() : 1);
            memcpy(p, d.data(), d.size());
            {
                std::vector<exg_inflate_member> members(d.size() / 18 + 8);
                uint64_t k = 0, total = 0;
                int open_ended = 0;
                (void)exg_gzip_index(p, d.size(), 0, members.data(), members.size(), &k, &total, &open_ended);
            }
            {
                // the BGZF member walk (exg_rd_bgzf.cpp): a member at every offset a header might be, the search for the first
                // member behind an offset, the index of the whole buffer by several threads
                exg_rd::Peek pk(p, -1, d.size());
                for (uint64_t at = 0; at < d.size() && at < 4096; at += 1 + rng() % 97) {
                    exg_inflate_member m;
                    uint32_t crc = 0;
                    const uint64_t nx = exg_rd::bgzf_member_at(pk, at, &m, &crc);
                    if (nx && (nx > d.size() || m.comp_off + m.comp_size > d.size())) return 7;  // an accepted member lies inside the input
                }
                const uint64_t found = exg_rd::bgzf_find(p, -1, d.size(), rng() % (d.size() + 1));
                if (found > d.size()) return 7;
                std::vector<exg_inflate_member> mm(d.size() / 18 + 8);
                std::vector<uint32_t> crcs;
                uint64_t k = 0, total = 0;
                if (exg_rd::bgzf_parallel_index(p, -1, d.size(), mm.data(), mm.size(), &k, &total, &crcs))
                    for (uint64_t i = 0; i < k; i++)
                        if (mm[i].comp_off + mm[i].comp_size > d.size()) return 7;
            }
            {
                exg::zst::Index idx;
                const bool whole = exg::zst::build_index(p, d.size(), idx);
                // (a damaged stream: what lies in front of the damage stays usable — the frames cover the blocks exactly)
                if (!whole && exg::zst::salvage_index(idx)) {
                    size_t nb = 0;
                    for (const auto &f : idx.frames) {
                        if (f.first_block != nb || !f.n_blocks) return 3;
                        nb += f.n_blocks;
                    }
                    if (nb != idx.blocks.size()) return 3;
                }
                for (const auto &b : idx.blocks)
                    if (b.src_off + (b.type == 1 ? 1 : b.src_size) > d.size()) return 3;  // a block the walk accepted must lie inside the input
                // the same walk over a file (small preads, what the reader runs; with the two fetching passes a big file gets in front
                // of it: EXG_ZSTD_INDEX_PREFETCH_MIN=0, set by main): the same answer, block for block
                if (runs % 8 == 0) {
                    char name[] = "/tmp/exg_asan_zst_XXXXXX";
                    const int fd = mkstemp(name);
                    if (fd < 0) return 9;
                    unlink(name);
                    if (!d.empty() && write(fd, p, d.size()) != (ssize_t)d.size()) return 9;
                    exg::zst::Index fi;
                    const bool whole_f = exg::zst::build_index_fd(fd, d.size(), fi);
                    close(fd);
                    exg::zst::Index mi;
                    (void)exg::zst::build_index(p, d.size(), mi);
                    if (whole_f != whole || fi.error != mi.error || fi.blocks.size() != mi.blocks.size() || fi.frames.size() != mi.frames.size()) return 9;
                    if (!fi.blocks.empty() && memcmp(fi.blocks.data(), mi.blocks.data(), fi.blocks.size() * sizeof fi.blocks[0])) return 9;
                    if (!fi.frames.empty() && memcmp(fi.frames.data(), mi.frames.data(), fi.frames.size() * sizeof fi.frames[0])) return 9;
                }
                // the PREFIX walk a reader begins its first round on (round 5: build_index_prefix_fd) — whatever the stop estimate, what
                // it returns is the whole walk's first blocks, field for field; it stops inside a frame (the frame it cuts is in the
                // index with the blocks seen so far and its header's real fields) or not at all (then it IS the whole index)
                if (runs % 8 == 2 && whole) {
                    char name[] = "/tmp/exg_asan_zst_XXXXXX";
                    const int fd = mkstemp(name);
                    if (fd < 0) return 9;
                    unlink(name);
                    if (!d.empty() && write(fd, p, d.size()) != (ssize_t)d.size()) return 9;
                    for (const uint64_t stop : {(uint64_t)1, (uint64_t)(128u << 10), (uint64_t)(1u << 20), (uint64_t)(64u << 20)}) {
                        exg::zst::Index pi;
                        bool stopped = false;
                        if (!exg::zst::build_index_prefix_fd(fd, d.size(), stop, pi, &stopped)) return 12;
                        if (pi.blocks.size() > idx.blocks.size() || pi.frames.size() > idx.frames.size()) return 12;
                        if (!pi.blocks.empty() && memcmp(pi.blocks.data(), idx.blocks.data(), pi.blocks.size() * sizeof pi.blocks[0])) return 12;
                        if (!stopped) {
                            if (pi.blocks.size() != idx.blocks.size() || pi.frames.size() != idx.frames.size()) return 12;
                            if (!pi.frames.empty() && memcmp(pi.frames.data(), idx.frames.data(), pi.frames.size() * sizeof pi.frames[0])) return 12;
                        } else {
                            if (pi.frames.empty() || pi.blocks.empty()) return 12;
                            const auto &cut = pi.frames.back(), &full = idx.frames[pi.frames.size() - 1];
                            if (cut.first_block != full.first_block || cut.src_off != full.src_off || cut.window != full.window ||
                                cut.has_checksum != full.has_checksum || cut.content_size != full.content_size)
                                return 12;
                            if (cut.first_block + cut.n_blocks != pi.blocks.size() || cut.n_blocks >= full.n_blocks) return 12;  // open: blocks follow
                            if (pi.frames.size() > 1 && memcmp(pi.frames.data(), idx.frames.data(), (pi.frames.size() - 1) * sizeof pi.frames[0])) return 12;
                        }
                    }
                    close(fd);
                }
                // a file that SHRANK behind its fstat (or an I/O error in the middle): the walk is told a size the file no longer has.
                // It stops at the first read that does not come — no block behind the failure enters the index (a zero block header
                // is a valid empty raw block: a walk that went on would push one block per three missing bytes) — and what was
                // read in front of it is exactly the memory walk's prefix.
                if (runs % 8 == 4 && whole && d.size() > 64) {
                    char name[] = "/tmp/exg_asan_zst_XXXXXX";
                    const int fd = mkstemp(name);
                    if (fd < 0) return 9;
                    unlink(name);
                    const size_t keep = 16 + rng() % (d.size() - 16);
                    if (write(fd, p, keep) != (ssize_t)keep) return 9;
                    exg::zst::Index fi;
                    const bool whole_f = exg::zst::build_index_fd(fd, d.size() + (64u << 20), fi);  // (64 MiB of bytes that are not there)
                    close(fd);
                    if (whole_f) return 10;
                    if (fi.error.find("short read") == std::string::npos) return 10;
                    if (fi.blocks.size() > idx.blocks.size() + 1) return 10;
                    for (size_t i = 0; i < fi.blocks.size(); i++) {
                        const auto &b = fi.blocks[i];
                        if (b.src_off > keep) return 10;   // its header was read: it begins inside what is there
                        if (i < idx.blocks.size() && memcmp(&fi.blocks[i], &idx.blocks[i], sizeof b)) return 10;
                    }
                    (void)exg::zst::salvage_index(fi);
                    size_t nb = 0;
                    for (const auto &f : fi.frames) {
                        if (f.first_block != nb || !f.n_blocks) return 10;
                        nb += f.n_blocks;
                    }
                    if (nb != fi.blocks.size()) return 10;
                }
            }
            free(p);
            runs++;
        }
    }
    // the VCF header -> typed INFO / FORMAT keys (exg_vcf_header.cpp) on a well-formed header and on mangled ones
    {
        const std::string hdr = "##fileformat=VCFv4.2\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d, \\\"q\\\"\">\n"
                                "##INFO=<ID=AF,Number=A,Type=Float,Description=\"a\">\n##INFO=<ID=DB,Number=0,Type=Flag>\r\n"
                                "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n##FORMAT=<ID=AD,Number=R,Type=Integer>\n#CHROM\tPOS\n";
        if (exg_rd::explain_vcf_header(hdr.data(), hdr.size()) != "INFO DP:i AF:[f] DB:b | FORMAT GT:u AD:[i]") return 8;
        for (int trial = 0; trial < 2000; trial++) {
            std::string t = hdr;
            for (int k = 0; k < 1 + (int)(rng() % 4); k++) {
                const size_t at = rng() % (t.size() + 1);
                switch (rng() % 4) {
                    case 0: t.insert(at, 1, (char)(rng() % 256)); break;
                    case 1: if (!t.empty()) t.erase(at % t.size(), 1 + rng() % 5); break;
                    case 2: t.resize(at); break;
                    default: if (!t.empty()) t[at % t.size()] = "<>=,\"\\#\n"[rng() % 8]; break;
                }
            }
            char *q = (char *)malloc(t.size() ? t.size() : 1);  // exactly t.size() readable bytes
            memcpy(q, t.data(), t.size());
            std::vector<exg_rd::KeyDef> info, format;
            exg_rd::parse_vcf_header(q, t.size(), &info, &format);
            free(q);
            runs++;
        }
    }
    // the filter grammar on well-formed and mangled predicates
    const std::vector<exg_rd::FilterColumn> cols = {{"chrom", 'u'}, {"pos", 'l'}, {"id", 'x'}, {"qual", 'f'}};
    const char *seeds[] = {"chrom='7' AND pos>=3000 AND pos<9000", "qual IS NULL OR qual>900.5", "\"chrom\"!='a''b' AND (pos<>1 OR qual<=1e3)",
                           "id='x'", "pos=", "((((chrom='1'", "qual>'abc'", "chrom IS NOT NULL AND chrom IS NULL OR pos>1 OR pos>2 OR pos>3"};
    for (const char *sd : seeds)
        for (int trial = 0; trial < 300; trial++) {
            std::string t = sd;
            if (trial) {
                for (int k = 0; k < 1 + (int)(rng() % 3); k++) {
                    const size_t at = rng() % (t.size() + 1);
                    switch (rng() % 3) {
                        case 0: t.insert(at, 1, (char)(32 + rng() % 95)); break;
                        case 1: if (!t.empty()) t.erase(at % t.size(), 1); break;
                        default: if (!t.empty()) t[at % t.size()] = (char)(rng() % 256); break;
                    }
                }
            }
            exg_rd::FilterParser fp(t, cols);
            (void)fp.parse();
            runs++;
        }
    // a long conjunction must be refused, not overflow the program
    std::string big = "pos>0";
    for (int i = 0; i < 100; i++) big += " AND pos>" + std::to_string(i);
    exg_rd::FilterParser fp(big, cols);
    if (fp.parse()) return 4;
    // host XXH64 (the checksum of big zstd frames): the specification's known answers, and any way of cutting the input
    // into updates gives the one-shot digest
    {
        auto one = [](const uint8_t *p, size_t n) {
            exg::Xxh64 h;
            h.update(p, n);
            return h.digest();
        };
        if (one((const uint8_t *)"", 0) != 0xEF46DB3751D8E999ull) return 5;
        if (one((const uint8_t *)"a", 1) != 0xD24EC4F1A98C6E5Bull) return 5;
        if (one((const uint8_t *)"abc", 3) != 0x44BC2CF5AD770999ull) return 5;
        std::vector<uint8_t> buf(5000);
        for (auto &b : buf) b = (uint8_t)rng();
        for (int trial = 0; trial < 300; trial++) {
            const size_t n = rng() % (buf.size() + 1);
            exg::Xxh64 h;
            for (size_t at = 0; at < n;) {
                const size_t want = 1 + rng() % 97, k = want < n - at ? want : n - at;
                h.update(buf.data() + at, k);
                at += k;
            }
            if (h.digest() != one(buf.data(), n)) return 6;
            runs++;
        }
    }
    // the shard planner (exg_rd_plan.cpp): every input file of this run, a directory of them, paths that do not exist, forced
    // shard counts, few and many devices — the plan never names more shards than the caller has room for, never a device
    // that does not exist, and every stripe of a fan-out points at a file of the input
    {
        const char *fmts[] = {"fastq", ""};
        std::vector<std::string> inputs;
        for (int i = 1; i < argc; i++) inputs.push_back(argv[i]);
        if (argc > 1) {
            std::string dir = argv[1];
            const size_t slash = dir.rfind('/');
            inputs.push_back(slash == std::string::npos ? "." : dir.substr(0, slash));  // the directory the test wrote them to
        }
        inputs.push_back("/nonexistent/x.fastq");
        inputs.push_back("");
        const char *forced[] = {nullptr, "7", "100000", "x"};
        for (const std::string &in : inputs)
            for (const char *f : forced)
                for (int n_dev : {1, 8}) {
                    g_devices = n_dev;
                    if (f) setenv("EXON_GPU_SHARDS", f, 1);
                    else unsetenv("EXON_GPU_SHARDS");
                    for (const char *fmt : fmts)
                        for (const char *comp : {(const char *)nullptr, "zstd", "??"}) {
                            exg_open_args a;
                            memset(&a, 0, sizeof a);
                            a.path = in.c_str();
                            a.file_format = fmt;
                            a.compression = comp;
                            a.device = n_dev - 1;
                            int devices[16];
                            uint32_t n = 0;
                            const uint32_t cap = 1 + (uint32_t)(rng() % 16);
                            if (exg_plan_shards(&a, &n, devices, cap) != EXG_OK) return 9;
                            if (n < 1 || n > cap) return 9;
                            for (uint32_t i = 0; i < n; i++)
                                if (devices[i] < 0 || devices[i] >= n_dev) return 9;
                            std::vector<std::string> files;
                            std::string err;
                            if (exg_rd::list_path(in, &files, &err) == EXG_OK) {
                                std::vector<exg_rd::Stripe> stripes;
                                unsigned workers = 0;
                                if (exg_rd::plan_stripes(files, exg_rd::compression_of(&a), &a, &stripes, &workers) != EXG_OK) return 9;
                                if (stripes.size() < files.size() || workers < 1) return 9;
                                for (const auto &st : stripes) {
                                    if (st.device < 0 || st.device >= n_dev || st.shard_index >= st.shard_count) return 9;
                                    bool known = false;
                                    for (const auto &fl : files) known = known || fl == st.path;
                                    if (!known) return 9;
                                }
                            }
                            runs++;
                        }
                }
        unsetenv("EXON_GPU_SHARDS");
        // the replacement scan of the reference's FFI on names of every shape
        const char *uris[] = {"a.fasta", "a.fa.gz", "x/y.fastq.zst", "a.vcf.bz2", "noext", "", ".", "..gz", "a.FASTQ", "a.fq", "s3://b/k.vcf.gz", "a.gz", "a.b.c.d"};
        for (const char *u : uris) {
            ReplacementScanResult rs = replacement_scan(u);
            (void)rs;
            runs++;
        }
        (void)replacement_scan(nullptr);
    }
    // the pinned-block pool (exg_block_pool.hpp) over malloc / free with the "device's node" said by the test: a block goes back
    // only to a taker on the node it was made on, the cap follows the devices that have taken blocks, every block is released exactly once
    {
        static int s_node = 0, s_dev = 0;
        static long s_live = 0;
        exg_rd::BlockPool::Hooks h;
        h.alloc = [](size_t n) -> void * { s_live++; return malloc(n > 4096 ? 4096 : n); };  // (the bookkeeping is what runs here)
        h.release = [](void *p) { s_live--; free(p); };
        h.current_node = [] { return s_node; };
        h.current_device = [] { return s_dev; };
        {
            exg_rd::BlockPool pool(h);
            if (pool.cap() != exg_rd::BlockPool::kPerDevice) return 10;   // nobody has taken a block: one device's worth
            size_t sz = 100u << 20;
            s_node = 0;
            char *a = pool.take(&sz);
            if (!a || sz != (128u << 20)) return 10;
            if (pool.cap() != exg_rd::BlockPool::kPerDevice || pool.devices_in_use() != 1) return 10;
            pool.give(a, sz);
            s_node = 1;                       // a reader on the other socket: must NOT get node 0's block
            size_t sz1 = 100u << 20;
            char *b = pool.take(&sz1);
            if (!b || b == a || pool.free_on_node(0) != 1) return 10;
            pool.give(b, sz1);
            s_node = 0;                       // back on node 0: the first block comes round
            size_t sz2 = 90u << 20;
            char *c = pool.take(&sz2);
            if (c != a || sz2 != (128u << 20) || pool.n_reused != 1) return 10;
            // a block given back from a thread whose device sits on another node keeps ITS node
            s_node = 1;
            pool.give(c, sz2);
            if (pool.free_on_node(0) != 1 || pool.free_on_node(1) != 1) return 10;
            // random traffic from "eight devices on two nodes"
            std::vector<std::pair<char *, size_t>> held;
            for (int it = 0; it < 4000; it++) {
                s_dev = (int)(rng() % 8);
                s_node = s_dev / 4;
                if (held.empty() || rng() % 2) {
                    size_t want = (size_t)(1 + rng() % 300) << 20;
                    char *p = pool.take(&want);
                    if (!p || want % (32u << 20)) return 10;
                    held.emplace_back(p, want);
                } else {
                    const size_t k = rng() % held.size();
                    pool.give(held[k].first, held[k].second);
                    held.erase(held.begin() + (long)k);
                }
                if (pool.pooled() > pool.cap()) return 10;
                runs++;
            }
            for (auto &x : held) pool.give(x.first, x.second);
            if (pool.devices_in_use() != 8 || pool.cap() != 8 * exg_rd::BlockPool::kPerDevice) return 10;  // the cap followed the devices in use
        }
        s_dev = 0;
        if (s_live != 0) return 10;           // the pool's destructor released what it still held; nothing twice (ASan), nothing lost
        {   // a block that does not fit under the cap takes the place of the ones that have lain free the longest (round 5)
            exg_rd::BlockPool pool(h, /*cap_override=*/(size_t)(160u << 20));
            s_node = 0;
            size_t s1 = 64u << 20, s2 = 64u << 20, s3 = 96u << 20;
            char *a = pool.take(&s1), *b = pool.take(&s2), *c = pool.take(&s3);
            pool.give(a, s1);
            pool.give(b, s2);
            if (pool.pooled() != (128u << 20)) return 11;
            pool.give(c, s3);  // 128 + 96 > 160: `a` (the oldest) goes, b and c stay
            if (pool.pooled() != (160u << 20) || pool.free_on_node(0) != 2) return 11;
            size_t s4 = 96u << 20;
            if (pool.take(&s4) != c) return 11;
            size_t s5 = 64u << 20;
            if (pool.take(&s5) != b) return 11;
            pool.give(c, s4);
            pool.give(b, s5);
            size_t big = 192u << 20;  // larger than the whole cap: never pooled
            char *d = pool.take(&big);
            pool.give(d, big);
            if (pool.pooled() != (160u << 20)) return 11;
        }
        if (s_live != 0) return 11;
        // a small cap: blocks beyond it are released at once
        {
            exg_rd::BlockPool pool(h, 64u << 20);
            size_t s1 = 1, s2 = 1, s3 = 1;
            char *a = pool.take(&s1), *b = pool.take(&s2), *c = pool.take(&s3);
            pool.give(a, s1), pool.give(b, s2), pool.give(c, s3);
            if (pool.pooled() != (64u << 20) || s_live != 2) return 10;
        }
        if (s_live != 0) return 10;
    }
    // a mapping whose file is truncated under it (exg_map_guard.cpp): the read gives zeros and marks the slot, nothing dies; a
    // mapping that is not registered is not touched (the guard hands the signal on)
    {
        char name[] = "/tmp/exg_asan_map_XXXXXX";
        const int fd = mkstemp(name);
        if (fd < 0) return 12;
        const size_t page = (size_t)sysconf(_SC_PAGESIZE), n = 64 * page;
        std::vector<char> fill(n, 'x');
        if (write(fd, fill.data(), n) != (ssize_t)n) return 12;
        char *m = (char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return 12;
        const int slot = exg_rd::MapGuard::add(m, n);
        if (slot < 0 || exg_rd::MapGuard::hit(slot)) return 12;
        volatile char c = m[10 * page];
        if (c != 'x' || exg_rd::MapGuard::hit(slot)) return 12;
        if (ftruncate(fd, (off_t)(8 * page)) != 0) return 12;
        c = m[3 * page];                       // still there
        if (c != 'x' || exg_rd::MapGuard::hit(slot)) return 12;
        c = m[40 * page + 17];                 // gone: SIGBUS -> a zero page, the slot is marked
        if (c != 0 || !exg_rd::MapGuard::hit(slot) || exg_rd::MapGuard::patched() != 1) return 12;
        c = m[40 * page + 99];                 // the patched page: no second fault
        if (c != 0 || exg_rd::MapGuard::patched() != 1) return 12;
        c = m[63 * page];
        if (c != 0 || exg_rd::MapGuard::patched() != 2) return 12;
        exg_rd::MapGuard::remove(slot);
        if (exg_rd::MapGuard::hit(slot)) { /* (a freed slot reports what it last saw until it is reused: harmless) */ }
        munmap(m, n);
        // slots are reused, and a full table refuses (-1) instead of overwriting
        std::vector<int> slots;
        for (int i = 0; i < 4200; i++) slots.push_back(exg_rd::MapGuard::add(fill.data(), 16));
        int ok = 0;
        for (int sl : slots) ok += sl >= 0;
        if (ok != 4096) return 12;
        for (int sl : slots) exg_rd::MapGuard::remove(sl);
        if (exg_rd::MapGuard::add(fill.data(), 16) < 0) return 12;
        close(fd);
        unlink(name);
        runs++;
    }
    // the fan-out's run-ahead (exg_rd_fanout.cpp): a worker never holds more than `depth` batches the consumer has not taken,
    // however short its stripes are (bounded per stripe, 40 stripes of one batch each would all be produced at once)
    {
        struct CountingSub : exg_rd::FanSub {
            int left;
            explicit CountingSub(int n) : left(n) {}
            int next(exg_rd::FanItem *out, std::string *) override {
                if (left-- > 0) out->batch = std::make_shared<int>(left), out->rows = 1;
                return EXG_OK;
            }
            int count(uint64_t *rows, std::string *) override { *rows = (uint64_t)left; return EXG_OK; }
            void stats(uint64_t *now, uint64_t *peak, uint64_t *nb, uint64_t *ns) override { *now = 10, *peak = 20, *nb = 1, *ns = 0; }
        };
        for (int per_stripe : {1, 2, 5}) {
            std::vector<exg_rd::Stripe> stripes(40);
            exg_rd::FanOpen open = [per_stripe](const exg_rd::Stripe &, std::unique_ptr<exg_rd::FanSub> *sub, std::string *) {
                sub->reset(new CountingSub(per_stripe));
                return EXG_OK;
            };
            exg_rd::FanOut fan(stripes, 4, open, 2);
            uint64_t rows = 0;
            for (;;) {
                exg_rd::FanItem it;
                std::string err;
                if (fan.next(&it, &err)) return 11;
                if (!it.batch) break;
                rows += it.rows;
                if (rows == 3) {  // let the workers run as far as they may
                    struct timespec ts = {0, 20000000};
                    nanosleep(&ts, nullptr);
                    (void)fan.stats();
                }
            }
            if (rows != 40u * (uint64_t)per_stripe || fan.max_outstanding() > 2) return 11;
            if (fan.stats().device_batches != 40) return 11;
            runs++;
        }
    }
    printf("%ld runs\n", runs);
    return 0;
}
