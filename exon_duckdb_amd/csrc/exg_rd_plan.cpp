// exg_rd_plan.cpp — host only: which decoder an input needs (rust/src/arrow_reader.rs:60-91), how many byte-range shards
// a scan is worth and where they run (in-process form), and the reference's `replacement_scan`
// (rust/src/arrow_reader.rs:173-197).
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <dirent.h>
#include <errno.h>

#include <algorithm>

#include "exg_rd_fanout.hpp"
#include "exg_rd_internal.hpp"
#include "exg_zstd.hpp"

namespace exg_rd {

// DataFusion 28 FileCompressionType::from_str as used at rust/src/arrow_reader.rs:87-88
bool parse_compression(const std::string &s, Compression *out) {
    std::string u;
    for (char ch : s) u.push_back((char)toupper((unsigned char)ch));
    if (u == "GZIP" || u == "GZ") return *out = kGzip, true;
    if (u == "ZSTD" || u == "ZST") return *out = kZstd, true;
    if (u == "BZIP2" || u == "BZ2") return *out = kBzip2, true;
    if (u == "XZ") return *out = kXz, true;
    if (u.empty()) return *out = kNone, true;
    return false;
}

// compression: NULL => extension sniffing (arrow_reader.rs:60-75); unknown string => uncompressed (:87-88)
Compression compression_of(const exg_open_args *args) {
    Compression c = kNone;
    if (!args->compression) {
        const std::string path = args->path;
        size_t dot = path.rfind('.');
        std::string ext = dot == std::string::npos ? path : path.substr(dot + 1);
        c = ext == "gz" ? kGzip : ext == "zst" ? kZstd : kNone;
    } else if (!parse_compression(args->compression, &c)) {
        c = kNone;
    }
    return c;
}

}  // namespace exg_rd

using namespace exg_rd;

// How many byte-range shards a scan of this input is worth and where they run (the table function's init_global asks:
// MaxThreads() = *n_shards, init_local i opens shard i on devices[i]).  One shard per visible device when the input can
// be sharded — text FASTQ / VCF / FASTA, or BGZF FASTQ / VCF (members carry their size) — and holds at least 256 MiB per
// shard; otherwise one.  EXON_GPU_SHARDS=n forces n shards (tests: several shards on one device).
// can a file of this format / compression be read as byte-range shards?  (text; BGZF — FEXTRA with a 'BC' subfield in the
// first member — by members; zstd with several frames by frames)
static bool file_is_shardable(const std::string &f, const std::string &fmt_lower, Compression comp) {
    (void)fmt_lower;
    if (comp == kNone) return true;
    if (comp == kZstd) {
        // by frames: worth it when the file has several (pzstd, the seekable format; the zstd CLI writes one) — the walk over
        // the frame / block headers reads a few bytes per block (exg_zstd_index.cpp)
        int fd = open(f.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        bool many = false;
        if (fstat(fd, &st) == 0 && st.st_size > 0) {
            exg::zst::Index idx;
            many = exg::zst::build_index_fd(fd, (uint64_t)st.st_size, idx) && idx.frames.size() > 1;
        }
        close(fd);
        return many;
    }
    if (comp != kGzip) return false;
    uint8_t h[18] = {0};
    FILE *fp = fopen(f.c_str(), "rb");
    const size_t got = fp ? fread(h, 1, sizeof h, fp) : 0;
    if (fp) fclose(fp);
    return got == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
}

namespace exg_rd {
// the files of an input path: itself, or — a directory, which the reference lists (test_fasta_scan.test:55-59,
// test_fastq_scan.test:65-68) — its regular files in name order
int list_path(const std::string &path, std::vector<std::string> *files, std::string *err) {
    struct stat st;
    if (path.empty() || stat(path.c_str(), &st) != 0) {
        *err = "could not register table: cannot open '" + path + "': " + strerror(errno);
        return EXG_E_IO;
    }
    if (S_ISDIR(st.st_mode)) {
        DIR *d = opendir(path.c_str());
        if (!d) {
            *err = "cannot list '" + path + "'";
            return EXG_E_IO;
        }
        const size_t first = files->size();
        while (dirent *e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            std::string p = path + (path.back() == '/' ? "" : "/") + e->d_name;
            struct stat s2;
            if (stat(p.c_str(), &s2) == 0 && S_ISREG(s2.st_mode)) files->push_back(p);
        }
        closedir(d);
        std::sort(files->begin() + (long)first, files->end());
    } else {
        files->push_back(path);
    }
    return EXG_OK;
}

// The stripes a reader with shard_count = 0 fans out over (exg_rd_fanout.hpp): every file that can be sharded is cut into
// stripes of about EXG_FANOUT_STRIPE_MB (1024) MiB, a multiple of the device count of them, stripe s on device s mod N; a
// file that cannot be sharded is one stripe.  One device and nothing forced: one stripe per file — the caller then reads
// the input itself.  EXON_GPU_SHARDS = n forces n stripes per shardable file (tests: several stripes on one device).
int plan_stripes(const std::vector<std::string> &files, Compression compression, const exg_open_args *args, std::vector<Stripe> *out, unsigned *n_workers) {
    out->clear();
    const int n_dev = exg_device_count();
    if (n_dev < 1) return EXG_E_NO_DEVICE;
    std::string fmt = args->file_format;
    for (char &ch : fmt) ch = (char)tolower((unsigned char)ch);
    const char *forced = getenv("EXON_GPU_SHARDS");
    const uint64_t stripe_bytes = (getenv("EXG_FANOUT_STRIPE_MB") ? std::max<uint64_t>(1, strtoull(getenv("EXG_FANOUT_STRIPE_MB"), nullptr, 10)) : 1024) << 20;
    uint32_t next_dev = 0;
    for (const std::string &f : files) {
        struct stat st;
        uint64_t bytes = stat(f.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0;
        uint32_t n = 1;
        if (file_is_shardable(f, fmt, compression)) {
            if (forced) n = (uint32_t)std::max(1, atoi(forced));
            else if (n_dev > 1 && bytes >= (512ull << 20)) {
                const uint64_t per_round = stripe_bytes * (uint64_t)n_dev;
                n = (uint32_t)std::min<uint64_t>((bytes + per_round - 1) / per_round * (uint64_t)n_dev, 1u << 20);
            }
        }
        for (uint32_t i = 0; i < n; i++) {
            Stripe s;
            s.path = f;
            s.shard_index = i;
            s.shard_count = n;
            s.device = (int)(next_dev++ % (uint32_t)n_dev);
            out->push_back(s);
        }
    }
    *n_workers = getenv("EXG_FANOUT_WORKERS") ? (unsigned)std::max(1, atoi(getenv("EXG_FANOUT_WORKERS"))) : (unsigned)n_dev;
    return EXG_OK;
}
}  // namespace exg_rd

extern "C" int exg_plan_shards(const exg_open_args *args, uint32_t *n_shards, int *devices, uint32_t devices_cap) {
    if (!args || !args->path || !args->file_format || !n_shards || !devices || !devices_cap) {
        exg::set_error("exg_plan_shards: null argument");
        return EXG_E_INVALID_ARG;
    }
    *n_shards = 1;
    devices[0] = args->device;
    const int n_dev = exg_device_count();
    if (n_dev < 1) return EXG_E_NO_DEVICE;
    std::string fmt = args->file_format;
    for (char &ch : fmt) ch = (char)tolower((unsigned char)ch);
    const Compression comp = compression_of(args);
    std::vector<std::string> files;
    std::string list_err;
    if (list_path(args->path, &files, &list_err) != EXG_OK) return EXG_OK;  // the open will report it
    uint64_t bytes = 0;
    bool shardable = true;
    for (const std::string &f : files) {
        struct stat st;
        if (stat(f.c_str(), &st) != 0) continue;
        bytes += (uint64_t)st.st_size;
        shardable = shardable && file_is_shardable(f, fmt, comp);
    }
    uint32_t want = 1;
    if (const char *e = getenv("EXON_GPU_SHARDS")) {
        want = (uint32_t)std::max(1, atoi(e));
    } else {
        static const uint64_t per_shard = 256ull << 20;
        want = (uint32_t)std::min<uint64_t>((uint64_t)n_dev, std::max<uint64_t>(1, bytes / per_shard));
    }
    if (!shardable) want = 1;
    want = std::min<uint32_t>(want, devices_cap);
    *n_shards = want;
    for (uint32_t i = 0; i < want; i++) devices[i] = want == 1 ? args->device : (int)(i % (uint32_t)n_dev);
    return EXG_OK;
}

// exon/include/rust.hpp:48, rust/src/arrow_reader.rs:173-197 — same symbol, same result struct: the file type named by
// the last extension, skipping one compression extension; NULL when it is not one of the path's formats
extern "C" ReplacementScanResult replacement_scan(const char *uri) {
    ReplacementScanResult res;
    res.file_type = nullptr;
    if (!uri) return res;
    std::string lower = uri;
    for (char &c : lower) c = (char)tolower((unsigned char)c);
    auto ext_of = [](const std::string &s, size_t end) {
        size_t dot = s.rfind('.', end == std::string::npos ? end : end - 1);
        return dot == std::string::npos ? std::make_pair(s.substr(0, end), (size_t)0)
                                        : std::make_pair(s.substr(dot + 1, (end == std::string::npos ? s.size() : end) - dot - 1), dot);
    };
    auto e1 = ext_of(lower, std::string::npos);
    std::string ext = e1.first;
    static const char *compressed[] = {"gz", "gzip", "zst", "zstd", "bz2", "bzip2", "xz"};
    if (std::find_if(std::begin(compressed), std::end(compressed), [&](const char *c) { return ext == c; }) != std::end(compressed) &&
        e1.second > 0)
        ext = ext_of(lower, e1.second).first;
    if (ext == "fasta" || ext == "fa" || ext == "fna") res.file_type = "FASTA";
    if (ext == "fastq" || ext == "fq") res.file_type = "FASTQ";
    if (ext == "vcf") res.file_type = "VCF";
    return res;
}

// ---- which scan first (include/exon_gpu.h: exg_scan_algo_hint) ---------------------------------------------------------------------
// What the lean scan cannot do in its single pass, and so marks for the any-shape run behind it (exg_fused_core.hpp): a record that
// begins in front of its half's 1 KiB window or whose last four newlines do not fit in it (FASTQ: four lines; VCF: lines of a few
// hundred bytes), a 16 KiB half with more lines than its list holds (FASTQ 512: lines of < 32 bytes on average; VCF 1024), bytes
// >= 0x80 (UTF-8 validation).  A sample of the first MiB tells all of that apart.
extern "C" int exg_scan_algo_hint(int format, const void *sample, uint64_t n_bytes) {
    if (format != EXG_FMT_FASTQ && format != EXG_FMT_VCF) return EXG_ALGO_FUSED;  // (FASTA has one scan)
    const uint8_t *p = (const uint8_t *)sample;
    const uint64_t n = n_bytes < (1u << 20) ? n_bytes : (1u << 20);
    if (!p || n < 4096) return EXG_ALGO_FUSED;  // (too little to tell: the batch's own result decides)
    uint64_t lines = 0, hi = 0, last_nl = 0;
    for (uint64_t i = 0; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        hi |= w & 0x8080808080808080ull;
        const uint64_t x = w ^ 0x0A0A0A0A0A0A0A0Aull;
        uint64_t m = ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);  // 0x80 in every byte equal to '\n' (exact)
        while (m) {
            lines++;
            last_nl = i + ((uint64_t)__builtin_ctzll(m) >> 3);
            m &= m - 1;
        }
    }
    if (hi) return EXG_ALGO_FUSED_FULL;
    if (lines < 4) return format == EXG_FMT_VCF ? EXG_ALGO_FUSED_INDEX : EXG_ALGO_FUSED_FULL;  // lines of hundreds of KiB
    const uint64_t avg = (last_nl + 1) / lines;
    if (format == EXG_FMT_FASTQ) return (avg * 4 >= 900 || avg < 34) ? EXG_ALGO_FUSED_FULL : EXG_ALGO_FUSED;
    if (avg >= 640) return getenv("EXG_NO_VCF_INDEX") ? EXG_ALGO_FUSED_FULL : EXG_ALGO_FUSED_INDEX;
    return (avg >= 200 || avg < 18) ? EXG_ALGO_FUSED_FULL : EXG_ALGO_FUSED;
}

