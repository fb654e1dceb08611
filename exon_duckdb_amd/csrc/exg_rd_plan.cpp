// exg_rd_plan.cpp — host only: which decoder an input needs (rust/src/arrow_reader.rs:60-91), how many byte-range shards
// a scan is worth and where they run (in-process form), and the reference's `replacement_scan`
// (rust/src/arrow_reader.rs:173-197).
#include <string.h>
#include <sys/stat.h>

#include <algorithm>

#include "exg_rd_internal.hpp"

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
    exg_reader tmp;
    if (list_files(&tmp, args->path) != EXG_OK) return EXG_OK;  // the open will report it
    uint64_t bytes = 0;
    bool shardable = comp == kNone || (comp == kGzip && fmt != "fasta");
    for (const std::string &f : tmp.files) {
        struct stat st;
        if (stat(f.c_str(), &st) != 0) continue;
        bytes += (uint64_t)st.st_size;
        if (comp == kGzip && shardable) {  // BGZF: FEXTRA with a 'BC' subfield in the first member
            uint8_t h[18] = {0};
            FILE *fp = fopen(f.c_str(), "rb");
            const size_t got = fp ? fread(h, 1, sizeof h, fp) : 0;
            if (fp) fclose(fp);
            shardable = got == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C';
        }
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
