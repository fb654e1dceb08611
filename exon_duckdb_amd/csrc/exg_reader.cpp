// exg_reader.cpp — reader level of the C-ABI (include/exon_gpu.h, layer 2): the entry points.
//   exg_open / exg_schema_of / exg_next_chunk / exg_release_chunk / exg_count_only / exg_close
// Replaces `new_reader` + the ArrowArrayStream it hands back (exon/include/rust.hpp:41-46,
// rust/src/arrow_reader.rs:38-166) and the per-batch pull in WTArrowTableFunction::Scan
// (exon/src/exon/arrow_table_function/module.cpp:257-294).  A reader is used by one thread at a time, like the
// reference's stream (MaxThreads() == 1 there); the work behind it is in exg_rd_*.cpp (see exg_rd_internal.hpp).
#include <sys/stat.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "exg_filter.hpp"
#include "exg_map_guard.hpp"
#include "exg_rd_fanout.hpp"
#include "exg_rd_source.hpp"

using namespace exg_rd;

static void flat_schema(const exg_reader *r, exg_schema *out);

extern "C" int exg_open(const exg_open_args *args, exg_reader **out) {
    if (!args || !out || !args->path || !args->file_format) {
        exg::set_error("exg_open: null argument");
        return EXG_E_INVALID_ARG;
    }
    *out = nullptr;
    std::unique_ptr<exg_reader> r(new exg_reader());
    std::string fmt = args->file_format;
    for (char &ch : fmt) ch = (char)tolower((unsigned char)ch);
    if (fmt == "fasta")
        r->format = EXG_FMT_FASTA;
    else if (fmt == "fastq")
        r->format = EXG_FMT_FASTQ;
    else if (fmt == "vcf")
        r->format = EXG_FMT_VCF;
    else {
        // rust/src/arrow_reader.rs:93-102
        exg::set_error("could not parse file_format %s", args->file_format);
        return EXG_E_INVALID_ARG;
    }
    std::string path = args->path;
    r->compression = compression_of(args);
    if (args->batch_rows) r->batch_rows = args->batch_rows;
    if (r->batch_rows % 64) {
        exg::set_error("exg_open: batch_rows must be a multiple of 64 (validity words)");
        return EXG_E_INVALID_ARG;
    }
    if (args->device_batch_bytes) r->device_batch_bytes = (args->device_batch_bytes + 15) / 16 * 16;
    else if (const char *e = getenv("EXG_DEVICE_BATCH_BYTES"))  // tuning / test knob
        r->device_batch_bytes = std::max<uint64_t>(4096, (strtoull(e, nullptr, 10) + 15) / 16 * 16);
    // EXG_DEVICE_MEM_CAP_MB: the device memory one reader may hold.  Device batches (and the decoded segments of a
    // compressed input) are sized from it: a batch in flight costs about ten times its bytes (segments queued and being
    // decoded, compressed windows, the scan's workspace and column vectors)
    if (const char *e = getenv("EXG_DEVICE_MEM_CAP_MB")) {
        r->mem_cap = strtoull(e, nullptr, 10) << 20;
        r->meter.cap = r->mem_cap;
        // (per input byte a scan provisions 16 B x columns / 32 (FASTQ) or / 16 (VCF: 9 columns + POS + QUAL) of column
        // vectors; a compressed input adds up to four segments and two compressed windows)
        // (24 for FASTQ: with 20 a single-member gzip under a 16 MiB cap peaked between 15.4 and 17.3 MB depending on how far the
        // decoder thread happened to run ahead of the scan — the first round's symbol buffer is sized for the worst ratio)
        const uint64_t div = r->format == EXG_FMT_VCF ? 64 : r->format == EXG_FMT_FASTA ? 32 : 24;
        if (r->mem_cap) r->device_batch_bytes = std::max<uint64_t>(64u << 10, std::min<uint64_t>(r->device_batch_bytes, (r->mem_cap / div) & ~15ull));
    }
    r->halo_want = getenv("EXG_SHARD_HALO") ? std::max<uint64_t>(16, strtoull(getenv("EXG_SHARD_HALO"), nullptr, 10)) : kShardHalo;
    r->device = args->device;
    r->shard_count = args->shard_count ? args->shard_count : 1;
    r->expect_chunks = (args->flags & EXG_OPEN_CHUNKS) != 0;
    r->want_cols = args->columns ? args->columns : ~0ull;
    r->shard_index = args->shard_index;
    if (r->shard_index >= r->shard_count) {
        exg::set_error("exg_open: shard_index %u is not below shard_count %u", r->shard_index, r->shard_count);
        return EXG_E_INVALID_ARG;
    }
    int rc = list_files(r.get(), path);
    if (rc) return rc;
    if (r->compression != kNone && r->compression != kGzip && r->compression != kZstd) {
        exg::set_error("compression is not supported: gzip and zstd have device decoders, bzip2 / xz do not, and there is no CPU fallback");
        return EXG_E_UNSUPPORTED;
    }
    const int n_dev = exg_device_count();
    if (n_dev < 1) return EXG_E_NO_DEVICE;
    if (r->device < 0 || r->device >= n_dev) {
        exg::set_error("exg_open: device %d does not exist (%d visible)", r->device, n_dev);
        return EXG_E_INVALID_ARG;
    }
    DeviceGuard guard(r->device);
    hipError_t he = exg_rd::stream_pool()->take(r->device, &r->stream, /*high=*/getenv("EXG_NO_SCAN_PRIORITY") == nullptr);
    if (he != hipSuccess) {
        exg::set_error("cannot initialise device %d: %s", r->device, hipGetErrorString(he));
        return EXG_E_HIP;
    }
    if (args->filters && *args->filters) {
        // `SELECT * FROM exon_table WHERE <filters>` (arrow_reader.rs:125-141), evaluated on the device
        exg_schema sch;
        flat_schema(r.get(), &sch);
        std::vector<FilterColumn> fcols;  // nested columns ('x') are refused by the parser, like in new_reader
        for (int c = 0; c < sch.n_columns; c++)
            fcols.push_back({sch.names[c], sch.types[c] == EXG_TYPE_BIGINT ? 'l' : sch.types[c] == EXG_TYPE_FLOAT ? 'f' : sch.types[c] == EXG_TYPE_VARCHAR ? 'u' : 'x'});
        const std::string text = args->filters;
        FilterParser fp(text, fcols);
        if (!fp.parse()) {
            exg::set_error("could not execute sql: %s", fp.err.c_str());
            return EXG_E_INVALID_ARG;
        }
        // (from the device pool: a filtered query does not pay two hipMalloc / hipFree pairs per open)
        r->filter_prog_bytes = sizeof fp.prog;
        r->filter_consts_bytes = fp.consts.size() + 16;
        if (!(r->d_filter_prog = exg_rd::dev_pool()->take(r->device, r->filter_prog_bytes)) ||
            !(r->d_filter_consts = exg_rd::dev_pool()->take(r->device, r->filter_consts_bytes)) ||
            hipMemcpy(r->d_filter_prog, &fp.prog, sizeof fp.prog, hipMemcpyHostToDevice) != hipSuccess ||
            (!fp.consts.empty() &&
             hipMemcpy(r->d_filter_consts, fp.consts.data(), fp.consts.size(), hipMemcpyHostToDevice) != hipSuccess)) {
            exg::set_error("could not execute sql: device allocation failed");
            return EXG_E_HIP;
        }
        r->has_filter = true;
        for (uint32_t k = 0; k < fp.prog.n_ops; k++)
            if (fp.prog.ops[k].op != exg::arrow::kOpAnd && fp.prog.ops[k].op != exg::arrow::kOpOr) r->filter_cols |= 1ull << fp.prog.ops[k].col;
    }
    if (args->shard_count == 0) {
        // several devices (or stripes forced): this reader becomes the front of a fan-out
        std::vector<Stripe> stripes;
        unsigned workers = 1;
        if ((rc = plan_stripes(r->files, r->compression, args, &stripes, &workers))) return rc;
        if (stripes.size() > r->files.size()) {
            struct Sub : FanSub {
                exg_reader *rd = nullptr;
                ~Sub() override { exg_close(rd); }
                int next(FanItem *item, std::string *err) override {
                    DeviceGuard guard(rd->device);
                    MeterScope meter_scope(&rd->meter);
                    bool end = false;
                    const int rc = advance_batch(rd, &end);
                    if (rc) {
                        *err = rd->error;
                        return rc;
                    }
                    if (!end) {
                        if (rd->batch->wait_landed()) {
                            *err = "the batch's vectors did not arrive (hipEventSynchronize failed)";
                            return EXG_E_HIP;
                        }
                        item->rows = rd->batch->n_rows;
                        item->batch = rd->batch;  // the batch's host buffers outlive its reader
                        rd->batch.reset();
                    }
                    return EXG_OK;
                }
                int count(uint64_t *rows, std::string *err) override {
                    const int rc = exg_count_only(rd, rows);
                    if (rc) *err = rd->error;
                    return rc;
                }
                void stats(uint64_t *now, uint64_t *peak, uint64_t *batches, uint64_t *segments) override {
                    *now = rd->meter.cur.load(), *peak = rd->meter.peak.load();
                    *batches = rd->n_batches.load(), *segments = rd->n_segments.load();
                }
            };
            const std::string format = args->file_format, compression = args->compression ? args->compression : "", filters = args->filters ? args->filters : "";
            const bool has_comp = args->compression != nullptr;
            const uint64_t batch_rows = r->batch_rows, dbb = args->device_batch_bytes, columns = args->columns;
            FanOpen open = [=](const Stripe &s, std::unique_ptr<FanSub> *sub, std::string *err) -> int {
                exg_open_args a;
                memset(&a, 0, sizeof a);
                a.path = s.path.c_str();
                a.file_format = format.c_str();
                a.compression = has_comp ? compression.c_str() : nullptr;
                a.batch_rows = batch_rows;
                a.device = s.device;
                a.device_batch_bytes = dbb;
                a.filters = filters.empty() ? nullptr : filters.c_str();
                a.shard_index = s.shard_index;
                a.shard_count = s.shard_count;
                a.columns = columns;
                if (a.shard_count < 1) a.shard_count = 1;
                std::unique_ptr<Sub> x(new Sub());
                const int orc = exg_open(&a, &x->rd);
                if (orc) {
                    *err = exg_last_error_message();
                    return orc;
                }
                *sub = std::move(x);
                return EXG_OK;
            };
            r->fan.reset(new FanOut(std::move(stripes), workers, std::move(open), 2));
        }
    }
    *out = r.release();
    return EXG_OK;
}

// names / types of the columns; no file is touched (the VCF trees come from nested_schema)
static void flat_schema(const exg_reader *r, exg_schema *out) {
    static const exg_type fastq_t[4] = {{EXG_TYPE_VARCHAR, 0, "name", 0, nullptr}, {EXG_TYPE_VARCHAR, 1, "description", 0, nullptr},
                                        {EXG_TYPE_VARCHAR, 0, "sequence", 0, nullptr}, {EXG_TYPE_VARCHAR, 0, "quality_scores", 0, nullptr}};
    static const exg_type fasta_t[3] = {{EXG_TYPE_VARCHAR, 0, "id", 0, nullptr}, {EXG_TYPE_VARCHAR, 1, "description", 0, nullptr},
                                        {EXG_TYPE_VARCHAR, 0, "sequence", 0, nullptr}};
    memset(out, 0, sizeof *out);
    if (r->format == EXG_FMT_FASTQ) {
        // order pinned by test_fastq_scan.test:35-41; names as exon 0.2.6 registers them
        out->n_columns = 4;
        for (int i = 0; i < 4; i++) out->names[i] = fastq_t[i].name, out->types[i] = EXG_TYPE_VARCHAR, out->nullable[i] = fastq_t[i].nullable, out->tree[i] = &fastq_t[i];
    } else if (r->format == EXG_FMT_FASTA) {
        // `id` pinned by test_fasta_scan.test:34-37, order + NULL description by test_fasta_copy.test:75-80
        out->n_columns = 3;
        for (int i = 0; i < 3; i++) out->names[i] = fasta_t[i].name, out->types[i] = EXG_TYPE_VARCHAR, out->nullable[i] = fasta_t[i].nullable, out->tree[i] = &fasta_t[i];
    } else {
        // test_vcf_record_scan.test:10-19: alt is a LIST, info a STRUCT (module.cpp:126-147 maps exon's Arrow schema)
        static const char *n[] = {"chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "formats"};
        static const int t[] = {EXG_TYPE_VARCHAR, EXG_TYPE_BIGINT, EXG_TYPE_LIST, EXG_TYPE_VARCHAR, EXG_TYPE_LIST, EXG_TYPE_FLOAT, EXG_TYPE_LIST, EXG_TYPE_STRUCT, EXG_TYPE_LIST};
        out->n_columns = 9;
        for (int i = 0; i < 9; i++) out->names[i] = n[i], out->types[i] = t[i], out->nullable[i] = !(i == 0 || i == 1 || i == 3);
    }
}

extern "C" int exg_schema_of(exg_reader *r, exg_schema *out) {
    if (!r || !out) return EXG_E_INVALID_ARG;
    r->join_ahead();
    flat_schema(r, out);
    if (r->format == EXG_FMT_VCF) {
        DeviceGuard guard(r->device);
        MeterScope meter_scope(&r->meter);
        int rc = nested_prepare(r);  // the INFO / FORMAT keys of the first file's header are part of the schema
        if (rc) return rc;
        nested_schema(r, out);
        if (r->fan) r->src.reset();  // (the front of a fan-out only needed the header: its decoder stops)
    }
    return EXG_OK;
}

// elements [e0, e1) of a nested node as the vector of DataChunk `chunk`: data is a slice of the batch-wide array; validity
// too when the slice begins on a word boundary, else its bits are shifted into words of their own; the children of a LIST
// are the elements between the chunk's bases, those of a STRUCT share the parent's range
static void slice_vector(const NVec &v, uint64_t e0, uint64_t e1, uint64_t chunk, ChunkKeep *keep, exg_vector *out) {
    memset(out, 0, sizeof *out);
    out->length = e1 - e0;
    out->data = v.data ? (void *)((const char *)v.data + (v.zero ? 0 : e0 * v.elem)) : nullptr;
    if (v.validity) {
        if (v.zero) {
            out->validity = (uint64_t *)v.validity;
        } else if ((e0 & 63) == 0) {
            out->validity = (uint64_t *)v.validity + e0 / 64;
        } else {
            const uint64_t n = e1 - e0, words = (n + 63) / 64, sh = e0 & 63;
            keep->words.emplace_back(new uint64_t[words ? words : 1]);
            uint64_t *w = keep->words.back().get();
            const uint64_t last_word = (e1 + 63) / 64;  // words of the source that exist
            for (uint64_t i = 0; i < words; i++) {
                const uint64_t lo = v.validity[e0 / 64 + i] >> sh;
                const uint64_t hi = e0 / 64 + i + 1 < last_word ? v.validity[e0 / 64 + i + 1] << (64 - sh) : 0;
                w[i] = lo | hi;
            }
            out->validity = w;
        }
    }
    if (v.children.empty()) return;
    keep->nodes.emplace_back(new exg_vector[v.children.size()]);
    exg_vector *kids = keep->nodes.back().get();
    for (size_t i = 0; i < v.children.size(); i++) {
        if (v.type == EXG_TYPE_LIST)
            slice_vector(v.children[i], v.child_base[chunk], v.child_base[chunk + 1], chunk, keep, &kids[i]);
        else
            slice_vector(v.children[i], e0, e1, chunk, keep, &kids[i]);
    }
    out->n_children = (int)v.children.size();
    out->children = kids;
}

// the batch behind `cur` on a thread of the reader's own (exg_reader.hpp: ahead)
static void start_ahead(exg_reader *r) {
    r->ahead_done = false;
    r->ahead = std::thread([r] {
        DeviceGuard guard(r->device);
        MeterScope meter_scope(&r->meter);
        r->ahead_end = false;
        r->ahead_rc = advance_batch(r, &r->ahead_end);
        r->ahead_done = true;
    });
}

extern "C" int exg_next_chunk(exg_reader *r, exg_chunk *out) {
    if (!r || !out) return EXG_E_INVALID_ARG;
    DeviceGuard guard(r->device);
    MeterScope meter_scope(&r->meter);
    memset(out, 0, sizeof *out);
    static const bool no_ahead = getenv("EXG_NO_RUN_AHEAD") != nullptr;
    for (;;) {
        if (r->cur && r->cur_row < r->cur->n_rows) {
            // A text file that shrank under the query (exg_map_guard.hpp): asked of the mapping THIS batch's strings point into —
            // the reader's current file may be another one by now, and a fan-out's front has none of its own
            if (r->cur->file && r->cur->file->guard >= 0 && MapGuard::hit(r->cur->file->guard))
                return fail(r, EXG_E_IO, "an input file was truncated while it was being read");
            const Batch &bt = *r->cur;
            uint64_t row0 = r->cur_row;
            uint64_t n = std::min<uint64_t>(r->batch_rows, bt.n_rows - row0);
            out->n_rows = n;
            out->n_columns = bt.n_cols;
            ChunkKeep *keep = new ChunkKeep();
            keep->batch = r->cur;
            const uint64_t chunk = row0 / r->batch_rows;
            for (int c = 0; c < bt.n_cols; c++) {
                exg_vector &v = keep->top[c];
                if ((size_t)c < bt.nested.size() && bt.nested[c].type) {
                    slice_vector(bt.nested[c], row0, row0 + n, chunk, keep, &v);
                } else if (!bt.cols[c]) {  // not in the projection (exg_open_args.columns)
                    out->data[c] = nullptr;
                    out->validity[c] = nullptr;
                    out->vectors[c] = nullptr;
                    continue;
                } else {
                    memset(&v, 0, sizeof v);
                    v.data = (char *)bt.cols[c] + row0 * bt.elem[c];
                    v.validity = bt.validity[c] ? (uint64_t *)bt.validity[c] + row0 / 64 : nullptr;
                    v.length = n;
                }
                out->data[c] = v.data;
                out->validity[c] = v.validity;
                out->vectors[c] = &v;
            }
            out->keepalive = keep;
            out->batch_no = bt.seq;
            r->cur_row += n;
            // the first chunk of a batch leaves: the batch behind it starts being made
            if (row0 == 0 && !no_ahead && !r->fan && !r->ahead.joinable() && !r->ahead_done) start_ahead(r);
            return EXG_OK;
        }
        r->cur.reset();
        if (r->fan) {
            // the batches of the stripes' readers, in file order
            FanItem item;
            std::string msg;
            const int rc = r->fan->next(&item, &msg);
            if (rc) return fail(r, rc, msg);
            if (!item.batch) return EXG_OK;  // n_rows == 0: end of stream
            r->cur = std::static_pointer_cast<Batch>(item.batch);
            r->cur->seq = r->batch_seq++;
            r->cur_row = 0;
            continue;
        }
        bool end = false;
        int rc;
        r->join_ahead();
        if (r->ahead_done) {
            r->ahead_done = false;
            rc = r->ahead_rc;
            end = r->ahead_end;
            if (rc) exg::set_error("%s", r->error.c_str());  // (the message was set on the other thread)
        } else {
            rc = advance_batch(r, &end);
        }
        if (rc) return rc;
        if (end) return EXG_OK;  // n_rows == 0: end of stream
        r->cur = std::move(r->batch);
        r->batch.reset();
        r->cur_row = 0;
        if (r->cur && r->cur->landed) {
            // read_vcf: the batch's vectors are still on their way — the batch behind it starts being made right now, beside them
            if (!no_ahead && r->cur->n_rows && !r->ahead.joinable() && !r->ahead_done) start_ahead(r);
            if (r->cur->wait_landed()) return fail(r, EXG_E_HIP, "the batch's vectors did not arrive (hipEventSynchronize failed)");
        }
    }
}

extern "C" void exg_release_chunk(exg_reader *, exg_chunk *chunk) {
    if (chunk && chunk->keepalive) {
        delete (ChunkKeep *)chunk->keepalive;
        chunk->keepalive = nullptr;
    }
}

extern "C" int exg_count_only(exg_reader *r, uint64_t *n_rows) {
    if (!r || !n_rows) return EXG_E_INVALID_ARG;
    DeviceGuard guard(r->device);
    MeterScope meter_scope(&r->meter);
    r->join_ahead();  // (chunks were pulled before: the batch that was being made behind them is dropped with the rest of them)
    r->ahead_done = false;
    r->cur.reset();
    if (r->fan) {
        std::string msg;
        const int rc = r->fan->count(n_rows, &msg);
        return rc ? fail(r, rc, msg) : EXG_OK;
    }
    uint64_t total = 0;
    for (;;) {
        if (r->pending_error) {
            std::string msg = std::string(exg_parse_error_string(r->pending_error)) + " at byte " +
                              std::to_string(r->pending_error_offset) + " of " + r->files[r->file_idx - 1];
            r->pending_error = 0;
            return fail(r, EXG_E_PARSE, msg);
        }
        if (r->file_done) {
            if (int jrc = r->finish_source()) return jrc;
            if (r->file_idx >= r->files.size()) break;
            int rc = open_next_file(r);
            if (rc) return rc;
        }
        uint64_t k;
        int rc = next_batch(r, true, &k);
        if (rc) return rc;
        total += k;
    }
    *n_rows = total;
    return EXG_OK;
}

// Give back what the process-wide pools hold (device buffers, pinned host blocks, streams of closed readers): for the
// extension's unload / idle path — a long-lived DuckDB process sharing the GPU should not sit on tens of GB it no longer
// uses.  Readers that are open keep what they hold.
extern "C" void exg_trim_pools(void) {
    exg_rd::dev_pool()->trim(0);
    exg_rd::global_pool()->trim();
    exg_rd::stream_pool()->trim();
}

extern "C" const char *exg_reader_error(exg_reader *r) { return r ? r->error.c_str() : ""; }

extern "C" int exg_reader_stats_of(exg_reader *r, exg_reader_stats *out) {
    if (!r || !out) return EXG_E_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->device_bytes_now = r->meter.cur.load();
    out->device_bytes_peak = r->meter.peak.load();
    out->device_mem_cap = r->mem_cap;
    out->device_batch_bytes = r->device_batch_bytes;
    out->device_batches = r->n_batches;
    out->decoded_segments = r->n_segments;
    out->scan_algo = r->fan ? 0 : r->fused_algo;
    out->nested_ns = r->nested_ns.load();
    out->host_vector_bytes = r->host_vector_bytes.load();
    for (const std::string &f : r->files) {
        struct stat sb;
        if (stat(f.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) out->input_bytes += (uint64_t)sb.st_size;
    }
    out->input_compression = r->compression == exg_rd::kNone ? 0 : r->compression == exg_rd::kZstd ? 2 : 1;
    if (r->fan) {
        // the front of a fan-out holds next to nothing itself: its stripes' readers (their own meters, on the workers' threads) do
        const exg_rd::FanOut::Stats fs = r->fan->stats();
        out->device_bytes_now += fs.device_bytes_now;
        out->device_bytes_peak += fs.device_bytes_peak;
        out->device_batches += fs.device_batches;
        out->decoded_segments += fs.decoded_segments;
    }
    return EXG_OK;
}

extern "C" void exg_free_device(void *d_ptr, uint64_t bytes) {
    if (!d_ptr) return;
    int dev = 0;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_ptr) == hipSuccess) dev = attr.device;
    else (void)hipGetLastError(), (void)hipGetDevice(&dev);
    exg_rd::dev_pool()->give(dev, d_ptr, (size_t)bytes);
}

extern "C" void exg_close(exg_reader *r) { delete r; }
