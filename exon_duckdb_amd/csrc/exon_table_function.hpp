// exon_table_function.hpp — the host glue of the drop-in: the reference's WTArrowTableFunction
// (exon/src/exon/arrow_table_function/module.cpp:75-318, exon/include/exon/arrow_table_function/module.hpp) re-designed on
// top of the reader-level C-ABI (include/exon_gpu.h), written ONCE against a traits struct `D` that names DuckDB's types:
//
//   * duckdb_shim/exon_extension.cpp instantiates it with DuckDB v0.8.1's own classes (the real `LOAD exon` binding; it
//     needs DuckDB's headers, which do not exist on the build box);
//   * csrc/testing/exon_tf_harness.cpp instantiates it with csrc/testing/duck_mini.hpp, the slice of that API restated for
//     the tests — so every line of logic below is compiled and exercised by the GPU test-suite, and the shim adds only the
//     adapters (how a LogicalType / a Vector is made in real DuckDB) and the registration calls.
//
// Same lifecycle as the reference — bind (schema from a reader that is closed again), init_global (FilterToString ->
// `filters`), init_local, scan (<= STANDARD_VECTOR_SIZE rows per call, zero rows = end) — with these deliberate differences
// (SURVEY.md Appendix B):
//   * no Arrow hop: the DataChunk's vectors reference the engine's host buffers (kept alive by the vector buffer);
//   * SEVERAL scan threads: the reference pins MaxThreads() to 1 (module.cpp:36 has an unused `max_threads = 6`).  Here
//     init_global plans byte-range shards (exg_plan_shards: as many as there are devices, when the input can be sharded and
//     is large enough) and MaxThreads() returns that number — an UPPER bound: DuckDB runs fewer scan threads when
//     `threads` is smaller or the pipeline is not parallel.  A scan thread therefore does not own a shard for life: whenever
//     its reader is exhausted (or it has none yet) it claims the next unclaimed shard and opens that shard's reader on that
//     shard's device, until no shard is left — one thread scans every shard in file order, N threads scan N shards at a
//     time on N GPUs, nothing exchanged between them.  COUNT(*) (only the row id projected) is counted per
//     shard and summed by DuckDB's aggregate;
//   * get_batch_index orders the chunks (shard-major, then the shard's device batches), so an order-preserving plan sees
//     the file order; a device batch (~256 MiB of input) is one DuckDB batch, and the index stays far below DuckDB's
//     per-pipeline increment (10^13) for any shard count the planner allows.
//
// What D provides: the types FunctionData, GlobalTableFunctionState, LocalTableFunctionState, LogicalType, DataChunk,
// TableFilter, ConstantFilter, ConjunctionFilter, TableFilterSet, TableFilterType, idx_t, the constants
// RowId / VectorSize, and the adapters
//   static LogicalType ToLogical(const exg_type &);                                       (module.cpp:126-147)
//   static void Reference(DataChunk &, idx_t out_col, const LogicalType &, const exg_vector &, std::shared_ptr<ExonChunk>);
//   static void SetCardinality(DataChunk &, idx_t);
//   static std::string ComparisonOperator(const ConstantFilter &);   static std::string ConstantSQL(const ConstantFilter &);
#pragma once
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "exon_gpu.h"

namespace exon_scan {

// one engine chunk: released when the last vector that references it is dropped
struct ExonChunk {
    exg_reader *reader = nullptr;
    exg_chunk chunk;
    ExonChunk() { memset(&chunk, 0, sizeof chunk); }
    ~ExonChunk() {
        if (chunk.keepalive) exg_release_chunk(reader, &chunk);
    }
    ExonChunk(const ExonChunk &) = delete;
    ExonChunk &operator=(const ExonChunk &) = delete;
};

template <class D>
struct ExonTableFunction {
    using idx_t = typename D::idx_t;
    using LogicalType = typename D::LogicalType;

    // module.cpp:29-41
    struct BindData : public D::FunctionData {
        std::string file_type;
        std::string compression;  // "auto_detect" when the named parameter is absent (module.cpp:85)
        std::string file_name;
        std::vector<std::string> all_names;
        std::vector<LogicalType> all_types;
        uint64_t input_bytes = 0;        // size of the input files on disk (exg_reader_stats.input_bytes at bind)
        uint64_t input_compression = 0;  // 0 text, 1 gzip, 2 zstd
    };

    struct GlobalState : public D::GlobalTableFunctionState {
        std::vector<idx_t> column_ids;
        std::string filter_clause;
        uint64_t columns = 0;  // exg_open_args.columns: the projection as a mask (0 = all)
        uint64_t open_flags = 0;  // exg_open_args.flags
        bool count_only = false;
        uint32_t n_shards = 1;
        int devices[64];
        std::atomic<uint32_t> next_shard{0};
        idx_t MaxThreads() const override { return n_shards; }
    };

    struct LocalState : public D::LocalTableFunctionState {
        exg_reader *reader = nullptr;
        uint32_t shard = 0;
        uint64_t batch_no = 0;  // the device batch (of the current shard's reader) the last chunk came from
        bool counted = false;
        uint64_t count_remaining = 0;
        ~LocalState() override {
            if (reader) exg_close(reader);
        }
    };
    static constexpr unsigned kBatchBits = 24;  // device batches per shard: 2^24 x 256 MiB = 4 PiB

    static exg_reader *OpenReader(const BindData &d, const std::string &filter_clause, uint64_t columns, uint64_t flags, uint32_t shard, uint32_t n_shards, int device) {
        exg_open_args a;
        memset(&a, 0, sizeof a);
        a.filters = filter_clause.empty() ? nullptr : filter_clause.c_str();                // module.cpp:239-243
        a.path = d.file_name.c_str();
        a.file_format = d.file_type.c_str();
        a.compression = d.compression == "auto_detect" ? nullptr : d.compression.c_str();  // module.cpp:95-103
        a.batch_rows = D::VectorSize;                                                      // module.cpp:83
        a.shard_index = shard;
        a.shard_count = n_shards;
        a.device = device;
        a.columns = columns;  // projection_pushdown (module.cpp:310): only these columns' vectors cross PCIe
        a.flags = flags;
        exg_reader *r = nullptr;
        if (exg_open(&a, &r) != EXG_OK) throw std::runtime_error(exg_last_error_message());  // module.cpp:105-108
        return r;
    }

    // module.cpp:75-156.  The reference learns the schema by opening a reader; so does this (and closes it again).
    static std::unique_ptr<BindData> Bind(const std::string &file_name, const std::string &compression, const std::string &file_type,
                                          std::vector<LogicalType> &return_types, std::vector<std::string> &names) {
        auto result = std::make_unique<BindData>();
        result->file_name = file_name;
        result->compression = compression.empty() ? "auto_detect" : compression;
        result->file_type = file_type;
        exg_reader *r = OpenReader(*result, "", 0, 0, 0, 1, 0);
        exg_schema sch;
        const int rc = exg_schema_of(r, &sch);
        if (rc == EXG_OK)
            for (int i = 0; i < sch.n_columns; i++) {
                return_types.push_back(D::ToLogical(*sch.tree[i]));  // (the trees are the reader's: convert before it closes)
                names.emplace_back(sch.names[i]);
            }
        const std::string why = rc == EXG_OK ? "" : exg_reader_error(r);
        exg_reader_stats st;
        if (exg_reader_stats_of(r, &st) == EXG_OK) result->input_bytes = st.input_bytes, result->input_compression = st.input_compression;
        exg_close(r);
        if (rc != EXG_OK) throw std::runtime_error("Failed to get schema: " + why);  // module.cpp:112-119
        result->all_names = names;
        result->all_types = return_types;
        return result;
    }

    static std::string Join(const std::vector<std::string> &v, const std::string &sep) {
        std::string out;
        for (size_t i = 0; i < v.size(); i++) out += (i ? sep : "") + v[i];
        return out;
    }

    // module.cpp:158-199: the predicate text the engine parses again (`filters` of exg_open / new_reader)
    static std::string FilterToString(const typename D::TableFilter &filter, const std::string &column_name) {
        using T = typename D::TableFilterType;
        switch (filter.filter_type) {
            case T::CONSTANT_COMPARISON: {
                auto &cf = static_cast<const typename D::ConstantFilter &>(filter);
                return column_name + D::ComparisonOperator(cf) + D::ConstantSQL(cf);
            }
            case T::CONJUNCTION_AND:
            case T::CONJUNCTION_OR: {
                auto &cj = static_cast<const typename D::ConjunctionFilter &>(filter);
                std::vector<std::string> parts;
                for (auto &c : cj.child_filters) parts.push_back(FilterToString(*c, column_name));
                return Join(parts, filter.filter_type == T::CONJUNCTION_AND ? " AND " : " OR ");
            }
            case T::IS_NOT_NULL: return column_name + " IS NOT NULL";
            case T::IS_NULL: return column_name + " IS NULL";
            default: break;
        }
        throw std::runtime_error("FilterToString: filter type not implemented");
    }
    // module.cpp:201-214
    static std::string FilterToString(const typename D::TableFilterSet &set, const std::vector<idx_t> &column_ids,
                                      const std::vector<std::string> &column_names) {
        std::vector<std::string> parts;
        for (auto &f : set.filters) parts.push_back(FilterToString(*f.second, column_names.at(column_ids.at(f.first))));
        return Join(parts, " AND ");
    }

    // module.cpp:216-255
    static std::unique_ptr<GlobalState> InitGlobal(const BindData &data, const std::vector<idx_t> &column_ids,
                                                   const typename D::TableFilterSet *filters) {
        auto gs = std::make_unique<GlobalState>();
        gs->column_ids = column_ids;
        gs->count_only = true;
        for (idx_t c : column_ids) {
            gs->count_only = gs->count_only && c == D::RowId;
            if (c != D::RowId && c < 63) gs->columns |= (uint64_t)1 << c;
        }
        // the scan is going to pull chunks (anything but COUNT(*)): a compressed input's decoded segments travel to the host from
        // the first one on, beside the decoder (include/exon_gpu.h: EXG_OPEN_CHUNKS)
        if (!gs->count_only) gs->open_flags |= EXG_OPEN_CHUNKS;
        if (filters) gs->filter_clause = FilterToString(*filters, column_ids, data.all_names);  // module.cpp:222-226
        exg_open_args a;
        memset(&a, 0, sizeof a);
        a.path = data.file_name.c_str();
        a.file_format = data.file_type.c_str();
        a.compression = data.compression == "auto_detect" ? nullptr : data.compression.c_str();
        a.filters = gs->filter_clause.empty() ? nullptr : gs->filter_clause.c_str();
        uint32_t n = 1;
        if (exg_plan_shards(&a, &n, gs->devices, 64) != EXG_OK) throw std::runtime_error(exg_last_error_message());
        gs->n_shards = n;
        return gs;
    }

    // one per scan thread (<= MaxThreads(), and DuckDB may run fewer): the shards are claimed in Scan, one after the other
    static std::unique_ptr<LocalState> InitLocal(const BindData &, GlobalState &) { return std::make_unique<LocalState>(); }

    // the next unclaimed shard becomes this thread's: false when none is left
    static bool ClaimShard(const BindData &data, GlobalState &gs, LocalState &ls) {
        if (ls.reader) {
            exg_close(ls.reader);  // (chunks it handed out keep their buffers alive on their own)
            ls.reader = nullptr;
        }
        const uint32_t shard = gs.next_shard.fetch_add(1);
        if (shard >= gs.n_shards) {
            gs.next_shard.store(gs.n_shards);  // (no wrap-around, however often exhausted threads ask)
            return false;
        }
        ls.shard = shard;
        ls.batch_no = 0;
        ls.counted = false;
        ls.reader = OpenReader(data, gs.filter_clause, gs.columns, gs.open_flags, shard, gs.n_shards, gs.devices[shard]);
        return true;
    }

    // module.cpp:257-294: leaves output.size() == 0 at the end of the (thread's) stream — here: when its reader is
    // exhausted AND no shard is left to claim
    static void Scan(const BindData &data, GlobalState &gs, LocalState *ls, typename D::DataChunk &output) {
        if (!ls) return;  // (module.cpp:259-261)
        D::SetCardinality(output, 0);
        for (;;) {
            if (!ls->reader && !ClaimShard(data, gs, *ls)) return;
            if (gs.count_only) {
                if (!ls->counted) {
                    if (exg_count_only(ls->reader, &ls->count_remaining) != EXG_OK) throw std::runtime_error(exg_reader_error(ls->reader));
                    ls->counted = true;
                }
                if (ls->count_remaining == 0) {
                    if (!ClaimShard(data, gs, *ls)) return;
                    continue;
                }
                const idx_t n = ls->count_remaining < (uint64_t)D::VectorSize ? (idx_t)ls->count_remaining : (idx_t)D::VectorSize;
                ls->count_remaining -= n;
                D::SetCardinality(output, n);
                return;
            }
            auto buf = std::make_shared<ExonChunk>();
            buf->reader = ls->reader;
            if (exg_next_chunk(ls->reader, &buf->chunk) != EXG_OK) throw std::runtime_error(exg_reader_error(ls->reader));
            if (buf->chunk.n_rows == 0) {
                if (!ClaimShard(data, gs, *ls)) return;
                continue;
            }
            D::SetCardinality(output, (idx_t)buf->chunk.n_rows);
            for (size_t i = 0; i < gs.column_ids.size(); i++) {
                const idx_t col = gs.column_ids[i];
                if (col == D::RowId) continue;
                D::Reference(output, (idx_t)i, data.all_types.at(col), *buf->chunk.vectors[col], buf);
            }
            ls->batch_no = buf->chunk.batch_no;
            return;
        }
    }

    // TableFunction::cardinality (registered at module.cpp:307: ArrowTableFunction::ArrowScanCardinality, which returns a
    // NodeStatistics without an estimate — the reference cannot know: its input is an opaque Arrow stream).  The reader knows
    // the input's size on disk, so the planner gets an estimate here: bytes (x the usual deflate / zstd ratio of sequence text)
    // / the usual bytes per row of the format.  0 = unknown (no estimate is set, like the reference).  Only the join order
    // and the sink's sizing look at it; no result depends on it.
    static idx_t EstimatedCardinality(const BindData &d) {
        if (!d.input_bytes) return 0;
        const double text = (double)d.input_bytes * (d.input_compression ? 3.5 : 1.0);
        const double per_row = d.file_type == "fastq" ? 300.0 : d.file_type == "vcf" ? 120.0 : 4096.0;
        const double rows = text / per_row;
        return rows < 1 ? (idx_t)1 : (idx_t)rows;
    }

    // TableFunction::get_batch_index: the chunks of shard s come before those of shard s + 1, a shard's device batches in
    // their order; the chunks of one device batch share an index (they are one DuckDB batch).  Non-decreasing per scan
    // thread: a thread claims shards in increasing order.  64 shards x 2^24 stays below 1.1e9 (DuckDB's pipelines are 1e13 apart).
    static idx_t BatchIndex(const LocalState &ls) {
        const uint64_t b = ls.batch_no < ((uint64_t)1 << kBatchBits) ? ls.batch_no : ((uint64_t)1 << kBatchBits) - 1;
        return ((idx_t)ls.shard << kBatchBits) + (idx_t)b;
    }

    // module.cpp:320-382 on rust/src/arrow_reader.rs:173-197 (`replacement_scan`, same symbol as exon/include/rust.hpp:48):
    // the table function that replaces a bare 'file' reference, or "" when the name is not one of ours
    static std::string ReplacementFunction(const std::string &table_name) {
        std::string lower = table_name;
        for (char &c : lower) c = (char)tolower((unsigned char)c);
        const ReplacementScanResult res = replacement_scan(lower.c_str());
        if (!res.file_type) return "";
        const std::string ft = res.file_type;
        if (ft == "FASTA") return "read_fasta";
        if (ft == "FASTQ") return "read_fastq";
        if (ft == "VCF") return "read_vcf_file_records";
        return "";
    }
};

// quality_score_string_to_list (exon/src/exon/fastq_functions/module.cpp:28-54, registered at exon_extension.cpp:60): one INTEGER
// per byte of the string, `c - 33` with `char` signed as on x86-64.  The per-row arithmetic of the SQL scalar the shim registers
// (host loop: the function runs on whatever VARCHAR vector DuckDB hands it); exg_quality_score_list is the same op on a scan
// column that is still in HBM.
static inline void QualityScores(const char *s, size_t n, int32_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = (int32_t)(signed char)s[i] - 33;
}

// exon/src/exon_extension.cpp:47-58: the registrations of the path (+ the read_vcf alias the north star names)
struct Registration {
    const char *name, *file_type;
};
static const Registration kRegistrations[] = {
    {"read_fasta", "fasta"}, {"read_fastq", "fastq"}, {"read_vcf_file_records", "vcf"}, {"read_vcf", "vcf"}};

}  // namespace exon_scan
