// exon_table_function.cpp — host side of the drop-in: the reference's WTArrowTableFunction
// (exon/src/exon/arrow_table_function/module.cpp, exon/include/exon/arrow_table_function/module.hpp)
// re-designed on top of the reader-level C-ABI.  Same registration names, same bind / init_global /
// init_local / scan lifecycle, same named parameter, same error behaviour (C++ exceptions, which
// DuckDB turns into query errors); no Arrow hop: Scan fills the DataChunk's vectors directly with
// references to the engine's host buffers (zero-copy, kept alive by the Vector's buffer).
//
// Deliberate differences from the reference, all listed in SURVEY.md Appendix B / §7.2:
//   * bind does not open a second full reader and leak it (module.cpp:82-155);
//   * filter_pushdown is false: DuckDB applies filters above the scan, results are identical
//     (the reference renders them to SQL for DataFusion, module.cpp:158-214);
//   * COUNT(*) (only COLUMN_IDENTIFIER_ROW_ID projected) never materialises a column.
#include <string.h>

#include <algorithm>
#include <stdexcept>

#include "duck_mini.hpp"

namespace exon_amd {

// exon/include/exon/arrow_table_function/module.hpp:29-35
struct WTArrowTableScanInfo : public TableFunctionInfo {
    explicit WTArrowTableScanInfo(std::string file_type_p) : file_type(std::move(file_type_p)) {}
    std::string file_type;
};

// module.cpp:29-41
struct ExonScanFunctionData : public FunctionData {
    std::string file_type;
    std::string compression;  // "auto_detect" when the named parameter is absent (module.cpp:85)
    std::string file_name;
    std::vector<std::string> all_names;
    std::vector<LogicalType> all_types;
};

struct ExonScanGlobalState : public GlobalTableFunctionState {
    exg_reader *reader = nullptr;
    std::vector<idx_t> column_ids;
    bool count_only = false;
    uint64_t count_remaining = 0;
    bool counted = false;
    ~ExonScanGlobalState() override {
        if (reader) exg_close(reader);
    }
    idx_t MaxThreads() const override { return 1; }  // like the reference's ArrowScanGlobalState
};

struct ExonScanLocalState : public LocalTableFunctionState {};

struct ChunkBuffer {  // VectorBuffer: releases the engine chunk when the last Vector drops it
    exg_reader *reader;
    exg_chunk chunk;
    ~ChunkBuffer() { exg_release_chunk(reader, &chunk); }
};

static exg_reader *open_reader(const ExonScanFunctionData &d) {
    exg_open_args a;
    memset(&a, 0, sizeof a);
    a.path = d.file_name.c_str();
    a.file_format = d.file_type.c_str();
    a.compression = d.compression == "auto_detect" ? nullptr : d.compression.c_str();  // module.cpp:95-103
    a.batch_rows = STANDARD_VECTOR_SIZE;                                               // module.cpp:83
    exg_reader *r = nullptr;
    if (exg_open(&a, &r) != EXG_OK) throw std::runtime_error(exg_last_error_message());  // module.cpp:105-108
    return r;
}

struct WTArrowTableFunction {
    // module.cpp:75-156
    static std::unique_ptr<FunctionData> FileTypeBind(TableFunctionBindInput &input, std::vector<LogicalType> &return_types,
                                                      std::vector<std::string> &names) {
        auto &info = static_cast<const WTArrowTableScanInfo &>(*input.info);
        auto result = std::make_unique<ExonScanFunctionData>();
        result->file_name = input.inputs.at(0);
        result->compression = "auto_detect";
        for (auto &kv : input.named_parameters)
            if (kv.first == "compression") result->compression = kv.second;
        result->file_type = info.file_type;
        // the reference learns the schema by opening a reader; so do we (and we close it again)
        exg_reader *r = open_reader(*result);
        exg_schema sch;
        int rc = exg_schema_of(r, &sch);
        exg_close(r);
        if (rc != EXG_OK) throw std::runtime_error("Failed to get schema");  // module.cpp:112-119
        for (int i = 0; i < sch.n_columns; i++) {
            return_types.push_back(LogicalType{(LogicalTypeId)sch.types[i]});
            names.emplace_back(sch.names[i]);
        }
        result->all_names = names;
        result->all_types = return_types;
        return result;
    }

    // module.cpp:216-255
    static std::unique_ptr<GlobalTableFunctionState> InitGlobal(TableFunctionInitInput &input) {
        auto &data = static_cast<const ExonScanFunctionData &>(*input.bind_data);
        auto gs = std::make_unique<ExonScanGlobalState>();
        gs->column_ids = input.column_ids;
        gs->count_only = std::all_of(input.column_ids.begin(), input.column_ids.end(),
                                     [](idx_t c) { return c == COLUMN_IDENTIFIER_ROW_ID; });
        gs->reader = open_reader(data);
        return gs;
    }

    static std::unique_ptr<LocalTableFunctionState> InitLocal(TableFunctionInitInput &, GlobalTableFunctionState *) {
        return std::make_unique<ExonScanLocalState>();
    }

    // module.cpp:257-294: leaves output.size() == 0 at the end of the stream
    static void Scan(TableFunctionInput &input, DataChunk &output) {
        if (!input.local_state) return;
        auto &gs = static_cast<ExonScanGlobalState &>(*input.global_state);
        auto &data = static_cast<const ExonScanFunctionData &>(*input.bind_data);
        output.Reset();
        if (gs.count_only) {
            if (!gs.counted) {
                if (exg_count_only(gs.reader, &gs.count_remaining) != EXG_OK)
                    throw std::runtime_error(exg_reader_error(gs.reader));
                gs.counted = true;
            }
            idx_t n = std::min<idx_t>(STANDARD_VECTOR_SIZE, gs.count_remaining);
            gs.count_remaining -= n;
            output.data.resize(gs.column_ids.size());
            output.SetCardinality(n);
            return;
        }
        auto buf = std::make_shared<ChunkBuffer>();
        buf->reader = gs.reader;
        if (exg_next_chunk(gs.reader, &buf->chunk) != EXG_OK) throw std::runtime_error(exg_reader_error(gs.reader));
        if (buf->chunk.n_rows == 0) return;
        output.SetCardinality(buf->chunk.n_rows);
        for (idx_t col : gs.column_ids) {
            Vector v;
            if (col != COLUMN_IDENTIFIER_ROW_ID) {
                v.type = data.all_types.at(col);
                v.data = buf->chunk.data[col];
                v.validity = buf->chunk.validity[col];
                v.buffer = buf;
            }
            output.data.push_back(std::move(v));
        }
    }

    // module.cpp:296-318
    static void Register(const std::string &name, const std::string &file_type, Catalog &catalog) {
        TableFunction scan;
        scan.name = name;
        scan.arguments = {LogicalType{LogicalTypeId::VARCHAR}};
        scan.function = Scan;
        scan.bind = FileTypeBind;
        scan.init_global = InitGlobal;
        scan.init_local = InitLocal;
        scan.function_info = std::make_shared<WTArrowTableScanInfo>(file_type);
        scan.named_parameters["compression"] = LogicalType{LogicalTypeId::VARCHAR};
        scan.projection_pushdown = true;
        scan.filter_pushdown = false;
        catalog.CreateTableFunction(scan);
    }

    // module.cpp:320-382 + rust/src/arrow_reader.rs:173-197.  Returns the table function that
    // replaces a bare 'file' reference, or "" when the name is not one of ours.
    static std::string ReplacementScan(const std::string &table_name) {
        std::string lower = table_name;
        for (char &c : lower) c = (char)tolower((unsigned char)c);
        auto ext_of = [](const std::string &s, size_t end) {
            size_t dot = s.rfind('.', end == std::string::npos ? end : end - 1);
            return dot == std::string::npos ? std::make_pair(s.substr(0, end), (size_t)0)
                                            : std::make_pair(s.substr(dot + 1, (end == std::string::npos ? s.size() : end) - dot - 1), dot);
        };
        auto e1 = ext_of(lower, std::string::npos);
        std::string ext = e1.first;
        static const char *compressed[] = {"gz", "gzip", "zst", "zstd", "bz2", "bzip2", "xz"};
        if (std::find_if(std::begin(compressed), std::end(compressed), [&](const char *c) { return ext == c; }) !=
                std::end(compressed) &&
            e1.second > 0)
            ext = ext_of(lower, e1.second).first;
        if (ext == "fasta" || ext == "fa" || ext == "fna") return "read_fasta";
        if (ext == "fastq" || ext == "fq") return "read_fastq";
        if (ext == "vcf") return "read_vcf_file_records";
        return "";
    }
};

// exon/src/exon_extension.cpp:25-96, restricted to the path: the three table functions (+ the
// read_vcf alias the north star names) and the replacement scan.
void LoadInternal(Catalog &catalog) {
    WTArrowTableFunction::Register("read_fasta", "fasta", catalog);
    WTArrowTableFunction::Register("read_fastq", "fastq", catalog);
    WTArrowTableFunction::Register("read_vcf_file_records", "vcf", catalog);
    WTArrowTableFunction::Register("read_vcf", "vcf", catalog);
}

}  // namespace exon_amd

// ---- C entry points used by the Python parity tests (tests/test_table_function*.py) -------------------
using namespace exon_amd;

struct exon_tf_handle {
    const TableFunction *fn = nullptr;
    std::unique_ptr<FunctionData> bind_data;
    std::unique_ptr<GlobalTableFunctionState> global;
    std::unique_ptr<LocalTableFunctionState> local;
    std::vector<LogicalType> types;
    std::vector<std::string> names;
    DataChunk chunk;
};

namespace exg {
void set_error(const char *fmt, ...);
}

static Catalog &the_catalog() {
    static Catalog c = [] {
        Catalog x;
        LoadInternal(x);
        return x;
    }();
    return c;
}

extern "C" int exon_tf_catalog_has(const char *name) { return the_catalog().GetTableFunction(name) != nullptr; }

extern "C" int exon_tf_bind(const char *fn_name, const char *path, const char *compression, exon_tf_handle **out) {
    *out = nullptr;
    const TableFunction *fn = the_catalog().GetTableFunction(fn_name);
    if (!fn) {
        exg::set_error("Catalog Error: Table Function with name %s does not exist!", fn_name);
        return EXG_E_INVALID_ARG;
    }
    auto h = std::make_unique<exon_tf_handle>();
    h->fn = fn;
    TableFunctionBindInput in;
    in.inputs.push_back(path);
    if (compression) in.named_parameters["compression"] = compression;
    in.info = fn->function_info.get();
    try {
        h->bind_data = fn->bind(in, h->types, h->names);
    } catch (const std::exception &e) {
        exg::set_error("%s", e.what());
        return EXG_E_IO;
    }
    *out = h.release();
    return EXG_OK;
}

extern "C" int exon_tf_schema(exon_tf_handle *h, exg_schema *out) {
    memset(out, 0, sizeof *out);
    out->n_columns = (int)h->names.size();
    for (int i = 0; i < out->n_columns; i++) {
        out->names[i] = h->names[i].c_str();
        out->types[i] = (int)h->types[i].id;
    }
    return EXG_OK;
}

extern "C" int exon_tf_init(exon_tf_handle *h, const uint64_t *column_ids, int n) {
    TableFunctionInitInput in;
    in.bind_data = h->bind_data.get();
    in.column_ids.assign(column_ids, column_ids + n);
    try {
        h->global = h->fn->init_global(in);
        h->local = h->fn->init_local(in, h->global.get());
    } catch (const std::exception &e) {
        exg::set_error("%s", e.what());
        return EXG_E_IO;
    }
    return EXG_OK;
}

// One call of TableFunction::function.  n_rows == 0 => end.  data/validity follow column_ids order.
extern "C" int exon_tf_scan(exon_tf_handle *h, exg_chunk *out) {
    memset(out, 0, sizeof *out);
    TableFunctionInput in;
    in.bind_data = h->bind_data.get();
    in.local_state = h->local.get();
    in.global_state = h->global.get();
    try {
        h->fn->function(in, h->chunk);
    } catch (const std::exception &e) {
        exg::set_error("%s", e.what());
        return EXG_E_PARSE;
    }
    out->n_rows = h->chunk.size();
    out->n_columns = (int)h->chunk.data.size();
    for (int i = 0; i < out->n_columns && i < 16; i++) {
        out->data[i] = h->chunk.data[i].data;
        out->validity[i] = h->chunk.data[i].validity;
    }
    return EXG_OK;
}

extern "C" void exon_tf_close(exon_tf_handle *h) { delete h; }

extern "C" int exon_replacement_scan(const char *table_name, char *out_fn, size_t cap) {
    std::string f = WTArrowTableFunction::ReplacementScan(table_name);
    if (f.empty() || f.size() + 1 > cap) return 0;
    memcpy(out_fn, f.c_str(), f.size() + 1);
    return 1;
}

// exon/include/rust.hpp:48 — same symbol, same result struct (file type upper-cased, NULL when unknown)
extern "C" ReplacementScanResult replacement_scan(const char *uri) {
    ReplacementScanResult r;
    r.file_type = nullptr;
    if (!uri) return r;
    std::string f = WTArrowTableFunction::ReplacementScan(uri);
    if (f == "read_fasta") r.file_type = "FASTA";
    if (f == "read_fastq") r.file_type = "FASTQ";
    if (f == "read_vcf_file_records") r.file_type = "VCF";
    return r;
}
