// duckdb_shim/exon_extension.cpp — the REAL DuckDB v0.8.1 binding of libexon_gpu.so.
//
// Compiled only where DuckDB's headers exist (the reference's `duckdb/` submodule is empty and no
// DuckDB header is present on the build box, so this file is not part of build()).  It is the
// mechanical mapping of csrc/exon_table_function.cpp (which runs against csrc/duck_mini.hpp and is
// what the tests drive) onto duckdb::TableFunction, and replaces, in the reference tree,
//   exon/src/exon/arrow_table_function/module.cpp   (Register / FileTypeBind / InitGlobal / Scan)
//   exon/src/exon_extension.cpp:47-58,79            (registrations of the three formats + replacement scan)
// Build (out-of-tree extension, like the reference's CMakeLists.txt:131-146, minus Rust/Corrosion):
//   c++ -std=c++17 -fPIC -shared -DDUCKDB_BUILD_LOADABLE_EXTENSION -I<duckdb>/src/include \
//       -I../include exon_extension.cpp -L../exon_duckdb_amd/lib -lexon_gpu -o exon.duckdb_extension
#define DUCKDB_EXTENSION_MAIN
#include "duckdb.hpp"
#include "duckdb/common/types/vector_buffer.hpp"
#include "duckdb/function/table_function.hpp"
#include "duckdb/main/extension_util.hpp"
#include "duckdb/parser/expression/constant_expression.hpp"
#include "duckdb/parser/expression/function_expression.hpp"
#include "duckdb/parser/parsed_data/create_table_function_info.hpp"
#include "duckdb/parser/tableref/table_function_ref.hpp"
#include "duckdb/planner/filter/conjunction_filter.hpp"
#include "duckdb/planner/filter/constant_filter.hpp"
#include "duckdb/planner/table_filter.hpp"

#include "exon_gpu.h"

namespace exon {
using namespace duckdb;

static_assert(sizeof(string_t) == sizeof(exg_string_t), "exg_string_t must be duckdb::string_t");
static_assert(STANDARD_VECTOR_SIZE == EXG_VECTOR_SIZE, "chunks are STANDARD_VECTOR_SIZE rows");

struct WTArrowTableScanInfo : public TableFunctionInfo {
	explicit WTArrowTableScanInfo(string file_type_p) : file_type(std::move(file_type_p)) {
	}
	string file_type;
};

struct ExonScanFunctionData : public TableFunctionData {
	string file_type, compression, file_name;
	vector<LogicalType> all_types;
	vector<string> all_names;
};

struct ExonScanGlobalState : public GlobalTableFunctionState {
	exg_reader *reader = nullptr;
	vector<column_t> column_ids;
	bool count_only = false, counted = false;
	uint64_t count_remaining = 0;
	~ExonScanGlobalState() override {
		if (reader) {
			exg_close(reader);
		}
	}
	idx_t MaxThreads() const override {
		return 1;
	}
};

// keeps the engine chunk (vectors + pinned payload) alive as long as a Vector references it
struct ExonChunkBuffer : public VectorBuffer {
	ExonChunkBuffer(exg_reader *r, exg_chunk c) : VectorBuffer(VectorBufferType::OPAQUE_BUFFER), reader(r), chunk(c) {
	}
	~ExonChunkBuffer() override {
		exg_release_chunk(reader, &chunk);
	}
	exg_reader *reader;
	exg_chunk chunk;
};

// the reference's FilterToString (module.cpp:158-214), unchanged in what it renders
static string FilterToString(const TableFilter &filter, const string &column_name) {
	switch (filter.filter_type) {
	case TableFilterType::CONSTANT_COMPARISON: {
		auto &cf = (const ConstantFilter &)filter;
		return column_name + ExpressionTypeToOperator(cf.comparison_type) + cf.constant.ToSQLString();
	}
	case TableFilterType::CONJUNCTION_AND: {
		vector<string> parts;
		for (auto &c : ((const ConjunctionAndFilter &)filter).child_filters) {
			parts.push_back(FilterToString(*c, column_name));
		}
		return StringUtil::Join(parts, " AND ");
	}
	case TableFilterType::CONJUNCTION_OR: {
		vector<string> parts;
		for (auto &c : ((const ConjunctionOrFilter &)filter).child_filters) {
			parts.push_back(FilterToString(*c, column_name));
		}
		return StringUtil::Join(parts, " OR ");
	}
	case TableFilterType::IS_NOT_NULL:
		return column_name + " IS NOT NULL";
	case TableFilterType::IS_NULL:
		return column_name + " IS NULL";
	default:
		throw NotImplementedException("FilterToString: filter type not implemented");
	}
}

static exg_reader *OpenReader(const ExonScanFunctionData &d, const string &filter_clause = "") {
	exg_open_args a {};
	a.filters = filter_clause.empty() ? nullptr : filter_clause.c_str(); // evaluated on the device
	a.path = d.file_name.c_str();
	a.file_format = d.file_type.c_str();
	a.compression = d.compression == "auto_detect" ? nullptr : d.compression.c_str();
	a.batch_rows = STANDARD_VECTOR_SIZE;
	exg_reader *r = nullptr;
	if (exg_open(&a, &r) != EXG_OK) {
		throw std::runtime_error(exg_last_error_message());
	}
	return r;
}

static LogicalType ToLogical(int t) {
	return t == EXG_TYPE_BIGINT ? LogicalType::BIGINT : t == EXG_TYPE_FLOAT ? LogicalType::FLOAT : LogicalType::VARCHAR;
}

static unique_ptr<FunctionData> FileTypeBind(ClientContext &, TableFunctionBindInput &input,
                                             vector<LogicalType> &return_types, vector<string> &names) {
	auto &info = input.info->Cast<WTArrowTableScanInfo>();
	auto result = make_uniq<ExonScanFunctionData>();
	result->file_name = input.inputs[0].GetValue<string>();
	result->compression = "auto_detect";
	for (auto &kv : input.named_parameters) {
		if (kv.first == "compression") {
			result->compression = kv.second.GetValue<string>();
		}
	}
	result->file_type = info.file_type;
	exg_reader *r = OpenReader(*result);
	exg_schema sch;
	int rc = exg_schema_of(r, &sch);
	exg_close(r);
	if (rc != EXG_OK) {
		throw std::runtime_error("Failed to get schema");
	}
	for (int i = 0; i < sch.n_columns; i++) {
		return_types.push_back(ToLogical(sch.types[i]));
		names.emplace_back(sch.names[i]);
	}
	result->all_types = return_types;
	result->all_names = names;
	return std::move(result);
}

static unique_ptr<GlobalTableFunctionState> InitGlobal(ClientContext &, TableFunctionInitInput &input) {
	auto &data = input.bind_data->Cast<ExonScanFunctionData>();
	auto gs = make_uniq<ExonScanGlobalState>();
	gs->column_ids = input.column_ids;
	gs->count_only = true;
	for (auto c : input.column_ids) {
		gs->count_only = gs->count_only && c == COLUMN_IDENTIFIER_ROW_ID;
	}
	string filter_clause;
	if (input.filters) { // module.cpp:222-226
		vector<string> parts;
		for (auto &f : input.filters->filters) {
			parts.push_back(FilterToString(*f.second, data.all_names[input.column_ids[f.first]]));
		}
		filter_clause = StringUtil::Join(parts, " AND ");
	}
	gs->reader = OpenReader(data, filter_clause);
	return std::move(gs);
}

static void Scan(ClientContext &, TableFunctionInput &input, DataChunk &output) {
	auto &gs = input.global_state->Cast<ExonScanGlobalState>();
	if (gs.count_only) {
		if (!gs.counted) {
			if (exg_count_only(gs.reader, &gs.count_remaining) != EXG_OK) {
				throw std::runtime_error(exg_reader_error(gs.reader));
			}
			gs.counted = true;
		}
		idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, gs.count_remaining);
		gs.count_remaining -= n;
		output.SetCardinality(n);
		return;
	}
	exg_chunk c;
	if (exg_next_chunk(gs.reader, &c) != EXG_OK) {
		throw std::runtime_error(exg_reader_error(gs.reader));
	}
	if (c.n_rows == 0) {
		return; // output.size() == 0 ends the scan
	}
	auto buffer = make_buffer<ExonChunkBuffer>(gs.reader, c);
	output.SetCardinality(c.n_rows);
	for (idx_t i = 0; i < gs.column_ids.size(); i++) {
		auto col = gs.column_ids[i];
		if (col == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		auto &vec = output.data[i];
		FlatVector::SetData(vec, data_ptr_cast(c.data[col])); // zero-copy: string_t array of the engine
		vec.SetAuxiliary(buffer);
		if (c.validity[col]) {
			FlatVector::Validity(vec).Initialize(reinterpret_cast<validity_t *>(c.validity[col]));
		}
	}
}

static void Register(const string &name, const string &file_type, DatabaseInstance &db) {
	TableFunction scan(name, {LogicalType::VARCHAR}, Scan, FileTypeBind, InitGlobal);
	scan.function_info = make_shared<WTArrowTableScanInfo>(file_type);
	scan.named_parameters["compression"] = LogicalType::VARCHAR;
	scan.projection_pushdown = true;
	scan.filter_pushdown = true; // like the reference (module.cpp:311); the predicate runs on the device
	ExtensionUtil::RegisterFunction(db, scan);
}

static unique_ptr<TableRef> ReplacementScan(ClientContext &, const string &table_name, ReplacementScanData *) {
	auto lower = StringUtil::Lower(table_name);
	auto res = replacement_scan(lower.c_str()); // same symbol and struct as exon/include/rust.hpp:11-13,48
	if (!res.file_type) {
		return nullptr;
	}
	string ft(res.file_type), fn;
	if (ft == "FASTA") {
		fn = "read_fasta";
	} else if (ft == "FASTQ") {
		fn = "read_fastq";
	} else if (ft == "VCF") {
		fn = "read_vcf_file_records";
	} else {
		return nullptr;
	}
	auto ref = make_uniq<TableFunctionRef>();
	vector<unique_ptr<ParsedExpression>> children;
	children.push_back(make_uniq<ConstantExpression>(Value(table_name)));
	ref->function = make_uniq<FunctionExpression>(fn, std::move(children));
	return std::move(ref);
}

static void LoadInternal(DatabaseInstance &db) {
	Register("read_fasta", "fasta", db);
	Register("read_fastq", "fastq", db);
	Register("read_vcf_file_records", "vcf", db);
	Register("read_vcf", "vcf", db);
	DBConfig::GetConfig(db).replacement_scans.emplace_back(ReplacementScan);
}
} // namespace exon

extern "C" {
DUCKDB_EXTENSION_API void exon_init(duckdb::DatabaseInstance &db) {
	exon::LoadInternal(db);
}
DUCKDB_EXTENSION_API const char *exon_version() {
	return duckdb::DuckDB::LibraryVersion();
}
}
