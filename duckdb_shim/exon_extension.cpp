// duckdb_shim/exon_extension.cpp — the REAL DuckDB v0.8.1 binding of libexon_gpu.so: `LOAD exon`.
//
// All of the glue's logic — bind / init_global (shard plan, FilterToString) / init_local (one reader per shard and device) /
// scan / batch index / replacement scan — is exon_duckdb_amd/csrc/exon_table_function.hpp, written once against a traits
// struct.  This file is that header instantiated over DuckDB's own classes: the traits (how a LogicalType and a Vector are
// made in DuckDB), thin wrappers with DuckDB's callback signatures, and the registrations.  The same header instantiated
// over csrc/testing/duck_mini.hpp is what the GPU test-suite drives (csrc/testing/exon_tf_harness.cpp), so the logic is
// compiled and tested on the build box; THIS file needs DuckDB's headers, which do not exist there (the reference's
// `duckdb/` submodule is empty), and is therefore not part of build().  It replaces, in the reference tree,
//   exon/src/exon/arrow_table_function/module.cpp   (Register / FileTypeBind / InitGlobal / Scan / ReplacementScan)
//   exon/src/exon_extension.cpp:47-58,79            (registrations of the three formats + replacement scan)
// Build (out-of-tree extension, like the reference's CMakeLists.txt:131-146, minus Rust/Corrosion):
//   c++ -std=c++17 -fPIC -shared -DDUCKDB_BUILD_LOADABLE_EXTENSION -I<duckdb>/src/include -I../include
//       -I../exon_duckdb_amd/csrc exon_extension.cpp -L../exon_duckdb_amd/lib -lexon_gpu -o exon.duckdb_extension
// tests/test_host_logic.py puts this file through `c++ -fsyntax-only` against declaration-only stand-ins of the ten DuckDB
// headers it includes (tests/duckdb_stub/: signatures as in v0.8.1): every template instantiation of the glue over DuckDB's
// types is compiled on the build box.
#define DUCKDB_EXTENSION_MAIN
#include "duckdb.hpp"
#include "duckdb/common/types/vector_buffer.hpp"
#include "duckdb/function/scalar_function.hpp"
#include "duckdb/function/table_function.hpp"
#include "duckdb/main/extension_util.hpp"
#include "duckdb/parser/expression/constant_expression.hpp"
#include "duckdb/parser/expression/function_expression.hpp"
#include "duckdb/parser/tableref/table_function_ref.hpp"
#include "duckdb/planner/filter/conjunction_filter.hpp"
#include "duckdb/planner/filter/constant_filter.hpp"
#include "duckdb/planner/table_filter.hpp"
#include "duckdb/storage/statistics/node_statistics.hpp"

#include "exon_table_function.hpp"

namespace exon {
using namespace duckdb;

static_assert(sizeof(string_t) == sizeof(exg_string_t), "exg_string_t must be duckdb::string_t");
static_assert(sizeof(list_entry_t) == sizeof(exg_list_entry_t), "exg_list_entry_t must be duckdb::list_entry_t");
static_assert(STANDARD_VECTOR_SIZE == EXG_VECTOR_SIZE, "chunks are STANDARD_VECTOR_SIZE rows");

// keeps the engine chunk (vectors + pinned payload) alive as long as a Vector references it
struct ExonChunkBuffer : public VectorBuffer {
	explicit ExonChunkBuffer(std::shared_ptr<exon_scan::ExonChunk> keep_p)
	    : VectorBuffer(VectorBufferType::OPAQUE_BUFFER), keep(std::move(keep_p)) {
	}
	std::shared_ptr<exon_scan::ExonChunk> keep;
};

struct RealDuck {
	using idx_t = duckdb::idx_t;
	using LogicalType = duckdb::LogicalType;
	using DataChunk = duckdb::DataChunk;
	using FunctionData = duckdb::TableFunctionData;
	using GlobalTableFunctionState = duckdb::GlobalTableFunctionState;
	using LocalTableFunctionState = duckdb::LocalTableFunctionState;
	using TableFilter = duckdb::TableFilter;
	using ConstantFilter = duckdb::ConstantFilter;
	using ConjunctionFilter = duckdb::ConjunctionFilter; // base of ConjunctionAndFilter / ConjunctionOrFilter
	using TableFilterSet = duckdb::TableFilterSet;
	using TableFilterType = duckdb::TableFilterType;
	static constexpr idx_t RowId = COLUMN_IDENTIFIER_ROW_ID;
	static constexpr idx_t VectorSize = STANDARD_VECTOR_SIZE;

	// module.cpp:126-147 (there: ArrowTableFunction::GetArrowLogicalType over the Arrow schema)
	static LogicalType ToLogical(const exg_type &t) {
		switch (t.type) {
		case EXG_TYPE_BIGINT:
			return LogicalType::BIGINT;
		case EXG_TYPE_FLOAT:
			return LogicalType::FLOAT;
		case EXG_TYPE_INTEGER:
			return LogicalType::INTEGER;
		case EXG_TYPE_BOOLEAN:
			return LogicalType::BOOLEAN;
		case EXG_TYPE_LIST:
			return LogicalType::LIST(ToLogical(t.children[0]));
		case EXG_TYPE_STRUCT: {
			child_list_t<LogicalType> fields;
			for (int i = 0; i < t.n_children; i++) {
				fields.emplace_back(t.children[i].name, ToLogical(t.children[i]));
			}
			return LogicalType::STRUCT(std::move(fields));
		}
		default:
			return LogicalType::VARCHAR;
		}
	}

	// an engine vector as a DuckDB Vector that references the engine's host buffers (zero-copy)
	static void Wrap(Vector &vec, const LogicalType &type, const exg_vector &src, const buffer_ptr<VectorBuffer> &keep) {
		if (src.validity) {
			FlatVector::Validity(vec).Initialize(reinterpret_cast<validity_t *>(src.validity));
		}
		switch (type.id()) {
		case LogicalTypeId::LIST: {
			FlatVector::SetData(vec, data_ptr_cast(src.data)); // list_entry_t[], offsets relative to this chunk's child
			auto &child = ListVector::GetEntry(vec);
			Wrap(child, ListType::GetChildType(type), src.children[0], keep);
			ListVector::SetListSize(vec, src.children[0].length);
			break;
		}
		case LogicalTypeId::STRUCT: {
			auto &entries = StructVector::GetEntries(vec);
			auto &fields = StructType::GetChildTypes(type);
			for (idx_t i = 0; i < entries.size(); i++) {
				Wrap(*entries[i], fields[i].second, src.children[i], keep);
			}
			break;
		}
		case LogicalTypeId::VARCHAR:
			FlatVector::SetData(vec, data_ptr_cast(src.data)); // string_t[]; payload kept alive through the aux buffer
			vec.SetAuxiliary(keep);
			break;
		default:
			FlatVector::SetData(vec, data_ptr_cast(src.data));
			vec.SetAuxiliary(keep);
			break;
		}
	}
	static void Reference(DataChunk &out, idx_t col, const LogicalType &type, const exg_vector &src,
	                      std::shared_ptr<exon_scan::ExonChunk> keep) {
		Wrap(out.data[col], type, src, make_buffer<ExonChunkBuffer>(std::move(keep)));
	}
	static void SetCardinality(DataChunk &out, idx_t n) {
		out.SetCardinality(n);
	}
	static std::string ComparisonOperator(const ConstantFilter &f) {
		return ExpressionTypeToOperator(f.comparison_type);
	}
	static std::string ConstantSQL(const ConstantFilter &f) {
		return f.constant.ToSQLString();
	}
};

using TF = exon_scan::ExonTableFunction<RealDuck>;

// exon/include/exon/arrow_table_function/module.hpp:29-35
struct WTArrowTableScanInfo : public TableFunctionInfo {
	explicit WTArrowTableScanInfo(string file_type_p) : file_type(std::move(file_type_p)) {
	}
	string file_type;
};

static unique_ptr<FunctionData> FileTypeBind(ClientContext &, TableFunctionBindInput &input,
                                             vector<LogicalType> &return_types, vector<string> &names) {
	auto &info = input.info->Cast<WTArrowTableScanInfo>();
	string compression;
	for (auto &kv : input.named_parameters) {
		if (kv.first == "compression") {
			compression = kv.second.GetValue<string>();
		}
	}
	std::vector<LogicalType> types;
	std::vector<std::string> col_names;
	auto result = TF::Bind(input.inputs[0].GetValue<string>(), compression, info.file_type, types, col_names);
	for (auto &t : types) {
		return_types.push_back(t);
	}
	for (auto &n : col_names) {
		names.push_back(n);
	}
	return unique_ptr<FunctionData>(result.release());
}

static unique_ptr<GlobalTableFunctionState> InitGlobal(ClientContext &, TableFunctionInitInput &input) {
	auto &data = input.bind_data->Cast<TF::BindData>();
	std::vector<idx_t> column_ids(input.column_ids.begin(), input.column_ids.end());
	return unique_ptr<GlobalTableFunctionState>(TF::InitGlobal(data, column_ids, input.filters.get()).release());
}

static unique_ptr<LocalTableFunctionState> InitLocal(ExecutionContext &, TableFunctionInitInput &input,
                                                     GlobalTableFunctionState *gs) {
	return unique_ptr<LocalTableFunctionState>(
	    TF::InitLocal(input.bind_data->Cast<TF::BindData>(), gs->Cast<TF::GlobalState>()).release());
}

static void Scan(ClientContext &, TableFunctionInput &input, DataChunk &output) {
	if (!input.local_state) { // module.cpp:259-261
		return;
	}
	TF::Scan(input.bind_data->Cast<TF::BindData>(), input.global_state->Cast<TF::GlobalState>(),
	         &input.local_state->Cast<TF::LocalState>(), output);
}

static idx_t GetBatchIndex(ClientContext &, const FunctionData *, LocalTableFunctionState *ls, GlobalTableFunctionState *) {
	return TF::BatchIndex(ls->Cast<TF::LocalState>());
}

// module.cpp:307 (there: ArrowTableFunction::ArrowScanCardinality = a NodeStatistics without an estimate)
static unique_ptr<NodeStatistics> Cardinality(ClientContext &, const FunctionData *bind_data) {
	const idx_t rows = TF::EstimatedCardinality(bind_data->Cast<TF::BindData>());
	return rows ? make_uniq<NodeStatistics>(rows) : make_uniq<NodeStatistics>();
}

// module.cpp:296-318
static void Register(const string &name, const string &file_type, DatabaseInstance &db) {
	TableFunction scan(name, {LogicalType::VARCHAR}, Scan, FileTypeBind, InitGlobal, InitLocal);
	scan.function_info = make_shared<WTArrowTableScanInfo>(file_type);
	scan.named_parameters["compression"] = LogicalType::VARCHAR;
	scan.cardinality = Cardinality;
	scan.get_batch_index = GetBatchIndex;
	scan.projection_pushdown = true;
	scan.filter_pushdown = true; // like the reference (module.cpp:311); the predicate runs on the device
	ExtensionUtil::RegisterFunction(db, scan);
}

// module.cpp:320-382
static unique_ptr<TableRef> ReplacementScan(ClientContext &, const string &table_name, ReplacementScanData *) {
	const string fn = TF::ReplacementFunction(table_name);
	if (fn.empty()) {
		return nullptr;
	}
	auto ref = make_uniq<TableFunctionRef>();
	vector<unique_ptr<ParsedExpression>> children;
	children.push_back(make_uniq<ConstantExpression>(Value(table_name)));
	ref->function = make_uniq<FunctionExpression>(fn, std::move(children));
	return std::move(ref);
}

// fastq_functions/module.cpp:28-54: quality_score_string_to_list(VARCHAR) -> LIST(INTEGER), one value per byte, c - 33.
// The reference goes through Value / SetValue per row; this writes the list entries and the child vector directly.  A NULL
// string gives a NULL list (the reference's GetValue path has no defined answer for it: StringValue::Get of a NULL).
static void QualityScoreStringToList(DataChunk &args, ExpressionState &, Vector &result) {
	const idx_t count = args.size();
	UnifiedVectorFormat in;
	args.data[0].ToUnifiedFormat(count, in);
	auto strings = reinterpret_cast<const string_t *>(in.data);
	result.SetVectorType(VectorType::FLAT_VECTOR);
	auto entries = FlatVector::GetData<list_entry_t>(result);
	idx_t total = 0;
	for (idx_t i = 0; i < count; i++) {
		const idx_t k = in.sel->get_index(i);
		total += in.validity.RowIsValid(k) ? strings[k].GetSize() : 0;
	}
	ListVector::Reserve(result, total);
	auto values = FlatVector::GetData<int32_t>(ListVector::GetEntry(result));
	idx_t at = 0;
	for (idx_t i = 0; i < count; i++) {
		const idx_t k = in.sel->get_index(i);
		if (!in.validity.RowIsValid(k)) {
			FlatVector::SetNull(result, i, true);
			entries[i].offset = at;
			entries[i].length = 0;
			continue;
		}
		const idx_t n = strings[k].GetSize();
		exon_scan::QualityScores(strings[k].GetData(), n, values + at);
		entries[i].offset = at;
		entries[i].length = n;
		at += n;
	}
	ListVector::SetListSize(result, total);
}

// exon/src/exon_extension.cpp:25-96, restricted to the path
static void LoadInternal(DatabaseInstance &db) {
	for (const auto &reg : exon_scan::kRegistrations) {
		Register(reg.name, reg.file_type, db);
	}
	// exon_extension.cpp:60 (FastqFunctions::GetQualityScoreStringToList)
	ExtensionUtil::RegisterFunction(db, ScalarFunction("quality_score_string_to_list", {LogicalType::VARCHAR},
	                                                   LogicalType::LIST(LogicalType::INTEGER), QualityScoreStringToList));
	DBConfig::GetConfig(db).replacement_scans.emplace_back(ReplacementScan);
}
} // namespace exon

extern "C" {
DUCKDB_EXTENSION_API void exon_init(duckdb::DatabaseInstance &db) { // exon_extension.cpp:110-116
	exon::LoadInternal(db);
}
DUCKDB_EXTENSION_API const char *exon_version() { // exon_extension.cpp:118-122
	return duckdb::DuckDB::LibraryVersion();
}
}
