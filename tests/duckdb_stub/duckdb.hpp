// tests/duckdb_stub/duckdb.hpp — TEST INFRASTRUCTURE, not DuckDB.
//
// Declaration-only stand-ins for the slice of DuckDB v0.8.1's C++ API that duckdb_shim/exon_extension.cpp touches, with
// the signatures as that release declares them (recalled: DuckDB's headers do not exist on the build box — the reference's
// `duckdb/` submodule is empty).  tests/test_host_logic.py runs `c++ -std=c++17 -fsyntax-only` over the shim with this
// directory on the include path: it proves nothing about DuckDB's behaviour, but every typo, wrong member name and
// template-instantiation error of the shim's `RealDuck` traits over exon_table_function.hpp is caught here.  Nothing in
// the product includes this file.
#pragma once
#include <stdint.h>

#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#ifndef DUCKDB_EXTENSION_API
#define DUCKDB_EXTENSION_API
#endif

namespace duckdb {
using std::string;
using idx_t = uint64_t;
using column_t = idx_t;
using data_t = uint8_t;
using data_ptr_t = data_t *;
using validity_t = uint64_t;
template <class T>
class vector : public std::vector<T> {
public:
	using std::vector<T>::vector;
};
template <class T, class D = std::default_delete<T>>
class unique_ptr : public std::unique_ptr<T, D> {
public:
	using std::unique_ptr<T, D>::unique_ptr;
};
using std::shared_ptr;
template <class T>
using buffer_ptr = shared_ptr<T>;
template <class T, class... A>
unique_ptr<T> make_uniq(A &&...a) {
	return unique_ptr<T>(new T(std::forward<A>(a)...));
}
using std::make_shared;
template <class T, class... A>
buffer_ptr<T> make_buffer(A &&...a) {
	return std::make_shared<T>(std::forward<A>(a)...);
}
template <class T>
class optional_ptr {
public:
	optional_ptr() : p(nullptr) {
	}
	optional_ptr(T *p_p) : p(p_p) { // NOLINT
	}
	T *get() const {
		return p;
	}
	T *operator->() const {
		return p;
	}
	T &operator*() const {
		return *p;
	}
	explicit operator bool() const {
		return p != nullptr;
	}

private:
	T *p;
};
template <class S>
data_ptr_t data_ptr_cast(S *p) {
	return reinterpret_cast<data_ptr_t>(p);
}

static constexpr idx_t STANDARD_VECTOR_SIZE = 2048;
static constexpr column_t COLUMN_IDENTIFIER_ROW_ID = (column_t)-1;

struct string_t {
	uint32_t length;
	char prefix[4];
	char *ptr;
	idx_t GetSize() const;
	const char *GetData() const;
};
struct list_entry_t {
	uint64_t offset;
	uint64_t length;
};

enum class LogicalTypeId : uint8_t { BOOLEAN, INTEGER, BIGINT, FLOAT, VARCHAR, STRUCT, LIST };
class LogicalType;
template <class T>
using child_list_t = vector<std::pair<string, T>>;
class LogicalType {
public:
	LogicalType();
	LogicalType(LogicalTypeId id); // NOLINT
	LogicalTypeId id() const;
	static const LogicalType BOOLEAN, INTEGER, BIGINT, FLOAT, VARCHAR;
	static LogicalType LIST(const LogicalType &child);
	static LogicalType STRUCT(child_list_t<LogicalType> children);
};
struct ListType {
	static const LogicalType &GetChildType(const LogicalType &type);
};
struct StructType {
	static const child_list_t<LogicalType> &GetChildTypes(const LogicalType &type);
};

enum class VectorBufferType : uint8_t { STANDARD_BUFFER, OPAQUE_BUFFER };
class VectorBuffer {
public:
	explicit VectorBuffer(VectorBufferType type);
	virtual ~VectorBuffer();
};
struct ValidityMask {
	void Initialize(validity_t *validity);
	bool RowIsValid(idx_t row_idx) const;
};
struct SelectionVector {
	idx_t get_index(idx_t idx) const;
};
struct UnifiedVectorFormat {
	const SelectionVector *sel;
	data_ptr_t data;
	ValidityMask validity;
};
enum class VectorType : uint8_t { FLAT_VECTOR, FSST_VECTOR, CONSTANT_VECTOR, DICTIONARY_VECTOR, SEQUENCE_VECTOR };
class Vector {
public:
	void SetAuxiliary(buffer_ptr<VectorBuffer> new_buffer);
	void SetVectorType(VectorType vector_type);
	void ToUnifiedFormat(idx_t count, UnifiedVectorFormat &data);
};
struct FlatVector {
	static ValidityMask &Validity(Vector &vector);
	static void SetData(Vector &vector, data_ptr_t data);
	template <class T>
	static T *GetData(Vector &vector);
	static void SetNull(Vector &vector, idx_t idx, bool is_null);
};
struct ListVector {
	static Vector &GetEntry(Vector &vector);
	static void SetListSize(Vector &vec, idx_t size);
	static void Reserve(Vector &vec, idx_t required_capacity);
};
struct StructVector {
	static vector<unique_ptr<Vector>> &GetEntries(Vector &vector);
};
class DataChunk {
public:
	vector<Vector> data;
	void SetCardinality(idx_t count);
	idx_t size() const;
};

class Value {
public:
	Value(string val); // NOLINT
	template <class T>
	T GetValue() const;
	string ToSQLString() const;
};
enum class ExpressionType : uint8_t {
	COMPARE_EQUAL = 25,
	COMPARE_NOTEQUAL = 26,
	COMPARE_LESSTHAN = 27,
	COMPARE_GREATERTHAN = 28,
	COMPARE_LESSTHANOREQUALTO = 29,
	COMPARE_GREATERTHANOREQUALTO = 30
};
string ExpressionTypeToOperator(ExpressionType type);

class ClientContext;
class ExecutionContext;
class DatabaseInstance;
class DuckDB {
public:
	static const char *LibraryVersion();
};

// ---- duckdb/function/function.hpp, table_function.hpp
struct FunctionData {
	virtual ~FunctionData();
	template <class TARGET>
	TARGET &Cast() {
		return reinterpret_cast<TARGET &>(*this);
	}
	template <class TARGET>
	const TARGET &Cast() const {
		return reinterpret_cast<const TARGET &>(*this);
	}
};
struct TableFunctionData : public FunctionData {
	~TableFunctionData() override;
};
struct TableFunctionInfo {
	virtual ~TableFunctionInfo();
	template <class TARGET>
	TARGET &Cast() {
		return reinterpret_cast<TARGET &>(*this);
	}
};
struct GlobalTableFunctionState {
	virtual ~GlobalTableFunctionState();
	virtual idx_t MaxThreads() const;
	template <class TARGET>
	TARGET &Cast() {
		return reinterpret_cast<TARGET &>(*this);
	}
};
struct LocalTableFunctionState {
	virtual ~LocalTableFunctionState();
	template <class TARGET>
	TARGET &Cast() {
		return reinterpret_cast<TARGET &>(*this);
	}
};
using named_parameter_map_t = std::unordered_map<string, Value>;
using named_parameter_type_map_t = std::unordered_map<string, LogicalType>;
struct TableFunctionBindInput {
	vector<Value> &inputs;
	named_parameter_map_t &named_parameters;
	vector<LogicalType> &input_table_types;
	vector<string> &input_table_names;
	optional_ptr<TableFunctionInfo> info;
};

// ---- duckdb/planner/table_filter.hpp, filter/*.hpp
enum class TableFilterType : uint8_t { CONSTANT_COMPARISON = 0, IS_NULL = 1, IS_NOT_NULL = 2, CONJUNCTION_OR = 3, CONJUNCTION_AND = 4 };
class TableFilter {
public:
	explicit TableFilter(TableFilterType filter_type_p);
	virtual ~TableFilter();
	TableFilterType filter_type;
};
class TableFilterSet {
public:
	std::unordered_map<idx_t, unique_ptr<TableFilter>> filters;
};
class ConstantFilter : public TableFilter {
public:
	ConstantFilter(ExpressionType comparison_type, Value constant);
	ExpressionType comparison_type;
	Value constant;
};
class ConjunctionFilter : public TableFilter {
public:
	explicit ConjunctionFilter(TableFilterType filter_type_p);
	vector<unique_ptr<TableFilter>> child_filters;
};

struct TableFunctionInitInput {
	optional_ptr<const FunctionData> bind_data;
	const vector<column_t> &column_ids;
	const vector<idx_t> projection_ids;
	optional_ptr<TableFilterSet> filters;
};
struct TableFunctionInput {
	optional_ptr<const FunctionData> bind_data;
	optional_ptr<LocalTableFunctionState> local_state;
	optional_ptr<GlobalTableFunctionState> global_state;
};
typedef unique_ptr<FunctionData> (*table_function_bind_t)(ClientContext &context, TableFunctionBindInput &input,
                                                          vector<LogicalType> &return_types, vector<string> &names);
typedef unique_ptr<GlobalTableFunctionState> (*table_function_init_global_t)(ClientContext &context, TableFunctionInitInput &input);
typedef unique_ptr<LocalTableFunctionState> (*table_function_init_local_t)(ExecutionContext &context, TableFunctionInitInput &input,
                                                                           GlobalTableFunctionState *global_state);
typedef void (*table_function_t)(ClientContext &context, TableFunctionInput &data, DataChunk &output);
// ---- duckdb/storage/statistics/node_statistics.hpp
class NodeStatistics {
public:
	NodeStatistics();
	explicit NodeStatistics(idx_t estimated_cardinality);
	NodeStatistics(idx_t estimated_cardinality, idx_t max_cardinality);
	bool has_estimated_cardinality;
	idx_t estimated_cardinality;
	bool has_max_cardinality;
	idx_t max_cardinality;
};
typedef unique_ptr<NodeStatistics> (*table_function_cardinality_t)(ClientContext &context, const FunctionData *bind_data);
typedef idx_t (*table_function_get_batch_index_t)(ClientContext &context, const FunctionData *bind_data,
                                                  LocalTableFunctionState *local_state, GlobalTableFunctionState *global_state);
class TableFunction {
public:
	TableFunction(string name, vector<LogicalType> arguments, table_function_t function, table_function_bind_t bind = nullptr,
	              table_function_init_global_t init_global = nullptr, table_function_init_local_t init_local = nullptr);
	named_parameter_type_map_t named_parameters;
	table_function_cardinality_t cardinality;
	table_function_get_batch_index_t get_batch_index;
	bool projection_pushdown;
	bool filter_pushdown;
	shared_ptr<TableFunctionInfo> function_info;
};
// ---- duckdb/function/scalar_function.hpp
struct ExpressionState;
typedef std::function<void(DataChunk &, ExpressionState &, Vector &)> scalar_function_t;
class ScalarFunction {
public:
	ScalarFunction(string name, vector<LogicalType> arguments, LogicalType return_type, scalar_function_t function);
};
struct ExtensionUtil {
	static void RegisterFunction(DatabaseInstance &db, TableFunction function);
	static void RegisterFunction(DatabaseInstance &db, ScalarFunction function);
};

// ---- parser
class ParsedExpression {
public:
	virtual ~ParsedExpression();
};
class ConstantExpression : public ParsedExpression {
public:
	explicit ConstantExpression(Value val);
};
class FunctionExpression : public ParsedExpression {
public:
	FunctionExpression(const string &function_name, vector<unique_ptr<ParsedExpression>> children);
};
class TableRef {
public:
	virtual ~TableRef();
};
class TableFunctionRef : public TableRef {
public:
	TableFunctionRef();
	unique_ptr<ParsedExpression> function;
};

// ---- duckdb/main/config.hpp (replacement scans)
struct ReplacementScanData {
	virtual ~ReplacementScanData();
};
typedef unique_ptr<TableRef> (*replacement_scan_t)(ClientContext &context, const string &table_name, ReplacementScanData *data);
struct ReplacementScan {
	explicit ReplacementScan(replacement_scan_t function, unique_ptr<ReplacementScanData> data = nullptr);
	replacement_scan_t function;
	unique_ptr<ReplacementScanData> data;
};
struct DBConfig {
	static DBConfig &GetConfig(DatabaseInstance &db);
	vector<ReplacementScan> replacement_scans;
};
} // namespace duckdb
