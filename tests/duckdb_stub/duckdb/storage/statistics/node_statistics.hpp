// test infrastructure: see tests/duckdb_stub/duckdb.hpp
#pragma once
#include "duckdb.hpp"
