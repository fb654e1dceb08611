// exg_arrow.hpp — device side of the Arrow export (new_reader): Arrow buffers (offsets, values,
// validity, list offsets, typed INFO / FORMAT children) are built in HBM from the scan kernels'
// duckdb::string_t columns; the host only copies them back (exg_arrow_stream.cpp).
// All functions enqueue on `stream` and return immediately.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exon_gpu.h"

namespace exg {
namespace arrow {

// a duckdb::string_t column on the device; pointers inside it address `payload_base + offset`, the
// bytes themselves are at `d_base + offset`
struct StrCol {
    const exg_string_t *d_col;
    const uint8_t *d_base;
    uint64_t payload_base;
};


// u64 entries of scratch the scans need for n elements
uint64_t scan_tmp_entries(uint64_t n);

// ---- strings -> Arrow Utf8 ------------------------------------------------------------------------
// d_goff[0..n] = exclusive prefix of the lengths (u64), rows taken through d_row_map when not NULL
void utf8_goff_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint64_t *d_goff, uint64_t *d_tmp,
                        hipStream_t stream);
// values[goff[j] .. goff[j+1]) = bytes of string j.  d_big: scratch for the indices of strings >= 8 KiB
// (big_cap entries + 1 counter word in front), copied by a second grid-wide kernel.
void utf8_copy_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff,
                        uint8_t *d_values, uint32_t *d_big, uint32_t big_cap, hipStream_t stream);
// The payload of ONE projected column of an input whose bytes live only in HBM (a decoded stream): goff = exclusive prefix of the
// lengths of the column's OUT-OF-LINE strings (> 12 bytes; inlined ones count 0), values = those strings closed up, and
// repoint_strings writes the column (gathered through d_row_map when given) with every out-of-line pointer = new_base + goff[j]
// — only the selected columns' bytes cross PCIe (exg_rd_batch.cpp).  d_out may be d_in when there is no row map.
void payload_goff_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint64_t *d_goff, uint64_t *d_tmp, hipStream_t stream);
void payload_copy_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff, uint8_t *d_values, uint32_t *d_big,
                           uint32_t big_cap, hipStream_t stream);
void repoint_strings(const exg_string_t *d_in, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff, uint64_t new_base,
                     exg_string_t *d_out, hipStream_t stream);
// per record batch of chunk_rows rows: off32[c * (chunk_rows + 1) + i] = goff[c * chunk_rows + i] - goff[c * chunk_rows]
// (i = 0..rows of the chunk), chunk_base[c] = goff[c * chunk_rows]
void rebase_offsets(const uint64_t *d_goff, uint64_t n, uint64_t chunk_rows, int32_t *d_off32, uint64_t *d_chunk_base,
                    hipStream_t stream);
// off32[i] = (int32) goff[i], i = 0..n (absolute offsets: list columns and their children)
void narrow_offsets(const uint64_t *d_goff, uint64_t n, int32_t *d_off32, hipStream_t stream);

// ---- quality_score_string_to_list (exon/src/exon/fastq_functions/module.cpp:28-54) ------------------------------
// DuckDB LIST(INTEGER) of a VARCHAR column: entries[j] = {goff[j], goff[j+1] - goff[j]} (list_entry_t) and
// values[goff[j] + i] = (int32)(signed char)byte i of string j - 33.  d_goff comes from utf8_goff_from_col.
// Nothing is written when goff[n] > values_cap (the caller reads goff[n] to find out).
struct ListEntry {
    uint64_t offset, length;
};
void quality_list(const StrCol &c, uint64_t n, const uint64_t *d_goff, ListEntry *d_entries, int32_t *d_values,
                  uint64_t values_cap, hipStream_t stream);

// ---- row selection (filters) ------------------------------------------------------------------------
enum : uint8_t { kColStr = 0, kColI64 = 1, kColF32 = 2 };
enum : uint8_t { kOpCmp = 0, kOpIsNull = 1, kOpIsNotNull = 2, kOpAnd = 3, kOpOr = 4 };
enum : uint8_t { kEq = 0, kNe = 1, kLt = 2, kLe = 3, kGt = 4, kGe = 5 };
enum : uint8_t { kLitStr = 0, kLitInt = 1, kLitFloat = 2 };

struct FilterOp {
    uint8_t op, col, cmp, lit;
    uint32_t str_off, str_len;  // kLitStr: bytes in d_consts
    int64_t i;
    double f;
};
static constexpr int kMaxFilterOps = 32;
static constexpr int kMaxFilterCols = 9;
struct FilterProgram {  // postfix
    uint32_t n_ops;
    FilterOp ops[kMaxFilterOps];
};
struct FilterCols {
    // 32-bit on purpose: hipcc 7.2 folds the address of a BYTE array element into the scalar base of the
    // neighbouring pointer arrays' s_load (base = &kind[c], soffset = 7c); SMEM ignores the two low bits
    // of its base, so every column but 0 read a torn pointer.  A dword array keeps the base aligned.
    uint32_t kind[kMaxFilterCols];
    const void *data[kMaxFilterCols];
    const uint64_t *validity[kMaxFilterCols];  // NULL => no nulls
    const uint8_t *d_base[kMaxFilterCols];
    uint64_t payload_base[kMaxFilterCols];
};
// d_row_map[0..n_out) = indices of the rows where the predicate is TRUE, d_goff_tmp[n] = n_out (u64);
// d_goff_tmp has n + 1 entries; program and column table live in device memory
void filter_rows(const FilterProgram *d_prog, const FilterCols *d_cols, const uint8_t *d_consts, uint64_t n,
                 uint64_t *d_goff_tmp, uint64_t *d_tmp, uint32_t *d_row_map, hipStream_t stream);

// ---- gathers through the row map ------------------------------------------------------------------------
void gather_bits(const uint64_t *d_in, const uint32_t *d_row_map, uint64_t n_out, uint64_t *d_out, hipStream_t stream);
void gather_u64(const uint64_t *d_in, const uint32_t *d_row_map, uint64_t n_out, uint64_t *d_out, hipStream_t stream);
void gather_u32(const uint32_t *d_in, const uint32_t *d_row_map, uint64_t n_out, uint32_t *d_out, hipStream_t stream);
void gather_u128(const void *d_in, const uint32_t *d_row_map, uint64_t n_out, void *d_out, hipStream_t stream);  // string_t

// (the nested VCF columns — id / alt / filter / info / formats — are exg_vcf_nested.hpp's)

}  // namespace arrow
}  // namespace exg
