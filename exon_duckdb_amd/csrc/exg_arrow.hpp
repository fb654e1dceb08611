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

// a string on the device (16 B)
struct View {
    const uint8_t *p;
    uint32_t len;
    uint32_t valid;
};

// u64 entries of scratch the scans need for n elements
uint64_t scan_tmp_entries(uint64_t n);

// ---- strings -> Arrow Utf8 ------------------------------------------------------------------------
// d_goff[0..n] = exclusive prefix of the lengths (u64), rows taken through d_row_map when not NULL
void utf8_goff_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint64_t *d_goff, uint64_t *d_tmp,
                        hipStream_t stream);
void utf8_goff_from_views(const View *d_views, uint64_t n, uint64_t *d_goff, uint64_t *d_tmp, hipStream_t stream);
// values[goff[j] .. goff[j+1]) = bytes of string j.  d_big: scratch for the indices of strings >= 8 KiB
// (big_cap entries + 1 counter word in front), copied by a second grid-wide kernel.
void utf8_copy_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff,
                        uint8_t *d_values, uint32_t *d_big, uint32_t big_cap, hipStream_t stream);
void utf8_copy_from_views(const View *d_views, uint64_t n, const uint64_t *d_goff, uint8_t *d_values, uint32_t *d_big,
                          uint32_t big_cap, hipStream_t stream);
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

// ---- VCF typed columns (exg_vcf_typed.hip) ------------------------------------------------------------------
enum : uint8_t { kVtFlag = 0, kVtInt = 1, kVtFloat = 2, kVtString = 3 };
static constexpr int kMaxVtKeys = 96;
struct VtKey {
    uint32_t name_off, name_len;  // in d_names
    uint8_t type, is_list, pad[2];
};
struct VtKeys {
    uint32_t n;
    const uint8_t *d_names;
    VtKey k[kMaxVtKeys];
};
// one cell of the key/value table: where the value of key k sits in row (or sample) j
struct VtCell {
    uint32_t off;  // byte offset of the value from the start of the row's field
    uint32_t len;  // 0xFFFFFFFF: key absent; 0xFFFFFFFE: present without '=' (flags)
};
static constexpr uint32_t kVtAbsent = 0xFFFFFFFFu, kVtBare = 0xFFFFFFFEu;

// split a column on `sep` ('.' alone => no elements): counts -> d_goff (n + 1), then views of the elements
void list_counts(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint8_t sep, uint64_t *d_goff, uint64_t *d_tmp,
                 hipStream_t stream);
void list_views(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint8_t sep, const uint64_t *d_goff, View *d_views,
                hipStream_t stream);

// INFO: d_cells[j * keys.n + k]
void info_cells(const StrCol &info, const uint32_t *d_row_map, uint64_t n, const VtKeys &keys, VtCell *d_cells,
                hipStream_t stream);
// FORMAT + samples out of the 9th column (NULL where the line has 8 fields): samples per row -> d_goff (n + 1)
void sample_counts(const StrCol &rest, const uint64_t *d_rest_valid, const uint32_t *d_row_map, uint64_t n, uint64_t *d_goff,
                   uint64_t *d_tmp, hipStream_t stream);
// d_cells[s * keys.n + k] for every sample s (global index = d_goff[row] + i); d_sample_row[s] = output row,
// d_sample_field[s] = view of the sample's text (the cells' offsets are relative to it)
void sample_cells(const StrCol &rest, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff, const VtKeys &keys,
                  VtCell *d_cells, View *d_sample_field, uint32_t *d_sample_row, hipStream_t stream);

// where the text of element j (an output row for INFO, a sample for FORMAT) comes from
struct CellSrc {
    const VtCell *d_cells;
    uint32_t n_keys, key;
    StrCol col;                  // INFO: the info column, read through d_row_map
    const uint32_t *d_row_map;
    const View *d_fields;        // FORMAT: the samples' texts (col unused)
    const uint32_t *d_elem_row;  // FORMAT: output row of every sample (error reporting); NULL: j itself
};
// Float literals the kernels cannot decide with 19 digits (exg_parse.hpp status 2) are decided exactly by a one-block
// fix-up launch behind them (exg_float_slow.hpp): the list lives right behind the error word — d_err points at an ErrBlock.
struct SlowF32 {
    const uint8_t *p;
    uint32_t len, pad;
    float *dst;
    unsigned long long row;
};
struct ErrBlock {
    unsigned long long err;  // atomicMin((row << 8) | code); ~0 = none
    unsigned int n_slow, pad;
    SlowF32 slow[4096];  // per kernel call; one more is a value error of its row
};
static constexpr unsigned int kSlowF32 = 4096;
// scalar children: values + validity (bit j); errors: atomicMin(*d_err, (row << 8) | err_code)
void cells_to_i32(const CellSrc &s, uint64_t n, int32_t *d_values, uint64_t *d_valid, unsigned long long *d_err,
                  uint32_t err_code, hipStream_t);
void cells_to_f32(const CellSrc &s, uint64_t n, float *d_values, uint64_t *d_valid, unsigned long long *d_err,
                  uint32_t err_code, hipStream_t);
void cells_to_flag(const CellSrc &s, uint64_t n, uint64_t *d_bits, uint64_t *d_valid, hipStream_t);
void cells_to_views(const CellSrc &s, uint64_t n, View *d_views, uint64_t *d_valid, hipStream_t);
// list children: element counts -> d_goff (n + 1) and validity of the list; then the elements
// (d_child_valid: bit per element, zeroed by the caller)
void cells_list_counts(const CellSrc &s, uint64_t n, uint64_t *d_goff, uint64_t *d_tmp, uint64_t *d_valid, hipStream_t);
void cells_list_i32(const CellSrc &s, uint64_t n, const uint64_t *d_goff, int32_t *d_values, uint32_t *d_child_valid,
                    unsigned long long *d_err, uint32_t err_code, hipStream_t);
void cells_list_f32(const CellSrc &s, uint64_t n, const uint64_t *d_goff, float *d_values, uint32_t *d_child_valid,
                    unsigned long long *d_err, uint32_t err_code, hipStream_t);
void cells_list_views(const CellSrc &s, uint64_t n, const uint64_t *d_goff, View *d_views, uint32_t *d_child_valid, hipStream_t);
// String / Character values are percent-decoded (noodles-vcf 0.34: percent_encoding::percent_decode(..).decode_utf8()):
// "%3B" -> ';'.  Almost no value holds an escape, so: pass 1 adds up the decoded lengths of the views that do
// (*d_total, zeroed by the caller; 0 = nothing to do); pass 2 decodes those into d_side (bump allocation through
// *d_cursor, zeroed by the caller) and points the views there.  A decoded value that is not UTF-8 is a value error:
// atomicMin(*d_err, (row << 8) | err_code), row = d_elem_row[parent] / parent, parent = the list (d_parent_goff, n_parent)
// the element belongs to / the element itself.
struct PercentRows {
    const uint64_t *d_parent_goff;  // NULL: the views are not list elements
    uint64_t n_parent;
    const uint32_t *d_elem_row;     // NULL: the parent index is the row
};
void percent_count(const View *d_views, uint64_t m, unsigned long long *d_total, hipStream_t);
void percent_decode(View *d_views, uint64_t m, uint8_t *d_side, unsigned long long *d_cursor, const PercentRows &rows,
                    unsigned long long *d_err, uint32_t err_code, hipStream_t);
// validity bits of a view array (bit j = views[j].valid)
void views_validity(const View *d_views, uint64_t n, uint64_t *d_valid, hipStream_t);

// ---- DuckDB vector layouts of the nested columns (the chunk boundary, exg_next_chunk) -------------------------------
// string views -> duckdb::string_t (payload zero-copy: ptr = payload_base + (view.p - d_base)); invalid views -> 16 zero bytes
// (views that point into [d_side, d_side + side_bytes) — percent-decoded values — get ptr = side_payload_base + offset)
void views_to_string_t(const View *d_views, uint64_t m, const uint8_t *d_base, uint64_t payload_base, const uint8_t *d_side,
                       uint64_t side_bytes, uint64_t side_payload_base, exg_string_t *d_out, hipStream_t stream);
// LIST parents over rows: entries[i] = {goff[i] - goff[i - i % chunk_rows], goff[i + 1] - goff[i]} — offsets are relative to
// the first child element of the row's DataChunk, so a chunk's child vector is a slice of the batch-wide child array
void list_entries_rows(const uint64_t *d_goff, uint64_t n, uint64_t chunk_rows, ListEntry *d_entries, hipStream_t stream);
// LIST parents over the elements of an outer list (FORMAT lists per sample): elem_row[s] = row of element s,
// outer_goff[row] = first element of a row; offsets relative to the first inner element of the row's DataChunk
void list_entries_elems(const uint64_t *d_goff, uint64_t m, const uint32_t *d_elem_row, const uint64_t *d_outer_goff,
                        uint64_t chunk_rows, ListEntry *d_entries, hipStream_t stream);
// out[c] = goff[min(c * chunk_rows, n)], c = 0..n_chunks: where every DataChunk's children begin
void chunk_bases_rows(const uint64_t *d_goff, uint64_t n, uint64_t chunk_rows, uint64_t n_chunks, uint64_t *d_out, hipStream_t stream);
// out[c] = goff[idx[c]], c = 0..n_chunks (second level)
void chunk_bases_pick(const uint64_t *d_goff, const uint64_t *d_idx, uint64_t n_chunks, uint64_t *d_out, hipStream_t stream);
// DuckDB BOOLEAN: one byte per value out of a bitmap
void bits_to_bytes(const uint64_t *d_bits, uint64_t m, uint8_t *d_out, hipStream_t stream);

}  // namespace arrow
}  // namespace exg
