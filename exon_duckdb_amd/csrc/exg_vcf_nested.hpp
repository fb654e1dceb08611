// exg_vcf_nested.hpp — the reference's nested VCF columns (id / alt / filter LIST(VARCHAR), info STRUCT of the
// header's ##INFO keys, formats LIST(STRUCT of the ##FORMAT keys)) built on the device in DuckDB's vector layouts, from the
// tokeniser's string_t columns (exg_vcf.hip).  Round 6 rebuild of round 3's thread-per-row byte loops (exg_vcf_typed.hip, gone):
//
//   * keys live in device memory (any number of them: no by-value table), looked up through an open-addressed hash of the
//     key text built on the host at bind; there is no (row x key) cell table in HBM — values are parsed where they are
//     found and stored straight into their child vector;
//   * three kernels, each in a counting and a writing form (lists need their offsets before their elements):
//       k_rows       thread = row: id / alt / filter, and INFO when the header declares <= kKA keys and the field is <= 128
//                    bytes — the field is staged in an LDS row of the thread's own, the (key -> value) cells of the row sit
//                    in LDS, and the typed children are written key by key (wave-uniform type switch, coalesced stores,
//                    validity words by ballot);
//       k_info_wide  wave = row, for the INFO fields k_rows leaves (wide headers, long fields): 1 KiB pieces of the field
//                    staged in LDS, ';' found with match + prefix sums, lane = entry;
//       k_samples    wave = row: FORMAT keys resolved once per line (lane = key), the samples of a 1 KiB piece found by
//                    their tabs, lane = sample, and the FORMAT positions walked in lockstep — the key, and so the type
//                    switch, is wave-uniform, consecutive samples store consecutive elements;
//   * one batched prefix sum for all list columns of a stage, one read-back of their totals (two for FORMAT keys that
//     are lists), one contiguous device region per top-level column mirrored by one D2H copy.
//
// Semantics: the header comment of exg_vcf_nested.hip (exon 0.2.6 VCFArrayBuilder over noodles-vcf 0.34; pinned by
// test_vcf_record_scan.test:10-19 and restated by oracle/pyoracle.py::vcf_typed_rows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/exon_gpu.h"

namespace exg {
namespace vn {

enum : uint8_t { kFlag = 0, kInt = 1, kFloat = 2, kString = 3 };  // (= exg::arrow::kVt*, exg_rd::kKey*)

// one declared key (device array, header order)
struct Key {
    uint32_t name_off;  // in KeyTab::names
    uint16_t name_len;
    uint8_t type, is_list;
    uint32_t hash;     // key_hash(name)
    int32_t list_idx;  // index among the list keys of its kind, -1: scalar
};
struct KeyTab {
    const Key *keys;
    const uint32_t *slots;  // open addressing, linear probing: key index + 1, 0 = empty; slot_mask + 1 entries (>= 2 x n_keys)
    const uint8_t *names;
    uint32_t n_keys, slot_mask, names_bytes, n_lists;
};
// host and device agree on the hash (FNV-1a) and the first slot
inline __host__ __device__ uint32_t key_hash_step(uint32_t h, uint32_t b) { return (h ^ b) * 16777619u; }
static constexpr uint32_t kKeyHashSeed = 2166136261u;
inline __host__ __device__ uint32_t key_slot(uint32_t h, uint32_t mask) { return (h ^ (h >> 15)) & mask; }

// where the children of one key go (device array per kind and batch)
struct KeyOut {
    void *vals;             // scalar: int32 / float / uint8 (Flag) / string_t per element;  list: unused (entries come from goff)
    uint64_t *valid;        // bit per element (scalar: value present; list: the list is not NULL)
    void *child_vals;       // list: elements, at goff[element]
    uint32_t *child_valid;  // list: bit per child element, preset to ones, cleared for '.'
};

static constexpr unsigned int kSlowCap = 4096;
struct SlowF32 {  // a Float literal left to the exact parser (exg_float_slow.hpp)
    const uint8_t *p;
    uint32_t len, code;  // code: EXG_PE_VCF_INFO / _FORMAT, should the literal turn out malformed
    float *dst;
    unsigned long long row;
};
struct Ctl {
    unsigned long long err;  // atomicMin((row << 8) | code); ~0 = none
    unsigned int n_slow, side_overflow;
    unsigned long long side_used;  // bytes of percent-decoded strings wanted so far (may exceed the side buffer)
    SlowF32 slow[kSlowCap];
};

// list columns over ROWS, in this order in cnt / goff: id, alt, filter, samples (formats), then the INFO list keys
enum { kColId = 0, kColAlt = 1, kColFilter = 2, kColSamples = 3, kColInfo0 = 4 };

struct Batch {
    const exg_string_t *col[5];  // id, alt, filter, info, FORMAT + samples remainder
    const uint64_t *rest_valid;  // bit per SCAN row: the line has a 9th field
    const uint32_t *row_map;     // NULL: output row j = scan row j
    const uint8_t *d_base;       // the scanned text ...
    uint64_t payload_base;       // ... and the host address string_t pointers give its byte 0
    uint64_t n;                  // output rows
    Ctl *ctl;
    uint8_t *d_side;  // percent-decoded String values (bump allocation through ctl->side_used)
    uint64_t side_cap, side_payload_base;
    uint32_t *cnt;   // cnt[c * cnt_stride + j]: elements of list column c in row j
    uint64_t cnt_stride;
    const uint64_t *goff;  // goff[c * goff_stride + j], n + 1 entries per column (after the scan)
    uint64_t goff_stride;
    exg_string_t *elems[3];  // WRITE: the elements of id / alt / filter
    uint32_t *key_any;  // WRITE: key_any[q] = 1 when INFO key q has a valid element in this batch (NULL: not wanted): a child without one does not travel
    unsigned long long *mid_rows;  // COUNT: += rows whose INFO field is longer than k_rows' small row (64 bytes) but not than its large one (NULL: not counted)
};

// FORMAT-level state of a batch (elements = samples)
struct Samples {
    uint64_t S;                // samples of the batch
    uint32_t *cnt;             // cnt[list_idx * cnt_stride + s]
    uint64_t cnt_stride;
    const uint64_t *goff;
    uint64_t goff_stride;
    uint32_t *srow;            // output row of every sample (NULL: not wanted)
};

uint64_t scan_tmp_entries(uint64_t n_cols, uint64_t n);

// ---- stage 1: counts over rows ---------------------------------------------------------------------------------------
// id / alt / filter (and the narrow INFO fields' list keys), the wide INFO fields' list keys, samples per row
// small_rows: k_rows with the 64-byte row (the fields of 65 - 128 bytes are then k_info_wide's too); count and write may differ
void rows_count(const Batch &b, const KeyTab &info, bool small_rows, hipStream_t s);
// (d_seen: info_wide_seen_bytes() of scratch, NULL when that is 0)
size_t info_wide_seen_bytes(const KeyTab &info, uint64_t n, uint32_t rows_per_group);
void info_wide_count(const Batch &b, const KeyTab &info, uint32_t rows_per_group, uint32_t *d_seen, bool small_rows, hipStream_t s);
void samples_count(const Batch &b, uint32_t rows_per_group, hipStream_t s);
// exclusive prefix sums of n_cols columns of counts (cnt[c * cnt_stride + i], i < n) -> goff[c * goff_stride + i] (n + 1 entries),
// totals[c] = goff[c][n].  d_tmp: scan_tmp_entries(n_cols, n) u64
void scan_counts(const uint32_t *d_cnt, uint64_t cnt_stride, uint64_t n_cols, uint64_t n, uint64_t *d_goff, uint64_t goff_stride,
                 uint64_t *d_totals, uint64_t *d_tmp, hipStream_t s);
// ---- stage 2: counts over samples (FORMAT keys that are lists) -------------------------------------------------------------
void samples_count_lists(const Batch &b, const Samples &sm, const KeyTab &format, uint32_t rows_per_group, hipStream_t s);
// ---- stage 3: the children --------------------------------------------------------------------------------------------------
void rows_write(const Batch &b, const KeyTab &info, const KeyOut *d_info_out, bool small_rows, hipStream_t s);
void info_wide_write(const Batch &b, const KeyTab &info, const KeyOut *d_info_out, uint32_t rows_per_group, uint32_t *d_seen, bool small_rows, hipStream_t s);
void samples_write(const Batch &b, const Samples &sm, const KeyTab &format, const KeyOut *d_format_out, uint32_t rows_per_group, hipStream_t s);
// the Float literals the kernels above left to the exact parser
void fix_slow_floats(Ctl *ctl, hipStream_t s);

// does k_rows take the INFO fields of a header with this many keys (else: every row is k_info_wide's)
bool rows_take_info(uint32_t n_info_keys);

// ---- DuckDB list entries for many list columns at once -----------------------------------------------------------------
struct EntryJob {
    const uint64_t *goff;  // n + 1 offsets of the column
    void *entries;         // ListEntry[n] out (NULL: none)
    uint64_t *bases;       // n_chunks + 1 out: where every DataChunk's children begin (NULL: none)
};
// over rows: entries[i] = {goff[i] - goff[i - i % chunk_rows], goff[i + 1] - goff[i]}, bases[c] = goff[min(c * chunk_rows, n)]
void entries_rows(const EntryJob *d_jobs, uint32_t n_jobs, uint64_t n, uint64_t chunk_rows, uint64_t n_chunks, hipStream_t s);
// over the samples of the rows: elem_row[s] = row, outer_goff = the samples' offsets per row
void entries_elems(const EntryJob *d_jobs, uint32_t n_jobs, uint64_t m, const uint32_t *d_elem_row, const uint64_t *d_outer_goff,
                   uint64_t n_rows, uint64_t chunk_rows, uint64_t n_chunks, hipStream_t s);
// Arrow's Boolean values out of the byte-per-value flags
void bytes_to_bits(const uint8_t *d_bytes, uint64_t m, uint64_t *d_bits, hipStream_t s);

}  // namespace vn
}  // namespace exg
