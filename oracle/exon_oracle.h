/*
 * exon_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the record-scan path of wheretrue/exon-duckdb v0.8.0.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libexon_gpu.so) never links or calls it.
 *
 * See exon_oracle.c for what is restated, from where, and how it is pinned.
 */
#ifndef EXON_ORACLE_H
#define EXON_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* same 16-byte layout as duckdb::string_t v0.8.1 */
typedef union orc_string_t {
    struct {
        uint32_t length;
        char prefix[4];
        uint64_t ptr;
    } pointer;
    struct {
        uint32_t length;
        char inlined[12];
    } inlined;
} orc_string_t;

/* An Arrow-style Utf8 column built the way exon's ArrayBuilders build it:
 * one offsets push + one memcpy into the value buffer per row. */
typedef struct orc_utf8_col {
    int64_t n_rows;
    int64_t *offsets;  /* n_rows + 1 */
    uint8_t *values;   /* offsets[n_rows] bytes */
    uint8_t *valid;    /* one byte per row: 1 valid, 0 NULL */
    int64_t *src_off;  /* byte offset of the field in the input, -1 when the value is not a contiguous slice */
    int64_t cap_rows, cap_values;
} orc_utf8_col;

typedef struct orc_i64_col {
    int64_t n_rows;
    int64_t *data;
    uint8_t *valid;
    int64_t cap_rows;
} orc_i64_col;

typedef struct orc_f32_col {
    int64_t n_rows;
    float *data;
    uint8_t *valid;
    int64_t cap_rows;
} orc_f32_col;

typedef struct orc_error {
    uint32_t code;       /* same numbering as EXG_PE_* in include/exon_gpu.h */
    uint64_t record;     /* index of the failing record */
    uint64_t offset;     /* byte offset of the failing line */
    char message[128];
} orc_error;

typedef struct orc_fastq_table {
    orc_utf8_col name, description, sequence, quality_scores;
    orc_error err;
} orc_fastq_table;

typedef struct orc_fasta_table {
    orc_utf8_col id, description, sequence;
    orc_error err;
} orc_fasta_table;

typedef struct orc_vcf_table {
    /* raw tab-separated fields, then typed POS/QUAL */
    orc_utf8_col fields[9]; /* chrom,pos,id,ref,alt,qual,filter,info,formats */
    orc_i64_col pos;
    orc_f32_col qual;
    int64_t header_bytes; /* offset of the first data line */
    int64_t n_header_lines;
    orc_error err;
} orc_vcf_table;

/* All parsers return the number of rows successfully parsed before the first
 * error (err.code != 0 tells whether there was one). */
int64_t orc_fastq_parse(const uint8_t *buf, uint64_t n, orc_fastq_table *out);
int64_t orc_fasta_parse(const uint8_t *buf, uint64_t n, orc_fasta_table *out);
int64_t orc_vcf_parse(const uint8_t *buf, uint64_t n, orc_vcf_table *out);
void orc_fastq_free(orc_fastq_table *t);
void orc_fasta_free(orc_fasta_table *t);
void orc_vcf_free(orc_vcf_table *t);

/* Arrow Utf8 -> duckdb::string_t, the per-row work of ArrowToDuckDB
 * (called at exon/src/exon/arrow_table_function/module.cpp:289).
 * mode 0: pointers into the column's own value buffer (what DuckDB does);
 * mode 1: canonical zero-copy view, ptr = payload_base + src_off (must be >= 0).
 * NULL rows become 16 zero bytes.  validity_words gets bit r = valid. */
void orc_utf8_to_string_t(const orc_utf8_col *col, int64_t row0, int64_t n, int mode, uint64_t payload_base,
                          orc_string_t *out, uint64_t *validity_words);

/* f32::from_str / i32::from_str acceptance + value (1 = parsed). */
int orc_parse_f32_text(const uint8_t *p, uint64_t n, float *out);
int orc_parse_i32_text(const uint8_t *p, uint64_t n, int32_t *out);

/* quality_score_string_to_list (reference: exon/src/exon/fastq_functions/module.cpp:28-54): for every row
 * one list of INTEGERs, value = (char)byte - 33 with `char` signed (x86-64).  entries[2r] = offset of row r's
 * first child value, entries[2r+1] = its length (DuckDB list_entry_t); out_values holds offsets[n_rows]
 * values.  Rows with valid[r] == 0 (valid may be NULL) give an empty entry. */
void orc_quality_score_list(const uint8_t *values, const int64_t *offsets, const uint8_t *valid, int64_t n_rows,
                            uint64_t *entries, int32_t *out_values);

/* Rust core::str::from_utf8 acceptance test. */
int orc_is_valid_utf8(const uint8_t *p, uint64_t n);

/* Deterministic synthetic inputs, SURVEY.md §8 D2. */
#define ORC_SYNTH_FASTQ_RECORD_BYTES 332
void orc_synth_fastq(uint8_t *out, uint64_t file_offset, uint64_t n_bytes, uint64_t seed);
/* ragged correctness variant: returns bytes written (<= cap) for records [0, n_records) */
uint64_t orc_synth_fastq_ragged(uint8_t *out, uint64_t cap, uint64_t n_records, uint64_t seed);
uint64_t orc_synth_vcf(uint8_t *out, uint64_t cap, uint64_t n_lines, uint64_t seed);
uint64_t orc_synth_fasta(uint8_t *out, uint64_t cap, uint64_t n_records, uint64_t seed);

/* cpu_baseline leg: the reference's loop shape — read_batch(2048) into Arrow
 * builders, then ArrowToDuckDB into a DataChunk — over an in-memory FASTQ.
 * Returns records parsed; *checksum folds every emitted string_t length so the
 * work cannot be optimised away. */
int64_t orc_fastq_scan_baseline(const uint8_t *buf, uint64_t n, uint64_t *checksum);
int64_t orc_fastq_scan_baseline_mt(const uint8_t *buf, uint64_t n, int n_threads, int reps);

/* replacement-scan / compression inference restated from rust/src/arrow_reader.rs:60-91,173-197 */
const char *orc_infer_compression(const char *uri, const char *compression_or_null);
const char *orc_replacement_scan(const char *uri);

#ifdef __cplusplus
}
#endif
#endif
