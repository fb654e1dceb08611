/*
 * exon_gpu.h — C-ABI of the MI355X-native record-scan engine (libexon_gpu.so).
 *
 * Drop-in boundary for ONE path of wheretrue/exon-duckdb: the read_fasta /
 * read_fastq / read_vcf_file_records table functions that tokenise raw file
 * bytes into DuckDB DataChunks.  Plain C, plain pointers and sizes; no torch,
 * no HIP and no DuckDB types appear in any signature.
 *
 * Three layers, lowest first:
 *
 *   (1) exg_*_scan  — device level.  Input bytes are already resident in HBM;
 *       output column vectors (arrays of 16-byte duckdb::string_t + validity
 *       words) are written to HBM.  This is the hot path bench.py measures.
 *       Replaces the per-batch work of the reference's Rust side:
 *       exon `BatchReader::read_batch` + noodles line readers + Arrow builders
 *       (external crates, reached through rust/src/arrow_reader.rs:116-153)
 *       and the Arrow->DataChunk conversion at
 *       exon/src/exon/arrow_table_function/module.cpp:257-294 (Scan).
 *
 *   (2) exg_open / exg_next_chunk — reader level.  File on the host ->
 *       pinned staging -> HBM -> kernels -> DataChunk-shaped host buffers.
 *       Replaces `new_reader` + the ArrowArrayStream it returns
 *       (exon/include/rust.hpp:41-46, rust/src/arrow_reader.rs:38-166).
 *
 *   (3) replacement_scan — same name, argument and result struct as the reference's FFI
 *       (exon/include/rust.hpp:11-13, :48), so the reference's ReplacementScan glue
 *       (module.cpp:320-382) binds to it unchanged; and `new_reader` (exon/include/rust.hpp:41-46),
 *       the reference's own entry point: an Arrow C stream with the reference's schema, built on the
 *       device (INTEGRATION.md also shows the binding that replaces its two call sites,
 *       module.cpp:98-102 and :239-243, with the faster chunk boundary exg_open).
 *
 * Error convention: every function returns 0 on success or a negative
 * EXG_E_* code; nothing throws across this boundary.  Parse errors found by a
 * kernel are reported through exg_scan_result (error_code > 0 = EXG_PE_*).
 */
#ifndef EXON_GPU_H
#define EXON_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXG_ABI_VERSION 9

/* DuckDB v0.8.1 STANDARD_VECTOR_SIZE; the reference asks the Rust side for
 * batches of exactly this many rows (module.cpp:83, :233). */
#define EXG_VECTOR_SIZE 2048

/* ---- API status codes (negative) ---------------------------------------- */
#define EXG_OK 0
#define EXG_E_INVALID_ARG (-1)
#define EXG_E_NO_DEVICE (-2)   /* HIP runtime / GPU missing: the product path fails loudly */
#define EXG_E_HIP (-3)         /* a HIP call failed; see exg_last_error_message() */
#define EXG_E_IO (-4)
#define EXG_E_UNSUPPORTED (-5) /* e.g. bzip2 / xz: no device decoder, and there is no CPU fallback */
#define EXG_E_PARSE (-6)       /* reader level: a record failed to parse */
#define EXG_E_CAPACITY (-7)    /* output arrays too small */
#define EXG_E_NOMEM (-8)

/* ---- parse error codes (positive; exg_scan_result.error_code) ----------- */
/* Mirrors the io::Error cases of the noodles readers at the pinned versions
 * (rust/Cargo.lock:1991-2194); each is a named oracle test. */
#define EXG_PE_NONE 0
#define EXG_PE_FASTQ_NAME_PREFIX 1    /* line 1 of a record does not start with '@' */
#define EXG_PE_FASTQ_PLUS_PREFIX 2    /* line 3 of a record does not start with '+' */
#define EXG_PE_UNEXPECTED_EOF 3       /* file ends inside a record (before line 3) */
#define EXG_PE_INVALID_UTF8 4         /* a field is not valid UTF-8 */
#define EXG_PE_FASTA_MISSING_PREFIX 5 /* definition line does not start with '>' */
#define EXG_PE_FASTA_MISSING_NAME 6   /* '>' followed by whitespace / nothing */
#define EXG_PE_FASTA_EMPTY_DEF 7      /* empty definition line */
#define EXG_PE_VCF_MISSING_FIELD 8    /* data line with fewer than 8 tab-separated fields */
#define EXG_PE_VCF_BAD_POS 9          /* POS is not a decimal integer */
#define EXG_PE_VCF_BAD_QUAL 10        /* QUAL is neither '.' nor a float */
#define EXG_PE_VCF_NO_HEADER 11       /* no '#CHROM' header line */
#define EXG_PE_FIELD_TOO_LONG 12      /* a field exceeds 2^32-1 bytes (string_t length is u32) */
#define EXG_PE_VCF_INFO 13            /* an INFO value does not parse as the type its ##INFO line declares */
#define EXG_PE_VCF_FORMAT 14          /* a sample value does not parse as the type its ##FORMAT line declares */

/* ---- duckdb::string_t, bit-for-bit (v0.8.1 duckdb/common/types/string_type.hpp)
 * length <= 12: bytes inlined, zero padded.  Otherwise 4-byte prefix + pointer.
 * Rows that are NULL hold 16 zero bytes. */
typedef union exg_string_t {
    struct {
        uint32_t length;
        char prefix[4];
        uint64_t ptr; /* payload_base + byte offset of the field in the input */
    } pointer;
    struct {
        uint32_t length;
        char inlined[12];
    } inlined;
} exg_string_t;

#define EXG_INLINE_LENGTH 12

/* ---- formats ------------------------------------------------------------- */
#define EXG_FMT_FASTA 1
#define EXG_FMT_FASTQ 2
#define EXG_FMT_VCF 3

/* ---- scan flags ------------------------------------------------------------ */
#define EXG_F_BOF 1u /* a line starts at d_input[0] (start of file, or a record-aligned batch) */
#define EXG_F_EOF 2u /* d_input[n_bytes-1] is the last byte of the file */
#define EXG_F_NO_STORE 4u /* COUNT(*) path: parse and validate every record, write no column (capacity ignored) */
#define EXG_F_ALL 7u      /* any other bit is refused (EXG_E_INVALID_ARG) */

/* result flags */
#define EXG_RF_NON_ASCII 1u /* a byte >= 0x80 was seen; the fields of the records around it were validated as UTF-8 */
#define EXG_RF_HEAD_UNRESOLVED 2u /* first owned record starts before d_input[0]: host must stitch */
#define EXG_RF_FALLBACK 4u  /* the fused kernels gave the launch up and the general kernels ran (no input shape does that any more:
                             * kept for EXG_ALGO_AUTO's gate) */
#define EXG_RF_CAPACITY 8u  /* more records than capacity_records: the surplus was not written */
#define EXG_RF_INDEX_OVERFLOW 16u /* general path: more lines than the workspace can index (enlarge d_workspace) */
#define EXG_RF_QUAL_RANGE 32u /* VCF: more QUAL literals of one launch needed the exact big-integer parser (> 19 digits astride a float
                                 rounding boundary) than its list holds (one per 32 bytes of input, at most 4096); the next was rejected */

/* algorithm selector (exg_*_scan_args.algo) */
#define EXG_RF_REDO 64u     /* the lean scan marked super-tiles — a record / line that begins more than 1 KiB in front of the 16 KiB
                             * half it ends in (long reads, multi-sample VCF), more lines in a half than its list holds (reads
                             * below ~45 bp), or a byte >= 0x80 (UTF-8 validation) — and the any-shape run behind it redid them: the output is complete; a caller
                             * with more batches of the same input does better with EXG_ALGO_FUSED_FULL from here on */
#define EXG_ALGO_AUTO 0
#define EXG_ALGO_MULTIPASS 1 /* count -> scan -> index -> fields: 4 launches, reads the input ~3x */
#define EXG_ALGO_FUSED 2     /* single pass: the lean scan, then the any-shape scan over the super-tiles the lean one marked */
#define EXG_ALGO_FUSED_FULL 3 /* single pass: the any-shape scan alone (any record length / line density; 17 % below the lean scan
                               * on 150 bp reads: 2.65 against 2.27 ms per 10 GB) */
#define EXG_ALGO_FUSED_INDEX 4 /* exg_vcf_scan only — wide lines (cohort VCFs: hundreds of bytes to 10 kB a line): the any-shape scan
                               * only notes where every line ends (8 bytes a line, in the workspace), the rows are parsed by a
                               * kernel of their own behind it, a thread per line.  Same rows, flags and errors as
                               * EXG_ALGO_FUSED_FULL; slower than it on short lines */

/* Written by the device (64 bytes, 8-byte aligned), copied back by exg_fetch_result. */
typedef struct exg_scan_result {
    uint64_t n_records;      /* records owned by this buffer (end at offset >= lead) */
    uint64_t n_lines;        /* '\n' count in [lead, n_bytes) (+1 for an unterminated last line at EOF) */
    uint64_t consumed_bytes; /* offset just past the last complete owned record */
    uint64_t error_offset;   /* byte offset (in d_input) of the first failing line; ~0 if none */
    uint64_t error_record;   /* index (within this buffer's output) of the first failing record */
    uint32_t error_code;     /* EXG_PE_* */
    uint32_t flags;          /* EXG_RF_* */
    uint64_t payload_bytes;  /* FASTA: bytes written to the compacted sequence payload */
    uint64_t redo_tiles;     /* EXG_RF_REDO: super-tiles (FASTQ: 48 KiB, VCF: 32 KiB of input each) the any-shape run redid behind the
                              * lean scan — few: the odd long read in a short-read file, cheaper redone than scanned any-shape
                              * throughout; most of them: an input of that shape (EXG_ALGO_FUSED_FULL from here on) */
} exg_scan_result;

/* FASTQ: 4 VARCHAR columns name, description, sequence, quality_scores
 * (order pinned by test/sql/exondb-release-with-deb-info/test_fastq_scan.test:35-41). */
typedef struct exg_fastq_scan_args {
    const void *d_input;       /* device pointer, 16-byte aligned, readable up to round_up(n_bytes,16) */
    uint64_t n_bytes;          /* halo + shard */
    uint64_t lead;             /* halo: records whose last line ends before this offset belong to the previous shard */
    uint64_t first_line_index; /* number of '\n' in the file before d_input[lead] (gives the 4-line phase) */
    uint64_t payload_base;     /* string_t.ptr = payload_base + offset in d_input */
    uint32_t flags;            /* EXG_F_* */
    uint32_t algo;             /* EXG_ALGO_* */
    exg_string_t *d_name;      /* device, capacity_records entries each */
    exg_string_t *d_description;
    exg_string_t *d_sequence;
    exg_string_t *d_quality;
    uint64_t *d_description_validity; /* device, ceil(capacity/64) words, bit r = row r valid */
    uint64_t capacity_records;
    void *d_workspace; /* device, exg_scan_workspace_bytes(EXG_FMT_FASTQ, n_bytes), 256-byte aligned (any hipMalloc pointer is) */
    uint64_t workspace_bytes;
    exg_scan_result *d_result; /* device, 64 bytes */
    void *stream;              /* hipStream_t (NULL = default stream) */
} exg_fastq_scan_args;

/* VCF: one row per data line; the 8 fixed columns (+ the FORMAT/samples rest)
 * as raw field slices, POS parsed to int64, QUAL to float32. */
typedef struct exg_vcf_scan_args {
    const void *d_input;
    uint64_t n_bytes;
    uint64_t lead;
    uint64_t payload_base;
    uint32_t flags;
    uint32_t algo;
    exg_string_t *d_fields[9]; /* chrom,pos,id,ref,alt,qual,filter,info,formats(rest) — any may be NULL (projection) */
    int64_t *d_pos;            /* parsed POS, may be NULL */
    float *d_qual;             /* parsed QUAL, may be NULL */
    uint64_t *d_qual_validity; /* '.' => NULL */
    uint64_t *d_formats_validity;
    uint64_t capacity_records;
    void *d_workspace;
    uint64_t workspace_bytes;
    exg_scan_result *d_result;
    void *stream;
} exg_vcf_scan_args;

/* FASTA: id, description, sequence.  The sequence is the concatenation of the
 * record's lines with terminators removed, so it is materialised in a
 * compacted payload buffer (d_seq_payload); id/description point into d_input.
 * A buffer begins with a record (lead = 0, EXG_F_BOF).  Without EXG_F_EOF it is a batch of a longer input: the last
 * record in it is still open (its sequence may go on behind the buffer) and is left to the next batch — n_records,
 * the columns and payload_bytes are those of the records in front of it, consumed_bytes = where its '>' line begins
 * (0 records when the buffer holds only that one: widen the batch — a FASTA record can be as long as the input). */
typedef struct exg_fasta_scan_args {
    const void *d_input;
    uint64_t n_bytes;
    uint64_t lead;
    uint64_t payload_base;     /* for id/description */
    uint64_t seq_payload_base; /* string_t.ptr of sequences = seq_payload_base + offset in d_seq_payload */
    uint32_t flags;
    uint32_t algo;
    exg_string_t *d_id;
    exg_string_t *d_description;
    exg_string_t *d_sequence;
    uint64_t *d_description_validity;
    uint8_t *d_seq_payload; /* device, >= n_bytes */
    uint64_t capacity_records;
    void *d_workspace;
    uint64_t workspace_bytes;
    exg_scan_result *d_result;
    void *stream;
} exg_fasta_scan_args;

/* ---- library / device ------------------------------------------------------ */
int exg_abi_version(void);
/* Number of visible HIP devices, or EXG_E_NO_DEVICE. Does not initialise a context. */
int exg_device_count(void);
/* Thread-local message for the last failing call on this thread. */
const char *exg_last_error_message(void);
const char *exg_parse_error_string(uint32_t pe_code);

/* ---- (1) device level -------------------------------------------------------- */
uint64_t exg_scan_workspace_bytes(int format, uint64_t n_bytes);
/* Enqueue on args->stream; asynchronous. */
int exg_fastq_scan(const exg_fastq_scan_args *args);
int exg_vcf_scan(const exg_vcf_scan_args *args);
int exg_fasta_scan(const exg_fasta_scan_args *args);
/* Synchronising copy of the 64-byte result to the host. */
int exg_fetch_result(const exg_scan_result *d_result, void *stream, exg_scan_result *out);
/* '\n' count of d_input[begin,end) into *d_count (device u64); used for the shard phase exchange. */
int exg_count_newlines(const void *d_input, uint64_t begin, uint64_t end, uint64_t *d_count, void *stream);
/* The first FASTA record start ('>' at the beginning of a line) at an offset in [begin, end) of d_bytes into *d_pos (device
 * u64; ~0 when there is none).  The byte in front of `begin` is read (begin = 0: at_bof says whether d_bytes[0] begins a
 * line).  What a shard of a compressed FASTA needs at its ends: a record belongs to the shard its '>' line begins in. */
int exg_fasta_find_record(const void *d_bytes, uint64_t begin, uint64_t end, int at_bof, uint64_t *d_pos, void *stream);
/* FASTQ 4-line phase of the first line that starts at or after `lead`, decided from local
 * structure ('@' on line 0, '+' on line 2 for 8 consecutive records).  *d_phase (device u32)
 * receives 0..3, or 0xFFFFFFFF when no or several phases fit (caller falls back to counting). */
int exg_fastq_guess_phase(const void *d_input, uint64_t n_bytes, uint64_t lead, uint32_t *d_phase, void *stream);
/* ABI 9: which scan to launch FIRST on an input, from a sample of its bytes on the HOST (the reader looks at the first MiB behind the
 * header it has mapped anyway; host only, no device is touched): EXG_ALGO_FUSED for ordinary shapes (150 bp reads, 50-byte VCF lines),
 * EXG_ALGO_FUSED_FULL where the lean scan would mark most super-tiles (records of ~1 KiB and more, lines denser than its list holds,
 * bytes >= 0x80), EXG_ALGO_FUSED_INDEX for VCF lines of >= 640 bytes.  Before round 6 a reader found this out from its first
 * batches' results: lean + redo, then the any-shape scan, then (cohort VCFs) the indexed one; a file of one batch never got its scan.
 * The result flags of every batch still correct the choice (exg_reader_stats.scan_algo). */
int exg_scan_algo_hint(int format, const void *sample, uint64_t n_bytes);

/* ---- quality_score_string_to_list on device-resident columns ----------------------------------------
 * Replaces the scalar function of exon/src/exon/fastq_functions/module.cpp:28-54 (one INTEGER per byte of the
 * string, value = (char)c - 33, `char` signed as on x86-64) for a VARCHAR column that is still in HBM — the
 * quality_scores column exg_fastq_scan just wrote.  Output is DuckDB's LIST(INTEGER) layout: one
 * list_entry_t {offset, length} per row + the child INTEGER vector.  NULL rows (16 zero bytes) give empty
 * entries; the list column's validity is the input column's validity. */
typedef struct exg_list_entry_t {
    uint64_t offset; /* first child value of the row */
    uint64_t length;
} exg_list_entry_t;
typedef struct exg_quality_list_args {
    const exg_string_t *d_strings; /* device, n_rows entries */
    uint64_t n_rows;
    const void *d_payload;       /* device bytes the non-inlined strings point into */
    uint64_t payload_base;       /* string_t.ptr - payload_base = byte offset in d_payload */
    exg_list_entry_t *d_entries; /* out: device, n_rows entries */
    int32_t *d_values;           /* out: device, 16-byte aligned, values_capacity entries */
    uint64_t values_capacity;
    uint64_t *d_total; /* out: device u64, number of child values; > values_capacity => entries and values untouched */
    void *d_workspace; /* device, exg_quality_list_workspace_bytes(n_rows) */
    uint64_t workspace_bytes;
    void *stream;
} exg_quality_list_args;
uint64_t exg_quality_list_workspace_bytes(uint64_t n_rows);
/* Enqueue on args->stream; asynchronous. */
int exg_quality_score_list(const exg_quality_list_args *args);

/* ---- device inflate (gzip / BGZF members; replaces flate2 behind rust/src/arrow_reader.rs:60-91) ---- */
typedef struct exg_inflate_member {
    uint64_t comp_off;  /* offset of the member's DEFLATE stream (after the gzip header) in d_comp */
    uint64_t comp_size; /* bytes available from comp_off */
    uint64_t out_off;   /* where its output starts in d_out */
    uint64_t out_cap;   /* bytes it may produce (ISIZE when known) */
} exg_inflate_member;
typedef struct exg_inflate_status {
    uint32_t code; /* 0 ok, 1 bad block, 2 bad code lengths, 3 bad symbol/distance, 4 output overflow */
    uint32_t pad;
    uint64_t produced; /* bytes written at out_off */
    uint64_t consumed; /* compressed bytes consumed from comp_off (byte aligned after the final block) */
} exg_inflate_status;
/* Host: index the gzip members of data[start, n) (RFC 1952 framing).  Members that carry their size (BGZF 'BC'
 * subfield) are all returned; a member of unknown size is returned last with *open_ended = 1 (inflate it, then
 * index again from comp_off + consumed + 8).  *total_out is in/out: running sum of the output offsets. */
int exg_gzip_index(const uint8_t *data, uint64_t n, uint64_t start, exg_inflate_member *members, uint64_t cap,
                   uint64_t *n_members, uint64_t *total_out, int *open_ended);
/* One wavefront per member; d_members / d_status are device arrays; d_comp 16-byte aligned. Asynchronous. */
int exg_inflate_members(const void *d_comp, void *d_out, const exg_inflate_member *d_members,
                        exg_inflate_status *d_status, uint32_t n_members, void *stream);

/* CRC-32 (the gzip trailer's checksum) of inflated bytes in HBM — the reference's decoders verify it (flate2 GzDecoder /
 * noodles-bgzf: "corrupt gzip stream does not have a matching checksum"), and so does the reader for every member.
 * One wavefront per segment; exg_crc32_members takes the segments from an exg_inflate_members call (out_off, produced).
 * Longer outputs are cut into segments whose checksums the host combines (exg_crc32_combine = zlib's crc32_combine). */
typedef struct exg_crc_segment {
    uint64_t off; /* first byte in d_data */
    uint64_t len;
} exg_crc_segment;
int exg_crc32_segments(const void *d_data, const exg_crc_segment *d_segs, uint32_t n_segs, uint32_t *d_crc, void *stream);
int exg_crc32_members(const void *d_out, const exg_inflate_member *d_members, const exg_inflate_status *d_status, uint32_t n_members,
                      uint32_t *d_crc, void *stream);
uint32_t exg_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

/* ONE big DEFLATE stream (a single-member gzip file) decoded by many wavefronts: block starts are searched near
 * every chunk_bytes of compressed input, the chunks are decoded concurrently with an unknown window and stitched
 * (pugz / rapidgzip method).  d_comp: the compressed bytes on the device, 16-byte aligned; the stream starts at
 * comp_off, at most comp_size bytes are read.  On success *d_out is a block of the library's device pool (*produced bytes
 * + 64 zeroed) that the caller gives back with exg_free_device() — hipFree() works too: pool blocks are whole allocations —
 * and *consumed the compressed bytes used.  On an error nothing stays allocated.  Synchronises the stream. */
int exg_inflate_stream(const void *d_comp, uint64_t comp_off, uint64_t comp_size, uint64_t chunk_bytes, void **d_out,
                       uint64_t *produced, uint64_t *consumed, void *stream);
/* The same stream in ROUNDS, so that neither the compressed nor the inflated bytes of a big member have to be resident at
 * once (the reference streams any size through a BufReader: rust/src/arrow_reader.rs:116-153): a round decodes from
 * start_bit (a block boundary; 0 for the first round) to a block boundary near the end of the bytes that are there.
 * Bit positions are relative to comp_off.  partial != 0: more compressed bytes follow behind comp_size — the round ends at
 * (or just behind) the last block start found and does not look at the stream's end; need_more != 0 on return means no
 * block start was found in the window (call again with more bytes; nothing was produced).  d_window: 32768 bytes of device
 * memory that carry the LZ77 window from round to round (have_window = 0 for a member's first round; updated in place).
 * front_reserve bytes stay free in front of the output: d_out + front_reserve is its first byte (any value; a caller that
 * wants an address congruent to the stream offset mod 16 — the decoded segments of the reader — passes reserve + (offset & 15)
 * with a 16-byte aligned reserve). */
typedef struct exg_inflate_round_args {
    const void *d_comp;
    uint64_t comp_off, comp_size;
    uint64_t start_bit;
    uint64_t chunk_bytes;
    int partial;
    int have_window;
    void *d_window;
    uint64_t front_reserve;
    double ratio_hint; /* inflated / compressed bytes seen so far; 0 = unknown */
    void *stream;
    /* results */
    void *d_out;        /* pool block of out_alloc bytes: exg_free_device_sized(d_out, out_alloc) */
    uint64_t out_alloc;
    uint64_t produced;
    uint64_t end_bit;   /* where the next round starts */
    int final_block;    /* the member's final block was decoded: end_bit (rounded up to a byte) is followed by the trailer */
    int need_more;
} exg_inflate_round_args;
int exg_inflate_round(exg_inflate_round_args *args);
/* Give a device block the library handed out (exg_inflate_stream, exg_inflate_round, exg_zstd_decode) back to its pool. */
void exg_free_device(void *d_ptr, uint64_t bytes);

/* ---- device zstd (RFC 8878 frames; replaces zstd 0.12.3 / libzstd 1.5.2 behind rust/src/arrow_reader.rs:73, :87-88;
 * pinned by test_fastq_scan.test:22-32, 55-59 and test_fasta_scan.test:22-26, 45-49) --------------------------------
 * All frames of the stream data[0, n) — concatenated frames and skippable frames included, as ZSTD_decompressStream
 * reads them.  h_comp: the compressed bytes on the host (only frame / block headers are read there: a zstd stream states
 * the size of every block, so the host finds all blocks by a pointer chase and the device entropy-decodes them all at
 * once); d_comp: the same bytes on the device, readable to n + 16.  On success *d_out is a block of the library's device
 * pool (*produced bytes + 64 zeroed): exg_free_device(*d_out, *produced + 64) — hipFree works too.  Every frame with a Content_Checksum is verified: XXH64 on the device up to
 * EXG_ZSTD_VERIFY_MAX bytes of content per frame (default 64 MiB: the hash is a serial recurrence, ~0.55 GB/s per frame on a
 * GPU), larger frames on the host from a copy that comes back in 32 MiB pieces (~20 GB/s; a reader does this on a thread
 * of its own while it scans, and reports a mismatch when the file's last batch is out).  Windows above
 * 128 MiB and dictionaries are refused like libzstd's defaults do.  Synchronises the stream.
 * Errors: EXG_E_PARSE, libzstd's wording in exg_last_error_message(). */
int exg_zstd_decode(const uint8_t *h_comp, const void *d_comp, uint64_t n, void **d_out, uint64_t *produced, void *stream);

/* ---- (2) reader level ----------------------------------------------------------- */
typedef struct exg_reader exg_reader;

typedef struct exg_open_args {
    const char *path;        /* local file or directory (the reference lists directories: test_fasta_scan.test:55-59) */
    const char *file_format; /* "fasta" | "fastq" | "vcf" (the reference's file_type strings, exon_extension.cpp:50,51,55) */
    const char *compression; /* NULL = infer from extension like arrow_reader.rs:60-75; "gzip","zstd","uncompressed",... */
    uint64_t batch_rows;     /* rows per chunk; 0 => EXG_VECTOR_SIZE */
    int device;              /* HIP device ordinal */
    uint64_t device_batch_bytes; /* bytes shipped to HBM per launch; 0 => default */
    const char *filters;     /* NULL / "" or the predicate FilterToString renders (module.cpp:158-214), same grammar as
                              * new_reader's: evaluated on the device, only the rows where it is TRUE are copied back.
                              * Nested columns (VCF id / alt / filter / info / formats) are refused, like in new_reader. */
    uint32_t shard_index;    /* byte-range shards of every file: this reader yields the records / lines that */
    uint32_t shard_count;    /* END in its 1/shard_count of the bytes behind the header.  One process per GPU opens the same
                              * path with its rank: the shards partition the rows, in file order, with no exchange (the FASTQ
                              * 4-line phase at a cut is found from the bytes around it).  The first batch of a shard carries
                              * 1 MiB (EXG_SHARD_HALO) of the bytes in front of its cut; when the record that ends behind the
                              * cut begins further back, the halo grows until it holds it: a record of any length across a
                              * cut is found, like the unsharded scan finds it.  BGZF inputs are sharded by members (a member
                              * belongs to the shard in whose bytes its header begins; each rank uploads and inflates only its
                              * own members + a halo of members in front, a VCF also the leading members that hold the
                              * header), multi-frame zstd inputs by frames.  FASTA: a record belongs to the shard in whose
                              * bytes (bgzip: members' bytes, zstd: frames' bytes) its '>' line begins; a shard is that run of
                              * whole records.  Not sharded (EXG_E_UNSUPPORTED): gzip without member sizes.
                              * shard_count = 1: the whole input through this one reader and its device.
                              * shard_count = 0: the whole input, and the reader may FAN OUT by itself: on a box with several
                              * devices (and an input that can be sharded) it cuts the input into stripes of ~1 GiB, reads
                              * them with worker threads on all devices and hands their batches out here in file order
                              * (exg_count_only sums them) — one consumer-facing stream, N GPUs. */
    uint64_t columns;        /* projection (DuckDB's projection_pushdown: module.cpp:310, input.column_ids): bit c = column c of
                              * exg_schema_of is wanted; 0 = all.  Every column is still tokenised, typed and validated on
                              * the device — a malformed INFO value is an error whether the column is selected or not, like in
                              * the reference, which parses everything and projects afterwards — but only the wanted ones are
                              * copied back: exg_chunk.vectors[c] of the others is NULL.  read_vcf's nested columns are
                              * two thirds of its bytes over PCIe.  Every bit is a column: ~0 = all, like 0. */
    uint64_t flags;          /* ABI 9 (a word of its own: ABI 8 kept this hint in bit 63 of `columns`, where columns = ~0 for "all"
                              * set it by accident).  EXG_OPEN_CHUNKS: the caller is going to pull chunks (exg_next_chunk), not
                              * exg_count_only — what a table function knows at init_global (column_ids != {ROW_ID}).  A compressed
                              * input's decoded segments then travel to the host from the FIRST one on, beside the decoder; without
                              * the hint that begins with the first exg_next_chunk call, and the segments decoded before it (one or
                              * two, up to 1 GiB each for zstd) are copied behind their scan. */
} exg_open_args;
#define EXG_OPEN_CHUNKS 1ull

#define EXG_TYPE_VARCHAR 1
#define EXG_TYPE_BIGINT 2
#define EXG_TYPE_FLOAT 3
#define EXG_TYPE_INTEGER 4 /* int32_t */
#define EXG_TYPE_BOOLEAN 5 /* one byte per value (DuckDB BOOLEAN) */
#define EXG_TYPE_LIST 6    /* data = exg_list_entry_t[] (DuckDB list_entry_t), children[0] = the elements */
#define EXG_TYPE_STRUCT 7  /* no data of its own; children = the fields, each as long as the parent */

/* The full type of a column.  The VCF columns carry the reference's schema (what DuckDB's Arrow conversion makes of
 * exon's Arrow schema, exon/src/exon/arrow_table_function/module.cpp:126-147; pinned by test_vcf_record_scan.test:10-19):
 * id / alt / filter LIST(VARCHAR), info STRUCT(<##INFO keys>), formats LIST(STRUCT(<##FORMAT keys>)), a key with
 * Number != 1 being a LIST of its type. */
typedef struct exg_type {
    int type; /* EXG_TYPE_* */
    int nullable;
    const char *name; /* column / field name; "item" for list elements */
    int n_children;
    const struct exg_type *children;
} exg_type;

/* One DuckDB vector of a chunk, same shape as its exg_type: flat data + validity (bit i = element i valid; NULL = all
 * valid) + children.  LIST: data = exg_list_entry_t[length] whose offsets are relative to children[0] as handed out
 * here (the child vector of THIS chunk).  STRUCT: data = NULL.  All memory is owned by the chunk's keepalive. */
typedef struct exg_vector {
    void *data;
    uint64_t *validity;
    uint64_t length;
    int n_children;
    const struct exg_vector *children;
} exg_vector;

typedef struct exg_schema {
    int n_columns;
    const char *names[16]; /* owned by the reader */
    int types[16];         /* EXG_TYPE_* of the column itself */
    int nullable[16];
    const exg_type *tree[16]; /* the column's full type (owned by the reader) */
} exg_schema;

typedef struct exg_chunk {
    uint64_t n_rows;           /* 0 => end of stream */
    int n_columns;
    void *data[16];            /* host pointers: exg_string_t[n_rows] / int64_t[] / float[] / exg_list_entry_t[] (STRUCT: NULL) */
    uint64_t *validity[16];    /* NULL => all valid; else ceil(n_rows/64) words */
    void *keepalive;           /* opaque; payload + vectors stay valid until exg_release_chunk */
    const exg_vector *vectors[16]; /* every column as a vector tree (data / validity above are vectors[c]'s own); NULL (and
                                * data[c] NULL) for a column outside exg_open_args.columns */
    uint64_t batch_no;         /* which device batch of this reader the rows come from (0, 1, ...: non-decreasing; what the
                                * table function reports as DuckDB's batch index, module.cpp has none: MaxThreads() == 1) */
} exg_chunk;

int exg_open(const exg_open_args *args, exg_reader **out);
/* How many byte-range shards a scan of this input is worth, and on which device each runs: what the table function's
 * init_global asks (MaxThreads() = *n_shards; init_local i opens shard i with device = devices[i]) — the in-process form
 * of SURVEY §8 E1: one host thread and one GPU per shard, nothing exchanged.  One shard per visible device when the input
 * can be sharded (text FASTQ / VCF / FASTA; BGZF FASTQ / VCF / FASTA, by members; multi-frame zstd, by frames) and holds
 * >= 256 MiB per shard, else 1 (single-member gzip, a single zstd frame).
 * EXON_GPU_SHARDS=n forces n (several shards on one device: how the tests exercise it on a 1-GPU box). */
int exg_plan_shards(const exg_open_args *args, uint32_t *n_shards, int *devices, uint32_t devices_cap);
/* VCF: opens the first file (the INFO / FORMAT keys of its header are the schema), so it can fail like a scan can. */
int exg_schema_of(exg_reader *r, exg_schema *out);
int exg_next_chunk(exg_reader *r, exg_chunk *out);
void exg_release_chunk(exg_reader *r, exg_chunk *chunk);
/* COUNT(*) fast path: no column is materialised. */
int exg_count_only(exg_reader *r, uint64_t *n_rows);
const char *exg_reader_error(exg_reader *r);
/* What a reader holds and has done so far.  Memory does not grow with the input: plain files travel in device batches,
 * compressed ones are decoded into a bounded stream of segments (like the reference's BufReader + convert_stream,
 * rust/src/arrow_reader.rs:60-91, 116-153).  EXG_DEVICE_MEM_CAP_MB=<MiB> (read at exg_open) sizes both from a budget. */
typedef struct exg_reader_stats {
    uint64_t device_bytes_now;   /* device memory held on behalf of this reader (pool size classes) */
    uint64_t device_bytes_peak;  /* ... its high-water mark */
    uint64_t device_mem_cap;     /* EXG_DEVICE_MEM_CAP_MB in bytes, 0 = not set */
    uint64_t device_batch_bytes; /* bytes per device batch / decoded segment */
    uint64_t device_batches;     /* scans launched */
    uint64_t decoded_segments;   /* segments of a compressed input consumed */
    uint64_t scan_algo;          /* EXG_ALGO_* the next device batch starts with: EXG_ALGO_FUSED until a batch came back with
                                  * EXG_RF_REDO (long reads, reads below ~45 bp, multi-sample VCF lines, bytes >= 0x80), then
                                  * EXG_ALGO_FUSED_FULL for the rest of the input (a fan-out reader: 0) */
    uint64_t input_bytes;        /* ABI 8: size of the reader's input files on disk (all files of a directory; a shard: the whole
                                  * files), what TableFunction::cardinality estimates rows from (module.cpp:307) */
    uint64_t input_compression;  /* ABI 8: 0 plain text, 1 gzip / BGZF, 2 zstd (input_bytes are compressed bytes then) */
    uint64_t nested_ns;          /* ABI 9: read_vcf — wall time the reader's thread spent making the nested columns (id / alt / filter /
                                  * info / formats) of its batches on the device, counting passes, prefix sums and children, up to the
                                  * point where their vectors start for the host (exg_vcf_nested.hpp) */
    uint64_t host_vector_bytes;  /* ABI 9: bytes of column vectors (flat and nested, validity included) sent to the host so far —
                                  * what a scan INTO DataChunks pays on the D2H link next to a decoded input's own bytes */
    uint64_t reserved[3];
} exg_reader_stats;
int exg_reader_stats_of(exg_reader *r, exg_reader_stats *out);
/* Device buffers, pinned host blocks and HIP streams of closed readers are recycled process-wide (size classes, at most
 * EXG_POOL_MAX_GB = 64 GB of HBM): this gives them back (the extension's unload / idle path). */
void exg_trim_pools(void);
void exg_close(exg_reader *r);

/* ---- (3) reference-FFI compatible ------------------------------------------------------ */
/* exon/include/rust.hpp:11-13 */
typedef struct ReplacementScanResult {
    const char *file_type; /* "FASTA" | "FASTQ" | "VCF" (static storage) or NULL */
} ReplacementScanResult;
/* exon/include/rust.hpp:48, rust/src/arrow_reader.rs:173-197: last extension, skipping one
 * compression extension (gz, gzip, zst, zstd, bz2, bzip2, xz). */
ReplacementScanResult replacement_scan(const char *uri);

/* The Arrow C data / stream interface (the published ABI the reference exchanges batches through:
 * `struct ArrowArrayStream stream;` at exon/src/exon/arrow_table_function/module.cpp:82, 228). */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
    const char *format;
    const char *name;
    const char *metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema **children;
    struct ArrowSchema *dictionary;
    void (*release)(struct ArrowSchema *);
    void *private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void **buffers;
    struct ArrowArray **children;
    struct ArrowArray *dictionary;
    void (*release)(struct ArrowArray *);
    void *private_data;
};
#endif
#ifndef ARROW_C_STREAM_INTERFACE
#define ARROW_C_STREAM_INTERFACE
struct ArrowArrayStream {
    int (*get_schema)(struct ArrowArrayStream *, struct ArrowSchema *out);
    int (*get_next)(struct ArrowArrayStream *, struct ArrowArray *out);
    const char *(*get_last_error)(struct ArrowArrayStream *);
    void (*release)(struct ArrowArrayStream *);
    void *private_data;
};
#endif

/* exon/include/rust.hpp:7-9 */
typedef struct ReaderResult {
    const char *error; /* NULL on success; else a heap C string the caller may free() (the reference leaks it) */
} ReaderResult;

/* exon/include/rust.hpp:41-46, rust/src/arrow_reader.rs:38-166 — the reference's own entry point,
 * same name, same arguments, same results: fills *stream_ptr with a stream of record batches of at
 * most batch_size rows (a multiple of 64).  The tokenising, the typed VCF columns (LIST / STRUCT),
 * the `filters` predicate and the Arrow buffers themselves (offsets, values, validity) are produced
 * on the device; the host only copies them back and wires the ArrowArray structs.
 *   compression: NULL = by extension (:60-75), else DataFusion's FileCompressionType names (:77-91)
 *   file_format: "fasta" | "fastq" | "vcf"
 *   filters:     NULL / "" or the predicate text FilterToString renders (module.cpp:158-214):
 *                <column> (= | != | <> | < | <= | > | >=) <literal>, <column> IS [NOT] NULL, AND, OR
 *                with SQL precedence; it is applied as `SELECT * FROM exon_table WHERE <filters>` (:125-141). */
ReaderResult new_reader(struct ArrowArrayStream *stream_ptr, const char *uri, uintptr_t batch_size,
                        const char *compression, const char *file_format, const char *filters);

#ifdef __cplusplus
}
#endif
#endif /* EXON_GPU_H */
