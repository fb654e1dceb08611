// exon_tf_bench.cpp — TEST / BENCH SCAFFOLDING (libexon_tf_test.so, not the product):
//   * consumers that walk a reader's DataChunks without an interpreter in the loop (bench.py's file -> DataChunks legs): one
//     that only counts, and ones that fold EVERY row's content — every string_t dereferenced: length, prefix, pointer,
//     payload bytes — into a digest that is compared with what the input's generator says the rows are;
//   * the host-only introspection the CPU tests use (the postfix program of a `filters` text, the keys of a VCF header);
//   * a probe of the host side of the upload path (page cache -> pinned memory [-> HBM]) with N readers at once.
// The product library exports none of this (round 2's verdict: exg_synth_*, exg_drain_chunks, *_explain shipped in it).
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "exg_filter.hpp"
#include "exg_vcf_header.hpp"
#include "exon_gpu.h"

// the test library's own copy of the error sink of exg_common.hpp (the product's is not exported)
namespace exg {
static thread_local char t_msg[512];
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_msg, sizeof t_msg, fmt, ap);
    va_end(ap);
}
}  // namespace exg
extern "C" const char *exon_tf_support_error(void) { return exg::t_msg; }

// ---- digests -------------------------------------------------------------------------------------------------------
static inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// a field's bytes folded 8 at a time (the verify passes hash tens of GB)
static inline uint64_t fold_bytes(uint64_t h, const uint8_t *p, size_t n) {
    h ^= n * 0x9E3779B97F4A7C15ull;
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0x100000001B3ull;
        h ^= h >> 29;
        p += 8, n -= 8;
    }
    uint64_t w = 0;
    if (n) memcpy(&w, p, n);
    h = (h ^ w ^ ((uint64_t)n << 56)) * 0x100000001B3ull;
    return h ^ (h >> 32);
}
static inline uint64_t fold_string_t(uint64_t h, const exg_string_t &s, bool valid) {
    if (!valid) return fold_bytes(h ^ 0xDEADull, nullptr, 0);
    const uint32_t len = s.inlined.length;
    const uint8_t *p = len <= EXG_INLINE_LENGTH ? (const uint8_t *)s.inlined.inlined : (const uint8_t *)(uintptr_t)s.pointer.ptr;
    if (len > EXG_INLINE_LENGTH && memcmp(s.pointer.prefix, p, 4) != 0) h ^= 0xBADBADull;  // the prefix must be the first four bytes
    return fold_bytes(h, p, len);
}
static inline bool bit(const uint64_t *words, uint64_t i) { return !words || ((words[i >> 6] >> (i & 63)) & 1); }

// ---- the FASTQ-150 generator of SURVEY §8 D2, on the host (csrc/testing/exg_synth.hip is the device form) -------------------
static inline uint64_t synth_word(uint64_t seed, uint64_t k, uint64_t j) { return mix64((seed ^ (k * 0x9E3779B97F4A7C15ull)) + j); }
static void synth_fastq_record(uint64_t seed, uint64_t k, uint8_t *rec) {  // 332 bytes
    memcpy(rec, "@SYN", 4);
    uint64_t v = k % 1000000000000ull;
    for (int i = 15; i >= 4; i--) rec[i] = (uint8_t)('0' + v % 10), v /= 10;
    rec[16] = ' ';
    rec[17] = (uint8_t)('0' + k % 4);
    memcpy(rec + 18, ":N:0:ACGT", 9);
    rec[27] = '\n';
    for (uint64_t j = 0; j < 5; j++) {
        const uint64_t w = synth_word(seed, k, j);
        for (uint64_t i = 0; i < 32 && j * 32 + i < 150; i++) rec[28 + j * 32 + i] = (uint8_t) "ACGT"[(w >> (2 * i)) & 3];
    }
    rec[178] = '\n', rec[179] = '+', rec[180] = '\n';
    for (uint64_t j = 0; j < 19; j++) {
        const uint64_t w = synth_word(seed, k, 8 + j);
        for (uint64_t i = 0; i < 8 && j * 8 + i < 150; i++) rec[181 + j * 8 + i] = (uint8_t)('!' + ((((w >> (8 * i)) & 0xFF) * 41) >> 8));
    }
    rec[331] = '\n';
}
static inline uint64_t fastq_row_digest(uint64_t k, const uint8_t *name, size_t n_name, const uint8_t *desc, size_t n_desc, bool desc_valid,
                                        const uint8_t *seq, size_t n_seq, const uint8_t *qual, size_t n_qual) {
    uint64_t h = mix64(k);
    h = fold_bytes(h, name, n_name);
    h = desc_valid ? fold_bytes(h, desc, n_desc) : fold_bytes(h ^ 0xDEADull, nullptr, 0);
    h = fold_bytes(h, seq, n_seq);
    h = fold_bytes(h, qual, n_qual);
    return mix64(h);
}

// records [first, first + n) of the synthetic FASTQ-150 file into out (332 n bytes): bench.py's workers deflate BASELINE config
// 4's input straight from these slices (the plain 19 GB file is never written)
extern "C" void exon_tf_synth_fastq150_host(uint64_t seed, uint64_t first, uint64_t n, uint8_t *out) {
    for (uint64_t i = 0; i < n; i++) synth_fastq_record(seed, first + i, out + 332 * i);
}

// what the rows [first, first + n) of the synthetic FASTQ-150 file are, as the digest exon_tf_drain_digest computes from a
// reader's chunks: the sum over rows of a hash of (row index, name, description, sequence, quality_scores)
extern "C" uint64_t exon_tf_expect_fastq150(uint64_t seed, uint64_t first, uint64_t n, int threads) {
    if (threads < 1) threads = 1;
    std::vector<uint64_t> part((size_t)threads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            uint8_t rec[332];
            uint64_t acc = 0;
            for (uint64_t i = (uint64_t)t; i < n; i += (uint64_t)threads) {
                const uint64_t k = first + i;
                synth_fastq_record(seed, k, rec);
                acc += fastq_row_digest(i, rec + 1, 15, rec + 17, 10, true, rec + 28, 150, rec + 181, 150);
            }
            part[(size_t)t] = acc;
        });
    uint64_t total = 0;
    for (int t = 0; t < threads; t++) th[(size_t)t].join(), total += part[(size_t)t];
    return total;
}

// A VCF file's data lines as that digest, by a split of its own (lines at '\n', fields at '\t'): row index, CHROM, POS as a
// number, REF — independent of the engine's tokeniser.
extern "C" int exon_tf_expect_vcf_file(const char *path, uint64_t *rows, uint64_t *digest) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    const uint8_t *d = n ? (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (n && d == MAP_FAILED) return -1;
    uint64_t k = 0, acc = 0;
    for (size_t pos = 0; pos < n;) {
        const uint8_t *nl = (const uint8_t *)memchr(d + pos, '\n', n - pos);
        size_t end = nl ? (size_t)(nl - d) : n;
        const size_t next = nl ? end + 1 : n;
        if (end > pos && d[end - 1] == '\r') end--;
        if (end > pos && d[pos] != '#') {
            const uint8_t *f[5];
            size_t fl[5];
            size_t q = pos;
            int nf = 0;
            while (nf < 5 && q <= end) {
                const uint8_t *tab = (const uint8_t *)memchr(d + q, '\t', end - q);
                const size_t fe = tab ? (size_t)(tab - d) : end;
                f[nf] = d + q, fl[nf] = fe - q, nf++;
                q = fe + 1;
            }
            if (nf >= 4) {
                int64_t p = 0;
                for (size_t i = 0; i < fl[1]; i++) p = p * 10 + (f[1][i] - '0');
                uint64_t h = mix64(k);
                h = fold_bytes(h, f[0], fl[0]);
                h = fold_bytes(h, (const uint8_t *)&p, 8);
                h = fold_bytes(h, f[3], fl[3]);
                acc += mix64(h);
                k++;
            }
        }
        pos = next;
    }
    if (d) munmap((void *)d, n);
    *rows = k;
    *digest = acc;
    return 0;
}

// A well-formed FASTQ file's records as that digest, by a split of its own (four lines per record at '\n', the name line at its
// first space, CR stripped): row index, name, description (NULL when empty), sequence, quality — independent of the engine's
// tokeniser and of the oracle; `threads` workers each take a run of whole records (found from a byte offset by the '@' ... '+'
// line pattern of this generator's files: names without '\n@' ambiguity are the caller's business).
extern "C" int exon_tf_expect_fastq_file(const char *path, uint64_t *rows, uint64_t *digest) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    const uint8_t *d = n ? (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (n && d == MAP_FAILED) return -1;
    uint64_t k = 0, acc = 0;
    size_t pos = 0;
    int rc = 0;
    while (pos < n) {
        const uint8_t *ls[4];
        size_t ll[4];
        for (int i = 0; i < 4; i++) {
            const uint8_t *nl = pos < n ? (const uint8_t *)memchr(d + pos, '\n', n - pos) : nullptr;
            size_t end = nl ? (size_t)(nl - d) : n;
            ls[i] = d + pos;
            ll[i] = end - pos;
            if (nl && ll[i] && ls[i][ll[i] - 1] == '\r') ll[i]--;
            pos = nl ? end + 1 : n;
        }
        if (!ll[0] || ls[0][0] != '@' || !ll[2] || ls[2][0] != '+') {
            rc = -2;
            break;
        }
        const uint8_t *name = ls[0] + 1;
        size_t n_name = ll[0] - 1;
        const uint8_t *sp = (const uint8_t *)memchr(name, ' ', n_name);
        const uint8_t *desc = sp ? sp + 1 : name + n_name;
        const size_t n_desc = sp ? (size_t)(name + n_name - desc) : 0;
        if (sp) n_name = (size_t)(sp - name);
        acc += fastq_row_digest(k, name, n_name, desc, n_desc, n_desc != 0, ls[1], ll[1], ls[3], ll[3]);
        k++;
    }
    if (d) munmap((void *)d, n);
    *rows = k;
    *digest = acc;
    return rc;
}

// A well-formed FASTA file's records as the digest of exon_tf_drain_digest_from's kind 2, by a split of its own: a '>' line begins
// a record, its id ends at the first blank, the description is what follows with the blanks around it dropped (NULL when nothing
// does), the sequence is the lines up to the next '>' line joined without their ends — independent of the engine's tokeniser and of
// the oracle.
extern "C" int exon_tf_expect_fasta_file(const char *path, uint64_t *rows, uint64_t *digest) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    const uint8_t *d = n ? (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (n && d == MAP_FAILED) return -1;
    uint64_t k = 0, acc = 0, h = 0;
    bool open_rec = false;
    std::vector<uint8_t> seq;
    auto flush = [&] {
        if (!open_rec) return;
        h = fold_bytes(h, seq.data(), seq.size());
        acc += mix64(h);
        k++;
        open_rec = false;
    };
    int rc = 0;
    for (size_t pos = 0; pos < n;) {
        const uint8_t *nl = (const uint8_t *)memchr(d + pos, '\n', n - pos);
        size_t end = nl ? (size_t)(nl - d) : n;
        const size_t next = nl ? end + 1 : n;
        if (end > pos && d[end - 1] == '\r') end--;
        if (end > pos && d[pos] == '>') {
            flush();
            const uint8_t *id = d + pos + 1, *e = d + end, *q = id;
            while (q < e && *q != ' ' && *q != '\t') q++;
            const uint8_t *ds = q, *de = e;
            while (ds < de && (*ds == ' ' || *ds == '\t')) ds++;
            while (de > ds && (de[-1] == ' ' || de[-1] == '\t')) de--;
            h = mix64(k);
            h = fold_bytes(h, id, (size_t)(q - id));
            h = de > ds ? fold_bytes(h, ds, (size_t)(de - ds)) : fold_bytes(h ^ 0xDEADull, nullptr, 0);
            seq.clear();
            open_rec = true;
        } else if (open_rec) {
            seq.insert(seq.end(), d + pos, d + end);
        } else if (end > pos) {
            rc = -2;  // bytes in front of the first '>'
            break;
        }
        pos = next;
    }
    flush();
    if (d) munmap((void *)d, n);
    *rows = k;
    *digest = acc;
    return rc;
}

// ---- consumers of a reader's chunks ------------------------------------------------------------------------------------------
// pull and release every remaining chunk — what a consumer that only walks the DataChunks does
extern "C" int exon_tf_drain_chunks(exg_reader *r, uint64_t *n_rows, uint64_t *n_chunks) {
    if (!r || !n_rows || !n_chunks) return EXG_E_INVALID_ARG;
    *n_rows = *n_chunks = 0;
    for (;;) {
        exg_chunk c;
        const int rc = exg_next_chunk(r, &c);
        if (rc) return rc;
        if (c.n_rows == 0) return EXG_OK;
        *n_rows += c.n_rows;
        *n_chunks += 1;
        exg_release_chunk(r, &c);
    }
}

// the same, folding every row's content.  kind 0: four VARCHAR columns (FASTQ: name, description, sequence, quality_scores);
// kind 1: VCF — chrom (column 0), pos (BIGINT, column 1), ref (column 3); kind 2: FASTA — id, description, sequence.  *bad: rows whose name / sequence / quality length is
// not the one given (0: not checked).
// first_row: the index of the reader's first row in the whole file (a shard's rows are rows [first_row, first_row + n) of the
// file: the digests of all shards then add up to the file's).
extern "C" int exon_tf_drain_digest_from(exg_reader *r, int kind, uint32_t want_seq_len, uint64_t first_row, uint64_t *n_rows, uint64_t *n_chunks,
                                         uint64_t *digest, uint64_t *bad) {
    if (!r || !n_rows || !n_chunks || !digest || !bad) return EXG_E_INVALID_ARG;
    *n_rows = *n_chunks = *digest = *bad = 0;
    uint64_t k = first_row, acc = 0;
    for (;;) {
        exg_chunk c;
        const int rc = exg_next_chunk(r, &c);
        if (rc) return rc;
        if (c.n_rows == 0) break;
        if (kind == 0) {
            const exg_string_t *name = (const exg_string_t *)c.data[0], *desc = (const exg_string_t *)c.data[1];
            const exg_string_t *seq = (const exg_string_t *)c.data[2], *qual = (const exg_string_t *)c.data[3];
            for (uint64_t i = 0; i < c.n_rows; i++) {
                uint64_t h = mix64(k + i);
                h = fold_string_t(h, name[i], true);
                h = fold_string_t(h, desc[i], bit(c.validity[1], i));
                h = fold_string_t(h, seq[i], true);
                h = fold_string_t(h, qual[i], true);
                acc += mix64(h);
                if (want_seq_len && (seq[i].inlined.length != want_seq_len || qual[i].inlined.length != want_seq_len)) (*bad)++;
            }
        } else if (kind == 2) {
            const exg_string_t *id = (const exg_string_t *)c.data[0], *desc = (const exg_string_t *)c.data[1], *seq = (const exg_string_t *)c.data[2];
            for (uint64_t i = 0; i < c.n_rows; i++) {
                uint64_t h = mix64(k + i);
                h = fold_string_t(h, id[i], true);
                h = fold_string_t(h, desc[i], bit(c.validity[1], i));
                h = fold_string_t(h, seq[i], true);
                acc += mix64(h);
            }
        } else {
            const exg_string_t *chrom = (const exg_string_t *)c.data[0], *ref = (const exg_string_t *)c.data[3];
            const int64_t *pos = (const int64_t *)c.data[1];
            for (uint64_t i = 0; i < c.n_rows; i++) {
                uint64_t h = mix64(k + i);
                h = fold_string_t(h, chrom[i], true);
                h = fold_bytes(h, (const uint8_t *)&pos[i], 8);
                h = fold_string_t(h, ref[i], true);
                acc += mix64(h);
            }
        }
        k += c.n_rows;
        *n_chunks += 1;
        exg_release_chunk(r, &c);
    }
    *n_rows = k - first_row;
    *digest = acc;
    return EXG_OK;
}
extern "C" int exon_tf_drain_digest(exg_reader *r, int kind, uint32_t want_seq_len, uint64_t *n_rows, uint64_t *n_chunks, uint64_t *digest,
                                    uint64_t *bad) {
    return exon_tf_drain_digest_from(r, kind, want_seq_len, 0, n_rows, n_chunks, digest, bad);
}

// read_vcf's `formats` column — LIST(STRUCT(<##FORMAT keys>)) — walked like a DuckDB operator would: every row's list entry,
// every sample's value of the struct's child `key` (a VARCHAR key: GT) folded into a digest; *samples = list elements in all.
extern "C" int exon_tf_drain_formats_digest(exg_reader *r, int key, uint64_t *n_rows, uint64_t *n_samples, uint64_t *digest) {
    if (!r || !n_rows || !n_samples || !digest) return EXG_E_INVALID_ARG;
    *n_rows = *n_samples = *digest = 0;
    uint64_t k = 0, acc = 0, ns = 0;
    for (;;) {
        exg_chunk c;
        const int rc = exg_next_chunk(r, &c);
        if (rc) return rc;
        if (c.n_rows == 0) break;
        const exg_vector *fl = c.vectors[8];
        if (!fl || fl->n_children != 1 || fl->children[0].n_children <= key) {
            exg_release_chunk(r, &c);
            return EXG_E_INVALID_ARG;
        }
        const exg_list_entry_t *ent = (const exg_list_entry_t *)fl->data;
        const exg_vector &item = fl->children[0];
        const exg_vector &gt = item.children[key];
        const exg_string_t *str = (const exg_string_t *)gt.data;
        for (uint64_t i = 0; i < c.n_rows; i++) {
            uint64_t h = mix64(k + i);
            if (ent[i].offset + ent[i].length > item.length) h ^= 0xBAD0FF5E7ull;  // an entry outside the chunk's child vector
            else
                for (uint64_t e = ent[i].offset; e < ent[i].offset + ent[i].length; e++) h = fold_string_t(h, str[e], bit(gt.validity, e));
            ns += ent[i].length;
            acc += mix64(h);
        }
        k += c.n_rows;
        exg_release_chunk(r, &c);
    }
    *n_rows = k;
    *n_samples = ns;
    *digest = acc;
    return EXG_OK;
}
// ... and what that digest must be, by a split of the file of its own: lines at '\n', fields at '\t', the FORMAT field at ':' to find
// where `key_name` stands, every sample at ':'; "." is NULL.
extern "C" int exon_tf_expect_vcf_formats_file(const char *path, const char *key_name, uint64_t *rows, uint64_t *samples, uint64_t *digest) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    const uint8_t *d = n ? (const uint8_t *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (n && d == MAP_FAILED) return -1;
    const size_t kn = strlen(key_name);
    uint64_t k = 0, acc = 0, ns = 0;
    for (size_t pos = 0; pos < n;) {
        const uint8_t *nl = (const uint8_t *)memchr(d + pos, '\n', n - pos);
        size_t end = nl ? (size_t)(nl - d) : n;
        const size_t next = nl ? end + 1 : n;
        if (end > pos && d[end - 1] == '\r') end--;
        if (end > pos && d[pos] != '#') {
            uint64_t h = mix64(k);
            size_t q = pos;
            int field = 0, key_pos = -1;
            while (q <= end) {
                const uint8_t *tab = (const uint8_t *)memchr(d + q, '\t', end - q);
                const size_t fe = tab ? (size_t)(tab - d) : end;
                if (field == 8) {  // FORMAT
                    int p = 0;
                    for (size_t a = q; a <= fe;) {
                        const uint8_t *col = (const uint8_t *)memchr(d + a, ':', fe - a);
                        const size_t ke = col ? (size_t)(col - d) : fe;
                        if (key_pos < 0 && ke - a == kn && memcmp(d + a, key_name, kn) == 0) key_pos = p;
                        p++;
                        a = ke + 1;
                    }
                } else if (field > 8) {
                    int p = 0;
                    bool found = false;
                    for (size_t a = q; a <= fe;) {
                        const uint8_t *col = (const uint8_t *)memchr(d + a, ':', fe - a);
                        const size_t ve = col ? (size_t)(col - d) : fe;
                        if (p == key_pos) {
                            const bool null = ve - a == 1 && d[a] == '.';
                            h = null ? fold_bytes(h ^ 0xDEADull, nullptr, 0) : fold_bytes(h, d + a, ve - a);
                            found = true;
                            break;
                        }
                        p++;
                        a = ve + 1;
                    }
                    if (!found) h = fold_bytes(h ^ 0xDEADull, nullptr, 0);
                    ns++;
                }
                field++;
                if (!tab) break;
                q = fe + 1;
            }
            acc += mix64(h);
            k++;
        }
        pos = next;
    }
    if (d) munmap((void *)d, n);
    *rows = k;
    *samples = ns;
    *digest = acc;
    return 0;
}

// ---- the reference's own boundary: new_reader -> Arrow C stream (exon/include/rust.hpp:41-46) ----------------------------------
// Pulls every record batch of a FASTQ stream through the Arrow callbacks and releases it — what DuckDB's ArrowToDuckDB consumer
// of the reference does per batch, minus the conversion.  with_digest: every row's four Utf8 values (int32 offsets + value bytes
// + the description's validity bitmap) folded into the digest exon_tf_expect_fastq150 predicts (untimed verification pass).
extern "C" int exon_tf_drain_arrow_fastq(const char *path, const char *compression, const char *filters, int with_digest, uint64_t *n_rows,
                                         uint64_t *n_batches, uint64_t *digest, char *err, size_t err_cap) {
    if (!path || !n_rows || !n_batches || !digest) return EXG_E_INVALID_ARG;
    *n_rows = *n_batches = *digest = 0;
    auto say = [&](const char *m) {
        if (err && err_cap) snprintf(err, err_cap, "%s", m ? m : "");
    };
    ArrowArrayStream stream;
    memset(&stream, 0, sizeof stream);
    const ReaderResult rr = new_reader(&stream, path, EXG_VECTOR_SIZE, compression, "fastq", filters);
    if (rr.error) {
        say(rr.error);
        return EXG_E_IO;
    }
    uint64_t k = 0, acc = 0;
    int rc = EXG_OK;
    for (;;) {
        ArrowArray a;
        memset(&a, 0, sizeof a);
        if (stream.get_next(&stream, &a) != 0) {
            say(stream.get_last_error ? stream.get_last_error(&stream) : "get_next failed");
            rc = EXG_E_PARSE;
            break;
        }
        if (!a.release) break;  // end of stream
        if (with_digest && a.n_children >= 4) {
            const uint8_t *vals[4];
            const int32_t *offs[4];
            const uint8_t *valid1 = nullptr;
            int64_t o[4];
            for (int c = 0; c < 4; c++) {
                const ArrowArray *ch = a.children[c];
                offs[c] = (const int32_t *)ch->buffers[1];
                vals[c] = (const uint8_t *)ch->buffers[2];
                o[c] = ch->offset;
                if (c == 1) valid1 = ch->null_count != 0 ? (const uint8_t *)ch->buffers[0] : nullptr;
            }
            for (int64_t i = 0; i < a.length; i++) {
                const uint8_t *f[4];
                size_t n[4];
                for (int c = 0; c < 4; c++) {
                    const int32_t b = offs[c][o[c] + i], e = offs[c][o[c] + i + 1];
                    f[c] = vals[c] + b, n[c] = (size_t)(e - b);
                }
                const bool dv = !valid1 || ((valid1[(o[1] + i) >> 3] >> ((o[1] + i) & 7)) & 1);
                acc += fastq_row_digest(k + (uint64_t)i, f[0], n[0], f[1], n[1], dv, f[2], n[2], f[3], n[3]);
            }
        }
        k += (uint64_t)a.length;
        *n_batches += 1;
        a.release(&a);
    }
    if (stream.release) stream.release(&stream);
    *n_rows = k;
    *digest = acc;
    return rc;
}

// The same for read_vcf's stream (the reference's schema: chrom Utf8, pos Int64, id List<Utf8>, ref Utf8, alt List<Utf8>, qual Float32,
// filter List<Utf8>, info Struct, formats List<Struct>): every record batch pulled through the Arrow callbacks and released.
// with_digest: chrom, pos and ref of every row folded into the digest exon_tf_expect_vcf_file predicts; *list_elems = elements of
// the id + alt + filter lists and samples of formats in all (their last offsets: what a consumer of the nested arrays walks).
extern "C" int exon_tf_drain_arrow_vcf(const char *path, const char *compression, const char *filters, int with_digest, uint64_t *n_rows,
                                       uint64_t *n_batches, uint64_t *digest, uint64_t *list_elems, char *err, size_t err_cap) {
    if (!path || !n_rows || !n_batches || !digest || !list_elems) return EXG_E_INVALID_ARG;
    *n_rows = *n_batches = *digest = *list_elems = 0;
    auto say = [&](const char *m) {
        if (err && err_cap) snprintf(err, err_cap, "%s", m ? m : "");
    };
    ArrowArrayStream stream;
    memset(&stream, 0, sizeof stream);
    const ReaderResult rr = new_reader(&stream, path, EXG_VECTOR_SIZE, compression, "vcf", filters);
    if (rr.error) {
        say(rr.error);
        return EXG_E_IO;
    }
    uint64_t k = 0, acc = 0, elems = 0;
    int rc = EXG_OK;
    for (;;) {
        ArrowArray a;
        memset(&a, 0, sizeof a);
        if (stream.get_next(&stream, &a) != 0) {
            say(stream.get_last_error ? stream.get_last_error(&stream) : "get_next failed");
            rc = EXG_E_PARSE;
            break;
        }
        if (!a.release) break;  // end of stream
        if (a.n_children >= 9) {
            for (int c : {2, 4, 6, 8}) {  // List arrays: offsets[length] - offsets[0] elements
                const ArrowArray *ch = a.children[c];
                const int32_t *off = (const int32_t *)ch->buffers[1] + ch->offset;
                elems += (uint64_t)(off[ch->length] - off[0]);
            }
            if (with_digest) {
                const ArrowArray *chrom = a.children[0], *pos = a.children[1], *ref = a.children[3];
                const int32_t *co = (const int32_t *)chrom->buffers[1] + chrom->offset, *ro = (const int32_t *)ref->buffers[1] + ref->offset;
                const uint8_t *cv = (const uint8_t *)chrom->buffers[2], *rv = (const uint8_t *)ref->buffers[2];
                const int64_t *pv = (const int64_t *)pos->buffers[1] + pos->offset;
                for (int64_t i = 0; i < a.length; i++) {
                    uint64_t h = mix64(k + (uint64_t)i);
                    h = fold_bytes(h, cv + co[i], (size_t)(co[i + 1] - co[i]));
                    h = fold_bytes(h, (const uint8_t *)&pv[i], 8);
                    h = fold_bytes(h, rv + ro[i], (size_t)(ro[i + 1] - ro[i]));
                    acc += mix64(h);
                }
            }
        }
        k += (uint64_t)a.length;
        *n_batches += 1;
        a.release(&a);
    }
    stream.release(&stream);
    *n_rows = k;
    *digest = acc;
    *list_elems = elems;
    return rc;
}

// ---- host-only introspection (no device is touched): what the CPU tests check ---------------------------------------
// The postfix program a `filters` text compiles to, e.g.  "name = 'a' | pos >= 5 | AND".  Columns are those of the
// format's schema (VCF: the flat ones; nested columns are refused like in new_reader).  Returns 0, or -1 with the
// parser's message in `out`.
extern "C" int exon_tf_filter_explain(const char *file_format, const char *filters, char *out, size_t cap) {
    namespace ea = exg::arrow;
    std::vector<exg_rd::FilterColumn> cols;
    const std::string fmt = file_format ? file_format : "";
    if (fmt == "fastq")
        cols = {{"name", 'u'}, {"description", 'u'}, {"sequence", 'u'}, {"quality_scores", 'u'}};
    else if (fmt == "fasta")
        cols = {{"id", 'u'}, {"description", 'u'}, {"sequence", 'u'}};
    else
        cols = {{"chrom", 'u'}, {"pos", 'l'}, {"id", 'x'}, {"ref", 'u'}, {"alt", 'x'}, {"qual", 'f'}, {"filter", 'x'}, {"info", 'x'}, {"formats", 'x'}};
    const std::string text = filters ? filters : "";
    exg_rd::FilterParser fp(text, cols);
    std::string res;
    int rc = 0;
    if (!fp.parse()) {
        res = fp.err;
        rc = -1;
    } else {
        static const char *cmp[] = {"=", "!=", "<", "<=", ">", ">="};
        for (uint32_t k = 0; k < fp.prog.n_ops; k++) {
            const ea::FilterOp &op = fp.prog.ops[k];
            if (k) res += " | ";
            if (op.op == ea::kOpAnd) res += "AND";
            else if (op.op == ea::kOpOr) res += "OR";
            else if (op.op == ea::kOpIsNull) res += cols[op.col].name + " isnull";
            else if (op.op == ea::kOpIsNotNull) res += cols[op.col].name + " notnull";
            else {
                res += cols[op.col].name + " " + cmp[op.cmp] + " ";
                if (op.lit == ea::kLitStr) res += "'" + fp.consts.substr(op.str_off, op.str_len) + "'";
                else if (op.lit == ea::kLitInt) res += std::to_string(op.i);
                else {
                    char b[64];
                    snprintf(b, sizeof b, "%g", op.f);
                    res += b;
                }
            }
        }
    }
    if (out && cap) snprintf(out, cap, "%s", res.c_str());
    return rc;
}

// The INFO / FORMAT keys a VCF header declares, as "INFO DP:i AF:[f] DB:b ANN:u | FORMAT GT:u AD:[i]".
extern "C" int exon_tf_vcf_header_explain(const char *header, size_t n, char *out, size_t cap) {
    const std::string res = exg_rd::explain_vcf_header(header, n);
    if (out && cap) snprintf(out, cap, "%s", res.c_str());
    return 0;
}

// ---- the host side of the upload path with N readers at once ----------------------------------------------------------------
// Every reader runs what exg_rd_io.cpp's pread_parallel runs per window: `threads` threads read 8 MiB slices of ITS range
// of the file into a pinned block; with h2d != 0 every slice is then sent on to a device buffer (hipMemcpyAsync on the
// reader's stream), like an upload.  Each byte is copied page cache -> pinned block (-> DMA): on a node with 8 GPUs the
// readers of all of them share the host's memory system, and this probe says how far that goes (bench.py
// `host_pipeline_scaling`: aggregate GB/s at 1 / 2 / 4 / 8 readers).  -> aggregate bytes per second, or < 0.
extern "C" double exon_tf_host_pipeline_probe(const char *path, int n_readers, int threads, int h2d, int device, double seconds) {
    if (!path || n_readers < 1 || threads < 1) return -1;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    fstat(fd, &st);
    const uint64_t n = (uint64_t)st.st_size;
    const size_t slice = 8u << 20, window = 256u << 20;
    if (n < (uint64_t)n_readers * slice) {
        close(fd);
        return -1;
    }
    struct Reader {
        char *pin = nullptr;
        void *dev = nullptr;
        hipStream_t st = nullptr;
    };
    std::vector<Reader> rd((size_t)n_readers);
    (void)hipSetDevice(device);
    bool ok = true;
    for (auto &x : rd) {
        ok = ok && hipHostMalloc((void **)&x.pin, window, hipHostMallocDefault) == hipSuccess;
        if (h2d) ok = ok && hipMalloc(&x.dev, window) == hipSuccess && hipStreamCreateWithFlags(&x.st, hipStreamNonBlocking) == hipSuccess;
    }
    std::atomic<uint64_t> moved{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (ok)
        for (int r = 0; r < n_readers; r++) {
            const uint64_t lo = n / (uint64_t)n_readers * (uint64_t)r, span = n / (uint64_t)n_readers / slice * slice;
            auto next = std::make_shared<std::atomic<uint64_t>>(0);
            for (int t = 0; t < threads; t++)
                th.emplace_back([&, r, lo, span, next] {
                    (void)hipSetDevice(device);
                    while (!stop.load(std::memory_order_relaxed)) {
                        const uint64_t i = next->fetch_add(1);
                        const uint64_t off = lo + (i * slice) % span, w = (i * slice) % window;
                        size_t got = 0;
                        while (got < slice) {
                            const ssize_t k = pread(fd, rd[(size_t)r].pin + w + got, slice - got, (off_t)(off + got));
                            if (k <= 0) break;
                            got += (size_t)k;
                        }
                        if (h2d && got)
                            (void)hipMemcpyAsync((char *)rd[(size_t)r].dev + w, rd[(size_t)r].pin + w, got, hipMemcpyHostToDevice, rd[(size_t)r].st);
                        moved.fetch_add(got, std::memory_order_relaxed);
                    }
                });
        }
    struct timespec ts = {(time_t)seconds, (long)((seconds - (time_t)seconds) * 1e9)};
    nanosleep(&ts, nullptr);
    stop = true;
    for (auto &t : th) t.join();
    for (auto &x : rd)
        if (x.st) (void)hipStreamSynchronize(x.st);
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double dt = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    for (auto &x : rd) {
        if (x.st) (void)hipStreamDestroy(x.st);
        if (x.dev) (void)hipFree(x.dev);
        if (x.pin) (void)hipHostFree(x.pin);
    }
    close(fd);
    return ok ? (double)moved.load() / dt : -1;
}


// ---- the same question without the bounce copy ------------------------------------------------------------------------------
// mode 1: zero-bounce — every reader maps the file and, window by window (256 MiB), REGISTERS the mapping's pages with the
//         runtime (hipHostRegister: pins the page cache pages themselves), sends the window to the device straight from them
//         (hipMemcpyAsync from registered memory is a DMA like one from a pinned block) and unregisters it behind the copy; a
//         helper thread per reader registers one window ahead.  No CPU touches the bytes.  The windows cycle over ONE mapping:
//         from the second pass on their page-table entries exist (what a file read twice through one mapping sees).
// mode 3: the same with a FRESH mapping per window (mmap, register, send, unregister, munmap): what a reader that sees every
//         byte once pays — the page-table entries of 65 536 pages per window are made by the registration;
// mode 4: mode 3 with the window's entries made first by four threads (madvise MADV_POPULATE_READ on a quarter each).
// mode 2: O_DIRECT — pread with O_DIRECT into the pinned block (no page cache, no second copy; what a cold file wants), then
//         the H2D copy.  Returns -2 when the file system refuses O_DIRECT (tmpfs does).
// -> aggregate bytes per second over `seconds`, or < 0.  *register_ms (mode 1): average ms per hipHostRegister + Unregister pair.
extern "C" double exon_tf_host_zero_bounce_probe(const char *path, int n_readers, int mode, int device, double seconds, double *register_ms) {
    if (register_ms) *register_ms = 0;
    if (!path || n_readers < 1 || mode < 1 || mode > 4) return -1;
    int fd = open(path, O_RDONLY | (mode == 2 ? O_DIRECT : 0));
    if (fd < 0) return mode == 2 ? -2 : -1;
    struct stat st;
    fstat(fd, &st);
    const uint64_t n = (uint64_t)st.st_size;
    const size_t window = 256u << 20;
    if (n < (uint64_t)n_readers * window) {
        close(fd);
        return -1;
    }
    (void)hipSetDevice(device);
    std::atomic<uint64_t> moved{0};
    std::atomic<bool> stop{false}, failed{false};
    std::atomic<uint64_t> reg_ns{0}, reg_n{0};
    auto now_ns = [] {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec;
    };
    char *map = nullptr;
    if (mode != 2) {
        map = (char *)mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) {
            close(fd);
            return -1;
        }
    }
    std::vector<std::thread> th;
    const uint64_t t0 = now_ns();
    for (int r = 0; r < n_readers; r++)
        th.emplace_back([&, r] {
            (void)hipSetDevice(device);
            void *dev = nullptr;
            hipStream_t s = nullptr;
            char *pin = nullptr;
            if (hipMalloc(&dev, window) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
                failed = true;
                return;
            }
            const uint64_t lo = n / (uint64_t)n_readers * (uint64_t)r, span = n / (uint64_t)n_readers / window * window;
            if (mode == 2) {
                if (hipHostMalloc((void **)&pin, window, hipHostMallocDefault) != hipSuccess) failed = true;
                for (uint64_t i = 0; !stop && !failed; i++) {
                    const uint64_t off = lo + (i * window) % span;
                    size_t got = 0;
                    while (got < window) {
                        const ssize_t k = pread(fd, pin + got, window - got, (off_t)(off + got));
                        if (k <= 0) {
                            failed = true;
                            break;
                        }
                        got += (size_t)k;
                    }
                    if (hipMemcpyAsync(dev, pin, window, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) failed = true;
                    moved += window;
                }
            } else {
                // window i is registered by the helper while window i - 1 travels
                std::mutex mu;
                std::condition_variable cv;
                uint64_t registered = 0, released = 0;  // windows registered / unregistered so far
                bool done = false;
                char *fresh[4] = {nullptr, nullptr, nullptr, nullptr};  // modes 3 / 4: the windows' own mappings
                std::thread helper([&] {
                    (void)hipSetDevice(device);
                    for (uint64_t i = 0;; i++) {
                        {
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return done || i < released + 2; });  // at most two windows pinned at a time
                            if (done) return;
                        }
                        const uint64_t t1 = now_ns();
                        char *w = map + lo + (i * window) % span;
                        if (mode >= 3) {
                            w = (char *)mmap(nullptr, window, PROT_READ, MAP_SHARED, fd, (off_t)(lo + (i * window) % span));
                            if (w == MAP_FAILED) {
                                failed = true;
                                w = nullptr;
                            }
                            if (w && mode == 4) {
                                std::vector<std::thread> pop;
                                for (int q = 0; q < 4; q++) pop.emplace_back([w, q] { (void)madvise(w + (size_t)q * (window / 4), window / 4, 22 /* MADV_POPULATE_READ */); });
                                for (auto &t : pop) t.join();
                            }
                            std::lock_guard<std::mutex> g(mu);
                            fresh[i & 3] = w;
                        }
                        if (w && hipHostRegister(w, window, hipHostRegisterDefault) != hipSuccess) {
                            (void)hipGetLastError();
                            failed = true;
                        }
                        reg_ns += now_ns() - t1;
                        std::lock_guard<std::mutex> g(mu);
                        registered = i + 1;
                        cv.notify_all();
                        if (failed) return;
                    }
                });
                for (uint64_t i = 0; !stop && !failed; i++) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return registered > i || failed.load(); });
                    }
                    if (failed) break;
                    char *src = map + lo + (i * window) % span;
                    if (mode >= 3) {
                        std::lock_guard<std::mutex> g(mu);
                        src = fresh[i & 3];
                    }
                    if (hipMemcpyAsync(dev, src, window, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) failed = true;
                    const uint64_t t1 = now_ns();
                    (void)hipHostUnregister(src);
                    if (mode >= 3) munmap(src, window);
                    reg_ns += now_ns() - t1;
                    reg_n++;
                    moved += window;
                    std::lock_guard<std::mutex> g(mu);
                    released = i + 1;
                    cv.notify_all();
                }
                {
                    std::lock_guard<std::mutex> g(mu);
                    done = true;
                    cv.notify_all();
                }
                helper.join();
                // (a window the helper registered and nobody sent)
                {
                    for (uint64_t i = released; i < registered; i++) {
                        char *w = mode >= 3 ? fresh[i & 3] : map + lo + (i * window) % span;
                        if (!w) continue;
                        (void)hipHostUnregister(w);
                        if (mode >= 3) munmap(w, window);
                    }
                }
            }
            if (pin) (void)hipHostFree(pin);
            (void)hipStreamDestroy(s);
            (void)hipFree(dev);
        });
    struct timespec ts = {(time_t)seconds, (long)((seconds - (time_t)seconds) * 1e9)};
    nanosleep(&ts, nullptr);
    stop = true;
    for (auto &t : th) t.join();
    const double dt = (double)(now_ns() - t0) * 1e-9;
    if (map) munmap(map, n);
    close(fd);
    if (register_ms && reg_n.load()) *register_ms = (double)reg_ns.load() / (double)reg_n.load() * 1e-6;
    return failed ? -1 : (double)moved.load() / dt;
}
