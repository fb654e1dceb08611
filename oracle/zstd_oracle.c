/*
 * zstd_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle/exon_oracle.c for the rule): a plain CPU restatement of the
 * Zstandard frame format, RFC 8878, used to cross-check the device decoder (exon_duckdb_amd/csrc/exg_zstd*.hip) stage
 * by stage.  The reference reaches zstd through DataFusion 28 FileCompressionType::ZSTD -> async-compression 0.4.0 ->
 * zstd 0.12.3+zstd.1.5.2 (rust/Cargo.lock:3875-3876; selected at rust/src/arrow_reader.rs:73 and :87-88); libzstd itself
 * is the checker of record in tests/ (ctypes on libzstd.so.1) — this file exists so that the intermediate products of a
 * decode (literals, sequences, per-block sizes) can be compared too, and to pin down the format before the kernels.
 *
 * Only tests/ may load this.  The product path never does.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ZSO_OK 0
#define ZSO_E_TRUNCATED 1
#define ZSO_E_MAGIC 2
#define ZSO_E_RESERVED 3
#define ZSO_E_WINDOW 4
#define ZSO_E_DICT 5
#define ZSO_E_BLOCK 6
#define ZSO_E_LITERALS 7
#define ZSO_E_HUF 8
#define ZSO_E_FSE 9
#define ZSO_E_SEQ 10
#define ZSO_E_OFFSET 11
#define ZSO_E_CAPACITY 12
#define ZSO_E_SIZE 13
#define ZSO_E_CHECKSUM 14

#define BLOCK_MAX (128u << 10)

static int hb(uint32_t v) { return 31 - __builtin_clz(v); }

/* ---- XXH64 (the frame's content checksum is its low 32 bits, seed 0) ---- */
#define P1 11400714785074694791ULL
#define P2 14029467366897019727ULL
#define P3 1609587929392839161ULL
#define P4 9650029242287828579ULL
#define P5 2870177450012600261ULL
static uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t xround(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
static uint64_t xmerge(uint64_t h, uint64_t v) { return (h ^ xround(0, v)) * P1 + P4; }
uint64_t zso_xxh64(const uint8_t *p, uint64_t n, uint64_t seed) {
    const uint8_t *end = p + n;
    uint64_t h;
    if (n >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = xround(v1, rd64(p));
            v2 = xround(v2, rd64(p + 8));
            v3 = xround(v3, rd64(p + 16));
            v4 = xround(v4, rd64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else {
        h = seed + P5;
    }
    h += n;
    while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p++) * P5; h = rotl(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* ---- bit readers ---- */
/* bits [off, off+n) of the little-endian bit string p[0..len); positions < 0 read as zero (RFC 8878 4.1: a backward
 * stream that is over-read delivers zeros) */
static uint64_t bits_at(const uint8_t *p, int64_t len, int64_t off, int n) {
    if (n == 0) return 0;
    if (off < 0) {
        if (off + n <= 0) return 0;
        return bits_at(p, len, 0, (int)(off + n)) << (-off);
    }
    uint64_t v = 0;
    int64_t b0 = off >> 3;
    for (int i = 0; i < 8; i++)
        if (b0 + i < len) v |= (uint64_t)p[b0 + i] << (8 * i);
    v >>= (off & 7);
    /* n <= 32 and (off & 7) <= 7: 64 bits are enough */
    return v & ((n >= 64) ? ~0ULL : ((1ULL << n) - 1));
}
typedef struct { const uint8_t *p; int64_t len; int64_t pos; } brev; /* pos = unread bits below the cursor */
static int brev_init(brev *b, const uint8_t *p, int64_t len) {
    if (len <= 0 || p[len - 1] == 0) return -1;
    b->p = p; b->len = len; b->pos = 8 * (len - 1) + hb(p[len - 1]);
    return 0;
}
static uint64_t brev_read(brev *b, int n) { b->pos -= n; return bits_at(b->p, b->len, b->pos, n); }
static uint64_t brev_peek(const brev *b, int n) { return bits_at(b->p, b->len, b->pos - n, n); }

/* ---- FSE ---- */
typedef struct { uint8_t sym[512]; uint8_t nbits[512]; uint16_t base[512]; int log; } fse_t;

/* RFC 8878 4.1.1: the normalised counts, then the decoding table.  Returns bytes consumed, or -1. */
static int fse_read_norm(const uint8_t *p, int64_t len, int max_log, int max_sym, int16_t *norm, int *n_sym, int *log_out) {
    if (len < 1) return -1;
    int64_t bit = 0;
    int log = 5 + (int)bits_at(p, len, bit, 4);
    bit += 4;
    if (log > max_log) return -1;
    int remaining = 1 << log, s = 0;
    while (remaining > 0 && s <= max_sym) {
        int nb = hb((uint32_t)remaining + 1) + 1;
        if (((bit + nb + 7) >> 3) > len + 4) return -1; /* far past the end */
        uint32_t val = (uint32_t)bits_at(p, len, bit, nb);
        uint32_t lower = (1u << (nb - 1)) - 1, thresh = (1u << nb) - 1 - ((uint32_t)remaining + 1);
        if ((val & lower) < thresh) { bit += nb - 1; val &= lower; }
        else { bit += nb; if (val > lower) val -= thresh; }
        int proba = (int)val - 1;
        remaining -= proba < 0 ? -proba : proba;
        norm[s++] = (int16_t)proba;
        if (proba == 0) {
            for (;;) {
                int rep = (int)bits_at(p, len, bit, 2);
                bit += 2;
                for (int i = 0; i < rep && s <= max_sym; i++) norm[s++] = 0;
                if (rep != 3) break;
            }
        }
    }
    if (remaining != 0 || s > max_sym + 1) return -1;
    int64_t bytes = (bit + 7) >> 3;
    if (bytes > len) return -1;
    *n_sym = s; *log_out = log;
    return (int)bytes;
}
static int fse_build(fse_t *t, const int16_t *norm, int n_sym, int log) {
    int size = 1 << log, high = size;
    uint16_t next[256];
    t->log = log;
    for (int s = 0; s < n_sym; s++)
        if (norm[s] == -1) { t->sym[--high] = (uint8_t)s; next[s] = 1; }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < n_sym; s++) {
        if (norm[s] <= 0) continue;
        next[s] = (uint16_t)norm[s];
        for (int i = 0; i < norm[s]; i++) {
            t->sym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return -1;
    for (int i = 0; i < size; i++) {
        uint16_t x = next[t->sym[i]]++;
        t->nbits[i] = (uint8_t)(log - hb(x));
        t->base[i] = (uint16_t)((x << t->nbits[i]) - size);
    }
    return 0;
}
static void fse_rle(fse_t *t, uint8_t sym) { t->log = 0; t->sym[0] = sym; t->nbits[0] = 0; t->base[0] = 0; }

static const int16_t LL_DEF[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEF[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* ---- Huffman (RFC 8878 4.2) ---- */
typedef struct { uint8_t sym[2048]; uint8_t nbits[2048]; int log; int valid; } huf_t;

static int huf_from_weights(huf_t *h, uint8_t *w, int n) { /* n weights given, the last one is implied */
    uint32_t total = 0;
    for (int i = 0; i < n; i++) {
        if (w[i] > 11) return -1;
        total += w[i] ? 1u << (w[i] - 1) : 0;
    }
    if (total == 0) return -1;
    int log = hb(total) + 1;
    if (log > 11) return -1;
    uint32_t rest = (1u << log) - total;
    if (rest & (rest - 1)) return -1; /* not a power of two */
    w[n] = (uint8_t)(hb(rest) + 1);
    n++;
    /* codes are handed out from the lowest weight (longest code) up, within a weight in symbol order */
    uint32_t rank_start[13] = {0}, cnt[13] = {0};
    for (int i = 0; i < n; i++) cnt[w[i]]++;
    if (cnt[1] < 2 || (cnt[1] & 1)) return -1; /* libzstd: at least two, and an even number, of the longest codes */
    uint32_t pos = 0;
    for (int k = 1; k <= log; k++) { rank_start[k] = pos; pos += cnt[k] << (k - 1); }
    for (int i = 0; i < n; i++) {
        if (!w[i]) continue;
        uint32_t len = 1u << (w[i] - 1), at = rank_start[w[i]];
        for (uint32_t j = 0; j < len; j++) { h->sym[at + j] = (uint8_t)i; h->nbits[at + j] = (uint8_t)(log + 1 - w[i]); }
        rank_start[w[i]] += len;
    }
    h->log = log; h->valid = 1;
    return 0;
}
/* returns bytes consumed by the tree description, or -1 */
static int huf_read_tree(huf_t *h, const uint8_t *p, int64_t len) {
    if (len < 1) return -1;
    uint8_t w[256];
    int hdr = p[0], n;
    if (hdr >= 128) {
        n = hdr - 127;
        int bytes = (n + 1) / 2;
        if (1 + bytes > len) return -1;
        for (int i = 0; i < n; i++) w[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
        if (huf_from_weights(h, w, n)) return -1;
        return 1 + bytes;
    }
    if (1 + hdr > len || hdr == 0) return -1;
    int16_t norm[16];
    int ns, log;
    int used = fse_read_norm(p + 1, hdr, 6, 12, norm, &ns, &log);
    if (used < 0) return -1;
    fse_t t;
    if (fse_build(&t, norm, ns, log)) return -1;
    brev b;
    if (brev_init(&b, p + 1 + used, hdr - used)) return -1;
    uint32_t s1 = (uint32_t)brev_read(&b, log), s2 = (uint32_t)brev_read(&b, log);
    n = 0;
    for (;;) {
        if (n >= 254) return -1;
        w[n++] = t.sym[s1];
        s1 = t.base[s1] + (uint32_t)brev_read(&b, t.nbits[s1]);
        if (b.pos < 0) { if (n >= 255) return -1; w[n++] = t.sym[s2]; break; }
        if (n >= 254) return -1;
        w[n++] = t.sym[s2];
        s2 = t.base[s2] + (uint32_t)brev_read(&b, t.nbits[s2]);
        if (b.pos < 0) { if (n >= 255) return -1; w[n++] = t.sym[s1]; break; }
    }
    if (huf_from_weights(h, w, n)) return -1;
    return 1 + hdr;
}
static int huf_stream(const huf_t *h, const uint8_t *p, int64_t len, uint8_t *out, int64_t n_out) {
    brev b;
    if (brev_init(&b, p, len)) return -1;
    for (int64_t i = 0; i < n_out; i++) {
        uint32_t idx = (uint32_t)brev_peek(&b, h->log);
        out[i] = h->sym[idx];
        b.pos -= h->nbits[idx];
        if (b.pos < -(int64_t)h->log) return -1;
    }
    return b.pos == 0 ? 0 : -1; /* libzstd: the stream must end exactly */
}

/* ---- frame state ---- */
typedef struct {
    huf_t huf;
    fse_t ll, of, ml;
    int have_ll, have_of, have_ml;
    uint64_t rep[3];
} zctx;

/* intermediate products of one compressed block, for stage-by-stage comparison with the device */
typedef struct {
    uint32_t n_lit, n_seq;
    uint8_t *lit;       /* n_lit */
    uint32_t *seq;      /* n_seq x {ll, ml, offset (resolved)} */
} zso_block_dump;

static int seq_table(fse_t *t, int *have, int mode, const uint8_t **pp, const uint8_t *end, const int16_t *def, int def_n, int def_log,
                     int max_log, int max_sym) {
    const uint8_t *p = *pp;
    if (mode == 0) { if (fse_build(t, def, def_n, def_log)) return -1; *have = 1; return 0; }
    if (mode == 1) { if (p >= end) return -1; if (*p > max_sym) return -1; fse_rle(t, *p); *pp = p + 1; *have = 1; return 0; }
    if (mode == 2) {
        int16_t norm[64];
        int ns, log;
        int used = fse_read_norm(p, end - p, max_log, max_sym, norm, &ns, &log);
        if (used < 0) return -1;
        if (fse_build(t, norm, ns, log)) return -1;
        *pp = p + used; *have = 1;
        return 0;
    }
    return *have ? 0 : -1; /* repeat */
}

static int compressed_block(zctx *c, const uint8_t *p, uint32_t bsize, uint8_t *dst, uint64_t dst_pos, uint64_t cap, uint64_t frame_start,
                            uint64_t *produced, zso_block_dump *dump) {
    static uint8_t lit[BLOCK_MAX + 32];
    const uint8_t *end = p + bsize;
    if (bsize < 1) return ZSO_E_LITERALS; /* libzstd: a compressed block holds at least ... */
    /* literals section */
    int ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
    uint32_t regen, csize = 0, hsz;
    int streams = 1;
    if (ltype < 2) {
        if (!(sf & 1)) { hsz = 1; regen = p[0] >> 3; }
        else if (sf == 1) { hsz = 2; if (bsize < 2) return ZSO_E_LITERALS; regen = (p[0] >> 4) | ((uint32_t)p[1] << 4); }
        else { hsz = 3; if (bsize < 3) return ZSO_E_LITERALS; regen = (p[0] >> 4) | ((uint32_t)p[1] << 4) | ((uint32_t)p[2] << 12); }
    } else {
        if (bsize < 3) return ZSO_E_LITERALS;
        uint64_t v = 0;
        for (int i = 0; i < 5 && i < (int)bsize; i++) v |= (uint64_t)p[i] << (8 * i);
        if (sf == 0) { hsz = 3; regen = (v >> 4) & 1023; csize = (v >> 14) & 1023; }
        else if (sf == 1) { hsz = 3; streams = 4; regen = (v >> 4) & 1023; csize = (v >> 14) & 1023; }
        else if (sf == 2) { hsz = 4; streams = 4; regen = (v >> 4) & 16383; csize = (uint32_t)(v >> 18) & 16383; }
        else { hsz = 5; streams = 4; regen = (v >> 4) & 262143; csize = (uint32_t)(v >> 22) & 262143; }
        if (hsz > bsize) return ZSO_E_LITERALS;
    }
    if (regen > BLOCK_MAX) return ZSO_E_LITERALS;
    const uint8_t *q = p + hsz;
    if (ltype == 0) { if (q + regen > end) return ZSO_E_LITERALS; memcpy(lit, q, regen); q += regen; }
    else if (ltype == 1) { if (q + 1 > end) return ZSO_E_LITERALS; memset(lit, *q, regen); q += 1; }
    else {
        if (q + csize > end) return ZSO_E_LITERALS;
        const uint8_t *s = q;
        int64_t left = csize;
        if (ltype == 2) {
            int used = huf_read_tree(&c->huf, s, left);
            if (used < 0) return ZSO_E_HUF;
            s += used; left -= used;
        } else if (!c->huf.valid) return ZSO_E_HUF;
        if (streams == 1) {
            if (huf_stream(&c->huf, s, left, lit, regen)) return ZSO_E_HUF;
        } else {
            if (left < 10) return ZSO_E_HUF; /* libzstd: jump table + at least one byte per stream */
            uint32_t s1 = s[0] | (s[1] << 8), s2 = s[2] | (s[3] << 8), s3 = s[4] | (s[5] << 8);
            int64_t s4 = left - 6 - (int64_t)s1 - s2 - s3;
            if (s4 < 1 || s1 < 1 || s2 < 1 || s3 < 1) return ZSO_E_HUF;
            uint32_t per = (regen + 3) / 4;
            if (3 * per > regen) return ZSO_E_HUF;
            const uint8_t *a = s + 6;
            if (huf_stream(&c->huf, a, s1, lit, per)) return ZSO_E_HUF;
            if (huf_stream(&c->huf, a + s1, s2, lit + per, per)) return ZSO_E_HUF;
            if (huf_stream(&c->huf, a + s1 + s2, s3, lit + 2 * per, per)) return ZSO_E_HUF;
            if (huf_stream(&c->huf, a + s1 + s2 + s3, s4, lit + 3 * per, regen - 3 * per)) return ZSO_E_HUF;
        }
        q += csize;
    }
    /* sequences section */
    if (q >= end) return ZSO_E_SEQ;
    uint32_t nseq = *q++;
    if (nseq >= 128) {
        if (nseq == 255) { if (q + 2 > end) return ZSO_E_SEQ; nseq = q[0] + (q[1] << 8) + 0x7F00; q += 2; }
        else { if (q + 1 > end) return ZSO_E_SEQ; nseq = ((nseq - 128) << 8) + q[0]; q += 1; }
    }
    if (dump) {
        dump->n_lit = regen; dump->n_seq = nseq;
        dump->lit = (uint8_t *)malloc(regen + 1); memcpy(dump->lit, lit, regen);
        dump->seq = (uint32_t *)malloc((size_t)nseq * 12 + 4);
    }
    uint64_t out = dst_pos;
    if (nseq == 0) {
        if (q != end) return ZSO_E_SEQ;
        if (out + regen > cap) return ZSO_E_CAPACITY;
        memcpy(dst + out, lit, regen);
        *produced = regen;
        return 0;
    }
    if (q >= end) return ZSO_E_SEQ;
    int modes = *q++;
    if (modes & 3) return ZSO_E_RESERVED;
    if (seq_table(&c->ll, &c->have_ll, modes >> 6, &q, end, LL_DEF, 36, 6, 9, 35)) return ZSO_E_FSE;
    if (seq_table(&c->of, &c->have_of, (modes >> 4) & 3, &q, end, OF_DEF, 29, 5, 8, 31)) return ZSO_E_FSE;
    if (seq_table(&c->ml, &c->have_ml, (modes >> 2) & 3, &q, end, ML_DEF, 53, 6, 9, 52)) return ZSO_E_FSE;
    brev b;
    if (brev_init(&b, q, end - q)) return ZSO_E_SEQ;
    uint32_t sl = (uint32_t)brev_read(&b, c->ll.log), so = (uint32_t)brev_read(&b, c->of.log), sm = (uint32_t)brev_read(&b, c->ml.log);
    if (b.pos < 0) return ZSO_E_SEQ;
    uint32_t lit_pos = 0;
    for (uint32_t i = 0; i < nseq; i++) {
        uint32_t oc = c->of.sym[so], mc = c->ml.sym[sm], lc = c->ll.sym[sl];
        if (oc > 31 || mc > 52 || lc > 35) return ZSO_E_SEQ;
        uint64_t ov = ((uint64_t)1 << oc) + brev_read(&b, (int)oc);
        uint32_t ml = ML_BASE[mc] + (uint32_t)brev_read(&b, ML_BITS[mc]);
        uint32_t ll = LL_BASE[lc] + (uint32_t)brev_read(&b, LL_BITS[lc]);
        if (i + 1 < nseq) {
            sl = c->ll.base[sl] + (uint32_t)brev_read(&b, c->ll.nbits[sl]);
            sm = c->ml.base[sm] + (uint32_t)brev_read(&b, c->ml.nbits[sm]);
            so = c->of.base[so] + (uint32_t)brev_read(&b, c->of.nbits[so]);
        }
        if (b.pos < 0) return ZSO_E_SEQ;
        uint64_t off;
        if (ov > 3) { off = ov - 3; c->rep[2] = c->rep[1]; c->rep[1] = c->rep[0]; c->rep[0] = off; }
        else {
            uint32_t idx = (uint32_t)ov - 1 + (ll == 0);
            if (idx == 0) off = c->rep[0];
            else {
                off = idx < 3 ? c->rep[idx] : c->rep[0] - 1;
                if (off == 0) return ZSO_E_OFFSET;
                if (idx > 1) c->rep[2] = c->rep[1];
                c->rep[1] = c->rep[0];
                c->rep[0] = off;
            }
        }
        if (dump) { dump->seq[3 * i] = ll; dump->seq[3 * i + 1] = ml; dump->seq[3 * i + 2] = (uint32_t)off; }
        if (lit_pos + ll > regen) return ZSO_E_SEQ;
        if (out + ll + ml > cap) return ZSO_E_CAPACITY;
        memcpy(dst + out, lit + lit_pos, ll);
        out += ll; lit_pos += ll;
        if (off > out - frame_start) return ZSO_E_OFFSET; /* libzstd: not beyond the start of the frame's content */
        for (uint32_t k = 0; k < ml; k++) dst[out + k] = dst[out + k - off];
        out += ml;
    }
    if (b.pos != 0) return ZSO_E_SEQ;
    uint32_t rest = regen - lit_pos;
    if (out + rest > cap) return ZSO_E_CAPACITY;
    memcpy(dst + out, lit + lit_pos, rest);
    out += rest;
    if (out - dst_pos > BLOCK_MAX) return ZSO_E_BLOCK;
    *produced = out - dst_pos;
    return 0;
}

/* Decode every frame of src[0, n) into dst (capacity cap).  dumps: NULL, or an array that receives one entry per
 * COMPRESSED block in file order (caller frees lit / seq), at most dump_cap entries; *n_dumps = blocks seen. */
int zso_decompress(const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t cap, uint64_t *produced, zso_block_dump *dumps, uint64_t dump_cap,
                   uint64_t *n_dumps) {
    uint64_t pos = 0, out = 0, nd = 0;
    while (pos < n) {
        if (n - pos < 4) return ZSO_E_TRUNCATED;
        uint32_t magic = rd32(src + pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (n - pos < 8) return ZSO_E_TRUNCATED;
            uint64_t sz = rd32(src + pos + 4);
            if (n - pos - 8 < sz) return ZSO_E_TRUNCATED;
            pos += 8 + sz;
            continue;
        }
        if (magic != 0xFD2FB528u) return ZSO_E_MAGIC;
        pos += 4;
        if (pos >= n) return ZSO_E_TRUNCATED;
        uint8_t fhd = src[pos++];
        int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did_flag = fhd & 3;
        if (fhd & 8) return ZSO_E_RESERVED;
        uint64_t window = 0, fcs = ~0ULL;
        if (!single) {
            if (pos >= n) return ZSO_E_TRUNCATED;
            uint8_t wd = src[pos++];
            int wlog = 10 + (wd >> 3);
            window = (1ULL << wlog) + ((1ULL << wlog) >> 3) * (wd & 7);
            if (wlog > 31) return ZSO_E_WINDOW;
        }
        static const int did_bytes[4] = {0, 1, 2, 4};
        if (n - pos < (uint64_t)did_bytes[did_flag]) return ZSO_E_TRUNCATED;
        uint32_t did = 0;
        for (int i = 0; i < did_bytes[did_flag]; i++) did |= (uint32_t)src[pos + i] << (8 * i);
        pos += did_bytes[did_flag];
        if (did) return ZSO_E_DICT;
        int fcs_bytes = fcs_flag == 0 ? single : fcs_flag == 1 ? 2 : fcs_flag == 2 ? 4 : 8;
        if (n - pos < (uint64_t)fcs_bytes) return ZSO_E_TRUNCATED;
        if (fcs_bytes) {
            fcs = 0;
            for (int i = 0; i < fcs_bytes; i++) fcs |= (uint64_t)src[pos + i] << (8 * i);
            if (fcs_bytes == 2) fcs += 256;
            pos += fcs_bytes;
        }
        if (single) window = fcs;
        if (window > (1ULL << 27) && !single) return ZSO_E_WINDOW; /* libzstd's default ZSTD_d_windowLogMax = 27 */
        zctx *c = (zctx *)calloc(1, sizeof(zctx));
        c->rep[0] = 1; c->rep[1] = 4; c->rep[2] = 8;
        uint64_t frame_start = out;
        int rc = 0;
        for (;;) {
            if (n - pos < 3) { rc = ZSO_E_TRUNCATED; break; }
            uint32_t bh = src[pos] | (src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
            pos += 3;
            int last = bh & 1, type = (bh >> 1) & 3;
            uint32_t bsize = bh >> 3;
            if (type == 3) { rc = ZSO_E_BLOCK; break; }
            if (bsize > BLOCK_MAX) { rc = ZSO_E_BLOCK; break; } /* Block_Maximum_Size; the format's hard limit */
            if (type == 0) {
                if (n - pos < bsize) { rc = ZSO_E_TRUNCATED; break; }
                if (out + bsize > cap) { rc = ZSO_E_CAPACITY; break; }
                memcpy(dst + out, src + pos, bsize);
                pos += bsize; out += bsize;
            } else if (type == 1) {
                if (n - pos < 1) { rc = ZSO_E_TRUNCATED; break; }
                if (out + bsize > cap) { rc = ZSO_E_CAPACITY; break; }
                memset(dst + out, src[pos], bsize);
                pos += 1; out += bsize;
            } else {
                if (n - pos < bsize) { rc = ZSO_E_TRUNCATED; break; }
                uint64_t got = 0;
                zso_block_dump *d = (dumps && nd < dump_cap) ? &dumps[nd] : NULL;
                rc = compressed_block(c, src + pos, bsize, dst, out, cap, frame_start, &got, d);
                nd++;
                if (rc) break;
                pos += bsize; out += got;
            }
            if (last) break;
        }
        free(c);
        if (rc) return rc;
        if (fcs != ~0ULL && out - frame_start != fcs) return ZSO_E_SIZE;
        if (checksum) {
            if (n - pos < 4) return ZSO_E_TRUNCATED;
            if ((uint32_t)zso_xxh64(dst + frame_start, out - frame_start, 0) != rd32(src + pos)) return ZSO_E_CHECKSUM;
            pos += 4;
        }
    }
    *produced = out;
    if (n_dumps) *n_dumps = nd;
    return ZSO_OK;
}
