// exg_gzip.cpp — gzip (RFC 1952) member framing on the host; the DEFLATE streams themselves are
// inflated on the device (exg_inflate.hip).  Replaces the stream framing done inside flate2 /
// noodles-bgzf for the reference (rust/src/arrow_reader.rs:60-91).
//
// BGZF members (and any member whose FEXTRA carries the 'BC' subfield) give their compressed size,
// so the whole file is indexed without decoding and every member is inflated in parallel.  A member
// without it runs to an unknown end: it is returned as the LAST entry with `open_ended` set; the
// caller inflates it, reads how many bytes it consumed, and indexes again from there.
#include <string.h>

#include "exg_common.hpp"

extern "C" int exg_gzip_index(const uint8_t *data, uint64_t n, uint64_t start, exg_inflate_member *members,
                              uint64_t cap, uint64_t *n_members, uint64_t *total_out, int *open_ended) {
    if (!data || !members || !n_members || !total_out || !open_ended) {
        exg::set_error("exg_gzip_index: null argument");
        return EXG_E_INVALID_ARG;
    }
    uint64_t pos = start, k = 0, out = *total_out;
    *open_ended = 0;
    while (pos < n) {
        if (n - pos < 18 || data[pos] != 0x1f || data[pos + 1] != 0x8b || data[pos + 2] != 8) {
            exg::set_error("invalid gzip header at byte %llu", (unsigned long long)pos);
            return EXG_E_PARSE;
        }
        const uint8_t flg = data[pos + 3];
        uint64_t p = pos + 10;
        int64_t bsize = -1;
        if (flg & 4) {  // FEXTRA
            if (p + 2 > n) goto truncated;
            uint64_t xlen = data[p] | ((uint64_t)data[p + 1] << 8);
            p += 2;
            if (p + xlen > n) goto truncated;
            uint64_t q = p;
            while (q + 4 <= p + xlen) {
                uint64_t slen = data[q + 2] | ((uint64_t)data[q + 3] << 8);
                if (data[q] == 'B' && data[q + 1] == 'C' && slen == 2 && q + 6 <= p + xlen)
                    bsize = (int64_t)(data[q + 4] | ((uint64_t)data[q + 5] << 8));
                q += 4 + slen;
            }
            p += xlen;
        }
        if (flg & 8) {  // FNAME
            while (p < n && data[p]) p++;
            p++;
        }
        if (flg & 16) {  // FCOMMENT
            while (p < n && data[p]) p++;
            p++;
        }
        if (flg & 2) p += 2;  // FHCRC
        if (p > n) goto truncated;
        if (k >= cap) {
            exg::set_error("exg_gzip_index: more than %llu members", (unsigned long long)cap);
            return EXG_E_CAPACITY;
        }
        exg_inflate_member &m = members[k];
        m.comp_off = p;
        m.out_off = out;
        if (bsize >= 0) {
            uint64_t end = pos + (uint64_t)bsize + 1;  // BSIZE = total block size - 1
            if (end > n || end < p + 8) goto truncated;
            m.comp_size = end - p;
            uint64_t isize = data[end - 4] | ((uint64_t)data[end - 3] << 8) | ((uint64_t)data[end - 2] << 16) |
                             ((uint64_t)data[end - 1] << 24);
            m.out_cap = isize;
            out += isize;
            pos = end;
            k++;
        } else {
            // unknown compressed size: runs (at most) to the end of the file; ISIZE of the file's last
            // member is in the last 4 bytes (mod 2^32) — a bound only if nothing follows
            m.comp_size = n - p;
            uint64_t isize = data[n - 4] | ((uint64_t)data[n - 3] << 8) | ((uint64_t)data[n - 2] << 16) |
                             ((uint64_t)data[n - 1] << 24);
            // DEFLATE expands at most 1032:1.  A small member (what stays on the one-wavefront path: below 128 KiB of
            // input) gets that bound — `cat a.vcf.gz b.vcf.gz` of highly compressible members must not be reported as
            // corrupt for outgrowing a guessed ratio; a big one is decoded in chunks (exg_inflate_stream sizes its own
            // output), and 8:1 + ISIZE only bounds the fallback with that path switched off
            uint64_t bound = (n - p) < (128u << 10) ? (n - p) * 1032 + 65536 : (n - p) * 8 + 65536;
            m.out_cap = isize > bound ? isize : bound;
            out += m.out_cap;
            k++;
            *open_ended = 1;
            break;
        }
    }
    *n_members = k;
    *total_out = out;
    return EXG_OK;
truncated:
    exg::set_error("truncated gzip member at byte %llu", (unsigned long long)pos);
    return EXG_E_PARSE;
}
