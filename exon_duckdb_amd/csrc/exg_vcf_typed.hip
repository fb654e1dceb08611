// exg_vcf_typed.hip — the reference's nested VCF columns, built on the device from the tokeniser's
// string_t columns (exg_vcf.hip):  id / alt / filter -> LIST(VARCHAR), info -> STRUCT of the header's
// ##INFO keys, formats -> LIST(STRUCT of the header's ##FORMAT keys), one entry per sample.
//
// Semantics restated (not the code; parity unpinned beyond test_vcf_record_scan.test:10-19, which pins
// alt = [<*>], info.indel = NULL, info.dp = 1 for row 1 of vcf/index.vcf): exon 0.2.6
// datasources::vcf::{VCFSchemaBuilder, VCFArrayBuilder} over noodles-vcf 0.34.0 record parsing
// (reached from rust/src/arrow_reader.rs:116-153):
//   * ID split on ';', ALT on ',', FILTER on ';' ("PASS" is one element); "." => empty list;
//   * INFO "." => every child NULL; else ';'-separated key[=value]; a header key that is absent => NULL,
//     a value "." => NULL; Flag => true when present; Integer => i32, Float => f32, String/Character
//     => the text; Number other than 1 (and not a Flag) => list split on ',', element "." => NULL;
//     keys that the header does not declare are not columns; a repeated key keeps its first value;
//   * FORMAT keys by position, each sample ':'-split the same way; trailing fields left out => NULL.
// Numbers that do not parse are a record error (EXG_PE_VCF_INFO / EXG_PE_VCF_FORMAT).
#include "exg_arrow.hpp"
#include "exg_parse.hpp"
#include "exg_float_slow.hpp"
#include "exg_scan.hpp"

namespace exg {
namespace arrow {

namespace {

struct ColGet2 {
    StrCol c;
    const uint32_t *row_map;
    __device__ __forceinline__ const uint8_t *ptr(uint64_t j, uint32_t *len_out) const {
        const uint64_t r = row_map ? (uint64_t)row_map[j] : j;
        const uint4 v = reinterpret_cast<const uint4 *>(c.d_col)[r];
        *len_out = v.x;
        if (v.x <= EXG_INLINE_LENGTH) return reinterpret_cast<const uint8_t *>(c.d_col) + r * 16 + 4;
        const uint64_t p = (uint64_t)v.z | ((uint64_t)v.w << 32);
        return c.d_base + (p - c.payload_base);
    }
    __device__ __forceinline__ uint64_t row(uint64_t j) const { return row_map ? (uint64_t)row_map[j] : j; }
};

struct PtrSrc {
    const uint8_t *p;
    int n;  // bytes of the literal: nothing behind them is read
    __device__ __forceinline__ uint32_t b(int i) const { return p[i]; }
    __device__ __forceinline__ void u96(int i, uint32_t *w0, uint32_t *w1, uint32_t *w2) const {
        uint32_t w[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 12; k++)
            if (i + k < n) w[k >> 2] |= (uint32_t)p[i + k] << (8 * (k & 3));
        *w0 = w[0], *w1 = w[1], *w2 = w[2];
    }
};

inline uint32_t grid_for(uint64_t n) { return (uint32_t)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

__device__ __forceinline__ bool is_missing(const uint8_t *p, uint32_t len) { return len == 1 && p[0] == '.'; }

// ---- id / alt / filter ---------------------------------------------------------------------------------------
struct ListCountF {
    ColGet2 g;
    uint8_t sep;
    __device__ uint64_t operator()(uint64_t j) const {
        uint32_t len;
        const uint8_t *p = g.ptr(j, &len);
        if (len == 0 || is_missing(p, len)) return 0;
        uint32_t c = 1;
        for (uint32_t i = 0; i < len; i++) c += p[i] == sep;
        return c;
    }
};

__global__ __launch_bounds__(256) void k_list_views(ColGet2 g, uint64_t n, uint8_t sep, const uint64_t *__restrict__ goff,
                                                    View *views) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) {
        uint64_t o = goff[j];
        if (goff[j + 1] == o) continue;
        uint32_t len;
        const uint8_t *p = g.ptr(j, &len);
        uint32_t s = 0;
        for (uint32_t i = 0; i <= len; i++) {
            if (i == len || p[i] == sep) {
                views[o++] = View{p + s, i - s, 1u};
                s = i + 1;
            }
        }
    }
}

// ---- INFO ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_key(const VtKeys &keys, const uint8_t *k, uint32_t klen) {
    for (uint32_t q = 0; q < keys.n; q++) {
        if (keys.k[q].name_len != klen) continue;
        const uint8_t *nm = keys.d_names + keys.k[q].name_off;
        bool eq = true;
        for (uint32_t i = 0; i < klen && eq; i++) eq = nm[i] == k[i];
        if (eq) return (int)q;
    }
    return -1;
}

__global__ __launch_bounds__(256) void k_info_cells(ColGet2 g, uint64_t n, VtKeys keys, VtCell *cells) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) {
        VtCell *row = cells + j * keys.n;
        for (uint32_t q = 0; q < keys.n; q++) row[q] = VtCell{0u, kVtAbsent};
        uint32_t len;
        const uint8_t *p = g.ptr(j, &len);
        if (len == 0 || is_missing(p, len)) continue;
        uint32_t s = 0;
        while (s <= len) {
            uint32_t e = s, eq = 0xFFFFFFFFu;
            while (e < len && p[e] != ';') {
                if (p[e] == '=' && eq == 0xFFFFFFFFu) eq = e;
                e++;
            }
            const uint32_t kend = eq != 0xFFFFFFFFu ? eq : e;
            if (kend > s) {
                const int q = find_key(keys, p + s, kend - s);
                if (q >= 0 && row[q].len == kVtAbsent)
                    row[q] = eq != 0xFFFFFFFFu ? VtCell{eq + 1, e - eq - 1} : VtCell{e, kVtBare};
            }
            s = e + 1;
        }
    }
}

// ---- FORMAT / samples -----------------------------------------------------------------------------------------------
struct SampleCountF {
    ColGet2 g;
    const uint64_t *valid;
    __device__ uint64_t operator()(uint64_t j) const {
        const uint64_t r = g.row(j);
        if (valid && !((valid[r >> 6] >> (r & 63)) & 1)) return 0;
        uint32_t len;
        const uint8_t *p = g.ptr(j, &len);
        uint32_t c = 0;
        for (uint32_t i = 0; i < len; i++) c += p[i] == '\t';
        return c;
    }
};

static constexpr int kMaxFormatPos = 64;

__global__ __launch_bounds__(256) void k_sample_cells(ColGet2 g, uint64_t n, const uint64_t *__restrict__ goff, VtKeys keys,
                                                      VtCell *cells, View *fields, uint32_t *elem_row) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) {
        uint64_t s_idx = goff[j];
        if (goff[j + 1] == s_idx) continue;
        uint32_t len;
        const uint8_t *p = g.ptr(j, &len);
        // FORMAT keys by position
        int16_t key_at[kMaxFormatPos];
        int n_pos = 0;
        uint32_t i = 0, s = 0;
        for (;; i++) {
            if (i == len || p[i] == '\t' || p[i] == ':') {
                if (n_pos < kMaxFormatPos) key_at[n_pos++] = (int16_t)find_key(keys, p + s, i - s);
                s = i + 1;
                if (i == len || p[i] == '\t') break;
            }
        }
        // samples
        while (i < len) {
            const uint32_t t0 = i + 1;  // first byte of the sample
            uint32_t e = t0;
            while (e < len && p[e] != '\t') e++;
            VtCell *row = cells + s_idx * keys.n;
            for (uint32_t q = 0; q < keys.n; q++) row[q] = VtCell{0u, kVtAbsent};
            fields[s_idx] = View{p + t0, e - t0, 1u};
            elem_row[s_idx] = (uint32_t)j;
            int pos = 0;
            uint32_t vs = t0;
            for (uint32_t q = t0; q <= e; q++) {
                if (q == e || p[q] == ':') {
                    if (pos < n_pos && key_at[pos] >= 0 && row[key_at[pos]].len == kVtAbsent)
                        row[key_at[pos]] = VtCell{vs - t0, q - vs};
                    pos++;
                    vs = q + 1;
                }
            }
            s_idx++;
            i = e;
        }
    }
}

// ---- typed children out of the cells --------------------------------------------------------------------------------
struct CellGet {
    CellSrc s;
    // value of element j: pointer + length; returns 0 absent, 1 bare (no '='), 2 value
    __device__ __forceinline__ int get(uint64_t j, const uint8_t **p, uint32_t *len) const {
        const VtCell c = s.d_cells[j * s.n_keys + s.key];
        if (c.len == kVtAbsent) return 0;
        if (c.len == kVtBare) return 1;
        const uint8_t *f;
        if (s.d_fields) {
            f = s.d_fields[j].p;
        } else {
            uint32_t fl;
            f = ColGet2{s.col, s.d_row_map}.ptr(j, &fl);
        }
        *p = f + c.off;
        *len = c.len;
        return 2;
    }
    __device__ __forceinline__ unsigned long long err_row(uint64_t j) const { return s.d_elem_row ? s.d_elem_row[j] : j; }
};

// status 2 of parse_f32: hand the literal to the exact parser (k_f32_slow); false when the list is full
__device__ __forceinline__ bool defer_f32(unsigned long long *err, const uint8_t *p, uint32_t len, float *dst, unsigned long long row) {
    ErrBlock *eb = reinterpret_cast<ErrBlock *>(err);
    const unsigned int k = atomicAdd(&eb->n_slow, 1u);
    if (k >= kSlowF32) return false;
    eb->slow[k] = SlowF32{p, len, 0u, dst, row};
    return true;
}
__global__ __launch_bounds__(64) void k_f32_slow(ErrBlock *eb, uint32_t err_code) {
    const unsigned int n = eb->n_slow < kSlowF32 ? eb->n_slow : kSlowF32;
    for (unsigned int i = threadIdx.x; i < n; i += 64) {
        const SlowF32 e = eb->slow[i];
        uint32_t bits = 0;
        if (f32_parse_exact(e.p, (int)e.len, &bits))
            atomicMin(&eb->err, (e.row << 8) | err_code);
        else
            *e.dst = __uint_as_float(bits);
    }
    __syncthreads();
    if (threadIdx.x == 0) eb->n_slow = 0;
}

template <int kType>  // kVtInt / kVtFloat
__global__ __launch_bounds__(256) void k_cells_scalar(CellGet g, uint64_t n, void *values, uint64_t *valid,
                                                      unsigned long long *err, uint32_t err_code) {
    const uint64_t n_pad = (n + 63) & ~63ull;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_pad; j += (uint64_t)gridDim.x * 256) {
        bool ok = false;
        if (j < n) {
            const uint8_t *p;
            uint32_t len;
            int iv = 0;
            float fv = 0.f;
            if (g.get(j, &p, &len) == 2 && !is_missing(p, len)) {
                bool parsed;
                if (kType == kVtInt)
                    parsed = parse_i32(PtrSrc{p, (int)len}, 0, (int)len, &iv);
                else {
                    const int st = parse_f32(PtrSrc{p, (int)len}, 0, (int)len, &fv);
                    parsed = st == 0 || (st == 2 && defer_f32(err, p, len, (float *)values + j, g.err_row(j)));
                }
                if (parsed)
                    ok = true;
                else
                    atomicMin(err, (g.err_row(j) << 8) | err_code);
            }
            if (kType == kVtInt)
                ((int32_t *)values)[j] = iv;
            else
                ((float *)values)[j] = fv;
        }
        const unsigned long long m = __ballot(ok);
        if ((threadIdx.x & 63) == 0) valid[j >> 6] = m;
    }
}

__global__ __launch_bounds__(256) void k_cells_flag(CellGet g, uint64_t n, uint64_t *bits, uint64_t *valid) {
    const uint64_t n_pad = (n + 63) & ~63ull;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_pad; j += (uint64_t)gridDim.x * 256) {
        bool present = false;
        if (j < n) {
            const uint8_t *p;
            uint32_t len;
            present = g.get(j, &p, &len) != 0;
        }
        const unsigned long long m = __ballot(present);
        if ((threadIdx.x & 63) == 0) bits[j >> 6] = m, valid[j >> 6] = m;
    }
}

__global__ __launch_bounds__(256) void k_cells_views(CellGet g, uint64_t n, View *views, uint64_t *valid) {
    const uint64_t n_pad = (n + 63) & ~63ull;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_pad; j += (uint64_t)gridDim.x * 256) {
        bool ok = false;
        if (j < n) {
            const uint8_t *p = nullptr;
            uint32_t len = 0;
            ok = g.get(j, &p, &len) == 2 && !is_missing(p, len);
            views[j] = ok ? View{p, len, 1u} : View{nullptr, 0u, 0u};
        }
        const unsigned long long m = __ballot(ok);
        if ((threadIdx.x & 63) == 0) valid[j >> 6] = m;
    }
}

struct CellListCountF {
    CellGet g;
    __device__ uint64_t operator()(uint64_t j) const {
        const uint8_t *p;
        uint32_t len;
        if (g.get(j, &p, &len) != 2 || is_missing(p, len)) return 0;
        uint32_t c = 1;
        for (uint32_t i = 0; i < len; i++) c += p[i] == ',';
        return c;
    }
};

__global__ __launch_bounds__(256) void k_cells_list_valid(CellGet g, uint64_t n, uint64_t *valid) {
    const uint64_t n_pad = (n + 63) & ~63ull;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_pad; j += (uint64_t)gridDim.x * 256) {
        bool ok = false;
        if (j < n) {
            const uint8_t *p;
            uint32_t len;
            ok = g.get(j, &p, &len) == 2 && !is_missing(p, len);
        }
        const unsigned long long m = __ballot(ok);
        if ((threadIdx.x & 63) == 0) valid[j >> 6] = m;
    }
}

template <int kType>  // kVtInt / kVtFloat / kVtString
__global__ __launch_bounds__(256) void k_cells_list(CellGet g, uint64_t n, const uint64_t *__restrict__ goff, void *values,
                                                    uint32_t *child_valid, unsigned long long *err, uint32_t err_code) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) {
        uint64_t o = goff[j];
        if (goff[j + 1] == o) continue;
        const uint8_t *p;
        uint32_t len;
        g.get(j, &p, &len);
        uint32_t s = 0;
        for (uint32_t i = 0; i <= len; i++) {
            if (i == len || p[i] == ',') {
                const uint32_t el = i - s;
                const bool miss = is_missing(p + s, el);
                bool ok = !miss;
                if (kType == kVtString) {
                    ((View *)values)[o] = ok ? View{p + s, el, 1u} : View{nullptr, 0u, 0u};
                } else if (kType == kVtInt) {
                    int v = 0;
                    if (ok && !parse_i32(PtrSrc{p + s, (int)el}, 0, (int)el, &v)) {
                        ok = false;
                        atomicMin(err, (g.err_row(j) << 8) | err_code);
                    }
                    ((int32_t *)values)[o] = v;
                } else {
                    float v = 0.f;
                    if (ok) {
                        const int st = parse_f32(PtrSrc{p + s, (int)el}, 0, (int)el, &v);
                        if (st != 0 && !(st == 2 && defer_f32(err, p + s, el, (float *)values + o, g.err_row(j)))) {
                            ok = false;
                            atomicMin(err, (g.err_row(j) << 8) | err_code);
                        }
                    }
                    ((float *)values)[o] = v;
                }
                if (ok) atomicOr(&child_valid[o >> 5], 1u << (o & 31));
                o++;
                s = i + 1;
            }
        }
    }
}

// ---- percent-decoding of String / Character values --------------------------------------------------------------------
__device__ __forceinline__ int hex_val(uint32_t c) {
    if (c - '0' <= 9u) return (int)(c - '0');
    c |= 0x20;
    if (c - 'a' <= 5u) return (int)(c - 'a') + 10;
    return -1;
}
// decoded length of p[0, len) when it holds at least one %XX escape, else 0 (nothing to decode)
__device__ __forceinline__ uint32_t percent_len(const uint8_t *p, uint32_t len) {
    uint32_t esc = 0;
    for (uint32_t i = 0; i + 2 < len; i++)
        if (p[i] == '%' && hex_val(p[i + 1]) >= 0 && hex_val(p[i + 2]) >= 0) esc++, i += 2;
    return esc ? len - 2 * esc : 0u;
}
__global__ __launch_bounds__(256) void k_percent_count(const View *__restrict__ views, uint64_t m, unsigned long long *total) {
    unsigned long long mine = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (uint64_t)gridDim.x * 256) {
        const View v = views[j];
        if (v.valid && v.len >= 3) mine += percent_len(v.p, v.len);
    }
    if (mine) atomicAdd(total, mine);
}
__global__ __launch_bounds__(256) void k_percent_decode(View *views, uint64_t m, uint8_t *side, unsigned long long *cursor, PercentRows rows,
                                                        unsigned long long *err, uint32_t err_code) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (uint64_t)gridDim.x * 256) {
        const View v = views[j];
        if (!v.valid || v.len < 3) continue;
        const uint32_t dl = percent_len(v.p, v.len);
        if (!dl) continue;
        uint8_t *dst = side + atomicAdd(cursor, (unsigned long long)dl);
        uint32_t o = 0;
        for (uint32_t i = 0; i < v.len; i++) {
            int h, l;
            if (v.p[i] == '%' && i + 2 < v.len && (h = hex_val(v.p[i + 1])) >= 0 && (l = hex_val(v.p[i + 2])) >= 0) {
                dst[o++] = (uint8_t)(h * 16 + l);
                i += 2;
            } else {
                dst[o++] = v.p[i];
            }
        }
        views[j] = View{dst, dl, 1u};
        if (!utf8_valid_global(dst, 0, dl)) {  // percent_decode(..).decode_utf8() fails: a value error of the row
            uint64_t parent = j;
            if (rows.d_parent_goff) {  // the list the element belongs to: last parent whose first element is <= j
                uint64_t lo = 0, hi = rows.n_parent;
                while (lo + 1 < hi) {
                    const uint64_t mid = (lo + hi) / 2;
                    if (rows.d_parent_goff[mid] <= j) lo = mid; else hi = mid;
                }
                parent = lo;
            }
            const unsigned long long row = rows.d_elem_row ? rows.d_elem_row[parent] : parent;
            atomicMin(err, (row << 8) | err_code);
        }
    }
}

__global__ __launch_bounds__(256) void k_views_validity(const View *views, uint64_t n, uint64_t *valid) {
    const uint64_t n_pad = (n + 63) & ~63ull;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_pad; j += (uint64_t)gridDim.x * 256) {
        const bool ok = j < n && views[j].valid;
        const unsigned long long m = __ballot(ok);
        if ((threadIdx.x & 63) == 0) valid[j >> 6] = m;
    }
}

}  // namespace

// ---- host wrappers ---------------------------------------------------------------------------------------------------
void list_counts(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint8_t sep, uint64_t *d_goff, uint64_t *d_tmp,
                 hipStream_t stream) {
    launch_xscan(ListCountF{ColGet2{c, d_row_map}, sep}, n, d_goff, d_tmp, stream);
}
void list_views(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint8_t sep, const uint64_t *d_goff, View *d_views,
                hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_list_views, dim3(grid_for(n)), dim3(256), 0, stream, ColGet2{c, d_row_map}, n, sep, d_goff, d_views);
}
void info_cells(const StrCol &info, const uint32_t *d_row_map, uint64_t n, const VtKeys &keys, VtCell *d_cells,
                hipStream_t stream) {
    if (!n || !keys.n) return;
    hipLaunchKernelGGL(k_info_cells, dim3(grid_for(n)), dim3(256), 0, stream, ColGet2{info, d_row_map}, n, keys, d_cells);
}
void sample_counts(const StrCol &rest, const uint64_t *d_rest_valid, const uint32_t *d_row_map, uint64_t n, uint64_t *d_goff,
                   uint64_t *d_tmp, hipStream_t stream) {
    launch_xscan(SampleCountF{ColGet2{rest, d_row_map}, d_rest_valid}, n, d_goff, d_tmp, stream);
}
void sample_cells(const StrCol &rest, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff, const VtKeys &keys,
                  VtCell *d_cells, View *d_sample_field, uint32_t *d_sample_row, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_sample_cells, dim3(grid_for(n)), dim3(256), 0, stream, ColGet2{rest, d_row_map}, n, d_goff, keys,
                       d_cells, d_sample_field, d_sample_row);
}

void cells_to_i32(const CellSrc &s, uint64_t n, int32_t *d_values, uint64_t *d_valid, unsigned long long *d_err,
                  uint32_t err_code, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_cells_scalar<kVtInt>, dim3(grid_for(n)), dim3(256), 0, stream, CellGet{s}, n, (void *)d_values, d_valid,
                       d_err, err_code);
}
void cells_to_f32(const CellSrc &s, uint64_t n, float *d_values, uint64_t *d_valid, unsigned long long *d_err,
                  uint32_t err_code, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_cells_scalar<kVtFloat>, dim3(grid_for(n)), dim3(256), 0, stream, CellGet{s}, n, (void *)d_values,
                       d_valid, d_err, err_code);
    hipLaunchKernelGGL(k_f32_slow, dim3(1), dim3(64), 0, stream, reinterpret_cast<ErrBlock *>(d_err), err_code);
}
void cells_to_flag(const CellSrc &s, uint64_t n, uint64_t *d_bits, uint64_t *d_valid, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_cells_flag, dim3(grid_for(n)), dim3(256), 0, stream, CellGet{s}, n, d_bits, d_valid);
}
void cells_to_views(const CellSrc &s, uint64_t n, View *d_views, uint64_t *d_valid, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_cells_views, dim3(grid_for(n)), dim3(256), 0, stream, CellGet{s}, n, d_views, d_valid);
}
void cells_list_counts(const CellSrc &s, uint64_t n, uint64_t *d_goff, uint64_t *d_tmp, uint64_t *d_valid, hipStream_t stream) {
    launch_xscan(CellListCountF{CellGet{s}}, n, d_goff, d_tmp, stream);
    if (!n) return;
    hipLaunchKernelGGL(k_cells_list_valid, dim3(grid_for(n)), dim3(256), 0, stream, CellGet{s}, n, d_valid);
}
void cells_list_i32(const CellSrc &s, uint64_t n, const uint64_t *d_goff, int32_t *d_values, uint32_t *d_child_valid,
                    unsigned long long *d_err, uint32_t err_code, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_cells_list<kVtInt>, dim3(grid_for(n)), dim3(256), 0, stream, CellGet{s}, n, d_goff, (void *)d_values,
                       d_child_valid, d_err, err_code);
}
void cells_list_f32(const CellSrc &s, uint64_t n, const uint64_t *d_goff, float *d_values, uint32_t *d_child_valid,
                    unsigned long long *d_err, uint32_t err_code, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_cells_list<kVtFloat>, dim3(grid_for(n)), dim3(256), 0, stream, CellGet{s}, n, d_goff, (void *)d_values,
                       d_child_valid, d_err, err_code);
    hipLaunchKernelGGL(k_f32_slow, dim3(1), dim3(64), 0, stream, reinterpret_cast<ErrBlock *>(d_err), err_code);
}
void cells_list_views(const CellSrc &s, uint64_t n, const uint64_t *d_goff, View *d_views, uint32_t *d_child_valid,
                      hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_cells_list<kVtString>, dim3(grid_for(n)), dim3(256), 0, stream, CellGet{s}, n, d_goff, (void *)d_views,
                       d_child_valid, (unsigned long long *)nullptr, 0u);
}
void percent_count(const View *d_views, uint64_t m, unsigned long long *d_total, hipStream_t stream) {
    if (!m) return;
    hipLaunchKernelGGL(k_percent_count, dim3(grid_for(m)), dim3(256), 0, stream, d_views, m, d_total);
}
void percent_decode(View *d_views, uint64_t m, uint8_t *d_side, unsigned long long *d_cursor, const PercentRows &rows,
                    unsigned long long *d_err, uint32_t err_code, hipStream_t stream) {
    if (!m) return;
    hipLaunchKernelGGL(k_percent_decode, dim3(grid_for(m)), dim3(256), 0, stream, d_views, m, d_side, d_cursor, rows, d_err, err_code);
}
void views_validity(const View *d_views, uint64_t n, uint64_t *d_valid, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_views_validity, dim3(grid_for(n)), dim3(256), 0, stream, d_views, n, d_valid);
}

}  // namespace arrow
}  // namespace exg
