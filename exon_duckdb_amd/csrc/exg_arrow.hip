// exg_arrow.hip — Arrow buffers built on the device from duckdb::string_t columns: Utf8 offsets and
// values, per-record-batch rebased offsets, row selection (the `filters` predicate of new_reader) and
// gathers through the row map.  What the reference gets from arrow-rs builders inside exon's
// FASTQArrayBuilder / FASTAArrayBuilder / VCFArrayBuilder and from DataFusion's FilterExec
// (rust/src/arrow_reader.rs:125-153), restated as HBM-resident scans and copies.
#include "exg_arrow.hpp"
#include "exg_scan.hpp"

namespace exg {
namespace arrow {

uint64_t scan_tmp_entries(uint64_t n) { return xscan_tmp_entries(n + 1); }

// ---- element accessors ------------------------------------------------------------------------------------
struct ColGet {
    StrCol c;
    const uint32_t *row_map;
    __device__ __forceinline__ uint64_t row(uint64_t j) const { return row_map ? (uint64_t)row_map[j] : j; }
    __device__ __forceinline__ uint32_t len(uint64_t j) const {
        return reinterpret_cast<const uint32_t *>(c.d_col)[row(j) * 4];
    }
    __device__ __forceinline__ const uint8_t *ptr(uint64_t j, uint32_t *len_out) const {
        const uint64_t r = row(j);
        const uint4 v = reinterpret_cast<const uint4 *>(c.d_col)[r];
        *len_out = v.x;
        if (v.x <= EXG_INLINE_LENGTH) return reinterpret_cast<const uint8_t *>(c.d_col) + r * 16 + 4;
        const uint64_t p = (uint64_t)v.z | ((uint64_t)v.w << 32);
        return c.d_base + (p - c.payload_base);
    }
};
template <class G>
struct LenF {
    G g;
    __device__ __forceinline__ uint64_t operator()(uint64_t j) const { return g.len(j); }
};

void utf8_goff_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint64_t *d_goff, uint64_t *d_tmp,
                        hipStream_t stream) {
    launch_xscan(LenF<ColGet>{ColGet{c, d_row_map}}, n, d_goff, d_tmp, stream);
}

// ---- values copy ----------------------------------------------------------------------------------------------
// Workgroup = 256 consecutive strings = one contiguous range of the values buffer.  Offsets and source
// pointers are staged in LDS; a thread assembles 4 output bytes at a time (binary search for the string
// that holds the first one) and stores an aligned dword.  Strings >= kBig are left to k_copy_big.
static constexpr uint32_t kBig = 8192;

template <class G>
__global__ __launch_bounds__(256) void k_utf8_copy(G g, uint64_t n, const uint64_t *__restrict__ goff, uint8_t *values,
                                                   uint32_t *d_big, uint32_t big_cap) {
    __shared__ uint64_t s_off[257];
    __shared__ const uint8_t *s_src[256];
    __shared__ uint32_t s_loc[257];  // prefix of the small strings' lengths
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_any_big;
    const uint64_t j0 = (uint64_t)blockIdx.x * 256;
    const uint32_t cnt = (uint32_t)(n - j0 < 256 ? n - j0 : 256);
    const uint32_t t = threadIdx.x;
    if (t == 0) s_any_big = 0;
    __syncthreads();
    uint32_t len = 0;
    if (t < cnt) {
        s_src[t] = g.ptr(j0 + t, &len);
        if (len >= kBig) {
            uint32_t slot = atomicAdd(&d_big[0], 1u);
            if (slot < big_cap) d_big[1 + slot] = (uint32_t)(j0 + t);
            s_any_big = 1;
            len = 0;
        }
    }
    for (uint32_t k = t; k <= cnt; k += 256) s_off[k] = goff[j0 + k];
    // local scan of the small lengths
    uint32_t incl = wave_incl_sum(len);
    if ((t & 63) == 63) s_wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t wo = 0;
    for (uint32_t k = 0; k < (t >> 6); k++) wo += s_wsum[k];
    s_loc[t] = wo + incl - len;
    if (t == 255) s_loc[256] = wo + incl;
    __syncthreads();
    if (!s_any_big) {
        const uint64_t A = s_off[0], B = s_off[cnt];
        for (uint64_t w = (A & ~3ull) + (uint64_t)t * 4; w < B; w += 1024) {
            const uint64_t b = w > A ? w : A;
            uint32_t lo = 0, hi = cnt;  // largest e < cnt with s_off[e] <= b
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (s_off[mid] <= b)
                    lo = mid;
                else
                    hi = mid;
            }
            uint32_t e = lo, word = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint64_t o = w + r;
                if (o >= A && o < B) {
                    while (s_off[e + 1] <= o) e++;
                    word |= (uint32_t)s_src[e][o - s_off[e]] << (8 * r);
                }
            }
            if (w >= A && w + 4 <= B) {
                *reinterpret_cast<uint32_t *>(values + w) = word;
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (w + r >= A && w + r < B) values[w + r] = (uint8_t)(word >> (8 * r));
            }
        }
    } else {
        const uint32_t total = s_loc[256];
        for (uint32_t lb = t; lb < total; lb += 256) {
            uint32_t lo = 0, hi = 256;
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (s_loc[mid] <= lb)
                    lo = mid;
                else
                    hi = mid;
            }
            values[s_off[lo] + (lb - s_loc[lo])] = s_src[lo][lb - s_loc[lo]];
        }
    }
}

template <class G>
__global__ __launch_bounds__(256) void k_copy_big(G g, const uint64_t *__restrict__ goff, uint8_t *values,
                                                  const uint32_t *d_big, uint32_t big_cap) {
    const uint32_t nbig = d_big[0] < big_cap ? d_big[0] : big_cap;
    for (uint32_t i = 0; i < nbig; i++) {
        const uint64_t j = d_big[1 + i];
        uint32_t len;
        const uint8_t *src = g.ptr(j, &len);
        uint8_t *dst = values + goff[j];
        const uint64_t mis = (uint64_t)(uintptr_t)dst & 15;  // start 16-byte groups on the destination's grid
        const uint64_t n_groups = (len + mis + 15) / 16;
        for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n_groups; q += (uint64_t)gridDim.x * 256) {
            const int64_t o = (int64_t)(q * 16) - (int64_t)mis;
            if (o >= 0 && o + 16 <= (int64_t)len) {
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < 16; r++) w[r >> 2] |= (uint32_t)src[o + r] << (8 * (r & 3));
                *reinterpret_cast<uint4 *>(dst + o) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                for (int r = 0; r < 16; r++)
                    if (o + r >= 0 && o + r < (int64_t)len) dst[o + r] = src[o + r];
            }
        }
    }
}

template <class G>
static void launch_copy(G g, uint64_t n, const uint64_t *d_goff, uint8_t *d_values, uint32_t *d_big, uint32_t big_cap,
                        hipStream_t stream) {
    (void)hipMemsetAsync(d_big, 0, 4, stream);
    if (!n) return;
    hipLaunchKernelGGL(k_utf8_copy<G>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, g, n, d_goff, d_values, d_big,
                       big_cap);
    hipLaunchKernelGGL(k_copy_big<G>, dim3(2048), dim3(256), 0, stream, g, d_goff, d_values, d_big, big_cap);
}

void utf8_copy_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff, uint8_t *d_values,
                        uint32_t *d_big, uint32_t big_cap, hipStream_t stream) {
    launch_copy(ColGet{c, d_row_map}, n, d_goff, d_values, d_big, big_cap, stream);
}

// ---- the payload of ONE projected column (a decoded input whose bytes live only in HBM) ----------------------------------------
// A string_t of more than 12 bytes points into the input.  When a projection selects a few columns of a compressed input, what
// crosses PCIe should be those columns' bytes, not the decoded file: the out-of-line strings of a column are closed up into a
// dense buffer (same scan + copy as the Arrow values buffer; inlined strings count as empty) and the column's pointers are
// rewritten to address that buffer's host copy.
struct ColGetLong {
    ColGet g;
    __device__ __forceinline__ uint32_t len(uint64_t j) const {
        const uint32_t l = g.len(j);
        return l > EXG_INLINE_LENGTH ? l : 0u;
    }
    __device__ __forceinline__ const uint8_t *ptr(uint64_t j, uint32_t *len_out) const {
        uint32_t l;
        const uint8_t *p = g.ptr(j, &l);
        *len_out = l > EXG_INLINE_LENGTH ? l : 0u;
        return p;
    }
};
void payload_goff_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, uint64_t *d_goff, uint64_t *d_tmp, hipStream_t stream) {
    launch_xscan(LenF<ColGetLong>{ColGetLong{ColGet{c, d_row_map}}}, n, d_goff, d_tmp, stream);
}
void payload_copy_from_col(const StrCol &c, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff, uint8_t *d_values, uint32_t *d_big,
                           uint32_t big_cap, hipStream_t stream) {
    launch_copy(ColGetLong{ColGet{c, d_row_map}}, n, d_goff, d_values, d_big, big_cap, stream);
}
__global__ __launch_bounds__(256) void k_repoint(const uint4 *__restrict__ in, const uint32_t *__restrict__ row_map, uint64_t n,
                                                 const uint64_t *__restrict__ goff, uint64_t new_base, uint4 *out) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) {
        uint4 v = in[row_map ? (uint64_t)row_map[j] : j];
        if (v.x > EXG_INLINE_LENGTH) {
            const uint64_t p = new_base + goff[j];
            v.z = (uint32_t)p, v.w = (uint32_t)(p >> 32);
        }
        out[j] = v;
    }
}
void repoint_strings(const exg_string_t *d_in, const uint32_t *d_row_map, uint64_t n, const uint64_t *d_goff, uint64_t new_base,
                     exg_string_t *d_out, hipStream_t stream) {
    if (!n) return;
    uint32_t grid = (uint32_t)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_repoint, dim3(grid), dim3(256), 0, stream, (const uint4 *)d_in, d_row_map, n, d_goff, new_base, (uint4 *)d_out);
}

// ---- quality_score_string_to_list -------------------------------------------------------------------------------
// Same shape as k_utf8_copy (workgroup = 256 consecutive strings = one contiguous range of the child vector),
// but every source byte widens to an int32, so a thread produces four values and stores them as one 16-byte
// group on the child vector's own 16-byte grid.  No big-string side path: the loop is output-centric.
__global__ __launch_bounds__(256) void k_quality_list(ColGet g, uint64_t n, const uint64_t *__restrict__ goff,
                                                      ListEntry *entries, int32_t *values, uint64_t values_cap) {
    __shared__ uint64_t s_off[257];
    __shared__ const uint8_t *s_src[256];
    if (goff[n] > values_cap) return;
    const uint64_t j0 = (uint64_t)blockIdx.x * 256;
    const uint32_t cnt = (uint32_t)(n - j0 < 256 ? n - j0 : 256);
    const uint32_t t = threadIdx.x;
    for (uint32_t k = t; k <= cnt; k += 256) s_off[k] = goff[j0 + k];
    if (t < cnt) {
        uint32_t len;
        s_src[t] = g.ptr(j0 + t, &len);
    }
    __syncthreads();
    if (t < cnt) entries[j0 + t] = ListEntry{s_off[t], s_off[t + 1] - s_off[t]};
    const uint64_t A = s_off[0], B = s_off[cnt];
    for (uint64_t w = (A & ~3ull) + (uint64_t)t * 4; w < B; w += 1024) {
        const uint64_t b = w > A ? w : A;
        uint32_t lo = 0, hi = cnt;  // largest e < cnt with s_off[e] <= b
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (s_off[mid] <= b)
                lo = mid;
            else
                hi = mid;
        }
        uint32_t e = lo;
        int32_t v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint64_t o = w + r;
            if (o >= A && o < B) {
                while (s_off[e + 1] <= o) e++;
                v[r] = (int32_t)(int8_t)s_src[e][o - s_off[e]] - 33;
            }
        }
        if (w >= A && w + 4 <= B) {
            *reinterpret_cast<int4 *>(values + w) = make_int4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (w + r >= A && w + r < B) values[w + r] = v[r];
        }
    }
}
void quality_list(const StrCol &c, uint64_t n, const uint64_t *d_goff, ListEntry *d_entries, int32_t *d_values,
                  uint64_t values_cap, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(k_quality_list, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, ColGet{c, nullptr}, n, d_goff,
                       d_entries, d_values, values_cap);
}

__global__ __launch_bounds__(256) void k_rebase(const uint64_t *__restrict__ goff, uint64_t n, uint64_t chunk_rows,
                                                int32_t *off32, uint64_t *chunk_base) {
    const uint64_t n_chunks = (n + chunk_rows - 1) / chunk_rows;
    const uint64_t total = n_chunks * (chunk_rows + 1);
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (uint64_t)gridDim.x * 256) {
        const uint64_t c = q / (chunk_rows + 1), i = q % (chunk_rows + 1);
        const uint64_t row = c * chunk_rows + i;
        const uint64_t base = goff[c * chunk_rows];
        if (i == 0) chunk_base[c] = base;
        off32[q] = row <= n ? (int32_t)(goff[row] - base) : 0;
    }
}
void rebase_offsets(const uint64_t *d_goff, uint64_t n, uint64_t chunk_rows, int32_t *d_off32, uint64_t *d_chunk_base,
                    hipStream_t stream) {
    if (!n) return;
    const uint64_t total = ((n + chunk_rows - 1) / chunk_rows) * (chunk_rows + 1);
    uint32_t grid = (uint32_t)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_rebase, dim3(grid), dim3(256), 0, stream, d_goff, n, chunk_rows, d_off32, d_chunk_base);
}

__global__ __launch_bounds__(256) void k_narrow(const uint64_t *__restrict__ goff, uint64_t n, int32_t *off32) {
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q <= n; q += (uint64_t)gridDim.x * 256)
        off32[q] = (int32_t)goff[q];
}
void narrow_offsets(const uint64_t *d_goff, uint64_t n, int32_t *d_off32, hipStream_t stream) {
    uint32_t grid = (uint32_t)((n + 256) / 256 < 8192 ? (n + 256) / 256 : 8192);
    hipLaunchKernelGGL(k_narrow, dim3(grid), dim3(256), 0, stream, d_goff, n, d_off32);
}

// ---- filters -----------------------------------------------------------------------------------------------------
// SQL three-valued logic over a postfix program; the operand stack is two bit fields (value, is-null).
struct FilterEval {
    const FilterProgram &prog;  // device memory
    const FilterCols &cols;
    const uint8_t *consts;

    __device__ bool is_null(uint32_t c, uint64_t r) const {
        const uint64_t *v = cols.validity[c];
        return v && !((v[r >> 6] >> (r & 63)) & 1);
    }
    __device__ int cmp_str(uint32_t c, uint64_t r, const FilterOp &op) const {
        ColGet g{StrCol{(const exg_string_t *)cols.data[c], cols.d_base[c], cols.payload_base[c]}, nullptr};
        uint32_t len;
        const uint8_t *p = g.ptr(r, &len);
        const uint8_t *q = consts + op.str_off;

        const uint32_t m = len < op.str_len ? len : op.str_len;
        for (uint32_t i = 0; i < m; i++) {
            int d = (int)p[i] - (int)q[i];
            if (d) return d < 0 ? -1 : 1;
        }
        return len < op.str_len ? -1 : len > op.str_len ? 1 : 0;
    }
    __device__ uint64_t operator()(uint64_t r) const {
        uint32_t vals = 0, nulls = 0;
        int sp = 0;
        for (uint32_t k = 0; k < prog.n_ops; k++) {
            const FilterOp &op = prog.ops[k];
            if (op.op == kOpAnd || op.op == kOpOr) {
                sp -= 2;
                const bool av = (vals >> sp) & 1, an = (nulls >> sp) & 1;
                const bool bv = (vals >> (sp + 1)) & 1, bn = (nulls >> (sp + 1)) & 1;
                bool rv, rn;
                if (op.op == kOpAnd) {
                    const bool any_false = (!an && !av) || (!bn && !bv);
                    rn = !any_false && (an || bn);
                    rv = !any_false && !rn;
                } else {
                    const bool any_true = (!an && av) || (!bn && bv);
                    rn = !any_true && (an || bn);
                    rv = any_true;
                }
                vals = (vals & ~(3u << sp)) | ((uint32_t)rv << sp);
                nulls = (nulls & ~(3u << sp)) | ((uint32_t)rn << sp);
                sp++;
                continue;
            }
            const uint32_t c = op.col;
            bool v = false, nul = false;
            if (op.op == kOpIsNull)
                v = is_null(c, r);
            else if (op.op == kOpIsNotNull)
                v = !is_null(c, r);
            else if (is_null(c, r))
                nul = true;
            else {
                int d;  // sign of column - literal; 2 = unordered (NaN)
                if (cols.kind[c] == kColStr) {
                    d = cmp_str(c, r, op);
                } else if (cols.kind[c] == kColI64 && op.lit == kLitInt) {
                    const int64_t x = ((const int64_t *)cols.data[c])[r];
                    d = x < op.i ? -1 : x > op.i ? 1 : 0;
                } else {
                    const double x = cols.kind[c] == kColI64 ? (double)((const int64_t *)cols.data[c])[r]
                                                             : (double)((const float *)cols.data[c])[r];
                    const double y = op.lit == kLitInt ? (double)op.i : op.f;
                    d = x < y ? -1 : x > y ? 1 : x == y ? 0 : 2;
                }
                switch (op.cmp) {
                    case kEq: v = d == 0; break;
                    case kNe: v = d != 0; break;
                    case kLt: v = d < 0; break;
                    case kLe: v = d <= 0; break;
                    case kGt: v = d > 0 && d != 2; break;
                    default: v = d >= 0 && d != 2; break;
                }
            }
            vals = (vals & ~(1u << sp)) | ((uint32_t)v << sp);
            nulls = (nulls & ~(1u << sp)) | ((uint32_t)nul << sp);
            sp++;
        }
        return (sp == 1 && (vals & 1) && !(nulls & 1)) ? 1 : 0;
    }
};

__global__ __launch_bounds__(256) void k_row_map(const uint64_t *__restrict__ goff, uint64_t n, uint32_t *row_map) {
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (uint64_t)gridDim.x * 256)
        if (goff[r + 1] != goff[r]) row_map[goff[r]] = (uint32_t)r;
}

void filter_rows(const FilterProgram *d_prog, const FilterCols *d_cols, const uint8_t *d_consts, uint64_t n,
                 uint64_t *d_goff_tmp, uint64_t *d_tmp, uint32_t *d_row_map, hipStream_t stream) {
    FilterEval f{*d_prog, *d_cols, d_consts};
    launch_xscan(f, n, d_goff_tmp, d_tmp, stream);
    if (!n) return;
    uint32_t grid = (uint32_t)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_row_map, dim3(grid), dim3(256), 0, stream, d_goff_tmp, n, d_row_map);
}

// ---- gathers -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_bits(const uint64_t *__restrict__ in, const uint32_t *__restrict__ row_map,
                                                     uint64_t n_out, uint64_t *out) {
    const uint64_t n_pad = (n_out + 63) & ~63ull;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_pad; j += (uint64_t)gridDim.x * 256) {
        bool bit = false;
        if (j < n_out) {
            const uint64_t r = row_map[j];
            bit = (in[r >> 6] >> (r & 63)) & 1;
        }
        const unsigned long long m = __ballot(bit);
        if ((threadIdx.x & 63) == 0) out[j >> 6] = m;
    }
}
void gather_bits(const uint64_t *d_in, const uint32_t *d_row_map, uint64_t n_out, uint64_t *d_out, hipStream_t stream) {
    if (!n_out) return;
    uint32_t grid = (uint32_t)((n_out + 255) / 256 < 8192 ? (n_out + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_gather_bits, dim3(grid), dim3(256), 0, stream, d_in, d_row_map, n_out, d_out);
}

template <class T>
__global__ __launch_bounds__(256) void k_gather(const T *__restrict__ in, const uint32_t *__restrict__ row_map, uint64_t n_out,
                                                T *out) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_out; j += (uint64_t)gridDim.x * 256)
        out[j] = in[row_map[j]];
}
void gather_u64(const uint64_t *d_in, const uint32_t *d_row_map, uint64_t n_out, uint64_t *d_out, hipStream_t stream) {
    if (!n_out) return;
    uint32_t grid = (uint32_t)((n_out + 255) / 256 < 8192 ? (n_out + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_gather<uint64_t>, dim3(grid), dim3(256), 0, stream, d_in, d_row_map, n_out, d_out);
}
void gather_u128(const void *d_in, const uint32_t *d_row_map, uint64_t n_out, void *d_out, hipStream_t stream) {
    if (!n_out) return;
    uint32_t grid = (uint32_t)((n_out + 255) / 256 < 8192 ? (n_out + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_gather<uint4>, dim3(grid), dim3(256), 0, stream, (const uint4 *)d_in, d_row_map, n_out, (uint4 *)d_out);
}
void gather_u32(const uint32_t *d_in, const uint32_t *d_row_map, uint64_t n_out, uint32_t *d_out, hipStream_t stream) {
    if (!n_out) return;
    uint32_t grid = (uint32_t)((n_out + 255) / 256 < 8192 ? (n_out + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_gather<uint32_t>, dim3(grid), dim3(256), 0, stream, d_in, d_row_map, n_out, d_out);
}

// ---- DuckDB vector layouts of the nested columns -------------------------------------------------------------------------
}  // namespace arrow
}  // namespace exg
