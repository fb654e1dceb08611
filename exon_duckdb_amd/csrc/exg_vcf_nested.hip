// exg_vcf_nested.hip — see exg_vcf_nested.hpp.  Semantics restated (not the code; parity unpinned beyond
// test_vcf_record_scan.test:10-19, which pins alt = [<*>], info.indel = NULL, info.dp = 1 for row 1 of vcf/index.vcf):
// exon 0.2.6 datasources::vcf::{VCFSchemaBuilder, VCFArrayBuilder} over noodles-vcf 0.34.0 (rust/src/arrow_reader.rs:116-153):
//   * ID split on ';', ALT on ',', FILTER on ';'; "." => empty list;
//   * INFO "." => every child NULL; else ';'-separated key[=value]; an absent key => NULL, a value "." => NULL; Flag => true
//     when present; Integer => i32, Float => f32, String / Character => the percent-decoded text; Number other than 1 => list
//     split on ',', element "." => NULL; undeclared keys are not columns; a repeated key keeps its first value;
//   * FORMAT keys by position, each sample ':'-split the same way; trailing fields left out => NULL.
// A number that does not parse is a record error (EXG_PE_VCF_INFO / EXG_PE_VCF_FORMAT) at its row.
#include "exg_vcf_nested.hpp"

#include "exg_common.hpp"
#include "exg_float_slow.hpp"
#include "exg_parse.hpp"

namespace exg {
namespace vn {

namespace {

enum { kCount = 0, kWrite = 1, kCountSamples = 2 };

#define EXG_LDS __attribute__((address_space(3)))
typedef EXG_LDS const uint8_t *lds_cptr;
typedef uint32_t u32_a1 __attribute__((aligned(1)));
typedef uint32_t v3u_a1 __attribute__((ext_vector_type(3), aligned(1)));

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a field of a row: where its bytes are on the device, how many, and the host address string_t pointers give byte 0
struct Fld {
    const uint8_t *g;
    uint32_t len;
    uint64_t hptr;  // (only fields of more than 12 bytes have one; values cut from a shorter field are inlined anyway)
};
__device__ __forceinline__ Fld field_of(const exg_string_t *slot, const uint8_t *d_base, uint64_t payload_base) {
    const uint4 v = *reinterpret_cast<const uint4 *>(slot);
    Fld f;
    f.len = v.x;
    if (v.x <= EXG_INLINE_LENGTH) {  // the bytes are in the column itself
        f.g = reinterpret_cast<const uint8_t *>(slot) + 4;
        f.hptr = 0;
    } else {
        const uint64_t p = (uint64_t)v.z | ((uint64_t)v.w << 32);
        f.g = d_base + (p - payload_base);
        f.hptr = p;
    }
    return f;
}

// Byte source of the parsers: field positions [lo, hi) are staged in LDS (readable up to `lim`: slack behind the staged
// bytes, whatever it holds), everything else is read from global memory.
struct Txt {
    lds_cptr l;  // LDS address of field position 0 (only [lo, lim) may be read through it)
    int lo, hi, lim;
    const uint8_t *g;  // global address of field position 0
    int len;
    __device__ __forceinline__ uint32_t b(int i) const { return (i >= lo && i < hi) ? (uint32_t)l[i] : (uint32_t)g[i]; }
    __device__ __forceinline__ uint32_t u32(int i) const {
        if (i >= lo && i + 4 <= lim) return *reinterpret_cast<EXG_LDS const u32_a1 *>(l + i);
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + k < len) w |= (uint32_t)g[i + k] << (8 * k);
        return w;
    }
    __device__ __forceinline__ void u96(int i, uint32_t *w0, uint32_t *w1, uint32_t *w2) const {
        if (i >= lo && i + 12 <= lim) {
            const v3u_a1 w = *reinterpret_cast<EXG_LDS const v3u_a1 *>(l + i);
            *w0 = w.x, *w1 = w.y, *w2 = w.z;
            return;
        }
        uint32_t w[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 12; k++)
            if (i + k < len) w[k >> 2] |= (uint32_t)g[i + k] << (8 * (k & 3));
        *w0 = w[0], *w1 = w[1], *w2 = w[2];
    }
};

// duckdb::string_t of t[s, s + n)
__device__ __forceinline__ uint4 make_str(const Txt &t, int s, uint32_t n, uint64_t hptr) {
    uint4 r;
    r.x = n;
    if (n <= EXG_INLINE_LENGTH) {
        uint32_t w0, w1, w2;
        t.u96(s, &w0, &w1, &w2);
        const uint32_t l1 = n > 4 ? n - 4 : 0, l2 = n > 8 ? n - 8 : 0;
        r.y = n >= 4 ? w0 : w0 & ((1u << (8 * n)) - 1u);
        r.z = l1 >= 4 ? w1 : w1 & ((1u << (8 * l1)) - 1u);
        r.w = l2 >= 4 ? w2 : w2 & ((1u << (8 * l2)) - 1u);
    } else {
        const uint64_t p = hptr + (uint64_t)s;
        r.y = t.u32(s);
        r.z = (uint32_t)p;
        r.w = (uint32_t)(p >> 32);
    }
    return r;
}

struct Env {
    Ctl *ctl;
    uint32_t err_code;
    uint32_t *cnt;  // the list columns of this kind (INFO: over rows; FORMAT: over samples)
    uint64_t cnt_stride;
    const uint64_t *goff;
    uint64_t goff_stride;
    uint8_t *d_side;
    uint64_t side_cap, side_payload_base;
};
__device__ __forceinline__ void report(const Env &e, unsigned long long row) { atomicMin(&e.ctl->err, (row << 8) | e.err_code); }

__device__ __forceinline__ bool is_hex(uint32_t c) { return c - '0' <= 9u || (c | 0x20u) - 'a' <= 5u; }
__device__ __forceinline__ uint32_t hex_val(uint32_t c) { return c - '0' <= 9u ? c - '0' : (c | 0x20u) - 'a' + 10u; }

// String / Character value t[s, s + n): percent-decoded like noodles-vcf 0.34 (percent_encoding::percent_decode(..).decode_utf8():
// '%' + two hex digits is that byte, any other '%' stays).  Almost no value holds an escape: those that do are decoded into the
// batch's side buffer; one that does not fit there sets side_overflow and the host repeats the pass with a larger buffer.
__device__ uint4 str_value(const Txt &t, int s, int n, uint64_t hptr, const Env &env, unsigned long long err_row) {
    int esc = 0;
    for (int i = s; i + 2 < s + n; i++)
        if (t.b(i) == '%' && is_hex(t.b(i + 1)) && is_hex(t.b(i + 2))) esc++, i += 2;
    if (!esc) return make_str(t, s, (uint32_t)n, hptr);
    const uint32_t dl = (uint32_t)(n - 2 * esc);
    const unsigned long long off = atomicAdd(&env.ctl->side_used, (unsigned long long)dl);
    if (off + dl > env.side_cap) {
        env.ctl->side_overflow = 1u;
        return make_str(t, s, (uint32_t)n, hptr);
    }
    uint8_t *dst = env.d_side + off;
    uint32_t o = 0;
    for (int i = s; i < s + n; i++) {
        const uint32_t c = t.b(i);
        if (c == '%' && i + 2 < s + n && is_hex(t.b(i + 1)) && is_hex(t.b(i + 2))) {
            dst[o++] = (uint8_t)(hex_val(t.b(i + 1)) * 16u + hex_val(t.b(i + 2)));
            i += 2;
        } else {
            dst[o++] = (uint8_t)c;
        }
    }
    if (!utf8_valid_global(dst, 0, dl)) report(env, err_row);  // .decode_utf8() fails: a value error of the row
    return make_string_global(dst, 0, dl, env.side_payload_base + off);
}

__device__ __forceinline__ bool defer_f32(const Env &env, const uint8_t *p, uint32_t len, float *dst, unsigned long long row) {
    const unsigned int k = atomicAdd(&env.ctl->n_slow, 1u);
    if (k >= kSlowCap) return false;
    env.ctl->slow[k] = SlowF32{p, len, env.err_code, dst, row};
    return true;
}

// The value t[vs, ve) of key k for element idx (a row, or a sample).  kCount: the element count of a list key; kWrite: the typed
// value(s).  Returns whether the element is valid (not NULL); scalars are stored only when they are.
template <int MODE>
__device__ bool put_value(const Key &k, const KeyOut &o, const Txt &t, int vs, int ve, bool has_val, uint64_t idx, unsigned long long err_row,
                          const Env &env, uint64_t hptr) {
    if (k.type == kFlag) {  // present, with or without a value
        if (MODE == kWrite) reinterpret_cast<uint8_t *>(o.vals)[idx] = 1;
        return true;
    }
    if (!has_val) return false;
    const int n = ve - vs;
    if (n == 1 && t.b(vs) == '.') return false;  // the missing value
    if (!k.is_list) {
        if (MODE != kWrite) return true;
        if (k.type == kInt) {
            int v = 0;
            if (!parse_i32(t, vs, ve, &v)) {
                report(env, err_row);
                return false;
            }
            reinterpret_cast<int32_t *>(o.vals)[idx] = v;
        } else if (k.type == kFloat) {
            float v = 0.f;
            float *dst = reinterpret_cast<float *>(o.vals) + idx;
            const int st = parse_f32(t, vs, ve, &v);
            if (st != 0 && !(st == 2 && defer_f32(env, t.g + vs, (uint32_t)n, dst, err_row))) {
                report(env, err_row);
                return false;
            }
            *dst = v;
        } else {
            reinterpret_cast<uint4 *>(o.vals)[idx] = str_value(t, vs, n, hptr, env, err_row);
        }
        return true;
    }
    if (MODE != kWrite) {
        uint32_t c = 1;
        for (int i = vs; i < ve; i++) c += t.b(i) == ',';
        env.cnt[(uint64_t)k.list_idx * env.cnt_stride + idx] = c;
        return true;
    }
    uint64_t out = env.goff[(uint64_t)k.list_idx * env.goff_stride + idx];
    int s = vs;
    for (int i = vs; i <= ve; i++) {
        if (i != ve && t.b(i) != ',') continue;
        const int el = i - s;
        if (el == 1 && t.b(s) == '.') {  // a NULL element
            atomicAnd(&o.child_valid[out >> 5], ~(1u << (out & 31)));
            if (k.type == kString)
                reinterpret_cast<uint4 *>(o.child_vals)[out] = make_uint4(0, 0, 0, 0);
            else
                reinterpret_cast<uint32_t *>(o.child_vals)[out] = 0u;
        } else if (k.type == kInt) {
            int v = 0;
            if (!parse_i32(t, s, i, &v)) report(env, err_row);
            reinterpret_cast<int32_t *>(o.child_vals)[out] = v;
        } else if (k.type == kFloat) {
            float v = 0.f;
            float *dst = reinterpret_cast<float *>(o.child_vals) + out;
            const int st = parse_f32(t, s, i, &v);
            if (st != 0 && !(st == 2 && defer_f32(env, t.g + s, (uint32_t)el, dst, err_row))) report(env, err_row);
            *dst = v;
        } else {
            reinterpret_cast<uint4 *>(o.child_vals)[out] = str_value(t, s, el, hptr, env, err_row);
        }
        out++;
        s = i + 1;
    }
    return true;
}

// "this key has a valid element in the batch": a word per key, written by whoever finds one first (read first: one store per key, not per wave)
__device__ __forceinline__ void note_any(uint32_t *key_any, uint32_t q) {
    if (key_any && !__builtin_nontemporal_load(key_any + q)) key_any[q] = 1u;
}

// index of the key t[s, s + klen) with hash h, or -1
__device__ __forceinline__ int lookup(const KeyTab &kt, uint32_t h, const Txt &t, int s, int klen) {
    uint32_t slot = key_slot(h, kt.slot_mask);
    for (;;) {
        const uint32_t e = kt.slots[slot];
        if (!e) return -1;
        const Key k = kt.keys[e - 1];
        if (k.hash == h && (int)k.name_len == klen) {
            const uint8_t *nm = kt.names + k.name_off;
            bool eq = true;
            for (int i = 0; i < klen && eq; i++) eq = (uint32_t)nm[i] == t.b(s + i);
            if (eq) return (int)(e - 1);
        }
        slot = (slot + 1) & kt.slot_mask;
    }
}

// ---- k_rows: thread = row -----------------------------------------------------------------------------------------------
static constexpr int kKA = 32;            // INFO keys a header may declare for k_rows to take its INFO fields
static constexpr int kRowThreads = 128;
// A thread stages its field in an LDS row of its own, and the rows are what bounds the kernel's occupancy (A/B in one box,
// 5.3 M lines of 49 bytes: rows of 128 bytes + cells for 32 keys = 30 KB a block: children 0.83 ms; rows of 64 bytes + 8 keys =
// 15 KB: 0.62 ms).  So the kernel comes in two row sizes and two key-table sizes; the host picks per batch — the small row when
// (nearly) no INFO field of the batch is longer than 64 bytes (the counting pass counts those that are), the small table
// when the header declares at most 8 keys.  Fields longer than the row are k_info_wide's either way.
static constexpr int kStageSmall = 64, kStageLarge = 128, kKASmall = 8;
template <int STAGE>
struct RowGeo {
    static constexpr int kStage = STAGE;          // bytes of a field staged per thread
    static constexpr int kBlocks = STAGE / 16 + 1;  // aligned 16-byte blocks that cover them at any alignment
    static constexpr int kStride = (kBlocks * 4 + 4) | 1;  // dwords per thread: the blocks + slack for 12-byte reads at the field's end; odd: no bank conflicts
};
static constexpr uint32_t kCellAbsent = 0xFFFFu, kCellBare = 0xFFFEu;

template <class G>
__device__ __forceinline__ Txt stage_row(uint32_t *row, const Fld &f) {
    constexpr int kRowStage = G::kStage, kRowBlocks = G::kBlocks, kRowStride = G::kStride;
    const uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(f.g) & 15u);
    const uint8_t *al = f.g - lead;
    const uint32_t want = f.len < (uint32_t)kRowStage ? f.len : (uint32_t)kRowStage;
    const uint32_t nblk = (lead + want + 15u) >> 4;  // blocks that hold a byte of the field: inside the buffer
    uint4 blk[kRowBlocks];
#pragma unroll
    for (int q = 0; q < kRowBlocks; q++)
        blk[q] = (uint32_t)q < nblk ? *reinterpret_cast<const uint4 *>(al + 16 * q) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < kRowBlocks; q++) {
        if ((uint32_t)q < nblk) {
            row[4 * q + 0] = blk[q].x;
            row[4 * q + 1] = blk[q].y;
            row[4 * q + 2] = blk[q].z;
            row[4 * q + 3] = blk[q].w;
        }
    }
    Txt t;
    t.l = (lds_cptr)(EXG_LDS const uint32_t *)row + lead;
    t.lo = 0;
    t.hi = (int)want;
    // (whole blocks were loaded: the bytes up to the last block's end are the buffer's; behind a field that is staged whole
    // the slack of the row may be read too — whatever it holds lies behind the field)
    t.lim = f.len <= (uint32_t)kRowStage ? kRowStride * 4 - (int)lead : 16 * (int)nblk - (int)lead;
    t.g = f.g;
    t.len = (int)f.len;
    return t;
}

__device__ __forceinline__ bool fld_missing(const Txt &t, uint32_t len) { return len == 0 || (len == 1 && t.b(0) == '.'); }

// id / alt / filter of one row
template <int MODE, class G>
__device__ __forceinline__ void row_list(uint32_t *row, const exg_string_t *slot, const Batch &a, uint32_t sep, uint32_t *cnt_out, uint4 *elems,
                                         uint64_t out) {
    if (MODE == kCount) {
        const uint4 v = *reinterpret_cast<const uint4 *>(slot);
        if (v.x <= EXG_INLINE_LENGTH) {  // counted in registers
            const uint32_t sep4 = sep * 0x01010101u;
            uint32_t m = nib4(match4(v.y, sep4)) | (nib4(match4(v.z, sep4)) << 4) | (nib4(match4(v.w, sep4)) << 8);
            m &= (1u << v.x) - 1u;
            *cnt_out = (v.x == 0 || (v.x == 1 && (v.y & 255u) == '.')) ? 0u : 1u + (uint32_t)__popc(m);
            return;
        }
    }
    const Fld f = field_of(slot, a.d_base, a.payload_base);
    const Txt t = stage_row<G>(row, f);
    const bool none = fld_missing(t, f.len);
    if (MODE == kCount) {
        uint32_t c = none ? 0u : 1u;
        for (uint32_t i = 0; i < f.len; i++) c += t.b((int)i) == sep;
        *cnt_out = c;
        return;
    }
    if (none) return;
    int s = 0;
    for (int i = 0; i <= (int)f.len; i++) {
        if (i != (int)f.len && t.b(i) != sep) continue;
        elems[out++] = make_str(t, s, (uint32_t)(i - s), f.hptr);
        s = i + 1;
    }
}

template <int STAGE, int KA>
struct RowsLds {
    uint32_t stage[kRowThreads * RowGeo<STAGE>::kStride];
    uint16_t cells[KA * kRowThreads];
    Key keys[KA];
    uint32_t slots[2 * KA];
    uint8_t names[512];
};

template <int MODE, int STAGE, int KA>
__global__ __launch_bounds__(kRowThreads) void k_rows(Batch a, KeyTab kt_in, const KeyOut *__restrict__ ko, int take_info) {
    using G = RowGeo<STAGE>;
    __shared__ RowsLds<STAGE, KA> s;
    const uint32_t tid = threadIdx.x;
    uint32_t *const row = s.stage + tid * G::kStride;
    // the key table in LDS (<= kKA keys): the walk looks a key up per entry
    KeyTab kt = kt_in;
    if (take_info) {
        for (uint32_t i = tid; i < kt.n_keys; i += kRowThreads) s.keys[i] = kt_in.keys[i];
        for (uint32_t i = tid; i <= kt.slot_mask; i += kRowThreads) s.slots[i] = kt_in.slots[i];
        kt.keys = s.keys;
        kt.slots = s.slots;
        if (kt.names_bytes <= sizeof s.names) {
            for (uint32_t i = tid; i < kt.names_bytes; i += kRowThreads) s.names[i] = kt_in.names[i];
            kt.names = s.names;
        }
        __syncthreads();
    }
    Env env;
    env.ctl = a.ctl;
    env.err_code = EXG_PE_VCF_INFO;
    env.cnt = a.cnt + kColInfo0 * a.cnt_stride;
    env.cnt_stride = a.cnt_stride;
    env.goff = a.goff + kColInfo0 * a.goff_stride;
    env.goff_stride = a.goff_stride;
    env.d_side = a.d_side;
    env.side_cap = a.side_cap;
    env.side_payload_base = a.side_payload_base;
    const bool info_pass = take_info && kt.n_keys && (MODE == kWrite || kt.n_lists);
    const uint64_t n_pad = (a.n + kRowThreads - 1) / kRowThreads * kRowThreads;
    for (uint64_t j0 = (uint64_t)blockIdx.x * kRowThreads; j0 < n_pad; j0 += (uint64_t)gridDim.x * kRowThreads) {
        const uint64_t j = j0 + tid;
        const bool act = j < a.n;
        const uint64_t r = act ? (a.row_map ? (uint64_t)a.row_map[j] : j) : 0;
        if (act) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint32_t sep = c == 1 ? ',' : ';';
                uint64_t out = 0;
                if (MODE == kWrite) {
                    out = a.goff[c * a.goff_stride + j];
                    if (a.goff[c * a.goff_stride + j + 1] == out) continue;
                }
                row_list<MODE, G>(row, a.col[c] + r, a, sep, a.cnt + c * a.cnt_stride + j, reinterpret_cast<uint4 *>(a.elems[c]), out);
            }
        }
        if (MODE == kCount && a.mid_rows) {
            // INFO fields too long for the small row but not for the large one: what the host picks the writing pass's row size by
            const uint32_t il = act ? reinterpret_cast<const uint4 *>(a.col[3] + r)->x : 0u;
            const unsigned long long mid = __ballot(il > (uint32_t)kStageSmall && il <= (uint32_t)kStageLarge);
            if ((tid & 63u) == 0 && mid) atomicAdd(a.mid_rows, (unsigned long long)__popcll(mid));
        }
        if (!info_pass) continue;
        // INFO: the walk fills the row's cells (first occurrence of a key wins) ...
        for (uint32_t q = 0; q < kt.n_keys; q++) s.cells[q * kRowThreads + tid] = (uint16_t)kCellAbsent;
        Txt t;
        t.l = (lds_cptr)(EXG_LDS const uint32_t *)row;
        t.lo = t.hi = t.lim = 0;
        t.g = nullptr;
        t.len = 0;
        uint64_t hptr = 0;
        if (act) {
            const Fld f = field_of(a.col[3] + r, a.d_base, a.payload_base);
            hptr = f.hptr;
            if (f.len <= (uint32_t)STAGE) {  // (longer: k_info_wide's)
                t = stage_row<G>(row, f);
                const int len = (int)f.len;
                if (!fld_missing(t, f.len)) {
                    int p = 0;
                    while (p <= len) {
                        uint32_t h = kKeyHashSeed;
                        int e = p, eq = -1;
                        while (e < len) {
                            const uint32_t c = t.b(e);
                            if (c == ';') break;
                            if (eq < 0) {
                                if (c == '=') eq = e;
                                else h = key_hash_step(h, c);
                            }
                            e++;
                        }
                        const int kend = eq >= 0 ? eq : e;
                        if (kend > p) {
                            const int q = lookup(kt, h, t, p, kend - p);
                            if (q >= 0 && s.cells[q * kRowThreads + tid] == kCellAbsent)
                                s.cells[q * kRowThreads + tid] = (uint16_t)(eq >= 0 ? (uint32_t)(eq + 1) | ((uint32_t)(e - eq - 1) << 8) : kCellBare);
                        }
                        p = e + 1;
                    }
                }
            }
        }
        // ... and the children are written key by key: the type is the wavefront's, the stores are consecutive
        for (uint32_t q = 0; q < kt.n_keys; q++) {
            const Key k = kt.keys[q];
            if (MODE == kCount && !k.is_list) continue;
            const uint32_t c = s.cells[q * kRowThreads + tid];
            const bool present = act && c != kCellAbsent;
            const int vs = (int)(c & 255u), ve = vs + (int)(c >> 8);
            if (MODE == kCount) {
                uint32_t *dst = env.cnt + (uint64_t)k.list_idx * env.cnt_stride + j;
                if (act) *dst = 0u;
                if (present) (void)put_value<kCount>(k, KeyOut{nullptr, nullptr, nullptr, nullptr}, t, vs, ve, c < kCellBare, j, j, env, hptr);
                continue;
            }
            const KeyOut o = ko[q];
            bool valid = false;
            if (present) valid = put_value<kWrite>(k, o, t, vs, ve, c < kCellBare, j, j, env, hptr);
            if (act && !valid && !k.is_list) {  // NULL: the value is zero
                if (k.type == kString) reinterpret_cast<uint4 *>(o.vals)[j] = make_uint4(0, 0, 0, 0);
                else if (k.type == kFlag) reinterpret_cast<uint8_t *>(o.vals)[j] = 0;
                else reinterpret_cast<uint32_t *>(o.vals)[j] = 0u;
            }
            const unsigned long long m = __ballot(valid);
            const uint64_t w0 = j0 + (tid & ~63u);
            if ((tid & 63u) == 0 && w0 < a.n) {
                o.valid[w0 >> 6] = m;
                if (m) note_any(a.key_any, q);
            }
        }
    }
}

// ---- wave = row: pieces of a field through LDS, items found by their separators -------------------------------------------------
static constexpr int kPiece = 1024, kHalo = 128;
static constexpr int kBufBytes = kPiece + kHalo + 48;
struct WaveLds {
    uint32_t buf[kBufBytes / 4];
    uint16_t seps[kPiece + kHalo + 8];
};

// field positions [pb, se) -> LDS (aligned 16-byte blocks, the ones that hold a byte of the range)
__device__ __forceinline__ Txt wave_stage(WaveLds &w, const Fld &f, int pb, int se, uint32_t *sh_out) {
    const uint8_t *p0 = f.g + pb;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p0) & 15u);
    const uint8_t *al = p0 - sh;
    const uint32_t nblk = (sh + (uint32_t)(se - pb) + 15u) >> 4;
    for (uint32_t q = lane_id(); q < nblk; q += 64) {
        const uint4 v = *reinterpret_cast<const uint4 *>(al + 16 * q);
        w.buf[4 * q + 0] = v.x;
        w.buf[4 * q + 1] = v.y;
        w.buf[4 * q + 2] = v.z;
        w.buf[4 * q + 3] = v.w;
    }
    wave_sync();
    Txt t;
    t.l = (lds_cptr)(EXG_LDS const uint32_t *)w.buf + sh - pb;
    t.lo = pb;
    t.hi = se;
    t.lim = pb - (int)sh + 16 * (int)nblk + (se == (int)f.len ? 16 : 0);  // (whole blocks: the buffer's own bytes; + slack behind the field's end)
    t.g = f.g;
    t.len = (int)f.len;
    *sh_out = sh;
    return t;
}

// the separators among the staged bytes [0, span) (LDS byte sh + i): positions, ascending, into w.seps; *m_in = how many lie in
// front of in_span.  Lane = 16-byte chunk; the places come from a prefix sum of the chunks' counts.
__device__ __forceinline__ uint32_t wave_seps(WaveLds &w, uint32_t sh, uint32_t span, uint32_t in_span, uint32_t sep4, uint32_t *m_in) {
    const uint32_t nch = (sh + span + 15u) >> 4;
    uint32_t total = 0, total_in = 0;
    for (uint32_t c0 = 0; c0 < nch; c0 += 64) {
        const uint32_t c = c0 + lane_id();
        uint32_t m = 0, lo = 16u * c;
        if (c < nch) {
            const uint4 v = make_uint4(w.buf[4 * c], w.buf[4 * c + 1], w.buf[4 * c + 2], w.buf[4 * c + 3]);
            m = match16(v, sep4);
            if (lo < sh) m &= ~((1u << (sh - lo)) - 1u);                     // bytes in front of position 0
            if (lo + 16u > sh + span) m &= (sh + span > lo) ? (1u << (sh + span - lo)) - 1u : 0u;  // bytes behind the range
        }
        uint32_t mi = m;
        if (lo + 16u > sh + in_span) mi &= (sh + in_span > lo) ? (1u << (sh + in_span - lo)) - 1u : 0u;
        const uint32_t cnt = (uint32_t)__popc(m);
        const uint32_t incl = wave_incl_sum(cnt | ((uint32_t)__popc(mi) << 16));
        uint32_t out = total + (incl & 0xFFFFu) - cnt;
        while (m) {
            const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            w.seps[out++] = (uint16_t)(lo + k - sh);
        }
        const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        total += last & 0xFFFFu;
        total_in += last >> 16;
    }
    wave_sync();
    *m_in = total_in;
    return total;
}

// first separator in field positions [from, b), or b: read from global memory, 1 KiB per step
__device__ __forceinline__ int wave_first_sep(const Fld &f, int from, int b, uint32_t sep4) {
    const uint8_t *p0 = f.g + from;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p0) & 15u);
    const uint8_t *al = p0 - sh;
    const int64_t span = (int64_t)b - from;  // chunk c holds the positions from + 16c - sh ...
    for (int64_t c0 = 0; 16 * c0 < (int64_t)sh + span; c0 += 64) {
        const int64_t c = c0 + lane_id();
        const int64_t lo = 16 * c;
        uint32_t m = 0;
        if (lo < (int64_t)sh + span) {
            m = match16(*reinterpret_cast<const uint4 *>(al + lo), sep4);
            if (lo < (int64_t)sh) m &= ~((1u << (sh - (uint32_t)lo)) - 1u);
            if (lo + 16 > (int64_t)sh + span) m &= (1u << (uint32_t)((int64_t)sh + span - lo)) - 1u;
        }
        const unsigned long long any = __ballot(m != 0);
        if (any) {
            const int L = __ffsll((long long)any) - 1;
            const int pos = from + (int)(lo - sh) + __ffs((int)m) - 1;
            return __builtin_amdgcn_readlane(pos, L);
        }
    }
    return b;
}

// The items of f[a, b) split on a separator (exactly what bytes.split(sep) gives: one more item than separators), 64 at a
// time: fn(t, act, s, e, ord) is called by the whole wavefront, lane = item [s, e) number `ord`.  Items are handed out with
// the piece their first byte lies in; an item that ends behind the piece's halo has its end looked up in global memory.
template <class F>
__device__ __forceinline__ uint32_t wave_items(WaveLds &w, const Fld &f, int a, int b, uint32_t sep, F &&fn) {
    const uint32_t sep4 = sep * 0x01010101u;
    uint32_t ord0 = 0;
    bool carry = true;  // an item begins at the piece's first byte
    int pb = a;
    for (;;) {
        const int pe = pb + kPiece < b ? pb + kPiece : b, se = pe + kHalo < b ? pe + kHalo : b;
        wave_sync();  // (the piece before is done with)
        uint32_t sh;
        const Txt t = wave_stage(w, f, pb, se, &sh);
        uint32_t m_in;
        const uint32_t m = wave_seps(w, sh, (uint32_t)(se - pb), (uint32_t)(pe - pb), sep4, &m_in);
        // a separator at the piece's last byte begins an item that is the next piece's (unless the range ends there: an empty last item)
        const bool carry_out = m_in > 0 && pe < b && (int)w.seps[m_in - 1] == pe - 1 - pb;
        const uint32_t n_it = (carry ? 1u : 0u) + m_in - (carry_out ? 1u : 0u);
        // item i begins behind separator i - 1 (carry) / i, and ends at separator i (carry) / i + 1 — or where the next one is found
        int tail = b;
        if (n_it) {
            const uint32_t ei_last = carry ? n_it - 1 : n_it;
            if (ei_last >= m && se < b) tail = wave_first_sep(f, se, b, sep4);
        }
        for (uint32_t base = 0; base < n_it; base += 64) {
            const uint32_t i = base + lane_id();
            const bool act = i < n_it;
            int s = pb, e = tail;
            if (act) {
                const uint32_t si = carry ? i - 1 : i, ei = carry ? i : i + 1;
                if (!(carry && i == 0)) s = pb + (int)w.seps[si] + 1;
                if (ei < m) e = pb + (int)w.seps[ei];
            }
            fn(t, act, s, e, ord0 + i);
        }
        ord0 += n_it;
        carry = carry_out;
        if (pe >= b) break;
        pb = pe;
    }
    return ord0;
}

// rows of a group that a wave kernel takes: lane < rows_per_group loads its row's field, the set lanes are looped over
struct RowPick {
    unsigned long long mask;
    uint64_t j0;
    const uint8_t *g;
    uint32_t len;
    uint64_t hptr;
};
__device__ __forceinline__ uint64_t bcast64(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

static constexpr uint32_t kSeenWordsLds = 512;  // 16 384 keys per wave in LDS; wider headers keep their bits in global scratch

template <int MODE>
__global__ __launch_bounds__(256) void k_info_wide(Batch a, KeyTab kt, const KeyOut *__restrict__ ko, uint32_t rpg, int all_rows,
                                                   uint32_t *g_seen, uint32_t seen_words, uint32_t row_stage) {
    __shared__ WaveLds s_w[4];
    __shared__ uint32_t s_seen[4][kSeenWordsLds];
    const uint32_t wv = threadIdx.x >> 6, lane = lane_id();
    WaveLds &w = s_w[wv];
    const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + wv, n_waves = (uint64_t)gridDim.x * 4;
    uint32_t *seen = g_seen ? g_seen + wave_id * seen_words : s_seen[wv];
    Env env;
    env.ctl = a.ctl;
    env.err_code = EXG_PE_VCF_INFO;
    env.cnt = a.cnt + kColInfo0 * a.cnt_stride;
    env.cnt_stride = a.cnt_stride;
    env.goff = a.goff + kColInfo0 * a.goff_stride;
    env.goff_stride = a.goff_stride;
    env.d_side = a.d_side;
    env.side_cap = a.side_cap;
    env.side_payload_base = a.side_payload_base;
    const uint64_t groups = (a.n + rpg - 1) / rpg;
    for (uint64_t g = wave_id; g < groups; g += n_waves) {
        const uint64_t jl = g * rpg + lane;
        Fld mine;
        mine.g = nullptr, mine.len = 0, mine.hptr = 0;
        bool need = false;
        if (lane < rpg && jl < a.n) {
            const uint64_t r = a.row_map ? (uint64_t)a.row_map[jl] : jl;
            mine = field_of(a.col[3] + r, a.d_base, a.payload_base);
            need = mine.len > 0 && !(mine.len == 1 && mine.g[0] == '.') && (all_rows || mine.len > row_stage);
        }
        unsigned long long todo = __ballot(need);
        while (todo) {
            const int L = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            Fld f;
            f.g = reinterpret_cast<const uint8_t *>(bcast64(reinterpret_cast<uint64_t>(mine.g), L));
            f.len = (uint32_t)__builtin_amdgcn_readlane((int)mine.len, L);
            f.hptr = bcast64(mine.hptr, L);
            const uint64_t j = g * rpg + (uint64_t)L;
            for (uint32_t i = lane; i < seen_words; i += 64) seen[i] = 0u;
            wave_sync();
            wave_items(w, f, 0, (int)f.len, ';', [&](const Txt &t, bool act, int s, int e, uint32_t) {
                uint32_t h = kKeyHashSeed;
                int eq = -1;
                if (act) {
                    for (int i = s; i < e; i++) {
                        const uint32_t c = t.b(i);
                        if (c == '=') {
                            eq = i;
                            break;
                        }
                        h = key_hash_step(h, c);
                    }
                }
                const int kend = eq >= 0 ? eq : e;
                const int q = (act && kend > s) ? lookup(kt, h, t, s, kend - s) : -1;
                // the first occurrence of a key wins: across steps by the row's seen bits, inside a step by lane order
                const bool fresh = q >= 0 && !((seen[q >> 5] >> (q & 31)) & 1u);
                bool loser = false;
                unsigned long long cand = __ballot(fresh);
                while (cand) {
                    const int c = __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const int qc = __builtin_amdgcn_readlane(q, c);
                    loser = loser || ((uint32_t)c < lane_id() && qc == q);
                }
                const bool win = fresh && !loser;
                if (win) atomicOr(&seen[q >> 5], 1u << (q & 31));
                wave_sync();
                if (win) {
                    const Key k = kt.keys[q];
                    if (MODE == kCount) {
                        if (k.is_list) (void)put_value<kCount>(k, KeyOut{nullptr, nullptr, nullptr, nullptr}, t, eq + 1, e, eq >= 0, j, j, env, f.hptr);
                    } else {
                        const KeyOut o = ko[q];
                        if (put_value<kWrite>(k, o, t, eq + 1, e, eq >= 0, j, j, env, f.hptr)) {
                            atomicOr(reinterpret_cast<unsigned long long *>(o.valid) + (j >> 6), 1ull << (j & 63));
                            note_any(a.key_any, (uint32_t)q);
                        }
                    }
                }
            });
        }
    }
}

// ---- k_samples -------------------------------------------------------------------------------------------------------------
static constexpr uint32_t kMaxPos = 1024;  // FORMAT positions of one line (a line with more is a record error)
static constexpr uint32_t kFmtCache = 256;  // bytes of the line before's FORMAT string a wavefront keeps (with its positions: key_at)
struct FmtCache {
    uint32_t len, n_pos;
    uint8_t bytes[kFmtCache];
};

// 64 consecutive validity bits from element `base` on (words shared with neighbours: OR)
__device__ __forceinline__ void or_bits64(uint64_t *words, uint64_t base, unsigned long long m) {
    if (lane_id() != 0 || !m) return;
    const uint32_t sh = (uint32_t)(base & 63);
    unsigned long long *w = reinterpret_cast<unsigned long long *>(words) + (base >> 6);
    atomicOr(w, m << sh);
    if (sh && (m >> (64 - sh))) atomicOr(w + 1, m >> (64 - sh));
}

template <int MODE>
__global__ __launch_bounds__(256) void k_samples(Batch a, Samples sm, KeyTab kt, const KeyOut *__restrict__ ko, uint32_t rpg) {
    __shared__ WaveLds s_w[4];
    __shared__ int16_t s_key_at[4][kMaxPos];
    __shared__ FmtCache s_fc[4];
    const uint32_t wv = threadIdx.x >> 6, lane = lane_id();
    WaveLds &w = s_w[wv];
    int16_t *key_at = s_key_at[wv];
    FmtCache &fc = s_fc[wv];
    if (lane == 0) fc.len = 0, fc.n_pos = 0;
    wave_sync();
    const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + wv, n_waves = (uint64_t)gridDim.x * 4;
    Env env;
    env.ctl = a.ctl;
    env.err_code = EXG_PE_VCF_FORMAT;
    env.cnt = sm.cnt;
    env.cnt_stride = sm.cnt_stride;
    env.goff = sm.goff;
    env.goff_stride = sm.goff_stride;
    env.d_side = a.d_side;
    env.side_cap = a.side_cap;
    env.side_payload_base = a.side_payload_base;
    const uint64_t groups = (a.n + rpg - 1) / rpg;
    for (uint64_t g = wave_id; g < groups; g += n_waves) {
        const uint64_t jl = g * rpg + lane;
        Fld mine;
        mine.g = nullptr, mine.len = 0, mine.hptr = 0;
        bool need = false;
        if (lane < rpg && jl < a.n) {
            const uint64_t r = a.row_map ? (uint64_t)a.row_map[jl] : jl;
            const bool has = !a.rest_valid || ((a.rest_valid[r >> 6] >> (r & 63)) & 1ull);
            if (has) mine = field_of(a.col[4] + r, a.d_base, a.payload_base);
            need = has && mine.len > 0;
            if (MODE == kCountSamples) {
                if (!need) a.cnt[kColSamples * a.cnt_stride + jl] = 0u;
                else if (mine.len <= EXG_INLINE_LENGTH) {  // counted in registers: no wavefront turn for a line without samples
                    const uint4 v = *reinterpret_cast<const uint4 *>(mine.g - 4);
                    uint32_t m = nib4(match4(v.y, 0x09090909u)) | (nib4(match4(v.z, 0x09090909u)) << 4) | (nib4(match4(v.w, 0x09090909u)) << 8);
                    m &= (1u << v.x) - 1u;
                    a.cnt[kColSamples * a.cnt_stride + jl] = (uint32_t)__popc(m);
                    need = false;
                }
            } else {
                need = need && a.goff[kColSamples * a.goff_stride + jl + 1] != a.goff[kColSamples * a.goff_stride + jl];
            }
        }
        unsigned long long todo = __ballot(need);
        while (todo) {
            const int L = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            Fld f;
            f.g = reinterpret_cast<const uint8_t *>(bcast64(reinterpret_cast<uint64_t>(mine.g), L));
            f.len = (uint32_t)__builtin_amdgcn_readlane((int)mine.len, L);
            f.hptr = bcast64(mine.hptr, L);
            const uint64_t j = g * rpg + (uint64_t)L;
            if (MODE == kCountSamples) {
                // samples = tabs of the field (FORMAT is what stands in front of the first)
                const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(f.g) & 15u);
                const uint8_t *al = f.g - sh;
                const uint64_t end = (uint64_t)sh + f.len;
                uint32_t c = 0;
                for (uint64_t lo = 16ull * lane; lo < end; lo += 1024) {
                    uint32_t m = match16(ld_stream16(al + lo), 0x09090909u);
                    if (lo < sh) m &= ~((1u << (sh - (uint32_t)lo)) - 1u);
                    if (lo + 16 > end) m &= (1u << (uint32_t)(end - lo)) - 1u;
                    c += (uint32_t)__popc(m);
                }
                c = wave_incl_sum(c);
                if (lane == 63) a.cnt[kColSamples * a.cnt_stride + j] = c;
                continue;
            }
            const uint64_t S0 = a.goff[kColSamples * a.goff_stride + j];
            // ONE walk over the field's tab-separated items: item 0 is FORMAT (lane 0 of the first step), the samples follow.  FORMAT is
            // resolved when the first step arrives — out of the staged bytes, lane = key start; the line before's FORMAT and its
            // positions are kept per wavefront (a cohort file repeats one FORMAT string for millions of lines: the lookups were a third
            // of a 100-sample line's time)
            uint32_t n_pos = 0;
            bool bad_line = false;
            wave_items(w, f, 0, (int)f.len, '\t', [&](const Txt &t, bool act, int s, int e, uint32_t ord) {
                const bool first_step = ord == lane_id();  // (item 0 is lane 0's)
                if (first_step) {
                    const int fs = __builtin_amdgcn_readlane(s, 0), fe = __builtin_amdgcn_readlane(e, 0), flen = fe - fs;
                    bool hit = false;
                    if (flen > 0 && flen <= (int)kFmtCache && fe <= t.hi && (uint32_t)flen == fc.len) {  // the same string as the line before?
                        bool same = true;
                        for (int i = (int)lane_id(); i < flen; i += 64) same = same && t.b(fs + i) == fc.bytes[i];
                        hit = __ballot(!same) == 0ull;
                    }
                    if (hit) {
                        n_pos = fc.n_pos;
                    } else if (fe <= t.hi) {
                        // lane = byte, 64 at a time: a key starts at the field's first byte and behind every ':'; its position is the
                        // number of ':' in front of it; the lane walks its key (hash + lookup)
                        uint32_t before = 0;
                        bool prev_colon = true;
                        for (int base = 0; base <= flen; base += 64) {  // (position flen too: a key may be the empty string behind a last ':')
                            const int i = base + (int)lane_id();
                            const bool colon = i < flen && t.b(fs + i) == ':';
                            const unsigned long long cm = __ballot(colon);
                            const bool start = i <= flen && (lane_id() ? ((cm >> (lane_id() - 1)) & 1ull) != 0 : prev_colon);
                            const uint32_t pos = before + (uint32_t)__popcll(cm & ((1ull << lane_id()) - 1ull));
                            if (start && pos < kMaxPos) {
                                int ke = fs + i;
                                uint32_t h = kKeyHashSeed;
                                while (ke < fe && t.b(ke) != ':') h = key_hash_step(h, t.b(ke)), ke++;
                                key_at[pos] = (int16_t)(ke > fs + i ? lookup(kt, h, t, fs + i, ke - (fs + i)) : -1);
                            }
                            before += (uint32_t)__popcll(cm);
                            prev_colon = ((cm >> 63) & 1ull) != 0;
                        }
                        n_pos = before + 1;
                    } else {
                        // (a FORMAT field that does not fit a staged piece — more than ~1 100 bytes: one lane walks it)
                        uint32_t np = 0;
                        if (lane_id() == 0) {
                            int ks = fs;
                            for (;;) {
                                int ke = ks;
                                uint32_t h = kKeyHashSeed;
                                while (ke < fe && t.b(ke) != ':') h = key_hash_step(h, t.b(ke)), ke++;
                                if (np < kMaxPos) key_at[np] = (int16_t)(ke > ks ? lookup(kt, h, t, ks, ke - ks) : -1);
                                np++;
                                if (ke >= fe) break;
                                ks = ke + 1;
                            }
                        }
                        n_pos = (uint32_t)__builtin_amdgcn_readlane((int)np, 0);
                    }
                    if (!hit) {
                        wave_sync();
                        if (n_pos > kMaxPos) {
                            bad_line = true;  // more FORMAT positions than a line may have: a record error, never a dropped value
                            fc.len = 0;
                            if (lane_id() == 0) report(env, j);
                        } else {
                            // a key that stands twice keeps its first position
                            for (uint32_t p = lane_id(); p < n_pos; p += 64) {
                                const int q = key_at[p];
                                bool dup = false;
                                for (uint32_t p2 = 0; p2 < p && !dup && q >= 0; p2++) dup = key_at[p2] == q;
                                if (dup) key_at[p] = -1;  // (first occurrences are never rewritten: the search above always finds one)
                            }
                            // ... and the string is remembered for the next line
                            fc.len = 0;
                            if (flen > 0 && flen <= (int)kFmtCache && fe <= t.hi) {
                                for (int i = (int)lane_id(); i < flen; i += 64) fc.bytes[i] = (uint8_t)t.b(fs + i);
                                fc.len = (uint32_t)flen;
                                fc.n_pos = n_pos;
                            }
                        }
                        wave_sync();
                    }
                }
                if (bad_line) return;
                const bool smp = act && ord >= 1;  // (ord 0 is FORMAT itself)
                const uint64_t idx = S0 + ord - 1;
                const uint64_t base_idx = first_step ? S0 : S0 + (ord - lane_id()) - 1;
                if (MODE == kWrite && sm.srow && smp) sm.srow[idx] = (uint32_t)j;
                int cur = s;
                for (uint32_t p = 0; p < n_pos; p++) {
                    const bool has = smp && cur <= e;
                    if (!__ballot(has)) break;
                    int ve = cur;
                    if (has)
                        while (ve < e && t.b(ve) != ':') ve++;
                    const int q = key_at[p];  // (the same address for every lane)
                    if (q >= 0) {
                        const Key k = kt.keys[q];
                        if (MODE == kCount) {
                            if (k.is_list && has) (void)put_value<kCount>(k, KeyOut{nullptr, nullptr, nullptr, nullptr}, t, cur, ve, true, idx, j, env, f.hptr);
                        } else {
                            const KeyOut o = ko[q];
                            const bool valid = has && put_value<kWrite>(k, o, t, cur, ve, true, idx, j, env, f.hptr);
                            unsigned long long vm = __ballot(valid);
                            if (first_step) vm >>= 1;  // (lane 0 held FORMAT: sample 0 is lane 1's)
                            or_bits64(o.valid, base_idx, vm);
                        }
                    }
                    cur = ve + 1;
                }
            });
        }
    }
}

__global__ __launch_bounds__(64) void k_slow_floats(Ctl *c) {
    const unsigned int n = c->n_slow < kSlowCap ? c->n_slow : kSlowCap;
    for (unsigned int i = threadIdx.x; i < n; i += 64) {
        const SlowF32 e = c->slow[i];
        uint32_t bits = 0;
        if (f32_parse_exact(e.p, (int)e.len, &bits))
            atomicMin(&c->err, (e.row << 8) | e.code);
        else
            *e.dst = __uint_as_float(bits);
    }
}

// ---- prefix sums of many columns of counts ----------------------------------------------------------------------------------
static constexpr uint32_t kScanChunk = 4096;  // 1024 threads x 4

__global__ __launch_bounds__(1024) void k_bscan_local(const uint32_t *__restrict__ cnt, uint64_t cnt_stride, uint64_t n, uint64_t *goff,
                                                      uint64_t goff_stride, uint64_t *bsum, uint64_t nb) {
    __shared__ unsigned long long s_w[16];
    const uint64_t col = blockIdx.y, base = (uint64_t)blockIdx.x * kScanChunk;
    const uint32_t *src = cnt + col * cnt_stride;
    uint64_t *dst = goff + col * goff_stride;
    uint64_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        v[k] = idx < n ? (uint64_t)src[idx] : 0;
        sum += v[k];
    }
    unsigned long long incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += o;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long off = 0, tot = 0;
    for (uint32_t k = 0; k < 16; k++) {
        if (k < (threadIdx.x >> 6)) off += s_w[k];
        tot += s_w[k];
    }
    uint64_t run = off + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        if (idx < n) dst[idx] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) bsum[col * (nb + 2) + blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_bscan_blocks(uint64_t *bsum_all, uint64_t nb, uint64_t *totals) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_run;
    uint64_t *bsum = bsum_all + (uint64_t)blockIdx.x * (nb + 2);
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nb; base += 1024) {
        const uint64_t idx = base + threadIdx.x;
        const unsigned long long c = idx < nb ? bsum[idx] : 0;
        unsigned long long incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(incl, d, 64);
            if ((int)(threadIdx.x & 63) >= d) incl += o;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long off = 0;
        for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) off += s_w[k];
        const unsigned long long run = s_run;
        if (idx < nb) bsum[idx] = run + off + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = run + off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bsum[nb] = s_run;
        totals[blockIdx.x] = s_run;
    }
}

__global__ __launch_bounds__(1024) void k_bscan_add(uint64_t *goff, uint64_t goff_stride, uint64_t n, const uint64_t *__restrict__ bsum_all, uint64_t nb) {
    const uint64_t col = blockIdx.y, base = (uint64_t)blockIdx.x * kScanChunk;
    const uint64_t *bsum = bsum_all + col * (nb + 2);
    uint64_t *dst = goff + col * goff_stride;
    const uint64_t add = nb ? bsum[blockIdx.x] : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t idx = base + (uint64_t)threadIdx.x * 4 + k;
        if (idx < n) dst[idx] += add;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dst[n] = bsum[nb];
}

// ---- DuckDB list entries -------------------------------------------------------------------------------------------------
struct ListEntry {
    uint64_t offset, length;
};
__global__ __launch_bounds__(256) void k_entries_rows(const EntryJob *__restrict__ jobs, uint64_t n, uint64_t chunk_rows, uint64_t n_chunks) {
    const EntryJob job = jobs[blockIdx.y];
    ListEntry *ent = reinterpret_cast<ListEntry *>(job.entries);
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (ent) {
            const uint64_t o = job.goff[i], base = job.goff[i - i % chunk_rows];
            ent[i] = ListEntry{o - base, job.goff[i + 1] - o};
        }
        if (job.bases && i % chunk_rows == 0) job.bases[i / chunk_rows] = job.goff[i];
    }
    if (job.bases && blockIdx.x == 0 && threadIdx.x == 0) job.bases[n_chunks] = job.goff[n];
}
__global__ __launch_bounds__(256) void k_entries_elems(const EntryJob *__restrict__ jobs, uint64_t m, const uint32_t *__restrict__ elem_row,
                                                       const uint64_t *__restrict__ outer_goff, uint64_t n_rows, uint64_t chunk_rows, uint64_t n_chunks) {
    const EntryJob job = jobs[blockIdx.y];
    ListEntry *ent = reinterpret_cast<ListEntry *>(job.entries);
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < m; s += (uint64_t)gridDim.x * 256) {
        const uint64_t row = elem_row[s];
        const uint64_t first_elem = outer_goff[row - row % chunk_rows];  // first outer element of the row's DataChunk
        const uint64_t o = job.goff[s];
        ent[s] = ListEntry{o - job.goff[first_elem], job.goff[s + 1] - o};
    }
    if (job.bases)
        for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c <= n_chunks; c += (uint64_t)gridDim.x * 256) {
            const uint64_t row = c * chunk_rows < n_rows ? c * chunk_rows : n_rows;
            job.bases[c] = job.goff[outer_goff[row]];
        }
}

__global__ __launch_bounds__(256) void k_bytes_to_bits(const uint8_t *__restrict__ bytes, uint64_t m, uint64_t *bits) {
    const uint64_t m_pad = (m + 63) & ~63ull;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m_pad; j += (uint64_t)gridDim.x * 256) {
        const unsigned long long b = __ballot(j < m && bytes[j] != 0);
        if ((threadIdx.x & 63) == 0) bits[j >> 6] = b;
    }
}

inline uint32_t wave_grid(uint64_t n, uint32_t rpg) {
    const uint64_t groups = (n + rpg - 1) / rpg, blocks = (groups + 3) / 4;
    return (uint32_t)(blocks < 4096 ? (blocks ? blocks : 1) : 4096);
}
inline uint32_t rows_grid(uint64_t n) {
    const uint64_t blocks = (n + kRowThreads - 1) / kRowThreads;
    return (uint32_t)(blocks < 16384 ? (blocks ? blocks : 1) : 16384);
}

}  // namespace

bool rows_take_info(uint32_t n_info_keys) { return n_info_keys <= (uint32_t)kKA; }
uint32_t rows_stage_bytes(bool small_rows) { return small_rows ? kStageSmall : kStageLarge; }

uint64_t scan_tmp_entries(uint64_t n_cols, uint64_t n) { return n_cols * ((n + kScanChunk - 1) / kScanChunk + 2); }

template <int MODE>
static void launch_rows(const Batch &b, const KeyTab &info, const KeyOut *d_out, bool small_rows, hipStream_t s) {
    if (!b.n) return;
    const int take = rows_take_info(info.n_keys) ? 1 : 0;
    const bool small_tab = info.n_keys <= (uint32_t)kKASmall || !take;
    const dim3 grid(rows_grid(b.n)), block(kRowThreads);
    if (small_rows && small_tab) hipLaunchKernelGGL((k_rows<MODE, kStageSmall, kKASmall>), grid, block, 0, s, b, info, d_out, take);
    else if (small_rows) hipLaunchKernelGGL((k_rows<MODE, kStageSmall, kKA>), grid, block, 0, s, b, info, d_out, take);
    else if (small_tab) hipLaunchKernelGGL((k_rows<MODE, kStageLarge, kKASmall>), grid, block, 0, s, b, info, d_out, take);
    else hipLaunchKernelGGL((k_rows<MODE, kStageLarge, kKA>), grid, block, 0, s, b, info, d_out, take);
}
void rows_count(const Batch &b, const KeyTab &info, bool small_rows, hipStream_t s) { launch_rows<kCount>(b, info, nullptr, small_rows, s); }
void rows_write(const Batch &b, const KeyTab &info, const KeyOut *d_info_out, bool small_rows, hipStream_t s) {
    launch_rows<kWrite>(b, info, d_info_out, small_rows, s);
}
// wider headers than 16 384 keys: the seen bits of every wave of the launch live in global scratch
static uint32_t seen_words_of(const KeyTab &info) { return (info.n_keys + 31) / 32; }
size_t info_wide_seen_bytes(const KeyTab &info, uint64_t n, uint32_t rpg) {
    const uint32_t words = seen_words_of(info);
    return words > kSeenWordsLds ? (size_t)wave_grid(n, rpg) * 4 * words * 4 : 0;
}
void info_wide_count(const Batch &b, const KeyTab &info, uint32_t rpg, uint32_t *d_seen, bool small_rows, hipStream_t s) {
    if (!b.n || !info.n_keys || !info.n_lists) return;
    hipLaunchKernelGGL(k_info_wide<kCount>, dim3(wave_grid(b.n, rpg)), dim3(256), 0, s, b, info, (const KeyOut *)nullptr, rpg,
                       rows_take_info(info.n_keys) ? 0 : 1, seen_words_of(info) > kSeenWordsLds ? d_seen : (uint32_t *)nullptr, seen_words_of(info),
                       rows_stage_bytes(small_rows));
}
void info_wide_write(const Batch &b, const KeyTab &info, const KeyOut *d_info_out, uint32_t rpg, uint32_t *d_seen, bool small_rows, hipStream_t s) {
    if (!b.n || !info.n_keys) return;
    hipLaunchKernelGGL(k_info_wide<kWrite>, dim3(wave_grid(b.n, rpg)), dim3(256), 0, s, b, info, d_info_out, rpg, rows_take_info(info.n_keys) ? 0 : 1,
                       seen_words_of(info) > kSeenWordsLds ? d_seen : (uint32_t *)nullptr, seen_words_of(info), rows_stage_bytes(small_rows));
}
void samples_count(const Batch &b, uint32_t rpg, hipStream_t s) {
    if (!b.n) return;
    Samples none;
    none.S = 0, none.cnt = nullptr, none.cnt_stride = 0, none.goff = nullptr, none.goff_stride = 0, none.srow = nullptr;
    KeyTab nk;
    nk.keys = nullptr, nk.slots = nullptr, nk.names = nullptr, nk.n_keys = nk.slot_mask = nk.names_bytes = nk.n_lists = 0;
    hipLaunchKernelGGL(k_samples<kCountSamples>, dim3(wave_grid(b.n, rpg)), dim3(256), 0, s, b, none, nk, (const KeyOut *)nullptr, rpg);
}
void samples_count_lists(const Batch &b, const Samples &sm, const KeyTab &format, uint32_t rpg, hipStream_t s) {
    if (!b.n || !sm.S) return;
    hipLaunchKernelGGL(k_samples<kCount>, dim3(wave_grid(b.n, rpg)), dim3(256), 0, s, b, sm, format, (const KeyOut *)nullptr, rpg);
}
void samples_write(const Batch &b, const Samples &sm, const KeyTab &format, const KeyOut *d_format_out, uint32_t rpg, hipStream_t s) {
    if (!b.n || !sm.S) return;
    hipLaunchKernelGGL(k_samples<kWrite>, dim3(wave_grid(b.n, rpg)), dim3(256), 0, s, b, sm, format, d_format_out, rpg);
}
void fix_slow_floats(Ctl *ctl, hipStream_t s) { hipLaunchKernelGGL(k_slow_floats, dim3(1), dim3(64), 0, s, ctl); }

void scan_counts(const uint32_t *d_cnt, uint64_t cnt_stride, uint64_t n_cols, uint64_t n, uint64_t *d_goff, uint64_t goff_stride, uint64_t *d_totals,
                 uint64_t *d_tmp, hipStream_t s) {
    if (!n_cols) return;
    const uint64_t nb = (n + kScanChunk - 1) / kScanChunk;
    if (nb) hipLaunchKernelGGL(k_bscan_local, dim3((uint32_t)nb, (uint32_t)n_cols), dim3(1024), 0, s, d_cnt, cnt_stride, n, d_goff, goff_stride, d_tmp, nb);
    hipLaunchKernelGGL(k_bscan_blocks, dim3((uint32_t)n_cols), dim3(1024), 0, s, d_tmp, nb, d_totals);
    hipLaunchKernelGGL(k_bscan_add, dim3((uint32_t)(nb ? nb : 1), (uint32_t)n_cols), dim3(1024), 0, s, d_goff, goff_stride, n, d_tmp, nb);
}

void entries_rows(const EntryJob *d_jobs, uint32_t n_jobs, uint64_t n, uint64_t chunk_rows, uint64_t n_chunks, hipStream_t s) {
    if (!n_jobs) return;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_entries_rows, dim3((uint32_t)(blocks < 2048 ? (blocks ? blocks : 1) : 2048), n_jobs), dim3(256), 0, s, d_jobs, n, chunk_rows, n_chunks);
}
void entries_elems(const EntryJob *d_jobs, uint32_t n_jobs, uint64_t m, const uint32_t *d_elem_row, const uint64_t *d_outer_goff, uint64_t n_rows,
                   uint64_t chunk_rows, uint64_t n_chunks, hipStream_t s) {
    if (!n_jobs) return;
    const uint64_t blocks = (m + 255) / 256;
    hipLaunchKernelGGL(k_entries_elems, dim3((uint32_t)(blocks < 2048 ? (blocks ? blocks : 1) : 2048), n_jobs), dim3(256), 0, s, d_jobs, m, d_elem_row,
                       d_outer_goff, n_rows, chunk_rows, n_chunks);
}
void bytes_to_bits(const uint8_t *d_bytes, uint64_t m, uint64_t *d_bits, hipStream_t s) {
    if (!m) return;
    const uint64_t blocks = (m + 255) / 256;
    hipLaunchKernelGGL(k_bytes_to_bits, dim3((uint32_t)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, d_bytes, m, d_bits);
}

}  // namespace vn
}  // namespace exg
