// exg_arrow_stream.cpp — `new_reader`, the reference's own FFI entry point (exon/include/rust.hpp:41-46,
// rust/src/arrow_reader.rs:38-166), on top of the device reader: an ArrowArrayStream whose record batches
// (<= batch_size rows) carry the reference's schema
//     FASTA  id, description, sequence                       (Utf8)
//     FASTQ  name, description, sequence, quality_scores     (Utf8)
//     VCF    chrom Utf8, pos Int64, id List<Utf8>, ref Utf8, alt List<Utf8>, qual Float32,
//            filter List<Utf8>, info Struct<##INFO keys>, formats List<Struct<##FORMAT keys>>
// so the unchanged C++ glue of the reference (WTArrowTableFunction::FileTypeBind / InitGlobal / Scan,
// module.cpp:75-294) can sit on top of it.  Every Arrow buffer is produced on the device (exg_arrow.hip,
// exg_vcf_typed.hip); this file parses the VCF header and the `filters` text, sizes the buffers, copies
// them back and wires the ArrowArray / ArrowSchema structs.
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "exg_arrow.hpp"
#include "exg_filter.hpp"
#include "exg_rd_fanout.hpp"
#include "exg_rd_source.hpp"
#include "exg_vcf_header.hpp"

using namespace exg_rd;
namespace ea = exg::arrow;

namespace {

// ---- schema -------------------------------------------------------------------------------------------------------------
struct Field {
    std::string name, format;
    bool nullable = true;
    std::vector<Field> children;
};

static_assert(kKeyFlag == ea::kVtFlag && kKeyInt == ea::kVtInt && kKeyFloat == ea::kVtFloat && kKeyString == ea::kVtString,
              "the header parser's value types are the emitter's");

Field key_field(const KeyDef &k) {
    Field f;
    f.name = k.id;
    const char *fmt = k.type == ea::kVtInt ? "i" : k.type == ea::kVtFloat ? "f" : k.type == ea::kVtFlag ? "b" : "u";
    if (k.is_list) {
        f.format = "+l";
        Field item;
        item.name = "item";
        item.format = fmt;
        f.children.push_back(item);
    } else {
        f.format = fmt;
    }
    return f;
}

// ---- memory ------------------------------------------------------------------------------------------------------------------
struct DevArena {  // bump allocator, reset per batch; what does not fit comes from the device pool, and the next batch's
                   // arena is as large as this batch turned out to need (hipMalloc / hipFree per batch cost milliseconds)
    char *base = nullptr;
    size_t cap = 0, used = 0, extra_bytes = 0, need = 0, min_extra = 1u << 20;
    std::vector<std::pair<void *, size_t>> extra;
    int dev = 0;
    void *alloc(size_t n) {
        n = (n + 255) & ~(size_t)255;
        if (n == 0) n = 256;
        if (used + n <= cap) {
            void *p = base + used;
            used += n;
            return p;
        }
        const size_t sz = n < min_extra ? min_extra : n;  // (1 MiB: small blocks would come one by one; 64 KiB under a memory budget)
        void *p = dev_pool()->take(dev, sz);
        if (!p) return nullptr;
        extra.emplace_back(p, sz);
        extra_bytes += sz;
        return p;
    }
    void reset() {
        need = used + extra_bytes;
        for (auto &e : extra) dev_pool()->give(dev, e.first, e.second);
        extra.clear();
        used = 0;
        extra_bytes = 0;
    }
    // before a batch: an arena of `want` bytes, or — when the batch before overflowed it — of what that one needed + 25 %
    void prepare(int device, size_t want) {
        dev = device;
        if (base && need > cap) {
            dev_pool()->give(dev, base, cap);
            base = nullptr, cap = 0;
            want = std::max(want, need + need / 4);
        }
        if (!base) {
            want = (want + 4095) & ~(size_t)4095;
            if ((base = (char *)dev_pool()->take(dev, want))) cap = want;
        }
    }
    ~DevArena() {
        reset();
        if (base) dev_pool()->give(dev, base, cap);
    }
};

// ---- one emitted column (host buffers of a whole device batch) ------------------------------------------------------
struct AColumn {
    enum Kind { kUtf8Top, kUtf8Abs, kPrim, kBool, kList, kStruct } kind = kPrim;
    int elem_size = 0;
    int64_t length = 0;                // elements of the batch-wide array
    const uint8_t *validity = nullptr;  // NULL: no nulls
    const uint8_t *offsets = nullptr;
    const uint8_t *data = nullptr;
    std::vector<uint64_t> chunk_base;  // kUtf8Top
    std::vector<AColumn> children;
};

struct ABatch {
    HostArena host;
    std::vector<AColumn> cols;
    uint64_t n_rows = 0;
};

struct StreamState {
    exg_reader *r = nullptr;
    std::vector<Field> schema;
    std::vector<KeyDef> info_keys, format_keys;
    bool has_filter = false;
    ea::FilterProgram prog;
    std::string consts;
    void *d_consts = nullptr, *d_prog = nullptr;
    void *d_info_names = nullptr, *d_format_names = nullptr;
    ea::VtKeys info_vt, format_vt;
    DevArena arena;
    std::shared_ptr<ABatch> batch;
    uint64_t batch_row = 0;
    // the device batch AFTER the one being handed out is produced on a thread of its own (scan, Arrow buffers, their way
    // back over PCIe: ~6 ms per 256 MiB) while the consumer walks the ~400 record batches of the current one
    std::shared_ptr<ABatch> produced;  // written by arrow_emit
    std::thread producer;
    int produced_state = 0;            // 1 a batch, 2 end of stream, 3 error (last_error)
    std::string last_error;
    size_t host_hint = 0;  // pinned bytes the previous batch needed
    hipStream_t copy_stream = nullptr;
    int copy_dev = 0;
    hipEvent_t copy_ev = nullptr;
    std::unique_ptr<FanOut> fan;  // several devices: the batches come from the stripes' streams (exg_rd_fanout.hpp)
    bool owns_reader = true;  // new_reader: the stream owns its reader; chunk mode: the reader owns this state
    // chunk mode: the columns' exg_type trees (node arrays and names live here)
    std::vector<std::unique_ptr<exg_type[]>> type_nodes;
    exg_type type_roots[16];
    ~StreamState() {
        fan.reset();  // (the workers' streams end first)
        if (copy_ev) (void)hipEventDestroy(copy_ev);
        if (copy_stream) stream_pool()->give(copy_dev, copy_stream);
        for (void *p : {d_consts, d_prog, d_info_names, d_format_names})
            if (p) (void)hipFree(p);
        arena.reset();
        if (owns_reader) delete r;
    }
};

#define EM_HIP(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) return fail(r, EXG_E_HIP, std::string(#expr " failed: ") + hipGetErrorString(_e)); \
    } while (0)

struct Emit {
    exg_reader *r;
    StreamState *st;
    HostArena *host;  // pinned memory of the batch being built
    hipStream_t s;
    uint64_t n;                 // output rows
    const uint32_t *d_row_map;  // NULL: identity
    unsigned long long *d_err;
    int rc = 0;
    bool copy = true;  // false: the column being built is not in the projection — built and validated on the device, not copied back

    void *dalloc(size_t bytes) {
        void *p = st->arena.alloc(bytes);
        if (!p && !rc) rc = fail(r, EXG_E_HIP, "out of device memory in the Arrow emitter");
        return p;
    }
    void *halloc(size_t bytes) {
        void *p = host->alloc(bytes);
        if (!p && !rc) rc = fail(r, EXG_E_HIP, "out of pinned host memory in the Arrow emitter");
        return p;
    }
    uint64_t fetch_u64(const uint64_t *d) {
        uint64_t v = 0;
        if (hipMemcpyAsync(&v, d, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            if (!rc) rc = fail(r, EXG_E_HIP, "device read failed in the Arrow emitter");
        }
        return v;
    }
    // copies back run on their own stream, behind an event on the kernels' stream: the next column's scans and copies
    // on the device overlap with this column's bytes crossing PCIe
    const uint8_t *to_host(const void *d, size_t bytes) {
        static const uint8_t nowhere[64] = {0};
        if (!copy) return nowhere;
        void *h = halloc(bytes);
        if (h && bytes) {
            hipError_t e = hipEventRecord(st->copy_ev, s);
            if (e == hipSuccess) e = hipStreamWaitEvent(st->copy_stream, st->copy_ev, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st->copy_stream);
            if (e != hipSuccess && !rc) rc = fail(r, EXG_E_HIP, "D2H copy failed in the Arrow emitter");
        }
        return (const uint8_t *)h;
    }
    static size_t bitmap_bytes(uint64_t m) { return (size_t)((m + 63) / 64) * 8; }

    // String / Character values are percent-decoded (noodles-vcf 0.34): the views that hold a %XX escape are decoded
    // into a side buffer on the device and point there afterwards.  *side_bytes == 0: no value held an escape.
    void decode_percent(ea::View *d_views, uint64_t m, const ea::PercentRows &rows, uint32_t err_code, uint8_t **d_side, uint64_t *side_bytes) {
        *d_side = nullptr;
        *side_bytes = 0;
        if (!m) return;
        unsigned long long *d_cnt = (unsigned long long *)dalloc(16);
        if (rc) return;
        (void)hipMemsetAsync(d_cnt, 0, 16, s);
        ea::percent_count(d_views, m, d_cnt, s);
        const uint64_t total = fetch_u64((const uint64_t *)d_cnt);
        if (!total || rc) return;
        uint8_t *side = (uint8_t *)dalloc(total + 16);
        if (rc) return;
        ea::percent_decode(d_views, m, side, d_cnt + 1, rows, d_err, err_code, s);
        *d_side = side;
        *side_bytes = total;
    }

    // validity of a row-space column whose source bitmap is indexed by scan rows
    const uint8_t *row_validity(const uint64_t *d_src) {
        if (!d_src) return nullptr;
        if (!d_row_map) return to_host(d_src, bitmap_bytes(n));
        uint64_t *d = (uint64_t *)dalloc(bitmap_bytes(n));
        if (!d) return nullptr;
        ea::gather_bits(d_src, d_row_map, n, d, s);
        return to_host(d, bitmap_bytes(n));
    }

    AColumn utf8_top(const ea::StrCol &c, const uint64_t *d_valid_src) {
        AColumn col;
        col.kind = AColumn::kUtf8Top;
        col.length = (int64_t)n;
        const uint64_t B = r->batch_rows, n_chunks = (n + B - 1) / B;
        uint64_t *d_goff = (uint64_t *)dalloc((n + 1) * 8);
        uint64_t *d_tmp = (uint64_t *)dalloc(ea::scan_tmp_entries(n) * 8);
        if (rc) return col;
        ea::utf8_goff_from_col(c, d_row_map, n, d_goff, d_tmp, s);
        const uint64_t total = fetch_u64(d_goff + n);
        const uint32_t big_cap = (uint32_t)(total / 8192 + 1);
        uint8_t *d_values = (uint8_t *)dalloc(total + 16);
        uint32_t *d_big = (uint32_t *)dalloc(4 * ((size_t)big_cap + 1));
        int32_t *d_off32 = (int32_t *)dalloc(n_chunks * (B + 1) * 4);
        uint64_t *d_cbase = (uint64_t *)dalloc(n_chunks * 8 + 8);
        if (rc) return col;
        ea::utf8_copy_from_col(c, d_row_map, n, d_goff, d_values, d_big, big_cap, s);
        ea::rebase_offsets(d_goff, n, B, d_off32, d_cbase, s);
        col.offsets = to_host(d_off32, n_chunks * (B + 1) * 4);
        col.data = to_host(d_values, total);
        col.chunk_base.resize(n_chunks + 1);
        if (n_chunks && hipMemcpyAsync(col.chunk_base.data(), d_cbase, n_chunks * 8, hipMemcpyDeviceToHost, s) != hipSuccess && !rc)
            rc = fail(r, EXG_E_HIP, "D2H copy failed in the Arrow emitter");
        col.chunk_base[n_chunks] = total;
        col.validity = row_validity(d_valid_src);
        return col;
    }

    // strings given as views; m elements; absolute int32 offsets
    AColumn utf8_abs(const ea::View *d_views, uint64_t m, const uint8_t *h_validity) {
        AColumn col;
        col.kind = AColumn::kUtf8Abs;
        col.length = (int64_t)m;
        col.validity = h_validity;
        uint64_t *d_goff = (uint64_t *)dalloc((m + 1) * 8);
        uint64_t *d_tmp = (uint64_t *)dalloc(ea::scan_tmp_entries(m) * 8);
        if (rc) return col;
        ea::utf8_goff_from_views(d_views, m, d_goff, d_tmp, s);
        const uint64_t total = fetch_u64(d_goff + m);
        if (total >= (1ull << 31)) {
            if (!rc) rc = fail(r, EXG_E_CAPACITY, "a nested string column exceeds Arrow's int32 offsets in one device batch");
            return col;
        }
        const uint32_t big_cap = (uint32_t)(total / 8192 + 1);
        uint8_t *d_values = (uint8_t *)dalloc(total + 16);
        uint32_t *d_big = (uint32_t *)dalloc(4 * ((size_t)big_cap + 1));
        int32_t *d_off32 = (int32_t *)dalloc((m + 1) * 4);
        if (rc) return col;
        ea::utf8_copy_from_views(d_views, m, d_goff, d_values, d_big, big_cap, s);
        ea::narrow_offsets(d_goff, m, d_off32, s);
        col.offsets = to_host(d_off32, (m + 1) * 4);
        col.data = to_host(d_values, total);
        return col;
    }

    // List<Utf8> out of a raw column split on `sep`
    AColumn list_of_strings(const ea::StrCol &c, uint8_t sep) {
        AColumn col;
        col.kind = AColumn::kList;
        col.length = (int64_t)n;
        uint64_t *d_goff = (uint64_t *)dalloc((n + 1) * 8);
        uint64_t *d_tmp = (uint64_t *)dalloc(ea::scan_tmp_entries(n) * 8);
        if (rc) return col;
        ea::list_counts(c, d_row_map, n, sep, d_goff, d_tmp, s);
        const uint64_t total = fetch_u64(d_goff + n);
        if (total >= (1ull << 31)) {
            if (!rc) rc = fail(r, EXG_E_CAPACITY, "a list column exceeds Arrow's int32 offsets in one device batch");
            return col;
        }
        ea::View *d_views = (ea::View *)dalloc(total * sizeof(ea::View));
        int32_t *d_off32 = (int32_t *)dalloc((n + 1) * 4);
        if (rc) return col;
        ea::list_views(c, d_row_map, n, sep, d_goff, d_views, s);
        ea::narrow_offsets(d_goff, n, d_off32, s);
        col.offsets = to_host(d_off32, (n + 1) * 4);
        col.children.push_back(utf8_abs(d_views, total, nullptr));
        return col;
    }

    // the typed children of INFO (elements = output rows) or FORMAT (elements = samples)
    std::vector<AColumn> cell_children(const std::vector<KeyDef> &keys, ea::CellSrc src, uint64_t m, uint32_t err_code) {
        std::vector<AColumn> kids;
        for (size_t k = 0; k < keys.size() && !rc; k++) {
            src.key = (uint32_t)k;
            const KeyDef &kd = keys[k];
            AColumn col;
            col.length = (int64_t)m;
            uint64_t *d_valid = (uint64_t *)dalloc(bitmap_bytes(m));
            if (rc) break;
            if (!kd.is_list) {
                if (kd.type == ea::kVtInt || kd.type == ea::kVtFloat) {
                    void *d_vals = dalloc(m * 4);
                    if (rc) break;
                    if (kd.type == ea::kVtInt)
                        ea::cells_to_i32(src, m, (int32_t *)d_vals, d_valid, d_err, err_code, s);
                    else
                        ea::cells_to_f32(src, m, (float *)d_vals, d_valid, d_err, err_code, s);
                    col.kind = AColumn::kPrim;
                    col.elem_size = 4;
                    col.data = to_host(d_vals, m * 4);
                    col.validity = to_host(d_valid, bitmap_bytes(m));
                } else if (kd.type == ea::kVtFlag) {
                    uint64_t *d_bits = (uint64_t *)dalloc(bitmap_bytes(m));
                    if (rc) break;
                    ea::cells_to_flag(src, m, d_bits, d_valid, s);
                    col.kind = AColumn::kBool;
                    col.data = to_host(d_bits, bitmap_bytes(m));
                    col.validity = to_host(d_valid, bitmap_bytes(m));
                } else {
                    ea::View *d_views = (ea::View *)dalloc(m * sizeof(ea::View));
                    if (rc) break;
                    ea::cells_to_views(src, m, d_views, d_valid, s);
                    uint8_t *d_side;
                    uint64_t side_bytes;
                    decode_percent(d_views, m, ea::PercentRows{nullptr, 0, src.d_elem_row}, err_code, &d_side, &side_bytes);
                    col = utf8_abs(d_views, m, to_host(d_valid, bitmap_bytes(m)));
                }
            } else {
                uint64_t *d_goff = (uint64_t *)dalloc((m + 1) * 8);
                uint64_t *d_tmp = (uint64_t *)dalloc(ea::scan_tmp_entries(m) * 8);
                if (rc) break;
                ea::cells_list_counts(src, m, d_goff, d_tmp, d_valid, s);
                const uint64_t total = fetch_u64(d_goff + m);
                if (total >= (1ull << 31)) {
                    rc = fail(r, EXG_E_CAPACITY, "a list column exceeds Arrow's int32 offsets in one device batch");
                    break;
                }
                col.kind = AColumn::kList;
                int32_t *d_off32 = (int32_t *)dalloc((m + 1) * 4);
                uint32_t *d_cv = (uint32_t *)dalloc(bitmap_bytes(total));
                if (rc) break;
                ea::narrow_offsets(d_goff, m, d_off32, s);
                (void)hipMemsetAsync(d_cv, 0, bitmap_bytes(total), s);
                col.offsets = to_host(d_off32, (m + 1) * 4);
                col.validity = to_host(d_valid, bitmap_bytes(m));
                AColumn child;
                child.length = (int64_t)total;
                if (kd.type == ea::kVtInt || kd.type == ea::kVtFloat) {
                    void *d_vals = dalloc(total * 4);
                    if (rc) break;
                    if (kd.type == ea::kVtInt)
                        ea::cells_list_i32(src, m, d_goff, (int32_t *)d_vals, d_cv, d_err, err_code, s);
                    else
                        ea::cells_list_f32(src, m, d_goff, (float *)d_vals, d_cv, d_err, err_code, s);
                    child.kind = AColumn::kPrim;
                    child.elem_size = 4;
                    child.data = to_host(d_vals, total * 4);
                    child.validity = to_host(d_cv, bitmap_bytes(total));
                } else {
                    ea::View *d_views = (ea::View *)dalloc(total * sizeof(ea::View));
                    if (rc) break;
                    ea::cells_list_views(src, m, d_goff, d_views, d_cv, s);
                    uint8_t *d_side;
                    uint64_t side_bytes;
                    decode_percent(d_views, total, ea::PercentRows{d_goff, m, src.d_elem_row}, err_code, &d_side, &side_bytes);
                    child = utf8_abs(d_views, total, to_host(d_cv, bitmap_bytes(total)));
                }
                col.children.push_back(std::move(child));
            }
            kids.push_back(std::move(col));
        }
        return kids;
    }
};

// ---- the same columns in DuckDB's vector layouts (chunk boundary) ----------------------------------------------------
struct DuckEmit {
    Emit &em;
    uint64_t B, n_chunks;       // rows per DataChunk, chunks of this batch
    const uint8_t *d_base;      // the scanned text on the device ...
    uint64_t payload_base;      // ... and the host address its bytes have in the chunk's payload

    // rows != NULL: String / Character values of INFO / FORMAT, percent-decoded first (the decoded bytes travel in a side
    // block of the batch's pinned arena; everything else stays a zero-copy pointer into the chunk's payload)
    NVec strings_from_views(ea::View *d_views, uint64_t m, const uint64_t *h_validity, const ea::PercentRows *rows = nullptr,
                            uint32_t err_code = 0) {
        NVec v;
        v.type = EXG_TYPE_VARCHAR;
        v.elem = 16;
        v.length = m;
        v.validity = h_validity;
        uint8_t *d_side = nullptr;
        uint64_t side_bytes = 0, side_host = 0;
        if (rows) em.decode_percent(d_views, m, *rows, err_code, &d_side, &side_bytes);
        if (side_bytes) side_host = (uint64_t)(uintptr_t)em.to_host(d_side, side_bytes);
        exg_string_t *d = (exg_string_t *)em.dalloc(m * 16 + 16);
        if (em.rc) return v;
        ea::views_to_string_t(d_views, m, d_base, payload_base, d_side, side_bytes, side_host, d, em.s);
        v.data = em.to_host(d, m * 16);
        return v;
    }
    // the n_chunks + 1 places where the chunks' elements begin, on the host (valid after the final sync) and on the device
    const uint64_t *bases_rows(const uint64_t *d_goff, uint64_t n, uint64_t **d_out) {
        uint64_t *d = (uint64_t *)em.dalloc((n_chunks + 1) * 8);
        if (em.rc) return nullptr;
        ea::chunk_bases_rows(d_goff, n, B, n_chunks, d, em.s);
        if (d_out) *d_out = d;
        return (const uint64_t *)em.to_host(d, (n_chunks + 1) * 8);
    }
    // LIST(VARCHAR) out of a raw column split on `sep`
    NVec list_of_strings(const ea::StrCol &c, uint8_t sep) {
        NVec v;
        v.type = EXG_TYPE_LIST;
        v.elem = 16;
        v.length = em.n;
        const uint64_t n = em.n;
        uint64_t *d_goff = (uint64_t *)em.dalloc((n + 1) * 8);
        uint64_t *d_tmp = (uint64_t *)em.dalloc(ea::scan_tmp_entries(n) * 8);
        if (em.rc) return v;
        ea::list_counts(c, em.d_row_map, n, sep, d_goff, d_tmp, em.s);
        const uint64_t total = em.fetch_u64(d_goff + n);
        ea::View *d_views = (ea::View *)em.dalloc(total * sizeof(ea::View) + 16);
        ea::ListEntry *d_entries = (ea::ListEntry *)em.dalloc(n * 16);
        if (em.rc) return v;
        ea::list_views(c, em.d_row_map, n, sep, d_goff, d_views, em.s);
        ea::list_entries_rows(d_goff, n, B, d_entries, em.s);
        v.data = em.to_host(d_entries, n * 16);
        v.child_base = bases_rows(d_goff, n, nullptr);
        v.children.push_back(strings_from_views(d_views, total, nullptr));  // (id / alt / filter are not percent-decoded)
        return v;
    }
    // the typed children of INFO (elements = output rows) or FORMAT (elements = samples: d_elem_row / d_outer_goff /
    // d_outer_bases describe the enclosing list)
    std::vector<NVec> cell_children(const std::vector<KeyDef> &keys, ea::CellSrc src, uint64_t m, uint32_t err_code,
                                    const uint32_t *d_elem_row, const uint64_t *d_outer_goff, const uint64_t *d_outer_bases) {
        std::vector<NVec> kids;
        for (size_t k = 0; k < keys.size() && !em.rc; k++) {
            src.key = (uint32_t)k;
            const KeyDef &kd = keys[k];
            NVec col;
            col.length = m;
            uint64_t *d_valid = (uint64_t *)em.dalloc(Emit::bitmap_bytes(m));
            if (em.rc) break;
            if (!kd.is_list) {
                if (kd.type == ea::kVtInt || kd.type == ea::kVtFloat) {
                    void *d_vals = em.dalloc(m * 4);
                    if (em.rc) break;
                    if (kd.type == ea::kVtInt)
                        ea::cells_to_i32(src, m, (int32_t *)d_vals, d_valid, em.d_err, err_code, em.s);
                    else
                        ea::cells_to_f32(src, m, (float *)d_vals, d_valid, em.d_err, err_code, em.s);
                    col.type = kd.type == ea::kVtInt ? EXG_TYPE_INTEGER : EXG_TYPE_FLOAT;
                    col.elem = 4;
                    col.data = em.to_host(d_vals, m * 4);
                    col.validity = (const uint64_t *)em.to_host(d_valid, Emit::bitmap_bytes(m));
                } else if (kd.type == ea::kVtFlag) {
                    uint64_t *d_bits = (uint64_t *)em.dalloc(Emit::bitmap_bytes(m));
                    uint8_t *d_bytes = (uint8_t *)em.dalloc(m + 16);
                    if (em.rc) break;
                    ea::cells_to_flag(src, m, d_bits, d_valid, em.s);
                    ea::bits_to_bytes(d_bits, m, d_bytes, em.s);
                    col.type = EXG_TYPE_BOOLEAN;
                    col.elem = 1;
                    col.data = em.to_host(d_bytes, m);
                    col.validity = (const uint64_t *)em.to_host(d_valid, Emit::bitmap_bytes(m));
                } else {
                    ea::View *d_views = (ea::View *)em.dalloc(m * sizeof(ea::View) + 16);
                    if (em.rc) break;
                    ea::cells_to_views(src, m, d_views, d_valid, em.s);
                    const ea::PercentRows pr{nullptr, 0, src.d_elem_row};
                    col = strings_from_views(d_views, m, (const uint64_t *)em.to_host(d_valid, Emit::bitmap_bytes(m)), &pr, err_code);
                }
            } else {
                uint64_t *d_goff = (uint64_t *)em.dalloc((m + 1) * 8);
                uint64_t *d_tmp = (uint64_t *)em.dalloc(ea::scan_tmp_entries(m) * 8);
                if (em.rc) break;
                ea::cells_list_counts(src, m, d_goff, d_tmp, d_valid, em.s);
                const uint64_t total = em.fetch_u64(d_goff + m);
                col.type = EXG_TYPE_LIST;
                col.elem = 16;
                ea::ListEntry *d_entries = (ea::ListEntry *)em.dalloc(m * 16 + 16);
                uint32_t *d_cv = (uint32_t *)em.dalloc(Emit::bitmap_bytes(total));
                uint64_t *d_cb = (uint64_t *)em.dalloc((n_chunks + 1) * 8);
                if (em.rc) break;
                if (d_elem_row) {
                    ea::list_entries_elems(d_goff, m, d_elem_row, d_outer_goff, B, d_entries, em.s);
                    ea::chunk_bases_pick(d_goff, d_outer_bases, n_chunks, d_cb, em.s);
                } else {
                    ea::list_entries_rows(d_goff, m, B, d_entries, em.s);
                    ea::chunk_bases_rows(d_goff, m, B, n_chunks, d_cb, em.s);
                }
                (void)hipMemsetAsync(d_cv, 0, Emit::bitmap_bytes(total), em.s);
                col.data = em.to_host(d_entries, m * 16);
                col.validity = (const uint64_t *)em.to_host(d_valid, Emit::bitmap_bytes(m));
                col.child_base = (const uint64_t *)em.to_host(d_cb, (n_chunks + 1) * 8);
                NVec child;
                child.length = total;
                if (kd.type == ea::kVtInt || kd.type == ea::kVtFloat) {
                    void *d_vals = em.dalloc(total * 4);
                    if (em.rc) break;
                    if (kd.type == ea::kVtInt)
                        ea::cells_list_i32(src, m, d_goff, (int32_t *)d_vals, d_cv, em.d_err, err_code, em.s);
                    else
                        ea::cells_list_f32(src, m, d_goff, (float *)d_vals, d_cv, em.d_err, err_code, em.s);
                    child.type = kd.type == ea::kVtInt ? EXG_TYPE_INTEGER : EXG_TYPE_FLOAT;
                    child.elem = 4;
                    child.data = em.to_host(d_vals, total * 4);
                    child.validity = (const uint64_t *)em.to_host(d_cv, Emit::bitmap_bytes(total));
                } else {
                    ea::View *d_views = (ea::View *)em.dalloc(total * sizeof(ea::View) + 16);
                    if (em.rc) break;
                    ea::cells_list_views(src, m, d_goff, d_views, d_cv, em.s);
                    const ea::PercentRows pr{d_goff, m, src.d_elem_row};
                    child = strings_from_views(d_views, total, (const uint64_t *)em.to_host(d_cv, Emit::bitmap_bytes(total)), &pr, err_code);
                }
                col.children.push_back(std::move(child));
            }
            kids.push_back(std::move(col));
        }
        return kids;
    }
};

int upload_keys(exg_reader *r, const std::vector<KeyDef> &keys, ea::VtKeys *vt, void **d_names) {
    if (keys.size() > (size_t)ea::kMaxVtKeys)
        return fail(r, EXG_E_UNSUPPORTED, "VCF header declares more than " + std::to_string(ea::kMaxVtKeys) + " INFO or FORMAT keys");
    std::string names;
    vt->n = (uint32_t)keys.size();
    for (size_t k = 0; k < keys.size(); k++) {
        vt->k[k].name_off = (uint32_t)names.size();
        vt->k[k].name_len = (uint32_t)keys[k].id.size();
        vt->k[k].type = keys[k].type;
        vt->k[k].is_list = keys[k].is_list;
        names += keys[k].id;
    }
    EM_HIP(hipMalloc(d_names, names.size() + 16));
    if (!names.empty()) EM_HIP(hipMemcpy(*d_names, names.data(), names.size(), hipMemcpyHostToDevice));
    vt->d_names = (const uint8_t *)*d_names;
    return EXG_OK;
}

// Called by next_batch with the scan's columns still in HBM.
static double em_now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
#define EM_TRACE(label)                                                                          \
    do {                                                                                         \
        if (getenv("EXG_TRACE")) {                                                               \
            (void)hipStreamSynchronize(r->stream);                                               \
            double _t = em_now();                                                                \
            fprintf(stderr, "[exg]   emit %-18s %.1f ms\n", label, (_t - em_t0) * 1e3);          \
            em_t0 = _t;                                                                          \
        }                                                                                        \
    } while (0)

int arrow_emit(exg_reader *r, const ScanCtx &ctx) {
    StreamState *st = (StreamState *)r->arrow_state.get();
    double em_t0 = em_now();
    st->arena.reset();
    // sized for the typical batch: offsets + values + views of every column; what a batch needs beyond that comes from the
    // pool and enlarges the arena of the next one
    st->arena.min_extra = r->mem_cap ? (64u << 10) : (1u << 20);
    st->arena.prepare(r->device, (size_t)std::min<uint64_t>(r->mem_cap ? r->d_in_cap * 6 + (1u << 20) : r->d_in_cap * 3 + (64u << 20), 6ull << 30));
    EM_TRACE("arena");
    if (!st->copy_stream) {
        EM_HIP(stream_pool()->take(r->device, &st->copy_stream));
        st->copy_dev = r->device;
        EM_HIP(hipEventCreateWithFlags(&st->copy_ev, hipEventDisableTiming));
    }
    auto batch = std::make_shared<ABatch>();
    struct CopyDrain {  // whatever way this function is left, no copy may still be writing into the batch's blocks
        hipStream_t cs;
        ~CopyDrain() { (void)hipStreamSynchronize(cs); }
    } drain{st->copy_stream};
    batch->host.reserve(st->host_hint);
    Emit em;
    em.r = r;
    em.st = st;
    em.host = &batch->host;
    em.s = r->stream;
    em.n = ctx.n_records;
    em.d_row_map = nullptr;
    const uint8_t *d_base = (const uint8_t *)ctx.d_input;
    const uint64_t pb = (uint64_t)(uintptr_t)ctx.h;
    auto str_col = [&](int c) {
        ea::StrCol sc{(const exg_string_t *)r->d_cols[c], d_base, pb};
        if (r->format == EXG_FMT_FASTA && c == 2) {
            sc.d_base = (const uint8_t *)r->d_payload;
            sc.payload_base = (uint64_t)(uintptr_t)ctx.h_seq_payload;
        }
        return sc;
    };
    em.d_err = (unsigned long long *)em.dalloc(sizeof(ea::ErrBlock));  // the error word + the list of literals for the exact float parser
    if (em.rc) return em.rc;
    EM_HIP(hipMemsetAsync(em.d_err, 0xFF, 8, r->stream));
    EM_HIP(hipMemsetAsync(em.d_err + 1, 0, 8, r->stream));

    if (st->has_filter) {
        ea::FilterCols fc;
        memset(&fc, 0, sizeof fc);
        for (size_t c = 0; c < st->schema.size() && c < (size_t)ea::kMaxFilterCols; c++) {
            const std::string &f = st->schema[c].format;
            ea::StrCol sc = str_col((int)c);
            fc.kind[c] = f == "l" ? ea::kColI64 : f == "f" ? ea::kColF32 : ea::kColStr;
            fc.data[c] = f == "l" ? r->d_pos : f == "f" ? r->d_qual : (const void *)sc.d_col;
            fc.d_base[c] = sc.d_base;
            fc.payload_base[c] = sc.payload_base;
            fc.validity[c] = nullptr;
        }
        if (r->format == EXG_FMT_VCF)
            fc.validity[5] = (const uint64_t *)r->d_valid[0];
        else
            fc.validity[1] = (const uint64_t *)r->d_valid[0];
        uint64_t *d_goff = (uint64_t *)em.dalloc((em.n + 1) * 8);
        uint64_t *d_tmp = (uint64_t *)em.dalloc(ea::scan_tmp_entries(em.n) * 8);
        uint32_t *d_map = (uint32_t *)em.dalloc(em.n * 4 + 4);
        if (em.rc) return em.rc;
        ea::FilterCols *d_fc = (ea::FilterCols *)em.dalloc(sizeof fc);
        if (em.rc) return em.rc;
        EM_HIP(hipMemcpyAsync(d_fc, &fc, sizeof fc, hipMemcpyHostToDevice, r->stream));
        EM_HIP(hipStreamSynchronize(r->stream));  // fc is a stack object
        ea::filter_rows((const ea::FilterProgram *)st->d_prog, d_fc, (const uint8_t *)st->d_consts, em.n, d_goff, d_tmp, d_map,
                        r->stream);
        em.n = em.fetch_u64(d_goff + ctx.n_records);
        em.d_row_map = d_map;
        if (em.rc) return em.rc;
    }
    const uint64_t n = em.n;
    if (n == 0) {
        st->produced.reset();
        return EXG_OK;
    }
    std::vector<AColumn> &cols = batch->cols;
    if (r->format == EXG_FMT_FASTQ || r->format == EXG_FMT_FASTA) {
        const int nc = r->format == EXG_FMT_FASTQ ? 4 : 3;
        for (int c = 0; c < nc && !em.rc; c++)
            cols.push_back(em.utf8_top(str_col(c), c == 1 ? (const uint64_t *)r->d_valid[0] : nullptr));
    } else {
        auto prim = [&](const void *d_src, int es, const uint64_t *d_valid_src) {
            AColumn col;
            col.kind = AColumn::kPrim;
            col.elem_size = es;
            col.length = (int64_t)n;
            const void *d = d_src;
            if (em.d_row_map) {
                void *g = em.dalloc(n * es);
                if (em.rc) return col;
                if (es == 8)
                    ea::gather_u64((const uint64_t *)d_src, em.d_row_map, n, (uint64_t *)g, r->stream);
                else
                    ea::gather_u32((const uint32_t *)d_src, em.d_row_map, n, (uint32_t *)g, r->stream);
                d = g;
            }
            col.data = em.to_host(d, n * es);
            col.validity = em.row_validity(d_valid_src);
            return col;
        };
        cols.push_back(em.utf8_top(str_col(0), nullptr));                              // chrom
        cols.push_back(prim(r->d_pos, 8, nullptr));                                     // pos
        cols.push_back(em.list_of_strings(str_col(2), ';'));                            // id
        cols.push_back(em.utf8_top(str_col(3), nullptr));                              // ref
        cols.push_back(em.list_of_strings(str_col(4), ','));                            // alt
        cols.push_back(prim(r->d_qual, 4, (const uint64_t *)r->d_valid[0]));            // qual
        cols.push_back(em.list_of_strings(str_col(6), ';'));                            // filter
        if (em.rc) return em.rc;
        EM_TRACE("flat + lists");
        {  // info
            AColumn info;
            info.kind = AColumn::kStruct;
            info.length = (int64_t)n;
            const uint32_t K = st->info_vt.n;
            if (K) {
                ea::VtCell *d_cells = (ea::VtCell *)em.dalloc((size_t)n * K * sizeof(ea::VtCell));
                if (em.rc) return em.rc;
                ea::info_cells(str_col(7), em.d_row_map, n, st->info_vt, d_cells, r->stream);
                ea::CellSrc src{d_cells, K, 0, str_col(7), em.d_row_map, nullptr, nullptr};
                info.children = em.cell_children(st->info_keys, src, n, EXG_PE_VCF_INFO);
            }
            cols.push_back(std::move(info));
        }
        if (em.rc) return em.rc;
        EM_TRACE("info");
        {  // formats
            AColumn fl;
            fl.kind = AColumn::kList;
            fl.length = (int64_t)n;
            uint64_t *d_goff = (uint64_t *)em.dalloc((n + 1) * 8);
            uint64_t *d_tmp = (uint64_t *)em.dalloc(ea::scan_tmp_entries(n) * 8);
            if (em.rc) return em.rc;
            ea::sample_counts(str_col(8), (const uint64_t *)r->d_valid[1], em.d_row_map, n, d_goff, d_tmp, r->stream);
            const uint64_t S = em.fetch_u64(d_goff + n);
            if (S >= (1ull << 31)) return fail(r, EXG_E_CAPACITY, "formats exceeds Arrow's int32 offsets in one device batch");
            int32_t *d_off32 = (int32_t *)em.dalloc((n + 1) * 4);
            if (em.rc) return em.rc;
            ea::narrow_offsets(d_goff, n, d_off32, r->stream);
            fl.offsets = em.to_host(d_off32, (n + 1) * 4);
            AColumn item;
            item.kind = AColumn::kStruct;
            item.length = (int64_t)S;
            const uint32_t K = st->format_vt.n;
            if (K) {
                ea::VtCell *d_cells = (ea::VtCell *)em.dalloc((size_t)S * K * sizeof(ea::VtCell));
                ea::View *d_fields = (ea::View *)em.dalloc((size_t)S * sizeof(ea::View));
                uint32_t *d_srow = (uint32_t *)em.dalloc((size_t)S * 4);
                if (em.rc) return em.rc;
                ea::sample_cells(str_col(8), em.d_row_map, n, d_goff, st->format_vt, d_cells, d_fields, d_srow, r->stream);
                ea::CellSrc src{d_cells, K, 0, ea::StrCol{nullptr, nullptr, 0}, nullptr, d_fields, d_srow};
                item.children = em.cell_children(st->format_keys, src, S, EXG_PE_VCF_FORMAT);
            }
            fl.children.push_back(std::move(item));
            cols.push_back(std::move(fl));
        }
    }
    if (em.rc) return em.rc;
    EM_TRACE("formats / columns");
    const uint64_t err = em.fetch_u64((const uint64_t *)em.d_err);
    if (em.rc) return em.rc;
    EM_HIP(hipStreamSynchronize(r->stream));
    EM_HIP(hipStreamSynchronize(st->copy_stream));  // every buffer has landed
    uint64_t n_rows = n;
    if (err != ~0ull) {
        // a typed value did not parse: rows before it are delivered, then the error (like the scan's own errors)
        n_rows = err >> 8;
        if (!r->pending_error) {
            r->pending_error = (uint32_t)(err & 0xFF);
            r->pending_error_offset = 0;
        }
    }
    for (AColumn &c : cols)
        if (c.kind == AColumn::kUtf8Top) {
            const uint64_t B = r->batch_rows;
            for (size_t k = 0; k + 1 < c.chunk_base.size(); k++)
                if (c.chunk_base[k + 1] - c.chunk_base[k] >= (1ull << 31) && k * B < n_rows)
                    return fail(r, EXG_E_CAPACITY, "a record batch holds more than 2 GiB of one string column (Arrow Utf8 offsets are int32)");
        }
    if (getenv("EXG_TRACE"))
        fprintf(stderr, "[exg] arrow emit: arena %zu of %zu MiB, %zu extra allocations (%zu MiB), %zu pinned blocks\n",
                st->arena.used >> 20, st->arena.cap >> 20, st->arena.extra.size(), st->arena.extra_bytes >> 20,
                batch->host.blocks.size());
    EM_TRACE("drain");
    st->host_hint = batch->host.total + batch->host.total / 8 + (1u << 20);
    batch->n_rows = n_rows;
    st->produced = n_rows ? batch : nullptr;
    EM_TRACE("swap batch");
    return EXG_OK;
}

// ---- ArrowSchema / ArrowArray export -------------------------------------------------------------------------------------
struct SchemaPriv {
    std::string name, format;
    std::vector<ArrowSchema> kids;
    std::vector<ArrowSchema *> kid_ptrs;
};
void release_schema(ArrowSchema *s) {
    if (!s || !s->release) return;
    for (int64_t i = 0; i < s->n_children; i++)
        if (s->children[i]->release) s->children[i]->release(s->children[i]);
    delete (SchemaPriv *)s->private_data;
    s->release = nullptr;
}
void export_field(const Field &f, ArrowSchema *out) {
    auto *p = new SchemaPriv();
    p->name = f.name;
    p->format = f.format;
    p->kids.resize(f.children.size());
    p->kid_ptrs.resize(f.children.size());
    for (size_t i = 0; i < f.children.size(); i++) {
        export_field(f.children[i], &p->kids[i]);
        p->kid_ptrs[i] = &p->kids[i];
    }
    memset(out, 0, sizeof *out);
    out->format = p->format.c_str();
    out->name = p->name.c_str();
    out->flags = f.nullable ? ARROW_FLAG_NULLABLE : 0;
    out->n_children = (int64_t)f.children.size();
    out->children = p->kid_ptrs.empty() ? nullptr : p->kid_ptrs.data();
    out->release = release_schema;
    out->private_data = p;
}

struct ArrayPriv {
    std::shared_ptr<ABatch> keep;
    const void *buffers[3] = {nullptr, nullptr, nullptr};
    std::vector<ArrowArray> kids;
    std::vector<ArrowArray *> kid_ptrs;
};
void release_array(ArrowArray *a) {
    if (!a || !a->release) return;
    for (int64_t i = 0; i < a->n_children; i++)
        if (a->children[i]->release) a->children[i]->release(a->children[i]);
    delete (ArrayPriv *)a->private_data;
    a->release = nullptr;
}
// rows [row0, row0 + len) of a row-space column (chunk = row0 / batch_rows), or the whole array of a child
void export_column(const AColumn &c, bool row_space, uint64_t row0, uint64_t len, uint64_t chunk, uint64_t batch_rows,
                   const std::shared_ptr<ABatch> &keep, ArrowArray *out) {
    auto *p = new ArrayPriv();
    p->keep = keep;
    memset(out, 0, sizeof *out);
    const uint64_t r0 = row_space ? row0 : 0;
    out->length = (int64_t)(row_space ? len : (uint64_t)c.length);
    out->null_count = c.validity ? -1 : 0;
    out->offset = 0;
    p->buffers[0] = c.validity ? c.validity + r0 / 8 : nullptr;
    switch (c.kind) {
        case AColumn::kUtf8Top:
            out->n_buffers = 3;
            p->buffers[1] = c.offsets + chunk * (batch_rows + 1) * 4;
            p->buffers[2] = c.data + c.chunk_base[chunk];
            break;
        case AColumn::kUtf8Abs:
            out->n_buffers = 3;
            p->buffers[1] = c.offsets + r0 * 4;
            p->buffers[2] = c.data;
            break;
        case AColumn::kPrim:
            out->n_buffers = 2;
            p->buffers[1] = c.data + r0 * c.elem_size;
            break;
        case AColumn::kBool:
            out->n_buffers = 2;
            p->buffers[1] = c.data + r0 / 8;
            break;
        case AColumn::kList:
            out->n_buffers = 2;
            p->buffers[1] = c.offsets + r0 * 4;
            break;
        case AColumn::kStruct:
            out->n_buffers = 1;
            break;
    }
    const bool kids_row_space = row_space && c.kind == AColumn::kStruct;
    p->kids.resize(c.children.size());
    p->kid_ptrs.resize(c.children.size());
    for (size_t i = 0; i < c.children.size(); i++) {
        export_column(c.children[i], kids_row_space, row0, len, chunk, batch_rows, keep, &p->kids[i]);
        p->kid_ptrs[i] = &p->kids[i];
    }
    out->n_children = (int64_t)c.children.size();
    out->children = p->kid_ptrs.empty() ? nullptr : p->kid_ptrs.data();
    out->buffers = p->buffers;
    out->release = release_array;
    out->private_data = p;
}

// ---- the stream -----------------------------------------------------------------------------------------------------------
int stream_get_schema(ArrowArrayStream *s, ArrowSchema *out) {
    StreamState *st = (StreamState *)s->private_data;
    Field root;
    root.format = "+s";
    root.nullable = false;
    root.children = st->schema;
    export_field(root, out);
    return 0;
}

// the next device batch with rows (st->produced), the end of the stream, or an error: everything that touches the reader
static void produce(StreamState *st) {
    exg_reader *r = st->r;
    DeviceGuard guard(r->device);
    st->produced.reset();
    for (;;) {
        if (r->pending_error) {
            st->last_error = std::string(exg_parse_error_string(r->pending_error)) + " in " + r->files[r->file_idx - 1];
            r->pending_error = 0;
            r->file_done = true;
            r->file_idx = r->files.size();
            st->produced_state = 3;
            return;
        }
        if (r->file_done) {
            if (r->finish_source()) {
                st->last_error = r->error;
                st->produced_state = 3;
                return;
            }
            if (r->file_idx >= r->files.size()) {
                st->produced_state = 2;
                return;
            }
            if (open_next_file(r)) {
                st->last_error = r->error;
                st->produced_state = 3;
                return;
            }
        }
        uint64_t k;
        if (next_batch(r, false, &k)) {
            st->last_error = r->error;
            st->produced_state = 3;
            return;
        }
        if (st->produced && st->produced->n_rows) {
            st->produced_state = 1;
            return;
        }
    }
}

int stream_get_next(ArrowArrayStream *s, ArrowArray *out) {
    StreamState *st = (StreamState *)s->private_data;
    exg_reader *r = st->r;
    memset(out, 0, sizeof *out);
    for (;;) {
        if (st->batch && st->batch_row < st->batch->n_rows) {
            // the first record batch of a device batch: the next device batch starts being made
            if (st->batch_row == 0 && !st->fan && !st->producer.joinable() && st->produced_state == 0) st->producer = std::thread(produce, st);
            const uint64_t row0 = st->batch_row, B = r->batch_rows;
            const uint64_t len = std::min<uint64_t>(B, st->batch->n_rows - row0);
            auto *p = new ArrayPriv();
            p->keep = st->batch;
            const size_t nc = st->batch->cols.size();
            p->kids.resize(nc);
            p->kid_ptrs.resize(nc);
            for (size_t c = 0; c < nc; c++) {
                export_column(st->batch->cols[c], true, row0, len, row0 / B, B, st->batch, &p->kids[c]);
                p->kid_ptrs[c] = &p->kids[c];
            }
            out->length = (int64_t)len;
            out->null_count = 0;
            out->n_buffers = 1;
            out->buffers = p->buffers;
            out->n_children = (int64_t)nc;
            out->children = p->kid_ptrs.data();
            out->release = release_array;
            out->private_data = p;
            st->batch_row += len;
            return 0;
        }
        st->batch.reset();
        if (st->fan) {
            if (st->produced_state == 3) return EIO;
            FanItem item;
            std::string msg;
            if (st->fan->next(&item, &msg)) {
                st->last_error = msg;
                st->produced_state = 3;
                return EIO;
            }
            if (!item.batch) return 0;  // out->release == NULL: end of stream
            st->batch = std::static_pointer_cast<ABatch>(item.batch);
            st->batch_row = 0;
            continue;
        }
        if (st->producer.joinable()) st->producer.join();
        if (st->produced_state == 0) produce(st);  // (the first batch of the stream: nothing is under way yet)
        const int state = st->produced_state;
        if (state == 3) return EIO;                // (sticky: the stream is over)
        if (state == 2) return 0;                  // out->release == NULL: end of stream
        st->produced_state = 0;
        st->batch = std::move(st->produced);
        st->batch_row = 0;
    }
}

const char *stream_last_error(ArrowArrayStream *s) {
    StreamState *st = (StreamState *)s->private_data;
    return st->last_error.empty() ? nullptr : st->last_error.c_str();
}

void stream_release(ArrowArrayStream *s) {
    if (!s || !s->release) return;
    StreamState *st = (StreamState *)s->private_data;
    if (st->producer.joinable()) st->producer.join();
    DeviceGuard guard(st->r->device);
    st->batch.reset();
    st->produced.reset();
    std::shared_ptr<void> last = std::move(st->r->arrow_state);
    last.reset();  // deletes st, and the reader with it
    s->release = nullptr;
}

ReaderResult result_error(const std::string &msg) {
    ReaderResult rr;
    rr.error = strdup(msg.c_str());
    return rr;
}

}  // namespace

// ---- the nested VCF columns at the chunk boundary (exg_next_chunk): DuckDB vector layouts built on the device -------
namespace exg_rd {

static exg_type *alloc_types(StreamState *st, size_t n) {
    st->type_nodes.emplace_back(new exg_type[n ? n : 1]());
    return st->type_nodes.back().get();
}
static exg_type leaf(int type, const char *name, int nullable) {
    exg_type t;
    memset(&t, 0, sizeof t);
    t.type = type, t.name = name, t.nullable = nullable;
    return t;
}
static exg_type list_of(StreamState *st, const char *name, const exg_type &item, int nullable) {
    exg_type t = leaf(EXG_TYPE_LIST, name, nullable);
    exg_type *c = alloc_types(st, 1);
    c[0] = item;
    t.n_children = 1;
    t.children = c;
    return t;
}
static exg_type key_type(StreamState *st, const KeyDef &k) {
    const int base = k.type == ea::kVtInt ? EXG_TYPE_INTEGER : k.type == ea::kVtFloat ? EXG_TYPE_FLOAT : k.type == ea::kVtFlag ? EXG_TYPE_BOOLEAN : EXG_TYPE_VARCHAR;
    if (!k.is_list) return leaf(base, k.id.c_str(), 1);
    return list_of(st, k.id.c_str(), leaf(base, "item", 1), 1);
}
static exg_type struct_of(StreamState *st, const char *name, const std::vector<KeyDef> &keys, int nullable) {
    exg_type t = leaf(EXG_TYPE_STRUCT, name, nullable);
    exg_type *c = alloc_types(st, keys.size());
    for (size_t k = 0; k < keys.size(); k++) c[k] = key_type(st, keys[k]);
    t.n_children = (int)keys.size();
    t.children = c;
    return t;
}

int nested_prepare(exg_reader *r) {
    if (r->nested_state || r->format != EXG_FMT_VCF) return EXG_OK;
    if (!r->file) {  // the schema is the first file's header (like register_exon_table, arrow_reader.rs:118-123)
        if (r->file_idx >= r->files.size()) return fail(r, EXG_E_IO, "no input file");
        int rc = open_next_file(r);
        if (rc) return rc;
    }
    auto st = std::make_shared<StreamState>();
    st->r = r;
    st->owns_reader = false;
    parse_vcf_header((const char *)r->file->p, (size_t)r->vcf_header_bytes, &st->info_keys, &st->format_keys);
    int rc;
    if ((rc = upload_keys(r, st->info_keys, &st->info_vt, &st->d_info_names)) ||
        (rc = upload_keys(r, st->format_keys, &st->format_vt, &st->d_format_names)))
        return rc;
    exg_type *t = st->type_roots;
    t[0] = leaf(EXG_TYPE_VARCHAR, "chrom", 0);
    t[1] = leaf(EXG_TYPE_BIGINT, "pos", 0);
    t[2] = list_of(st.get(), "id", leaf(EXG_TYPE_VARCHAR, "item", 1), 1);
    t[3] = leaf(EXG_TYPE_VARCHAR, "ref", 0);
    t[4] = list_of(st.get(), "alt", leaf(EXG_TYPE_VARCHAR, "item", 1), 1);
    t[5] = leaf(EXG_TYPE_FLOAT, "qual", 1);
    t[6] = list_of(st.get(), "filter", leaf(EXG_TYPE_VARCHAR, "item", 1), 1);
    t[7] = struct_of(st.get(), "info", st->info_keys, 1);
    t[8] = list_of(st.get(), "formats", struct_of(st.get(), "item", st->format_keys, 1), 1);
    r->nested_state = st;
    return EXG_OK;
}

void nested_schema(exg_reader *r, exg_schema *out) {
    StreamState *st = (StreamState *)r->nested_state.get();
    if (!st) return;
    for (int c = 0; c < 9; c++) {
        out->tree[c] = &st->type_roots[c];
        out->types[c] = st->type_roots[c].type;
        out->nullable[c] = st->type_roots[c].nullable;
    }
}

// Columns id, alt, filter, info, formats of one scanned batch -> b->nested (the flat columns are copied by next_batch).
// *n_rows: in = rows of the batch, out = rows to hand out (a typed value that does not parse ends the stream there).
int nested_emit(exg_reader *r, const ScanCtx &ctx, Batch *b, const uint32_t *d_row_map, uint64_t *n_rows) {
    StreamState *st = (StreamState *)r->nested_state.get();
    if (!st) return fail(r, EXG_E_INVALID_ARG, "nested_emit without nested_prepare");
    if (getenv("EXG_TRACE"))
        fprintf(stderr, "[exg] nested emit: the batch before used %zu KiB of a %zu KiB arena + %zu extra blocks (%zu KiB)\n", st->arena.used >> 10,
                st->arena.cap >> 10, st->arena.extra.size(), st->arena.extra_bytes >> 10);
    st->arena.reset();
    st->arena.min_extra = r->mem_cap ? (64u << 10) : (1u << 20);
    st->arena.prepare(r->device, (size_t)std::min<uint64_t>(r->mem_cap ? r->d_in_cap * 6 + (1u << 20) : r->d_in_cap * 3 + (64u << 20), 6ull << 30));
    if (!st->copy_stream) {
        EM_HIP(stream_pool()->take(r->device, &st->copy_stream));
        st->copy_dev = r->device;
        EM_HIP(hipEventCreateWithFlags(&st->copy_ev, hipEventDisableTiming));
    }
    struct CopyDrain {
        hipStream_t cs;
        ~CopyDrain() { (void)hipStreamSynchronize(cs); }
    } drain{st->copy_stream};
    const uint64_t n = *n_rows, B = r->batch_rows;
    Emit em;
    em.r = r;
    em.st = st;
    em.host = &b->host;
    em.s = r->stream;
    em.n = n;
    em.d_row_map = d_row_map;
    em.d_err = (unsigned long long *)em.dalloc(sizeof(ea::ErrBlock));  // the error word + the list of literals for the exact float parser
    if (em.rc) return em.rc;
    EM_HIP(hipMemsetAsync(em.d_err, 0xFF, 8, r->stream));
    EM_HIP(hipMemsetAsync(em.d_err + 1, 0, 8, r->stream));
    const uint8_t *d_base = (const uint8_t *)ctx.d_input;
    const uint64_t pb = (uint64_t)(uintptr_t)ctx.h;
    auto str_col = [&](int c) { return ea::StrCol{(const exg_string_t *)r->d_cols[c], d_base, pb}; };
    DuckEmit de{em, B, (n + B - 1) / B, d_base, pb};
    b->nested.assign(9, NVec());
    // (columns outside the projection are built like the others — a malformed value is an error whether or not its column is
    // selected, like in the reference — but em.copy is off for them and their NVec is dropped at the end)
    em.copy = r->want(2);
    b->nested[2] = de.list_of_strings(str_col(2), ';');
    em.copy = r->want(4);
    b->nested[4] = de.list_of_strings(str_col(4), ',');
    em.copy = r->want(6);
    b->nested[6] = de.list_of_strings(str_col(6), ';');
    if (em.rc) return em.rc;
    em.copy = r->want(7);
    {  // info
        NVec info;
        info.type = EXG_TYPE_STRUCT;
        info.length = n;
        const uint32_t K = st->info_vt.n;
        if (K) {
            ea::VtCell *d_cells = (ea::VtCell *)em.dalloc((size_t)n * K * sizeof(ea::VtCell));
            if (em.rc) return em.rc;
            ea::info_cells(str_col(7), d_row_map, n, st->info_vt, d_cells, r->stream);
            ea::CellSrc src{d_cells, K, 0, str_col(7), d_row_map, nullptr, nullptr};
            info.children = de.cell_children(st->info_keys, src, n, EXG_PE_VCF_INFO, nullptr, nullptr, nullptr);
        }
        b->nested[7] = std::move(info);
    }
    if (em.rc) return em.rc;
    em.copy = r->want(8);
    {  // formats
        NVec fl;
        fl.type = EXG_TYPE_LIST;
        fl.elem = 16;
        fl.length = n;
        uint64_t *d_goff = (uint64_t *)em.dalloc((n + 1) * 8);
        uint64_t *d_tmp = (uint64_t *)em.dalloc(ea::scan_tmp_entries(n) * 8);
        ea::ListEntry *d_entries = (ea::ListEntry *)em.dalloc(n * 16 + 16);
        if (em.rc) return em.rc;
        ea::sample_counts(str_col(8), (const uint64_t *)r->d_valid[1], d_row_map, n, d_goff, d_tmp, r->stream);
        const uint64_t S = em.fetch_u64(d_goff + n);
        ea::list_entries_rows(d_goff, n, B, d_entries, r->stream);
        fl.data = em.to_host(d_entries, n * 16);
        uint64_t *d_sb = nullptr;
        fl.child_base = de.bases_rows(d_goff, n, &d_sb);
        NVec item;
        item.type = EXG_TYPE_STRUCT;
        item.length = S;
        const uint32_t K = st->format_vt.n;
        if (K && !em.rc) {
            ea::VtCell *d_cells = (ea::VtCell *)em.dalloc((size_t)S * K * sizeof(ea::VtCell) + 16);
            ea::View *d_fields = (ea::View *)em.dalloc((size_t)S * sizeof(ea::View) + 16);
            uint32_t *d_srow = (uint32_t *)em.dalloc((size_t)S * 4 + 16);
            if (em.rc) return em.rc;
            ea::sample_cells(str_col(8), d_row_map, n, d_goff, st->format_vt, d_cells, d_fields, d_srow, r->stream);
            ea::CellSrc src{d_cells, K, 0, ea::StrCol{nullptr, nullptr, 0}, nullptr, d_fields, d_srow};
            item.children = de.cell_children(st->format_keys, src, S, EXG_PE_VCF_FORMAT, d_srow, d_goff, d_sb);
        }
        fl.children.push_back(std::move(item));
        b->nested[8] = std::move(fl);
    }
    if (em.rc) return em.rc;
    const uint64_t err = em.fetch_u64((const uint64_t *)em.d_err);
    if (em.rc) return em.rc;
    EM_HIP(hipStreamSynchronize(r->stream));
    EM_HIP(hipStreamSynchronize(st->copy_stream));
    for (int c : {2, 4, 6, 7, 8})
        if (!r->want(c)) b->nested[(size_t)c] = NVec();  // (validated, not handed out)
    if (err != ~0ull) {
        // a typed value did not parse: the rows in front of it are handed out, then the error (like the scan's own errors)
        *n_rows = err >> 8;
        if (!r->pending_error) {
            r->pending_error = (uint32_t)(err & 0xFF);
            r->pending_error_offset = 0;
        }
    }
    return EXG_OK;
}

}  // namespace exg_rd

// A reader in Arrow mode + its stream state: schema (VCF: from the first file's header), the `filters` program, the
// emitter.  shard_index / shard_count / device of `oa` make it the reader of one stripe of a fan-out.
static int build_stream(const exg_open_args &oa, const char *filters, std::shared_ptr<StreamState> *out, std::string *err) {
    exg_reader *r = nullptr;
    int rc = exg_open(&oa, &r);
    if (rc) {
        std::string m = exg_last_error_message();
        // arrow_reader.rs:93-102 / :118-123
        if (m.rfind("could not", 0) != 0) m = "could not register table: " + m;
        *err = m;
        return rc;
    }
    DeviceGuard guard(r->device);
    MeterScope meter_scope(&r->meter);
    auto st = std::make_shared<StreamState>();
    st->r = r;
    // The VCF schema needs the first file's header, like register_exon_table (arrow_reader.rs:118-123).  The file
    // is let go again right after: the reference's bind opens a stream only for its schema and never releases
    // it (module.cpp:82-155), so a stream that was not read must not pin a mapping or inflated bytes in HBM.
    if (r->format == EXG_FMT_VCF && open_next_file(r)) {
        *err = "could not register table: " + r->error;
        return EXG_E_IO;
    }
    auto utf8 = [](const char *name, bool nullable) {
        Field f;
        f.name = name;
        f.format = "u";
        f.nullable = nullable;
        return f;
    };
    if (r->format == EXG_FMT_FASTQ) {
        st->schema = {utf8("name", false), utf8("description", true), utf8("sequence", false), utf8("quality_scores", false)};
    } else if (r->format == EXG_FMT_FASTA) {
        st->schema = {utf8("id", false), utf8("description", true), utf8("sequence", false)};
    } else {
        parse_vcf_header((const char *)r->file->p, (size_t)r->vcf_header_bytes, &st->info_keys, &st->format_keys);
        auto list_utf8 = [&](const char *name) {
            Field f;
            f.name = name;
            f.format = "+l";
            Field item = utf8("item", true);
            f.children.push_back(item);
            return f;
        };
        Field pos, qual, info, formats, item;
        pos.name = "pos", pos.format = "l", pos.nullable = false;
        qual.name = "qual", qual.format = "f";
        info.name = "info", info.format = "+s";
        for (auto &k : st->info_keys) info.children.push_back(key_field(k));
        item.name = "item", item.format = "+s";
        for (auto &k : st->format_keys) item.children.push_back(key_field(k));
        formats.name = "formats", formats.format = "+l";
        formats.children.push_back(item);
        st->schema = {utf8("chrom", false), pos, list_utf8("id"), utf8("ref", false), list_utf8("alt"), qual,
                      list_utf8("filter"), info, formats};
        if ((rc = upload_keys(r, st->info_keys, &st->info_vt, &st->d_info_names)) ||
            (rc = upload_keys(r, st->format_keys, &st->format_vt, &st->d_format_names))) {
            *err = "could not register table: " + r->error;
            return rc;
        }
        r->file.reset();
        r->src.reset();  // (the decoder of a compressed file stops: it reads through the descriptor that closes next)
        r->fd_keep.reset();
        r->file_idx = 0;
        r->file_pos = 0;
        r->file_done = true;
    }
    if (filters && *filters) {
        // `SELECT * FROM exon_table WHERE <filters>` (arrow_reader.rs:125-141)
        std::string text = filters;
        std::vector<exg_rd::FilterColumn> fcols;
        for (auto &f : st->schema) fcols.push_back({f.name, f.format == "u" ? 'u' : f.format == "l" ? 'l' : f.format == "f" ? 'f' : 'x'});
        exg_rd::FilterParser fp(text, fcols);
        if (!fp.parse()) {
            *err = "could not execute sql: " + fp.err;
            return EXG_E_INVALID_ARG;
        }
        st->has_filter = true;
        st->prog = fp.prog;
        st->consts = fp.consts;
        if (hipMalloc(&st->d_consts, st->consts.size() + 16) != hipSuccess ||
            hipMalloc(&st->d_prog, sizeof(ea::FilterProgram)) != hipSuccess ||
            hipMemcpy(st->d_prog, &st->prog, sizeof(ea::FilterProgram), hipMemcpyHostToDevice) != hipSuccess ||
            (!st->consts.empty() &&
             hipMemcpy(st->d_consts, st->consts.data(), st->consts.size(), hipMemcpyHostToDevice) != hipSuccess)) {
            *err = "could not execute sql: device allocation failed";
            return EXG_E_HIP;
        }
    }
    r->arrow_emit = arrow_emit;
    // the reader is owned by the stream state from here on
    r->arrow_state = std::shared_ptr<void>(st, st.get());
    *out = st;
    return EXG_OK;
}

extern "C" ReaderResult new_reader(ArrowArrayStream *stream_ptr, const char *uri, uintptr_t batch_size, const char *compression,
                                   const char *file_format, const char *filters) {
    if (!stream_ptr || !uri || !file_format) return result_error("new_reader: null argument");
    exg_open_args oa;
    memset(&oa, 0, sizeof oa);
    oa.path = uri;
    oa.file_format = file_format;
    oa.compression = compression;
    oa.batch_rows = batch_size;
    oa.shard_count = 1;  // (the front reader itself reads the whole input unless it fans out below)
    std::shared_ptr<StreamState> st;
    std::string err;
    if (build_stream(oa, filters, &st, &err)) return result_error(err);
    // The reference's FFI has no shard argument and its glue pulls the stream from one thread (rust.hpp:41-46,
    // module.cpp:36): with several devices the stream fans out by itself — stripes of the input are read by streams of
    // their own, one worker thread and one device each, and their record batches are handed out here in file order.
    {
        std::vector<Stripe> stripes;
        unsigned workers = 1;
        if (plan_stripes(st->r->files, st->r->compression, &oa, &stripes, &workers) == EXG_OK && stripes.size() > st->r->files.size()) {
            struct Sub : FanSub {
                std::shared_ptr<StreamState> st;
                ~Sub() override {
                    if (!st) return;
                    DeviceGuard guard(st->r->device);  // (no MeterScope: the reader, and its meter, go away in here)
                    st->produced.reset();
                    std::shared_ptr<void> last = std::move(st->r->arrow_state);
                    st.reset();
                    last.reset();  // deletes the state, and the reader with it
                }
                int next(FanItem *item, std::string *e) override {
                    MeterScope meter_scope(&st->r->meter);
                    st->produced_state = 0;
                    produce(st.get());
                    if (st->produced_state == 3) {
                        *e = st->last_error;
                        return EXG_E_PARSE;
                    }
                    if (st->produced_state == 1) {
                        item->rows = st->produced->n_rows;
                        item->batch = std::move(st->produced);
                    }
                    return EXG_OK;
                }
                int count(uint64_t *, std::string *e) override {
                    *e = "an Arrow stream is not counted";
                    return EXG_E_UNSUPPORTED;
                }
            };
            const std::string format = file_format, comp = compression ? compression : "", flt = filters ? filters : "";
            const bool has_comp = compression != nullptr;
            const uint64_t rows = batch_size;
            FanOpen open = [=](const Stripe &s, std::unique_ptr<FanSub> *sub, std::string *e) -> int {
                exg_open_args a;
                memset(&a, 0, sizeof a);
                a.path = s.path.c_str();
                a.file_format = format.c_str();
                a.compression = has_comp ? comp.c_str() : nullptr;
                a.batch_rows = rows;
                a.device = s.device;
                a.shard_index = s.shard_index;
                a.shard_count = s.shard_count ? s.shard_count : 1;
                std::unique_ptr<Sub> x(new Sub());
                const int rc = build_stream(a, flt.empty() ? nullptr : flt.c_str(), &x->st, e);
                if (rc) {
                    x->st.reset();
                    return rc;
                }
                *sub = std::move(x);
                return EXG_OK;
            };
            st->fan.reset(new FanOut(std::move(stripes), workers, std::move(open), 2));
        }
    }
    stream_ptr->get_schema = stream_get_schema;
    stream_ptr->get_next = stream_get_next;
    stream_ptr->get_last_error = stream_last_error;
    stream_ptr->release = stream_release;
    stream_ptr->private_data = st.get();
    ReaderResult ok;
    ok.error = nullptr;
    return ok;
}
