// exg_arrow_stream.cpp — `new_reader`, the reference's own FFI entry point (exon/include/rust.hpp:41-46,
// rust/src/arrow_reader.rs:38-166), on top of the device reader: an ArrowArrayStream whose record batches
// (<= batch_size rows) carry the reference's schema
//     FASTA  id, description, sequence                       (Utf8)
//     FASTQ  name, description, sequence, quality_scores     (Utf8)
//     VCF    chrom Utf8, pos Int64, id List<Utf8>, ref Utf8, alt List<Utf8>, qual Float32,
//            filter List<Utf8>, info Struct<##INFO keys>, formats List<Struct<##FORMAT keys>>
// so the unchanged C++ glue of the reference (WTArrowTableFunction::FileTypeBind / InitGlobal / Scan,
// module.cpp:75-294) can sit on top of it.  Every Arrow buffer is produced on the device (exg_arrow.hip,
// exg_vcf_nested.hip); this file parses the VCF header and the `filters` text, sizes the buffers, copies
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
#include "exg_vcf_nested.hpp"

using namespace exg_rd;
namespace ea = exg::arrow;
namespace vn = exg::vn;

namespace {

// ---- schema -------------------------------------------------------------------------------------------------------------
struct Field {
    std::string name, format;
    bool nullable = true;
    std::vector<Field> children;
};

static_assert(kKeyFlag == vn::kFlag && kKeyInt == vn::kInt && kKeyFloat == vn::kFloat && kKeyString == vn::kString,
              "the header parser's value types are the emitter's");

Field key_field(const KeyDef &k) {
    Field f;
    f.name = k.id;
    const char *fmt = k.type == vn::kInt ? "i" : k.type == vn::kFloat ? "f" : k.type == vn::kFlag ? "b" : "u";
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
    bool owns(const void *p) const {  // (a block of this arena: what stays put until its reset)
        const char *q = (const char *)p;
        if (base && q >= base && q < base + cap) return true;
        for (const auto &e : extra)
            if (q >= (const char *)e.first && q < (const char *)e.first + e.second) return true;
        return false;
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
    // round 6: the batch is handed on while its buffers still travel (an event behind the last copy): whoever hands out its first
    // record batch waits for it — after the batch BEHIND it has begun to be made, whose kernels then run beside these copies
    hipEvent_t landed = nullptr;
    void wait_landed() {
        if (!landed) return;
        (void)hipEventSynchronize(landed);
        (void)hipEventDestroy(landed);
        landed = nullptr;
    }
    ~ABatch() { wait_landed(); }  // (before `host` gives its pinned blocks back)
};

struct StreamState {
    exg_reader *r = nullptr;
    std::vector<Field> schema;
    std::vector<KeyDef> info_keys, format_keys;
    bool has_filter = false;
    ea::FilterProgram prog;
    std::string consts;
    void *d_consts = nullptr, *d_prog = nullptr;
    // the header's keys on the device (exg_vcf_nested.hpp): any number of them, looked up by a hash of their text
    struct NestedKeys {
        vn::KeyTab tab;
        void *d_keys = nullptr, *d_slots = nullptr, *d_names = nullptr;
        NestedKeys() { memset(&tab, 0, sizeof tab); }
    } nk_info, nk_format;
    size_t side_cap = 64u << 10;  // bytes of percent-decoded String values a batch may hold (grows when a batch needs more)
    bool small_rows = false;      // k_rows' row size: the batch before had (nearly) no INFO field of 65 - 128 bytes (exg_vcf_nested.hip)
    // two arenas, taken in turn: a batch's regions are still being copied back while the batch behind it is built (Batch::landed)
    DevArena arena_a, arena_b;
    DevArena *arena_p = &arena_a;
    DevArena &arena() { return *arena_p; }
    hipEvent_t arena_done[2] = {nullptr, nullptr};  // (Arrow mode) behind the last copy out of arena_a / arena_b; arena_busy: recorded, not yet waited for
    bool arena_busy[2] = {false, false};
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
        if (copy_stream) stream_pool()->give_d2h(copy_dev, copy_stream);  // (synchronises it: nothing reads the arenas any more)
        for (hipEvent_t e : arena_done)
            if (e) (void)hipEventDestroy(e);
        for (void *p : {d_consts, d_prog, nk_info.d_keys, nk_info.d_slots, nk_info.d_names, nk_format.d_keys, nk_format.d_slots, nk_format.d_names})
            if (p) (void)hipFree(p);
        arena_a.reset();
        arena_b.reset();
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
    int rc = 0;
    bool copy = true;  // false: the column being built is not in the projection — built and validated on the device, not copied back
    hipStream_t d2h = nullptr;  // the stream build_nested's mirrors travel on (NULL: st->copy_stream)
    bool lazy = false;  // arrow_emit: the copies are not waited for here (ABatch::landed) — a source outside the arena is staged in it first

    void *dalloc(size_t bytes) {
        void *p = st->arena().alloc(bytes);
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
        // (round 6) the big buffers leave by a kernel's 16-byte stores (exg::stream_to_host), not by a copy engine: a text file's
        // next batch is uploaded beside them in slices that each take whichever engine is free — every other batch they queued
        // behind these copies and the upload landed 5 ms late (FASTQ at this boundary: batches of 5.4 and 11 ms in turn).
        // EXG_ARROW_D2H_ENGINE=1: hipMemcpyAsync as before (A/B)
        static const bool by_engine = getenv("EXG_ARROW_D2H_ENGINE") != nullptr;
        // (only beside a text file's uploads: a decoded stream sends up a fraction of the bytes, on its producer's streams, and the
        // engine's 57 GB/s beat the kernel's 50 there — bgzip FASTQ at this boundary 144 ms by kernel, 122 by engine)
        const bool by_kernel = !by_engine && bytes >= (64u << 10) && !r->src;
        void *h = halloc(by_kernel ? (bytes + 15) & ~(size_t)15 : bytes);
        if (h && bytes && lazy && !st->arena().owns(d)) {
            // (the scan's own vectors — POS, QUAL, validity words —: the next batch's scan writes them while this copy may still read)
            void *g = dalloc(bytes + 16);
            if (!g) return (const uint8_t *)h;
            if (hipMemcpyAsync(g, d, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess && !rc) rc = fail(r, EXG_E_HIP, "device copy failed in the Arrow emitter");
            d = g;
        }
        if (h && bytes) {
            hipError_t e = hipEventRecord(st->copy_ev, s);
            if (e == hipSuccess) e = hipStreamWaitEvent(st->copy_stream, st->copy_ev, 0);
            if (e == hipSuccess) {
                if (by_kernel && !(((uintptr_t)h | (uintptr_t)d) & 15)) {
                    if (exg::stream_to_host(h, d, bytes, st->copy_stream) != EXG_OK) e = hipErrorUnknown;
                } else {
                    e = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st->copy_stream);
                }
            }
            if (e != hipSuccess && !rc) rc = fail(r, EXG_E_HIP, "D2H copy failed in the Arrow emitter");
        }
        return (const uint8_t *)h;
    }
    static size_t bitmap_bytes(uint64_t m) { return (size_t)((m + 63) / 64) * 8; }

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

    // a string_t array on the device (the children the nested kernels write) -> Utf8 with absolute int32 offsets
    AColumn utf8_abs(const exg_string_t *d_col, const uint8_t *d_base, uint64_t payload_base, uint64_t m, const uint8_t *h_validity) {
        AColumn col;
        col.kind = AColumn::kUtf8Abs;
        col.length = (int64_t)m;
        col.validity = h_validity;
        uint64_t *d_goff = (uint64_t *)dalloc((m + 1) * 8);
        uint64_t *d_tmp = (uint64_t *)dalloc(ea::scan_tmp_entries(m) * 8);
        if (rc) return col;
        const ea::StrCol sc{d_col, d_base, payload_base};
        ea::utf8_goff_from_col(sc, nullptr, m, d_goff, d_tmp, s);
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
        ea::utf8_copy_from_col(sc, nullptr, m, d_goff, d_values, d_big, big_cap, s);
        ea::narrow_offsets(d_goff, m, d_off32, s);
        col.offsets = to_host(d_off32, (m + 1) * 4);
        col.data = to_host(d_values, total);
        return col;
    }
};

// ---- the nested VCF columns of one device batch (exg_vcf_nested.hpp), for both boundaries ------------------------------------
// Everything is made in DuckDB's layouts — list_entry_t + child vectors, string_t, a byte per BOOLEAN, validity bits — one
// device region per top-level column (id, alt, filter, info, formats), mirrored to pinned memory by one copy when the chunk
// boundary wants the column; the Arrow boundary converts what differs (int32 offsets, Utf8 values, Boolean bits) on the device.
struct NKeyBuf {  // where the children of one key are in the region of its column
    size_t vals = 0, valid = 0, entries = 0, bases = 0, child_vals = 0, child_valid = 0;
    uint64_t total = 0;  // list keys: child elements of the batch
    bool zero = false;   // no element of the key is valid in this batch: its values / entries and validity did not travel (NVec::zero)
};
struct NGroup {
    char *d = nullptr, *h = nullptr;
    size_t bytes = 0;
};
struct NestedOut {
    uint64_t n = 0, S = 0, n_chunks = 0;
    NGroup g[5];  // id, alt, filter, info, formats
    size_t l_entries[3] = {0, 0, 0}, l_bases[3] = {0, 0, 0}, l_elems[3] = {0, 0, 0};
    uint64_t l_total[3] = {0, 0, 0};
    size_t f_entries = 0, f_bases = 0;
    std::vector<NKeyBuf> info, format;
    const uint64_t *d_goff = nullptr;  // list columns over rows (vn::kCol*), n + 1 offsets each
    const uint64_t *d_fgoff = nullptr;  // FORMAT list keys over samples, S + 1 offsets each
    uint64_t err = ~0ull;
    uint64_t side_payload_base = 0;
    bool group_zero[5] = {false, false, false, false, false};  // id / alt / filter: no element at all; formats: no sample at all
    const void *h_zero = nullptr;  // the batch's block of zeros (B x 16 bytes): what NVec::zero nodes point at
};
struct Layout {  // offsets inside a region: what must be zero first, then what the kernels write whole, then what starts as ones
    struct It {
        size_t *slot, bytes;
    };
    std::vector<It> cat[3];
    size_t end[3] = {0, 0, 0};
    void add(int c, size_t *slot, size_t bytes) { cat[c].push_back(It{slot, bytes}); }
    // the byte ranges of the region that travel: everything but the buffers whose slot is in `skip`, neighbours merged
    template <class Skip>
    std::vector<std::pair<size_t, size_t>> ranges(const Skip &skip) const {
        std::vector<std::pair<size_t, size_t>> r;
        for (int c = 0; c < 3; c++)
            for (const It &it : cat[c]) {
                if (skip(it.slot)) continue;
                const size_t len = (it.bytes + 63) & ~(size_t)63;
                if (!r.empty() && r.back().first + r.back().second == *it.slot) r.back().second += len;
                else r.emplace_back(*it.slot, len);
            }
        return r;
    }
    size_t finish() {
        size_t off = 0;
        for (int c = 0; c < 3; c++) {
            for (It &it : cat[c]) {
                *it.slot = off;
                off += (it.bytes + 63) & ~(size_t)63;
            }
            end[c] = off;
        }
        return off;
    }
};
inline size_t key_elem_size(uint8_t type) { return type == vn::kString ? 16 : type == vn::kFlag ? 1 : 4; }

int upload_keys(exg_reader *r, const std::vector<KeyDef> &defs, StreamState::NestedKeys *nk) {
    std::vector<vn::Key> keys(defs.size());
    std::string names;
    uint32_t n_lists = 0;
    for (size_t k = 0; k < defs.size(); k++) {
        if (defs[k].id.size() > 0xFFFFu) return fail(r, EXG_E_UNSUPPORTED, "a VCF header key of more than 65 535 bytes");
        vn::Key &key = keys[k];
        key.name_off = (uint32_t)names.size();
        key.name_len = (uint16_t)defs[k].id.size();
        key.type = defs[k].type;
        key.is_list = defs[k].is_list ? 1 : 0;
        uint32_t h = vn::kKeyHashSeed;
        for (unsigned char c : defs[k].id) h = vn::key_hash_step(h, c);
        key.hash = h;
        key.list_idx = defs[k].is_list ? (int32_t)n_lists++ : -1;
        names += defs[k].id;
    }
    size_t n_slots = 2;
    while (n_slots < 2 * keys.size()) n_slots <<= 1;
    std::vector<uint32_t> slots(n_slots, 0u);
    for (size_t k = 0; k < keys.size(); k++) {  // (the header parser keeps the first definition of an ID: the names are distinct)
        uint32_t sl = vn::key_slot(keys[k].hash, (uint32_t)n_slots - 1);
        while (slots[sl]) sl = (sl + 1) & ((uint32_t)n_slots - 1);
        slots[sl] = (uint32_t)k + 1;
    }
    EM_HIP(hipMalloc(&nk->d_keys, keys.size() * sizeof(vn::Key) + 16));
    EM_HIP(hipMalloc(&nk->d_slots, n_slots * 4));
    EM_HIP(hipMalloc(&nk->d_names, names.size() + 16));
    if (!keys.empty()) EM_HIP(hipMemcpy(nk->d_keys, keys.data(), keys.size() * sizeof(vn::Key), hipMemcpyHostToDevice));
    EM_HIP(hipMemcpy(nk->d_slots, slots.data(), n_slots * 4, hipMemcpyHostToDevice));
    if (!names.empty()) EM_HIP(hipMemcpy(nk->d_names, names.data(), names.size(), hipMemcpyHostToDevice));
    nk->tab.keys = (const vn::Key *)nk->d_keys;
    nk->tab.slots = (const uint32_t *)nk->d_slots;
    nk->tab.names = (const uint8_t *)nk->d_names;
    nk->tab.n_keys = (uint32_t)keys.size();
    nk->tab.slot_mask = (uint32_t)n_slots - 1;
    nk->tab.names_bytes = (uint32_t)names.size();
    nk->tab.n_lists = n_lists;
    return EXG_OK;
}

// want[c]: the chunk boundary hands column c out (its region is mirrored to pinned memory); mirror == false: the Arrow boundary
// (nothing is copied here).  em.n rows (through em.d_row_map), B rows per DataChunk.
int build_nested(Emit &em, const ScanCtx &ctx, uint64_t B, const bool *want, bool mirror, NestedOut *o) {
    exg_reader *r = em.r;
    StreamState *st = em.st;
    hipStream_t s = em.s;
    const uint64_t n = em.n, nc = (n + B - 1) / B;
    const std::vector<KeyDef> &ik = st->info_keys, &fk = st->format_keys;
    const vn::KeyTab &it = st->nk_info.tab, &ft = st->nk_format.tab;
    const uint32_t nil = it.n_lists, nfl = ft.n_lists;
    const uint64_t C = (uint64_t)vn::kColInfo0 + nil;
    const bool rows_info = vn::rows_take_info(it.n_keys);
    o->n = n;
    o->n_chunks = nc;
    const uint8_t *d_base = (const uint8_t *)ctx.d_input;
    const uint64_t pb = (uint64_t)(uintptr_t)ctx.h;
    // rows a wavefront looks at per turn of the wave kernels: one for cohort lines, 64 where most lines are k_rows' own
    const uint64_t avg = ctx.res.n_records ? ctx.res.consumed_bytes / ctx.res.n_records : 64;
    const uint32_t rpg = avg >= 2048 ? 1u : avg >= 256 ? 8u : 64u;

    const bool trace = getenv("EXG_TRACE") != nullptr;
    struct timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    auto since = [&]() {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (ts.tv_sec - ts0.tv_sec) * 1e3 + (ts.tv_nsec - ts0.tv_nsec) * 1e-6;
    };
    double t_stage1 = 0, t_stage2 = 0, t_stage3 = 0;
    vn::Batch b;
    memset(&b, 0, sizeof b);
    const int src_col[5] = {2, 4, 6, 7, 8};
    for (int c = 0; c < 5; c++) b.col[c] = (const exg_string_t *)r->d_cols[src_col[c]];
    b.rest_valid = (const uint64_t *)r->d_valid[1];
    b.row_map = em.d_row_map;
    b.d_base = d_base;
    b.payload_base = pb;
    b.n = n;
    vn::Ctl *d_ctl = (vn::Ctl *)em.dalloc(sizeof(vn::Ctl));
    uint32_t *d_cnt = (uint32_t *)em.dalloc(C * n * 4);
    uint64_t *d_goff = (uint64_t *)em.dalloc(C * (n + 1) * 8);
    uint64_t *d_tmp = (uint64_t *)em.dalloc(vn::scan_tmp_entries(C, n) * 8);
    uint64_t *d_tot = (uint64_t *)em.dalloc((C + nfl + 1) * 8);
    uint64_t *h_tot = (uint64_t *)em.halloc((C + nfl + 1) * 8 + 64);
    const size_t seen_bytes = vn::info_wide_seen_bytes(it, n, rpg);
    uint32_t *d_seen = seen_bytes ? (uint32_t *)em.dalloc(seen_bytes) : nullptr;
    if (em.rc) return em.rc;
    b.ctl = d_ctl;
    b.cnt = d_cnt;
    b.cnt_stride = n;
    b.goff = d_goff;
    b.goff_stride = n + 1;
    EM_HIP(hipMemsetAsync(d_ctl, 0xFF, 8, s));
    EM_HIP(hipMemsetAsync((char *)d_ctl + 8, 0, 16, s));
    if (!rows_info && nil) EM_HIP(hipMemsetAsync(d_cnt + (uint64_t)vn::kColInfo0 * n, 0, (size_t)nil * n * 4, s));  // (k_info_wide writes the keys a row has)
    // ---- stage 1: elements per row of every list column over rows
    b.mid_rows = (unsigned long long *)(d_tot + C + nfl);
    EM_HIP(hipMemsetAsync(b.mid_rows, 0, 8, s));
    vn::rows_count(b, it, st->small_rows, s);
    vn::info_wide_count(b, it, rpg, d_seen, st->small_rows, s);
    vn::samples_count(b, rpg, s);
    vn::scan_counts(d_cnt, n, C, n, d_goff, n + 1, d_tot, d_tmp, s);
    EM_HIP(hipMemcpyAsync(h_tot, d_tot, (C + nfl + 1) * 8, hipMemcpyDeviceToHost, s));
    EM_HIP(hipStreamSynchronize(s));
    // the writing pass's row size: small when at most one INFO field in 64 would be pushed to the wave kernel by it
    const bool small_rows = rows_info && h_tot[C + nfl] * 64 <= n;
    st->small_rows = small_rows;
    const uint64_t S = h_tot[vn::kColSamples];
    o->S = S;
    o->d_goff = d_goff;
    t_stage1 = since();
    // ---- stage 2: elements per sample of the FORMAT keys that are lists
    vn::Samples sm;
    memset(&sm, 0, sizeof sm);
    sm.S = S;
    uint64_t *h_ftot = h_tot + C;
    for (uint32_t k = 0; k < nfl; k++) h_ftot[k] = 0;
    if (S && nfl) {
        uint32_t *d_fcnt = (uint32_t *)em.dalloc((size_t)nfl * S * 4);
        uint64_t *d_fgoff = (uint64_t *)em.dalloc((size_t)nfl * (S + 1) * 8);
        uint64_t *d_tmp2 = (uint64_t *)em.dalloc(vn::scan_tmp_entries(nfl, S) * 8);
        uint32_t *d_srow = (uint32_t *)em.dalloc(S * 4);
        if (em.rc) return em.rc;
        EM_HIP(hipMemsetAsync(d_fcnt, 0, (size_t)nfl * S * 4, s));
        sm.cnt = d_fcnt;
        sm.cnt_stride = S;
        vn::samples_count_lists(b, sm, ft, rpg, s);
        vn::scan_counts(d_fcnt, S, nfl, S, d_fgoff, S + 1, d_tot + C, d_tmp2, s);
        EM_HIP(hipMemcpyAsync(h_ftot, d_tot + C, (size_t)nfl * 8, hipMemcpyDeviceToHost, s));
        EM_HIP(hipStreamSynchronize(s));
        sm.goff = d_fgoff;
        sm.goff_stride = S + 1;
        sm.srow = d_srow;
        o->d_fgoff = d_fgoff;
    }
    t_stage2 = since();
    // ---- stage 3: the children
    Layout L[5];
    for (int c = 0; c < 3; c++) {
        o->l_total[c] = h_tot[c];
        L[c].add(1, &o->l_entries[c], n * 16);
        L[c].add(1, &o->l_bases[c], (nc + 1) * 8);
        L[c].add(1, &o->l_elems[c], h_tot[c] * 16 + 16);
    }
    auto lay_keys = [&](Layout &lay, const std::vector<KeyDef> &defs, const vn::KeyTab &, std::vector<NKeyBuf> *bufs, uint64_t m, const uint64_t *totals,
                        int scalar_cat) {
        bufs->assign(defs.size(), NKeyBuf());
        uint32_t li = 0;
        for (size_t q = 0; q < defs.size(); q++) {
            NKeyBuf &kb = (*bufs)[q];
            const size_t es = key_elem_size(defs[q].type);
            if (!defs[q].is_list) {
                lay.add(scalar_cat, &kb.vals, m * es + 16);
                lay.add(scalar_cat, &kb.valid, Emit::bitmap_bytes(m));
            } else {
                kb.total = totals[li++];
                lay.add(1, &kb.entries, m * 16 + 16);
                lay.add(0, &kb.bases, (nc + 1) * 8);  // (zero first: a batch without samples runs no entries kernel for the FORMAT keys)
                lay.add(scalar_cat, &kb.valid, Emit::bitmap_bytes(m));
                lay.add(1, &kb.child_vals, kb.total * es + 16);
                lay.add(2, &kb.child_valid, Emit::bitmap_bytes(kb.total));
            }
        }
    };
    // (k_rows stores every row's values and validity words of a header it takes; the wave kernels store what a row / sample has)
    lay_keys(L[3], ik, it, &o->info, n, h_tot + vn::kColInfo0, rows_info ? 1 : 0);
    L[4].add(1, &o->f_entries, n * 16);
    L[4].add(1, &o->f_bases, (nc + 1) * 8);
    lay_keys(L[4], fk, ft, &o->format, S, h_ftot, 0);
    for (int attempt = 0;; attempt++) {
        // (a second turn: a batch with more percent-decoded bytes than the side buffer held — everything is made again)
        for (int c = 0; c < 5; c++) {
            NGroup &g = o->g[c];
            g.bytes = L[c].finish();
            g.d = (char *)em.dalloc(g.bytes + 64);
            g.h = mirror && want[c] ? (char *)em.halloc(g.bytes + 64) : nullptr;
            if (em.rc) return em.rc;
            if (L[c].end[0]) EM_HIP(hipMemsetAsync(g.d, 0, L[c].end[0], s));
            if (L[c].end[2] > L[c].end[1]) EM_HIP(hipMemsetAsync(g.d + L[c].end[1], 0xFF, L[c].end[2] - L[c].end[1], s));
        }
        const size_t side_cap = st->side_cap;
        uint8_t *d_side = (uint8_t *)em.dalloc(side_cap + 16);
        char *h_side = mirror ? (char *)em.halloc(side_cap + 16) : nullptr;
        vn::KeyOut *h_ko = (vn::KeyOut *)em.halloc((ik.size() + fk.size() + 1) * sizeof(vn::KeyOut));
        vn::KeyOut *d_ko = (vn::KeyOut *)em.dalloc((ik.size() + fk.size() + 1) * sizeof(vn::KeyOut));
        const size_t n_jobs_max = 4 + nil + nfl;
        vn::EntryJob *h_jobs = (vn::EntryJob *)em.halloc(n_jobs_max * sizeof(vn::EntryJob));
        vn::EntryJob *d_jobs = (vn::EntryJob *)em.dalloc(n_jobs_max * sizeof(vn::EntryJob));
        uint64_t *h_ctl = (uint64_t *)em.halloc(64);
        uint32_t *d_any = (uint32_t *)em.dalloc(ik.size() * 4 + 16);
        uint32_t *h_any = (uint32_t *)em.halloc(ik.size() * 4 + 16);
        if (em.rc) return em.rc;
        EM_HIP(hipMemsetAsync(d_any, 0, ik.size() * 4 + 16, s));
        b.key_any = mirror ? d_any : nullptr;
        auto fill_ko = [&](vn::KeyOut *ko, const std::vector<KeyDef> &defs, const std::vector<NKeyBuf> &bufs, char *base) {
            for (size_t q = 0; q < defs.size(); q++) {
                const NKeyBuf &kb = bufs[q];
                ko[q].vals = defs[q].is_list ? nullptr : base + kb.vals;
                ko[q].valid = (uint64_t *)(base + kb.valid);
                ko[q].child_vals = defs[q].is_list ? base + kb.child_vals : nullptr;
                ko[q].child_valid = defs[q].is_list ? (uint32_t *)(base + kb.child_valid) : nullptr;
            }
        };
        fill_ko(h_ko, ik, o->info, o->g[3].d);
        fill_ko(h_ko + ik.size(), fk, o->format, o->g[4].d);
        EM_HIP(hipMemcpyAsync(d_ko, h_ko, (ik.size() + fk.size()) * sizeof(vn::KeyOut) + 1, hipMemcpyHostToDevice, s));
        for (int c = 0; c < 3; c++) b.elems[c] = (exg_string_t *)(o->g[c].d + o->l_elems[c]);
        b.d_side = d_side;
        b.side_cap = side_cap;
        // (Arrow: the decoded strings become Utf8 values through the text's own (device base, payload base) pair)
        o->side_payload_base = b.side_payload_base = mirror ? (uint64_t)(uintptr_t)h_side : pb + (uint64_t)((const uint8_t *)d_side - d_base);
        vn::rows_write(b, it, d_ko, small_rows, s);
        vn::info_wide_write(b, it, d_ko, rpg, d_seen, small_rows, s);
        vn::samples_write(b, sm, ft, d_ko + ik.size(), rpg, s);
        vn::fix_slow_floats(d_ctl, s);
        // DuckDB's list_entry_t of every list column + where every DataChunk's children begin
        uint32_t nj = 0;
        for (int c = 0; c < 3; c++) h_jobs[nj++] = vn::EntryJob{d_goff + (uint64_t)c * (n + 1), o->g[c].d + o->l_entries[c], (uint64_t *)(o->g[c].d + o->l_bases[c])};
        h_jobs[nj++] = vn::EntryJob{d_goff + (uint64_t)vn::kColSamples * (n + 1), o->g[4].d + o->f_entries, (uint64_t *)(o->g[4].d + o->f_bases)};
        for (size_t q = 0, li = 0; q < ik.size(); q++)
            if (ik[q].is_list) {
                h_jobs[nj++] = vn::EntryJob{d_goff + (vn::kColInfo0 + li) * (n + 1), o->g[3].d + o->info[q].entries, (uint64_t *)(o->g[3].d + o->info[q].bases)};
                li++;
            }
        const uint32_t nj_rows = nj;
        if (sm.goff)
            for (size_t q = 0, li = 0; q < fk.size(); q++)
                if (fk[q].is_list) {
                    h_jobs[nj++] = vn::EntryJob{sm.goff + li * (S + 1), o->g[4].d + o->format[q].entries, (uint64_t *)(o->g[4].d + o->format[q].bases)};
                    li++;
                }
        EM_HIP(hipMemcpyAsync(d_jobs, h_jobs, nj * sizeof(vn::EntryJob), hipMemcpyHostToDevice, s));
        vn::entries_rows(d_jobs, nj_rows, n, B, nc, s);
        if (nj > nj_rows) vn::entries_elems(d_jobs + nj_rows, nj - nj_rows, S, sm.srow, d_goff + (uint64_t)vn::kColSamples * (n + 1), n, B, nc, s);
        EM_HIP(hipMemcpyAsync(h_ctl, d_ctl, 24, hipMemcpyDeviceToHost, s));
        if (mirror && !ik.empty()) EM_HIP(hipMemcpyAsync(h_any, d_any, ik.size() * 4, hipMemcpyDeviceToHost, s));
        EM_HIP(hipStreamSynchronize(s));
        const uint64_t side_used = h_ctl[2];
        if ((h_ctl[1] >> 32) && attempt == 0) {  // side_overflow
            st->side_cap = (size_t)(side_used + side_used / 4 + 4096);
            EM_HIP(hipMemsetAsync(d_ctl, 0xFF, 8, s));
            EM_HIP(hipMemsetAsync((char *)d_ctl + 8, 0, 16, s));
            continue;
        }
        o->err = h_ctl[0];
        t_stage3 = since();
        trace_at("N nested kernels done", n);
        r->nested_ns += (uint64_t)(t_stage3 * 1e6);

        if (trace) {
            size_t bytes = 0;
            for (int c = 0; c < 5; c++) bytes += o->g[c].h ? o->g[c].bytes : 0;
            fprintf(stderr, "[exg]   nested: counts %.2f ms, sample lists %.2f ms, children %.2f ms (%llu rows, %llu samples, %.1f MiB to the host)\n", t_stage1,
                    t_stage2 - t_stage1, t_stage3 - t_stage2, (unsigned long long)n, (unsigned long long)S, bytes / 1048576.0);
        }
        if (mirror) {
            // What is all zeros stays at home (round 6): a list column without a single element in the batch (`formats` of a file without
            // samples, `id` of a file of "."s), an INFO key that no row of the batch has — 16 bytes a row each that said nothing; their
            // NVec nodes point every chunk at one block of zeros instead (VCF-8: 32 of a line's 171 bytes)
            static const bool no_zero_skip = getenv("EXG_VCF_NO_ZERO_SKIP") != nullptr;
            std::vector<const size_t *> skip;
            if (!no_zero_skip) {
                void *hz = em.halloc(B * 16 + 64);
                if (em.rc) return em.rc;
                memset(hz, 0, B * 16 + 64);
                o->h_zero = hz;
                for (int c = 0; c < 3; c++) o->group_zero[c] = o->l_total[c] == 0;
                o->group_zero[4] = S == 0;
                for (size_t q = 0; q < ik.size(); q++) {
                    NKeyBuf &kb = o->info[q];
                    kb.zero = ik[q].is_list ? kb.total == 0 : h_any[q] == 0;
                    if (!kb.zero) continue;
                    if (ik[q].is_list) skip.push_back(&kb.entries), skip.push_back(&kb.child_vals), skip.push_back(&kb.child_valid);
                    else skip.push_back(&kb.vals);
                    skip.push_back(&kb.valid);
                }
            }
            const hipStream_t cs = em.d2h ? em.d2h : st->copy_stream;
            for (int c = 0; c < 5; c++) {
                if (!o->g[c].h || !o->g[c].bytes || o->group_zero[c]) continue;
                for (const auto &rg : L[c].ranges([&](const size_t *slot) { return std::find(skip.begin(), skip.end(), slot) != skip.end(); })) {
                    EM_HIP(hipMemcpyAsync(o->g[c].h + rg.first, o->g[c].d + rg.first, rg.second, hipMemcpyDeviceToHost, cs));
                    r->host_vector_bytes += rg.second;
                }
            }
            if (side_used) EM_HIP(hipMemcpyAsync(h_side, d_side, (size_t)std::min<uint64_t>(side_used, side_cap), hipMemcpyDeviceToHost, em.d2h ? em.d2h : st->copy_stream));
        }
        break;
    }
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
    // Round 6: the batch leaves this function while its buffers still cross PCIe (ABatch::landed), the batch behind it is scanned and
    // emitted beside them out of the OTHER arena — an arena is reset only when the last copy out of it has landed (arena_done).  A
    // text file: FASTQ at this boundary 111 -> 101 ms (A/B in one box).  NOT a decoded stream: its decoder's kernels run ahead on the
    // producer's streams all the time, and the emitter's kernels beside the copies AND the decoder measured slower (bgzip VCF 72 -> 86 ms,
    // bgzip FASTQ 118 -> 127).  Not under a memory cap (one arena).  EXG_ARROW_EAGER_LANDING=1: as before (A/B)
    static const bool eager_landing = getenv("EXG_ARROW_EAGER_LANDING") != nullptr;
    const bool lazy = !eager_landing && !r->mem_cap && !r->src;
    int arena_idx = 0;
    if (lazy) {
        st->arena_p = st->arena_p == &st->arena_a ? &st->arena_b : &st->arena_a;
        arena_idx = st->arena_p == &st->arena_a ? 0 : 1;
        if (st->arena_busy[arena_idx]) {
            EM_HIP(hipEventSynchronize(st->arena_done[arena_idx]));
            st->arena_busy[arena_idx] = false;
        }
    }
    st->arena().reset();
    // sized for the typical batch: offsets + values + views of every column; what a batch needs beyond that comes from the
    // pool and enlarges the arena of the next one
    st->arena().min_extra = r->mem_cap ? (64u << 10) : (1u << 20);
    st->arena().prepare(r->device, (size_t)std::min<uint64_t>(r->mem_cap ? r->d_in_cap * 6 + (1u << 20) : r->d_in_cap * 3 + (64u << 20), 6ull << 30));
    EM_TRACE("arena");
    if (!st->copy_stream) {
        EM_HIP(stream_pool()->take_d2h(r->device, &st->copy_stream, /*calibrate=*/r->file && r->file->n >= (512ull << 20) && !r->mem_cap));
        st->copy_dev = r->device;
        EM_HIP(hipEventCreateWithFlags(&st->copy_ev, hipEventDisableTiming));
    }
    auto batch = std::make_shared<ABatch>();
    struct CopyDrain {  // whatever way this function is left, no copy may still be writing into the batch's blocks
        hipStream_t cs;
        ~CopyDrain() {
            if (cs) (void)hipStreamSynchronize(cs);
        }
    } drain{st->copy_stream};
    batch->host.reserve(st->host_hint);
    Emit em;
    em.lazy = lazy;
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
    uint64_t err = ~0ull;  // (row << 8) | code of the first typed VCF value that does not parse

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
        AColumn c_ref = em.utf8_top(str_col(3), nullptr);
        AColumn c_qual = prim(r->d_qual, 4, (const uint64_t *)r->d_valid[0]);
        if (em.rc) return em.rc;
        EM_TRACE("flat");
        // id / alt / filter / info / formats: made on the device in DuckDB's layouts (exg_vcf_nested.hpp), converted here to Arrow's
        // (absolute int32 list offsets, Utf8 values closed up, Boolean bits)
        const bool want_all[5] = {true, true, true, true, true};
        NestedOut no;
        if (int nrc = build_nested(em, ctx, r->batch_rows, want_all, false, &no)) return nrc;
        err = no.err;
        EM_TRACE("nested kernels");
        auto off32_of = [&](const uint64_t *d_goff, uint64_t m) -> const uint8_t * {
            if (!d_goff || !m) {  // (no element at all: one zero offset per entry)
                void *h = em.halloc((m + 1) * 4);
                if (h) memset(h, 0, (m + 1) * 4);
                return (const uint8_t *)h;
            }
            int32_t *d = (int32_t *)em.dalloc((m + 1) * 4);
            if (em.rc) return nullptr;
            ea::narrow_offsets(d_goff, m, d, r->stream);
            return em.to_host(d, (m + 1) * 4);
        };
        auto strings = [&](const char *d, uint64_t m, const uint8_t *h_valid) { return em.utf8_abs((const exg_string_t *)d, d_base, pb, m, h_valid); };
        auto too_long = [&](uint64_t total) {
            if (total >= (1ull << 31) && !em.rc) em.rc = fail(r, EXG_E_CAPACITY, "a list column exceeds Arrow's int32 offsets in one device batch");
            return em.rc != 0;
        };
        auto list_utf8 = [&](int c) {
            AColumn col;
            col.kind = AColumn::kList;
            col.length = (int64_t)n;
            if (too_long(no.l_total[c])) return col;
            col.offsets = off32_of(no.d_goff + (uint64_t)c * (n + 1), n);
            col.children.push_back(strings(no.g[c].d + no.l_elems[c], no.l_total[c], nullptr));
            return col;
        };
        auto key_cols = [&](const std::vector<KeyDef> &defs, const std::vector<NKeyBuf> &bufs, const char *base, uint64_t m, const uint64_t *goffs) {
            std::vector<AColumn> kids;
            uint64_t li = 0;
            for (size_t q = 0; q < defs.size() && !em.rc; q++) {
                const NKeyBuf &kb = bufs[q];
                const KeyDef &kd = defs[q];
                AColumn col;
                col.length = (int64_t)m;
                const uint8_t *valid = em.to_host(base + kb.valid, Emit::bitmap_bytes(m));
                if (!kd.is_list) {
                    if (kd.type == vn::kInt || kd.type == vn::kFloat) {
                        col.kind = AColumn::kPrim;
                        col.elem_size = 4;
                        col.data = em.to_host(base + kb.vals, m * 4);
                        col.validity = valid;
                    } else if (kd.type == vn::kFlag) {
                        uint64_t *d_bits = (uint64_t *)em.dalloc(Emit::bitmap_bytes(m));
                        if (em.rc) break;
                        vn::bytes_to_bits((const uint8_t *)base + kb.vals, m, d_bits, r->stream);
                        col.kind = AColumn::kBool;
                        col.data = em.to_host(d_bits, Emit::bitmap_bytes(m));
                        col.validity = valid;
                    } else {
                        col = strings(base + kb.vals, m, valid);
                    }
                } else {
                    if (too_long(kb.total)) break;
                    col.kind = AColumn::kList;
                    col.offsets = off32_of(goffs ? goffs + li * (m + 1) : nullptr, m);
                    col.validity = valid;
                    const uint8_t *cv = em.to_host(base + kb.child_valid, Emit::bitmap_bytes(kb.total));
                    AColumn child;
                    child.length = (int64_t)kb.total;
                    if (kd.type == vn::kInt || kd.type == vn::kFloat) {
                        child.kind = AColumn::kPrim;
                        child.elem_size = 4;
                        child.data = em.to_host(base + kb.child_vals, kb.total * 4);
                        child.validity = cv;
                    } else {
                        child = strings(base + kb.child_vals, kb.total, cv);
                    }
                    col.children.push_back(std::move(child));
                    li++;
                }
                kids.push_back(std::move(col));
            }
            return kids;
        };
        cols.push_back(list_utf8(0));   // id
        cols.push_back(std::move(c_ref));
        cols.push_back(list_utf8(1));   // alt
        cols.push_back(std::move(c_qual));
        cols.push_back(list_utf8(2));   // filter
        {
            AColumn info;
            info.kind = AColumn::kStruct;
            info.length = (int64_t)n;
            info.children = key_cols(st->info_keys, no.info, no.g[3].d, n, no.d_goff + (uint64_t)vn::kColInfo0 * (n + 1));
            cols.push_back(std::move(info));
        }
        {
            AColumn fl;
            fl.kind = AColumn::kList;
            fl.length = (int64_t)n;
            if (too_long(no.S)) return em.rc;
            fl.offsets = off32_of(no.d_goff + (uint64_t)vn::kColSamples * (n + 1), n);
            AColumn item;
            item.kind = AColumn::kStruct;
            item.length = (int64_t)no.S;
            item.children = key_cols(st->format_keys, no.format, no.g[4].d, no.S, no.d_fgoff);
            fl.children.push_back(std::move(item));
            cols.push_back(std::move(fl));
        }
    }
    if (em.rc) return em.rc;
    EM_TRACE("formats / columns");
    EM_HIP(hipStreamSynchronize(r->stream));
    if (lazy) {
        if (!st->arena_done[arena_idx]) EM_HIP(hipEventCreateWithFlags(&st->arena_done[arena_idx], hipEventDisableTiming));
        EM_HIP(hipEventCreateWithFlags(&batch->landed, hipEventDisableTiming));
        EM_HIP(hipEventRecord(batch->landed, st->copy_stream));
        EM_HIP(hipEventRecord(st->arena_done[arena_idx], st->copy_stream));
        st->arena_busy[arena_idx] = true;
    } else {
        EM_HIP(hipStreamSynchronize(st->copy_stream));  // every buffer has landed
    }
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
                st->arena().used >> 20, st->arena().cap >> 20, st->arena().extra.size(), st->arena().extra_bytes >> 20,
                batch->host.blocks.size());
    EM_TRACE("drain");
    st->host_hint = batch->host.total + batch->host.total / 8 + (1u << 20);
    batch->n_rows = n_rows;
    st->produced = n_rows ? batch : nullptr;
    EM_TRACE("swap batch");
    if (lazy) drain.cs = nullptr;  // (the batch carries its landing event; a batch that is dropped waits for it itself: ~ABatch)
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
            if (st->batch_row == 0) st->batch->wait_landed();  // (behind the start of the next batch: its kernels run beside these copies)
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
    const int base = k.type == vn::kInt ? EXG_TYPE_INTEGER : k.type == vn::kFloat ? EXG_TYPE_FLOAT : k.type == vn::kFlag ? EXG_TYPE_BOOLEAN : EXG_TYPE_VARCHAR;
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
    if ((rc = upload_keys(r, st->info_keys, &st->nk_info)) || (rc = upload_keys(r, st->format_keys, &st->nk_format))) return rc;
    // A wide header makes wide rows of vectors whatever the lines hold — every key is a child with a value slot per row (per sample)
    // and, on the device, a count and an offset per row for each list key: a 1 000-key header over 256 MiB of SHORT lines would ask
    // for tens of GB of both.  The device batch shrinks with the number of keys (64 keys: as before).
    {
        const uint64_t keys = st->info_keys.size() + st->format_keys.size();
        if (keys > 64) r->device_batch_bytes = std::max<uint64_t>(8ull << 20, (r->device_batch_bytes * 64 / keys) & ~0xFFFFFull);
    }
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
        fprintf(stderr, "[exg] nested emit: the batch before used %zu KiB of a %zu KiB arena + %zu extra blocks (%zu KiB)\n", st->arena().used >> 10,
                st->arena().cap >> 10, st->arena().extra.size(), st->arena().extra_bytes >> 10);
    // (the other arena: the one of the batch before may still be read by its copies — whose end the consumer waits for before this
    // function is entered a second time after it)
    st->arena_p = st->arena_p == &st->arena_a ? &st->arena_b : &st->arena_a;
    st->arena().reset();
    st->arena().min_extra = r->mem_cap ? (64u << 10) : (1u << 20);
    st->arena().prepare(r->device, (size_t)std::min<uint64_t>(r->mem_cap ? r->d_in_cap * 6 + (1u << 20) : r->d_in_cap * 3 + (64u << 20), 6ull << 30));
    if (!st->copy_stream) {
        EM_HIP(stream_pool()->take_d2h(r->device, &st->copy_stream, /*calibrate=*/r->file && r->file->n >= (512ull << 20) && !r->mem_cap));
        st->copy_dev = r->device;
        EM_HIP(hipEventCreateWithFlags(&st->copy_ev, hipEventDisableTiming));
    }
    struct CopyDrain {
        hipStream_t cs;
        ~CopyDrain() { (void)hipStreamSynchronize(cs); }
    } drain{st->copy_stream};
    const uint64_t n = *n_rows;
    Emit em;
    em.r = r;
    em.st = st;
    em.host = &b->host;
    em.s = r->stream;
    em.n = n;
    em.d_row_map = d_row_map;
    // (columns outside the projection are built like the others — a malformed value is an error whether or not its column is
    // selected, like in the reference — but their regions stay on the device)
    const int top[5] = {2, 4, 6, 7, 8};
    bool want[5];
    for (int c = 0; c < 5; c++) want[c] = r->want(top[c]);
    // Every byte that goes back to the host travels on ONE stream — behind the flat columns' copies when next_batch gave them a
    // stream of their own: a copy occupies an SDMA engine, a second stream of D2H copies takes a second engine, and that was the
    // one the next batch's upload runs on (EXG_TRACE=2: the upload landed 2 ms after the columns had left, the link never duplex)
    em.d2h = r->col_stream && !getenv("EXG_VCF_TWO_D2H") ? r->col_stream : st->copy_stream;
    struct D2hDrain {  // (an error return: nothing may still be writing the batch's blocks)
        hipStream_t cs;
        ~D2hDrain() {
            if (cs) (void)hipStreamSynchronize(cs);
        }
    } drain2{em.d2h};
    NestedOut no;
    if (int rc = build_nested(em, ctx, r->batch_rows, want, true, &no)) return rc;
    if (r->lazy_landing && em.d2h == r->col_stream)
        drain2.cs = nullptr;  // the caller records the batch's landing event behind these copies (Batch::landed)
    else
        EM_HIP(hipStreamSynchronize(em.d2h));
    b->nested.assign(9, NVec());
    auto leaf = [](int type, uint32_t elem, uint64_t length, const void *data, const void *validity) {
        NVec v;
        v.type = type;
        v.elem = elem;
        v.length = length;
        v.data = data;
        v.validity = (const uint64_t *)validity;
        return v;
    };
    auto duck_type = [](uint8_t t) { return t == vn::kInt ? EXG_TYPE_INTEGER : t == vn::kFloat ? EXG_TYPE_FLOAT : t == vn::kFlag ? EXG_TYPE_BOOLEAN : EXG_TYPE_VARCHAR; };
    // (a node whose buffers stayed at home — NKeyBuf::zero, NestedOut::group_zero — points at the batch's block of zeros)
    const char *hz = (const char *)no.h_zero;
    // (all chunks' children begin at 0 when a list column has no element at all)
    const uint64_t *zero_bases = nullptr;
    if (hz && (no.group_zero[0] || no.group_zero[1] || no.group_zero[2] || no.group_zero[4])) {
        void *zb = em.halloc((no.n_chunks + 1) * 8);
        if (em.rc) return em.rc;
        memset(zb, 0, (no.n_chunks + 1) * 8);
        zero_bases = (const uint64_t *)zb;
    }
    auto key_vecs = [&](const std::vector<KeyDef> &defs, const std::vector<NKeyBuf> &bufs, const char *h, uint64_t m, bool all_zero) {
        std::vector<NVec> kids;
        for (size_t q = 0; q < defs.size(); q++) {
            const NKeyBuf &kb = bufs[q];
            const uint32_t es = (uint32_t)key_elem_size(defs[q].type);
            const bool z = (kb.zero || all_zero) && hz;
            if (!defs[q].is_list) {
                kids.push_back(leaf(duck_type(defs[q].type), es, m, z ? hz : h + kb.vals, z ? hz : h + kb.valid));
                kids.back().zero = z;
            } else {
                NVec v = leaf(EXG_TYPE_LIST, 16, m, z ? hz : h + kb.entries, z ? hz : h + kb.valid);
                v.zero = z;
                v.child_base = all_zero ? zero_bases : (const uint64_t *)(h + kb.bases);
                v.children.push_back(leaf(duck_type(defs[q].type), es, kb.total, z ? hz : h + kb.child_vals, z ? hz : h + kb.child_valid));
                v.children.back().zero = z;
                kids.push_back(std::move(v));
            }
        }
        return kids;
    };
    for (int c = 0; c < 3; c++) {
        if (!want[c]) continue;
        const char *h = no.g[c].h;
        const bool z = no.group_zero[c] && hz;
        NVec v = leaf(EXG_TYPE_LIST, 16, n, z ? hz : h + no.l_entries[c], nullptr);
        v.zero = z;
        v.child_base = z ? zero_bases : (const uint64_t *)(h + no.l_bases[c]);
        v.children.push_back(leaf(EXG_TYPE_VARCHAR, 16, no.l_total[c], z ? hz : h + no.l_elems[c], nullptr));  // (id / alt / filter are not percent-decoded)
        v.children.back().zero = z;
        b->nested[(size_t)top[c]] = std::move(v);
    }
    if (want[3]) {
        NVec info = leaf(EXG_TYPE_STRUCT, 0, n, nullptr, nullptr);
        info.children = key_vecs(st->info_keys, no.info, no.g[3].h, n, false);
        b->nested[7] = std::move(info);
    }
    if (want[4]) {
        const char *h = no.g[4].h;
        const bool z = no.group_zero[4] && hz;  // (no line of the batch has a sample: every entry {0, 0}, every child empty)
        NVec fl = leaf(EXG_TYPE_LIST, 16, n, z ? hz : h + no.f_entries, nullptr);
        fl.zero = z;
        fl.child_base = z ? zero_bases : (const uint64_t *)(h + no.f_bases);
        NVec item = leaf(EXG_TYPE_STRUCT, 0, no.S, nullptr, nullptr);
        item.children = key_vecs(st->format_keys, no.format, h, no.S, z);
        fl.children.push_back(std::move(item));
        b->nested[8] = std::move(fl);
    }
    if (no.err != ~0ull) {
        // a typed value did not parse: the rows in front of it are handed out, then the error (like the scan's own errors)
        *n_rows = no.err >> 8;
        if (!r->pending_error) {
            r->pending_error = (uint32_t)(no.err & 0xFF);
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
        if ((rc = upload_keys(r, st->info_keys, &st->nk_info)) || (rc = upload_keys(r, st->format_keys, &st->nk_format))) {
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
