// exg_fastq_fused.hip — FASTQ record scan in ONE pass over the input (EXG_ALGO_FUSED).
//
// Replaces, for FASTQ, everything the reference does per batch on one CPU core: noodles-fastq
// 0.8.0 `read_record` (memchr per line, '@' / '+' checks, first-space split), exon 0.2.6
// FASTQArrayBuilder (one memcpy + offset push per field; empty description => NULL), both
// reached through rust/src/arrow_reader.rs:116-153, and DuckDB's ArrowToDuckDB string_t
// construction called at exon/src/exon/arrow_table_function/module.cpp:289.
//
// Design (MI355X / gfx950, wave64, HBM-bound byte work, no MFMA):
//   * one 256-thread workgroup per 48 KiB SUPER-TILE (block b+1 -> super-tile b).  Its bytes are
//     loaded with coalesced 16 B/lane loads (12 per thread, all issued up front) and stay in
//     REGISTERS; the '\n' SWAR match, popcounts and the non-ASCII test run on the registers.
//     The super-tile is then processed as three 16 KiB halves through ONE 17 KiB LDS buffer
//     (bytes are only parked in LDS for the random access that field extraction needs).
//     Why: the global line index arrives ~5 us after a tile's count is published (below), and a
//     CU can hold at most 160 KiB of LDS-staged bytes; keeping the waiting bytes in registers
//     multiplies the bytes in flight per CU (6 workgroups x 48 KiB) at 24.7 KiB of LDS each, which is
//     what hides that latency (A/B in one box: 2 halves 52.1 %, 3 halves 53.7 %, 4 halves 50.6 % of peak);
//   * the 1 KiB that precedes a half is staged too (from memory for the first half, from the
//     previous half's tail for the others), so a record straddling the left edge is resolved from
//     LDS.  A record that begins in front of that window (a long read: every PacBio / ONT file) is
//     not a reason to give the launch up: at most ONE such record ends in a half, its fields are
//     slices — nothing needs its bytes but the '@' / '+' checks, the name's split and the string
//     prefixes — so the half writes the record's newline positions as far as it knows them
//     (FarRec, exg_fastq_ws.hpp) and k_fastq_far, a small kernel behind this one, finds the others
//     by looking back over the tiles' counts and last-four lists and emits the row from global
//     memory.  A half with more lines than its list holds (reads below ~45 bp) is emitted in passes;
//   * newline ranks: packed 4x16-bit wave shuffle scan + wave totals, positions -> u16 list;
//   * the global line index (hence the exact 4-line phase — '@' is also a quality character, so
//     the phase is never guessed — and the output row) comes from an ordered prefix over the
//     super-tile newline counts computed by a CENTRAL SCANNER wave (block 0): workgroups publish
//     A[t], the scanner streams over A[] with the running sum in registers and publishes P[t]; a
//     workgroup polls only its own P[t].  Measured alternatives on MI355X: a ticket counter (one
//     hot atomic word) held the kernel at 0.3 TB/s; a flat decoupled look-back lost 50 % (its
//     dependency chain hops across XCDs at 1-2 us per hop against ~250 tiles/us); a 64-tile group
//     hierarchy lost 80 % (2000 waves polling the same dozen cache lines); reading the input twice
//     (count pass + lagged emit pass) costs 2x: L2 misses are capped near 5.3 TB/s whether HBM or
//     the Infinity Cache serves them.  Descriptor words are their own payload (relaxed
//     agent-scope 8-byte atomics), so no fences are needed;
//   * no dependence on dispatch order: the scanner counts a tile itself when its descriptor is
//     40 us late, and a workgroup whose prefix is 2 ms late sums its predecessors itself;
//   * emission is wave-per-column: wave 0 name, 1 description (+ validity ballot), 2 sequence
//     (+ '+' check), 3 quality; lane = record, so every store instruction writes consecutive
//     16-byte string_t of one column; fields are cut out of LDS dwords with v_alignbyte.
#include "exg_fastq.hpp"
#include "exg_fused_core.hpp"

namespace exg {

struct FastqFormat;
static constexpr int kFastqHalves = 3;  // 48 KiB per workgroup: A/B on one box 2 -> 3 halves +3 %, 4 halves -7 %


// a field that is not UTF-8 (exon's FASTQArrayBuilder: from_utf8 on name, description, sequence, quality): the record's error
// unless a structural one (smaller code) is its
__device__ __forceinline__ void fastq_report_utf8(ScanWsHeader *hdr, long long out, int64_t rec_off) {
    atomicMin(&hdr->err_word, ((unsigned long long)out << 8) | EXG_PE_INVALID_UTF8);
    atomicMin(&hdr->err_off, (unsigned long long)rec_off);
}

// Records that END in the half staged in LDS: wave = column, lane = record.
template <int kMode, class L>
__device__ __forceinline__ void fastq_emit_half(const L &s, const FastqDev &a, ScanWsHeader *hdr, const TileCtx &c,
                                          unsigned long long halo_nl, uint32_t dev_mode, uint32_t lane,
                                          uint32_t wave, unsigned long long *__restrict__ tile_qend,
                                          uint64_t tile_index) {
    const unsigned long long P0 = a.first_line_index - halo_nl;  // line index of d_input[0]
    const uint32_t i0 = (uint32_t)((3 - P0) & 3);                // first quality line of the buffer
    const uint32_t i_first = (uint32_t)((3 - (P0 + c.P)) & 3);   // first quality line of the half
    const unsigned long long q_before = c.P > i0 ? (c.P - i0 + 3) / 4 : 0;  // quality lines before the half
    const unsigned long long n_hc = halo_nl > i0 ? (halo_nl - i0 + 3) / 4 : 0;
    const uint32_t n_rec = c.n_lines > i_first ? (c.n_lines - i_first + 3) / 4 : 0;
    const uint64_t ptr_of_e0 = a.payload_base + c.tile_off - kWin;

    if (threadIdx.x == 0 && (c.pass_base == 0 || n_rec)) {  // offset just past the last quality line that ends in this half (0: none)
        long long e = 0;
        if (n_rec) {
            e = (long long)c.tile_off + (int)s.nlist[4 + i_first + 4 * (n_rec - 1)] - kWin + 1;
            if ((unsigned long long)e > a.n_bytes) e = (long long)a.n_bytes;
        }
        if (c.pass_base) e |= (long long)(tile_qend[tile_index] & kFarBit);  // (a later pass keeps the first pass's mark)
        // (any-shape scan) The first record that ends here begins in front of the LDS window — a long read; only a half's first
        // record in its first pass can —: its row is k_fastq_far's.  What is known here of its five newlines — inside the
        // half, or as codes (prev32) — goes into the half's FarRec; the loop below stores zeros in the row.
        if constexpr (kMode != kLean) if (n_rec && s.nlist[i_first] == kNoneE) {
            const long long out0 = (long long)q_before - (long long)n_hc;
            const int e4 = s.nlist[4 + i_first];
            if ((uint64_t)((int64_t)c.tile_off + e4 - kWin) >= a.lead && out0 >= 0 &&
                ((a.flags & EXG_F_NO_STORE) || (unsigned long long)out0 < a.capacity)) {
                FarRec f;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint32_t idx = i_first + k;
                    f.pos[k] = idx >= 4 ? c.half * kTile + (int32_t)s.nlist[idx] - kWin : s.prev32[idx];
                }
                f.flags = c.is_eof_tile ? 1u : 0u;
                f.out = out0;
                far_rec_of(tile_qend, a.n_bytes)[tile_index] = f;
                hdr->any_far = 1u;
                e |= (long long)kFarBit;
            }
        }
        tile_qend[tile_index] = (unsigned long long)e;
    }
    if (dev_mode >= 3) return;
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;

    for (uint32_t jb = 0; jb < n_rec; jb += 64) {
        const uint32_t j = jb + lane;
        const long long out = (long long)(q_before + j) - (long long)n_hc;
        bool act = j < n_rec;
        int i = 0;
        if (act) {
            i = (int)(i_first + 4 * j);
            int e4 = s.nlist[4 + i];
            act = (uint64_t)((int64_t)c.tile_off + e4 - kWin) >= a.lead && out >= 0;
            if (act && !no_store && (unsigned long long)out >= a.capacity) {
                if (wave == 0) atomicOr(&hdr->flags, EXG_RF_CAPACITY);
                act = false;
            }
        }
        bool desc_valid = false;
        if (act) {
            const uint32_t q0 = s.nlist[i];  // newline before the name line
            uint4 val = make_uint4(0, 0, 0, 0);
            if (q0 == kNoneE) {
                // the record begins in front of the LDS window (val stays zero).  Lean scan: the any-shape run redoes this
                // super-tile; any-shape scan: k_fastq_far's row (the FarRec above)
                if constexpr (kMode == kLean) {
                    if (wave == 0) {
                        tile_redo_of(tile_qend, a.n_bytes)[opaque_s((uint32_t)(tile_index / kFastqHalves))] = kRedoFar;
                        hdr->any_redo = 1u;
                    }
                }
            } else if (wave <= 1) {
                // name line [s0, e0): '@' check, CR strip, split at the first ' '
                int s0 = (int)q0 + 1, e0 = s.nlist[i + 1];
                bool name_ok = s0 < e0 && ldb(s, s0) == '@';
                if (e0 > s0 && !(c.is_eof_tile && e0 == c.lim_e) && ldb(s, e0 - 1) == '\r') e0--;
                int ns = s0 + 1 < e0 ? s0 + 1 : e0;
                int sp = e0;
                for (uint32_t aa = (uint32_t)ns & ~3u; aa < (uint32_t)e0 && sp == e0; aa += 32) {
                    uint32_t w8[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) w8[q] = ldw(s, aa + 4 * q);  // in-bounds: the buffer has slack
                    uint32_t bits = 0;  // one bit per byte of the 32-byte block
#pragma unroll
                    for (int q = 0; q < 8; q++) bits |= nib4(match4(w8[q], 0x20202020u)) << (4 * q);
                    if (aa < (uint32_t)ns) bits &= 0xFFFFFFFFu << ((uint32_t)ns - aa);
                    if (bits) {
                        uint32_t pos = aa + (uint32_t)__ffs(bits) - 1;
                        if (pos < (uint32_t)e0) sp = (int)pos;
                        break;
                    }
                }
                int ds = sp < e0 ? sp + 1 : e0;
                if (wave == 0) {
                    uint32_t code = 0;
                    if (!name_ok)
                        code = EXG_PE_FASTQ_NAME_PREFIX;
                    if (code) {
                        atomicMin(&hdr->err_word, ((unsigned long long)out << 8) | code);
                        atomicMin(&hdr->err_off, (unsigned long long)((int64_t)c.tile_off + s0 - kWin));
                    }
                    val = make_string_lds(s, ns, (uint32_t)(sp - ns), ptr_of_e0);
                    if constexpr (kMode != kLean) {
                        if (c.non_ascii && !utf8_valid_lds(s, ns, sp)) fastq_report_utf8(hdr, out, (int64_t)c.tile_off + s0 - kWin);
                    }
                } else {
                    desc_valid = e0 > ds;
                    if (desc_valid) val = make_string_lds(s, ds, (uint32_t)(e0 - ds), ptr_of_e0);
                    if constexpr (kMode != kLean) {
                        if (c.non_ascii && !utf8_valid_lds(s, ds, e0)) fastq_report_utf8(hdr, out, (int64_t)c.tile_off + s0 - kWin);
                    }
                }
            } else {
                // wave 2: sequence = line 1 (+ '+' check on line 2); wave 3: quality = line 3
                const int k = wave == 2 ? 1 : 3;
                int sk = (int)s.nlist[i + k] + 1, ek = s.nlist[i + k + 1];
                if (sk > ek) sk = ek;  // virtual EOF lines are empty
                if (ek > sk && !(c.is_eof_tile && ek == c.lim_e) && ldb(s, ek - 1) == '\r') ek--;
                uint32_t code = 0;
                if (wave == 2) {
                    int s2 = (int)s.nlist[i + 2] + 1, e2 = s.nlist[i + 3];
                    if (!(s2 < e2 && ldb(s, s2) == '+')) code = EXG_PE_FASTQ_PLUS_PREFIX;
                }
                if (code) {
                    atomicMin(&hdr->err_word, ((unsigned long long)out << 8) | code);
                    atomicMin(&hdr->err_off, (unsigned long long)((int64_t)c.tile_off + (int)q0 + 1 - kWin));
                }
                val = make_string_lds(s, sk, (uint32_t)(ek - sk), ptr_of_e0);
                if constexpr (kMode != kLean) {
                    if (c.non_ascii && !utf8_valid_lds(s, sk, ek)) fastq_report_utf8(hdr, out, (int64_t)c.tile_off + (int)q0 + 1 - kWin);
                }
            }
            // dev_mode 2 keeps the values live but (practically) never stores
            if (!no_store && (dev_mode != 2 || (val.x ^ val.y ^ val.z ^ val.w) == 0x9E3779B9u)) {
                exg_string_t *col = wave == 0 ? a.d_name : wave == 1 ? a.d_desc : wave == 2 ? a.d_seq : a.d_qual;
                st_stream16(reinterpret_cast<uint4 *>(col) + out, val);
            }
        }
        if (wave == 1 && !no_store) {
            // description validity bits of these 64 consecutive records
            unsigned long long b = __ballot(desc_valid);
            if (b) {
                long long out_base = (long long)(q_before + jb) - (long long)n_hc;
                if (out_base < 0) {
                    b >>= (unsigned long long)(-out_base);
                    out_base = 0;
                }
                if (lane == 0 && b) {
                    uint32_t sh = (uint32_t)(out_base & 63);
                    unsigned long long lo = b << sh, hi2 = sh ? b >> (64 - sh) : 0;
                    if (lo) atomicOr((unsigned long long *)&a.d_desc_valid[out_base >> 6], lo);
                    if (hi2) atomicOr((unsigned long long *)&a.d_desc_valid[(out_base >> 6) + 1], hi2);
                }
            }
        }
    }
}


struct FastqFormat {
    using Dev = FastqDev;
#ifndef EXG_FASTQ_NLCAP
#define EXG_FASTQ_NLCAP 512
#endif
    static constexpr int kNlCap = EXG_FASTQ_NLCAP;  // 197 lines per half for 150 bp reads
    static constexpr bool kTabMapLean = false, kTabMapFull = false;
    static constexpr bool kBarriers = true;   // (exg_fused_core.hpp opaque: its lean scan spills without them)
    static constexpr int kHalves = kFastqHalves;
    static constexpr int kHalvesFull = kHalves;
    static constexpr int kMinWavesPerSimd = 6;  // 80 VGPRs, no scratch: 6 x 48 KiB in flight per CU
#ifndef EXG_FASTQ_WAVES_FULL
#define EXG_FASTQ_WAVES_FULL 5
#endif
    static constexpr int kMinWavesPerSimdRedo = 5;
    static constexpr int kMinWavesPerSimdFull = EXG_FASTQ_WAVES_FULL;  // the any-shape instances: 96 VGPRs (their pass loop and FarRec code spill at 80)
    // noodles-fastq at EOF: a record that has its '+' line but no quality line gets an empty one
    __device__ static __forceinline__ uint32_t eof_extra_lines(unsigned long long total_lines) {
        return (total_lines & 3) == 3 ? 1u : 0u;
    }
    template <int kMode, class L>
    __device__ static __forceinline__ void emit_half(const L &s, const FastqDev &a, ScanWsHeader *hdr,
                                                     const TileCtx &c, unsigned long long halo_nl, uint32_t dev_mode,
                                                     uint32_t lane, uint32_t wave,
                                                     unsigned long long *__restrict__ tile_qend, uint64_t tile_index) {
        fastq_emit_half<kMode>(s, a, hdr, c, halo_nl, dev_mode, lane, wave, tile_qend, tile_index);
    }
    __device__ static __forceinline__ unsigned long long analytic_prefix(uint64_t off) {  // FASTQ-150 synthetic
        uint64_t k = off / 332, w = off % 332;
        return 4 * k + (w > 27) + (w > 178) + (w > 180);
    }
};

// The rows k_fused left out: one record per marked half, lines read from global memory (thread = record).  Runs behind
// k_fused on the stream — every tile's count and last-four list is final, the look-back is plain reads.
__global__ __launch_bounds__(256) void k_fastq_far(FastqDev a, const unsigned int *__restrict__ tileA, const int32_t *__restrict__ tileL,
                                                   const unsigned long long *__restrict__ tile_qend, const FarRec *__restrict__ far_rec,
                                                   ScanWsHeader *hdr, uint32_t n_halves) {
    if (!hdr->any_far) return;
    constexpr uint32_t kHalves = FastqFormat::kHalves;
    constexpr uint64_t kSuper = (uint64_t)kHalves * kTile;
    const bool no_store = (a.flags & EXG_F_NO_STORE) != 0;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n_halves; x += (uint64_t)gridDim.x * blockDim.x) {
        if (!(tile_qend[x] & kFarBit)) continue;
        const FarRec f = far_rec[x];
        const uint32_t st = (uint32_t)(x / kHalves);
        int64_t p[5];
        const bool resolved = far_positions<5>(f, st, kSuper, tileA, tileL, (a.flags & EXG_F_BOF) != 0, p);
        const unsigned long long out = (unsigned long long)f.out;
        uint4 z = {0, 0, 0, 0};
        if (!resolved) {  // the record begins in front of d_input[0]: the caller widens the halo (like the general path says it)
            atomicAdd(&hdr->n_unresolved, 1ull);
            atomicOr(&hdr->flags, EXG_RF_HEAD_UNRESOLVED);
            if (!no_store) {
                reinterpret_cast<uint4 *>(a.d_name)[out] = z;
                reinterpret_cast<uint4 *>(a.d_desc)[out] = z;
                reinterpret_cast<uint4 *>(a.d_seq)[out] = z;
                reinterpret_cast<uint4 *>(a.d_qual)[out] = z;
            }
            continue;
        }
        const FastqGeom g = fastq_geometry_at(a.d_in, a.n_bytes, p);
        uint32_t code = 0;
        if (!g.name_ok)
            code = EXG_PE_FASTQ_NAME_PREFIX;
        else if (!g.plus_ok)
            code = EXG_PE_FASTQ_PLUS_PREFIX;
        const uint64_t lens[4] = {g.name_e - g.s[0], g.e[0] - g.desc_s, g.e[1] - g.s[1], g.e[3] - g.s[3]};
        if (!code && (lens[0] > 0xFFFFFFFFull || lens[1] > 0xFFFFFFFFull || lens[2] > 0xFFFFFFFFull || lens[3] > 0xFFFFFFFFull))
            code = EXG_PE_FIELD_TOO_LONG;
        if (!code && tiles_non_ascii(tileA, kSuper, p[0] + 1, p[4]) &&
            !(utf8_valid_global(a.d_in, g.s[0], g.name_e) && utf8_valid_global(a.d_in, g.desc_s, g.e[0]) &&
              utf8_valid_global(a.d_in, g.s[1], g.e[1]) && utf8_valid_global(a.d_in, g.s[3], g.e[3])))
            code = EXG_PE_INVALID_UTF8;
        if (code) {
            atomicMin(&hdr->err_word, (out << 8) | code);
            atomicMin(&hdr->err_off, (unsigned long long)(p[0] + 1));
        }
        if (no_store) continue;
        const bool desc_valid = lens[1] != 0;
        reinterpret_cast<uint4 *>(a.d_name)[out] = make_string_global(a.d_in, g.s[0], lens[0], a.payload_base);
        reinterpret_cast<uint4 *>(a.d_desc)[out] = desc_valid ? make_string_global(a.d_in, g.desc_s, lens[1], a.payload_base) : z;
        reinterpret_cast<uint4 *>(a.d_seq)[out] = make_string_global(a.d_in, g.s[1], lens[2], a.payload_base);
        reinterpret_cast<uint4 *>(a.d_qual)[out] = make_string_global(a.d_in, g.s[3], lens[3], a.payload_base);
        if (desc_valid) atomicOr((unsigned long long *)&a.d_desc_valid[out >> 6], 1ull << (out & 63));
    }
}

// Runs after k_fastq_fused on the same stream: folds the header into the 64-byte result.
__global__ __launch_bounds__(256) void k_fastq_finalize_fused(FastqDev a, ScanWsHeader *hdr,
                                                                const unsigned long long *__restrict__ tile_qend,
                                                                uint32_t n_tiles, exg_scan_result *res) {
    // last tile (searching backwards) in which a quality line ends
    __shared__ unsigned long long s_qend;
    __shared__ int s_found;
    if (threadIdx.x == 0) {
        s_qend = 0;
        s_found = 0;
    }
    __syncthreads();
    if (!hdr->overflow) {
        for (int64_t base = (int64_t)n_tiles - 1; base >= 0; base -= 256) {
            int64_t t = base - threadIdx.x;
            unsigned long long q = t >= 0 ? tile_qend[t] & ~kFarBit : 0;
            if (q) atomicMax(&s_qend, q);
            if (q) s_found = 1;
            __syncthreads();
            if (s_found) break;
        }
    }
    __syncthreads();
    const unsigned long long last_qend = s_qend;
    if (threadIdx.x) return;
    if (hdr->overflow) {  // the general kernels that follow on the stream overwrite this
        exg_scan_result r = {};
        r.flags = EXG_RF_FALLBACK;
        r.error_offset = ~0ull;
        r.error_record = ~0ull;
        *res = r;
        return;
    }
    uint64_t halo_nl = hdr->halo_nl, T = hdr->total_lines;
    uint64_t p0 = a.first_line_index - halo_nl;
    uint64_t i0 = (3 - p0) & 3;
    uint64_t n_cand = T > i0 ? (T - i0 + 3) / 4 : 0;
    uint64_t n_hc = halo_nl > i0 ? (halo_nl - i0 + 3) / 4 : 0;
    uint64_t n_owned = n_cand - (n_hc < n_cand ? n_hc : n_cand);
    unsigned long long err = hdr->err_word, err_off = hdr->err_off;
    uint64_t consumed = last_qend > a.lead ? last_qend : a.lead;
    if ((a.flags & EXG_F_EOF) && ((p0 + T) & 3) != 0 && T > halo_nl) {
        // the truncated record starts where the last complete one ended; the reader checks its '@'
        // before it can run out of lines
        uint64_t start = last_qend;
        uint32_t code = (start < a.n_bytes && a.d_in[start] == '@') ? EXG_PE_UNEXPECTED_EOF : EXG_PE_FASTQ_NAME_PREFIX;
        unsigned long long w = ((unsigned long long)n_owned << 8) | code;
        if (w < err) {
            err = w;
            err_off = start;
        }
    }
    exg_scan_result r;
    r.n_lines = T - halo_nl;
    r.flags = hdr->flags | (hdr->any_redo ? EXG_RF_REDO : 0u);
    r.payload_bytes = 0;
    r.redo_tiles = hdr->n_redo;
    r.error_code = 0;
    r.error_offset = ~0ull;
    r.error_record = ~0ull;
    uint64_t n_rec = (n_owned < a.capacity || (a.flags & EXG_F_NO_STORE)) ? n_owned : a.capacity;
    if (err != kNoError) {
        uint64_t rec = err >> 8;
        r.error_code = (uint32_t)(err & 0xFF);
        r.error_record = rec;
        r.error_offset = err_off;
        if (rec < n_rec) {
            n_rec = rec;
            consumed = err_off > a.lead ? err_off : a.lead;
        }
    }
    r.n_records = n_rec;
    r.consumed_bytes = n_rec ? consumed : a.lead;
    *res = r;
}

int run_fastq_fused(const exg_fastq_scan_args *args, const FastqDev &dev, uint8_t *ws, const FastqWsLayout &l,
                    hipStream_t stream, bool full) {
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    constexpr uint64_t kSuperBytes = (uint64_t)FastqFormat::kHalves * kTile;
    constexpr uint32_t kHalvesHost = FastqFormat::kHalves;
    uint64_t n_super64 = (dev.n_bytes + kSuperBytes - 1) / kSuperBytes;
    if (n_super64 == 0) n_super64 = 1;
    if (n_super64 > 0x7FFFFFF0ull) {
        set_error("exg_fastq_scan: buffer too large for one launch (%llu super-tiles)", (unsigned long long)n_super64);
        return EXG_E_INVALID_ARG;
    }
    uint32_t n_super = (uint32_t)n_super64;
    // descriptor block (exg_fastq_ws.hpp): u64 tileA[n] (u32 counts), u64 tileP[n], u64 tile_redo[n] (u32 marks), u64 tile_qend[n],
    // int32 tileL[n][4], FarRec[n]
    const uint64_t n = l.n_tiles_fused;
    unsigned int *tileA = reinterpret_cast<unsigned int *>(ws + l.off_tile_desc);
    unsigned long long *tileP = reinterpret_cast<unsigned long long *>(ws + l.off_tile_desc) + n;
    unsigned long long *tile_qend = tileP + 2 * n;
    int32_t *tileL = reinterpret_cast<int32_t *>(ws + l.off_tile_last4);
    FarRec *far_rec = reinterpret_cast<FarRec *>(ws + l.off_far);
    hipLaunchKernelGGL(k_init_hdr, dim3(1), dim3(1), 0, stream, hdr, l.lines_cap, 0u);
    EXG_HIP_CHECK(hipMemsetAsync(tileA, 0, (size_t)n * 24, stream));
    if (dev.lead) {
        int rc = exg_count_newlines(dev.d_in, 0, dev.lead, (uint64_t *)&hdr->halo_nl, stream);
        if (rc) return rc;
    }
    if (full) {
        hipLaunchKernelGGL((k_fused<FastqFormat, kFullPrimary>), dim3(n_super + 1), dim3(kThreads), 0, stream, dev, tileA, tileP, tile_qend, hdr, n_super);
    } else {
        hipLaunchKernelGGL((k_fused<FastqFormat, kLean>), dim3(n_super + 1), dim3(kThreads), 0, stream, dev, tileA, tileP, tile_qend, hdr, n_super);
        // the super-tiles the lean scan marked, any shape (returns at once when it marked none)
        hipLaunchKernelGGL((k_fused<FastqFormat, kFullRedo>), dim3(n_super < 1280 ? n_super : 1280), dim3(kThreads), 0, stream, dev, tileA, tileP,
                           tile_qend, hdr, n_super);
    }
    {   // the rows of records that begin in front of their half's window (returns at once when there is none)
        const uint32_t n_halves = n_super * kHalvesHost;
        const uint32_t grid = (n_halves + 255) / 256 < 4096 ? (n_halves + 255) / 256 : 4096;
        hipLaunchKernelGGL(k_fastq_far, dim3(grid), dim3(256), 0, stream, dev, tileA, tileL, tile_qend, far_rec, hdr, n_halves);
    }
    hipLaunchKernelGGL(k_fastq_finalize_fused, dim3(1), dim3(256), 0, stream, dev, hdr, tile_qend, n_super * kHalvesHost,
                       args->d_result);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

}  // namespace exg
