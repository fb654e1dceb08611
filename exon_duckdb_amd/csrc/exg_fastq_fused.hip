// exg_fastq_fused.hip — FASTQ record scan in ONE pass over the input (EXG_ALGO_FUSED).
//
// Replaces, for FASTQ, everything the reference does per batch on one CPU core: noodles-fastq
// 0.8.0 `read_record` (memchr per line, '@' / '+' checks, first-space split), exon 0.2.6
// FASTQArrayBuilder (one memcpy + offset push per field; empty description => NULL), both
// reached through rust/src/arrow_reader.rs:116-153, and DuckDB's ArrowToDuckDB string_t
// construction called at exon/src/exon/arrow_table_function/module.cpp:289.
//
// Design (MI355X / gfx950, wave64, HBM-bound byte work, no MFMA):
//   * one 256-thread workgroup per 32 KiB SUPER-TILE (block b+1 -> super-tile b).  Its bytes are
//     loaded with coalesced 16 B/lane loads (8 per thread, all issued up front) and stay in
//     REGISTERS; the '\n' SWAR match, popcounts and the non-ASCII test run on the registers.
//     The super-tile is then processed as two 16 KiB halves through ONE 17 KiB LDS buffer
//     (bytes are only parked in LDS for the random access that field extraction needs).
//     Why: the global line index arrives ~5 us after a tile's count is published (below), and a
//     CU can hold at most 160 KiB of LDS-staged bytes; keeping the waiting bytes in registers
//     doubles the bytes in flight per CU (8 workgroups x 32 KiB) at 20 KiB of LDS each, which is
//     what hides that latency;
//   * the 1 KiB that precedes a half is staged too (from memory for the first half, from the
//     first half's tail for the second), so a record straddling the left edge is resolved from
//     LDS; a record larger than that window raises `overflow` and the general multipass kernels
//     redo the buffer;
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

namespace exg {

static constexpr int kTile = kFusedTileBytes;  // 16384: one half, the unit of LDS staging and of tile_qend
static constexpr int kHalves = 2;
static constexpr int kSuper = kTile * kHalves;  // 32768 bytes per workgroup
static constexpr int kWin = kFusedWindow;       // 1024
static constexpr int kThreads = 256;
static constexpr int kRows = kTile / (kThreads * 16);  // 4 chunk rows (4 KiB each) per half
static constexpr int kLdsBytes = kWin + kTile + 96;
static constexpr int kNlCap = 1024;  // newline positions kept per half
static constexpr uint32_t kNoneE = 0xFFFFu;

static constexpr unsigned long long kFlag = 1ull << 63;  // descriptor word is published
static constexpr unsigned long long kVal = (1ull << 48) - 1;

struct FusedLds {
    uint8_t bytes[kLdsBytes];        // [0,kWin) window, then the half; e = p + kWin
    uint16_t nlist[4 + kNlCap + 4];  // e-offsets of newlines: [0..3] the 4 before the half (oldest first)
    uint32_t wtot[4];   // per-wave newline counts of the staged half
    uint32_t wcnt[4];   // per-wave packed (half 0 | half 1 << 16) newline counts
    unsigned long long prefix;            // '\n' in the buffer before this super-tile
    uint32_t hi_or[4];
    uint16_t carry[4];  // the 4 newlines before the second half, relative to it
};

__device__ __forceinline__ uint32_t ldw(const FusedLds &s, uint32_t e_aligned) {
    return *reinterpret_cast<const uint32_t *>(s.bytes + e_aligned);
}
__device__ __forceinline__ uint32_t ldb(const FusedLds &s, int e) { return s.bytes[e]; }
// 4 bytes at extended offset e (any alignment)
__device__ __forceinline__ uint32_t ldu32(const FusedLds &s, int e) {
    uint32_t a = (uint32_t)e & ~3u;
    uint32_t lo = ldw(s, a), hi = ldw(s, a + 4);
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)e & 3u);
}

// duckdb::string_t of the field [e, e+len); ptr_of_e0 = payload pointer of extended offset 0
__device__ __forceinline__ uint4 make_string_lds(const FusedLds &s, int e, uint32_t len, uint64_t ptr_of_e0) {
    uint4 r;
    r.x = len;
    uint32_t w0 = ldu32(s, e);
    if (len <= EXG_INLINE_LENGTH) {
        uint32_t w1 = ldu32(s, e + 4), w2 = ldu32(s, e + 8);
        uint32_t m0 = len >= 4 ? 0xFFFFFFFFu : ((1u << (8 * len)) - 1u);
        uint32_t l1 = len > 4 ? len - 4 : 0, l2 = len > 8 ? len - 8 : 0;
        uint32_t m1 = l1 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * l1)) - 1u);
        uint32_t m2 = l2 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * l2)) - 1u);
        r.y = w0 & m0;
        r.z = w1 & m1;
        r.w = w2 & m2;
    } else {
        uint64_t ptr = ptr_of_e0 + (uint64_t)e;
        r.y = w0;
        r.z = (uint32_t)ptr;
        r.w = (uint32_t)(ptr >> 32);
    }
    return r;
}

// '\n' count of bytes [b, e), by one wave, straight from global memory (helping path)
__device__ unsigned long long help_count_bytes(const uint8_t *__restrict__ d_in, uint64_t n_bytes, uint64_t b,
                                               uint64_t e, uint32_t lane) {
    unsigned long long cnt = 0;
    if (e > n_bytes) e = n_bytes;
    for (uint64_t off = b + (uint64_t)lane * 16; off < e; off += 1024) {
        uint4 q = *reinterpret_cast<const uint4 *>(d_in + off);
        uint32_t mm = match16(q, 0x0A0A0A0Au);
        if (off + 16 > e) mm &= (1u << (uint32_t)(e - off)) - 1u;
        cnt += __popc(mm);
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    return __shfl(cnt, 0, 64);
}

__device__ __forceinline__ unsigned long long rfl64(unsigned long long x) {  // wave-uniform value -> SGPRs
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long ld_desc(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_desc(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- ordered prefix: central scanner --------------------------------------------------------------
static constexpr int kScanBatches = 8;  // 512 descriptors per scanner probe

__device__ void scanner_wave(const uint8_t *__restrict__ d_in, uint64_t n_bytes,
                             const unsigned long long *__restrict__ tileA, unsigned long long *__restrict__ tileP,
                             uint32_t n_super, uint32_t lane) {
    __builtin_amdgcn_s_setprio(3);
    uint64_t next = 0;
    unsigned long long running = 0;
    unsigned long long t_last = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (next < n_super) {
        unsigned long long d[kScanBatches];
#pragma unroll
        for (int k = 0; k < kScanBatches; k++) {
            uint64_t idx = next + (uint64_t)k * 64 + lane;
            d[k] = idx < n_super ? ld_desc(&tileA[idx]) : 0ull;
        }
        bool progressed = false;
#pragma unroll
        for (int k = 0; k < kScanBatches; k++) {
            unsigned long long rdy = __ballot((d[k] & kFlag) != 0);
            int r = rdy == ~0ull ? 64 : __ffsll((long long)~rdy) - 1;  // leading run of published counts
            if (r > 0) {
                uint32_t c = (int)lane < r ? (uint32_t)(d[k] & kVal) : 0u;
                uint32_t inc = wave_incl_sum(c);
                if ((int)lane < r) st_desc(&tileP[next + lane], kFlag | (running + inc - c));
                running += __shfl(inc, 63, 64);
                next += (uint64_t)r;
                progressed = true;
            }
            if (r < 64) break;
        }
        if (progressed) {
            t_last = __builtin_amdgcn_s_memrealtime();
        } else if (__builtin_amdgcn_s_memrealtime() - t_last > 4000) {
            // ~40 us without the next count: that block may not have been dispatched; count its bytes
            // ourselves so that progress never depends on the dispatch order.
            unsigned long long c = help_count_bytes(d_in, n_bytes, next * kSuper, (next + 1) * kSuper, lane);
            if (lane == 0) st_desc(&tileP[next], kFlag | running);
            running += c;
            next++;
            t_last = __builtin_amdgcn_s_memrealtime();
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// Workgroup side (wave 0): wait for the exclusive prefix of super-tile st (its count is published).
__device__ unsigned long long wait_prefix(const uint8_t *__restrict__ d_in, uint64_t n_bytes,
                                          unsigned long long *__restrict__ tileA,
                                          unsigned long long *__restrict__ tileP, uint32_t st, uint32_t lane) {
    if (st == 0) return 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        unsigned long long x = lane == 0 ? ld_desc(&tileP[st]) : 0ull;
        x = (unsigned long long)__shfl((long long)x, 0, 64);
        if (x & kFlag) return x & kVal;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000) break;  // 2 ms: the scanner is not running
        __builtin_amdgcn_s_sleep(2);
    }
    // Last resort (never seen with in-order dispatch): sum every predecessor ourselves.
    unsigned long long sum = 0;
    for (uint64_t b = 0; b < st; b += 64) {
        uint64_t idx = b + lane;
        unsigned long long x = idx < st ? ld_desc(&tileA[idx]) : kFlag;
        unsigned long long miss = __ballot((x & kFlag) == 0);
        while (miss) {
            int l = __ffsll((long long)miss) - 1;
            miss &= miss - 1;
            unsigned long long c = help_count_bytes(d_in, n_bytes, (b + l) * kSuper, (b + l + 1) * kSuper, lane);
            if ((int)lane == l) x = kFlag | c;
        }
        unsigned long long v = idx < st ? (x & kVal) : 0;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        sum += __shfl(v, 0, 64);
    }
    return sum;
}

struct TileCtx {  // what emission needs besides the LDS contents (all workgroup-uniform)
    uint64_t tile_off;          // offset of the half in d_input
    unsigned long long P;       // '\n' in the buffer before the half
    uint32_t n_lines;           // newline entries of the half (incl. virtual EOF lines)
    int lim_e;                  // extended offset of the end of input inside this half (kWin + min(lim, kTile))
    bool is_eof_tile;           // the input ends in this half and EXG_F_EOF
    bool first_of_buffer;       // half 0 of super-tile 0: what precedes is before d_input[0]
};

// Records that END in the half staged in LDS: wave = column, lane = record.
__device__ __forceinline__ void emit_half(const FusedLds &s, const FastqDev &a, ScanWsHeader *hdr, const TileCtx &c,
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

    if (threadIdx.x == 0) {  // offset just past the last quality line that ends in this half (0: none)
        long long e = 0;
        if (n_rec) {
            e = (long long)c.tile_off + (int)s.nlist[4 + i_first + 4 * (n_rec - 1)] - kWin + 1;
            if ((unsigned long long)e > a.n_bytes) e = (long long)a.n_bytes;
        }
        tile_qend[tile_index] = (unsigned long long)e;
    }
    if (dev_mode >= 3) return;

    for (uint32_t jb = 0; jb < n_rec; jb += 64) {
        const uint32_t j = jb + lane;
        const long long out = (long long)(q_before + j) - (long long)n_hc;
        bool act = j < n_rec;
        int i = 0;
        if (act) {
            i = (int)(i_first + 4 * j);
            int e4 = s.nlist[4 + i];
            act = (uint64_t)((int64_t)c.tile_off + e4 - kWin) >= a.lead && out >= 0;
            if (act && (unsigned long long)out >= a.capacity) {
                if (wave == 0) atomicOr(&hdr->flags, EXG_RF_CAPACITY);
                act = false;
            }
        }
        bool desc_valid = false;
        if (act) {
            const uint32_t q0 = s.nlist[i];  // newline before the name line
            uint4 val = make_uint4(0, 0, 0, 0);
            if (q0 == kNoneE) {
                // the record starts before the window (val stays zero)
                if (wave == 0) {
                    if (c.first_of_buffer) {
                        atomicAdd(&hdr->n_unresolved, 1ull);
                        atomicOr(&hdr->flags, EXG_RF_HEAD_UNRESOLVED);
                    } else {
                        atomicOr(&hdr->overflow, 1u);
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
                } else {
                    desc_valid = e0 > ds;
                    if (desc_valid) val = make_string_lds(s, ds, (uint32_t)(e0 - ds), ptr_of_e0);
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
            }
            // dev_mode 2 keeps the values live but (practically) never stores
            if (dev_mode != 2 || (val.x ^ val.y ^ val.z ^ val.w) == 0x9E3779B9u) {
                exg_string_t *col = wave == 0 ? a.d_name : wave == 1 ? a.d_desc : wave == 2 ? a.d_seq : a.d_qual;
                reinterpret_cast<uint4 *>(col)[out] = val;
            }
        }
        if (wave == 1) {
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

__global__ __launch_bounds__(kThreads, 7) void k_fastq_fused(FastqDev a, unsigned long long *__restrict__ tileA,
                                                             unsigned long long *__restrict__ tileP,
                                                             unsigned long long *__restrict__ tile_qend,
                                                             ScanWsHeader *hdr, uint32_t n_super) {
    __shared__ __attribute__((aligned(16))) FusedLds s;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    // DEV ONLY (tools/dev_probe.py): flags bits 8..11 select ablations on the synthetic FASTQ-150 file
    //   1: analytic prefix instead of the scanner   2: 1 + no output stores
    //   3: 1 + no emission at all                    4: scanner, no emission
    const uint32_t dev_mode = (a.flags >> 8) & 15u;
    if (blockIdx.x == 0) {  // the scanner: one wave, no tile
        if (wave == 0 && !(dev_mode >= 1 && dev_mode <= 3)) scanner_wave(a.d_in, a.n_bytes, tileA, tileP, n_super, lane);
        return;
    }
    const uint32_t st = blockIdx.x - 1;
    const uint64_t super_off = (uint64_t)st * kSuper;
    const uint8_t *__restrict__ d_in = a.d_in;
    const uint64_t n_pad = (a.n_bytes + 15) & ~15ull;
    const int64_t lim64 = (int64_t)a.n_bytes - (int64_t)super_off;
    const int lim_s = lim64 < kSuper ? (int)lim64 : kSuper;  // super-tile-relative end of input (> 0)
    const bool last_super = st + 1 == n_super;

    // ---- loads: 8 strided 16 B chunks per thread, all in flight at once (+ window by wave 3) ------
    uint4 v[kHalves * kRows];
#pragma unroll
    for (int j = 0; j < kHalves * kRows; j++) {
        uint64_t off = super_off + (uint64_t)(j * kThreads + tid) * 16;
        v[j] = off < n_pad ? *reinterpret_cast<const uint4 *>(d_in + off) : make_uint4(0, 0, 0, 0);
    }
    uint4 wv = make_uint4(0, 0, 0, 0);
    const int64_t woff = (int64_t)super_off - kWin + (int64_t)lane * 16;  // wave 3 only
    if (wave == 3 && woff >= 0) wv = *reinterpret_cast<const uint4 *>(d_in + woff);

    // ---- count in registers (all that is needed to publish) ----------------------------------------
    // zero-byte SWAR without compaction: z has bit 7 of a byte clear iff the byte matched, every
    // other bit set, so matches in a dword = 32 - popcount(z): 5 VALU ops per dword.
    uint32_t hi = 0, cnt = 0;
    if (lim_s == kSuper) {
        uint32_t z0 = 0, z1 = 0;  // popcount accumulators (half 0, half 1)
#pragma unroll
        for (int j = 0; j < kHalves * kRows; j++) {
            const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            hi |= (w[0] | w[1] | w[2] | w[3]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t x = w[q] ^ 0x0A0A0A0Au;
                uint32_t y = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
                uint32_t z = y | x | 0x7F7F7F7Fu;
                if (j < kRows)
                    z0 += __popc(z);
                else
                    z1 += __popc(z);
            }
        }
        hi &= 0x80808080u;
        cnt = (kRows * 4 * 32 - z0) | ((kRows * 4 * 32 - z1) << 16);  // half 0 low 16 bits, half 1 high
    } else {
        // the input ends inside this super-tile: mask the bytes past the end
#pragma unroll
        for (int j = 0; j < kHalves * kRows; j++) {
            uint32_t mj = match16(v[j], 0x0A0A0A0Au);
            hi |= (v[j].x | v[j].y | v[j].z | v[j].w) & 0x80808080u;
            int rem = lim_s - (int)(j * kThreads + tid) * 16;  // bytes of this chunk inside the input
            if (rem < 16) mj &= rem <= 0 ? 0u : ((1u << rem) - 1u);
            cnt += __popc(mj) << (j < kRows ? 0 : 16);
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if (wave == 3) hi |= (wv.x | wv.y | wv.z | wv.w) & 0x80808080u;
    uint32_t any_hi = __any(hi != 0);
    if (lane == 0) {
        s.hi_or[wave] = any_hi;
        s.wcnt[wave] = cnt;
    }
    __syncthreads();  // #1
    const uint32_t both = __builtin_amdgcn_readfirstlane(s.wcnt[0] + s.wcnt[1] + s.wcnt[2] + s.wcnt[3]);
    const uint32_t n_nl[kHalves] = {both & 0xFFFFu, both >> 16};
    const bool non_ascii = (s.hi_or[0] | s.hi_or[1] | s.hi_or[2] | s.hi_or[3]) != 0;

    // ---- publish the super-tile count; its prefix is awaited after half 0 has been staged ---------------
    const bool analytic = dev_mode >= 1 && dev_mode <= 3;
    if (tid == 0 && !analytic) st_desc(&tileA[st], kFlag | (unsigned long long)(n_nl[0] + n_nl[1]));
    const unsigned long long halo_nl = rfl64(hdr->halo_nl);
    // Bytes >= 0x80 need UTF-8 validation of every field (the reference builds Arrow Utf8 columns).
    // That is rare in FASTQ and is left to the general path: raise `overflow`, which gates it in.
    if (non_ascii && tid == 0) {
        atomicOr(&hdr->flags, EXG_RF_NON_ASCII);
        atomicOr(&hdr->overflow, 1u);
    }

#pragma unroll
    for (int h = 0; h < kHalves; h++) {
        const int lim_h = lim_s - h * kTile;  // half-relative end of input
        if (h > 0 && lim_h <= 0) {
            // no input in this half: nothing ends here
            if (tid == 0) tile_qend[(uint64_t)st * kHalves + h] = 0;
            break;
        }
        // ---- stage the half: window, bytes, newline list ------------------------------------------
        if (h == 0) {
            if (wave == 3) {
                *reinterpret_cast<uint4 *>(s.bytes + lane * 16) = wv;
                if (lane < 4) s.nlist[lane] = (uint16_t)kNoneE;
                uint32_t wm = woff >= 0 ? match16(wv, 0x0A0A0A0Au) : 0u;
                uint32_t wc = __popc(wm);
                uint32_t wincl = wave_incl_sum(wc);
                uint32_t W = __shfl(wincl, 63, 64);
                uint32_t r = wincl - wc;  // rank, oldest first; goes to slot 4 - (W - r) when >= 0
                while (wm) {
                    uint32_t b = __ffs(wm) - 1;
                    wm &= wm - 1;
                    int slot = 4 - (int)(W - r);
                    if (slot >= 0) s.nlist[slot] = (uint16_t)(lane * 16 + b);
                    r++;
                }
                // a line starts at d_input[0] when EXG_F_BOF: model it as a newline at offset -1 (W == 0 here)
                if (st == 0 && (a.flags & EXG_F_BOF) && lane == 0) s.nlist[3] = (uint16_t)(kWin - 1);
            }
        } else {
            // window = last 1 KiB of the previous half (still in LDS); its last 4 newlines were saved
            uint4 t = make_uint4(0, 0, 0, 0);
            if (wave == 3) t = *reinterpret_cast<const uint4 *>(s.bytes + kTile + lane * 16);
            __syncthreads();
            if (wave == 3) {
                *reinterpret_cast<uint4 *>(s.bytes + lane * 16) = t;
                if (lane < 4) s.nlist[lane] = s.carry[lane];
            }
        }
#pragma unroll
        for (int j = 0; j < kRows; j++)
            *reinterpret_cast<uint4 *>(s.bytes + kWin + (j * kThreads + tid) * 16) = v[h * kRows + j];
        __syncthreads();  // bytes staged
        {
            // Each thread now owns 64 CONTIGUOUS bytes of the half, so newline ranks follow from one
            // 32-bit wave scan.  The four 16 B chunks are read in the order (t>>2)+k mod 4: any 16 lanes
            // of a ds_read_b128 group then touch 16 different bank slots (lane stride alone is 4-way).
            const uint8_t *mine = s.bytes + kWin + tid * 64;
            unsigned long long mask = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t cidx = ((tid >> 2) + k) & 3u;
                uint4 q = *reinterpret_cast<const uint4 *>(mine + cidx * 16);
                mask |= (unsigned long long)match16(q, 0x0A0A0A0Au) << (16 * cidx);
            }
            int rem = lim_h - (int)tid * 64;  // bytes of my 64 inside the input
            if (rem < 64) mask &= rem <= 0 ? 0ull : ((1ull << rem) - 1ull);
            uint32_t c = (uint32_t)__popcll(mask);
            uint32_t inc = wave_incl_sum(c);
            if (lane == 63) s.wtot[wave] = inc;
            __syncthreads();
            uint32_t r = inc - c + (wave > 0 ? s.wtot[0] : 0) + (wave > 1 ? s.wtot[1] : 0) + (wave > 2 ? s.wtot[2] : 0);
            const uint32_t e0 = kWin + tid * 64;
            while (mask) {
                uint32_t b = (uint32_t)__ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if (r < (uint32_t)kNlCap) s.nlist[4 + r] = (uint16_t)(e0 + b);
                r++;
            }
        }
        if (h == 0) {
            // the half is staged while the scanner turns the published count into our prefix
            if (wave == 0) {
                unsigned long long pre;
                if (analytic) {
                    uint64_t k = super_off / 332, w = super_off % 332;
                    pre = 4 * k + (w > 27) + (w > 178) + (w > 180);
                } else {
                    pre = wait_prefix(d_in, a.n_bytes, tileA, tileP, st, lane);
                }
                if (lane == 0) s.prefix = pre;
            }
        }
        __syncthreads();  // staged (and, for h == 0, the prefix has arrived)

        TileCtx c;
        c.tile_off = super_off + (uint64_t)h * kTile;
        c.P = rfl64(s.prefix) + (h ? n_nl[0] : 0);
        c.first_of_buffer = st == 0 && h == 0;
        const bool ends_here = last_super && lim_h <= kTile;  // the input ends inside (or at the end of) this half
        c.is_eof_tile = ends_here && (a.flags & EXG_F_EOF);
        c.lim_e = (lim_h < kTile ? lim_h : kTile) + kWin;
        uint32_t n_lines = n_nl[h];
        if (c.is_eof_tile) {
            // noodles EOF rules: an unterminated last line is a line; a record with its '+' line but no
            // quality line gets an empty one (read_line returns 0 bytes at EOF without error).
            const unsigned long long P0 = a.first_line_index - halo_nl;
            bool unterminated = a.n_bytes > 0 && ldb(s, c.lim_e - 1) != '\n';
            uint32_t n0 = n_lines;
            if (unterminated) n_lines++;
            if (((P0 + c.P + n_lines) & 3) == 3) n_lines++;
            if (tid == 0 && n_lines <= (uint32_t)kNlCap)
                for (uint32_t q = n0; q < n_lines; q++) s.nlist[4 + q] = (uint16_t)c.lim_e;
            __syncthreads();
        }
        c.n_lines = n_lines;
        if (ends_here && tid == 0) {
            hdr->total_nl = c.P + n_nl[h];
            hdr->total_lines = c.P + n_lines;
        }
        if (n_lines > (uint32_t)kNlCap) {
            if (tid == 0) atomicOr(&hdr->overflow, 1u);
            return;
        }
        if (h + 1 < kHalves && tid < 4) {
            // the 4 newlines before the next half, relative to it (entries 4+n-4 .. 4+n-1 of this list)
            uint32_t e = s.nlist[n_lines + tid];
            s.carry[tid] = (e != kNoneE && e >= (uint32_t)kTile) ? (uint16_t)(e - kTile) : (uint16_t)kNoneE;
        }
        emit_half(s, a, hdr, c, halo_nl, dev_mode, lane, wave, tile_qend, (uint64_t)st * kHalves + h);
        if (h + 1 < kHalves) __syncthreads();  // everyone is done reading this half
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
            unsigned long long q = t >= 0 ? tile_qend[t] : 0;
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
    r.flags = hdr->flags;
    r.payload_bytes = 0;
    r.reserved = 0;
    r.error_code = 0;
    r.error_offset = ~0ull;
    r.error_record = ~0ull;
    uint64_t n_rec = n_owned < a.capacity ? n_owned : a.capacity;
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
                    hipStream_t stream) {
    ScanWsHeader *hdr = reinterpret_cast<ScanWsHeader *>(ws);
    uint64_t n_super64 = (dev.n_bytes + kSuper - 1) / kSuper;
    if (n_super64 == 0) n_super64 = 1;
    if (n_super64 > 0x7FFFFFF0ull) {
        set_error("exg_fastq_scan: buffer too large for one launch (%llu super-tiles)", (unsigned long long)n_super64);
        return EXG_E_INVALID_ARG;
    }
    uint32_t n_super = (uint32_t)n_super64;
    // descriptor block: tileA[n_tiles_fused], tileP[n_tiles_fused], tile_qend[n_tiles_fused]
    unsigned long long *tileA = reinterpret_cast<unsigned long long *>(ws + l.off_tile_desc);
    unsigned long long *tileP = tileA + l.n_tiles_fused;
    unsigned long long *tile_qend = tileP + l.n_tiles_fused;
    hipLaunchKernelGGL(k_init_hdr, dim3(1), dim3(1), 0, stream, hdr, l.lines_cap, 0u);
    EXG_HIP_CHECK(hipMemsetAsync(tileA, 0, (size_t)l.n_tiles_fused * 16, stream));
    if (dev.lead) {
        int rc = exg_count_newlines(dev.d_in, 0, dev.lead, (uint64_t *)&hdr->halo_nl, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_fastq_fused, dim3(n_super + 1), dim3(kThreads), 0, stream, dev, tileA, tileP, tile_qend, hdr,
                       n_super);
    hipLaunchKernelGGL(k_fastq_finalize_fused, dim3(1), dim3(256), 0, stream, dev, hdr, tile_qend, n_super * kHalves,
                       args->d_result);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

}  // namespace exg
