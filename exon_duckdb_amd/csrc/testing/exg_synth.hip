// exg_synth.hip — TEST / BENCH SCAFFOLDING (libexon_tf_test.so): the deterministic synthetic inputs of SURVEY.md §8 D2 generated
// in HBM, so that bench.py's device-level legs need no PCIe traffic.  Byte for byte the generators the section specifies;
// tests compare them with the oracle's.
//   int exg_synth_fastq(void *d_out, uint64_t file_offset, uint64_t n_bytes, uint64_t seed, void *stream);
//       file bytes [file_offset, file_offset + n_bytes) of the FASTQ-150 file (332 B per record)
//   int exg_synth_vcf(void *d_out, uint64_t cap, uint64_t n_lines, uint64_t seed, uint64_t *n_bytes, void *stream);
//   int exg_synth_fasta(void *d_out, uint64_t cap, uint64_t n_records, uint64_t seed, uint64_t *n_bytes, void *stream);
//       lines / records vary in length: lengths -> scan -> write; *n_bytes = bytes written (EXG_E_CAPACITY beyond cap)
#include <string.h>

#include <algorithm>
#include <string>

#include "exg_common.hpp"
#include "exg_scan.hpp"

namespace exg {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t synth_word(uint64_t seed, uint64_t k, uint64_t j) {
    return splitmix64((seed ^ (k * 0x9E3779B97F4A7C15ull)) + j);
}

__device__ uint32_t synth_fastq_byte(uint64_t seed, uint64_t off) {
    uint64_t k = off / 332, w = off % 332;
    if (w < 28) {
        if (w < 4) return (uint32_t) "@SYN"[w];
        if (w < 16) {
            uint64_t v = k % 1000000000000ull;
            for (uint64_t i = 15; i > w; i--) v /= 10;
            return (uint32_t)('0' + v % 10);
        }
        if (w == 16) return ' ';
        if (w == 17) return (uint32_t)('0' + k % 4);
        if (w < 27) return (uint32_t) ":N:0:ACGT"[w - 18];
        return '\n';
    }
    if (w < 178) {
        uint64_t i = w - 28;
        return (uint32_t) "ACGT"[(synth_word(seed, k, i / 32) >> (2 * (i % 32))) & 3];
    }
    if (w == 178) return '\n';
    if (w == 179) return '+';
    if (w == 180) return '\n';
    if (w < 331) {
        uint64_t i = w - 181;
        uint64_t b = (synth_word(seed, k, 8 + i / 8) >> (8 * (i % 8))) & 0xFF;
        return (uint32_t)('!' + ((b * 41) >> 8));
    }
    return '\n';
}

// one thread per 16 output bytes
__global__ __launch_bounds__(256) void k_synth_fastq(uint8_t *__restrict__ out, uint64_t file_offset, uint64_t n_bytes,
                                                     uint64_t seed) {
    uint64_t n_chunks = (n_bytes + 15) / 16;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks;
         c += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t w[4] = {0, 0, 0, 0};
        uint64_t base = c * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint64_t o = base + i;
            uint32_t b = o < n_bytes ? synth_fastq_byte(seed, file_offset + o) : 0u;
            w[i >> 2] |= b << (8 * (i & 3));
        }
        *reinterpret_cast<uint4 *>(out + base) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- VCF-8 and FASTA: variable-length lines / records, so the generator runs in two passes — lengths ->
// exclusive scan -> every line (record) written at its offset.  Byte for byte the oracle's orc_synth_vcf / orc_synth_fasta.
__device__ __forceinline__ uint32_t dec_digits(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10) v /= 10, n++;
    return n;
}
__device__ __forceinline__ uint32_t put_dec(uint8_t *p, uint64_t v) {
    const uint32_t n = dec_digits(v);
    for (uint32_t i = n; i-- > 0;) p[i] = (uint8_t)('0' + v % 10), v /= 10;
    return n;
}
__device__ __forceinline__ uint32_t put_dec_pad(uint8_t *p, uint64_t v, uint32_t width) {
    for (uint32_t i = width; i-- > 0;) p[i] = (uint8_t)('0' + v % 10), v /= 10;
    return width;
}
__device__ __forceinline__ uint32_t put_str(uint8_t *p, const char *s) {
    uint32_t n = 0;
    while (s[n]) p[n] = (uint8_t)s[n], n++;
    return n;
}
// one data line into buf (<= 96 bytes); returns its length
__device__ uint32_t synth_vcf_line(uint64_t seed, uint64_t i, uint64_t per_chrom, uint8_t *buf) {
    const uint64_t h = synth_word(seed, i, 0), h2 = synth_word(seed, i, 1);
    uint32_t n = 0;
    n += put_dec(buf + n, i / per_chrom + 1);
    buf[n++] = '\t';
    n += put_dec(buf + n, (i % per_chrom) * 37 + 1 + (h & 31));
    buf[n++] = '\t';
    if (h & 0x100) {
        buf[n++] = 'r', buf[n++] = 's';
        n += put_dec_pad(buf + n, h2 % 1000000000ull, 9);
    } else {
        buf[n++] = '.';
    }
    buf[n++] = '\t';
    buf[n++] = (uint8_t) "ACGT"[(h >> 10) & 3];
    buf[n++] = '\t';
    if (((h >> 12) & 7) == 0)
        n += put_str(buf + n, "A,C");
    else
        buf[n++] = (uint8_t) "ACGT"[(h >> 15) & 3];
    buf[n++] = '\t';
    if (((h >> 20) & 15) != 0) {
        const uint64_t v = (h >> 24) % 10000;  // "%.1f" of v / 10.0
        n += put_dec(buf + n, v / 10);
        buf[n++] = '.';
        buf[n++] = (uint8_t)('0' + v % 10);
    } else {
        buf[n++] = '.';
    }
    buf[n++] = '\t';
    const uint32_t f = (uint32_t)(h >> 40) & 3;
    n += put_str(buf + n, f == 1 ? "." : f == 2 ? "q10" : "PASS");
    buf[n++] = '\t';
    n += put_str(buf + n, "DP=");
    n += put_dec(buf + n, h2 % 500);
    n += put_str(buf + n, ";AF=0.");  // "%.4f" of a value below 1
    n += put_dec_pad(buf + n, (h2 >> 16) % 10000, 4);
    if ((h2 >> 40) & 1) n += put_str(buf + n, ";DB");
    buf[n++] = '\n';
    return n;
}
struct VcfLenF {
    uint64_t seed, per_chrom;
    __device__ uint64_t operator()(uint64_t i) const {
        uint8_t buf[96];
        return synth_vcf_line(seed, i, per_chrom, buf);
    }
};
__global__ __launch_bounds__(256) void k_synth_vcf(uint8_t *out, uint64_t n_lines, uint64_t seed, uint64_t per_chrom,
                                                   const uint64_t *__restrict__ off) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_lines; i += (uint64_t)gridDim.x * 256) {
        uint8_t buf[96];
        const uint32_t n = synth_vcf_line(seed, i, per_chrom, buf);
        uint8_t *dst = out + off[i];
        for (uint32_t k = 0; k < n; k++) dst[k] = buf[k];
    }
}

__device__ __forceinline__ void synth_fasta_shape(uint64_t seed, uint64_t k, uint64_t *total, uint64_t *lines) {
    const uint64_t h = synth_word(seed, k, 1000);
    *lines = 5 + h % 46;
    *total = (*lines - 1) * 60 + 1 + (h >> 8) % 60;
}
__device__ uint32_t synth_fasta_header(uint64_t k, uint64_t total, uint8_t *buf) {
    uint32_t n = put_str(buf, ">seq");
    n += put_dec(buf + n, k);
    if (k % 3 != 2) {
        n += put_str(buf + n, " synthetic record ");
        n += put_dec(buf + n, k);
        n += put_str(buf + n, " len=");
        n += put_dec(buf + n, total);
    }
    buf[n++] = '\n';
    return n;
}
struct FastaLenF {
    uint64_t seed;
    __device__ uint64_t operator()(uint64_t k) const {
        uint64_t total, lines;
        synth_fasta_shape(seed, k, &total, &lines);
        uint8_t buf[96];
        return synth_fasta_header(k, total, buf) + total + lines;
    }
};
__global__ __launch_bounds__(256) void k_synth_fasta(uint8_t *out, uint64_t n_records, uint64_t seed, const uint64_t *__restrict__ off) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n_records; k += (uint64_t)gridDim.x * 256) {
        uint64_t total, lines;
        synth_fasta_shape(seed, k, &total, &lines);
        uint8_t buf[96];
        const uint32_t hn = synth_fasta_header(k, total, buf);
        uint8_t *dst = out + off[k];
        for (uint32_t i = 0; i < hn; i++) dst[i] = buf[i];
        dst += hn;
        uint64_t w = 0;
        for (uint64_t i = 0; i < total; i++) {
            if ((i & 31) == 0) w = synth_word(seed, k, i / 32);
            *dst++ = (uint8_t) "ACGT"[(w >> (2 * (i % 32))) & 3];
            if (i % 60 == 59 || i + 1 == total) *dst++ = '\n';
        }
    }
}

template <class F, class K>
static int synth_two_pass(F len_f, K write, uint8_t *d_out, uint64_t cap, uint64_t n, const char *header, uint64_t *n_bytes, hipStream_t st) {
    const uint64_t hn = header ? strlen(header) : 0;
    uint64_t *d_off = nullptr, *d_tmp = nullptr;
    EXG_HIP_CHECK(hipMalloc(&d_off, (n + 1) * 8 + 16));
    if (hipMalloc(&d_tmp, xscan_tmp_entries(n) * 8 + 16) != hipSuccess) {
        (void)hipFree(d_off);
        set_error("exg_synth: out of device memory");
        return EXG_E_HIP;
    }
    launch_xscan(len_f, n, d_off, d_tmp, st);
    uint64_t total = 0;
    hipError_t e = hipMemcpyAsync(&total, d_off + n, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    int rc = EXG_OK;
    if (e != hipSuccess) {
        set_error("exg_synth: %s", hipGetErrorString(e));
        rc = EXG_E_HIP;
    } else if (hn + total > cap) {
        set_error("exg_synth: %llu bytes do not fit the buffer of %llu", (unsigned long long)(hn + total), (unsigned long long)cap);
        rc = EXG_E_CAPACITY;
    } else {
        if (hn) e = hipMemcpyAsync(d_out, header, hn, hipMemcpyHostToDevice, st);
        if (n) write(d_out + hn, d_off);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            set_error("exg_synth: %s", hipGetErrorString(e));
            rc = EXG_E_HIP;
        }
        *n_bytes = hn + total;
    }
    (void)hipFree(d_off);
    (void)hipFree(d_tmp);
    return rc;
}

}  // namespace exg

// VCF-8 of SURVEY.md §8 D2 (header + n_lines data lines of ~49 bytes) generated on the device; *n_bytes = what was
// written (<= cap, else EXG_E_CAPACITY).  Synchronises the stream (bench / tests input only).
extern "C" int exg_synth_vcf(void *d_out, uint64_t cap, uint64_t n_lines, uint64_t seed, uint64_t *n_bytes, void *stream) {
    using namespace exg;
    if (!d_out || !n_bytes) {
        set_error("exg_synth_vcf: null argument");
        return EXG_E_INVALID_ARG;
    }
    std::string header = "##fileformat=VCFv4.2\n";
    for (int c = 1; c <= 22; c++) header += "##contig=<ID=" + std::to_string(c) + ">\n";
    header +=
        "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Depth\">\n"
        "##INFO=<ID=AF,Number=1,Type=Float,Description=\"Allele frequency\">\n"
        "##INFO=<ID=DB,Number=0,Type=Flag,Description=\"dbSNP membership\">\n"
        "##INFO=<ID=ANN,Number=1,Type=String,Description=\"Annotation\">\n"
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
    const uint64_t per_chrom = n_lines / 22 + 1;
    hipStream_t st = (hipStream_t)stream;
    auto write = [&](uint8_t *out, const uint64_t *d_off) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n_lines + 255) / 256, 65536);
        hipLaunchKernelGGL(k_synth_vcf, dim3(grid), dim3(256), 0, st, out, n_lines, seed, per_chrom, d_off);
    };
    return synth_two_pass(VcfLenF{seed, per_chrom}, write, (uint8_t *)d_out, cap, n_lines, header.c_str(), n_bytes, st);
}

// FASTA of SURVEY.md §8 D2 (60-column wrapped sequences of 5..50 lines, every 3rd record without description).
extern "C" int exg_synth_fasta(void *d_out, uint64_t cap, uint64_t n_records, uint64_t seed, uint64_t *n_bytes, void *stream) {
    using namespace exg;
    if (!d_out || !n_bytes) {
        set_error("exg_synth_fasta: null argument");
        return EXG_E_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    auto write = [&](uint8_t *out, const uint64_t *d_off) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n_records + 255) / 256, 65536);
        hipLaunchKernelGGL(k_synth_fasta, dim3(grid), dim3(256), 0, st, out, n_records, seed, d_off);
    };
    return synth_two_pass(FastaLenF{seed}, write, (uint8_t *)d_out, cap, n_records, nullptr, n_bytes, st);
}

extern "C" int exg_synth_fastq(void *d_out, uint64_t file_offset, uint64_t n_bytes, uint64_t seed, void *stream) {
    if (!d_out || ((uintptr_t)d_out & 15)) {
        exg::set_error("exg_synth_fastq: output must be a 16-byte aligned device pointer");
        return EXG_E_INVALID_ARG;
    }
    if (!n_bytes) return EXG_OK;
    uint64_t n_chunks = (n_bytes + 15) / 16;
    uint64_t blocks = (n_chunks + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(exg::k_synth_fastq, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, (uint8_t *)d_out,
                       file_offset, n_bytes, seed);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}

// ---- the host link, each way and both ways at once (bench.py `pcie_link`, tools/duplex_probe.py) -----------------------------
// out[0] H2D by hipMemcpyAsync (SDMA) alone, out[1] D2H by hipMemcpyAsync alone, out[2] both at once (aggregate bytes / s),
// out[3] D2H by a copy KERNEL alone (16 B per lane stores into the pinned block, which is mapped into the device's address space),
// out[4] H2D by hipMemcpyAsync + D2H by the kernel at once (aggregate), out[5] two D2H hipMemcpyAsync at once (aggregate).
namespace exg {
typedef uint32_t copy_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const copy_v4 *__restrict__ src, copy_v4 *__restrict__ dst, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
}  // namespace exg
extern "C" int exon_tf_link_probe(int device, uint64_t bytes, int copy_blocks, double *out) {
    if (!out || bytes < (1u << 20)) return EXG_E_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return EXG_E_HIP;
    void *h_up = nullptr, *h_dn = nullptr, *h_dn2 = nullptr, *d_up = nullptr, *d_dn = nullptr, *d_dn2 = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;
    int rc = EXG_E_HIP;
    auto now = [] {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec + ts.tv_nsec * 1e-9;
    };
    do {
        if (hipHostMalloc(&h_up, bytes, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&h_dn, bytes, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&h_dn2, bytes, hipHostMallocDefault) != hipSuccess)
            break;
        if (hipMalloc(&d_up, bytes) != hipSuccess || hipMalloc(&d_dn, bytes) != hipSuccess || hipMalloc(&d_dn2, bytes) != hipSuccess) break;
        if (hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) break;
        memset(h_up, 1, bytes);
        (void)hipMemset(d_dn, 2, bytes);
        (void)hipMemset(d_dn2, 3, bytes);
        const uint64_t n16 = bytes / 16;
        const int grid = copy_blocks > 0 ? copy_blocks : 256;
        auto up = [&] { (void)hipMemcpyAsync(d_up, h_up, bytes, hipMemcpyHostToDevice, s1); };
        auto dn = [&] { (void)hipMemcpyAsync(h_dn, d_dn, bytes, hipMemcpyDeviceToHost, s2); };
        auto dn2 = [&] { (void)hipMemcpyAsync(h_dn2, d_dn2, bytes, hipMemcpyDeviceToHost, s1); };
        auto dnk = [&] { hipLaunchKernelGGL(exg::k_copy16, dim3(grid), dim3(256), 0, s2, (const exg::copy_v4 *)d_dn, (exg::copy_v4 *)h_dn, n16); };
        auto best = [&](auto fn, double moved) {
            double b = 1e30;
            for (int r = 0; r < 4; r++) {
                (void)hipDeviceSynchronize();
                const double t0 = now();
                fn();
                (void)hipDeviceSynchronize();
                b = std::min(b, now() - t0);
            }
            return moved / b;
        };
        out[0] = best([&] { up(); }, (double)bytes);
        out[1] = best([&] { dn(); }, (double)bytes);
        out[2] = best([&] { up(); dn(); }, 2.0 * bytes);
        out[3] = best([&] { dnk(); }, (double)bytes);
        out[4] = best([&] { up(); dnk(); }, 2.0 * bytes);
        out[5] = best([&] { dn(); dn2(); }, 2.0 * bytes);
        rc = hipGetLastError() == hipSuccess ? EXG_OK : EXG_E_HIP;
    } while (0);
    if (s1) (void)hipStreamDestroy(s1);
    if (s2) (void)hipStreamDestroy(s2);
    for (void *p : {d_up, d_dn, d_dn2})
        if (p) (void)hipFree(p);
    for (void *p : {h_up, h_dn, h_dn2})
        if (p) (void)hipHostFree(p);
    return rc;
}
