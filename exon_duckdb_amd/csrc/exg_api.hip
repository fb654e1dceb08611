// exg_api.hip — C-ABI entry points of the device level (include/exon_gpu.h, layer 1) and the
// library plumbing (errors, device probe, result fetch).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "exg_arrow.hpp"
#include "exg_fastq.hpp"

namespace exg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

}  // namespace exg

using namespace exg;

// A reader keeps six to eight HIP streams busy on its device (its scan stream, the upload stream, three decoder lanes, the
// decoded segments' host copies, a checksum stage); HIP maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default and
// streams that share one wait for each other's packets: measured on a BGZF FASTQ read into DataChunks, the 0.1 ms scan of a
// batch sat 23-28 ms behind the decoder lanes' kernels and the 5 ms host copies (EXG_TRACE).  The runtime reads the
// variable at the process's first HIP call.  Until round 5 a constructor of this library called setenv() for it: in a DuckDB
// process that is already multi-threaded when `LOAD exon` runs that is a data race with every getenv (glibc may move `environ`),
// and it changed the queue mapping of every other HIP user of the process.  The library does not touch the environment any
// more: whoever starts the process exports GPU_MAX_HW_QUEUES=8 (bench.py and the Python loader do, before HIP
// is initialised).

extern "C" int exg_abi_version(void) { return EXG_ABI_VERSION; }

extern "C" const char *exg_last_error_message(void) { return g_err; }

extern "C" int exg_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s): libexon_gpu has no CPU fallback", hipGetErrorString(e));
        return EXG_E_NO_DEVICE;
    }
    return n;
}

extern "C" const char *exg_parse_error_string(uint32_t code) {
    switch (code) {
        case EXG_PE_NONE: return "ok";
        case EXG_PE_FASTQ_NAME_PREFIX: return "invalid name prefix";           // noodles-fastq InvalidData
        case EXG_PE_FASTQ_PLUS_PREFIX: return "invalid description prefix";    // noodles-fastq InvalidData
        case EXG_PE_UNEXPECTED_EOF: return "unexpected end of file";
        case EXG_PE_INVALID_UTF8: return "invalid utf-8";
        case EXG_PE_FASTA_MISSING_PREFIX: return "missing prefix ('>')";
        case EXG_PE_FASTA_MISSING_NAME: return "missing name";
        case EXG_PE_FASTA_EMPTY_DEF: return "empty input";
        case EXG_PE_VCF_MISSING_FIELD: return "missing field";
        case EXG_PE_VCF_BAD_POS: return "invalid position";
        case EXG_PE_VCF_BAD_QUAL: return "invalid quality score";
        case EXG_PE_VCF_NO_HEADER: return "missing header";
        case EXG_PE_FIELD_TOO_LONG: return "field longer than 4 GiB";
        case EXG_PE_VCF_INFO: return "invalid info field value";
        case EXG_PE_VCF_FORMAT: return "invalid genotype field value";
        default: return "unknown parse error";
    }
}

extern "C" uint64_t exg_scan_workspace_bytes(int format, uint64_t n_bytes) {
    return fastq_ws_layout(n_bytes, 0, format == EXG_FMT_FASTA ? 4 : 1).total_bytes;
}

extern "C" int exg_fetch_result(const exg_scan_result *d_result, void *stream, exg_scan_result *out) {
    if (!d_result || !out) {
        set_error("exg_fetch_result: null pointer");
        return EXG_E_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    EXG_HIP_CHECK(hipMemcpyAsync(out, d_result, sizeof(*out), hipMemcpyDeviceToHost, s));
    EXG_HIP_CHECK(hipStreamSynchronize(s));
    return EXG_OK;
}

extern "C" int exg_fastq_scan(const exg_fastq_scan_args *a) {
    if (!a || !a->d_result || !a->d_workspace || ((uintptr_t)a->d_workspace & 255) || (a->n_bytes && !a->d_input) || ((uintptr_t)a->d_input & 15) ||
        a->lead > a->n_bytes) {
        set_error("exg_fastq_scan: bad arguments (null pointer, unaligned input or workspace, or lead > n_bytes)");
        return EXG_E_INVALID_ARG;
    }
    if (a->flags & ~EXG_F_ALL) {
        set_error("exg_fastq_scan: unknown flag bits 0x%x", a->flags & ~EXG_F_ALL);
        return EXG_E_INVALID_ARG;
    }
    if (a->capacity_records && !(a->flags & EXG_F_NO_STORE) &&
        (!a->d_name || !a->d_description || !a->d_sequence || !a->d_quality || !a->d_description_validity)) {
        set_error("exg_fastq_scan: null output column");
        return EXG_E_INVALID_ARG;
    }
    FastqWsLayout l = fastq_ws_layout(a->n_bytes, a->workspace_bytes);
    if (a->workspace_bytes < fastq_ws_layout(a->n_bytes, 0).off_nl_pos + 64) {
        set_error("exg_fastq_scan: workspace too small (%llu bytes, need %llu)",
                  (unsigned long long)a->workspace_bytes,
                  (unsigned long long)exg_scan_workspace_bytes(EXG_FMT_FASTQ, a->n_bytes));
        return EXG_E_INVALID_ARG;
    }
    FastqDev dev;
    dev.d_in = (const uint8_t *)a->d_input;
    dev.n_bytes = a->n_bytes;
    dev.lead = a->lead;
    dev.first_line_index = a->first_line_index;
    dev.payload_base = a->payload_base;
    dev.flags = a->flags;
    dev.pad = 0;
    dev.d_name = a->d_name;
    dev.d_desc = a->d_description;
    dev.d_seq = a->d_sequence;
    dev.d_qual = a->d_quality;
    dev.d_desc_valid = a->d_description_validity;
    dev.capacity = a->capacity_records;
    hipStream_t stream = (hipStream_t)a->stream;
    uint8_t *ws = (uint8_t *)a->d_workspace;
    if (a->capacity_records && !(a->flags & EXG_F_NO_STORE))
        EXG_HIP_CHECK(hipMemsetAsync(a->d_description_validity, 0, (size_t)((a->capacity_records + 63) / 64) * 8, stream));
    switch (a->algo) {
        case EXG_ALGO_MULTIPASS:
            return run_fastq_multipass(a, dev, ws, l, stream, false);
        case EXG_ALGO_FUSED:
            return run_fastq_fused(a, dev, ws, l, stream, false);
        case EXG_ALGO_FUSED_FULL:
            return run_fastq_fused(a, dev, ws, l, stream, true);
        case EXG_ALGO_AUTO: {
            int rc = run_fastq_fused(a, dev, ws, l, stream, false);
            if (rc) return rc;
            // general kernels, gated on the device by the fused kernel's overflow word
            return run_fastq_multipass(a, dev, ws, l, stream, true);
        }
        default:
            set_error("exg_fastq_scan: unknown algo %u", a->algo);
            return EXG_E_INVALID_ARG;
    }
}

// ---- quality_score_string_to_list --------------------------------------------------------------------------------
// workspace: goff (n + 1 u64) | scan scratch
extern "C" uint64_t exg_quality_list_workspace_bytes(uint64_t n_rows) {
    return (n_rows + 1 + arrow::scan_tmp_entries(n_rows)) * 8 + 64;
}

extern "C" int exg_quality_score_list(const exg_quality_list_args *a) {
    if (!a || !a->d_total || !a->d_workspace || (a->n_rows && (!a->d_strings || !a->d_entries)) ||
        (a->values_capacity && !a->d_values) || ((uintptr_t)a->d_values & 15) || ((uintptr_t)a->d_workspace & 7)) {
        set_error("exg_quality_score_list: bad arguments (null pointer or unaligned values / workspace)");
        return EXG_E_INVALID_ARG;
    }
    if (a->workspace_bytes < exg_quality_list_workspace_bytes(a->n_rows)) {
        set_error("exg_quality_score_list: workspace too small (%llu bytes, need %llu)", (unsigned long long)a->workspace_bytes,
                  (unsigned long long)exg_quality_list_workspace_bytes(a->n_rows));
        return EXG_E_INVALID_ARG;
    }
    hipStream_t s = (hipStream_t)a->stream;
    uint64_t *d_goff = (uint64_t *)a->d_workspace;
    uint64_t *d_tmp = d_goff + a->n_rows + 1;
    if (!a->n_rows) {
        EXG_HIP_CHECK(hipMemsetAsync(a->d_total, 0, 8, s));
        return EXG_OK;
    }
    const arrow::StrCol col{a->d_strings, (const uint8_t *)a->d_payload, a->payload_base};
    arrow::utf8_goff_from_col(col, nullptr, a->n_rows, d_goff, d_tmp, s);
    EXG_HIP_CHECK(hipMemcpyAsync(a->d_total, d_goff + a->n_rows, 8, hipMemcpyDeviceToDevice, s));
    arrow::quality_list(col, a->n_rows, d_goff, (arrow::ListEntry *)a->d_entries, a->d_values, a->values_capacity, s);
    EXG_HIP_CHECK(hipGetLastError());
    return EXG_OK;
}
