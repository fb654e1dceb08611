"""ctypes mirror of include/exon_gpu.h (keep in sync with the header; tests check sizes)."""
import ctypes as C

EXG_ABI_VERSION = 9
EXG_TYPE_VARCHAR, EXG_TYPE_BIGINT, EXG_TYPE_FLOAT, EXG_TYPE_INTEGER, EXG_TYPE_BOOLEAN, EXG_TYPE_LIST, EXG_TYPE_STRUCT = 1, 2, 3, 4, 5, 6, 7
EXG_VECTOR_SIZE = 2048
EXG_OPEN_CHUNKS = 1   # exg_open_args.flags

EXG_OK = 0
EXG_E_INVALID_ARG, EXG_E_NO_DEVICE, EXG_E_HIP, EXG_E_IO = -1, -2, -3, -4
EXG_E_UNSUPPORTED, EXG_E_PARSE, EXG_E_CAPACITY, EXG_E_NOMEM = -5, -6, -7, -8

EXG_PE_NONE = 0
EXG_PE_FASTQ_NAME_PREFIX = 1
EXG_PE_FASTQ_PLUS_PREFIX = 2
EXG_PE_UNEXPECTED_EOF = 3
EXG_PE_INVALID_UTF8 = 4
EXG_PE_FASTA_MISSING_PREFIX = 5
EXG_PE_FASTA_MISSING_NAME = 6
EXG_PE_FASTA_EMPTY_DEF = 7
EXG_PE_VCF_MISSING_FIELD = 8
EXG_PE_VCF_BAD_POS = 9
EXG_PE_VCF_BAD_QUAL = 10
EXG_PE_VCF_NO_HEADER = 11
EXG_PE_FIELD_TOO_LONG = 12
EXG_PE_VCF_INFO = 13
EXG_PE_VCF_FORMAT = 14

EXG_FMT_FASTA, EXG_FMT_FASTQ, EXG_FMT_VCF = 1, 2, 3
EXG_F_BOF, EXG_F_EOF, EXG_F_NO_STORE = 1, 2, 4
EXG_RF_NON_ASCII, EXG_RF_HEAD_UNRESOLVED, EXG_RF_FALLBACK, EXG_RF_CAPACITY, EXG_RF_INDEX_OVERFLOW = 1, 2, 4, 8, 16
EXG_RF_QUAL_RANGE = 32
EXG_RF_REDO = 64
EXG_ALGO_AUTO, EXG_ALGO_MULTIPASS, EXG_ALGO_FUSED, EXG_ALGO_FUSED_FULL, EXG_ALGO_FUSED_INDEX = 0, 1, 2, 3, 4

EXG_SYNTH_FASTQ_SEED = 0xE0A5EED0001
EXG_SYNTH_VCF_SEED = 0xE0A5EED0002
EXG_SYNTH_FASTQ_RECORD_BYTES = 332


class ScanResult(C.Structure):
    _fields_ = [
        ("n_records", C.c_uint64),
        ("n_lines", C.c_uint64),
        ("consumed_bytes", C.c_uint64),
        ("error_offset", C.c_uint64),
        ("error_record", C.c_uint64),
        ("error_code", C.c_uint32),
        ("flags", C.c_uint32),
        ("payload_bytes", C.c_uint64),
        ("redo_tiles", C.c_uint64),
    ]


class FastqScanArgs(C.Structure):
    _fields_ = [
        ("d_input", C.c_void_p),
        ("n_bytes", C.c_uint64),
        ("lead", C.c_uint64),
        ("first_line_index", C.c_uint64),
        ("payload_base", C.c_uint64),
        ("flags", C.c_uint32),
        ("algo", C.c_uint32),
        ("d_name", C.c_void_p),
        ("d_description", C.c_void_p),
        ("d_sequence", C.c_void_p),
        ("d_quality", C.c_void_p),
        ("d_description_validity", C.c_void_p),
        ("capacity_records", C.c_uint64),
        ("d_workspace", C.c_void_p),
        ("workspace_bytes", C.c_uint64),
        ("d_result", C.c_void_p),
        ("stream", C.c_void_p),
    ]


class VcfScanArgs(C.Structure):
    _fields_ = [
        ("d_input", C.c_void_p),
        ("n_bytes", C.c_uint64),
        ("lead", C.c_uint64),
        ("payload_base", C.c_uint64),
        ("flags", C.c_uint32),
        ("algo", C.c_uint32),
        ("d_fields", C.c_void_p * 9),
        ("d_pos", C.c_void_p),
        ("d_qual", C.c_void_p),
        ("d_qual_validity", C.c_void_p),
        ("d_formats_validity", C.c_void_p),
        ("capacity_records", C.c_uint64),
        ("d_workspace", C.c_void_p),
        ("workspace_bytes", C.c_uint64),
        ("d_result", C.c_void_p),
        ("stream", C.c_void_p),
    ]


class FastaScanArgs(C.Structure):
    _fields_ = [
        ("d_input", C.c_void_p),
        ("n_bytes", C.c_uint64),
        ("lead", C.c_uint64),
        ("payload_base", C.c_uint64),
        ("seq_payload_base", C.c_uint64),
        ("flags", C.c_uint32),
        ("algo", C.c_uint32),
        ("d_id", C.c_void_p),
        ("d_description", C.c_void_p),
        ("d_sequence", C.c_void_p),
        ("d_description_validity", C.c_void_p),
        ("d_seq_payload", C.c_void_p),
        ("capacity_records", C.c_uint64),
        ("d_workspace", C.c_void_p),
        ("workspace_bytes", C.c_uint64),
        ("d_result", C.c_void_p),
        ("stream", C.c_void_p),
    ]


class QualityListArgs(C.Structure):
    _fields_ = [
        ("d_strings", C.c_void_p),
        ("n_rows", C.c_uint64),
        ("d_payload", C.c_void_p),
        ("payload_base", C.c_uint64),
        ("d_entries", C.c_void_p),
        ("d_values", C.c_void_p),
        ("values_capacity", C.c_uint64),
        ("d_total", C.c_void_p),
        ("d_workspace", C.c_void_p),
        ("workspace_bytes", C.c_uint64),
        ("stream", C.c_void_p),
    ]


class OpenArgs(C.Structure):
    _fields_ = [("path", C.c_char_p), ("file_format", C.c_char_p), ("compression", C.c_char_p), ("batch_rows", C.c_uint64),
                ("device", C.c_int), ("device_batch_bytes", C.c_uint64), ("filters", C.c_char_p),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("columns", C.c_uint64), ("flags", C.c_uint64)]


class InflateMember(C.Structure):
    _fields_ = [("comp_off", C.c_uint64), ("comp_size", C.c_uint64), ("out_off", C.c_uint64), ("out_cap", C.c_uint64)]


class InflateStatus(C.Structure):
    _fields_ = [("code", C.c_uint32), ("pad", C.c_uint32), ("produced", C.c_uint64), ("consumed", C.c_uint64)]


# every symbol include/exon_gpu.h declares -> (restype, argtypes); None = not yet bound by name only
SIGNATURES = {
    "exg_abi_version": (C.c_int, []),
    "exg_scan_algo_hint": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64]),
    "exg_device_count": (C.c_int, []),
    "exg_last_error_message": (C.c_char_p, []),
    "exg_parse_error_string": (C.c_char_p, [C.c_uint32]),
    "exg_scan_workspace_bytes": (C.c_uint64, [C.c_int, C.c_uint64]),
    "exg_fastq_scan": (C.c_int, [C.POINTER(FastqScanArgs)]),
    "exg_vcf_scan": (C.c_int, [C.POINTER(VcfScanArgs)]),
    "exg_fasta_scan": (C.c_int, [C.POINTER(FastaScanArgs)]),
    "exg_gzip_index": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "exg_inflate_members": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "exg_zstd_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]),
    "exg_fetch_result": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(ScanResult)]),
    "exg_count_newlines": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "exg_quality_list_workspace_bytes": (C.c_uint64, [C.c_uint64]),
    "exg_quality_score_list": (C.c_int, [C.POINTER(QualityListArgs)]),
    "exg_plan_shards": (C.c_int, [C.POINTER(OpenArgs), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.c_uint32]),
}


class ReaderStats(C.Structure):
    _fields_ = [("device_bytes_now", C.c_uint64), ("device_bytes_peak", C.c_uint64), ("device_mem_cap", C.c_uint64),
                ("device_batch_bytes", C.c_uint64), ("device_batches", C.c_uint64), ("decoded_segments", C.c_uint64),
                ("scan_algo", C.c_uint64), ("input_bytes", C.c_uint64), ("input_compression", C.c_uint64),
                ("nested_ns", C.c_uint64), ("host_vector_bytes", C.c_uint64), ("reserved", C.c_uint64 * 3)]

# libexon_tf_test.so (csrc/testing/): test / bench scaffolding — synthetic inputs generated in HBM, consumers that drain a
# reader's chunks (counting, or folding every row into a digest), host-only introspection, the host-pipeline probe
TEST_SIGNATURES = {
    "exg_synth_fastq": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "exg_synth_vcf": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]),
    "exg_synth_fasta": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]),
    "exon_tf_support_error": (C.c_char_p, []),
    "exon_tf_drain_chunks": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "exon_tf_drain_arrow_vcf": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "exon_tf_drain_formats_digest": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "exon_tf_expect_vcf_formats_file": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "exon_tf_drain_arrow_fastq": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
    "exon_tf_drain_digest": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64)]),
    "exon_tf_drain_digest_from": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "exon_tf_synth_fastq150_host": (None, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "exon_tf_expect_fastq150": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]),
    "exon_tf_expect_fastq_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "exon_tf_expect_vcf_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "exon_tf_filter_explain": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "exon_tf_vcf_header_explain": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "exon_tf_link_probe": (C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "exon_tf_host_pipeline_probe": (C.c_double, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
    "exon_tf_host_zero_bounce_probe": (C.c_double, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double)]),
}
