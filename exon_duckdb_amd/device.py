"""Device-level scans on torch-owned HBM buffers (torch = allocator + streams only)."""
import ctypes as C

import numpy as np

try:  # torch (allocator + streams) brings a HIP runtime of its own: it must be in the process BEFORE libexon_gpu.so binds to
    import torch as _torch_module  # one, so the module that needs it imports it first (see _lib.load_library)
except ImportError:  # pragma: no cover
    _torch_module = None

from . import abi
from ._lib import ExgError, check, load_library, load_test_library


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise ExgError(abi.EXG_E_NO_DEVICE, "no GPU visible to torch: the record scan has no CPU fallback")
    return torch


def stream_ptr():
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def upload(data: bytes, device="cuda", pad=64):
    """Host bytes -> 16-byte-aligned device uint8 tensor, zero padded (API contract: readable to
    round_up(n,16))."""
    torch = _torch()
    n = len(data)
    t = torch.zeros(n + pad + 16, dtype=torch.uint8, device=device)
    if n:
        t[:n].copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
    assert t.data_ptr() % 16 == 0
    return t


def _check_synth(rc):
    if rc != 0:
        raise ExgError(rc, (load_test_library().exon_tf_support_error() or b"").decode("utf-8", "replace"))


def synth_fastq(n_bytes, file_offset=0, seed=abi.EXG_SYNTH_FASTQ_SEED, device="cuda"):
    """file bytes [file_offset, file_offset + n_bytes) of the synthetic FASTQ-150 file, generated in HBM by the test / bench
    scaffolding library (csrc/testing/exg_synth.hip)"""
    torch = _torch()
    lib = load_test_library()
    t = torch.empty(((n_bytes + 15) // 16) * 16 + 64, dtype=torch.uint8, device=device)
    t[n_bytes:].zero_()
    _check_synth(lib.exg_synth_fastq(C.c_void_p(t.data_ptr()), file_offset, n_bytes, seed, stream_ptr()))
    return t


def _synth_two_pass(fn, n_units, cap, seed, device):
    torch = _torch()
    out = torch.zeros(cap + 64, dtype=torch.uint8, device=device)
    n = C.c_uint64(0)
    _check_synth(fn(C.c_void_p(out.data_ptr()), cap, n_units, seed, C.byref(n), stream_ptr()))
    return out, int(n.value)


def synth_vcf(n_lines, seed=abi.EXG_SYNTH_VCF_SEED, device="cuda"):
    """VCF-8 of SURVEY.md §8 D2 generated in HBM -> (uint8 tensor, n_bytes); same bytes as the oracle's synth_vcf."""
    return _synth_two_pass(load_test_library().exg_synth_vcf, n_lines, 1024 + 64 * n_lines, seed, device)


def synth_fasta(n_records, seed=0xE0A5EED0003, device="cuda"):
    """FASTA of SURVEY.md §8 D2 generated in HBM -> (uint8 tensor, n_bytes)."""
    return _synth_two_pass(load_test_library().exg_synth_fasta, n_records, 4096 + 3200 * n_records, seed, device)


class FastqScan:
    """Reusable output + workspace buffers for exg_fastq_scan on one device buffer size."""

    def __init__(self, n_bytes, capacity_records=None, device="cuda"):
        torch = _torch()
        self.lib = load_library()
        self.n_bytes = n_bytes
        self.capacity = int(capacity_records if capacity_records is not None else n_bytes // 4 + 16)
        cap = max(self.capacity, 1)
        self.cols = [torch.empty((cap, 2), dtype=torch.int64, device=device) for _ in range(4)]
        self.validity = torch.empty(((cap + 63) // 64,), dtype=torch.int64, device=device)
        self.ws_bytes = int(self.lib.exg_scan_workspace_bytes(abi.EXG_FMT_FASTQ, n_bytes))
        self.ws = torch.empty((self.ws_bytes + 255) // 8, dtype=torch.int64, device=device)
        self.result = torch.zeros(8, dtype=torch.int64, device=device)
        self.args = abi.FastqScanArgs()

    def launch(self, d_input, n_bytes=None, lead=0, first_line_index=0, payload_base=0,
               flags=abi.EXG_F_BOF | abi.EXG_F_EOF, algo=abi.EXG_ALGO_AUTO):
        a = self.args
        a.d_input = d_input.data_ptr()
        a.n_bytes = self.n_bytes if n_bytes is None else n_bytes
        a.lead = lead
        a.first_line_index = first_line_index
        a.payload_base = payload_base
        a.flags = flags
        a.algo = algo
        a.d_name, a.d_description, a.d_sequence, a.d_quality = (c.data_ptr() for c in self.cols)
        a.d_description_validity = self.validity.data_ptr()
        a.capacity_records = self.capacity
        a.d_workspace = self.ws.data_ptr()
        a.workspace_bytes = self.ws_bytes
        a.d_result = self.result.data_ptr()
        a.stream = stream_ptr().value
        check(self.lib.exg_fastq_scan(C.byref(a)))

    def fetch(self):
        r = abi.ScanResult()
        check(self.lib.exg_fetch_result(C.c_void_p(self.result.data_ptr()), stream_ptr(), C.byref(r)))
        return r

    def columns_host(self, n):
        """(4 x [n,16] uint8 string_t arrays, validity words) on the host."""
        cols = [c[:n].cpu().numpy().view(np.uint8).reshape(n, 16) for c in self.cols]
        words = self.validity[: (n + 63) // 64].cpu().numpy().view(np.uint64)
        return cols, words


def quality_score_string_to_list(strings, n_rows, d_payload, payload_base, values_capacity=None):
    """quality_score_string_to_list (reference fastq_functions/module.cpp:28-54) on a VARCHAR column that is still
    in HBM: `strings` is a [cap, 2] int64 tensor of string_t (e.g. FastqScan.cols[3]), `d_payload` the buffer its
    pointers address.  Returns (entries [n_rows, 2] int64 = DuckDB list_entry_t {offset, length}, values int32,
    total): total > len(values) means the capacity was too small and nothing was written."""
    torch = _torch()
    lib = load_library()
    dev = strings.device
    cap = int(values_capacity if values_capacity is not None else d_payload.numel())
    entries = torch.empty((max(n_rows, 1), 2), dtype=torch.int64, device=dev)
    values = torch.empty(max(cap, 4), dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    ws_bytes = int(lib.exg_quality_list_workspace_bytes(n_rows))
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=dev)
    a = abi.QualityListArgs()
    a.d_strings = strings.data_ptr()
    a.n_rows = n_rows
    a.d_payload = d_payload.data_ptr()
    a.payload_base = payload_base
    a.d_entries = entries.data_ptr()
    a.d_values = values.data_ptr()
    a.values_capacity = cap
    a.d_total = total.data_ptr()
    a.d_workspace = ws.data_ptr()
    a.workspace_bytes = ws_bytes
    a.stream = stream_ptr().value
    check(lib.exg_quality_score_list(C.byref(a)))
    t = int(total.item())
    return entries[:n_rows], values[:min(t, cap)], t


class VcfScan:
    """Reusable output + workspace buffers for exg_vcf_scan."""

    FIELDS = ["chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "formats"]

    def __init__(self, n_bytes, capacity_records=None, device="cuda"):
        torch = _torch()
        self.lib = load_library()
        self.n_bytes = n_bytes
        self.capacity = int(capacity_records if capacity_records is not None else n_bytes // 8 + 16)
        cap = max(self.capacity, 1)
        self.cols = [torch.empty((cap, 2), dtype=torch.int64, device=device) for _ in range(9)]
        self.pos = torch.empty((cap,), dtype=torch.int64, device=device)
        self.qual = torch.empty((cap,), dtype=torch.float32, device=device)
        self.qual_valid = torch.empty(((cap + 63) // 64,), dtype=torch.int64, device=device)
        self.formats_valid = torch.empty(((cap + 63) // 64,), dtype=torch.int64, device=device)
        self.ws_bytes = int(self.lib.exg_scan_workspace_bytes(abi.EXG_FMT_VCF, n_bytes))
        self.ws = torch.empty((self.ws_bytes + 255) // 8, dtype=torch.int64, device=device)
        self.result = torch.zeros(8, dtype=torch.int64, device=device)
        self.args = abi.VcfScanArgs()

    def launch(self, d_input, n_bytes=None, lead=0, payload_base=0, flags=abi.EXG_F_BOF | abi.EXG_F_EOF,
               algo=abi.EXG_ALGO_AUTO, project=None):
        a = self.args
        a.d_input = d_input.data_ptr()
        a.n_bytes = self.n_bytes if n_bytes is None else n_bytes
        a.lead = lead
        a.payload_base = payload_base
        a.flags = flags
        a.algo = algo
        for k in range(9):
            a.d_fields[k] = self.cols[k].data_ptr() if (project is None or k in project) else None
        a.d_pos = self.pos.data_ptr()
        a.d_qual = self.qual.data_ptr()
        a.d_qual_validity = self.qual_valid.data_ptr()
        a.d_formats_validity = self.formats_valid.data_ptr()
        a.capacity_records = self.capacity
        a.d_workspace = self.ws.data_ptr()
        a.workspace_bytes = self.ws_bytes
        a.d_result = self.result.data_ptr()
        a.stream = stream_ptr().value
        check(self.lib.exg_vcf_scan(C.byref(a)))

    def fetch(self):
        r = abi.ScanResult()
        check(self.lib.exg_fetch_result(C.c_void_p(self.result.data_ptr()), stream_ptr(), C.byref(r)))
        return r

    def host(self, n):
        cols = [c[:n].cpu().numpy().view(np.uint8).reshape(n, 16) for c in self.cols]
        nw = (n + 63) // 64
        return dict(cols=cols, pos=self.pos[:n].cpu().numpy(), qual=self.qual[:n].cpu().numpy(),
                    qual_valid=self.qual_valid[:nw].cpu().numpy().view(np.uint64),
                    formats_valid=self.formats_valid[:nw].cpu().numpy().view(np.uint64))


class FastaScan:
    """Reusable output + workspace buffers for exg_fasta_scan (whole-file buffers)."""

    def __init__(self, n_bytes, capacity_records=None, device="cuda"):
        torch = _torch()
        self.lib = load_library()
        self.n_bytes = n_bytes
        self.capacity = int(capacity_records if capacity_records is not None else n_bytes // 2 + 16)
        cap = max(self.capacity, 1)
        self.cols = [torch.empty((cap, 2), dtype=torch.int64, device=device) for _ in range(3)]
        self.validity = torch.empty(((cap + 63) // 64,), dtype=torch.int64, device=device)
        self.payload = torch.empty((n_bytes + 64,), dtype=torch.uint8, device=device)
        self.ws_bytes = int(self.lib.exg_scan_workspace_bytes(abi.EXG_FMT_FASTA, n_bytes))
        self.ws = torch.empty((self.ws_bytes + 255) // 8, dtype=torch.int64, device=device)
        self.result = torch.zeros(8, dtype=torch.int64, device=device)
        self.args = abi.FastaScanArgs()

    def launch(self, d_input, payload_base=0, seq_payload_base=0, flags=abi.EXG_F_BOF | abi.EXG_F_EOF, lead=0,
               algo=abi.EXG_ALGO_AUTO):
        a = self.args
        a.d_input = d_input.data_ptr()
        a.n_bytes = self.n_bytes
        a.lead = lead
        a.payload_base = payload_base
        a.seq_payload_base = seq_payload_base
        a.flags = flags
        a.algo = algo
        a.d_id, a.d_description, a.d_sequence = (c.data_ptr() for c in self.cols)
        a.d_description_validity = self.validity.data_ptr()
        a.d_seq_payload = self.payload.data_ptr()
        a.capacity_records = self.capacity
        a.d_workspace = self.ws.data_ptr()
        a.workspace_bytes = self.ws_bytes
        a.d_result = self.result.data_ptr()
        a.stream = stream_ptr().value
        check(self.lib.exg_fasta_scan(C.byref(a)))

    def fetch(self):
        r = abi.ScanResult()
        check(self.lib.exg_fetch_result(C.c_void_p(self.result.data_ptr()), stream_ptr(), C.byref(r)))
        return r

    def host(self, n, payload_bytes):
        cols = [c[:n].cpu().numpy().view(np.uint8).reshape(n, 16) for c in self.cols]
        words = self.validity[: (n + 63) // 64].cpu().numpy().view(np.uint64)
        return cols, words, self.payload[:payload_bytes].cpu().numpy()
