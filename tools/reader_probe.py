"""End-to-end (PCIe-inclusive) rates of the two reader boundaries on a 4 GB FASTQ-150 file in the page
cache: exg_open / exg_next_chunk (DuckDB-shaped chunks) and new_reader (Arrow C stream, consumed by pyarrow)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exon_duckdb_amd import device, table_function
from exon_duckdb_amd.arrow import new_reader

path = "/tmp/exg_bench.fastq"
nb = 332 * 12_000_000
with open(path, "wb") as f:
    f.write(device.synth_fastq(nb)[:nb].cpu().numpy().tobytes())
con = table_function.connect()
rel = con.table_function("read_fastq", path)


def arrow_all():
    return sum(b.num_rows for b in new_reader(path, "fastq"))


def arrow_filtered():
    return sum(b.num_rows for b in new_reader(path, "fastq", filters="description='3:N:0:ACGT' AND sequence<'C'"))


import ctypes as C  # noqa: E402


class _ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                ("n_children", C.c_int64), ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p),
                ("release", C.CFUNCTYPE(None, C.c_void_p)), ("private_data", C.c_void_p)]


class _ArrowStream(C.Structure):
    _fields_ = [("get_schema", C.c_void_p), ("get_next", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)),
                ("get_last_error", C.c_void_p), ("release", C.CFUNCTYPE(None, C.c_void_p)), ("private_data", C.c_void_p)]


def arrow_raw(filters=None):
    """the stream pulled through its C callbacks only (no pyarrow objects): the library's own rate"""
    from exon_duckdb_amd import arrow as A
    l = A._lib()
    st = _ArrowStream()
    res = l.new_reader(C.addressof(st), path.encode(), 2048, None, b"fastq", filters.encode() if filters else None)
    assert not res.error
    n = 0
    while True:
        arr = _ArrowArray()
        assert st.get_next(C.addressof(st), C.addressof(arr)) == 0
        if not arr.release:
            break
        n += arr.length
        arr.release(C.addressof(arr))
    st.release(C.addressof(st))
    return n


for label, fn in (("count", rel.count), ("arrow (C callbacks only)", arrow_raw), ("chunks", lambda: sum(rel.chunk_sizes())), ("arrow", arrow_all),
                  ("arrow+filter", arrow_filtered)):
    fn()
    for _ in range(2):
        t0 = time.time(); n = fn(); dt = time.time() - t0
        print(label, n, f"{dt:.3f}s {nb/dt/1e9:.2f} GB/s {n/dt/1e6:.1f} M rec/s", flush=True)
os.remove(path)

# VCF: typed nested columns
from oracle import pyoracle  # noqa: E402  (test infrastructure: only used to write the synthetic input)
vpath = "/tmp/exg_bench.vcf"
data = bytes(pyoracle.synth_vcf(4_000_000))
open(vpath, "wb").write(data)
for label, fn in (("vcf arrow", lambda: sum(b.num_rows for b in new_reader(vpath, "vcf"))),):
    fn()
    for _ in range(2):
        t0 = time.time(); n = fn(); dt = time.time() - t0
        print(label, n, f"{dt:.3f}s {len(data)/dt/1e9:.2f} GB/s {n/dt/1e6:.1f} M rows/s", flush=True)
os.remove(vpath)
