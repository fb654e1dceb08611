"""`new_reader` — the reference's own FFI entry point (exon/include/rust.hpp:41-46) — from Python.

The C function fills an Arrow C stream; pyarrow imports it, which is what DuckDB's Arrow scan does with
the reference's stream (exon/src/exon/arrow_table_function/module.cpp:228-250).

    rdr = new_reader("x/test.fastq", "fastq")                       # pyarrow.RecordBatchReader
    rdr = new_reader("x/a.vcf.gz", "vcf", filters="chrom='1' AND pos>=1000")
    table = rdr.read_all()
"""
import ctypes as C

from ._lib import ExgError, load_library
from . import abi


class ReaderResult(C.Structure):
    _fields_ = [("error", C.c_void_p)]


_bound = False


def _lib():
    global _bound
    l = load_library()
    if not _bound:
        l.new_reader.restype = ReaderResult
        l.new_reader.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p]
        _bound = True
    return l


def new_reader(uri, file_format, batch_size=abi.EXG_VECTOR_SIZE, compression=None, filters=None):
    import pyarrow as pa

    l = _lib()
    stream = (C.c_uint64 * 8)()  # struct ArrowArrayStream (5 pointers), filled by the callee
    enc = lambda s: None if s is None else s.encode()  # noqa: E731
    res = l.new_reader(C.addressof(stream), enc(uri), batch_size, enc(compression), enc(file_format), enc(filters))
    if res.error:
        msg = C.string_at(res.error).decode("utf-8", "replace")
        C.CDLL(None).free(C.c_void_p(res.error))
        raise ExgError(abi.EXG_E_INVALID_ARG, msg)
    return pa.RecordBatchReader._import_from_c(C.addressof(stream))
