"""exon_duckdb_amd — MI355X-native record scan (read_fasta / read_fastq / read_vcf_file_records).

Thin Python plumbing over libexon_gpu.so (C-ABI: include/exon_gpu.h).  The compute is hand-written
HIP for gfx950; this package only loads the library, allocates device memory through torch and
mirrors the reference's table-function surface for tests.  There is NO CPU fallback: importing
works anywhere, but every scan raises if the library or a GPU is missing.
"""
from ._lib import lib, load_library, load_test_library, ExgError, LIB_PATH  # noqa: F401
from . import abi  # noqa: F401

__all__ = ["lib", "load_library", "load_test_library", "ExgError", "LIB_PATH", "abi"]
