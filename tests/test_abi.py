"""CPU checks of the drop-in boundary: libexon_gpu.so loads, exports every function that
include/exon_gpu.h declares, and the ctypes mirrors (exon_duckdb_amd/abi.py) have the C layout."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "exon_gpu.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n.startswith("exg_") or n in ("replacement_scan", "new_reader")))


def test_header_declares_the_expected_surface():
    fns = declared_functions()
    for must in ["exg_fastq_scan", "exg_vcf_scan", "exg_fasta_scan", "exg_open", "exg_next_chunk", "exg_count_only",
                 "exg_close", "exg_count_newlines", "exg_fastq_guess_phase", "replacement_scan", "new_reader"]:
        assert must in fns


def test_library_loads_and_exports_every_declared_symbol():
    from exon_duckdb_amd import load_library, LIB_PATH

    lib = load_library()
    assert os.path.exists(LIB_PATH)
    missing = [f for f in declared_functions() if not hasattr(lib, f)]
    assert not missing, f"declared in include/exon_gpu.h but not exported: {missing}"
    from exon_duckdb_amd import abi
    assert lib.exg_abi_version() == abi.EXG_ABI_VERSION == 9
    lib.exg_parse_error_string.restype = C.c_char_p
    assert lib.exg_parse_error_string(1) == b"invalid name prefix"


def test_product_library_exports_only_the_declared_c_abi():
    """test / bench scaffolding (synthetic-input generators, chunk-draining consumers, introspection helpers) lives in
    libexon_tf_test.so: libexon_gpu.so exports exactly what include/exon_gpu.h declares"""
    from exon_duckdb_amd import LIB_PATH, load_library
    load_library()
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if " T " in ln)
    assert exported == declared_functions(), sorted(set(exported) ^ set(declared_functions()))


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    from exon_duckdb_amd import ExgError, device, load_library

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ExgError):
        device.FastqScan(1024)
    lib = load_library()
    assert lib.exg_device_count() < 0 and b"no CPU fallback" in lib.exg_last_error_message()


def test_ctypes_mirrors_match_the_c_layout(tmp_path):
    from exon_duckdb_amd import abi
    from exon_duckdb_amd.table_function import Chunk, ExgType, ExgVector, Schema

    structs = {"exg_scan_result": abi.ScanResult, "exg_fastq_scan_args": abi.FastqScanArgs,
               "exg_vcf_scan_args": abi.VcfScanArgs, "exg_fasta_scan_args": abi.FastaScanArgs,
               "exg_chunk": Chunk, "exg_schema": Schema, "exg_type": ExgType, "exg_vector": ExgVector}
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, cls in structs.items():
        prog.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            prog.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    prog.append('printf("exg_string_t %zu\\n", sizeof(exg_string_t)); return 0; }')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    assert got["exg_string_t"] == "16"
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_error_code_numbering_is_shared_with_the_oracle():
    from exon_duckdb_amd import abi
    src = open(HEADER).read()
    for name, val in re.findall(r"#define (EXG_PE_[A-Z0-9_]+) (\d+)", src):
        assert getattr(abi, name) == int(val)
    for name, val in re.findall(r"#define (EXG_RF_[A-Z0-9_]+) (\d+)u", src):
        assert getattr(abi, name) == int(val)


def test_catalog_and_replacement_scan_on_cpu():
    # LoadInternal registers the three table functions (+ alias); ReplacementScan maps extensions
    from exon_duckdb_amd import table_function, load_library

    con = table_function.connect()
    for fn in ["read_fasta", "read_fastq", "read_vcf_file_records", "read_vcf"]:
        assert con.has_table_function(fn)
    assert not con.has_table_function("read_gff")          # out of scope
    assert con.replacement_scan("./t/test.fasta") == "read_fasta"
    assert con.replacement_scan("./t/TEST.FASTA.GZ") == "read_fasta"      # module.cpp:323 lower-cases
    assert con.replacement_scan("./t/test.fq.zst") == "read_fastq"
    assert con.replacement_scan("./t/index.vcf.gz") == "read_vcf_file_records"
    assert con.replacement_scan("./t/table.parquet") is None

    class RS(C.Structure):
        _fields_ = [("file_type", C.c_char_p)]
    lib = load_library()
    lib.replacement_scan.restype = RS
    lib.replacement_scan.argtypes = [C.c_char_p]
    assert lib.replacement_scan(b"a/b.fastq.gz").file_type == b"FASTQ"
    assert lib.replacement_scan(b"a/b.txt").file_type is None


def test_bind_errors_are_raised_at_bind_time_without_a_gpu():
    # SELECT count(*) FROM read_fastq('')  ->  statement error (test_fastq_scan.test:61-62)
    from exon_duckdb_amd import ExgError, table_function

    con = table_function.connect()
    with pytest.raises(ExgError):
        con.table_function("read_fastq", "")
    with pytest.raises(ExgError):
        con.table_function("read_fasta", "/nonexistent/x.fasta")
    with pytest.raises(ExgError):
        con.table_function("read_gff", "x.gff")


def test_oracle_and_product_replacement_scan_agree(oracle):
    from exon_duckdb_amd import table_function
    con = table_function.connect()
    names = {"FASTA": "read_fasta", "FASTQ": "read_fastq", "VCF": "read_vcf_file_records", None: None}
    for uri in ["a.fasta", "a.fa", "a.fna.gz", "b.fastq", "b.fq.zst", "c.vcf", "c.vcf.gz", "d.txt", "noext", "x.gz"]:
        assert con.replacement_scan(uri) == names[oracle.replacement_scan(uri)], uri
