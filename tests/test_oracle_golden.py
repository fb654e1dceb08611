"""Pins the oracle (oracle/exon_oracle.c) against every expectation the reference's own
sqllogictests hold for the record-scan path, on the reference's fixture files (tests/golden/)."""
import gzip
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def expected(golden_dir):
    with open(os.path.join(golden_dir, "expected.json")) as f:
        return json.load(f)


def _read(golden_dir, name):
    with open(os.path.join(golden_dir, name), "rb") as f:
        data = f.read()
    if name.endswith((".gz", ".gzip")):
        data = gzip.decompress(data)  # multi-member aware (BGZF .vcf.gz)
    return data


# ---- FASTQ: test_fastq_scan.test ------------------------------------------------------------

def test_fastq_count(oracle, golden_dir, expected):
    # test_fastq_scan.test:5-8
    r = oracle.fastq_parse(_read(golden_dir, "test.fastq"))
    assert r.error_code == 0
    assert r.n_rows == expected["fastq"]["test.fastq"]["count"] == 2


@pytest.mark.parametrize("name", ["test.fastq.gz", "test.fastq.gzip"])
def test_fastq_gz_count(oracle, golden_dir, expected, name):
    # test_fastq_scan.test:11-20
    r = oracle.fastq_parse(_read(golden_dir, name))
    assert r.error_code == 0 and r.n_rows == expected["fastq"][name]["count"]


def test_fastq_row0_values_and_column_order(oracle, golden_dir, expected):
    # test_fastq_scan.test:35-41 — SELECT * LIMIT 1: name, description, sequence, quality
    r = oracle.fastq_parse(_read(golden_dir, "test.fastq"))
    order = list(r.columns.keys())
    assert order == ["name", "description", "sequence", "quality_scores"]
    got = [r.columns[k].row(0).decode() for k in order]
    assert got == expected["fastq"]["row0"]["values"]


def test_fastq_directory_count(oracle, golden_dir, expected):
    # test_fastq_scan.test:65-68 — read_fastq('…/fastq/') lists the directory: 2 files x 2 records
    d = os.path.join(golden_dir, "fastq")
    total = sum(oracle.fastq_parse(_read(golden_dir, os.path.join("fastq", f))).n_rows for f in sorted(os.listdir(d)))
    assert total == expected["fastq"]["directory"]["count"] == 4


# ---- FASTA: test_fasta_scan.test, test_fasta_copy.test -----------------------------------------

@pytest.mark.parametrize("name", ["test.fasta", "test.fasta.gz", "test.fasta.gzip"])
def test_fasta_count(oracle, golden_dir, expected, name):
    # test_fasta_scan.test:5-20
    r = oracle.fasta_parse(_read(golden_dir, name))
    assert r.error_code == 0 and r.n_rows == expected["fasta"][name]["count"] == 2


def test_fasta_where_id(oracle, golden_dir, expected):
    # test_fasta_scan.test:34-37 — column is called `id`; WHERE id = 'a' keeps one row
    r = oracle.fasta_parse(_read(golden_dir, "test.fasta"))
    assert list(r.columns.keys())[0] == "id"
    assert sum(1 for v in r.columns["id"].to_list() if v == b"a") == expected["fasta"]["where_id_a_count"]["count"]


def test_fasta_null_description(oracle, golden_dir, expected):
    # test_fasta_copy.test:75-80 — `WHERE description IS NULL` returns (b, NULL, ATCG)
    r = oracle.fasta_parse(_read(golden_dir, "test.mixed-desc.fasta"))
    assert list(r.columns.keys()) == expected["fasta"]["mixed_desc_null_row"]["columns"]
    rows = [[r.columns[k].row(i) for k in r.columns] for i in range(r.n_rows)]
    nulls = [row for row in rows if row[1] is None]
    assert nulls == [[b"b", None, b"ATCG"]]


def test_fasta_directory_count(oracle, golden_dir, expected):
    # test_fasta_scan.test:55-59
    d = os.path.join(golden_dir, "fasta")
    total = sum(oracle.fasta_parse(_read(golden_dir, os.path.join("fasta", f))).n_rows for f in sorted(os.listdir(d)))
    assert total == expected["fasta"]["directory"]["count"] == 4


# ---- VCF: test_vcf_record_scan.test -------------------------------------------------------------

@pytest.mark.parametrize("name", ["vcf/index.vcf", "vcf/index.vcf.gz"])
def test_vcf_count_and_row0(oracle, golden_dir, expected, name):
    # test_vcf_record_scan.test:4-19 and :32-41 (the .gz is BGZF = multi-member gzip)
    r = oracle.vcf_parse(_read(golden_dir, name))
    assert r.error_code == 0
    assert r.n_rows == expected["vcf"]["vcf/index.vcf"]["count"] == 621
    e = expected["vcf"]["row0"]
    assert r.columns["chrom"].row(0).decode() == e["chrom"]
    assert int(r.extra["pos"][0]) == e["pos"]
    assert r.columns["ref"].row(0).decode() == e["ref"]
    assert r.columns["alt"].row(0).decode().split(",") == e["alt"]
    assert float(r.extra["qual"][0]) == e["qual"] and r.extra["qual_valid"][0] == 1
    info = dict(kv.split("=", 1) if "=" in kv else (kv, None) for kv in r.columns["info"].row(0).decode().split(";"))
    assert "INDEL" not in info            # info.indel IS NULL
    assert int(info["DP"]) == e["info.dp"]


def test_vcf_other_fixtures_parse(oracle, golden_dir):
    r = oracle.vcf_parse(_read(golden_dir, "vcf/vcf_file.vcf"))
    assert r.error_code == 0 and r.n_rows > 0
    assert r.columns["alt"].to_list().count(b"T,C") == 1
    r = oracle.vcf_parse(_read(golden_dir, "vcf/vcf_meta_meta.vcf"))
    assert r.error_code == 0 and r.n_rows == 1
    assert r.columns["formats"].row(0) is None and r.extra["qual_valid"][0] == 0


# ---- plumbing restated from rust/src/arrow_reader.rs ---------------------------------------------

def test_compression_inference(oracle):
    # arrow_reader.rs:60-75 (extension) and :77-91 (explicit string, unknown => uncompressed)
    assert oracle.infer_compression("x/test.fastq.gz") == "GZIP"
    assert oracle.infer_compression("x/test.fastq.zst") == "ZSTD"
    assert oracle.infer_compression("x/test.fastq.gzip") == "UNCOMPRESSED"   # only `gz` / `zst` are sniffed
    assert oracle.infer_compression("x/test.fastq.gzip", "gzip") == "GZIP"   # test_fastq_scan.test:17-20
    assert oracle.infer_compression("x/test.fastq.zstd", "zstd") == "ZSTD"
    assert oracle.infer_compression("x/test.fastq", "bogus") == "UNCOMPRESSED"


def test_replacement_scan(oracle):
    # arrow_reader.rs:173-197 + module.cpp:336-375; test_fasta_scan.test:29-32,40-43; test_fastq_scan.test:44-59
    assert oracle.replacement_scan("./t/test.fasta") == "FASTA"
    assert oracle.replacement_scan("./t/test.fasta.gz") == "FASTA"
    assert oracle.replacement_scan("./t/test.fastq.zst") == "FASTQ"
    assert oracle.replacement_scan("./t/index.vcf.gz") == "VCF"
    assert oracle.replacement_scan("./t/table.parquet") is None


# ---- the synthetic generators are self-consistent --------------------------------------------------

def test_synth_fastq_layout(oracle):
    n = 332 * 64
    buf = oracle.synth_fastq(n)
    r = oracle.fastq_parse(buf)
    assert r.error_code == 0 and r.n_rows == 64
    assert r.columns["name"].row(5) == b"SYN000000000005"
    assert r.columns["description"].row(5) == b"1:N:0:ACGT"
    assert set(np.unique(r.columns["sequence"].values)) <= set(b"ACGT")
    assert (r.columns["sequence"].lengths() == 150).all() and (r.columns["quality_scores"].lengths() == 150).all()
    # any byte range can be generated independently (shards)
    part = oracle.synth_fastq(1000, file_offset=777)
    assert (part == buf[777:1777]).all()
