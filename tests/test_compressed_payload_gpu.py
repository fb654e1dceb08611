"""How the decoded bytes of a COMPRESSED input reach the host when DataChunks are asked for (round 5):

* all string columns: the producer sends every decoded segment to the host as it hands it over (HostMirror,
  exg_rd_source.hpp) — the batch's strings point into that block; only the tail carried over from the segment before is
  copied behind the scan;
* a projection that leaves payload-bearing columns out: the decoded bytes stay in HBM, the selected columns' out-of-line
  strings are closed up into a side buffer and only that crosses PCIe (exg_arrow.hip: payload_*_from_col, repoint_strings).

The reference materialises every column from a decoded stream (module.cpp:257-294; rust/src/arrow_reader.rs:60-91): the rows
must be the plain file's rows whatever the route, bit for bit."""
import gzip

import pytest

from test_streaming_gpu import _bgzf

pytestmark = pytest.mark.gpu


def _zstd(data, frame=1 << 20):
    from zstd_util import compress
    return b"".join(compress(data[o:o + frame], 3, o % 2 == 0) for o in range(0, len(data), frame))


def _rows(path, fmt, **kw):
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(str(path), fmt, **kw)
    try:
        return r.rows()
    finally:
        r.close()


@pytest.fixture(scope="module")
def ragged(oracle):
    # unpadded names, descriptions missing every eighth record (NULL), CRLF every 1024th, no final newline
    data = bytes(oracle.synth_fastq_ragged(60000)) if hasattr(oracle, "synth_fastq_ragged") else None
    if data is None:
        from zstd_util import fastq_text
        data = fastq_text(60000, 7, 20, 180)
    exp = oracle.fastq_parse(data, want_string_t=False)
    cols = [exp.columns[c].to_list() for c in ["name", "description", "sequence", "quality_scores"]]
    return data, list(zip(*cols))


@pytest.mark.parametrize("wrap", ["bgzf", "gzip", "zstd"])
@pytest.mark.parametrize("batch", [1 << 20, 0])
def test_all_columns_of_a_compressed_fastq_through_the_host_mirror(gpu, ragged, tmp_path, monkeypatch, wrap, batch):
    """many small segments (1 MiB device batches: every segment carries a tail into the next, the first segments are pushed before
    the consumer's first call and have no mirror) and the default size"""
    data, want = ragged
    if batch:
        monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(batch))
    p = tmp_path / ("x.fastq." + {"bgzf": "gz", "gzip": "gz", "zstd": "zst"}[wrap])
    p.write_bytes({"bgzf": _bgzf(data, 30000), "gzip": gzip.compress(data, 1, mtime=0), "zstd": _zstd(data)}[wrap])
    got = _rows(p, "fastq")
    assert len(got) == len(want)
    assert got == want
    # with the caller's word that chunks will be pulled (EXG_OPEN_CHUNKS: every segment has a mirror, the first one too)
    assert _rows(p, "fastq", expect_chunks=True) == want
    # ... and without the mirror (the copy behind the scan): the same rows
    monkeypatch.setenv("EXG_NO_HOST_MIRROR", "1")
    assert _rows(p, "fastq") == want


@pytest.mark.parametrize("wrap", ["bgzf", "zstd"])
@pytest.mark.parametrize("cols", [[0], [2], [1], [0, 1], [1, 3], [0, 2, 3]])
def test_projection_of_a_compressed_fastq_brings_only_its_columns(gpu, ragged, tmp_path, monkeypatch, wrap, cols):
    data, want = ragged
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(2 << 20))
    p = tmp_path / ("x.fastq." + ("gz" if wrap == "bgzf" else "zst"))
    p.write_bytes(_bgzf(data, 30000) if wrap == "bgzf" else _zstd(data))
    got = _rows(p, "fastq", columns=cols)
    assert got == [tuple(row[c] for c in cols) for row in want]
    # with a pushed-down predicate (the side buffer is built through the row map)
    sel = [row for row in want if row[2] < b"C"]
    got = _rows(p, "fastq", columns=cols, filters="sequence<'C'")
    assert len(sel) > 100 and got == [tuple(row[c] for c in cols) for row in sel]
    # the route can be switched off: same rows
    monkeypatch.setenv("EXG_NO_PAYLOAD_COMPACT", "1")
    assert _rows(p, "fastq", columns=cols) == [tuple(row[c] for c in cols) for row in want]


def test_projection_of_a_bgzip_vcf_with_long_ref_alleles(gpu, oracle, tmp_path, monkeypatch):
    """chrom, pos, ref of a bgzip VCF: CHROM / REF are usually inlined (<= 12 bytes: nothing but the vectors travels); a long
    deletion's REF is out of line and comes through the side buffer"""
    import random
    rng = random.Random(5)
    hdr = (b"##fileformat=VCFv4.2\n##contig=<ID=chr1_KI270706v1_random>\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
           b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
    lines, want = [], []
    for i in range(30000):
        chrom = b"chr1_KI270706v1_random" if i % 7 == 0 else b"chr1"
        ref = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([1, 1, 1, 2, 12, 13, 40, 300])))
        lines.append(b"%s\t%d\t.\t%s\t%s\t%d\tPASS\tDP=%d\n" % (chrom, 100 + i, ref, b"A", i % 90, i % 50))
        want.append((chrom, 100 + i, ref))
    data = hdr + b"".join(lines)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(512 << 10))
    p = tmp_path / "l.vcf.gz"
    p.write_bytes(_bgzf(data, 20000))
    assert _rows(p, "vcf", columns=[0, 1, 3]) == want
    assert _rows(p, "vcf", columns=[3]) == [(w[2],) for w in want]
    got = _rows(p, "vcf", columns=[0, 3], filters="pos>=20000")
    assert got == [(w[0], w[2]) for w in want if w[1] >= 20000]
    # a nested column in the projection: the whole payload travels (the emitter's views are cut out of the line's text)
    full = _rows(p, "vcf", columns=[0, 3, 4])
    assert [(f[0], f[1]) for f in full] == [(w[0], w[2]) for w in want] and full[0][2] == [b"A"]


@pytest.mark.parametrize("cols", [[0, 1, 3], None])
def test_bgzip_cohort_vcf_under_the_indexed_scan(gpu, oracle, tmp_path, monkeypatch, cols):
    """a bgzip VCF of 4 - 12 kB lines in 1 MiB device batches: the reader switches to EXG_ALGO_FUSED_INDEX (round 5: the rows in a
    kernel of their own) on decoded segments too — with a projection (the side buffer of long REF alleles) and with all columns"""
    import random
    from exon_duckdb_amd import abi
    from exon_duckdb_amd.reader import ShardReader
    rng = random.Random(9)
    hdr = (b"##fileformat=VCFv4.2\n##contig=<ID=chr2>\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
           b"##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\n")
    lines = []
    for i in range(1200):
        ref = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([1, 1, 2, 13, 40])))
        lines.append(b"chr2\t%d\t.\t%s\tT\t%d\tPASS\tDP=%d\tGT" % (500 + i, ref, i % 60, i % 31) + b"\t0|1" * rng.randrange(1000, 3000) + b"\n")
    data = hdr + b"".join(lines)
    p = tmp_path / "c.vcf.gz"
    p.write_bytes(_bgzf(data, 60000))
    plain = tmp_path / "c.vcf"
    plain.write_bytes(data)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(1 << 20))
    kw = {"columns": cols} if cols else {}
    want = _rows(plain, "vcf", **kw)
    assert len(want) == 1200
    r = ShardReader(str(p), "vcf", **kw)
    try:
        got = r.rows()
        algo = r.stats()["scan_algo"]
    finally:
        r.close()
    assert got == want and algo == abi.EXG_ALGO_FUSED_INDEX
