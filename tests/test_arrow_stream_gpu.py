"""`new_reader` (the reference's own FFI entry point, exon/include/rust.hpp:41-46) end to end on the GPU:
the Arrow C stream is imported with pyarrow — as DuckDB's Arrow scan imports the reference's — and every
record batch is compared with the oracle: flat Utf8 columns for FASTA / FASTQ, the nested LIST / STRUCT
schema for VCF (pinned row: test_vcf_record_scan.test:10-19), `filters` predicates as
WTArrowTableFunction::InitGlobal renders them (module.cpp:158-226)."""
import gzip
import math
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nr(gpu):
    from exon_duckdb_amd.arrow import new_reader
    return new_reader


def G(golden_dir, name):
    return os.path.join(golden_dir, name)


def same(a, b):
    """deep equality where NaN == NaN"""
    if isinstance(a, float) and isinstance(b, float):
        return a == b or (math.isnan(a) and math.isnan(b))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return a == b


def fastq_rows(oracle, data):
    res = oracle.fastq_parse(data, want_string_t=False)
    cols = [res.columns[k].to_list() for k in ("name", "description", "sequence", "quality_scores")]
    dec = lambda v: None if v is None else v.decode("utf-8")  # noqa: E731
    return [dict(zip(("name", "description", "sequence", "quality_scores"), map(dec, t))) for t in zip(*cols)]


def fasta_rows(oracle, data):
    res = oracle.fasta_parse(data)
    cols = [res.columns[k].to_list() for k in ("id", "description", "sequence")]
    dec = lambda v: None if v is None else v.decode("utf-8")  # noqa: E731
    return [dict(zip(("id", "description", "sequence"), map(dec, t))) for t in zip(*cols)]


# ---- schema + the reference's own statements -------------------------------------------------------------------------

def test_fastq_schema_and_rows(nr, oracle, golden_dir):
    rdr = nr(G(golden_dir, "test.fastq"), "fastq")
    assert rdr.schema.names == ["name", "description", "sequence", "quality_scores"]
    assert [str(t) for t in rdr.schema.types] == ["string"] * 4
    rows = rdr.read_all().to_pylist()
    assert rows == fastq_rows(oracle, open(G(golden_dir, "test.fastq"), "rb").read())
    assert rows[0] == {"name": "SEQ_ID", "description": "This is a description",
                       "sequence": "GATTTGGGGTTCAAAGCAGTATCGATCAAATAGTAAATCCATTTGTTCAACTCACAGTTT",
                       "quality_scores": "!''*((((***+))%%%++)(%%%%).1***-+*''))**55CCF>>>>>>CCCCCCC65"}
    assert rows[1]["description"] is None


def test_fasta_schema_and_rows(nr, oracle, golden_dir):
    rdr = nr(G(golden_dir, "test.fasta"), "fasta")
    assert rdr.schema.names == ["id", "description", "sequence"]
    rows = rdr.read_all().to_pylist()
    assert rows == fasta_rows(oracle, open(G(golden_dir, "test.fasta"), "rb").read())
    rows = nr(G(golden_dir, "test.mixed-desc.fasta"), "fasta").read_all().to_pylist()
    assert [r["description"] for r in rows].count(None) >= 1


def test_gzip_and_explicit_compression(nr, oracle, golden_dir):
    exp = fastq_rows(oracle, gzip.open(G(golden_dir, "test.fastq.gz")).read())
    assert nr(G(golden_dir, "test.fastq.gz"), "fastq").read_all().to_pylist() == exp
    assert nr(G(golden_dir, "test.fastq.gz"), "fastq", compression="gzip").read_all().to_pylist() == exp
    exp = fasta_rows(oracle, gzip.open(G(golden_dir, "test.fasta.gz")).read())
    assert nr(G(golden_dir, "test.fasta.gz"), "fasta", compression="GZIP").read_all().to_pylist() == exp


def test_directory_listing(nr, golden_dir):
    # SELECT COUNT(*) FROM read_fastq('…/fastq/') -> 4   (test_fastq_scan.test:65-68)
    assert nr(G(golden_dir, "fastq") + "/", "fastq").read_all().num_rows == 4


def test_errors_come_back_as_reader_result_text(nr, golden_dir):
    from exon_duckdb_amd import ExgError
    with pytest.raises(ExgError, match="could not register table"):
        nr("", "fastq")
    with pytest.raises(ExgError, match="could not parse file_format"):
        nr(G(golden_dir, "test.fastq"), "bogus")
    with pytest.raises(ExgError, match="could not execute sql"):
        nr(G(golden_dir, "test.fastq"), "fastq", filters="nope = 'x'")
    with pytest.raises(ExgError, match="could not execute sql"):
        nr(G(golden_dir, "test.fastq"), "fastq", filters="name = ")


def test_parse_error_surfaces_from_get_next(nr, tmp_path):
    import pyarrow as pa
    p = tmp_path / "bad.fastq"
    p.write_bytes(b"@a\nAC\n+\n!!\n@b\nAC\nX\n!!\n")
    rdr = nr(str(p), "fastq")
    got = []
    with pytest.raises((pa.ArrowInvalid, OSError, pa.ArrowException)):
        for b in rdr:
            got.extend(b.to_pylist())
    assert [r["name"] for r in got] == ["a"]


# ---- many record batches, several device batches -------------------------------------------------------------------------

@pytest.mark.parametrize("batch_size", [2048, 64])
def test_fastq_ragged_many_batches(nr, oracle, tmp_path, monkeypatch, batch_size):
    data = bytes(oracle.synth_fastq_ragged(20000))
    (tmp_path / "r.fastq").write_bytes(data)
    exp = fastq_rows(oracle, data)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(1 << 20))
    rdr = nr(str(tmp_path / "r.fastq"), "fastq", batch_size=batch_size)
    rows, sizes = [], []
    for b in rdr:
        sizes.append(b.num_rows)
        rows.extend(b.to_pylist())
    assert max(sizes) <= batch_size and len(sizes) > 5
    assert rows == exp


def test_fasta_long_sequences(nr, oracle, tmp_path):
    # sequences >= 8 KiB take the big-string copy kernel
    recs = [b">chr%d some text\n" % i + b"\n".join([b"ACGTTGCA" * 10] * (150 * (i + 1))) + b"\n" for i in range(6)]
    recs.insert(3, b">tiny\nAC\n")
    data = b"".join(recs)
    (tmp_path / "l.fasta").write_bytes(data)
    rows = nr(str(tmp_path / "l.fasta"), "fasta").read_all().to_pylist()
    assert rows == fasta_rows(oracle, data)
    assert max(len(r["sequence"]) for r in rows) > 8192


def test_fasta_synthetic(nr, oracle, tmp_path):
    data = bytes(oracle.synth_fasta(3000))
    (tmp_path / "s.fasta").write_bytes(data)
    assert nr(str(tmp_path / "s.fasta"), "fasta", batch_size=512).read_all().to_pylist() == fasta_rows(oracle, data)


# ---- VCF: the nested schema --------------------------------------------------------------------------------------------------

def test_vcf_pinned_row(nr, golden_dir):
    # SELECT chrom, pos, ref, alt, qual, info.indel, info.dp FROM read_vcf_file_records('…/vcf/index.vcf') LIMIT 1;
    #   -> 1, 9999919, G, [<*>], 0.0, NULL, 1                                      (test_vcf_record_scan.test:10-19)
    for name in ("vcf/index.vcf", "vcf/index.vcf.gz"):
        t = nr(G(golden_dir, name), "vcf").read_all()
        assert t.num_rows == 621
        r = t.slice(0, 1).to_pylist()[0]
        assert (r["chrom"], r["pos"], r["ref"], r["alt"], r["qual"], r["info"]["INDEL"], r["info"]["DP"]) == \
               ("1", 9999919, "G", ["<*>"], 0.0, None, 1)


def test_vcf_schema(nr, golden_dir):
    import pyarrow as pa
    s = nr(G(golden_dir, "vcf/vcf_file.vcf"), "vcf").schema
    assert s.names == ["chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "formats"]
    assert s.field("pos").type == pa.int64() and s.field("qual").type == pa.float32()
    for c in ("id", "alt", "filter"):
        assert s.field(c).type == pa.list_(pa.string())
    info = s.field("info").type
    assert [f.name for f in info] == ["TEST", "DP4", "AC", "AN", "INDEL", "STR"]
    assert info.field("TEST").type == pa.int32() and info.field("DP4").type == pa.list_(pa.int32())
    assert info.field("INDEL").type == pa.bool_() and info.field("STR").type == pa.string()
    item = s.field("formats").type.value_type
    assert [f.name for f in item] == ["TT", "GT", "GQ", "DP", "GL"]
    assert item.field("GL").type == pa.list_(pa.float32()) and item.field("GT").type == pa.string()


@pytest.mark.parametrize("name", ["vcf/index.vcf", "vcf/vcf_file.vcf", "vcf/vcf_meta_meta.vcf", "vcf/index.vcf.gz"])
def test_vcf_fixtures_against_oracle(nr, oracle, golden_dir, name):
    raw = open(G(golden_dir, name), "rb").read()
    data = gzip.decompress(raw) if name.endswith(".gz") else raw
    exp, err = oracle.vcf_typed_rows(data)
    assert err is None
    got = nr(G(golden_dir, name), "vcf").read_all().to_pylist()
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert same(g, e), (g, e)


def test_vcf_synthetic_many_batches(nr, oracle, tmp_path, monkeypatch):
    data = bytes(oracle.synth_vcf(30000))
    (tmp_path / "s.vcf").write_bytes(data)
    exp, err = oracle.vcf_typed_rows(data)
    assert err is None
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(256 << 10))
    rows = []
    for b in nr(str(tmp_path / "s.vcf"), "vcf", batch_size=1024):
        assert b.num_rows <= 1024
        rows.extend(b.to_pylist())
    assert len(rows) == len(exp) == 30000
    assert all(same(g, e) for g, e in zip(rows, exp))


HEADER = (b"##fileformat=VCFv4.2\n"
          b"##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
          b"##INFO=<ID=AF,Number=A,Type=Float,Description=\"a, with comma\">\n"
          b"##INFO=<ID=DB,Number=0,Type=Flag,Description=\"f\">\n"
          b"##INFO=<ID=ANN,Number=.,Type=String,Description=\"s\">\n"
          b"##INFO=<ID=CH,Number=1,Type=Character,Description=\"c\">\n"
          b"##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n"
          b"##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"r\">\n"
          b"##FORMAT=<ID=PL,Number=G,Type=Float,Description=\"p\">\n"
          b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\tS3\n")


def test_vcf_typed_edge_cases(nr, oracle, tmp_path):
    lines = [
        b"1\t10\trs1;rs2\tA\tC,G,<DEL>\t1e-3\tq10;s50\tDP=5;AF=0.5,.,1e-2;DB;ANN=a|b,c;CH=x;ZZ=9\tGT:AD:PL\t0/1:1,2:.\t.\t1|1:.,3",
        b"2\t20\t.\tA\t.\t.\t.\t.\tGT\t.\t./.\t0",
        b"3\t30\tx\tAC\tA\t-0\tPASS\tDB;DP=.;AF=.;ANN=.\tAD:GT:XX\t1:0/0:q\t2,3,4\t.",
        b"4\t40\tx\tAC\tA\t7\tPASS\tDP=-12;DP=13;AF=3\tPL:GT\t1.5,2.5e1,-0.125:1\t.:\t.:.",
        b"5\t50\tx\tAC\tA\t7\tPASS\tANN=;DP",
        b"6\t60\tx\tAC\tA\t7\tPASS\tDP=2147483647;AF=inf,NaN,-infinity\tGT",
        # String / Character values are percent-decoded (INFO and samples); ids, alts and filters are not
        b"7\t70\ta%3Bb\tA\t<%41>\t7\tq%31\tANN=a%3Bb,c%2C%25,%zz,%4,100%,%e2%82%ac" + b"x" * 20 + b";CH=%41\tGT:AD\t0%2F1:1\t%7c:2\t" + b"%2e" * 9,
    ]
    data = HEADER + b"\n".join(lines) + b"\n"
    (tmp_path / "e.vcf").write_bytes(data)
    exp, err = oracle.vcf_typed_rows(data)
    assert err is None and len(exp) == 7
    got = nr(str(tmp_path / "e.vcf"), "vcf").read_all().to_pylist()
    assert len(got) == 7
    assert got[6]["info"]["ANN"] == ["a;b", "c,%", "%zz", "%4", "100%", "\u20ac" + "x" * 20] and got[6]["formats"][0]["GT"] == "0/1"
    for g, e in zip(got, exp):
        assert same(g, e), (g, e)
    assert got[0]["info"]["AF"] == [0.5, None, pytest.approx(0.01)] and got[0]["info"]["DB"] is True
    assert got[0]["formats"][1] == {"GT": None, "AD": None, "PL": None}
    assert got[1]["formats"][2]["GT"] == "0" and got[1]["alt"] == [] and got[1]["filter"] == []


def test_vcf_typed_value_error_stops_the_stream(nr, oracle, tmp_path):
    import pyarrow as pa
    lines = [b"1\t10\t.\tA\tC\t1\tPASS\tDP=5", b"1\t11\t.\tA\tC\t1\tPASS\tDP=5", b"1\t12\t.\tA\tC\t1\tPASS\tDP=five",
             b"1\t13\t.\tA\tC\t1\tPASS\tDP=6"]
    data = HEADER + b"\n".join(lines) + b"\n"
    (tmp_path / "bad.vcf").write_bytes(data)
    exp, err = oracle.vcf_typed_rows(data)
    assert err == 2
    got = []
    with pytest.raises((pa.ArrowInvalid, OSError, pa.ArrowException)):
        for b in nr(str(tmp_path / "bad.vcf"), "vcf"):
            got.extend(b.to_pylist())
    assert [r["pos"] for r in got] == [10, 11]


# ---- filters ------------------------------------------------------------------------------------------------------------------

def test_fasta_filter_from_the_reference_test(nr, golden_dir):
    # SELECT * FROM read_fasta('…/test.fasta') WHERE id = 'a'      (test_fasta_scan.test:34-37; FilterToString -> "id='a'")
    rows = nr(G(golden_dir, "test.fasta"), "fasta", filters="id='a'").read_all().to_pylist()
    assert [r["id"] for r in rows] == ["a"]


FASTQ_FILTERS = [
    ("description IS NULL", lambda r: r["description"] is None),
    ("description IS NOT NULL", lambda r: r["description"] is not None),
    ("name>='r5' AND name<'r7'", lambda r: "r5" <= r["name"] < "r7"),
    ("name='SYNTH_RAGGED_17' OR name='r4242' OR description='3:N:0:ACGT extra words' AND name<'SYNTH_RAGGED_2'",
     lambda r: r["name"] in ("SYNTH_RAGGED_17", "r4242") or
     (r["description"] == "3:N:0:ACGT extra words" and r["name"] < "SYNTH_RAGGED_2")),
    ("description!='d1'", lambda r: r["description"] is not None and r["description"] != "d1"),
    ("description<'d'", lambda r: r["description"] is not None and r["description"] < "d"),
    ("(name='r1' OR description IS NULL) AND sequence>'G'",
     lambda r: (r["name"] == "r1" or r["description"] is None) and r["sequence"] > "G"),
    ("quality_scores<='5' OR sequence<'AC'", lambda r: r["quality_scores"] <= "5" or r["sequence"] < "AC"),
    ("name='it''s'", lambda r: False),
]


@pytest.mark.parametrize("sql,pred", FASTQ_FILTERS)
def test_fastq_filters(nr, oracle, tmp_path, monkeypatch, sql, pred):
    data = bytes(oracle.synth_fastq_ragged(6000))
    (tmp_path / "r.fastq").write_bytes(data)
    exp = [r for r in fastq_rows(oracle, data) if pred(r)]
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(512 << 10))
    rows = nr(str(tmp_path / "r.fastq"), "fastq", filters=sql, batch_size=256).read_all().to_pylist()
    assert rows == exp


VCF_FILTERS = [
    ("chrom='7'", lambda r: r["chrom"] == "7"),
    ("pos>=5000 AND pos<9000", lambda r: 5000 <= r["pos"] < 9000),
    ("pos>9000.5", lambda r: r["pos"] > 9000.5),
    ("qual>50.5", lambda r: r["qual"] is not None and r["qual"] > 50.5),
    ("qual IS NULL OR ref='A' AND pos<=100000", lambda r: r["qual"] is None or (r["ref"] == "A" and r["pos"] <= 100000)),
    ("qual<=10", lambda r: r["qual"] is not None and r["qual"] <= 10),
]


@pytest.mark.parametrize("sql,pred", VCF_FILTERS)
def test_vcf_filters(nr, oracle, tmp_path, monkeypatch, sql, pred):
    data = bytes(oracle.synth_vcf(8000))
    (tmp_path / "s.vcf").write_bytes(data)
    exp = [r for r in oracle.vcf_typed_rows(data)[0] if pred(r)]
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(128 << 10))
    rows = nr(str(tmp_path / "s.vcf"), "vcf", filters=sql).read_all().to_pylist()
    assert len(rows) == len(exp)
    assert all(same(g, e) for g, e in zip(rows, exp))


def test_filter_on_nested_column_is_refused(nr, golden_dir):
    from exon_duckdb_amd import ExgError
    with pytest.raises(ExgError, match="could not execute sql"):
        nr(G(golden_dir, "vcf/index.vcf"), "vcf", filters="alt='A'")


# ---- several devices behind the reference's FFI: the stream fans out by itself (exg_rd_fanout.hpp) ------------------------

@pytest.mark.parametrize("workers", [1, 4])
def test_new_reader_fans_out_over_stripes(nr, oracle, tmp_path, monkeypatch, workers):
    """`new_reader` has no shard argument (rust.hpp:41-46) and the reference's glue pulls the stream from one thread
    (module.cpp:36): the stream cuts the input into stripes itself — forced here: 6 stripes on the one device of the test
    box — reads them through streams of their own on worker threads and hands their record batches out in file order.  The
    rows are those of the unsharded stream, nested VCF columns and a pushed-down filter included."""
    fq = bytes(oracle.synth_fastq_ragged(20000))
    vcf = bytes(oracle.synth_vcf(12000))
    (tmp_path / "r.fastq").write_bytes(fq)
    (tmp_path / "s.vcf").write_bytes(vcf)
    exp_fq = fastq_rows(oracle, fq)
    exp_vcf, err = oracle.vcf_typed_rows(vcf)
    assert err is None
    monkeypatch.setenv("EXON_GPU_SHARDS", "6")
    monkeypatch.setenv("EXG_FANOUT_WORKERS", str(workers))
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(256 << 10))
    rows = []
    for b in nr(str(tmp_path / "r.fastq"), "fastq", batch_size=512):
        assert b.num_rows <= 512
        rows.extend(b.to_pylist())
    assert rows == exp_fq
    rows = []
    for b in nr(str(tmp_path / "s.vcf"), "vcf", batch_size=1024):
        rows.extend(b.to_pylist())
    assert len(rows) == 12000 and all(same(g, e) for g, e in zip(rows, exp_vcf))
    rows = nr(str(tmp_path / "r.fastq"), "fastq", filters="sequence < 'C'").read_all().to_pylist()
    assert rows == [r for r in exp_fq if r["sequence"] < "C"] and rows
