"""The reference's VCF schema AT THE DATACHUNK BOUNDARY (exg_next_chunk and the table functions on top of it): id / alt /
filter LIST(VARCHAR), info STRUCT(<##INFO keys>), formats LIST(STRUCT(<##FORMAT keys>)) in DuckDB's vector layouts
(list_entry_t + child vectors, struct children, validity), built on the device — bit for bit what
oracle.pyoracle.vcf_typed_rows states, which is pinned by test_vcf_record_scan.test:10-19, 32-41
(exon/src/exon/arrow_table_function/module.cpp:126-147 maps the Arrow schema to these DuckDB types)."""
import gzip
import os

import pytest

from test_arrow_stream_gpu import HEADER, same

pytestmark = pytest.mark.gpu


def G(golden_dir, name):
    return os.path.join(golden_dir, name)


def norm(v):
    """chunk-boundary values (bytes) -> the oracle's (str)"""
    if isinstance(v, bytes):
        return v.decode("utf-8")
    if isinstance(v, list):
        return [norm(x) for x in v]
    if isinstance(v, dict):
        return {k: norm(x) for k, x in v.items()}
    return v


def reader_rows(path, batch_rows=2048, **kw):
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(path, "vcf", batch_rows=batch_rows, **kw)
    rows = [dict(zip(r.names, map(norm, t))) for t in r.rows()]
    r.close()
    return rows


def test_schema_is_the_reference_schema(gpu, golden_dir):
    from exon_duckdb_amd.reader import ShardReader
    from exon_duckdb_amd.table_function import type_sql
    r = ShardReader(G(golden_dir, "vcf/vcf_file.vcf"), "vcf")
    assert r.names == ["chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "formats"]
    sql = dict(zip(r.names, map(type_sql, r.trees)))
    assert sql["chrom"] == "VARCHAR" and sql["pos"] == "BIGINT" and sql["qual"] == "FLOAT"
    assert sql["id"] == sql["alt"] == sql["filter"] == "VARCHAR[]"
    assert sql["info"] == "STRUCT(TEST INTEGER, DP4 INTEGER[], AC INTEGER[], AN INTEGER, INDEL BOOLEAN, STR VARCHAR)"
    assert sql["formats"] == "STRUCT(TT INTEGER[], GT VARCHAR, GQ INTEGER, DP INTEGER, GL FLOAT[])[]"


def test_pinned_row_through_the_table_function(gpu, golden_dir):
    # SELECT chrom, pos, ref, alt, qual, info.indel, info.dp FROM read_vcf_file_records('…/vcf/index.vcf') LIMIT 1;
    #   -> 1, 9999919, G, [<*>], 0.0, NULL, 1                                      (test_vcf_record_scan.test:10-19, 32-41)
    from exon_duckdb_amd.table_function import connect
    con = connect()
    for name in ("vcf/index.vcf", "vcf/index.vcf.gz"):
        rel = con.table_function("read_vcf_file_records", G(golden_dir, name))
        chrom, pos, ref, alt, qual, info = rel.fetchall(columns=["chrom", "pos", "ref", "alt", "qual", "info"], limit=1)[0]
        assert (chrom, pos, ref, alt, qual, info["INDEL"], info["DP"]) == (b"1", 9999919, b"G", [b"<*>"], 0.0, None, 1)


@pytest.mark.parametrize("batch_rows", [64, 2048])
@pytest.mark.parametrize("name", ["vcf/index.vcf", "vcf/vcf_file.vcf", "vcf/vcf_meta_meta.vcf", "vcf/index.vcf.gz"])
def test_fixtures_against_oracle(gpu, oracle, golden_dir, name, batch_rows):
    raw = open(G(golden_dir, name), "rb").read()
    data = gzip.decompress(raw) if name.endswith(".gz") else raw
    exp, err = oracle.vcf_typed_rows(data)
    assert err is None
    got = reader_rows(G(golden_dir, name), batch_rows)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert same(g, e), (g, e)


@pytest.mark.parametrize("batch_rows,device_batch", [(64, 64 << 10), (2048, 256 << 10), (2048, 0)])
def test_synthetic_many_batches_and_chunks(gpu, oracle, tmp_path, batch_rows, device_batch):
    data = bytes(oracle.synth_vcf(30000))
    (tmp_path / "s.vcf").write_bytes(data)
    exp, err = oracle.vcf_typed_rows(data)
    assert err is None
    got = reader_rows(str(tmp_path / "s.vcf"), batch_rows, device_batch_bytes=device_batch)
    assert len(got) == len(exp) == 30000
    assert all(same(g, e) for g, e in zip(got, exp))


def edge_lines():
    return [
        b"1\t10\trs1;rs2\tA\tC,G,<DEL>\t1e-3\tq10;s50\tDP=5;AF=0.5,.,1e-2;DB;ANN=a|b,c;CH=x;ZZ=9\tGT:AD:PL\t0/1:1,2:.\t.\t1|1:.,3",
        b"2\t20\t.\tA\t.\t.\t.\t.\tGT\t.\t./.\t0",
        b"3\t30\tx\tAC\tA\t-0\tPASS\tDB;DP=.;AF=.;ANN=.\tAD:GT:XX\t1:0/0:q\t2,3,4\t.",
        b"4\t40\tx\tAC\tA\t7\tPASS\tDP=-12;DP=13;AF=3\tPL:GT\t1.5,2.5e1,-0.125:1\t.:\t.:.",
        b"5\t50\tx\tAC\tA\t7\tPASS\tANN=;DP",
        b"6\t60\tx\tAC\tA\t7\tPASS\tDP=2147483647;AF=inf,NaN,-infinity\tGT",
        # Float values of more than 19 digits astride a rounding boundary: decided exactly (exg_float_slow.hpp)
        b"8\t80\tx\tA\tC\t7\tPASS\tAF=1.00000005960464477539062500000000000000001,0.5,2.00000011920928955078125000000001\tGT:PL\t0:1.0000000596046447753906250000001",
        # String / Character values are percent-decoded (INFO and samples); ids, alts and filters are not
        b"7\t70\ta%3Bb\tA\t<%41>\t7\tq%31\tANN=a%3Bb,c%2C%25,%zz,%4,100%,%e2%82%ac" + b"x" * 20 + b";CH=%41\tGT:AD\t0%2F1:1\t%7c:2\t" + b"%2e" * 9,
    ]


@pytest.mark.parametrize("batch_rows", [64, 2048])
def test_typed_edge_cases(gpu, oracle, tmp_path, batch_rows):
    # the six shapes, repeated so that chunks begin in the middle of words of the children's validity
    lines = edge_lines() * 37
    data = HEADER + b"\n".join(lines) + b"\n"
    (tmp_path / "e.vcf").write_bytes(data)
    exp, err = oracle.vcf_typed_rows(data)
    assert err is None and len(exp) == len(lines)
    got = reader_rows(str(tmp_path / "e.vcf"), batch_rows)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert same(g, e), (g, e)
    assert got[0]["info"]["AF"] == [0.5, None, pytest.approx(0.01)] and got[0]["info"]["DB"] is True
    assert got[0]["formats"][1] == {"GT": None, "AD": None, "PL": None}
    assert got[1]["formats"][2]["GT"] == "0" and got[1]["alt"] == [] and got[1]["filter"] == []
    # long strings in the nested columns are zero-copy pointers into the chunk's payload like the flat ones
    long = b"1\t10\t" + b"r" * 40 + b";" + b"s" * 13 + b"\tA\t<" + b"D" * 50 + b">\t1\tPASS\tANN=" + b"z" * 100 + b",yy\tGT\t" + b"0/1" * 9
    data = HEADER + long + b"\n"
    (tmp_path / "l.vcf").write_bytes(data)
    exp, _ = oracle.vcf_typed_rows(data)
    assert all(same(g, e) for g, e in zip(reader_rows(str(tmp_path / "l.vcf"), batch_rows), exp))
    (tmp_path / "l.vcf.gz").write_bytes(gzip.compress(data))
    assert all(same(g, e) for g, e in zip(reader_rows(str(tmp_path / "l.vcf.gz"), batch_rows), exp))


def test_percent_decoding(gpu, oracle, tmp_path):
    from exon_duckdb_amd import ExgError
    from exon_duckdb_amd.reader import ShardReader
    data = HEADER + edge_lines()[7] + b"\n"
    (tmp_path / "p.vcf").write_bytes(data)
    row = reader_rows(str(tmp_path / "p.vcf"))[0]
    assert row["info"]["ANN"] == ["a;b", "c,%", "%zz", "%4", "100%", "\u20ac" + "x" * 20] and row["info"]["CH"] == "A"
    assert row["id"] == ["a%3Bb"] and row["alt"] == ["<%41>"] and row["filter"] == ["q%31"]
    assert [s["GT"] for s in row["formats"]] == ["0/1", "|", "." * 9]
    # a decoded value that is not UTF-8 is a value error of its row (percent_decode(..).decode_utf8())
    bad = HEADER + b"1\t10\t.\tA\tC\t1\tPASS\tDP=5\n1\t11\t.\tA\tC\t1\tPASS\tANN=ok,%ff\n1\t12\t.\tA\tC\t1\tPASS\tDP=6\n"
    (tmp_path / "b.vcf").write_bytes(bad)
    assert oracle.vcf_typed_rows(bad)[1] == 1
    with pytest.raises(ExgError):
        ShardReader(str(tmp_path / "b.vcf"), "vcf").rows()


def test_typed_value_error_after_the_rows_in_front(gpu, oracle, tmp_path):
    from exon_duckdb_amd import ExgError
    from exon_duckdb_amd.reader import ShardReader
    lines = [b"1\t10\t.\tA\tC\t1\tPASS\tDP=5", b"1\t11\t.\tA\tC\t1\tPASS\tDP=5", b"1\t12\t.\tA\tC\t1\tPASS\tDP=five",
             b"1\t13\t.\tA\tC\t1\tPASS\tDP=6"]
    data = HEADER + b"\n".join(lines) + b"\n"
    (tmp_path / "bad.vcf").write_bytes(data)
    assert oracle.vcf_typed_rows(data)[1] == 2
    r = ShardReader(str(tmp_path / "bad.vcf"), "vcf")
    with pytest.raises(ExgError):
        r.rows()


def test_projection_and_filters(gpu, oracle, tmp_path, monkeypatch):
    from exon_duckdb_amd import ExgError
    from exon_duckdb_amd.table_function import F, connect
    data = bytes(oracle.synth_vcf(8000))
    (tmp_path / "s.vcf").write_bytes(data)
    exp, _ = oracle.vcf_typed_rows(data)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(100 << 10))
    rel = connect().table_function("read_vcf", str(tmp_path / "s.vcf"))
    got = rel.fetchall(columns=["info", "alt", "pos"])
    assert [(norm(i), norm(a), p) for i, a, p in got] == [(e["info"], e["alt"], e["pos"]) for e in exp] or \
        all(same((norm(i), norm(a), p), (e["info"], e["alt"], e["pos"])) for (i, a, p), e in zip(got, exp))
    # a pushed-down filter on flat columns selects rows before the nested columns are built
    want = [e for e in exp if e["chrom"] == "7" and 3000 <= e["pos"] < 9000]
    got = rel.fetchall(columns=["pos", "filter", "info"], filters={"chrom": F.cmp("=", b"7"), "pos": F.and_(F.cmp(">=", 3000), F.cmp("<", 9000))})
    assert len(got) == len(want) > 0
    assert all(same((p, norm(f), norm(i)), (e["pos"], e["filter"], e["info"])) for (p, f, i), e in zip(got, want))
    # nested columns cannot be compared (DuckDB pushes no filter on LIST / STRUCT; new_reader refuses them too)
    with pytest.raises(ExgError, match="nested"):
        rel.fetchall(filters={"info": F.cmp(">=", b"DP=5")})


def test_projection_is_pushed_into_the_reader(gpu, oracle, tmp_path, monkeypatch):
    """exg_open_args.columns (DuckDB's projection_pushdown, module.cpp:310): only the wanted columns' vectors come back — the
    others are NULL in the chunk — with the values of the full scan; every column is still parsed and typed on the device, so
    a malformed INFO value is an error whether INFO is selected or not, like in the reference (which parses everything and
    projects afterwards).  Through the table function a projection reaches the reader as that mask."""
    from exon_duckdb_amd import ExgError
    from exon_duckdb_amd.reader import ShardReader
    from exon_duckdb_amd.table_function import connect
    data = bytes(oracle.synth_vcf(6000))
    p = tmp_path / "p.vcf"
    p.write_bytes(data)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(128 << 10))
    full = ShardReader(str(p), "vcf")
    names = full.names
    all_rows = full.rows()
    full.close()
    for want in (["chrom", "pos"], ["pos", "qual"], ["info"], ["id", "ref", "formats"], ["alt", "filter"]):
        idx = sorted(names.index(c) for c in want)
        r = ShardReader(str(p), "vcf", columns=idx)
        got = r.rows()     # (asserts that the other columns' vectors are NULL)
        r.close()
        assert len(got) == len(all_rows)
        assert all(same(g, tuple(row[k] for k in idx)) for g, row in zip(got, all_rows)), want
    # the same projections through the DuckDB-shaped boundary, gzip-compressed (the inflated payload travels only for strings)
    import gzip
    pz = tmp_path / "p.vcf.gz"
    pz.write_bytes(gzip.compress(data, 6, mtime=0))
    rel = connect().table_function("read_vcf", str(pz))
    exp, _ = oracle.vcf_typed_rows(data)
    assert rel.fetchall(columns=["pos", "qual"]) == [(e["pos"], e["qual"]) for e in exp] or \
        all(same(g, (e["pos"], e["qual"])) for g, e in zip(rel.fetchall(columns=["pos", "qual"]), exp))
    got = rel.fetchall(columns=["chrom", "ref"])
    assert [(c.decode(), r_.decode()) for c, r_ in got] == [(e["chrom"], e["ref"]) for e in exp]
    # a value that does not parse, in a column that is NOT selected: still the reference's error
    bad = HEADER + b"1\t10\t.\tA\tC\t1\tPASS\tDP=5\n1\t11\t.\tA\tC\t1\tPASS\tDP=five\n"
    pb = tmp_path / "bad.vcf"
    pb.write_bytes(bad)
    r = ShardReader(str(pb), "vcf", columns=[0, 1])
    with pytest.raises(ExgError):
        r.rows()
    r.close()


def test_fastq_and_fasta_projections(gpu, oracle, tmp_path, monkeypatch):
    from exon_duckdb_amd.reader import ShardReader
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(256 << 10))
    fq = bytes(oracle.synth_fastq_ragged(5000))
    fa = bytes(oracle.synth_fasta(300))
    (tmp_path / "a.fastq").write_bytes(fq)
    (tmp_path / "a.fasta").write_bytes(fa)
    import gzip
    (tmp_path / "a.fastq.gz").write_bytes(gzip.compress(fq, 6, mtime=0))
    for path, fmt, ncol in (("a.fastq", "fastq", 4), ("a.fastq.gz", "fastq", 4), ("a.fasta", "fasta", 3)):
        r = ShardReader(str(tmp_path / path), fmt)
        all_rows = r.rows()
        r.close()
        for idx in ([0], [1], [ncol - 1], [0, ncol - 1], [1, 2]):
            r = ShardReader(str(tmp_path / path), fmt, columns=idx)
            got = r.rows()
            r.close()
            assert got == [tuple(row[k] for k in idx) for row in all_rows], (path, idx)


def test_projection_through_a_fan_out(gpu, oracle, tmp_path, monkeypatch):
    """the stripes' readers of a fan-out (exg_open with shard_count = 0) take the projection along"""
    from exon_duckdb_amd.reader import ShardReader
    data = bytes(oracle.synth_vcf(9000))
    p = tmp_path / "f.vcf"
    p.write_bytes(data)
    full = ShardReader(str(p), "vcf")
    all_rows = full.rows()
    full.close()
    monkeypatch.setenv("EXON_GPU_SHARDS", "4")
    monkeypatch.setenv("EXG_FANOUT_WORKERS", "2")
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(96 << 10))
    r = ShardReader(str(p), "vcf", shard_count=0, columns=[1, 3, 7])
    got = r.rows()
    r.close()
    assert len(got) == len(all_rows) and all(same(g, (row[1], row[3], row[7])) for g, row in zip(got, all_rows))
