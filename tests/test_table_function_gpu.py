"""The reference's sqllogictests for the path, statement by statement, against the host-side table
functions (bind / init_global / init_local / scan over the reader-level C-ABI and the HIP kernels).

test/sql/exondb-release-with-deb-info/test_fastq_scan.test, test_fasta_scan.test,
test_vcf_record_scan.test — same fixtures (tests/golden/), same expected values.  Statements on
.gz and .zst inputs are decoded on the device (there is no CPU fallback)."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def con(gpu):
    from exon_duckdb_amd import table_function
    return table_function.connect()


def G(golden_dir, name):
    return os.path.join(golden_dir, name)


# ---- test_fastq_scan.test ---------------------------------------------------------------------------

def test_fastq_count(con, golden_dir):
    # SELECT count(*) FROM read_fastq('…/test.fastq');   -> 2          (:5-8)
    assert con.table_function("read_fastq", G(golden_dir, "test.fastq")).count() == 2


def test_fastq_table_structure(con, golden_dir):
    # SELECT * FROM read_fastq('…/test.fastq') LIMIT 1;                (:35-41)
    rel = con.table_function("read_fastq", G(golden_dir, "test.fastq"))
    assert rel.names == ["name", "description", "sequence", "quality_scores"]
    assert rel.fetchall(limit=1) == [(
        b"SEQ_ID", b"This is a description",
        b"GATTTGGGGTTCAAAGCAGTATCGATCAAATAGTAAATCCATTTGTTCAACTCACAGTTT",
        b"!''*((((***+))%%%++)(%%%%).1***-+*''))**55CCF>>>>>>CCCCCCC65")]
    assert rel.fetchall(columns=["description"]) == [(b"This is a description",), (None,)]


def test_fastq_replacement_scan(con, golden_dir):
    # SELECT count(*) FROM '…/test.fastq';                             (:44-47)
    assert con.from_path(G(golden_dir, "test.fastq")).count() == 2
    assert con.replacement_scan("x/test.fastq.gz") == "read_fastq"     # (:50-53)
    assert con.replacement_scan("x/test.fastq.zst") == "read_fastq"    # (:56-59)


def test_fastq_empty_path_is_an_error(con):
    # statement error: SELECT count(*) FROM read_fastq('');            (:61-62)
    from exon_duckdb_amd import ExgError
    with pytest.raises(ExgError):
        con.table_function("read_fastq", "")


def test_fastq_directory(con, golden_dir):
    # SELECT COUNT(*) FROM read_fastq('…/fastq/') LIMIT 1;  -> 4       (:65-68)
    assert con.table_function("read_fastq", G(golden_dir, "fastq") + "/").count() == 4
    names = con.table_function("read_fastq", G(golden_dir, "fastq")).fetchall(columns=["name"])
    assert names == [(b"SEQ_ID",), (b"SEQ_ID2",)] * 2


@pytest.mark.parametrize("name,compression", [("test.fastq.gz", None), ("test.fastq.gzip", "gzip")])
def test_fastq_gzip(con, golden_dir, name, compression):
    # SELECT count(*) FROM read_fastq('…/test.fastq.gz'); / ('…/test.fastq.gzip', compression='gzip')  -> 2   (:11-20)
    rel = con.table_function("read_fastq", G(golden_dir, name), compression=compression)
    assert rel.count() == 2
    # this fixture inflates to a different file than test.fastq: record 1 has no description
    assert rel.fetchall(columns=["name", "description"]) == [(b"SEQ_ID", None), (b"SEQ_ID2", None)]
    if compression is None:
        assert con.from_path(G(golden_dir, name)).count() == 2                                           # (:50-53)


@pytest.mark.parametrize("name,compression", [("test.fastq.zst", None), ("test.fastq.zstd", "zstd")])
def test_fastq_zstd(con, golden_dir, name, compression):
    # SELECT count(*) FROM read_fastq('…/test.fastq.zst'); / ('…/test.fastq.zstd', compression='zstd')  -> 2   (:22-32)
    rel = con.table_function("read_fastq", G(golden_dir, name), compression=compression)
    assert rel.count() == 2
    # the fixture is test.fastq.gz's text compressed with zstd (265 bytes; record 1 has no description there)
    assert rel.fetchall() == con.table_function("read_fastq", G(golden_dir, "test.fastq.gz")).fetchall()
    assert rel.fetchall(columns=["name", "description"]) == [(b"SEQ_ID", None), (b"SEQ_ID2", None)]
    if compression is None:
        assert con.from_path(G(golden_dir, name)).count() == 2     # SELECT count(*) FROM '…/test.fastq.zst';   (:55-59)


def test_fastq_zstd_without_the_option_is_not_zstd(con, golden_dir):
    # only `gz` / `zst` are sniffed (arrow_reader.rs:71-75): '.zstd' without compression='zstd' is read as text and fails
    from exon_duckdb_amd import ExgError
    with pytest.raises(ExgError):
        con.table_function("read_fastq", G(golden_dir, "test.fastq.zstd")).count()


# ---- test_fasta_scan.test / test_fasta_copy.test -------------------------------------------------------

def test_fasta_count(con, golden_dir):
    assert con.table_function("read_fasta", G(golden_dir, "test.fasta")).count() == 2        # (:5-8)
    assert con.from_path(G(golden_dir, "test.fasta")).count() == 2                            # (:29-32)


def test_fasta_where_id(con, golden_dir):
    # SELECT count(*) FROM '…/test.fasta' WHERE id = 'a';   -> 1       (:34-37)
    rel = con.from_path(G(golden_dir, "test.fasta"))
    assert rel.names == ["id", "description", "sequence"]
    assert len(rel.fetchall(where=lambda r: r["id"] == b"a")) == 1
    assert rel.fetchall() == [(b"a", b"description", b"ATCG"), (b"b", b"description2", b"ATCG")]


def test_fasta_null_description(con, golden_dir):
    # FROM read_fasta(…mixed-desc…) WHERE description IS NULL -> b, NULL, ATCG   (test_fasta_copy.test:75-80)
    rel = con.table_function("read_fasta", G(golden_dir, "test.mixed-desc.fasta"))
    assert rel.fetchall(where=lambda r: r["description"] is None) == [(b"b", None, b"ATCG")]


def test_fasta_empty_path_is_an_error(con):
    from exon_duckdb_amd import ExgError
    with pytest.raises(ExgError):
        con.table_function("read_fasta", "")                                                  # (:51-53)


def test_fasta_gzip(con, golden_dir):
    assert con.table_function("read_fasta", G(golden_dir, "test.fasta.gzip"), compression="gzip").count() == 2   # (:11-14)
    assert con.table_function("read_fasta", G(golden_dir, "test.fasta.gz")).count() == 2                          # (:17-20)
    assert con.from_path(G(golden_dir, "test.fasta.gz")).count() == 2                                             # (:40-43)
    # SELECT COUNT(*) FROM read_fasta('…/fasta/', compression='gzip');  -> 4                                      (:55-59)
    rel = con.table_function("read_fasta", G(golden_dir, "fasta") + "/", compression="gzip")
    assert rel.count() == 4
    assert rel.fetchall() == [(b"a", b"description", b"ATCG"), (b"b", b"description2", b"ATCG")] * 2


def test_fasta_zstd(con, golden_dir):
    assert con.table_function("read_fasta", G(golden_dir, "test.fasta.zstd"), compression="zstd").count() == 2   # (:22-26)
    assert con.table_function("read_fasta", G(golden_dir, "test.fasta.zst")).count() == 2                          # (:46-49)
    assert con.from_path(G(golden_dir, "test.fasta.zst")).count() == 2
    assert con.table_function("read_fasta", G(golden_dir, "test.fasta.zst")).fetchall() == [
        (b"a", b"description", b"ATCG"), (b"b", b"description2", b"ATCG")]


# ---- test_vcf_record_scan.test ---------------------------------------------------------------------------

def test_vcf_count(con, golden_dir):
    # SELECT COUNT(*) FROM read_vcf_file_records('…/vcf/index.vcf');  -> 621     (:4-7)
    assert con.table_function("read_vcf_file_records", G(golden_dir, "vcf/index.vcf")).count() == 621
    assert con.table_function("read_vcf", G(golden_dir, "vcf/index.vcf")).count() == 621     # the north star's name
    assert con.from_path(G(golden_dir, "vcf/index.vcf")).count() == 621


def test_vcf_row0(con, golden_dir):
    # SELECT chrom, pos, ref, alt, qual, info.indel, info.dp … LIMIT 1            (:10-19)
    rel = con.table_function("read_vcf_file_records", G(golden_dir, "vcf/index.vcf"))
    assert rel.names == ["chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "formats"]
    chrom, pos, ref, alt, qual, info = rel.fetchall(columns=["chrom", "pos", "ref", "alt", "qual", "info"], limit=1)[0]
    # alt is a LIST, info a STRUCT: -> 1, 9999919, G, [<*>], 0.0, NULL, 1
    assert (chrom, pos, ref, alt, qual, info["INDEL"], info["DP"]) == (b"1", 9999919, b"G", [b"<*>"], 0.0, None, 1)


def test_vcf_bgzf(con, golden_dir):
    # same statements on '…/vcf/index.vcf.gz' (BGZF = multi-member gzip)                         (:32-41)
    rel = con.table_function("read_vcf_file_records", G(golden_dir, "vcf/index.vcf.gz"))
    assert rel.count() == 621
    chrom, pos, ref, alt, qual = rel.fetchall(columns=["chrom", "pos", "ref", "alt", "qual"], limit=1)[0]
    assert (chrom, pos, ref, alt, qual) == (b"1", 9999919, b"G", [b"<*>"], 0.0)
    plain = con.table_function("read_vcf_file_records", G(golden_dir, "vcf/index.vcf")).fetchall()
    assert rel.fetchall() == plain


def test_vcf_nulls(con, golden_dir):
    rows = con.table_function("read_vcf_file_records", G(golden_dir, "vcf/vcf_meta_meta.vcf")).fetchall()
    # id [test], alt [T], QUAL '.' -> NULL, FILTER '.' -> [], INFO '.' -> a struct of NULLs, no FORMAT column -> no samples
    assert len(rows) == 1 and rows[0][:7] == (b"1", 123, [b"test"], b"TC", [b"T"], None, [])
    assert rows[0][7] is not None and all(v is None for v in rows[0][7].values()) and rows[0][8] == []


# ---- chunking: the reference asks for STANDARD_VECTOR_SIZE rows per batch (module.cpp:83) ----------------------

def test_chunks_are_2048_rows(con, oracle, tmp_path):
    p = tmp_path / "many.fastq"
    p.write_bytes(oracle.synth_fastq(332 * 5000).tobytes())
    rel = con.table_function("read_fastq", str(p))
    assert rel.chunk_sizes() == [2048, 2048, 904]
    assert rel.count() == 5000
    rows = rel.fetchall(columns=["name", "description"])
    assert rows[4999] == (b"SYN000000004999", b"3:N:0:ACGT")


def test_parse_error_surfaces_after_the_good_rows(con, tmp_path):
    from exon_duckdb_amd import ExgError
    p = tmp_path / "bad.fastq"
    p.write_bytes(b"@x\nAC\n+\n!!\n" * 3000 + b"oops\nAC\n+\n!!\n")
    rel = con.table_function("read_fastq", str(p))
    with pytest.raises(ExgError, match="invalid name prefix"):
        rel.count()
    with pytest.raises(ExgError, match="invalid name prefix"):
        rel.fetchall()


def test_streaming_batches_match_one_batch(con, oracle, tmp_path, monkeypatch):
    # several record-aligned device batches must give the same rows as one
    data = bytes(oracle.synth_fastq_ragged(3000))
    p = tmp_path / "ragged.fastq"
    p.write_bytes(data)
    exp = oracle.fastq_parse(data)
    rows = con.table_function("read_fastq", str(p)).fetchall()
    assert len(rows) == 3000
    for i in (0, 1, 2999):
        assert rows[i] == tuple(exp.columns[k].row(i) for k in exp.columns)


def test_gzip_streams_through_several_batches(con, oracle, tmp_path, monkeypatch):
    # BGZF-framed and plain single-member gzip of the same FASTQ, scanned in place on the device in
    # record-aligned batches (batch starts are only byte aligned there)
    import gzip
    import struct
    import zlib

    data = bytes(oracle.synth_fastq_ragged(6000))
    exp = oracle.fastq_parse(data)
    blocks = []
    for i in range(0, len(data), 65280):
        chunk = data[i:i + 65280]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = co.compress(chunk) + co.flush()
        blocks.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" +
                      struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1) + raw + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    (tmp_path / "bgzf.fastq.gz").write_bytes(b"".join(blocks))
    (tmp_path / "plain.fastq.gz").write_bytes(gzip.compress(data, mtime=0))
    (tmp_path / "two_members.fastq.gz").write_bytes(gzip.compress(data[:len(data) // 2 + 7], mtime=0) +
                                                    gzip.compress(data[len(data) // 2 + 7:], mtime=0))
    (tmp_path / "plain.fastq").write_bytes(data)
    want = [tuple(exp.columns[k].row(i) for k in exp.columns) for i in range(exp.n_rows)]
    for batch in (None, "65536", "4096"):
        if batch:
            monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", batch)
        for name in ["bgzf.fastq.gz", "plain.fastq.gz", "two_members.fastq.gz", "plain.fastq"]:
            rel = con.table_function("read_fastq", str(tmp_path / name))
            assert rel.count() == 6000, (name, batch)
            assert rel.fetchall() == want, (name, batch)


def test_vcf_streams_through_several_batches(con, oracle, tmp_path, monkeypatch):
    import gzip
    data = bytes(oracle.synth_vcf(5000))
    exp = oracle.vcf_parse(data)
    typed, _ = oracle.vcf_typed_rows(data)
    (tmp_path / "v.vcf").write_bytes(data)
    (tmp_path / "v.vcf.gz").write_bytes(gzip.compress(data, mtime=0))
    want_pos = exp.extra["pos"].tolist()
    for batch in (None, "32768"):
        if batch:
            monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", batch)
        for name in ["v.vcf", "v.vcf.gz"]:
            rel = con.table_function("read_vcf", str(tmp_path / name))
            assert rel.count() == 5000
            rows = rel.fetchall(columns=["chrom", "pos", "info"])
            assert [r[1] for r in rows] == want_pos
            assert {k: (v.decode() if isinstance(v, bytes) else v) for k, v in rows[4999][2].items()}.keys() == typed[4999]["info"].keys()
            assert rows[4999][2]["DP"] == typed[4999]["info"]["DP"]


def test_prefetch_miss_and_batch_growth(con, oracle, tmp_path, monkeypatch):
    """Records far longer than the prefetch slack and than the device batch itself: the reader has to fall
    back to synchronous uploads and to double the batch until a whole record fits; rows must not change."""
    import random
    rnd = random.Random(5)
    recs = []
    for i in range(40):
        ln = rnd.choice([10, 150, 30_000, 90_000, 200_000])
        seq = "".join(rnd.choice("ACGT") for _ in range(64)) * (ln // 64 + 1)
        recs.append(f"@r{i} d{i}\n{seq[:ln]}\n+\n{'I' * ln}\n")
    data = "".join(recs).encode()
    (tmp_path / "long.fastq").write_bytes(data)
    exp = oracle.fastq_parse(data, want_string_t=False)
    want = list(zip(*[exp.columns[k].to_list() for k in ("name", "description", "sequence", "quality_scores")]))
    for batch in ("65536", "262144", None):
        if batch:
            monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", batch)
        else:
            monkeypatch.delenv("EXG_DEVICE_BATCH_BYTES")
        rel = con.table_function("read_fastq", str(tmp_path / "long.fastq"))
        assert rel.count() == 40
        assert rel.fetchall() == want


# ---- filter pushdown (scan.filter_pushdown = true, module.cpp:311): TableFilterSet -> FilterToString -> device ------

def test_filter_pushdown_reference_statement(con, golden_dir):
    # SELECT * FROM read_fasta('…/test.fasta') WHERE id = 'a'                      (test_fasta_scan.test:34-37)
    from exon_duckdb_amd.table_function import F
    rel = con.table_function("read_fasta", G(golden_dir, "test.fasta"))
    rows = rel.fetchall(filters={"id": F.cmp("=", "a")})
    assert [r[0] for r in rows] == [b"a"]
    assert rel.count(filters={"id": F.cmp("=", "a")}) == 1
    assert rel.count(filters={"id": F.cmp("!=", "a")}) == 1


def _fq_cases():
    from exon_duckdb_amd.table_function import F
    return [
        ({"description": F.isnull()}, lambda r: r["description"] is None),
        ({"description": F.notnull(), "name": F.cmp(">", b"r5")}, lambda r: r["description"] is not None and r["name"] > b"r5"),
        ({"name": F.or_(F.cmp("=", b"SYNTH_RAGGED_17"), F.cmp("=", b"r4242"), F.and_(F.cmp(">=", b"r10"), F.cmp("<", b"r11")))},
         lambda r: r["name"] in (b"SYNTH_RAGGED_17", b"r4242") or b"r10" <= r["name"] < b"r11"),
        ({"sequence": F.cmp("<", b"AC"), "quality_scores": F.cmp(">=", b"5")}, lambda r: r["sequence"] < b"AC" and r["quality_scores"] >= b"5"),
        ({"name": F.cmp("=", b"it's")}, lambda r: False),
    ]


@pytest.mark.parametrize("case", range(5))
def test_filter_pushdown_fastq(con, oracle, tmp_path, monkeypatch, case):
    filters, pred = _fq_cases()[case]
    data = bytes(oracle.synth_fastq_ragged(6000))
    (tmp_path / "r.fastq").write_bytes(data)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(300 << 10))
    rel = con.table_function("read_fastq", str(tmp_path / "r.fastq"))
    want = rel.fetchall(where=pred)                      # the same predicate applied above an unfiltered scan
    assert rel.fetchall(filters=filters) == want
    assert rel.count(filters=filters) == len(want)
    assert rel.fetchall(columns=["sequence"], filters=filters) == [(r[2],) for r in want]


def test_filter_pushdown_vcf(con, oracle, tmp_path, monkeypatch):
    from exon_duckdb_amd.table_function import F
    data = bytes(oracle.synth_vcf(8000))
    (tmp_path / "s.vcf").write_bytes(data)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(100 << 10))
    rel = con.table_function("read_vcf", str(tmp_path / "s.vcf"))
    cases = [
        ({"chrom": F.cmp("=", b"7"), "pos": F.and_(F.cmp(">=", 3000), F.cmp("<", 9000))},
         lambda r: r["chrom"] == b"7" and 3000 <= r["pos"] < 9000),
        ({"qual": F.or_(F.isnull(), F.cmp(">", 900.5))}, lambda r: r["qual"] is None or r["qual"] > 900.5),
        ({"ref": F.cmp(">=", b"C"), "chrom": F.cmp("!=", b"1")}, lambda r: r["ref"] >= b"C" and r["chrom"] != b"1"),
    ]
    for filters, pred in cases:
        want = rel.fetchall(where=pred)
        assert len(want) > 0
        assert rel.fetchall(filters=filters) == want
        assert rel.count(filters=filters) == len(want)


def test_filter_on_unknown_column_is_a_bind_time_error(gpu, golden_dir):
    import ctypes as C
    from exon_duckdb_amd.table_function import _lib  # noqa: F401  (binds the library)
    from exon_duckdb_amd import load_library

    from exon_duckdb_amd.abi import OpenArgs
    lib = load_library()
    a = OpenArgs(G(golden_dir, "test.fastq").encode(), b"fastq", None, 2048, 0, 0, b"nope='x'", 0, 0)
    r = C.c_void_p()
    assert lib.exg_open(C.byref(a), C.byref(r)) != 0
    assert b"could not execute sql" in lib.exg_last_error_message()


def test_gzip_vcf_with_a_header_longer_than_the_first_host_prefix(con, oracle, tmp_path, monkeypatch):
    # gzip + VCF: only the header prefix of the inflated bytes comes back to the host (4 MiB first, then more);
    # the rows' payload travels batch by batch
    import gzip

    body = bytes(oracle.synth_vcf(3000))
    lines = body.split(b"\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith(b"#CHROM"))
    filler = b"".join(b"##contig=<ID=scaffold_%07d,length=%d>\n" % (i, 1000 + i) for i in range(120_000))   # ~5.6 MB
    assert len(filler) > (4 << 20)
    data = b"\n".join(lines[:k]) + b"\n" + filler + b"\n".join(lines[k:])
    exp = oracle.vcf_parse(data)
    assert exp.error_code == 0 and exp.n_rows == 3000
    (tmp_path / "long_header.vcf.gz").write_bytes(gzip.compress(data, 6, mtime=0))
    (tmp_path / "long_header.vcf").write_bytes(data)
    for batch in (None, "65536"):
        if batch:
            monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", batch)
        plain = con.table_function("read_vcf", str(tmp_path / "long_header.vcf")).fetchall()
        rel = con.table_function("read_vcf", str(tmp_path / "long_header.vcf.gz"))
        assert rel.count() == 3000
        assert rel.fetchall() == plain
    assert plain[0][0] == exp.columns["chrom"].row(0)


# ---- several scan threads inside one process: init_global plans shards, every init_local opens its shard on its device ----

def test_small_inputs_scan_on_one_thread(con, golden_dir):
    rel = con.table_function("read_fastq", G(golden_dir, "test.fastq"))
    assert rel.count() == 2 and rel.last_max_threads == 1      # MaxThreads() == 1: nothing to shard


@pytest.mark.parametrize("n_shards", [2, 5])
def test_parallel_scan_threads_partition_the_rows(con, oracle, tmp_path, monkeypatch, n_shards):
    """EXON_GPU_SHARDS forces what a multi-GPU box plans by itself (one shard per device): MaxThreads() scan threads, each
    with its own reader on its own byte range; together they return the unsharded rows, in file order by batch index,
    and COUNT(*) is the sum of the shards' counts."""
    import gzip
    fq = bytes(oracle.synth_fastq(332 * 30000))
    vcf = bytes(oracle.synth_vcf(20000))
    fa = bytes(oracle.synth_fasta(3000, seed=77))
    (tmp_path / "p.fastq").write_bytes(fq)
    (tmp_path / "p.vcf").write_bytes(vcf)
    (tmp_path / "p.fasta").write_bytes(fa)
    (tmp_path / "one.fastq.gz").write_bytes(gzip.compress(fq[:332 * 3000], mtime=0))   # not BGZF: one shard
    want = {}
    for name, fn in (("p.fastq", "read_fastq"), ("p.vcf", "read_vcf"), ("p.fasta", "read_fasta")):
        rel = con.table_function(fn, str(tmp_path / name))
        want[name] = rel.fetchall()
        assert rel.last_max_threads == 1
    monkeypatch.setenv("EXON_GPU_SHARDS", str(n_shards))
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(1 << 20))
    for name, fn in (("p.fastq", "read_fastq"), ("p.vcf", "read_vcf"), ("p.fasta", "read_fasta")):
        rel = con.table_function(fn, str(tmp_path / name))
        rows = rel.fetchall()
        assert rel.last_max_threads == n_shards
        assert rows == want[name], name
        assert rel.count() == len(want[name])
    rel = con.table_function("read_fastq", str(tmp_path / "one.fastq.gz"))
    assert rel.count() == 3000 and rel.last_max_threads == 1
    # a pushed-down filter runs in every shard
    from exon_duckdb_amd.table_function import F
    rel = con.table_function("read_vcf", str(tmp_path / "p.vcf"))
    got = rel.fetchall(columns=["chrom", "pos"], filters={"chrom": F.cmp("=", b"7")})
    assert got == [(c, p) for c, p, *_ in want["p.vcf"] if c == b"7"] and len(got) > 0


def test_parallel_scan_of_bgzf(con, oracle, tmp_path, monkeypatch):
    from test_reader_shards_gpu import _bgzf as bgzf
    fq = bytes(oracle.synth_fastq(332 * 20000))
    (tmp_path / "b.fastq.gz").write_bytes(bgzf(fq, 65280))
    rel = con.table_function("read_fastq", str(tmp_path / "b.fastq.gz"))
    want = rel.fetchall()
    assert len(want) == 20000
    monkeypatch.setenv("EXON_GPU_SHARDS", "4")
    rel = con.table_function("read_fastq", str(tmp_path / "b.fastq.gz"))
    assert rel.fetchall() == want and rel.last_max_threads == 4
    assert rel.count() == 20000


@pytest.mark.parametrize("scan_threads", [1, 2, 3])
def test_fewer_scan_threads_than_shards(con, oracle, tmp_path, monkeypatch, scan_threads):
    """MaxThreads() is an upper bound: DuckDB runs fewer scan threads under `SET threads=1`, with threads < GPUs, or
    when the sink is not parallel (Pipeline::ScheduleParallel).  Every shard must still be scanned — a thread whose
    reader is exhausted claims the next unclaimed shard — with the rows in file order by batch index, and every device
    batch one DuckDB batch (its chunks share an index; indices never decrease on a thread)."""
    from test_reader_shards_gpu import _bgzf as bgzf
    fq = bytes(oracle.synth_fastq(332 * 30000))
    vcf = bytes(oracle.synth_vcf(20000))
    (tmp_path / "p.fastq").write_bytes(fq)
    (tmp_path / "p.vcf").write_bytes(vcf)
    (tmp_path / "b.fastq.gz").write_bytes(bgzf(fq, 65280))
    cases = (("p.fastq", "read_fastq"), ("p.vcf", "read_vcf"), ("b.fastq.gz", "read_fastq"))
    want = {name: con.table_function(fn, str(tmp_path / name)).fetchall() for name, fn in cases}
    monkeypatch.setenv("EXON_GPU_SHARDS", "5")
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(1 << 20))
    for name, fn in cases:
        rel = con.table_function(fn, str(tmp_path / name))
        rel.scan_threads = scan_threads
        rows = rel.fetchall()
        assert rel.last_max_threads == 5
        assert rows == want[name], name
        assert rel.count() == len(want[name])
        idx = [b for b, _, _ in rel._scan([0])]
        assert idx == sorted(idx) and max(idx) < 10 ** 13
        assert len(set(idx)) < len(idx), "a device batch of several chunks is one batch"


def test_plan_shards_policy(gpu, golden_dir, tmp_path, monkeypatch):
    import ctypes as C
    from exon_duckdb_amd.abi import OpenArgs
    gpu.exg_plan_shards.argtypes = [C.POINTER(OpenArgs), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.c_uint32]

    def plan(path, fmt, compression=None):
        a = OpenArgs(path.encode(), fmt.encode(), compression, 2048, 0, 0, None, 0, 0)
        n = C.c_uint32(0)
        dev = (C.c_int * 64)()
        assert gpu.exg_plan_shards(C.byref(a), C.byref(n), dev, 64) == 0
        return n.value, list(dev[:n.value])

    assert plan(G(golden_dir, "test.fastq"), "fastq") == (1, [0])
    monkeypatch.setenv("EXON_GPU_SHARDS", "3")
    n_dev = gpu.exg_device_count()
    assert plan(G(golden_dir, "test.fastq"), "fastq") == (3, [i % n_dev for i in range(3)])
    assert plan(G(golden_dir, "vcf/index.vcf.gz"), "vcf")[0] == 3            # BGZF: members carry their size
    assert plan(G(golden_dir, "test.fastq.gz"), "fastq")[0] == 1             # plain gzip: not sharded
    assert plan(G(golden_dir, "test.fastq.zst"), "fastq")[0] == 1            # zstd: not sharded
    assert plan(G(golden_dir, "test.fasta.gz"), "fasta")[0] == 1


def test_bgzf_of_many_windows_in_flight(gpu, tmp_path, monkeypatch):
    """A BGZF file of several hundred MB — the input builder of bench.py's config 4: members deflated straight from the
    generator by a pool of processes — read as a stream of segments, three windows of members in flight (and one, two:
    EXG_GZ_LANES): COUNT(*), every row's content (the digest of the scaffolding library against the generator's), the
    first / last rows."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from exon_duckdb_amd import abi, load_library, load_test_library
    from exon_duckdb_amd.reader import ShardReader
    n_rec = 16320 * 110                                 # 596 MB of FASTQ-150 -> ~310 MB of BGZF
    gz = str(tmp_path / "big.fastq.gz")
    comp = bench.build_bgzf(abi.EXG_SYNTH_FASTQ_SEED, n_rec, gz, max(1, min(bench.effective_cores(), 64)))
    assert comp > (256 << 20), comp
    want = int(load_test_library().exon_tf_expect_fastq150(abi.EXG_SYNTH_FASTQ_SEED, 0, n_rec, 8))
    lib = load_library()
    for lanes in ("3", "1", "2"):
        monkeypatch.setenv("EXG_GZ_LANES", lanes)
        rows, chunks, got, bad = bench.reader_digest(lib, gz, "fastq", 150)
        assert rows == n_rec and got == want and bad == 0, lanes
        assert ShardReader(gz, "fastq").count() == n_rec
    rd = ShardReader(gz, "fastq", device_batch_bytes=64 << 20)
    rows = rd.rows()
    rd.close()
    assert len(rows) == n_rec and rows[0][0] == b"SYN000000000000" and rows[-1][0] == b"SYN%012d" % (n_rec - 1)


def test_cardinality_estimate(con, golden_dir, oracle, tmp_path):
    """TableFunction::cardinality (module.cpp:307).  The reference registers ArrowScanCardinality, which has no estimate to give
    (an opaque Arrow stream); this glue knows the input's size on disk and estimates rows from it — within a small factor of
    the truth on the usual record shapes, 0 (no estimate, like the reference) only when the size is unknown."""
    from exon_duckdb_amd.table_function import Relation
    data = bytes(oracle.synth_fastq(332 * 50000))
    p = tmp_path / "c.fastq"
    p.write_bytes(data)
    est = Relation("read_fastq", str(p)).estimated_cardinality
    assert 50000 / 2 <= est <= 50000 * 2, est
    import gzip
    pz = tmp_path / "c.fastq.gz"
    pz.write_bytes(gzip.compress(data, 6))
    est = Relation("read_fastq", str(pz)).estimated_cardinality
    assert 50000 / 3 <= est <= 50000 * 3, est
    est = Relation("read_vcf_file_records", os.path.join(golden_dir, "vcf", "index.vcf")).estimated_cardinality
    assert 621 / 3 <= est <= 621 * 3, est
    assert Relation("read_fasta", os.path.join(golden_dir, "test.fasta")).estimated_cardinality >= 1


def test_quality_score_string_to_list_scalar(gpu):
    """quality_score_string_to_list as `LOAD exon` registers it in SQL (exon_extension.cpp:60, fastq_functions/module.cpp:28-54:
    `c - 33` per byte with `char` signed): the shim's host arithmetic against the closed form, and against the device op on
    a column in HBM (tests/test_quality_list_gpu.py checks that one against the oracle)."""
    from exon_duckdb_amd.table_function import quality_score_string_to_list
    assert quality_score_string_to_list(b"!I5@") == [0, 40, 20, 31]
    assert quality_score_string_to_list(b"") == []
    s = bytes(range(256))
    assert quality_score_string_to_list(s) == [(c - 256 if c >= 128 else c) - 33 for c in s]
