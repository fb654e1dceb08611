"""Record shapes outside the synthetic 150 bp FASTQ / 49-byte VCF line, through the C-ABI against the oracle.

The reference's line readers run at one rate whatever the record length (noodles, reached at
rust/src/arrow_reader.rs:116-153).  The fused kernels stage 16 KiB halves with a 1 KiB window in LDS; these tests pin
that nothing but the bytes' content decides the output — long reads (HiFi-like 15 kb, ONT-like 1-100 kb, a 3 MB read that
spans dozens of super-tiles), short reads (36 bp and shorter: more lines per half than the LDS list holds, emitted in
passes), multi-sample VCF lines (100 and 2 504 samples) — and that NO launch gives up (`EXG_RF_FALLBACK` clear) with the
fused kernel alone.  Bit-exact: every column vector, validity word and the result block.
"""
import numpy as np
import pytest

from exon_duckdb_amd import abi
from exon_duckdb_amd.testing.shapes import fastq_records, vcf_lines

from test_fastq_gpu import BASE as FQ_BASE, NAMES, check_against_oracle, run_gpu as run_fastq
from test_vcf_gpu import HDR, check as check_vcf, header_bytes, run_gpu as run_vcf

pytestmark = pytest.mark.gpu

FUSED_AND_PARTNER = [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_FUSED_FULL, abi.EXG_ALGO_MULTIPASS]


def no_fallback(res):
    assert not (res.flags & abi.EXG_RF_FALLBACK), "the fused kernel gave the launch up"


def test_the_lean_scan_says_when_it_marked_tiles(gpu, oracle):
    """EXG_RF_REDO: the lean scan marked super-tiles and the any-shape run redid them (complete output; a reader with more
    batches of the input switches to EXG_ALGO_FUSED_FULL).  Clear on the synthetic 150 bp shape and under FUSED_FULL."""
    short = bytes(oracle.synth_fastq(332 * 3000))
    long_ = fastq_records([15000] * 20, seed=2)
    tiny = b"@a\n\n+\n\n" * 6000
    for data, redo in ((short, False), (long_, True), (tiny, True), (short + long_ + short, True)):
        res, _, _ = run_fastq(data, abi.EXG_ALGO_FUSED)
        assert bool(res.flags & abi.EXG_RF_REDO) == redo and res.error_code == 0
        res, _, _ = run_fastq(data, abi.EXG_ALGO_FUSED_FULL)
        assert not (res.flags & abi.EXG_RF_REDO) and res.error_code == 0
    res, _ = run_vcf(vcf_lines(40, 2504, seed=1), abi.EXG_ALGO_FUSED)
    assert res.flags & abi.EXG_RF_REDO
    res, _ = run_vcf(vcf_lines(400, 3, seed=1), abi.EXG_ALGO_FUSED)
    assert not (res.flags & abi.EXG_RF_REDO)


# ---- long reads ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
@pytest.mark.parametrize("shape", ["hifi", "ont", "at_window", "mixed_short_long"])
def test_long_reads(gpu, oracle, shape, algo):
    rng = np.random.default_rng(11)
    if shape == "hifi":          # ~15 kb +- 3 kb
        lengths = np.clip(rng.normal(15000, 3000, 60), 2000, 40000)
    elif shape == "ont":         # log-uniform 1 kb .. 100 kb
        lengths = np.exp(rng.uniform(np.log(1000), np.log(100000), 48))
    elif shape == "at_window":   # record sizes around the 1 KiB window and the 16 KiB half
        lengths = [470, 480, 490, 500, 505, 510, 515, 520, 1000, 1020, 1030, 8180, 8190, 8200, 16380, 16390, 24570, 24580] * 3
    else:                        # one long read among short ones costs nothing but itself
        lengths = [150] * 300 + [20000] + [150] * 300 + [70000, 36, 36, 150000] + [150] * 200
    data = fastq_records(lengths, seed=5, crlf_every=7)
    res = check_against_oracle(oracle, data, algo)
    assert res.n_records == len(lengths)
    no_fallback(res)


@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
def test_a_read_that_spans_dozens_of_super_tiles(gpu, oracle, algo):
    # 3 MB and 1.2 MB reads: the look-back walks over ~130 tiles without a newline
    data = fastq_records([100, 3_000_000, 150, 1_200_000, 1, 0, 150], seed=9)
    res = check_against_oracle(oracle, data, algo)
    assert res.n_records == 7
    no_fallback(res)


@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
@pytest.mark.parametrize("case", ["bad_at", "bad_plus", "empty_plus", "truncated_in_qual", "truncated_after_plus", "no_final_newline",
                                  "cr_at_eof", "long_name_line", "first_error_wins"])
def test_long_read_rules(gpu, oracle, case, algo):
    good = fastq_records([30000, 200, 45000], seed=3)
    seq = b"ACGT" * 9000
    if case == "bad_at":
        data = good + b"xlong\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n" + good
    elif case == "bad_plus":
        data = good + b"@long\n" + seq + b"\n-\n" + b"I" * len(seq) + b"\n" + good
    elif case == "empty_plus":
        data = good + b"@long\n" + seq + b"\n\n" + b"I" * len(seq) + b"\n"
    elif case == "truncated_in_qual":
        data = good + b"@long\n" + seq + b"\n+\n" + b"I" * 20000       # quality line unterminated at EOF: still a record
    elif case == "truncated_after_plus":
        data = good + b"@long\n" + seq + b"\n+\n"                      # missing quality line => empty
    elif case == "no_final_newline":
        data = good[:-1]
    elif case == "cr_at_eof":
        data = good[:-1] + b"\r"
    elif case == "long_name_line":                                      # a 40 kB name line with its first space far in
        data = good + b"@" + b"n" * 25000 + b" " + b"d" * 15000 + b"\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n" + good
    else:
        data = good + b"@a\n" + seq + b"\n-\n" + b"I" * len(seq) + b"\nzz\n" + seq + b"\n+\n" + b"I" * len(seq) + b"\n"
    res = check_against_oracle(oracle, data, algo)
    no_fallback(res)


@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
def test_long_reads_in_unaligned_shards(gpu, oracle, algo):
    """byte-range shards with a halo over long reads: rows of the shards = rows of the file; a shard whose halo does not reach
    the beginning of its first record says EXG_RF_HEAD_UNRESOLVED (the reader then widens the halo)"""
    lengths = [12000, 150, 33000, 150, 150, 9000, 70000, 150, 20000, 5000, 150, 41000]
    data = bytes(fastq_records(lengths, seed=21))
    exp = oracle.fastq_parse(data, payload_base=FQ_BASE)
    arr = np.frombuffer(data, np.uint8)
    nl = np.flatnonzero(arr == 10)
    n = len(data)
    cuts = [0] + [16 * (n * k // 5 // 16) for k in range(1, 5)] + [n]
    halo = 160 * 1024
    got_cols = [[] for _ in range(4)]
    total = 0
    for s, e in zip(cuts[:-1], cuts[1:]):
        h = min(halo, s)
        h -= h % 16
        fli = int(np.searchsorted(nl, s))             # '\n' in front of the shard's first byte (buffer offset `lead`)
        flags = (abi.EXG_F_BOF if s - h == 0 else 0) | (abi.EXG_F_EOF if e == n else 0)
        res, cols, words = run_fastq(data[s - h:e], algo, lead=h, first_line_index=fli, flags=flags, payload_base=FQ_BASE + s - h)
        assert res.error_code == 0 and not (res.flags & (abi.EXG_RF_HEAD_UNRESOLVED | abi.EXG_RF_FALLBACK)), (s, e, res.flags)
        k = int(res.n_records)
        for c in range(4):
            got_cols[c].append(cols[c])
        total += k
    assert total == exp.n_rows == len(lengths)
    for c, name in enumerate(NAMES):
        assert np.array_equal(np.concatenate(got_cols[c]), exp.string_t[name][0]), name
    # a 4 KiB halo cannot hold the head of the 70 kb read when the cut falls into its quality line: said, not guessed
    s = (int(exp.columns["quality_scores"].src_off[6]) + 30000) // 16 * 16
    res, _, _ = run_fastq(data[s - 4096:], algo, lead=4096, first_line_index=int(np.searchsorted(nl, s)), flags=abi.EXG_F_EOF,
                          payload_base=FQ_BASE + s - 4096)
    assert res.flags & abi.EXG_RF_HEAD_UNRESOLVED and int(res.n_records) == len(lengths) - 6


@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
def test_long_reads_record_aligned_batches(gpu, oracle, algo):
    # streaming: not at EOF, the tail record is incomplete; consumed_bytes says where to resume
    lengths = [20000, 150, 31000, 18000, 150, 52000]
    data = bytes(fastq_records(lengths, seed=4))
    exp = oracle.fastq_parse(data, payload_base=FQ_BASE)
    cut = int(exp.columns["sequence"].src_off[3]) + 5000      # inside the 4th record
    res, cols, _ = run_fastq(data[:cut], algo, flags=abi.EXG_F_BOF)
    assert int(res.n_records) == 3 and res.error_code == 0
    assert int(res.consumed_bytes) == int(exp.columns["name"].src_off[3]) - 1
    assert np.array_equal(cols[2], exp.string_t["sequence"][0][:3])
    no_fallback(res)


@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
def test_long_reads_capacity_and_count_only(gpu, oracle, algo):
    lengths = [9000] * 40
    data = fastq_records(lengths, seed=8)
    res, cols, _ = run_fastq(data, algo, capacity=17)
    assert res.flags & abi.EXG_RF_CAPACITY and res.n_records == 17
    exp = oracle.fastq_parse(bytes(data), payload_base=FQ_BASE)
    assert np.array_equal(cols[3], exp.string_t["quality_scores"][0][:17])
    # COUNT(*): every record validated, nothing stored
    bad = bytes(data) + b"@x\n" + b"A" * 5000 + b"\n*\n" + b"I" * 5000 + b"\n"
    res, _, _ = run_fastq(bad, algo, flags=abi.EXG_F_BOF | abi.EXG_F_EOF | abi.EXG_F_NO_STORE)
    assert res.error_code == abi.EXG_PE_FASTQ_PLUS_PREFIX and res.error_record == 40 and res.n_records == 40
    no_fallback(res)


# ---- short reads: more lines in a half than the LDS list holds ------------------------------------------------------

@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
@pytest.mark.parametrize("shape", ["36bp", "36bp_short_names", "1bp", "empty_fields", "mixed", "crlf"])
def test_short_reads(gpu, oracle, shape, algo):
    rng = np.random.default_rng(2)
    if shape == "36bp":
        data = fastq_records([36] * 3000, seed=6, name_len=24)
    elif shape == "36bp_short_names":
        data = fastq_records([36] * 5000, seed=6, desc_every=0)
    elif shape == "1bp":
        data = fastq_records([1] * 20000, seed=6, desc_every=0)
    elif shape == "empty_fields":
        data = b"@\n\n+\n\n" * 12000 + b"@a b\nAC\n+\n!!\n" * 3000 + b"@\n\n+\n\n" * 7000
    elif shape == "mixed":   # runs of tiny records between ordinary and long ones: halves of 1, 2 and 20 passes
        data = b"".join(fastq_records([int(x)] * int(c), seed=int(c)) for x, c in
                        [(150, 200), (0, 4000), (20000, 2), (3, 3000), (150, 100), (0, 9000), (36, 800), (9000, 3)])
    else:
        data = fastq_records([20] * 4000, seed=6, crlf_every=3, desc_every=5)
    res = check_against_oracle(oracle, data, algo)
    no_fallback(res)


@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
def test_short_reads_errors_and_eof(gpu, oracle, algo):
    body = b"@r\nA\n+\n!\n" * 9000
    for tail in [b"@r\nA\n+\n", b"@r\nA\n+", b"@r\nA\n", b"@r", b"x\nA\n+\n!\n", b"@r\nA\n-\n!\n" + b"@r\nA\n+\n!\n" * 3000]:
        res = check_against_oracle(oracle, body + tail, algo)
        no_fallback(res)


# ---- wide VCF lines ---------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
@pytest.mark.parametrize("n_samples,n_lines", [(100, 600), (2504, 120), (300, 300), (40000, 6)])
def test_multisample_vcf(gpu, oracle, n_samples, n_lines, algo):
    data = vcf_lines(n_lines, n_samples, seed=n_samples, crlf_every=11)
    res = check_vcf(oracle, data, algo)
    assert res.n_records == n_lines
    no_fallback(res)


@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
@pytest.mark.parametrize("case", ["bad_pos", "bad_qual", "missing_field", "no_final_newline", "tabs_far_in", "first_error_wins"])
def test_wide_vcf_line_rules(gpu, oracle, case, algo):
    good = vcf_lines(30, 2504, seed=3)
    wide = b"\tGT" + b"\t0|1" * 3000
    if case == "bad_pos":
        data = good + b"1\tx5\t.\tA\tC\t1.5\tPASS\tDP=1" + wide + b"\n"
    elif case == "bad_qual":
        data = good + b"1\t5\t.\tA\tC\t-1\tPASS\tDP=1" + wide + b"\n" + b"1\t6\t.\tA\tC\t2\tPASS\tDP=1" + wide + b"\n"
    elif case == "missing_field":
        data = good + b"1\t5\t.\tA\tC\t" + b"9" * 5000 + b"\n"
    elif case == "no_final_newline":
        data = good[:-1]
    elif case == "tabs_far_in":     # the eighth tab lies 30 kB into the line (a huge INFO)
        data = good + b"1\t5\t.\tA\tC\t3\tPASS\t" + b";".join(b"K%d=%d" % (i, i) for i in range(3000)) + wide + b"\n" + good[header_bytes(good):]
    else:
        data = good + b"1\t5\t.\tA\tC\tzz\tPASS\tDP=1" + wide + b"\n" + b"1\t\t.\tA\tC\t1\tPASS\tDP=1" + wide + b"\n"
    res = check_vcf(oracle, data, algo)
    no_fallback(res)


@pytest.mark.parametrize("algo", FUSED_AND_PARTNER)
def test_tiny_vcf_lines(gpu, oracle, algo):
    # 15-byte lines: 1 092 per half, more than the list's 1 024: two passes
    data = HDR + b"".join(b"%d\t%d\t.\tA\tC\t.\t.\t.\n" % (k % 9 + 1, k % 10) for k in range(9000))
    res = check_vcf(oracle, data, algo)
    assert res.n_records == 9000
    no_fallback(res)
    # blank lines are lines (an error at the first): 16 384 per half
    res = check_vcf(oracle, HDR + b"1\t5\t.\tA\tC\t.\t.\t.\n" * 3 + b"\n" * 40000, algo)
    no_fallback(res)


# ---- through the reader: small device batches, the sticky choice of the scan ----------------------------------------

def _reader_rows(path, fmt, **kw):
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(str(path), fmt, **kw)
    rows = r.rows()
    st = r.stats()
    r.close()
    return rows, st


def _fastq_rows(oracle, data):
    t = oracle.fastq_parse(bytes(data), want_string_t=False)
    assert t.error_code == 0
    return list(zip(*[t.columns[c].to_list() for c in NAMES]))


@pytest.mark.parametrize("compressed", [False, True])
def test_reader_long_reads_switch_to_the_any_shape_scan(gpu, oracle, tmp_path, compressed):
    """a long-read file in 2 MiB device batches: the first batch comes back marked (EXG_RF_REDO), the batches behind it start
    with EXG_ALGO_FUSED_FULL; the rows are the oracle's, text and BGZF"""
    from test_streaming_gpu import _bgzf
    rng = np.random.default_rng(5)
    lengths = np.exp(rng.uniform(np.log(2000), np.log(60000), 900))
    data = fastq_records(lengths, seed=12, crlf_every=13)
    path = tmp_path / ("long.fastq.gz" if compressed else "long.fastq")
    path.write_bytes(_bgzf(bytes(data)) if compressed else bytes(data))
    rows, st = _reader_rows(path, "fastq", device_batch_bytes=2 << 20)
    assert rows == _fastq_rows(oracle, data)
    assert st["device_batches"] >= 8 and st["scan_algo"] == abi.EXG_ALGO_FUSED_FULL
    # 150 bp reads: the lean scan all the way
    short = bytes(oracle.synth_fastq(332 * 40000))
    p2 = tmp_path / "short.fastq"
    p2.write_bytes(short)
    rows, st = _reader_rows(p2, "fastq", device_batch_bytes=2 << 20)
    assert len(rows) == 40000 and st["scan_algo"] == abi.EXG_ALGO_FUSED


def test_reader_the_odd_long_read_does_not_change_the_scan(gpu, oracle, tmp_path):
    """150 bp reads with a 30 kb read every ~4 MB: the lean scan marks a tile or two per batch, the any-shape run redoes them
    (exg_scan_result.redo_tiles), and the reader stays on the lean scan — it switches when the marks are the input's shape"""
    parts = []
    for k in range(12):
        parts.append(bytes(oracle.synth_fastq(332 * 12000, file_offset=332 * 12000 * k)))
        parts.append(bytes(fastq_records([30000 + 1000 * k], seed=k)))
    data = b"".join(parts)
    p = tmp_path / "mostly_short.fastq"
    p.write_bytes(data)
    rows, st = _reader_rows(p, "fastq", device_batch_bytes=4 << 20)
    assert rows == _fastq_rows(oracle, data)
    assert st["device_batches"] >= 10 and st["scan_algo"] == abi.EXG_ALGO_FUSED
    # device level: the count of redone tiles
    res, _, _ = run_fastq(data, abi.EXG_ALGO_FUSED)
    assert res.flags & abi.EXG_RF_REDO and 12 <= res.redo_tiles <= 40, res.redo_tiles
    res, _, _ = run_fastq(fastq_records([15000] * 200, seed=2), abi.EXG_ALGO_FUSED)
    assert res.redo_tiles >= (200 * 30000 // 49152) * 0.9


def test_reader_short_reads_and_wide_vcf(gpu, oracle, tmp_path):
    data = fastq_records([36] * 60000, seed=3, desc_every=0)
    p = tmp_path / "short36.fastq"
    p.write_bytes(bytes(data))
    rows, st = _reader_rows(p, "fastq", device_batch_bytes=1 << 20)
    assert rows == _fastq_rows(oracle, data) and st["scan_algo"] == abi.EXG_ALGO_FUSED_FULL
    vcf = vcf_lines(900, 2504, seed=4)
    pv = tmp_path / "wide.vcf"
    pv.write_bytes(vcf)
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(str(pv), "vcf", device_batch_bytes=1 << 20, columns=[0, 1, 3, 5])
    got = r.rows()
    st = r.stats()
    r.close()
    t = oracle.vcf_parse(vcf, want_string_t=False)
    want = list(zip(t.columns["chrom"].to_list(), [int(x) for x in t.extra["pos"]], t.columns["ref"].to_list(),
                    [float(q) if v else None for q, v in zip(t.extra["qual"], t.extra["qual_valid"])]))
    assert len(got) == 900 and got == want
    # lines of 10 kB: the any-shape scan, then (round 5) with the rows left to a kernel of their own
    assert st["scan_algo"] == abi.EXG_ALGO_FUSED_INDEX
    # ... the same rows without that switch, and all columns of lines of ~500 bytes (which stay with the any-shape scan)
    import os
    os.environ["EXG_NO_VCF_INDEX"] = "1"
    try:
        r = ShardReader(str(pv), "vcf", device_batch_bytes=1 << 20, columns=[0, 1, 3, 5])
        assert r.rows() == want and r.stats()["scan_algo"] == abi.EXG_ALGO_FUSED_FULL
        r.close()
    finally:
        del os.environ["EXG_NO_VCF_INDEX"]


def test_reader_non_ascii_batches(gpu, oracle, tmp_path):
    """bytes >= 0x80 (valid UTF-8) in the first megabyte of a file read in 256 KiB batches: the lean scan marks their tiles,
    the any-shape scan validates the fields — no batch is given up —, and the reader stays on the any-shape scan; an invalid
    sequence far into the file is that row's error, behind the rows in front of it"""
    head = b"".join(("@r%d caf\u00e9\n" % k).encode() + b"ACGT" * 30 + b"\n+\n" + b"I" * 120 + b"\n" for k in range(4000))
    tail = bytes(oracle.synth_fastq(332 * 30000))
    data = head + tail
    p = tmp_path / "mixed.fastq"
    p.write_bytes(data)
    rows, st = _reader_rows(p, "fastq", device_batch_bytes=256 << 10)
    assert rows == _fastq_rows(oracle, data)
    assert st["scan_algo"] == abi.EXG_ALGO_FUSED_FULL and st["device_batches"] >= 12
    bad = bytearray(data)
    at = len(head) + 332 * 20000 + 40      # inside a sequence line
    bad[at] = 0xC3
    p2 = tmp_path / "bad.fastq"
    p2.write_bytes(bytes(bad))
    from exon_duckdb_amd._lib import ExgError
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(str(p2), "fastq", device_batch_bytes=256 << 10)
    with pytest.raises(ExgError) as e:
        r.rows()
    r.close()
    t = oracle.fastq_parse(bytes(bad), want_string_t=False)
    assert t.error_code == abi.EXG_PE_INVALID_UTF8 and t.n_rows == 4000 + 20000
    assert "utf-8" in str(e.value).lower() or "utf8" in str(e.value).lower(), str(e.value)


# ---- differential fuzz over shapes: random length mixes, mutations, shard cuts ------------------------------------------------

def _random_fastq(rng, n_rec):
    """records of wildly different sizes (0 .. 200 kb, heavy tail), names with and without descriptions, sporadic CRLF"""
    kinds = rng.integers(0, 6, n_rec)
    lengths = np.where(kinds == 0, rng.integers(0, 4, n_rec),
              np.where(kinds == 1, rng.integers(20, 60, n_rec),
              np.where(kinds == 2, rng.integers(100, 400, n_rec),
              np.where(kinds == 3, rng.integers(900, 1200, n_rec),
              np.where(kinds == 4, rng.integers(8000, 40000, n_rec), rng.integers(40000, 200000, n_rec))))))
    return fastq_records(lengths, seed=int(rng.integers(1 << 30)), crlf_every=int(rng.integers(0, 9)), desc_every=int(rng.integers(0, 4)))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EXG_SHAPES_FUZZ", "10"))))
def test_fastq_shapes_fuzz(gpu, oracle, seed):
    from test_fuzz_gpu import mutate
    rng = np.random.default_rng(7000 + seed)
    base = bytes(_random_fastq(rng, int(rng.integers(5, 60))))
    for trial in range(4):
        data = base if trial == 0 else mutate(base, rng, int(rng.integers(1, 4)))
        for algo in FUSED_AND_PARTNER:
            res = check_against_oracle(oracle, data, algo)
            no_fallback(res)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EXG_SHAPES_FUZZ", "6"))))
def test_fastq_shapes_shard_cuts_fuzz(gpu, oracle, seed):
    """random 16-byte-aligned cuts with random halos over a random mix of record sizes: the shards' rows are the file's rows, or
    a shard says that its halo does not reach the head of its first record (and with the whole prefix as halo it has them)"""
    rng = np.random.default_rng(8000 + seed)
    data = bytes(_random_fastq(rng, int(rng.integers(20, 70))))
    exp = oracle.fastq_parse(data, payload_base=FQ_BASE)
    assert exp.error_code == 0
    arr = np.frombuffer(data, np.uint8)
    nl = np.flatnonzero(arr == 10)
    n = len(data)
    cuts = sorted({0, n} | {int(x) // 16 * 16 for x in rng.integers(1, n, int(rng.integers(1, 6)))})
    for algo in (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_FUSED_FULL):
        got = [[] for _ in range(4)]
        for s, e in zip(cuts[:-1], cuts[1:]):
            halo = int(rng.choice([1024, 16384, 262144]))
            while True:
                h = min(halo, s) // 16 * 16
                flags = (abi.EXG_F_BOF if s - h == 0 else 0) | (abi.EXG_F_EOF if e == n else 0)
                res, cols, _ = run_fastq(data[s - h:e], algo, lead=h, first_line_index=int(np.searchsorted(nl, s)), flags=flags,
                                         payload_base=FQ_BASE + s - h)
                assert res.error_code == 0 and not (res.flags & abi.EXG_RF_FALLBACK)
                if not (res.flags & abi.EXG_RF_HEAD_UNRESOLVED):
                    break
                assert h < s, "a halo that reaches the first byte of the file cannot be too short"
                halo *= 8      # what the reader does
            for c in range(4):
                got[c].append(cols[c])
        for c, name in enumerate(NAMES):
            assert np.array_equal(np.concatenate(got[c]), exp.string_t[name][0]), (name, cuts)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("EXG_SHAPES_FUZZ", "8"))))
def test_vcf_shapes_fuzz(gpu, oracle, seed):
    from test_fuzz_gpu import mutate
    rng = np.random.default_rng(9000 + seed)
    lines = []
    for k in range(int(rng.integers(5, 80))):
        ns = int(rng.choice([0, 1, 3, 50, 300, 3000, 20000]))
        info = b";".join(b"K%d=%d" % (i, i) for i in range(int(rng.choice([1, 3, 400]))))
        qual = [b".", b"3", b"1e2", b"59.25"][int(rng.integers(0, 4))]
        lines.append(b"%d\t%d\t.\tA\tC\t%s\tPASS\t%s" % (k % 22 + 1, 100 + k, qual, info) + (b"\tGT" + b"\t0/1" * ns if ns else b"") +
                     (b"\r\n" if rng.integers(0, 7) == 0 else b"\n"))
    base = HDR + b"".join(lines)
    hdr = header_bytes(base)
    for trial in range(4):
        data = base if trial == 0 else base[:hdr] + mutate(base[hdr:], rng, int(rng.integers(1, 4)))
        if header_bytes(data) != hdr:
            continue
        for algo in FUSED_AND_PARTNER:
            res = check_vcf(oracle, data, algo)
            no_fallback(res)


def test_the_scan_is_chosen_before_the_first_launch(gpu, oracle, tmp_path):
    """round 6: a reader looks at the first MiB behind the header (exg_scan_algo_hint) — a cohort VCF of ONE device batch runs the
    indexed scan, a long-read file the any-shape scan, from their first launch (before: lean + redo, then any-shape, then indexed:
    a file of one batch never reached its scan)"""
    from exon_duckdb_amd.reader import ShardReader
    vcf = vcf_lines(300, 2504, seed=9)
    pv = tmp_path / "cohort.vcf"
    pv.write_bytes(vcf)
    r = ShardReader(str(pv), "vcf", columns=[0, 1, 3, 5])
    assert r.stats()["scan_algo"] == abi.EXG_ALGO_FUSED_INDEX and r.stats()["device_batches"] == 0   # (the header is read at open)
    got = r.rows()
    st = r.stats()
    r.close()
    t = oracle.vcf_parse(vcf, want_string_t=False)
    assert len(got) == 300 and [g[1] for g in got] == [int(x) for x in t.extra["pos"]]
    assert st["device_batches"] == 1 and st["scan_algo"] == abi.EXG_ALGO_FUSED_INDEX
    data = fastq_records([20000] * 30, seed=10)
    pf = tmp_path / "long1.fastq"
    pf.write_bytes(bytes(data))
    rows, st = _reader_rows(pf, "fastq")
    assert rows == _fastq_rows(oracle, data) and st["device_batches"] == 1 and st["scan_algo"] == abi.EXG_ALGO_FUSED_FULL
