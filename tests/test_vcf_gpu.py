"""Parity of the HIP VCF scan (C-ABI exg_vcf_scan) against the oracle: nine string_t column vectors,
parsed POS (int64) and QUAL (float32, exact), QUAL / formats validity words, result block."""
import os

import numpy as np
import pytest

from exon_duckdb_amd import abi

pytestmark = pytest.mark.gpu

BASE = 0x7E0000000000
ALGOS = [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_AUTO, abi.EXG_ALGO_FUSED_FULL, abi.EXG_ALGO_FUSED_INDEX]
# the lean scan + the any-shape run over what it marked; the any-shape scan alone; the any-shape scan noting the line ends for a
# kernel behind it that parses the rows (round 5: wide lines)
FUSED = (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_FUSED_FULL, abi.EXG_ALGO_FUSED_INDEX)
HDR = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"


def header_bytes(data: bytes) -> int:
    """Host side of the boundary: leading '#' lines (what the reader does before calling the kernel)."""
    pos = 0
    while pos < len(data) and data[pos:pos + 1] == b"#":
        nl = data.find(b"\n", pos)
        pos = len(data) if nl < 0 else nl + 1
    return pos


def run_gpu(data, algo, capacity=None, flags=abi.EXG_F_BOF | abi.EXG_F_EOF):
    from exon_duckdb_amd import device

    data = bytes(data)
    d_in = device.upload(data)
    scan = device.VcfScan(len(data), capacity_records=capacity)
    scan.launch(d_in, lead=header_bytes(data), payload_base=BASE, flags=flags, algo=algo)
    res = scan.fetch()
    return res, scan.host(int(res.n_records))


def bits(words, n):
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n]


def check(oracle, data, algo, expect_fallback=False):
    data = bytes(data)
    exp = oracle.vcf_parse(data, payload_base=BASE)
    res, got = run_gpu(data, algo)
    assert not (res.flags & abi.EXG_RF_FALLBACK), "a fused launch asked for the general path"
    assert res.error_code == exp.error_code, (res.error_code, exp.error_code, exp.error_message)
    assert res.n_records == exp.n_rows
    n = exp.n_rows
    if exp.error_code:
        assert res.error_record == exp.error_record and res.error_offset == exp.error_offset
    for k, name in enumerate(oracle.VCF_FIELDS):
        assert np.array_equal(got["cols"][k], exp.string_t[name][0]), name
    assert np.array_equal(got["pos"], exp.extra["pos"])
    qv = bits(got["qual_valid"], n)
    assert np.array_equal(qv, exp.extra["qual_valid"])
    assert np.array_equal(got["qual"].view(np.uint32)[qv == 1], exp.extra["qual"].view(np.uint32)[qv == 1])  # bit exact
    assert np.array_equal(bits(got["formats_valid"], n), exp.columns["formats"].valid)
    if not exp.error_code:
        assert res.consumed_bytes == len(data) and res.n_lines == n
    return res


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("name", ["vcf/index.vcf", "vcf/vcf_file.vcf", "vcf/vcf_meta_meta.vcf"])
def test_reference_fixtures(gpu, oracle, golden_dir, name, algo):
    with open(os.path.join(golden_dir, name), "rb") as f:
        data = f.read()
    res = check(oracle, data, algo)
    if name == "vcf/index.vcf":
        assert res.n_records == 621        # test_vcf_record_scan.test:4-7


@pytest.mark.parametrize("algo", ALGOS)
def test_reference_row0(gpu, golden_dir, algo):
    # test_vcf_record_scan.test:10-19, straight from the device output
    with open(os.path.join(golden_dir, "vcf/index.vcf"), "rb") as f:
        data = f.read()
    res, got = run_gpu(data, algo)

    def text(k):
        st = got["cols"][k][0]
        ln = int(st[:4].view(np.uint32)[0])
        if ln <= 12:
            return st[4:4 + ln].tobytes().decode()
        off = int(st[8:16].view(np.uint64)[0]) - BASE
        return data[off:off + ln].decode()
    assert text(0) == "1" and int(got["pos"][0]) == 9999919 and text(3) == "G" and text(4) == "<*>"
    assert float(got["qual"][0]) == 0.0 and (int(got["qual_valid"][0]) & 1) == 1
    info = text(7)
    assert "INDEL" not in info.split(";") and "DP=1" in info.split(";")


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n_lines", [1, 2, 63, 64, 65, 1000, 30011])
def test_synth_vcf(gpu, oracle, n_lines, algo):
    data = oracle.synth_vcf(n_lines)
    res = check(oracle, data, algo)
    assert res.n_records == n_lines


EDGE = {
    "header_only": HDR,
    "one_line": HDR + b"1\t5\t.\tA\tC\t.\tPASS\tDP=1\n",
    "no_trailing_newline": HDR + b"1\t5\t.\tA\tC\t3.5\tPASS\tX\n1\t7\t.\tA\tC\t.\t.\tY",
    "crlf": HDR + b"1\t5\t.\tA\tC\t3.5\tPASS\tX\r\n1\t7\trs1\tA\tC,G\t1e2\tq10\t.\tGT\t0/1\t1/1\r\n",
    "with_samples": HDR + b"2\t6\trs1\tA\tC,G\t1e2\tq10\t.\tGT\t0/1\t1/1\n",
    "empty_fields": HDR + b"\t5\t\t\t\t.\t\t\n",
    "missing_field": HDR + b"1\t5\t.\tA\tC\t.\tPASS\tOK\n1\t5\t.\tA\tC\t.\tPASS\n",
    "blank_line": HDR + b"1\t5\t.\tA\tC\t.\tPASS\t.\n\n",
    "bad_pos": HDR + b"1\tx5\t.\tA\tC\t.\tPASS\t.\n",
    "bad_pos_empty": HDR + b"1\t\t.\tA\tC\t.\tPASS\t.\n",
    "pos_plus_sign": HDR + b"1\t+42\t.\tA\tC\t.\tPASS\t.\n",
    "pos_big": HDR + b"1\t9223372036854775807\t.\tA\tC\t.\tPASS\t.\n",
    "pos_overflow": HDR + b"1\t9223372036854775808\t.\tA\tC\t.\tPASS\t.\n",
    "bad_qual": HDR + b"1\t5\t.\tA\tC\tabc\tPASS\t.\n",
    "qual_forms": HDR + b"".join(b"1\t5\t.\tA\tC\t" + q + b"\tPASS\t.\n" for q in
                                 [b"0", b"0.0", b"59.2", b"12.9", b"1e2", b"1E+2", b"2.5e-3", b".5", b"5.", b"+7.25",
                                  b"-0", b"inf", b"Infinity", b"NaN", b"3000", b"16777217", b"0.1", b"0.3",
                                  b"123456.789", b"1.17549435e-38"[:0] + b"9.999999e9", b"000012.50"]),
    "qual_bad_forms": HDR + b"1\t5\t.\tA\tC\t1e\tPASS\t.\n",
    # noodles-vcf 0.34 QualityScore refuses n < 0.0: "-1" and "-inf" are errors; "-0", "-0.0", "nan", "-nan" are not
    "qual_negative": HDR + b"1\t5\t.\tA\tC\t-0\tPASS\t.\n1\t6\t.\tA\tC\t-1\tPASS\t.\n1\t7\t.\tA\tC\t2\tPASS\t.\n",
    "qual_negative_small": HDR + b"1\t5\t.\tA\tC\t-nan\tPASS\t.\n1\t6\t.\tA\tC\t-1e-50\tPASS\t.\n1\t6\t.\tA\tC\t-1e-45\tPASS\t.\n",
    "qual_negative_inf": HDR + b"1\t5\t.\tA\tC\t-0.0\tPASS\t.\n1\t6\t.\tA\tC\t-Infinity\tPASS\t.\n",
    "qual_dot_only": HDR + b"1\t5\t.\tA\tC\t..\tPASS\t.\n",
    "long_info": HDR + b"1\t5\t.\tA\tC\t1\tPASS\t" + b"K=" + b"v" * 700 + b"\tGT\t0/1\n" + b"1\t6\t.\tA\tC\t1\tPASS\tS\n",
    "first_error_wins": HDR + b"1\t5\t.\tA\tC\t.\tPASS\t.\n1\tzz\t.\tA\tC\t.\tPASS\t.\n1\t5\t.\tA\n",
    "hash_line_after_header_is_data": HDR + b"1\t5\t.\tA\tC\t.\tPASS\t.\n#oops\n",
}


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("case", sorted(EDGE))
def test_edge_cases(gpu, oracle, case, algo):
    check(oracle, EDGE[case], algo)


def test_no_header_is_reported_by_the_oracle_only(oracle):
    # header detection is host work (the reader finds the '#' prefix): documented split of labour
    assert oracle.vcf_parse(b"1\t5\t.\tA\tC\t.\tPASS\tDP=1\n").error_code == abi.EXG_PE_VCF_NO_HEADER


@pytest.mark.parametrize("algo", ALGOS)
def test_long_lines_stay_on_the_single_pass(gpu, oracle, algo):
    # multi-sample lines longer than the fused kernel's straddle window: k_vcf_far emits them, no launch is given up
    # (more shapes: tests/test_record_shapes_gpu.py)
    rng = np.random.default_rng(3)
    lines = []
    for k in range(200):
        ns = int(rng.integers(1, 4000)) if k % 2 else 2
        lines.append(b"%d\t%d\t.\tA\tC\t%d.5\tPASS\tDP=%d\tGT" % (k % 22 + 1, 100 + k, k, k) + b"\t0/1" * ns + b"\n")
    data = HDR + b"".join(lines)
    res = check(oracle, data, algo)
    assert res.n_records == 200 and not (res.flags & abi.EXG_RF_FALLBACK)


def test_indexed_scan_on_wide_lines_and_when_its_index_is_too_small(gpu, oracle):
    """EXG_ALGO_FUSED_INDEX (round 5): the any-shape scan notes where every line ends (8 bytes a line in the workspace), k_vcf_lines
    parses the rows behind it — lines of 10 kB that straddle halves and super-tiles, CRLF, a last line without a newline; and a
    buffer with more lines than the workspace's index holds (one per 8 bytes of input: only empty lines get there) says
    EXG_RF_FALLBACK like every fused launch that gives a batch up, it does not write past the index"""
    rng = np.random.default_rng(11)
    lines = []
    for k in range(300):
        ns = int(rng.integers(2000, 2600))
        eol = b"\r\n" if k % 7 == 0 else b"\n"
        lines.append(b"chr%d\t%d\trs%d\tA\tC,G\t%s\tPASS\tDP=%d;AF=0.5\tGT:DP" % (k % 22 + 1, 1000 + k, k, b"." if k % 5 == 0 else b"%d.25" % k, k)
                     + b"\t0/1:12" * ns + eol)
    data = HDR + b"".join(lines)
    for d in (data, data[:-1]):
        res = check(oracle, d, abi.EXG_ALGO_FUSED_INDEX)
        assert res.n_records == 300 and not (res.flags & abi.EXG_RF_FALLBACK)
    # (the index holds a line per 8 bytes of input, and never fewer than min(n, 1 Mi) lines: 12 MiB of empty lines overflow it)
    res, _ = run_gpu(HDR + b"\n" * (12 << 20), abi.EXG_ALGO_FUSED_INDEX, capacity=1024)
    assert res.flags & abi.EXG_RF_FALLBACK


@pytest.mark.parametrize("seed", range(8))
def test_random_line_widths_every_selector(gpu, oracle, seed):
    """lines of 16 bytes .. 24 kB in random order (narrow runs between wide lines: dense halves, halves without a line end, lines
    across super-tiles), random CRLF / '.' QUAL / long REF / non-ASCII INFO, every other seed with ONE broken line (too few fields,
    a bad POS, a bad QUAL, invalid UTF-8) at a random row: rows, flags, the error's code, row and offset against the oracle under
    all five selectors — the row kernel behind the indexed scan reads its lines from global memory with aligned loads, whatever
    their alignment and however close to the buffer's end"""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(300, 900))
    bad_at = int(rng.integers(0, n)) if seed % 2 else -1
    lines = []
    for k in range(n):
        wide = rng.random() < 0.35
        ns = int(rng.integers(300, 4000)) if wide else int(rng.integers(0, 4))
        ref = b"ACGT"[k % 4:k % 4 + 1] * int(rng.choice([1, 1, 1, 2, 12, 13, 60]))
        qual = b"." if rng.random() < 0.3 else (b"%d" % int(rng.integers(0, 10000)) if rng.random() < 0.5 else b"%.3f" % float(rng.random() * 99))
        info = b"DP=%d" % k + (b";NOTE=caf\xc3\xa9" if rng.random() < 0.1 else b"") + (b";AF=" + b",".join(b"0.%d" % j for j in range(int(rng.integers(1, 40)))) if wide else b"")
        f = [b"chr%d" % (k % 22 + 1), b"%d" % (1 + k * 7), b"rs%d" % k if k % 3 else b".", ref, b"T", qual, b"PASS", info]
        if k == bad_at:
            kind = seed % 8 // 2
            if kind == 0:
                f = f[:5]
            elif kind == 1:
                f[1] = b"12x4"
            elif kind == 2:
                f[5] = b"-3"
            else:
                f[7] = b"DP=1;X=\xff\xfe"
        line = b"\t".join(f)
        if ns:
            line += b"\tGT:DP" + b"\t0/1:%d" % (k % 50) * ns
        lines.append(line + (b"\r\n" if rng.random() < 0.1 else b"\n"))
    data = HDR + b"".join(lines)
    if seed % 3 == 0:
        data = data[:-1]  # no last newline
    for algo in ALGOS:
        check(oracle, data, algo)


@pytest.mark.parametrize("algo", [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_FULL, abi.EXG_ALGO_FUSED_INDEX])
def test_qual_parse_is_correctly_rounded(gpu, oracle, algo):
    rng = np.random.default_rng(11)
    quals = []
    for _ in range(4000):
        kind = rng.integers(0, 4)
        if kind == 0:
            quals.append(b"%d.%0*d" % (rng.integers(0, 100000), int(rng.integers(1, 7)), rng.integers(0, 10 ** 6) % 10 ** int(rng.integers(1, 7))))
        elif kind == 1:
            quals.append(b"%d" % rng.integers(0, 2 ** 40))
        elif kind == 2:
            quals.append(b"%.9g" % float(np.float32(rng.random() * 10.0 ** int(rng.integers(-6, 9)))))
        else:
            quals.append(b"%de%d" % (rng.integers(1, 10 ** 9), rng.integers(-20, 15)))
    data = HDR + b"".join(b"1\t5\t.\tA\tC\t" + q + b"\tPASS\t.\n" for q in quals)
    check(oracle, data, algo)


def test_long_and_extreme_qual_literals_are_exact(gpu, oracle):
    """Beyond Clinger's fast path: 8 .. 19 significant digits, exponents of any size, subnormals, overflow to infinity,
    exact halfway points (Eisel-Lemire on the first 19 digits), and literals of 20 .. 700 digits (decided by the two ends
    of the interval their first 19 digits pin down) — bit for bit the oracle's strtof."""
    rng = np.random.default_rng(77)
    quals = [b"16777217", b"16777216.000000000000000000000", b"16777217.0000000000000000000000001", b"16777218.99999999999999999999",
             b"1.00000005960464477539062500000000000000001",
             b"0.100000001490116119384765625", b"0.1000000014901161193847656250000000000000000000000000001",
             b"3.4028234663852886e38", b"3.4028235677973366e38", b"3.40282357e38", b"1e39", b"123456789e31", b"1e-45", b"7e-46", b"7.1e-46",
             b"1.401298464324817e-45", b"1.17549435e-38", b"1.1754942e-38", b"5.877471754111438e-39", b"1e-60", b"0e999999", b"1e+38", b"1E-38",
             b"9007199254740993", b"9007199254740992.5", b"18446744073709551615", b"18446744073709551616", b"99999999999999999999999999999999999999",
             b"0." + b"0" * 40 + b"123456789", b"1" + b"0" * 38, b"1" + b"0" * 39, b"0.3333333333333333333333333333333333", b"2.7182818284590452353602874713527",
             b"1." + b"7" * 600, b"8" * 30 + b"." + b"8" * 30 + b"e-25"]
    for _ in range(3000):
        nd = int(rng.integers(8, 40))
        digits = "".join(str(int(d)) for d in rng.integers(0, 10, nd))
        dot = int(rng.integers(0, nd + 1))
        lit = (digits[:dot] or "0") + "." + digits[dot:] if rng.random() < 0.8 else digits
        if rng.random() < 0.5:
            lit += "e%d" % int(rng.integers(-60, 45))
        quals.append(lit.encode())
    # python-style reprs (17 significant digits) of floats and of float32 values
    for _ in range(1500):
        x = float(rng.random() * 10.0 ** int(rng.integers(-30, 30)))
        quals.append(repr(x).encode())
        quals.append(repr(float(np.float32(x))).encode())
    lines = [b"1\t5\t.\tA\tC\t" + q + b"\tPASS\t.\n" for q in quals]
    for algo in (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_INDEX):
        res = check(oracle, HDR + b"".join(lines), algo)
        assert res.error_code == 0 and res.n_records == len(quals)


def test_long_literals_astride_a_rounding_boundary_are_decided_exactly(gpu, oracle):
    """More than 19 significant digits AND a float rounding boundary strictly inside the interval their first 19 digits pin
    down: the scan kernel hands the literal to the exact (big-integer) parser that runs in the finalize kernel
    (exg_float_slow.hpp) — what Rust's dec2flt slow path decides.  Up to 4096 per launch; one more is reported
    (EXG_PE_VCF_BAD_QUAL + EXG_RF_QUAL_RANGE), never mis-rounded."""
    from exon_duckdb_amd import device
    base = "1.000000059604644775390625"          # halfway between 1.0 and its successor
    lits = [base + "0" * 15 + "1", base[:-1] + "4" + "9" * 20, base + "0" * 40, "2.00000011920928955078125" + "0" * 9 + "3",
            "-0.000000000000000000000000000000000000000000000700649232162408535461864791644958065640130970938257885878534141944895541342930300743319094181060791015626",
            "340282356779733661637539395458142568447.9999999", "16777219.000000000000000000001", "8388609.4999999999999999999999", "0.1" + "0" * 30 + "1"]
    lits = [l.encode() for l in lits]
    for algo in (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_INDEX):
        data = HDR + b"".join(b"1\t%d\t.\tA\tC\t" % (5 + k) + q + b"\tPASS\t.\n" for k, q in enumerate(lits) if not q.startswith(b"-"))
        res = check(oracle, data, algo)
        assert res.error_code == 0 and res.n_records == len(lits) - 1 and not (res.flags & abi.EXG_RF_QUAL_RANGE)
        # the negative one (-7.006e-46 rounds to -1.4e-45 < 0): an error of ITS row, found by the exact parser
        data = HDR + b"1\t5\t.\tA\tC\t2\tPASS\t.\n1\t6\t.\tA\tC\t" + lits[4] + b"\tPASS\t.\n1\t7\t.\tA\tC\t3\tPASS\t.\n"
        res = check(oracle, data, algo)
        assert res.error_code == abi.EXG_PE_VCF_BAD_QUAL and res.error_record == 1
    # hundreds of such literals in one launch: all decided exactly (the list lives in the workspace: one entry per 32 bytes
    # of input, at most 4096)
    many = [lits[k % 4] for k in range(700)]
    data = HDR + b"".join(b"1\t%d\t.\tA\tC\t" % (5 + k) + q + b"\tPASS\t.\n" for k, q in enumerate(many))
    for algo in (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_INDEX):
        res = check(oracle, data, algo)
        assert res.error_code == 0 and res.n_records == len(many) and not (res.flags & abi.EXG_RF_QUAL_RANGE)
    # more than the list can hold (4096 per launch): reported, never mis-rounded
    data = HDR + b"".join(b"1\t%d\t.\tA\tC\t" % k + lits[0] + b"\tPASS\t.\n" for k in range(4200))
    d_in = device.upload(data)
    scan = device.VcfScan(len(data))
    scan.launch(d_in, lead=header_bytes(data))
    res = scan.fetch()
    assert res.error_code == abi.EXG_PE_VCF_BAD_QUAL and (res.flags & abi.EXG_RF_QUAL_RANGE)


def test_short_decimal_qual_and_pos_fast_paths_are_exact(gpu, oracle):
    # the 32-bit paths of the number parsers (<= 7 significant digits / <= 9 POS digits) and their neighbours:
    # every m / 10^k for m < 30000, k = 1..4, forms around the 7-digit and 12-character limits, POS of 1..19 digits
    quals = [b"%d.%0*d" % (m // 10 ** k, k, m % 10 ** k) for k in (1, 2, 3, 4) for m in range(0, 30000, 1 if k < 3 else 7)]
    quals += [b"9999999", b"10000000", b"1234567.8", b"0.1234567", b"0.12345678", b"0.00000001", b".5", b"5.", b"+7.25",
              b"-0.0", b"00012.50", b"0.0000000001", b"16777217", b"16777216.0", b"8388608.5", b"0.1", b"0.3", b"100.0"]
    poss = [b"1", b"+1", b"999999999", b"1000000000", b"4294967295", b"4294967296", b"123456789012345678",
            b"9223372036854775807", b"000000000000000000012"]
    lines = [b"1\t5\t.\tA\tC\t" + q + b"\tPASS\t.\n" for q in quals]
    lines += [b"1\t" + p + b"\t.\tA\tC\t1.5\tPASS\t.\n" for p in poss]
    for algo in (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_INDEX):
        check(oracle, HDR + b"".join(lines), algo)
    for bad in (b"9223372036854775808", b"12a", b"+", b""):
        res = check(oracle, HDR + b"1\t" + bad + b"\t.\tA\tC\t1.5\tPASS\t.\n", abi.EXG_ALGO_AUTO)
        assert res.error_code != 0


@pytest.mark.parametrize("seed", range(6))
def test_random_number_fields(gpu, oracle, seed):
    """POS and QUAL literals of random shape — the register fast paths (one 12-byte read, <= 9 digits / the usual decimal),
    their limits, and everything that falls through to the general parsers — in lines of random neighbours so that the wave's
    longest literal, the lanes' activity and the 12-byte reads past short fields all vary: bit-exact against the oracle,
    valid files and (one bad literal at a random row) failing ones."""
    rng = np.random.default_rng(9000 + seed)

    def digits(n, lead_zero=False):
        d = "".join(str(int(x)) for x in rng.integers(0, 10, n))
        return d if lead_zero or n == 0 else (str(int(rng.integers(1, 10))) + d[1:])

    def pos():
        k = rng.integers(0, 10)
        if k < 6:
            p = digits(int(rng.integers(1, 10)), lead_zero=bool(rng.integers(0, 2)))
        elif k < 8:
            p = digits(int(rng.integers(10, 19)))
        else:
            p = "0" * int(rng.integers(1, 12)) + digits(int(rng.integers(1, 8)))
        return ("+" if rng.integers(0, 6) == 0 else "") + p

    def qual():
        k = rng.integers(0, 12)
        if k == 0:
            return "."
        sign = ["", "", "", "+"][int(rng.integers(0, 4))]
        ip, fp = digits(int(rng.integers(0, 8)), True), digits(int(rng.integers(0, 8)), True)
        if k < 7:
            body = (ip or "0") + ("." + fp if rng.integers(0, 3) else "")
        elif k < 9:
            body = (ip + "." + fp) if (ip or fp) else "0."
        elif k == 9:
            body = digits(int(rng.integers(1, 4)), True) + "." + digits(int(rng.integers(1, 4)), True) + "e" + ["", "-", "+"][int(rng.integers(0, 3))] + str(int(rng.integers(0, 40)))
        elif k == 10:
            body = digits(int(rng.integers(8, 30))) + "." + digits(int(rng.integers(0, 30)), True)
        else:
            body = ["inf", "Infinity", "nan", "NaN", "0", "00", "1e0", "12345678", "0.000001"][int(rng.integers(0, 9))]
        return sign + body

    rows = [b"%d\t%s\t.\tA\tC\t%s\tPASS\tDP=%d\n" % (1 + i % 22, pos().encode(), qual().encode(), int(rng.integers(0, 1000)))
            for i in range(3000)]
    data = HDR + b"".join(rows)
    for algo in (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_INDEX):
        res = check(oracle, data, algo)
        assert res.error_code == 0 and res.n_records == len(rows)
    bad_lits = [b"12a4", b"1..2", b"--1", b"1e", b"e5", b"+", b"1 2", b"0x10", b"-5", b"-inf"]
    for _ in range(4):
        r = int(rng.integers(0, len(rows)))
        bad = bad_lits[int(rng.integers(0, len(bad_lits)))]
        field = 1 if (rng.integers(0, 2) and not bad.startswith(b"-")) else 5
        cols = rows[r].split(b"\t")
        cols[field] = bad
        broken = HDR + b"".join(rows[:r]) + b"\t".join(cols) + b"".join(rows[r + 1:])
        for algo in (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_INDEX):
            res = check(oracle, broken, algo)
            assert res.error_code != 0 and res.error_record == r, (bad, field)


@pytest.mark.parametrize("algo", [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_FULL, abi.EXG_ALGO_FUSED_INDEX])
def test_projection_and_capacity(gpu, oracle, algo):
    from exon_duckdb_amd import device

    data = bytes(oracle.synth_vcf(500))
    exp = oracle.vcf_parse(data, payload_base=BASE)
    d_in = device.upload(data)
    scan = device.VcfScan(len(data), capacity_records=100)
    for c in scan.cols:
        c.fill_(-1)
    scan.launch(d_in, lead=header_bytes(data), payload_base=BASE, algo=algo, project={0, 7})
    res = scan.fetch()
    assert res.flags & abi.EXG_RF_CAPACITY and res.n_records == 100
    got = scan.host(100)
    assert np.array_equal(got["cols"][0], exp.string_t["chrom"][0][:100])
    assert np.array_equal(got["cols"][7], exp.string_t["info"][0][:100])
    assert (got["cols"][3] == 0xFF).all()       # unprojected column untouched
    assert np.array_equal(got["pos"], exp.extra["pos"][:100])


def test_config3_full_size_properties(gpu, oracle):
    """BASELINE.json configs[2] at its full size: ~5 GB of 8-column VCF built in HBM as header + T copies of one
    synthetic body of L lines.  The first copy is checked bit for bit against the oracle (it parses 400 k lines);
    the other T - 1 copies through periodicity: row t L + j must equal row j with every out-of-line pointer
    advanced by t x body bytes (inlined strings, POS, QUAL and validity identical)."""
    import torch
    from exon_duckdb_amd import device

    L = 400_000
    body = oracle.synth_vcf(L)
    exp = oracle.vcf_parse(bytes(body), payload_base=BASE)
    assert exp.n_rows == L and exp.error_code == 0
    hdr = int(exp.extra["header_bytes"])
    blen = len(body) - hdr
    T = 5_000_000_000 // blen
    n = hdr + T * blen
    d_body = torch.frombuffer(bytearray(bytes(body)), dtype=torch.uint8).cuda()
    d_in = torch.zeros(n + 80, dtype=torch.uint8, device="cuda")
    d_in[:hdr] = d_body[:hdr]
    d_in[hdr:n].view(T, blen)[:] = d_body[hdr:]
    assert d_in.data_ptr() % 16 == 0
    scan = device.VcfScan(n, capacity_records=T * L + 16)
    scan.launch(d_in, lead=hdr, payload_base=BASE, algo=abi.EXG_ALGO_AUTO)
    res = scan.fetch()
    assert res.error_code == 0 and res.n_records == T * L and res.consumed_bytes == n
    assert not (res.flags & abi.EXG_RF_FALLBACK)
    # copy 0 against the oracle
    got = scan.host(L)
    for k, name in enumerate(oracle.VCF_FIELDS):
        assert np.array_equal(got["cols"][k], exp.string_t[name][0]), name
    assert np.array_equal(got["pos"], exp.extra["pos"])
    qv = bits(got["qual_valid"], L)
    assert np.array_equal(qv, exp.extra["qual_valid"])
    assert np.array_equal(got["qual"].view(np.uint32)[qv == 1], exp.extra["qual"].view(np.uint32)[qv == 1])
    # periodicity of the rest
    shift = (torch.arange(T, device="cuda", dtype=torch.int64) * blen).view(T, 1)
    for k in range(9):
        c = scan.cols[k][: T * L].view(T, L, 2)
        lens = c[0, :, 0] & 0xFFFFFFFF
        assert bool((c[:, :, 0] == c[0, :, 0]).all())                       # length + prefix / first inlined bytes
        outline = (lens > 12).view(1, L)
        want = torch.where(outline, c[0, :, 1].view(1, L) + shift, c[0, :, 1].view(1, L))
        assert bool((c[:, :, 1] == want).all())
    assert bool((scan.pos[: T * L].view(T, L) == scan.pos[:L]).all())
    assert L % 64 == 0
    assert bool((scan.qual_valid[: T * L // 64].view(T, L // 64) == scan.qual_valid[: L // 64]).all())
    q = scan.qual[: T * L].view(T, L).view(torch.int32)
    valid = torch.from_numpy(qv.astype(np.bool_)).cuda()
    assert bool((q[:, valid] == q[0, valid]).all())


def test_config3_full_size_generated(gpu, oracle):
    """BASELINE.json configs[2] at its full size on a NON-periodic input: ~5 GB (about 103 M lines) from the device
    generator.  Three windows of whole lines (head, middle, tail; 60 k lines each) are parsed by the oracle and compared bit
    for bit with the same rows of the 5 GB launch; every row is covered by size-independent properties: the generator's
    closed forms for CHROM and POS, and the byte ledger (the lengths of a row's eight fields + its eight separators, summed
    over all rows, is the body's byte count; out-of-line pointers increase strictly with the row)."""
    import torch
    from exon_duckdb_amd import device

    n_lines = int(5e9 / 48.65)
    d_in, n = device.synth_vcf(n_lines)
    head = bytes(d_in[:4096].cpu().numpy())
    hdr = header_bytes(head[: head.index(b"\n1\t") + 1])
    scan = device.VcfScan(n, capacity_records=n_lines + 16)
    scan.launch(d_in, n_bytes=n, lead=hdr, payload_base=BASE, algo=abi.EXG_ALGO_AUTO)
    res = scan.fetch()
    assert res.error_code == 0 and res.n_records == n_lines and res.consumed_bytes == n
    assert not (res.flags & abi.EXG_RF_FALLBACK)

    def newlines_before(a):
        tot = 0
        for o in range(hdr, a, 1 << 30):
            tot += int((d_in[o:min(a, o + (1 << 30))] == 10).sum().item())
        return tot

    K = 60_000
    for start in (hdr, n // 2, None):
        if start is None:                       # the last K lines
            w = bytes(d_in[n - K * 80:n].cpu().numpy())
            cut = len(w)
            for _ in range(K + 1):
                cut = w.rfind(b"\n", 0, cut)
            a, b = n - K * 80 + cut + 1, n
        else:
            a = start
            if a != hdr:
                a += bytes(d_in[a:a + 4096].cpu().numpy()).index(b"\n") + 1
            w = bytes(d_in[a:a + K * 80].cpu().numpy())
            cut = -1
            for _ in range(K):
                cut = w.index(b"\n", cut + 1)
            b = a + cut + 1
        r0 = newlines_before(a)
        window = head[:hdr] + bytes(d_in[a:b].cpu().numpy())
        exp = oracle.vcf_parse(window, payload_base=BASE + a - hdr)
        assert exp.error_code == 0 and exp.n_rows == K
        for k, name in enumerate(oracle.VCF_FIELDS):
            got = scan.cols[k][r0:r0 + K].cpu().numpy().view(np.uint8).reshape(K, 16)
            assert np.array_equal(got, exp.string_t[name][0]), (name, start)
        assert np.array_equal(scan.pos[r0:r0 + K].cpu().numpy(), exp.extra["pos"])
        qv = exp.extra["qual_valid"].astype(bool)
        gq = scan.qual[r0:r0 + K].cpu().numpy().view(np.uint32)
        assert np.array_equal(gq[qv], exp.extra["qual"].view(np.uint32)[qv])
        if r0 % 64 == 0:
            assert np.array_equal(bits(scan.qual_valid[r0 // 64:(r0 + K + 63) // 64].cpu().numpy().view(np.uint64), K),
                                  exp.extra["qual_valid"])
    # every row: closed forms + the byte ledger
    per_chrom = n_lines // 22 + 1
    i = torch.arange(n_lines, device="cuda", dtype=torch.int64)
    assert bool((((scan.pos[:n_lines] - 1) // 37) == (i % per_chrom)).all())
    chrom = i // per_chrom + 1
    want = torch.where(chrom < 10, 1 | ((0x30 + chrom) << 32), 2 | ((0x30 + chrom // 10) << 32) | ((0x30 + chrom % 10) << 40))
    assert bool((scan.cols[0][:n_lines, 0] == want).all())
    del i, chrom, want
    total = 0
    for k in range(8):
        lens = scan.cols[k][:n_lines, 0] & 0xFFFFFFFF
        total += int(lens.sum().item())
        ptr = scan.cols[k][:n_lines, 1]
        out = lens > 12
        p = ptr[out]
        if p.numel() > 1:
            assert bool((p[1:] > p[:-1]).all()), k
            assert int(p[0].item()) >= BASE + hdr and int(p[-1].item()) < BASE + n
        del lens, ptr, out, p
    assert total + 8 * n_lines == n - hdr


def test_device_generators_match_the_oracle(gpu, oracle):
    """exg_synth_vcf / exg_synth_fasta (two passes: lengths -> scan -> write) produce the oracle's bytes: bench.py takes its
    VCF and FASTA inputs from them (the bench may not touch the oracle outside its cpu_baseline leg)."""
    from exon_duckdb_amd import device
    for n in (1, 22, 1000, 50001):
        t, nb = device.synth_vcf(n)
        want = bytes(oracle.synth_vcf(n))
        assert nb == len(want) and bytes(t[:nb].cpu().numpy()) == want, n
    for n in (1, 3, 500, 4001):
        t, nb = device.synth_fasta(n)
        want = bytes(oracle.synth_fasta(n))
        assert nb == len(want) and bytes(t[:nb].cpu().numpy()) == want, n
