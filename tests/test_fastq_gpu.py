"""Parity of the HIP FASTQ scan (through the C-ABI exg_fastq_scan) against the oracle.

Bit-exact bar: the four string_t column vectors, the description validity words and the result
block must equal what the oracle produces on the same bytes (canonical zero-copy view:
ptr = payload_base + field offset).  Both device implementations are checked.
"""
import os

import numpy as np
import pytest

from exon_duckdb_amd import abi

pytestmark = pytest.mark.gpu

BASE = 0x7F0000000000  # payload_base used by all tests (pointers become BASE + file offset)
ALGOS = [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_AUTO, abi.EXG_ALGO_FUSED_FULL]
FUSED = (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_FUSED_FULL)   # the lean scan + the any-shape run over what it marked; the any-shape scan alone
NAMES = ["name", "description", "sequence", "quality_scores"]


def run_gpu(data, algo, lead=0, first_line_index=0, flags=abi.EXG_F_BOF | abi.EXG_F_EOF, capacity=None,
            payload_base=BASE):
    from exon_duckdb_amd import device

    buf = bytes(data)
    d_in = device.upload(buf)
    scan = device.FastqScan(len(buf), capacity_records=capacity)
    scan.launch(d_in, lead=lead, first_line_index=first_line_index, payload_base=payload_base, flags=flags, algo=algo)
    res = scan.fetch()
    n = int(res.n_records)
    cols, words = scan.columns_host(n)
    return res, cols, words


def check_against_oracle(oracle, data, algo, expect_fallback=None):
    data = bytes(data)
    exp = oracle.fastq_parse(data, payload_base=BASE)
    res, cols, words = run_gpu(data, algo)
    # no launch is given up any more: long records, dense halves and bytes >= 0x80 (UTF-8 validation) are the any-shape scan's
    assert not (res.flags & abi.EXG_RF_FALLBACK), "a fused launch asked for the general path"
    assert res.error_code == exp.error_code, (res.error_code, exp.error_code, exp.error_message)
    assert res.n_records == exp.n_rows
    if exp.error_code:
        assert res.error_record == exp.error_record
        assert res.error_offset == exp.error_offset
    for k, name in enumerate(NAMES):
        want, want_words = exp.string_t[name]
        assert np.array_equal(cols[k], want), f"column {name} differs (algo {algo})"
        if name == "description":
            nw = (exp.n_rows + 63) // 64
            got = words[:nw].copy()
            if exp.n_rows % 64:
                got[-1] &= np.uint64((1 << (exp.n_rows % 64)) - 1)   # bits of rows past an error are unspecified
            assert np.array_equal(got, want_words[:nw]), "description validity differs"
    if not exp.error_code:
        assert res.consumed_bytes == len(data)
        assert bool(res.flags & abi.EXG_RF_NON_ASCII) == any(b >= 0x80 for b in data)
    return res


# ---- fixtures of the reference ---------------------------------------------------------------------

@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("name", ["test.fastq", "test2.fastq", "fastq/copy-a.fastq", "fastq/copy-b.fastq"])
def test_reference_fixtures(gpu, oracle, golden_dir, name, algo):
    with open(os.path.join(golden_dir, name), "rb") as f:
        data = f.read()
    res = check_against_oracle(oracle, data, algo)
    assert res.n_records == 2          # test_fastq_scan.test:5-8


@pytest.mark.parametrize("algo", ALGOS)
def test_reference_row0_values(gpu, oracle, golden_dir, algo):
    # test_fastq_scan.test:35-41 straight from the device output (no oracle in between)
    with open(os.path.join(golden_dir, "test.fastq"), "rb") as f:
        data = f.read()
    res, cols, words = run_gpu(data, algo)
    def text(col, row):
        st = cols[col][row]
        ln = int(st[:4].view(np.uint32)[0])
        if ln <= 12:
            return st[4:4 + ln].tobytes().decode()
        off = int(st[8:16].view(np.uint64)[0]) - BASE
        assert st[4:8].tobytes() == data[off:off + 4]
        return data[off:off + ln].decode()
    assert [text(c, 0) for c in range(4)] == [
        "SEQ_ID", "This is a description",
        "GATTTGGGGTTCAAAGCAGTATCGATCAAATAGTAAATCCATTTGTTCAACTCACAGTTT",
        "!''*((((***+))%%%++)(%%%%).1***-+*''))**55CCF>>>>>>CCCCCCC65"]
    assert int(words[0]) & 3 == 1       # row 0 has a description, row 1 does not


# ---- synthetic FASTQ-150 (BASELINE config shape) at oracle-friendly sizes ------------------------------

@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n_records", [1, 3, 49, 50, 197, 198, 1000, 20011])
def test_synth_fastq150(gpu, oracle, n_records, algo):
    data = oracle.synth_fastq(332 * n_records)
    res = check_against_oracle(oracle, data, algo)
    assert res.n_records == n_records and res.n_lines == 4 * n_records


@pytest.mark.parametrize("algo", ALGOS)
def test_synth_fastq150_truncated_mid_record(gpu, oracle, algo):
    # cut inside the sequence line of the last record: UnexpectedEof on that record
    data = oracle.synth_fastq(332 * 100 + 60)
    res = check_against_oracle(oracle, data, algo)
    assert res.error_code == abi.EXG_PE_UNEXPECTED_EOF and res.n_records == 100


def test_device_generator_matches_oracle(gpu, oracle):
    from exon_duckdb_amd import device

    for off, n in [(0, 332 * 300), (12345, 100001), (332 * 10**9 + 7, 5000)]:
        t = device.synth_fastq(n, file_offset=off)
        got = t[:n].cpu().numpy()
        assert np.array_equal(got, oracle.synth_fastq(n, file_offset=off))


# ---- ragged inputs: CRLF, missing descriptions, inline-length fields, no final newline ------------------

@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n_records", [1, 7, 64, 65, 500, 5000])
def test_ragged(gpu, oracle, n_records, algo):
    data = oracle.synth_fastq_ragged(n_records)
    res = check_against_oracle(oracle, data, algo)
    assert res.n_records == n_records


# ---- the [RECALLED] edge rules, device vs oracle -------------------------------------------------------

EDGE_CASES = {
    "empty": b"",
    "one_record": b"@a\nAC\n+\n!!\n",
    "no_trailing_newline": b"@a\nAC\n+\n!!",
    "cr_before_eof_kept": b"@a\nAC\n+\n!!\r",
    "crlf": b"@id d\r\nAC\r\n+\r\n!!\r\n@id2\r\nACGT\r\n+\r\n!!!!\r\n",
    "empty_name": b"@\nAC\n+\n!!\n@ d\nAC\n+\n!!\n",
    "trailing_space_null_desc": b"@id \nAC\n+\n!!\n",
    "split_first_space_only": b"@id a b  c\nAC\n+\n!!\n",
    "tab_not_delimiter": b"@id\tx\nAC\n+\n!!\n",
    "plus_line_content": b"@id\nAC\n+id again\n!!\n",
    "quality_starts_with_at": b"@a\nAC\n+\n@!\n@b\nGT\n+\n@@\n",
    "missing_quality_line": b"@a\nAC\n+\n",
    "missing_quality_line_no_nl": b"@a\nAC\n+",
    "empty_sequence_and_quality": b"@a\n\n+\n\n@b\n\n+\n\n",
    "truncated_1_line": b"@x\nAC\n+\n!!\n@a\n",
    "truncated_1_line_no_nl": b"@x\nAC\n+\n!!\n@a",
    "truncated_2_lines": b"@x\nAC\n+\n!!\n@a\nAC\n",
    "truncated_2_lines_no_nl": b"@x\nAC\n+\n!!\n@a\nAC",
    "bad_name_prefix": b"@x\nAC\n+\n!!\nx\nAC\n+\n!!\n",
    "blank_line_between": b"@x\nAC\n+\n!!\n\n@y\nAC\n+\n!!\n",
    "trailing_blank_line": b"@x\nAC\n+\n!!\n\n",
    "bad_plus_prefix": b"@x\nAC\n-\n!!\n",
    "empty_plus_line": b"@x\nAC\n\n!!\n",
    "first_error_wins": b"@x\nAC\n+\n!!\n@y\nAC\n-\n!!\nz\nAC\n+\n!!\n",
    "utf8_ok": "@é ü\nAC\n+\n!!\n".encode(),
    "utf8_bad_desc": b"@x\nAC\n+\n!!\n@y \xff\nAC\n+\n!!\n",
    "utf8_bad_seq": b"@x\nA\xc3\n+\n!!\n",
    "utf8_bad_but_plus_line_only": b"@x\nAC\n+\xff\xfe\n!!\n",
    "structural_beats_utf8": b"@y \xff\nAC\n-\n!!\n",
    "inline_lengths": b"".join(b"@" + b"n" * k + b" " + b"d" * (13 - k) + b"\n" + b"A" * k + b"\n+\n" + b"!" * k + b"\n"
                               for k in range(0, 14)),
    "only_newlines": b"\n" * 37,
    "truncated_bad_prefix_2_lines": b"@x\nAC\n+\n!!\nx\nAC\n",
    "truncated_bad_prefix_1_line": b"@x\nAC\n+\n!!\nxyz",
    "name_only_at": b"@",
}


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("case", sorted(EDGE_CASES))
def test_edge_cases(gpu, oracle, case, algo):
    check_against_oracle(oracle, EDGE_CASES[case], algo)


# ---- records larger than the fused kernel's LDS window, halves with more lines than its list: still one pass ----------
# (more shapes: tests/test_record_shapes_gpu.py)

@pytest.mark.parametrize("algo", ALGOS)
def test_long_reads_stay_on_the_single_pass(gpu, oracle, algo):
    rng = np.random.default_rng(7)
    recs = []
    for k in range(40):
        ln = int(rng.integers(1, 60000)) if k % 3 else int(rng.integers(1, 200))
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), ln).tobytes()
        recs.append(b"@long%d desc\n" % k + seq + b"\n+\n" + b"I" * ln + b"\n")
    data = b"".join(recs)
    res = check_against_oracle(oracle, data, algo)
    assert res.n_records == 40 and not (res.flags & abi.EXG_RF_FALLBACK)


@pytest.mark.parametrize("algo", ALGOS)
def test_many_tiny_lines_in_one_tile(gpu, oracle, algo):
    # 10 900 newlines per 16 KiB half: the fused kernel's list holds 512, the half is emitted in 22 passes
    data = b"@a\n\n+\n\n" * 5000
    res = check_against_oracle(oracle, data, algo)
    assert res.n_records == 5000 and not (res.flags & abi.EXG_RF_FALLBACK)


# ---- byte-range shards with a halo (multi-GPU layout) and record-aligned batches ---------------------------

@pytest.mark.parametrize("algo", [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_FULL])
@pytest.mark.parametrize("ragged", [False, True])
def test_unaligned_shards_reassemble(gpu, oracle, algo, ragged):
    n_rec = 3000
    data = bytes(oracle.synth_fastq_ragged(n_rec) if ragged else oracle.synth_fastq(332 * n_rec))
    exp = oracle.fastq_parse(data, payload_base=BASE)
    arr = np.frombuffer(data, np.uint8)
    nl = np.flatnonzero(arr == 10)
    n = len(data)
    cuts = [0, 16 * 1021, 16 * 4099 + 16, 16 * 20000, n]   # 16-byte aligned, NOT record aligned
    halo = 1024 if not ragged else 2048
    got_cols = [[] for _ in range(4)]
    got_valid = []
    total = 0
    for s, e in zip(cuts[:-1], cuts[1:]):
        h = min(halo, s)
        h -= h % 16
        buf = data[s - h:e]
        fli = int(np.searchsorted(nl, s))           # '\n' before the shard start
        flags = (abi.EXG_F_BOF if s - h == 0 else 0) | (abi.EXG_F_EOF if e == n else 0)
        res, cols, words = run_gpu(buf, algo, lead=h, first_line_index=fli, flags=flags,
                                   payload_base=BASE + s - h)
        assert res.error_code == 0 and not (res.flags & (abi.EXG_RF_HEAD_UNRESOLVED | abi.EXG_RF_FALLBACK))
        k = int(res.n_records)
        for c in range(4):
            got_cols[c].append(cols[c])
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[:k]
        got_valid.append(bits)
        total += k
    assert total == exp.n_rows == n_rec
    for c, name in enumerate(NAMES):
        assert np.array_equal(np.concatenate(got_cols[c]), exp.string_t[name][0]), name
    assert np.array_equal(np.concatenate(got_valid), exp.columns["description"].valid)


@pytest.mark.parametrize("algo", [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_FULL])
def test_head_record_before_buffer_is_reported(gpu, oracle, algo):
    data = bytes(oracle.synth_fastq(332 * 200))
    s = 16 * 700   # mid record, no halo and no BOF: the first owned record cannot be resolved
    res, cols, words = run_gpu(data[s:], algo, lead=0, first_line_index=int((np.frombuffer(data[:s], np.uint8) == 10).sum()),
                               flags=abi.EXG_F_EOF, payload_base=BASE + s)
    assert res.flags & abi.EXG_RF_HEAD_UNRESOLVED
    exp = oracle.fastq_parse(data, payload_base=BASE)
    first = (s + 331) // 332 if s % 332 else s // 332      # first record that starts inside the buffer
    k = int(res.n_records)
    assert k == 200 - first + 1
    assert np.array_equal(cols[2][1:], exp.string_t["sequence"][0][first:])
    assert not cols[0][0].any()          # the unresolved row is zeroed


@pytest.mark.parametrize("algo", [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_FULL])
def test_record_aligned_batches(gpu, oracle, algo):
    # streaming: not at EOF, the tail record is incomplete; consumed_bytes says where to resume
    data = bytes(oracle.synth_fastq_ragged(400))
    cut = len(data) * 2 // 3
    res, cols, words = run_gpu(data[:cut], algo, flags=abi.EXG_F_BOF)
    exp = oracle.fastq_parse(data, payload_base=BASE)
    k = int(res.n_records)
    assert 0 < k < 400 and res.error_code == 0
    assert np.array_equal(cols[0], exp.string_t["name"][0][:k])
    resume = int(res.consumed_bytes)
    assert resume == int(exp.columns["name"].src_off[k]) - 1     # the '@' of the next record
    res2, cols2, _ = run_gpu(data[resume:], algo, flags=abi.EXG_F_BOF | abi.EXG_F_EOF, payload_base=BASE + resume)
    assert int(res2.n_records) == 400 - k
    assert np.array_equal(cols2[3], exp.string_t["quality_scores"][0][k:])


@pytest.mark.parametrize("algo", [abi.EXG_ALGO_FUSED, abi.EXG_ALGO_MULTIPASS, abi.EXG_ALGO_FUSED_FULL])
def test_capacity_is_enforced(gpu, oracle, algo):
    data = oracle.synth_fastq(332 * 500)
    res, cols, _ = run_gpu(data, algo, capacity=123)
    assert res.flags & abi.EXG_RF_CAPACITY and res.n_records == 123
    exp = oracle.fastq_parse(data, payload_base=BASE)
    assert np.array_equal(cols[0], exp.string_t["name"][0][:123])


# ---- size-independent properties at a size the oracle does not parse -----------------------------------

def test_large_buffer_properties(gpu):
    """1 GiB of FASTQ-150 generated in HBM: record count, line count, analytic string_t of sampled
    records (record k starts at byte 332 k), fused == multipass on the whole output."""
    import torch
    from exon_duckdb_amd import device

    n_rec = (1 << 30) // 332
    n = n_rec * 332
    d_in = device.synth_fastq(n)
    scan = device.FastqScan(n, capacity_records=n_rec + 8)
    scan.launch(d_in, payload_base=BASE, algo=abi.EXG_ALGO_FUSED)
    res = scan.fetch()
    assert res.error_code == 0 and res.n_records == n_rec and res.n_lines == 4 * n_rec
    assert res.consumed_bytes == n and res.flags == 0
    fused = [c[:n_rec].clone() for c in scan.cols]
    fused_valid = scan.validity[: (n_rec + 63) // 64].clone()
    # analytic check of every record's string_t header (length + pointer), on the device
    k = torch.arange(n_rec, device="cuda", dtype=torch.int64)
    for col, (off, ln) in zip(fused, [(1, 15), (17, 10), (28, 150), (181, 150)]):
        lengths = col[:, 0] & 0xFFFFFFFF
        assert bool((lengths == ln).all())
        if ln > 12:
            assert bool((col[:, 1] == BASE + 332 * k + off).all())
    assert bool((fused_valid[:-1] == -1).all())
    scan.launch(d_in, payload_base=BASE, algo=abi.EXG_ALGO_MULTIPASS)
    res2 = scan.fetch()
    assert res2.n_records == n_rec
    for a, b in zip(fused, scan.cols):
        assert torch.equal(a, b[:n_rec])
    assert torch.equal(fused_valid, scan.validity[: (n_rec + 63) // 64])


def test_config2_full_size_properties(gpu):
    """BASELINE.json configs[1] at its full size: 9 999 999 692 bytes = 30 120 481 records of FASTQ-150 generated in
    HBM.  The oracle does not parse 10 GB; checked through size-independent properties: record / line counts,
    every record's string_t (length, and for out-of-line strings the pointer = base + 332 k + field offset), the
    4-byte prefix of every name against its closed form, all-valid descriptions, and a checksum of checksums
    that equals the same expression evaluated on the first GiB alone scaled by position arithmetic."""
    import torch
    from exon_duckdb_amd import device

    n_rec = 30_120_481
    n = n_rec * 332
    assert n == 9_999_999_692
    d_in = device.synth_fastq(n)
    scan = device.FastqScan(n, capacity_records=n_rec + 8)
    scan.launch(d_in, payload_base=BASE, algo=abi.EXG_ALGO_AUTO)
    res = scan.fetch()
    assert res.error_code == 0 and res.n_records == n_rec and res.n_lines == 4 * n_rec
    assert res.consumed_bytes == n and not (res.flags & abi.EXG_RF_FALLBACK)
    k = torch.arange(n_rec, device="cuda", dtype=torch.int64)
    for col, (off, ln) in zip(scan.cols, [(1, 15), (17, 10), (28, 150), (181, 150)]):
        c = col[:n_rec]
        assert bool(((c[:, 0] & 0xFFFFFFFF) == ln).all())
        if ln > 12:
            assert bool((c[:, 1] == BASE + 332 * k + off).all())
    # names are "SYN" + 12 decimal digits of k: the string_t prefix (bytes 4..7 of the struct) is "SYN" + first digit
    first_digit = (k // 10 ** 11) % 10
    want_prefix = 0x53 | (0x59 << 8) | (0x4E << 16) | ((0x30 + first_digit) << 24)
    assert bool((((scan.cols[0][:n_rec, 0] >> 32) & 0xFFFFFFFF) == want_prefix).all())
    # descriptions "d:N:0:ACGT" with d = k mod 4 are inlined: their first payload dword is d ':' 'N' ':'
    want_desc = (0x30 + (k & 3)) | (0x3A << 8) | (0x4E << 16) | (0x3A << 24)
    assert bool((((scan.cols[1][:n_rec, 0] >> 32) & 0xFFFFFFFF) == want_desc).all())
    words = (n_rec + 63) // 64
    assert bool((scan.validity[: words - 1] == -1).all())
    # sequence / quality prefixes equal the input bytes they point at (gather from the input itself)
    flat = d_in[:n].view(torch.uint8)
    idx = torch.randint(0, n_rec, (1 << 20,), device="cuda", dtype=torch.int64)
    for col, off in ((scan.cols[2], 28), (scan.cols[3], 181)):
        pref = (col[idx, 0] >> 32) & 0xFFFFFFFF
        got = sum(flat[332 * idx + off + j].to(torch.int64) << (8 * j) for j in range(4))
        assert bool((pref == got).all())


def _newlines_in(lo, hi):
    """'\\n' bytes of the synthetic FASTQ-150 file in [lo, hi): a record has them at offsets 27, 178, 180 and 331"""
    def upto(x):   # newlines in [0, x)
        q, r = divmod(x, 332)
        return 4 * q + sum(1 for o in (27, 178, 180, 331) if o < r)
    return upto(hi) - upto(lo)


@pytest.mark.parametrize("rank", [3, 7])
def test_config5_shard_full_size_properties(gpu, rank):
    """BASELINE.json configs[4] as ONE of its eight GPUs sees it: shard `rank` of plan_shards(100 GB, 8) —
    12.5 GB generated in HBM at its file offset, cut at 16-byte (NOT record) boundaries, a 1 KiB halo in front as `lead`,
    no BOF (and no EOF for the middle shard 3; shard 7 is the file's last), the 4-line phase guessed from the shard's own
    bytes by exg_fastq_guess_phase exactly as bench.py --gpus 8 does it.  Checked through the closed forms of
    test_config2_full_size_properties: the shard owns the records whose last line ENDS in it, rows in file order."""
    import ctypes as C

    import torch
    from exon_duckdb_amd import device, load_library, sharding

    lib = load_library()
    file_bytes = int(8 * 12.5e9) // 332 * 332
    assert file_bytes // 332 == 301_204_819          # config 5: 100 GB of 332-byte records
    sh = sharding.plan_shards(file_bytes, 8, halo=1024)[rank]
    assert sh.halo == 1024 and not sh.is_first and sh.start % 332 != 0 and sh.is_last == (rank == 7)
    n_bytes = sh.n_bytes
    d_in = device.synth_fastq(n_bytes, file_offset=sh.load_offset)
    first = sh.start // 332                           # the record that holds byte `start` ends at or behind it
    n_rec = sh.end // 332 - first                     # ... and the one that straddles `end` is the next shard's
    scan = device.FastqScan(n_bytes, capacity_records=n_rec + 16)
    ph = torch.zeros(1, dtype=torch.int32, device="cuda")
    device.check(lib.exg_fastq_guess_phase(C.c_void_p(d_in.data_ptr()), n_bytes, sh.halo, C.c_void_p(ph.data_ptr()), device.stream_ptr()))
    guess = int(ph.item()) & 0xFFFFFFFF
    assert guess < 4
    prev_is_nl = bool(int(d_in[sh.halo - 1].item()) == 10)
    fli = guess if prev_is_nl else (guess - 1) % 4
    # the exact line index of the line that holds the shard's first byte, from the generator's geometry
    assert fli == _newlines_in(0, sh.start) % 4
    flags = abi.EXG_F_EOF if sh.is_last else 0
    base = BASE + sh.load_offset
    for algo in (abi.EXG_ALGO_FUSED, abi.EXG_ALGO_FUSED_FULL):
        scan.launch(d_in, n_bytes=n_bytes, lead=sh.halo, first_line_index=fli, payload_base=base, flags=flags, algo=algo)
        res = scan.fetch()
        assert res.error_code == 0 and not (res.flags & (abi.EXG_RF_FALLBACK | abi.EXG_RF_HEAD_UNRESOLVED)), (res.error_code, res.flags)
        assert res.n_records == n_rec, (res.n_records, n_rec)
        assert res.n_lines == _newlines_in(sh.start, sh.end)
        k = first + torch.arange(n_rec, device="cuda", dtype=torch.int64)
        for col, (off, ln) in zip(scan.cols, [(1, 15), (17, 10), (28, 150), (181, 150)]):
            c = col[:n_rec]
            assert bool(((c[:, 0] & 0xFFFFFFFF) == ln).all())
            if ln > 12:
                assert bool((c[:, 1] == BASE + 332 * k + off).all())
        first_digit = (k // 10 ** 11) % 10
        want_prefix = 0x53 | (0x59 << 8) | (0x4E << 16) | ((0x30 + first_digit) << 24)
        assert bool((((scan.cols[0][:n_rec, 0] >> 32) & 0xFFFFFFFF) == want_prefix).all())
        want_desc = (0x30 + (k & 3)) | (0x3A << 8) | (0x4E << 16) | (0x3A << 24)
        assert bool((((scan.cols[1][:n_rec, 0] >> 32) & 0xFFFFFFFF) == want_desc).all())
        # the inlined description's remaining bytes: "0:ACGT" + two zero bytes (bytes 8..15 of the string_t)
        assert bool((scan.cols[1][:n_rec, 1] == int.from_bytes(b"0:ACGT\0\0", "little")).all())
        assert bool((scan.validity[: (n_rec + 63) // 64 - 1] == -1).all())
        flat = d_in[:n_bytes].view(torch.uint8)
        idx = torch.randint(0, n_rec, (1 << 20,), device="cuda", dtype=torch.int64)
        for col, off in ((scan.cols[2], 28), (scan.cols[3], 181)):
            pref = (col[idx, 0] >> 32) & 0xFFFFFFFF
            got = sum(flat[332 * (first + idx) - sh.load_offset + off + j].to(torch.int64) << (8 * j) for j in range(4))
            assert bool((pref == got).all())
        del k, first_digit, want_prefix, want_desc, idx
    # the eight shards' row counts add up to the file's
    assert sum(s.end // 332 - (0 if s.is_first else s.start // 332) for s in sharding.plan_shards(file_bytes, 8, halo=1024)) == file_bytes // 332
