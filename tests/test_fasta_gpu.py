"""Parity of the HIP FASTA scan (C-ABI exg_fasta_scan) against the oracle.

id / description are slices of the input (ptr = BASE + offset); the sequence is the record's lines
concatenated in a compacted payload buffer (ptr = SEQ_BASE + payload offset).  The oracle's Arrow-style
columns give lengths + bytes + validity; string_t values are resolved and compared field by field, and
the whole payload buffer must equal the concatenation of the oracle's sequence values."""
import os

import numpy as np
import pytest

from exon_duckdb_amd import abi

pytestmark = pytest.mark.gpu

BASE = 0x7D0000000000
SEQ_BASE = 0x7C0000000000


# both device implementations: EXG_ALGO_AUTO = one pass over super-tiles (k_fa_fused), EXG_ALGO_MULTIPASS = line index
ALGOS = [abi.EXG_ALGO_AUTO, abi.EXG_ALGO_MULTIPASS]


def run_gpu(data, capacity=None, flags=abi.EXG_F_BOF | abi.EXG_F_EOF, algo=abi.EXG_ALGO_AUTO):
    from exon_duckdb_amd import device

    data = bytes(data)
    d_in = device.upload(data)
    scan = device.FastaScan(len(data), capacity_records=capacity)
    scan.launch(d_in, payload_base=BASE, seq_payload_base=SEQ_BASE, flags=flags, algo=algo)
    res = scan.fetch()
    cols, words, payload = scan.host(int(res.n_records), int(res.payload_bytes))
    return res, cols, words, payload


def resolve(st, data, payload):
    ln = int(st[:4].view(np.uint32)[0])
    if ln <= 12:
        assert not st[4 + ln:].any(), "inlined string_t must be zero padded"
        return st[4:4 + ln].tobytes()
    ptr = int(st[8:16].view(np.uint64)[0])
    if ptr >= BASE:
        off, buf = ptr - BASE, data
    else:
        off, buf = ptr - SEQ_BASE, payload
    got = bytes(buf[off:off + ln])
    assert st[4:8].tobytes() == got[:4], "prefix must be the first 4 bytes"
    return got


def check(oracle, data, algos=None):
    for algo in (algos or ALGOS):
        res = check_one(oracle, data, algo)
    return res


def check_one(oracle, data, algo):
    data = bytes(data)
    exp = oracle.fasta_parse(data)
    res, cols, words, payload = run_gpu(data, algo=algo)
    assert res.error_code == exp.error_code, (res.error_code, exp.error_code, exp.error_message)
    assert res.n_records == exp.n_rows
    if exp.error_code:
        assert res.error_record == exp.error_record and res.error_offset == exp.error_offset
    n = exp.n_rows
    valid = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
    assert np.array_equal(valid, exp.columns["description"].valid)
    pl = payload.tobytes()
    for k, name in enumerate(["id", "description", "sequence"]):
        col = exp.columns[name]
        for i in range(n):
            want = col.row(i)
            if want is None:
                assert not cols[k][i].any()          # NULL rows are 16 zero bytes
            else:
                assert resolve(cols[k][i], data, pl) == want, (name, i)
    if not exp.error_code:
        assert pl == exp.columns["sequence"].values.tobytes()
        assert res.consumed_bytes == len(data)
    return res


@pytest.mark.parametrize("name", ["test.fasta", "test.mixed-desc.fasta"])
def test_reference_fixtures(gpu, oracle, golden_dir, name):
    with open(os.path.join(golden_dir, name), "rb") as f:
        data = f.read()
    res = check(oracle, data)
    assert res.n_records == 2      # test_fasta_scan.test:5-8


def test_reference_null_description_row(gpu, golden_dir):
    # test_fasta_copy.test:75-80: (b, NULL, ATCG), straight from the device output
    with open(os.path.join(golden_dir, "test.mixed-desc.fasta"), "rb") as f:
        data = f.read()
    res, cols, words, payload = run_gpu(data)
    assert int(words[0]) & 3 == 1
    assert resolve(cols[0][1], data, payload.tobytes()) == b"b"
    assert not cols[1][1].any()
    assert resolve(cols[2][1], data, payload.tobytes()) == b"ATCG"


@pytest.mark.parametrize("n_records", [1, 2, 50, 700])
def test_synth_fasta(gpu, oracle, n_records):
    res = check(oracle, oracle.synth_fasta(n_records))
    assert res.n_records == n_records


def test_config1_one_megabyte(gpu, oracle):
    # BASELINE config 1 shape: SELECT COUNT(*) FROM read_fasta() on a 1 MB FASTA
    data = bytes(oracle.synth_fasta(620))
    assert 0.9e6 < len(data) < 1.3e6
    res = check(oracle, data)
    from exon_duckdb_amd import device
    d_in = device.upload(data)
    scan = device.FastaScan(len(data))
    scan.launch(d_in, flags=abi.EXG_F_BOF | abi.EXG_F_EOF | abi.EXG_F_NO_STORE)
    assert scan.fetch().n_records == res.n_records == 620


EDGE = {
    "empty": b"",
    "multiline": b">a d\nAC\nGT\n\nTT\n>b\nA\n",
    "desc_trim_and_tab": b">a\t  two words \nAC\n",
    "trailing_space_empty_desc": b">a \nAC\n",
    "crlf": b">a d\r\nAC\r\nGT\r\n>b\r\nA\r\n",
    "empty_sequence": b">a\n>b\nAC\n>c\n",
    "no_trailing_newline": b">a d\nACGT",
    "cr_before_eof_kept": b">a\nAC\r",
    "gt_inside_sequence": b">a\nAC>GT\n",
    "missing_prefix": b"ACGT\n>a\nAC\n",
    "empty_first_line": b"\n>a\nAC\n",
    "empty_first_line_crlf": b"\r\n>a\nAC\n",
    "missing_name": b">a\nAC\n> desc only\nAC\n",
    "missing_name_bare": b">a\nAC\n>\nAC\n",
    "bad_utf8_definition": b">a \xff\nAC\n",
    "bad_utf8_sequence": b">a\nAC\n>b\nA\xc3\n",
    "utf8_ok": ">é ü \nAC\n".encode(),
    "unicode_trim": ">a 　x y \nAC\n".encode(),
    "inline_lengths": b"".join(b">" + b"n" * k + b" " + b"d" * (13 - k) + b"\n" + b"A" * k + b"\n" for k in range(1, 14)),
    "long_single_line": b">chr1 one line\n" + b"ACGT" * 50000 + b"\n>chr2\n" + b"TTGCA" * 3000 + b"\n",
    "many_short_lines": b">x\n" + b"A\n" * 5000,
    "cr_mid_line_splits_definition": b">a\rb\nAC\n",
}


@pytest.mark.parametrize("case", sorted(EDGE))
def test_edge_cases(gpu, oracle, case):
    check(oracle, EDGE[case])


def test_capacity(gpu, oracle):
    data = bytes(oracle.synth_fasta(100))
    exp = oracle.fasta_parse(data)
    res, cols, words, payload = run_gpu(data, capacity=10)
    assert res.flags & abi.EXG_RF_CAPACITY and res.n_records == 10
    for i in range(10):
        assert resolve(cols[0][i], data, payload.tobytes()) == exp.columns["id"].row(i)


def test_what_a_buffer_may_be(gpu, oracle):
    """a buffer begins with a record (lead = 0, EXG_F_BOF); without EXG_F_EOF it is a batch (one-pass form only)"""
    from exon_duckdb_amd import device, ExgError

    data = bytes(oracle.synth_fasta(10))
    d_in = device.upload(data)
    scan = device.FastaScan(len(data))
    with pytest.raises(ExgError):
        scan.launch(d_in, flags=abi.EXG_F_EOF)                                       # not at a record start
    with pytest.raises(ExgError):
        scan.launch(d_in, flags=abi.EXG_F_BOF | abi.EXG_F_EOF, lead=16)
    with pytest.raises(ExgError):
        scan.launch(d_in, flags=abi.EXG_F_BOF, algo=abi.EXG_ALGO_MULTIPASS)          # the line-index form scans whole inputs


def test_open_tail_batches(gpu, oracle):
    """Without EXG_F_EOF the buffer is a batch of a longer input: the last record in it is still open (its sequence may
    go on behind the buffer) and is nobody's row yet — n_records / the columns / the payload are those of the records in
    front of it, consumed_bytes = where its definition line begins (the next batch starts there).  A batch that holds
    only one record reports nothing (consumed 0: the caller widens it).  Cuts at every kind of place: inside a sequence
    line, inside a definition line, right behind a newline, right in front of a '>'."""
    import random
    data = bytes(oracle.synth_fasta(400, seed=11))
    starts = [0] + [i + 1 for i in range(len(data) - 1) if data[i] == 10 and data[i + 1] == 62]
    rng = random.Random(5)
    cuts = [rng.randrange(1, len(data)) for _ in range(40)]
    cuts += [starts[7], starts[7] + 1, starts[7] - 1, starts[30] + 5, starts[1] - 3, len(data) - 1, 1, 17]
    for cut in cuts:
        buf = data[:cut]
        res, cols, words, payload = run_gpu(buf, flags=abi.EXG_F_BOF)
        # the open record: the last one whose '>' lies inside the buffer AND is announced by the newline in front of it
        inside = [s for s in starts if s < cut]
        last = inside[-1]
        exp = oracle.fasta_parse(data[:last])
        assert res.error_code == 0, (cut, res.error_code)
        assert res.n_records == exp.n_rows == len(inside) - 1, cut
        assert res.consumed_bytes == last, (cut, res.consumed_bytes, last)
        pl = payload.tobytes()
        assert pl == b"".join(exp.columns["sequence"].to_list()), cut
        n = exp.n_rows
        valid = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
        assert np.array_equal(valid, exp.columns["description"].valid)
        for k, name in enumerate(["id", "description", "sequence"]):
            col = exp.columns[name]
            for i in range(0, n, max(1, n // 25)):
                want = col.row(i)
                if want is None:
                    assert not cols[k][i].any()
                else:
                    assert resolve(cols[k][i], buf, pl) == want, (cut, name, i)


# ---- features placed exactly on the boundaries of the tiled implementation (16 B chunks, 1 KiB rows, 8 KiB wave spans,
# ---- 32 KiB super-tiles = workgroups; 16 KiB was the first form's tile) ----

def _pad_to(buf: bytearray, target: int, line=60):
    """append sequence lines until len(buf) == target (the last line is cut to fit, still newline-terminated)"""
    while len(buf) < target:
        room = target - len(buf)
        if room == 1:
            buf += b"\n"           # an empty line
            break
        n = min(line, room - 1)
        buf += b"ACGT" * (n // 4) + b"A" * (n % 4) + b"\n"
    assert len(buf) == target
    return buf


@pytest.mark.parametrize("boundary", [16, 1024, 4096, 8192, 16384, 24576, 32768, 65536, 98304])
@pytest.mark.parametrize("shift", [-2, -1, 0, 1])
def test_features_on_tile_boundaries(gpu, oracle, boundary, shift):
    # a definition line whose '>' sits at boundary + shift: the newline in front of it, the '>' and the line body
    # fall on different sides of a chunk / row / tile edge
    buf = bytearray(b">f r\n")
    _pad_to(buf, boundary + shift)
    buf += b">second some description\r\nACGTACGT\r\nTT\n>third\nGG"
    check(oracle, bytes(buf))
    # the same with the definition line itself running across the boundary
    buf = bytearray(b">f\n")
    _pad_to(buf, max(boundary + shift - 10, len(buf)))
    buf += b">a_long_identifier_that_crosses description text\nACGT\n"
    check(oracle, bytes(buf))


def test_tiles_without_a_newline(gpu, oracle):
    # single-line sequences far longer than a tile: dozens of tiles have no line start at all, in both states
    # (inside a sequence line, and inside a 40 KB definition line)
    rng = np.random.default_rng(3)
    seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 200_000))
    long_def = b">id " + b"x" * 40_000
    data = b">one\n" + seq + b"\n" + long_def + b"\n" + seq[:70_000] + b"\r\n>three d\n" + seq[:20] + b"\n"
    res = check(oracle, data)
    assert res.n_records == 3


def test_cr_and_lf_split_by_boundaries(gpu, oracle):
    for boundary in (64, 1024, 8192, 16384, 32768, 65536):
        buf = bytearray(b">r\n")
        _pad_to(buf, boundary - 20)
        buf += b"ACGTACGTACGTACGTACG\r\nACGT\r\n>x\r\nAA\r"      # "\r" is the last byte of the chunk / row / tile, "\n" the first of the next
        assert buf[boundary - 1:boundary + 1] == b"\r\n"
        check(oracle, bytes(buf))


@pytest.mark.parametrize("seed", range(12))
def test_multi_super_tile_fuzz(gpu, oracle, seed):
    """100-400 KB inputs (several super-tiles, so the scanner's prefixes, the line state across workgroups and the waves'
    own prefixes all matter) with line lengths from 1 to 20 000, CRLF, empty lines, definition lines of any length, and a
    few byte-level mutations: both device forms against the oracle."""
    rng = np.random.default_rng(7000 + seed)
    parts = []
    total = 0
    target = int(rng.integers(100_000, 400_000))
    while total < target:
        kind = rng.integers(0, 10)
        if kind < 2:
            ln = b">" + bytes(rng.choice(np.frombuffer(b"abcXYZ09 _\t", np.uint8), int(rng.integers(1, 300 if kind else 40_000))))
        else:
            width = int(rng.choice([1, 2, 15, 16, 17, 60, 61, 70, 80, 1023, 1024, 1025, 8191, 8192, 20_000]))
            ln = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), int(rng.integers(0, width + 1))))
        ln += b"\r\n" if rng.integers(0, 8) == 0 else b"\n"
        parts.append(ln)
        total += len(ln)
    data = b">first record\n" + b"".join(parts)
    check(oracle, data)
    import test_fuzz_gpu as FZ
    for _ in range(3):
        check(oracle, FZ.mutate(data, rng, int(rng.integers(1, 5))))


def test_large_input_both_forms_agree(gpu, oracle):
    """~100 MB (60 k records, > 6 000 tiles: several tile-scan workgroups, each composing the descriptors in front of it):
    the tiled form against the line-index form column for column on the device, and against the oracle through the
    payload bytes, the id / description / sequence lengths and the description validity."""
    import torch
    from exon_duckdb_amd import device

    n_rec = 60_000
    d_in, n = device.synth_fasta(n_rec)
    outs = []
    for algo in ALGOS:
        scan = device.FastaScan(n, capacity_records=n_rec + 16)
        scan.launch(d_in, payload_base=BASE, seq_payload_base=SEQ_BASE, algo=algo)
        res = scan.fetch()
        assert res.error_code == 0 and res.n_records == n_rec and res.consumed_bytes == n
        outs.append((scan, res))
    (s0, r0), (s1, r1) = outs
    assert r0.payload_bytes == r1.payload_bytes
    for k in range(3):
        assert torch.equal(s0.cols[k][:n_rec], s1.cols[k][:n_rec]), k
    nw = (n_rec + 63) // 64
    assert torch.equal(s0.validity[:nw - 1], s1.validity[:nw - 1])
    assert torch.equal(s0.payload[:r0.payload_bytes], s1.payload[:r1.payload_bytes])
    data = bytes(d_in[:n].cpu().numpy())
    exp = oracle.fasta_parse(data)
    assert exp.n_rows == n_rec and exp.error_code == 0
    cols, words, payload = s0.host(n_rec, int(r0.payload_bytes))
    assert payload.tobytes() == exp.columns["sequence"].values.tobytes()
    valid = np.unpackbits(words.view(np.uint8), bitorder="little")[:n_rec]
    assert np.array_equal(valid, exp.columns["description"].valid)
    for k, name in enumerate(["id", "description", "sequence"]):
        lens = cols[k][:, :4].copy().view(np.uint32)[:, 0]
        want = exp.columns[name].lengths() * (exp.columns[name].valid != 0)
        assert np.array_equal(lens, want), name
    # sequence string_t: pointers are the oracle's payload offsets
    seq = cols[2]
    ptr = seq[:, 8:16].copy().view(np.uint64)[:, 0]
    long_rows = exp.columns["sequence"].lengths() > 12
    assert np.array_equal(ptr[long_rows] - SEQ_BASE, exp.columns["sequence"].offsets[:-1][long_rows].astype(np.uint64))


def test_scanner_helping_path(gpu, oracle):
    """k_fa_fused's scanner computes a super-tile's aggregate itself when the super-tile's workgroup has not published it in
    time (progress must not depend on the dispatch order).  With EXG_FASTA_HELP_TICKS=0 it does so for every aggregate it
    does not find at once: the output must not change.  (A process of its own: the library reads the variable once.)"""
    import subprocess
    import sys
    code = (
        "import sys, os\n"
        "sys.path.insert(0, %r)\n"
        "import torch\n"
        "from exon_duckdb_amd import abi, device\n"
        "n_rec = 40000\n"
        "d_in, n = device.synth_fasta(n_rec)\n"
        "outs = []\n"
        "for algo in (abi.EXG_ALGO_AUTO, abi.EXG_ALGO_MULTIPASS):\n"
        "    s = device.FastaScan(n, capacity_records=n_rec + 16)\n"
        "    for rep in range(5 if algo == abi.EXG_ALGO_AUTO else 1):\n"
        "        s.launch(d_in, algo=algo)\n"
        "        r = s.fetch()\n"
        "        assert r.error_code == 0 and r.n_records == n_rec and r.consumed_bytes == n, (r.error_code, r.n_records)\n"
        "    outs.append((s, r))\n"
        "(a, ra), (b, rb) = outs\n"
        "assert ra.payload_bytes == rb.payload_bytes\n"
        "for k in range(3):\n"
        "    assert torch.equal(a.cols[k][:n_rec], b.cols[k][:n_rec]), k\n"
        "assert torch.equal(a.payload[:ra.payload_bytes], b.payload[:rb.payload_bytes])\n"
        "print('helping ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, EXG_FASTA_HELP_TICKS="0"))
    assert res.returncode == 0 and "helping ok" in res.stdout, res.stdout[-1000:] + res.stderr[-3000:]


@pytest.mark.parametrize("batch", [64 << 10, 300 << 10])
def test_reader_batches_uploaded_ahead(gpu, oracle, tmp_path, batch):
    # round 6: a text FASTA's next batch is uploaded — from exactly where this batch's last whole record ends, into the other input
    # slot — while this batch's joined sequences travel back (they leave by a kernel's stores, not by a copy engine): dozens of
    # batches, a record longer than a batch in the middle (the batch widens; no upload ahead for it), every row the oracle's
    from exon_duckdb_amd.reader import ShardReader
    a, b = bytes(oracle.synth_fasta(900, seed=11)), bytes(oracle.synth_fasta(700, seed=12))
    long_rec = b">long one\n" + b"\n".join(b"ACGTTGCA" * 8 for _ in range(12000)) + b"\n"   # 780 kB
    data = a + long_rec + b
    exp = oracle.fasta_parse(data)
    assert exp.error_code == 0
    want = list(zip(*(exp.columns[k].to_list() for k in ("id", "description", "sequence"))))
    p = tmp_path / "ahead.fasta"
    p.write_bytes(data)
    r = ShardReader(str(p), "fasta", device_batch_bytes=batch)
    got = r.rows()
    st = r.stats()
    r.close()
    assert len(got) == len(want) == 1601
    assert got == want
    assert st["device_batches"] > 10
    r = ShardReader(str(p), "fasta", device_batch_bytes=batch)
    assert r.count() == 1601
    r.close()


def test_reader_batches_uploaded_ahead_across_files_and_shards(gpu, oracle, tmp_path):
    # the upload ahead ends with a file (nothing of the next file is asked of this one's slots) and with a shard's run of records
    from exon_duckdb_amd.reader import ShardReader
    parts = [bytes(oracle.synth_fasta(400 + 150 * k, seed=20 + k)) for k in range(3)]
    d = tmp_path / "dir"
    d.mkdir()
    for k, p in enumerate(parts):
        (d / f"f{k}.fasta").write_bytes(p)
    want = []
    for p in parts:
        exp = oracle.fasta_parse(p)
        want += list(zip(*(exp.columns[c].to_list() for c in ("id", "description", "sequence"))))
    r = ShardReader(str(d) + "/", "fasta", device_batch_bytes=96 << 10)
    got = r.rows()
    r.close()
    assert got == want
    whole = tmp_path / "whole.fasta"
    whole.write_bytes(b"".join(parts))
    rows = []
    for s in range(5):
        r = ShardReader(str(whole), "fasta", shard_index=s, shard_count=5, device_batch_bytes=64 << 10)
        rows += r.rows()
        r.close()
    assert rows == want
