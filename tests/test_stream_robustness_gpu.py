"""The streaming sources and the fan-out under the things a consumer and a disk do to them: a reader that is closed in the
middle of its stream (the producer thread, its lanes and queued segments must go away at once — DuckDB closes a scan as soon
as a LIMIT is satisfied), a member / frame that is corrupt far into a stream that is decoded piece by piece (the rows in front
of it arrive, then the error — where a streaming decoder like the reference's flate2 / zstd readers reports it — and nothing
hangs), a file that ends in the middle of a member, and the bounded stream behind the reference's own FFI (`new_reader`)."""
import gzip
import time

import pytest

from test_streaming_gpu import CAP_MB, _bgzf, _open, _oracle_digest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fastq_mid(oracle):
    data = bytes(oracle.synth_fastq(332 * 200000))   # 66 MB: many segments under a 16 MiB cap
    exp = oracle.fastq_parse(data, want_string_t=False)
    return data, _oracle_digest(exp, ["name", "description", "sequence", "quality_scores"])


def _zstd_frames(data, frame=4 << 20, level=1):
    from zstd_util import compress
    return b"".join(compress(data[o:o + frame], level, True) for o in range(0, len(data), frame))


def _files(tmp_path, data):
    (tmp_path / "a.fastq.gz").write_bytes(_bgzf(data))
    (tmp_path / "b.fastq.gz").write_bytes(gzip.compress(data, 1, mtime=0))
    (tmp_path / "c.fastq.zst").write_bytes(_zstd_frames(data))
    (tmp_path / "d.fastq").write_bytes(data)
    return ["a.fastq.gz", "b.fastq.gz", "c.fastq.zst", "d.fastq"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fan", [False, True])
def test_close_in_the_middle_of_a_stream(gpu, fastq_mid, tmp_path, monkeypatch, fan):
    """one chunk, then exg_close: at once, and the next reader of the same file still returns every row"""
    from exon_duckdb_amd.reader import ShardReader
    data, want = fastq_mid
    names = _files(tmp_path, data)
    monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(CAP_MB))
    if fan:
        monkeypatch.setenv("EXON_GPU_SHARDS", "5")
        monkeypatch.setenv("EXG_FANOUT_WORKERS", "3")
    for name in names:
        if fan and name == "b.fastq.gz":
            continue   # (a single gzip member has no cut points: one stripe)
        for n_chunks in (0, 1, 40):
            r = ShardReader(str(tmp_path / name), "fastq", shard_count=0 if fan else 1)
            it = 0
            while it < n_chunks:
                from exon_duckdb_amd.table_function import Chunk
                import ctypes as C
                ch = Chunk()
                assert r._l.exg_next_chunk(r._r, C.byref(ch)) == 0
                assert ch.n_rows > 0
                r._l.exg_release_chunk(r._r, C.byref(ch))
                it += 1
            t0 = time.perf_counter()
            r.close()
            assert time.perf_counter() - t0 < 5.0, (name, n_chunks, "close waited for the stream")
        r = ShardReader(str(tmp_path / name), "fastq", shard_count=0 if fan else 1)
        assert r.digest() == want, name
        r.close()


@pytest.mark.timeout(300)
def test_close_in_the_middle_of_overlapped_zstd_rounds(gpu, fastq_mid, tmp_path, monkeypatch):
    """without a memory cap two zstd rounds overlap (round n + 1's entropy stages and execution beside round n's resolve launches,
    which a thread of their own issues; three compressed windows, a read-ahead thread): exg_close with rounds in flight returns at
    once, nothing is left behind for the next reader, and a whole read gives every row"""
    from exon_duckdb_amd.reader import ShardReader
    from exon_duckdb_amd.table_function import Chunk
    import ctypes as C
    data, want = fastq_mid
    _files(tmp_path, data)
    monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(1 << 20))   # rounds of ~1 MiB: dozens of them in flight one after the other
    p = str(tmp_path / "c.fastq.zst")
    for n_chunks in (0, 1, 7, 60):
        r = ShardReader(p, "fastq")
        for _ in range(n_chunks):
            ch = Chunk()
            assert r._l.exg_next_chunk(r._r, C.byref(ch)) == 0
            assert ch.n_rows > 0
            r._l.exg_release_chunk(r._r, C.byref(ch))
        t0 = time.perf_counter()
        r.close()
        assert time.perf_counter() - t0 < 5.0, (n_chunks, "close waited for the stream")
    for _ in range(2):
        r = ShardReader(p, "fastq")
        assert r.digest() == want
        r.close()


@pytest.mark.timeout(300)
def test_early_close_of_a_zstd_reader_beside_another_reader_of_the_device(gpu, fastq_mid, tmp_path, monkeypatch):
    """advisor, round 4: a multi-round .zst reader that is closed after its first batch may have a read-ahead window travelling
    on its I/O stream; its pinned block and device window go back to the process-wide pools when the producer unwinds.  A second
    reader of the same device takes blocks from those pools all the time: had a DMA still been reading / writing a returned
    block, that reader's rows would be damaged.  Many early closes beside a reader that is digested again and again."""
    import threading
    from exon_duckdb_amd.reader import ShardReader
    from exon_duckdb_amd.table_function import Chunk
    import ctypes as C
    data, want = fastq_mid
    _files(tmp_path, data)
    monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(2 << 20))
    p = str(tmp_path / "c.fastq.zst")
    bad, stop = [], threading.Event()

    def other():
        while not stop.is_set():
            for name in ("c.fastq.zst", "a.fastq.gz"):
                r = ShardReader(str(tmp_path / name), "fastq")
                if r.digest() != want:
                    bad.append(name)
                r.close()

    th = threading.Thread(target=other)
    th.start()
    try:
        for it in range(40):
            r = ShardReader(p, "fastq")
            for _ in range(1 + it % 3):
                ch = Chunk()
                assert r._l.exg_next_chunk(r._r, C.byref(ch)) == 0 and ch.n_rows > 0
                r._l.exg_release_chunk(r._r, C.byref(ch))
            r.close()
    finally:
        stop.set()
        th.join()
    assert not bad, bad


def _expect_error(path, fmt, min_rows_before, compression=None, **kw):
    """reads to the error: returns (rows delivered before it, message)"""
    import ctypes as C

    from exon_duckdb_amd._lib import ExgError
    from exon_duckdb_amd.table_function import Chunk
    r = _open(path, fmt, compression=compression, **kw)
    rows = 0
    try:
        while True:
            ch = Chunk()
            rc = r._l.exg_next_chunk(r._r, C.byref(ch))
            if rc != 0:
                msg = (r._l.exg_reader_error(r._r) or b"").decode("utf-8", "replace")
                assert msg, "an error code without a message"
                assert rows >= min_rows_before, (rows, msg)
                return rows, msg
            if ch.n_rows == 0:
                raise AssertionError(f"the corrupt input was read to its end without an error ({rows} rows)")
            rows += int(ch.n_rows)
            r._l.exg_release_chunk(r._r, C.byref(ch))
    except ExgError:
        raise
    finally:
        t0 = time.perf_counter()
        r.close()
        assert time.perf_counter() - t0 < 5.0


@pytest.mark.timeout(300)
def test_corruption_far_into_a_capped_stream(gpu, fastq_mid, tmp_path, monkeypatch):
    """BGZF: a payload byte of a member at 70 % of the file (its CRC-32 no longer matches, or its codes break); single
    member: a byte at 70 %; zstd: a byte inside the frame at 70 %; a file cut in the middle of a member.  The rows of the
    segments in front arrive, then the error; the reader closes at once; a clean file reads fine afterwards."""
    data, want = fastq_mid
    n_rows = want[0]
    monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(CAP_MB))
    cases = []
    bg = bytearray(_bgzf(data))
    at = int(len(bg) * 0.7)
    bg[at] ^= 0x5A
    cases.append(("bgzf_flip.fastq.gz", bytes(bg)))
    cases.append(("bgzf_cut.fastq.gz", _bgzf(data)[: int(len(bg) * 0.7)]))
    one = bytearray(gzip.compress(data, 1, mtime=0))
    one[int(len(one) * 0.7)] ^= 0x5A
    cases.append(("one_flip.fastq.gz", bytes(one)))
    cases.append(("one_cut.fastq.gz", gzip.compress(data, 1, mtime=0)[: int(len(one) * 0.7)]))
    zs = bytearray(_zstd_frames(data))
    zs[int(len(zs) * 0.7)] ^= 0x5A
    cases.append(("z_flip.fastq.zst", bytes(zs)))
    cases.append(("z_cut.fastq.zst", _zstd_frames(data)[: int(len(zs) * 0.7)]))
    from zstd_util import compress
    z1 = compress(data, 1, True)                      # one frame, like the zstd CLI writes: the blocks in front of the cut
    cases.append(("z1_cut.fastq.zst", z1[: int(len(z1) * 0.7)]))
    for name, blob in cases:
        p = tmp_path / name
        p.write_bytes(blob)
        # (a flipped byte inside a stored / literal run can leave the codes valid: then the checksum is what reports it, behind
        # its member's or frame's rows — at most the whole input for the single member)
        rows, msg = _expect_error(p, "fastq", min_rows_before=int(n_rows * 0.3))
        assert rows < n_rows or "one_" in name or "check" in msg.lower() or "crc" in msg.lower(), (name, rows, msg)
    good = tmp_path / "good.fastq.gz"
    good.write_bytes(_bgzf(data))
    r = _open(good, "fastq")
    assert r.digest() == want
    r.close()


@pytest.mark.timeout(300)
def test_corruption_inside_a_fan_out(gpu, fastq_mid, tmp_path, monkeypatch):
    """stripes on worker threads: the error of the stripe that holds the corrupt member comes behind the rows of the stripes
    in front of it, the workers of the stripes behind it are stopped by the close"""
    data, want = fastq_mid
    monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(CAP_MB))
    monkeypatch.setenv("EXON_GPU_SHARDS", "6")
    monkeypatch.setenv("EXG_FANOUT_WORKERS", "3")
    bg = bytearray(_bgzf(data))
    bg[int(len(bg) * 0.55)] ^= 0xA5
    p = tmp_path / "fan_flip.fastq.gz"
    p.write_bytes(bytes(bg))
    rows, msg = _expect_error(p, "fastq", min_rows_before=int(want[0] * 0.3), shard_count=0)
    assert rows < want[0], msg


@pytest.mark.timeout(300)
def test_new_reader_under_the_cap(gpu, oracle, fastq_mid, tmp_path, monkeypatch):
    """path (A), the reference's FFI: the Arrow stream over a BGZF input 4x the cap returns the oracle's rows, batch by batch"""
    import hashlib

    from exon_duckdb_amd.arrow import new_reader
    data, want = fastq_mid
    p = tmp_path / "arrow.fastq.gz"
    p.write_bytes(_bgzf(data))
    monkeypatch.setenv("EXG_DEVICE_MEM_CAP_MB", str(CAP_MB))
    hs = [hashlib.blake2b(digest_size=16) for _ in range(4)]
    n = 0
    for b in new_reader(str(p), "fastq", batch_size=8192):
        n += b.num_rows
        for k, col in enumerate(b.columns):
            for v in col.to_pylist():
                hs[k].update(b"\xff\x00NULL" if v is None else v.encode())
                hs[k].update(b"\x00")
    assert (n, hashlib.blake2b(b"".join(h.digest() for h in hs), digest_size=16).hexdigest()) == want


@pytest.mark.timeout(300)
@pytest.mark.parametrize("round_out", [1 << 20, 5 << 20])
def test_round_size_does_not_change_the_rows(gpu, fastq_mid, tmp_path, monkeypatch, round_out):
    """No memory cap: a big gzip member is decoded in rounds by two lanes that take turns (the next window read ahead while
    the current one is decoded), a zstd frame in rounds whose checksum is folded by the stage behind the decoder.  Small
    rounds (EXG_STREAM_ROUND_OUT) put a 66 MB input through a dozen to sixty of them: the rows are the oracle's."""
    from zstd_util import compress
    data, want = fastq_mid
    monkeypatch.delenv("EXG_DEVICE_MEM_CAP_MB", raising=False)
    monkeypatch.setenv("EXG_STREAM_ROUND_OUT", str(round_out))
    one = tmp_path / "one.fastq.gz"
    one.write_bytes(gzip.compress(data, 1, mtime=0))
    z1 = tmp_path / "one.fastq.zst"
    z1.write_bytes(compress(data, 3, True, window_log=20))   # one frame with a checksum, a 1 MiB window carried from round to round
    for p in (one, z1):
        r = _open(p, "fastq")
        got = r.digest()
        st = r.stats()
        r.close()
        assert got == want, p.name
        assert st["decoded_segments"] >= (len(data) // round_out) // 2, st
    # a wrong checksum in the zstd frame's last four bytes: reported, behind the rows
    bad = bytearray(z1.read_bytes())
    bad[-1] ^= 0x40
    zb = tmp_path / "bad.fastq.zst"
    zb.write_bytes(bytes(bad))
    rows, msg = _expect_error(zb, "fastq", min_rows_before=int(want[0] * 0.9))
    assert "checksum" in msg.lower(), msg
    # ... and in the gzip member's trailer (CRC-32, then ISIZE): the rows of the final round too come before the error (advisor,
    # round 3: the last round's segment — a single-member file's only one without a cap — went back to the pool unseen)
    for at in (-5, -1):
        bad = bytearray(one.read_bytes())
        bad[at] ^= 0x40
        gb = tmp_path / "bad.fastq.gz"
        gb.write_bytes(bytes(bad))
        rows, msg = _expect_error(gb, "fastq", min_rows_before=int(want[0] * 0.99))
        assert "checksum" in msg.lower(), msg


@pytest.mark.timeout(300)
def test_a_text_file_truncated_while_its_rows_are_out(gpu, fastq_mid, tmp_path, monkeypatch):
    """A plain text file is mapped and the DataChunk strings point into the mapping (zero-copy payload).  Another process that
    truncates the file while a query runs used to be a SIGBUS — the end of the DuckDB process — as soon as anybody touched a
    string behind the new end; the reference's buffered reader returns an I/O error.  Now: the strings of the chunk that is
    out read as zeros where the bytes are gone, and the reader's next call fails with EXG_E_IO (exg_map_guard.hpp)."""
    import ctypes as C
    import os
    from exon_duckdb_amd import abi
    from exon_duckdb_amd.reader import ShardReader
    from exon_duckdb_amd.table_function import Chunk
    data, _ = fastq_mid
    p = tmp_path / "t.fastq"
    p.write_bytes(data)
    monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", str(8 << 20))
    r = ShardReader(str(p), "fastq")
    # walk to a chunk whose strings lie well inside the file
    ch = None
    for _ in range(30):
        ch = Chunk()
        assert r._l.exg_next_chunk(r._r, C.byref(ch)) == 0 and ch.n_rows > 0
        last = ch
        if _ < 29:
            r._l.exg_release_chunk(r._r, C.byref(ch))
    n = int(last.n_rows)
    seq = C.cast(last.data[2], C.POINTER(C.c_uint8 * 16))
    raw = bytes(seq[n - 1])
    ln = int.from_bytes(raw[:4], "little")
    ptr = int.from_bytes(raw[8:], "little")
    assert ln == 150 and C.string_at(ptr, ln) in data            # the payload is the mapping's bytes
    os.truncate(p, 4096)                                         # ... which are gone now
    got = C.string_at(ptr, ln)                                   # a SIGBUS without the guard
    assert got == b"\0" * ln
    r._l.exg_release_chunk(r._r, C.byref(last))
    rc, msg = 0, ""
    for _ in range(100000):
        ch = Chunk()
        rc = r._l.exg_next_chunk(r._r, C.byref(ch))
        if rc != 0:
            msg = (r._l.exg_reader_error(r._r) or b"").decode()
            break
        if ch.n_rows == 0:
            break
        r._l.exg_release_chunk(r._r, C.byref(ch))
    assert rc == abi.EXG_E_IO and ("truncated" in msg or "short read" in msg), (rc, msg)
    r.close()
    # the process lives, and the next reader of a whole file is unharmed
    p.write_bytes(data)
    r = ShardReader(str(p), "fastq")
    assert r.count() == 200000
    r.close()


def test_three_hundred_text_readers_open_at_once(gpu, oracle, tmp_path):
    """every mapped text file is registered with the SIGBUS guard (exg_map_guard.hpp): until round 5 its table held 256 mappings and the
    257th file was silently read unguarded — now it holds 4096 and a full table refuses the file (EXG_E_NOMEM)"""
    from exon_duckdb_amd.reader import ShardReader
    data = bytes(oracle.synth_fastq(332 * 50))
    p = tmp_path / "small.fastq"
    p.write_bytes(data)
    readers = []
    try:
        for _ in range(300):
            r = ShardReader(str(p), "fastq", device_batch_bytes=64 << 10)
            assert r.count() == 50          # (the file is mapped when it is first read)
            readers.append(r)
    finally:
        for r in readers:
            r.close()
