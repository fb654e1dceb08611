"""Differential fuzzing of the three scans against the oracle: valid synthetic inputs with seeded byte-level
mutations (separators turned into data and back, CRs, stray prefixes, non-UTF-8 bytes, deletions, truncation).
Everything must agree: row count, error code / record / offset, and every column of the rows in front of the
error — on both device implementations.  Seeds are fixed: a failure is reproducible by its parameter."""
import numpy as np
import pytest

from exon_duckdb_amd import abi

import test_fasta_gpu as T_FA
import test_fastq_gpu as T_FQ
import test_vcf_gpu as T_VCF

pytestmark = pytest.mark.gpu

SPECIAL = [b"\n", b"\r", b"\r\n", b"@", b"+", b">", b"\t", b" ", b".", b";", b",", b"=", b"#", b"\x00", b"\xc3\xa9", b"\xff",
           b"\xe2\x80\xa8", b"\xc2", b"1e400", b"-", b"nan"]


def mutate(data: bytes, rng, n_mut):
    b = bytearray(data)
    for _ in range(n_mut):
        if not b:
            break
        kind = rng.integers(0, 6)
        pos = int(rng.integers(0, len(b)))
        tok = SPECIAL[int(rng.integers(0, len(SPECIAL)))]
        if kind == 0:
            b[pos:pos + 1] = tok                      # replace a byte
        elif kind == 1:
            b[pos:pos] = tok                          # insert
        elif kind == 2:
            del b[pos:pos + int(rng.integers(1, 8))]  # delete a few bytes
        elif kind == 3:
            nl = b.find(b"\n", pos)                   # drop a newline
            if nl >= 0:
                del b[nl]
        elif kind == 4:
            del b[pos:]                               # truncate
        else:
            b[pos:pos + 1] = bytes([int(rng.integers(0, 256))])
    return bytes(b)


@pytest.mark.parametrize("seed", range(48))
def test_fastq_fuzz(gpu, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    base = bytes(oracle.synth_fastq_ragged(int(rng.integers(3, 120))))
    for _ in range(6):
        data = mutate(base, rng, int(rng.integers(1, 6)))
        for algo in T_FQ.ALGOS:
            T_FQ.check_against_oracle(oracle, data, algo)


@pytest.mark.parametrize("seed", range(48))
def test_vcf_fuzz(gpu, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    base = bytes(oracle.synth_vcf(int(rng.integers(2, 80))))
    hdr = T_VCF.header_bytes(base)
    for _ in range(4):
        data = base[:hdr] + mutate(base[hdr:], rng, int(rng.integers(1, 5)))
        if T_VCF.header_bytes(data) != hdr:
            continue                                  # a mutation made a new header line: host-side territory
        for algo in T_VCF.ALGOS:
            T_VCF.check(oracle, data, algo)


@pytest.mark.parametrize("seed", range(48))
def test_fasta_fuzz(gpu, oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    base = bytes(oracle.synth_fasta(int(rng.integers(1, 12))))
    for _ in range(6):
        data = mutate(base, rng, int(rng.integers(1, 6)))
        T_FA.check(oracle, data)


# ---- the same through the reader level: many tiny device batches, prefetch, errors after the good rows -------------

def _stream_rows(nr, path, fmt, **kw):
    import pyarrow as pa
    rows, failed = [], False
    try:
        for b in nr(path, fmt, **kw):
            rows.extend(b.to_pylist())
    except (pa.ArrowException, OSError):
        failed = True
    return rows, failed


@pytest.mark.parametrize("seed", range(16))
def test_reader_fastq_fuzz(gpu, oracle, tmp_path, monkeypatch, seed):
    from exon_duckdb_amd.arrow import new_reader
    rng = np.random.default_rng(4000 + seed)
    base = bytes(oracle.synth_fastq_ragged(int(rng.integers(50, 400))))
    data = mutate(base, rng, int(rng.integers(0, 4)))
    if any(c >= 0x80 for c in data):
        data = bytes(c if c < 0x80 else 0x41 for c in data)      # Arrow Utf8 -> python str: keep it ASCII here
    p = tmp_path / "f.fastq"
    p.write_bytes(data)
    exp = oracle.fastq_parse(data, want_string_t=False)
    cols = [exp.columns[k].to_list() for k in T_FQ.NAMES]
    dec = lambda v: None if v is None else v.decode()  # noqa: E731
    want = [dict(zip(T_FQ.NAMES, map(dec, t))) for t in zip(*cols)]
    for batch in ("4096", "20000", None):
        if batch:
            monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", batch)
        else:
            monkeypatch.delenv("EXG_DEVICE_BATCH_BYTES", raising=False)
        rows, failed = _stream_rows(new_reader, str(p), "fastq", batch_size=64)
        assert rows == want, (seed, batch)
        assert failed == bool(exp.error_code), (seed, batch, exp.error_code)


@pytest.mark.parametrize("seed", range(16))
def test_reader_vcf_typed_fuzz(gpu, oracle, tmp_path, monkeypatch, seed):
    from exon_duckdb_amd.arrow import new_reader
    from test_arrow_stream_gpu import HEADER, same
    rng = np.random.default_rng(5000 + seed)
    lines = []
    for i in range(int(rng.integers(20, 200))):
        info = ";".join(rng.permutation([f"DP={int(rng.integers(-5, 500))}", "AF=" + ",".join(
            rng.choice(["0.5", ".", "1e-3", "7", "-0.25"], int(rng.integers(1, 4)))), "DB", "ANN=x|y,z", "CH=q", "ZZ=1"])[
            : int(rng.integers(0, 6))]) or "."
        samples = "\t".join(rng.choice(["0/1:1,2:0.5,1.5", ".", "1|1:.:.", "0:3", "./.:1,.,3:."], 3))
        lines.append(f"{1 + i % 3}\t{100 + i}\t{rng.choice(['.', 'rs1', 'a;b'])}\tA\t{rng.choice(['C', 'C,G', '.', '<DEL>'])}\t"
                     f"{rng.choice(['.', '10', '3.5e1', '0'])}\t{rng.choice(['PASS', '.', 'q10;s5'])}\t{info}\tGT:AD:PL\t{samples}")
    body = ("\n".join(lines) + "\n").encode()
    body = mutate(body, rng, int(rng.integers(0, 3)))
    if any(c >= 0x80 for c in body):
        body = bytes(c if c < 0x80 else 0x41 for c in body)
    data = HEADER + body
    if T_VCF.header_bytes(data) != len(HEADER):
        pytest.skip("mutation produced a header line")
    p = tmp_path / "f.vcf"
    p.write_bytes(data)
    want, err_row = oracle.vcf_typed_rows(data)
    tok = oracle.vcf_parse(data, want_string_t=False)
    for batch in ("4096", None):
        if batch:
            monkeypatch.setenv("EXG_DEVICE_BATCH_BYTES", batch)
        else:
            monkeypatch.delenv("EXG_DEVICE_BATCH_BYTES", raising=False)
        rows, failed = _stream_rows(new_reader, str(p), "vcf", batch_size=64)
        assert len(rows) == len(want), (seed, batch, len(rows), len(want))
        assert all(same(g, e) for g, e in zip(rows, want)), (seed, batch)
        assert failed == (err_row is not None or bool(tok.error_code)), (seed, batch)
