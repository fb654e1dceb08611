"""Soak of the nested VCF columns (round 6: exg_vcf_nested.hip) against oracle.pyoracle.vcf_typed_rows: random headers (0 to 600
##INFO keys — both sides of k_rows' 32-key limit —, 0 to 90 ##FORMAT keys), random lines from the generators of
tests/test_vcf_nested_wide_gpu.py, then structural mutations INSIDE the INFO / FORMAT / sample fields (separators doubled, dropped,
swapped, '.', '%41', empty values, digits turned into letters), random DataChunk sizes and device batch sizes.  Everything must agree:
the rows in front of the first value error, and whether there is one.    SOAK_SEEDS=200 SOAK_FIRST=0 python tools/vcf_nested_soak.py
SOAK_ARROW=1: through new_reader's Arrow stream (the nested columns converted to Arrow's layouts on the device) instead of the chunk boundary."""
import ctypes as C
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (first: the HIP runtime the library binds to)
from oracle import pyoracle  # noqa: E402
from exon_duckdb_amd.reader import ShardReader  # noqa: E402
from exon_duckdb_amd.table_function import Chunk, decode_vector  # noqa: E402
from test_arrow_stream_gpu import same  # noqa: E402
from test_vcf_nested_gpu import norm  # noqa: E402
from test_vcf_nested_wide_gpu import make_header, make_line  # noqa: E402

TOKENS = [b";", b";;", b"=", b"==", b":", b"::", b",", b",,", b".", b"%41", b"%3B", b"%zz", b"", b"x", b"-1", b"1e5", b"=;", b";=", b":.", b".:", b" "]


def mutate_fields(line, rng):
    f = line.split(b"\t")
    for _ in range(rng.choice([0, 0, 1, 1, 2, 4])):
        c = rng.choice([7] + list(range(8, len(f)))) if len(f) > 8 else 7
        if c >= len(f):
            continue
        b = bytearray(f[c])
        kind = rng.randrange(5)
        pos = rng.randrange(len(b) + 1)
        tok = rng.choice(TOKENS)
        if kind == 0:
            b[pos:pos] = tok
        elif kind == 1 and b:
            b[pos % len(b):pos % len(b) + 1] = tok
        elif kind == 2 and b:
            del b[pos % len(b):pos % len(b) + rng.randrange(1, 6)]
        elif kind == 3:
            b = bytearray(b"." if rng.random() < 0.5 else b"")
        elif b:
            i = pos % len(b)
            if 48 <= b[i] <= 57:
                b[i] = rng.choice(b"aZ-+e.")
        f[c] = bytes(b)
    return b"\t".join(f)


def read_rows(path, **kw):
    """-> (rows as dicts, failed?)"""
    r = ShardReader(path, "vcf", **kw)
    out, failed = [], False
    try:
        while True:
            ch = Chunk()
            rc = r._l.exg_next_chunk(r._r, C.byref(ch))
            if rc != 0:
                failed = True
                break
            n = int(ch.n_rows)
            if n == 0:
                break
            cols = [decode_vector(ch.vectors[k].contents, r.trees[k]) for k in range(len(r.names))]
            out.extend(dict(zip(r.names, map(norm, t))) for t in zip(*cols))
            r._l.exg_release_chunk(r._r, C.byref(ch))
    finally:
        r.close()
    return out, failed


def read_rows_arrow(path, batch_rows=2048, device_batch_bytes=0):
    """the same through the reference's own boundary (new_reader -> Arrow C stream, pyarrow) -> (rows as dicts, failed?)"""
    from exon_duckdb_amd.arrow import new_reader
    if device_batch_bytes:
        os.environ["EXG_DEVICE_BATCH_BYTES"] = str(device_batch_bytes)
    else:
        os.environ.pop("EXG_DEVICE_BATCH_BYTES", None)
    out, failed = [], False
    try:
        for b in new_reader(path, "vcf", batch_size=batch_rows):
            out.extend(b.to_pylist())
    except Exception:  # noqa: BLE001  (pyarrow raises what get_next returned)
        failed = True
    finally:
        os.environ.pop("EXG_DEVICE_BATCH_BYTES", None)
    return out, failed


def make_input(seed):
    """-> (file bytes, reader keywords, header shape) of a seed"""
    rng = random.Random(seed)
    n_info = rng.choice([0, 1, 3, 8, 9, 31, 32, 33, 60, 150, 600])
    n_fmt = rng.choice([0, 1, 4, 12, 90])
    n_smp = rng.choice([0, 0, 1, 3, 70, 300]) if n_fmt or rng.random() < 0.3 else 0
    hdr, info, fmt = make_header(n_info, n_fmt, n_smp, seed=seed)
    n_fk = rng.choice([1, 2, 6, 60])
    n_lines = min(rng.choice([1, 7, 64, 65, 300, 1200]), max(5, 150000 // max(1, n_smp * min(n_fk, max(1, n_fmt)))))
    lines, rate = [], rng.choice([0.0, 0.004, 0.02, 0.02, 0.1, 1.0])   # (a value error ends the scan: few mutated lines -> deep files)
    for k in range(n_lines):
        ln = make_line(rng, k, info, fmt, n_smp, rng.choice([0, 1, 2, 5, 20, 120]), n_fk if rng.random() < 0.8 else rng.choice([1, 2, 6, 60]), long_strings=rng.random() < 0.03)
        lines.append(mutate_fields(ln, rng) if rng.random() < rate else ln)
    data = hdr + b"\n".join(lines) + b"\n"
    kw = dict(batch_rows=rng.choice([64, 2048]), device_batch_bytes=rng.choice([0, 0, 16 << 10, 200 << 10]))
    return data, kw, (n_info, n_fmt, n_smp)


def run_seed(seed, path, dry=False):
    """None when the reader and the oracle agree on the seed's file, else what differs"""
    data, kw, shape = make_input(seed)
    want, err_row = pyoracle.vcf_typed_rows(data)
    tok = pyoracle.vcf_parse(data, want_string_t=False)
    with open(path, "wb") as f:
        f.write(data)
    if dry:   # the generator and the oracle alone (no GPU)
        return None
    got, failed = read_rows_arrow(path, **kw) if os.environ.get("SOAK_ARROW") else read_rows(path, **kw)
    if len(got) == len(want) and all(same(g, e) for g, e in zip(got, want)) and failed == (err_row is not None or bool(tok.error_code)):
        return None
    bad = next((i for i, (g, e) in enumerate(zip(got, want)) if not same(g, e)), None)
    msg = (f"seed {seed}: rows {len(got)} / {len(want)}, failed {failed} (oracle: error row {err_row}, tokeniser {tok.error_code}), "
           f"first bad row {bad}, {kw}, keys {shape[0]}/{shape[1]}, samples {shape[2]}")
    if bad is not None:
        g, e = got[bad], want[bad]
        msg += "\n    " + repr({k: (g[k], e[k]) for k in g if not same(g[k], e[k])})[:2000]
    return msg


def main():
    n_seeds, first = int(os.environ.get("SOAK_SEEDS", "100")), int(os.environ.get("SOAK_FIRST", "0"))
    budget = float(os.environ.get("SOAK_SECONDS", "1e9"))
    t0 = time.time()
    d = tempfile.mkdtemp(prefix="exg_vn_soak_")
    p = os.path.join(d, "s.vcf")
    done = 0
    for seed in range(first, first + n_seeds):
        if time.time() - t0 > budget:
            break
        msg = run_seed(seed, p, dry=bool(os.environ.get("SOAK_DRY")))
        if msg:
            keep = os.path.join(ROOT, "gpurun_out", f"vn_soak_seed{seed}.vcf")
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            os.replace(p, keep)
            print("MISMATCH " + msg + f"\n    input kept as {keep}", flush=True)
            sys.exit(1)
        done += 1
    if os.path.exists(p):
        os.unlink(p)
    os.rmdir(d)
    print(f"vcf nested soak: seeds {first}..{first + done - 1} ({done}) agree with the oracle, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
