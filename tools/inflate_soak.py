"""One-off soak of both DEFLATE decoders against zlib: the randomised cases of tests/test_inflate_random_gpu.py over
SOAK_SEEDS (default 300) more seeds."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (its HIP runtime first: the library binds to the one that is there)
from exon_duckdb_amd import load_library
from tests.test_inflate_gpu import bgzf, roundtrip
from tests.test_inflate_stream_gpu import stream_inflate
from tests.test_inflate_random_gpu import payload, STRATEGIES

lib = load_library()
n = int(os.environ.get("SOAK_SEEDS", "300"))
t0 = time.time()
for seed in range(n):
    rng = np.random.default_rng(50_000 + seed)
    data = payload(rng, seed % 6, int(rng.integers(1, 400_000)))
    level = int(rng.integers(1, 10))
    roundtrip(lib, data, bgzf(data, block=int(rng.choice([997, 8191, 32768, 65280])), level=level))
    strat = STRATEGIES[seed % len(STRATEGIES)]
    piece = data[:65000]
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strat)
    roundtrip(lib, piece, co.compress(piece) + co.flush())
    if seed % 3 == 0:
        parts = [payload(rng, int(rng.integers(0, 5)), int(rng.integers(50_000, 600_000))) for _ in range(int(rng.integers(2, 5)))]
        big = b"".join(parts)
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, STRATEGIES[int(rng.integers(0, 3))])
        comp = co.compress(big) + co.flush()
        rc, got, consumed = stream_inflate(lib, comp, int(rng.integers(32768, 300_000)), pad_front=int(rng.integers(0, 16)))
        assert rc == 0 and got == big and consumed == len(comp), (seed, rc)
print(f"{n} seeds: both decoders bit-exact against zlib ({time.time() - t0:.0f} s)")
