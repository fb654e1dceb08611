"""Randomised inflate parity against zlib: payload generators x levels x strategies x member sizes through both decoders
(exg_inflate_members: byte ring + HBM window, tokens placed by prefix sum; exg_inflate_stream: chunked symbol decode)."""
import zlib

import numpy as np
import pytest

from tests.test_inflate_gpu import bgzf, roundtrip            # noqa: E402  (helpers: BGZF framing, member round trip)
from tests.test_inflate_stream_gpu import stream_inflate      # noqa: E402

pytestmark = pytest.mark.gpu

STRATEGIES = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]


def payload(rng, kind, n):
    if kind == 0:      # DNA-like
        return rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()
    if kind == 1:      # runs (dist 1-4 matches, overlapping copies, length 258)
        out = bytearray()
        while len(out) < n:
            out += bytes([int(rng.integers(32, 127))]) * int(rng.integers(1, 700))
        return bytes(out[:n])
    if kind == 2:      # short period repeats with mutations (distances of every size, matches ending at any phase)
        unit = rng.integers(32, 127, int(rng.integers(2, 300)), dtype=np.uint8)
        buf = np.tile(unit, n // len(unit) + 1)[:n].copy()
        idx = rng.integers(0, n, n // 97 + 1)
        buf[idx] = rng.integers(32, 127, len(idx), dtype=np.uint8)
        return buf.tobytes()
    if kind == 3:      # far matches: a 20-40 KB block repeated (distances near the 32 KiB window)
        blk = rng.integers(32, 127, int(rng.integers(20000, 32768)), dtype=np.uint8)
        return np.tile(blk, n // len(blk) + 1)[:n].tobytes()
    if kind == 4:      # text with a skewed alphabet (long and short codes, > 10-bit codes at HUFFMAN_ONLY)
        p = 1.0 / np.arange(1, 97) ** 1.7
        return (rng.choice(96, n, p=p / p.sum()) + 32).astype(np.uint8).tobytes()
    return rng.integers(0, 256, n, dtype=np.uint8).tobytes()   # incompressible: stored blocks


@pytest.mark.parametrize("seed", range(24))
def test_members_random(gpu, seed):
    rng = np.random.default_rng(1000 + seed)
    data = payload(rng, seed % 6, int(rng.integers(1, 400_000)))
    level = int(rng.integers(1, 10))
    block = int(rng.choice([997, 8191, 32768, 65280]))
    gz = bgzf(data, block=block, level=level)
    roundtrip(gpu, data, gz)
    # one raw member per strategy (fixed / huffman-only / rle produce block shapes the BGZF default never does)
    for strat in STRATEGIES:
        piece = data[:65000]
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strat)          # gzip wrapper, one member
        roundtrip(gpu, piece, co.compress(piece) + co.flush())


@pytest.mark.parametrize("seed", range(12))
def test_stream_random(gpu, seed):
    rng = np.random.default_rng(2000 + seed)
    kinds = [0, 1, 2, 3, 4]
    parts = [payload(rng, int(rng.choice(kinds)), int(rng.integers(50_000, 900_000))) for _ in range(int(rng.integers(2, 6)))]
    data = b"".join(parts)
    level = int(rng.integers(1, 10))
    strat = STRATEGIES[int(rng.integers(0, 3))]
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
    comp = co.compress(data) + co.flush()
    for chunk in (32768, int(rng.integers(40_000, 300_000))):
        rc, got, consumed = stream_inflate(gpu, comp, chunk, pad_front=int(rng.integers(0, 16)))
        assert rc == 0, gpu.exg_last_error_message()
        assert got == data and consumed == len(comp), (seed, level, strat, chunk, len(got), len(data))
