"""A BGZF writer for the probes: `path` -> `path_gz` in members of 65 280 bytes (what `bgzip` writes), deflated by a pool of processes."""
import multiprocessing as mp
import os
import struct
import zlib


def _members(args):
    path, lo, hi, level = args
    out = []
    with open(path, "rb") as f:
        f.seek(lo)
        raw = f.read(hi - lo)
    for o in range(0, len(raw), 65280):
        chunk = raw[o:o + 65280]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 18 + len(body) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body +
                   struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


def bgzip(path, path_gz, level=1):
    n = os.path.getsize(path)
    step = 65280 * 256
    jobs = [(path, o, min(n, o + step), level) for o in range(0, n, step)]
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as pool, open(path_gz, "wb") as f:
        for part in pool.imap(_members, jobs):
            f.write(part)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # the EOF block
    return os.path.getsize(path_gz)
