"""What a reader holds under EXG_DEVICE_MEM_CAP_MB: run with EXG_TRACE=1 to see the per-batch device bytes."""
import os
import struct
import sys
import tempfile
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle  # noqa: E402  (the probe only needs its generators)


def bgzf(data, block=65280, level=1):
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        d = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" +
                   struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1) + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


if __name__ == "__main__":
    fmt = sys.argv[1] if len(sys.argv) > 1 else "vcf"
    cap = sys.argv[2] if len(sys.argv) > 2 else "16"
    data = bytes({"vcf": lambda: pyoracle.synth_vcf(300000), "fastq": lambda: pyoracle.synth_fastq(332 * 60000),
                  "fasta": lambda: pyoracle.synth_fasta(12000, seed=5)}[fmt]())
    os.environ["EXG_DEVICE_MEM_CAP_MB"] = cap
    from exon_duckdb_amd.reader import ShardReader  # noqa: E402

    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x." + fmt + ".gz")
        open(p, "wb").write(bgzf(data))
        r = ShardReader(p, fmt)
        print(r.digest(), r.stats())
        r.close()

