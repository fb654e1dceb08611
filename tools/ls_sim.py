"""development tool: writes raw DEFLATE members of synthetic FASTQ / VCF / FASTA (BGZF-sized, zlib level 6) for tools/ls_sim.cpp"""
import struct
import sys
import zlib

sys.path.insert(0, ".")
from oracle import pyoracle  # noqa: E402  (tools only: the generators' host form)

kind, n_members, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
level = int(sys.argv[4]) if len(sys.argv) > 4 else 6
n = 65280 * n_members
if kind == "fastq":
    data = pyoracle.synth_fastq(n).tobytes()
elif kind == "vcf":
    data = pyoracle.synth_vcf(n // 40).tobytes()[:n]
elif kind == "fasta":
    data = pyoracle.synth_fasta(n // 1500).tobytes()[:n]
else:
    data = open(kind, "rb").read()[:n]
with open(out, "wb") as f:
    ms = []
    for o in range(0, len(data), 65280):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        ms.append((c.compress(data[o:o + 65280]) + c.flush(), len(data[o:o + 65280])))
    f.write(struct.pack("<I", len(ms)))
    for d, u in ms:
        f.write(struct.pack("<II", len(d), u) + d)
print(kind, len(data), "->", sum(len(d) for d, _ in ms))
