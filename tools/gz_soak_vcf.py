"""bgzip-style VCF through the reader at size: header once, the body of 400 000 synthetic lines repeated, all in
65 280-byte BGZF members; COUNT(*), chunks, Arrow stream (typed nested columns)."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import table_function
from exon_duckdb_amd.arrow import new_reader
from oracle import pyoracle            # writes the input only


def bgzf(data):
    out = []
    for i in range(0, len(data), 65280):
        chunk = data[i:i + 65280]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1)
                   + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


L = 400_000
text = bytes(pyoracle.synth_vcf(L))
hdr = int(pyoracle.vcf_parse(text[:4096], want_string_t=False).extra["header_bytes"])
head, body = bgzf(text[:hdr]), bgzf(text[hdr:])
K = max(1, int(float(os.environ.get("GZ_SOAK_GB", "2")) * 1e9 / (len(text) - hdr)))
path = "/tmp/exg_soak.vcf.gz"
with open(path, "wb") as f:
    f.write(head)
    for _ in range(K):
        f.write(body)
infl, comp = hdr + K * (len(text) - hdr), len(head) + K * len(body)
print(f"{comp / 1e9:.2f} GB compressed, {infl / 1e9:.2f} GB of VCF, {K * L} lines", flush=True)
con = table_function.connect()
rel = con.table_function("read_vcf", path)
for label, fn in (("COUNT(*)", rel.count), ("chunks", lambda: sum(rel.chunk_sizes())),
                  ("arrow (nested columns)", lambda: sum(b.num_rows for b in new_reader(path, "vcf")))):
    fn()
    t0 = time.time(); n = fn(); dt = time.time() - t0
    assert n == K * L, (label, n)
    print(f"{label}: {dt:.3f} s = {infl / dt / 1e9:.1f} GB/s of VCF, {n / dt / 1e6:.0f} M rows/s", flush=True)
os.unlink(path)
