"""How fast do the two gzip shapes inflate + scan end to end: one member (gzip/pigz output) vs BGZF members."""
import gzip, os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exon_duckdb_amd import device, table_function
n_rec = int(os.environ.get("GZ_RECORDS", "400000"))
raw = device.synth_fastq(332 * n_rec)[: 332 * n_rec].cpu().numpy().tobytes()
single = "/tmp/exg_single.fastq.gz"
open(single, "wb").write(gzip.compress(raw, 6, mtime=0))
parts = []
for i in range(0, len(raw), 65280):
    chunk = raw[i:i + 65280]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    d = co.compress(chunk) + co.flush()
    parts.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1)
                 + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
bg = "/tmp/exg_bgzf.fastq.gz"
open(bg, "wb").write(b"".join(parts))
con = table_function.connect()
for label, path in ((("single member", single),) if os.environ.get("GZ_ONLY_SINGLE") else (("single member", single), ("bgzf", bg))):
    rel = con.table_function("read_fastq", path)
    rel.count()
    t0 = time.time(); n = rel.count(); dt = time.time() - t0
    assert n == n_rec
    print(f"{label}: {len(raw)/1e6:.0f} MB inflated, count in {dt:.3f} s = {len(raw)/dt/1e6:.0f} MB/s of FASTQ", flush=True)
