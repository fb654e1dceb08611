"""Secondary measurements (GPU box): config 3 (VCF 8-column scan) and config 4 (BGZF inflate feeding the
FASTQ scan) of BASELINE.json, plus the PCIe-inclusive reader rate.  Prints one JSON object."""
import ctypes as C
import json
import os
import struct
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from exon_duckdb_amd import abi, device, load_library
from oracle import pyoracle

lib = load_library()
torch.cuda.set_device(0)
out = {}


def time_ms(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2]


# ---- config 3: VCF -----------------------------------------------------------------------------------
body = pyoracle.synth_vcf(400_000)
hdr_end = pyoracle.vcf_parse(body[:4096], want_string_t=False).extra["header_bytes"]   # end of the '#' lines
reps = 5_000_000_000 // (len(body) - hdr_end)
vcf = np.concatenate([body[:hdr_end]] + [body[hdr_end:]] * reps)
n = len(vcf)
n_lines = 400_000 * reps
d_in = device.upload(vcf.tobytes())
del vcf
scan = device.VcfScan(n, capacity_records=n_lines + 16)
for label, proj in (("all_columns", None), ("chrom_pos_only", {0})):
    ms = time_ms(lambda: scan.launch(d_in, lead=hdr_end, algo=abi.EXG_ALGO_AUTO, project=proj), reps=5, warm=1)
    r = scan.fetch()
    assert r.error_code == 0 and r.n_records == n_lines and not (r.flags & abi.EXG_RF_FALLBACK), (r.error_code, r.n_records, r.flags)
    out[f"vcf_{label}"] = {"bytes": n, "lines": n_lines, "ms": ms, "read_GBps": n / ms / 1e6, "lines_per_s": n_lines / ms * 1e3}
del scan, d_in
torch.cuda.empty_cache()

# ---- config 4: BGZF FASTQ -> inflate -> scan ------------------------------------------------------------
sample = device.synth_fastq(332 * 100_000)[: 332 * 100_000].cpu().numpy().tobytes()   # 33 MB
t0 = time.time()
members, comp = [], []
pos = 0
for i in range(0, len(sample), 65280):
    chunk = sample[i:i + 65280]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(chunk) + co.flush()
    blk = (b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1)
           + raw + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    members.append((pos + 18, len(raw) + 8, i, len(chunk)))
    comp.append(blk)
    pos += len(blk)
comp = b"".join(comp)
K = 32                                                     # replicate the stream: members are independent
U, Cn = len(sample), len(comp)
all_members = (abi.InflateMember * (len(members) * K))()
for k in range(K):
    for j, (co_, cs, oo, oc) in enumerate(members):
        m = all_members[k * len(members) + j]
        m.comp_off, m.comp_size, m.out_off, m.out_cap = k * Cn + co_, cs, k * U + oo, oc
d_comp = device.upload(comp * K)
d_out = torch.empty(U * K + 64, dtype=torch.uint8, device="cuda")
d_members = torch.frombuffer(bytearray(bytes(all_members)), dtype=torch.uint8).cuda()
d_status = torch.zeros(len(all_members) * 24, dtype=torch.uint8, device="cuda")
ms = time_ms(lambda: device.check(lib.exg_inflate_members(C.c_void_p(d_comp.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                                          C.c_void_p(d_members.data_ptr()), C.c_void_p(d_status.data_ptr()),
                                                          len(all_members), device.stream_ptr())), reps=5, warm=1)
st = np.frombuffer(d_status.cpu().numpy().tobytes(), dtype=np.dtype([("code", "<u4"), ("pad", "<u4"), ("produced", "<u8"), ("consumed", "<u8")]))
assert (st["code"] == 0).all() and bytes(d_out[:U].cpu().numpy().tobytes()) == sample
scan = device.FastqScan(U * K, capacity_records=U * K // 332 + 16)
ms_scan = time_ms(lambda: scan.launch(d_out, n_bytes=U * K), reps=5, warm=1)
r = scan.fetch()
assert r.error_code == 0 and r.n_records == U * K // 332
out["bgzf_fastq"] = {"compressed_bytes": Cn * K, "inflated_bytes": U * K, "ratio": U / Cn, "members": len(all_members),
                     "inflate_ms": ms, "inflate_out_GBps": U * K / ms / 1e6, "scan_ms": ms_scan,
                     "inflate_plus_scan_out_GBps": U * K / (ms + ms_scan) / 1e6,
                     "records_per_s": (U * K // 332) / (ms + ms_scan) * 1e3}
del d_comp, d_out, scan
torch.cuda.empty_cache()

# ---- reader level, PCIe inclusive: file in host RAM -> DataChunks on the host ----------------------------------
path = "/tmp/exg_bench.fastq"
nb = 332 * 6_000_000
with open(path, "wb") as f:
    f.write(device.synth_fastq(nb)[:nb].cpu().numpy().tobytes())
from exon_duckdb_amd import table_function
con = table_function.connect()
rel = con.table_function("read_fastq", path)
for label, fn in (("count", rel.count), ("all_columns_chunks", lambda: sum(rel.chunk_sizes()))):
    fn()
    t0 = time.time()
    nrec = fn()
    dt = time.time() - t0
    out[f"reader_fastq_{label}"] = {"bytes": nb, "records": nrec, "s": dt, "GBps": nb / dt / 1e9, "records_per_s": nrec / dt}
os.remove(path)
print(json.dumps(out, indent=1))
