"""BGZF inflate alone (for profiling): 33 MB of FASTQ-150 (INFLATE_DATA: vcf, fasta, zeros) in 65 280-byte members, replicated
INFLATE_K (32) times = 1 GB of output."""
import ctypes as C, json, os, struct, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from exon_duckdb_amd import abi, device, load_library
lib = load_library()
torch.cuda.set_device(0)
kind = os.environ.get("INFLATE_DATA", "fastq")   # fastq | vcf | fasta | zeros (how the step copes with other token mixes)
if kind == "vcf":
    t, n = device.synth_vcf(600_000)
    sample = t[:n].cpu().numpy().tobytes()
elif kind == "fasta":
    t, n = device.synth_fasta(20_000)
    sample = t[:n].cpu().numpy().tobytes()
elif kind == "zeros":
    sample = bytes(32 << 20)
else:
    sample = device.synth_fastq(332 * 100_000)[: 332 * 100_000].cpu().numpy().tobytes()
members, comp, pos = [], [], 0
for i in range(0, len(sample), 65280):
    chunk = sample[i:i + 65280]
    co = zlib.compressobj(6, zlib.DEFLATED, -15, int(os.environ.get("INFLATE_MEMLEVEL", "8")))  # (memLevel 9: half as many blocks per member)
    raw = co.compress(chunk) + co.flush()
    blk = (b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1)
           + raw + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    members.append((pos + 18, len(raw) + 8, i, len(chunk)))
    comp.append(blk)
    pos += len(blk)
comp = b"".join(comp)
K = int(os.environ.get("INFLATE_K", "32"))
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
def run():
    device.check(lib.exg_inflate_members(C.c_void_p(d_comp.data_ptr()), C.c_void_p(d_out.data_ptr()), C.c_void_p(d_members.data_ptr()),
                                         C.c_void_p(d_status.data_ptr()), len(all_members), device.stream_ptr()))
run(); torch.cuda.synchronize()
ev = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); ev.append((a, b))
torch.cuda.synchronize()
ms = sorted(x.elapsed_time(y) for x, y in ev)[1]
if not os.environ.get("EXG_INFLATE_NOOUT"):  # (development build: a decode that keeps nothing)
    assert bytes(d_out[:U].cpu().numpy().tobytes()) == sample
print(json.dumps({"data": kind, "ratio": round(U / Cn, 2), "members": len(all_members), "out_bytes": U * K, "ms": ms, "out_GBps": U * K / ms / 1e6}))
