"""AddressSanitizer + UBSan build of the host-only parsers of libexon_gpu (gzip member index, the BGZF member walk, zstd
frame / block walk, the VCF header parser, the shard planner, the `filters` grammar), driven over valid streams, truncations and random mutations (tests/host_asan_driver.cpp).  GPU
sanitizers do not exist on the pool; everything these parsers read is user input."""
import gzip
import os
import struct
import subprocess
import zlib

import pytest

from zstd_util import compress, fastq_text, skippable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "exon_duckdb_amd", "csrc")


def bgzf(data, block):
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1)
                   + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


def test_host_parsers_under_asan_ubsan(tmp_path):
    if not os.path.isdir("/opt/rocm/include"):
        pytest.skip("HIP headers not found")
    exe = tmp_path / "host_asan"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D__HIP_PLATFORM_AMD__",
           "-I", "/opt/rocm/include", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", str(exe),
           os.path.join(ROOT, "tests", "host_asan_driver.cpp"), os.path.join(CSRC, "exg_gzip.cpp"), os.path.join(CSRC, "exg_zstd_index.cpp"),
           os.path.join(CSRC, "exg_rd_bgzf.cpp"), os.path.join(CSRC, "exg_vcf_header.cpp"), os.path.join(CSRC, "exg_rd_plan.cpp"), os.path.join(CSRC, "exg_rd_fanout.cpp"), os.path.join(CSRC, "exg_map_guard.cpp"), "-lpthread"]
    subprocess.check_call(cmd)
    text = fastq_text(3000, 4)
    files = {
        "a.gz": gzip.compress(text, 6, mtime=0),
        "b.gz": bgzf(text, 7000),
        "c.gz": gzip.compress(text[:5000], 6, mtime=0) + bgzf(text[5000:60000], 9000) + gzip.compress(b"", 6, mtime=0),
        "a.zst": compress(text, 3, True),
        "b.zst": compress(text[:40000], 19, True, window_log=12, content_size=False) + skippable(b"meta") + compress(text[40000:], 1, False),
        "c.zst": b"".join(compress(text[i:i + 5001], 1 + i % 5, i % 2 == 0) for i in range(0, 60000, 5001)),
    }
    paths = []
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
        paths.append(str(tmp_path / name))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:handle_sigbus=0", UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run([str(exe)] + paths, env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "runs" in res.stdout
