"""A library variant for an in-box A/B (tools/ab_lib.sh): one source recompiled with extra flags, linked with the current objects
of every other source into exon_duckdb_amd/lib/<name>.so.    python tools/build_variant.py <name> <source.hip> -DFOO=1 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from exon_duckdb_amd import build as B  # noqa: E402

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(verbose=False)
src_path = os.path.join(B.CSRC, src)
obj = os.path.join(B.OBJ_DIR, f"{src}.{name}.o")
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
subprocess.check_call([hipcc] + B._flags() + B.FILE_FLAGS.get(src, []) + flags + ["-c", src_path, "-o", obj])
objs = [obj if os.path.basename(s) == src else B._obj(s) for s in B.sources()]
out = os.path.join(B.LIB_DIR, name + ".so")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-Wl,--version-script=" + B.EXPORTS, "-lpthread", "-ldl"])
print(out)
