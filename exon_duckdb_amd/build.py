"""Build libexon_gpu.so for gfx950 with hipcc (in-tree, so the .so travels with gpurun).

One object per source under lib/obj/ (compiled in parallel, rebuilt when the source or any header is newer),
linked with an export list (csrc/exports.map): only the C-ABI of include/exon_gpu.h leaves the library.
Test and bench scaffolding (csrc/testing/: the DuckDB-API mirror duck_mini + the exon_tf_* harness, the synthetic-input
generators exg_synth_*, the chunk-draining / digesting consumers, the host-pipeline probe) is linked into a library of
its own, libexon_tf_test.so, on top of libexon_gpu.so — it is not part of the product.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
TESTING = os.path.join(CSRC, "testing")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libexon_gpu.so")
TEST_LIB = os.path.join(LIB_DIR, "libexon_tf_test.so")
INCLUDE = os.path.join(HERE, "..", "include")
EXPORTS = os.path.join(CSRC, "exports.map")


def _srcs(d):
    if not os.path.isdir(d):
        return []
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".hip") or f.endswith(".cpp"))


def sources():
    return _srcs(CSRC)


def test_sources():
    return _srcs(TESTING)


# host-only units of the product that the test library links a copy of (their C++ symbols do not leave libexon_gpu.so)
TEST_SHARED = ["exg_vcf_header.cpp"]


def _headers():
    hs = [os.path.join(INCLUDE, "exon_gpu.h")]
    for d in (CSRC, TESTING):
        if os.path.isdir(d):
            hs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".hpp") or f.endswith(".h")]
    return hs


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.basename(src) + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def needs_build():
    return _stale(LIB, sources() + _headers() + [EXPORTS]) or (
        bool(test_sources()) and _stale(TEST_LIB, test_sources() + _headers() + [LIB]))


def _flags():
    extra = os.environ.get("EXG_CXXFLAGS", "").split()
    return ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function",
            "-I", INCLUDE, "-I", CSRC] + extra


# Per-file flags.  exg_inflate.hip: the machine scheduler's max-ILP strategy — k_inflate's step is long dependent chains on two
# issue ports that are both ~85-90 % busy; A/B in one box: FASTQ 76.1 -> 76.9 GB/s, VCF 99.7 -> 101.4 (the
# max-memory-clause strategy: +0.5 %).
FILE_FLAGS = {"exg_inflate.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    todo = []
    for s in sources() + test_sources():
        o = _obj(s)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            todo.append(s)

    def cc(s):
        cmd = [hipcc] + _flags() + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", _obj(s)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    jobs = int(os.environ.get("EXG_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(cc, todo))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in sources()] + [
        "-Wl,--version-script=" + EXPORTS, "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    if test_sources():
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TEST_LIB] + [_obj(s) for s in test_sources()] + [
            _obj(os.path.join(CSRC, f)) for f in TEST_SHARED] + [
            "-L", LIB_DIR, "-lexon_gpu", "-Wl,-rpath,$ORIGIN", "-lpthread"]
        if verbose:
            print(" ".join(link), file=sys.stderr)
        subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
