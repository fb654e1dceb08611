"""CPU checks of the host logic behind new_reader / exg_open: the `filters` grammar (what the reference's FilterToString
renders, exon/src/exon/arrow_table_function/module.cpp:158-214, evaluated by DataFusion with SQL precedence) and the
VCF header -> typed INFO / FORMAT keys mapping.  No device is touched."""
import ctypes as C
import os

import pytest


@pytest.fixture(scope="module")
def lib():
    from exon_duckdb_amd._lib import load_test_library
    return load_test_library()   # the introspection helpers live in the scaffolding library, not in the product


def explain(lib, fmt, text):
    buf = C.create_string_buffer(2048)
    rc = lib.exon_tf_filter_explain(fmt.encode(), text.encode(), buf, 2048)
    return rc, buf.value.decode()


def test_filter_to_string_forms(lib):
    # FilterToString output for a ConstantFilter / IS NULL / IS NOT NULL (module.cpp:162-199)
    assert explain(lib, "fasta", "id='a'") == (0, "id = 'a'")
    assert explain(lib, "fastq", "description IS NULL") == (0, "description isnull")
    assert explain(lib, "fastq", "description IS NOT NULL") == (0, "description notnull")
    assert explain(lib, "vcf", "pos>=1000") == (0, "pos >= 1000")
    assert explain(lib, "vcf", "qual>30.5") == (0, "qual > 30.5")
    assert explain(lib, "vcf", "pos!=7") == (0, "pos != 7")


def test_and_binds_tighter_than_or_without_parentheses(lib):
    # the reference joins with " AND " / " OR " and never parenthesises (module.cpp:173-189): SQL precedence decides
    assert explain(lib, "fastq", "name='a' OR name='b' AND sequence<'C'") == \
        (0, "name = 'a' | name = 'b' | sequence < 'C' | AND | OR")
    assert explain(lib, "fastq", "name='a' AND name='b' OR sequence<'C'") == \
        (0, "name = 'a' | name = 'b' | AND | sequence < 'C' | OR")
    assert explain(lib, "fastq", "(name='a' OR name='b') AND sequence<'C'")[1].endswith("OR | sequence < 'C' | AND")


def test_literals_and_names(lib):
    assert explain(lib, "fastq", "name='it''s'") == (0, "name = 'it's'")          # Value::ToSQLString doubles quotes
    assert explain(lib, "fastq", 'NAME = \'x\'') == (0, "name = 'x'")             # column names are case-insensitive
    assert explain(lib, "fastq", '"name"<>\'x\'') == (0, "name != 'x'")
    assert explain(lib, "vcf", "qual<=1e-3")[1] == "qual <= 0.001"
    assert explain(lib, "vcf", "pos>-5") == (0, "pos > -5")


@pytest.mark.parametrize("fmt,text", [
    ("fastq", "nope='x'"), ("fastq", "name="), ("fastq", "name='x"), ("fastq", "name='x' AND"), ("fastq", "name 'x'"),
    ("vcf", "alt='A'"), ("vcf", "info IS NULL"), ("vcf", "pos='a'"), ("fastq", "name=5"), ("fastq", "(name='a'"),
])
def test_filter_errors(lib, fmt, text):
    rc, msg = explain(lib, fmt, text)
    assert rc == -1 and msg


def test_vcf_header_keys(lib):
    hdr = (b"##fileformat=VCFv4.2\n"
           b"##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
           b"##INFO=<ID=AF,Number=A,Type=Float,Description=\"with, comma and \\\"quotes\\\"\">\n"
           b"##INFO=<ID=DB,Number=0,Type=Flag,Description=\"f\">\n"
           b"##INFO=<ID=ANN,Number=.,Type=String,Description=\"s\">\n"
           b"##INFO=<ID=DP,Number=1,Type=Float,Description=\"second definition ignored\">\n"
           b"##FILTER=<ID=q10,Description=\"x\">\n"
           b"##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\r\n"
           b"##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"r\">\n"
           b"##FORMAT=<ID=C,Number=1,Type=Character,Description=\"c\">\n"
           b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
           b"1\t5\t.\tA\tC\t.\t.\tDP=1\n")
    buf = C.create_string_buffer(2048)
    assert lib.exon_tf_vcf_header_explain(hdr, len(hdr), buf, 2048) == 0
    assert buf.value.decode() == "INFO DP:i AF:[f] DB:b ANN:[u] | FORMAT GT:u AD:[i] C:u"


def test_vcf_header_keys_match_the_oracle(lib, oracle, golden_dir):
    import os
    for name in ("vcf/index.vcf", "vcf/vcf_file.vcf", "vcf/vcf_meta_meta.vcf"):
        data = open(os.path.join(golden_dir, name), "rb").read()
        info, fmt = oracle.vcf_header_keys(data)
        t = {"Integer": "i", "Float": "f", "Flag": "b", "String": "u"}
        want = "INFO" + "".join(f" {k}:{'[' + t[ty] + ']' if ls else t[ty]}" for k, ty, ls in info) + " | FORMAT" + \
               "".join(f" {k}:{'[' + t[ty] + ']' if ls else t[ty]}" for k, ty, ls in fmt)
        buf = C.create_string_buffer(8192)
        assert lib.exon_tf_vcf_header_explain(data, len(data), buf, 8192) == 0
        assert buf.value.decode() == want, name


def test_duckdb_shim_compiles_against_the_api_stubs():
    """duckdb_shim/exon_extension.cpp — the real `LOAD exon` binding: the glue of csrc/exon_table_function.hpp instantiated over
    DuckDB's classes — goes through a compiler: `-fsyntax-only` against declaration-only stand-ins of the ten DuckDB v0.8.1
    headers it includes (tests/duckdb_stub/).  DuckDB itself does not exist on the build box, so this says nothing about
    DuckDB's behaviour; it catches every typo, wrong member and template-instantiation error of the `RealDuck` traits."""
    import shutil
    import subprocess
    cxx = shutil.which("c++") or shutil.which("g++")
    if not cxx:
        pytest.skip("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [cxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(root, "tests", "duckdb_stub"), "-I", os.path.join(root, "include"),
           "-I", os.path.join(root, "exon_duckdb_amd", "csrc"), os.path.join(root, "duckdb_shim", "exon_extension.cpp")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_batch_index_stays_below_duckdbs_pipeline_increment():
    """get_batch_index = (shard << 24) + device batch: 64 shards (the planner's limit) of 2^24 batches stay eleven thousand
    times below the 10^13 DuckDB 0.8.1 puts between two pipelines' batch ranges (PipelineBuildState::BATCH_INCREMENT)."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exon_duckdb_amd", "csrc", "exon_table_function.hpp")).read()
    assert "kBatchBits = 24" in src
    assert (64 << 24) + (1 << 24) < 10 ** 13 // 1000


def test_bench_refuses_a_world_that_is_not_gpus():
    """`bench.py --gpus 2` under a 1-rank launcher (WORLD_SIZE=1) must fail loudly, before anything touches a GPU: a line
    that says n_gpus = 1 for a run that was asked for 2 would be a wasted scaling run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 2 and "WORLD_SIZE=1" in res.stderr and not res.stdout.strip(), (res.returncode, res.stderr[-500:])


def test_bench_without_a_launcher_starts_its_ranks():
    """`bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself (a child torch.distributed.run; nothing is exec'ed).  There
    is no GPU here, so each rank stops at "bench.py needs a GPU" — said twice, by ranks that see WORLD_SIZE=2 — and the parent
    exits with the launcher's non-zero code."""
    import os
    import subprocess
    import sys
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this test is tests/test_reader_shards_gpu.py::test_bench_launches_its_own_ranks")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--backend", "gloo"], env=env,
                         cwd=root, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert res.stderr.count("bench.py needs a GPU") >= 2, res.stderr[-2000:]


def test_falsifiability_kit_writes_every_rule_case(tmp_path):
    """tools/falsify_kit.py: the inputs of the [RECALLED] rule tests as files + the oracle's rows + one script for a real exon
    build; its compare.py says SAME for every case when the oracle's own rows come back"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kit = tmp_path / "kit"
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "falsify_kit.py"), str(kit)])
    cases = json.load(open(kit / "cases.json"))
    assert len(cases) >= 90 and {c["format"] for c in cases} == {"fastq", "fasta", "vcf"}
    assert all(os.path.exists(kit / c["file"]) for c in cases)
    # round 5: decoder-level cases (the compressed file is the case) and schema cases (DESCRIBE) ride along
    assert sum(1 for c in cases if c["file"].endswith((".gz", ".zst"))) >= 18 and sum(1 for c in cases if c.get("compression")) >= 2
    assert sum(1 for c in cases if c.get("schema")) == 3 and "DESCRIBE SELECT * FROM read_vcf_file_records" in open(kit / "run.sh").read()
    os.makedirs(kit / "got")
    for c in cases:   # what a run.sh against a build that agrees with the oracle leaves behind
        e = json.load(open(kit / "expected" / (c["case"] + ".json")))
        if c.get("schema"):
            (kit / "got" / (c["case"] + ".json")).write_text(json.dumps([dict(r, null="YES", key=None, default=None, extra=None) for r in e["schema"]]))
            continue
        if e["error"]:
            (kit / "got" / (c["case"] + ".err")).write_text("Error: ...")
        else:
            (kit / "got" / (c["case"] + ".json")).write_text("".join(json.dumps(r) + "\n" for r in e["rows"]))
    res = subprocess.run([sys.executable, "compare.py"], cwd=kit, capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.count("SAME") == len(cases), res.stdout[-2000:]
    # and a build that disagrees on one rule is told so
    c = next(c for c in cases if c["case"].startswith("fastq_split_at_first_space_only"))
    (kit / "got" / (c["case"] + ".json")).write_text(json.dumps({"name": "id a", "description": "b  c", "sequence": "AC", "quality_scores": "!!"}) + "\n")
    res = subprocess.run([sys.executable, "compare.py"], cwd=kit, capture_output=True, text=True)
    assert res.returncode == 1 and "DIFFERENT test_fastq_split_at_first_space_only" in res.stdout
    # ... and one whose `pos` is INTEGER, or that stops behind the first gzip member
    c = next(c for c in cases if c.get("schema") and c["format"] == "vcf")
    e = json.load(open(kit / "expected" / (c["case"] + ".json")))["schema"]
    (kit / "got" / (c["case"] + ".json")).write_text(json.dumps([dict(r, column_type="INTEGER") if r["column_name"] == "pos" else r for r in e]))
    c2 = next(c for c in cases if c["case"].startswith("gzip_every_member_of_a_concatenation_is_read_0"))
    e2 = json.load(open(kit / "expected" / (c2["case"] + ".json")))["rows"]
    (kit / "got" / (c2["case"] + ".json")).write_text("".join(json.dumps(r) + "\n" for r in e2[:2]))
    res = subprocess.run([sys.executable, "compare.py"], cwd=kit, capture_output=True, text=True)
    assert "pos: expected BIGINT, got INTEGER" in res.stdout and "DIFFERENT test_gzip_every_member_of_a_concatenation_is_read" in res.stdout


def test_scan_algo_hint_from_a_host_sample():
    """exg_scan_algo_hint (ABI 9): the scan a reader launches first, from the first MiB of the input on the host — no device touched.
    The five shapes of bench.py's record_shapes + the ordinary ones (what the lean scan cannot do in one pass: exg_fused_core.hpp)"""
    import ctypes as C
    from exon_duckdb_amd import abi, load_library
    from exon_duckdb_amd.testing import shapes
    lib = load_library()
    lib.exg_scan_algo_hint.restype = C.c_int
    lib.exg_scan_algo_hint.argtypes = [C.c_int, C.c_char_p, C.c_uint64]

    def hint(fmt, data):
        data = bytes(data[:1 << 20])
        return lib.exg_scan_algo_hint(fmt, data, len(data))

    fq150 = shapes.fastq_records([150] * 3000, seed=1)
    assert hint(abi.EXG_FMT_FASTQ, fq150) == abi.EXG_ALGO_FUSED
    assert hint(abi.EXG_FMT_FASTQ, shapes.fastq_records([15000] * 40, seed=2)) == abi.EXG_ALGO_FUSED_FULL     # HiFi
    assert hint(abi.EXG_FMT_FASTQ, shapes.fastq_records([36] * 20000, seed=3, desc_every=0)) == abi.EXG_ALGO_FUSED_FULL   # dense lines
    assert hint(abi.EXG_FMT_FASTQ, shapes.fastq_records([250] * 3000, seed=4)) == abi.EXG_ALGO_FUSED
    assert hint(abi.EXG_FMT_FASTQ, b"@r caf\xc3\xa9\n" + fq150) == abi.EXG_ALGO_FUSED_FULL                      # bytes >= 0x80
    body = lambda v: v[v.index(b"#CHROM"):].split(b"\n", 1)[1]   # noqa: E731
    assert hint(abi.EXG_FMT_VCF, body(shapes.vcf_lines(5000, 0, seed=5))) == abi.EXG_ALGO_FUSED
    assert hint(abi.EXG_FMT_VCF, body(shapes.vcf_lines(2000, 100, seed=6))) == abi.EXG_ALGO_FUSED_FULL           # 483-byte lines
    assert hint(abi.EXG_FMT_VCF, body(shapes.vcf_lines(200, 2504, seed=7))) == abi.EXG_ALGO_FUSED_INDEX         # 10 kB lines
    assert hint(abi.EXG_FMT_VCF, b"1\t1\t.\tA\tC\t.\t.\t" + b"x" * (2 << 20)) == abi.EXG_ALGO_FUSED_INDEX     # one line of megabytes
    assert hint(abi.EXG_FMT_FASTQ, b"@r\nAC\n+\nII\n") == abi.EXG_ALGO_FUSED                                   # too little to tell
    assert hint(abi.EXG_FMT_FASTA, b">a\nACGT\n" * 10000) == abi.EXG_ALGO_FUSED


def _mix64(x):
    m = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


def _fold_bytes(h, b):
    m = (1 << 64) - 1
    n = len(b)
    h ^= (n * 0x9E3779B97F4A7C15) & m
    o = 0
    while n - o >= 8:
        h = ((h ^ int.from_bytes(b[o:o + 8], "little")) * 0x100000001B3) & m
        h ^= h >> 29
        o += 8
    w = int.from_bytes(b[o:], "little") if n > o else 0
    h = ((h ^ w ^ (((n - o) << 56) & m)) * 0x100000001B3) & m
    return h ^ (h >> 32)


def test_the_bench_legs_fasta_expectation_is_the_oracles_rows(lib, tmp_path):
    # bench.py's end_to_end_fasta leg checks the reader's rows against exon_tf_expect_fasta_file — a split of the file of its own.
    # Here that split is held against the oracle: the digest it gives must be the fold of the oracle's rows (id, description or
    # NULL, the joined sequence), on the reference's fixtures and on a generated file
    import os
    from oracle import pyoracle
    lib.exon_tf_expect_fasta_file.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    files = [os.path.join(g, "test.fasta"), os.path.join(g, "test.mixed-desc.fasta")]
    p = tmp_path / "synth.fasta"
    p.write_bytes(bytes(pyoracle.synth_fasta(300, seed=9)).replace(b"\n", b"\r\n", 40))
    files.append(str(p))
    for f in files:
        with open(f, "rb") as fh:
            data = fh.read()
        exp = pyoracle.fasta_parse(data)
        assert exp.error_code == 0
        acc = 0
        cols = [exp.columns[k].to_list() for k in ("id", "description", "sequence")]
        for k, (i, d, s) in enumerate(zip(*cols)):
            h = _fold_bytes(_mix64(k), i)
            h = _fold_bytes(h, d) if d is not None else _fold_bytes(h ^ 0xDEAD, b"")
            h = _fold_bytes(h, s)
            acc = (acc + _mix64(h)) & ((1 << 64) - 1)
        rows, dg = C.c_uint64(0), C.c_uint64(0)
        assert lib.exon_tf_expect_fasta_file(f.encode(), C.byref(rows), C.byref(dg)) == 0
        assert rows.value == exp.n_rows and dg.value == acc, f
