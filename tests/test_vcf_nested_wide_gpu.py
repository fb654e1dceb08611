"""The nested VCF columns on inputs the round-3 kernels could not take or were never measured on:
headers with hundreds of ##INFO / ##FORMAT keys (exon derives the info STRUCT / formats LIST(STRUCT) from however many the header
has: rust/src/arrow_reader.rs:116-123, exon/src/exon/arrow_table_function/module.cpp:126-147), FORMAT columns of 80 keys, INFO
fields of several KiB, cohort lines — every value against oracle.pyoracle.vcf_typed_rows, at the chunk boundary and through
new_reader's Arrow stream."""
import random

import pytest

from test_arrow_stream_gpu import same
from test_vcf_nested_gpu import reader_rows

pytestmark = pytest.mark.gpu

TYPES = ["Integer", "Float", "Flag", "String", "Character"]


def make_header(n_info, n_format, n_samples, seed=1):
    rng = random.Random(seed)
    info, fmt = [], []
    lines = [b"##fileformat=VCFv4.2"]
    for k in range(n_info):
        ty = TYPES[k % 5] if k >= 5 else ["Integer", "Float", "Flag", "String", "Integer"][k]
        num = "0" if ty == "Flag" else rng.choice(["1", "1", "A", ".", "2"])
        name = "I%d" % k if k % 7 else "INFO_KEY_WITH_A_LONG_NAME_%d" % k
        info.append((name, ty, num))
        lines.append(b'##INFO=<ID=%s,Number=%s,Type=%s,Description="x">' % (name.encode(), num.encode(), ty.encode()))
    for k in range(n_format):
        ty = ["String", "Integer", "Float", "Integer"][k % 4]
        num = rng.choice(["1", "1", "R", "G", "."])
        name = "F%d" % k
        fmt.append((name, ty, num))
        lines.append(b'##FORMAT=<ID=%s,Number=%s,Type=%s,Description="x">' % (name.encode(), num.encode(), ty.encode()))
    cols = b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"
    if n_samples:
        cols += b"\tFORMAT\t" + b"\t".join(b"S%d" % i for i in range(n_samples))
    lines.append(cols)
    return b"\n".join(lines) + b"\n", info, fmt


def value_of(rng, ty, num, long_strings=False):
    def one():
        if ty == "Integer":
            return rng.choice([b"0", b"-7", b"123456", b"2147483647", b"."]) if rng.random() < 0.3 else b"%d" % rng.randrange(-1000, 100000)
        if ty == "Float":
            return rng.choice([b"0.5", b"1e-3", b"-2.25", b"3", b".", b"nan", b"1.17549435e-38"]) if rng.random() < 0.5 else b"%.4f" % rng.random()
        n = rng.choice([0, 1, 3, 12, 13, 40]) if not long_strings else rng.choice([5, 200, 1500])
        s = bytes(rng.choice(b"ACGTacgt|_-+*/()[]xyz0123456789") for _ in range(n))
        if rng.random() < 0.05:
            s += b"%3B%zz%41"
        return s if s != b"." else b"x"
    if num == "1":
        return one()
    return b",".join(one() for _ in range(rng.choice([1, 1, 2, 3, 7])))


def make_line(rng, k, info, fmt, n_samples, n_entries, n_fmt_keys, long_strings=False):
    ents = []
    picks = rng.sample(range(len(info)), min(n_entries, len(info))) if info else []
    for q in picks:
        name, ty, num = info[q]
        if ty == "Flag":
            ents.append(name.encode() + (b"" if rng.random() < 0.8 else b"=1"))
        elif rng.random() < 0.03:
            ents.append(name.encode())   # a key without a value
        else:
            ents.append(name.encode() + b"=" + value_of(rng, ty, num, long_strings))
    if rng.random() < 0.3:
        ents.insert(rng.randrange(len(ents) + 1), b"UNDECLARED=5")
    if picks and rng.random() < 0.2:   # a repeated key: the first occurrence wins
        q = rng.choice(picks)
        name, ty, num = info[q]
        ents.append(name.encode() + (b"" if ty == "Flag" else b"=" + value_of(rng, ty, num)))
    if rng.random() < 0.05:
        ents.append(b"")
    inf = b";".join(ents) if ents else b"."
    line = b"%d\t%d\trs%d\tA\t%s\t%s\t%s\t%s" % (k % 22 + 1, 100 + k, k, rng.choice([b"C", b"C,G", b"<DEL>,AAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"."]),
                                                  rng.choice([b".", b"30", b"1e2"]), rng.choice([b"PASS", b".", b"q10;s50"]), inf)
    if n_samples:
        fk = rng.sample(range(len(fmt)), min(n_fmt_keys, len(fmt))) if fmt else []
        names = [fmt[q][0].encode() for q in fk]
        if rng.random() < 0.2:
            names.insert(rng.randrange(len(names) + 1), b"ZZ")
        samples = []
        for _ in range(n_samples):
            r = rng.random()
            if r < 0.05:
                samples.append(b".")
                continue
            nv = len(names) if r < 0.8 or not names else rng.randrange(1, len(names) + 1)
            vals = []
            for nm in names[:nv]:
                d = [f for f in fmt if f[0].encode() == nm]
                vals.append(value_of(rng, d[0][1], d[0][2]) if d else b"junk")
            samples.append(b":".join(vals))
        line += b"\t" + b":".join(names) + b"\t" + b"\t".join(samples)
    return line


def check(oracle, tmp_path, data, name="w.vcf", **kw):
    (tmp_path / name).write_bytes(data)
    exp, err = oracle.vcf_typed_rows(data)
    assert err is None
    got = reader_rows(str(tmp_path / name), **kw)
    assert len(got) == len(exp)
    for i, (g, e) in enumerate(zip(got, exp)):
        assert same(g, e), (i, {k: (g[k], e[k]) for k in g if not same(g[k], e[k])})
    return got


@pytest.mark.parametrize("batch_rows", [64, 2048])
def test_wide_header_1000_info_200_format_keys(gpu, oracle, tmp_path, batch_rows):
    # a gnomAD-class header: the file must open, COUNT(*) and flat projections included, and every child must be right
    rng = random.Random(5)
    hdr, info, fmt = make_header(1000, 200, 3)
    lines = [make_line(rng, k, info, fmt, 3, rng.choice([0, 3, 40, 150, 400]), rng.choice([1, 5, 80])) for k in range(240)]
    data = hdr + b"\n".join(lines) + b"\n"
    got = check(oracle, tmp_path, data, batch_rows=batch_rows)
    assert len(got[0]["info"]) == 1000 and len(got[0]["formats"][0]) == 200
    from exon_duckdb_amd.reader import ShardReader
    r = ShardReader(str(tmp_path / "w.vcf"), "vcf")
    assert r.count() == 240
    r.close()
    r = ShardReader(str(tmp_path / "w.vcf"), "vcf", columns=[0, 1, 3])
    assert [t[1] for t in r.rows()] == [100 + k for k in range(240)]
    r.close()


def test_format_column_of_80_keys(gpu, oracle, tmp_path):
    # no FORMAT position is dropped (round 5 kept the first 64 silently)
    rng = random.Random(6)
    hdr, info, fmt = make_header(6, 120, 5)
    lines = [make_line(rng, k, info, fmt, 5, 3, 80) for k in range(60)]
    data = hdr + b"\n".join(lines) + b"\n"
    got = check(oracle, tmp_path, data)
    assert any(sum(v is not None for v in s.values()) > 64 for row in got for s in row["formats"])


@pytest.mark.parametrize("device_batch", [0, 96 << 10])
def test_narrow_header_with_long_and_short_info_fields(gpu, oracle, tmp_path, device_batch):
    # <= 32 keys: k_rows takes the fields of up to 128 bytes, k_info_wide the others — lists and repeated keys on both sides
    rng = random.Random(7)
    hdr, info, fmt = make_header(30, 4, 2)
    lines = []
    for k in range(3000):
        n_ent = rng.choice([0, 1, 2, 3, 5, 12, 30])
        lines.append(make_line(rng, k, info, fmt, 2, n_ent, 3, long_strings=(k % 97 == 0)))
    data = hdr + b"\n".join(lines) + b"\n"
    check(oracle, tmp_path, data, device_batch_bytes=device_batch)


def test_repeated_keys_first_wins_everywhere(gpu, oracle, tmp_path):
    hdr, info, fmt = make_header(40, 3, 1)   # > 32 keys: every row is the wave kernel's
    name = [i for i in info if i[1] == "Integer" and i[2] == "1"][0][0].encode()
    lname = [i for i in info if i[1] == "Integer" and i[2] != "1"][0][0].encode()
    filler = b";".join(b"U%d=%d" % (i, i) for i in range(200))   # > 1 KiB of undeclared keys: the repeat lies in another piece
    lines = [
        b"1\t1\t.\tA\tC\t.\t.\t" + name + b"=1;" + name + b"=2",
        b"1\t2\t.\tA\tC\t.\t.\t" + name + b"=.;" + name + b"=2",
        b"1\t3\t.\tA\tC\t.\t.\t" + name + b"=3;" + filler + b";" + name + b"=bad",
        b"1\t4\t.\tA\tC\t.\t.\t" + lname + b"=1,2,3;" + lname + b"=4;" + filler + b";" + lname + b"=5,6",
        b"1\t5\t.\tA\tC\t.\t.\t" + b";".join([name + b"=%d" % i for i in range(70)]),
    ]
    data = hdr[:hdr.rindex(b"\tFORMAT")] + b"\n" + b"\n".join(lines) + b"\n"
    got = check(oracle, tmp_path, data)
    assert got[0]["info"][name.decode()] == 1 and got[1]["info"][name.decode()] is None and got[4]["info"][name.decode()] == 0


def test_sample_edge_shapes(gpu, oracle, tmp_path):
    hdr, info, fmt = make_header(3, 4, 4)
    f = [x[0].encode() for x in fmt]
    lines = [
        b"1\t1\t.\tA\tC\t.\t.\t.\t" + f[0] + b"\t0/1\t\t.\t",                      # empty samples, a trailing tab
        b"1\t2\t.\tA\tC\t.\t.\t.\t\t.\t.",                                          # an empty FORMAT
        b"1\t3\t.\tA\tC\t.\t.\t.\t" + f[0] + b":" + f[0] + b"\ta:b\tc:d:e:f",       # a key twice; more values than keys
        b"1\t4\t.\tA\tC\t.\t.\t.\t" + b":".join(f) + b"\tx",                        # fewer values than keys
        b"1\t5\t.\tA\tC\t.\t.\t.\t" + f[0],                                         # FORMAT and no sample
        b"1\t6\t.\tA\tC\t.\t.\t.",                                                  # eight fields
        b"1\t7\t.\tA\tC\t.\t.\t.\t" + f[0] + b"\t" + b"\t".join(b"g%d" % i for i in range(700)),   # more samples than a piece holds
    ]
    data = hdr + b"\n".join(lines) + b"\n"
    got = check(oracle, tmp_path, data)
    assert len(got[0]["formats"]) == 4 and len(got[6]["formats"]) == 700 and got[4]["formats"] == [] and got[5]["formats"] == []


@pytest.mark.parametrize("n_samples,n_lines", [(100, 400), (2504, 12)])
def test_cohort_lines_with_formats(gpu, oracle, tmp_path, n_samples, n_lines):
    # the reference's real schema on multi-sample input: formats LIST(STRUCT) with scalar and list keys, thousands of samples a line
    rng = random.Random(8)
    hdr = (b"##fileformat=VCFv4.2\n"
           b'##INFO=<ID=AC,Number=A,Type=Integer,Description="x">\n##INFO=<ID=AF,Number=A,Type=Float,Description="x">\n'
           b'##INFO=<ID=AN,Number=1,Type=Integer,Description="x">\n##INFO=<ID=VT,Number=.,Type=String,Description="x">\n'
           b'##FORMAT=<ID=GT,Number=1,Type=String,Description="x">\n##FORMAT=<ID=AD,Number=R,Type=Integer,Description="x">\n'
           b'##FORMAT=<ID=DP,Number=1,Type=Integer,Description="x">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="x">\n'
           b'##FORMAT=<ID=PL,Number=G,Type=Integer,Description="x">\n'
           b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + b"\t".join(b"S%05d" % i for i in range(n_samples)) + b"\n")
    lines = []
    for k in range(n_lines):
        head = b"%d\t%d\trs%d\tA\tC\t50\tPASS\tAC=%d;AF=0.%04d;AN=%d;VT=SNP" % (k % 22 + 1, 1000 + k, k, k, k % 10000, 2 * n_samples)
        if k % 3 == 0:
            fmt, mk = b"GT", lambda: rng.choice([b"0|0", b"0|1", b"1|1", b".|."])
        else:
            fmt = b"GT:AD:DP:GQ:PL"
            mk = lambda: (b"./.:.:.:.:." if rng.random() < 0.1 else b"%s:%d,%d:%d:%d:%d,%d,%d" % (
                rng.choice([b"0/0", b"0/1", b"1/1"]), rng.randrange(60), rng.randrange(60), rng.randrange(120), rng.randrange(99),
                rng.randrange(999), rng.randrange(999), rng.randrange(999)))
        lines.append(head + b"\t" + fmt + b"\t" + b"\t".join(mk() for _ in range(n_samples)))
    data = hdr + b"\n".join(lines) + b"\n"
    got = check(oracle, tmp_path, data)
    assert all(len(r["formats"]) == n_samples for r in got)


def test_long_string_values_behind_the_staged_piece(gpu, oracle, tmp_path):
    # annotation-like values of several KiB (scalar and list): bytes behind a piece's halo are read from global memory
    rng = random.Random(9)
    hdr, info, fmt = make_header(45, 2, 1)
    s_key = [i for i in info if i[1] == "String" and i[2] == "1"][0][0].encode()
    l_key = [i for i in info if i[1] == "String" and i[2] != "1"][0][0].encode()
    i_key = [i for i in info if i[1] == "Integer" and i[2] == "1"][0][0].encode()
    big = lambda n: bytes(rng.choice(b"ACGT|&=:") for _ in range(n)).replace(b";", b"x")
    lines = []
    for k in range(40):
        ents = [i_key + b"=%d" % k, s_key + b"=" + big(rng.choice([900, 1100, 3000])), l_key + b"=" + b",".join(big(rng.choice([10, 400, 2000])) for _ in range(4)),
                b"TAIL=1"]
        rng.shuffle(ents)
        lines.append(b"1\t%d\t.\tA\tC\t.\t.\t" % (k + 1) + b";".join(ents) + b"\t" + fmt[0][0].encode() + b"\tv")
    data = hdr + b"\n".join(lines) + b"\n"
    check(oracle, tmp_path, data)


def test_more_percent_decoded_bytes_than_the_side_buffer(gpu, oracle, tmp_path):
    # > 64 KiB of decoded String values in one device batch: the writing pass is repeated with a buffer that holds them
    hdr, info, fmt = make_header(5, 1, 0)
    s_key = [i for i in info if i[1] == "String"][0][0].encode()
    lines = [b"1\t%d\t.\tA\tC\t.\t.\t" % (k + 1) + s_key + b"=" + b"a%3Bb%2C" * 12 + b"%d" % k for k in range(3000)]
    data = hdr + b"\n".join(lines) + b"\n"
    got = check(oracle, tmp_path, data)
    assert got[7]["info"][s_key.decode()] == "a;b," * 12 + "7"


def test_arrow_boundary_wide_and_cohort(gpu, oracle, tmp_path):
    from exon_duckdb_amd.arrow import new_reader
    rng = random.Random(10)
    hdr, info, fmt = make_header(300, 60, 7)
    lines = [make_line(rng, k, info, fmt, 7, rng.choice([0, 5, 60, 200]), rng.choice([1, 6, 40])) for k in range(150)]
    data = hdr + b"\n".join(lines) + b"\n"
    (tmp_path / "a.vcf").write_bytes(data)
    exp, err = oracle.vcf_typed_rows(data)
    assert err is None
    got = new_reader(str(tmp_path / "a.vcf"), "vcf", batch_size=64).read_all().to_pylist()
    assert len(got) == len(exp)
    for i, (g, e) in enumerate(zip(got, exp)):
        assert same(g, e), (i, {k: (g[k], e[k]) for k in g if not same(g[k], e[k])})


def test_value_error_in_a_wide_row_and_in_a_sample(gpu, oracle, tmp_path):
    from exon_duckdb_amd import ExgError
    from exon_duckdb_amd.reader import ShardReader
    hdr, info, fmt = make_header(50, 3, 2)
    i_key = [i for i in info if i[1] == "Integer" and i[2] == "1"][0][0].encode()
    f_int = [f for f in fmt if f[1] == "Integer"][0][0].encode()
    ok = b"1\t%d\t.\tA\tC\t.\t.\t" + i_key + b"=5\t" + f_int + b"\t1\t2"
    for bad in (b"1\t3\t.\tA\tC\t.\t.\t" + i_key + b"=five\t" + f_int + b"\t1\t2", b"1\t3\t.\tA\tC\t.\t.\t" + i_key + b"=5\t" + f_int + b"\t1\tx"):
        data = hdr + b"\n".join([ok % 1, ok % 2, bad, ok % 4]) + b"\n"
        (tmp_path / "bad.vcf").write_bytes(data)
        assert oracle.vcf_typed_rows(data)[1] == 2
        r = ShardReader(str(tmp_path / "bad.vcf"), "vcf")
        with pytest.raises(ExgError):
            r.rows()
        r.close()


@pytest.mark.parametrize("device_batch", [0, 64 << 10])
def test_small_rows_with_a_few_longer_fields(gpu, oracle, tmp_path, device_batch):
    # nearly every INFO field <= 64 bytes: k_rows runs with its 64-byte row (picked per batch from the counting pass's tally of the
    # fields of 65 - 128 bytes) and the few longer fields — 65 to 128 bytes, and beyond — are the wave kernel's; lists on both sides
    rng = random.Random(11)
    hdr, info, fmt = make_header(8, 2, 1)
    lines = []
    for k in range(4000):
        mid, long_ = k % 400 == 7, k % 1000 == 13
        while True:
            line = make_line(rng, k, info, fmt, 1, 8 if long_ else rng.choice([3, 4, 5, 6]) if mid else rng.choice([0, 1, 2, 3]), 2, long_strings=long_)
            n_info = len(line.split(b"\t")[7])
            if long_ or (mid and 64 < n_info <= 128) or (not mid and n_info <= 64):
                break
        lines.append(line)
    data = hdr + b"\n".join(lines) + b"\n"
    lens = [len(l.split(b"\t")[7]) for l in lines]
    assert sum(64 < x <= 128 for x in lens) * 64 <= len(lines) and any(64 < x <= 128 for x in lens) and any(x > 128 for x in lens)
    check(oracle, tmp_path, data, device_batch_bytes=device_batch)


def test_wide_header_over_many_short_lines(gpu, oracle, tmp_path):
    # 1 000 declared keys, lines that carry two of them: the children are a value slot per key and row whatever the lines hold, so the
    # device batch shrinks with the header (a 256 MiB batch of such lines would ask for tens of GB of vectors)
    from exon_duckdb_amd.reader import ShardReader
    hdr, info, fmt = make_header(1000, 0, 0)
    ints = [i[0].encode() for i in info if i[1] == "Integer" and i[2] == "1"]
    n = 120000
    lines = [b"%d\t%d\t.\tA\tC\t.\t.\t%s=%d;%s=%d" % (k % 22 + 1, k + 1, ints[k % len(ints)], k, ints[(k + 7) % len(ints)], -k) for k in range(n)]
    data = hdr + b"\n".join(lines) + b"\n"
    p = tmp_path / "ws.vcf"
    p.write_bytes(data)
    r = ShardReader(str(p), "vcf")
    assert r.stats()["device_batch_bytes"] <= (256 << 20) * 64 // 1000
    rows = 0
    import ctypes as C
    from exon_duckdb_amd.table_function import Chunk, decode_vector
    # (120 000 rows x 1 000 children in Python would take minutes: the first and the last chunk are decoded, the others counted)
    first = last = None
    while True:
        ch = Chunk()
        assert r._l.exg_next_chunk(r._r, C.byref(ch)) == 0
        if ch.n_rows == 0:
            break
        if first is None or rows + int(ch.n_rows) == n:
            info_col = decode_vector(ch.vectors[7].contents, r.trees[7])
            if first is None:
                first = info_col[0]
            last = info_col[-1]
        rows += int(ch.n_rows)
        r._l.exg_release_chunk(r._r, C.byref(ch))
    r.close()
    assert rows == n
    k = n - 1
    assert first[ints[0].decode()] == 0 and first[ints[7 % len(ints)].decode()] == 0 and sum(v is not None for v in first.values()) <= 2
    assert last[ints[k % len(ints)].decode()] == k and last[ints[(k + 7) % len(ints)].decode()] == -k


def test_format_strings_change_from_line_to_line(gpu, oracle, tmp_path):
    # the line before's FORMAT string is kept per wavefront: lines that repeat it, change it, end it with ':' (an empty last key),
    # make it exactly 64 / 65 / 256 / 257 bytes, or longer than a staged piece
    hdr, info, fmt = make_header(2, 90, 2)
    f = [x[0].encode() for x in fmt]
    def line(k, keys, vals=b"1"):
        return b"1\t%d\t.\tA\tC\t.\t.\t.\t" % (k + 1) + b":".join(keys) + b"\t" + b":".join([vals] * max(1, len(keys))) + b"\t."
    pad = lambda n: [b"Z" * n]   # noqa: E731  (an undeclared key of n bytes)
    shapes_ = [f[:1], f[:1], f[:3], f[:3], f[:1], f[:2] + [b""], f[:2] + [b""], [b""], f[:1] * 3, f[2:5],
               f[:1] + pad(62), f[:1] + pad(63), f[:1] + pad(254), f[:1] + pad(255), f[:80], f[:80], f[:1] + pad(1500), f[:1] + pad(1500), f[:1]]
    lines = [line(k, shapes_[k % len(shapes_)]) for k in range(len(shapes_) * 40)]
    data = hdr + b"\n".join(lines) + b"\n"
    check(oracle, tmp_path, data)


@pytest.mark.parametrize("first", range(0, 48, 8))
def test_soak_seeds_against_the_oracle(gpu, oracle, tmp_path, first):
    # tools/vcf_nested_soak.py's first seeds: random headers (0 - 600 INFO keys, 0 - 90 FORMAT keys, 0 - 300 samples), random lines,
    # structural mutations inside the INFO / FORMAT / sample fields, random chunk and device batch sizes: the rows in front of the
    # first value error and whether there is one must be the oracle's
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import vcf_nested_soak
    for seed in range(first, first + 8):
        assert vcf_nested_soak.run_seed(seed, str(tmp_path / "s.vcf")) is None
