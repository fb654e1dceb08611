"""The falsifiability kit (tools/falsify_kit.py) against THIS build: every case file the kit hands to a real exon build — the text
rules, the decoder-level rules (multi-member gzip, BGZF with and without its EOF block, trailing bytes, truncations, wrong
trailers, concatenated / skippable zstd frames, `compression =` on plain text) and the schema cases — goes through the device
reader, and what comes back must be what the kit's expected/ says: the kit describes the build it ships with."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kit(tmp_path_factory):
    out = tmp_path_factory.mktemp("kit")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "falsify_kit.py"), str(out)])
    return out, json.load(open(out / "cases.json"))


def _plain(x):
    """rows of the reader -> the kit's JSON form"""
    if isinstance(x, bytes):
        try:
            return x.decode("utf-8")
        except UnicodeDecodeError:
            return {"bytes_hex": x.hex()}
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def _same(a, b):
    if isinstance(a, float) and isinstance(b, float):
        return (a != a and b != b) or a == b or abs(a - b) <= 1e-6 * max(abs(a), abs(b))
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, list) and isinstance(b, list):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


def test_every_case_of_the_kit_through_the_device_reader(gpu, kit):
    from exon_duckdb_amd import ExgError
    from exon_duckdb_amd.reader import ShardReader
    from exon_duckdb_amd.table_function import type_sql
    out, cases = kit
    n_rows_cases = n_err_cases = n_schema = 0
    for c in cases:
        exp = json.load(open(out / "expected" / (c["case"] + ".json")))
        path = str(out / c["file"])
        if c.get("schema"):
            r = ShardReader(path, c["format"])
            got = [{"column_name": n, "column_type": type_sql(t)} for n, t in zip(r.names, r.trees)]
            r.close()
            assert got == exp["schema"], c["case"]
            n_schema += 1
            continue
        rows, err = None, None
        try:
            r = ShardReader(path, c["format"], compression=c.get("compression"))
            try:
                rows = r.rows()
            finally:
                r.close()
        except ExgError as e:
            err = str(e)
        if exp["error"]:
            assert err is not None, (c["case"], "expected an error: " + exp["error"], rows)
            n_err_cases += 1
        else:
            assert err is None, (c["case"], err)
            names = {"fastq": ["name", "description", "sequence", "quality_scores"], "fasta": ["id", "description", "sequence"],
                     "vcf": ["chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "formats"]}[c["format"]]
            got = [_plain(dict(zip(names, row))) for row in rows]
            assert len(got) == len(exp["rows"]) and all(_same(g, e) for g, e in zip(got, exp["rows"])), (c["case"], got[:2], exp["rows"][:2])
            n_rows_cases += 1
    assert n_schema == 3 and n_err_cases >= 20 and n_rows_cases >= 50
