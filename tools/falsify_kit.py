#!/usr/bin/env python3
"""Falsifiability kit: every input of tests/test_oracle_rules.py — the [RECALLED] noodles / exon rules that the reference's
own sqllogictests do not pin (parity unpinned) — written out as files, with the oracle's answer beside each and ONE
script that puts them through a real exon build:

    python tools/falsify_kit.py kit/              # here (needs only the oracle): kit/cases/*, kit/expected/*, kit/run.sh
    cd kit && DUCKDB=/path/to/duckdb ./run.sh     # on a machine with DuckDB v0.8.1 + the reference's exon.duckdb_extension
    python compare.py                             # -> one line per case: SAME / DIFFERENT (+ what differs); exit 1 on any

Nothing of the reference travels anywhere: the kit holds inputs made by this repository's tests and this repository's oracle's
rows.  A DIFFERENT line names a rule of tests/test_oracle_rules.py to fix in the oracle (the GPU path follows the oracle
bit for bit, so it is fixed with it).

How the cases are found: the rule tests are plain functions of the `oracle` fixture; they are run here against a recording
proxy of oracle/pyoracle.py that notes every parse call (format, input bytes) and what it returned.
"""
import inspect
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Recorder:
    """pyoracle with its three parsers wrapped"""

    def __init__(self, mod):
        self._m = mod
        self.calls = []          # (test name, format, bytes)
        self.current = ""

    def __getattr__(self, k):
        v = getattr(self._m, k)
        fmt = {"fastq_parse": "fastq", "fasta_parse": "fasta", "vcf_parse": "vcf", "vcf_typed_rows": "vcf"}.get(k)
        if not fmt:
            return v

        def wrapped(data, *a, **kw):
            self.calls.append((self.current, fmt, bytes(data)))
            return v(data, *a, **kw)
        return wrapped


def jsonable(x):
    if isinstance(x, bytes):
        try:
            return x.decode("utf-8")
        except UnicodeDecodeError:
            return {"bytes_hex": x.hex()}
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if hasattr(x, "item"):
        return x.item()
    return x


def expected_of(o, fmt, data):
    """-> (rows as a list of dicts in the reference's schema, error text or None)"""
    if fmt == "fastq":
        t = o.fastq_parse(data, want_string_t=False)
        names = ["name", "description", "sequence", "quality_scores"]
        rows = [dict(zip(names, r)) for r in zip(*[t.columns[c].to_list() for c in names])]
    elif fmt == "fasta":
        t = o.fasta_parse(data)
        names = ["id", "description", "sequence"]
        rows = [dict(zip(names, r)) for r in zip(*[t.columns[c].to_list() for c in names])]
    else:
        t = o.vcf_parse(data, want_string_t=False)
        if t.error_code:
            rows = []
        else:
            rows, err_row = o.vcf_typed_rows(data)
            if err_row is not None:
                return jsonable(rows), f"a value of row {err_row} does not parse"
    err = None
    if t.error_code:
        err = f"{t.error_message} (record {t.error_record})"
    return jsonable(rows), err


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "kit"
    from oracle import pyoracle
    pyoracle.lib()
    import test_oracle_rules as rules
    rec = Recorder(pyoracle)
    golden = os.path.join(ROOT, "tests", "golden")
    for name, fn in sorted(inspect.getmembers(rules, inspect.isfunction)):
        if not name.startswith("test_"):
            continue
        params = list(inspect.signature(fn).parameters)
        rec.current = name
        try:
            if params == ["oracle"]:
                fn(rec)
            elif params == ["oracle", "golden_dir"]:
                fn(rec, golden)
            # (parametrised helpers — UTF-8 acceptance tables — call no parser)
        except AssertionError:
            print(f"warning: {name} does not hold against this oracle", file=sys.stderr)
    os.makedirs(os.path.join(out, "cases"), exist_ok=True)
    os.makedirs(os.path.join(out, "expected"), exist_ok=True)
    seen, cases = {}, []
    for test, fmt, data in rec.calls:
        key = (fmt, data)
        if key in seen:
            continue
        k = sum(1 for c in cases if c["test"] == test)
        case = f"{test[5:]}_{k}"
        seen[key] = case
        ext = {"fastq": "fastq", "fasta": "fasta", "vcf": "vcf"}[fmt]
        with open(os.path.join(out, "cases", f"{case}.{ext}"), "wb") as f:
            f.write(data)
        rows, err = expected_of(pyoracle, fmt, data)
        with open(os.path.join(out, "expected", f"{case}.json"), "w") as f:
            json.dump({"rows": rows, "error": err}, f, indent=1, sort_keys=True)
        cases.append({"test": test, "case": case, "format": fmt, "file": f"cases/{case}.{ext}"})
    with open(os.path.join(out, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    fn_of = {"fastq": "read_fastq", "fasta": "read_fasta", "vcf": "read_vcf_file_records"}
    with open(os.path.join(out, "run.sh"), "w") as f:
        f.write("#!/bin/sh\n# one DuckDB invocation per case (an error must not end the others); DUCKDB = the CLI of a build with the exon extension\n"
                ": ${DUCKDB:=duckdb}\nmkdir -p got\n")
        for c in cases:
            sql = (f"LOAD exon; COPY (SELECT * FROM {fn_of[c['format']]}('{c['file']}')) TO 'got/{c['case']}.json' (FORMAT JSON);")
            f.write(f"$DUCKDB -unsigned -c \"{sql}\" > got/{c['case']}.log 2>&1 || cp got/{c['case']}.log got/{c['case']}.err\n")
    os.chmod(os.path.join(out, "run.sh"), 0o755)
    with open(os.path.join(out, "compare.py"), "w") as f:
        f.write(COMPARE)
    print(f"{len(cases)} cases of {len(set(c['test'] for c in cases))} rule tests -> {out}/ (cases/, expected/, run.sh, compare.py)")


COMPARE = r'''#!/usr/bin/env python3
"""expected/ (this repository's oracle) against got/ (a real exon build, written by run.sh): one line per case"""
import json, os, struct, sys
cases = json.load(open("cases.json"))
bad = 0


def f32(x):
    return struct.unpack("f", struct.pack("f", float(x)))[0] if isinstance(x, (int, float)) and not isinstance(x, bool) else x


def norm(x):
    if isinstance(x, dict):
        if set(x) == {"bytes_hex"}:
            return x
        return {k: norm(v) for k, v in x.items()}
    if isinstance(x, list):
        return [norm(v) for v in x]
    if isinstance(x, float):
        return "nan" if x != x else f32(x)
    return x


for c in cases:
    exp = json.load(open(f"expected/{c['case']}.json"))
    got_err = os.path.exists(f"got/{c['case']}.err")
    rows = []
    if os.path.exists(f"got/{c['case']}.json"):
        with open(f"got/{c['case']}.json") as f:
            rows = [json.loads(line) for line in f if line.strip()]
    if exp["error"]:
        same = got_err     # the reference fails the query; which rows came first is not visible through SQL
        detail = "" if same else f"expected an error ({exp['error']}), got {len(rows)} rows"
    else:
        same = not got_err and norm(rows) == norm(exp["rows"])
        detail = "" if same else ("the query failed: " + open(f"got/{c['case']}.err").read()[-200:].strip() if got_err else
                                  f"rows differ: expected {json.dumps(exp['rows'])[:300]} got {json.dumps(rows)[:300]}")
    bad += not same
    print(("SAME      " if same else "DIFFERENT ") + f"{c['test']} [{c['case']}] {detail}")
sys.exit(1 if bad else 0)
'''

if __name__ == "__main__":
    main()
