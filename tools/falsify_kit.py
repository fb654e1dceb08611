#!/usr/bin/env python3
"""Falsifiability kit: every input of tests/test_oracle_rules.py — the [RECALLED] noodles / exon rules that the reference's
own sqllogictests do not pin (parity unpinned) — written out as files, with the oracle's answer beside each and ONE
script that puts them through a real exon build.  Since round 5 also the DECODER-level rules (what the rows of a multi-member
gzip / BGZF / multi-frame zstd file are, what trailing bytes, truncations and wrong checksums give: the case is the compressed
file itself) and the SCHEMA rules (column names and types: `DESCRIBE SELECT * FROM read_*(...)`, diffed by compare.py):

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
        self.calls = []          # (test name, format, bytes[, compression, file extension])
        self.schemas = []        # (test name, format, bytes)
        self.current = ""

    def __getattr__(self, k):
        v = getattr(self._m, k)
        if k == "compressed_parse":   # decoder-level rules: the case is the COMPRESSED file
            def wrapped_c(fmt, data, compression, ext=None):
                self.calls.append((self.current, fmt, bytes(data), compression, ext or fmt))
                return v(fmt, data, compression, ext)
            return wrapped_c
        if k == "schema_of":          # schema rules: DESCRIBE of the table function over this input
            def wrapped_s(fmt, data=b""):
                self.schemas.append((self.current, fmt, bytes(data)))
                return v(fmt, data)
            return wrapped_s
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


def rows_of_table(fmt, t):
    names = {"fastq": ["name", "description", "sequence", "quality_scores"], "fasta": ["id", "description", "sequence"]}[fmt]
    return [dict(zip(names, r)) for r in zip(*[t.columns[c].to_list() for c in names])]


def expected_of(o, fmt, data, compression=None):
    """-> (rows as a list of dicts in the reference's schema, error text or None)"""
    if compression:
        r = o.compressed_parse(fmt, data, compression)
        if fmt == "vcf":
            rows = r.typed_rows or []
        else:
            rows = rows_of_table(fmt, r.table)
        return jsonable(rows), r.error
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
    for call in rec.calls:
        test, fmt, data = call[:3]
        compression, ext = (call[3], call[4]) if len(call) > 3 else (None, fmt)
        key = (fmt, data, compression, ext)
        if key in seen:
            continue
        k = sum(1 for c in cases if c["test"] == test)
        case = f"{test[5:]}_{k}"
        seen[key] = case
        with open(os.path.join(out, "cases", f"{case}.{ext}"), "wb") as f:
            f.write(data)
        rows, err = expected_of(pyoracle, fmt, data, compression)
        with open(os.path.join(out, "expected", f"{case}.json"), "w") as f:
            json.dump({"rows": rows, "error": err}, f, indent=1, sort_keys=True)
        c = {"test": test, "case": case, "format": fmt, "file": f"cases/{case}.{ext}"}
        # the `compression` named parameter is passed when the file's extension does not say it (read_fastq('x.fastq', compression = 'gzip'))
        if compression and pyoracle.infer_compression("x." + ext) != pyoracle.infer_compression("x." + ext, compression):
            c["compression"] = compression
        cases.append(c)
    # schema cases: DESCRIBE SELECT * FROM read_*(file) -> column names and types
    schemas = []
    for test, fmt, data in rec.schemas:
        if any(sc["format"] == fmt and sc["data"] == data for sc in schemas):
            continue
        case = f"schema_{test[5:]}_{fmt}"
        with open(os.path.join(out, "cases", f"{case}.{fmt}"), "wb") as f:
            f.write(data)
        with open(os.path.join(out, "expected", f"{case}.json"), "w") as f:
            json.dump({"schema": [{"column_name": n, "column_type": t} for n, t in pyoracle.schema_of(fmt, data)]}, f, indent=1)
        schemas.append({"test": test, "case": case, "format": fmt, "file": f"cases/{case}.{fmt}", "data": data})
    for sc in schemas:
        del sc["data"]
        sc["schema"] = True
        cases.append(sc)
    with open(os.path.join(out, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    fn_of = {"fastq": "read_fastq", "fasta": "read_fasta", "vcf": "read_vcf_file_records"}
    with open(os.path.join(out, "run.sh"), "w") as f:
        f.write("#!/bin/sh\n# one DuckDB invocation per case (an error must not end the others); DUCKDB = the CLI of a build with the exon extension\n"
                ": ${DUCKDB:=duckdb}\nmkdir -p got\n")
        for c in cases:
            arg = f"'{c['file']}'" + (f", compression = '{c['compression']}'" if c.get("compression") else "")
            if c.get("schema"):   # column names + types as DESCRIBE prints them (-json: one array of objects)
                f.write(f"$DUCKDB -unsigned -json -c \"LOAD exon; DESCRIBE SELECT * FROM {fn_of[c['format']]}({arg});\" > got/{c['case']}.json 2> got/{c['case']}.log "
                        f"|| cp got/{c['case']}.log got/{c['case']}.err\n")
                continue
            sql = (f"LOAD exon; COPY (SELECT * FROM {fn_of[c['format']]}({arg})) TO 'got/{c['case']}.json' (FORMAT JSON);")
            f.write(f"$DUCKDB -unsigned -c \"{sql}\" > got/{c['case']}.log 2>&1 || cp got/{c['case']}.log got/{c['case']}.err\n")
    os.chmod(os.path.join(out, "run.sh"), 0o755)
    with open(os.path.join(out, "compare.py"), "w") as f:
        f.write(COMPARE)
    n_dec = sum(1 for c in cases if not c.get("schema") and c["file"].rsplit(".", 1)[-1] not in ("fastq", "fasta", "vcf") or c.get("compression"))
    print(f"{len(cases)} cases of {len(set(c['test'] for c in cases))} rule tests ({n_dec} decoder-level, {sum(1 for c in cases if c.get('schema'))} schema) "
          f"-> {out}/ (cases/, expected/, run.sh, compare.py)")


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
    if c.get("schema"):
        got = []
        if not got_err and os.path.exists(f"got/{c['case']}.json"):
            try:
                got = [{"column_name": r["column_name"], "column_type": r["column_type"]} for r in json.load(open(f"got/{c['case']}.json"))]
            except Exception as e:  # noqa: BLE001
                got = [{"unreadable": str(e)}]
        same = got == exp["schema"]
        diff = [f"{e['column_name']}: expected {e['column_type']}, got {g.get('column_type')} ({g.get('column_name')})"
                for e, g in zip(exp["schema"], got + [{}] * len(exp["schema"])) if e != g]
        bad += not same
        print(("SAME      " if same else "DIFFERENT ") + f"{c['test']} [{c['case']}] " + ("" if same else "; ".join(diff)[:600] or f"{len(got)} columns against {len(exp['schema'])}"))
        continue
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
