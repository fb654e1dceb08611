#!/usr/bin/env python3
"""Condense gpurun_out/<tag>_* (tools/collect_profiles.sh) into profiles/.

  python tools/pmc_summary.py r01_d

Copies the kernel-stats CSV and the per-dispatch counter rows of the dominant kernel, and rewrites
profiles/pmc_fastq_fused.json (read by bench.py for roofline.traffic).  rocprofv3 reports
FETCH_SIZE/WRITE_SIZE in kilobytes (x1024 -> bytes); on gfx950 FETCH_SIZE under-reports wide
streaming reads by 2x (MI355X_MICROARCH.md, HBM section), so reads are doubled.
"""
import csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_d"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
KERNEL = "k_fused<exg::FastqFormat, 0>"   # the lean scan (1: the any-shape scan, 2: its redo run)


def find(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


def counter_rows(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Kernel_Name"]:
                rows.append(r)
    return rows


res = {}
stats = find(f"{tag}_kt/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(prof, f"{tag}_kernel_stats.csv"))
    with open(stats) as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Name"]:
                res["kernel_avg_ns"] = float(r["AverageNs"])
                res["kernel_calls"] = int(r["Calls"])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = find(f"{tag}_pmc_{c}/**/*counter_collection.csv")
    if not p:
        continue
    rows = counter_rows(p)
    with open(os.path.join(prof, f"{tag}_pmc_{c.lower()}_k_fused.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c]
    res[c] = sum(vals) / len(vals)
for leg in ("configs", "zstd", "gzstream", "fasta", "fasta_e2e", "vcf", "inflate", "shapes", "vcf_nested", "vcf_cohort100", "vcf_cohort2504"):
    st = find(f"{tag}_kt_{leg}/**/*kernel_stats.csv")
    if st:
        shutil.copy(st, os.path.join(prof, f"{tag}_{leg}_kernel_stats.csv"))
# the nested VCF kernels' HBM traffic (round 6): per dispatch of k_rows / k_info_wide / k_samples / the batched scans, both counters
nested = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = find(f"{tag}_pmcn_{c}/**/*counter_collection.csv")
    if not p:
        continue
    rows = [r for r in csv.DictReader(open(p)) if "exg::vn::" in r["Kernel_Name"] and r["Counter_Name"] == c]
    if rows:
        with open(os.path.join(prof, f"{tag}_pmc_{c.lower()}_vcf_nested.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    for r in rows:
        mm = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
        k = mm.group(1) if mm else r["Kernel_Name"][:40]
        nested.setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
if nested:
    res["vcf_nested_traffic"] = {k: {c: {"dispatches": len(v), "avg_KiB": sum(v) / len(v), "avg_bytes": sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)}
                                     for c, v in d.items()} for k, d in nested.items()}
# SQ activity counters of k_inflate (tools/pmc_inflate.sh): averages per counter
import collections
acc = collections.defaultdict(list)
for fcsv in glob.glob(os.path.join(out, f"{tag}_pmcinf_*", "**", "*counter_collection.csv"), recursive=True):
    with open(fcsv) as f:
        for r in csv.DictReader(f):
            if "k_inflate" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc:
    with open(os.path.join(prof, f"{tag}_pmc_sq_k_inflate.csv"), "w") as f:
        f.write("counter,average_per_dispatch,dispatches\n")
        for k in sorted(acc):
            f.write(f"{k},{sum(acc[k]) / len(acc[k])},{len(acc[k])}\n")
b = os.path.join(out, f"{tag}_bench.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join(prof, f"{tag}_bench.json"))
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    old = json.load(open(os.path.join(prof, "pmc_fastq_fused.json")))
    rd = res["FETCH_SIZE"] * 1024 * 2
    wr = res["WRITE_SIZE"] * 1024
    old.update({
        "tag": tag,
        "FETCH_SIZE_bytes_raw": res["FETCH_SIZE"] * 1024,
        "read_bytes_per_launch": rd,
        "write_bytes_per_launch": wr,
        "hbm_bytes_per_launch_10GB": rd + wr,
        "kernel_avg_ns_rocprofv3": res.get("kernel_avg_ns"),
    })
    old.pop("sq_counters_per_launch", None)
    json.dump(old, open(os.path.join(prof, "pmc_fastq_fused.json"), "w"), indent=1)
if res.get("vcf_nested_traffic"):
    json.dump({"tag": tag, "what": "HBM traffic per dispatch of the nested-VCF kernels over one all-columns drain of the 2.08 GB VCF-8 file (device batches of 256 MiB "
                                   "= ~5.3 M lines; the first batches of a file are smaller: exg_reader.hpp ramp_bytes): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                   "passes of their own, KiB x 1024, FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md)",
               "kernels": res["vcf_nested_traffic"]}, open(os.path.join(prof, f"{tag}_vcf_nested_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
