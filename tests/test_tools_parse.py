"""tools/ holds the probes and A/B scripts the measurements in DESIGN.md came from; none of them runs in the suites (they need a
GPU box and minutes), so a rename in the package can break one unnoticed.  Here: every Python tool parses and every shell tool
passes `bash -n`; the helpers they import from the package exist."""
import ast
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) > 20
    for f in files:
        with open(f) as fh:
            ast.parse(fh.read(), filename=f)


def test_shell_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert files
    for f in files:
        subprocess.run(["bash", "-n", f], check=True)


def test_package_helpers_the_tools_import_exist():
    import importlib
    for mod, names in (("exon_duckdb_amd.testing.bgzf", ["bgzip"]), ("exon_duckdb_amd.testing.shapes", ["vcf_cohort_header", "vcf_multisample_block", "fastq_long_block"]),
                       ("exon_duckdb_amd.arrow", ["new_reader"]), ("exon_duckdb_amd.reader", ["ShardReader"])):
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
