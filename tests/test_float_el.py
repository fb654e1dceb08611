"""exg_float_el.hpp (Eisel-Lemire decimal -> float32: what makes the device's f32::from_str exact beyond 15 digits and
|exponent| 22) compiled for the host and checked against strtof on ~1.5 M literals (tests/float_el_check.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eisel_lemire_matches_strtof(tmp_path):
    exe = tmp_path / "float_el_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "exon_duckdb_amd", "csrc"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "float_el_check.cpp")])
    out = subprocess.check_output([str(exe)]).decode()
    assert out.strip().endswith("0 bad"), out


def test_pow5_table_is_what_the_generator_writes(tmp_path):
    # the committed table is the generator's output (tools/gen_pow5_table.py)
    path = os.path.join(ROOT, "exon_duckdb_amd", "csrc", "exg_pow5_table.hpp")
    out = tmp_path / "table.hpp"
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "gen_pow5_table.py"), str(out)], stdout=subprocess.DEVNULL)
    assert open(path).read() == out.read_text()
