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


def test_exact_parser_on_literals_beside_rounding_boundaries(tmp_path):
    """exg_float_slow.hpp (the big-integer decision for literals of more than 19 digits whose first 19 straddle a rounding
    boundary) against strtof: for random floats of every magnitude, the exact decimal expansion of the midpoint to the
    next float — itself, with a 1 appended after zeros, one unit below in the last place, cut after 19 / 20 / 40 digits."""
    import random
    import struct
    from fractions import Fraction
    exe = tmp_path / "float_el_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "exon_duckdb_amd", "csrc"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "float_el_check.cpp")])
    rnd = random.Random(31)

    def dec(fr):  # exact decimal expansion of a dyadic rational
        num, den = fr.numerator, fr.denominator
        k = den.bit_length() - 1
        assert den == 1 << k
        digits = str(num * 5 ** k)
        if k == 0:
            return digits
        digits = digits.rjust(k + 1, "0")
        return digits[:-k] + "." + digits[-k:]

    lits = []
    for _ in range(700):
        bits = rnd.choice([rnd.getrandbits(31) % 0x7F800000, rnd.getrandbits(23), 0x7F7FFFFF - rnd.getrandbits(4), (rnd.randrange(100, 160) << 23) | rnd.getrandbits(23)])
        f = Fraction(struct.unpack("<f", struct.pack("<I", bits))[0])
        nxt = struct.unpack("<f", struct.pack("<I", bits + 1))[0] if bits + 1 < 0x7F800000 else None
        g = Fraction(nxt) if nxt is not None else Fraction(2) ** 128
        mid = dec((f + g) / 2)
        lits += [mid, mid + "0" * 25 + "1", mid + "9", mid[:-1] + str(int(mid[-1]) - 1) + "9" * 30 if mid[-1] != "0" else mid + "0"]
        digits_only = mid.replace(".", "")
        for cut in (19, 20, 40):
            if len(digits_only.lstrip("0")) > cut and "." in mid:
                lits.append(mid[:len(mid) - (len(digits_only.lstrip("0")) - cut)] if len(mid) - (len(digits_only.lstrip("0")) - cut) > mid.index(".") + 1 else mid)
        if "." in mid:
            i = mid.index(".")
            lits.append(mid.replace(".", "") + "e-%d" % (len(mid) - i - 1))      # the same value with an exponent
    lits += ["1.00000005960464477539062500000000000000001", "16777217.0000000000000000000000001", "16777218.99999999999999999999",
             "0." + "0" * 44 + "7006492321624085354618647916449580656401309709382578858785341419448955413429303", "1e-46", "4e38",
             "340282356779733661637539395458142568447.99999", "340282356779733661637539395458142568448", "1." + "0" * 300 + "1",
             "0." + "3" * 400, "123456789012345678901234567890.123456789e-10"]
    path = tmp_path / "lits.txt"
    path.write_text("\n".join(lits) + "\n")
    out = subprocess.run([str(exe), str(path)], capture_output=True, text=True)
    assert out.returncode == 0 and " 0 bad" in out.stdout, out.stdout[-2000:]


def test_pow5_table_is_what_the_generator_writes(tmp_path):
    # the committed table is the generator's output (tools/gen_pow5_table.py)
    path = os.path.join(ROOT, "exon_duckdb_amd", "csrc", "exg_pow5_table.hpp")
    out = tmp_path / "table.hpp"
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "gen_pow5_table.py"), str(out)], stdout=subprocess.DEVNULL)
    assert open(path).read() == out.read_text()
