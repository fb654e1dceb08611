"""Named tests for the [RECALLED] noodles / exon rules the oracle restates beyond what the
reference's sqllogictests pin (parity UNPINNED): a run against a real exon build can falsify each."""
import numpy as np
import pytest

PE = dict(NAME=1, PLUS=2, EOF=3, UTF8=4, FA_PREFIX=5, FA_NAME=6, FA_EMPTY=7, VCF_FIELD=8, VCF_POS=9, VCF_QUAL=10,
          VCF_HDR=11)


def rows(r):
    return [[r.columns[k].row(i) for k in r.columns] for i in range(r.n_rows)]


# ---- FASTQ (noodles-fastq 0.8.0 read_record; exon FASTQArrayBuilder) -------------------------------

def test_fastq_empty_input_is_zero_rows(oracle):
    r = oracle.fastq_parse(b"")
    assert r.n_rows == 0 and r.error_code == 0


def test_fastq_split_at_first_space_only(oracle):
    r = oracle.fastq_parse(b"@id a b  c\nAC\n+\n!!\n")
    assert rows(r) == [[b"id", b"a b  c", b"AC", b"!!"]]


def test_fastq_tab_is_not_a_delimiter(oracle):
    r = oracle.fastq_parse(b"@id\tx\nAC\n+\n!!\n")
    assert rows(r) == [[b"id\tx", None, b"AC", b"!!"]]


def test_fastq_trailing_space_gives_null_description(oracle):
    r = oracle.fastq_parse(b"@id \nAC\n+\n!!\n")
    assert rows(r) == [[b"id", None, b"AC", b"!!"]]


def test_fastq_empty_name_is_allowed(oracle):
    r = oracle.fastq_parse(b"@\nAC\n+\n!!\n@ d\nAC\n+\n!!\n")
    assert rows(r) == [[b"", None, b"AC", b"!!"], [b"", b"d", b"AC", b"!!"]]


def test_fastq_crlf_is_stripped(oracle):
    r = oracle.fastq_parse(b"@id d\r\nAC\r\n+\r\n!!\r\n")
    assert rows(r) == [[b"id", b"d", b"AC", b"!!"]]


def test_fastq_plus_line_content_is_discarded(oracle):
    r = oracle.fastq_parse(b"@id\nAC\n+id again\n!!\n")
    assert rows(r) == [[b"id", None, b"AC", b"!!"]]


def test_fastq_quality_may_start_with_at(oracle):
    r = oracle.fastq_parse(b"@a\nAC\n+\n@!\n@b\nGT\n+\n@@\n")
    assert rows(r) == [[b"a", None, b"AC", b"@!"], [b"b", None, b"GT", b"@@"]]


def test_fastq_no_trailing_newline(oracle):
    r = oracle.fastq_parse(b"@a\nAC\n+\n!!")
    assert rows(r) == [[b"a", None, b"AC", b"!!"]] and r.error_code == 0


def test_fastq_cr_before_eof_without_lf_is_kept(oracle):
    r = oracle.fastq_parse(b"@a\nAC\n+\n!!\r")
    assert rows(r) == [[b"a", None, b"AC", b"!!\r"]]


def test_fastq_missing_quality_line_is_empty_quality(oracle):
    # read_line returns 0 bytes at EOF without error
    r = oracle.fastq_parse(b"@a\nAC\n+\n")
    assert rows(r) == [[b"a", None, b"AC", b""]] and r.error_code == 0
    r = oracle.fastq_parse(b"@a\nAC\n+")
    assert rows(r) == [[b"a", None, b"AC", b""]] and r.error_code == 0


def test_fastq_truncated_before_plus_is_unexpected_eof(oracle):
    for data in (b"@a\n", b"@a", b"@a\nAC\n", b"@a\nAC"):
        r = oracle.fastq_parse(b"@x\nAC\n+\n!!\n" + data)
        assert r.n_rows == 1 and r.error_code == PE["EOF"] and r.error_record == 1 and r.error_offset == 11


def test_fastq_bad_name_prefix(oracle):
    r = oracle.fastq_parse(b"@x\nAC\n+\n!!\nx\nAC\n+\n!!\n")
    assert r.n_rows == 1 and r.error_code == PE["NAME"] and r.error_record == 1 and r.error_offset == 11


def test_fastq_blank_line_between_records_is_an_error(oracle):
    r = oracle.fastq_parse(b"@x\nAC\n+\n!!\n\n@y\nAC\n+\n!!\n")
    assert r.n_rows == 1 and r.error_code == PE["NAME"]


def test_fastq_trailing_blank_line_is_an_error(oracle):
    r = oracle.fastq_parse(b"@x\nAC\n+\n!!\n\n")
    assert r.n_rows == 1 and r.error_code == PE["NAME"] and r.error_record == 1


def test_fastq_bad_plus_prefix(oracle):
    r = oracle.fastq_parse(b"@x\nAC\n-\n!!\n")
    assert r.n_rows == 0 and r.error_code == PE["PLUS"] and r.error_record == 0 and r.error_offset == 0
    r = oracle.fastq_parse(b"@x\nAC\n\n!!\n")
    assert r.error_code == PE["PLUS"]


def test_fastq_invalid_utf8_is_an_error(oracle):
    r = oracle.fastq_parse(b"@x\nAC\n+\n!!\n@y \xff\nAC\n+\n!!\n")
    assert r.n_rows == 1 and r.error_code == PE["UTF8"] and r.error_record == 1
    r = oracle.fastq_parse("@é ü\nAC\n+\n!!\n".encode())
    assert r.error_code == 0 and rows(r) == [["é".encode(), "ü".encode(), b"AC", b"!!"]]


def test_fastq_structural_error_wins_over_utf8_in_the_same_record(oracle):
    r = oracle.fastq_parse(b"@y \xff\nAC\n-\n!!\n")
    assert r.error_code == PE["PLUS"]


# ---- FASTA (noodles-fasta 0.27.0 read_definition / read_sequence; Definition::from_str) ---------------

def test_fasta_multiline_sequence_is_concatenated(oracle):
    r = oracle.fasta_parse(b">a d\nAC\nGT\n\nTT\n>b\nA\n")
    assert rows(r) == [[b"a", b"d", b"ACGTTT"], [b"b", None, b"A"]]


def test_fasta_description_is_trimmed_and_split_on_ascii_whitespace(oracle):
    r = oracle.fasta_parse(b">a\t  two words \nAC\n")
    assert rows(r) == [[b"a", b"two words", b"AC"]]


def test_fasta_trailing_space_gives_empty_not_null_description(oracle):
    r = oracle.fasta_parse(b">a \nAC\n")
    assert rows(r) == [[b"a", b"", b"AC"]]


def test_fasta_crlf(oracle):
    r = oracle.fasta_parse(b">a d\r\nAC\r\nGT\r\n")
    assert rows(r) == [[b"a", b"d", b"ACGT"]]


def test_fasta_empty_sequence_is_allowed(oracle):
    r = oracle.fasta_parse(b">a\n>b\nAC\n")
    assert rows(r) == [[b"a", None, b""], [b"b", None, b"AC"]]


def test_fasta_errors(oracle):
    assert oracle.fasta_parse(b"ACGT\n>a\nAC\n").error_code == PE["FA_PREFIX"]
    assert oracle.fasta_parse(b"\n>a\nAC\n").error_code == PE["FA_EMPTY"]
    r = oracle.fasta_parse(b">a\nAC\n> desc only\nAC\n")
    assert r.n_rows == 1 and r.error_code == PE["FA_NAME"] and r.error_record == 1
    assert oracle.fasta_parse(b">a \xff\nAC\n").error_code == PE["UTF8"]
    assert oracle.fasta_parse(b"").n_rows == 0


def test_fasta_gt_inside_a_sequence_line_is_data(oracle):
    r = oracle.fasta_parse(b">a\nAC>GT\n")
    assert rows(r) == [[b"a", None, b"AC>GT"]]


# ---- VCF (noodles-vcf 0.34.0, tokenising level) ------------------------------------------------------

HDR = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"


def test_vcf_eight_columns_and_rest(oracle):
    r = oracle.vcf_parse(HDR + b"1\t5\t.\tA\tC\t.\tPASS\tDP=1\n2\t6\trs1\tA\tC,G\t1e2\tq10\t.\tGT\t0/1\t1/1\n")
    assert r.error_code == 0 and r.n_rows == 2
    assert rows(r)[0] == [b"1", b"5", b".", b"A", b"C", b".", b"PASS", b"DP=1", None]
    assert rows(r)[1][-1] == b"GT\t0/1\t1/1"
    assert list(r.extra["pos"]) == [5, 6]
    assert list(r.extra["qual_valid"]) == [0, 1] and float(r.extra["qual"][1]) == 100.0


def test_vcf_errors(oracle):
    assert oracle.vcf_parse(b"1\t5\t.\tA\tC\t.\tPASS\tDP=1\n").error_code == PE["VCF_HDR"]
    assert oracle.vcf_parse(HDR + b"1\t5\t.\tA\tC\t.\tPASS\n").error_code == PE["VCF_FIELD"]
    assert oracle.vcf_parse(HDR + b"1\tx5\t.\tA\tC\t.\tPASS\t.\n").error_code == PE["VCF_POS"]
    assert oracle.vcf_parse(HDR + b"1\t5\t.\tA\tC\tabc\tPASS\t.\n").error_code == PE["VCF_QUAL"]
    r = oracle.vcf_parse(HDR + b"1\t5\t.\tA\tC\t.\tPASS\t.\n\n")
    assert r.n_rows == 1 and r.error_code == PE["VCF_FIELD"]   # blank data line


def test_vcf_qual_negative_is_an_error_but_negative_zero_and_nan_are_not(oracle):
    # noodles-vcf 0.34 record::QualityScore: f32::from_str, then TryFrom<f32> refuses n < 0.0
    line = lambda q: HDR + b"1\t5\t.\tA\tC\t" + q + b"\tPASS\t.\n"  # noqa: E731
    for bad in (b"-1", b"-0.5", b"-1e-45", b"-inf", b"-Infinity"):
        assert oracle.vcf_parse(line(bad)).error_code == PE["VCF_QUAL"], bad
    for ok in (b"-0", b"-0.0", b"-0e5", b"-1e-50", b"nan", b"-nan", b"NaN", b"inf", b"+Infinity", b"0", b"1e39"):
        r = oracle.vcf_parse(line(ok))
        assert r.error_code == 0 and r.n_rows == 1, ok


def test_vcf_float_literals_of_any_length_are_correctly_rounded(oracle):
    # f32::from_str is correctly rounded whatever the number of digits: checked against exact rational arithmetic
    import struct
    from fractions import Fraction

    def nearest_f32(lit):
        v = Fraction(lit)
        if v == 0:
            return 0.0
        import math
        e = math.floor(math.log2(v)) if v > 0 else 0
        while Fraction(2) ** e > v:
            e -= 1
        while Fraction(2) ** (e + 1) <= v:
            e += 1
        e = max(e, -126)
        q = v / Fraction(2) ** (e - 23)           # in units of the ulp
        n = q.numerator // q.denominator
        rem = q - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
            n += 1
        x = float(n) * 2.0 ** (e - 23)
        return struct.unpack("<f", struct.pack("<f", x))[0] if x < 3.5e38 else float("inf")

    for lit in ("16777217", "16777217.00000000000000000000001", "0.1000000014901161193847656250000000000000000000000000001",
                "1.00000005960464477539062500000000000000001", "1.000000059604644775390625", "9007199254740993", "1e-45", "7e-46",
                "7.1e-46", "3.4028235677973366e38", "1." + "7" * 600, "2.7182818284590452353602874713527"):
        got = oracle.parse_f32_text(lit.encode())
        assert got == nearest_f32(lit), lit


def test_vcf_last_line_without_newline_and_crlf(oracle):
    r = oracle.vcf_parse(HDR + b"1\t5\t.\tA\tC\t3.5\tPASS\tX\r\n1\t7\t.\tA\tC\t.\t.\tY")
    assert r.n_rows == 2 and rows(r)[0][7] == b"X" and rows(r)[1][7] == b"Y"


# ---- from_utf8 ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("b,ok", [
    (b"plain", True), ("é€😀".encode(), True), (b"\xc0\xaf", False), (b"\xed\xa0\x80", False),
    (b"\xf4\x90\x80\x80", False), (b"\xe2\x82", False), (b"\x80", False), (b"\xf0\x9f\x98\x80", True),
])
def test_utf8_acceptance(oracle, b, ok):
    assert oracle.is_valid_utf8(b) == ok
    try:
        b.decode("utf-8")
        py_ok = True
    except UnicodeDecodeError:
        py_ok = False
    assert py_ok == ok


# ---- nested VCF columns: every [RECALLED] rule of oracle.vcf_typed_rows by name -----------------

VCF_HDR = (b"##fileformat=VCFv4.2\n"
           b"##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
           b"##INFO=<ID=AF,Number=A,Type=Float,Description=\"a, with a comma and \\\"quotes\\\"\">\n"
           b"##INFO=<ID=DB,Number=0,Type=Flag,Description=\"f\">\n"
           b"##INFO=<ID=ANN,Number=.,Type=String,Description=\"s\">\n"
           b"##INFO=<ID=DP,Number=1,Type=Float,Description=\"a second definition of DP is ignored\">\n"
           b"##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n"
           b"##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"r\">\n"
           b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\n")


def _typed(oracle, *lines):
    return oracle.vcf_typed_rows(VCF_HDR + b"\n".join(lines) + b"\n")


def test_vcf_typed_pinned_row(oracle, golden_dir):
    # test_vcf_record_scan.test:10-19 — chrom 1, pos 9999919, ref G, alt [<*>], qual 0.0, info.indel NULL, info.dp 1
    import os
    rows, err = oracle.vcf_typed_rows(open(os.path.join(golden_dir, "vcf/index.vcf"), "rb").read())
    assert err is None and len(rows) == 621
    r = rows[0]
    assert (r["chrom"], r["pos"], r["ref"], r["alt"], r["qual"], r["info"]["INDEL"], r["info"]["DP"]) == \
           ("1", 9999919, "G", ["<*>"], 0.0, None, 1)


def test_vcf_typed_header_keys_in_header_order_first_definition_wins(oracle):
    info, fmt = oracle.vcf_header_keys(VCF_HDR)
    assert info == [("DP", "Integer", False), ("AF", "Float", True), ("DB", "Flag", False), ("ANN", "String", True)]
    assert fmt == [("GT", "String", False), ("AD", "Integer", True)]


def test_vcf_typed_strings_are_percent_decoded(oracle):
    # noodles-vcf 0.34 (rust/Cargo.lock:2193-2194): String / Character values of INFO and of the samples go through
    # percent_encoding::percent_decode(..).decode_utf8(); ids, alts, filters and keys do not
    assert oracle.percent_decode(b"a%3Bb%2c%25") == b"a;b,%" and oracle.percent_decode(b"%zz%4%") == b"%zz%4%"
    hdr = (b"##fileformat=VCFv4.2\n##INFO=<ID=S,Number=.,Type=String,Description=\"s\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n"
           b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tX\n")
    rows, err = oracle.vcf_typed_rows(hdr + b"1\t5\ti%3B\tA\tC\t.\tf%3B\tS=a%3Bb,%e2%82%ac\tGT\t0%2F1\n")
    assert err is None and rows[0]["info"]["S"] == ["a;b", "\u20ac"] and rows[0]["formats"][0]["GT"] == "0/1"
    assert rows[0]["id"] == ["i%3B"] and rows[0]["filter"] == ["f%3B"]
    assert oracle.vcf_typed_rows(hdr + b"1\t5\t.\tA\tC\t.\t.\tS=%ff\tGT\t0\n")[1] == 0      # not UTF-8 after decoding: a value error


def test_vcf_typed_lists_split_and_missing_is_empty(oracle):
    rows, _ = _typed(oracle, b"1\t5\ta;b\tA\tC,<DEL>\t1\tq10;s50\t.", b"1\t6\t.\tA\t.\t1\t.\t.", b"1\t7\tx\tA\tT\t1\tPASS\t.")
    assert rows[0]["id"] == ["a", "b"] and rows[0]["alt"] == ["C", "<DEL>"] and rows[0]["filter"] == ["q10", "s50"]
    assert rows[1]["id"] == [] and rows[1]["alt"] == [] and rows[1]["filter"] == []
    assert rows[2]["filter"] == ["PASS"]


def test_vcf_typed_info_rules(oracle):
    rows, err = _typed(oracle, b"1\t5\t.\tA\tC\t1\t.\tDP=7;AF=0.5,.;DB;ANN=x,y;ZZ=1;DP=9", b"1\t6\t.\tA\tC\t1\t.\tDP=.;AF=.;ANN=",
                       b"1\t7\t.\tA\tC\t1\t.\t.")
    assert err is None
    assert rows[0]["info"] == {"DP": 7, "AF": [0.5, None], "DB": True, "ANN": ["x", "y"]}   # undeclared ZZ, repeated DP dropped
    assert rows[1]["info"] == {"DP": None, "AF": None, "DB": None, "ANN": [""]}
    assert rows[2]["info"] == {"DP": None, "AF": None, "DB": None, "ANN": None}


def test_vcf_typed_formats_rules(oracle):
    rows, err = _typed(oracle, b"1\t5\t.\tA\tC\t1\t.\t.\tGT:AD:XX\t0/1:3,.:q\t.", b"1\t6\t.\tA\tC\t1\t.\t.\tAD\t1,2\t3",
                       b"1\t7\t.\tA\tC\t1\t.\t.")
    assert err is None
    assert rows[0]["formats"] == [{"GT": "0/1", "AD": [3, None]}, {"GT": None, "AD": None}]
    assert rows[1]["formats"] == [{"GT": None, "AD": [1, 2]}, {"GT": None, "AD": [3]}]
    assert rows[2]["formats"] == []


def test_vcf_typed_bad_number_is_a_record_error(oracle):
    rows, err = _typed(oracle, b"1\t5\t.\tA\tC\t1\t.\tDP=3", b"1\t6\t.\tA\tC\t1\t.\tDP=3.5")
    assert err == 1 and len(rows) == 1
    rows, err = _typed(oracle, b"1\t5\t.\tA\tC\t1\t.\t.\tAD\tx")
    assert err == 0 and rows == []
    assert oracle.parse_i32_text(b"2147483647") == 2147483647 and oracle.parse_i32_text(b"2147483648") is None
    assert oracle.parse_i32_text(b"-2147483648") == -2147483648 and oracle.parse_i32_text(b"+") is None


# ---- quality_score_string_to_list (fastq_functions/module.cpp:28-54) — parity unpinned: no sqllogictest calls it ----
def test_quality_score_list_is_byte_minus_33_signed(oracle):
    class Col:
        offsets = np.array([0, 3, 3, 5], np.int64)
        values = np.frombuffer(b"!I@\x80\xff", np.uint8)
        valid = None

    entries, values = oracle.quality_score_string_to_list(Col)
    assert entries.tolist() == [[0, 3], [3, 0], [3, 2]]
    assert values.tolist() == [0, 40, 31, -128 - 33, -1 - 33]


def test_quality_score_list_null_rows_are_empty(oracle):
    class Col:
        offsets = np.array([0, 2, 4], np.int64)
        values = np.frombuffer(b"!!II", np.uint8)
        valid = np.array([0, 1], np.uint8)

    entries, values = oracle.quality_score_string_to_list(Col)
    assert entries.tolist() == [[0, 0], [0, 2]]
    assert values.tolist() == [40, 40]


def test_quality_score_list_golden_record(oracle, golden_dir):
    exp = oracle.fastq_parse(open(f"{golden_dir}/test.fastq", "rb").read())
    entries, values = oracle.quality_score_string_to_list(exp.columns["quality_scores"])
    q0 = b"!''*((((***+))%%%++)(%%%%).1***-+*''))**55CCF>>>>>>CCCCCCC65"   # test_fastq_scan.test:35-41
    assert entries[0].tolist() == [0, len(q0)]
    assert values[: len(q0)].tolist() == [c - 33 for c in q0]


# ---- decoder-level rules (oracle/pyoracle.py decode_by_rule): what the ROWS of a compressed input are -------------------------------
# [RECALLED / open]: the reference decodes through DataFusion 28 -> async-compression 0.4.0 (rust/src/arrow_reader.rs:60-91); whether
# its GzipDecoder reads past the first member decides the first two rules.  tools/falsify_kit.py writes each
# input as a file with these rows beside it.
FQ_A = b"@a x\nAC\n+\n!!\n@b\nGT\n+\n##\n"
FQ_B = b"@c y z\nTTT\n+\nIII\n"
FA_A, FA_B = b">s1 d\nACGT\nAC\n", b">s2\nGG\n"


def _bgzf(data, block, eof=True):
    import struct
    import zlib
    out = []
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(d) + 8 - 1)
                   + d + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out) + (bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000") if eof else b"")


def _crows(r):
    return rows(r.table)


def test_gzip_every_member_of_a_concatenation_is_read(oracle):
    import gzip
    r = oracle.compressed_parse("fastq", gzip.compress(FQ_A, 6, mtime=0) + gzip.compress(FQ_B, 6, mtime=0), "gzip", "fastq.gz")
    assert r.error is None and _crows(r) == [[b"a", b"x", b"AC", b"!!"], [b"b", None, b"GT", b"##"], [b"c", b"y z", b"TTT", b"III"]]
    r = oracle.compressed_parse("fasta", gzip.compress(FA_A, 6, mtime=0) + gzip.compress(FA_B, 6, mtime=0), "gzip", "fasta.gz")
    assert r.error is None and _crows(r) == [[b"s1", b"d", b"ACGTAC"], [b"s2", None, b"GG"]]


def test_gzip_a_record_may_span_members(oracle):
    import gzip
    cut = FQ_A.index(b"@b") + 4   # inside record b
    r = oracle.compressed_parse("fastq", gzip.compress(FQ_A[:cut], 6, mtime=0) + gzip.compress(FQ_A[cut:] + FQ_B, 6, mtime=0), "gzip", "fastq.gz")
    assert r.error is None and len(_crows(r)) == 3


def test_bgzf_is_read_member_by_member_with_or_without_its_eof_block(oracle):
    for eof in (True, False):
        r = oracle.compressed_parse("fastq", _bgzf(FQ_A + FQ_B, 11, eof), "gzip", "fastq.gz")
        assert r.error is None and len(_crows(r)) == 3
        r = oracle.compressed_parse("fasta", _bgzf(FA_A + FA_B, 7, eof), "gzip", "fasta.gz")
        assert r.error is None and _crows(r) == [[b"s1", b"d", b"ACGTAC"], [b"s2", None, b"GG"]]


def test_gzip_bytes_behind_the_last_member_are_an_error(oracle):
    import gzip
    for tail in (b"\n", b"\0" * 4, b"not gzip"):
        r = oracle.compressed_parse("fastq", gzip.compress(FQ_A, 6, mtime=0) + tail, "gzip", "fastq.gz")
        assert r.error == "invalid gzip header" and len(_crows(r)) == 2   # (the rows in front come first; through SQL the query fails)


def test_gzip_truncated_member_and_wrong_trailer_are_errors(oracle):
    import gzip
    z = gzip.compress(FQ_A + FQ_B, 6, mtime=0)
    assert oracle.compressed_parse("fastq", z[:-5], "gzip", "fastq.gz").error == "truncated gzip member"
    bad = bytearray(z)
    bad[-8] ^= 1   # CRC-32
    assert "corrupt gzip stream" in oracle.compressed_parse("fastq", bytes(bad), "gzip", "fastq.gz").error
    bad = bytearray(z)
    bad[-4] ^= 1   # ISIZE
    assert "corrupt gzip stream" in oracle.compressed_parse("fastq", bytes(bad), "gzip", "fastq.gz").error


def test_gzip_option_on_plain_text_is_an_error(oracle):
    # read_fastq('x.fastq', compression = 'gzip'): the option wins over the extension (rust/src/arrow_reader.rs:60-75)
    r = oracle.compressed_parse("fastq", FQ_A, "gzip", "fastq")
    assert r.error == "invalid gzip header" and _crows(r) == []


def test_empty_gzip_file_and_empty_member(oracle):
    import gzip
    assert oracle.compressed_parse("fastq", b"", "gzip", "fastq.gz").error == "empty gzip file"   # (zero bytes: no member at all)
    r = oracle.compressed_parse("fastq", gzip.compress(b"", 6, mtime=0) + gzip.compress(FQ_B, 6, mtime=0) + gzip.compress(b"", 6, mtime=0), "gzip", "fastq.gz")
    assert r.error is None and _crows(r) == [[b"c", b"y z", b"TTT", b"III"]]


def test_zstd_concatenated_and_skippable_frames_are_read_through(oracle):
    from zstd_util import compress, skippable
    z = compress(FQ_A, 3, True) + skippable(b"index", 3) + compress(FQ_B, 19, False, content_size=False) + skippable(b"")
    r = oracle.compressed_parse("fastq", z, "zstd", "fastq.zst")
    assert r.error is None and len(_crows(r)) == 3
    cut = 9
    r = oracle.compressed_parse("fasta", compress((FA_A + FA_B)[:cut], 1) + compress((FA_A + FA_B)[cut:], 1), "zstd", "fasta.zst")
    assert r.error is None and _crows(r) == [[b"s1", b"d", b"ACGTAC"], [b"s2", None, b"GG"]]


def test_zstd_bytes_that_begin_no_frame_and_truncated_frames_are_errors(oracle):
    from zstd_util import compress
    z = compress(FQ_A + FQ_B, 3, True)
    assert oracle.compressed_parse("fastq", z + b"tail", "zstd", "fastq.zst").error is not None
    assert oracle.compressed_parse("fastq", z[:-2], "zstd", "fastq.zst").error is not None
    assert oracle.compressed_parse("fastq", FQ_A, "zstd", "fastq").error is not None   # compression = 'zstd' on plain text


def test_zstd_content_checksum_is_verified(oracle):
    from zstd_util import compress
    bad = bytearray(compress(FQ_A + FQ_B, 3, True))
    bad[-1] ^= 0x40
    assert oracle.compressed_parse("fastq", bytes(bad), "zstd", "fastq.zst").error is not None


# ---- schema rules (oracle/pyoracle.py schema_of): names and DuckDB types — DESCRIBE SELECT * FROM read_*(...) ------------------------
def test_schema_of_fastq_and_fasta_is_all_varchar(oracle):
    assert oracle.schema_of("fastq", FQ_A) == [("name", "VARCHAR"), ("description", "VARCHAR"), ("sequence", "VARCHAR"), ("quality_scores", "VARCHAR")]
    assert oracle.schema_of("fasta", FA_A) == [("id", "VARCHAR"), ("description", "VARCHAR"), ("sequence", "VARCHAR")]


VCF_TYPES = (b"##fileformat=VCFv4.2\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"a\">\n"
             b"##INFO=<ID=DB,Number=0,Type=Flag,Description=\"f\">\n##INFO=<ID=ANN,Number=.,Type=String,Description=\"s\">\n"
             b"##INFO=<ID=CH,Number=1,Type=Character,Description=\"c\">\n"
             b"##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"r\">\n"
             b"##FORMAT=<ID=GL,Number=G,Type=Float,Description=\"l\">\n"
             b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n1\t5\trs1;rs2\tA\tC,G\t3.5\tq10;s50\tDP=3;AF=0.5,0.25;DB;CH=x\tGT:AD\t0/1:3,4,5\n")


def test_schema_of_vcf_follows_the_header(oracle):
    assert oracle.schema_of("vcf", VCF_TYPES) == [
        ("chrom", "VARCHAR"), ("pos", "BIGINT"), ("id", "VARCHAR[]"), ("ref", "VARCHAR"), ("alt", "VARCHAR[]"), ("qual", "FLOAT"), ("filter", "VARCHAR[]"),
        ("info", "STRUCT(DP INTEGER, AF FLOAT[], DB BOOLEAN, ANN VARCHAR[], CH VARCHAR)"),
        ("formats", "STRUCT(GT VARCHAR, AD INTEGER[], GL FLOAT[])[]")]
    rows_, err = oracle.vcf_typed_rows(VCF_TYPES)
    assert err is None and rows_[0]["id"] == ["rs1", "rs2"] and rows_[0]["alt"] == ["C", "G"] and rows_[0]["filter"] == ["q10", "s50"]
    assert rows_[0]["info"] == {"DP": 3, "AF": [0.5, 0.25], "DB": True, "ANN": None, "CH": "x"}
    assert rows_[0]["formats"] == [{"GT": "0/1", "AD": [3, 4, 5], "GL": None}]
