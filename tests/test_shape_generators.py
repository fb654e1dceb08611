"""The bench's record-shape generators (exon_duckdb_amd/testing/shapes.py) say what the rows of their blocks must be — field
offsets and lengths — so that bench.py can verify a scan without the oracle.  Here the oracle checks the generators."""
import numpy as np

from exon_duckdb_amd.testing import shapes


def test_fastq_blocks_say_what_the_oracle_parses(oracle):
    for data, expect in (shapes.fastq_fixed_block(5000, 36), shapes.fastq_fixed_block(300, 1, name_digits=3),
                         shapes.fastq_long_block(shapes.ont_lengths(60)), shapes.fastq_long_block(shapes.hifi_lengths(40))):
        t = oracle.fastq_parse(data, want_string_t=False)
        assert t.error_code == 0
        for c in ("name", "description", "sequence", "quality_scores"):
            col = t.columns[c]
            if expect[c] is None:
                assert not col.valid.any()
                continue
            off, ln = expect[c]
            assert t.n_rows == len(off) and col.valid.all()
            assert np.array_equal(col.src_off, off) and np.array_equal(col.lengths(), ln), c


def test_vcf_blocks_say_what_the_oracle_parses(oracle):
    for n_lines, n_samples in ((300, 100), (40, 2504)):
        hdr, lines, e = shapes.vcf_multisample_block(n_lines, n_samples)
        t = oracle.vcf_parse(hdr + lines, want_string_t=False)
        assert t.error_code == 0 and t.n_rows == n_lines
        assert np.array_equal(t.extra["pos"], e["pos"]) and np.array_equal(t.extra["qual_valid"].astype(bool), e["qual_valid"])
        assert np.array_equal(t.columns["formats"].src_off, e["formats"][0] + len(hdr))
        assert np.array_equal(t.columns["formats"].lengths(), e["formats"][1])
        assert np.array_equal(t.columns["chrom"].src_off, e["start"] + len(hdr))
        assert [int(x) for x in t.columns["chrom"].to_list()] == list(e["chrom"])
