"""Host-side generators of test / bench inputs of other shapes than the synthetic 150 bp FASTQ / 49-byte VCF line (TEST AND BENCH
SCAFFOLDING, like csrc/testing/): long reads, short reads, multi-sample VCF lines.  Seeded numpy; the parity tests
(tests/test_record_shapes_gpu.py) and bench.py's `record_shapes` legs use the same functions, the oracle checks what they
make."""
import numpy as np

VCF_HDR = b"##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"


def fastq_records(lengths, seed=1, crlf_every=0, desc_every=2, name_len=None):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for k, ln in enumerate(lengths):
        ln = int(ln)
        eol = b"\r\n" if crlf_every and k % crlf_every == 0 else b"\n"
        name = b"@r%d" % k if name_len is None else b"@" + (b"%d" % k).rjust(name_len, b"n")
        if desc_every and k % desc_every == 0:
            name += b" len=%d ch=%d" % (ln, k % 512)
        seq = rng.choice(acgt, ln).tobytes()
        qual = (rng.integers(33, 74, ln, dtype=np.uint8)).tobytes()   # includes '@' and '+'
        out.append(name + eol + seq + eol + b"+" + eol + qual + eol)
    return b"".join(out)


def vcf_lines(n_lines, n_samples, seed=1, crlf_every=0):
    rng = np.random.default_rng(seed)
    gts = [b"0|0", b"0|1", b"1|0", b"1|1", b".|."]
    smp = b"\t".join(b"S%05d" % i for i in range(n_samples))
    hdr = VCF_HDR[:-1] + (b"\tFORMAT\t" + smp if n_samples else b"") + b"\n"
    out = [hdr]
    for k in range(n_lines):
        eol = b"\r\n" if crlf_every and k % crlf_every == 0 else b"\n"
        qual = b"." if k % 9 == 0 else b"%d.%d" % (k % 1000, k % 10)
        info = b"AC=%d;AF=0.%04d;AN=%d;NS=%d;DP=%d;VT=SNP" % (k % 5008, k % 10000, 2 * n_samples, n_samples, 1000 + k)
        line = b"%d\t%d\trs%d\t%s\t%s\t%s\tPASS\t%s" % (k % 22 + 1, 10000 + 37 * k, k, b"ACGT"[k % 4:k % 4 + 1],
                                                           b"ACGT"[(k + 1) % 4:(k + 1) % 4 + 1], qual, info)
        if n_samples:
            idx = rng.integers(0, 5, n_samples)
            line += b"\tGT\t" + b"\t".join(gts[i] for i in idx)
        out.append(line + eol)
    return b"".join(out)




def hifi_lengths(n, seed=11):
    """PacBio HiFi-like: ~15 kb +- 3 kb"""
    return np.clip(np.random.default_rng(seed).normal(15000, 3000, n), 2000, 40000)


def ont_lengths(n, seed=11):
    """ONT-like: log-uniform 1 kb .. 100 kb"""
    return np.exp(np.random.default_rng(seed).uniform(np.log(1000), np.log(100000), n))


# ---- bench-sized blocks (tens of MB in about a second): vectorised, every record whole.  Each returns what its rows must be —
# field offsets and lengths known to the generator, not taken from any parser — so that bench.py can check a scan's output
# without the oracle (which only tests/, smoke() and the cpu_baseline leg may use); the bit-exact parity of these shapes against
# the oracle is tests/test_record_shapes_gpu.py's -------------------------------------------------------------------------------

def fastq_fixed_block(n_records, read_len, name_digits=9, seed=7):
    """n_records records of one shape — `@` + name_digits decimal digits, read_len bases, `+`, read_len qualities — as one byte
    string (36 bp short reads: 700 k records = 61 MB in ~1 s) -> (bytes, expect): expect[col] = (offsets int64[n], lengths
    int64[n]) for name / sequence / quality_scores; the description is NULL in every row"""
    rng = np.random.default_rng(seed)
    rec = 1 + name_digits + 1 + read_len + 1 + 2 + read_len + 1
    a = np.empty((n_records, rec), np.uint8)
    a[:, 0] = ord("@")
    k = np.arange(n_records, dtype=np.int64)
    for d in range(name_digits):
        a[:, name_digits - d] = ord("0") + (k // 10 ** d) % 10
    o = 1 + name_digits
    a[:, o] = 10
    a[:, o + 1:o + 1 + read_len] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n_records, read_len), dtype=np.uint8)]
    seq_o = o + 1
    o += 1 + read_len
    a[:, o] = 10
    a[:, o + 1] = ord("+")
    a[:, o + 2] = 10
    a[:, o + 3:o + 3 + read_len] = rng.integers(33, 74, (n_records, read_len), dtype=np.uint8)
    qual_o = o + 3
    a[:, o + 3 + read_len] = 10
    base = k * rec
    expect = {"name": (base + 1, np.full(n_records, name_digits, np.int64)), "description": None,
              "sequence": (base + seq_o, np.full(n_records, read_len, np.int64)),
              "quality_scores": (base + qual_o, np.full(n_records, read_len, np.int64))}
    return a.tobytes(), expect


def fastq_long_block(lengths, seed=7):
    """long reads (a few thousand records: the per-record loop is fine), names like an instrument's -> (bytes, expect) like
    fastq_fixed_block; every row has a description"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    out = []
    cols = {c: ([], []) for c in ("name", "description", "sequence", "quality_scores")}
    at = 0
    for k, ln in enumerate(lengths):
        ln = int(ln)
        name = b"m64011_190830_220126/%d/ccs" % (k * 73 + 1)
        desc = b"np=%d rq=0.99%d" % (3 + k % 17, k % 10)
        out.append(b"@" + name + b" " + desc + b"\n")
        out.append(acgt[rng.integers(0, 4, ln, dtype=np.uint8)].tobytes())
        out.append(b"\n+\n")
        out.append(rng.integers(33, 127, ln, dtype=np.uint8).tobytes())
        out.append(b"\n")
        for c, off, n in (("name", at + 1, len(name)), ("description", at + 2 + len(name), len(desc)),
                          ("sequence", at + 3 + len(name) + len(desc), ln), ("quality_scores", at + 3 + len(name) + len(desc) + ln + 3, ln)):
            cols[c][0].append(off)
            cols[c][1].append(n)
        at += 3 + len(name) + len(desc) + 2 * ln + 4
    return b"".join(out), {c: (np.asarray(o, np.int64), np.asarray(n, np.int64)) for c, (o, n) in cols.items()}


def vcf_cohort_header(n_samples):
    """the header vcf_multisample_block's lines need for the reference's real schema: its INFO keys typed (AC / AF per-allele lists,
    like 1000 Genomes declares them) and FORMAT GT, so that `info` is a STRUCT of six children and `formats` a LIST(STRUCT(GT VARCHAR))"""
    smp = b"\t".join(b"S%05d" % i for i in range(n_samples))
    return (b"##fileformat=VCFv4.2\n"
            b'##INFO=<ID=AC,Number=A,Type=Integer,Description="Allele count">\n##INFO=<ID=AF,Number=A,Type=Float,Description="Allele frequency">\n'
            b'##INFO=<ID=AN,Number=1,Type=Integer,Description="Alleles">\n##INFO=<ID=NS,Number=1,Type=Integer,Description="Samples">\n'
            b'##INFO=<ID=DP,Number=1,Type=Integer,Description="Depth">\n##INFO=<ID=VT,Number=.,Type=String,Description="Variant type">\n'
            b'##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
            b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + smp + b"\n")


def vcf_multisample_block(n_lines, n_samples, seed=7):
    """-> (header bytes, data-line bytes, expect): 1000-Genomes-like lines, GT per sample ("0|0\\t" ...), vectorised genotypes.
    expect: line starts, CHROM values, POS, QUAL validity, and where each line's FORMAT + samples remainder lies"""
    rng = np.random.default_rng(seed)
    gts = np.frombuffer(b"0|0\t0|1\t1|0\t1|1\t.|.\t", np.uint8).reshape(5, 4)
    smp = b"\t".join(b"S%05d" % i for i in range(n_samples))
    hdr = VCF_HDR[:-1] + b"\tFORMAT\t" + smp + b"\n"
    g = gts[rng.choice(5, size=(n_lines, n_samples), p=[0.7, 0.1, 0.1, 0.08, 0.02])].reshape(n_lines, n_samples * 4)
    g[:, -1] = 10   # the last sample's tab is the line's newline
    out = []
    start, chrom, pos, qual_valid, fmt_off, fmt_len = [], [], [], [], [], []
    at = 0
    for k in range(n_lines):
        qual = b"." if k % 9 == 0 else b"%d.%d" % (k % 1000, k % 10)
        info = b"AC=%d;AF=0.%04d;AN=%d;NS=%d;DP=%d;VT=SNP" % (k % 5008, k % 10000, 2 * n_samples, n_samples, 1000 + k)
        head = b"%d\t%d\trs%d\t%s\t%s\t%s\tPASS\t%s\t" % (k % 22 + 1, 10000 + 37 * k, k, b"ACGT"[k % 4:k % 4 + 1],
                                                                  b"ACGT"[(k + 1) % 4:(k + 1) % 4 + 1], qual, info)
        out.append(head + b"GT\t")
        out.append(g[k].tobytes())
        start.append(at)
        chrom.append(k % 22 + 1)
        pos.append(10000 + 37 * k)
        qual_valid.append(k % 9 != 0)
        fmt_off.append(at + len(head))
        fmt_len.append(3 + n_samples * 4 - 1)
        at += len(head) + 3 + n_samples * 4
    expect = {"start": np.asarray(start, np.int64), "chrom": np.asarray(chrom, np.int64), "pos": np.asarray(pos, np.int64),
              "qual_valid": np.asarray(qual_valid, bool), "formats": (np.asarray(fmt_off, np.int64), np.asarray(fmt_len, np.int64))}
    return hdr, b"".join(out), expect
