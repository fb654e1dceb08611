"""ctypes front-end of oracle/libexon_oracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libexon_oracle.so")


class Utf8Col(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64),
        ("offsets", C.POINTER(C.c_int64)),
        ("values", C.POINTER(C.c_uint8)),
        ("valid", C.POINTER(C.c_uint8)),
        ("src_off", C.POINTER(C.c_int64)),
        ("cap_rows", C.c_int64),
        ("cap_values", C.c_int64),
    ]


class I64Col(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("data", C.POINTER(C.c_int64)), ("valid", C.POINTER(C.c_uint8)),
                ("cap_rows", C.c_int64)]


class F32Col(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("data", C.POINTER(C.c_float)), ("valid", C.POINTER(C.c_uint8)),
                ("cap_rows", C.c_int64)]


class Error(C.Structure):
    _fields_ = [("code", C.c_uint32), ("record", C.c_uint64), ("offset", C.c_uint64), ("message", C.c_char * 128)]


class FastqTable(C.Structure):
    _fields_ = [("name", Utf8Col), ("description", Utf8Col), ("sequence", Utf8Col), ("quality_scores", Utf8Col),
                ("err", Error)]


class FastaTable(C.Structure):
    _fields_ = [("id", Utf8Col), ("description", Utf8Col), ("sequence", Utf8Col), ("err", Error)]


class VcfTable(C.Structure):
    _fields_ = [("fields", Utf8Col * 9), ("pos", I64Col), ("qual", F32Col), ("header_bytes", C.c_int64),
                ("n_header_lines", C.c_int64), ("err", Error)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.orc_fastq_parse.restype = C.c_int64
        l.orc_fastq_parse.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(FastqTable)]
        l.orc_fasta_parse.restype = C.c_int64
        l.orc_fasta_parse.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(FastaTable)]
        l.orc_vcf_parse.restype = C.c_int64
        l.orc_vcf_parse.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(VcfTable)]
        l.orc_fastq_free.argtypes = [C.POINTER(FastqTable)]
        l.orc_fasta_free.argtypes = [C.POINTER(FastaTable)]
        l.orc_vcf_free.argtypes = [C.POINTER(VcfTable)]
        l.orc_utf8_to_string_t.argtypes = [C.POINTER(Utf8Col), C.c_int64, C.c_int64, C.c_int, C.c_uint64,
                                           C.c_void_p, C.c_void_p]
        l.orc_parse_f32_text.restype = C.c_int
        l.orc_parse_f32_text.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_float)]
        l.orc_quality_score_list.restype = None
        l.orc_quality_score_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        l.orc_parse_i32_text.restype = C.c_int
        l.orc_parse_i32_text.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_int32)]
        l.orc_is_valid_utf8.restype = C.c_int
        l.orc_is_valid_utf8.argtypes = [C.c_void_p, C.c_uint64]
        l.orc_synth_fastq.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        for f in ("orc_synth_fastq_ragged", "orc_synth_vcf", "orc_synth_fasta"):
            getattr(l, f).restype = C.c_uint64
            getattr(l, f).argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        l.orc_fastq_scan_baseline.restype = C.c_int64
        l.orc_fastq_scan_baseline.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        l.orc_fastq_scan_baseline_mt.restype = C.c_int64
        l.orc_fastq_scan_baseline_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        l.orc_infer_compression.restype = C.c_char_p
        l.orc_infer_compression.argtypes = [C.c_char_p, C.c_char_p]
        l.orc_replacement_scan.restype = C.c_char_p
        l.orc_replacement_scan.argtypes = [C.c_char_p]
        _lib = l
    return _lib


SYNTH_FASTQ_SEED = 0xE0A5EED0001
SYNTH_VCF_SEED = 0xE0A5EED0002
SYNTH_FASTA_SEED = 0xE0A5EED0003


class Column:
    """Host copy of an Arrow-style Utf8 column: offsets / values / valid / src_off."""

    def __init__(self, c: Utf8Col, n_rows=None):
        n = int(c.n_rows if n_rows is None else min(n_rows, c.n_rows))
        self.n = n
        if n:
            self.offsets = np.ctypeslib.as_array(c.offsets, shape=(n + 1,)).copy()
            nb = int(self.offsets[n])
            self.values = np.ctypeslib.as_array(c.values, shape=(max(nb, 1),))[:nb].copy() if nb else np.zeros(0, np.uint8)
            self.valid = np.ctypeslib.as_array(c.valid, shape=(n,)).copy()
            self.src_off = np.ctypeslib.as_array(c.src_off, shape=(n,)).copy()
        else:
            self.offsets = np.zeros(1, np.int64)
            self.values = np.zeros(0, np.uint8)
            self.valid = np.zeros(0, np.uint8)
            self.src_off = np.zeros(0, np.int64)

    def row(self, i):
        if not self.valid[i]:
            return None
        return self.values[self.offsets[i]:self.offsets[i + 1]].tobytes()

    def lengths(self):
        return np.diff(self.offsets)

    def to_list(self):
        return [self.row(i) for i in range(self.n)]


def _string_t(c: Utf8Col, n, mode, payload_base):
    out = np.zeros((max(n, 1), 16), np.uint8)
    words = np.zeros(((max(n, 1) + 63) // 64,), np.uint64)
    if n:
        lib().orc_utf8_to_string_t(C.byref(c), 0, n, mode, payload_base, out.ctypes.data, words.ctypes.data)
    return out[:n], words[: (n + 63) // 64]


class ParseResult:
    def __init__(self, n_rows, columns, err, string_t=None, extra=None):
        self.n_rows = n_rows
        self.columns = columns      # dict name -> Column
        self.error_code = int(err.code)
        self.error_record = int(err.record)
        self.error_offset = int(err.offset)
        self.error_message = err.message.decode()
        self.string_t = string_t or {}   # dict name -> ([n,16] u8, validity words)  (canonical zero-copy view)
        self.extra = extra or {}


def _as_buf(data):
    if isinstance(data, np.ndarray):
        arr = np.ascontiguousarray(data, dtype=np.uint8)
    else:
        arr = np.frombuffer(bytes(data), dtype=np.uint8)
    return arr, arr.ctypes.data if arr.size else None


def fastq_parse(data, payload_base=0, want_string_t=True):
    arr, ptr = _as_buf(data)
    t = FastqTable()
    n = int(lib().orc_fastq_parse(ptr, arr.size, C.byref(t)))
    names = ["name", "description", "sequence", "quality_scores"]
    cols = {k: Column(getattr(t, k), n) for k in names}
    st = {k: _string_t(getattr(t, k), n, 1, payload_base) for k in names} if want_string_t else {}
    res = ParseResult(n, cols, t.err, st)
    lib().orc_fastq_free(C.byref(t))
    return res


def fasta_parse(data):
    arr, ptr = _as_buf(data)
    t = FastaTable()
    n = int(lib().orc_fasta_parse(ptr, arr.size, C.byref(t)))
    names = ["id", "description", "sequence"]
    cols = {k: Column(getattr(t, k), n) for k in names}
    res = ParseResult(n, cols, t.err)
    lib().orc_fasta_free(C.byref(t))
    return res


VCF_FIELDS = ["chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "formats"]


def vcf_parse(data, payload_base=0, want_string_t=True):
    arr, ptr = _as_buf(data)
    t = VcfTable()
    n = int(lib().orc_vcf_parse(ptr, arr.size, C.byref(t)))
    cols = {VCF_FIELDS[k]: Column(t.fields[k], n) for k in range(9)}
    st = {VCF_FIELDS[k]: _string_t(t.fields[k], n, 1, payload_base) for k in range(9)} if want_string_t else {}
    extra = {
        "pos": np.ctypeslib.as_array(t.pos.data, shape=(max(n, 1),))[:n].copy() if n else np.zeros(0, np.int64),
        "qual": np.ctypeslib.as_array(t.qual.data, shape=(max(n, 1),))[:n].copy() if n else np.zeros(0, np.float32),
        "qual_valid": np.ctypeslib.as_array(t.qual.valid, shape=(max(n, 1),))[:n].copy() if n else np.zeros(0, np.uint8),
        "header_bytes": int(t.header_bytes),
        "n_header_lines": int(t.n_header_lines),
    }
    res = ParseResult(n, cols, t.err, st, extra)
    lib().orc_vcf_free(C.byref(t))
    return res


# ---- nested VCF columns -----------------------------------------------------------------
# Restates exon 0.2.6 datasources::vcf::{VCFSchemaBuilder, VCFArrayBuilder} over noodles-vcf 0.34.0 at the
# level the reference's Arrow schema shows it (parity unpinned beyond test_vcf_record_scan.test:10-19):
#   id / alt / filter: List<Utf8> ("." => []), info: Struct of the header's ##INFO keys, formats:
#   List<Struct of the ##FORMAT keys>, one entry per sample.  Rows come out as pyarrow's to_pylist() would
#   print them.  Pure Python over the C tokeniser above: small inputs only.
def parse_f32_text(b: bytes):
    v = C.c_float(0)
    return float(v.value) if lib().orc_parse_f32_text(b, len(b), C.byref(v)) else None


def parse_i32_text(b: bytes):
    v = C.c_int32(0)
    return int(v.value) if lib().orc_parse_i32_text(b, len(b), C.byref(v)) else None


def quality_score_string_to_list(col):
    """(entries u64[n, 2] = {offset, length}, values i32) of a Column (fastq_functions/module.cpp:28-54)."""
    n = len(col.offsets) - 1
    offsets = np.ascontiguousarray(col.offsets, np.int64)
    values = np.ascontiguousarray(col.values, np.uint8)
    valid = None if getattr(col, "valid", None) is None else np.ascontiguousarray(col.valid, np.uint8)
    entries = np.zeros((n, 2), np.uint64)
    out = np.zeros(max(int(offsets[n]), 1), np.int32)
    lib().orc_quality_score_list(values.ctypes.data, offsets.ctypes.data, valid.ctypes.data if valid is not None else None,
                                 n, entries.ctypes.data, out.ctypes.data)
    return entries, out[:int(entries[:, 1].sum())]


def vcf_header_keys(data: bytes):
    """([(id, type, is_list)] for ##INFO, same for ##FORMAT), header order, first definition of an ID wins."""
    out = {"INFO": [], "FORMAT": []}
    for raw in bytes(data).split(b"\n"):
        if not raw.startswith(b"#"):
            break
        line = raw.rstrip(b"\r").decode("utf-8", "replace")
        for kind in ("INFO", "FORMAT"):
            pre = "##" + kind + "=<"
            if not line.startswith(pre):
                continue
            body, fields, i = line[len(pre):], {}, 0
            while i < len(body) and body[i] != ">":
                eq = body.find("=", i)
                if eq < 0:
                    break
                key, j, val = body[i:eq], eq + 1, ""
                if j < len(body) and body[j] == '"':
                    j += 1
                    while j < len(body) and body[j] != '"':
                        if body[j] == "\\" and j + 1 < len(body):
                            j += 1
                        val += body[j]
                        j += 1
                    j += 1
                else:
                    while j < len(body) and body[j] not in ",>":
                        val += body[j]
                        j += 1
                fields.setdefault(key, val)
                i = j + 1 if j < len(body) and body[j] == "," else j
            if "ID" in fields and fields["ID"] and all(k[0] != fields["ID"] for k in out[kind]):
                ty = fields.get("Type", "String")
                ty = ty if ty in ("Integer", "Float", "Flag") else "String"
                out[kind].append((fields["ID"], ty, ty != "Flag" and fields.get("Number", "1") != "1"))
    return out["INFO"], out["FORMAT"]


class TypedValueError(ValueError):
    pass


def percent_decode(t: bytes) -> bytes:
    """percent_encoding::percent_decode (what noodles-vcf 0.34 applies to String / Character values of INFO and of the
    samples, rust/Cargo.lock:2193-2194): '%' followed by two hex digits becomes that byte; any other '%' stays."""
    out = bytearray()
    i = 0
    hexd = b"0123456789abcdefABCDEF"
    while i < len(t):
        if t[i] == 0x25 and i + 2 < len(t) and t[i + 1] in hexd and t[i + 2] in hexd:
            out.append(int(t[i + 1:i + 3], 16))
            i += 3
        else:
            out.append(t[i])
            i += 1
    return bytes(out)


def _typed(text: bytes, ty, is_list):
    """value text (after '=') -> python value; None for the missing value '.'"""
    def one(t):
        if ty == "Integer":
            v = parse_i32_text(t)
        elif ty == "Float":
            v = parse_f32_text(t)
        else:
            try:
                return percent_decode(t).decode("utf-8")   # .decode_utf8(): a decoded value that is not UTF-8 is an error
            except UnicodeDecodeError:
                raise TypedValueError(t)
        if v is None:
            raise TypedValueError(t)
        return v

    if text == b".":
        return None
    if not is_list:
        return one(text)
    return [None if t == b"." else one(t) for t in text.split(b",")]


def vcf_typed_rows(data):
    """-> (rows, error_row): rows as dicts in the reference's schema; error_row = index of the first row whose
    INFO / FORMAT values do not parse (rows stop there), else None."""
    data = bytes(data)
    base = vcf_parse(data, want_string_t=False)
    info_keys, format_keys = vcf_header_keys(data)
    cols = {k: base.columns[k].to_list() for k in VCF_FIELDS}
    rows = []
    for r in range(base.n_rows):
        def split(field, sep):
            t = cols[field][r]
            return [] if t in (b".", b"") else [x.decode("utf-8") for x in t.split(sep)]

        try:
            info = {k: None for k, _, _ in info_keys}
            raw = cols["info"][r]
            if raw not in (b".", b""):
                seen = set()
                for ent in raw.split(b";"):
                    key, eq, val = ent.partition(b"=")
                    name = key.decode("utf-8", "replace")
                    if not key or name in seen:
                        continue
                    seen.add(name)
                    for k, ty, is_list in info_keys:
                        if k == name:
                            if ty == "Flag":
                                info[k] = True
                            elif eq:
                                info[k] = _typed(val, ty, is_list)
            formats = []
            rest = cols["formats"][r]
            if rest is not None:
                parts = rest.split(b"\t")
                fkeys = [p.decode("utf-8", "replace") for p in parts[0].split(b":")]
                for sample in parts[1:]:
                    d = {k: None for k, _, _ in format_keys}
                    vals, done = sample.split(b":"), set()
                    for p, name in enumerate(fkeys):
                        if p >= len(vals) or name in done:
                            continue
                        for k, ty, is_list in format_keys:
                            if k == name:
                                done.add(name)
                                d[k] = _typed(vals[p], ty, is_list)
                    formats.append(d)
        except TypedValueError:
            return rows, r
        rows.append({
            "chrom": cols["chrom"][r].decode("utf-8"),
            "pos": int(base.extra["pos"][r]),
            "id": split("id", b";"),
            "ref": cols["ref"][r].decode("utf-8"),
            "alt": split("alt", b","),
            "qual": float(base.extra["qual"][r]) if base.extra["qual_valid"][r] else None,
            "filter": split("filter", b";"),
            "info": info,
            "formats": formats,
        })
    return rows, None


def is_valid_utf8(b: bytes) -> bool:
    arr, ptr = _as_buf(b)
    return bool(lib().orc_is_valid_utf8(ptr, arr.size))


def synth_fastq(n_bytes, file_offset=0, seed=SYNTH_FASTQ_SEED):
    out = np.empty(n_bytes, np.uint8)
    if n_bytes:
        lib().orc_synth_fastq(out.ctypes.data, file_offset, n_bytes, seed)
    return out


def _synth_var(fn, n, seed, cap):
    out = np.empty(cap, np.uint8)
    got = int(fn(out.ctypes.data, cap, n, seed))
    return out[:got].copy()


def synth_fastq_ragged(n_records, seed=SYNTH_FASTQ_SEED + 1):
    return _synth_var(lib().orc_synth_fastq_ragged, n_records, seed, n_records * 720 + 4096)


def synth_vcf(n_lines, seed=SYNTH_VCF_SEED):
    return _synth_var(lib().orc_synth_vcf, n_lines, seed, n_lines * 160 + 8192)


def synth_fasta(n_records, seed=SYNTH_FASTA_SEED):
    return _synth_var(lib().orc_synth_fasta, n_records, seed, n_records * 3300 + 4096)


def fastq_scan_baseline(data):
    arr, ptr = _as_buf(data)
    chk = C.c_uint64(0)
    n = int(lib().orc_fastq_scan_baseline(ptr, arr.size, C.byref(chk)))
    return n, int(chk.value)


def fastq_scan_baseline_mt(data, n_threads, reps):
    arr, ptr = _as_buf(data)
    return int(lib().orc_fastq_scan_baseline_mt(ptr, arr.size, n_threads, reps))


def infer_compression(uri, compression=None):
    return lib().orc_infer_compression(uri.encode(), compression.encode() if compression is not None else None).decode()


def replacement_scan(uri):
    r = lib().orc_replacement_scan(uri.encode())
    return r.decode() if r else None


# ---- decoder-level rules (round 5): what a compressed input's ROWS are -----------------------------------------------------------
# The reference hands compressed files to DataFusion 28's FileCompressionType::convert_stream -> async-compression 0.4.0 (flate2 /
# zstd) and bgzip'ed VCFs to noodles-bgzf (rust/src/arrow_reader.rs:60-91); which of these rules those decoders follow at the pinned
# versions is [RECALLED] or open (an async GzipDecoder without multiple_members may stop behind the FIRST member).
# What this build does, stated here so that tools/falsify_kit.py can put it in front of a real exon build:
#   gzip  every member of a concatenation is decoded, in order (BGZF is such a concatenation; its empty EOF member is optional;
#         a file of zero bytes holds no member: an error);
#         behind the last complete member only another member may follow: anything else is "invalid gzip header"; a member that
#         ends early, or whose CRC-32 / ISIZE trailer does not match, is an error;
#   zstd  every frame is decoded, skippable frames are skipped; bytes that begin no frame, a frame that ends early, a
#         Content_Checksum that does not match are errors;
#   an error is reported when the rows in front of it have been handed out (through SQL: the query fails).
def decode_by_rule(data: bytes, compression: str):
    """-> (decoded bytes in front of the first error, error text or None)"""
    import zlib
    data = bytes(data)
    if compression == "gzip":
        out, buf = bytearray(), data
        if not buf:
            return b"", "empty gzip file"   # (no member at all: what `gzip -t` calls an unexpected end of file)
        while buf:
            if len(buf) < 18 or buf[0] != 0x1F or buf[1] != 0x8B or buf[2] != 8:
                return bytes(out), "invalid gzip header"
            d = zlib.decompressobj(31)
            try:
                out += d.decompress(buf)
            except zlib.error as e:   # a bad block, a CRC-32 / ISIZE mismatch
                return bytes(out), f"corrupt gzip stream ({e})"
            if not d.eof:
                return bytes(out), "truncated gzip member"
            buf = d.unused_data
        return bytes(out), None
    if compression == "zstd":
        import ctypes as CC
        so = os.path.join(HERE, "libzstd_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", HERE, "libzstd_oracle.so"])
        o = CC.CDLL(so)
        o.zso_decompress.argtypes = [CC.c_char_p, CC.c_uint64, CC.c_void_p, CC.c_uint64, CC.POINTER(CC.c_uint64), CC.c_void_p, CC.c_uint64, CC.c_void_p]
        cap = max(1 << 20, 64 * len(data))
        while True:
            buf = CC.create_string_buffer(cap + 64)
            p = CC.c_uint64(0)
            rc = o.zso_decompress(data, len(data), buf, cap, CC.byref(p), None, 0, None)
            if rc == 0 or p.value < cap or cap >= (1 << 31):
                break
            cap *= 8   # (the output did not fit)
        return buf.raw[:p.value], (None if rc == 0 else f"corrupt or truncated zstd stream (oracle code {rc})")
    if compression in (None, "", "none", "uncompressed"):
        return data, None
    raise ValueError(compression)


class CompressedResult:
    """rows of a compressed input by the rules above: `table` = the parse of the bytes decoded in front of the first decoder
    error (None when the stream could not even be opened), `error` = the decoder's or the parser's error text (None: none)"""

    def __init__(self, table, typed_rows, error):
        self.table, self.typed_rows, self.error = table, typed_rows, error


def compressed_parse(fmt: str, data: bytes, compression: str, ext: str = None):
    """what `read_<fmt>('file.<ext>' [, compression = ...])` returns for these file bytes.  `ext` only names the case's file
    for the falsifiability kit (e.g. "fastq.gz"; None: <fmt> + the usual extension of the compression)"""
    text, derr = decode_by_rule(data, compression)
    if fmt == "fastq":
        t = fastq_parse(text, want_string_t=False)
    elif fmt == "fasta":
        t = fasta_parse(text)
    else:
        t = vcf_parse(text, want_string_t=False)
    typed, perr = None, None
    if fmt == "vcf" and not t.error_code:
        typed, bad = vcf_typed_rows(text)
        if bad is not None:
            perr = f"a value of row {bad} does not parse"
    if t.error_code and derr is None:
        perr = f"{t.error_message} (record {t.error_record})"
    # a stream that ends in an error: the rows in front of it come first; the text cut at the error may itself end inside a
    # record — that record's error is the decoder's, not a parse error
    return CompressedResult(t, typed, derr or perr)


# ---- schema rules: column names and DuckDB types of the three table functions ----------------------------------------------------
# test_fastq_scan.test:35-41 / test_fasta_scan.test:35 / test_vcf_record_scan.test:10-19 pin the names they select and that `alt` is a
# list, `qual` a float, `info` a struct; every other type here is [RECALLED] from exon 0.2.6's schema builders
# (rust/src/arrow_reader.rs:116-153 is where the schema comes from) as DuckDB's Arrow import renders it.
def schema_of(fmt: str, data: bytes = b""):
    """-> [(column name, DuckDB type as DESCRIBE prints it)]; VCF: from the header lines of `data`"""
    if fmt == "fastq":
        return [(c, "VARCHAR") for c in ("name", "description", "sequence", "quality_scores")]
    if fmt == "fasta":
        return [(c, "VARCHAR") for c in ("id", "description", "sequence")]
    info, fmts = vcf_header_keys(data)
    ty = {"Integer": "INTEGER", "Float": "FLOAT", "Flag": "BOOLEAN", "String": "VARCHAR"}

    def struct(keys):
        return "STRUCT(" + ", ".join(f"{k} {ty[t]}{'[]' if is_list else ''}" for k, t, is_list in keys) + ")"
    return [("chrom", "VARCHAR"), ("pos", "BIGINT"), ("id", "VARCHAR[]"), ("ref", "VARCHAR"), ("alt", "VARCHAR[]"), ("qual", "FLOAT"),
            ("filter", "VARCHAR[]"), ("info", struct(info)), ("formats", struct(fmts) + "[]")]
