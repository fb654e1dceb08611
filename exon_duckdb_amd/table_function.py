"""Python front-end of the host-side table functions (csrc/exon_table_function.cpp).

It plays the role DuckDB plays for the reference: look a table function up in the catalog, bind it,
init the scan with a projection, pull DataChunks until an empty one comes back.  Used by the parity
tests, which read like the reference's sqllogictests:

    con.table_function("read_fastq", path).count()
    con.table_function("read_fasta", path, compression="gzip").fetchall()
    con.from_path("x/test.fasta")                     # replacement scan
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import ExgError, load_library

ROW_ID = (1 << 64) - 1


class ExgType(C.Structure):
    pass


ExgType._fields_ = [("type", C.c_int), ("nullable", C.c_int), ("name", C.c_char_p), ("n_children", C.c_int),
                    ("children", C.POINTER(ExgType))]


class ExgVector(C.Structure):
    pass


ExgVector._fields_ = [("data", C.c_void_p), ("validity", C.c_void_p), ("length", C.c_uint64), ("n_children", C.c_int),
                      ("children", C.POINTER(ExgVector))]


class Schema(C.Structure):
    _fields_ = [("n_columns", C.c_int), ("names", C.c_char_p * 16), ("types", C.c_int * 16), ("nullable", C.c_int * 16),
                ("tree", C.POINTER(ExgType) * 16)]


class Chunk(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_columns", C.c_int), ("data", C.c_void_p * 16),
                ("validity", C.c_void_p * 16), ("keepalive", C.c_void_p), ("vectors", C.POINTER(ExgVector) * 16),
                ("batch_no", C.c_uint64)]


def type_tree(t):
    """exg_type -> a plain Python description that outlives the reader: (type id, name, [children])"""
    return (int(t.type), (t.name or b"").decode(), [type_tree(t.children[i]) for i in range(t.n_children)])


def type_sql(tree):
    """the DuckDB spelling of a type tree, e.g. STRUCT(DP INTEGER, AF FLOAT[])"""
    tid, _, kids = tree
    if tid == abi.EXG_TYPE_LIST:
        return type_sql(kids[0]) + "[]"
    if tid == abi.EXG_TYPE_STRUCT:
        return "STRUCT(" + ", ".join(f"{k[1]} {type_sql(k)}" for k in kids) + ")"
    return {abi.EXG_TYPE_VARCHAR: "VARCHAR", abi.EXG_TYPE_BIGINT: "BIGINT", abi.EXG_TYPE_FLOAT: "FLOAT",
            abi.EXG_TYPE_INTEGER: "INTEGER", abi.EXG_TYPE_BOOLEAN: "BOOLEAN"}[tid]


def _valid_bits(validity_ptr, n):
    if not validity_ptr or n == 0:
        return None
    words = np.ctypeslib.as_array(C.cast(validity_ptr, C.POINTER(C.c_uint64)), shape=((n + 63) // 64,))
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n]


def decode_vector(vec, tree):
    """One exg_vector (DuckDB layout) -> list of Python values: bytes / int / float / bool / list / dict / None."""
    tid, _, kids = tree
    n = int(vec.length)
    if tid == abi.EXG_TYPE_VARCHAR:
        return _decode_strings(vec.data, vec.validity, n) if n else []
    valid = _valid_bits(vec.validity, n)
    if tid == abi.EXG_TYPE_LIST:
        child = decode_vector(vec.children[0], kids[0])
        out = []
        if n:
            ent = np.ctypeslib.as_array(C.cast(vec.data, C.POINTER(C.c_uint64)), shape=(2 * n,)).reshape(n, 2)
            for i in range(n):
                if valid is not None and not valid[i]:
                    out.append(None)
                else:
                    o, l = int(ent[i, 0]), int(ent[i, 1])
                    assert o + l <= len(child), "list entry outside the chunk's child vector"
                    out.append(child[o:o + l])
        return out
    if tid == abi.EXG_TYPE_STRUCT:
        cols = [decode_vector(vec.children[i], kids[i]) for i in range(vec.n_children)]
        names = [k[1] for k in kids]
        return [None if (valid is not None and not valid[i]) else {nm: c[i] for nm, c in zip(names, cols)} for i in range(n)]
    dt = {abi.EXG_TYPE_BIGINT: np.int64, abi.EXG_TYPE_FLOAT: np.float32, abi.EXG_TYPE_INTEGER: np.int32,
          abi.EXG_TYPE_BOOLEAN: np.uint8}[tid]
    if n == 0:
        return []
    arr = np.ctypeslib.as_array(C.cast(vec.data, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).copy()
    vals = [bool(x) for x in arr] if tid == abi.EXG_TYPE_BOOLEAN else arr.tolist()
    if valid is not None:
        vals = [x if ok else None for x, ok in zip(vals, valid)]
    return vals


class FilterNode(C.Structure):
    _fields_ = [("kind", C.c_int), ("column", C.c_int), ("cmp", C.c_int), ("const_type", C.c_int), ("n_children", C.c_int),
                ("constant", C.c_char_p)]


class F:
    """Builders of DuckDB TableFilters (what the optimizer hands a scan with filter_pushdown = true):
    F.cmp('>=', 'r5'), F.isnull(), F.notnull(), F.and_(...), F.or_(...)."""
    OPS = {"=": 25, "!=": 26, "<": 27, ">": 28, "<=": 29, ">=": 30}

    @staticmethod
    def cmp(op, value):
        return ("cmp", F.OPS[op], value)

    @staticmethod
    def isnull():
        return ("isnull",)

    @staticmethod
    def notnull():
        return ("notnull",)

    @staticmethod
    def and_(*children):
        return ("and", children)

    @staticmethod
    def or_(*children):
        return ("or", children)


def _flatten(node, column, const_type, out):
    if node[0] == "cmp":
        v = node[2]
        text = v if isinstance(v, bytes) else str(v).encode()
        out.append(FilterNode(0, column, node[1], const_type, 0, text))
    elif node[0] == "isnull":
        out.append(FilterNode(1, column, 0, 0, 0, None))
    elif node[0] == "notnull":
        out.append(FilterNode(2, column, 0, 0, 0, None))
    else:
        out.append(FilterNode(4 if node[0] == "and" else 3, column, 0, 0, len(node[1]), None))
        for ch in node[1]:
            _flatten(ch, column, const_type, out)


_tf = None


def _lib():
    """libexon_tf_test.so: the table-function glue (csrc/exon_table_function.hpp) instantiated over the DuckDB API slice
    of csrc/testing/duck_mini.hpp — test scaffolding, built next to the product library and linked against it."""
    global _tf
    if _tf is None:
        from ._lib import load_test_library
        l = load_test_library()
        l.exon_tf_last_error.restype = C.c_char_p
        l.exon_tf_catalog_has.restype = C.c_int
        l.exon_tf_catalog_has.argtypes = [C.c_char_p]
        l.exon_tf_bind.restype = C.c_int
        l.exon_tf_bind.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        l.exon_tf_schema.argtypes = [C.c_void_p, C.POINTER(Schema)]
        l.exon_tf_init_global.restype = C.c_int
        l.exon_tf_init_global.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.POINTER(FilterNode), C.c_int, C.POINTER(C.c_uint64)]
        l.exon_tf_init_local.restype = C.c_int
        l.exon_tf_init_local.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.exon_tf_scan.restype = C.c_int
        l.exon_tf_scan.argtypes = [C.c_void_p, C.c_int, C.POINTER(Chunk), C.POINTER(C.c_uint64)]
        l.exon_tf_close.argtypes = [C.c_void_p]
        l.exon_tf_cardinality.restype = C.c_uint64
        l.exon_tf_cardinality.argtypes = [C.c_void_p]
        l.exon_tf_quality_scores.restype = None
        l.exon_tf_quality_scores.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_int32)]
        l.exon_replacement_scan.restype = C.c_int
        l.exon_replacement_scan.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        _tf = l
    return _tf


def _err(l):
    return (l.exon_tf_last_error() or b"").decode("utf-8", "replace")


def _decode_strings(ptr, validity_ptr, n):
    """n duckdb::string_t at host address ptr -> list of bytes / None."""
    raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * 16,)).reshape(n, 16)
    lens = raw[:, :4].copy().view(np.uint32).reshape(n)
    ptrs = raw[:, 8:16].copy().view(np.uint64).reshape(n)
    valid = None
    if validity_ptr:
        words = np.ctypeslib.as_array(C.cast(validity_ptr, C.POINTER(C.c_uint64)), shape=((n + 63) // 64,))
        valid = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
    out = []
    for i in range(n):
        if valid is not None and not valid[i]:
            out.append(None)
        elif lens[i] <= 12:
            out.append(raw[i, 4:4 + lens[i]].tobytes())
        else:
            out.append(C.string_at(int(ptrs[i]), int(lens[i])))   # zero-copy payload in pinned host memory
    return out


class Relation:
    def __init__(self, fn_name, path, compression=None):
        self._l = _lib()
        self.fn_name, self.path, self.compression = fn_name, path, compression
        h = C.c_void_p()
        rc = self._l.exon_tf_bind(fn_name.encode(), path.encode(), compression.encode() if compression else None, C.byref(h))
        if rc != 0:
            raise ExgError(rc, _err(self._l))          # bind-time error, e.g. read_fastq('')
        sch = Schema()
        self._l.exon_tf_schema(h, C.byref(sch))
        self.names = [sch.names[i].decode() for i in range(sch.n_columns)]
        self.types = [sch.types[i] for i in range(sch.n_columns)]
        self.trees = [type_tree(sch.tree[i].contents) for i in range(sch.n_columns)]
        #: TableFunction::cardinality (module.cpp:307): the row estimate handed to the planner, 0 = none
        self.estimated_cardinality = int(self._l.exon_tf_cardinality(h))
        self._l.exon_tf_close(h)

    #: how many scan threads `_scan` runs: None = MaxThreads() (DuckDB's upper bound); DuckDB itself runs fewer when
    #: `SET threads` is smaller or the pipeline is sequential — the tests set 1 (and other numbers below MaxThreads())
    scan_threads = None

    def _scan(self, column_ids, filters=None, decode=None):
        """Plays DuckDB: init_global once, init_local + the scan loop on up to MaxThreads() threads, every chunk handed
        to `decode` while it is alive.  -> [(batch_index, n_rows, decoded)] in batch order (stable: the chunks of one
        batch keep the order their thread produced them in)."""
        import threading
        h = C.c_void_p()
        rc = self._l.exon_tf_bind(self.fn_name.encode(), self.path.encode(),
                                  self.compression.encode() if self.compression else None, C.byref(h))
        if rc != 0:
            raise ExgError(rc, _err(self._l))
        try:
            ids = (C.c_uint64 * len(column_ids))(*column_ids)
            nodes = []
            for name, node in (filters or {}).items():
                # TableFilterSet: keyed by the position of the column in column_ids (module.cpp:201-214)
                cid = self.names.index(name)
                _flatten(node, list(column_ids).index(cid), self.types[cid], nodes)
            arr = (FilterNode * max(1, len(nodes)))(*nodes)
            max_threads = C.c_uint64(1)
            if self._l.exon_tf_init_global(h, ids, len(column_ids), arr, len(nodes), C.byref(max_threads)) != 0:
                raise ExgError(abi.EXG_E_IO, _err(self._l))
            self.last_max_threads = int(max_threads.value)
            out, errors = [], []
            lock = threading.Lock()

            def worker():
                lid = C.c_int(0)
                if self._l.exon_tf_init_local(h, C.byref(lid)) != 0:
                    with lock:
                        errors.append(ExgError(abi.EXG_E_IO, _err(self._l)))
                    return
                while True:
                    ch = Chunk()
                    bi = C.c_uint64(0)
                    if self._l.exon_tf_scan(h, lid, C.byref(ch), C.byref(bi)) != 0:
                        with lock:
                            errors.append(ExgError(abi.EXG_E_PARSE, _err(self._l)))
                        return
                    if ch.n_rows == 0:
                        return
                    item = (int(bi.value), int(ch.n_rows), decode(ch) if decode else None)
                    with lock:
                        out.append(item)

            n_threads = self.last_max_threads if self.scan_threads is None else max(1, min(self.scan_threads, self.last_max_threads))
            threads = [threading.Thread(target=worker) for _ in range(n_threads)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            out.sort(key=lambda x: x[0])
            return out
        finally:
            self._l.exon_tf_close(h)

    def count(self, filters=None):
        """SELECT count(*) [WHERE filters]: only the row id (and the filter columns) are projected."""
        ids = [ROW_ID]
        if filters:
            ids += [self.names.index(c) for c in filters]
        return sum(n for _, n, _ in self._scan(ids, filters))

    def chunk_sizes(self, columns=None):
        cols = self.names if columns is None else columns
        return [n for _, n, _ in self._scan([self.names.index(c) for c in cols])]

    def fetchall(self, columns=None, limit=None, where=None, filters=None):
        """SELECT columns ... [WHERE where(row_dict)] [LIMIT limit] -> list of tuples (bytes/None/int/float/list/dict),
        in file order (chunks sorted by get_batch_index).  `filters` = {column: F....}: pushed down into the scan like
        DuckDB's TableFilterSet; `where` = a Python predicate applied above the scan."""
        cols = self.names if columns is None else columns
        ids = [self.names.index(c) for c in cols]
        for c in (filters or {}):                     # DuckDB keeps filter columns in column_ids
            if self.names.index(c) not in ids:
                ids.append(self.names.index(c))

        def decode(ch):
            return [decode_vector(ch.vectors[k].contents, self.trees[cid]) for k, cid in enumerate(ids[:len(cols)])]

        rows = []
        for _, n, decoded in self._scan(ids, filters, decode):
            for i in range(n):
                row = tuple(d[i] for d in decoded)
                if where is None or where(dict(zip(cols, row))):
                    rows.append(row)
                    if limit is not None and len(rows) >= limit:
                        return rows
        return rows


def abi_type(name):
    return {"VARCHAR": 1, "BIGINT": 2, "FLOAT": 3, "INTEGER": 4, "BOOLEAN": 5, "LIST": 6, "STRUCT": 7}[name]


class Connection:
    """LOAD exon; then call table functions by name."""

    def has_table_function(self, name):
        return bool(_lib().exon_tf_catalog_has(name.encode()))

    def table_function(self, name, path, compression=None):
        if not self.has_table_function(name):
            raise ExgError(abi.EXG_E_INVALID_ARG, f"Catalog Error: Table Function with name {name} does not exist!")
        return Relation(name, path, compression)

    def replacement_scan(self, table_name):
        buf = C.create_string_buffer(64)
        if _lib().exon_replacement_scan(table_name.encode(), buf, 64):
            return buf.value.decode()
        return None

    def from_path(self, path):
        """SELECT ... FROM 'path'  (WTArrowTableFunction::ReplacementScan, module.cpp:320-382)."""
        fn = self.replacement_scan(path)
        if fn is None:
            raise ExgError(abi.EXG_E_INVALID_ARG, f"Catalog Error: Table with name {path} does not exist!")
        return Relation(fn, path)


def connect():
    return Connection()


def quality_score_string_to_list(s: bytes):
    """the SQL scalar of exon/src/exon/fastq_functions/module.cpp:28-54 as the DuckDB shim registers it (host arithmetic of
    duckdb_shim/exon_extension.cpp: QualityScoreStringToList) -> list of int"""
    out = (C.c_int32 * max(1, len(s)))()
    _lib().exon_tf_quality_scores(s, len(s), out)
    return list(out[:len(s)])
