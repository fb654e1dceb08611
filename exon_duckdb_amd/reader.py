"""The chunk boundary of libexon_gpu (exg_open / exg_next_chunk / exg_count_only) from Python, with byte-range shards:
one process per GPU opens the same path with shard_index = its rank (include/exon_gpu.h, exg_open_args)."""
import ctypes as C

from . import abi
from ._lib import ExgError, load_library
from .table_function import Chunk, Schema, decode_vector, type_tree


class ShardReader:
    def __init__(self, path, file_format, shard_index=0, shard_count=1, compression=None, device=0, device_batch_bytes=0,
                 filters=None, batch_rows=2048, columns=None, expect_chunks=False):
        """columns: indices (exg_schema_of order) of the columns to copy back; None = all (exg_open_args.columns);
        expect_chunks: exg_open_args.flags = EXG_OPEN_CHUNKS — chunks will be pulled (a compressed input mirrors its decoded segments from the first on)"""
        self._l = load_library()
        self._l.exg_open.argtypes = [C.POINTER(abi.OpenArgs), C.POINTER(C.c_void_p)]
        self._l.exg_next_chunk.argtypes = [C.c_void_p, C.POINTER(Chunk)]
        self._l.exg_release_chunk.argtypes = [C.c_void_p, C.POINTER(Chunk)]
        self._l.exg_count_only.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self._l.exg_schema_of.argtypes = [C.c_void_p, C.POINTER(Schema)]
        self._l.exg_reader_error.restype = C.c_char_p
        self._l.exg_reader_error.argtypes = [C.c_void_p]
        self._l.exg_close.argtypes = [C.c_void_p]
        a = abi.OpenArgs(path.encode(), file_format.encode(), compression.encode() if compression else None, batch_rows, device,
                         device_batch_bytes, filters.encode() if filters else None, shard_index, shard_count,
                         sum(1 << int(c) for c in columns) if columns is not None else 0, abi.EXG_OPEN_CHUNKS if expect_chunks else 0)
        self.columns = None if columns is None else sorted(int(c) for c in columns)
        self._r = C.c_void_p()
        rc = self._l.exg_open(C.byref(a), C.byref(self._r))
        if rc != 0:
            raise ExgError(rc, self._l.exg_last_error_message().decode("utf-8", "replace"))
        sch = Schema()
        rc = self._l.exg_schema_of(self._r, C.byref(sch))
        if rc != 0:
            msg = (self._l.exg_reader_error(self._r) or b"").decode("utf-8", "replace")
            self._l.exg_close(self._r)
            self._r = C.c_void_p()
            raise ExgError(rc, msg)
        self.names = [sch.names[i].decode() for i in range(sch.n_columns)]
        self.types = [sch.types[i] for i in range(sch.n_columns)]
        self.trees = [type_tree(sch.tree[i].contents) for i in range(sch.n_columns)]

    def _fail(self, rc):
        raise ExgError(rc, (self._l.exg_reader_error(self._r) or b"").decode("utf-8", "replace"))

    def count(self):
        n = C.c_uint64(0)
        rc = self._l.exg_count_only(self._r, C.byref(n))
        if rc != 0:
            self._fail(rc)
        return int(n.value)

    def rows(self):
        """All rows of this shard as tuples (bytes / None / int / float), in file order."""
        out = []
        while True:
            ch = Chunk()
            rc = self._l.exg_next_chunk(self._r, C.byref(ch))
            if rc != 0:
                self._fail(rc)
            n = int(ch.n_rows)
            if n == 0:
                return out
            want = range(len(self.names)) if self.columns is None else self.columns
            for k in range(len(self.names)):
                if k not in want:
                    assert not ch.vectors[k], f"column {k} was not asked for but came back"
            cols = [decode_vector(ch.vectors[k].contents, self.trees[k]) for k in want]
            out.extend(zip(*cols))
            self._l.exg_release_chunk(self._r, C.byref(ch))

    def stats(self):
        """exg_reader_stats_of: device bytes held now / at their peak, batches scanned, decoded segments consumed."""
        st = abi.ReaderStats()
        self._l.exg_reader_stats_of.argtypes = [C.c_void_p, C.POINTER(abi.ReaderStats)]
        rc = self._l.exg_reader_stats_of(self._r, C.byref(st))
        if rc != 0:
            self._fail(rc)
        return {f: getattr(st, f) for f, _ in abi.ReaderStats._fields_ if f != "reserved"}

    def digest(self, nested=False, per_column=False):
        """(rows, blake2b over the columns of every row in file order) without building Python rows for all of them: the
        string_t vectors are resolved with numpy, BIGINT / FLOAT columns hashed as their bytes; LIST / STRUCT columns only
        with nested=True (decoded to Python values: slow).  What a content check of a big input needs."""
        import hashlib

        import numpy as np
        hs = [hashlib.blake2b(digest_size=16) for _ in self.types]   # one per column: independent of the chunking
        rows = 0
        while True:
            ch = Chunk()
            rc = self._l.exg_next_chunk(self._r, C.byref(ch))
            if rc != 0:
                self._fail(rc)
            n = int(ch.n_rows)
            if n == 0:
                if per_column:
                    return rows, [x.hexdigest() for x in hs]
                return rows, hashlib.blake2b(b"".join(x.digest() for x in hs), digest_size=16).hexdigest()
            rows += n
            for k, t in enumerate(self.types):
                if self.columns is not None and k not in self.columns:
                    continue   # (not asked for: exg_chunk.data / vectors are NULL for it; its hash stays the empty one)
                h = hs[k]
                if t in (abi.EXG_TYPE_BIGINT, abi.EXG_TYPE_FLOAT):
                    width = 8 if t == abi.EXG_TYPE_BIGINT else 4
                    vals = np.ctypeslib.as_array(C.cast(ch.data[k], C.POINTER(C.c_uint8)), shape=(n * width,)).copy()
                    if ch.validity[k]:   # a NULL row's value is unspecified: hash zeros there
                        words = np.ctypeslib.as_array(C.cast(ch.validity[k], C.POINTER(C.c_uint64)), shape=((n + 63) // 64,))
                        ok = np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)
                        vals.reshape(n, width)[~ok] = 0
                        # (the validity byte travels with its row: independent of the chunking)
                        vals = np.concatenate([ok.astype(np.uint8)[:, None], vals.reshape(n, width)], axis=1)
                    h.update(vals.tobytes())
                    continue
                if t != abi.EXG_TYPE_VARCHAR:
                    if nested:   # row by row: the digest must not depend on how the rows were cut into chunks
                        for v in decode_vector(ch.vectors[k].contents, self.trees[k]):
                            h.update(repr(v).encode())
                            h.update(b"\x00")
                    continue
                raw = np.ctypeslib.as_array(C.cast(ch.data[k], C.POINTER(C.c_uint8)), shape=(n * 16,)).reshape(n, 16)
                lens = raw[:, :4].copy().view(np.uint32).reshape(n)
                ptrs = raw[:, 8:16].copy().view(np.uint64).reshape(n)
                valid = None
                if ch.validity[k]:
                    words = np.ctypeslib.as_array(C.cast(ch.validity[k], C.POINTER(C.c_uint64)), shape=((n + 63) // 64,))
                    valid = np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
                for i in range(n):
                    if valid is not None and not valid[i]:
                        h.update(b"\xff\x00NULL")
                    elif lens[i] <= 12:
                        h.update(raw[i, 4:4 + lens[i]].tobytes())
                    else:
                        h.update(C.string_at(int(ptrs[i]), int(lens[i])))
                    h.update(b"\x00")
            self._l.exg_release_chunk(self._r, C.byref(ch))

    def close(self):
        if self._r:
            self._l.exg_close(self._r)
            self._r = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
