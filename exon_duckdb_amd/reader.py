"""The chunk boundary of libexon_gpu (exg_open / exg_next_chunk / exg_count_only) from Python, with byte-range shards:
one process per GPU opens the same path with shard_index = its rank (include/exon_gpu.h, exg_open_args)."""
import ctypes as C

from . import abi
from ._lib import ExgError, load_library
from .table_function import Chunk, Schema, decode_vector, type_tree


class ShardReader:
    def __init__(self, path, file_format, shard_index=0, shard_count=1, compression=None, device=0, device_batch_bytes=0,
                 filters=None, batch_rows=2048):
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
                         device_batch_bytes, filters.encode() if filters else None, shard_index, shard_count)
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
            cols = [decode_vector(ch.vectors[k].contents, self.trees[k]) for k in range(len(self.names))]
            out.extend(zip(*cols))
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
