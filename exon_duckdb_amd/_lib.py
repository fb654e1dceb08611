"""Loads libexon_gpu.so (in-tree build) and binds the C-ABI.  Fails loudly when it is missing."""
import ctypes as C
import os

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libexon_gpu.so")


class ExgError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libexon_gpu error {code}: {message}")
        self.code = code


_lib = None


def load_library(path=None):
    """dlopen the HIP library; raises ExgError if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p) and path is None:
        # in-tree build on first use (hipcc cross-compiles gfx950 without a GPU); never a CPU fallback
        try:
            from . import build as _build

            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise ExgError(abi.EXG_E_NO_DEVICE, f"{p} is missing and could not be built with hipcc: {e}") from e
    if not os.path.exists(p):
        raise ExgError(abi.EXG_E_NO_DEVICE, f"{p} not found: run `python __graft_entry__.py build` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # PyTorch bundles a HIP runtime of its own; two HIP runtimes in one process do not share the GPU (whichever comes second
    # sees no device).  Whoever uses this package beside torch imports torch FIRST (tests/conftest.py, bench.py and smoke()
    # do): libexon_gpu.so then binds to the copy that is already there.  This loader does not import torch by itself — that
    # would cost every user seconds and silently change which HIP runtime the product binds to; EXG_PRELOAD_TORCH=1 asks
    # for it (a process that will import torch later and cannot order its imports).
    import sys
    # a reader keeps six to eight streams busy: more hardware queues than HIP's default of four (csrc/exg_api.hip; read by the
    # runtime at its first HIP call — a process that initialised HIP before this import exports it itself, like bench.py)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if "torch" not in sys.modules and os.environ.get("EXG_PRELOAD_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception as e:  # noqa: BLE001
            print(f"exon_duckdb_amd: EXG_PRELOAD_TORCH is set but torch does not import: {e}", file=sys.stderr)
    l = C.CDLL(p)
    for name, (res, args) in abi.SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    if l.exg_abi_version() != abi.EXG_ABI_VERSION:
        raise ExgError(abi.EXG_E_INVALID_ARG, "ABI version mismatch between abi.py and libexon_gpu.so")
    if path is None:
        _lib = l
    return l


_test_lib = None


def load_test_library():
    """libexon_tf_test.so: test / bench scaffolding built next to the product library and linked against it (synthetic
    inputs generated in HBM, chunk-draining consumers, the DuckDB-API harness, host-only introspection)."""
    global _test_lib
    if _test_lib is None:
        load_library()  # the product library first (the scaffolding links against it)
        from . import build as _build
        if not os.path.exists(_build.TEST_LIB):
            _build.build(verbose=False)
        l = C.CDLL(_build.TEST_LIB)
        for name, (res, args) in abi.TEST_SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _test_lib = l
    return _test_lib


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _LazyLib()


def check(rc):
    if rc != 0:
        raise ExgError(rc, load_library().exg_last_error_message().decode("utf-8", "replace"))
