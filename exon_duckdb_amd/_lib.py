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
    # sees no device).  Whoever uses this package beside torch (the tests, bench.py, smoke()) therefore gets torch's runtime
    # loaded first — libexon_gpu.so then binds to the copy that is already there.  Without torch nothing changes.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001
        pass
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


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _LazyLib()


def check(rc):
    if rc != 0:
        raise ExgError(rc, load_library().exg_last_error_message().decode("utf-8", "replace"))
