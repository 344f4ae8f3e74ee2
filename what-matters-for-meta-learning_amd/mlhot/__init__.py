"""mlhot - MI355X-native CNP/ANP meta-batch hot path (host side).

`lib()` returns the ctypes binding of csrc/libmlhot.so.  There is no fallback: if the
HIP library is missing, or a tensor is not on a HIP device, the call raises.
MLHOT_LIB=<path> selects another build of the same library (e.g. the -DMLHOT_TS instrumented one);
MLHOT_OPTS="name=value,..." applies mlhot_set_option() calls right after loading (kernel A/B runs of bench.py).
"""
import os
import threading

from .binding import MlhotError, MlhotLib  # noqa: F401
from .build import PRODUCT_SO, build_product  # noqa: F401

_lock = threading.Lock()
_lib = None


def lib():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                loaded = MlhotLib(os.environ.get("MLHOT_LIB") or PRODUCT_SO)
                for item in filter(None, os.environ.get("MLHOT_OPTS", "").split(",")):
                    name, _, value = item.partition("=")
                    loaded.set_option(name.strip(), int(value))
                _lib = loaded
    return _lib
