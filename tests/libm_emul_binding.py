"""ctypes binding of tests/_build/liblibm_emul.so: csrc/rt_libm.h compiled for the host, beside the host's libm.
Test infrastructure only."""
import ctypes as C
from pathlib import Path

import numpy as np

_L = C.CDLL(str(Path(__file__).resolve().parent / "_build" / "liblibm_emul.so"))
_P = C.POINTER(C.c_double)
_L.libm_emul_count_diffs.restype = C.c_int64
_L.libm_emul_count_diffs.argtypes = [C.c_int, _P, _P, C.c_int64, C.POINTER(C.c_int64)]
_L.libm_emul_eval.restype = None
_L.libm_emul_eval.argtypes = [C.c_int, _P, _P, C.c_int64, _P, _P]
FUNCTIONS = ("log", "sin", "acos", "atan2", "cos", "pow")


def _arr(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def count_diffs(which: str, a, b=None):
    """(number of arguments where rt_libm.h and the host libm differ, index of the first)"""
    a = _arr(a)
    b = a if b is None else _arr(b)
    first = C.c_int64(-1)
    n = _L.libm_emul_count_diffs(FUNCTIONS.index(which), a.ctypes.data_as(_P), b.ctypes.data_as(_P), a.size, C.byref(first))
    return int(n), int(first.value)


def evaluate(which: str, a, b=None):
    """(rt_libm.h on the host, the host's libm) for every argument"""
    a = _arr(np.atleast_1d(a))
    b = a if b is None else _arr(np.atleast_1d(b))
    mine, host = np.empty_like(a), np.empty_like(a)
    _L.libm_emul_eval(FUNCTIONS.index(which), a.ctypes.data_as(_P), b.ctypes.data_as(_P), a.size, mine.ctypes.data_as(_P),
                      host.ctypes.data_as(_P))
    return mine, host


def same_bits(x, y):
    """equal bit patterns, NaN == NaN whatever its sign and payload (those are not claimed)"""
    x, y = _arr(x), _arr(y)
    return (x.view(np.uint64) == y.view(np.uint64)) | (np.isnan(x) & np.isnan(y))
