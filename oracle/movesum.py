"""ctypes loader for oracle/movesum.c with a pure-Python fallback (small cases only)."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_movesum.so")
_lib = None


def _load():
    global _lib
    if _lib is None and os.path.exists(_SO):
        lib = ctypes.CDLL(_SO)
        lib.oracle_move_sum.restype = ctypes.c_int
        lib.oracle_move_sum.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_long,
                                        ctypes.c_long, ctypes.c_void_p]
        lib.oracle_bin_add.restype = None
        lib.oracle_bin_add.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_long,
                                       ctypes.c_long, ctypes.c_void_p]
        _lib = lib
    return _lib


def move_sum(a, window, min_count=1):
    """Stand-in for `bottleneck.move_sum(a, window, min_count=1)` on a 1-D float64 view
    (possibly negative-strided)."""
    assert min_count == 1 and a.ndim == 1 and a.dtype == np.float64
    n = a.shape[0]
    window = int(window)
    if window < 1 or window > n:
        raise ValueError("Moving window (=%d) must between 1 and %d, inclusive" % (window, n))
    y = np.empty(n, dtype=np.float64)
    lib = _load()
    if lib is not None:
        stride = a.strides[0] // 8
        rc = lib.oracle_move_sum(a.ctypes.data, stride, n, window, y.ctypes.data)
        assert rc == 0
        return y
    x = a.tolist()
    asum = 0.0
    for i in range(window):
        asum += x[i]
        y[i] = asum
    for i in range(window, n):
        asum += x[i] - x[i - window]
        y[i] = asum
    return y
