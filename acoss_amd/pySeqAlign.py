"""
Drop-in for the reference's Cython module `pySeqAlign` (benchmarking/pySeqAlign.pyx:7,14,21):
same three callables, same argument order, same in-place-D contract -- executed on the MI355X
through the C ABI's reference entry points (include/acoss_mi355x.h group 1).

    from acoss_amd.pySeqAlign import qmax, dmax, swconstrained

S: np.ndarray[uint8, ndim=1, C-contiguous] of rows*cols entries; D: np.ndarray[float32, ndim=1,
C-contiguous] of rows*cols (qmax, dmax) or (rows+1)*(cols+1) (swconstrained) entries, zeroed by
the caller and overwritten in place; returns a Python float.  Wrong dtype / ndim / contiguity
raises ValueError, None raises TypeError (as the Cython buffer checks do).
"""
import math

import numpy as np

from . import _lib


def _buffer(arr, dtype, name):
    if arr is None:
        raise TypeError("Argument '%s' must not be None" % name)
    if not isinstance(arr, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(arr).__name__))
    if arr.dtype != dtype:
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s'" % (np.dtype(dtype).name, arr.dtype.name))
    if arr.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % arr.ndim)
    if not arr.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    return arr


def _call(fn_name, S, D, a, b, need_s, need_d):
    S = _buffer(S, np.uint8, "SParam")
    D = _buffer(D, np.float32, "DParam")
    a, b = int(a), int(b)
    # the Cython wrapper does no bounds checking; here a short buffer is an error, not a fault
    if a > 0 and b > 0 and (S.size < need_s(a, b) or D.size < need_d(a, b)):
        raise ValueError("%s: buffers too small for a %d x %d matrix" % (fn_name, a, b))
    if not D.flags.writeable:
        raise ValueError("buffer source array is read-only")
    res = getattr(_lib.load(), fn_name)(S.ctypes.data, D.ctypes.data, a, b)
    if math.isnan(res):
        raise _lib.AcossError("%s failed: %s" % (fn_name, _lib.last_error()))
    return float(res)


def qmax(SParam, DParam, N, M):
    """pySeqAlign.pyx:14 (rows first)."""
    return _call("qmax_c", SParam, DParam, N, M, lambda r, c: r * c, lambda r, c: r * c)


def dmax(SParam, DParam, N, M):
    """pySeqAlign.pyx:21 (rows first)."""
    return _call("dmax_c", SParam, DParam, N, M, lambda r, c: r * c, lambda r, c: r * c)


def swconstrained(SParam, DParam, N, M):
    """pySeqAlign.pyx:7: S is N x M, D is (N+1) x (M+1)."""
    return _call("swalignimpconstrained", SParam, DParam, N, M, lambda r, c: r * c,
                 lambda r, c: (r + 1) * (c + 1))
