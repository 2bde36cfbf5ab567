"""
ctypes front-end of the CPU oracle (oracle/acoss_oracle.c) and of the reference alignment
kernel compiled into oracle/_ref/ (oracle/Makefile).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from acoss_amd/.  Parity status: pinned against
tests/golden/ (generated from the reference itself by tests/golden/make_golden.py).

Function names mirror the reference's (CRPUtils.py / pySeqAlign.pyx) so that tests read like
calls into the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = {}

_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)
_up = ctypes.POINTER(ctypes.c_ubyte)


def build(quiet=True):
    """Compile the oracle (and oracle/_ref when /root/reference is mounted)."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libacoss_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_np_sum.restype = ctypes.c_double
        L.orc_np_sum.argtypes = [_dp, ctypes.c_long]
        L.orc_global_chroma.restype = ctypes.c_int
        L.orc_global_chroma.argtypes = [_dp, ctypes.c_long, ctypes.c_int, _dp]
        L.orc_get_oti.restype = ctypes.c_int
        L.orc_get_oti.argtypes = [_dp, _dp, ctypes.c_int]
        L.orc_csm_f64.restype = None
        L.orc_csm_f64.argtypes = [_dp, ctypes.c_long, _dp, ctypes.c_long, ctypes.c_int,
                                  ctypes.c_int, _dp]
        L.orc_csm_f32.restype = None
        L.orc_csm_f32.argtypes = [_fp, ctypes.c_long, _fp, ctypes.c_long, ctypes.c_int,
                                  ctypes.c_int, _fp]
        L.orc_sliding_csm_f64.restype = ctypes.c_int
        L.orc_sliding_csm_f64.argtypes = [_dp, ctypes.c_long, ctypes.c_long, ctypes.c_int, _dp]
        L.orc_sliding_csm_f32.restype = ctypes.c_int
        L.orc_sliding_csm_f32.argtypes = [_fp, ctypes.c_long, ctypes.c_long, ctypes.c_int, _dp]
        L.orc_nneighbs.restype = ctypes.c_long
        L.orc_nneighbs.argtypes = [ctypes.c_double, ctypes.c_long]
        for name in ("orc_csm_to_binary", "orc_csm_to_binary_mutual"):
            fn = getattr(L, name)
            fn.restype = None
            fn.argtypes = [_dp, ctypes.c_long, ctypes.c_long, ctypes.c_double, _up]
        for name in ("orc_qmax", "orc_dmax", "orc_swc"):
            fn = getattr(L, name)
            fn.restype = ctypes.c_float
            fn.argtypes = [_up, _fp, ctypes.c_int, ctypes.c_int]
        L.orc_serra09_pair.restype = ctypes.c_int
        L.orc_serra09_pair.argtypes = [_dp, _dp, ctypes.c_long, _dp, _dp, ctypes.c_long,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_int, _dp, _dp]
        L.orc_serra09_pairs.restype = ctypes.c_int
        L.orc_serra09_pairs.argtypes = [_dp, ctypes.POINTER(ctypes.c_int64), _dp,
                                        ctypes.POINTER(ctypes.c_int32), ctypes.c_long,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_int, ctypes.c_int, _dp, _dp]
        _LIB = L
    return _LIB


def ref_lib(variant="Ofast"):
    """The reference's SequenceAlignment.c compiled as-is (oracle/_ref).  None if absent."""
    if variant not in _REF:
        name = "libseqalign_ref.so" if variant == "Ofast" else "libseqalign_ref_O2.so"
        path = os.path.join(_HERE, "_ref", name)
        if not os.path.exists(path):
            _REF[variant] = None
        else:
            R = ctypes.CDLL(path)
            for fn in (R.qmax_c, R.dmax_c, R.swalignimpconstrained):
                fn.restype = ctypes.c_float  # the C functions return float (pxd says double)
                fn.argtypes = [_up, _fp, ctypes.c_int, ctypes.c_int]
            _REF[variant] = R
    return _REF[variant]


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def _u(a):
    return a.ctypes.data_as(_up)


# ---------------------------------------------------------------------------------------
# CRPUtils.py mirrors
# ---------------------------------------------------------------------------------------
def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().orc_np_sum(_d(a), a.size)


def global_chroma(chroma):
    """Serra09.py:24-28; chroma is (nframes, nbins)."""
    chroma = np.ascontiguousarray(chroma, dtype=np.float64)
    out = np.zeros(chroma.shape[1])
    if lib().orc_global_chroma(_d(chroma), chroma.shape[0], chroma.shape[1], _d(out)) != 0:
        raise IOError("Wrong axis for the input chroma array. Expected shape "
                      "'(frame_size, bin_size)'")
    return out


def get_oti(C1, C2):
    """CRPUtils.py:109-136."""
    C1 = np.ascontiguousarray(C1, dtype=np.float64)
    C2 = np.ascontiguousarray(C2, dtype=np.float64)
    return int(lib().orc_get_oti(_d(C1), _d(C2), C1.size))


def get_csm(X, Y, shift=0):
    """CRPUtils.py:67-84; X (M,d), Y (N,d); dtype follows the inputs (f32 or f64)."""
    if X.dtype == np.float32 and Y.dtype == np.float32:
        X = np.ascontiguousarray(X)
        Y = np.ascontiguousarray(Y)
        out = np.empty((X.shape[0], Y.shape[0]), dtype=np.float32)
        lib().orc_csm_f32(_f(X), X.shape[0], _f(Y), Y.shape[0], X.shape[1], shift, _f(out))
        return out
    X = np.ascontiguousarray(X, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    out = np.empty((X.shape[0], Y.shape[0]))
    lib().orc_csm_f64(_d(X), X.shape[0], _d(Y), Y.shape[0], X.shape[1], shift, _d(out))
    return out


def sliding_csm(D, win):
    """CRPUtils.py:24-45; always returns float64."""
    M, N = D.shape[0] - win + 1, D.shape[1] - win + 1
    S = np.zeros((max(M, 0), max(N, 0)))
    if D.dtype == np.float32:
        D = np.ascontiguousarray(D)
        lib().orc_sliding_csm_f32(_f(D), D.shape[0], D.shape[1], win, _d(S))
    else:
        D = np.ascontiguousarray(D, dtype=np.float64)
        lib().orc_sliding_csm_f64(_d(D), D.shape[0], D.shape[1], win, _d(S))
    return S


def csm_to_binary(D, kappa):
    """CRPUtils.py:169-199."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    B = np.zeros(D.shape, dtype=np.uint8)
    lib().orc_csm_to_binary(_d(D), D.shape[0], D.shape[1], float(kappa), _u(B))
    return B


def csm_to_binary_mutual(D, kappa):
    """CRPUtils.py:201-219."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    B = np.zeros(D.shape, dtype=np.uint8)
    lib().orc_csm_to_binary_mutual(_d(D), D.shape[0], D.shape[1], float(kappa), _u(B))
    return B


# ---------------------------------------------------------------------------------------
# pySeqAlign.pyx mirrors (in-place D, rows first)
# ---------------------------------------------------------------------------------------
def _check(S, D):
    assert S.dtype == np.uint8 and S.ndim == 1 and S.flags.c_contiguous
    assert D.dtype == np.float32 and D.ndim == 1 and D.flags.c_contiguous


def qmax(S, D, M, N, impl="oracle"):
    """pySeqAlign.pyx:14 -> SequenceAlignment.c:113."""
    _check(S, D)
    fn = lib().orc_qmax if impl == "oracle" else ref_lib(impl).qmax_c
    return float(fn(_u(S), _f(D), int(M), int(N)))


def dmax(S, D, M, N, impl="oracle"):
    """pySeqAlign.pyx:21 -> SequenceAlignment.c:147."""
    _check(S, D)
    fn = lib().orc_dmax if impl == "oracle" else ref_lib(impl).dmax_c
    return float(fn(_u(S), _f(D), int(M), int(N)))


def swconstrained(S, D, N, M, impl="oracle"):
    """pySeqAlign.pyx:7 -> SequenceAlignment.c:73.  S is (N,M) flattened, D is (N+1)(M+1)."""
    _check(S, D)
    fn = lib().orc_swc if impl == "oracle" else ref_lib(impl).swalignimpconstrained
    return float(fn(_u(S), _f(D), int(N), int(M)))


# ---------------------------------------------------------------------------------------
# Serra09.py:166-175 chain
# ---------------------------------------------------------------------------------------
def serra09_pair(Xi, gi, Xj, gj, m=9, kappa=0.095, do_oti=True):
    """(qmax/(M+N), dmax/(M+N)) for one pair; Xi (ni,d) frames-major float64."""
    Xi = np.ascontiguousarray(Xi, dtype=np.float64)
    Xj = np.ascontiguousarray(Xj, dtype=np.float64)
    gi = np.ascontiguousarray(gi, dtype=np.float64)
    gj = np.ascontiguousarray(gj, dtype=np.float64)
    q = ctypes.c_double(0.0)
    dm = ctypes.c_double(0.0)
    rc = lib().orc_serra09_pair(_d(Xi), _d(gi), Xi.shape[0], _d(Xj), _d(gj), Xj.shape[0],
                                Xi.shape[1], m, float(kappa), int(bool(do_oti)),
                                ctypes.byref(q), ctypes.byref(dm))
    if rc != 0:
        raise ValueError("songs shorter than the embedding window")
    return q.value, dm.value


def serra09_pairs(feats, frame_off, gchroma, pairs, m=9, kappa=0.095, do_oti=True,
                  nthreads=1, want_dmax=True):
    """Chain over many pairs of a concatenated corpus.  Returns (qmax, dmax, threads_used);
    want_dmax=False skips the dmax recurrence (dmax is then all zeros)."""
    feats = np.ascontiguousarray(feats, dtype=np.float64)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
    gchroma = np.ascontiguousarray(gchroma, dtype=np.float64)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    K = pairs.shape[0]
    q = np.zeros(K)
    dm = np.zeros(K)
    used = lib().orc_serra09_pairs(
        _d(feats), frame_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _d(gchroma),
        pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), K, feats.shape[1], m,
        float(kappa), int(bool(do_oti)), int(nthreads), _d(q), _d(dm) if want_dmax else None)
    return q, dm, used
