"""
GPU-backed mirror of the reference's benchmarking/CRPUtils.py for the functions on the scoring
hot path: same names, argument meaning and return types (numpy in, numpy out), each one a thin
host wrapper over one batched stage kernel (a batch of one pair here; Serra09.similarity uses
the batched forms directly).

get_ssm, get_csm_cosine and sliding_window (off the Serra09 path; EarlySNF.py:74-75 and the other plugins import them) are
thin forms over the same kernel / plain data movement, so that `from CRPUtils import *` against this module finds every name
the reference's module defines.
"""
import numpy as np

from . import engine


def _two_song_corpus(X, Y):
    X = np.ascontiguousarray(X)
    Y = np.ascontiguousarray(Y)
    if X.ndim != 2 or Y.ndim != 2 or X.shape[1] != Y.shape[1]:
        raise ValueError("X and Y must be (M, d) and (N, d)")
    if X.shape[0] < 1 or Y.shape[0] < 1:
        raise ValueError("empty point cloud")
    dtype = np.float32 if (X.dtype == np.float32 and Y.dtype == np.float32) else np.float64
    feats = np.concatenate([X.astype(dtype, copy=False), Y.astype(dtype, copy=False)], axis=0)
    off = np.array([0, X.shape[0], X.shape[0] + Y.shape[0]], dtype=np.int64)
    return engine.DeviceCorpus(feats, off)


def _matrix_batch(D, win):
    """A one-pair batch describing an existing (M, N) matrix as a 'CSM' (win=1: CSM == CRP)."""
    M, N = D.shape
    off = np.array([0, M, M + N], dtype=np.int64)
    return engine.PairBatch(off, np.array([[0, 1]], dtype=np.int32), win, "cuda:%d" % engine.torch.cuda.current_device(),
                            pitch_align=1)


def get_oti(C1, C2, do_plot=False):
    """CRPUtils.py:109-136: rotation of C1 that best matches C2 (first maximum wins)."""
    C1 = np.asarray(C1, dtype=np.float64).reshape(-1)
    C2 = np.asarray(C2, dtype=np.float64).reshape(-1)
    engine.require_gpu()
    g = np.stack([C1, C2])
    corpus = engine.DeviceCorpus(np.zeros((2, len(C1))), np.array([0, 1, 2], dtype=np.int64), gchroma=g)
    batch = engine.PairBatch(corpus.frame_off, np.array([[0, 1]], dtype=np.int32), 1, corpus.device)
    engine.oti(corpus, batch)
    return int(batch.fetch_shifts()[0])


def get_csm(X, Y, shift=0):
    """CRPUtils.py:67-84: Euclidean cross-similarity of X (M, d) and Y (N, d); dtype follows the
    inputs.  `shift` (extension) rotates X's bins first, as Serra09.py:167 does before the call."""
    corpus = _two_song_corpus(X, Y)
    batch = engine.PairBatch(corpus.frame_off, np.array([[0, 1]], dtype=np.int32), 1, corpus.device, pitch_align=1)
    if shift:
        batch.set_shifts([int(shift) % corpus.d])
    C = engine.csm(corpus, batch)
    return C[:batch.total_csm].cpu().numpy().reshape(X.shape[0], Y.shape[0])


get_csm_euclidean = get_csm


def get_ssm(X):
    """CRPUtils.py:48-65: Euclidean self-similarity matrix, zero diagonal (the reference sets it explicitly, :63)."""
    D = get_csm(X, X)
    np.fill_diagonal(D, 0)
    return D


def get_csm_cosine(X, Y):
    """CRPUtils.py:88-107: 1 - cosine of every pair of rows; rows of zero norm count as unit vectors of nothing (their distance
    to everything is 1, :101-104).  On the normalised clouds 1 - x.y = |x - y|^2 / 2: the Euclidean kernel does the products."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    xn, yn = np.sqrt(np.sum(X ** 2, 1)), np.sqrt(np.sum(Y ** 2, 1))
    zx, zy = xn == 0, yn == 0
    xn[zx] = 1
    yn[zy] = 1
    C = get_csm(X / xn[:, None], Y / yn[:, None])
    D = 0.5 * C * C
    D[zx, :] = 1.0
    D[:, zy] = 1.0
    return D


def sliding_window(X, win):
    """CRPUtils.py:8-22: delay embedding of a point cloud, (N - win + 1, d * win) float64 (data movement only)."""
    X = np.asarray(X)
    M = X.shape[0] - win + 1
    Y = np.zeros((max(M, 0), X.shape[1] * win))
    for i in range(win):
        Y[:, i * X.shape[1]:(i + 1) * X.shape[1]] = X[i:i + M, :]
    return Y


def sliding_csm(D, win):
    """CRPUtils.py:24-45: delay-embedding effect on a CSM by summing squares along diagonals."""
    D = np.ascontiguousarray(D)
    if D.dtype != np.float32:
        D = D.astype(np.float64, copy=False)
    M, N = D.shape[0] - win + 1, D.shape[1] - win + 1
    if M < 1 or N < 1:
        return np.zeros((max(M, 0), max(N, 0)))
    engine.require_gpu()
    batch = _matrix_batch(D, win)
    Dd = engine.torch.from_numpy(D.reshape(-1)).to(batch.device)
    S = engine.sliding(Dd, batch)
    return S[:batch.total_crp].cpu().numpy().reshape(M, N)


def _binarize(D, kappa, mutual):
    D = np.ascontiguousarray(D, dtype=np.float64)
    if kappa == 0:
        return np.ones_like(D)                      # CRPUtils.py:188-189 (float matrix, as the reference)
    engine.require_gpu()
    batch = _matrix_batch(D, 1)
    Dd = engine.torch.from_numpy(D.reshape(-1)).to(batch.device)
    B = engine.binarize(Dd, batch, kappa, mutual=mutual)
    return B[:batch.total_crp].cpu().numpy().reshape(D.shape)


def csm_to_binary(D, kappa):
    """CRPUtils.py:169-199."""
    return _binarize(D, kappa, False)


def csm_to_binary_mutual(D, kappa):
    """CRPUtils.py:201-219."""
    return _binarize(D, kappa, True)


def get_csm_blocked_oti(X, Y, C1, C2, csm_fn=None):
    """CRPUtils.py:138-166 with csm_fn = get_csm_euclidean (the ChenFusion.py:59 use): every
    n_chroma_bins-wide block of X is rotated by the global OTI, then the Euclidean CSM."""
    if csm_fn is not None and csm_fn is not get_csm and csm_fn is not get_csm_euclidean:
        raise NotImplementedError("only the Euclidean CSM is on the accelerated path")
    nb = len(C1)
    blocks = int(X.shape[1] / nb)
    oti = get_oti(C1, C2)
    X1 = np.reshape(X, (X.shape[0], blocks, nb))
    X1 = np.roll(X1, oti, axis=2)                   # data movement only; arithmetic is in get_csm
    X1 = np.reshape(X1, [X.shape[0], blocks * nb])
    return get_csm(X1, Y)
