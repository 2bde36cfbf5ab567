"""
CPU restatement of the reference's similarity network fusion (benchmarking/SimilarityFusion.py) and of the
EarlySNF per-pair chain (benchmarking/EarlySNF.py:41-90) -- TEST INFRASTRUCTURE, like everything under oracle/:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path.

Pinned: tests/test_oracle_golden.py checks every function here against tests/golden/snf.npz, which
tests/golden/make_golden_snf.py produced by calling the reference's SimilarityFusion / CRPUtils functions and its
compiled SequenceAlignment.c.
"""
import numpy as np

from . import oracle as orc


def _k_smallest_mean(D, k, axis):
    """np.mean(np.partition(D, k, axis)[first k along axis], axis): the mean of the k smallest entries."""
    part = np.partition(D, k, axis)
    part = part[:, 0:k] if axis == 1 else part[0:k, :]
    return np.mean(part, axis)


def get_W(D, K, Mu=0.5):
    """SimilarityFusion.py:50-73: affinity matrix of a self-similarity matrix."""
    DSym = 0.5 * (D + D.T)
    np.fill_diagonal(DSym, 0)
    MeanDist = _k_smallest_mean(DSym, K + 1, 1) * float(K + 1) / float(K)      # :60-61 (the zero diagonal is among the K+1)
    Eps = (MeanDist[:, None] + MeanDist[None, :] + DSym) / 3
    Denom = 2 * (Mu * Eps) ** 2
    Denom[Denom == 0] = 1
    return np.exp(-DSym ** 2 / Denom)


def get_WCSM(CSMAB, k1, k2, Mu=0.5):
    """SimilarityFusion.py:76-92: affinity block of a cross-similarity matrix."""
    MeanDist1 = _k_smallest_mean(CSMAB, k2, 1)
    MeanDist2 = _k_smallest_mean(CSMAB, k1, 0)
    Eps = (MeanDist1[:, None] + MeanDist2[None, :] + CSMAB) / 3
    return np.exp(-CSMAB ** 2 / (2 * (Mu * Eps) ** 2))


def get_WCSMSSM(SSMA, SSMB, CSMAB, K, Mu=0.5):
    """SimilarityFusion.py:94-134: [[W(SSMA), W(CSM)], [W(CSM)^T, W(SSMB)]] with the neighbours split by size."""
    M, N = SSMA.shape[0], SSMB.shape[0]
    k1 = int(K * float(M) / (M + N))
    k2 = K - k1
    W = np.zeros((N + M, N + M))
    WC = get_WCSM(CSMAB, k1, k2, Mu)
    W[0:M, 0:M] = get_W(SSMA, k1, Mu)
    W[0:M, M::] = WC
    W[M::, 0:M] = WC.T
    W[M::, M::] = get_W(SSMB, k2, Mu)
    return W


def get_P(W, reg_diag=False):
    """SimilarityFusion.py:136-157."""
    if reg_diag:
        WNoDiag = np.array(W)
        np.fill_diagonal(WNoDiag, 0)
        RowSum = np.sum(WNoDiag, 1)
        RowSum[RowSum == 0] = 1
        return 0.5 * np.eye(W.shape[0]) + 0.5 * WNoDiag / RowSum[:, None]
    RowSum = np.sum(W, 1)
    RowSum[RowSum == 0] = 1
    return W / RowSum[:, None]


def get_S(W, K):
    """SimilarityFusion.py:159-180 as a dense matrix: every row keeps its K largest entries, L1-normalised."""
    N = W.shape[0]
    J = np.argpartition(-W, K, 1)[:, 0:K]
    I = np.tile(np.arange(N)[:, None], (1, K))
    V = W[I, J]
    SNorm = np.sum(V, 1)
    SNorm[SNorm == 0] = 1
    S = np.zeros((N, N))
    S[I, J] = V / SNorm[:, None]
    return S


def snf_ws(Ws, K=5, niters=20, reg_diag=True):
    """SimilarityFusion.py:207-277.  From the second iteration on `Pts` and `nextPts` are the same list (:270), so
    matrix i is updated from the already-updated matrices k < i of the same iteration; kept as is."""
    Ps = [get_P(W, reg_diag) for W in Ws]
    Ss = [get_S(W, K) for W in Ws]
    Pts = [np.array(P) for P in Ps]
    nextPts = [np.zeros(P.shape) for P in Pts]
    N = len(Pts)
    for it in range(niters):
        for i in range(N):
            nextPts[i] *= 0
            for k in range(N):
                if i == k:
                    continue
                nextPts[i] += Pts[k]
            nextPts[i] /= float(N - 1)
            A = Ss[i].dot(nextPts[i].T)
            nextPts[i] = Ss[i].dot(A.T)
            if reg_diag:
                PNoDiag = np.array(nextPts[i])
                np.fill_diagonal(PNoDiag, 0)
                RowSum = np.sum(PNoDiag, 1)
                RowSum[RowSum == 0] = 1
                nextPts[i] = 0.5 * np.eye(PNoDiag.shape[0]) + 0.5 * PNoDiag / RowSum[:, None]
        Pts = nextPts
    Fused = np.zeros(Pts[0].shape)
    for Pt in Pts:
        Fused += Pt
    return Fused / len(Pts)


def early_snf_pair(Si, Sj, m=9, kappa=0.095, keep=None):
    """EarlySNF.py:41-90 for one pair: (snf_qmax, snf_dmax); S* = {'gchroma', 'chroma' (12, n), 'ssms' (n-m+1, d)}."""
    oti = orc.get_oti(Si['gchroma'], Sj['gchroma'])
    X = np.ascontiguousarray(Si['chroma'].T)
    Y = np.ascontiguousarray(Sj['chroma'].T)
    Xr = np.ascontiguousarray(np.roll(Si['chroma'], oti, axis=0).T)
    csm = orc.sliding_csm(orc.get_csm(X, Y, oti), m)
    M, N = csm.shape
    K = int(kappa * (M + N))
    ssma = orc.sliding_csm(orc.get_csm(Xr, Xr), m)
    ssmb = orc.sliding_csm(orc.get_csm(Y, Y), m)
    Ws = [get_WCSMSSM(ssma, ssmb, csm, K)]
    A, B = np.ascontiguousarray(Si['ssms'], dtype=np.float64), np.ascontiguousarray(Sj['ssms'], dtype=np.float64)
    sa, sb = orc.get_csm(A, A), orc.get_csm(B, B)
    np.fill_diagonal(sa, 0)            # CRPUtils.py:64
    np.fill_diagonal(sb, 0)
    Ws.append(get_WCSMSSM(sa, sb, orc.get_csm(A, B), K))
    fused = snf_ws(Ws, K=K, niters=3, reg_diag=True)
    cross = np.ascontiguousarray(-fused[0:M, M::])
    Bm = orc.csm_to_binary_mutual(cross, kappa)
    Bf = np.ascontiguousarray(Bm.flatten())
    D = np.zeros(M * N, dtype=np.float32)
    q = orc.qmax(Bf, D, M, N) / (M + N)
    d = orc.dmax(Bf, D, M, N) / (M + N)
    if keep is not None:
        keep.update(csm=csm, ssma=ssma, ssmb=ssmb, W0=Ws[0], W1=Ws[1], fused=fused, B=Bm, K=K)
    return q, d
