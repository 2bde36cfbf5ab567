"""
Loop-for-loop restatement of CoverAlgorithm.getEvalStatistics (CoverAlgorithm.py:330-418),
minus the prints and the CSV append.

TEST INFRASTRUCTURE ONLY (see oracle/acoss_oracle.h).  Pure-Python O(N^2) loops: small N only.
Parity status: pinned -- tests/golden/evalstats_*.npz hold the reference's own return values.
"""
import warnings
from itertools import chain

import numpy as np


def get_eval_statistics(Dmat, cliques, topsidx=(1, 10, 100, 1000)):
    """
    Parameters
    ----------
    Dmat: ndarray(N, N)
        Pairwise similarity scores (higher = more similar), as Ds[similarity_type]
    cliques: {label: set(int)}
        As CoverAlgorithm.cliques (insertion order matters: CoverAlgorithm.py:345)
    Returns
    -------
    (MR, MRR, MDR, MAP, tops)
    """
    D = np.array(Dmat, dtype=np.float32)                      # :340
    N = D.shape[0]
    groups = [list(cliques[s]) for s in cliques]              # :345
    Ks = np.array([len(c) for c in groups])                   # :346
    order = np.argsort(-Ks)                                   # :349
    Ks = Ks[order]
    groups = [groups[i] for i in order]
    perm = np.array(list(chain(*groups)), dtype=int)          # :355
    D = D[perm, :]
    D = D[:, perm]
    np.fill_diagonal(D, -np.inf)                              # :361
    idx = np.argsort(-D, 1)                                   # :362
    ranks = np.nan * np.ones(N)
    AllMap = np.nan * np.ones(N)
    startidx = 0
    kidx = 0
    for i in range(N):                                        # :367
        if i >= startidx + Ks[kidx]:
            startidx += Ks[kidx]
            kidx += 1
            if Ks[kidx] < 2:
                break
        iranks = []
        for k in range(N):
            diff = idx[i, k] - startidx
            if diff >= 0 and diff < Ks[kidx]:
                iranks.append(k + 1)
        iranks = iranks[0:-1]                                 # :381 drop the song itself
        if len(iranks) == 0:
            warnings.warn("Recalling 0 songs for clique of size %i at song index %i" % (Ks[kidx], i))
            break
        ranks[i] = iranks[0]
        P = np.array([float(j) / float(r) for (j, r) in zip(range(1, Ks[kidx]), iranks)])
        AllMap[i] = np.mean(P)
    MAP = np.nanmean(AllMap)                                  # :391
    ranks = ranks[np.isnan(ranks) == 0]
    MR = np.mean(ranks)
    MRR = 1.0 / N * (np.sum(1.0 / ranks))                     # :395 divides by ALL songs
    MDR = np.median(ranks)
    tops = np.zeros(len(topsidx))
    for t in range(len(tops)):
        tops[t] = np.sum(ranks <= topsidx[t])
    return (MR, MRR, MDR, MAP, tops)
