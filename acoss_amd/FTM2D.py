# -*- coding: utf-8 -*-
"""
Mirror of the reference's FTM2D plugin (benchmarking/FTM2D.py:51-127) on the MI355X path: same constructor
keywords, same cache prefix, `load_features(i)` returns the song's 900-number shingle, `similarity(idxs)` writes
exp(-|s_i - s_j|^2) into Ds['main'] -- but the shingles of all songs come from one GPU call
(engine.ftm2d_shingles: chrompwr, |fft2| of every 12 x 75 window, log, median, normalisation) and the
similarities of a call from one kernel; `all_pairwise` uses the all-pairs product on the matrix cores.

Host side, per song: the beat synchronisation (librosa.util.sync with the madmom onsets, FTM2D.py:91).  librosa is
not available here; `sync_median` restates its documented behaviour (median over the frames between consecutive
boundaries of {0} U onsets U {n}).
"""

import numpy as np

from .CoverAlgorithm import CoverAlgorithm
from . import engine

CHROMA_WIN = 75          # FTM2D.py:27


def sync_median(data, onsets):
    """librosa.util.sync(data, onsets, aggregate=np.median) for data (12, n)  (FTM2D.py:91)."""
    n = data.shape[-1]
    b = np.unique(np.concatenate([[0], np.clip(np.asarray(onsets, dtype=int), 0, n), [n]]))
    return np.stack([np.median(data[:, s:e], axis=1) for s, e in zip(b[:-1], b[1:])], axis=1)


class FTM2D(CoverAlgorithm):
    """
    Attributes
    ----------
    Same as CoverAlgorithms, plus
    shingles: {int: ndarray(WIN*chromabins)}
        A map from the song index to the FFT2DM shingles, so that they are cached
    chroma_type: string
        Type of chroma to use (key into features)
    """
    def __init__(self, datapath="../features_covers80", chroma_type='hpcp', shortname='Covers80', PWR=1.96, WIN=75, C=5,
                 do_memmaps=True, cachedir="cache"):
        if WIN != CHROMA_WIN:
            raise ValueError("the MI355X path implements the reference's window of 75 beats")
        self.PWR = PWR
        self.WIN = WIN
        self.C = C
        self.chroma_type = chroma_type
        self.shingles = {}
        self._dev = None          # (n_songs, 900) device tensor of all shingles
        CoverAlgorithm.__init__(self, "FTM2D", datapath=datapath, shortname=shortname, do_memmaps=do_memmaps,
                                cachedir=cachedir)

    def get_cacheprefix(self):
        """Return a descriptive file prefix to use for caching features and distance matrices (FTM2D.py:71-76)."""
        return "%s/%s_%s_%s" % (self.cachedir, self.name, self.shortname, self.chroma_type)

    # ------------------------------------------------------------------------------------------
    def beat_chroma(self, i):
        """Beat-synchronous chroma (12, nbeats) of song i, or None when there are not enough beats (FTM2D.py:85-91)."""
        if self.corpus is not None and getattr(self.corpus, "btchroma", None) is not None:
            return self.corpus.btchroma[i]
        feats = CoverAlgorithm.load_features(self, i)
        hpcp_orig = np.asarray(feats[self.chroma_type]).T
        onsets = np.asarray(feats['madmom_features']['onsets'] if 'madmom_features' in feats else feats['onsets'])
        if onsets.size > CHROMA_WIN:
            return sync_median(hpcp_orig, onsets)
        print("Warning: Not enough beats")
        return None

    def compute_all_shingles(self):
        """Every song's shingle in one GPU call."""
        if self._dev is None:
            bts = [self.beat_chroma(i) for i in range(self.N)]
            bts = [b if b is not None else np.zeros((12, 0)) for b in bts]
            self._dev = engine.ftm2d_shingles(bts, self.PWR, self.C)
            host = self._dev.cpu().numpy()
            for i in range(self.N):
                self.shingles[i] = host[i]
        return self._dev

    def load_features(self, i, do_plot=False):
        if i not in self.shingles:
            self.compute_all_shingles()
        return self.shingles[i]

    def similarity(self, idxs):
        idxs = np.asarray(idxs).reshape(-1, 2)
        sims = engine.ftm2d_pairs(self.compute_all_shingles(), idxs)
        if self.do_memmaps and len(idxs):
            self.Ds['main'][idxs[:, 0], idxs[:, 1]] = sims             # FTM2D.py:127
        return {'main': sims}

    def all_pairwise(self, parallel=0, n_cores=12, symmetric=False, precomputed=False, **kw):
        """All pairs as one N x 900 x N product (CoverAlgorithm.py:138-184 fills the same matrix pair by pair).
        Not sharded over ranks: at 15 000 songs the product is 0.4 TFLOP, milliseconds on one GPU."""
        if precomputed:
            return CoverAlgorithm.all_pairwise(self, parallel, n_cores, symmetric, precomputed, **kw)
        import time
        tic = time.time()
        G = engine.ftm2d_gram(self.compute_all_shingles()).cpu().numpy()
        if symmetric:
            # the reference scores i < j and mirrors (CoverAlgorithm.py:166, :180-182); the diagonal stays zero
            G = np.triu(G, 1)
            G = G + G.T
        else:
            np.fill_diagonal(G, 0.0)                               # permutations: no (i, i) pairs (:168)
        if not hasattr(self, "Ds"):
            self.Ds = {'main': np.zeros((self.N, self.N), dtype=np.float32)}
        self.Ds['main'][:, :] = G
        self.get_all_clique_ids()
        np.savez("%s_Ds.npz" % self.get_cacheprefix(), **{k: np.asarray(v) for k, v in self.Ds.items()})
        print("Elapsed Time All Pairwise: %.3g" % (time.time() - tic))


if __name__ == '__main__':
    from ._cli import run
    run(lambda a, mm: FTM2D(a.datapath, a.chroma_type, a.shortname, do_memmaps=mm),
        "Benchmarking with 2D Fourier Transform Magnitude Coefficients", "hpcp", "Covers80")
