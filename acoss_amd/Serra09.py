# -*- coding: utf-8 -*-
"""
Mirror of the reference's Serra09 plugin (benchmarking/Serra09.py:73-196) on the MI355X path:
same constructor keywords, same six similarity-type keys, same `similarity(idxs)` contract
(returns {key: float64 ndarray(K)} and writes Ds[key][i][j] when do_memmaps) -- but the K pairs of
a call go through the GPU as ONE batch instead of a Python loop over pairs (Serra09.py:161).

Per pair (Serra09.py:166-175): oti -> roll -> get_csm -> sliding_csm(m) -> csm_to_binary_mutual
(kappa) -> qmax/(M+N) and dmax/(M+N), where dmax runs on the D that qmax just filled (the
reference does not re-zero it; the kernels reproduce that boundary).

Feature preparation (Serra09.py:96-156) keeps the reference's per-song semantics for chroma and
MFCC (global chroma on the full-resolution chroma, median / mean aggregation over blocks of
`downsample_fac` frames with librosa.util.sync's block boundaries).  The scattering-transform SSM
features (kymatio, skimage) are outside the hot path's scope: songs that carry a precomputed
'ssms' array get ssms_scatter_* scores, others get zeros for those two keys.
"""
import warnings

import numpy as np

from .CoverAlgorithm import CoverAlgorithm
from . import engine


def global_chroma(chroma):
    """Computes global chroma of a input chroma vector (Serra09.py:24-28)."""
    if chroma.shape[1] not in [12, 24, 36]:
        raise IOError("Wrong axis for the input chroma array. Expected shape '(frame_size, bin_size)'")
    return np.divide(chroma.sum(axis=0), np.max(chroma.sum(axis=0)))


def block_aggregate(data, fac, aggregate):
    """librosa.util.sync(data, np.arange(0, n, fac), aggregate=...) along the last axis: blocks
    [0,fac), [fac,2fac), ..., the last one possibly shorter (Serra09.py:104,110)."""
    n = data.shape[-1]
    bounds = list(range(0, n, fac)) + [n]
    return np.stack([aggregate(data[..., a:b], axis=-1) for a, b in zip(bounds[:-1], bounds[1:])], axis=-1)


class Serra09(CoverAlgorithm):
    """
    Attributes
    ----------
    Same as CoverAlgorithm, plus
    chroma_type: string
        Type of chroma to use (key into features)
    downsample_fac: int
        The factor by which to downsample the HPCPs with median aggregation
    all_feats: {int: dictionary}
        Cached features
    """
    KEYS = ["ssms_scatter_qmax", "ssms_scatter_dmax", "chroma_qmax", "chroma_dmax", "mfcc_qmax", "mfcc_dmax"]
    alignments = ("qmax", "dmax")          # (subclasses that keep the reference's constructor: the reference's pair)

    def __init__(self, datapath="../features_covers80", chroma_type='crema', shortname='benchmark',
                 oti=True, kappa=0.095, m=9, downsample_fac=40, do_memmaps=True, cachedir="cache",
                 alignments=("qmax", "dmax")):
        """The reference's keywords (Serra09.py:86-94) plus `alignments`: the recurrences run on every mask.  The default is
        the reference's pair (qmax, then dmax on the same D).  "swc" adds SequenceAlignment.c's constrained Smith-Waterman
        (BASELINE config 3, "Serra09 Smith-Waterman constrained") on the same mutual mask, called as its one caller in the
        reference does (EarlySNF_Old.py:198-203: a fresh (M+1) x (N+1) D, rows first) and normalised like the other two:
        three more keys, `chroma_swc`, `mfcc_swc`, `ssms_scatter_swc` = swconstrained / (M + N)."""
        bad = [a for a in alignments if a not in ("qmax", "dmax", "swc")]
        if bad or "qmax" not in alignments or "dmax" not in alignments:
            raise ValueError("Serra09: alignments must hold 'qmax' and 'dmax' (the reference's keys) and may add 'swc'; got %r" % (alignments,))
        self.alignments = tuple(a for a in ("qmax", "dmax", "swc") if a in alignments)
        self.oti = oti
        self.m = m
        self.chroma_type = chroma_type
        self.kappa = kappa
        self.downsample_fac = downsample_fac
        self._dev = {}
        self._warned_ssms = False
        keys = list(self.KEYS)
        if "swc" in self.alignments:
            keys += ["ssms_scatter_swc", "chroma_swc", "mfcc_swc"]
        CoverAlgorithm.__init__(self, "Serra09", datapath=datapath, shortname=shortname, do_memmaps=do_memmaps,
                                similarity_types=keys, cachedir=cachedir)

    # ------------------------------------------------------------------------------------------
    def load_features(self, i):
        if i not in self.all_feats:
            feats = CoverAlgorithm.load_features(self, i)
            if self.corpus is not None:
                # synthetic corpora are generated at the aggregated frame rate already
                chroma = np.ascontiguousarray(self.corpus.song(i).T)
                out = {'gchroma': self.corpus.gchroma[i], 'chroma': chroma}
                mfcc = getattr(self.corpus, "mfcc", None)
                if mfcc is not None:
                    out['mfcc'] = mfcc[i]
                ssms = getattr(self.corpus, "ssms", None)           # precomputed (n - m + 1, d) scattering features
                if ssms is not None:
                    out['ssms'] = ssms[i]
                self.all_feats[i] = out
                return out
            chroma = feats[self.chroma_type]
            gchroma = global_chroma(chroma)                                     # Serra09.py:102
            chroma = block_aggregate(chroma.T, self.downsample_fac, np.median)  # :104 -> (12, n)
            out = {'gchroma': gchroma}
            if 'mfcc_htk' in feats:
                mfcc_orig = np.array(feats['mfcc_htk'])
                mfcc_orig[np.isnan(mfcc_orig)] = 0                              # :108-109
                mfcc_orig[np.isinf(mfcc_orig)] = 0
                mfcc = block_aggregate(mfcc_orig, self.downsample_fac, np.mean)  # :110 -> (13, n)
                N = min(chroma.shape[1], mfcc.shape[1])                         # :111
                chroma = chroma[:, 0:N]
                out['mfcc'] = mfcc[:, 0:N]
            out['chroma'] = chroma
            if 'ssms' in feats:
                M = chroma.shape[1] - self.m + 1
                ssms = np.array(feats['ssms'])
                if ssms.shape[0] < M:                                           # :147-151
                    pad = np.zeros((M, ssms.shape[1]), dtype=ssms.dtype)
                    pad[0:ssms.shape[0], :] = ssms
                    pad[ssms.shape[0]::, :] = ssms[-1, :]
                    ssms = pad
                out['ssms'] = ssms[0:M, :]                                      # :152
            self.all_feats[i] = out
        return self.all_feats[i]

    # ------------------------------------------------------------------------------------------
    def _device_corpus(self, key, songs):
        """Frames-major device copy of feature `key` for the given songs (cached while the song set
        does not grow).  Returns (DeviceCorpus, position of every song in it or -1: int32 array of N)."""
        cached = self._dev.get(key)
        if cached is not None and np.all(cached[1][songs] >= 0):
            return cached
        have = np.flatnonzero(cached[1] >= 0) if cached is not None else np.zeros(0, dtype=np.int64)
        songs = np.union1d(songs, have)
        mats, g = [], []
        for s in songs:
            f = self.load_features(int(s))
            x = f[key]
            mats.append(np.ascontiguousarray(x.T if key != 'ssms' else x))      # (n, d) frames-major
            if key == 'chroma':
                g.append(f['gchroma'])
        dtype = np.float32 if all(mm.dtype == np.float32 for mm in mats) else np.float64
        feats = np.concatenate([mm.astype(dtype, copy=False) for mm in mats], axis=0)
        off = np.zeros(len(mats) + 1, dtype=np.int64)
        off[1:] = np.cumsum([mm.shape[0] for mm in mats])
        corpus = engine.DeviceCorpus(feats, off, gchroma=np.stack(g) if g else None)
        where = np.full(max(self.N, int(songs.max()) + 1), -1, dtype=np.int32)
        where[songs] = np.arange(len(songs), dtype=np.int32)
        self._dev[key] = (corpus, where)
        return self._dev[key]

    def _chain(self, key, idxs, win, do_oti):
        corpus, where = self._device_corpus(key, np.unique(idxs).astype(np.int64))
        return engine.serra09_scores(corpus, where[idxs], m=win, kappa=self.kappa, do_oti=do_oti, want=self.alignments)

    def similarity(self, idxs):
        idxs = np.asarray(idxs).reshape(-1, 2)
        K = idxs.shape[0]
        similarities = {key: np.zeros(K) for key in self.similarity_types}
        if K == 0:
            return similarities
        have = self.load_features(int(idxs[0, 0]))

        def take(prefix, res):
            for a in self.alignments:
                similarities["%s_%s" % (prefix, a)] = res[a]
        # Step 1: chroma (OTI)                                   Serra09.py:165-175
        take('chroma', self._chain('chroma', idxs, self.m, self.oti))
        # Step 2: MFCC (no OTI)                                  Serra09.py:177-184
        if 'mfcc' in have:
            take('mfcc', self._chain('mfcc', idxs, self.m, False))
        # Step 3: SSM-scatter features, no sliding window        Serra09.py:186-192
        if 'ssms' in have:
            take('ssms_scatter', self._chain('ssms', idxs, 1, False))
        elif not self._warned_ssms:
            warnings.warn("no 'ssms' features: ssms_scatter_* scores are left at zero "
                          "(scattering features are outside the accelerated path)")
            self._warned_ssms = True
        if self.do_memmaps:
            for key in self.Ds.keys():
                self.Ds[key][idxs[:, 0], idxs[:, 1]] = similarities[key]       # Serra09.py:193-195
        return similarities

    def _pair_costs(self, pairs):
        lens = np.array([self.load_features(i)['chroma'].shape[1] for i in range(self.N)], dtype=np.int64)
        return (lens[pairs[:, 0]] - self.m + 1) * (lens[pairs[:, 1]] - self.m + 1)


if __name__ == '__main__':
    from ._cli import run
    run(lambda a, mm: Serra09(a.datapath, a.chroma_type, a.shortname, do_memmaps=mm),
        "Benchmarking with Joan Serra's Cover id algorithm", "crema", "covers80", batch_options=True)
