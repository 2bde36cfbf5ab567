# -*- coding: utf-8 -*-
"""
Mirror of the reference's EarlySNF plugin (benchmarking/EarlySNF.py:11-97) on the MI355X path: same constructor
keywords, the same eight similarity-type keys, the same `similarity(idxs)` contract as Serra09 -- the K pairs of a
call go through the GPU in batches.

Per pair (EarlySNF.py:41-90): the Serra09 chains on chroma, MFCC and the SSM-scatter features, plus the fusion: block
affinity matrices [[W(SSM_A), W(CSM)], [W(CSM)^T, W(SSM_B)]] of the chroma (with the sliding window) and of the
SSM-scatter features (without), 3 cross-diffusion iterations (SimilarityFusion.snf_ws), and the Serra09 mask +
qmax / dmax on the negated cross block of the fused matrix.  As in the reference, MFCC stays out of the fusion (:64).
Songs need the precomputed 'ssms' features (the scattering transform itself is outside the accelerated path).
"""

import numpy as np

from .CoverAlgorithm import CoverAlgorithm
from .Serra09 import Serra09
from . import engine


class EarlySNF(Serra09):
    KEYS = ["chroma_qmax", "chroma_dmax", "mfcc_qmax", "mfcc_dmax", "ssms_scatter_qmax", "ssms_scatter_dmax",
            "snf_qmax", "snf_dmax"]

    def __init__(self, datapath="../features_covers80", chroma_type='crema', shortname='benchmark',
                 oti=True, kappa=0.095, m=9, downsample_fac=40, do_memmaps=True, cachedir="cache"):
        self.oti = oti
        self.m = m
        self.chroma_type = chroma_type
        self.kappa = kappa
        self.downsample_fac = downsample_fac
        self._dev = {}
        self._warned_ssms = False
        CoverAlgorithm.__init__(self, "EarlySNF", datapath=datapath, shortname=shortname, do_memmaps=do_memmaps,
                                similarity_types=list(self.KEYS), cachedir=cachedir)

    def similarity(self, idxs):
        idxs = np.asarray(idxs).reshape(-1, 2)
        K = idxs.shape[0]
        similarities = {key: np.zeros(K) for key in self.KEYS}
        if K == 0:
            return similarities
        have = self.load_features(int(idxs[0, 0]))
        res = self._chain('chroma', idxs, self.m, self.oti)                          # EarlySNF.py:45-61
        similarities['chroma_qmax'], similarities['chroma_dmax'] = res['qmax'], res['dmax']
        if 'mfcc' in have:                                                           # :63-69
            res = self._chain('mfcc', idxs, self.m, False)
            similarities['mfcc_qmax'], similarities['mfcc_dmax'] = res['qmax'], res['dmax']
        if 'ssms' not in have:
            raise KeyError("EarlySNF needs the 'ssms' features of every song (EarlySNF.py:72)")
        res = self._chain('ssms', idxs, 1, False)                                    # :71-81
        similarities['ssms_scatter_qmax'], similarities['ssms_scatter_dmax'] = res['qmax'], res['dmax']
        songs = [int(s) for s in np.unique(idxs)]                                    # :82-89
        chroma, where = self._device_corpus('chroma', songs)
        ssms, where2 = self._device_corpus('ssms', songs)
        if ssms.dtype != np.float64:
            ssms = engine.DeviceCorpus(ssms.feats.cpu().numpy().astype(np.float64), ssms.frame_off)
            self._dev['ssms'] = (ssms, where2)
        local = np.array([[where[int(a)], where[int(b)]] for a, b in idxs], dtype=np.int32)
        assert all(where[s] == where2[s] for s in songs)
        res = engine.early_snf_scores(chroma, ssms, local, m=self.m, kappa=self.kappa, do_oti=self.oti)
        similarities['snf_qmax'], similarities['snf_dmax'] = res['qmax'], res['dmax']
        if self.do_memmaps:
            for key in self.Ds.keys():
                self.Ds[key][idxs[:, 0], idxs[:, 1]] = similarities[key]            # :93-95
        return similarities


if __name__ == '__main__':
    from ._cli import run
    run(lambda a, mm: EarlySNF(a.datapath, a.chroma_type, a.shortname, do_memmaps=mm),
        "Benchmarking with early fusion + QMax/DMax", "crema", "covers80")
