"""
CPU restatement of the reference's FTM2D feature and similarity (benchmarking/FTM2D.py) -- TEST INFRASTRUCTURE,
like everything under oracle/: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product path (acoss_amd/).

Pinned: tests/test_oracle_golden.py checks every function here against tests/golden/ftm2d.npz, which
tests/golden/make_golden_ftm2d.py produced by calling the reference's own chrompwr / btchroma_to_fftmat and its
load_features / similarity expressions.  Not pinned by the reference: sync_median restates librosa.util.sync
(librosa >= 0.6, not installed here: "parity unpinned" for that one step), used at FTM2D.py:91.
"""
import numpy as np


def chrompwr(X, P=.5):
    """FTM2D.py:9-25: raise chroma columns to a power, preserving each column's norm.  X: (12, nbeats)."""
    nchr = X.shape[0]
    CMn = np.tile(np.sqrt(np.sum(X * X, axis=0)), (nchr, 1))
    CMn[CMn == 0] = 1
    CMp = np.power(X / CMn, P)
    CMpn = np.tile(np.sqrt(np.sum(CMp * CMp, axis=0)), (nchr, 1))
    CMpn[CMpn == 0] = 1.
    return CMn * (CMp / CMpn)


def btchroma_to_fftmat(btchroma, win=75):
    """FTM2D.py:29-48: |fft2| of every 12 x win patch, fftshift-ed and flattened row-major -> (12*win, nbeats-win+1)."""
    nchrm, nbeats = btchroma.shape
    assert nchrm == 12
    if nbeats < win:
        return None
    fftmat = np.zeros((nchrm * win, nbeats - win + 1))
    for i in range(nbeats - win + 1):
        F = np.abs(np.fft.fft2(btchroma[:, i:i + win]))
        fftmat[:, i] = np.fft.fftshift(F).flatten()
    return fftmat


def sync_median(data, onsets):
    """librosa.util.sync(data, onsets, aggregate=np.median) (FTM2D.py:91) for data (12, n): the median over the
    frames between consecutive boundaries of sorted(unique({0} U onsets U {n}))."""
    n = data.shape[-1]
    b = np.unique(np.concatenate([[0], np.clip(np.asarray(onsets, dtype=int), 0, n), [n]]))
    return np.stack([np.median(data[:, s:e], axis=1) for s, e in zip(b[:-1], b[1:])], axis=1)


def shingle_from_btchroma(btchroma, PWR=1.96, WIN=75, C=5):
    """FTM2D.py:92-100 from beat-synchronous chroma (12, nbeats); zeros(900) when there are too few beats (:89-90)."""
    if btchroma.shape[1] < WIN:
        return np.zeros(12 * WIN)
    chroma = chrompwr(btchroma, PWR)
    shingles = btchroma_to_fftmat(chroma, WIN).T
    Norm = np.sqrt(np.sum(shingles**2, 1))
    Norm[Norm == 0] = 1
    shingles = np.log(C * shingles / Norm[:, None] + 1)
    shingle = np.median(shingles, 0)
    return shingle / np.sqrt(np.sum(shingle**2))


def similarity(s1, s2):
    """FTM2D.py:122-126."""
    return np.exp(-np.sum((s1 - s2)**2))
