"""
Seeded synthetic chroma corpora shaped like the BASELINE.json configs.

There is no dataset in the container or on the GPU box, so bench.py, the smoke check and the
parity tests all draw from this generator.  A corpus is a set of cover cliques: every clique has
a base "song" (a chain of chord segments) and each version of it is transposed, tempo-warped and
re-noised, which exercises the optimal-transposition index and produces the diagonal structures
the alignment scores respond to.  Frames are max-normalised like essentia HPCP, values in
[0, 1], no all-zero frames and no exactly repeated frames (so the kappa-nearest-neighbour
selections are tie-free, the condition under which mask parity with the reference is defined).
"""
import numpy as np


class Corpus(object):
    """
    Attributes
    ----------
    feats: ndarray(total_frames, nbins) float64
        All songs' frames, concatenated, frames-major (song i is feats[frame_off[i]:frame_off[i+1]],
        i.e. the transpose of the reference's (12, n) 'chroma' feature, Serra09.py:154)
    frame_off: ndarray(n_songs+1) int64
    gchroma: ndarray(n_songs, nbins) float64
        Global chroma per song (Serra09.py:24-28)
    labels: list(str)
        Clique label per song (the 'label' field of the reference's h5 files)
    """

    def __init__(self, feats, frame_off, gchroma, labels):
        self.feats = feats
        self.frame_off = frame_off
        self.gchroma = gchroma
        self.labels = labels

    @property
    def n_songs(self):
        return len(self.labels)

    def song(self, i):
        return self.feats[self.frame_off[i]:self.frame_off[i + 1]]

    def cliques(self):
        out = {}
        for i, lab in enumerate(self.labels):
            out.setdefault(lab, set()).add(i)
        return out


HARD_NOISE, HARD_TEMPO = 1.5, (0.5, 2.0)          # config2_hard(): set by the search recorded in tests/golden/README.md


def _global_chroma(chroma):
    s = chroma.sum(axis=0)
    return np.divide(s, np.max(s))


def _ar1(rng, shape, scale, rho):
    """Noise with the marginal spread of U(0, scale) that is correlated from frame to frame: an AR(1) sequence per bin,
    e_t = rho e_{t-1} + sqrt(1 - rho^2) u_t with u_t = U(0, scale) - scale / 2, shifted back to mean scale / 2.  rho = 0: U(0, scale) itself."""
    u = rng.uniform(0.0, scale, size=shape)
    if rho <= 0.0:
        return u
    from scipy.signal import lfilter
    u -= 0.5 * scale
    g = np.sqrt(1.0 - rho * rho)
    u[0] /= g                                            # so that e_0 = u_0
    return lfilter([g], [1.0, -rho], u, axis=0) + 0.5 * scale


def _base_song(rng, n_frames, nbins, rho=0.0):
    base = np.zeros((n_frames, nbins))
    pos = 0
    while pos < n_frames:
        seg = int(rng.integers(4, 17))
        n_active = int(rng.integers(3, 5))
        bins = rng.choice(nbins, size=n_active, replace=False)
        template = np.zeros(nbins)
        template[bins] = rng.uniform(0.5, 1.0, size=n_active)
        base[pos:pos + seg] = template[None, :]
        pos += seg
    base += np.clip(_ar1(rng, base.shape, 0.15, rho), 0.0, None)
    return base


def _version(rng, base, n_frames, noise=0.2, tempo_range=(0.8, 1.25), rho=0.0):
    nbins = base.shape[1]
    shift = int(rng.integers(0, nbins))
    tempo = rng.uniform(tempo_range[0], tempo_range[1])
    # linear time-resampling of the base by the tempo factor, wrapped to n_frames
    src = (np.arange(n_frames) * tempo) % (base.shape[0] - 1)
    lo = np.floor(src).astype(int)
    frac = (src - lo)[:, None]
    x = (1.0 - frac) * base[lo] + frac * base[lo + 1]
    x = np.roll(x, shift, axis=1)
    x = x + _ar1(rng, x.shape, noise, rho)
    x = np.clip(x, 0.0, None)
    x = x / np.max(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float64)


def make_corpus(n_cliques, versions, n_frames=1000, nbins=12, seed=20260, lengths=None,
                singletons=0, noise=0.2, tempo_range=(0.8, 1.25), rho=0.0):
    """
    Parameters
    ----------
    n_cliques, versions: int
        n_cliques cliques of `versions` songs each, then `singletons` one-song cliques
    n_frames: int
        Frames per song when `lengths` is None
    lengths: callable(rng) -> int, optional
        Per-song length draw for ragged corpora
    noise, tempo_range: float, (float, float)
        Per-version additive noise U(0, noise) and tempo factor U(tempo_range): the defaults are the BASELINE configs'
        (SURVEY.md section 8d); config2_hard() raises them until retrieval is no longer perfect
    """
    rng = np.random.default_rng(seed)
    songs, labels = [], []
    sizes = [versions] * n_cliques + [1] * singletons
    for c, size in enumerate(sizes):
        base = _base_song(rng, 1000 if lengths is not None else max(n_frames, 32), nbins, rho)
        for _ in range(size):
            n = int(lengths(rng)) if lengths is not None else n_frames
            songs.append(_version(rng, base, n, noise, tempo_range, rho))
            labels.append("clique_%05d" % c)
    frame_off = np.zeros(len(songs) + 1, dtype=np.int64)
    frame_off[1:] = np.cumsum([s.shape[0] for s in songs])
    feats = np.ascontiguousarray(np.concatenate(songs, axis=0))
    gchroma = np.ascontiguousarray(np.stack([_global_chroma(s) for s in songs]))
    return Corpus(feats, frame_off, gchroma, labels)


def config2(n_songs=1000, n_frames=1000, seed=20260):
    """BASELINE config 2: 1k songs x 1000-frame 12-bin HPCP, 250 cliques x 4."""
    return make_corpus(n_songs // 4, 4, n_frames=n_frames, seed=seed)


def config2_smooth(n_songs=1000, n_frames=1000, seed=20262, rho=0.9):
    """Config 2's shape with TEMPORALLY SMOOTH frames: the per-frame noise of the base songs and of the versions is an AR(1)
    sequence (rho = 0.9) instead of independent draws -- what real HPCP / crema frames look like from one frame to the next.
    (rho > 0 changes the random stream, so this is its own corpus, not config2() smoothed.)"""
    return make_corpus(n_songs // 4, 4, n_frames=n_frames, seed=seed, rho=rho)


def config2_hard(n_songs=64, n_frames=1000, seed=20261, noise=HARD_NOISE, tempo_range=HARD_TEMPO):
    """A config-2-shaped slice (cliques of 4, 1000 frames x 12 bins) whose covers are HARD: heavier per-version noise and a
    wider tempo spread, chosen so that the reference's MAP lands between 0.6 and 0.9 and the evaluation statistics
    discriminate (with config 2's own settings MAP is 1.0).  tests/golden/config2_hard64.npz holds the reference's scores."""
    return make_corpus(n_songs // 4, 4, n_frames=n_frames, seed=seed, noise=noise, tempo_range=tempo_range)


def config1(seed=80):
    """covers80-shaped: 160 songs = 80 cliques x 2, ragged lengths U{300..700}."""
    return make_corpus(80, 2, seed=seed, lengths=lambda r: r.integers(300, 701))


def config3(n_cliques=1000, singletons=2000, seed=15000):
    """DA-TACOS benchmark_subset-shaped: cliques of 13 + singletons, lengths ~N(520,120) in [200,1200]."""
    return make_corpus(n_cliques, 13, seed=seed, singletons=singletons,
                       lengths=lambda r: int(np.clip(r.normal(520, 120), 200, 1200)))


def all_pairs(n_songs):
    """Upper-triangular pair list in the order of itertools.combinations (CoverAlgorithm.py:166)."""
    i, j = np.triu_indices(n_songs, k=1)
    return np.ascontiguousarray(np.stack([i, j], axis=1).astype(np.int32))
