"""The float32 filter with 16-bit keys (csrc/keys16.h, csrc/keys16_kernels.hip, the row-band strip kernel's OUT = 1 form):
the key plane against its definition, and the masks against the float64 path's bit for bit -- golden 1000-frame pairs,
ragged pairs with every pitch alignment, kappa conventions, crafted exact ties and 1e-11 perturbations (everything in
reach of the error band: float64 refinement), a worthless approximation (whole rows in reach: the general refinement),
13-dimensional features without OTI, thresholds outside the key range (clamped keys: handed over), a side buffer too
small for the batch (the strided refinement kernels), corpora scaled by 1e+-25."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from acoss_amd import engine
    engine.require_gpu()
    return engine


def _cells(d):
    M, N = int(d["nx"]) - 8, int(d["ny"]) - 8
    return M, N, (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)


def _chain(eng, corpus, batch):
    xp32 = eng.pack_x32(corpus, batch)
    if corpus.dtype == np.float32:
        koff, band = eng.keys16_koff_f32(corpus, batch, xp32), eng.keys16_band_f32(corpus, batch)
    else:
        koff, band = eng.keys16_koff(corpus, batch), eng.planar32_band(corpus, batch)
    return xp32, koff, band, eng.crp_keys16(corpus, batch, xp32, koff)


def _tie_corpus():
    """Periodic songs (every row holds a handful of distinct values, each many times: exact ties) and copies of them with
    1e-11 perturbations (values the float32 filter cannot tell apart)."""
    rng = np.random.default_rng(5)
    pat7, pat5 = rng.random((7, 12)) + 0.1, rng.random((5, 12)) + 0.1
    A = np.tile(pat7, (30, 1))[:200]
    Bs = np.tile(pat5, (31, 1))[:151]
    C = A + 1e-11 * rng.random(A.shape)
    Dn = Bs + 1e-11 * rng.random(Bs.shape)
    feats = np.concatenate([A, Bs, C, Dn])
    off = np.cumsum([0, len(A), len(Bs), len(C), len(Dn)]).astype(np.int64)
    gc = np.stack([x.sum(0) / x.sum(0).max() for x in (A, Bs, C, Dn)])
    return feats, off, gc, np.array([(i, j) for i in range(4) for j in range(4)], dtype=np.int32)


def test_key_plane_is_the_quantised_float32_matrix(eng, golden):
    """key16 = min(max(k' >> 11, (k' >> 9) -sat 49152), 0xFFFE), k' = float32 bits -sat koff, of exactly the values
    crp_planar32 writes (same arithmetic in the row-band and the column-strip kernel), for every pitch alignment; koff is
    the pattern of 2 W 2^-7."""
    from acoss_amd import synth
    g = golden("pairs_1000")
    lens = iter([9, 40, 65, 129, 300, 1032])
    small = synth.make_corpus(3, 2, seed=83, lengths=lambda r: next(lens))
    cases = [(g["feats"], g["frame_off"], g["gchroma"], g["pairs"]),
             (small.feats, small.frame_off, small.gchroma, np.array([(i, j) for i in range(6) for j in range(6)], dtype=np.int32))]
    for (feats, off, gc, pairs), align in [(c, a) for c in cases for a in (32, 2, 1)]:
        corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=align)
        eng.oti(corpus, batch)
        xp32, koff, band, k16 = _chain(eng, corpus, batch)
        k32 = eng.crp_planar32(corpus, batch, xp32).cpu().numpy().view(np.uint32).astype(np.int64) & 0x7fffffff
        k16 = k16.cpu().numpy().view(np.uint16).astype(np.int64)
        ko = koff.cpu().numpy().view(np.uint32).astype(np.int64)
        w = corpus.song_wmax(9)
        for p in range(batch.K):
            M, N, idx = _cells(batch.descs[p])
            kp = np.maximum(k32[idx] - ko[p], 0)
            want = np.minimum(np.maximum(kp >> 11, np.maximum((kp >> 9) - 49152, 0)), 0xFFFE)
            assert np.array_equal(k16[idx], want), (align, p)
            # behind the last key of a row: 0xFFFF up to the next multiple of 16 columns (where the pitch has room)
            d = batch.descs[p]
            pitch, pad_to = int(d["crp_pitch"]), min((N + 15) & ~15, int(d["crp_pitch"]))
            if pad_to > N:
                rows = int(d["crp_off"]) + np.arange(M)[:, None] * pitch + np.arange(N, pad_to)[None, :]
                assert (k16[rows] == 0xFFFF).all(), (align, p)
            W2 = 2.0 * (w[batch.descs["song_x"][p]] + w[batch.descs["song_y"][p]])
            top = np.array([ko[p] + (7 << 23)], dtype=np.uint32).view(np.float32)[0]
            assert W2 <= top <= W2 * (1 + 2.0 ** -22)


def test_masks_equal_the_float64_masks(eng, golden):
    import torch
    from acoss_amd import synth
    g = golden("pairs_1000")
    lens = iter([9, 40, 65, 129, 300, 1032])
    small = synth.make_corpus(3, 2, seed=83, lengths=lambda r: next(lens))
    tf, to, tg, tp = _tie_corpus()
    cases = [(g["feats"], g["frame_off"], g["gchroma"], g["pairs"], (0.095, 0.5, 3, 0)),
             (small.feats, small.frame_off, small.gchroma, np.array([(i, j) for i in range(6) for j in range(6)], dtype=np.int32), (0.095, 0.5, 3, 0)),
             (tf, to, tg, tp, (0.095,))]
    for ci, (feats, off, gc, pairs, kappas) in enumerate(cases):
        corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
        for align in (32, 2, 1):
            batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=align)
            eng.oti(corpus, batch)
            T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
            xp32, koff, band, k16 = _chain(eng, corpus, batch)
            for mutual in (True, False):
                for kappa in kappas:
                    want, _ = eng.mask_bits(T, batch, kappa, mutual=mutual)
                    got, _ = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, kappa, mutual=mutual)
                    if not torch.equal(got, want):
                        for p in range(batch.K):
                            assert np.array_equal(eng.unpack_mask_bits(got, batch, p), eng.unpack_mask_bits(want, batch, p)), (ci, align, mutual, kappa, p)


def test_useless_approximation_and_signed_13d_features(eng):
    """Features with a large per-bin offset (norms ~1e6, distances ~20): the float32 values are worthless, whole rows lie in
    reach of the error band (more than 64 cells: hand-over, then the general refinement); and 13-dimensional signed features
    (the MFCC shape) without OTI."""
    rng = np.random.default_rng(77)
    offs = 100.0 * (np.arange(12) + 1.0)
    lens = [210, 180, 333]
    feats = np.concatenate([offs[None, :] + rng.standard_normal((n, 12)) for n in lens])
    off = np.cumsum([0] + lens).astype(np.int64)
    gc = np.stack([np.abs(rng.standard_normal(12)) for _ in lens])
    mf = np.concatenate([np.cumsum(rng.standard_normal((n, 13)), axis=0) * 3.0 for n in lens])
    pairs = np.array([(0, 1), (1, 2), (2, 0), (1, 1)], dtype=np.int32)
    for F, do_oti in ((feats, True), (mf, False)):
        corpus = eng.DeviceCorpus(F, off, gchroma=gc)
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
        if do_oti:
            eng.oti(corpus, batch)
        T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
        xp32, koff, band, k16 = _chain(eng, corpus, batch)
        for mutual in (True, False):
            want, _ = eng.mask_bits(T, batch, 0.095, mutual=mutual)
            got, _ = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, mutual=mutual)
            for p in range(batch.K):
                assert np.array_equal(eng.unpack_mask_bits(got, batch, p), eng.unpack_mask_bits(want, batch, p)), (do_oti, mutual, p)


def test_thresholds_outside_the_key_range(eng, orc):
    """A song with long near-silent passages next to loud ones: the windowed sums of whole rows lie more than eight octaves
    below 2 W (their keys clamp to 0, so does the threshold: handed over and refined exactly), others clamp at the top.
    Masks equal the float64 path's, scores the oracle's."""
    from acoss_amd import synth
    rng = np.random.default_rng(12)
    ch = synth.make_corpus(2, 2, seed=12, lengths=lambda r: 260)
    feats = ch.feats.copy()
    off = ch.frame_off
    # songs 0 and 2: frames 60..200 almost identical to each other and tiny in spread (quiet passage), the rest as generated
    for s_ in (0, 2):
        a = int(off[s_])
        feats[a + 60:a + 200] = 0.3 + 1e-4 * rng.random((140, 12))
    gc = np.stack([feats[off[i]:off[i + 1]].sum(0) / feats[off[i]:off[i + 1]].sum(0).max() for i in range(4)])
    corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
    pairs = np.array([(0, 2), (2, 0), (0, 1), (1, 2), (0, 0), (3, 1)], dtype=np.int32)
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    eng.oti(corpus, batch)
    T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    kh = k16.cpu().numpy().view(np.uint16)
    M, N, idx = _cells(batch.descs[0])
    assert (np.sort(kh[idx], axis=1)[:, int(round(0.095 * N)) - 1] == 0).sum() > 50       # thresholds that left the range
    for mutual in (True, False):
        want, _ = eng.mask_bits(T, batch, 0.095, mutual=mutual)
        got, _ = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, mutual=mutual)
        for p in range(batch.K):
            assert np.array_equal(eng.unpack_mask_bits(got, batch, p), eng.unpack_mask_bits(want, batch, p)), (mutual, p)
    res = eng.serra09_scores(corpus, pairs)
    for t, (i, j) in enumerate(pairs):
        q, d = orc.serra09_pair(feats[off[i]:off[i + 1]], gc[i], feats[off[j]:off[j + 1]], gc[j])
        assert res["qmax"][t] == q and res["dmax"][t] == d, t


def test_side_buffer_overflow_goes_the_strided_way(eng):
    """More undecided rows than the side buffer holds (the tie corpus repeated: every row and column is undecided, the buffer
    takes 2 % of them): the rest is finished by the strided refinement kernels from the key plane itself."""
    import torch
    tf, to, tg, tp = _tie_corpus()
    corpus = eng.DeviceCorpus(tf, to, gchroma=tg)
    pairs = np.tile(np.array([(0, 2), (2, 0), (1, 3), (0, 0), (2, 2), (3, 1)], dtype=np.int32), (40, 1))      # 240 pairs, ~80 000 rows + columns
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    eng.oti(corpus, batch)
    T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    want, _ = eng.mask_bits(T, batch, 0.095, mutual=True)
    got, _ = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, mutual=True)
    assert torch.equal(got, want)


def test_scale_free(eng, orc):
    """Corpora of very large and very small magnitude through the product chain (16-bit keys are its default)."""
    from acoss_amd import synth
    lens = iter([300, 220, 410])
    ch = synth.make_corpus(3, 1, seed=19, lengths=lambda r: next(lens))
    pairs = np.array([(0, 1), (1, 2), (2, 0)], dtype=np.int32)
    assert eng.keys16_default()
    for mag in (1e25, 1e-25):
        feats = ch.feats * mag
        corpus = eng.DeviceCorpus(feats, ch.frame_off, gchroma=ch.gchroma)
        got = eng.serra09_scores(corpus, pairs, approx32=True)
        for t, (i, j) in enumerate(pairs):
            q, d = orc.serra09_pair(feats[ch.frame_off[i]:ch.frame_off[i + 1]], ch.gchroma[i],
                                    feats[ch.frame_off[j]:ch.frame_off[j + 1]], ch.gchroma[j])
            assert got["qmax"][t] == q and got["dmax"][t] == d, (mag, t)


def test_pairs_far_below_the_corpus_scale(eng):
    """Two songs 1e-20 of the loudness of the rest of the corpus: after the centring of the float32 copy their frames are
    almost constant vectors with ordinary norms, so their windowed sums cancel catastrophically in float32 (every key of
    the quiet pair clamps to 0 or carries no information) and everything is decided by the exact refinement; pairs mixing
    them with ordinary songs too (rows of equal values up to the last bits of float64: exact ties, cut lowest position
    first).  Masks equal the float64 kernels' (the oracle forms its window sums by cumulative differences and breaks such
    ties differently: section 3 of DESIGN.md)."""
    import torch
    from acoss_amd import synth
    lens = iter([240, 300, 210, 280])
    ch = synth.make_corpus(2, 2, seed=33, lengths=lambda r: next(lens))
    feats = ch.feats.copy()
    off = ch.frame_off
    for s_ in (1, 3):
        feats[off[s_]:off[s_ + 1]] *= 1e-20
    corpus = eng.DeviceCorpus(feats, off, gchroma=ch.gchroma)
    pairs = np.array([(1, 3), (3, 1), (0, 1), (3, 2), (0, 2), (1, 1)], dtype=np.int32)
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    eng.oti(corpus, batch)
    T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    for mutual in (True, False):
        want, _ = eng.mask_bits(T, batch, 0.095, mutual=mutual)
        got, _ = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, mutual=mutual)
        for p in range(batch.K):
            assert np.array_equal(eng.unpack_mask_bits(got, batch, p), eng.unpack_mask_bits(want, batch, p)), (mutual, p)
    a = eng.serra09_scores(corpus, pairs)
    b = eng.serra09_scores(corpus, pairs, approx32=False)
    assert np.array_equal(a["qmax"], b["qmax"]) and np.array_equal(a["dmax"], b["dmax"])


def _f32_reference_masks(eng, corpus, batch, kappa, mutual):
    """The float32-input chain as the staged path runs it (crp_kernel<float>: CRPUtils.py:82 + :40-41, then the float64
    selection): the definition the filter must reproduce."""
    xp = eng.pack_x(corpus, batch)
    T = eng.crp(corpus, batch, xp, sqrt_out=False)
    return eng.mask_bits(T, batch, kappa, mutual=mutual)[0]


@pytest.mark.parametrize("d,do_oti", [(13, False), (12, True)])
def test_float32_corpus_masks_equal_the_float32_input_chain(eng, d, do_oti):
    """Round 4: the 16-bit-key filter for corpora of float32 features (the reference's mfcc_htk, 13-d, and essentia hpcp, 12-d
    with OTI).  Operand = the corpus itself, so the cross-similarity values are the exact path's; tiers 2 and 3 must
    reproduce the float32-input windowed sums (sum of (double)(sqrtf(c)^2)).  Masks against crp_kernel<float> + float64
    selection, bit for bit: smooth random walks, features with a large common offset (window norm sums far above the
    distances: thresholds fall into the coarse keys or below the key range -- handed over), tiny and huge magnitudes, exact
    ties (repeated frames), ragged lengths, both kappa conventions."""
    rng = np.random.default_rng(100 + d)
    lens = [60, 131, 257, 400, 1032, 333]
    walk = [np.cumsum(rng.standard_normal((n, d)), axis=0).astype(np.float32) for n in lens]
    fams = {"walk": walk,
            "offset": [(w + np.float32(200.0)).astype(np.float32) for w in walk],
            "tiny": [(w * np.float32(1e-12)).astype(np.float32) for w in walk],
            "huge": [(w * np.float32(1e+12)).astype(np.float32) for w in walk],
            "ties": [np.tile(w[:11], (len(w) // 11 + 1, 1))[:len(w)].copy() for w in walk]}
    pairs = np.array([(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0), (4, 4), (2, 0)], dtype=np.int32)
    for name, songs in fams.items():
        if d == 12:
            songs = [np.abs(s) for s in songs]
        feats = np.concatenate(songs)
        off = np.cumsum([0] + [len(s) for s in songs]).astype(np.int64)
        gc = np.stack([s.astype(np.float64).sum(0) / max(float(s.astype(np.float64).sum(0).max()), 1e-300) for s in songs]) if do_oti else None
        corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
        assert corpus.dtype == np.float32
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
        if do_oti:
            eng.oti(corpus, batch)
        if not eng.keys16_supported(corpus, batch):
            assert name in ("huge",)                       # squared norms beyond float32: stays on the staged kernels
            continue
        xp32, koff, band, k16 = _chain(eng, corpus, batch)
        for kappa, mutual in ((0.095, True), (0.095, False), (7, True)):
            want = _f32_reference_masks(eng, corpus, batch, kappa, mutual)
            got, _ = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, kappa, mutual=mutual)
            for p in range(batch.K):
                a, b = eng.unpack_mask_bits(got, batch, p), eng.unpack_mask_bits(want, batch, p)
                if name == "ties":
                    # exact ties: both forms cut lowest position first, but their float64 sums of equal terms are formed in the
                    # same order -- still identical
                    pass
                assert np.array_equal(a, b), (name, kappa, mutual, p, int((a != b).sum()))


def test_float32_corpus_scores_through_the_engine(eng, orc):
    """engine.serra09_scores on a float32 corpus takes the filter path and returns the scores of the oracle's float32 chain."""
    rng = np.random.default_rng(77)
    lens = [90, 150, 220, 317]
    songs = [np.cumsum(rng.standard_normal((n, 13)), axis=0).astype(np.float32) for n in lens]
    feats = np.concatenate(songs)
    off = np.cumsum([0] + lens).astype(np.int64)
    corpus = eng.DeviceCorpus(feats, off)
    pairs = np.array([(i, j) for i in range(4) for j in range(4) if i != j], dtype=np.int32)
    got = eng.serra09_scores(corpus, pairs, do_oti=False, want=("qmax", "dmax", "swc"))
    for t, (i, j) in enumerate(pairs):
        B = orc.csm_to_binary_mutual(orc.sliding_csm(orc.get_csm(songs[i], songs[j]), 9), 0.095)
        M, N = B.shape
        Bf = np.ascontiguousarray(B.flatten())
        D = np.zeros(M * N, dtype=np.float32)
        q = orc.qmax(Bf, D, M, N) / (M + N)
        dm = orc.dmax(Bf, D, M, N) / (M + N)
        assert got["qmax"][t] == q and got["dmax"][t] == dm, (t, i, j)


def _wrapped_bin_planes():
    """Two 1024 x 1024 key planes (no padding) for the regression test below.  Every row and column of the base plane is a
    permutation of 20000 + 4 t (gaps of four keys: nothing within reach of anything).  Plane A, rows 0 and 1; plane B (its
    transpose), columns 0 and 1: the first has its third-largest key at 65522; the second holds 65522, then 65524 and 65526 in
    ONE lane (neighbouring positions 200, 201), 65530, 65534, and a key 0 -- with k = 1022 its k-th smallest key is 65526, and the
    histogram window predicted from 65522 puts it into the bin [65522, 65537], which reaches past the top of the key range; the
    key 0 sits 14 below that bin modulo 2^16."""
    n = 1024
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    base = (20000 + 4 * ((i * 5 + j * 7) % n)).astype(np.uint16)
    A = base.copy()
    A[0, 500], A[0, 550], A[0, 600] = 65530, 65526, 65522
    A[1, 100], A[1, 200], A[1, 201], A[1, 300], A[1, 350], A[1, 400] = 65522, 65524, 65526, 65530, 65534, 0
    return A, np.ascontiguousarray(A.T)


@pytest.mark.parametrize("radix", [False, True])
def test_histogram_bin_that_reaches_past_the_top_of_the_key_range(eng, radix):
    """Regression test for commit d26ca8d (csrc/keys16.h, k16_bin_keys): offsets from the bin's first key are taken modulo 2^16,
    so a bin that reaches past key 0xFFFF took keys far BELOW it for its own; with two of the bin's real keys in one lane the
    lane count still matched and the threshold came out as 65530 instead of 65526 (one cell too many selected).  Checked by building the library with
    -DK16_REGRESSION_D26CA8D (the old line) and running this test against it (ACOSS_LIB_PATH): the wave-per-row form fails there
    (profiles/r05_regression_d26ca8d.txt).  The key planes are synthetic (decided by the keys alone: no cell within reach of a
    threshold), k = 1022 of 1024; also through the radix selection, which has no such bins."""
    import os
    import torch
    rng = np.random.default_rng(9)
    feats = rng.random((2 * 1032, 12)) + 0.1
    off = np.array([0, 1032, 2064], dtype=np.int64)
    corpus = eng.DeviceCorpus(feats, off, gchroma=np.ones((2, 12)))
    batch = eng.PairBatch(corpus.frame_off, np.array([(0, 1), (1, 0)], dtype=np.int32), 9, corpus.device, pitch_align=32)
    A, B = _wrapped_bin_planes()
    plane = np.zeros(batch.total_crp + 64, dtype=np.uint16)
    for p, P in enumerate((A, B)):
        d = batch.descs[p]
        assert int(d["crp_pitch"]) == 1024
        plane[int(d["crp_off"]):int(d["crp_off"]) + 1024 * 1024] = P.reshape(-1)
    k16 = torch.from_numpy(plane.view(np.int16)).to(corpus.device)
    xp32 = eng.pack_x32(corpus, batch)
    band = torch.tensor([0.0, 1e-9] * 2, dtype=torch.float32, device=corpus.device)
    koff = torch.tensor([0x3C000000, 0x3C000000], dtype=torch.int32, device=corpus.device)
    old = os.environ.get("ACOSS_RADIX16")
    os.environ["ACOSS_RADIX16"] = "1" if radix else "0"
    try:
        rows_only, _ = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 1022, mutual=False)
        rows_only = rows_only.clone()
        both, _ = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 1022, mutual=True)
    finally:
        if old is None:
            os.environ.pop("ACOSS_RADIX16", None)
        else:
            os.environ["ACOSS_RADIX16"] = old

    def kth(P, axis):
        return np.sort(P.astype(np.int64), axis=axis).take(1021, axis=axis)
    want_rows = (A.astype(np.int64) <= kth(A, 1)[:, None]).astype(np.uint8)
    assert kth(A, 1)[0] == 65522 and kth(A, 1)[1] == 65526 and want_rows[1].sum() == 1022 and want_rows[1, 300] == 0
    assert np.array_equal(eng.unpack_mask_bits(rows_only, batch, 0), want_rows)
    want_b = ((B.astype(np.int64) <= kth(B, 1)[:, None]) & (B.astype(np.int64) <= kth(B, 0)[None, :])).astype(np.uint8)
    assert kth(B, 0)[1] == 65526
    assert np.array_equal(eng.unpack_mask_bits(both, batch, 1), want_b)
