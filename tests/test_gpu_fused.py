"""GPU parity of the fused band kernel (csrc/band_kernels.hip: get_csm + sliding_csm + csm_to_binary_mutual with nothing but
bit planes leaving the chip): identical masks to the reference's own B / B1 matrices and to the float64 materialising path,
through the exact-refinement side buffer, its overflow re-run, ragged shapes, exact ties and degenerate kappas."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from acoss_amd import engine
    engine.require_gpu()
    return engine


def _fused_masks(eng, corpus, pairs, kappa, mutual=True, m=9, **kw):
    batch = eng.PairBatch(corpus.frame_off, pairs, m, corpus.device, pitch_align=32)
    if corpus.gchroma is not None:
        eng.oti(corpus, batch)
    assert eng.fused_supported(corpus, batch)
    bits, work = eng.mask_bits_fused(corpus, batch, kappa, mutual=mutual, **kw)
    return batch, bits, int(eng.fused_counter(work).item())


def _f64_masks(eng, corpus, pairs, kappa, mutual=True, m=9):
    batch = eng.PairBatch(corpus.frame_off, pairs, m, corpus.device, pitch_align=32)
    if corpus.gchroma is not None:
        eng.oti(corpus, batch)
    planes = eng.crp_planar(corpus, batch, eng.pack_x(corpus, batch))
    bits, _ = eng.mask_bits_planar(planes, corpus, batch, kappa, mutual=mutual)
    return batch, bits


@pytest.mark.parametrize("c", [0, 1, 2])
def test_fused_masks_equal_reference_masks(eng, golden, c):
    """CRPUtils.py:169-219 on the reference's own stage dumps: B (mutual) and B1 (rows only)."""
    g = golden("stages")
    p = "c%d_" % c
    X, Y, m, kappa = g[p + "X"], g[p + "Y"], int(g[p + "m"]), float(g[p + "kappa"])
    if m != 9:
        pytest.skip("the fused kernel is built for the reference's window of 9")
    corpus = eng.DeviceCorpus(np.concatenate([X, Y]), np.array([0, len(X), len(X) + len(Y)]),
                              gchroma=np.stack([g[p + "gX"], g[p + "gY"]]))
    b, bits, _ = _fused_masks(eng, corpus, [[0, 1]], kappa, mutual=True, m=m)
    assert np.array_equal(eng.unpack_mask_bits(bits, b, 0), g[p + "B"])
    b, bits, _ = _fused_masks(eng, corpus, [[0, 1]], kappa, mutual=False, m=m)
    assert np.array_equal(eng.unpack_mask_bits(bits, b, 0), g[p + "B1"])


def test_fused_equals_float64_path_on_1000_frame_pairs(eng, golden):
    g = golden("pairs_1000")
    corpus = eng.DeviceCorpus(g["feats"], g["frame_off"], gchroma=g["gchroma"])
    b0, bits0 = _f64_masks(eng, corpus, g["pairs"], 0.095)
    b1, bits1, asked = _fused_masks(eng, corpus, g["pairs"], 0.095)
    for p in range(b0.K):
        assert np.array_equal(eng.unpack_mask_bits(bits0, b0, p), eng.unpack_mask_bits(bits1, b1, p)), p
    assert 0 < asked < 200          # a few rows per pair go through the exact refinement
    # the side buffer too small for them: the call reports it and is repeated with room for all
    b2, bits2, asked2 = _fused_masks(eng, corpus, g["pairs"], 0.095, side_rows=1)
    assert asked2 == asked
    for p in range(b0.K):
        assert np.array_equal(eng.unpack_mask_bits(bits0, b0, p), eng.unpack_mask_bits(bits2, b2, p)), p


def test_fused_ragged_shapes_and_scores(eng, orc):
    """Every length class of the register layout (9 frames = a 1 x 1 matrix ... 1022 frames = the widest), all ordered
    pairs; masks against the float64 path, scores against the CPU oracle."""
    from acoss_amd import synth
    lens = iter([9, 10, 12, 33, 64, 65, 100, 131, 257, 300, 511, 640, 777, 1000, 1021, 1022])
    ch = synth.make_corpus(8, 2, seed=79, lengths=lambda r: next(lens))
    corpus = eng.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = np.array([(i, j) for i in range(16) for j in range(16)], dtype=np.int32)
    b0, bits0 = _f64_masks(eng, corpus, pairs, 0.095)
    b1, bits1, _ = _fused_masks(eng, corpus, pairs, 0.095)
    for p in range(b0.K):
        assert np.array_equal(eng.unpack_mask_bits(bits0, b0, p), eng.unpack_mask_bits(bits1, b1, p)), pairs[p]
    sel = pairs[::5]
    b, bits, _ = _fused_masks(eng, corpus, sel, 0.095)
    q, d = eng.align_bits_qd(bits, b, boundary=1)
    denom = (b.M + b.N).astype(np.float64)
    qo, do, _ = orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, sel, nthreads=4)
    assert np.array_equal(q.cpu().numpy().astype(np.float64) / denom, qo)
    assert np.array_equal(d.cpu().numpy().astype(np.float64) / denom, do)


def test_fused_exact_ties_and_kappa_conventions(eng):
    """Periodic songs give many exactly equal windowed sums (every such row goes through the float64 refinement, ties cut
    lowest position first as in the float64 path); kappa = 0 / >= 1 / tiny follow CRPUtils.py:188-193."""
    rng = np.random.default_rng(5)
    base = rng.random((25, 12))
    X = np.tile(base, (8, 1))                   # 200 frames, period 25
    Y = np.tile(base[::-1], (6, 1)) * 0.5 + 0.25 * np.tile(base, (6, 1))
    feats = np.concatenate([X, Y, X[:37] + 0.01 * rng.random((37, 12))])
    off = np.array([0, len(X), len(X) + len(Y), len(feats)])
    corpus = eng.DeviceCorpus(feats, off)
    pairs = [[0, 1], [1, 0], [0, 2], [2, 2], [0, 0]]
    for kappa in (0.095, 0.3, 0.0, 5.0, 1e-4):
        b0, bits0 = _f64_masks(eng, corpus, pairs, kappa)
        b1, bits1, _ = _fused_masks(eng, corpus, pairs, kappa)
        for p in range(b0.K):
            assert np.array_equal(eng.unpack_mask_bits(bits0, b0, p), eng.unpack_mask_bits(bits1, b1, p)), (kappa, p)


def test_fused_mfcc_width_and_scaled_corpora(eng):
    """13-dimensional features (the MFCC chain, no OTI) and corpora of extreme magnitude (the float32 copy is centred and
    rescaled by a power of two)."""
    from acoss_amd import synth
    rng = np.random.default_rng(11)
    lens = [150, 333, 1000]
    feats = rng.standard_normal((sum(lens), 13)) * 20.0
    off = np.concatenate([[0], np.cumsum(lens)])
    pairs = [[0, 1], [2, 1], [1, 2], [2, 2]]
    for scale in (1.0, 1e20, 1e-20):
        corpus = eng.DeviceCorpus(feats * scale, off)
        b0, bits0 = _f64_masks(eng, corpus, pairs, 0.095)
        b1, bits1, _ = _fused_masks(eng, corpus, pairs, 0.095)
        for p in range(b0.K):
            assert np.array_equal(eng.unpack_mask_bits(bits0, b0, p), eng.unpack_mask_bits(bits1, b1, p)), (scale, p)
