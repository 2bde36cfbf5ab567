"""GPU parity of the fast forms: packed-x CSM (bit-identical to the plain CSM kernel) and the fused
CSM + sliding-window kernel, then the whole fast chain against the reference's scores."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from acoss_amd import engine
    engine.require_gpu()
    return engine


def _mat(buf, batch, p, what):
    d = batch.descs[p]
    if what == "csm":
        rows, cols, off, pitch = d["nx"], d["ny"], d["csm_off"], d["csm_pitch"]
    else:
        rows, cols, off, pitch = d["nx"] - batch.win + 1, d["ny"] - batch.win + 1, d["crp_off"], d["crp_pitch"]
    return buf[off:off + rows * pitch].cpu().numpy().reshape(rows, pitch)[:, :cols]


@pytest.mark.parametrize("c", [0, 1, 2])
def test_fused_kernels_against_reference_stages(eng, golden, c):
    g = golden("stages")
    p = "c%d_" % c
    X, Y, m, kappa = g[p + "X"], g[p + "Y"], int(g[p + "m"]), float(g[p + "kappa"])
    corpus = eng.DeviceCorpus(np.concatenate([X, Y]), np.array([0, len(X), len(X) + len(Y)]),
                              gchroma=np.stack([g[p + "gX"], g[p + "gY"]]))
    batch = eng.PairBatch(corpus.frame_off, [[0, 1]], m, corpus.device)
    eng.oti(corpus, batch)
    xp = eng.pack_x(corpus, batch)
    C0 = eng.csm(corpus, batch)
    C1 = eng.csm_packed(corpus, batch, xp)
    assert np.array_equal(_mat(C0, batch, 0, "csm"), _mat(C1, batch, 0, "csm"))     # same arithmetic, bit for bit
    C2 = eng.csm_strip(corpus, batch, xp)                                           # matrix-core strip kernel
    assert np.array_equal(_mat(C0, batch, 0, "csm"), _mat(C2, batch, 0, "csm"))
    C3 = eng.csm_rows(corpus, batch, xp)                                            # row-band kernel (round 4)
    assert np.array_equal(_mat(C0, batch, 0, "csm"), _mat(C3, batch, 0, "csm"))
    S = _mat(eng.crp(corpus, batch, xp, sqrt_out=True), batch, 0, "crp")
    assert np.max(np.abs(S - g[p + "S"])) <= 1e-9
    Tbuf = eng.crp(corpus, batch, xp, sqrt_out=False)
    T = _mat(Tbuf, batch, 0, "crp")
    assert np.max(np.abs(T - g[p + "S"] ** 2)) <= 1e-9
    # the three forms of the float64 kernel (persistent strips on the matrix cores = default, tiled on
    # the matrix cores, tiled all-VALU) agree bit for bit: a float64 MFMA is a k-ordered FMA chain
    Tv = _mat(eng.crp(corpus, batch, xp, sqrt_out=False, force_valu=True), batch, 0, "crp")
    Tt = _mat(eng.crp(corpus, batch, xp, sqrt_out=False, force_tile=True), batch, 0, "crp")
    assert np.array_equal(T, Tv) and np.array_equal(T, Tt)
    # selection on the squared sums gives the reference's masks
    assert np.array_equal(_mat(eng.binarize(Tbuf, batch, kappa, mutual=False), batch, 0, "crp"), g[p + "B1"])
    assert np.array_equal(_mat(eng.binarize(Tbuf, batch, kappa, mutual=True), batch, 0, "crp"), g[p + "B"])


@pytest.mark.parametrize("d", [12, 13])
def test_row_band_csm_equals_the_valu_kernel_on_ragged_batches(eng, d):
    """get_csm in row-band form (csm_rows_kernel; CRPUtils.py:67-84) against csm_kernel on a batch of ragged pairs -- lengths
    1 .. 1000 including non-multiples of every tile size, OTI shifts, d = 12 and 13 -- whole matrices bit for bit, and no
    byte written outside them (columns past the end of a row and the rows of other pairs keep their fill pattern)."""
    import torch
    rng = np.random.default_rng(40 + d)
    lens = [1, 2, 15, 16, 17, 31, 33, 64, 97, 127, 128, 129, 255, 300, 513, 1000, 999]
    feats = rng.random((sum(lens), d))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    corpus = eng.DeviceCorpus(feats, off, gchroma=rng.random((len(lens), 12)) if d == 12 else None)
    pairs = [(i, j) for i in range(len(lens)) for j in (0, 3, 7, 11, 15, 16) if i != j][:60] + [(5, 5), (12, 12), (15, 15), (16, 16)]
    # (self pairs: exact zeros on the diagonal -- the values the kernel's fast square root must not be given)
    for pitch_align in (16, 32):
        batch = eng.PairBatch(corpus.frame_off, pairs, 1, corpus.device, pitch_align=pitch_align)
        if d == 12:
            eng.oti(corpus, batch)
        xp = eng.pack_x(corpus, batch)
        want = eng.csm(corpus, batch)
        fill = float.fromhex("0x1.5p+40")
        got = torch.full((batch.total_csm,), fill, dtype=torch.float64, device=corpus.device)
        eng.csm_rows(corpus, batch, xp, out=got)
        gh, wh = got.cpu().numpy(), want.cpu().numpy()
        untouched = np.ones(batch.total_csm, dtype=bool)
        for p in range(batch.K):
            dsc = batch.descs[p]
            nx, ny, o, pt = int(dsc["nx"]), int(dsc["ny"]), int(dsc["csm_off"]), int(dsc["csm_pitch"])
            G = gh[o:o + nx * pt].reshape(nx, pt)
            W = wh[o:o + nx * pt].reshape(nx, pt)
            assert np.array_equal(G[:, :ny], W[:, :ny]), (p, nx, ny)
            if pairs[p][0] == pairs[p][1] and d == 13:
                assert np.all(np.diag(G[:, :ny]) == 0.0)
            untouched[o:o + nx * pt].reshape(nx, pt)[:, :ny] = False
        assert np.all(gh[untouched] == fill)


def test_fused_float32_features(eng, golden):
    g = golden("stages")
    X, Y = g["f32_X"], g["f32_Y"]
    corpus = eng.DeviceCorpus(np.concatenate([X, Y]), np.array([0, len(X), len(X) + len(Y)]))
    assert corpus.dtype == np.float32
    batch = eng.PairBatch(corpus.frame_off, [[0, 1]], 9, corpus.device)
    xp = eng.pack_x(corpus, batch)
    assert np.array_equal(_mat(eng.csm(corpus, batch), batch, 0, "csm"), _mat(eng.csm_packed(corpus, batch, xp), batch, 0, "csm"))
    S = _mat(eng.crp(corpus, batch, xp, sqrt_out=True), batch, 0, "crp")
    assert np.max(np.abs(S - g["f32_S"])) <= 2e-6          # the CSM itself is float32 (2e-6 to the reference's BLAS)
    # against our own staged float32 chain the fused kernel is exact
    S2 = _mat(eng.sliding(eng.csm(corpus, batch), batch), batch, 0, "crp")
    assert np.max(np.abs(S - S2)) <= 1e-15


def test_fast_chain_scores(eng, golden, orc):
    g = golden("serra09_mini")
    corpus = eng.DeviceCorpus(g["feats"], g["frame_off"], gchroma=g["gchroma"])
    res = eng.serra09_scores(corpus, g["pairs"], batch_pairs=30)
    assert np.array_equal(res["qmax"], g["chroma_qmax"]) and np.array_equal(res["dmax"], g["chroma_dmax"])
    mf = eng.DeviceCorpus(g["mfcc"], g["frame_off"])
    res = eng.serra09_scores(mf, g["pairs"], do_oti=False)
    assert np.array_equal(res["qmax"], g["mfcc_qmax"]) and np.array_equal(res["dmax"], g["mfcc_dmax"])
    g = golden("pairs_1000")
    corpus = eng.DeviceCorpus(g["feats"], g["frame_off"], gchroma=g["gchroma"])
    res = eng.serra09_scores(corpus, g["pairs"])
    assert np.array_equal(res["qmax"], g["chroma_qmax"]) and np.array_equal(res["dmax"], g["chroma_dmax"])


def test_fast_chain_ragged_and_other_windows(eng, orc):
    from acoss_amd import synth
    lens = iter([9, 10, 12, 33, 64, 65, 100, 131, 257, 300])
    ch = synth.make_corpus(5, 2, seed=79, lengths=lambda r: next(lens))
    corpus = eng.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = np.array([(i, j) for i in range(10) for j in range(10)], dtype=np.int32)
    res = eng.serra09_scores(corpus, pairs)
    q, d, _ = orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, pairs, nthreads=4)
    assert np.array_equal(res["qmax"], q) and np.array_equal(res["dmax"], d)
    for m, kappa in ((1, 0.2), (4, 0.1), (16, 0.15)):      # window sizes other than the templated 9
        sel = pairs[(pairs[:, 0] >= 3) & (pairs[:, 1] >= 3)]
        res = eng.serra09_scores(corpus, sel, m=m, kappa=kappa)
        q, d, _ = orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, sel, m=m, kappa=kappa, nthreads=4)
        assert np.array_equal(res["qmax"], q) and np.array_equal(res["dmax"], d), m


def test_fused_alignment_equals_mask_then_alignment(eng, golden):
    """qmax / dmax computed straight from the windowed sums and thresholds == the same recurrences on
    the materialised mask, in every boundary mode, one-sided and mutual."""
    g = golden("pairs_1000")
    corpus = eng.DeviceCorpus(g["feats"], g["frame_off"], gchroma=g["gchroma"])
    batch = eng.PairBatch(corpus.frame_off, g["pairs"], 9, corpus.device)
    eng.oti(corpus, batch)
    T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
    mats, _ = batch.mats()
    for mutual in (True, False):
        B = eng.binarize(T, batch, 0.095, mutual=mutual)
        work = eng.thresholds(T, batch, 0.095, mutual=mutual)
        assert np.array_equal(eng.align_fused("qmax", T, batch, work, mutual=mutual).cpu().numpy(),
                              eng.align("qmax", B, mats).cpu().numpy())
        for boundary in (0, 1):
            assert np.array_equal(eng.align_fused("dmax", T, batch, work, mutual=mutual, boundary=boundary).cpu().numpy(),
                                  eng.align("dmax", B, mats, boundary=boundary).cpu().numpy())


def test_results_do_not_depend_on_stale_device_memory(eng, orc):
    """Buffers come from torch.empty: padding elements (odd widths, pitch rounding) hold whatever the
    allocator's previous tenant left.  Poison the pool with adversarial bit patterns (negative, huge,
    NaN, small positive) and check that scores on odd-sized ragged pairs do not move."""
    import torch
    from acoss_amd import synth
    lens = iter([41, 57, 99, 123, 201, 333])
    ch = synth.make_corpus(3, 2, seed=81, lengths=lambda r: next(lens))
    pairs = np.array([(i, j) for i in range(6) for j in range(6)], dtype=np.int32)
    q, d, _ = orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, pairs, nthreads=4)
    for fill in (-1e300, 1e-300, float("nan"), -3.5, 0.0):
        junk = [torch.full((1 << 24,), fill, dtype=torch.float64, device="cuda") for _ in range(6)]
        rnd = torch.randint(-2 ** 62, 2 ** 62, (1 << 24,), dtype=torch.int64, device="cuda")
        del junk, rnd
        corpus = eng.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
        for fn in (eng.serra09_scores, eng.serra09_scores_staged):
            res = fn(corpus, pairs, batch_pairs=13)
            assert np.array_equal(res["qmax"], q) and np.array_equal(res["dmax"], d), (fill, fn.__name__)
        del corpus


def test_bit_mask_path_equals_byte_mask_path(eng, golden):
    """Bit-packed masks from the selection ballots == the uint8 masks, and alignment from bits == alignment
    from bytes (mutual and one-sided, both dmax boundaries), on ragged 1000-frame and small pairs."""
    from acoss_amd import synth
    g = golden("pairs_1000")
    lens = iter([9, 40, 65, 129, 300, 1032])
    small = synth.make_corpus(3, 2, seed=83, lengths=lambda r: next(lens))
    cases = [(g["feats"], g["frame_off"], g["gchroma"], g["pairs"]),
             (small.feats, small.frame_off, small.gchroma, np.array([(i, j) for i in range(6) for j in range(6)], dtype=np.int32))]
    for feats, off, gc, pairs in cases:
        corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device)
        eng.oti(corpus, batch)
        T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
        mats, _ = batch.mats()
        for mutual in (True, False):
            B = eng.binarize(T, batch, 0.095, mutual=mutual)
            bits, _ = eng.mask_bits(T, batch, 0.095, mutual=mutual)
            for p in range(0, batch.K, max(1, batch.K // 7)):
                d = batch.descs[p]
                M, N = d["nx"] - 8, d["ny"] - 8
                Bp = B[d["crp_off"]:d["crp_off"] + M * d["crp_pitch"]].cpu().numpy().reshape(M, -1)[:, :N]
                assert np.array_equal(eng.unpack_mask_bits(bits, batch, p), Bp), (p, mutual)
            assert np.array_equal(eng.align_bits("qmax", bits, batch).cpu().numpy(), eng.align("qmax", B, mats).cpu().numpy())
            for boundary in (0, 1):
                assert np.array_equal(eng.align_bits("dmax", bits, batch, boundary=boundary).cpu().numpy(),
                                      eng.align("dmax", B, mats, boundary=boundary).cpu().numpy())
                q, d = eng.align_bits_qd(bits, batch, boundary=boundary)          # both in one sweep
                assert np.array_equal(q.cpu().numpy(), eng.align("qmax", B, mats).cpu().numpy())
                assert np.array_equal(d.cpu().numpy(), eng.align("dmax", B, mats, boundary=boundary).cpu().numpy())


def _key_hi(T):
    """High words of the order-preserving keys of a float64 device vector (include/acoss_mi355x.h,
    acoss_crp_planar_batch_f64), as int64 numpy."""
    import torch
    b = (T + 0.0).view(torch.int64)
    key = torch.where(b < 0, ~b, b | (-0x8000000000000000))
    return ((key >> 32) & 0xffffffff).cpu().numpy()


def test_planar_planes_decode_to_float64_result(eng, golden):
    """crp_planar writes exactly the high words of crp()'s values (order-preserving keys), indexed like the float64 matrix."""
    import torch
    from acoss_amd import synth
    g = golden("pairs_1000")
    lens = iter([9, 40, 65, 129, 300, 1032])
    small = synth.make_corpus(3, 2, seed=83, lengths=lambda r: next(lens))
    cases = [(g["feats"], g["frame_off"], g["gchroma"], g["pairs"]),
             (small.feats, small.frame_off, small.gchroma, np.array([(i, j) for i in range(6) for j in range(6)], dtype=np.int32))]
    for (feats, off, gc, pairs), align in [(c, a) for c in cases for a in (32, 16, 1)]:
        corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=align)
        eng.oti(corpus, batch)
        xp = eng.pack_x(corpus, batch)
        T = eng.crp(corpus, batch, xp)
        planes = eng.crp_planar(corpus, batch, xp, out=torch.full((eng.planar_elems(batch),), -1, dtype=torch.int32, device=corpus.device))
        planes = planes.cpu().numpy().astype(np.int64) & 0xffffffff
        want = _key_hi(T)
        for p in range(batch.K):
            d = batch.descs[p]
            M, N = d["nx"] - 8, d["ny"] - 8
            idx = (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)
            assert np.array_equal(planes[idx], want[idx]), p


def test_planar_mask_equals_float64_mask(eng, golden):
    """mask_bits_planar == mask_bits: on the golden pairs, on ragged small pairs, and on matrices crafted so
    that the k-th smallest shares its high word with other elements or is an exact tie (fix-up pass)."""
    import torch
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
        xp = eng.pack_x(corpus, batch)
        T = eng.crp(corpus, batch, xp)
        planes = eng.crp_planar(corpus, batch, xp)
        for mutual in (True, False):
            for kappa in (0.095, 0.5, 3, 0):
                want, _ = eng.mask_bits(T, batch, kappa, mutual=mutual)
                got, _ = eng.mask_bits_planar(planes, corpus, batch, kappa, mutual=mutual)
                for p in range(batch.K):
                    assert np.array_equal(eng.unpack_mask_bits(got, batch, p), eng.unpack_mask_bits(want, batch, p)), (p, mutual, kappa)
    # songs crafted so that the k-th smallest value is an exact tie (periodic frames: every row holds a handful of
    # distinct values, each many times) or shares its high word with its neighbours (copies of a song with 1e-11
    # perturbations): those rows and columns go through the fix-up kernel, which recomputes them from the features
    rng = np.random.default_rng(5)
    pat7, pat5 = rng.random((7, 12)) + 0.1, rng.random((5, 12)) + 0.1
    A = np.tile(pat7, (30, 1))[:200]
    Bs = np.tile(pat5, (31, 1))[:151]
    C = A + 1e-11 * rng.random(A.shape)
    Dn = Bs + 1e-11 * rng.random(Bs.shape)
    feats = np.concatenate([A, Bs, C, Dn])
    off = np.cumsum([0, len(A), len(Bs), len(C), len(Dn)]).astype(np.int64)
    corpus = eng.DeviceCorpus(feats, off, gchroma=np.stack([x.sum(0) / x.sum(0).max() for x in (A, Bs, C, Dn)]))
    pairs = np.array([(i, j) for i in range(4) for j in range(4)], dtype=np.int32)
    for align in (32, 2):
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=align)
        eng.oti(corpus, batch)
        xp = eng.pack_x(corpus, batch)
        T = eng.crp(corpus, batch, xp)
        planes = eng.crp_planar(corpus, batch, xp)
        for mutual in (True, False):
            want, _ = eng.mask_bits(T, batch, 0.095, mutual=mutual)
            got, _ = eng.mask_bits_planar(planes, corpus, batch, 0.095, mutual=mutual)
            assert torch.equal(got, want), (align, mutual)


def test_smith_waterman_from_bits_equals_from_bytes(eng, golden):
    """swalignimpconstrained from the bit mask == from the byte mask (same float32 expression order), mutual and
    one-sided masks, ragged sizes."""
    from acoss_amd import synth
    g = golden("pairs_1000")
    lens = iter([9, 40, 65, 129, 300, 1032])
    small = synth.make_corpus(3, 2, seed=83, lengths=lambda r: next(lens))
    cases = [(g["feats"], g["frame_off"], g["gchroma"], g["pairs"]),
             (small.feats, small.frame_off, small.gchroma, np.array([(i, j) for i in range(6) for j in range(6)], dtype=np.int32))]
    for feats, off, gc, pairs in cases:
        corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device)
        eng.oti(corpus, batch)
        T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
        mats, _ = batch.mats()
        for mutual in (True, False):
            B = eng.binarize(T, batch, 0.095, mutual=mutual)
            bits, _ = eng.mask_bits(T, batch, 0.095, mutual=mutual)
            a = eng.align_bits("swc", bits, batch).cpu().numpy()
            b = eng.align("swc", B, mats).cpu().numpy()
            assert np.array_equal(a, b), mutual


def test_full_size_properties(eng, orc):
    """BASELINE config 2 shape (1000-frame songs, 992 x 992 cross-recurrence plots, k = 94) through properties that do not
    need a full oracle run: exactly k ones per row / column of the one-sided masks, mutual = AND of the two, scores on
    the 0.5 grid, transpose symmetry without OTI, fast path == staged path, and the oracle on a few pairs."""
    from acoss_amd import synth
    ch = synth.make_corpus(6, 2, n_frames=1000, seed=20260)
    corpus = eng.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = synth.all_pairs(ch.n_songs)[:48]
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    xp = eng.pack_x(corpus, batch)                       # no OTI: shifts stay 0
    planes = eng.crp_planar(corpus, batch, xp)
    rows_only, _ = eng.mask_bits_planar(planes, corpus, batch, 0.095, mutual=False)
    mutual, _ = eng.mask_bits_planar(planes, corpus, batch, 0.095, mutual=True)
    swapped = eng.PairBatch(corpus.frame_off, pairs[:, ::-1].copy(), 9, corpus.device, pitch_align=32)
    planes_t = eng.crp_planar(corpus, swapped, eng.pack_x(corpus, swapped))
    cols_only_t, _ = eng.mask_bits_planar(planes_t, corpus, swapped, 0.095, mutual=False)       # rows of the transposed pair = columns
    for p in range(0, 48, 5):
        R = eng.unpack_mask_bits(rows_only, batch, p)
        Ct = eng.unpack_mask_bits(cols_only_t, swapped, p)
        Mu = eng.unpack_mask_bits(mutual, batch, p)
        assert R.shape == (992, 992) and np.all(R.sum(1) == 94) and np.all(Ct.sum(1) == 94)
        assert np.array_equal(Mu, R & Ct.T)
    q = eng.align_bits("qmax", mutual, batch).cpu().numpy()
    d = eng.align_bits("dmax", mutual, batch, boundary=1).cpu().numpy()
    assert np.all(q * 2 == np.round(q * 2)) and np.all(d * 2 == np.round(d * 2)) and np.all(d >= q)
    mutual_t, _ = eng.mask_bits_planar(planes_t, corpus, swapped, 0.095, mutual=True)
    assert np.array_equal(eng.align_bits("qmax", mutual_t, swapped).cpu().numpy(), q)
    fast = eng.serra09_scores(corpus, pairs, do_oti=True)
    staged = eng.serra09_scores_staged(corpus, pairs, do_oti=True)
    assert np.array_equal(fast["qmax"], staged["qmax"]) and np.array_equal(fast["dmax"], staged["dmax"])
    oq, od, _ = orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, pairs[:6], nthreads=6)
    assert np.array_equal(fast["qmax"][:6], oq) and np.array_equal(fast["dmax"][:6], od)


def test_wide_forms_up_to_2048(eng, orc):
    """Matrices between 1024 x 1024 and 2048 x 2048 (songs of 1033 .. 2056 frames): the 32-values-per-lane forms of
    the selection kernels, the 32-word bit planes and the alignment kernels against the byte-mask path (bit-serial
    selection on the float64 matrix, `dp_wave_kernel`), incl. the crafted tie cases; then the chain against the
    oracle on two pairs."""
    import torch
    from acoss_amd import synth
    lens = iter([2056, 1033, 300, 1500, 9, 1990])
    ch = synth.make_corpus(3, 2, seed=611, lengths=lambda r: next(lens))
    rng = np.random.default_rng(6)
    pat7 = rng.random((7, 12)) + 0.1
    A = np.tile(pat7, (200, 1))[:1300]                       # periodic: exact ties in every row and column
    C = A + 1e-11 * rng.random(A.shape)                      # shared high words
    feats = np.concatenate([ch.feats, A, C])
    off = np.concatenate([ch.frame_off, ch.frame_off[-1] + np.cumsum([len(A), len(C)])]).astype(np.int64)
    gc = np.concatenate([ch.gchroma, np.stack([x.sum(0) / x.sum(0).max() for x in (A, C)])])
    corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
    pairs = np.array([(0, 1), (1, 0), (2, 3), (3, 5), (5, 0), (4, 0), (0, 4), (0, 0), (6, 7), (7, 6), (6, 6), (1, 6), (3, 3)], dtype=np.int32)
    for align in (32, 1):
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=align)
        assert eng.bits_words(batch) == 32 and eng.planar_supported(corpus, batch)
        eng.oti(corpus, batch)
        xp = eng.pack_x(corpus, batch)
        T = eng.crp(corpus, batch, xp)
        planes = eng.crp_planar(corpus, batch, xp)
        mats, _ = batch.mats()
        for mutual in (True, False):
            for kappa in ((0.095, 0.5, 3, 0) if align == 32 else (0.095,)):
                B = eng.binarize(T, batch, kappa, mutual=mutual)
                bits, _ = eng.mask_bits_planar(planes, corpus, batch, kappa, mutual=mutual)
                for p in range(batch.K):
                    d = batch.descs[p]
                    M, N = int(batch.M[p]), int(batch.N[p])
                    want = B[int(d["crp_off"]):int(d["crp_off"]) + M * int(d["crp_pitch"])].cpu().numpy().reshape(M, -1)[:, :N]
                    assert np.array_equal(eng.unpack_mask_bits(bits, batch, p), want), (align, mutual, kappa, p)
                if kappa != 0.095:
                    continue
                for kind, kw in (("qmax", {}), ("dmax", {}), ("dmax", {"boundary": 1}), ("swc", {})):
                    a = eng.align_bits(kind, bits, batch, **kw).cpu().numpy()
                    b = eng.align(kind, B, mats, **kw).cpu().numpy()
                    assert np.array_equal(a, b), (align, mutual, kind, kw)
                q, dm = eng.align_bits_qd(bits, batch, boundary=1)
                assert torch.equal(q, eng.align_bits("qmax", bits, batch)) and torch.equal(dm, eng.align_bits("dmax", bits, batch, boundary=1))
    got = eng.serra09_scores(corpus, pairs[[0, 3]])
    for t, (i, j) in enumerate(pairs[[0, 3]]):
        q, dm = orc.serra09_pair(feats[off[i]:off[i + 1]], gc[i], feats[off[j]:off[j + 1]], gc[j])
        assert got["qmax"][t] == q and got["dmax"][t] == dm, (i, j)


def test_float32_approximate_keys_give_identical_masks(eng, golden):
    """crp_planar32 (float32 windowed sums, bound 2^-24 * (16.5 * window norm sums + 9.5 * value)) + mask_bits_planar32 (error-band check,
    exact float64 refinement inside the band) == mask_bits on the float64 matrix, bit for bit: golden 1000-frame pairs,
    ragged small pairs, crafted exact ties and 1e-11 perturbations (everything inside the band), 1033 .. 2056-frame songs;
    the approximation stays inside its bound."""
    import torch
    from acoss_amd import synth
    g = golden("pairs_1000")
    lens = iter([9, 40, 65, 129, 300, 1032])
    small = synth.make_corpus(3, 2, seed=83, lengths=lambda r: next(lens))
    rng = np.random.default_rng(5)
    pat7, pat5 = rng.random((7, 12)) + 0.1, rng.random((5, 12)) + 0.1
    A = np.tile(pat7, (30, 1))[:200]
    Bs = np.tile(pat5, (31, 1))[:151]
    C = A + 1e-11 * rng.random(A.shape)
    Dn = Bs + 1e-11 * rng.random(Bs.shape)
    tie_feats = np.concatenate([A, Bs, C, Dn])
    tie_off = np.cumsum([0, len(A), len(Bs), len(C), len(Dn)]).astype(np.int64)
    tie_gc = np.stack([x.sum(0) / x.sum(0).max() for x in (A, Bs, C, Dn)])
    wl = iter([2056, 1033, 300, 1500])
    wide = synth.make_corpus(2, 2, seed=611, lengths=lambda r: next(wl))
    cases = [(g["feats"], g["frame_off"], g["gchroma"], g["pairs"], (0.095, 0.5, 3, 0)),
             (small.feats, small.frame_off, small.gchroma, np.array([(i, j) for i in range(6) for j in range(6)], dtype=np.int32), (0.095, 0.5, 3, 0)),
             (tie_feats, tie_off, tie_gc, np.array([(i, j) for i in range(4) for j in range(4)], dtype=np.int32), (0.095,)),
             (wide.feats, wide.frame_off, wide.gchroma, np.array([(0, 1), (1, 0), (2, 3), (3, 0), (0, 0)], dtype=np.int32), (0.095,))]
    for ci, (feats, off, gc, pairs, kappas) in enumerate(cases):
        corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
        for align in (32, 1):
            batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=align)
            eng.oti(corpus, batch)
            T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
            keys = eng.crp_planar32(corpus, batch, eng.pack_x32(corpus, batch))
            band = eng.planar32_band(corpus, batch)
            # the approximation against its bound (per pair: band / 2)
            Th, Kh, bh = T.cpu().numpy(), keys.cpu().numpy().view(np.uint32), band.cpu().numpy().astype(np.float64)
            for p in range(batch.K):
                d = batch.descs[p]
                M, N = int(d["nx"]) - 8, int(d["ny"]) - 8
                idx = (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)
                approx = (Kh[idx] & 0x7fffffff).astype(np.uint32).view(np.float32).astype(np.float64)
                Ts = Th[idx] * corpus._f32_scale2          # the kernel works on a power-of-two rescaled copy
                bound = (bh[2 * p] + bh[2 * p + 1] * Ts) / 2
                assert np.all(Kh[idx] >> 31 == 1) and np.all(np.abs(approx - Ts) <= bound), (ci, p)
            planes = eng.crp_planar(corpus, batch, eng.pack_x(corpus, batch)) if ci < 3 else None
            for mutual in (True, False):
                for kappa in kappas:
                    got, _ = eng.mask_bits_planar32(keys, band, corpus, batch, kappa, mutual=mutual)
                    if planes is not None:
                        want, _ = eng.mask_bits_planar(planes, corpus, batch, kappa, mutual=mutual)
                        for p in range(batch.K):
                            assert np.array_equal(eng.unpack_mask_bits(got, batch, p), eng.unpack_mask_bits(want, batch, p)), (ci, align, mutual, kappa, p)
                    else:
                        B = eng.binarize(T, batch, kappa, mutual=mutual)
                        for p in range(batch.K):
                            d = batch.descs[p]
                            M, N = int(batch.M[p]), int(batch.N[p])
                            want = B[int(d["crp_off"]):int(d["crp_off"]) + M * int(d["crp_pitch"])].cpu().numpy().reshape(M, -1)[:, :N]
                            assert np.array_equal(eng.unpack_mask_bits(got, batch, p), want), (ci, align, mutual, kappa, p)


def test_float32_filter_with_useless_approximation(eng):
    """Features with a large per-bin offset (norms ~1e6, distances ~20): the float32 approximation is worthless, its
    error band covers whole rows (more than 64 band elements: the general refinement path), and the masks still equal the
    float64 path's; also 13-dimensional signed features (the MFCC shape), no OTI."""
    from acoss_amd import synth
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
        keys = eng.crp_planar32(corpus, batch, eng.pack_x32(corpus, batch))
        band = eng.planar32_band(corpus, batch)
        Th, Kh, bh = T.cpu().numpy(), keys.cpu().numpy().view(np.uint32), band.cpu().numpy().astype(np.float64)
        for p in range(batch.K):
            d = batch.descs[p]
            M, N = int(d["nx"]) - 8, int(d["ny"]) - 8
            idx = (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)
            approx = (Kh[idx] & 0x7fffffff).astype(np.uint32).view(np.float32).astype(np.float64)
            Ts = Th[idx] * corpus._f32_scale2
            assert np.all(np.abs(approx - Ts) <= (bh[2 * p] + bh[2 * p + 1] * Ts) / 2), p
        for mutual in (True, False):
            want, _ = eng.mask_bits(T, batch, 0.095, mutual=mutual)
            got, _ = eng.mask_bits_planar32(keys, band, corpus, batch, 0.095, mutual=mutual)
            for p in range(batch.K):
                assert np.array_equal(eng.unpack_mask_bits(got, batch, p), eng.unpack_mask_bits(want, batch, p)), (do_oti, mutual, p)


def test_float32_strip_kernel_is_a_round_to_nearest_fma_chain(eng):
    """The error bound of the float32 filter assumes that v_mfma_f32_16x16x4_f32 accumulates like a chain of
    round-to-nearest FMAs over k (as the float64 form does).  Pinned here bit for bit: the kernel's keys equal a host
    emulation of its arithmetic in float32 (an FMA = one rounding of the exact product-sum; the product of two float32
    is exact in float64), window sums included (pairwise, shared between the rows of a wave: kernel_utils.h)."""
    from acoss_amd import synth
    lens = iter([200, 150, 173])
    ch = synth.make_corpus(3, 1, seed=5, lengths=lambda r: next(lens))
    corpus = eng.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = np.array([(0, 1), (1, 2), (2, 0)], dtype=np.int32)
    b = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    eng.oti(corpus, b)
    xp32 = eng.pack_x32(corpus, b)
    keys = eng.crp_planar32(corpus, b, xp32).cpu().numpy().view(np.uint32)
    f32, n32 = [t.cpu().numpy() for t in eng.float32_copy(corpus)]
    shifts = b.descs_dev.cpu().numpy().view(eng.PAIR_DESC)["shift"]

    def fma32(a, bb, c):
        return (a.astype(np.float64) * bb.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    for p in range(b.K):
        d = b.descs[p]
        nx, ny, sh = int(d["nx"]), int(d["ny"]), int(shifts[p])
        X = np.roll(f32[int(d["x_row0"]):int(d["x_row0"]) + nx], sh, axis=1)
        Y = f32[int(d["y_row0"]):int(d["y_row0"]) + ny]
        acc = np.zeros((nx, ny), dtype=np.float32)
        for bin_ in range(12):
            acc = fma32(X[:, bin_][:, None], Y[:, bin_][None, :], acc)
        nsum = (n32[int(d["x_row0"]):int(d["x_row0"]) + nx][:, None] + n32[int(d["y_row0"]):int(d["y_row0"]) + ny][None, :]).astype(np.float32)
        C = np.maximum(fma32(np.full_like(acc, -2.0), acc, nsum), np.float32(0))
        M, N = nx - 8, ny - 8
        # window sums as window_sum9() of csrc/kernel_utils.h associates them: by the row, (i mod 7) & 1
        c = [C[k:k + M, k:k + N] for k in range(9)]
        even = c[0] + (((c[1] + c[2]) + (c[3] + c[4])) + ((c[5] + c[6]) + (c[7] + c[8])))
        odd = (((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]))) + c[8]
        assert even.dtype == np.float32 and odd.dtype == np.float32
        T = np.where((((np.arange(M) % 7) & 1) == 1)[:, None], odd, even)
        idx = (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)
        assert np.array_equal(keys[idx] & 0x7fffffff, T.view(np.uint32)), p


def test_float32_filter_is_scale_free(eng, orc):
    """Corpora of very large and very small magnitude (features x 1e25, x 1e-25: squared norms beyond float32's range
    either way): the filter works on a power-of-two rescaled copy, the scores equal the oracle's on the original data."""
    from acoss_amd import synth
    lens = iter([300, 220, 410])
    ch = synth.make_corpus(3, 1, seed=19, lengths=lambda r: next(lens))
    pairs = np.array([(0, 1), (1, 2), (2, 0)], dtype=np.int32)
    for mag in (1e25, 1e-25):
        feats = ch.feats * mag
        corpus = eng.DeviceCorpus(feats, ch.frame_off, gchroma=ch.gchroma)
        got = eng.serra09_scores(corpus, pairs, approx32=True)
        for t, (i, j) in enumerate(pairs):
            q, d = orc.serra09_pair(feats[ch.frame_off[i]:ch.frame_off[i + 1]], ch.gchroma[i],
                                    feats[ch.frame_off[j]:ch.frame_off[j + 1]], ch.gchroma[j])
            assert got["qmax"][t] == q and got["dmax"][t] == d, (mag, t)


def test_float32_error_bound_under_adversarial_features(eng, orc):
    """The float32 filter's certificate, |T~ - T| <= 2^-24 (16.5 W + 9.5 T), on features built to stress it -- near-duplicate
    frames of large norm (every distance is a cancellation), two far-apart clusters (after centring: norms huge against the
    within-cluster distances), one dominant bin, heavy-tailed magnitudes, alternating signs (13-d MFCC-like) -- and, with the
    bound holding, the scores of the whole chain against the oracle.  Reports how much of the bound the worst cell uses."""
    rng = np.random.default_rng(2026)
    n = 230

    def chroma_like(x):
        return np.abs(x) + 1e-3

    base = rng.lognormal(0.0, 2.0, (n, 12))
    fams = {
        "near_duplicates": [chroma_like(base * (1.0 + 1e-4 * rng.standard_normal((n, 12)))) for _ in range(3)],
        "two_clusters": [chroma_like(np.where((np.arange(n) % 2 == s)[:, None], 100.0, 1.0) * (1.0 + 1e-3 * rng.random((n, 12)))) for s in range(3)],
        "dominant_bin": [chroma_like(np.concatenate([1e3 * (1 + 1e-5 * rng.random((n, 1))), 1e-2 * rng.random((n, 11))], axis=1)) for _ in range(3)],
        "heavy_tail": [chroma_like(rng.lognormal(0.0, 3.0, (n, 12))) for _ in range(3)],
    }
    worst = {}
    for name, songs in fams.items():
        feats = np.concatenate(songs)
        off = np.arange(len(songs) + 1, dtype=np.int64) * n
        gc = np.stack([s.sum(0) / s.sum(0).max() for s in songs])
        corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
        pairs = np.array([(0, 1), (1, 2), (2, 0), (1, 1)], dtype=np.int32)
        batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
        eng.oti(corpus, batch)
        if not eng.keys16_supported(corpus, batch):
            continue                                   # (the engine itself refuses the filter for this corpus: float64 path)
        T = eng.crp(corpus, batch, eng.pack_x(corpus, batch)).cpu().numpy()
        Kh = eng.crp_planar32(corpus, batch, eng.pack_x32(corpus, batch)).cpu().numpy().view(np.uint32)
        bh = eng.planar32_band(corpus, batch).cpu().numpy().astype(np.float64)
        ratio = 0.0
        for p in range(batch.K):
            d = batch.descs[p]
            M, N = int(d["nx"]) - 8, int(d["ny"]) - 8
            idx = (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)
            approx = (Kh[idx] & 0x7fffffff).astype(np.uint32).view(np.float32).astype(np.float64)
            Ts = T[idx] * corpus._f32_scale2
            bound = (bh[2 * p] + bh[2 * p + 1] * Ts) / 2
            ratio = max(ratio, float(np.max(np.abs(approx - Ts) / bound)))
        worst[name] = ratio
        assert ratio <= 1.0, (name, ratio)
        got = eng.serra09_scores(corpus, pairs)
        q, dm, _ = orc.serra09_pairs(feats, off, gc, pairs, nthreads=4)
        assert np.array_equal(got["qmax"], q) and np.array_equal(got["dmax"], dm), name
    print("worst |T~ - T| / bound per family:", {k: round(v, 3) for k, v in worst.items()})
    assert len(worst) >= 3
