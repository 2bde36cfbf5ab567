"""
GPU parity tests of the stage kernels (through the C ABI) against the golden vectors produced by
the reference and against the CPU oracle on seeded inputs.  Integer / mask / D results must be
bit-exact; float64 CSM and sliding values are compared at 1e-9 (contract: 1e-5); the constrained
Smith-Waterman at 1e-5 (the reference is built -Ofast and -0.7 is inexact).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from acoss_amd import engine
    engine.require_gpu()
    return engine


def _pair_corpus(eng, X, Y, gX=None, gY=None):
    feats = np.concatenate([X, Y], axis=0)
    off = np.array([0, X.shape[0], X.shape[0] + Y.shape[0]], dtype=np.int64)
    g = None if gX is None else np.stack([gX, gY])
    return eng.DeviceCorpus(feats, off, gchroma=g)


def _unpack(buf, batch, p, what):
    """host view of pair p's matrix out of a flat device buffer"""
    d = batch.descs[p]
    if what == "csm":
        rows, cols, off, pitch = d["nx"], d["ny"], d["csm_off"], d["csm_pitch"]
    else:
        rows, cols, off, pitch = d["nx"] - batch.win + 1, d["ny"] - batch.win + 1, d["crp_off"], d["crp_pitch"]
    flat = buf[off:off + rows * pitch].cpu().numpy()
    return flat.reshape(rows, pitch)[:, :cols]


def test_oti_batch(eng, golden):
    g = golden("stages")
    G1, G2 = g["oti_G1"], g["oti_G2"]
    n = len(G1)
    corpus = eng.DeviceCorpus(np.zeros((2 * n, 12)), np.arange(2 * n + 1, dtype=np.int64),
                              gchroma=np.concatenate([G1, G2]))
    pairs = np.stack([np.arange(n), np.arange(n) + n], axis=1).astype(np.int32)
    batch = eng.PairBatch(corpus.frame_off, pairs, 1, corpus.device)
    eng.oti(corpus, batch)
    assert np.array_equal(batch.fetch_shifts(), g["oti_expected"])
    from acoss_amd import CRPUtils
    assert CRPUtils.get_oti(G1[5], G2[5]) == 5


@pytest.mark.parametrize("c", [0, 1, 2])
def test_stage_chain_against_reference(eng, golden, orc, c):
    g = golden("stages")
    p = "c%d_" % c
    X, Y, m, kappa = g[p + "X"], g[p + "Y"], int(g[p + "m"]), float(g[p + "kappa"])
    corpus = _pair_corpus(eng, X, Y, g[p + "gX"], g[p + "gY"])
    batch = eng.PairBatch(corpus.frame_off, [[0, 1]], m, corpus.device)
    eng.oti(corpus, batch)
    assert int(batch.fetch_shifts()[0]) == int(g[p + "oti"])
    C = eng.csm(corpus, batch)
    csm = _unpack(C, batch, 0, "csm")
    assert np.max(np.abs(csm - g[p + "CSM"])) <= 1e-9
    S = eng.sliding(C, batch)
    Sh = _unpack(S, batch, 0, "crp")
    assert np.max(np.abs(Sh - g[p + "S"])) <= 1e-9
    B1 = _unpack(eng.binarize(S, batch, kappa, mutual=False), batch, 0, "crp")
    assert np.array_equal(B1, g[p + "B1"])
    Bbuf = eng.binarize(S, batch, kappa, mutual=True)
    B = _unpack(Bbuf, batch, 0, "crp")
    assert np.array_equal(B, g[p + "B"])
    M, N = B.shape
    sc = g[p + "scores"]
    # scores-only batched path (what Serra09.similarity uses)
    mats, _ = batch.mats()
    q = float(eng.align("qmax", Bbuf, mats).cpu()[0])
    d = float(eng.align("dmax", Bbuf, mats, boundary=1).cpu()[0])
    df = float(eng.align("dmax", Bbuf, mats, boundary=0).cpu()[0])
    assert q / (M + N) == sc[0] and d / (M + N) == sc[1] and df / (M + N) == sc[2]
    # D-producing path with the reference's in-place semantics
    import torch
    mats_d, total = batch.mats(with_d=True)
    D = torch.zeros(total, dtype=torch.float32, device=corpus.device)
    assert float(eng.align("qmax", Bbuf, mats_d, D=D).cpu()[0]) == q
    assert np.array_equal(D.cpu().numpy().reshape(M, N), g[p + "Dq"])
    assert float(eng.align("dmax", Bbuf, mats_d, D=D).cpu()[0]) == d      # D reused, Serra09.py:173-175
    assert np.array_equal(D.cpu().numpy().reshape(M, N), g[p + "Dd_reused"])
    D.zero_()
    assert float(eng.align("dmax", Bbuf, mats_d, D=D).cpu()[0]) == df
    assert np.array_equal(D.cpu().numpy().reshape(M, N), g[p + "Dd_fresh"])
    mats_w, total_w = batch.mats(with_d=True, sw=True)
    Dw = torch.zeros(total_w, dtype=torch.float32, device=corpus.device)
    w = float(eng.align("swc", Bbuf, mats_w, D=Dw).cpu()[0])
    assert abs(w - sc[3]) <= 1e-5
    assert np.max(np.abs(Dw.cpu().numpy().reshape(M + 1, N + 1) - g[p + "Dsw_mutual"])) <= 1e-5


def test_crputils_mirror_functions(golden):
    from acoss_amd import CRPUtils
    g = golden("stages")
    p = "c1_"
    X, Y, m, kappa, oti = g[p + "X"], g[p + "Y"], int(g[p + "m"]), float(g[p + "kappa"]), int(g[p + "oti"])
    csm = CRPUtils.get_csm(X, Y, shift=oti)
    assert csm.dtype == np.float64 and csm.shape == (X.shape[0], Y.shape[0])
    assert np.max(np.abs(csm - g[p + "CSM"])) <= 1e-9
    # rolling X on the host, as Serra09.py:167 does, is the same thing (up to the order in which
    # the rolled frame's squared norm is accumulated)
    assert np.max(np.abs(CRPUtils.get_csm(np.roll(X, oti, axis=1), Y) - csm)) <= 1e-12
    S = CRPUtils.sliding_csm(g[p + "CSM"], m)
    assert S.dtype == np.float64 and np.max(np.abs(S - g[p + "S"])) <= 1e-9
    assert np.array_equal(CRPUtils.csm_to_binary(g[p + "S"], kappa), g[p + "B1"])
    assert np.array_equal(CRPUtils.csm_to_binary_mutual(g[p + "S"], kappa), g[p + "B"])
    assert CRPUtils.csm_to_binary(g[p + "S"], kappa).dtype == np.uint8


def test_crputils_functions_off_the_serra09_path(golden):
    """get_ssm (CRPUtils.py:48-65), get_csm_cosine (:88-107), sliding_window (:8-22) against the reference's own outputs."""
    from acoss_amd import CRPUtils
    g = golden("crputils_extra")
    X, Y = g["X"], g["Y"]
    ssm = CRPUtils.get_ssm(X)
    assert ssm.shape == (57, 57) and np.all(np.diag(ssm) == 0) and np.max(np.abs(ssm - g["ssm"])) <= 1e-9
    cos = CRPUtils.get_csm_cosine(X, Y)
    assert np.max(np.abs(cos - g["cosine"])) <= 1e-9 and np.all(cos[11] == 1.0) and np.all(cos[:, 7] == 1.0)
    assert np.array_equal(CRPUtils.sliding_window(X, 9), g["window9"]) and np.array_equal(CRPUtils.sliding_window(Y, 1), g["window1"])
    import acoss_amd.CRPUtils as mod
    for name in ("sliding_window", "sliding_csm", "get_ssm", "get_csm", "get_csm_euclidean", "get_csm_cosine", "get_oti",
                 "get_csm_blocked_oti", "csm_to_binary", "csm_to_binary_mutual"):       # every def of the reference's module
        assert callable(getattr(mod, name))


def test_float32_inputs(golden):
    from acoss_amd import CRPUtils
    g = golden("stages")
    csm = CRPUtils.get_csm(g["f32_X"], g["f32_Y"])
    assert csm.dtype == np.float32                       # dtype follows the inputs (CRPUtils.py:82)
    assert np.max(np.abs(csm - g["f32_CSM"])) <= 2e-6
    S = CRPUtils.sliding_csm(g["f32_CSM"], 9)            # promoted to float64 (CRPUtils.py:40-41)
    assert S.dtype == np.float64 and np.max(np.abs(S - g["f32_S"])) <= 1e-9
    assert np.array_equal(CRPUtils.csm_to_binary_mutual(g["f32_S"], 0.095), g["f32_B"])


def test_kappa_conventions(golden):
    from acoss_amd import CRPUtils
    g = golden("stages")
    D = g["kap_D"]
    assert np.array_equal(CRPUtils.csm_to_binary(D, 0.25), g["kap_B_frac"])     # round(12.5) = 12
    assert np.array_equal(CRPUtils.csm_to_binary(D, 0.11), g["kap_B_frac2"])    # round(5.5) = 6
    assert np.array_equal(CRPUtils.csm_to_binary(D, 7), g["kap_B_int"])
    assert np.array_equal(CRPUtils.csm_to_binary_mutual(D, 7), g["kap_Bm_int"])
    assert np.array_equal(CRPUtils.csm_to_binary_mutual(D, 0.25), g["kap_Bm_frac"])
    assert np.all(CRPUtils.csm_to_binary(D, 0) == 1)


def test_binarize_ties_and_negative_values(orc):
    """Exact ties (quantised values, zero-padded frames) and negative entries (EarlySNF feeds -W):
    the GPU resolves ties lowest-index first, exactly like the oracle."""
    from acoss_amd import CRPUtils
    rng = np.random.default_rng(3)
    # (beyond 2048 rows / columns: the any-size radix selection, select_generic_kernel)
    for shape, kappa in [((37, 53), 0.2), ((64, 64), 0.1), ((100, 129), 9), ((30, 2500), 0.1), ((2500, 30), 0.3),
                         ((2100, 2060), 0.095), ((2049, 2049), 700)]:
        D = np.round(rng.standard_normal(shape) * 2) / 2      # many exact ties, both signs, +-0
        D[3, :] = 0.0
        D[:, 5] = -0.0
        assert np.array_equal(CRPUtils.csm_to_binary(D, kappa), orc.csm_to_binary(D, kappa))
        assert np.array_equal(CRPUtils.csm_to_binary_mutual(D, kappa), orc.csm_to_binary_mutual(D, kappa))


def test_alignment_golden_cases_through_pyseqalign(golden):
    """The reference's native interface (host pointers, in-place D) on the GPU."""
    from acoss_amd import pySeqAlign
    g = golden("dp_cases")
    for k in range(int(g["n_cases"])):
        p = "k%d_" % k
        S = g[p + "S"]
        M, N = S.shape
        Sf = np.ascontiguousarray(S.flatten())
        sc = g[p + "scores"]
        D = np.zeros(M * N, dtype=np.float32)
        assert pySeqAlign.qmax(Sf, D, M, N) == sc[0], (k, M, N)
        assert np.array_equal(D.reshape(M, N), g[p + "Dq"]), (k, M, N)
        assert pySeqAlign.dmax(Sf, D, M, N) == sc[1], (k, M, N)
        assert np.array_equal(D.reshape(M, N), g[p + "Dd_reused"]), (k, M, N)
        Df = np.zeros(M * N, dtype=np.float32)
        assert pySeqAlign.dmax(Sf, Df, M, N) == sc[2]
        assert np.array_equal(Df.reshape(M, N), g[p + "Dd_fresh"])
        Dw = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        assert abs(pySeqAlign.swconstrained(Sf, Dw, M, N) - sc[3]) <= 1e-5
        assert np.max(np.abs(Dw.reshape(M + 1, N + 1) - g[p + "Dsw"])) <= 1e-5
    for tag in ("ones", "zeros", "eye"):
        S = g["ka_" + tag + "_S"]
        M, N = S.shape
        Sf = np.ascontiguousarray(S.flatten())
        sc = g["ka_" + tag + "_scores"]
        assert pySeqAlign.qmax(Sf, np.zeros(M * N, np.float32), M, N) == sc[0]
        assert pySeqAlign.dmax(Sf, np.zeros(M * N, np.float32), M, N) == sc[1]
        assert abs(pySeqAlign.swconstrained(Sf, np.zeros((M + 1) * (N + 1), np.float32), M, N) - sc[2]) <= 1e-5


def test_alignment_batch_scores_only_and_boundary_modes(eng, golden):
    """All golden masks in ONE launch per recurrence, D == NULL, dmax with both boundaries."""
    import torch
    from acoss_amd._lib import MAT_DESC
    g = golden("dp_cases")
    n = int(g["n_cases"])
    chunks, mats = [], np.zeros(n, dtype=MAT_DESC)
    off = 0
    for k in range(n):
        S = g["k%d_S" % k]
        M, N = S.shape
        pitch = (N + 15) // 16 * 16
        buf = np.zeros((M, pitch), dtype=np.uint8)
        buf[:, :N] = S
        chunks.append(buf.reshape(-1))
        mats[k] = (off, 0, M, N, pitch, 0)
        off += M * pitch
    Sdev = torch.from_numpy(np.concatenate(chunks)).cuda()
    exp = np.stack([g["k%d_scores" % k] for k in range(n)])
    assert np.array_equal(eng.align("qmax", Sdev, mats).cpu().numpy(), exp[:, 0].astype(np.float32))
    assert np.array_equal(eng.align("dmax", Sdev, mats, boundary=1).cpu().numpy(), exp[:, 1].astype(np.float32))
    assert np.array_equal(eng.align("dmax", Sdev, mats, boundary=0).cpu().numpy(), exp[:, 2].astype(np.float32))
    assert np.max(np.abs(eng.align("swc", Sdev, mats).cpu().numpy() - exp[:, 3])) <= 1e-5


@pytest.mark.parametrize("shape", [(40, 1024), (37, 1025), (23, 2048), (19, 2049), (12, 3000)])
def test_alignment_wide_matrices(orc, shape):
    """Column counts around the kernel-variant boundaries (16 / 32 columns per lane, then the
    workgroup-per-matrix kernel), unaligned pitches included."""
    from acoss_amd import pySeqAlign
    rng = np.random.default_rng(shape[1])
    M, N = shape
    S = (rng.random((M, N)) < 0.3).astype(np.uint8)
    Sf = np.ascontiguousarray(S.flatten())
    Dg, Do = np.zeros(M * N, np.float32), np.zeros(M * N, np.float32)
    assert pySeqAlign.qmax(Sf, Dg, M, N) == orc.qmax(Sf, Do, M, N)
    assert np.array_equal(Dg, Do)
    assert pySeqAlign.dmax(Sf, Dg, M, N) == orc.dmax(Sf, Do, M, N)
    assert np.array_equal(Dg, Do)
    Wg, Wo = np.zeros((M + 1) * (N + 1), np.float32), np.zeros((M + 1) * (N + 1), np.float32)
    assert abs(pySeqAlign.swconstrained(Sf, Wg, M, N) - orc.swconstrained(Sf, Wo, M, N)) <= 1e-5
    assert np.max(np.abs(Wg - Wo)) <= 1e-5


def test_custom_gap_penalties(eng, orc):
    """Penalties are kernel arguments; with onset != extension the mask history matters."""
    import torch
    from acoss_amd._lib import MAT_DESC, AlignParams
    rng = np.random.default_rng(8)
    M, N = 60, 80
    S = (rng.random((M, N)) < 0.2).astype(np.uint8)
    prm = AlignParams(0.5, 0.5, 1.0, -1.0, -0.5, -0.7)
    mats = np.zeros(1, dtype=MAT_DESC)
    mats[0] = (0, 0, M, N, N, 0)
    Sd = torch.from_numpy(S.reshape(-1)).cuda()
    base = float(eng.align("qmax", Sd, mats, params=prm).cpu()[0])
    assert base == orc.qmax(np.ascontiguousarray(S.flatten()), np.zeros(M * N, np.float32), M, N)
    prm2 = AlignParams(1.0, 0.25, 1.0, -1.0, -0.5, -0.7)
    other = float(eng.align("qmax", Sd, mats, params=prm2).cpu()[0])
    # brute-force restatement with the changed penalties
    D = np.zeros((M, N), np.float32)
    for i in range(2, M):
        for j in range(2, N):
            if S[i, j] == 1:
                D[i, j] = max(D[i - 1, j - 1], D[i - 2, j - 1], D[i - 1, j - 2]) + 1
            else:
                gam = lambda s: 1.0 if s == 1 else 0.25
                D[i, j] = max(0.0, D[i - 1, j - 1] - gam(S[i - 1, j - 1]), D[i - 2, j - 1] - gam(S[i - 2, j - 1]),
                              D[i - 1, j - 2] - gam(S[i - 1, j - 2]))
    assert other == float(D.max())


def test_serra09_mini_corpus_scores(eng, golden):
    g = golden("serra09_mini")
    corpus = eng.DeviceCorpus(g["feats"], g["frame_off"], gchroma=g["gchroma"])
    res = eng.serra09_scores_staged(corpus, g["pairs"], m=9, kappa=0.095, batch_pairs=25)
    assert np.array_equal(res["qmax"], g["chroma_qmax"])
    assert np.array_equal(res["dmax"], g["chroma_dmax"])
    # MFCC chain: float32 13-d features, no OTI (Serra09.py:178-184)
    mf = eng.DeviceCorpus(g["mfcc"], g["frame_off"])
    assert mf.dtype == np.float32
    res = eng.serra09_scores_staged(mf, g["pairs"], m=9, kappa=0.095, do_oti=False)
    assert np.array_equal(res["qmax"], g["mfcc_qmax"])
    assert np.array_equal(res["dmax"], g["mfcc_dmax"])


def test_pairs_1000_frames(eng, golden):
    g = golden("pairs_1000")
    corpus = eng.DeviceCorpus(g["feats"], g["frame_off"], gchroma=g["gchroma"])
    keep = {}
    res = eng.serra09_scores_staged(corpus, g["pairs"], keep=keep)
    assert np.array_equal(res["qmax"], g["chroma_qmax"])
    assert np.array_equal(res["dmax"], g["chroma_dmax"])
    batch = keep["batch"]
    assert np.array_equal(batch.fetch_shifts(), g["oti"])
    for t in range(3):
        B = _unpack(keep["B"], batch, t, "crp")
        assert B.shape == (992, 992) and np.all(B.sum(axis=1) <= 94)
        assert np.array_equal(np.packbits(B, axis=1), g["B_packed_%d" % t])
        S = _unpack(keep["S"], batch, t, "crp")
        assert np.max(np.abs(np.diag(S) - g["S_diag_%d" % t])) <= 1e-9
        assert np.max(np.abs(_unpack(keep["C"], batch, t, "csm")[0] - g["CSM_row0_%d" % t])) <= 1e-9


def test_ragged_batch_against_oracle(eng, orc):
    """Ragged lengths in one batch, including a song exactly as long as the window and i == j
    pairs (do_batch includes the diagonal, CoverAlgorithm.py:244)."""
    from acoss_amd import synth
    lens = iter([9, 10, 12, 33, 64, 65, 100, 131])
    corpus_h = synth.make_corpus(4, 2, seed=77, lengths=lambda r: next(lens))
    corpus = eng.DeviceCorpus(corpus_h.feats, corpus_h.frame_off, gchroma=corpus_h.gchroma)
    pairs = np.array([(i, j) for i in range(8) for j in range(8)], dtype=np.int32)
    res = eng.serra09_scores_staged(corpus, pairs, m=9, kappa=0.095)
    q, d, _ = orc.serra09_pairs(corpus_h.feats, corpus_h.frame_off, corpus_h.gchroma, pairs, nthreads=4)
    assert np.array_equal(res["qmax"], q)
    assert np.array_equal(res["dmax"], d)


def test_wide_crp_selection_paths(eng, orc):
    """More than 1024 columns / rows: the 32-elements-per-lane selection and 32-column DP variants."""
    from acoss_amd import synth
    lens = iter([1100, 40, 1500, 1030])
    corpus_h = synth.make_corpus(2, 2, seed=78, lengths=lambda r: next(lens))
    corpus = eng.DeviceCorpus(corpus_h.feats, corpus_h.frame_off, gchroma=corpus_h.gchroma)
    pairs = np.array([[0, 1], [1, 0], [2, 3], [0, 3]], dtype=np.int32)
    res = eng.serra09_scores_staged(corpus, pairs)
    q, d, _ = orc.serra09_pairs(corpus_h.feats, corpus_h.frame_off, corpus_h.gchroma, pairs, nthreads=4)
    assert np.array_equal(res["qmax"], q)
    assert np.array_equal(res["dmax"], d)


def test_wide_float32_feature_csm_on_matrix_cores(eng, orc):
    """get_csm for float32 features wider than 32 (the reference keeps its scattering features in float32, Serra09.py:187-192;
    the CSM follows the dtype, CRPUtils.py:82): the v_mfma_f32_16x16x4_f32 kernel, quads (no roll, d % 4 == 0) and the
    element-wise loader (roll, odd d), ragged tiles.  Tolerance, on the SQUARED distances: a float32 dot product of d terms is
    within (d + 2) u |x||y| of the exact one whatever its order (u = 2^-24), the norms' sum and the final FMA add two more
    roundings: |c - c_exact| <= (d + 4) u (|x|^2 + |y|^2) against float64, twice that between two float32 evaluations."""
    rng = np.random.default_rng(29)
    u = 2.0 ** -24
    for d, shift, (M, N) in ((32, 0, (70, 131)), (52, 0, (129, 128)), (200, 0, (300, 17)), (333, 41, (70, 131)), (64, 5, (40, 260)),
                             (2048, 0, (150, 140))):
        X = (np.cumsum(rng.standard_normal((M, d)), axis=0) * 0.1).astype(np.float32)
        Y = (np.cumsum(rng.standard_normal((N, d)), axis=0) * 0.1).astype(np.float32)
        corpus = _pair_corpus(eng, X, Y)
        assert corpus.dtype == np.float32
        batch = eng.PairBatch(corpus.frame_off, np.array([[0, 1]], dtype=np.int32), 1, corpus.device)
        batch.set_shifts([shift])
        C = eng.csm(corpus, batch)
        assert C.dtype == eng.torch.float32
        dsc = batch.descs[0]
        got = C.cpu().numpy()[int(dsc["csm_off"]):int(dsc["csm_off"]) + M * int(dsc["csm_pitch"])].reshape(M, -1)[:, :N].astype(np.float64)
        Xr = np.roll(X.astype(np.float64), shift, axis=1)
        Y64 = Y.astype(np.float64)
        nx, ny = np.sum(Xr * Xr, axis=1), np.sum(Y64 * Y64, axis=1)
        exact = np.maximum(nx[:, None] + ny[None, :] - 2.0 * Xr.dot(Y64.T), 0.0)
        bound = (d + 4) * u * (nx[:, None] + ny[None, :])
        assert np.all(np.abs(got * got - exact) <= bound * (1 + 1e-6) + 4 * u * exact), (d, shift)      # + the rounding of sqrt, squared back
        want = orc.get_csm(X, Y, shift).astype(np.float64)
        assert np.all(np.abs(got * got - want * want) <= 2 * bound + 8 * u * exact), (d, shift)


def test_wide_feature_csm_on_matrix_cores(eng, orc):
    """get_csm for float64 features wider than 32 (the scattering-feature case, Serra09.py:187): the MFMA kernel
    against the oracle, with and without a roll, ragged sizes, d not a multiple of the k-chunk; and the m = 1 chain."""
    rng = np.random.default_rng(12)
    for d, shift in ((32, 0), (50, 7), (200, 0), (333, 41)):
        X = np.cumsum(rng.standard_normal((70, d)), axis=0) * 0.1
        Y = np.cumsum(rng.standard_normal((131, d)), axis=0) * 0.1
        corpus = _pair_corpus(eng, X, Y)
        batch = eng.PairBatch(corpus.frame_off, np.array([[0, 1]], dtype=np.int32), 1, corpus.device)
        batch.set_shifts([shift])
        C = eng.csm(corpus, batch).cpu().numpy()
        dsc = batch.descs[0]
        got = C[int(dsc["csm_off"]):int(dsc["csm_off"]) + 70 * int(dsc["csm_pitch"])].reshape(70, -1)[:, :131]
        want = orc.get_csm(X, Y, shift)
        assert np.max(np.abs(got - want)) <= 1e-9, (d, shift)
    # the Serra09 chain without window on wide features (ssms_scatter_* scores)
    A = np.cumsum(rng.standard_normal((90, 64)), axis=0) * 0.1
    B = np.cumsum(rng.standard_normal((75, 64)), axis=0) * 0.1
    corpus = _pair_corpus(eng, A, B)
    res = eng.serra09_scores(corpus, np.array([[0, 1], [1, 0]], dtype=np.int32), m=1, do_oti=False)
    for t, (x, y) in enumerate(((A, B), (B, A))):
        q, dm = orc.serra09_pair(x, np.zeros(64), y, np.zeros(64), m=1, do_oti=False)
        assert res["qmax"][t] == q and res["dmax"][t] == dm
