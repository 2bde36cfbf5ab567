"""
The CPU oracle (oracle/) against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare
the HIP path with the oracle and with the same fixtures.
"""
import numpy as np
import pytest


def test_np_sum_matches_numpy_bitwise(orc):
    rng = np.random.default_rng(0)
    for n in [1, 2, 7, 8, 9, 12, 13, 24, 36, 100, 128, 129, 500, 1000]:
        for _ in range(20):
            a = rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3, n)
            assert orc.np_sum(a) == np.sum(a)


def test_global_chroma(orc):
    rng = np.random.default_rng(1)
    for nb in (12, 24, 36):
        c = rng.random((777, nb))
        expect = np.divide(c.sum(axis=0), np.max(c.sum(axis=0)))  # Serra09.py:28
        assert np.array_equal(orc.global_chroma(c), expect)
    with pytest.raises(IOError):
        orc.global_chroma(rng.random((12, 100)))  # wrong axis, Serra09.py:26-27


def test_oti(orc, golden):
    g = golden("stages")
    got = [orc.get_oti(a, b) for a, b in zip(g["oti_G1"], g["oti_G2"])]
    assert np.array_equal(got, g["oti_expected"])
    # a profile rotated by k has OTI k with respect to itself
    assert list(g["oti_expected"][:12]) == list(range(12))


@pytest.mark.parametrize("c", [0, 1, 2])
def test_stage_chain(orc, golden, c):
    g = golden("stages")
    p = "c%d_" % c
    X, Y, m, kappa = g[p + "X"], g[p + "Y"], int(g[p + "m"]), float(g[p + "kappa"])
    oti = orc.get_oti(g[p + "gX"], g[p + "gY"])
    assert oti == int(g[p + "oti"])
    csm = orc.get_csm(X, Y, shift=oti)
    assert np.max(np.abs(csm - g[p + "CSM"])) <= 1e-9   # BLAS order differs; contract is 1e-5
    # sliding on the reference's own CSM is bit-exact (cumsum order is sequential)
    assert np.array_equal(orc.sliding_csm(g[p + "CSM"], m), g[p + "S"])
    S = orc.sliding_csm(csm, m)
    assert np.max(np.abs(S - g[p + "S"])) <= 1e-9
    assert np.array_equal(orc.csm_to_binary(S, kappa), g[p + "B1"])
    B = orc.csm_to_binary_mutual(S, kappa)
    assert np.array_equal(B, g[p + "B"])
    M, N = B.shape
    D = np.zeros(M * N, dtype=np.float32)
    q = orc.qmax(B.flatten(), D, M, N)
    assert np.array_equal(D.reshape(M, N), g[p + "Dq"])
    d = orc.dmax(B.flatten(), D, M, N)                   # reused D, Serra09.py:173-175
    assert np.array_equal(D.reshape(M, N), g[p + "Dd_reused"])
    Df = np.zeros(M * N, dtype=np.float32)
    df = orc.dmax(B.flatten(), Df, M, N)
    assert np.array_equal(Df.reshape(M, N), g[p + "Dd_fresh"])
    sc = g[p + "scores"]
    assert q / (M + N) == sc[0] and d / (M + N) == sc[1] and df / (M + N) == sc[2]
    for key, mask, ref_score in (("Dsw_mutual", B, sc[3]), ("Dsw_onesided", g[p + "B1"], sc[4])):
        Dw = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        w = orc.swconstrained(np.ascontiguousarray(mask.flatten()), Dw, M, N)
        # -0.7 is inexact and the reference is built -Ofast (reassociation): tolerance 1e-5
        assert np.max(np.abs(Dw.reshape(M + 1, N + 1) - g[p + key])) <= 1e-5
        assert abs(w - ref_score) <= 1e-5


def test_float32_inputs(orc, golden):
    g = golden("stages")
    csm = orc.get_csm(g["f32_X"], g["f32_Y"])
    assert csm.dtype == np.float32
    assert np.max(np.abs(csm - g["f32_CSM"])) <= 2e-6
    assert np.array_equal(orc.sliding_csm(g["f32_CSM"], 9), g["f32_S"])
    assert np.array_equal(orc.csm_to_binary_mutual(g["f32_S"], 0.095), g["f32_B"])


def test_kappa_conventions(orc, golden):
    g = golden("stages")
    D = g["kap_D"]
    assert orc.lib().orc_nneighbs(0.25, 50) == 12 and orc.lib().orc_nneighbs(0.11, 50) == 6
    assert np.array_equal(orc.csm_to_binary(D, 0.25), g["kap_B_frac"])
    assert np.array_equal(orc.csm_to_binary(D, 0.11), g["kap_B_frac2"])
    assert np.array_equal(orc.csm_to_binary(D, 7), g["kap_B_int"])
    assert np.array_equal(orc.csm_to_binary_mutual(D, 7), g["kap_Bm_int"])
    assert np.array_equal(orc.csm_to_binary_mutual(D, 0.25), g["kap_Bm_frac"])
    assert np.all(orc.csm_to_binary(D, 0) == 1)          # CRPUtils.py:188-189


def test_alignment_cases(orc, golden):
    g = golden("dp_cases")
    for k in range(int(g["n_cases"])):
        p = "k%d_" % k
        S = g[p + "S"]
        M, N = S.shape
        Sf = np.ascontiguousarray(S.flatten())
        sc = g[p + "scores"]
        D = np.zeros(M * N, dtype=np.float32)
        assert orc.qmax(Sf, D, M, N) == sc[0]
        assert np.array_equal(D.reshape(M, N), g[p + "Dq"])
        assert orc.dmax(Sf, D, M, N) == sc[1]
        assert np.array_equal(D.reshape(M, N), g[p + "Dd_reused"])
        Df = np.zeros(M * N, dtype=np.float32)
        assert orc.dmax(Sf, Df, M, N) == sc[2]
        assert np.array_equal(Df.reshape(M, N), g[p + "Dd_fresh"])
        Dw = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        assert abs(orc.swconstrained(Sf, Dw, M, N) - sc[3]) <= 1e-5
        assert np.max(np.abs(Dw.reshape(M + 1, N + 1) - g[p + "Dsw"])) <= 1e-5
    for tag in ("ones", "zeros", "eye"):
        S = g["ka_" + tag + "_S"]
        M, N = S.shape
        Sf = np.ascontiguousarray(S.flatten())
        sc = g["ka_" + tag + "_scores"]
        assert orc.qmax(Sf, np.zeros(M * N, np.float32), M, N) == sc[0]
        assert orc.dmax(Sf, np.zeros(M * N, np.float32), M, N) == sc[1]
        assert abs(orc.swconstrained(Sf, np.zeros((M + 1) * (N + 1), np.float32), M, N) - sc[2]) <= 1e-5


def test_alignment_values_are_half_integers(golden):
    g = golden("dp_cases")
    for k in range(int(g["n_cases"])):
        for key in ("Dq", "Dd_reused", "Dd_fresh"):
            D = g["k%d_%s" % (k, key)]
            assert np.array_equal(D * 2, np.round(D * 2))


def test_serra09_mini_chain(orc, golden):
    g = golden("serra09_mini")
    feats, off, gc, pairs = g["feats"], g["frame_off"], g["gchroma"], g["pairs"]
    q, d, _ = orc.serra09_pairs(feats, off, gc, pairs, m=9, kappa=0.095, do_oti=True, nthreads=2)
    assert np.array_equal(q, g["chroma_qmax"])
    assert np.array_equal(d, g["chroma_dmax"])


def test_serra09_swc_chain(orc, golden):
    """BASELINE config 3 at chain level: the oracle's chain (stage by stage) + its swalignimpconstrained restatement against the
    reference's compiled one on the reference's masks (serra09_swc.npz: CRPUtils chain + SequenceAlignment.c -Ofast, called
    as EarlySNF_Old.py:198-203 does).  The -0.7 penalty is inexact in float32 and -Ofast reassociates: 1e-5."""
    import zlib
    g, mini = golden("serra09_swc"), golden("serra09_mini")
    assert zlib.crc32(np.ascontiguousarray(mini["feats"]).tobytes()) == int(g["corpus_crc"][0])
    assert np.array_equal(g["pairs"], mini["pairs"])
    feats, off, gc = mini["feats"], mini["frame_off"], mini["gchroma"]
    for t, (i, j) in enumerate(g["pairs"][:24]):
        X, Y = feats[off[i]:off[i + 1]], feats[off[j]:off[j + 1]]
        S = orc.sliding_csm(orc.get_csm(X, Y, orc.get_oti(gc[i], gc[j])), 9)
        for key, B in (("chroma_swc_raw", orc.csm_to_binary_mutual(S, 0.095)), ("onesided_swc_raw", orc.csm_to_binary(S, 0.095))):
            M, N = B.shape
            assert (M, N) == tuple(g["shapes"][t])
            D = np.zeros((M + 1) * (N + 1), dtype=np.float32)
            got = orc.swconstrained(np.ascontiguousarray(B.flatten()), D, M, N)
            assert abs(got - g[key][t]) <= 1e-5, (t, key, got, g[key][t])
    assert np.allclose(g["chroma_swc"], g["chroma_swc_raw"] / g["shapes"].sum(axis=1), rtol=0, atol=0)


def test_pairs_1000(orc, golden):
    g = golden("pairs_1000")
    feats, off, gc, pairs = g["feats"], g["frame_off"], g["gchroma"], g["pairs"]
    q, d, _ = orc.serra09_pairs(feats, off, gc, pairs, nthreads=3)
    assert np.array_equal(q, g["chroma_qmax"])
    assert np.array_equal(d, g["chroma_dmax"])
    i, j = pairs[0]
    X, Y = feats[off[i]:off[i + 1]], feats[off[j]:off[j + 1]]
    assert orc.get_oti(gc[i], gc[j]) == int(g["oti"][0])
    S = orc.sliding_csm(orc.get_csm(X, Y, shift=int(g["oti"][0])), 9)
    B = orc.csm_to_binary_mutual(S, 0.095)
    assert np.array_equal(np.packbits(B, axis=1), g["B_packed_0"])
    assert np.max(np.abs(np.diag(S) - g["S_diag_0"])) <= 1e-9


def test_evalstats(golden):
    from oracle.evalstats import get_eval_statistics
    g = golden("evalstats")
    for c in range(int(g["n_cases"])):
        labels = g["e%d_labels" % c]
        cliques = {}
        for i, lab in enumerate(labels):
            cliques.setdefault("clique_%d" % lab, set()).add(i)
        MR, MRR, MDR, MAP, tops = get_eval_statistics(g["e%d_D" % c], cliques)
        assert np.array_equal(np.array([MR, MRR, MDR, MAP] + list(tops)), g["e%d_stats" % c])


def test_ftm2d_oracle_against_reference_functions(golden):
    """oracle/ftm2d.py vs the reference's chrompwr / btchroma_to_fftmat / shingle / similarity outputs."""
    from oracle import ftm2d
    g = golden("ftm2d")
    n = int(g["n_songs"])
    sh = []
    for s in range(n):
        X = g["bt%d" % s]
        if "chrompwr%d" % s in g.files:
            assert np.array_equal(ftm2d.chrompwr(X, 1.96), g["chrompwr%d" % s])
            # numpy's pocketfft vs scipy.fftpack: same transform, last-bit differences
            np.testing.assert_allclose(ftm2d.btchroma_to_fftmat(g["chrompwr%d" % s], 75), g["fftmat%d" % s], rtol=0, atol=1e-12)
        sh.append(ftm2d.shingle_from_btchroma(X))
        np.testing.assert_allclose(sh[-1], g["shingle%d" % s], rtol=0, atol=1e-13)
    sims = np.array([ftm2d.similarity(sh[i], sh[j]) for i, j in g["pairs"]])
    np.testing.assert_allclose(sims, g["sims"], rtol=0, atol=1e-13)
    assert np.array_equal(ftm2d.shingle_from_btchroma(np.ones((12, 74))), np.zeros(900))     # too few beats (:89)


def test_snf_oracle_against_reference(golden):
    """oracle/snf.py vs SimilarityFusion.py's own outputs (affinities, P, S, cross-diffusion) and the EarlySNF chain."""
    from oracle import snf
    g = golden("snf")
    tol = dict(rtol=0, atol=1e-12)
    np.testing.assert_allclose(snf.get_W(g["u_ssma"], 4), g["u_W_ssma"], **tol)
    np.testing.assert_allclose(snf.get_WCSM(g["u_csm"], 4, 5), g["u_WCSM"], **tol)
    W = snf.get_WCSMSSM(g["u_ssma"], g["u_ssmb"], g["u_csm"], 9)
    np.testing.assert_allclose(W, g["u_W"], **tol)
    np.testing.assert_allclose(snf.get_P(g["u_W"], True), g["u_P"], **tol)
    np.testing.assert_allclose(snf.get_P(g["u_W"], False), g["u_P_noreg"], **tol)
    np.testing.assert_allclose(snf.get_S(g["u_W"], 9), g["u_S"], **tol)
    for it in (1, 2, 3):
        np.testing.assert_allclose(snf.snf_ws([g["u_W"], g["u_W2"]], K=9, niters=it), g["u_fused_it%d" % it], **tol)
    # the chain, pair 0 with intermediates, then all scores
    off = g["c_frame_off"]
    songs = [{'gchroma': g["c_gchroma"][s], 'chroma': np.ascontiguousarray(g["c_feats"][off[s]:off[s + 1]].T),
              'ssms': g["c_ssms%d" % s]} for s in range(len(off) - 1)]
    keep = {}
    q, d = [], []
    for t, (i, j) in enumerate(g["c_pairs"]):
        a, b = snf.early_snf_pair(songs[i], songs[j], keep=keep if t == 0 else None)
        q.append(a); d.append(b)
    np.testing.assert_allclose(keep["csm"], g["c0_csm"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(keep["W0"], g["c0_W0"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(keep["W1"], g["c0_W1"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(keep["fused"], g["c0_fused"], rtol=0, atol=1e-11)
    assert np.array_equal(keep["B"], g["c0_B"])
    assert np.array_equal(np.array(q), g["c_snf_qmax"]) and np.array_equal(np.array(d), g["c_snf_dmax"])


def test_config2_slice_scores_and_map(orc, golden):
    """BASELINE config 2 (1000-frame songs): a sample of the 2 016 reference scores of the 64-song slice through the
    oracle, and the oracle's evaluation statistics on the reference's full slice score matrix (the -m gpu test runs
    all 2 016 pairs through the plugin)."""
    import zlib
    from acoss_amd import synth
    from oracle.evalstats import get_eval_statistics
    g = golden("config2_slice64")
    n = int(g["n_songs"])
    corpus = synth.make_corpus(n // 4, 4, n_frames=1000, seed=20260)
    assert zlib.crc32(corpus.feats.tobytes()) == int(g["corpus_crc"][0])
    pairs = synth.all_pairs(n)
    pick = np.r_[0:6, 40:44, np.arange(100, len(pairs), 97)]            # in-clique and cross-clique pairs
    q, d, _ = orc.serra09_pairs(corpus.feats, corpus.frame_off, corpus.gchroma, pairs[pick], nthreads=8)
    assert np.array_equal(q, g["chroma_qmax"][pick]) and np.array_equal(d, g["chroma_dmax"][pick])
    for key in ("qmax", "dmax"):
        D = np.zeros((n, n), dtype=np.float32)
        D[pairs[:, 0], pairs[:, 1]] = g["chroma_" + key]
        D += D.T
        MR, MRR, MDR, MAP, tops = get_eval_statistics(D, corpus.cliques())
        assert np.array_equal(np.array([MR, MRR, MDR, MAP] + list(tops)), g["stats_" + key])


def test_config2_hard_slice_scores_and_map(orc, golden):
    """The oracle on the hard slice (synth.config2_hard(): MAP 0.6-0.9 in the reference): a sample of the 2 016 scores of both
    keys equal the reference's, and the reference's statistics follow from its scores through the loop-form evaluation."""
    import zlib
    from acoss_amd import synth
    from oracle import evalstats
    g = golden("config2_hard64")
    corpus = synth.config2_hard()
    assert zlib.crc32(corpus.feats.tobytes()) == int(g["corpus_crc"][0])
    pairs = synth.all_pairs(corpus.n_songs)
    sel = np.random.default_rng(3).permutation(len(pairs))[:96]
    q, d, _ = orc.serra09_pairs(corpus.feats, corpus.frame_off, corpus.gchroma, pairs[sel], nthreads=8)
    assert np.array_equal(q, g["chroma_qmax"][sel]) and np.array_equal(d, g["chroma_dmax"][sel])
    n = corpus.n_songs
    for key, want in (("chroma_qmax", g["stats_qmax"]), ("chroma_dmax", g["stats_dmax"])):
        D = np.zeros((n, n), dtype=np.float32)
        D[pairs[:, 0], pairs[:, 1]] = g[key]
        D += D.T
        MR, MRR, MDR, MAP, tops = evalstats.get_eval_statistics(D, corpus.cliques())
        assert np.array_equal(np.array([MR, MRR, MDR, MAP] + list(tops)), want)
    assert 0.6 <= float(g["stats_qmax"][3]) <= 0.9


def test_reference_similarity_fixture(orc, golden):
    """The six score vectors the reference's own Serra09.similarity returned (make_golden_config2.py --similarity)
    against the oracle: chroma and MFCC chains exactly; the float32 'ssms' chain (get_csm in float32, no window,
    Serra09.py:186-192) on the pairs whose masks do not depend on the float32 summation order."""
    g = golden("serra09_similarity_ref")
    feats, off, gc, pairs = g["feats"], g["frame_off"], g["gchroma"], g["pairs"]
    q, d, _ = orc.serra09_pairs(feats, off, gc, pairs, nthreads=4)
    assert np.array_equal(q, g["chroma_qmax"]) and np.array_equal(d, g["chroma_dmax"])
    mf, so, ss = g["mfcc"], g["ssms_off"], g["ssms"]
    for t, (i, j) in enumerate(pairs):
        a, b = mf[off[i]:off[i + 1]], mf[off[j]:off[j + 1]]
        S = orc.sliding_csm(orc.get_csm(a, b), 9)                       # float32 CSM, float64 window sums (:178-179)
        B = orc.csm_to_binary_mutual(S, 0.095)
        M, N = B.shape
        D = np.zeros(M * N, dtype=np.float32)
        assert orc.qmax(np.ascontiguousarray(B.flatten()), D, M, N) / (M + N) == g["mfcc_qmax"][t]
        assert orc.dmax(np.ascontiguousarray(B.flatten()), D, M, N) / (M + N) == g["mfcc_dmax"][t]
        if not g["ssms_robust"][t]:
            continue
        B = orc.csm_to_binary_mutual(orc.get_csm(ss[so[i]:so[i + 1]], ss[so[j]:so[j + 1]]), 0.095)
        M, N = B.shape
        D = np.zeros(M * N, dtype=np.float32)
        assert orc.qmax(np.ascontiguousarray(B.flatten()), D, M, N) / (M + N) == g["ssms_scatter_qmax"][t]
        assert orc.dmax(np.ascontiguousarray(B.flatten()), D, M, N) / (M + N) == g["ssms_scatter_dmax"][t]
