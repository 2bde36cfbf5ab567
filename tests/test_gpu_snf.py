"""Similarity network fusion / EarlySNF on the GPU against the reference's own outputs
(tests/golden/snf.npz, made by tests/golden/make_golden_snf.py from SimilarityFusion.py / CRPUtils.py / the compiled
SequenceAlignment.c) and against the oracle restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from acoss_amd import engine
    engine.require_gpu()
    return engine


def _layout(eng, lens, pairs):
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return eng.PairBatch(off, np.array(pairs, dtype=np.int32), 1, "cuda:0")


def _place(batch, mats):
    import torch
    buf = np.zeros(max(batch.total_crp, 1))
    for p, A in enumerate(mats):
        d = batch.descs[p]
        view = buf[int(d["crp_off"]):int(d["crp_off"]) + A.shape[0] * int(d["crp_pitch"])].reshape(A.shape[0], -1)
        view[:, :A.shape[1]] = A
    return torch.from_numpy(buf).cuda()


def test_affinity_and_fusion_match_reference(eng, golden):
    """get_WCSMSSM and snf_ws (1, 2, 3 iterations) from given distance matrices."""
    g = golden("snf")
    M, N = g["u_ssma"].shape[0], g["u_ssmb"].shape[0]
    la, lb, lc = _layout(eng, [M, N], [(0, 0)]), _layout(eng, [M, N], [(1, 1)]), _layout(eng, [M, N], [(0, 1)])
    feats = []
    for pre in ("u", "u2"):
        feats.append(dict(ssma=_place(la, [g[pre + "_ssma"]]), ssmb=_place(lb, [g[pre + "_ssmb"]]), csm=_place(lc, [g[pre + "_csm"]]),
                          da=la, db=lb, dc=lc, win=1))
    L = M + N
    # K = int(kappa * L) must be 9 as in the fixture
    kappa = 9.5 / L
    for it in (1, 2, 3):
        cross, W, fused = eng.snf_cross(feats, [M], [N], kappa, lc, niters=it, debug=True)
        W = W.cpu().numpy().reshape(2, L, L)
        np.testing.assert_allclose(W[0], g["u_W"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(W[1], g["u_W2"], rtol=0, atol=1e-12)
        fused = fused.cpu().numpy().reshape(L, L)
        np.testing.assert_allclose(fused, g["u_fused_it%d" % it], rtol=0, atol=1e-12)
        d = lc.descs[0]
        got = cross.cpu().numpy()[int(d["crp_off"]):int(d["crp_off"]) + M * int(d["crp_pitch"])].reshape(M, -1)[:, :N]
        np.testing.assert_allclose(got, -g["u_fused_it%d" % it][0:M, M:], rtol=0, atol=1e-12)


def test_early_snf_chain_matches_reference(eng, golden):
    """EarlySNF.py:41-90 end to end on the fixture's ragged pairs: snf_qmax / snf_dmax equal the reference's."""
    g = golden("snf")
    off = g["c_frame_off"]
    n = len(off) - 1
    chroma = eng.DeviceCorpus(g["c_feats"], off, gchroma=g["c_gchroma"])
    ss = [g["c_ssms%d" % s] for s in range(n)]
    soff = np.concatenate([[0], np.cumsum([x.shape[0] for x in ss])]).astype(np.int64)
    ssms = eng.DeviceCorpus(np.concatenate(ss, axis=0), soff)
    res = eng.early_snf_scores(chroma, ssms, g["c_pairs"])
    assert np.array_equal(res["qmax"], g["c_snf_qmax"]) and np.array_equal(res["dmax"], g["c_snf_dmax"])


def test_plugin_contract_against_oracle(eng, tmp_path, monkeypatch):
    """EarlySNF class on feature files with chroma + mfcc + 'ssms': all eight score vectors against the oracle chain."""
    from oracle import oracle as orc, snf as osnf
    from acoss_amd.EarlySNF import EarlySNF
    from acoss_amd import synth
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(8)
    lens = iter([64, 81, 100, 73, 90])
    ch = synth.make_corpus(5, 1, seed=5, lengths=lambda r: next(lens))
    d = tmp_path / "feats"
    d.mkdir()
    songs = []
    for i in range(ch.n_songs):
        X = ch.song(i)                                       # (n, 12) at the aggregated rate: downsample_fac = 1
        n = X.shape[0]
        mf = np.cumsum(rng.standard_normal((13, n)), axis=1)
        ss = np.cumsum(rng.standard_normal((n - 8, 30)), axis=0) * 0.1
        np.savez(d / ("s%02d.npz" % i), crema=X, mfcc_htk=mf, ssms=ss, label="clique_%d" % (i // 2), track_id="t%d" % i)
        songs.append({'gchroma': orc.global_chroma(X), 'chroma': np.ascontiguousarray(X.T), 'mfcc': mf, 'ssms': ss})
    alg = EarlySNF(datapath=str(d), chroma_type="crema", shortname="t", downsample_fac=1, cachedir=str(tmp_path / "cache"))
    idxs = np.array([[0, 1], [1, 2], [2, 4], [3, 0], [4, 3]])
    res = alg.similarity(idxs)
    assert sorted(res) == sorted(EarlySNF.KEYS)
    for t, (i, j) in enumerate(idxs):
        q, dm = osnf.early_snf_pair(songs[i], songs[j])
        assert res["snf_qmax"][t] == q and res["snf_dmax"][t] == dm, (t, res["snf_qmax"][t], q)
        q, dm = orc.serra09_pair(songs[i]['chroma'].T, songs[i]['gchroma'], songs[j]['chroma'].T, songs[j]['gchroma'])
        assert res["chroma_qmax"][t] == q and res["chroma_dmax"][t] == dm
        q, dm = orc.serra09_pair(songs[i]['mfcc'].T, np.zeros(13), songs[j]['mfcc'].T, np.zeros(13), do_oti=False)
        assert res["mfcc_qmax"][t] == q and res["mfcc_dmax"][t] == dm
        q, dm = orc.serra09_pair(songs[i]['ssms'], np.zeros(30), songs[j]['ssms'], np.zeros(30), m=1, do_oti=False)
        assert res["ssms_scatter_qmax"][t] == q and res["ssms_scatter_dmax"][t] == dm
    assert np.allclose(alg.Ds["snf_qmax"][idxs[:, 0], idxs[:, 1]], res["snf_qmax"])
