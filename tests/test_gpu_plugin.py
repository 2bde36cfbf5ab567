"""The plugin surface on the GPU: Serra09.similarity / all_pairwise / getEvalStatistics against the
reference's own scores for the covers80-shaped corpus (BASELINE config 0/1) and against the oracle
for file-based features."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config1_all_pairwise_scores_and_map_equal_reference(golden, tmp_path, monkeypatch):
    from acoss_amd import synth
    from acoss_amd.Serra09 import Serra09
    monkeypatch.chdir(tmp_path)
    g = golden("config1_scores")
    corpus = synth.config1()
    assert zlib.crc32(corpus.feats.tobytes()) == int(g["corpus_crc"][0])     # same inputs as the reference saw
    alg = Serra09(corpus, shortname="config1", do_memmaps=True, cachedir=str(tmp_path / "cache"))
    alg.all_pairwise(symmetric=True)
    pairs = synth.all_pairs(corpus.n_songs)
    for key in ("chroma_qmax", "chroma_dmax"):
        got = np.asarray(alg.Ds[key])[pairs[:, 0], pairs[:, 1]]
        assert np.array_equal(got, g[key].astype(np.float32)), key              # every one of 12720 scores
        assert np.array_equal(np.asarray(alg.Ds[key]), np.asarray(alg.Ds[key]).T)   # CoverAlgorithm.py:180-182
    MR, MRR, MDR, MAP, tops = alg.getEvalStatistics("chroma_qmax", verbose=False)
    assert np.array_equal(np.array([MR, MRR, MDR, MAP] + list(tops)), g["stats_qmax"])
    MR, MRR, MDR, MAP, tops = alg.getEvalStatistics("chroma_dmax", verbose=False)
    assert np.array_equal(np.array([MR, MRR, MDR, MAP] + list(tops)), g["stats_dmax"])
    alg.cleanup_memmap()


def test_similarity_contract(golden, tmp_path):
    """Six keys, float64 arrays of length K, Ds written in place when do_memmaps (Serra09.py:158-196)."""
    from acoss_amd import synth
    from acoss_amd.Serra09 import Serra09
    g = golden("serra09_mini")
    corpus = synth.Corpus(g["feats"], g["frame_off"], g["gchroma"], [str(x) for x in g["labels"]])
    off = g["frame_off"]
    corpus.mfcc = [np.ascontiguousarray(g["mfcc"][off[i]:off[i + 1]].T) for i in range(corpus.n_songs)]
    alg = Serra09(corpus, shortname="mini", do_memmaps=True, cachedir=str(tmp_path / "cache"))
    idxs = g["pairs"][:20].astype(np.int64)
    with pytest.warns(UserWarning):
        sims = alg.similarity(idxs)
    assert sorted(sims) == sorted(Serra09.KEYS)
    for key in sims:
        assert sims[key].dtype == np.float64 and sims[key].shape == (20,)
    assert np.array_equal(sims["chroma_qmax"], g["chroma_qmax"][:20])
    assert np.array_equal(sims["chroma_dmax"], g["chroma_dmax"][:20])
    assert np.array_equal(sims["mfcc_qmax"], g["mfcc_qmax"][:20])
    assert np.array_equal(sims["mfcc_dmax"], g["mfcc_dmax"][:20])
    assert np.all(sims["ssms_scatter_qmax"] == 0)
    assert np.array_equal(np.asarray(alg.Ds["chroma_qmax"])[idxs[:, 0], idxs[:, 1]],
                          g["chroma_qmax"][:20].astype(np.float32))
    # do_memmaps=False: no Ds attribute at all (CoverAlgorithm.py:48-51)
    alg2 = Serra09(corpus, shortname="mini2", do_memmaps=False)
    assert not hasattr(alg2, "Ds")
    assert np.array_equal(alg2.similarity(idxs[:3])["chroma_qmax"], g["chroma_qmax"][:3])


def test_feature_files_and_downsampling(orc, tmp_path):
    """Songs on disk (.npz with the reference's field names), full-resolution chroma aggregated by
    block medians, MFCC by block means, OTI from the full-resolution global chroma."""
    from acoss_amd import Serra09 as S9
    from acoss_amd import synth
    full = synth.make_corpus(3, 2, seed=31, lengths=lambda r: r.integers(900, 1400))
    rng = np.random.default_rng(32)
    data = tmp_path / "features"
    data.mkdir()
    for i in range(full.n_songs):
        x = full.song(i)
        np.savez(str(data / ("song_%02d.npz" % i)), hpcp=x.astype(np.float32), crema=x,
                 mfcc_htk=np.cumsum(rng.standard_normal((13, x.shape[0])), axis=1).astype(np.float32),
                 label=full.labels[i])
    alg = S9.Serra09(str(data), chroma_type="crema", shortname="files", downsample_fac=8, do_memmaps=False,
                     cachedir=str(tmp_path / "cache"))
    assert alg.N == 6
    idxs = np.array([[0, 1], [2, 3], [4, 1], [5, 5]])
    with pytest.warns(UserWarning):
        sims = alg.similarity(idxs)
    assert alg.cliques[full.labels[0]] == {0, 1}
    exp_q, exp_d = np.zeros(len(idxs)), np.zeros(len(idxs))
    for t, (i, j) in enumerate(idxs):
        fi, fj = alg.load_features(i), alg.load_features(j)
        assert np.array_equal(fi["gchroma"], S9.global_chroma(full.song(i)))
        exp_q[t], exp_d[t] = orc.serra09_pair(fi["chroma"].T, fi["gchroma"], fj["chroma"].T, fj["gchroma"], m=9, kappa=0.095)
    assert np.array_equal(sims["chroma_qmax"], exp_q), (sims["chroma_qmax"], exp_q)
    assert np.array_equal(sims["chroma_dmax"], exp_d), (sims["chroma_dmax"], exp_d)


def test_config3_shape_smith_waterman_chain(orc):
    """BASELINE config 3 (DA-TACOS benchmark_subset shape: lengths ~N(520,120) in [200,1200], Smith-Waterman
    constrained on the mutual mask): engine chain vs the oracle composed stage by stage.  qmax / dmax exact, the
    Smith-Waterman score within 1e-5 (its -0.7 penalty is inexact in float32)."""
    from acoss_amd import engine, synth
    engine.require_gpu()
    lens_it = iter([200, 1200, 520, 640, 1100, 333, 415, 777, 560, 999, 250, 480])       # the clip limits and in between
    ch = synth.make_corpus(4, 3, seed=15000, lengths=lambda r: next(lens_it))
    rng = np.random.default_rng(0)
    allp = synth.all_pairs(ch.n_songs)
    pairs = allp[rng.permutation(len(allp))[:40]]
    corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    got = engine.serra09_scores(corpus, pairs, want=("qmax", "dmax", "swc"))
    lens = np.diff(ch.frame_off)
    assert lens.min() == 200 and lens.max() == 1200
    for t, (i, j) in enumerate(pairs):
        X, Y = ch.song(i), ch.song(j)
        oti = orc.get_oti(ch.gchroma[i], ch.gchroma[j])
        S = orc.sliding_csm(orc.get_csm(X, Y, oti), 9)
        B = orc.csm_to_binary_mutual(S, 0.095)
        M, N = B.shape
        Bf = np.ascontiguousarray(B.flatten())
        D = np.zeros(M * N, dtype=np.float32)
        q = orc.qmax(Bf, D, M, N) / (M + N)
        d = orc.dmax(Bf, D, M, N) / (M + N)                        # D not re-zeroed (Serra09.py:173-175)
        sw = orc.swconstrained(Bf, np.zeros((M + 1) * (N + 1), dtype=np.float32), M, N) / (M + N)
        assert got["qmax"][t] == q and got["dmax"][t] == d, (t, i, j)
        assert abs(got["swc"][t] - sw) <= 1e-5, (t, i, j, got["swc"][t], sw)


def test_mixed_size_batch_splits_by_size_class(orc):
    """A pair list mixing <= 1032-frame songs (bit-mask path) with longer ones (byte-mask path): same scores as the
    oracle, in the caller's order."""
    from acoss_amd import engine, synth
    engine.require_gpu()
    lens_it = iter([300, 1100, 420, 1050, 200])
    ch = synth.make_corpus(5, 1, seed=7, lengths=lambda r: next(lens_it))
    pairs = np.array([(0, 2), (1, 0), (2, 4), (3, 1), (4, 0), (2, 3)], dtype=np.int32)
    corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    got = engine.serra09_scores(corpus, pairs)
    for t, (i, j) in enumerate(pairs):
        q, d = orc.serra09_pair(ch.song(i), ch.gchroma[i], ch.song(j), ch.gchroma[j])
        assert got["qmax"][t] == q and got["dmax"][t] == d, (t, i, j)


def test_randomised_ragged_pairs_against_oracle(orc):
    """4000 random pairs (both orientations) of a 350-song corpus with lengths 60..1032: the product chain's qmax and dmax
    equal the oracle's exactly (tools/soak.py runs the same check on 30 000 pairs)."""
    from acoss_amd import engine, synth
    engine.require_gpu()
    rng = np.random.default_rng(3)
    ch = synth.make_corpus(40, 8, seed=3, singletons=30, lengths=lambda r: int(np.clip(r.normal(520, 160), 60, 1032)))
    corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    allp = synth.all_pairs(ch.n_songs)
    pairs = allp[rng.permutation(len(allp))[:4000]]
    pairs = np.where(rng.random((len(pairs), 1)) < 0.5, pairs, pairs[:, ::-1]).astype(np.int32)
    got = engine.serra09_scores(corpus, pairs)
    q, d, _ = orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, pairs, nthreads=min(os.cpu_count() or 1, 16))
    assert np.array_equal(got["qmax"], q) and np.array_equal(got["dmax"], d)


def test_songs_longer_than_2056_frames(orc):
    """No length limit in the reference (it scores whole tracks): matrices beyond 2048 x 2048 go through the any-size
    radix selection, the byte mask and the LDS-resident alignment kernel.  Scores equal the oracle's; swalignimpconstrained
    within 1e-5.  (Exact ties on this path: tests/test_gpu_stages.py::test_binarize_ties_and_negative_values.)"""
    from acoss_amd import engine, synth
    engine.require_gpu()
    lens_it = iter([2300, 400, 2070, 3000, 2057])
    ch = synth.make_corpus(5, 1, seed=99, lengths=lambda r: next(lens_it))
    corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = np.array([(0, 1), (1, 0), (2, 3), (3, 0), (4, 4), (1, 1), (4, 2)], dtype=np.int32)
    got = engine.serra09_scores(corpus, pairs, want=("qmax", "dmax", "swc"))
    for t, (i, j) in enumerate(pairs):
        X, Y = ch.song(i), ch.song(j)
        q, d = orc.serra09_pair(X, ch.gchroma[i], Y, ch.gchroma[j])
        assert got["qmax"][t] == q and got["dmax"][t] == d, (t, i, j, got["qmax"][t], q, got["dmax"][t], d)
        if t in (0, 6):
            B = orc.csm_to_binary_mutual(orc.sliding_csm(orc.get_csm(X, Y, orc.get_oti(ch.gchroma[i], ch.gchroma[j])), 9), 0.095)
            M, N = B.shape
            sw = orc.swconstrained(np.ascontiguousarray(B.flatten()), np.zeros((M + 1) * (N + 1), dtype=np.float32), M, N) / (M + N)
            assert abs(got["swc"][t] - sw) <= 1e-5


def test_config3_swc_through_the_plugin(golden, tmp_path):
    """BASELINE config 3 ("Serra09 Smith-Waterman constrained") through the plugin surface: Serra09(alignments=(..., "swc"))
    returns and stores `chroma_swc` (and `mfcc_swc`) beside the reference's six keys; values within 1e-5 of the reference's
    chain (tests/golden/serra09_swc.npz: CRPUtils chain + compiled SequenceAlignment.c as EarlySNF_Old.py:198-203 calls it),
    the other keys unchanged bit for bit; the default constructor keeps exactly the reference's six keys."""
    from acoss_amd import synth
    from acoss_amd.Serra09 import Serra09
    g, w = golden("serra09_mini"), golden("serra09_swc")
    corpus = synth.Corpus(g["feats"], g["frame_off"], g["gchroma"], [str(x) for x in g["labels"]])
    off = g["frame_off"]
    corpus.mfcc = [np.ascontiguousarray(g["mfcc"][off[i]:off[i + 1]].T) for i in range(corpus.n_songs)]
    alg = Serra09(corpus, shortname="mini_swc", do_memmaps=True, cachedir=str(tmp_path / "cache"), alignments=("qmax", "dmax", "swc"))
    assert alg.similarity_types == Serra09.KEYS + ["ssms_scatter_swc", "chroma_swc", "mfcc_swc"]
    idxs = g["pairs"].astype(np.int64)
    with pytest.warns(UserWarning):
        sims = alg.similarity(idxs)
    assert sorted(sims) == sorted(alg.similarity_types)
    assert np.array_equal(sims["chroma_qmax"], g["chroma_qmax"]) and np.array_equal(sims["chroma_dmax"], g["chroma_dmax"])
    assert np.array_equal(sims["mfcc_qmax"], g["mfcc_qmax"]) and np.array_equal(sims["mfcc_dmax"], g["mfcc_dmax"])
    assert sims["chroma_swc"].dtype == np.float64
    assert np.max(np.abs(sims["chroma_swc"] - w["chroma_swc"])) <= 1e-5
    assert np.max(sims["chroma_swc"]) > 0.05 and np.all(sims["mfcc_swc"] >= 0) and np.any(sims["mfcc_swc"] > 0)
    assert np.allclose(np.asarray(alg.Ds["chroma_swc"])[idxs[:, 0], idxs[:, 1]], w["chroma_swc"].astype(np.float32), rtol=0, atol=1e-5)
    assert sorted(Serra09(corpus, shortname="mini_d", do_memmaps=False).similarity_types) == sorted(Serra09.KEYS)
    with pytest.raises(ValueError):
        Serra09(corpus, shortname="bad", do_memmaps=False, alignments=("qmax", "swc"))


def test_all_pairwise_twice_gives_the_same_matrices(tmp_path, monkeypatch):
    """A second all_pairwise() on the same object starts its matrices from zero again: `Ds += Ds.T` (CoverAlgorithm.py:180-182)
    must not add the new upper triangle to the previous call's lower one (round 3's advisor finding)."""
    from acoss_amd import synth
    from acoss_amd.Serra09 import Serra09
    monkeypatch.chdir(tmp_path)
    corpus = synth.make_corpus(3, 2, seed=21, lengths=lambda r: r.integers(80, 160))
    alg = Serra09(corpus, shortname="twice", do_memmaps=False, cachedir=str(tmp_path / "cache"))
    alg.all_pairwise(symmetric=True)
    first = {k: np.array(v) for k, v in alg.Ds.items()}
    alg.all_pairwise(symmetric=True)
    for k in first:
        assert np.array_equal(np.asarray(alg.Ds[k]), first[k]), k
    assert np.max(first["chroma_qmax"]) > 0 and np.array_equal(first["chroma_qmax"], first["chroma_qmax"].T)
