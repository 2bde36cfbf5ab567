#!/usr/bin/env python
"""
Generates the golden fixtures in this directory by running THE REFERENCE ITSELF in the build
container (it cannot travel to the GPU box):

  * /root/reference/benchmarking/CRPUtils.py is imported as-is (numpy + scipy only);
  * /root/reference/benchmarking/SequenceAlignment.c is used through oracle/_ref/
    libseqalign_ref.so, compiled unmodified with the reference's flags by oracle/Makefile;
  * CoverAlgorithm.getEvalStatistics is called on the reference class; CoverAlgorithm.py's
    top-level `import deepdish` (absent here, never touched by that method) is satisfied by
    an empty module object registered under that name for the duration of this script;
  * Serra09.py is NOT imported (it instantiates kymatio at import); its per-pair chain
    (Serra09.py:166-184) is composed here from the reference's own functions in the same
    order, including the D-reuse between qmax and dmax (Serra09.py:173-175).

Usage:  python tests/golden/make_golden.py [--config1]
Outputs are data only (inputs + expected outputs); see README.md in this directory.
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/benchmarking"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import CRPUtils as ref  # noqa: E402  (the reference module)
from oracle import oracle as orc  # noqa: E402  (only for ref_lib(): the compiled reference C)
from acoss_amd import synth  # noqa: E402

R = orc.ref_lib("Ofast")
assert R is not None, "run `make -C oracle` first (needs /root/reference)"


def meta():
    return np.array([
        "numpy=" + np.__version__, "scipy=" + scipy.__version__,
        "reference=ctralie/acoss@/root/reference", "seqalign=-Ofast (benchmarking/setup.py:45)",
    ])


def ref_qmax(S, D, M, N):
    return float(R.qmax_c(orc._u(S), orc._f(D), int(M), int(N)))


def ref_dmax(S, D, M, N):
    return float(R.dmax_c(orc._u(S), orc._f(D), int(M), int(N)))


def ref_swc(S, D, N, M):
    return float(R.swalignimpconstrained(orc._u(S), orc._f(D), int(N), int(M)))


def global_chroma_ref(chroma):
    """Serra09.py:24-28 (expression restated; Serra09.py itself is not importable here)."""
    return np.divide(chroma.sum(axis=0), np.max(chroma.sum(axis=0)))


def chain(Xi, gi, Xj, gj, m, kappa, do_oti, dump=None):
    """Serra09.py:166-175 composed from the reference's functions.  Xi is (n,d) frames-major,
    i.e. Si['chroma'].T."""
    Ci, Cj = Xi.T, Xj.T                                  # (d, n) as in Serra09.py:154
    oti = int(ref.get_oti(gi, gj)) if do_oti else 0      # :166
    C1 = np.roll(Ci, oti, axis=0)                        # :167
    csm = ref.get_csm(C1.T, Cj.T)                        # :169
    S = ref.sliding_csm(csm, m)                          # :170
    B = ref.csm_to_binary_mutual(S, kappa)               # :171
    M, N = B.shape
    D = np.zeros(M * N, dtype=np.float32)                # :173
    q = ref_qmax(B.flatten(), D, M, N)                   # :174
    Dq = D.copy()
    d = ref_dmax(B.flatten(), D, M, N)                   # :175 (D reused)
    if dump is not None:
        dump.update(oti=oti, CSM=csm, S=S, B=B, Dq=Dq.reshape(M, N), Dd_reused=D.reshape(M, N))
    return q / (M + N), d / (M + N)


# -----------------------------------------------------------------------------------------
def gen_stages():
    out = {"meta": meta()}
    cases = [(40, 53, 5, 0.2), (97, 120, 9, 0.095), (200, 173, 9, 0.095)]
    corpus = synth.make_corpus(3, 2, seed=4242, lengths=lambda r: 256)
    for c, (ni, nj, m, kappa) in enumerate(cases):
        Xi = np.ascontiguousarray(corpus.song(2 * c)[:ni])
        Xj = np.ascontiguousarray(corpus.song(2 * c + 1)[:nj])
        gi, gj = global_chroma_ref(Xi), global_chroma_ref(Xj)
        dump = {}
        q, d = chain(Xi, gi, Xj, gj, m, kappa, True, dump)
        B1 = ref.csm_to_binary(dump["S"], kappa)
        M, N = dump["B"].shape
        Dfresh = np.zeros(M * N, dtype=np.float32)
        d_fresh = ref_dmax(dump["B"].flatten(), Dfresh, M, N)
        Dsw_m = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        sw_m = ref_swc(dump["B"].flatten(), Dsw_m, M, N)
        Dsw_1 = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        sw_1 = ref_swc(B1.flatten(), Dsw_1, M, N)        # EarlySNF_Old.py:200-201 convention
        p = "c%d_" % c
        out.update({
            p + "X": Xi, p + "Y": Xj, p + "gX": gi, p + "gY": gj, p + "m": m, p + "kappa": kappa,
            p + "oti": dump["oti"], p + "CSM": dump["CSM"], p + "S": dump["S"],
            p + "B1": B1, p + "B": dump["B"], p + "Dq": dump["Dq"],
            p + "Dd_reused": dump["Dd_reused"], p + "Dd_fresh": Dfresh.reshape(M, N),
            p + "Dsw_mutual": Dsw_m.reshape(M + 1, N + 1), p + "Dsw_onesided": Dsw_1.reshape(M + 1, N + 1),
            p + "scores": np.array([q, d, d_fresh / (M + N), sw_m, sw_1]),
        })
    out["n_cases"] = len(cases)
    # float32 inputs (essentia-HPCP-like): get_csm in f32, sliding promotes to f64
    Xi = corpus.song(0)[:90].astype(np.float32)
    Xj = corpus.song(3)[:75].astype(np.float32)
    csm32 = ref.get_csm(Xi, Xj)
    assert csm32.dtype == np.float32
    S32 = ref.sliding_csm(csm32, 9)
    out.update(f32_X=Xi, f32_Y=Xj, f32_CSM=csm32, f32_S=S32,
               f32_B=ref.csm_to_binary_mutual(S32, 0.095))
    # kappa conventions (CRPUtils.py:186-193): fraction with half-even rounding, integer count
    Dk = np.random.default_rng(7).random((30, 50))
    out.update(kap_D=Dk,
               kap_B_frac=ref.csm_to_binary(Dk, 0.25),          # round(12.5) -> 12
               kap_B_frac2=ref.csm_to_binary(Dk, 0.11),         # round(5.5)  -> 6
               kap_B_int=ref.csm_to_binary(Dk, 7),
               kap_Bm_int=ref.csm_to_binary_mutual(Dk, 7),
               kap_Bm_frac=ref.csm_to_binary_mutual(Dk, 0.25))
    # OTI known answers: every rotation of a profile against itself, plus random profiles
    rng = np.random.default_rng(11)
    G1 = rng.random((40, 12))
    G2 = rng.random((40, 12))
    G2[:12] = np.stack([np.roll(G1[k], k) for k in range(12)])
    out.update(oti_G1=G1, oti_G2=G2,
               oti_expected=np.array([ref.get_oti(a, b) for a, b in zip(G1, G2)]))
    np.savez_compressed(os.path.join(HERE, "stages.npz"), **out)
    print("stages.npz")


def gen_dp():
    """Alignment recurrences on random masks: tiny sizes, every density, exotic byte values."""
    rng = np.random.default_rng(99)
    out = {"meta": meta()}
    shapes = [(1, 1), (2, 2), (2, 9), (3, 3), (3, 4), (4, 4), (4, 3), (5, 17), (17, 5), (33, 64),
              (64, 65), (70, 129), (128, 31), (150, 150)]
    n = 0
    for (M, N) in shapes:
        for dens, exotic in [(0.07, False), (0.5, False), (0.95, False), (0.3, True)]:
            if exotic:
                S = rng.choice(np.array([0, 1, 2, 3, 255], dtype=np.uint8), size=(M, N),
                               p=[0.55, 0.3, 0.05, 0.05, 0.05])
            else:
                S = (rng.random((M, N)) < dens).astype(np.uint8)
            Sf = np.ascontiguousarray(S.flatten())
            Dq = np.zeros(M * N, dtype=np.float32)
            q = ref_qmax(Sf, Dq, M, N)
            Dr = Dq.copy()
            dr = ref_dmax(Sf, Dr, M, N)                    # Serra09-style reuse
            Df = np.zeros(M * N, dtype=np.float32)
            df = ref_dmax(Sf, Df, M, N)
            Dw = np.zeros((M + 1) * (N + 1), dtype=np.float32)
            w = ref_swc(Sf, Dw, M, N)
            p = "k%d_" % n
            out.update({p + "S": S, p + "Dq": Dq.reshape(M, N), p + "Dd_reused": Dr.reshape(M, N),
                        p + "Dd_fresh": Df.reshape(M, N), p + "Dsw": Dw.reshape(M + 1, N + 1),
                        p + "scores": np.array([q, dr, df, w])})
            n += 1
    # all-ones / all-zeros known answers
    for tag, S in (("ones", np.ones((20, 31), np.uint8)), ("zeros", np.zeros((20, 31), np.uint8)),
                   ("eye", np.eye(40, dtype=np.uint8))):
        M, N = S.shape
        Sf = np.ascontiguousarray(S.flatten())
        D = np.zeros(M * N, dtype=np.float32)
        q = ref_qmax(Sf, D, M, N)
        Df = np.zeros(M * N, dtype=np.float32)
        d = ref_dmax(Sf, Df, M, N)
        Dw = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        w = ref_swc(Sf, Dw, M, N)
        out["ka_" + tag + "_S"] = S
        out["ka_" + tag + "_scores"] = np.array([q, d, w])
    out["n_cases"] = n
    np.savez_compressed(os.path.join(HERE, "dp_cases.npz"), **out)
    print("dp_cases.npz", n)


def gen_serra09_mini():
    """12-song ragged mini-corpus through the Serra09 chain: chroma (OTI, f64) and an
    MFCC-shaped 13-d float32 feature without OTI (Serra09.py:178-184)."""
    corpus = synth.make_corpus(4, 3, seed=1212, lengths=lambda r: r.integers(60, 141))
    rng = np.random.default_rng(1213)
    n = corpus.n_songs
    mfcc = [np.cumsum(rng.standard_normal((corpus.song(i).shape[0], 13)), axis=0).astype(np.float32)
            for i in range(n)]
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)] + [(3, 3), (7, 2), (11, 0)]
    pairs = np.array(pairs, dtype=np.int32)
    res = {k: np.zeros(len(pairs)) for k in ("chroma_qmax", "chroma_dmax", "mfcc_qmax", "mfcc_dmax")}
    for t, (i, j) in enumerate(pairs):
        q, d = chain(corpus.song(i), corpus.gchroma[i], corpus.song(j), corpus.gchroma[j],
                     9, 0.095, True)
        res["chroma_qmax"][t], res["chroma_dmax"][t] = q, d
        q, d = chain(mfcc[i], None, mfcc[j], None, 9, 0.095, False)
        res["mfcc_qmax"][t], res["mfcc_dmax"][t] = q, d
    np.savez_compressed(os.path.join(HERE, "serra09_mini.npz"), meta=meta(), feats=corpus.feats,
                        frame_off=corpus.frame_off, gchroma=corpus.gchroma,
                        labels=np.array(corpus.labels), mfcc=np.concatenate(mfcc, axis=0),
                        pairs=pairs, **res)
    print("serra09_mini.npz")


def gen_serra09_swc():
    """BASELINE config 3, "Serra09 Smith-Waterman constrained", at chain level: the Serra09 chain of the 12-song mini-corpus
    (Serra09.py:166-171: oti, roll, get_csm, sliding_csm, csm_to_binary_mutual -- the reference's CRPUtils functions) with the
    reference's compiled swalignimpconstrained as the alignment, called the way its one caller does (EarlySNF_Old.py:198-203:
    D = zeros((M+1)*(N+1)), alignment_fn(mask.flatten(), D, M, N), rows first).  Raw scores and the mask shapes; the plugin's
    `chroma_swc` key is raw / (M + N), the normalisation Serra09 applies to its other alignments (Serra09.py:174-175).
    Same corpus and pair list as serra09_mini.npz (that fixture holds the features)."""
    corpus = synth.make_corpus(4, 3, seed=1212, lengths=lambda r: r.integers(60, 141))
    n = corpus.n_songs
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)] + [(3, 3), (7, 2), (11, 0)]
    pairs = np.array(pairs, dtype=np.int32)
    raw, one_sided, shapes = np.zeros(len(pairs)), np.zeros(len(pairs)), np.zeros((len(pairs), 2), dtype=np.int64)
    for t, (i, j) in enumerate(pairs):
        dump = {}
        chain(corpus.song(i), corpus.gchroma[i], corpus.song(j), corpus.gchroma[j], 9, 0.095, True, dump)
        B = dump["B"]
        M, N = B.shape
        D = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        raw[t] = ref_swc(np.ascontiguousarray(B.flatten()), D, M, N)
        B1 = ref.csm_to_binary(dump["S"], 0.095)                   # the one-sided mask EarlySNF_Old.py:201 aligns
        D = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        one_sided[t] = ref_swc(np.ascontiguousarray(B1.flatten()), D, M, N)
        shapes[t] = (M, N)
    import zlib
    np.savez_compressed(os.path.join(HERE, "serra09_swc.npz"), meta=meta(), pairs=pairs, shapes=shapes,
                        corpus_crc=np.array([zlib.crc32(corpus.feats.tobytes())]),
                        chroma_swc_raw=raw, chroma_swc=raw / shapes.sum(axis=1), onesided_swc_raw=one_sided)
    print("serra09_swc.npz", raw[:5], one_sided[:5])


def gen_pairs_1000():
    """Three 1000-frame pairs of the BASELINE config-2 generator (cover / cover / non-cover)."""
    corpus = synth.make_corpus(3, 2, n_frames=1000, seed=20260)
    pairs = np.array([[0, 1], [2, 3], [1, 4]], dtype=np.int32)
    out = {"meta": meta(), "feats": corpus.feats, "frame_off": corpus.frame_off,
           "gchroma": corpus.gchroma, "pairs": pairs}
    qs, ds, otis = [], [], []
    for t, (i, j) in enumerate(pairs):
        dump = {}
        q, d = chain(corpus.song(i), corpus.gchroma[i], corpus.song(j), corpus.gchroma[j],
                     9, 0.095, True, dump)
        qs.append(q); ds.append(d); otis.append(dump["oti"])
        out["B_packed_%d" % t] = np.packbits(dump["B"], axis=1)
        out["Dq_rowmax_%d" % t] = dump["Dq"].max(axis=1)
        out["S_diag_%d" % t] = np.diag(dump["S"]).copy()
        out["CSM_row0_%d" % t] = dump["CSM"][0].copy()
    out.update(chroma_qmax=np.array(qs), chroma_dmax=np.array(ds), oti=np.array(otis))
    np.savez_compressed(os.path.join(HERE, "pairs_1000.npz"), **out)
    print("pairs_1000.npz", qs, ds)


def gen_evalstats():
    """CoverAlgorithm.getEvalStatistics on seeded score matrices (CoverAlgorithm.py:330-418)."""
    sys.modules.setdefault("deepdish", types.ModuleType("deepdish"))  # see module docstring
    import CoverAlgorithm as refca
    out = {"meta": meta()}
    rng = np.random.default_rng(5)
    specs = [  # (clique sizes, noise)
        ([4] * 6, 0.3), ([5, 3, 3, 2, 2, 1, 1, 1], 0.5), ([2] * 20, 1.0), ([13] * 3 + [1] * 11, 0.4),
        ([3, 3, 3, 3], 0.0),
    ]
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)  # getEvalStatistics appends results_<shortname>.csv to the cwd
    try:
        for c, (sizes, noise) in enumerate(specs):
            N = int(np.sum(sizes))
            songs = rng.permutation(N)
            labels = np.zeros(N, dtype=int)
            pos = 0
            for lab, sz in enumerate(sizes):
                labels[songs[pos:pos + sz]] = lab
                pos += sz
            same = (labels[:, None] == labels[None, :]).astype(float)
            D = same + noise * rng.standard_normal((N, N))
            D = np.round(D * 8) / 8 if c == 4 else D       # case 4: heavy score ties
            D = np.triu(D, 1)
            D = (D + D.T).astype(np.float32)
            alg = refca.CoverAlgorithm.__new__(refca.CoverAlgorithm)
            alg.name, alg.shortname = "Golden", "golden%d" % c
            alg.Ds = {"main": D}
            alg.cliques = {}
            for i in range(N):                              # insertion order as load_features would
                alg.cliques.setdefault("clique_%d" % labels[i], set()).add(i)
            with contextlib.redirect_stdout(io.StringIO()):
                MR, MRR, MDR, MAP, tops = alg.getEvalStatistics("main")
            out["e%d_D" % c] = D
            out["e%d_labels" % c] = labels
            out["e%d_stats" % c] = np.array([MR, MRR, MDR, MAP] + list(tops))
    finally:
        os.chdir(cwd)
    out["n_cases"] = len(specs)
    np.savez_compressed(os.path.join(HERE, "evalstats.npz"), **out)
    print("evalstats.npz")


def _config1_worker(args):
    lo, hi = args
    corpus = synth.config1()
    pairs = synth.all_pairs(corpus.n_songs)[lo:hi]
    q = np.zeros(len(pairs)); d = np.zeros(len(pairs))
    for t, (i, j) in enumerate(pairs):
        q[t], d[t] = chain(corpus.song(i), corpus.gchroma[i], corpus.song(j), corpus.gchroma[j],
                           9, 0.095, True)
    return q, d


def gen_config1():
    """BASELINE config 0/1: covers80-shaped corpus, every pair through the reference chain, then
    the reference's MAP.  ~12.7k pairs of reference Python: minutes on 8 processes."""
    import multiprocessing as mp
    import zlib
    sys.modules.setdefault("deepdish", types.ModuleType("deepdish"))
    import CoverAlgorithm as refca
    corpus = synth.config1()
    pairs = synth.all_pairs(corpus.n_songs)
    K = len(pairs)
    bounds = np.linspace(0, K, 65).astype(int)
    with mp.Pool(8) as pool:
        parts = pool.map(_config1_worker, list(zip(bounds[:-1], bounds[1:])))
    q = np.concatenate([p[0] for p in parts]); d = np.concatenate([p[1] for p in parts])
    stats = {}
    cwd = os.getcwd(); tmp = tempfile.mkdtemp(); os.chdir(tmp)
    try:
        for key, v in (("chroma_qmax", q), ("chroma_dmax", d)):
            D = np.zeros((corpus.n_songs, corpus.n_songs), dtype=np.float32)
            D[pairs[:, 0], pairs[:, 1]] = v
            D += D.T                                       # CoverAlgorithm.py:180-182
            alg = refca.CoverAlgorithm.__new__(refca.CoverAlgorithm)
            alg.name, alg.shortname = "Golden", "config1"
            alg.Ds = {key: D}
            alg.cliques = corpus.cliques()
            with contextlib.redirect_stdout(io.StringIO()):
                MR, MRR, MDR, MAP, tops = alg.getEvalStatistics(key)
            stats[key] = np.array([MR, MRR, MDR, MAP] + list(tops))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "config1_scores.npz"), meta=meta(),
                        corpus_crc=np.array([zlib.crc32(corpus.feats.tobytes())]),
                        chroma_qmax=q, chroma_dmax=d, stats_qmax=stats["chroma_qmax"],
                        stats_dmax=stats["chroma_dmax"])
    print("config1_scores.npz", stats)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config1", action="store_true", help="also run the slow config-1 corpus")
    ap.add_argument("--swc-only", action="store_true", help="only serra09_swc.npz (round 4)")
    args = ap.parse_args()
    if args.swc_only:
        gen_serra09_swc()
        sys.exit(0)
    gen_stages()
    gen_dp()
    gen_serra09_mini()
    gen_serra09_swc()
    gen_pairs_1000()
    gen_evalstats()
    if args.config1:
        gen_config1()
