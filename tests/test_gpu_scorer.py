"""The one-call scorer of the C ABI (include/acoss_mi355x.h group (3): acoss_corpus_create + acoss_serra09_scores) driven
through raw ctypes -- no engine, no torch tensors in the call -- against the CPU oracle: the binding a non-Python host
would write (INTEGRATION.md level 2)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import torch
    assert torch.cuda.is_available()
    from acoss_amd import _lib
    return _lib.load()


def _scores(lib, feats, frame_off, gchroma, pairs, want=3, m=9, kappa=0.095, do_oti=1, batch_pairs=0):
    import torch
    feats = np.ascontiguousarray(feats, dtype=np.float64)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    K = len(pairs)
    h = ctypes.c_void_p()
    g = np.ascontiguousarray(gchroma, dtype=np.float64) if gchroma is not None else None
    rc = lib.acoss_corpus_create(feats.ctypes.data, frame_off.ctypes.data, len(frame_off) - 1, feats.shape[1],
                                 g.ctypes.data if g is not None else None, g.shape[1] if g is not None else 0, ctypes.byref(h))
    assert rc == 0, lib.acoss_last_error()
    try:
        need = lib.acoss_serra09_scratch_bytes(h, pairs.ctypes.data, K, m, batch_pairs)
        assert need > 0, lib.acoss_last_error()
        scratch = torch.empty(need, dtype=torch.uint8, device="cuda")       # any device allocation will do
        out = [np.full(K, np.nan) for _ in range(3)]
        rc = lib.acoss_serra09_scores(h, pairs.ctypes.data, K, m, float(kappa), do_oti, want, batch_pairs,
                                      ctypes.c_void_p(scratch.data_ptr()), need,
                                      out[0].ctypes.data if want & 1 else None, out[1].ctypes.data if want & 2 else None,
                                      out[2].ctypes.data if want & 4 else None, None)
        assert rc == 0, lib.acoss_last_error()
        return out
    finally:
        lib.acoss_corpus_destroy(h)


def test_scorer_on_ragged_pairs_equals_oracle(lib, orc):
    """4000 random pairs (both orientations) of a 350-song corpus with lengths 60..1032, in batches of 1500."""
    from acoss_amd import synth
    rng = np.random.default_rng(3)
    ch = synth.make_corpus(40, 8, seed=3, singletons=30, lengths=lambda r: int(np.clip(r.normal(520, 160), 60, 1032)))
    allp = synth.all_pairs(ch.n_songs)
    pairs = allp[rng.permutation(len(allp))[:4000]]
    pairs = np.where(rng.random((len(pairs), 1)) < 0.5, pairs, pairs[:, ::-1]).astype(np.int32)
    q, d, _ = _scores(lib, ch.feats, ch.frame_off, ch.gchroma, pairs, want=3, batch_pairs=1500)
    qo, do, _ = orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, pairs, nthreads=min(os.cpu_count() or 1, 16))
    assert np.array_equal(q, qo) and np.array_equal(d, do)


def test_scorer_size_classes_and_smith_waterman(lib, orc):
    """Songs of 400 .. 3000 frames in one call: the three size classes each take their own kernels; qmax / dmax equal the
    oracle's, swalignimpconstrained within 1e-5 (its -0.7 penalty is inexact, SequenceAlignment.c:46)."""
    from acoss_amd import synth
    lens_it = iter([2300, 400, 1500, 3000, 900])
    ch = synth.make_corpus(5, 1, seed=99, lengths=lambda r: next(lens_it))
    pairs = np.array([(0, 1), (1, 0), (2, 4), (3, 0), (4, 4), (1, 1), (4, 2), (1, 4)], dtype=np.int32)
    q, d, s = _scores(lib, ch.feats, ch.frame_off, ch.gchroma, pairs, want=7)
    for t, (i, j) in enumerate(pairs):
        X, Y = ch.song(i), ch.song(j)
        qo, do = orc.serra09_pair(X, ch.gchroma[i], Y, ch.gchroma[j])
        assert q[t] == qo and d[t] == do, (t, i, j)
        if t in (1, 2, 7):
            B = orc.csm_to_binary_mutual(orc.sliding_csm(orc.get_csm(X, Y, orc.get_oti(ch.gchroma[i], ch.gchroma[j])), 9), 0.095)
            M, N = B.shape
            D = np.zeros((M + 1) * (N + 1), dtype=np.float32)
            so = orc.swconstrained(np.ascontiguousarray(B.flatten()), D, M, N) / (M + N)
            assert abs(s[t] - so) <= 1e-5, (t, s[t], so)


def test_scorer_long_songs_with_exact_ties_come_round_again(lib, orc):
    """Songs of 1033 .. 2056 frames take the 16-bit keys' long form (round 5); a periodic song's pairs hold exact ties by the
    hundred, the radix selection lists them as unresolved and the scorer scores them again on the float64 path: every pair
    equals the oracle, with the radix selection on and off."""
    from acoss_amd import synth
    rng = np.random.default_rng(12)
    pat = rng.random((7, 12)) + 0.1
    A = np.tile(pat, (200, 1))[:1300]
    ch = synth.make_corpus(3, 1, seed=77, lengths=lambda r: int(r.integers(1040, 1300)))
    feats = np.concatenate([A, ch.feats])
    off = np.concatenate([[0, len(A)], len(A) + ch.frame_off[1:]]).astype(np.int64)
    gc = np.concatenate([(A.sum(0) / A.sum(0).max())[None, :], ch.gchroma])
    pairs = np.array([(0, 1), (1, 2), (0, 0), (2, 3), (3, 0), (3, 1)], dtype=np.int32)
    free = [1, 3, 5]                # (pairs with the periodic song: which of a line's equal values np.argpartition takes is numpy's
                                    #  choice -- CRPUtils.py:192; here the lowest positions, and the float64 path is the reference)
    want_q, want_d = zip(*[orc.serra09_pair(feats[off[i]:off[i + 1]], gc[i], feats[off[j]:off[j + 1]], gc[j]) for i, j in pairs[free]])
    old = {k: os.environ.get(k) for k in ("ACOSS_RADIX16", "ACOSS_PLANAR32")}
    try:
        os.environ["ACOSS_PLANAR32"] = "0"                      # the float64 keys for every pair
        q64, d64, _ = _scores(lib, feats, off, gc, pairs, want=3)
        os.environ.pop("ACOSS_PLANAR32")
        assert np.array_equal(q64[free], np.array(want_q)) and np.array_equal(d64[free], np.array(want_d))
        for flag in ("1", "0"):
            os.environ["ACOSS_RADIX16"] = flag
            q, d, _ = _scores(lib, feats, off, gc, pairs, want=3)
            assert np.array_equal(q, q64) and np.array_equal(d, d64), flag
        os.environ["ACOSS_RADIX16"] = "1"
        q, d, _ = _scores(lib, feats, off, gc, pairs, want=3, batch_pairs=2)      # three batches, each with pairs that come round again
        assert np.array_equal(q, q64) and np.array_equal(d, d64)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_scorer_switches_on_fresh_handles(lib, orc):
    """ACOSS_KEYS16 / ACOSS_PLANAR32 are read when a corpus handle is made ("0", "false", "no", "" switch off, as in the Python
    engine): 32-bit keys and the all-float64 keys give the default's scores, which are the oracle's."""
    from acoss_amd import synth
    ch = synth.make_corpus(5, 2, seed=41, lengths=lambda r: int(r.integers(80, 700)))
    pairs = synth.all_pairs(ch.n_songs).astype(np.int32)
    q0, d0, _ = _scores(lib, ch.feats, ch.frame_off, ch.gchroma, pairs, want=3)
    qo, do, _ = orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, pairs, nthreads=min(os.cpu_count() or 1, 16))
    assert np.array_equal(q0, qo) and np.array_equal(d0, do)
    for name, value in (("ACOSS_KEYS16", "0"), ("ACOSS_KEYS16", "false"), ("ACOSS_PLANAR32", "0"), ("ACOSS_PLANAR32", "no"), ("ACOSS_RADIX16", "")):
        old = os.environ.get(name)
        os.environ[name] = value
        try:
            q, d, _ = _scores(lib, ch.feats, ch.frame_off, ch.gchroma, pairs, want=3)
        finally:
            if old is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = old
        assert np.array_equal(q, q0) and np.array_equal(d, d0), (name, value)


def test_scorer_other_widths_windows_and_errors(lib, orc):
    """20-dimensional features and a window of 5 have no fused kernel: one kernel per function, all three recurrences
    (the advisor's case: a silent zero for swc there).  Bad indices and short songs are errors, not zeros."""
    rng = np.random.default_rng(8)
    lens = [90, 140, 33]
    feats = rng.random((sum(lens), 20))
    off = np.concatenate([[0], np.cumsum(lens)])
    pairs = np.array([(0, 1), (1, 2), (2, 0)], dtype=np.int32)
    q, d, s = _scores(lib, feats, off, None, pairs, want=7, m=5, kappa=0.1, do_oti=0)
    for t, (i, j) in enumerate(pairs):
        X, Y = feats[off[i]:off[i + 1]], feats[off[j]:off[j + 1]]
        B = orc.csm_to_binary_mutual(orc.sliding_csm(orc.get_csm(X, Y, 0), 5), 0.1)
        M, N = B.shape
        Bf = np.ascontiguousarray(B.flatten())
        D = np.zeros(M * N, dtype=np.float32)
        assert q[t] == orc.qmax(Bf, D, M, N) / (M + N)
        assert d[t] == orc.dmax(Bf, D, M, N) / (M + N)          # on the D qmax left behind (Serra09.py:173-175)
        D = np.zeros((M + 1) * (N + 1), dtype=np.float32)
        assert abs(s[t] - orc.swconstrained(Bf, D, M, N) / (M + N)) <= 1e-5 and s[t] > 0
    h = ctypes.c_void_p()
    f64 = np.ascontiguousarray(feats)
    o64 = np.ascontiguousarray(off, dtype=np.int64)
    assert lib.acoss_corpus_create(f64.ctypes.data, o64.ctypes.data, 3, 20, None, 0, ctypes.byref(h)) == 0
    bad = np.array([[0, 7]], dtype=np.int32)
    assert lib.acoss_serra09_scratch_bytes(h, bad.ctypes.data, 1, 5, 0) == 0 and b"names song" in lib.acoss_last_error()
    short = np.array([[0, 2]], dtype=np.int32)
    assert lib.acoss_serra09_scratch_bytes(h, short.ctypes.data, 1, 40, 0) == 0 and b"shorter" in lib.acoss_last_error()
    one = np.zeros(1)
    assert lib.acoss_serra09_scores(h, pairs.ctypes.data, 1, 5, 0.1, 1, 1, 0, None, 0, one.ctypes.data, None, None, None) == -22
    lib.acoss_corpus_destroy(h)


def test_scorer_handles_run_concurrently_and_one_handle_serialises(lib, orc):
    """Thread safety as include/acoss_mi355x.h states it: two handles, each with its own scratch and stream, scored from
    two host threads at once give the results of the sequential calls; two threads calling on ONE handle are serialised
    by its mutex (they share its pinned staging) and both get correct scores."""
    import threading
    import torch
    from acoss_amd import synth
    corpora = [synth.make_corpus(6, 3, seed=s, lengths=lambda r: int(r.integers(200, 640))) for s in (41, 42)]
    rng = np.random.default_rng(5)
    plists = []
    for ch in corpora:
        allp = synth.all_pairs(ch.n_songs)
        plists.append(np.ascontiguousarray(allp[rng.permutation(len(allp))[:120]], dtype=np.int32))
    expect = [orc.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, p, nthreads=8)[:2] for ch, p in zip(corpora, plists)]
    handles, scratch, streams = [], [], []
    for ch, p in zip(corpora, plists):
        h = ctypes.c_void_p()
        fo = np.ascontiguousarray(ch.frame_off, dtype=np.int64)
        assert lib.acoss_corpus_create(ch.feats.ctypes.data, fo.ctypes.data, ch.n_songs, 12, ch.gchroma.ctypes.data, 12, ctypes.byref(h)) == 0
        need = lib.acoss_serra09_scratch_bytes(h, p.ctypes.data, len(p), 9, 40)
        handles.append(h)
        scratch.append([torch.empty(need, dtype=torch.uint8, device="cuda") for _ in range(2)])
        streams.append(torch.cuda.Stream())
    results = {}

    def call(tag, which, buf, stream_ptr):
        p = plists[which]
        q, d = np.full(len(p), np.nan), np.full(len(p), np.nan)
        for _ in range(3):
            rc = lib.acoss_serra09_scores(handles[which], p.ctypes.data, len(p), 9, 0.095, 1, 3, 40, ctypes.c_void_p(buf.data_ptr()),
                                          buf.numel(), q.ctypes.data, d.ctypes.data, None, ctypes.c_void_p(stream_ptr))
            assert rc == 0, lib.acoss_last_error()
        results[tag] = (q, d)

    try:
        threads = [threading.Thread(target=call, args=("a", 0, scratch[0][0], streams[0].cuda_stream)),
                   threading.Thread(target=call, args=("b", 1, scratch[1][0], streams[1].cuda_stream)),
                   threading.Thread(target=call, args=("a2", 0, scratch[0][1], streams[1].cuda_stream))]     # same handle as "a"
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        for tag, which in (("a", 0), ("b", 1), ("a2", 0)):
            assert np.array_equal(results[tag][0], expect[which][0]), tag
            assert np.array_equal(results[tag][1], expect[which][1]), tag
    finally:
        for h in handles:
            lib.acoss_corpus_destroy(h)


def test_first_product_call_makes_one_scratch_allocation_and_no_trial_batches(monkeypatch):
    """engine.serra09_scores on a fresh process state: the scorer's scratch is ONE plain allocation and the C scorer is entered
    exactly once (rounds 2-3 allocated several candidates and timed trial batches in each: retired with the row-band kernel)."""
    import torch
    from acoss_amd import engine, synth
    engine.require_gpu()
    engine.release_scratch()
    ch = synth.make_corpus(6, 2, seed=3, lengths=lambda r: r.integers(200, 420))
    corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = synth.all_pairs(ch.n_songs)
    lib = engine._lib.load()
    calls, allocs = [], []
    real_scores, real_empty = lib.acoss_serra09_scores, torch.empty

    def counting_scores(*a):
        calls.append(1)
        return real_scores(*a)

    def counting_empty(*a, **kw):
        t = real_empty(*a, **kw)
        if t.is_cuda and t.dtype == torch.uint8:
            allocs.append(t.numel())
        return t
    monkeypatch.setattr(lib, "acoss_serra09_scores", counting_scores)
    monkeypatch.setattr(engine.torch, "empty", counting_empty)
    got = engine.serra09_scores(corpus, pairs)
    assert len(calls) == 1 and len(allocs) == 1, (calls, allocs)
    again = engine.serra09_scores(corpus, pairs)                     # the scratch is reused
    assert len(calls) == 2 and len(allocs) == 1
    assert np.array_equal(got["qmax"], again["qmax"]) and np.max(got["qmax"]) > 0
