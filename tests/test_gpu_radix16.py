"""The radix form of the kNN selection on the 16-bit key plane (csrc/radix16_kernels.hip, round 5; CRPUtils.py:169-219):
the row / column bounds t1 and the work items against numpy on the key plane itself, the masks against the wave-per-row
kernels of round 3-4 (ACOSS_RADIX16=0) and against the float64 masks -- ragged batches at every pitch alignment, one-sided
masks, absolute and fractional kappa, temporally smooth features (threads with more hits than slots: the third sweep), and
corpora with exact ties, whose pairs the radix kernels must hand back (list kernels)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from acoss_amd import engine
    engine.require_gpu()
    return engine


def _chain(eng, corpus, batch):
    xp32 = eng.pack_x32(corpus, batch)
    koff, band = eng.keys16_koff(corpus, batch), eng.planar32_band(corpus, batch)
    return xp32, koff, band, eng.crp_keys16(corpus, batch, xp32, koff)


def _masks(eng, k16, band, koff, xp32, corpus, batch, kappa, mutual, radix):
    old = os.environ.get("ACOSS_RADIX16")
    os.environ["ACOSS_RADIX16"] = "1" if radix else "0"
    try:
        bits, work = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, kappa, mutual=mutual)
        st = (ctypes.c_int * 20)()
        if radix:
            eng.check(eng._lib.load().acoss_mask_bits_keys16_stats(eng._ptr(work), batch.K, batch.max_nx, batch.max_ny, 9, st), "stats")
        return bits.clone(), list(st)
    finally:
        if old is None:
            os.environ.pop("ACOSS_RADIX16", None)
        else:
            os.environ["ACOSS_RADIX16"] = old


def test_bounds_and_items_against_numpy_on_the_key_plane(eng):
    """Stage entry points: every row's / column's bound t1 is th + 1 where exactly k keys are <= th (the k-th smallest key) and
    no key equals th + 1, else th - 1 with a work item; nothing is flagged on tie-free features."""
    import torch
    from acoss_amd import synth, _lib
    lib = _lib.load()
    ch = synth.make_corpus(6, 3, seed=11, lengths=lambda r: int(r.integers(60, 1033)))
    corpus = eng.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = synth.all_pairs(ch.n_songs)[:120]
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    eng.oti(corpus, batch)
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    K = batch.K
    nb = lib.acoss_radix16_work_bytes(K, batch.max_nx, batch.max_ny, 9)
    work = torch.empty(nb, dtype=torch.uint8, device=k16.device)
    bits = torch.zeros(K * (batch.max_nx - 8) * 16, dtype=torch.int64, device=k16.device)
    eng.check(lib.acoss_radix16_stage(7, eng._ptr(k16), eng._ptr(band), eng._ptr(koff), eng._ptr(corpus.feats), eng._ptr(corpus.norms),
                                      corpus.d, eng._ptr(batch.descs_dev), K, 9, batch.max_nx, batch.max_ny, 0.095, 1, eng._ptr(bits),
                                      eng._ptr(work), work.numel(), eng._stream()), "radix16_stage")
    ptrs, dims = (ctypes.c_void_p * 8)(), (ctypes.c_int * 4)()
    lib.acoss_radix16_layout(eng._ptr(work), K, batch.max_nx, batch.max_ny, 9, ptrs, dims)
    ldm, ldn = dims[0], dims[1]

    def view(i, dtype, count):
        off = ptrs[i] - work.data_ptr()
        return work[off:off + count * torch.tensor([], dtype=dtype).element_size()].view(dtype).cpu().numpy()
    t1r, t1c = view(0, torch.int16, K * ldm).view(np.uint16).reshape(K, ldm), view(1, torch.int16, K * ldn).view(np.uint16).reshape(K, ldn)
    ir, ic = view(2, torch.int32, K * ldm).reshape(K, ldm), view(3, torch.int32, K * ldn).reshape(K, ldn)
    counters = view(4, torch.int32, 20)
    assert counters[1] == 0 and counters[2] == 0, " ".join(str(int(c)) for c in counters)      # no line flagged its pair
    kh = k16.cpu().numpy().view(np.uint16)
    for p in range(0, K, 7):
        d = batch.descs[p]
        M, N, pitch = int(d["nx"]) - 8, int(d["ny"]) - 8, int(d["crp_pitch"])
        pl = kh[int(d["crp_off"]):int(d["crp_off"]) + M * pitch].reshape(M, pitch)[:, :N].astype(np.int64)
        for mat, t1, it, k in ((pl.T, t1c[p, :N], ic[p, :N], int(np.rint(0.095 * M))), (pl, t1r[p, :M], ir[p, :M], int(np.rint(0.095 * N)))):
            if k <= 0:
                assert (t1 == 0).all() and (it < 0).all()
                continue
            s = np.sort(mat, axis=1)
            th = s[:, k - 1]
            clean = ((mat <= th[:, None]).sum(1) == k) & ~(mat == th[:, None] + 1).any(1)
            assert np.array_equal(it < 0, clean), p
            assert np.array_equal(t1.astype(np.int64), np.where(clean, th + 1, th - 1)), p


@pytest.mark.parametrize("align", [32, 2, 1])
def test_masks_equal_the_wave_per_row_kernels_and_float64(eng, align):
    import torch
    from acoss_amd import synth
    lens = iter([9, 33, 40, 65, 129, 300, 1000, 1032])
    small = synth.make_corpus(4, 2, seed=83, lengths=lambda r: next(lens))
    corpus = eng.DeviceCorpus(small.feats, small.frame_off, gchroma=small.gchroma)
    pairs = np.array([(i, j) for i in range(8) for j in range(8)], dtype=np.int32)
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=align)
    eng.oti(corpus, batch)
    T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    for mutual in (True, False):
        for kappa in (0.095, 0.5, 3, 200, 0):
            want, _ = eng.mask_bits(T, batch, kappa, mutual=mutual)
            old, _ = _masks(eng, k16, band, koff, xp32, corpus, batch, kappa, mutual, radix=False)
            new, st = _masks(eng, k16, band, koff, xp32, corpus, batch, kappa, mutual, radix=True)
            assert torch.equal(old, want), (mutual, kappa)
            if not torch.equal(new, want):
                for p in range(batch.K):
                    assert np.array_equal(eng.unpack_mask_bits(new, batch, p), eng.unpack_mask_bits(want, batch, p)), (mutual, kappa, p, st)


def test_smooth_features_take_the_third_sweep(eng):
    """AR(1) frames (synth.config2_smooth) and 13-dimensional random walks: runs of neighbouring cells inside one key window give
    some threads more hits than slots; their blocks find the items' cells by a third sweep.  Masks as the old kernels'."""
    import torch
    from acoss_amd import synth
    ch = synth.config2_smooth(n_songs=16, n_frames=1000, rho=0.97)
    rng = np.random.default_rng(4)
    walks = [np.cumsum(rng.standard_normal((n, 12)) * 0.05, axis=0) for n in (700, 1000, 1032, 400)]
    feats = np.concatenate([ch.feats] + [w - w.min() + 0.1 for w in walks])
    off = np.concatenate([ch.frame_off, ch.frame_off[-1] + np.cumsum([len(w) for w in walks])]).astype(np.int64)
    gc = np.concatenate([ch.gchroma, np.ones((4, 12))])
    corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
    pairs = np.array([(i, j) for i in range(20) for j in range(i + 1, 20)], dtype=np.int32)
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    eng.oti(corpus, batch)
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    old, _ = _masks(eng, k16, band, koff, xp32, corpus, batch, 0.095, True, radix=False)
    new, st = _masks(eng, k16, band, koff, xp32, corpus, batch, 0.095, True, radix=True)
    assert torch.equal(new, old), st


def test_exact_ties_hand_their_pairs_back(eng):
    """Periodic songs: every row holds a handful of distinct values many times over -- more cells in reach than an item holds.
    The radix kernels flag those pairs, the list kernels redo them; tie-free pairs of the same batch stay on the radix path."""
    import torch
    from acoss_amd import synth
    rng = np.random.default_rng(5)
    pat7, pat5 = rng.random((7, 12)) + 0.1, rng.random((5, 12)) + 0.1
    A, B = np.tile(pat7, (30, 1))[:200], np.tile(pat5, (31, 1))[:151]
    ch = synth.make_corpus(2, 2, seed=3, lengths=lambda r: int(r.integers(150, 400)))
    feats = np.concatenate([A, B, ch.feats])
    off = np.concatenate([[0, len(A), len(A) + len(B)], len(A) + len(B) + ch.frame_off[1:]]).astype(np.int64)
    gc = np.concatenate([np.stack([x.sum(0) / x.sum(0).max() for x in (A, B)]), ch.gchroma])
    corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
    pairs = np.tile(np.array([(i, j) for i in range(6) for j in range(6)], dtype=np.int32), (10, 1))        # 360 pairs: more than the list kernels cover at a time
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    eng.oti(corpus, batch)
    T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    for mutual in (True, False):
        want, _ = eng.mask_bits(T, batch, 0.095, mutual=mutual)
        new, st = _masks(eng, k16, band, koff, xp32, corpus, batch, 0.095, mutual, radix=True)
        assert 0 < st[2] < batch.K, st                      # some pairs handed back, not all
        assert torch.equal(new, want), (mutual, st)


def _byte_mask(B, batch, p):
    d = batch.descs[p]
    M, N = int(batch.M[p]), int(batch.N[p])
    return B[int(d["crp_off"]):int(d["crp_off"]) + M * int(d["crp_pitch"])].cpu().numpy().reshape(M, -1)[:, :N]


def test_long_form_up_to_2048(eng):
    """Songs of 1033 .. 2056 frames (a side of 1025 .. 2048): the radix selection with 64 dwords of keys per thread, 32-word mask
    rows.  Masks as the float64 path's for every pair it resolves -- all of them on tie-free features, both one-sided and mutual."""
    import torch
    from acoss_amd import synth
    lens = iter([2056, 1033, 300, 1500, 9, 1990, 1040, 1100])
    ch = synth.make_corpus(4, 2, seed=29, lengths=lambda r: next(lens))
    corpus = eng.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = np.array([(i, j) for i in range(8) for j in range(8)], dtype=np.int32)
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    assert eng.bits_words(batch) == 32
    eng.oti(corpus, batch)
    T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    for mutual in (True, False):
        for kappa in (0.095, 0.4, 7):
            B = eng.binarize(T, batch, kappa, mutual=mutual)           # (bit-serial selection on the float64 sums, byte mask)
            new, st = _masks(eng, k16, band, koff, xp32, corpus, batch, kappa, mutual, radix=True)
            assert st[2] == 0, st
            for p in range(batch.K):
                assert np.array_equal(eng.unpack_mask_bits(new, batch, p), _byte_mask(B, batch, p)), (mutual, kappa, p, st)


def test_long_form_lists_the_pairs_it_cannot_express(eng):
    """Periodic 1500-frame songs (exact ties by the hundred): their pairs are listed as unresolved, the others' masks are the
    float64 path's; the scorer (acoss_serra09_scores) redoes the listed pairs on the float64 path -- tests/test_gpu_scorer.py."""
    import torch
    from acoss_amd import synth
    rng = np.random.default_rng(6)
    pat = rng.random((7, 12)) + 0.1
    A = np.tile(pat, (215, 1))[:1500]
    ch = synth.make_corpus(3, 1, seed=31, lengths=lambda r: int(r.integers(1100, 1400)))
    feats = np.concatenate([A, ch.feats])
    off = np.concatenate([[0, len(A)], len(A) + ch.frame_off[1:]]).astype(np.int64)
    gc = np.concatenate([(A.sum(0) / A.sum(0).max())[None, :], ch.gchroma])
    corpus = eng.DeviceCorpus(feats, off, gchroma=gc)
    pairs = np.array([(i, j) for i in range(4) for j in range(4)], dtype=np.int32)
    batch = eng.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    eng.oti(corpus, batch)
    T = eng.crp(corpus, batch, eng.pack_x(corpus, batch))
    xp32, koff, band, k16 = _chain(eng, corpus, batch)
    B = eng.binarize(T, batch, 0.095, mutual=True)
    bits, work = eng.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, mutual=True)
    un = set(int(p) for p in eng.mask_bits_keys16_unresolved(work, batch))
    assert un and 0 in un and len(un) < batch.K, un             # song 0 against itself: ties everywhere
    for p in range(batch.K):
        if p not in un:
            assert np.array_equal(eng.unpack_mask_bits(bits, batch, p), _byte_mask(B, batch, p)), p


def test_long_float32_corpus_through_the_engine(eng):
    """13-dimensional float32 features (the reference's mfcc_htk) with songs of 1033 .. 1600 frames through engine.serra09_scores: the
    long form of the filter on the corpus itself, a periodic song's pairs (exact ties) redone without the filter; every score equals
    the plain float32-input chain's (approx32=False)."""
    rng = np.random.default_rng(9)
    lens = [1033, 1600, 1200, 1100, 700]
    songs = [(np.cumsum(rng.standard_normal((n, 13)), axis=0) * 0.3 + rng.standard_normal((n, 13))).astype(np.float32) for n in lens]
    pat = rng.random((7, 13)).astype(np.float32) + 0.1
    songs.append(np.tile(pat, (200, 1))[:1300])
    feats = np.concatenate(songs)
    off = np.cumsum([0] + [len(s) for s in songs]).astype(np.int64)
    corpus = eng.DeviceCorpus(feats, off)
    pairs = np.array([(i, j) for i in range(6) for j in range(6) if i != j] + [(5, 5)], dtype=np.int32)
    want = eng.serra09_scores(corpus, pairs, do_oti=False, approx32=False)
    got = eng.serra09_scores(corpus, pairs, do_oti=False)
    assert np.array_equal(got["qmax"], want["qmax"]) and np.array_equal(got["dmax"], want["dmax"])
