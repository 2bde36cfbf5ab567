"""qmax and dmax in 16-bit integers (dp_bits_q16_kernel / dp_bits_d16_kernel: E = 2 D on packed u16 instructions) against the
oracle's qmax_c / dmax_c restatements (SequenceAlignment.c:113-180) on masks the selection never produces: dense ones, where the alignment values grow to the
matrix size, all-ones and all-zeros, and every width class (a lane boundary, a register boundary, one column)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pack(masks, max_m, W=16):
    """[K][max_m][W] uint64 words, bit c of word e of row i = mask[i][64 e + c] (engine.unpack_mask_bits' inverse)."""
    out = np.zeros((len(masks), max_m, W * 8), dtype=np.uint8)
    for p, B in enumerate(masks):
        M, N = B.shape
        padded = np.zeros((M, W * 64), dtype=np.uint8)
        padded[:, :N] = B
        out[p, :M] = np.packbits(padded, axis=1, bitorder="little")
    return out.reshape(-1).view(np.int64)


def test_integer_qmax_and_dmax_equal_oracle_on_dense_and_degenerate_masks(orc):
    import torch
    from acoss_amd import engine
    engine.require_gpu()
    rng = np.random.default_rng(16)
    shapes = [(1024, 1024), (992, 992), (1000, 17), (17, 1000), (3, 3), (2, 50), (50, 2), (64, 64), (65, 63), (300, 511), (300, 513),
              (129, 1023), (700, 33), (4, 1024), (1024, 4)]
    masks = []
    for t, (M, N) in enumerate(shapes):
        dens = [1.0, 0.0, 0.9, 0.5, 0.3, 0.07][t % 6]
        masks.append((rng.random((M, N)) < dens).astype(np.uint8))
    masks.append(np.ones((1024, 1024), dtype=np.uint8))                     # values reach 1022.0: E = 2044
    masks.append(np.eye(1024, dtype=np.uint8))
    masks.append(np.triu(np.ones((800, 900), dtype=np.uint8)))
    win = 9
    lens = np.array([[B.shape[0] + win - 1, B.shape[1] + win - 1] for B in masks], dtype=np.int64)
    # a corpus of dummy songs of the wanted lengths: only the descriptors matter here
    song_len = lens.reshape(-1)
    off = np.concatenate([[0], np.cumsum(song_len)]).astype(np.int64)
    pairs = np.array([(2 * p, 2 * p + 1) for p in range(len(masks))], dtype=np.int32)
    batch = engine.PairBatch(off, pairs, win, torch.device("cuda:0"))
    max_m = batch.max_nx - win + 1
    assert engine.bits_words(batch) == 16
    bits = torch.from_numpy(_pack(masks, max_m)).to("cuda:0")
    got = engine.align_bits("qmax", bits, batch).cpu().numpy()
    for p, B in enumerate(masks):
        M, N = B.shape
        D = np.zeros(M * N, dtype=np.float32)
        want = orc.qmax(np.ascontiguousarray(B.reshape(-1)), D, M, N)
        assert got[p] == want, (p, B.shape, got[p], want)
    assert got[len(shapes)] == 1022.0
    # dmax (dp_bits_d16_kernel; SequenceAlignment.c:147-180) on the same masks: on a fresh D (boundary 0) and on the D that
    # qmax leaves behind, as Serra09 drives it (Serra09.py:173-175: boundary 1)
    for boundary in (0, 1):
        gd = engine.align_bits("dmax", bits, batch, boundary=boundary).cpu().numpy()
        for p, B in enumerate(masks):
            M, N = B.shape
            Bf = np.ascontiguousarray(B.reshape(-1))
            D = np.zeros(M * N, dtype=np.float32)
            if boundary:
                orc.qmax(Bf, D, M, N)
            want = orc.dmax(Bf, D, M, N)
            assert gd[p] == want, (boundary, p, B.shape, gd[p], want)
        # ... and both recurrences in one sweep (dp_bits_qd16_kernel: the carried (i-3, j-1) term, the second table)
        q2, d2 = engine.align_bits_qd(bits, batch, boundary=boundary)
        assert np.array_equal(q2.cpu().numpy(), got) and np.array_equal(d2.cpu().numpy(), gd), boundary
