"""Quick GPU check of the fused band kernel (csrc/band_kernels.hip) against the materialising float64 path:
identical mask bits on the golden 1000-frame pairs and on a ragged corpus, then kernel times at bench size.
    python tools/fused_check.py [pairs]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acoss_amd import engine, synth  # noqa: E402


def bits_of(corpus, pairs, kappa=0.095, fused=True, mutual=True):
    batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    if corpus.gchroma is not None:
        engine.oti(corpus, batch)
    if fused:
        bits, work = engine.mask_bits_fused(corpus, batch, kappa, mutual=mutual)
        asked = int(engine.fused_counter(work).item())
    else:
        xp = engine.pack_x(corpus, batch)
        planes = engine.crp_planar(corpus, batch, xp)
        bits, _ = engine.mask_bits_planar(planes, corpus, batch, kappa, mutual=mutual)
        asked = -1
    torch.cuda.synchronize()
    return batch, bits, asked


def compare(corpus, pairs, name, mutual=True):
    b0, bits0, _ = bits_of(corpus, pairs, fused=False, mutual=mutual)
    b1, bits1, asked = bits_of(corpus, pairs, fused=True, mutual=mutual)
    bad = 0
    for p in range(b0.K):
        m0 = engine.unpack_mask_bits(bits0, b0, p)
        m1 = engine.unpack_mask_bits(bits1, b1, p)
        if not np.array_equal(m0, m1):
            bad += 1
            if bad <= 3:
                diff = np.argwhere(m0 != m1)
                print("  pair %d %s: %d cells differ, first %s; rows %s cols %s" % (p, m0.shape, len(diff), diff[:4].tolist(),
                      np.unique(diff[:, 0])[:8].tolist(), np.unique(diff[:, 1])[:8].tolist()))
    print("%s: %d pairs, %d differ, undecided rows %d (mutual=%s)" % (name, b0.K, bad, asked, mutual))
    return bad == 0


def main():
    ok = True
    g = np.load(os.path.join(ROOT, "tests", "golden", "pairs_1000.npz"))
    corpus = engine.DeviceCorpus(g["feats"], g["frame_off"], gchroma=g["gchroma"])
    ok &= compare(corpus, g["pairs"], "golden pairs_1000")
    ok &= compare(corpus, g["pairs"], "golden pairs_1000", mutual=False)
    lens = iter([9, 10, 12, 33, 64, 65, 100, 131, 257, 300, 511, 640, 777, 1000, 1021, 1022])
    ch = synth.make_corpus(8, 2, seed=79, lengths=lambda r: next(lens))
    corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = np.array([(i, j) for i in range(16) for j in range(16)], dtype=np.int32)
    ok &= compare(corpus, pairs, "ragged 9..1022")
    if not ok:
        print("MISMATCH")
        sys.exit(1)
    # timing at bench size
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ch = synth.config2(n_songs=200, n_frames=1000)
    corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
    pairs = synth.all_pairs(ch.n_songs)[:P]
    batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
    engine.oti(corpus, batch)
    band = engine.planar32_band(corpus, batch, fused=True)
    bits, work = engine.mask_bits_fused(corpus, batch, 0.095, band=band)
    print("undecided rows: %d of %d" % (int(engine.fused_counter(work).item()), batch.K * 2 * 992))
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.mask_bits_fused(corpus, batch, 0.095, band=band, out=bits, work=work, verify=False)
        e1.record()
        torch.cuda.synchronize()
        print("mask_bits_fused %d pairs: %.3f ms" % (P, e0.elapsed_time(e1)))
    q = engine.align_bits("qmax", bits, batch)
    torch.cuda.synchronize()
    # reference chain
    xp = engine.pack_x(corpus, batch)
    planes = engine.crp_planar(corpus, batch, xp)
    bits0, _ = engine.mask_bits_planar(planes, corpus, batch, 0.095)
    q0 = engine.align_bits("qmax", bits0, batch)
    print("scores identical to the float64 path on %d pairs: %s" % (P, bool(torch.equal(q, q0))))
    print("bits identical: %s" % bool(torch.equal(bits[:batch.K * 992 * 16], bits0[:batch.K * 992 * 16])))


if __name__ == "__main__":
    main()
