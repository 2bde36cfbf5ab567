"""Per-stage kernel timing of the staged Serra09 chain on synthetic 1000-frame pairs (dev tool)."""
import sys
import time

import numpy as np
import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
corpus_h = synth.make_corpus(8, 4, n_frames=1000, seed=20260)
corpus = engine.DeviceCorpus(corpus_h.feats, corpus_h.frame_off, gchroma=corpus_h.gchroma)
allp = synth.all_pairs(corpus_h.n_songs)
pairs = allp[np.arange(K) % len(allp)]
batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device)


def timed(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        out = fn()
        ev[r + 1].record()
    torch.cuda.synchronize()
    ms = np.array([ev[r].elapsed_time(ev[r + 1]) for r in range(reps)])
    print("%-22s median %8.3f ms  min %8.3f ms   per pair %8.2f us" % (name, np.median(ms), ms.min(), 1e3 * np.median(ms) / K))
    return out, float(np.median(ms))


tot = 0.0
_, t = timed("oti", lambda: engine.oti(corpus, batch)); tot += t
C, t = timed("csm_f64", lambda: engine.csm(corpus, batch)); tot += t
print("   csm: %.1f GB/s algorithmic (8.192 MB/pair)" % (K * 8.192e6 / (t * 1e-3) / 1e9))
S, t = timed("sliding", lambda: engine.sliding(C, batch)); tot += t
work = torch.empty(int(engine._lib.load().acoss_binarize_work_bytes(batch.K, batch.max_nx, batch.max_ny, 9)), dtype=torch.uint8, device=corpus.device)
Bout = torch.zeros(batch.total_crp, dtype=torch.uint8, device=corpus.device)
B, t = timed("binarize(mutual)", lambda: engine.binarize(S, batch, 0.095, True, out=Bout, work=work)); tot += t
mats, _ = batch.mats()
_, t = timed("qmax", lambda: engine.align("qmax", B, mats)); tot += t
_, t2 = timed("dmax", lambda: engine.align("dmax", B, mats, boundary=1))
_, t3 = timed("swc", lambda: engine.align("swc", B, mats))
print("chain (oti+csm+sliding+binarize+qmax): %.3f ms for %d pairs -> %.0f pair-scores/s" % (tot, K, K / (tot * 1e-3)))
# fast forms
xp, t = timed("pack_x", lambda: engine.pack_x(corpus, batch)); tf = t
_, t = timed("csm_packed_f64", lambda: engine.csm_packed(corpus, batch, xp, out=C))
print("   csm_packed: %.1f GB/s algorithmic" % (K * 8.192e6 / (t * 1e-3) / 1e9))
_, t = timed("csm_strip_f64", lambda: engine.csm_strip(corpus, batch, xp, out=C))
print("   csm_strip: %.1f GB/s algorithmic" % (K * 8.192e6 / (t * 1e-3) / 1e9))
Tb = torch.empty(batch.total_crp, dtype=torch.float64, device=corpus.device)
_, t = timed("crp (squared)", lambda: engine.crp(corpus, batch, xp, False, out=Tb)); tf += t
print("   crp: %.1f GB/s of output (7.87 MB/pair)" % (K * 7.872e6 / (t * 1e-3) / 1e9))
_, t2 = timed("crp (sqrt)", lambda: engine.crp(corpus, batch, xp, True, out=Tb))
_, t2 = timed("crp (squared, VALU tile)", lambda: engine.crp(corpus, batch, xp, False, out=Tb, force_valu=True))
_, t2 = timed("crp (squared, MFMA tile)", lambda: engine.crp(corpus, batch, xp, False, out=Tb, force_tile=True))
engine.crp(corpus, batch, xp, False, out=Tb)
_, t = timed("binarize(T)", lambda: engine.binarize(Tb, batch, 0.095, True, out=Bout, work=work))
_, t = timed("thresholds(T)", lambda: engine.thresholds(Tb, batch, 0.095, True, work=work)); tf += t
_, t = timed("qmax fused", lambda: engine.align_fused("qmax", Tb, batch, work)); tf += t
_, t2 = timed("dmax fused", lambda: engine.align_fused("dmax", Tb, batch, work, boundary=1))
(bits, wb), t = timed("mask_bits(T)", lambda: engine.mask_bits(Tb, batch, 0.095, True)); tb = t
_, t = timed("qmax bits", lambda: engine.align_bits("qmax", bits, batch)); tb += t
_, t2 = timed("dmax bits", lambda: engine.align_bits("dmax", bits, batch, boundary=1))
print("bits chain (oti+pack+crp+mask_bits+qmax_bits): %.3f ms -> %.0f pair-scores/s" % (tf - 0 + 0.03, 0))
print("fast chain (oti+pack+crp+thresholds+qmax_fused): %.3f ms for %d pairs -> %.0f pair-scores/s" % (tf + 0.03, K, K / ((tf + 0.03) * 1e-3)))
# f32 csm
c32 = engine.DeviceCorpus(corpus_h.feats.astype(np.float32), corpus_h.frame_off, gchroma=corpus_h.gchroma)
_, t = timed("csm_f32", lambda: engine.csm(c32, batch))
print("   csm_f32: %.1f GB/s algorithmic (4.096 MB/pair)" % (K * 4.096e6 / (t * 1e-3) / 1e9))
