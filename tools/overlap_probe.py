"""Does running two batches' kernel chains on two streams beat running them back to back? (dev tool)
usage: python tools/overlap_probe.py [pairs per batch]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)


class Chain(object):
    def __init__(self, lo):
        self.batch = engine.PairBatch(corpus.frame_off, allp[lo:lo + K], 9, corpus.device, pitch_align=32)
        engine.oti(corpus, self.batch)
        self.band = engine.planar32_band(corpus, self.batch)
        self.xp = engine.pack_x32(corpus, self.batch)
        self.T = engine.crp_planar32(corpus, self.batch, self.xp)
        self.bits, self.work = engine.mask_bits_planar32(self.T, self.band, corpus, self.batch, 0.095)
        self.scores = engine.align_bits("qmax", self.bits, self.batch)

    def run(self):
        engine.pack_x32(corpus, self.batch, out=self.xp)
        engine.crp_planar32(corpus, self.batch, self.xp, out=self.T)
        engine.mask_bits_planar32(self.T, self.band, corpus, self.batch, 0.095, out=self.bits, work=self.work)
        engine.align_bits("qmax", self.bits, self.batch, scores=self.scores)


a, b = Chain(0), Chain(K)
ref_a, ref_b = a.scores.clone(), b.scores.clone()
torch.cuda.synchronize()
R = 6
for name in ("sequential", "two streams", "sequential", "two streams"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if name == "sequential":
        for r in range(R):
            a.run(); b.run()
    else:
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for r in range(R):
            with torch.cuda.stream(s1):
                a.run()
            with torch.cuda.stream(s2):
                b.run()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = bool(torch.equal(a.scores, ref_a) and torch.equal(b.scores, ref_b))
    print("%-12s %d x 2 x %d pairs in %.1f ms -> %.0f pair-scores/s  (scores unchanged: %s)" % (name, R, K, el * 1e3, 2 * R * K / el, ok), flush=True)
