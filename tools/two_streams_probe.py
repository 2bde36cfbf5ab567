"""Does the chain gain from two half-batches on two streams (kernel tails of one overlapping kernel heads of the other)?
One 4096-pair batch on one stream against 2 x 2048 on two.  usage: python tools/two_streams_probe.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
engine.float32_copy(corpus)


class Chain(object):
    def __init__(self, pairs):
        self.b = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
        self.band = engine.planar32_band(corpus, self.b)
        self.koff = engine.keys16_koff(corpus, self.b)
        self.xp = None; self.keys = None; self.bits = None; self.work = None; self.sc = None

    def run(self):
        b = self.b
        engine.oti(corpus, b)
        self.xp = engine.pack_x32(corpus, b, out=self.xp)
        self.keys = engine.crp_keys16(corpus, b, self.xp, self.koff, out=self.keys)
        self.bits, self.work = engine.mask_bits_keys16(self.keys, self.band, self.koff, self.xp, corpus, b, 0.095, out=self.bits, work=self.work)
        self.sc = engine.align_bits("qmax", self.bits, b)
        return self.sc


def wall(fn, reps=8):
    ts = []
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[2:]))


one = Chain(allp[:4096])
halves = [Chain(allp[:2048]), Chain(allp[2048:4096])]
quarters = [Chain(allp[1024 * q:1024 * (q + 1)]) for q in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]


def split(chains):
    for c, s in zip(chains, streams):
        with torch.cuda.stream(s):
            c.run()


ref = one.run().cpu().numpy()
split(halves); torch.cuda.synchronize()
got = np.concatenate([c.sc.cpu().numpy() for c in halves])
print("scores equal:", np.array_equal(ref, got))
print("one stream, 4096 pairs: %.3f ms" % wall(one.run))
print("one stream, 2 x 2048 back to back: %.3f ms" % wall(lambda: [c.run() for c in halves]))
print("two streams, 2 x 2048: %.3f ms" % wall(lambda: split(halves)))
print("four streams, 4 x 1024: %.3f ms" % wall(lambda: split(quarters)))
