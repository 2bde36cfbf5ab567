"""One-call scorer (acoss_serra09_scores) rate by call size and pair order (dev tool)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs).astype(np.int32)
perm = np.random.default_rng(0).permutation(len(allp))
for K in (131072, 32768, 4096 * 4):
    for name, sel in (("in order", allp[:K]), ("shuffled", allp[perm[:K]])):
        sel = np.ascontiguousarray(sel)
        engine.serra09_scores(corpus, sel, want=("qmax",))
        torch.cuda.synchronize()
        ts = []
        for rnd in range(3):
            t0 = time.perf_counter()
            engine.serra09_scores(corpus, sel, want=("qmax",))
            ts.append(time.perf_counter() - t0)
        print("%7d pairs %-9s: %.1f ms -> %.0f pair-scores/s" % (K, name, min(ts) * 1e3, K / min(ts)), flush=True)
