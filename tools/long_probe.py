"""Throughput of the chain on songs longer than 1032 frames (32-values-per-lane kernels up to 2056 frames, any-size kernels beyond) (dev tool)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ch = synth.make_corpus(8, 4, n_frames=n, seed=3)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
pairs = allp[np.arange(K) % len(allp)]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    res = engine.serra09_scores(corpus, pairs, want=("qmax",))
    torch.cuda.synchronize(); t1 = time.time()
    print("%d-frame songs: %d pairs in %.3f s -> %.0f pairs/s (cells/s %.3g)" % (n, K, t1 - t0, K / (t1 - t0), K * (n - 8.0) ** 2 / (t1 - t0)))
