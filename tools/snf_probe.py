"""EarlySNF at full size (dev tool): pairs of 1000-frame songs, L = 1984."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ch = synth.make_corpus(4, 4, n_frames=1000, seed=20260)
rng = np.random.default_rng(0)
chroma = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
ss = [np.cumsum(rng.standard_normal((992, 64)), axis=0) * 0.1 for _ in range(ch.n_songs)]
ssms = engine.DeviceCorpus(np.concatenate(ss), np.arange(ch.n_songs + 1, dtype=np.int64) * 992)
allp = synth.all_pairs(ch.n_songs)
pairs = allp[np.arange(K) % len(allp)]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    res = engine.early_snf_scores(chroma, ssms, pairs)
    torch.cuda.synchronize(); t1 = time.time()
    print("early_snf_scores: %d pairs in %.3f s -> %.1f pairs/s (%.1f ms / pair; 12 products of 1984^3 = 187 GFLOP / pair -> %.1f TFLOP/s f64 overall)"
          % (K, t1 - t0, K / (t1 - t0), 1e3 * (t1 - t0) / K, 187.4e9 * K / (t1 - t0) / 1e12))
