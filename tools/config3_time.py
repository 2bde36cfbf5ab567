"""Kernel times of the one-call scorer on a ragged corpus (synth.config3's length distribution): run under
rocprofv3 --kernel-trace --stats.  usage: python tools/config3_time.py [pairs] [reps] [want]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
want = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("qmax", "dmax", "swc")
ch = synth.config3(n_cliques=133, singletons=271)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
rng = np.random.default_rng(1)
allp = synth.all_pairs(ch.n_songs)
pairs = allp[rng.permutation(len(allp))[:K]]
engine.serra09_scores(corpus, pairs, want=want)
torch.cuda.synchronize()
for r in range(reps):
    t0 = time.perf_counter()
    engine.serra09_scores(corpus, pairs, want=want)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%d pairs %s: %.1f ms  %.0f pairs/s" % (K, want, el * 1e3, K / el), flush=True)
lens = np.diff(ch.frame_off)
cells = ((lens[pairs[:, 0]] - 8) * (lens[pairs[:, 1]] - 8)).sum()
print("cells %.3e  (config2 pair = 9.84e5): %.2f ns / Mcell" % (cells, el * 1e9 / (cells / 1e6)))
