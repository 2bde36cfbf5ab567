"""End-to-end throughput through the plugin class (dev tool): Serra09.similarity on a synthetic config-2 corpus."""
import os, sys, time, warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
from acoss_amd.Serra09 import Serra09
warnings.simplefilter("ignore")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
corpus = synth.config2(n_songs=1000, n_frames=1000)
alg = Serra09(corpus, shortname="probe", do_memmaps=False, cachedir="/tmp/acoss_probe_cache")
pairs = synth.all_pairs(corpus.n_songs)
rng = np.random.default_rng(0)
sel = pairs[rng.permutation(len(pairs))[:K]].astype(np.int64)
alg.similarity(sel[:4096])          # warm: feature upload, allocations
torch.cuda.synchronize()
t0 = time.time()
res = alg.similarity(sel)
torch.cuda.synchronize()
t1 = time.time()
print("Serra09.similarity: %d pairs (chroma qmax + dmax) in %.2f s -> %.0f pairs/s" % (K, t1 - t0, K / (t1 - t0)))
t0 = time.time()
out = engine.serra09_scores(alg._device_corpus('chroma', list(range(1000)))[0], sel.astype(np.int32), want=("qmax",))
torch.cuda.synchronize()
t1 = time.time()
print("engine.serra09_scores (qmax only): %.2f s -> %.0f pairs/s" % (t1 - t0, K / (t1 - t0)))
