"""EarlySNF on 32 pairs of 1000-frame songs (bench.py's early_snf workload), three timed calls (dev tool; run it under
rocprofv3 --pmc for the product kernel's matrix-core utilisation)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
odd = "odd" in sys.argv          # songs of 1000 and 999 frames alternate: M + N is odd for half of the pairs
lens = iter([1000, 999] * 8) if odd else None
ch = synth.make_corpus(4, 4, seed=20260, lengths=(lambda r: next(lens))) if odd else synth.make_corpus(4, 4, n_frames=1000, seed=20260)
rng = np.random.default_rng(0)
chroma = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
nfr = np.diff(ch.frame_off) - 8
ss = [np.cumsum(rng.standard_normal((int(n), 64)), axis=0) * 0.1 for n in nfr]
ssms = engine.DeviceCorpus(np.concatenate(ss), np.concatenate([[0], np.cumsum(nfr)]).astype(np.int64))
allp = synth.all_pairs(ch.n_songs)
K = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
pairs = allp[np.arange(K) % len(allp)]
engine.early_snf_scores(chroma, ssms, pairs[:4])
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    res = engine.early_snf_scores(chroma, ssms, pairs)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    gflop = 12.0 * 2.0 * 1984.0 ** 3 / 1e9
    print("%d pairs: %.1f ms = %.1f pairs/s, %.1f TFLOP/s end to end (%.3f of 78.6)" % (K, 1e3 * el, K / el, gflop * K / el / 1e3, gflop * K / el / 1e3 / 78.6), flush=True)
print("qmax[0] %.9f dmax[0] %.9f" % (res["qmax"][0], res["dmax"][0]))
