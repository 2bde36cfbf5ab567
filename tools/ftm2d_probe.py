"""FTM2D at scale (dev tool): shingles of 1000 songs x ~400 beats, all-pairs product at N songs."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine
engine.require_gpu()
rng = np.random.default_rng(0)
bts = [rng.random((12, int(rng.integers(300, 500)))) for _ in range(1000)]
for rep in range(2):
    t0 = time.time(); S = engine.ftm2d_shingles(bts); torch.cuda.synchronize(); t1 = time.time()
    print("shingles of 1000 songs (%d windows): %.3f s incl. host packing + upload" % (sum(b.shape[1] - 74 for b in bts), t1 - t0))
for N in (1000, 15000):
    X = torch.from_numpy(rng.random((N, 900))).cuda()
    X /= X.norm(dim=1, keepdim=True)
    engine.ftm2d_gram(X); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); G = engine.ftm2d_gram(X); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("gram N=%5d: %.3f ms  %.1f TFLOP/s f64 (2*N*N*900)  %.1f M pair-similarities/s" % (N, ms, 2.0 * N * N * 900 / ms / 1e9, N * N / ms / 1e3))
