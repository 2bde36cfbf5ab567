"""Evaluation at the DA-TACOS benchmark-subset shape (15000 songs = 1000 cliques x 13 + 2000 singletons): GPU ranks
vs the argsort form on the host (dev tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_plugin_cpu import _alg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
rng = np.random.default_rng(1)
labels = np.concatenate([np.repeat(np.arange(N // 15), 13), N + np.arange(N - 13 * (N // 15))])[rng.permutation(N)]
D = rng.random((N, N), dtype=np.float32)
D += np.float32(0.5) * (labels[:, None] == labels[None, :])
alg = _alg("/tmp", N)
alg.Ds = {"main": D}
for i, lab in enumerate(labels):
    alg.cliques.setdefault("c%d" % lab, set()).add(i)
import torch
from acoss_amd import engine
engine.require_gpu()
cl = [sorted(v) for v in alg.cliques.values()]
for rep in range(2):
    t0 = time.time(); alg._mate_ranks_device(D, cl); torch.cuda.synchronize(); t1 = time.time()
    print("GPU ranks incl. upload of %.0f MB: %.3f s" % (D.nbytes / 1e6, t1 - t0))
Dd = torch.from_numpy(D).cuda()
t0 = time.time(); g = alg.getEvalStatistics("main", verbose=False, write_csv=False, on_gpu=True); t1 = time.time()
print("getEvalStatistics(on_gpu=True): %.2f s  MAP %.6f" % (t1 - t0, g[3]))
if N <= 15000:
    t0 = time.time(); c = alg.getEvalStatistics("main", verbose=False, write_csv=False); t1 = time.time()
    print("getEvalStatistics (host argsort): %.2f s  MAP %.6f  equal: %s" % (t1 - t0, c[3], np.array_equal(np.array(list(g[:4]) + list(g[4])), np.array(list(c[:4]) + list(c[4])))))
