"""Float32 wide-feature CSM (Serra09.py:187-192: 20 736-d scattering features) on the matrix cores: rate at
992 x 20736 x 992 per pair (dev tool).  usage: python tools/scatter_csm_probe.py [songs] [frames] [d]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine
engine.require_gpu()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
F = int(sys.argv[2]) if len(sys.argv) > 2 else 992
D = int(sys.argv[3]) if len(sys.argv) > 3 else 20736
rng = np.random.default_rng(5)
feats = rng.standard_normal((S * F, D), dtype=np.float32)
off = np.arange(S + 1, dtype=np.int64) * F
corpus = engine.DeviceCorpus(feats, off)
pairs = np.array([(i, j) for i in range(S) for j in range(S) if i < j], dtype=np.int32)
batch = engine.PairBatch(corpus.frame_off, pairs, 1, corpus.device)
out = engine.csm(corpus, batch)
ts = []
for rnd in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.csm(corpus, batch, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = float(np.median(ts[1:]))
flop = 2.0 * F * F * D * len(pairs)
print("%d pairs of %d x %d x %d: %.2f ms, %.1f TFLOP/s float32 (%.3f of the 157.3 peak), %.1f pairs/s" % (
    len(pairs), F, D, F, t, flop / t / 1e9, flop / t / 1e9 / 157.3, len(pairs) / t * 1e3))
# spot check against float64 on one row block
C = out.cpu().numpy()
d0 = batch.descs[0]
got = C[int(d0["csm_off"]):int(d0["csm_off"]) + F * int(d0["csm_pitch"])].reshape(F, -1)[:8, :F].astype(np.float64)
x = feats[pairs[0, 0] * F: pairs[0, 0] * F + 8].astype(np.float64)
y = feats[pairs[0, 1] * F: pairs[0, 1] * F + F].astype(np.float64)
exact = np.sqrt(np.maximum((x * x).sum(1)[:, None] + (y * y).sum(1)[None, :] - 2 * x.dot(y.T), 0))
print("max |csm - float64| on 8 rows: %.3g (values ~ %.1f)" % (np.max(np.abs(got - exact)), exact.mean()))
