"""The float32 wide-feature CSM (20 736-d, csm_gemm32_kernel) alone: 28 pairs of 992 x 20736 x 992 (bench.py's scatter_csm workload), for A/B runs of
timing-only builds (-DGM32_PROBE=1/2/3: ACOSS_LIB_PATH) (dev tool, round 5)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine
rng = np.random.default_rng(5)
S, F, D = 8, 992, 20736
feats = rng.standard_normal((S * F, D), dtype=np.float32)
corpus = engine.DeviceCorpus(feats, np.arange(S + 1, dtype=np.int64) * F)
pairs = np.array([(i, j) for i in range(S) for j in range(S) if i < j], dtype=np.int32)
batch = engine.PairBatch(corpus.frame_off, pairs, 1, corpus.device)
out = engine.csm(corpus, batch)
ms = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.csm(corpus, batch, out=out); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
m = float(np.median(ms[1:]))
flop = 2.0 * F * F * D * len(pairs)
print("%d pairs: %.3f ms = %.1f TFLOP/s (%.3f of 157.3); checksum %.6e" % (len(pairs), m, flop / m / 1e9, flop / m / 1e9 / 157.3, float(out[:1000].double().sum())))
