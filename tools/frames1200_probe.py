"""1200-frame songs through the scorer: the long form of the radix selection (round 5) against the float64 planar keys it
replaces for this size class (ACOSS_RADIX16=0).  usage: python tools/frames1200_probe.py [frames] [pairs] [ragged]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
if "ragged" in sys.argv:          # lengths across the size class (and a few below it: mixed batches)
    ch = synth.make_corpus(40, 4, seed=frames, lengths=lambda r: int(r.integers(900, 2057)))
else:
    ch = synth.config2(n_songs=256, n_frames=frames)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
sel = allp[np.random.default_rng(2).permutation(len(allp))[:P]]
res = {}
for flag in ("1", "0"):
    os.environ["ACOSS_RADIX16"] = flag
    engine.serra09_scores(corpus, sel[:4096])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res[flag] = engine.serra09_scores(corpus, sel)
    el = time.perf_counter() - t0
    print("ACOSS_RADIX16=%s: %d pairs of %d frames, qmax + dmax: %.3f s = %.0f pair-scores/s" % (flag, len(sel), frames, el, 2 * len(sel) / el), flush=True)
print("identical:", all(np.array_equal(res["1"][k], res["0"][k]) for k in res["1"]))
