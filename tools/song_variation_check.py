"""engine.song_variation (segmented sums on the device) against numpy per song (dev tool, round 5)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from acoss_amd import engine
rng = np.random.default_rng(0)
lens = rng.integers(9, 400, size=300)
feats = np.concatenate([rng.standard_normal((n, 13)).astype(np.float32) * rng.uniform(0.1, 5) for n in lens])
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
c = engine.DeviceCorpus(feats, off)
v = engine.song_variation(c)
ref = np.array([((feats[off[i]:off[i+1]].astype(np.float64) - feats[off[i]:off[i+1]].astype(np.float64).mean(0)) ** 2).sum(1).mean() for i in range(len(lens))])
print("max rel err", np.max(np.abs(v - ref) / ref))
