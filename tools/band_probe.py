"""Float32 keys: how many rows / columns have another value inside the error band of their k-th smallest (dev tool)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
ch = synth.config2(n_songs=64, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
pairs = synth.all_pairs(ch.n_songs)[:16]
b = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
engine.oti(corpus, b)
keys = engine.crp_planar32(corpus, b, engine.pack_x32(corpus, b)).cpu().numpy().view(np.uint32)
band2 = engine.planar32_band(corpus, b).cpu().numpy().astype(np.float64).reshape(-1, 2)
band = band2[:, 0] + band2[:, 1] * 16.5 * corpus._f32_scale2          # at a typical threshold value
T = engine.crp(corpus, b, engine.pack_x(corpus, b)).cpu().numpy() * corpus._f32_scale2
print("band (2 x bound) per pair: min %.3g max %.3g; song wmax mean %.3g" % (band.min(), band.max(), corpus.song_wmax(9).mean()))
for scale in (1.0, 0.25, 1 / 16.0):
    tot_r = amb_r = tot_c = amb_c = 0
    worst = 0.0
    for p in range(b.K):
        d = b.descs[p]
        M, N = int(d["nx"]) - 8, int(d["ny"]) - 8
        idx = (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)
        A = (keys[idx] & 0x7fffffff).astype(np.uint32).view(np.float32).astype(np.float64)
        worst = max(worst, np.max(np.abs(A - T[idx])) / (band[p] / 2))
        for X, k, name in ((A, int(np.round(0.095 * N)), "r"), (A.T, int(np.round(0.095 * M)), "c")):
            thr = np.partition(X, k - 1, axis=1)[:, k - 1]
            cnt = np.sum(np.abs(X - thr[:, None]) <= band[p] * scale, axis=1)
            if name == "r":
                tot_r += len(cnt); amb_r += int(np.sum(cnt > 1))
            else:
                tot_c += len(cnt); amb_c += int(np.sum(cnt > 1))
    print("band x %.4f: rows with company in the band %.2f %%, columns %.2f %%   (T at threshold ~ %.3g; max error / bound %.3f)"
          % (scale, 100.0 * amb_r / tot_r, 100.0 * amb_c / tot_c, float(np.median(thr)), worst))
