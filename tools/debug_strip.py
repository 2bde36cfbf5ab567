"""Where does a strip-kernel build differ from the float64 tile kernels? (dev tool)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "serra09_mini.npz"))
corpus = engine.DeviceCorpus(g["feats"], g["frame_off"], gchroma=g["gchroma"])
for align in (32, 16):
    batch = engine.PairBatch(corpus.frame_off, g["pairs"], 9, corpus.device, pitch_align=align)
    engine.oti(corpus, batch)
    xp = engine.pack_x(corpus, batch)
    ref = engine.crp(corpus, batch, xp, force_tile=True).cpu().numpy()       # MFMA tile kernel (not the strip kernel)
    got = engine.crp(corpus, batch, xp).cpu().numpy()                        # strip kernel, float64 out
    planes = engine.crp_planar(corpus, batch, xp).cpu().numpy().astype(np.int64) & 0xffffffff
    nbad = 0
    for p in range(batch.K):
        d = batch.descs[p]
        M, N = int(d["nx"]) - 8, int(d["ny"]) - 8
        idx = (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)
        a, b = ref[idx], got[idx]
        w = idx
        key = planes[w]
        bits = (a.view(np.uint64) | (1 << 63)) >> 32
        bad1 = np.argwhere(a != b)
        bad2 = np.argwhere(key.astype(np.uint64) != bits)
        if len(bad1) or len(bad2):
            nbad += 1
            if nbad <= 4 and len(bad2):
                i, j = bad2[0]
                print("   at (%d,%d): key got %016x want %016x ; neighbours want %016x %016x" % (i, j, int(key[i, j]), int(bits[i, j]), int(bits[i, j - 1]), int(bits[i, j + 1])))
            if nbad <= 4:
                print("align %d pair %d (M %d, N %d, off %d pitch %d): f64 strip mismatches %d %s | planar mismatches %d %s" % (
                    align, p, M, N, d["crp_off"], d["crp_pitch"], len(bad1), bad1[:6].tolist(), len(bad2), bad2[:6].tolist()))
    print("align %d: %d of %d pairs differ" % (align, nbad, batch.K))
