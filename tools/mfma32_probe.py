"""Is v_mfma_f32_16x16x4_f32's accumulation a k-ordered chain of round-to-nearest FMAs (as the float64 form is)?  Emulates
crp_strip32_kernel's arithmetic on the host in float32 (FMA = one rounding of the exact product-sum, emulated through
float64: the product of two float32 is exact there) and compares with the kernel's keys bit for bit (dev tool)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
lens = iter([200, 150, 173])
ch = synth.make_corpus(3, 1, seed=5, lengths=lambda r: next(lens))
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
pairs = np.array([(0, 1), (1, 2), (2, 0)], dtype=np.int32)
b = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
engine.oti(corpus, b)
keys = engine.crp_planar32(corpus, b, engine.pack_x32(corpus, b)).cpu().numpy().view(np.uint32)
f32, n32 = [t.cpu().numpy() for t in engine.float32_copy(corpus)]
shifts = b.descs_dev.cpu().numpy().view(engine.PAIR_DESC)["shift"] if hasattr(b, "descs_dev") else b.descs["shift"]
def fma32(a, bb, c):
    return (a.astype(np.float64) * bb.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
tot = diff = 0
for p in range(b.K):
    d = b.descs[p]
    nx, ny, sh = int(d["nx"]), int(d["ny"]), int(shifts[p])
    X = np.roll(f32[int(d["x_row0"]):int(d["x_row0"]) + nx], sh, axis=1)
    Y = f32[int(d["y_row0"]):int(d["y_row0"]) + ny]
    acc = np.zeros((nx, ny), dtype=np.float32)
    for bin_ in range(12):
        acc = fma32(X[:, bin_][:, None], Y[:, bin_][None, :], acc)
    nsum = (n32[int(d["x_row0"]):int(d["x_row0"]) + nx][:, None] + n32[int(d["y_row0"]):int(d["y_row0"]) + ny][None, :]).astype(np.float32)
    C = np.maximum(fma32(np.float32(-2.0) * np.ones_like(acc), acc, nsum), np.float32(0))
    M, N = nx - 8, ny - 8
    T = C[0:M, 0:N].copy()
    for k in range(1, 9):
        T = (T + C[k:k + M, k:k + N]).astype(np.float32)
    idx = (int(d["crp_off"]) + np.arange(M)[:, None] * int(d["crp_pitch"]) + np.arange(N)[None, :]).astype(np.int64)
    got = (keys[idx] & 0x7fffffff).astype(np.uint32)
    want = T.view(np.uint32)
    tot += M * N
    diff += int(np.sum(got != want))
    if np.any(got != want):
        dd = np.abs(got.astype(np.int64) - want.astype(np.int64))
        print("pair %d: %d of %d keys differ, max %d ulp" % (p, int(np.sum(got != want)), M * N, int(dd.max())))
print("%d of %d keys differ from the round-to-nearest FMA-chain emulation" % (diff, tot))
