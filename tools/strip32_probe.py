"""Float32 strip kernel (dev tool): error against the float64 windowed sums relative to the documented bound, and time."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.make_corpus(16, 4, n_frames=1000, seed=20260)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
# accuracy on a few ragged pairs
lens = iter([9, 40, 65, 129, 300, 1032, 1000, 777])
rag = synth.make_corpus(4, 2, seed=83, lengths=lambda r: next(lens))
rc = engine.DeviceCorpus(rag.feats, rag.frame_off, gchroma=rag.gchroma)
pairs = np.array([(i, j) for i in range(8) for j in range(8)], dtype=np.int32)
for align in (() if os.environ.get("ACOSS_STRIP32_NOSTORE") else (32, 1)):
    b = engine.PairBatch(rc.frame_off, pairs, 9, rc.device, pitch_align=align)
    engine.oti(rc, b)
    T = engine.crp(rc, b, engine.pack_x(rc, b)).cpu().numpy()
    A = engine.crp_planar32(rc, b, engine.pack_x32(rc, b)).cpu().numpy().view(np.uint32)
    worst = 0.0
    bnd = engine.planar32_band(rc, b).cpu().numpy().astype(np.float64).reshape(-1, 2) / 2      # the bound per pair: base + slope * value
    for p in range(b.K):
        d = b.descs[p]
        Mm, Nn = int(d["nx"]) - 8, int(d["ny"]) - 8
        idx = (int(d["crp_off"]) + np.arange(Mm)[:, None] * int(d["crp_pitch"]) + np.arange(Nn)[None, :]).astype(np.int64)
        approx = (A[idx] & 0x7fffffff).astype(np.uint32).view(np.float32).astype(np.float64)
        assert np.all(A[idx] >> 31 == 1)
        Ts = T[idx] * rc._f32_scale2
        worst = max(worst, float(np.max(np.abs(approx - Ts) / (bnd[p, 0] + bnd[p, 1] * Ts))))
    print("pitch_align %d: max |approx - exact| / bound = %.4f over %d pairs" % (align, worst, b.K))
batch = engine.PairBatch(corpus.frame_off, allp[np.arange(K) % len(allp)], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
out = torch.empty(engine.planar_elems(batch), dtype=torch.int32, device=corpus.device)
res = {"f64": [], "f32": [], "pack32": []}
for rnd in range(7):
    for name, fn in (("f64", lambda: engine.crp_planar(corpus, batch, xp, out=out)), ("f32", lambda: engine.crp_planar32(corpus, batch, xp32, out=out)),
                     ("pack32", lambda: engine.pack_x32(corpus, batch, out=xp32))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if rnd: res[name].append(e0.elapsed_time(e1))
for k, v in res.items():
    print("%-7s median %.3f ms (K = %d)" % (k, np.median(v), K))
