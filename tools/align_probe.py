"""Strip-kernel time vs the placement of its output buffer (dev tool)."""
import ctypes, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
K = 2048
ch = synth.make_corpus(16, 4, n_frames=1000, seed=20260)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[np.arange(K) % len(allp)], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x(corpus, batch)
n = batch.total_crp
big = torch.empty(2 * n + (64 << 20), dtype=torch.int32, device=corpus.device)
print("base address %#x" % big.data_ptr())
offs = [0, 64, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 3 << 20, 1 << 22, 1 << 24]
res = {o: [] for o in offs}
for rnd in range(5):
    for o in offs:
        out = big[o // 4: o // 4 + 2 * n]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.crp_planar(corpus, batch, xp, out=out); e1.record(); torch.cuda.synchronize()
        if rnd: res[o].append(e0.elapsed_time(e1))
for o in offs:
    print("offset %9d B: median %.3f ms" % (o, np.median(res[o])))
# several fresh allocations
for i in range(6):
    buf = torch.empty(2 * n, dtype=torch.int32, device=corpus.device)
    ts = []
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.crp_planar(corpus, batch, xp, out=buf); e1.record(); torch.cuda.synchronize()
        if rnd: ts.append(e0.elapsed_time(e1))
    print("alloc %d at %#x: median %.3f ms" % (i, buf.data_ptr(), np.median(ts)))
    keep = buf if i % 2 == 0 else None
