"""CSM kernel probes: where does the time go? (dev tool)"""
import ctypes, sys
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ch = synth.make_corpus(16, 4, n_frames=1000, seed=20260)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[np.arange(K) % len(allp)], 9, corpus.device)
engine.oti(corpus, batch)
C = torch.empty(batch.total_csm, dtype=torch.float64, device=corpus.device)
lib = _lib.load()
if not hasattr(lib, "acoss_dev_csm_probe"):
    raise SystemExit("development probes are not in this build: python -m acoss_amd.build --probes")
fn = lib.acoss_dev_csm_probe
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
names = {0: "normal", 1: "stores only", 2: "arith only", 3: "shift=0 contiguous x"}
res = {m: [] for m in names}
for rnd in range(6):
    for m in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(m, engine._ptr(corpus.feats), engine._ptr(corpus.norms), engine._ptr(batch.descs_dev), K, 1000, 1000, engine._ptr(C), engine._stream())
        e1.record(); torch.cuda.synchronize()
        if rnd: res[m].append(e0.elapsed_time(e1))
for m in names:
    t = np.array(res[m])
    print("mode %d %-22s median %.3f ms  min %.3f  -> %.0f GB/s" % (m, names[m], np.median(t), t.min(), K * 8.192e6 / np.median(t) / 1e6))
# plain fill for reference
x = torch.empty(batch.total_csm, dtype=torch.float64, device=corpus.device)
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); x.fill_(1.5); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("torch fill_ %.1f GB: median %.3f ms -> %.0f GB/s" % (x.numel() * 8 / 1e9, np.median(ts), x.numel() * 8 / np.median(ts) / 1e6))
y = torch.empty_like(x); ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y.copy_(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("torch copy_ (read+write %.1f GB): median %.3f ms -> %.0f GB/s" % (2 * x.numel() * 8 / 1e9, np.median(ts), 2 * x.numel() * 8 / np.median(ts) / 1e6))
