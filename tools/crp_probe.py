"""crp strip kernel probes (dev tool)."""
import ctypes, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ch = synth.make_corpus(16, 4, n_frames=1000, seed=20260)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[np.arange(K) % len(allp)], 9, corpus.device)
engine.oti(corpus, batch)
xp = engine.pack_x(corpus, batch)
T = torch.empty(batch.total_crp, dtype=torch.float64, device=corpus.device)
lib = _lib.load()
if not hasattr(lib, "acoss_dev_crp_probe"):
    raise SystemExit("development probes are not in this build: python -m acoss_amd.build --probes")
fn = lib.acoss_dev_crp_probe
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
names = {0: "normal", 1: "no stores", 2: "no window sums", 3: "no MFMA phase", 4: "no stores, no sums", 5: "no stores, no MFMA"}
res = {m: [] for m in names}
for rnd in range(5):
    for m in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(m, engine._ptr(xp), engine._ptr(corpus.feats), engine._ptr(corpus.norms), engine._ptr(batch.descs_dev), K, 1000, 1000, engine._ptr(T), engine._stream())
        e1.record(); torch.cuda.synchronize()
        if rnd: res[m].append(e0.elapsed_time(e1))
for m in names:
    t = np.array(res[m]); print("mode %d %-16s median %.3f ms" % (m, names[m], np.median(t)))
