"""select_rows probes (dev tool)."""
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
T = engine.crp(corpus, batch, xp)
lib = _lib.load()
work = torch.empty(int(lib.acoss_binarize_work_bytes(K, 1000, 1000, 9)), dtype=torch.uint8, device=corpus.device)
if not hasattr(lib, "acoss_dev_select_probe"):
    raise SystemExit("development probes are not in this build: python -m acoss_amd.build --probes")
fn = lib.acoss_dev_select_probe
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
names = {0: "rows", 1: "rows loads only", 2: "rows select only", 3: "rows sorted/probing", 10: "cols", 11: "cols loads only", 13: "cols sorted/probing"}
res = {m: [] for m in names}
for rnd in range(5):
    for m in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(m, engine._ptr(T), engine._ptr(batch.descs_dev), K, 9, 1000, 1000, 0.095, engine._ptr(work), engine._stream())
        e1.record(); torch.cuda.synchronize()
        if rnd: res[m].append(e0.elapsed_time(e1))
for m in names:
    t = np.array(res[m]); print("mode %2d %-22s median %.3f ms" % (m, names[m], np.median(t)))

fn(0, engine._ptr(T), engine._ptr(batch.descs_dev), K, 9, 1000, 1000, 0.095, engine._ptr(work), engine._stream())
torch.cuda.synchronize()
mm = 992
off = K * mm * 8 * 2
cut = work[off:off + K * mm * 4].view(torch.int32).cpu().numpy()
vals, cnt = np.unique(cut, return_counts=True)
for v, c in zip(vals, cnt):
    print("cut %11d (0x%08x): %9d  %.3f%%" % (int(v), int(v) & 0xffffffff, c, 100.0 * c / cut.size))

fn(10, engine._ptr(T), engine._ptr(batch.descs_dev), K, 9, 1000, 1000, 0.095, engine._ptr(work), engine._stream())
torch.cuda.synchronize()
off2 = off + K * mm * 4
cut = work[off2:off2 + K * mm * 4].view(torch.int32).cpu().numpy()
vals, cnt = np.unique(cut, return_counts=True)
for v, c in zip(vals, cnt):
    print("col cut %11d (0x%08x): %9d  %.3f%%" % (int(v), int(v) & 0xffffffff, c, 100.0 * c / cut.size))
