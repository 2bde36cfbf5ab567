"""planar selection probes (dev tool)."""
import ctypes, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ch = synth.make_corpus(16, 4, n_frames=1000, seed=20260)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[np.arange(K) % len(allp)], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x(corpus, batch)
planes = engine.crp_planar(corpus, batch, xp)
lib = _lib.load()
work = torch.empty(int(lib.acoss_mask_bits_work_bytes(K, 1000, 1000, 9)), dtype=torch.uint8, device=corpus.device)
if not hasattr(lib, "acoss_dev_planar_probe"):
    raise SystemExit("development probes are not in this build: python -m acoss_amd.build --probes")
fn = lib.acoss_dev_planar_probe
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
               ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
names = {1: "rows loads only", 2: "rows select", 3: "rows select, made-up keys (no loads)", 11: "cols loads only", 13: "cols w/o selection", 12: "cols select"}
res = {m: [] for m in names}
for rnd in range(5):
    for m in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(m, engine._ptr(planes), engine._ptr(corpus.feats), engine._ptr(corpus.norms), corpus.d, engine._ptr(batch.descs_dev), K, 9, 1000, 1000, 0.095, engine._ptr(work), work.numel(), engine._stream())
        assert rc == 0, rc
        e1.record(); torch.cuda.synchronize()
        if rnd: res[m].append(e0.elapsed_time(e1))
for m in names:
    t = np.array(res[m]); print("mode %2d %-18s median %.3f ms" % (m, names[m], np.median(t)))
for name, fnc in (("crp f64", lambda: engine.crp(corpus, batch, xp)), ("crp planar", lambda: engine.crp_planar(corpus, batch, xp, out=planes))):
    ts = []
    for rnd in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fnc(); e1.record(); torch.cuda.synchronize()
        if rnd: ts.append(e0.elapsed_time(e1))
    print("%-12s median %.3f ms" % (name, np.median(ts)))

Tf = torch.empty(batch.total_crp + 32, dtype=torch.float64, device=corpus.device)
for rnd in range(6):
    engine.crp(corpus, batch, xp, out=Tf)
    engine.crp_planar(corpus, batch, xp, out=planes)
torch.cuda.synchronize()
