"""A/B of library builds on get_csm's row-band kernel (acoss_csm_rows_batch_f64), one process, same buffers (dev tool, round 4).
usage: python tools/ab_csm.py [pairs] name=path.so ...    (probe builds return wrong values: timing only)"""
import ctypes, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
args = sys.argv[1:]
K = int(args.pop(0)) if args and args[0].isdigit() else 4096
libs = {"base": _lib.load()}
for a in args:
    n, pth = a.split("=", 1)
    libs[n] = ctypes.CDLL(os.path.abspath(pth))
fn = "acoss_csm_rows_batch_f64"
for lib in libs.values():
    f = getattr(lib, fn)
    f.restype = ctypes.c_int
    f.argtypes = _lib.SIGNATURES[fn][1]
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
batch = engine.PairBatch(corpus.frame_off, synth.all_pairs(ch.n_songs)[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x(corpus, batch)
C = torch.empty(batch.total_csm, dtype=torch.float64, device=corpus.device)
nx, ny = batch.descs["nx"].astype(np.float64), batch.descs["ny"].astype(np.float64)
algo = float(np.sum(8.0 * (nx * ny + corpus.d * (nx + ny))))
P = engine._ptr
res = {}
order = list(libs)
for rnd in range(7):
    for name in (order if rnd % 2 else order[::-1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert getattr(libs[name], fn)(P(xp), P(corpus.feats), P(corpus.norms), corpus.d, P(batch.descs_dev), batch.K, batch.max_nx, batch.max_ny, P(C), engine._stream()) == 0
        e1.record(); torch.cuda.synchronize()
        if rnd:
            res.setdefault(name, []).append(e0.elapsed_time(e1))
for name in order:
    ms = float(np.median(res[name]))
    print("%-10s %.3f ms  %.2f TB/s  %.3f of 8 TB/s" % (name, ms, algo / ms / 1e9, algo / ms / 1e9 / 8.0))
