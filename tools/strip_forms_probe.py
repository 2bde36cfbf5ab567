"""Compute-only (no stores) against full times of the two strip kernel forms (probes build).  usage: python tools/strip_forms_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
batch = engine.PairBatch(corpus.frame_off, synth.all_pairs(ch.n_songs)[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
out = torch.empty(engine.planar_elems(batch) + 1024, dtype=torch.int32, device=corpus.device)
def timed(fn, reps=6):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for form in ("cols", "rows"):
    os.environ["ACOSS_STRIP32_FORM"] = form
    os.environ.pop("ACOSS_STRIP32_NOSTORE", None)
    t_full = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
    os.environ["ACOSS_STRIP32_NOSTORE"] = "1"
    t_ns = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
    print("%s: full %.3f ms, without stores %.3f ms" % (form, t_full, t_ns), flush=True)
