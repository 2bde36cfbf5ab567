"""A/B of two builds of the strip kernel in one process (dev tool): the in-tree library vs tools/ab/libacoss_old.so."""
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
new = _lib.load()
old = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libacoss_old.so"))
outs = {}
for name, lib in (("new", new), ("old", old)):
    for fn_name in ("acoss_crp_planar_batch_f64", "acoss_crp_batch_f64"):
        fn = getattr(lib, fn_name)
        fn.restype = ctypes.c_int
        fn.argtypes = _lib.SIGNATURES[fn_name][1]
one = torch.empty(engine.planar_elems(batch) + 2 * batch.total_crp, dtype=torch.int32, device=corpus.device)
bufs = {"new": one, "old": one}      # same output buffer: kernel time depends on the allocation
def run(lib, which, out):
    if which == "planar":
        rc = lib.acoss_crp_planar_batch_f64(engine._ptr(xp), engine._ptr(corpus.feats), engine._ptr(corpus.norms), corpus.d,
                                            engine._ptr(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny, engine._ptr(out), engine._stream())
    else:
        rc = lib.acoss_crp_batch_f64(engine._ptr(xp), engine._ptr(corpus.feats), engine._ptr(corpus.norms), corpus.d,
                                     engine._ptr(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny, 0, engine._ptr(out), engine._stream())
    assert rc == 0
res = {}
for rnd in range(9):
    for which in ("planar", "f64"):
        for name, lib in ((("new", new), ("old", old)) if rnd % 2 else (("old", old), ("new", new))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(lib, which, bufs[name]); e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault((which, name), []).append(e0.elapsed_time(e1))
for k in sorted(res):
    print("%-8s %-4s median %.3f ms  min %.3f" % (k[0], k[1], np.median(res[k]), np.min(res[k])))
