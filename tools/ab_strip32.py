"""A/B of two builds of the float32 strip kernel in one process on the same buffers (dev tool): in-tree library against
tools/ab/libacoss_old.so."""
import ctypes, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=200, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
f32, n32 = engine.float32_copy(corpus)
keys = engine.crp_planar32(corpus, batch, xp32)
ref = keys.clone()
import glob
libs = {"tree": _lib.load()}
for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libacoss_*.so"))):
    libs[os.path.basename(path)[9:-3]] = ctypes.CDLL(path)
name_fn = "acoss_crp_planar32_batch"
for lib in libs.values():
    fn = getattr(lib, name_fn)
    fn.restype = ctypes.c_int
    fn.argtypes = _lib.SIGNATURES[name_fn][1]


def run(lib):
    rc = getattr(lib, name_fn)(engine._ptr(xp32), engine._ptr(f32), engine._ptr(n32), corpus.d, engine._ptr(batch.descs_dev), batch.K, 9,
                               batch.max_nx, batch.max_ny, engine._ptr(keys), engine._stream())
    assert rc == 0


res = {}
order = list(libs.items())
for rnd in range(13):
    order = order[1:] + order[:1]
    for name, lib in order:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(lib); e1.record(); torch.cuda.synchronize()
        if rnd: res.setdefault(name, []).append(e0.elapsed_time(e1))
    assert torch.equal(keys, ref)
for k in sorted(res):
    print("crp_planar32 %-6s median %.3f ms  min %.3f" % (k, np.median(res[k]), np.min(res[k])))
