"""A/B of two builds of the selection kernels in one process on the same buffers (dev tool): the in-tree library against
tools/ab/libacoss_old.so, acoss_mask_bits_planar32_batch on the float32 keys of 4096 pairs."""
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
keys = engine.crp_planar32(corpus, batch, engine.pack_x32(corpus, batch))
band = engine.planar32_band(corpus, batch)
bits, work = engine.mask_bits_planar32(keys, band, corpus, batch, 0.095)
ref = bits.clone()
new = _lib.load()
old = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libacoss_old.so"))
name_fn = "acoss_mask_bits_planar32_batch"
for lib in (new, old):
    fn = getattr(lib, name_fn)
    fn.restype = ctypes.c_int
    fn.argtypes = _lib.SIGNATURES[name_fn][1]


def run(lib):
    rc = getattr(lib, name_fn)(engine._ptr(keys), engine._ptr(band), engine._ptr(corpus.feats), engine._ptr(corpus.norms), corpus.d,
                               engine._ptr(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny, 0.095, 1, engine._ptr(bits),
                               engine._ptr(work), work.numel(), engine._stream())
    assert rc == 0


res = {}
for rnd in range(9):
    for name, lib in ((("new", new), ("old", old)) if rnd % 2 else (("old", old), ("new", new))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(lib); e1.record(); torch.cuda.synchronize()
        assert torch.equal(bits, ref)
        if rnd: res.setdefault(name, []).append(e0.elapsed_time(e1))
for k in sorted(res):
    print("mask_bits_planar32 %-4s median %.3f ms  min %.3f" % (k, np.median(res[k]), np.min(res[k])))
