"""A/B of builds on the whole kernel chain of one batch, same buffers, one process (dev tool): the in-tree library against
every tools/ab/libacoss_*.so; per-stage HIP-event times (the stages run back to back as in the product path, so cache
state between kernels is the product's)."""
import ctypes, glob, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
f32, n32 = engine.float32_copy(corpus)
keys = engine.crp_planar32(corpus, batch, xp32)
band = engine.planar32_band(corpus, batch)
bits, work = engine.mask_bits_planar32(keys, band, corpus, batch, 0.095)
scores = engine.align_bits("qmax", bits, batch).clone()
ref = scores.clone()
libs = {"tree": _lib.load()}
for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libacoss_*.so"))):
    libs[os.path.basename(path)[9:-3]] = ctypes.CDLL(path)
names = ("acoss_pack_x_f32", "acoss_crp_planar32_batch", "acoss_mask_bits_planar32_batch", "acoss_align_bits_batch")
for lib in libs.values():
    for n in names:
        fn = getattr(lib, n)
        fn.restype = ctypes.c_int
        fn.argtypes = _lib.SIGNATURES[n][1]
P = engine._ptr


def chain(lib, ev):
    st = engine._stream()
    ev[0].record()
    assert lib.acoss_pack_x_f32(P(f32), P(n32), corpus.d, P(batch.descs_dev), batch.K, batch.max_nx, P(xp32), st) == 0
    assert lib.acoss_crp_planar32_batch(P(xp32), P(f32), P(n32), corpus.d, P(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny, P(keys), st) == 0
    ev[1].record()
    assert lib.acoss_mask_bits_planar32_batch(P(keys), P(band), P(corpus.feats), P(corpus.norms), corpus.d, P(batch.descs_dev), batch.K, 9,
                                              batch.max_nx, batch.max_ny, 0.095, 1, P(bits), P(work), work.numel(), st) == 0
    ev[2].record()
    assert lib.acoss_align_bits_batch(0, P(bits), P(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny, 0, None, P(scores), st) == 0
    ev[3].record()


res = {}
order = list(libs.items())
for rnd in range(9):
    order = order[1:] + order[:1]
    for name, lib in order:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for rep in range(3):            # three chains back to back, the last one timed: steady state of this build
            chain(lib, ev)
        torch.cuda.synchronize()
        assert torch.equal(scores, ref)
        if rnd:
            res.setdefault(name, []).append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)] + [ev[0].elapsed_time(ev[3])])
for k in sorted(res):
    m = np.median(np.array(res[k]), axis=0)
    print("%-8s pack+crp %.3f  mask_bits %.3f  qmax %.3f  chain %.3f ms" % (k, m[0], m[1], m[2], m[3]))
