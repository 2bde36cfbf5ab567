"""A/B of library builds on the 16-bit-key chain of one batch, in one process on the same buffers (dev tool, round 4).
usage: python tools/ab_k16.py [pairs] name=path.so ...     (the in-tree library is always "base")
Per build: strip kernel, row selection alone, column selection alone, the whole mask_bits call, qmax, dmax; masks and scores
must equal the in-tree build's."""
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
NAMES = ["acoss_crp_keys16_batch", "acoss_mask_bits_keys16_batch", "acoss_align_bits_batch", "acoss_align_bits_qd_batch"]
for lib in libs.values():
    for fn in NAMES:
        f = getattr(lib, fn)
        f.restype = ctypes.c_int
        f.argtypes = _lib.SIGNATURES[fn][1]
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
f32, n32 = engine.float32_copy(corpus)
xp = engine.pack_x32(corpus, batch)
koff = engine.keys16_koff(corpus, batch)
band = engine.planar32_band(corpus, batch)
keys = engine.crp_keys16(corpus, batch, xp, koff)
bits, work = engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095)
ref_bits = bits.clone()
sq = engine.align_bits("qmax", bits, batch).clone()
sd = engine.align_bits("dmax", bits, batch, boundary=1).clone()
ref_q, ref_d = sq.clone(), sd.clone()
P = engine._ptr
st = engine._stream


def strip(lib):
    assert lib.acoss_crp_keys16_batch(P(xp), P(f32), P(n32), corpus.d, P(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny,
                                      P(koff), P(keys), st()) == 0


def mask(lib, mutual):
    assert lib.acoss_mask_bits_keys16_batch(P(keys), P(band), P(koff), P(xp), P(f32), P(n32), P(corpus.feats), P(corpus.norms),
                                            corpus.d, P(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny, 0.095, mutual, P(bits),
                                            P(work), work.numel(), st()) == 0


def align(lib, kind, out, boundary):
    assert lib.acoss_align_bits_batch(kind, P(bits), P(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny, boundary, None, P(out), st()) == 0


def align_qd(lib, oq, od):
    assert lib.acoss_align_bits_qd_batch(P(bits), P(batch.descs_dev), batch.K, 9, batch.max_nx, batch.max_ny, 1, None, P(oq), P(od), st()) == 0


sq2, sd2 = sq.clone(), sd.clone()
res = {}
order = list(libs)
for rnd in range(7):
    for name in (order if rnd % 2 else order[::-1]):
        lib = libs[name]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        ev[0].record(); strip(lib)
        ev[1].record(); mask(lib, 2)
        ev[2].record(); mask(lib, 3)
        ev[3].record(); mask(lib, 1)
        ev[4].record(); align(lib, 0, sq, 0)
        ev[5].record(); align(lib, 1, sd, 1)
        ev[6].record(); align_qd(lib, sq2, sd2)
        ev[7].record()
        torch.cuda.synchronize()
        assert torch.equal(sq2, ref_q) and torch.equal(sd2, ref_d), "one-sweep scores differ in build %s" % name
        assert torch.equal(bits, ref_bits), "mask bits differ in build %s" % name
        assert torch.equal(sq, ref_q) and torch.equal(sd, ref_d), "scores differ in build %s" % name
        if rnd:
            res.setdefault(name, []).append([ev[i].elapsed_time(ev[i + 1]) for i in range(7)])
print("%-12s %8s %8s %8s %10s %8s %8s %8s   (ms per %d pairs, medians of 6)" % ("build", "strip", "rows", "cols", "mask_bits", "qmax", "dmax", "q+d call", K))
for name in order:
    m = np.median(np.array(res[name]), axis=0)
    print("%-12s %8.3f %8.3f %8.3f %10.3f %8.3f %8.3f %8.3f" % ((name,) + tuple(m)))
