"""A/B of library builds on the 16-bit-key selection of a FLOAT32 corpus (13-d random walks, the bench's MFCC stand-in), in one
process on the same buffers (dev tool).   usage: python tools/ab_k16_f32.py [frames, 0 = ragged] name=path.so ...
Per build: row selection alone, column selection alone, the whole mask_bits call; masks must equal the in-tree build's."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
args = sys.argv[1:]
frames = int(args.pop(0)) if args and args[0].isdigit() else 1000
libs = {"base": _lib.load()}
for a in args:
    n, pth = a.split("=", 1)
    libs[n] = ctypes.CDLL(os.path.abspath(pth))
FN = "acoss_mask_bits_keys16_f32_batch"
for lib in libs.values():
    f = getattr(lib, FN); f.restype = ctypes.c_int; f.argtypes = _lib.SIGNATURES[FN][1]
rng = np.random.default_rng(1)
songs = [np.cumsum(rng.standard_normal((frames or int(rng.integers(60, 1033)), 13)) * rng.uniform(0.2, 3.0), axis=0).astype(np.float32) for _ in range(220)]
feats = np.concatenate(songs); off = np.cumsum([0] + [len(s) for s in songs]).astype(np.int64)
corpus = engine.DeviceCorpus(feats, off)
batch = engine.PairBatch(corpus.frame_off, synth.all_pairs(220)[:4096], 9, corpus.device, pitch_align=32)
xp = engine.pack_x32(corpus, batch)
koff = engine.keys16_koff_f32(corpus, batch, xp); band = engine.keys16_band_f32(corpus, batch)
keys = engine.crp_keys16(corpus, batch, xp, koff)
bits, work = engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095)
ref = bits.clone()
f32, n32 = engine.float32_copy(corpus)
P = engine._ptr; st = engine._stream


def mask(lib, mutual):
    assert getattr(lib, FN)(P(keys), P(band), P(koff), P(xp), P(f32), P(n32), corpus.d, P(batch.descs_dev), batch.K, 9, batch.max_nx,
                            batch.max_ny, 0.095, mutual, P(bits), P(work), work.numel(), st()) == 0


res = {}
order = list(libs)
for rnd in range(7):
    for name in (order if rnd % 2 else order[::-1]):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record(); mask(libs[name], 2)
        ev[1].record(); mask(libs[name], 3)
        ev[2].record(); mask(libs[name], 1)
        ev[3].record(); torch.cuda.synchronize()
        assert torch.equal(bits, ref), "mask bits differ in build %s" % name
        if rnd: res.setdefault(name, []).append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
print("%-10s %8s %8s %10s   (ms per 4096 pairs of %s-frame 13-d float32 random walks, medians of 6)" % ("build", "rows", "cols", "mask_bits", frames or "60..1032"))
for name in order:
    print("%-10s %8.3f %8.3f %10.3f" % ((name,) + tuple(np.median(np.array(res[name]), axis=0))))
