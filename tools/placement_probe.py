"""Time of the float32 strip kernel and of the two selection kernels against the row pitch of the key matrix and the
placement of its buffer (dev tool; the rows / cols split needs `python -m acoss_amd.build --probes`).
usage: python tools/placement_probe.py [pairs]"""
import ctypes, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.make_corpus(16, 4, n_frames=1000, seed=20260)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
lib = _lib.load()
probe = getattr(lib, "acoss_dev_planar_probe", None)
if probe is not None:
    probe.restype = ctypes.c_int
    probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                      ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]


def timed(fn, n=6):
    ts = []
    for rnd in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if rnd: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def case(align, offset_bytes=0, label=""):
    batch = engine.PairBatch(corpus.frame_off, allp[np.arange(K) % len(allp)], 9, corpus.device, pitch_align=align)
    engine.oti(corpus, batch)
    xp32 = engine.pack_x32(corpus, batch)
    n = engine.planar_elems(batch)
    big = torch.empty(n + (8 << 20), dtype=torch.int32, device=corpus.device)
    out = big[offset_bytes // 4: offset_bytes // 4 + n]
    band = engine.planar32_band(corpus, batch)
    t_crp = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
    bits, work = engine.mask_bits_planar32(out, band, corpus, batch, 0.095)
    t_mask = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, out=bits, work=work))
    line = "%-22s pitch %4d  base %#x +%-8d crp32 %.3f (min %.3f)  mask_bits %.3f (min %.3f)" % (
        label or "align %d" % align, int(batch.descs["crp_pitch"][0]), big.data_ptr(), offset_bytes, t_crp[0], t_crp[1], t_mask[0], t_mask[1])
    if probe is not None:
        for m, nm in ((2, "rows"), (12, "cols")):
            t = timed(lambda: probe(m, engine._ptr(out), engine._ptr(corpus.feats), engine._ptr(corpus.norms), corpus.d, engine._ptr(batch.descs_dev), K, 9,
                                    1000, 1000, 0.095, engine._ptr(work), work.numel(), engine._stream()))
            line += "  %s %.3f" % (nm, t[0])
    print(line, flush=True)
    del big, out, bits, work
    engine.release_scratch()
    torch.cuda.empty_cache()


for align in (32, 64, 96, 160, 224):
    case(align)
for off in (0, 128, 256, 1024, 2048, 4096, 65536, 1 << 20, 3 << 20):
    case(32, off, "align 32 offset")
for i in range(4):
    hold = torch.empty((i + 1) * (37 << 20), dtype=torch.uint8, device=corpus.device)       # shifts where the next allocation lands
    case(32, 0, "align 32 fresh #%d" % i)
