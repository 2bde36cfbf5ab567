"""A/A and offset reproducibility of the strip / selection kernels inside ONE allocation (dev tool, --probes build)."""
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
probe = lib.acoss_dev_planar_probe
probe.restype = ctypes.c_int
probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                  ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]


def timed(fn, n=6):
    ts = []
    for rnd in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if rnd: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


batch = engine.PairBatch(corpus.frame_off, allp[np.arange(K) % len(allp)], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
n = engine.planar_elems(batch)
big = torch.empty(n + (8 << 20), dtype=torch.int32, device=corpus.device)
band = engine.planar32_band(corpus, batch)
bits, work = engine.mask_bits_planar32(big[:n], band, corpus, batch, 0.095)
print("base %#x  bits %#x  work %#x  xp32 %#x" % (big.data_ptr(), bits.data_ptr(), work.data_ptr(), xp32.data_ptr()))
offs = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 65536, 1 << 20, 2 << 20, 3 << 20]
for rnd in range(3):
    for off in offs:
        out = big[off // 4: off // 4 + n]
        t_crp = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
        t = [timed(lambda: probe(m, engine._ptr(out), engine._ptr(corpus.feats), engine._ptr(corpus.norms), corpus.d, engine._ptr(batch.descs_dev), K, 9,
                                 1000, 1000, 0.095, engine._ptr(work), work.numel(), engine._stream())) for m in (2, 12, 1, 11)]
        print("round %d offset %8d: crp32 %.3f rows %.3f cols %.3f rows-loads %.3f cols-loads %.3f" % (rnd, off, t_crp, t[0], t[1], t[2], t[3]), flush=True)
