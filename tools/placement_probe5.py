"""Strip / selection times at offset 0 of fresh allocations of different SIZES (dev tool)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
band = engine.planar32_band(corpus, batch)
n = engine.planar_elems(batch)
tmp = torch.empty(n, dtype=torch.int32, device=corpus.device)
bits, work = engine.mask_bits_planar32(tmp, band, corpus, batch, 0.095)
del tmp
torch.cuda.empty_cache()


def timed(fn, reps=4):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


GB = 1 << 30
for rnd in range(2):
    for size_gb in (15.1, 16, 20, 24, 32, 48, 64, 100, 16, 15.1):
        arena = torch.empty(int(size_gb * GB) // 4, dtype=torch.int32, device=corpus.device)
        out = arena[:n]
        t_crp = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
        t_rows = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, False, out=bits, work=work))
        t_both = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, True, out=bits, work=work))
        print("allocation of %5.1f GB at %#x: strip %.3f  rows call %.3f  cols %.3f  sum %.3f" % (size_gb, arena.data_ptr(), t_crp, t_rows, t_both - t_rows, t_crp + t_both), flush=True)
        del arena, out
        torch.cuda.empty_cache()
