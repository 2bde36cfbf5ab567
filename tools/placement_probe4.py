"""Strip / selection times in windows of ONE large allocation at many offsets (dev tool): is the placement effect a function
of the address?  usage: python tools/placement_probe4.py [GB of the arena]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = 4096
ARENA_GB = float(sys.argv[1]) if len(sys.argv) > 1 else 96.0
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
band = engine.planar32_band(corpus, batch)
n = engine.planar_elems(batch)
arena = torch.empty(int(ARENA_GB * (1 << 30)) // 4, dtype=torch.int32, device=corpus.device)
print("arena at %#x, %.1f GB; window %.2f GB" % (arena.data_ptr(), ARENA_GB, n * 4 / 2 ** 30), flush=True)
bits, work = engine.mask_bits_planar32(arena[:n], band, corpus, batch, 0.095)


def timed(fn, reps=4):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


GB = 1 << 30
offs = [0, 2 << 20, 64 << 20, 512 << 20, 1 * GB, 2 * GB, 4 * GB, 8 * GB, 16 * GB, 16 * GB + (256 << 20), 17 * GB, 20 * GB, 24 * GB, 32 * GB, 33 * GB,
        40 * GB, 48 * GB, 56 * GB, 64 * GB, 72 * GB]
for off in offs:
    if off + n * 4 > arena.numel() * 4:
        continue
    out = arena[off // 4: off // 4 + n]
    t_crp = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
    t_rows = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, False, out=bits, work=work))
    t_both = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, True, out=bits, work=work))
    print("offset %7.3f GB: strip %.3f  rows call %.3f  cols %.3f  sum %.3f" % (off / GB, t_crp, t_rows, t_both - t_rows, t_crp + t_both), flush=True)
