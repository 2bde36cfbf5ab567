"""Map of the placement effect at 2 GB resolution (dev tool): strip and selection time of a 512-pair batch in every 2 GB chunk
of one arena.  usage: python tools/placement_probe6.py [arena GB]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = 512
ARENA_GB = float(sys.argv[1]) if len(sys.argv) > 1 else 160.0
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
band = engine.planar32_band(corpus, batch)
n = engine.planar_elems(batch)
arena = torch.empty(int(ARENA_GB * (1 << 30)) // 4, dtype=torch.int32, device=corpus.device)
bits, work = engine.mask_bits_planar32(arena[:n], band, corpus, batch, 0.095)
print("arena %.0f GB at %#x; chunk batch = %d pairs = %.2f GB" % (ARENA_GB, arena.data_ptr(), K, n * 4 / 2 ** 30), flush=True)


def timed(fn, reps=5):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    return float(np.min(ts))


GB = 1 << 30
rows = []
for c in range(int(ARENA_GB // 2) - 1):
    off = c * 2 * GB
    out = arena[off // 4: off // 4 + n]
    t_crp = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
    t_sel = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, True, out=bits, work=work))
    rows.append((c * 2, t_crp, t_sel))
for i in range(0, len(rows), 4):
    print("   ".join("%3d GB: %.3f %.3f" % r for r in rows[i:i + 4]), flush=True)
a = np.array(rows)
print("strip: min %.3f median %.3f max %.3f   selection: min %.3f median %.3f max %.3f   corr %.2f" % (
    a[:, 1].min(), np.median(a[:, 1]), a[:, 1].max(), a[:, 2].min(), np.median(a[:, 2]), a[:, 2].max(), np.corrcoef(a[:, 1], a[:, 2])[0, 1]))
