"""Stage times of the 16-bit-key chain on songs beyond 1032 frames (the long form of the radix selection, round 5).
usage: python tools/long_stages.py [frames] [pairs]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ch = synth.config2(n_songs=128, n_frames=frames)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
batch = engine.PairBatch(corpus.frame_off, synth.all_pairs(ch.n_songs)[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x32(corpus, batch)
koff, band = engine.keys16_koff(corpus, batch), engine.planar32_band(corpus, batch)
keys = engine.crp_keys16(corpus, batch, xp, koff)
bits, work = engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095)
print("unresolved:", len(engine.mask_bits_keys16_unresolved(work, batch)))
rwork = engine.radix16_work(batch)


def timed(fn, reps=5):
    out = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return float(np.median(out[1:]))


st = lambda what: engine.radix16_stage(what, keys, band, koff, corpus, batch, 0.095, bits, rwork)
t_strip = timed(lambda: engine.crp_keys16(corpus, batch, xp, koff, out=keys))
tc, tcr, tall = timed(lambda: st(1)), timed(lambda: st(3)), timed(lambda: st(7))
t_mask = timed(lambda: engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095, out=bits, work=work))
t_qd = timed(lambda: engine.align_bits_qd(bits, batch, boundary=1))
t_q = timed(lambda: engine.align_bits("qmax", bits, batch))
cells = float(np.sum(batch.M.astype(np.float64) * batch.N))
print("%d pairs of %d frames (%.2f G cells): strip %.3f ms (%.0f GB/s written), cols %.3f, rows %.3f, exact+apply %.3f, mask call %.3f, qmax+dmax %.3f, qmax %.3f"
      % (batch.K, frames, cells / 1e9, t_strip, 2 * cells / t_strip / 1e6, tc, tcr - tc, tall - tcr, t_mask, t_qd, t_q))
print("per G cells: strip %.3f cols %.3f rows %.3f exact %.3f qd %.3f q %.3f" % tuple(x / (cells / 1e9) for x in (t_strip, tc, tcr - tc, tall - tcr, t_qd, t_q)))
