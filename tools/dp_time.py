"""Times of the alignment kernels on the bit-packed masks of 4096 config-2 pairs (dev tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
batch = engine.PairBatch(corpus.frame_off, synth.all_pairs(ch.n_songs)[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
koff = engine.keys16_koff(corpus, batch); band = engine.planar32_band(corpus, batch)
k16 = engine.crp_keys16(corpus, batch, xp32, koff)
bits, _ = engine.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095)
def timed(fn, reps=6):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
q = engine.align_bits("qmax", bits, batch).cpu().numpy()
d = engine.align_bits("dmax", bits, batch, boundary=1).cpu().numpy()
q2, d2 = engine.align_bits_qd(bits, batch, boundary=1)
print("qd == separate:", np.array_equal(q, q2.cpu().numpy()), np.array_equal(d, d2.cpu().numpy()), "checksum", float(q.sum()), float(d.sum()))
print("K=%d: qmax %.3f ms  dmax %.3f ms  swc %.3f ms  qmax+dmax in one sweep %.3f ms" % (K, timed(lambda: engine.align_bits("qmax", bits, batch)),
      timed(lambda: engine.align_bits("dmax", bits, batch, boundary=1)), timed(lambda: engine.align_bits("swc", bits, batch)),
      timed(lambda: engine.align_bits_qd(bits, batch, boundary=1))))
