"""Does the mask stage's time depend on where its buffers lie?  (round 5: a fresh box's first bench process sometimes shows mask_bits
6.8-7.2 ms instead of 4.4.)  One process; per trial a dummy allocation of a different size shifts the addresses of the key plane, the
mask and the workspace, which are then allocated afresh.  usage: python tools/placement_probe_r05.py [trials]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
batch = engine.PairBatch(corpus.frame_off, synth.all_pairs(ch.n_songs)[:4096], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x32(corpus, batch)
koff, band = engine.keys16_koff(corpus, batch), engine.planar32_band(corpus, batch)


def timed(fn, reps=5):
    out = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return float(np.median(out[1:]))


for trial in range(trials):
    dummy = torch.empty(int((trial * 0.37 + 0.001) * (1 << 30)), dtype=torch.uint8, device=corpus.device)
    keys = engine.crp_keys16(corpus, batch, xp, koff)
    bits, work = engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095)
    t_strip = timed(lambda: engine.crp_keys16(corpus, batch, xp, koff, out=keys))
    t_mask = timed(lambda: engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095, out=bits, work=work))
    rwork = engine.radix16_work(batch)
    st = lambda what: engine.radix16_stage(what, keys, band, koff, corpus, batch, 0.095, bits, rwork)
    tc, tcr = timed(lambda: st(1)), timed(lambda: st(3))
    print("trial %d: keys %#x bits %#x work %#x (%.2f GB): strip %.3f  mask call %.3f  | with a workspace of its own: cols %.3f rows %.3f"
          % (trial, keys.data_ptr(), bits.data_ptr(), work.data_ptr(), work.numel() / 2**30, t_strip, t_mask, tc, tcr - tc), flush=True)
    del keys, bits, work, rwork, dummy
    torch.cuda.empty_cache()
