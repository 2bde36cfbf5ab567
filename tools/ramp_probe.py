"""Per-step time of the product chain from the first step of a process (dev tool, round 5): does a fresh box's first process ramp up?
usage: python tools/ramp_probe.py [steps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
batch = engine.PairBatch(corpus.frame_off, synth.all_pairs(ch.n_songs)[:4096], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x32(corpus, batch)
koff, band = engine.keys16_koff(corpus, batch), engine.planar32_band(corpus, batch)
keys = engine.crp_keys16(corpus, batch, xp, koff)
bits, work = engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095)
torch.cuda.synchronize()
t_start = time.perf_counter()
ev = []
for s in range(steps):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    engine.crp_keys16(corpus, batch, xp, koff, out=keys)
    e[1].record()
    engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095, out=bits, work=work)
    e[2].record()
    engine.align_bits("qmax", bits, batch)
    e[3].record()
    ev.append(e)
torch.cuda.synchronize()
for s in list(range(0, min(steps, 12))) + list(range(12, steps, 10)):
    e = ev[s]
    print("step %3d: strip %.3f  mask %.3f  qmax %.3f" % (s, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])))
print("wall %.2f s for %d steps" % (time.perf_counter() - t_start, steps))
