"""Time mask_bits_fused alone at bench size (for rocprofv3): python tools/fused_time.py [pairs] [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acoss_amd import engine, synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ch = synth.config2(n_songs=200, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
pairs = synth.all_pairs(ch.n_songs)[:P]
batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
band = engine.planar32_band(corpus, batch, fused=True)
bits, work = engine.mask_bits_fused(corpus, batch, 0.095, band=band)
ms = []
for rep in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    engine.mask_bits_fused(corpus, batch, 0.095, band=band, out=bits, work=work, verify=False)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print("mask_bits_fused %d pairs: %s ms; undecided %d" % (P, ["%.3f" % m for m in ms], int(engine.fused_counter(work).item())))
