"""Alignment of batch i on a second stream while the strip kernel of batch i+1 runs on the first (dev tool)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
band = engine.planar32_band(corpus, batch)
xp = engine.pack_x32(corpus, batch)
T = engine.crp_planar32(corpus, batch, xp)
bits0, work = engine.mask_bits_planar32(T, band, corpus, batch, 0.095)
bits = [bits0, bits0.clone()]
scores = [engine.align_bits("qmax", bits0, batch).clone() for _ in range(2)]
ref = scores[0].clone()
R = 12


def front(b):
    engine.pack_x32(corpus, batch, out=xp)
    engine.crp_planar32(corpus, batch, xp, out=T)
    engine.mask_bits_planar32(T, band, corpus, batch, 0.095, out=bits[b], work=work)


for name in ("one stream", "alignment on a second stream", "one stream", "alignment on a second stream"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if name == "one stream":
        for r in range(R):
            front(r & 1)
            engine.align_bits("qmax", bits[r & 1], batch, scores=scores[r & 1])
    else:
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        done = [None, None]
        for r in range(R):
            b = r & 1
            with torch.cuda.stream(s1):
                if done[b] is not None:
                    s1.wait_event(done[b])          # the alignment that read bits[b] two rounds ago
                front(b)
                ready = torch.cuda.Event(); ready.record(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(ready)
                engine.align_bits("qmax", bits[b], batch, scores=scores[b])
                done[b] = torch.cuda.Event(); done[b].record(s2)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = bool(torch.equal(scores[0], ref) and torch.equal(scores[1], ref))
    print("%-30s %d x %d pairs in %.1f ms -> %.0f pair-scores/s (%.3f ms / batch; scores unchanged: %s)" % (name, R, K, el * 1e3, R * K / el, el * 1e3 / R, ok), flush=True)
