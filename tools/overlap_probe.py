"""Two steps of the chain in flight on two streams (dev probe, round 5): the strip kernel is bound by instruction issue, the radix
selection kernels by HBM -- does the GPU overlap step n's selection with step n+1's strip?
usage: python tools/overlap_probe.py [pairs] [streams]"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
engine.float32_copy(corpus)
NB = 4
sets = []
for s in range(NB):
    batch = engine.PairBatch(corpus.frame_off, allp[s * K:(s + 1) * K], 9, corpus.device, pitch_align=32)
    engine.oti(corpus, batch)
    xp = engine.pack_x32(corpus, batch)
    koff = engine.keys16_koff(corpus, batch)
    band = engine.planar32_band(corpus, batch)
    keys = engine.crp_keys16(corpus, batch, xp, koff)
    bits, work = engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095)
    sc = engine.align_bits("qmax", bits, batch).clone()
    sets.append(dict(batch=batch, xp=xp, koff=koff, band=band, keys=keys, bits=bits, work=work, ref=sc.clone()))
torch.cuda.synchronize()


def step(d):
    b = d["batch"]
    engine.oti(corpus, b)
    engine.pack_x32(corpus, b, out=d["xp"])
    engine.crp_keys16(corpus, b, d["xp"], d["koff"], out=d["keys"])
    engine.mask_bits_keys16(d["keys"], d["band"], d["koff"], d["xp"], corpus, b, 0.095, out=d["bits"], work=d["work"])
    return engine.align_bits("qmax", d["bits"], b)


def run(n_streams, steps=24):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % n_streams]):
            outs.append((i % NB, step(sets[i % NB]) if n_streams == 1 or True else None))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = all(torch.equal(o, sets[s]["ref"]) for s, o in outs[-NB:])
    return 1e3 * el / steps, ok


for ns in (1, 2, 1, 2, 3, 4):
    run(ns, 8)
    ms, ok = run(ns)
    print("streams %d: %.3f ms per %d-pair step (%.0f k pair-scores/s), scores ok %s" % (ns, ms, K, K / ms, ok))
