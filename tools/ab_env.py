"""A/B of an environment switch on the whole kernel chain of one batch, same buffers, one process (dev tool).
usage: python tools/ab_env.py VAR [pairs]   (the chain runs alternately with VAR unset and VAR=1)"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
VAR = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
keys = engine.crp_planar32(corpus, batch, xp32)
band = engine.planar32_band(corpus, batch)
os.environ.pop(VAR, None)
bits, work = engine.mask_bits_planar32(keys, band, corpus, batch, 0.095)
ref_bits = bits.clone()
scores = engine.align_bits("qmax", bits, batch).clone()
ref = scores.clone()


def chain(ev):
    ev[0].record()
    engine.pack_x32(corpus, batch, out=xp32)
    engine.crp_planar32(corpus, batch, xp32, out=keys)
    ev[1].record()
    engine.mask_bits_planar32(keys, band, corpus, batch, 0.095, out=bits, work=work)
    ev[2].record()
    engine.align_bits("qmax", bits, batch, scores=scores)
    ev[3].record()


res = {}
for rnd in range(9):
    for name in (("off", "on") if rnd % 2 else ("on", "off")):
        if name == "on":
            os.environ[VAR] = "1"
        else:
            os.environ.pop(VAR, None)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for rep in range(3):
            chain(ev)
        torch.cuda.synchronize()
        assert torch.equal(bits, ref_bits), "mask bits differ with %s=%s" % (VAR, name)
        assert torch.equal(scores, ref)
        if rnd:
            res.setdefault(name, []).append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)] + [ev[0].elapsed_time(ev[3])])
os.environ.pop(VAR, None)
for k in sorted(res):
    m = np.median(np.array(res[k]), axis=0)
    print("%s %-4s pack+crp %.3f  mask_bits %.3f  qmax %.3f  chain %.3f ms" % (VAR, k, m[0], m[1], m[2], m[3]))
