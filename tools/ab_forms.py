"""A/B of the two forms of the float32 strip kernel (column strips of round 2 / row bands of round 3) on the SAME buffers in
one process, plus the selection kernels that read the result.  usage: python tools/ab_forms.py [pairs] [buffers]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth  # noqa: E402

engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
band = engine.planar32_band(corpus, batch)
n = engine.planar_elems(batch)
bufs = [torch.empty(n + 1024, dtype=torch.int32, device=corpus.device) for _ in range(NB)]
bits, work = engine.mask_bits_planar32(bufs[0][:n], band, corpus, batch, 0.095)


def timed(fn, reps=6):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


bytes_alg = 4.0 * K * 992 * 992 + 4.0 * 12 * 2000 * K
for bi, buf in enumerate(bufs):
    out = buf[:n]
    res = {}
    for form in ("cols", "rows", "cols", "rows"):
        os.environ["ACOSS_STRIP32_FORM"] = form
        t = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
        res.setdefault(form, []).append(t)
    ref = out.clone()
    os.environ["ACOSS_STRIP32_FORM"] = "cols"
    engine.crp_planar32(corpus, batch, xp32, out=out)
    torch.cuda.synchronize()
    same = bool(torch.equal(ref.view(-1, 1024)[:, :992], out.view(-1, 1024)[:, :992])) if n % 1024 == 0 else None
    os.environ["ACOSS_STRIP32_FORM"] = "rows"
    t_sel = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, True, out=bits, work=work))
    print("buffer %d: column strips %s ms (%.2f TB/s)   row bands %s ms (%.2f TB/s = %.3f of 8 TB/s)   keys identical: %s   selection %.3f ms"
          % (bi, ["%.3f" % t for t in res["cols"]], bytes_alg / min(res["cols"]) / 1e9, ["%.3f" % t for t in res["rows"]],
             bytes_alg / min(res["rows"]) / 1e9, bytes_alg / min(res["rows"]) / 1e9 / 8000.0, same, t_sel), flush=True)
