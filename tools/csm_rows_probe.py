"""Dev tool (round 4): get_csm at the bench's 4096 pairs per launch -- the row-band kernel against the column-strip kernel and the
VALU kernel, same batch, same output buffer, alternating; fraction of the 8 TB/s HBM peak on the algorithmic bytes."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x(corpus, batch)
C = torch.empty(batch.total_csm, dtype=torch.float64, device=corpus.device)
nx, ny = batch.descs["nx"].astype(np.float64), batch.descs["ny"].astype(np.float64)
algo = float(np.sum(8.0 * (nx * ny + corpus.d * (nx + ny))))
forms = {"rows": lambda: engine.csm_rows(corpus, batch, xp, out=C), "strip": lambda: engine.csm_strip(corpus, batch, xp, out=C),
         "valu": lambda: engine.csm(corpus, batch, out=C), "fill_": lambda: C.fill_(1.0)}
ref = engine.csm(corpus, batch).clone()
res = {}
for rnd in range(6):
    for name in (list(forms) if rnd % 2 else list(forms)[::-1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); forms[name](); e1.record(); torch.cuda.synchronize()
        if name in ("rows", "strip") and rnd == 0:
            # whole matrices: compare inside each pair's (nx, ny) window
            d0 = batch.descs[0]
            o, pt = int(d0["csm_off"]), int(d0["csm_pitch"])
            a = C[o:o + int(d0["nx"]) * pt].view(int(d0["nx"]), pt)[:, :int(d0["ny"])]
            b = ref[o:o + int(d0["nx"]) * pt].view(int(d0["nx"]), pt)[:, :int(d0["ny"])]
            assert torch.equal(a, b), name
        if rnd:
            res.setdefault(name, []).append(e0.elapsed_time(e1))
for name in forms:
    ms = float(np.median(res[name]))
    by = C.numel() * 8 if name == "fill_" else algo
    print("%-6s %.3f ms  %.2f TB/s  %.3f of 8 TB/s (%s bytes)" % (name, ms, by / ms / 1e9, by / ms / 1e9 / 8.0, "buffer" if name == "fill_" else "algorithmic"))
