import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
from acoss_amd import Serra09 as S9
# dirty the allocator with NaNs
junk = [torch.full((1 << 26,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(8)]
del junk
full = synth.make_corpus(3, 2, seed=31, lengths=lambda r: r.integers(900, 1400))
songs = [np.ascontiguousarray(S9.block_aggregate(full.song(i).T, 8, np.median).T) for i in range(6)]
g = np.stack([S9.global_chroma(full.song(i)) for i in range(6)])
off = np.concatenate([[0], np.cumsum([s.shape[0] for s in songs])]).astype(np.int64)
corpus = engine.DeviceCorpus(np.concatenate(songs), off, gchroma=g)
pairs = np.array([[0, 1], [2, 3], [4, 1], [5, 5]], dtype=np.int32)
for rep in range(3):
    a = engine.serra09_scores(corpus, pairs)
    b = engine.serra09_scores_staged(corpus, pairs)
    print(rep, "qmax equal", np.array_equal(a["qmax"], b["qmax"]), "dmax equal", np.array_equal(a["dmax"], b["dmax"]), a["qmax"], b["qmax"])
# stage-level comparison
batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device)
engine.oti(corpus, batch)
xp = engine.pack_x(corpus, batch)
T = torch.full((batch.total_crp,), float("nan"), dtype=torch.float64, device="cuda")
engine.crp(corpus, batch, xp, out=T)
Tv = torch.full((batch.total_crp,), float("nan"), dtype=torch.float64, device="cuda")
engine.crp(corpus, batch, xp, out=Tv, force_valu=True)
for p in range(4):
    d = batch.descs[p]; M, N = d["nx"] - 8, d["ny"] - 8
    A = T[d["crp_off"]:d["crp_off"] + M * d["crp_pitch"]].cpu().numpy().reshape(M, -1)[:, :N]
    B = Tv[d["crp_off"]:d["crp_off"] + M * d["crp_pitch"]].cpu().numpy().reshape(M, -1)[:, :N]
    print("pair", p, M, N, "strip==tile", np.array_equal(A, B), "nan in strip", np.isnan(A).sum(), "nan in tile", np.isnan(B).sum())
work = engine.thresholds(T, batch, 0.095)
Bm = engine.binarize(T, batch, 0.095)
mats, _ = batch.mats()
print("fused", engine.align_fused("qmax", T, batch, work).cpu().numpy(), "unfused", engine.align("qmax", Bm, mats).cpu().numpy())
