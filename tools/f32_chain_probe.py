import sys, os, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from acoss_amd import engine, synth
rng = np.random.default_rng(13)
S, n = 200, 1000
feats = np.concatenate([np.cumsum(rng.standard_normal((n, 13)), axis=0).astype(np.float32) for _ in range(S)])
corpus = engine.DeviceCorpus(feats, np.arange(S + 1, dtype=np.int64) * n)
pairs = synth.all_pairs(S)[:4096]
batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
xp = engine.pack_x(corpus, batch)
T = engine.crp(corpus, batch, xp, sqrt_out=False)
bits, work = engine.mask_bits(T, batch, 0.095, mutual=True)
def t(fn, reps=4):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return float(np.median(ms[1:]))
print("pack_x   %.3f ms" % t(lambda: engine.pack_x(corpus, batch, out=xp)))
print("crp f32  %.3f ms" % t(lambda: engine.crp(corpus, batch, xp, sqrt_out=False, out=T)))
print("mask f64 %.3f ms" % t(lambda: engine.mask_bits(T, batch, 0.095, mutual=True, out=bits, work=work)))
print("qd       %.3f ms" % t(lambda: engine.align_bits_qd(bits, batch, boundary=1)))
t0 = time.perf_counter(); r = engine.serra09_scores(corpus, pairs, do_oti=False); torch.cuda.synchronize(); t1 = time.perf_counter()
t0 = time.perf_counter(); r = engine.serra09_scores(corpus, pairs, do_oti=False); torch.cuda.synchronize(); t1 = time.perf_counter()
print("serra09_scores(4096 pairs, f32 corpus) %.1f ms wall" % (1e3 * (t1 - t0)))
# where do thresholds sit relative to the window norm sums?  (key-range question for a float32 filter on raw operands)
Th = T[:int(batch.descs[0]['crp_off']) + 992 * int(batch.descs[0]['crp_pitch'])]
d0 = batch.descs[0]; M = int(d0['nx']) - 8
mat = T[int(d0['crp_off']):int(d0['crp_off']) + M * int(d0['crp_pitch'])].view(M, -1)[:, :M].cpu().numpy()
thr = np.sort(mat, axis=1)[:, 93]
nr = corpus.norms.cpu().numpy().astype(np.float64)
W = 9 * (nr[:n].max() + nr[n:2 * n].max())
print("row thresholds / (2 W): log2 min %.2f median %.2f max %.2f" % (np.log2(thr.min() / (2 * W)), np.log2(np.median(thr) / (2 * W)), np.log2(thr.max() / (2 * W))))
