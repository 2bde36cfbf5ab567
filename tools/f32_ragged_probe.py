"""Stage times of the 16-bit-key chain on a float32 corpus of 13-d random walks (the MFCC stand-in), ragged or PROBE_FRAMES long, with the radix
selection on and off, against the plain float32-input chain (dev tool, rounds 4-5).  usage: [PROBE_FRAMES=1000] python tools/f32_ragged_probe.py"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
rng = np.random.default_rng(1)
S = 220
songs = []
for s_ in range(S):
    n = int(os.environ.get("PROBE_FRAMES", 0)) or int(rng.integers(60, 1033))
    songs.append(np.cumsum(rng.standard_normal((n, 13)) * rng.uniform(0.2, 3.0), axis=0).astype(np.float32))
feats = np.concatenate(songs); off = np.cumsum([0] + [len(s) for s in songs]).astype(np.int64)
corpus = engine.DeviceCorpus(feats, off)
pairs = synth.all_pairs(S)[:4096]
batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
def t(fn, reps=4):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return float(np.median(ms[1:]))
xp32 = engine.pack_x32(corpus, batch)
koff = engine.keys16_koff_f32(corpus, batch, xp32); band = engine.keys16_band_f32(corpus, batch)
k16 = engine.crp_keys16(corpus, batch, xp32, koff)
bits, work = engine.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095)
import ctypes
st = (ctypes.c_int * 20)()
engine._lib.load().acoss_mask_bits_keys16_stats(engine._ptr(work), batch.K, batch.max_nx, batch.max_ny, 9, st)
print("radix stats: extra items %d, flagged lines %d, flagged pairs %d of %d; reasons %s" % (st[0], st[1], st[2], batch.K, list(st[8:18])))
for env in ("1", "0"):
    os.environ["ACOSS_RADIX16"] = env
    print("ACOSS_RADIX16=%s mask %.3f ms" % (env, t(lambda: engine.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, True, out=bits, work=work))))
os.environ["ACOSS_RADIX16"] = "1"
print("koff   %.3f ms" % t(lambda: engine.keys16_koff_f32(corpus, batch, xp32)))
print("strip  %.3f ms" % t(lambda: engine.crp_keys16(corpus, batch, xp32, koff, out=k16)))
print("rows   %.3f ms" % t(lambda: engine.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, "rows_kernel_only", out=bits, work=work)))
print("cols   %.3f ms" % t(lambda: engine.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, "cols_kernel_only", out=bits, work=work)))
print("mask   %.3f ms" % t(lambda: engine.mask_bits_keys16(k16, band, koff, xp32, corpus, batch, 0.095, True, out=bits, work=work)))
print("qd     %.3f ms" % t(lambda: engine.align_bits_qd(bits, batch, boundary=1)))
xp = engine.pack_x(corpus, batch); T = engine.crp(corpus, batch, xp, sqrt_out=False)
b2, w2 = engine.mask_bits(T, batch, 0.095)
print("old: crp f32 %.3f ms, mask f64 %.3f ms" % (t(lambda: engine.crp(corpus, batch, xp, sqrt_out=False, out=T)), t(lambda: engine.mask_bits(T, batch, 0.095, out=b2, work=w2))))
print("masks equal:", bool(torch.equal(bits, b2)))
for name, fn in (("filter", lambda: engine.serra09_scores(corpus, pairs, do_oti=False)), ("old", lambda: engine.serra09_scores(corpus, pairs, do_oti=False, approx32=False))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); print(name, "serra09_scores 4096 ragged pairs: %.1f ms wall" % (1e3 * (time.perf_counter() - t0)))
