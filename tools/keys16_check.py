"""Development check of the 16-bit key path: masks against the 32-bit filter's on the same batch, timings of each kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
if frames > 0:
    ch = synth.config2(n_songs=100, n_frames=frames)
else:
    ch = synth.make_corpus(20, 5, seed=3, lengths=lambda r: int(np.clip(r.normal(520, 160), 60, 1032)))
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
rng = np.random.default_rng(0)
pairs = allp[rng.permutation(len(allp))[:K]]
batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
band = engine.planar32_band(corpus, batch)
koff = engine.keys16_koff(corpus, batch)
keys32 = engine.crp_planar32(corpus, batch, xp32)
bits32, work32 = engine.mask_bits_planar32(keys32, band, corpus, batch, 0.095)
bits32 = bits32.clone()
keys16 = engine.crp_keys16(corpus, batch, xp32, koff)
torch.cuda.synchronize()
# the plane against the quantised 32-bit keys
d0 = batch.descs[0]
M0, N0, pitch = int(d0["nx"]) - 8, int(d0["ny"]) - 8, int(d0["crp_pitch"])
k32 = keys32[int(d0["crp_off"]):int(d0["crp_off"]) + M0 * pitch].cpu().numpy().view(np.uint32).reshape(M0, pitch)[:, :N0] & 0x7fffffff
k16 = keys16[int(d0["crp_off"]):int(d0["crp_off"]) + M0 * pitch].cpu().numpy().view(np.uint16).reshape(M0, pitch)[:, :N0]
ko = int(koff[0].item()) & 0xffffffff
kp = np.maximum(k32.astype(np.int64) - ko, 0)
exp = np.minimum(np.maximum(kp >> 11, np.maximum((kp >> 9) - 49152, 0)), 0xFFFE)
print("plane of pair 0 equals the quantised 32-bit keys:", np.array_equal(exp, k16.astype(np.int64)), "key range", k16.min(), k16.max(), flush=True)
bits16, work16 = engine.mask_bits_keys16(keys16, band, koff, xp32, corpus, batch, 0.095)
torch.cuda.synchronize()
same = torch.equal(bits16, bits32)
print("masks identical:", same, flush=True)
if not same:
    W = engine.bits_words(batch)
    mm = batch.max_nx - 8
    a = bits16.view(batch.K, mm, W).cpu().numpy(); b = bits32.view(batch.K, mm, W).cpu().numpy()
    bad = np.argwhere((a != b).any(axis=2))
    print("differing (pair,row):", len(bad), bad[:10])
    p, r = bad[0]
    x = (a[p, r] ^ b[p, r]).view(np.uint64)
    print("xor words", [hex(int(v)) for v in x], "M,N", batch.M[p], batch.N[p])
    # which direction is wrong? compare against the non-mutual (rows only) masks
    r16, _ = engine.mask_bits_keys16(keys16, band, koff, xp32, corpus, batch, 0.095, mutual=False)
    r32, _ = engine.mask_bits_planar32(keys32, band, corpus, batch, 0.095, mutual=False)
    print("row-only masks identical:", torch.equal(r16, r32))
lib = engine._lib.load()
if hasattr(lib, "acoss_dev_side_counter") and not os.environ.get("ACOSS_K16_FLAGS"):
    import ctypes
    lib.acoss_dev_side_counter.restype = ctypes.c_void_p
    lib.acoss_dev_side_counter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    os.environ["ACOSS_K16_STATS"] = "1"
    for mutual in (False, True):
        engine.mask_bits_keys16(keys16, band, koff, xp32, corpus, batch, 0.095, mutual, out=bits16, work=work16)
        torch.cuda.synchronize()
        off = lib.acoss_dev_side_counter(work16.data_ptr(), batch.K, batch.max_nx, batch.max_ny, 9) - work16.data_ptr()
        c = work16[off:off + 256].view(torch.int32).cpu().numpy()[::16]
        nrc = int(batch.M.sum()) + (int(batch.N.sum()) if mutual else 0)
        print("mutual=%s: rows+cols %d  hand-overs %d (%.2f%%)  float32 recomputes %d (%.2f%%)  full passes %d (%.2f%%)  finer passes %d (%.2f%%)" % (
            mutual, nrc, c[0], 100.0 * c[0] / nrc, c[1], 100.0 * c[1] / nrc, c[2], 100.0 * c[2] / nrc, c[3], 100.0 * c[3] / nrc), flush=True)
def timed(fn, reps=5):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
os.environ.pop("ACOSS_K16_STATS", None)
print("K=%d: strip32 %.3f ms  strip16 %.3f ms | mask32 rows %.3f both %.3f | mask16 rows %.3f both %.3f" % (
    batch.K, timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=keys32)), timed(lambda: engine.crp_keys16(corpus, batch, xp32, koff, out=keys16)),
    timed(lambda: engine.mask_bits_planar32(keys32, band, corpus, batch, 0.095, False, out=bits32, work=work32)),
    timed(lambda: engine.mask_bits_planar32(keys32, band, corpus, batch, 0.095, True, out=bits32, work=work32)),
    timed(lambda: engine.mask_bits_keys16(keys16, band, koff, xp32, corpus, batch, 0.095, False, out=bits16, work=work16)),
    timed(lambda: engine.mask_bits_keys16(keys16, band, koff, xp32, corpus, batch, 0.095, True, out=bits16, work=work16))), flush=True)
