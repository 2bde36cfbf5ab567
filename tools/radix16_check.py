"""Radix selection (csrc/radix16_kernels.hip) against the wave-per-row selection on one batch of the config-2 workload, in one
process on the same buffers (dev tool, round 5).
usage: python tools/radix16_check.py [pairs] [ragged]
Checks: final mask bits equal; per sampled pair the column / row bounds t1 against numpy on the downloaded key plane.
Times: column kernel, row kernel, exact + apply, against rows / cols / whole mask_bits call of the old path."""
import ctypes, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth, _lib

args = sys.argv[1:]
for a in list(args):                    # lib=path: another build of the library (tools/build_variant.py)
    if a.startswith("lib="):
        _lib.LIB_PATH = os.path.abspath(a[4:])
        args.remove(a)
K = int(args[0]) if args and args[0].isdigit() else 4096
ragged = "ragged" in args
smooth = "smooth" in args
lib = _lib.load()
if ragged:
    ch = synth.make_corpus(40, 4, seed=7, lengths=lambda r: int(r.integers(60, 1033)))
elif smooth:
    ch = synth.config2_smooth(n_songs=1000 if K > 1000 else 64, n_frames=1000)
else:
    ch = synth.config2(n_songs=1000 if K > 1000 else 64, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
if ragged:
    rng = np.random.default_rng(3)
    allp = allp[rng.permutation(len(allp))]
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
K = batch.K
engine.oti(corpus, batch)
f32, n32 = engine.float32_copy(corpus)
xp = engine.pack_x32(corpus, batch)
koff = engine.keys16_koff(corpus, batch)
band = engine.planar32_band(corpus, batch)
keys = engine.crp_keys16(corpus, batch, xp, koff)
os.environ["ACOSS_RADIX16"] = "0"
bits, work = engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095)
os.environ["ACOSS_RADIX16"] = "1"
ref_bits = bits.clone()
P = engine._ptr
st = engine._stream
max_m = batch.max_nx - 8
nb = lib.acoss_radix16_work_bytes(K, batch.max_nx, batch.max_ny, 9)
rwork = torch.empty(nb, dtype=torch.uint8, device=keys.device)
ptrs = (ctypes.c_void_p * 8)()
dims = (ctypes.c_int * 4)()
lib.acoss_radix16_layout(P(rwork), K, batch.max_nx, batch.max_ny, 9, ptrs, dims)
base = rwork.data_ptr()
ldm, ldn, icap, isz = dims[0], dims[1], dims[2], dims[3]


def view(idx, dtype, count):
    off = ptrs[idx] - base
    return rwork[off:off + count * torch.tensor([], dtype=dtype).element_size()].view(dtype)


t1_row, t1_col = view(0, torch.int16, K * ldm), view(1, torch.int16, K * ldn)
item_row, item_col = view(2, torch.int32, K * ldm), view(3, torch.int32, K * ldn)
counters = view(4, torch.int32, 64)
flags = view(6, torch.uint8, K)
nbits = torch.zeros_like(ref_bits)


def stage(what, mutual=1):
    rc = lib.acoss_radix16_stage(what, P(keys), P(band), P(koff), P(corpus.feats), P(corpus.norms), corpus.d, P(batch.descs_dev), K, 9,
                                 batch.max_nx, batch.max_ny, 0.095, mutual, P(nbits), P(rwork), rwork.numel(), st())
    assert rc == 0, _lib.last_error()


def old(mutual):
    os.environ["ACOSS_RADIX16"] = "0"
    engine.check(lib.acoss_mask_bits_keys16_batch(P(keys), P(band), P(koff), P(xp), P(f32), P(n32), P(corpus.feats), P(corpus.norms),
                                                  corpus.d, P(batch.descs_dev), K, 9, batch.max_nx, batch.max_ny, 0.095, mutual, P(bits),
                                                  P(work), work.numel(), st()), "old")
    os.environ["ACOSS_RADIX16"] = "1"


def new_call():
    engine.check(lib.acoss_mask_bits_keys16_batch(P(keys), P(band), P(koff), P(xp), P(f32), P(n32), P(corpus.feats), P(corpus.norms),
                                                  corpus.d, P(batch.descs_dev), K, 9, batch.max_nx, batch.max_ny, 0.095, 1, P(bits),
                                                  P(work), work.numel(), st()), "new")


stage(7)
torch.cuda.synchronize()
cn = counters.cpu().numpy()
print("counters", cn[:4])
nitems = int(((item_row.view(K, ldm) >= 0).sum() + (item_col.view(K, ldn) >= 0).sum()).item()) if cn[2] == 0 else -1
print("pairs %d  items %d (%.2f %% of rows + columns; %d beyond the tiles' slots)  flagged lines %d  flagged pairs %d" % (
    K, nitems, 100.0 * nitems / max(1, sum(int(d["nx"]) + int(d["ny"]) - 16 for d in batch.descs)), cn[0], cn[1], cn[2]))
fl = flags.cpu().numpy().astype(bool)
diff = (nbits.view(K, -1) != ref_bits.view(K, -1)).any(1).cpu().numpy()
bad_pairs = [int(p) for p in np.nonzero(diff & ~fl)[0]]
print("mask bits: %d of %d unflagged pairs differ" % (len(bad_pairs), K - int(fl.sum())))
# the bounds against numpy, on a few pairs (and on the first that differ)
t1r, t1c = t1_row.cpu().numpy().view(np.uint16).reshape(K, ldm), t1_col.cpu().numpy().view(np.uint16).reshape(K, ldn)
ir, ic = item_row.cpu().numpy().reshape(K, ldm), item_col.cpu().numpy().reshape(K, ldn)
for p in (bad_pairs[:3] + [0, K // 2, K - 1]):
    d = batch.descs[p]
    M, N = int(d["nx"]) - 8, int(d["ny"]) - 8
    pl = keys[int(d["crp_off"]):int(d["crp_off"]) + M * int(d["crp_pitch"])].cpu().numpy().view(np.uint16).reshape(M, int(d["crp_pitch"]))[:, :N].astype(np.int64)
    for name, mat, t1, it, k in (("cols", pl.T, t1c[p, :N], ic[p, :N], int(np.rint(0.095 * M))), ("rows", pl, t1r[p, :M], ir[p, :M], int(np.rint(0.095 * N)))):
        s = np.sort(mat, axis=1)
        th = s[:, k - 1]
        cle = (mat <= th[:, None]).sum(1)
        nxt = (mat == th[:, None] + 1).any(1)
        clean = (cle == k) & ~nxt
        exp = np.where(clean, th + 1, th - 1)
        got_clean = it < 0
        wrong = (got_clean != clean) | (t1.astype(np.int64) != exp)
        print("  pair %d %s: %d lines, clean %d (expected %d), wrong %d" % (p, name, len(th), int(got_clean.sum()), int(clean.sum()), int(wrong.sum())))
        if wrong.any():
            w = np.nonzero(wrong)[0][:5]
            for q in w:
                print("     line %d: th %d cle %d k %d nxt %s -> expected t1 %d clean %s; got t1 %d item %d" % (q, th[q], cle[q], k, nxt[q], exp[q], clean[q], t1[q], it[q]))


def timed(fn, reps=6):
    out = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return float(np.median(out[1:]))


tc, tcr, tall = timed(lambda: stage(1)), timed(lambda: stage(3)), timed(lambda: stage(7))
stage(3); torch.cuda.synchronize()
print("direct: list + exact %.3f, list + exact + apply + flags %.3f;  by difference: list + exact %.3f" % (
    timed(lambda: stage(4 | 64)), timed(lambda: stage(4)), timed(lambda: stage(7 | 64)) - tcr))
print("ms per %d pairs:  radix cols %.3f  rows %.3f  exact+apply %.3f  all %.3f   | old rows %.3f cols %.3f mask_bits %.3f" % (
    K, tc, tcr - tc, tall - tcr, tall, timed(lambda: old(2)), timed(lambda: old(3)), timed(lambda: old(1))))
print("whole call: radix %.3f  old %.3f" % (timed(new_call), timed(lambda: old(1))))
new_call()
stt = (ctypes.c_int * 20)()
lib.acoss_mask_bits_keys16_stats(P(work), K, batch.max_nx, batch.max_ny, 9, stt)
print("whole call stats: extra items %d, flagged lines %d, flagged pairs %d; reasons %s; masks equal %s" % (stt[0], stt[1], stt[2], list(stt[8:18]), bool(torch.equal(bits, ref_bits))))
for dbg in (1, 2, 3, 4, 8, 5):
    print("phase cut %d: cols %.3f rows %.3f" % (dbg, timed(lambda: stage(1 | (dbg << 8))), timed(lambda: stage(3 | (dbg << 12))) - tc))
