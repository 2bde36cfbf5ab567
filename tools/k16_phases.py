"""Where the selection kernels' waves spend their cycles (dev tool; needs the probes build: python -m acoss_amd.build --probes).
K16Probe (csrc/keys16.h) stamps s_memtime at the phase boundaries of every row / column; this prints the sums per phase as
fractions of the waves' lifetimes, for the row kernel and the column kernel alone, on the bench workload.
usage: python tools/k16_phases.py [pairs]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
engine.require_gpu()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = engine._lib.load()
if not hasattr(lib, "acoss_dev_side_counter"):
    sys.exit("needs the probes build (python -m acoss_amd.build --probes)")
lib.acoss_dev_side_counter.restype = ctypes.c_void_p
lib.acoss_dev_side_counter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
batch = engine.PairBatch(corpus.frame_off, synth.all_pairs(ch.n_songs)[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp = engine.pack_x32(corpus, batch)
koff = engine.keys16_koff(corpus, batch)
band = engine.planar32_band(corpus, batch)
keys = engine.crp_keys16(corpus, batch, xp, koff)
bits, work = engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095)
off = lib.acoss_dev_side_counter(work.data_ptr(), batch.K, batch.max_nx, batch.max_ny, 9) - work.data_ptr()
NAMES = {"rows_kernel_only": ["waiting for the row's keys", "histogram pass (bin, LDS atomics, scan, decode)", "bin's keys + rank", "decision (reach, mask bits)", "stores + loop"],
         "cols_kernel_only": ["staging (loads -> LDS -> registers, barriers)", "histogram pass (bin, LDS atomics, scan, decode)", "bin's keys + rank", "decision (reach, mask bits)", "stores"]}


def timed(fn, reps=5):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for which in ("rows_kernel_only", "cols_kernel_only"):
    run = lambda: engine.mask_bits_keys16(keys, band, koff, xp, corpus, batch, 0.095, which, out=bits, work=work)
    os.environ.pop("ACOSS_K16_STATS", None); os.environ.pop("ACOSS_K16_FLAGS", None)
    t_plain = timed(run)
    os.environ["ACOSS_K16_STATS"] = "1"; os.environ["ACOSS_K16_FLAGS"] = "4"
    t_probe = timed(run)
    run(); torch.cuda.synchronize()
    c = work[off:off + 256].view(torch.int32).cpu().numpy().astype(np.int64)
    ph = c[4:11] * 64
    tot = ph.sum()
    units = max(int(c[11]), 1)
    print("%s: %.3f ms plain, %.3f ms with the stamps; %d %s sampled (one block in 64); %.0f cycles per %s and wave in all" % (
        which, t_plain, t_probe, units, "rows" if which.startswith("rows") else "columns", tot / units, "row" if which.startswith("rows") else "column"))
    for n, v in zip(NAMES[which], ph):
        print("    %-52s %5.1f %%   %7.0f cycles" % (n, 100.0 * v / tot, v / units))
    print("    (counters: hand-overs %d, float32 recomputes %d, full passes %d, finer passes %d)" % (c[0], c[16], c[32], c[48]))
