"""Phase shares of the band kernel (needs `python -m acoss_amd.build --probes`; sets ACOSS_BAND_STAMP=1):
python tools/band_stamps.py [pairs]"""
import ctypes
import os
import sys

os.environ["ACOSS_BAND_STAMP"] = "1"
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acoss_amd import engine, synth, _lib  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=200, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
pairs = synth.all_pairs(ch.n_songs)[:P]
batch = engine.PairBatch(corpus.frame_off, pairs, 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
band = engine.planar32_band(corpus, batch, fused=True)
bits, work = engine.mask_bits_fused(corpus, batch, 0.095, band=band)
lib = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(lib, "acoss_dev_band_stamps"):
    raise SystemExit("development probes are not in this build: python -m acoss_amd.build --probes")
out = (ctypes.c_ulonglong * 12)()
lib.acoss_dev_band_stamps(out, 12)          # reset
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
engine.mask_bits_fused(corpus, batch, 0.095, band=band, out=bits, work=work, verify=False)
e1.record()
torch.cuda.synchronize()
lib.acoss_dev_band_stamps(out, 12)
v = np.array(list(out), dtype=np.float64)
names = ["0 band top (issue tile load)", "1 sync (x frames / band above)", "2 A fragments + first y stage", "3 phase A: mfma + epilogue (per chunk)",
         "4 barrier (per chunk)", "5 stage next y tile + load (per chunk)", "6 phase B: window sums (per chunk)", "7 validity + cold start",
         "8 select3", "9 per-row: slow paths + emission", "10 column staging / band end", "11 -"]
print("stamped launch: %.3f ms; total wave-cycles %.3e" % (e0.elapsed_time(e1), v.sum()))
for n, x in zip(names, v):
    print("  %-48s %6.2f %%" % (n, 100.0 * x / v.sum()))
