"""Placement of the key matrix by allocation path (VERDICT r2 task 5; one bounded run): the strip kernel and the two
selection kernels timed on 16 GB buffers obtained from (a) torch.empty, (b) hipMalloc, (c) hipMemCreate of ONE physical
chunk mapped into a reserved range, (d) hipMemCreate in 1 GB chunks -- several of each, all alive at once, one process.
usage: python tools/vmm_probe.py [pairs]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acoss_amd import engine, synth  # noqa: E402

engine.require_gpu()
V = ctypes.CDLL(os.path.join(ROOT, "tools", "vmm", "libvmm.so"))
V.vmm_alloc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
V.vmm_ptr.restype = ctypes.c_void_p
V.vmm_ptr.argtypes = [ctypes.c_void_p]
V.vmm_bytes.restype = ctypes.c_size_t
V.vmm_bytes.argtypes = [ctypes.c_void_p]
V.vmm_free.argtypes = [ctypes.c_void_p]
V.plain_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
V.vmm_granularity.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ch = synth.config2(n_songs=1000, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
batch = engine.PairBatch(corpus.frame_off, allp[:K], 9, corpus.device, pitch_align=32)
engine.oti(corpus, batch)
xp32 = engine.pack_x32(corpus, batch)
band = engine.planar32_band(corpus, batch)
n = engine.planar_elems(batch)
nbytes = n * 4 + 4096
gmin, grec = ctypes.c_size_t(0), ctypes.c_size_t(0)
rc = V.vmm_granularity(0, ctypes.byref(gmin), ctypes.byref(grec))
print("granularity rc %d: minimum %d recommended %d bytes; buffer %.2f GB" % (rc, gmin.value, grec.value, nbytes / 2 ** 30), flush=True)


class Raw(object):
    def __init__(self, ptr, device):
        self._p, self.device = ptr, device

    def data_ptr(self):
        return self._p


def timed(fn, reps=4):
    ts = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1))
    return float(np.min(ts))


first = torch.empty(n + 1024, dtype=torch.int32, device=corpus.device)
bits, work = engine.mask_bits_planar32(first[:n], band, corpus, batch, 0.095)
keep = []
rows = []


def measure(kind, ptr):
    out = Raw(ptr, corpus.device)
    t_crp = timed(lambda: engine.crp_planar32(corpus, batch, xp32, out=out))
    t_rows = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, False, out=bits, work=work))
    t_both = timed(lambda: engine.mask_bits_planar32(out, band, corpus, batch, 0.095, True, out=bits, work=work))
    rows.append((kind, ptr, t_crp, t_rows, t_both - t_rows))
    print("%-28s %#16x  strip %.3f  rows-call %.3f  cols %.3f  sum %.3f ms" % (kind, ptr, t_crp, t_rows, t_both - t_rows, t_crp + t_both), flush=True)


measure("torch.empty", first.data_ptr())
REPS = 3
for r in range(REPS - 1):
    t = torch.empty(n + 1024, dtype=torch.int32, device=corpus.device)
    keep.append(t)
    measure("torch.empty", t.data_ptr())
for r in range(REPS):
    p = ctypes.c_void_p()
    if V.plain_alloc(nbytes, ctypes.byref(p)) != 0:
        print("hipMalloc failed"); break
    measure("hipMalloc", p.value)
for label, chunk, align in (("vmm one chunk", 0, 0), ("vmm 1 GB chunks", 1 << 30, 0), ("vmm 2 MB chunks", 2 << 20, 0),
                            ("vmm one chunk, 1 GB aligned", 0, 1 << 30)):
    for r in range(REPS if chunk != (2 << 20) and align == 0 else 1):
        if torch.cuda.mem_get_info()[0] < nbytes + (8 << 30):
            print("%s: out of memory budget, skipped" % label, flush=True)
            break
        b = ctypes.c_void_p()
        rc = V.vmm_alloc(0, nbytes, chunk, align, ctypes.byref(b))
        if rc != 0:
            print("%s: vmm_alloc rc %d" % (label, rc), flush=True)
            break
        keep.append(b)
        measure(label, V.vmm_ptr(b))
a = {}
for kind, _, t1, t2, t3 in rows:
    a.setdefault(kind, []).append((t1, t2, t3, t1 + t2 + t3))
for kind, v in a.items():
    v = np.array(v)
    print("%-28s n=%d  strip %.3f..%.3f  rows %.3f..%.3f  cols %.3f..%.3f  sum %.3f..%.3f" % (
        kind, len(v), v[:, 0].min(), v[:, 0].max(), v[:, 1].min(), v[:, 1].max(), v[:, 2].min(), v[:, 2].max(), v[:, 3].min(), v[:, 3].max()))
