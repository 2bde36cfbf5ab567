"""Write bandwidth by store pattern (dev tool): acoss_dev_store_probe modes on K matrices of 1000 x 1000 float64."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import _lib, engine
engine.require_gpu()
lib = _lib.load()
if not hasattr(lib, "acoss_dev_store_probe"):
    raise SystemExit("development probes are not in this build: python -m acoss_amd.build --probes")
fn = lib.acoss_dev_store_probe
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rows = 1000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
buf = torch.empty(K * rows * cols, dtype=torch.float64, device="cuda")
nbytes = buf.numel() * 8
st = torch.cuda.current_stream().cuda_stream
names = {0: "linear", 1: "tiles 128x128", 2: "strips 120 cols", 3: "bands 16 rows", 4: "band64 chunk-major", 5: "band32 walk right", 6: "u32 strips 120 (480 B)", 7: "u32 strips 240 (960 B)", 8: "u32 strips 112 (448 B)", 9: "f64 strips 112 (896 B)", 10: "u32 band32 walk right 112"}
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms))
t = timed(lambda: buf.fill_(1.0))
print("K=%d (%.1f GB): torch fill_ %.3f ms -> %.2f TB/s" % (K, nbytes / 1e9, t, nbytes / t / 1e9))
for mode in range(11):
    def f():
        rc = fn(buf.data_ptr(), K, rows, cols, mode, st)
        assert rc == 0, _lib.last_error()
    t = timed(f)
    nb = nbytes / 2 if mode in (6, 7, 8, 10) else nbytes
    print("  mode %d %-24s %.3f ms -> %.2f TB/s" % (mode, names[mode], t, nb / t / 1e9))
