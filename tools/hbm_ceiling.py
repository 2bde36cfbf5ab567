"""What a plain streaming kernel reaches on this GPU (dev tool): torch fill (write only), copy (read + write) and sum (read only)
on buffers the size of the fast path's matrices.  Context for the roofline fractions in bench.py, which divide by the 8 TB/s peak."""
import sys, time
import torch
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(16e9)
a = torch.empty(n // 8, dtype=torch.float64, device="cuda")
b = torch.empty(n // 8, dtype=torch.float64, device="cuda")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: a.fill_(1.5)); print("fill  %5.1f GB: %.3f ms -> %.2f TB/s written" % (n / 1e9, t * 1e3, n / t / 1e12))
t = timed(lambda: b.copy_(a)); print("copy  %5.1f GB: %.3f ms -> %.2f TB/s moved (read + write)" % (n / 1e9, t * 1e3, 2 * n / t / 1e12))
t = timed(lambda: a.sum()); print("sum   %5.1f GB: %.3f ms -> %.2f TB/s read" % (n / 1e9, t * 1e3, n / t / 1e12))
