"""HBM bandwidth probes with stock torch kernels (dev tool): pure read, pure write, copy."""
import numpy as np, torch
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for gb in (0.25, 1, 4, 16):
    n = int(gb * (1 << 30) / 8)
    x = torch.empty(n, dtype=torch.float64, device="cuda").fill_(1.0)
    y = torch.empty_like(x)
    xi = x.view(torch.int64)
    r = t(lambda: x.sum()); w = t(lambda: y.fill_(2.0)); c = t(lambda: y.copy_(x)); m = t(lambda: xi.max())
    a = t(lambda: torch.add(x, 1.0, out=y))
    print("%5.2f GiB: read(sum) %.0f GB/s  read(max i64) %.0f GB/s  write(fill) %.0f GB/s  copy %.0f GB/s (r+w)  add %.0f GB/s (r+w)" % (
        gb, n * 8 / r / 1e6, n * 8 / m / 1e6, n * 8 / w / 1e6, 2 * n * 8 / c / 1e6, 2 * n * 8 / a / 1e6))
    del x, y
