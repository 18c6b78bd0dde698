"""GPU-box probe: what a pure streaming WRITE reaches on this part (calibrates the feature-extraction roof: that kernel reads
0.7 KB and writes 26 KB per row)."""
import torch
x = torch.empty(16384 * 6498, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, fn, nbytes in (("fill_ (write only)", lambda: x.fill_(1.0), x.numel() * 4), ("zero_ (memset)", lambda: x.zero_(), x.numel() * 4),
                         ("copy_ (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4)):
    ms = timeit(fn)
    print("%-22s %.4f ms  %.2f TB/s" % (name, ms, nbytes / ms / 1e9))
