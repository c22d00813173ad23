"""Dev probe: practical HBM ceilings of the box (write-only, copy, read-only) with torch ops."""
import torch, time
dev = torch.device("cuda", 0)
n = 1 << 29  # 4 GiB of f64
a = torch.empty(n, dtype=torch.float64, device=dev); b = torch.empty_like(a)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = n * 8 / 1e9
ms = t(lambda: a.fill_(1.5)); print("fill  (write %.1f GB): %.3f ms  %.0f GB/s" % (gb, ms, gb / ms * 1e3))
ms = t(lambda: a.zero_()); print("zero  (write %.1f GB): %.3f ms  %.0f GB/s" % (gb, ms, gb / ms * 1e3))
ms = t(lambda: b.copy_(a)); print("copy  (r+w %.1f GB): %.3f ms  %.0f GB/s" % (2 * gb, ms, 2 * gb / ms * 1e3))
ms = t(lambda: a.sum()); print("sum   (read %.1f GB): %.3f ms  %.0f GB/s" % (gb, ms, gb / ms * 1e3))
ms = t(lambda: torch.add(a, 1.0, out=b)); print("add   (r+w %.1f GB): %.3f ms  %.0f GB/s" % (2 * gb, ms, 2 * gb / ms * 1e3))
