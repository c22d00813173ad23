"""How the flight of calibrate_batch empties: models still active per iteration and wall time per iteration (8192 x (8,2), T = 1000)."""
import os
import sys
import io
import time
import contextlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from metran_amd import calibrate as cal  # noqa: E402
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

dev = torch.device("cuda", 0)
d = make_dfm_batch_torch(8192, 8, 2, 1000, seed=5000, device=dev, missing=0.0)
kf = BatchedKalman(0, layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
cal.calibrate_batch(kf, maxiter=2)
stamps = []
real_print = print


def stamp(*a, **k):
    torch.cuda.synchronize()
    stamps.append((time.perf_counter(), a[0] % a[1:] if len(a) > 1 else a[0]))


buf = io.StringIO()
import builtins  # noqa: E402

builtins.print = lambda *a, **k: (torch.cuda.synchronize(), stamps.append((time.perf_counter(), " ".join(str(x) for x in a))))
t0 = time.perf_counter()
res = cal.calibrate_batch(kf, maxiter=200, fd_below=4096, verbose=True)
builtins.print = real_print
prev = t0
cum = 0.0
for k, (t, line) in enumerate(stamps):
    act = int(line.split("active")[1].split()[0])
    cum += t - prev
    if k % 5 == 0 or k > len(stamps) - 5:
        print("it %3d active %5d  this iteration %.2f ms  cumulative %.3f s" % (k + 1, act, 1e3 * (t - prev), cum))
    prev = t
print("iterations", res.nit, "launches", res.launches, "total", stamps[-1][0] - t0)
