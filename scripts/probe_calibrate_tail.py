"""How the flight of calibrate_batch empties: models still active per iteration (8192 x (8,2), T = 1000)."""
import os
import sys
import io
import contextlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from metran_amd.calibrate import calibrate_batch  # noqa: E402
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

dev = torch.device("cuda", 0)
d = make_dfm_batch_torch(8192, 8, 2, 1000, seed=5000, device=dev, missing=0.0)
kf = BatchedKalman(0, layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    res = calibrate_batch(kf, maxiter=200, fd_below=4096, verbose=True)
act = [int(ln.split("active")[1].split()[0]) for ln in buf.getvalue().splitlines() if "active" in ln]
print("iterations", res.nit, "launches", res.launches, "active by iteration (every 5th):", act[::5])
