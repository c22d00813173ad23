import sys, time
sys.path.insert(0, "/root/repo")
import torch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
dev = torch.device("cuda", 0)
for miss in (0.0, 0.3):
    d = make_dfm_batch_torch(8192, 8, 2, 1000, seed=5000, device=dev, missing=miss)
    kf = BatchedKalman(0, layout="time_major")
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    for name, fn in (("f64", kf.loglik), ("f32", kf.loglik_f32)):
        for _ in range(5): m = fn(d["phi"], d["q"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m = fn(d["phi"], d["q"])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        if name == "f64": ref = m
        print(miss, name, "ms", round(dt * 1e3, 3), "max rel err", float(((m - ref).abs() / ref.abs()).max()))
    kf.close()
