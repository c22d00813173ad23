"""Dev probe: adjoint gradient vs batched finite differences at the benchmark shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
B, N, K, T = int(os.environ.get("B", 4096)), 8, 2, 1000
dev = torch.device("cuda", 0)
d = make_dfm_batch_torch(B, N, K, T, seed=7, device=dev)
kf = BatchedKalman(0, layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
kf.enable_timing(True)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
n = N + K
t_adj = timeit(lambda: kf.loglik_grad_alpha(d["alpha"]))
f, s = kf.last_kernel_ms()
pts = torch.cat([d["alpha"][None], d["alpha"][None] + 1e-8 * torch.eye(n, device=dev, dtype=torch.float64)[:, None, :]], 0).reshape((n + 1) * B, n)
def fd():
    phi, q = kf.params_from_alpha(pts)
    return kf.loglik(phi, q)
t_fd = timeit(fd)
print("B=%d  adjoint gradient %.2f ms (forward filter %.2f + backward %.2f)   finite differences (%d instances) %.2f ms   -> %.1fx" % (B, t_adj, f, s, (n + 1) * B, t_fd, t_fd / t_adj))
