"""Dev probe: objective latency on examples/data (one record, P+1 = 7 finite-difference instances)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metran_amd.engine import BatchedKalman
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g1_real.npz"))
kf = BatchedKalman(0).set_observations(g["obs"][None]).set_loadings(g["loadings"][None])
kf.enable_timing(True)
alpha = np.tile(g["alpha_star"], (7, 1)); alpha[1:] += 1e-8 * np.eye(6)
phi, q = kf.params_from_alpha(alpha)
for _ in range(3): m = kf.loglik(phi, q)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m = kf.loglik(phi, q)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("objective of 7 instances on examples/data (T=6255, 343 observed steps): kernel %.3f ms, wall %.3f ms, mle %.9f" % (kf.last_kernel_ms()[0], dt * 1e3, float(m[0])))
two = np.stack([g["obs"], g["obs"]])                       # two records -> the step-by-step filter
kf2 = BatchedKalman(0).set_observations(two).set_loadings(np.stack([g["loadings"]] * 2))
kf2.enable_timing(True)
phi2, q2 = kf2.params_from_alpha(np.tile(g["alpha_star"], (14, 1)))
for _ in range(3): m2 = kf2.loglik(phi2, q2)
print("step-by-step filter on the same record: kernel %.3f ms, mle %.9f" % (kf2.last_kernel_ms()[0], float(m2[0])))
