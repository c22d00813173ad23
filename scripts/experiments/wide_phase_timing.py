"""Phase costs of the wide smoother: libraries built with -DMK_TUNE=<mask> (4: no fused factorisation / forward sweep, 2: no
backward sweep, 1: no products) against the shipped one, same box, configs[3] size.  Results of the tuned libraries are wrong
by construction; only the times mean anything.
  gpurun -- 'python scripts/experiments/wide_phase_timing.py ab/lib_tune4.so ab/lib_tune2.so ...'"""
import os
import subprocess
import sys

CHILD = r'''
import sys, torch
sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
B, N, K, T = 4096, 32, 4, 1000
d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=torch.device("cuda", 0), missing=0.3)
kf = BatchedKalman(layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
bufs = kf.alloc_projection(B)
kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
torch.cuda.synchronize()
kf.enable_timing(True, accumulate=True)
for _ in range(3):
    kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
torch.cuda.synchronize()
f_tot, f_n, s_tot, s_n = kf.kernel_ms_totals()
print("filter %.2f ms smoother %.2f ms (T = %d)" % (f_tot / f_n, s_tot / s_n, T))
'''
for lib in ["default"] + sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["METRAN_HIP_LIBRARY"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("%-22s %s" % (os.path.basename(lib), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
