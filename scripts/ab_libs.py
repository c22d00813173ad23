"""Same-box A/B of several builds of libmetran_hip.so on the tape paths of configs[3] (4096 x (32,4), 30 % missing, T = 2000):
every library in its own process (METRAN_HIP_LIBRARY), interleaved over two rounds, kernel ms from hipEvents, and a checksum of
the outputs so that a variant that computes something else shows.
  gpurun -- 'python scripts/ab_libs.py ab/lib_BASE.so ab/lib_X.so ... [--state] [--T 2000] [--B 4096]'"""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, sys, torch
sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
T, B, state = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1"
d = make_dfm_batch_torch(B, 32, 4, T, seed=4000, device=torch.device("cuda", 0), missing=0.3)
kf = BatchedKalman(layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
if state:
    bufs = kf.alloc_state_variances(B)
    run = lambda: kf.smooth_state_variances(d["phi"], d["q"], buffers=bufs)
else:
    bufs = kf.alloc_projection(B)
    run = lambda: kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
run(); torch.cuda.synchronize()
kf.enable_timing(True, accumulate=True)
for _ in range(4):
    run()
torch.cuda.synchronize()
f, fn, s, sn = kf.kernel_ms_totals()
key = ("S", "var") if state else ("sim_means", "sim_vars")
print(json.dumps({"filter_ms": f / fn, "smoother_ms": s / sn, "mle_sum": float(bufs["mle"].sum()),
                  "chk": [float(bufs[k].double().abs().sum()) for k in key]}))
'''

args = [a for a in sys.argv[1:] if not a.startswith("--")]
state = "--state" in sys.argv
T = int(sys.argv[sys.argv.index("--T") + 1]) if "--T" in sys.argv else 2000
B = int(sys.argv[sys.argv.index("--B") + 1]) if "--B" in sys.argv else 4096
libs = [a for a in args if a.endswith(".so")]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, METRAN_HIP_LIBRARY=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", CHILD, str(T), str(B), "1" if state else "0"], env=env, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(lib, "FAILED", out.stderr[-600:], flush=True)
            continue
        r = json.loads(line[0])
        print("%-28s round %d  filter %.2f  smoother %.2f  -> %.0f models/s   mle_sum %.10e  chk %.12e %.12e" % (
            os.path.basename(lib), rnd, r["filter_ms"], r["smoother_ms"], B / ((r["filter_ms"] + r["smoother_ms"]) / 1e3),
            r["mle_sum"], r["chk"][0], r["chk"][1]), flush=True)
