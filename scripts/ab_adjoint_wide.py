"""Same-box A/B of builds of libmetran_hip.so on the WIDE adjoint gradient (mk_loglik_grad, 16 < n <= 64): kernel ms of the recording
forward pass and of the backward pass at the flight sizes a wide calibration runs at, with the gradient's checksum.
  gpurun -- 'python scripts/ab_adjoint_wide.py ab/lib_base.so ab/lib_new.so'"""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, sys, torch
sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
out = {}
for B, T in ((1, 2000), (64, 500), (512, 500), (2048, 500)):
    d = make_dfm_batch_torch(B, 32, 4, T, seed=77, device=torch.device("cuda", 0), missing=0.3)
    kf = BatchedKalman(layout="time_major")
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    kf.loglik_grad(d["phi"], d["q"]); torch.cuda.synchronize()
    kf.enable_timing(True, accumulate=True)
    for _ in range(4):
        mle, gphi, gq = kf.loglik_grad(d["phi"], d["q"])
    torch.cuda.synchronize()
    f, fn, s, sn = kf.kernel_ms_totals()
    out["%dx%d" % (B, T)] = {"forward_ms": round(f / fn, 3), "backward_ms": round(s / sn, 3), "chk": [float(gphi.abs().sum()), float(gq.abs().sum())]}
    kf.close()
print(json.dumps(out))
'''
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, METRAN_HIP_LIBRARY=os.path.abspath(lib))
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [ln for ln in o.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(lib, "FAILED", o.stderr[-800:], flush=True)
            continue
        r = json.loads(line[0])
        print(os.path.basename(lib), "round", rnd, " ".join("%s: fwd %.2f bwd %.2f (chk %.9e %.9e)" % (k, v["forward_ms"], v["backward_ms"], v["chk"][0], v["chk"][1])
                                                              for k, v in r.items()), flush=True)
