"""Same-box A/B of builds of libmetran_hip.so on the adjoint gradient (mk_loglik_grad): kernel ms of the recording forward pass
and of the backward (adjoint) pass at several flight sizes, the gradient's checksum, and the end-to-end calibration of 8192 models.
  gpurun -- 'python scripts/ab_adjoint.py ab/lib_ADJ_BASE.so metran_amd/libmetran_hip.so [--calibrate]'"""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, sys, time, torch
sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman
from metran_amd.synthetic import make_dfm_batch_torch
from metran_amd.calibrate import calibrate_batch
out = {}
for B in (2048, 4096, 8192):
    d = make_dfm_batch_torch(B, 8, 2, 1000, seed=77, device=torch.device("cuda", 0))
    kf = BatchedKalman(layout="time_major")
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    kf.loglik_grad(d["phi"], d["q"]); torch.cuda.synchronize()
    kf.enable_timing(True, accumulate=True)
    for _ in range(5):
        mle, gphi, gq = kf.loglik_grad(d["phi"], d["q"])
    torch.cuda.synchronize()
    f, fn, s, sn = kf.kernel_ms_totals()
    out[str(B)] = {"forward_ms": f / fn, "backward_ms": s / sn, "chk": [float(gphi.abs().sum()), float(gq.abs().sum())]}
    kf.close()
if sys.argv[1] == "1":
    d = make_dfm_batch_torch(8192, 8, 2, 1000, seed=2000, device=torch.device("cuda", 0))
    kf = BatchedKalman(layout="time_major")
    kf.set_observations(d["obs"]).set_loadings(d["loadings"])
    calibrate_batch(kf, maxiter=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = calibrate_batch(kf, maxiter=200, fd_below=4096)
    torch.cuda.synchronize()
    out["calibrate_8192"] = {"seconds": time.perf_counter() - t0, "nit": int(res.nit), "launches": int(res.launches),
                             "converged_frac": float(res.converged.double().mean()), "obj_sum": float(res.obj.sum())}
print(json.dumps(out))
'''
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
cal = "1" if "--calibrate" in sys.argv else "0"
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, METRAN_HIP_LIBRARY=os.path.abspath(lib))
        o = subprocess.run([sys.executable, "-c", CHILD, cal], env=env, capture_output=True, text=True)
        line = [ln for ln in o.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(lib, "FAILED", o.stderr[-800:], flush=True)
            continue
        r = json.loads(line[0])
        print("%-24s round %d  " % (os.path.basename(lib), rnd) + "  ".join(
            "B=%s fwd %.2f bwd %.2f" % (b, r[b]["forward_ms"], r[b]["backward_ms"]) for b in ("2048", "4096", "8192")), flush=True)
        print("    chk %s" % r["8192"]["chk"] + ("   calibrate %s" % r["calibrate_8192"] if "calibrate_8192" in r else ""), flush=True)
