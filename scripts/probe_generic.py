"""The size-generic kernel family (mk_generic.hip) timed at a few shapes, with its results against the specialised kernels
(where the shape has them) or the oracle (beyond 64 states).  hipEvents per launch.
  gpurun -- 'python scripts/probe_generic.py [N,K,T,B ...]'"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

dev = torch.device("cuda", 0)
KEYS = ("Xp", "Pp", "F", "Pf", "S", "Ps")
cases = [(8, 2, 1000, 4096), (5, 1, 1000, 4096), (14, 3, 500, 2048), (32, 4, 2000, 512), (48, 3, 500, 256), (70, 3, 300, 256),
         (96, 4, 300, 256), (120, 8, 100, 128)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]
for (N, K, T, B) in cases:
    d = make_dfm_batch_torch(B, N, K, T, seed=5, device=dev, missing=0.3)
    out = {}
    fams = ("generic", "specialised") if N + K <= 36 else ("generic",)
    for fam in fams:
        kf = BatchedKalman(0, layout="time_major")
        kf.set_variant("kernel_family", fam)
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        r = kf.filter_smooth(d["phi"], d["q"])
        torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            r = kf.filter_smooth(d["phi"], d["q"])
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        f, fn, s, sn = kf.kernel_ms_totals()
        out[fam] = {k: r[k].clone() for k in KEYS + ("mle",)}
        print("(%d,%d) T=%d B=%d  %-11s filter %9.3f ms  smoother %9.3f ms  %10.1f models/s" % (
            N, K, T, B, fam, f / fn, s / sn, B / wall), flush=True)
        kf.close()
        del r
    if len(fams) == 2:
        for k in KEYS + ("mle",):
            a, b = out["generic"][k], out["specialised"][k]
            print("      %-3s max |generic - specialised| = %.2e (scale %.1e)" % (k, float((a - b).abs().max()), float(b.abs().max())))
    else:
        import oracle
        obs = d["obs"].cpu().numpy()
        for b in (0, B - 1):
            Z = np.concatenate([np.eye(N), d["loadings"][b].cpu().numpy()], axis=1)
            o, oi, oc = oracle.set_observations(obs[b])
            phi, q = d["phi"][b].cpu().numpy(), d["q"][b].cpu().numpy()
            sg, df, sc, F, Pf, Xp, Pp = oracle.seqkalmanfilter(o, np.diag(phi), np.diag(q), Z, np.zeros(N), oi, oc, np.zeros(N + K), np.eye(N + K))
            S, Ps = oracle.kalmansmoother(F, Pf, Xp, Pp, np.diag(phi))
            mle = oracle.get_mle(sg[:sc], df[:sc], oc)
            ref = dict(Xp=Xp, Pp=Pp, F=F, Pf=Pf, S=S, Ps=Ps)
            errs = {k: float(np.abs(out["generic"][k][b].cpu().numpy() - ref[k]).max()) for k in KEYS}
            print("      model %d vs oracle: mle rel %.1e  " % (b, abs(float(out["generic"]["mle"][b]) - mle) / abs(mle))
                  + "  ".join("%s %.1e" % kv for kv in errs.items()), flush=True)
    del out, d
    torch.cuda.empty_cache()
