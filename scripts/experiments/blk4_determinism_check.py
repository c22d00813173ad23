"""The test that exposed the block path's round-3 irreproducibility, and must now come out clean: full-occupancy runs
(4096 models = two wavefronts per SIMD) of the block path against the shipped tile kernel, three times each, on the
record outputs AND on the projection path, plus the oracle on a subset.
  METRAN_HIP_LIBRARY=ab/libmetran_hip_blk4.so python scripts/experiments/blk4_determinism_check.py
Round 3: with <= 1024 models the two paths agreed to 8e-16; with 4096 a quarter of the model-steps had covariance entries
off by <= 2e-7, differently every run.  Cause (found statically, scripts/check_asm_hazards.py rule M2): the write-back's
inline-asm ds_write stored MFMA accumulators 3-8 wait states behind their last MFMA, 9 are required."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

N, K, T = 32, 4, 60
dev = torch.device("cuda", 0)
print("library:", os.environ.get("METRAN_HIP_LIBRARY", "default"))
ok = True
for B in (1024, 4096):
    d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=dev, missing=0.3)
    out = {}
    for variant in ("mfma", "mfma_blk4", "mfma_blk4_unfolded"):
        kf = BatchedKalman(layout="time_major").set_variant("wide_smoother", variant)
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        runs = []
        for _ in range(3):
            r = kf.filter_smooth(d["phi"], d["q"])
            torch.cuda.synchronize()
            runs.append((r["S"].clone(), r["Ps"].clone()))
        bufs = kf.alloc_projection(B)
        proj = []
        for _ in range(3):
            kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs)
            torch.cuda.synchronize()
            proj.append((bufs["sim_means"].clone(), bufs["sim_vars"].clone()))
        same = all(torch.equal(runs[0][i], x[i]) for x in runs[1:] for i in (0, 1)) and \
            all(torch.equal(proj[0][i], x[i]) for x in proj[1:] for i in (0, 1))
        print("  B %5d %-20s three runs bit-identical: %s" % (B, variant, same), flush=True)
        ok &= same
        out[variant] = (runs[0], proj[0])
        kf.close()
        del kf, r, bufs
    for variant in ("mfma_blk4", "mfma_blk4_unfolded"):
        (S0, P0), (m0, v0) = out["mfma"]
        (S1, P1), (m1, v1) = out[variant]
        scale = float(P0.abs().max())
        dP, dS = float((P1 - P0).abs().max()), float((S1 - S0).abs().max())
        dm, dv = float((m1 - m0).abs().max()), float((v1 - v0).abs().max())
        bad = ((P1 - P0).abs() > 1e-10 * max(scale, 1.0)).any(3).any(2)
        print("  B %5d %-20s vs tile kernel: max dS %.2e dPs %.2e (scale %.2e) dmean %.2e dvar %.2e; model-steps off by > 1e-10: %d of %d"
              % (B, variant, dS, dP, scale, dm, dv, int(bad.sum()), bad.numel()), flush=True)
        ok &= int(bad.sum()) == 0 and dS < 1e-9 and dm < 1e-9 and dv < 1e-9
    del out, d
    torch.cuda.empty_cache()
# the oracle (CPU restatement of the reference) on a subset
try:
    import oracle
    B = 64
    d = make_dfm_batch_torch(B, N, K, T, seed=4000, device=dev, missing=0.3)
    host = {k: d[k].cpu().numpy() for k in ("obs", "phi", "q", "loadings")}
    ref = oracle.dfm_batch(host["obs"], host["phi"], host["q"], host["loadings"], smooth=True)
    for variant in ("mfma_blk4", "mfma_blk4_unfolded"):
        kf = BatchedKalman(layout="time_major").set_variant("wide_smoother", variant)
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        r = kf.filter_smooth(d["phi"], d["q"])
        eS = float(np.abs(r["S"].cpu().numpy() - ref["S"]).max())
        eP = float(np.abs(r["Ps"].cpu().numpy() - ref["Ps"]).max())
        print("  %-20s vs oracle (64 models): max |dS| %.2e  max |dPs| %.2e" % (variant, eS, eP))
        ok &= eS < 1e-9 and eP < 1e-9
        kf.close()
except Exception as e:  # noqa: BLE001 -- the oracle's signature is test infrastructure; the tile-kernel comparison above stands alone
    print("  (oracle comparison skipped: %s: %s)" % (type(e).__name__, e))
print("BLK4 CHECK", "PASSED" if ok else "FAILED")
sys.exit(0 if ok else 1)
