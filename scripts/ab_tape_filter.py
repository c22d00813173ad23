"""Same-box A/B of the two writers of the backward tape (mk_set_kernel_variant MK_VARIANT_TAPE_FILTER: "observable" = filter_obs_kernel,
round 6; "state" = filter_split_kernel OUT = 4, round 4) on configs[3]'s batch, interleaved, kernel ms from hipEvents, with checksums of
the outputs and the largest difference between the two variants' results.
  gpurun -- 'python scripts/ab_tape_filter.py [--state] [--T 2000] [--B 4096]'"""
import json
import sys

import torch

sys.path.insert(0, ".")
from metran_amd.engine import BatchedKalman  # noqa: E402
from metran_amd.synthetic import make_dfm_batch_torch  # noqa: E402

state = "--state" in sys.argv
T = int(sys.argv[sys.argv.index("--T") + 1]) if "--T" in sys.argv else 2000
B = int(sys.argv[sys.argv.index("--B") + 1]) if "--B" in sys.argv else 4096
d = make_dfm_batch_torch(B, 32, 4, T, seed=4000, device=torch.device("cuda", 0), missing=0.3)
kf = BatchedKalman(layout="time_major")
kf.set_observations(d["obs"]).set_loadings(d["loadings"])
bufs = kf.alloc_state_variances(B) if state else kf.alloc_projection(B)
run = (lambda: kf.smooth_state_variances(d["phi"], d["q"], buffers=bufs)) if state else (lambda: kf.simulate_smoothed(d["phi"], d["q"], buffers=bufs))
keys = ("S", "var") if state else ("sim_means", "sim_vars")
keep = {}
for rnd in range(3):
    for variant in ("state", "observable"):
        kf.set_variant("tape_filter", variant)
        run()
        torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        f, fn, s, sn = kf.kernel_ms_totals()
        kf.enable_timing(False)
        print("%-11s round %d  filter %.2f  smoother %.2f  -> %.0f models/s   mle_sum %.10e  chk %.12e %.12e" % (
            variant, rnd, f / fn, s / sn, B / ((f / fn + s / sn) / 1e3), float(bufs["mle"].sum()),
            float(bufs[keys[0]].double().abs().sum()), float(bufs[keys[1]].double().abs().sum())), flush=True)
        if rnd == 0:
            keep[variant] = {k: bufs[k].clone() for k in keys + ("mle",)}
a, b = keep["state"], keep["observable"]
print(json.dumps({"max_rel_diff_mle": float(((a["mle"] - b["mle"]).abs() / a["mle"].abs()).max()),
                  **{"max_abs_diff_" + k: float((a[k] - b[k]).abs().max()) for k in keys}}))
