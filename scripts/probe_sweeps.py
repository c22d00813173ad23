"""Filter + smoother kernel times of a 16-lane shape with 11 <= n <= 16 states: default build
(sweeps tiled into a few asm statements, mk_sweeps.h) against ``METRAN_HIP_JIT_FLAGS=-DMK_NO_TILED_SWEEPS`` (one statement
per FMA, an `s_nop` at every sweep boundary -- what these shapes had before round 3).  The shapes are compiled at run time
(metran_amd/jit.py); run the script once per setting.

    python scripts/probe_sweeps.py [--shapes 12,3 14,2] [--batch 4096] [--T 1000]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["12,3", "14,2"])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch

    import oracle
    from metran_amd.engine import BatchedKalman
    from metran_amd.synthetic import make_dfm_batch_torch

    for sh in args.shapes:
        N, K = (int(v) for v in sh.split(","))
        d = make_dfm_batch_torch(args.batch, N, K, args.T, seed=1200 + N, device=torch.device("cuda", 0))
        kf = BatchedKalman(0, layout="time_major")
        kf.set_observations(d["obs"]).set_loadings(d["loadings"])
        bufs = kf._alloc_outputs(args.batch, ["F", "Pf", "Xp", "Pp", "S", "Ps"])
        for _ in range(5):
            r = kf.filter_smooth(d["phi"], d["q"], buffers=bufs)
        torch.cuda.synchronize()
        kf.enable_timing(True, accumulate=True)
        for _ in range(args.steps):
            r = kf.filter_smooth(d["phi"], d["q"], buffers=bufs)
        torch.cuda.synchronize()
        f_tot, f_n, s_tot, s_n = kf.kernel_ms_totals()
        kf.enable_timing(False)
        # parity of a sub-sample against the oracle while we are here
        m = 8
        ref = oracle.dfm_batch(d["obs"][:m].cpu().numpy(), d["phi"][:m].cpu().numpy(), d["q"][:m].cpu().numpy(),
                               d["loadings"][:m].cpu().numpy())
        errS = float(np.abs(r["S"][:m].cpu().numpy() - ref["S"]).max())
        errP = float(np.abs(r["Ps"][:m].cpu().numpy() - ref["Ps"]).max())
        print(json.dumps({"shape": [N, K], "n": N + K, "batch": args.batch, "T": args.T,
                          "jit_flags": os.environ.get("METRAN_HIP_JIT_FLAGS", ""),
                          "filter_ms": f_tot / f_n, "smoother_ms": s_tot / s_n,
                          "models_per_s": args.batch / ((f_tot / f_n + s_tot / s_n) / 1e3),
                          "max_abs_err_smoothed_mean_vs_oracle": errS, "max_abs_err_smoothed_cov_vs_oracle": errP}))


if __name__ == "__main__":
    main()
