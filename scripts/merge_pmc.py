"""Box-side helper: merge the FETCH_SIZE and WRITE_SIZE passes of one bench configuration into
profiles-style ``pmc_hbm_<config>.json``, stamped with the hash of the kernel sources (bench.py refuses to quote
traffic from a profile whose stamp does not match the kernels it runs).
    python scripts/merge_pmc.py <config> <packed_sym 0|1> <fetch.json> <write.json> <out.json> "<command>" """
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg, sym, fetch, write, out, cmd = sys.argv[1], sys.argv[2] == "1", sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
f, w = json.load(open(fetch)), json.load(open(write))
B, N, K, T, missing, mode = bench.CONFIGS[cfg]
kernels = {}
for name, c in f.items():
    if name in w and "FETCH_SIZE" in c and "WRITE_SIZE" in w[name]:
        kernels[name] = {"FETCH_SIZE": c["FETCH_SIZE"], "WRITE_SIZE": w[name]["WRITE_SIZE"], "dispatches": c["dispatches"],
                         "duration_ms_mean_fetch_pass": c["duration_ms_mean"], "duration_ms_mean_write_pass": w[name]["duration_ms_mean"],
                         "hbm_GB": (2.0 * c["FETCH_SIZE"] + w[name]["WRITE_SIZE"]) * 1024 / 1e9}
json.dump({"command": cmd, "config": cfg, "packed_sym": sym, "kernel_source_sha256": bench.kernel_source_sha(),
           "units": "KiB per dispatch (rocprofv3 FETCH_SIZE/WRITE_SIZE); HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 -- FETCH_SIZE "
                    "counts half of a wide coalesced read stream on gfx950 (MI355X_MICROARCH.md, HBM section)",
           "workload": {"batch": B, "series": N, "factors": K, "T": T, "missing": missing, "mode": mode},
           "kernels": kernels}, open(out, "w"), indent=1)
print(out, {k[:50]: round(v["hbm_GB"], 3) for k, v in kernels.items()})
