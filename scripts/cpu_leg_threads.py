#!/usr/bin/env python
"""The optimised CPU leg (oracle/kalman_fast.c) per OpenMP thread count on this host, with what the host says about its CPUs
(cores, affinity, cgroup quota): a box whose quota is below its core count runs 128 threads slower than 32."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from metran_amd.synthetic import make_dfm_batch

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|MHz' | head -8", shell=True, capture_output=True, text=True).stdout)
try:
    oracle.build(native=True); native = True
except Exception:
    native = False
B = 2048
d = make_dfm_batch(B, 8, 2, 1000, seed=1)
out = oracle.fast_dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"], native=native)
for t in (8, 16, 32, 64, 96, 128, 192, 256):
    if t > 2 * (os.cpu_count() or 1):
        break
    oracle.fast_set_num_threads(t, native)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); oracle.fast_dfm_batch(d["obs"], d["phi"], d["q"], d["loadings"], native=native, out=out); best = min(best, time.perf_counter() - t0)
    print("threads %3d  %.0f models/s  (%.1f per thread)" % (t, B / best, B / best / t), flush=True)
