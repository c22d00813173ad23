#!/usr/bin/env python
"""cProfile of Metran.solve() through the installed HIP engine on examples/data (where the host time of one get_mle goes)."""
import cProfile
import glob
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import pandas as pd

import _refshim

metran = _refshim.install()
import metran_amd.kalmanfilter as hip

files = sorted(glob.glob(os.path.join(_refshim.REFERENCE_ROOT, "examples", "data", "*_res.csv")))
series = []
for f in files:
    x = pd.read_csv(f, header=0, index_col=0, parse_dates=True).squeeze()
    x.name = os.path.basename(f).split("_")[0]
    series.append(x)
hip.install(metran)
mt = metran.Metran(series, name="B21B0214")
mt.solve(report=False)
for rep in range(2):
    mt = metran.Metran(series, name="B21B0214")
    t0 = time.perf_counter()
    mt.solve(report=False)
    print("solve_s %.4f nfev %d" % (time.perf_counter() - t0, mt.fit.nfev))
mt = metran.Metran(series, name="B21B0214")
pr = cProfile.Profile()
pr.enable()
mt.solve(report=False)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
