import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# a test that calls __graft_entry__.build() must not spend minutes prebuilding the shape grid (the driver's build() does that)
os.environ.setdefault("METRAN_BUILD_SHAPES", "none")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The wide filter's default ("auto") keeps the split-layout kernels for batches of more than two models per SIMD; the
    # test tier's batches are small, so its engines start with the split layout forced -- the kernels the large batches run.
    # tests/test_hip_layouts.py::test_wide_filter_auto_rule checks the default rule itself.
    from metran_amd.engine import BatchedKalman

    BatchedKalman.default_variants = {"wide_filter": "split"}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def golden_models(name):
    """Yield per-model dicts of a synthetic golden file written by make_golden.synthetic_case."""
    g = load_golden(name)
    for i in range(int(g["nmodels"])):
        pre = f"m{i}_"
        yield i, {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}


SYNTH_GOLDENS = ["c2_small.npz", "c2_T1000.npz", "c4_missing.npz", "c4_T400.npz", "edge_cases.npz", "n17_k3.npz"]


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_real.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_seeded.npz")


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
