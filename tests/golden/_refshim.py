"""Import shim that lets the *unmodified* reference package at /root/reference
run in this container (no pastas, no numba, NumPy 2).

TEST INFRASTRUCTURE ONLY.  Used by ``make_golden.py`` (golden-vector generation,
run once in the build container) and by the optional ``test_reference_live.py``
differential tests, which skip when /root/reference is absent (i.e. on the GPU
box).  Nothing in ``metran_amd`` imports this.

Recipe: stub the ~10 ``pastas`` symbols the reference
imports (none of them does arithmetic on the Kalman path), make ``njit`` an
identity decorator so ``seqkalmanfilter`` (metran/kalmanfilter.py:236-400)
executes as plain Python with IEEE fp64 semantics, and restore the two NumPy-1
aliases the reference still uses (``np.int`` kalmanfilter.py:99, ``np.NaN``
solver.py:257).
"""
import logging
import os
import sys
import types

import numpy as np

_STAGED = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref")


def _default_root():
    """The reference itself where it is mounted (build container), else the verbatim copy that
    ``oracle/make_ref.sh`` staged into the git-ignored ``oracle/_ref`` (travels to the GPU box)."""
    if os.path.isdir("/root/reference/metran"):
        return "/root/reference"
    return _STAGED


REFERENCE_ROOT = os.environ.get("METRAN_REFERENCE_ROOT") or _default_root()


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "metran"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Install stubs and return the imported reference ``metran`` module."""
    if "metran" in sys.modules and getattr(sys.modules["metran"], "_shimmed", False):
        return sys.modules["metran"]
    if not reference_available():
        raise ImportError("reference not present at %s" % REFERENCE_ROOT)

    if not hasattr(np, "NaN"):
        np.NaN = np.nan
    if not hasattr(np, "int"):
        np.int = int

    def njit(*args, **kwargs):
        # used as @njit("signature") in the reference -> identity decorator
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    def initialize_logger(logger=None, level=logging.INFO):
        return None

    def validate_name(name, raise_error=False):
        return str(name)

    class TimeSeries:  # only used in isinstance() checks
        pass

    pastas = _mod("pastas", __version__="1.4.0")
    pastas.decorators = _mod("pastas.decorators", njit=njit)
    pastas.utils = _mod(
        "pastas.utils", initialize_logger=initialize_logger, validate_name=validate_name
    )
    pastas.timeseries = _mod("pastas.timeseries", TimeSeries=TimeSeries)
    pastas.version = _mod("pastas.version", __version__="1.4.0")
    pastas.timeseries_utils = _mod(
        "pastas.timeseries_utils", _frequency_is_supported=lambda f: f
    )
    pastas.plotting = _mod("pastas.plotting")
    pastas.plotting.plotutil = _mod(
        "pastas.plotting.plotutil", _get_height_ratios=lambda ylims: [1.0] * len(ylims)
    )
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import metran  # noqa: E402

    metran._shimmed = True
    return metran
