"""Run-time shape specialisation: kernels for an (N series, K factors) pair that is not in the
ahead-of-time list are compiled on demand from the SAME source (``csrc/mk_kernels.hip``) by hipcc,
statically checked for the hazards hipcc does not pad around inline asm (``scripts/check_dpp_hazards.py``: the fused
``v_fmac_f64_dpp``; ``scripts/check_asm_hazards.py``: MFMA results, MFMA operands, lane selects and transcendental
results with one side inside an asm block) and registered with the library (``mk_register_shape_module``).
Built modules are cached by a hash of the sources under ``$METRAN_HIP_CACHE`` (default
``~/.cache/metran_amd``).  Set ``METRAN_HIP_JIT=0`` to forbid compilation (unsupported shapes raise).
"""
import hashlib
import logging
import os
import shutil
import subprocess
import sys
import tempfile

from . import _lib
from ._lib import MetranHipError

logger = logging.getLogger(__name__)
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# every file a shape module is compiled from, directly or through an #include (mk_prims.h includes the GENERATED
# mk_sweeps.h: regenerating the sweeps must invalidate cached modules too)
_SOURCES = [os.path.join(_HERE, "csrc", "mk_kernels.hip"), os.path.join(_HERE, "csrc", "mk_wide.hip"),
            os.path.join(_HERE, "csrc", "mk_split.hip"), os.path.join(_HERE, "csrc", "mk_dk.hip")] + sorted(
    os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith(".h")) + [
    os.path.join(_ROOT, "include", "metran_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise MetranHipError("hipcc not found: cannot build a kernel module for this model shape")


def cache_dir():
    d = os.environ.get("METRAN_HIP_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "metran_amd")
    os.makedirs(d, exist_ok=True)
    return d


def _extra_flags():
    """Extra hipcc flags for the shape modules (``METRAN_HIP_JIT_FLAGS``, build-time A/B measurements such as
    ``-DMK_NO_TILED_SWEEPS``); part of the cache key."""
    return os.environ.get("METRAN_HIP_JIT_FLAGS", "").split()


def _source_hash():
    h = hashlib.sha256()
    for f in _SOURCES:
        h.update(open(f, "rb").read())
    h.update(" ".join(_extra_flags()).encode())
    return h.hexdigest()[:16]


def module_path(N, K):
    return os.path.join(cache_dir(), "mk_shape_%d_%d_%s.so" % (N, K, _source_hash()))


def build_shape_module(N, K):
    """Compile, hazard-check and cache the shape module for (N, K); returns its path."""
    out = module_path(N, K)
    if os.path.exists(out):
        return out
    if os.environ.get("METRAN_HIP_JIT", "1") == "0":
        raise MetranHipError("no kernel for (N=%d, K=%d) and METRAN_HIP_JIT=0 forbids building one" % (N, K))
    n = N + K
    if N < 1 or K < 1 or n > 64:
        raise MetranHipError("unsupported model shape N=%d, K=%d (need N, K >= 1 and N + K <= 64)" % (N, K))
    logger.info("building HIP kernels for a %d-series / %d-factor model (one-off, cached)", N, K)
    tmp = tempfile.mkdtemp(prefix="mkjit_")
    try:
        obj = os.path.join(tmp, "mod.o")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
               "-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_HERE, "csrc"), "-DMK_SHAPE_MODULE",
               "-DMK_SHAPES(X)=X(%d,%d)" % (N, K), "-save-temps=obj", "-Wno-unused-command-line-argument"] + _extra_flags() + [
               "-c", _SOURCES[0], "-o", obj]
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise MetranHipError("hipcc failed for shape (%d,%d):\n%s" % (N, K, r.stderr[-2000:]))
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        for script, what in (("check_dpp_hazards.py", "DPP"), ("check_asm_hazards.py", "inline-asm")):
            checker = os.path.join(_ROOT, "scripts", script)
            if asm and os.path.exists(checker):
                c = subprocess.run([sys.executable, checker, os.path.join(tmp, asm[0])], capture_output=True, text=True)
                if c.returncode != 0:
                    raise MetranHipError("%s hazard check failed for shape (%d,%d):\n%s" % (what, N, K, c.stdout[-2000:]))
        so = os.path.join(tmp, "mod.so")
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise MetranHipError("link failed for shape (%d,%d):\n%s" % (N, K, r.stderr[-2000:]))
        # publish atomically: every rank of a multi-GPU job builds the same shape at the same moment, and a
        # reader must never dlopen a half-written file (temp file INSIDE the cache directory + rename)
        fd, stage = tempfile.mkstemp(prefix=".mk_shape_", suffix=".so", dir=cache_dir())
        os.close(fd)
        shutil.copyfile(so, stage)
        os.chmod(stage, 0o755)
        os.replace(stage, out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def ensure_shape(N, K):
    """Make kernels for (N, K) available in the loaded library (no-op for ahead-of-time shapes)."""
    L = _lib.lib()
    if L.mk_shape_supported(N, K):
        return False
    path = build_shape_module(int(N), int(K))
    _lib.check(L.mk_register_shape_module(path.encode()))
    if not L.mk_shape_supported(N, K):
        raise MetranHipError("shape module %s did not register (N=%d, K=%d)" % (path, N, K))
    return True
