"""Batched MI355X engine: thousands of independent Metran dynamic-factor models per launch.

Host-side plumbing only (device memory and streams come from PyTorch-ROCm); every
number is produced by the HIP kernels behind the C ABI (``include/metran_hip.h``).

Mirrors, batched over a leading axis, the reference call chain
``SPKalmanFilter.set_observations -> set_matrices -> run_filter / run_smoother -> get_mle /
simulate / decompose`` (/root/reference/metran/kalmanfilter.py:520-778) and
``Metran._get_matrices`` (metran/metran.py:386-416).
"""
import ctypes
import logging

import numpy as np

from . import _lib
from ._lib import MetranHipError, Outputs, Problem, check

logger = logging.getLogger(__name__)

__all__ = ["BatchedKalman", "MetranHipError", "FLAG_NONPOSITIVE_F", "FLAG_NOT_SPD", "FLAG_RANK_DEFICIENT"]

FLAG_NONPOSITIVE_F = 1
FLAG_NOT_SPD = 2
FLAG_RANK_DEFICIENT = 4  # informational: a null direction of Pp was dropped, as the reference's pinv does

_STATE_OUTPUTS = ("F", "Pf", "Xp", "Pp", "S", "Ps")


def _torch():
    import torch

    return torch


class BatchedKalman:
    """One context per (process, GPU).

    Parameters
    ----------
    device : int or torch.device, optional
        CUDA/HIP device index; default ``torch.cuda.current_device()``.

    Typical use::

        kf = BatchedKalman()
        kf.set_observations(obs)            # [R,T,N], NaN = missing; uploaded once
        kf.set_loadings(loadings)           # [R,N,K]
        mle = kf.loglik(phi, q)             # [B] -2 log L, B = k*R instances
        out = kf.filter_smooth(phi, q)      # dict of device tensors F,Pf,Xp,Pp,S,Ps,mle,...
    """

    def __init__(self, device=None, layout="model_major", packed_sym=False):
        """layout: "model_major" -- per-step arrays are ``[B,T,...]`` contiguous (the reference's
        per-model arrays stacked); "time_major" -- the memory is ``[T,B,...]`` (one time step of all
        models contiguous: every wavefront's stores land next to its neighbours', which is what HBM
        wants) and the tensors handed back are ``[B,T,...]`` *views* of it, so indexing is unchanged."""
        torch = _torch()
        if layout not in ("model_major", "time_major"):
            raise ValueError("layout must be 'model_major' or 'time_major'")
        self.time_major = layout == "time_major"
        self.packed_sym = bool(packed_sym)
        # simulate_smoothed of wide models: "auto" = the inverse-free tape path where it applies (MK_OUT_TAPE: 16 < n
        # <= 63), else filtered records + RTS smoother; "tape" insists, "records" never uses it
        self.projection_path = "auto"
        L = _lib.lib()  # raises MetranHipError when the HIP library is not built
        if not torch.cuda.is_available():
            raise MetranHipError("no GPU visible to PyTorch-ROCm; metran_amd has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self._L = L
        ctx = ctypes.c_void_p()
        check(L.mk_create(self.device.index, ctypes.byref(ctx)))
        self._ctx = ctx
        self.obs = None
        self.loadings = None
        self.obsvar = None
        self.scale = None
        self.offset = None
        self.R = self.T = self.N = self.K = None
        self._timing = False
        for which, name in type(self).default_variants.items():
            self.set_variant(which, name)

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_ctx", None):
            self._L.mk_destroy(self._ctx)
            self._ctx = None
            self._has_comm = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ kernel variants (A/B measurements)
    _VARIANTS = {"smoother16": (0, ("record", "blk")),
                 "wide_smoother": (1, ("mfma", "v1", "mfma_unfolded")),
                 "wide_filter": (2, ("auto", "lane_per_state", "split")),
                 "single_record": (3, ("sparse", "stepwise")),
                 "kernel_family": (4, ("specialised", "generic")),
                 "tape_filter": (5, ("observable", "state"))}
    # variants every new engine starts with (name -> value); empty = the library's defaults.  The GPU test tier sets
    # {"wide_filter": "split"} so that its small batches keep exercising the split kernels, which "auto" reserves for
    # batches of more than two models per SIMD (tests/conftest.py; tests/test_hip_layouts.py checks "auto" itself)
    default_variants = {}

    def set_variant(self, which, name):
        """Choose between two equivalent kernels of a shape class (``mk_set_kernel_variant``): ``"smoother16"``:
        ``"record"`` (default) | ``"blk"``; ``"wide_smoother"``: ``"mfma"`` (default) | ``"v1"`` | ``"mfma_unfolded"``; ``"wide_filter"``: ``"auto"`` (default: the split
        layout for more than two models per SIMD, one state per lane below) | ``"lane_per_state"`` | ``"split"``; ``"tape_filter"``
        (the writer of the backward tape, N <= 32): ``"observable"`` (default: the filter in the observable basis) | ``"state"``.
        Every member is tested against the oracle; there is no environment switch."""
        sel, names = self._VARIANTS[which]
        check(self._L.mk_set_kernel_variant(self._ctx, sel, names.index(name)))
        return self

    def resolved_wide_filter(self, B):
        """The wide-filter kernel ``"auto"`` picks for B instances (``mk_kernels.hip`` dispatch_filter: the split layout above
        two wavefronts per SIMD, one state per lane below)."""
        v = self.get_variant("wide_filter")
        if v != "auto":
            return v
        torch = _torch()
        simds = 4 * torch.cuda.get_device_properties(self.device).multi_processor_count
        return "split" if B > 2 * simds else "lane_per_state"

    def get_variant(self, which):
        sel, names = self._VARIANTS[which]
        v = ctypes.c_int(-1)
        check(self._L.mk_get_kernel_variant(self._ctx, sel, ctypes.byref(v)))
        return names[v.value]

    # ------------------------------------------------------------------ helpers
    def _dev(self, a, shape=None, name="array"):
        """float64 contiguous tensor on this device (accepts numpy / torch)."""
        torch = _torch()
        if a is None:
            return None
        if not isinstance(a, torch.Tensor):
            a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        a = a.to(device=self.device, dtype=torch.float64).contiguous()
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError("%s has shape %s, expected %s" % (name, tuple(a.shape), tuple(shape)))
        return a

    def _bind_stream(self):
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self._L.mk_set_stream(self._ctx, ctypes.c_void_p(s)))

    @staticmethod
    def _p(t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    # ------------------------------------------------------------------ inputs
    def set_observations(self, obs):
        """Upload observation records ``[R,T,N]`` (or ``[T,N]``); NaN/inf = missing
        (reference packing: metran/kalmanfilter.py:646-674)."""
        torch = _torch()
        if not isinstance(obs, torch.Tensor):
            obs = np.asarray(obs, dtype=np.float64)
        if obs.ndim == 2:
            obs = obs[None]
        if obs.ndim != 3:
            raise ValueError("observations must be [R,T,N] or [T,N]")
        obs = self._dev(obs)
        # the reference finds valid entries with "(x + 1e10).nonzero()" (kalmanfilter.py:666-667): a finite
        # observation of exactly -1e10 is dropped there, so it is a missing value here too
        obs = torch.where(obs == -1e10, torch.full_like(obs, float("nan")), obs)
        self.obs = self._layout(obs)
        self._obs_unmasked = None
        self._grad_pending = self._grad_alpha = None  # a pending forward pass belongs to the previous dataset
        self.R, self.T, self.N = (int(s) for s in self.obs.shape)
        check(self._L.mk_observations_changed(self._ctx))  # a recycled allocation may carry a new record at an old address
        return self

    # ------------------------------------------------------------------ ingestion
    def standardize(self, obs=None):
        """``Metran.standardize`` (metran/metran.py:102-121) for every record on the device: per series,
        subtract the mean and divide by the standard deviation (NaN skipped, ddof = 1).  ``obs`` are raw
        series ``[R,T,N]`` (default: the records already set).  The standardised records become the
        engine's observations and ``(std, mean)`` its scaling (``set_scaling``), so that
        ``simulate_smoothed`` returns values in the original units; returns ``(mean, std)`` ``[R,N]``."""
        torch = _torch()
        if obs is not None:
            self.set_observations(obs)
        if self.obs is None:
            raise MetranHipError("call set_observations first")
        if self.N > 64:
            raise MetranHipError("standardize supports N <= 64 series per model")
        mean = torch.empty((self.R, self.N), dtype=torch.float64, device=self.device)
        std = torch.empty_like(mean)
        self._bind_stream()
        check(self._L.mk_standardize(self._ctx, self.R, self.T, self.N, int(self.time_major), self._p(self.obs),
                                     self._p(self.obs), self._p(mean), self._p(std)))
        self._obs_unmasked = None
        self.scale, self.offset = std, mean
        check(self._L.mk_observations_changed(self._ctx))  # standardised in place
        return mean, std

    def mask_observations(self, mask):
        """``Metran.mask_observations`` (metran/metran.py:464-494): hide the observations where ``mask``
        ``[R,T,N]`` is non-zero.  The unmasked records stay on the device; ``unmask_observations`` is free."""
        torch = _torch()
        if self.obs is None:
            raise MetranHipError("call set_observations first")
        base = self._obs_unmasked if getattr(self, "_obs_unmasked", None) is not None else self.obs
        if not isinstance(mask, torch.Tensor):
            mask = torch.from_numpy(np.ascontiguousarray(np.asarray(mask) != 0).view(np.uint8))
        mask = (mask != 0).to(device=self.device, dtype=torch.uint8)
        if mask.ndim == 2:
            mask = mask[None]
        if tuple(mask.shape) != (self.R, self.T, self.N):
            raise ValueError("Dimensions of mask %s do not equal dimensions of series %s"
                             % (tuple(mask.shape), (self.R, self.T, self.N)))  # metran.py:484-491
        mask = self._layout(mask)
        out = torch.empty_like(base)
        self._bind_stream()
        check(self._L.mk_mask_observations(self._ctx, int(base.numel()), self._p(base), self._p(mask), self._p(out)))
        self._obs_unmasked = base
        self.obs = out
        check(self._L.mk_observations_changed(self._ctx))
        return self

    def unmask_observations(self):
        """``Metran.unmask_observations`` (metran/metran.py:496-506)."""
        if getattr(self, "_obs_unmasked", None) is not None:
            self.obs = self._obs_unmasked
            self._obs_unmasked = None
            check(self._L.mk_observations_changed(self._ctx))
        return self

    def pack_observations(self):
        """``SPKalmanFilter.set_observations`` (metran/kalmanfilter.py:646-674) for every record:
        ``(observations [R,T,N], observation_indices [R,T,N] float64, observation_count [R,T] int64)``."""
        torch = _torch()
        if self.obs is None:
            raise MetranHipError("call set_observations first")
        observations = torch.empty_like(self.obs)
        indices = torch.empty_like(self.obs)
        count = self._empty_bt(self.R, self.T, dtype=torch.int64)
        # the kernel walks the R*T rows in memory order, whichever layout that is
        self._bind_stream()
        check(self._L.mk_pack_observations(self._ctx, self.R * self.T, 1, self.N, self._p(self.obs),
                                           self._p(observations), self._p(indices), self._p(count)))
        return observations, indices, count

    def _layout(self, t):
        """Logical ``[B,T,...]`` tensor whose memory follows the engine's layout."""
        if not self.time_major:
            return t.contiguous()
        if t.transpose(0, 1).is_contiguous():
            return t
        return t.transpose(0, 1).contiguous().transpose(0, 1)

    def _empty_bt(self, B, T, *rest, dtype=None):
        torch = _torch()
        dtype = dtype or torch.float64
        if self.time_major:
            return torch.empty((T, B) + tuple(rest), dtype=dtype, device=self.device).transpose(0, 1)
        return torch.empty((B, T) + tuple(rest), dtype=dtype, device=self.device)

    def set_loadings(self, loadings, obsvar=None):
        """Factor loadings ``[R,N,K]`` (``Z = [I | loadings]``, metran/metran.py:365-370) and
        optional observation variances ``[R,N]`` (zeros in Metran, metran/metran.py:382-384)."""
        torch = _torch()
        if self.obs is None:
            raise MetranHipError("call set_observations first")
        if not isinstance(loadings, torch.Tensor):
            loadings = np.asarray(loadings, dtype=np.float64)
        if loadings.ndim == 2:
            loadings = loadings[None]
        self.loadings = self._dev(loadings)
        self._grad_pending = self._grad_alpha = None  # ... and to the previous loadings
        if tuple(self.loadings.shape[:2]) != (self.R, self.N):
            raise ValueError("loadings must be [R=%d,N=%d,K], got %s" % (self.R, self.N, tuple(self.loadings.shape)))
        self.K = int(self.loadings.shape[2])
        self._ensure_kernels()
        if obsvar is not None:
            if not isinstance(obsvar, torch.Tensor):
                obsvar = np.asarray(obsvar, dtype=np.float64)
            if obsvar.ndim == 1:
                obsvar = obsvar[None]
            self.obsvar = self._dev(obsvar, (self.R, self.N), "obsvar")
        else:
            self.obsvar = None
        return self

    def _ensure_kernels(self):
        """Kernels for this engine's (N, K).  A shape of the ahead-of-time list runs as it is; any other shape with
        N + K <= 64 gets its own specialised module (``metran_amd/jit.py``: prebuilt under ``metran_amd/_shape_cache``, cached
        per user, or compiled now by hipcc); where that is not possible -- no hipcc on the machine, ``METRAN_HIP_JIT=0``,
        or N + K > 64 -- the size-generic kernels serve the shape (``mk_generic.hip``, N + K <= 128: correct for every
        shape, an order of magnitude slower; no packed-symmetric records, tape or adjoint gradient)."""
        if self._L.mk_shape_specialised(self.N, self.K):
            return
        why = None
        if self.n <= 64:
            from . import jit

            try:
                jit.ensure_shape(self.N, self.K)  # builds + registers a specialised kernel module (cached)
                return
            except jit.ShapeUnavailable as e:   # no compiler / METRAN_HIP_JIT=0: the size-generic kernels serve the shape.
                why = str(e).splitlines()[0]    # A FAILED build (hipcc, link, hazard check: jit.ShapeBuildError) propagates --
                                                # it must not hide behind kernels ten times slower (round-5 advice)
        if not self._L.mk_shape_supported(self.N, self.K):
            raise MetranHipError("a model of N=%d series and K=%d factors has %d states; the library serves N + K <= %d"
                                 % (self.N, self.K, self.n, int(self._L.mk_generic_max_states())))
        if self.packed_sym:
            raise MetranHipError("packed-symmetric records need specialised kernels; (N=%d, K=%d) runs the size-generic ones%s"
                                 % (self.N, self.K, " (%s)" % why if why else ""))
        logger.warning("(N=%d, K=%d) runs the size-generic kernels%s -- correct, about an order of magnitude slower, no adjoint "
                       "gradient / tape / packed-symmetric records", self.N, self.K, ": " + why if why else " (N + K > 64)")

    def specialised(self):
        """True when this engine's shape runs the specialised (unrolled, one-state-per-lane) kernels, False for the size-generic ones."""
        return bool(self._L.mk_shape_specialised(self.N, self.K))

    def subset(self, index):
        """A new engine holding the records ``index`` (1-D integer tensor / array of record numbers) of this one: the
        observations, loadings and observation variances are gathered on the device (what ``calibrate_batch`` does when
        most of its models have converged: the remaining iterations run on the still-active records only)."""
        torch = _torch()
        index = torch.as_tensor(index, dtype=torch.long, device=self.device)
        sub = BatchedKalman(self.device.index, layout="time_major" if self.time_major else "model_major",
                            packed_sym=self.packed_sym)
        # the sub-engine runs the SAME kernels as its parent: the user's variants and projection path carry over, and
        # "auto" (a rule on the batch size) is pinned to what it resolves to for the parent's record count -- so a model
        # sees bit-identical results whether or not calibrate_batch compacted the flight around it
        sub.projection_path = self.projection_path
        for which in self._VARIANTS:
            sub.set_variant(which, self.get_variant(which))
        if self.get_variant("wide_filter") == "auto" and self.R is not None:
            sub.set_variant("wide_filter", self.resolved_wide_filter(self.R))
        sub.set_observations(self.obs[index])
        sub.set_loadings(self.loadings[index], None if self.obsvar is None else self.obsvar[index])
        if self.scale is not None:
            sub.scale, sub.offset = self.scale[index].contiguous(), self.offset[index].contiguous()
        # the sub-engine takes over the parent's adjoint workspace (a subset never needs more of it; at 8192 x (8,2), T = 1000 it is
        # 7.3 GB, and a fresh 3 GB allocation for the first compacted flight was the largest single item of a calibration's wall
        # time).  The two are used one after the other (calibrate_batch returns to the parent when the flight has finished); a
        # forward pass still waiting for its backward pass on the parent is void from here on.  Interleaved use is not supported:
        # The buffer carries an OWNER token shared by everybody who holds it (round-5 advice): whoever starts a forward pass
        # takes the token, and a backward pass whose engine no longer holds it raises instead of walking somebody else's records.
        work = getattr(self, "_grad_work", None)
        if work is not None:
            sub._grad_work = work
            sub._grad_share = self._grad_share = getattr(self, "_grad_share", None) or {"owner": None}
            self._grad_pending = None
        sub.adjoint_updates = self.adjoint_updates
        if getattr(self, "_grad_upd", None) is not None:   # ... and so does the update tape of the wide adjoint gradient
            sub._grad_upd = self._grad_upd
        return sub

    @property
    def n(self):
        return self.N + self.K

    def params_from_alpha(self, alpha, dt=1.0):
        """``alpha [B,n] -> (phi, q)`` on device (Metran._get_matrices diagonals, metran.py:246-322)."""
        torch = _torch()
        alpha = self._dev(alpha)
        if alpha.ndim == 1:
            alpha = alpha[None]
        B = int(alpha.shape[0])
        if alpha.shape[1] != self.n:
            raise ValueError("alpha must be [B,%d]" % self.n)
        phi = torch.empty_like(alpha)
        q = torch.empty_like(alpha)
        self._bind_stream()
        check(self._L.mk_params_from_alpha(self._ctx, B, self.R, self.N, self.K, self._p(alpha),
                                           self._p(self.loadings), float(dt), self._p(phi), self._p(q)))
        return phi, q

    def _problem(self, phi, q, warmup, x0, P0):
        if self.obs is None or self.loadings is None:
            raise MetranHipError("set_observations and set_loadings must be called before running the filter")
        phi = self._dev(phi)
        q = self._dev(q)
        if phi.ndim == 1:
            phi = phi[None]
        if q.ndim == 1:
            q = q[None]
        B = int(phi.shape[0])
        if tuple(phi.shape) != (B, self.n) or tuple(q.shape) != (B, self.n):
            raise ValueError("phi and q must be [B,%d]; got %s, %s" % (self.n, tuple(phi.shape), tuple(q.shape)))
        if B % self.R != 0:
            raise ValueError("number of instances B=%d must be a multiple of the number of records R=%d" % (B, self.R))
        x0 = self._dev(x0, (B, self.n), "x0") if x0 is not None else None
        P0 = self._dev(P0, (B, self.n, self.n), "P0") if P0 is not None else None
        prob = Problem(B, self.R, self.T, self.N, self.K, int(warmup), self._p(self.obs), self._p(phi), self._p(q),
                       self._p(self.loadings), self._p(self.obsvar), self._p(x0), self._p(P0),
                       1 if self.time_major else 0, self._p(self.scale), self._p(self.offset))
        keep = (phi, q, x0, P0)
        return prob, keep, B

    # ------------------------------------------------------------------ hot path
    def loglik(self, phi, q, warmup=1, x0=None, P0=None, out=None):
        """-2 log L per instance (``Metran.get_mle`` objective, metran/metran.py:605-622)."""
        torch = _torch()
        prob, keep, B = self._problem(phi, q, warmup, x0, P0)
        mle = out if out is not None else torch.empty(B, dtype=torch.float64, device=self.device)
        self._bind_stream()
        check(self._L.mk_loglik(self._ctx, ctypes.byref(prob), self._p(mle)))
        return mle

    def loglik_grad(self, phi, q, warmup=1, x0=None, P0=None):
        """Objective and its ADJOINT gradient: ``(mle [B], gphi [B,n], gq [B,n])`` with
        ``gphi = d(-2 log L)/d phi``, ``gq = d(-2 log L)/d q`` in two launches (forward filter writing the
        filtered records into a workspace that is kept between calls, then the backward adjoint kernel).
        The reference has no gradient (metran/solver.py:248-255 leaves scipy to difference P+1 runs).
        Any supported shape: four models per wavefront for n <= 16, one per wavefront for 16 < n <= 64."""
        torch = _torch()
        prob, keep, B = self._problem(phi, q, warmup, x0, P0)
        self._grad_pending = None  # the shared workspace is about to be overwritten: a forward pass waiting for its backward is void
        need = B * self.T * self.record_stride()
        work = self._take_grad_work(need)
        self._ensure_grad_updates(B)
        mle = torch.empty(B, dtype=torch.float64, device=self.device)
        sc = torch.empty(B, dtype=torch.int64, device=self.device)
        gphi = torch.empty((B, self.n), dtype=torch.float64, device=self.device)
        gq = torch.empty_like(gphi)
        self._bind_stream()
        check(self._L.mk_loglik_grad(self._ctx, ctypes.byref(prob), self._p(work), 1 if self.time_major else 0,
                                     self._p(mle), self._p(sc), self._p(gphi), self._p(gq), None))
        return mle, gphi, gq

    # the update tape of the wide adjoint gradient (C ABI mk_set_adjoint_updates; round 6): True = allocate and use it where the
    # shape has one (16 < N + K <= 64) and it fits in 40 % of the free memory, False = the backward walk recomputes every step
    adjoint_updates = True

    def _ensure_grad_updates(self, B):
        torch = _torch()
        us = int(self._L.mk_adjoint_update_stride(self.N, self.K)) if self.adjoint_updates else 0
        buf = getattr(self, "_grad_upd", None)
        need = B * self.T * us
        if us and (buf is None or buf.numel() < need):
            free = float(torch.cuda.mem_get_info(self.device)[0])
            buf = torch.empty(need, dtype=torch.float64, device=self.device) if 8.0 * need <= 0.4 * free else None
            self._grad_upd = buf
            self._grad_upd_set = None
        if not us:
            buf = None
        key = None if buf is None else (buf.data_ptr(), buf.numel())
        if getattr(self, "_grad_upd_set", "unset") != key:   # (the context keeps the pointer: tell it only when it changes)
            check(self._L.mk_set_adjoint_updates(self._ctx, self._p(buf), 0 if buf is None else int(buf.numel())))
            self._grad_upd_set = key

    def _take_grad_work(self, need):
        """The adjoint workspace (at least ``need`` doubles), with this engine as the owner of its contents from now on: an
        engine sharing the buffer (``subset``) that still waits for a backward pass will refuse to run it."""
        torch = _torch()
        work = getattr(self, "_grad_work", None)
        if work is None or work.numel() < need:
            self._grad_work = work = torch.empty(need, dtype=torch.float64, device=self.device)
            self._grad_share = None          # a fresh buffer is nobody else's
        self._grad_token = object()
        if getattr(self, "_grad_share", None) is not None:
            self._grad_share["owner"] = self._grad_token
        return work

    def loglik_forward(self, phi, q, warmup=1, x0=None, P0=None):
        """The forward half of ``loglik_grad`` alone (``mk_loglik_grad_phases``, MK_GRAD_FORWARD): ``mle [B]``, with the
        filtered records left in the workspace.  ``loglik_backward()`` then returns the gradient AT THESE parameters
        without filtering them again -- a line search evaluates several trial points and wants the gradient of the last
        one only."""
        torch = _torch()
        prob, keep, B = self._problem(phi, q, warmup, x0, P0)
        need = B * self.T * self.record_stride()
        work = self._take_grad_work(need)
        self._ensure_grad_updates(B)
        mle = torch.empty(B, dtype=torch.float64, device=self.device)
        sc = torch.empty(B, dtype=torch.int64, device=self.device)
        self._bind_stream()
        check(self._L.mk_loglik_grad_phases(self._ctx, ctypes.byref(prob), self._p(work), 1 if self.time_major else 0,
                                            self._p(mle), self._p(sc), None, None, None, 1))
        # everything the backward launch dereferences stays alive with the pending record (the parameter tensors in `keep`,
        # the observation / loadings / variance tensors the problem struct points at); set_observations, set_loadings and
        # loglik_grad void it (round-3 advice: a backward pass over records of another parameter set or dataset returned a
        # wrong gradient without an error)
        self._grad_pending = (prob, keep + (self.obs, self.loadings, self.obsvar), B, sc, work)
        self._grad_alpha = None
        return mle

    def loglik_backward(self):
        """``(gphi [B,n], gq [B,n])`` at the parameters of the last ``loglik_forward`` (MK_GRAD_BACKWARD)."""
        torch = _torch()
        pending = getattr(self, "_grad_pending", None)
        if pending is None:
            raise MetranHipError("loglik_backward without a preceding loglik_forward")
        share = getattr(self, "_grad_share", None)
        if share is not None and share["owner"] is not self._grad_token:
            self._grad_pending = None
            raise MetranHipError("the records of this engine's forward pass have been overwritten: the adjoint workspace is shared with "
                                 "another engine (BatchedKalman.subset) that ran a forward pass since -- use the two one after the other")
        prob, keep, B, sc, work = pending
        gphi = torch.empty((B, self.n), dtype=torch.float64, device=self.device)
        gq = torch.empty_like(gphi)
        self._bind_stream()
        check(self._L.mk_loglik_grad_phases(self._ctx, ctypes.byref(prob), self._p(work), 1 if self.time_major else 0,
                                            None, self._p(sc), self._p(gphi), self._p(gq), None, 2))
        self._grad_pending = None  # one backward pass per forward pass
        return gphi, gq

    def loglik_forward_alpha(self, alpha, dt=1.0, warmup=1):
        """``loglik_forward`` in Metran's parametrisation; ``loglik_backward_alpha()`` completes it."""
        alpha = self._dev(alpha)
        if alpha.ndim == 1:
            alpha = alpha[None]
        phi, q = self.params_from_alpha(alpha, dt=dt)
        mle = self.loglik_forward(phi, q, warmup=warmup)
        self._grad_alpha = (alpha, float(dt))
        return mle

    def loglik_backward_alpha(self):
        """``d mle / d alpha [B,n]`` at the point of the last ``loglik_forward_alpha``."""
        torch = _torch()
        if getattr(self, "_grad_alpha", None) is None:
            raise MetranHipError("loglik_backward_alpha without a preceding loglik_forward_alpha")
        alpha, dt = self._grad_alpha
        gphi, gq = self.loglik_backward()
        self._grad_alpha = None
        galpha = torch.empty_like(gphi)
        check(self._L.mk_alpha_grad(self._ctx, int(alpha.shape[0]), self.R, self.N, self.K, self._p(alpha),
                                    self._p(self.loadings), dt, self._p(gphi), self._p(gq), self._p(galpha)))
        return galpha

    def has_adjoint(self):
        """Whether ``loglik_grad`` (``mk_loglik_grad``) serves this engine's shape: ``adjoint_kernel`` for n <= 16 (four
        models per wavefront), ``adjoint_wide_kernel`` for 16 < n <= 64 (one model per wavefront)."""
        return self.N is not None and self.n <= 64 and self.specialised() and self.get_variant("kernel_family") == "specialised"

    def loglik_grad_alpha(self, alpha, dt=1.0, warmup=1):
        """``(mle [B], d mle / d alpha [B,n])`` for Metran's parametrisation (``params_from_alpha`` forward,
        ``mk_alpha_grad`` chain rule backward)."""
        torch = _torch()
        alpha = self._dev(alpha)
        if alpha.ndim == 1:
            alpha = alpha[None]
        phi, q = self.params_from_alpha(alpha, dt=dt)
        mle, gphi, gq = self.loglik_grad(phi, q, warmup=warmup)
        galpha = torch.empty_like(gphi)
        check(self._L.mk_alpha_grad(self._ctx, int(alpha.shape[0]), self.R, self.N, self.K, self._p(alpha),
                                    self._p(self.loadings), float(dt), self._p(gphi), self._p(gq), self._p(galpha)))
        return mle, galpha

    def record_stride(self):
        """Doubles per packed (model, step) record for this state dimension (C ABI ``mk_record_stride``, or
        ``mk_record_stride_sym`` for an engine created with ``packed_sym=True``)."""
        if self.packed_sym:
            return int(self._L.mk_record_stride_sym(self.n))
        return int(self._L.mk_record_stride(self.n))

    def unpack_sym(self, packed):
        """Packed upper triangle ``[..., n(n+1)/2]`` (what the covariance entries of a ``packed_sym`` engine's results
        are) -> full symmetric ``[..., n, n]``.  Lazy by design: the packed records are what moves through HBM."""
        torch = _torch()
        n = self.n
        idx = getattr(self, "_sym_index", None)
        if idx is None or idx.shape[0] != n:
            r = torch.arange(n, device=self.device)
            lo, hi = torch.minimum(r[:, None], r[None, :]), torch.maximum(r[:, None], r[None, :])
            idx = self._sym_index = lo * n - (lo * (lo - 1)) // 2 + (hi - lo)
        return packed[..., idx]

    def _alloc_records(self, B):
        """One record array ``[B,T,RS]`` (logical; memory follows the engine layout) and its views
        ``(mean [B,T,n], cov [B,T,n,n], pad0 [B,T], pad1 [B,T])``."""
        T, n = self.T, self.n
        RS = self.record_stride()
        rec = self._empty_bt(B, T, RS)
        if self.packed_sym:  # covariance entry = the packed upper triangle [B,T,n(n+1)/2]; see unpack_sym
            nv = n + n * (n + 1) // 2
            return rec, rec[..., :n], rec[..., n:nv], rec[..., nv], rec[..., nv + 1]
        nv = n + n * n
        return rec, rec[..., :n], rec[..., n:nv].unflatten(-1, (n, n)), rec[..., nv], rec[..., nv + 1]

    def _alloc_outputs(self, B, want, bookkeeping=True):
        """Output tensors for one launch.  When the predicted and filtered moments (and, if any, both
        smoothed moments) are all requested, they are allocated as PACKED RECORDS (the kernels' fast
        path, see ``mk_outputs.record_stride``) and the entries of the returned dict are views."""
        torch = _torch()
        T, n = self.T, self.n
        f64 = dict(dtype=torch.float64, device=self.device)
        want = list(want)
        for k in want:
            if k not in _STATE_OUTPUTS:
                raise ValueError("unknown output %r (choose from %s)" % (k, _STATE_OUTPUTS))
        res = {"mle": torch.empty(B, **f64)}
        res["status"] = torch.zeros(B, dtype=torch.int32, device=self.device)
        if bookkeeping:
            res["sigmacount"] = torch.empty(B, dtype=torch.int64, device=self.device)
        smooth_pair = ("S" in want) + ("Ps" in want)
        records = all(k in want for k in ("F", "Pf", "Xp", "Pp")) and smooth_pair in (0, 2)
        if self.packed_sym and not records:
            raise MetranHipError("a packed_sym engine writes record outputs only: ask for F, Pf, Xp, Pp (and S, Ps), "
                                 "or use simulate_smoothed / smooth_state_variances")
        if records:
            res["_rs"] = self.record_stride()
            res["_rec_pred"], res["Xp"], res["Pp"], _, _ = self._alloc_records(B)
            res["_rec_filt"], res["F"], res["Pf"], sig, det = self._alloc_records(B)
            if bookkeeping:
                res["sigmas"], res["detfs"] = sig, det
            if smooth_pair:
                res["_rec_smooth"], res["S"], res["Ps"], _, _ = self._alloc_records(B)
            return res
        if bookkeeping:
            res["sigmas"] = self._empty_bt(B, T)
            res["detfs"] = self._empty_bt(B, T)
        for k in want:
            res[k] = self._empty_bt(B, T, n) if k in ("F", "Xp", "S") else self._empty_bt(B, T, n, n)
        return res

    def _outputs_struct(self, res):
        g = res.get
        return Outputs(self._p(g("mle")), self._p(g("sigmas")), self._p(g("detfs")), self._p(g("sigmacount")),
                       self._p(g("F")), self._p(g("Pf")), self._p(g("Xp")), self._p(g("Pp")), self._p(g("S")),
                       self._p(g("Ps")), self._p(g("status")), 1 if self.time_major else 0,
                       self._p(g("sim_means")), self._p(g("sim_vars")), int(g("_rs", 0)),
                       (1 if (self.packed_sym and g("_rs", 0) and not g("_tape")) else 0) | (2 if g("_var_only") else 0)
                       | (4 if g("_tape") else 0))

    def filter(self, phi, q, warmup=1, x0=None, P0=None, outputs=("F", "Pf", "Xp", "Pp"), buffers=None):
        """``run_filter`` for B instances (kalmanfilter.py:696-778).  Returns a dict of device tensors."""
        prob, keep, B = self._problem(phi, q, warmup, x0, P0)
        res = buffers if buffers is not None else self._alloc_outputs(B, [o for o in outputs if o not in ("S", "Ps")])
        o = self._outputs_struct(res)
        self._bind_stream()
        check(self._L.mk_filter(self._ctx, ctypes.byref(prob), ctypes.byref(o)))
        return res

    def filter_smooth(self, phi, q, warmup=1, x0=None, P0=None, outputs=_STATE_OUTPUTS, buffers=None):
        """``run_smoother`` for B instances (kalmanfilter.py:676-694): filter, then RTS smoother."""
        want = set(outputs) | {"F", "Pf"}  # the backward pass re-reads the filtered moments
        if {"Xp", "Pp"} <= want:
            want |= {"S", "Ps"}  # full output set -> packed records for all three moment sets
        prob, keep, B = self._problem(phi, q, warmup, x0, P0)
        res = buffers if buffers is not None else self._alloc_outputs(B, [k for k in _STATE_OUTPUTS if k in want])
        o = self._outputs_struct(res)
        self._bind_stream()
        check(self._L.mk_filter_smooth(self._ctx, ctypes.byref(prob), ctypes.byref(o)))
        return res

    def set_scaling(self, scale=None, offset=None):
        """Series standard deviations / means ``[R,N]`` for the fused projection (``Metran.oseries_std`` /
        ``oseries_mean``; the scaled observation matrix of metran/metran.py:944-961).  None = 1 / 0."""
        def prep(a):
            if a is None:
                return None
            torch = _torch()
            if not isinstance(a, torch.Tensor):
                a = np.asarray(a, dtype=np.float64)
            if a.ndim == 1:
                a = a[None]
            return self._dev(a, (self.R, self.N), "scale/offset")
        self.scale, self.offset = prep(scale), prep(offset)
        return self

    def tape_path(self):
        """True when ``simulate_smoothed`` runs the inverse-free backward pass over the filter's tape (``MK_OUT_TAPE``:
        ``mk_split.hip`` OUT = 4 + ``mk_dk.hip``) instead of filtered records + the RTS smoother: wide models served by the
        split filter -- with more than 32 series: by the lane-per-state filter, same tape -- (16 < n <= 63), full-square engine,
        ``projection_path`` not "records"."""
        ok = (self.loadings is not None and not self.packed_sym and bool(self._L.mk_tape_supported(self.N, self.K))
              and self.get_variant("kernel_family") == "specialised")
        if self.projection_path == "tape" and not ok:
            raise MetranHipError("projection_path='tape' needs a model with 16 < N + K <= 63 and a full-square engine "
                                 "(got N=%s, K=%s)" % (self.N, getattr(self, "K", None)))
        return ok and self.projection_path != "records"

    def alloc_projection(self, B):
        """Buffers of ``simulate_smoothed`` for B instances (filtered record array -- or the backward tape, see
        ``tape_path`` -- + projected moments), for callers that run it repeatedly (pass them back as ``buffers=``)."""
        torch = _torch()
        res = {"mle": torch.empty(B, dtype=torch.float64, device=self.device),
               "status": torch.zeros(B, dtype=torch.int32, device=self.device),
               "sigmacount": torch.empty(B, dtype=torch.int64, device=self.device), "_rs": self.record_stride()}
        if self.tape_path():
            res["_rs"] = int(self._L.mk_tape_stride(self.N, self.K))
            res["_tape"] = True
            res["F"] = res["_rec_filt"] = self._empty_bt(B, self.T, res["_rs"])  # d_F = the tape
            res["sim_means"] = self._empty_bt(B, self.T, self.N)
            res["sim_vars"] = self._empty_bt(B, self.T, self.N)
            return res
        res["_rec_filt"], res["F"], res["Pf"], res["sigmas"], res["detfs"] = self._alloc_records(B)
        res["sim_means"] = self._empty_bt(B, self.T, self.N)
        res["sim_vars"] = self._empty_bt(B, self.T, self.N)
        return res

    def simulate_smoothed(self, phi, q, warmup=1, x0=None, P0=None, buffers=None):
        """``Metran.get_simulated_means/variances(method="smoother")`` for B instances in two launches
        WITHOUT materialising the smoothed states: the filter writes only the filtered records, the
        smoother reads them and writes the projected means/variances ``[B,T,N]`` (fused epilogue,
        kalmanfilter.py:569-603 with the scaling set by ``set_scaling``).  Returns a dict with
        ``sim_means, sim_vars, mle, sigmacount, status`` (and the filtered views ``F, Pf``; on the tape path of wide
        models -- ``tape_path`` -- ``F`` is the tape and there is no ``Pf``)."""
        prob, keep, B = self._problem(phi, q, warmup, x0, P0)
        res = buffers if buffers is not None else self.alloc_projection(B)
        o = self._outputs_struct(res)
        self._bind_stream()
        check(self._L.mk_filter_smooth(self._ctx, ctypes.byref(prob), ctypes.byref(o)))
        return res

    def state_tape_path(self):
        """True when ``smooth_state_variances`` runs over the STATE tape (``MK_OUT_TAPE | MK_OUT_VAR_ONLY``: the tape of
        ``tape_path`` plus K factor entries per step; ``mk_dk.hip`` STATE = true): the shapes of ``tape_path`` with zero
        observation variances (Metran's, metran.py:382-384)."""
        return self.tape_path() and self.obsvar is None

    def alloc_state_variances(self, B, projection=False):
        """Buffers of ``smooth_state_variances`` for B instances (pass them back as ``buffers=``).  ``projection``: on the
        state-tape path the same backward pass can write the projected moments ``sim_means / sim_vars [B,T,N]`` as well."""
        torch = _torch()
        res = {"mle": torch.empty(B, dtype=torch.float64, device=self.device),
               "status": torch.zeros(B, dtype=torch.int32, device=self.device),
               "sigmacount": torch.empty(B, dtype=torch.int64, device=self.device), "_rs": self.record_stride(),
               "_var_only": True}
        if self.state_tape_path():
            res["_rs"] = int(self._L.mk_state_tape_stride(self.N, self.K))
            res["_tape"] = True
            res["F"] = res["_rec_filt"] = self._empty_bt(B, self.T, res["_rs"])  # d_F = the state tape
            if projection:
                res["sim_means"] = self._empty_bt(B, self.T, self.N)
                res["sim_vars"] = self._empty_bt(B, self.T, self.N)
        else:
            if projection:
                raise MetranHipError("projection=True needs the state-tape path (state_tape_path())")
            res["_rec_filt"], res["F"], res["Pf"], res["sigmas"], res["detfs"] = self._alloc_records(B)
        res["S"] = self._empty_bt(B, self.T, self.n)
        res["var"] = self._empty_bt(B, self.T, self.n)
        return res

    def smooth_state_variances(self, phi, q, warmup=1, x0=None, P0=None, buffers=None):
        """``Metran.get_state_means`` / ``get_state_variances`` (metran.py:655-711, method="smoother") for B instances
        without materialising the smoothed covariances (``MK_OUT_VAR_ONLY``): smoothed state means ``S [B,T,n]`` and
        variances ``var [B,T,n]``.  n <= 16 and the shapes without a tape: the filter writes the filtered records, the
        RTS smoother reads them.  Wide models on the tape path (``state_tape_path``): the filter writes the STATE tape and
        the inverse-free backward pass turns it into the same moments -- no filtered record, no LDL^T chain."""
        prob, keep, B = self._problem(phi, q, warmup, x0, P0)
        res = buffers if buffers is not None else self.alloc_state_variances(B)
        o = self._outputs_struct(dict(res, Ps=res["var"]))
        self._bind_stream()
        check(self._L.mk_filter_smooth(self._ctx, ctypes.byref(prob), ctypes.byref(o)))
        return res

    def supported_shapes(self):
        """(N, K) pairs with kernels compiled into the library."""
        shapes = (ctypes.c_int64 * 256)()
        cnt = self._L.mk_supported_shapes(shapes, 128)
        return [(int(shapes[2 * i]), int(shapes[2 * i + 1])) for i in range(min(cnt, 128))]

    def smooth(self, phi, q, F, Pf, outputs=("S", "Ps")):
        """``kalmansmoother`` (kalmanfilter.py:403-476) from existing filtered moments
        ``F [B,T,n]``, ``Pf [B,T,n,n]``; needs no observations (the kernel depends on n only)."""
        F = self._layout(self._dev(F) if not isinstance(F, _torch().Tensor) else F.to(self.device))
        Pf = self._layout(self._dev(Pf) if not isinstance(Pf, _torch().Tensor) else Pf.to(self.device))
        B, T, n = (int(s) for s in F.shape)
        if tuple(Pf.shape) != (B, T, n, n):
            raise ValueError("Pf must be [B,T,n,n]")
        phi = self._dev(phi, (B, n), "phi")
        q = self._dev(q, (B, n), "q")
        shape = [(N, K) for (N, K) in self.supported_shapes() if N + K == n]
        if not shape and n >= 2 and self._L.mk_shape_supported(n - 1, 1):
            shape = [(n - 1, 1)]  # no specialised kernel of this state dimension: the size-generic smoother (it depends on n only)
        if not shape:
            raise MetranHipError("no HIP smoother kernel for state dimension n=%d" % n)
        N, K = shape[0]
        prob = Problem(B, 1, T, N, K, 1, None, self._p(phi), self._p(q), None, None, None, None, 0, None, None)
        torch = _torch()
        res = {"F": F, "Pf": Pf, "status": torch.zeros(B, dtype=torch.int32, device=self.device)}
        if "S" in outputs:
            res["S"] = self._empty_bt(B, T, n)
        if "Ps" in outputs:
            res["Ps"] = self._empty_bt(B, T, n, n)
        o = self._outputs_struct(res)
        self._bind_stream()
        check(self._L.mk_smooth(self._ctx, ctypes.byref(prob), ctypes.byref(o)))
        return res

    def smooth_dense(self, phi, F, Pf, Xp, Pp):
        """``kalmansmoother`` in its literal 5-argument form (kalmanfilter.py:403-476) for B models: the predicted moments
        ``Xp [B,T,n]``, ``Pp [B,T,n,n]`` are READ as handed in (:454-474), not recomputed from the filtered ones -- for
        moments that did not come out of this engine's own filter.  ``mk_smooth_dense`` (size-generic kernel, n <= 128).
        Returns a dict with ``S [B,T,n]``, ``Ps [B,T,n,n]``, ``status [B]``."""
        torch = _torch()
        F, Pf, Xp, Pp = (self._dev(a) for a in (F, Pf, Xp, Pp))
        B, T, n = (int(s) for s in F.shape)
        if tuple(Pf.shape) != (B, T, n, n) or tuple(Xp.shape) != (B, T, n) or tuple(Pp.shape) != (B, T, n, n):
            raise ValueError("F, Xp must be [B,T,n] and Pf, Pp [B,T,n,n]")
        phi = self._dev(phi, (B, n), "phi")
        res = {"S": torch.empty((B, T, n), dtype=torch.float64, device=self.device),
               "Ps": torch.empty((B, T, n, n), dtype=torch.float64, device=self.device),
               "status": torch.zeros(B, dtype=torch.int32, device=self.device)}
        self._bind_stream()
        check(self._L.mk_smooth_dense(self._ctx, B, T, n, self._p(phi), self._p(F), self._p(Pf), self._p(Xp), self._p(Pp),
                                      self._p(res["S"]), self._p(res["Ps"]), self._p(res["status"])))
        return res

    # ------------------------------------------------------------------ projection epilogues
    def simulate(self, Z, means, covs):
        """``SPKalmanFilter.simulate`` (kalmanfilter.py:569-603): Z ``[RZ,N,n]`` or ``[N,n]``."""
        torch = _torch()
        Z = self._dev(Z)
        if Z.ndim == 2:
            Z = Z[None]
        means = self._dev(means)
        covs = self._dev(covs)
        B, T, n = (int(s) for s in means.shape)
        RZ, N = int(Z.shape[0]), int(Z.shape[1])
        sm = torch.empty((B, T, N), dtype=torch.float64, device=self.device)
        sv = torch.empty((B, T, N), dtype=torch.float64, device=self.device)
        self._bind_stream()
        check(self._L.mk_simulate(self._ctx, B, RZ, T, N, n, self._p(Z), self._p(means), self._p(covs), self._p(sm),
                                  self._p(sv)))
        return sm, sv

    def decompose(self, Z, means):
        """``SPKalmanFilter.decompose`` (kalmanfilter.py:605-644) -> (sdf [B,T,N], cdf [B,K,T,N])."""
        torch = _torch()
        Z = self._dev(Z)
        if Z.ndim == 2:
            Z = Z[None]
        means = self._dev(means)
        B, T, n = (int(s) for s in means.shape)
        RZ, N = int(Z.shape[0]), int(Z.shape[1])
        sdf = torch.empty((B, T, N), dtype=torch.float64, device=self.device)
        cdf = torch.empty((B, n - N, T, N), dtype=torch.float64, device=self.device)
        self._bind_stream()
        check(self._L.mk_decompose(self._ctx, B, RZ, T, N, n, self._p(Z), self._p(means), self._p(sdf), self._p(cdf)))
        return sdf, cdf

    def sum(self, values):
        """Deterministic device sum (fixed order) of a 1-D tensor -> 0-d tensor."""
        torch = _torch()
        values = self._dev(values).reshape(-1)
        out = torch.empty(1, dtype=torch.float64, device=self.device)
        self._bind_stream()
        check(self._L.mk_sum(self._ctx, int(values.numel()), self._p(values), self._p(out)))
        return out[0]

    # ------------------------------------------------------------------ the one collective (C ABI mk_allreduce_sum)
    @staticmethod
    def comm_unique_id():
        """128 bytes identifying a new RCCL communicator (``mk_comm_unique_id`` = ``ncclGetUniqueId``): ONE rank calls it
        and hands the bytes to the others (``distributed.attach_communicator`` does that over the process group's store)."""
        buf = ctypes.create_string_buffer(128)
        check(_lib.lib().mk_comm_unique_id(buf))
        return buf.raw

    def init_communicator(self, nranks, rank, unique_id):
        """``mk_comm_init_rank``: join the RCCL communicator ``unique_id`` as ``rank`` of ``nranks`` on this engine's device
        (collective: every rank calls it).  The context owns the communicator until ``close()``."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        check(self._L.mk_comm_init_rank(self._ctx, int(nranks), int(rank), ctypes.c_char_p(bytes(unique_id))))
        self._has_comm = True
        return self

    def set_communicator(self, nccl_comm):
        """``mk_set_communicator``: use a caller-owned ``ncclComm_t`` (an integer address / ``c_void_p``); None detaches."""
        check(self._L.mk_set_communicator(self._ctx, ctypes.c_void_p(nccl_comm) if nccl_comm else None))
        self._has_comm = bool(nccl_comm)
        return self

    def has_communicator(self):
        return bool(getattr(self, "_has_comm", False))

    def allreduce_sum(self, t):
        """In-place all-reduce(sum) of a float64 device tensor over the engine's communicator (``mk_allreduce_sum``, on the
        current stream).  Raises without a communicator: there is no single-rank shortcut."""
        torch = _torch()
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("allreduce_sum takes a contiguous float64 tensor on the engine's device")
        self._bind_stream()
        check(self._L.mk_allreduce_sum(self._ctx, self._p(t), int(t.numel())))
        return t

    # ------------------------------------------------------------------ lock-step L-BFGS (calibrate_batch; mk_lbfgs.hip)
    def lbfgs_direction(self, x, g, lo, active, Sh, Yh, rho, hlen, hpos, gtol, pg, d, phase=None, step=None, nback=None):
        """Projected gradient into ``pg``, ``active &= max|pg| > gtol``, the two-loop recursion over every model's own history
        ring (``hlen`` live pairs from slot ``hpos``) with its safeguards into ``d``.  With ``phase / step / nback`` (own line
        search per model) a model in the middle of its search keeps direction and step, the others start a new one.  Returns the
        number of active models."""
        R, n = (int(v) for v in x.shape)
        cnt = ctypes.c_int(0)
        self._bind_stream()
        check(self._L.mk_lbfgs_direction(self._ctx, R, n, int(Sh.shape[0]), self._p(x), self._p(g), self._p(lo), self._p(active), self._p(Sh),
                                         self._p(Yh), self._p(rho), self._p(hlen), self._p(hpos), float(gtol), self._p(pg), self._p(d),
                                         self._p(phase), self._p(step), self._p(nback), ctypes.byref(cnt)))
        return int(cnt.value)

    def lbfgs_trial(self, x, d, step, lo, searching, x_new, xt, xe):
        """``xt = max(x + step d, lo)``; ``xe = xt`` for the searching models, their accepted point for the settled ones."""
        R, n = (int(v) for v in x.shape)
        self._bind_stream()
        check(self._L.mk_lbfgs_trial(self._ctx, R, n, self._p(x), self._p(d), self._p(step), self._p(lo), self._p(searching), self._p(x_new),
                                     self._p(xt), self._p(xe)))

    def lbfgs_armijo(self, ft, f, pg, xt, x, searching, step, x_new, f_new, nback=None, max_backtracks=0, accepted=None):
        """Armijo test of the searching models at their trial points (in place: ``x_new, f_new, searching, step``).  Lock-step form:
        returns the number of models still searching.  Own-line-search form (``nback``, ``accepted`` given): an accepted model is
        marked in ``accepted`` and stays in ``searching`` (the active mask), a model out of trial points leaves it; returns
        ``(still searching, accepted)``."""
        R, n = (int(v) for v in x.shape)
        ns, na = ctypes.c_int(0), ctypes.c_int(0)
        self._bind_stream()
        own = nback is not None
        check(self._L.mk_lbfgs_armijo(self._ctx, R, n, self._p(ft), self._p(f), self._p(pg), self._p(xt), self._p(x), self._p(searching),
                                      self._p(step), self._p(x_new), self._p(f_new), self._p(nback), int(max_backtracks), self._p(accepted),
                                      ctypes.byref(ns), ctypes.byref(na) if own else None))
        return (int(ns.value), int(na.value)) if own else int(ns.value)

    def lbfgs_update(self, x, f, g, x_new, f_new, g_new, keep_old, searching, active, ftol, Sh, Yh, rho, hlen, hpos, mask=None, phase=None,
                     nit=None, maxiter=0):
        """For the models of ``mask`` (None: all): the pair of the accepted point into the model's ring if it is usable,
        ``(x, f, g) <- (x_new, f_new, g_new)`` (lock-step form: models still searching keep their old gradient if ``keep_old`` and
        leave ``active``), ``active &= relative reduction > ftol``.  ``nit [R]`` int32: every active model updated here has taken one
        more quasi-Newton iteration; at ``maxiter`` (> 0) it leaves ``active``.  Returns the number of usable pairs."""
        R, n = (int(v) for v in x.shape)
        cnt = ctypes.c_int(0)
        self._bind_stream()
        check(self._L.mk_lbfgs_update(self._ctx, R, n, int(Sh.shape[0]), self._p(x), self._p(f), self._p(g), self._p(x_new), self._p(f_new),
                                      self._p(g_new), 1 if keep_old else 0, self._p(searching), self._p(mask), self._p(active), float(ftol),
                                      self._p(Sh), self._p(Yh), self._p(rho), self._p(hlen), self._p(hpos), self._p(phase), self._p(nit),
                                      int(maxiter), ctypes.byref(cnt)))
        return int(cnt.value)

    # ------------------------------------------------------------------ instrumentation
    def enable_timing(self, enable=True, accumulate=False):
        """hipEvents around every hot-kernel launch.  ``accumulate``: every launch keeps its own event pair until
        ``kernel_ms_totals`` collects them (no host synchronisation inside a timed loop); otherwise only the most
        recent launch of each kind is kept (``last_kernel_ms``)."""
        self._timing = bool(enable)
        check(self._L.mk_enable_timing(self._ctx, (2 if accumulate else 1) if enable else 0))

    def kernel_ms_totals(self):
        """(filter_ms, filter_launches, smoother_ms, smoother_launches) since the previous call (accumulate mode)."""
        f, s = ctypes.c_double(0.0), ctypes.c_double(0.0)
        nf, ns = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._L.mk_kernel_ms_totals(self._ctx, ctypes.byref(f), ctypes.byref(nf), ctypes.byref(s), ctypes.byref(ns)))
        return float(f.value), int(nf.value), float(s.value), int(ns.value)

    def last_kernel_ms(self):
        """(filter_ms, smoother_ms) of the most recent launches, measured with hipEvents on the launch stream."""
        f, s = ctypes.c_float(-1.0), ctypes.c_float(-1.0)
        check(self._L.mk_last_kernel_ms(self._ctx, ctypes.byref(f), ctypes.byref(s)))
        return float(f.value), float(s.value)

    def synchronize(self):
        check(self._L.mk_sync(self._ctx))
