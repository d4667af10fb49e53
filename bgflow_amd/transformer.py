"""Coupling-layer transformers backed by the hand-written HIP kernels (no CPU path).

API mirrors bgflow/nn/flow/transformer/{base,affine,spline}.py: same class names, constructor
arguments, attribute / state_dict names (``_params_net``, ``_shift_transformation``,
``_scale_transformation``, ``_log_alpha``) and ``_forward(x_cond, y, *cond, **kwargs) ->
(y', dlogp[B,1])`` protocol.  Conditioners are arbitrary torch modules; when the spline conditioner
is a bgflow_amd DenseNet (optionally wrapped in WrapPeriodic) and no gradient is required, the
whole coupling layer runs as ONE fused kernel (MLP on the f32 matrix cores + spline epilogue).
"""
import warnings

import numpy as np
import torch

from . import _lib
from .utils import row_pitch
from .flow import ACC_KW, Flow, as_tensor

def _invalidate_fused_cache(self):
    """Forget the packed-operand cache of the fused kernels.  The cache is keyed on the parameters' (data_ptr, _version): optimizer
    steps, load_state_dict and ordinary in-place ops are picked up automatically; edits through ``.data`` (EMA / weight-swap code)
    bump no version counter -- call this after them."""
    self._fused_cache.clear()
    self.__dict__.pop("_train_cache", None)


__all__ = ["Transformer", "AffineTransformer", "ConditionalSplineTransformer"]

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


class Transformer(Flow):
    """transformer/base.py:7-16"""

    def _forward(self, x, y, *args, **kwargs):
        raise NotImplementedError()

    def _inverse(self, x, y, *args, **kwargs):
        raise NotImplementedError()


def _flat2d(t):
    """[..., d] -> ([N, d] row-major view/copy, leading shape)"""
    lead = t.shape[:-1]
    t2 = t.reshape(-1, t.shape[-1])
    return t2, lead


# ------------------------------------------------------------------------------------------------
# affine
# ------------------------------------------------------------------------------------------------
def _dlogp_target(acc, B, device):
    """([B] buffer, accumulate flag) of a launch: the pass's running log-det buffer (flow._LogDetAcc) or a fresh tensor"""
    if acc is None:
        return torch.empty((B,), dtype=torch.float32, device=device), 0
    buf, started = acc.peek()
    assert buf.shape[0] == B and buf.device == device, "running log-det buffer does not match the batch"
    return buf, int(started)


def affine_transform(y, mu, s_raw, log_alpha, preserve_volume, is_circular, inverse, acc=None):
    """Launch bgk_affine_transform.  y, mu, s_raw: [..., d] (mu / s_raw may be None).  ``acc``: the pass's running log-det
    (the kernel adds to it; returned in place of the dlogp tensor)."""
    _lib.require_hip(y, mu, s_raw, log_alpha)
    y2, lead = _flat2d(y)
    y2, ldy = _lib.rowmajor(y2)
    B, d = y2.shape
    mu2 = s2 = None
    ldmu = lds = 0
    if mu is not None:
        mu2, ldmu = _lib.rowmajor(mu.reshape(-1, d))
    if s_raw is not None:
        s2, lds = _lib.rowmajor(s_raw.reshape(-1, d))
    out = torch.empty((B, d), dtype=torch.float32, device=y.device)
    dlogp, accumulate = _dlogp_target(acc, B, y.device)
    with torch.cuda.device(y.device):
        st = _lib.lib().bgk_affine_transform(
            _lib.ptr(y2), ldy, _lib.ptr(mu2), ldmu, _lib.ptr(s2), lds, _lib.ptr(log_alpha),
            int(preserve_volume), int(is_circular), int(inverse), B, d, _lib.ptr(out), d,
            _lib.ptr(dlogp), accumulate, _lib.stream_ptr(y.device))
    _lib.check(st, "bgk_affine_transform")
    if acc is not None:
        acc.commit()
        return out.reshape(*lead, d), acc
    return out.reshape(*lead, d), dlogp.reshape(*lead, 1)


class _AffineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, mu, s_raw, log_alpha, preserve_volume, is_circular, inverse):
        ctx.save_for_backward(y, mu, s_raw, log_alpha)
        ctx.cfg = (preserve_volume, is_circular, inverse)
        out, dlogp = affine_transform(y, mu, s_raw, log_alpha, preserve_volume, is_circular, inverse)
        return out, dlogp

    @staticmethod
    def backward(ctx, g_out, g_dlogp):
        y, mu, s_raw, log_alpha = ctx.saved_tensors
        pv, circ, inverse = ctx.cfg
        d = y.shape[-1]
        lead = y.shape[:-1]
        y2, ldy = _lib.rowmajor(y.reshape(-1, d))
        B = y2.shape[0]
        g_out2, ldgo = _lib.rowmajor(g_out.reshape(-1, d).contiguous())
        g_dl = g_dlogp.reshape(-1).contiguous()
        mu2 = s2 = None
        ldmu = lds = 0
        if mu is not None:
            mu2, ldmu = _lib.rowmajor(mu.reshape(-1, d))
        if s_raw is not None:
            s2, lds = _lib.rowmajor(s_raw.reshape(-1, d))
        dev = y.device
        g_y = torch.empty((B, d), dtype=torch.float32, device=dev)
        g_mu = torch.empty((B, d), dtype=torch.float32, device=dev) if mu is not None else None
        g_s = torch.empty((B, d), dtype=torch.float32, device=dev) if s_raw is not None else None
        g_la = torch.zeros((1,), dtype=torch.float32, device=dev) if s_raw is not None else None
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_affine_backward(
                _lib.ptr(y2), ldy, _lib.ptr(mu2), ldmu, _lib.ptr(s2), lds, _lib.ptr(log_alpha),
                int(pv), int(circ), int(inverse), B, d, _lib.ptr(g_out2), ldgo, _lib.ptr(g_dl),
                _lib.ptr(g_y), d, _lib.ptr(g_mu), d, _lib.ptr(g_s), d, _lib.ptr(g_la), None, None, _lib.stream_ptr(dev))
        _lib.check(st, "bgk_affine_backward")
        shp = (*lead, d)
        return (g_y.reshape(shp), None if g_mu is None else g_mu.reshape(shp),
                None if g_s is None else g_s.reshape(shp), g_la, None, None, None)


class AffineTransformer(Transformer):
    """RealNVP / NICE transformer (transformer/affine.py:11-70).

    ``y' = exp(log_sigma) * y + mu`` with ``log_sigma = tanh(scale(x)) * exp(_log_alpha)``;
    ``dlogp = sum(log_sigma)``; optional volume preservation and periodic wrap ``% 1``.
    The conditioner networks are torch modules; the elementwise tail + row reduction is one HIP
    kernel (bgk_affine_transform)."""

    def __init__(self, shift_transformation=None, scale_transformation=None, init_downscale=1.0,
                 preserve_volume=False, is_circular=False):
        if scale_transformation is not None and is_circular:
            raise ValueError("Scaling is not compatible with periodicity.")
        super().__init__()
        self._shift_transformation = shift_transformation
        self._scale_transformation = scale_transformation
        self._log_alpha = torch.nn.Parameter(torch.zeros(1) - init_downscale)
        self._preserve_volume = preserve_volume
        self._is_circular = is_circular
        self._fused_cache = {}
        self.allow_fused = True           # set False to force conditioner networks + bgk_affine_transform

    _bgk_acc = True
    _bgk_multi_cond = True

    def _run(self, x, y, cond, inverse, acc=None):
        grad = torch.is_grad_enabled() and (
            y.requires_grad or x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if grad:
            acc = None                        # the autograd Functions return fresh tensors
        if not grad and self.allow_fused and not cond:
            from .dense import fused_affine_coupling   # late import (dense imports nothing from here)
            fused = fused_affine_coupling(self, x, y, inverse, acc=acc)
            if fused is not None:
                return fused
        x = as_tensor(x)
        if grad and self.allow_fused and not cond:
            from .dense import fused_affine_coupling_train
            fused = fused_affine_coupling_train(self, x, y, inverse)      # forward and backward on the hand-written kernels (round 6)
            if fused is not None:
                return fused
        mu = self._shift_transformation(x, *cond) if self._shift_transformation is not None else None
        s_raw = self._scale_transformation(x, *cond) if self._scale_transformation is not None else None
        if mu is not None:
            assert mu.shape[-1] == y.shape[-1]
        if s_raw is not None:
            assert s_raw.shape[-1] == y.shape[-1]
        log_alpha = self._log_alpha.to(device=y.device, dtype=torch.float32)
        needs_grad = torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (y, mu, s_raw, log_alpha))
        if needs_grad:
            return _AffineFn.apply(y, mu, s_raw, log_alpha, self._preserve_volume, self._is_circular, inverse)
        return affine_transform(y, mu, s_raw, log_alpha, self._preserve_volume, self._is_circular, inverse,
                                acc=acc if y.dim() == 2 else None)

    def _forward(self, x, y, *cond, **kwargs):
        return self._run(x, y, cond, False, acc=kwargs.get(ACC_KW))

    def _inverse(self, x, y, *cond, **kwargs):
        return self._run(x, y, cond, True, acc=kwargs.get(ACC_KW))


# ------------------------------------------------------------------------------------------------
# rational-quadratic spline
# ------------------------------------------------------------------------------------------------
def rqs_transform(y, params, nc_slot, n_bins, inverse, left, right, bottom, top, settings,
                  want_bin_idx=False, oob_counter=None, acc=None):
    """Launch bgk_rqs_transform.  y [..., d], params [..., P]; returns (out, dlogp[...,1][, bin_idx]).  ``acc``: the pass's
    running log-det buffer (returned in place of dlogp)."""
    _lib.require_hip(y, params, nc_slot)
    y2, lead = _flat2d(y)
    y2, ldy = _lib.rowmajor(y2)
    B, d = y2.shape
    P = params.shape[-1]
    # (any bin count: rows that leave no room for an LDS tile -- more than 64 bins for ~17 dims -- run on the kernel's direct
    # variant, every lane walking its element's parameters in memory; the backward has the same two forms, see rqs_backward)
    p2, ldp = _lib.rowmajor(params.reshape(-1, P))
    out = torch.empty((B, d), dtype=torch.float32, device=y.device)
    if y.dim() != 2:
        acc = None
    dlogp, accumulate = _dlogp_target(acc, B, y.device)
    bins = torch.empty((B, d), dtype=torch.int32, device=y.device) if want_bin_idx else None
    with torch.cuda.device(y.device):
        st = _lib.lib().bgk_rqs_transform(
            _lib.ptr(y2), ldy, _lib.ptr(p2), ldp, P, _lib.ptr(nc_slot), B, d, n_bins, int(inverse),
            left, right, bottom, top, settings["min_bin_width"], settings["min_bin_height"],
            settings["min_derivative"], int(settings.get("enable_identity_init", False)),
            _lib.ptr(out), d, _lib.ptr(dlogp), accumulate, _lib.ptr(bins), _lib.ptr(oob_counter),
            _lib.stream_ptr(y.device))
    _lib.check(st, "bgk_rqs_transform")
    if acc is not None:
        acc.commit()
        res = (out.reshape(*lead, d), acc)
    else:
        res = (out.reshape(*lead, d), dlogp.reshape(*lead, 1))
    return res + (bins.reshape(*lead, d),) if want_bin_idx else res


class _RQSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, params, nc_slot, n_bins, inverse, left, right, bottom, top, settings, oob_counter):
        ctx.save_for_backward(y, params, nc_slot)
        ctx.cfg = (n_bins, inverse, left, right, bottom, top, dict(settings))
        return rqs_transform(y, params, nc_slot, n_bins, inverse, left, right, bottom, top, settings,
                             oob_counter=oob_counter)

    @staticmethod
    def backward(ctx, g_out, g_dlogp):
        y, params, nc_slot = ctx.saved_tensors
        g_y, g_p = rqs_backward(y, params, nc_slot, ctx.cfg, g_out, g_dlogp)
        return (g_y, g_p) + (None,) * 9


def rqs_backward(y, params, nc_slot, cfg, g_out, g_dlogp, absmax=None, packed_width=None):
    """Launch bgk_rqs_backward: VJP of rqs_transform w.r.t. (y, params); cfg = (n_bins, inverse, left, right, bottom,
    top, settings).  Any bin count: 4 / 8 / 12 / 16 / 32 bins on the register-resident streaming kernel, every other count on the
    kernel's direct variant (the parameters stay in memory and are walked) -- no device torch ops in the backward either.
    ``packed_width`` = P: ``params`` is the element-major tensor [B, d (3 K + 1)] the fused training forward saved (its
    ``params_layout = 1``), P the width of the conditioner's output; the gradient comes back as [B, P] in the reference's order."""
    n_bins, inverse, left, right, bottom, top, settings = cfg
    d = y.shape[-1]
    P = params.shape[-1] if packed_width is None else int(packed_width)
    y2, ldy = _lib.rowmajor(y.reshape(-1, d))
    p2, ldp = _lib.rowmajor(params.reshape(-1, params.shape[-1]))
    B = y2.shape[0]
    g_out2 = g_out.reshape(-1, d).contiguous()
    g_dl = g_dlogp.reshape(-1).contiguous()
    g_y = torch.empty((B, d), dtype=torch.float32, device=y.device)
    # rows padded to a multiple of 4 floats: the 16-byte accesses of this kernel and of bgk_dense_backward_dx stay aligned
    # (P = 425 gives 1700-byte rows otherwise)
    ldgp = row_pitch(P)
    g_p = torch.empty((B, ldgp), dtype=torch.float32, device=y.device)[:, :P]
    with torch.cuda.device(y.device):
        st = _lib.lib().bgk_rqs_backward(
            _lib.ptr(y2), ldy, _lib.ptr(p2), ldp, P, _lib.ptr(nc_slot), B, d, n_bins, int(inverse),
            left, right, bottom, top, settings["min_bin_width"], settings["min_bin_height"],
            settings["min_derivative"], int(settings.get("enable_identity_init", False)),
            _lib.ptr(g_out2), d, _lib.ptr(g_dl), _lib.ptr(g_y), d, _lib.ptr(g_p), ldgp, _lib.ptr(absmax),
            0 if packed_width is None else 1, _lib.stream_ptr(y.device))
    _lib.check(st, "bgk_rqs_backward")
    return g_y.reshape(y.shape), (g_p.reshape(params.shape) if packed_width is None else g_p)


class ConditionalSplineTransformer(Transformer):
    """Rational-quadratic spline transformer on [left, right] -> [bottom, top]
    (transformer/spline.py:14-204; arithmetic = nflows ``rational_quadratic_spline``).

    ``params_net(x)`` must return ``3 * n_bins * d (+ number of non-circular dims)`` values per
    sample, packed ``[widths | heights | slopes | non-circular extra slopes]``; ``n_bins`` is
    inferred from the width.  bgflow's *forward* is the spline's root-solve branch
    (nflows ``inverse=True``), bgflow's *inverse* the direct evaluation (spline.py:133-144,164-175).
    Inputs outside the domain are clamped and a UserWarning is raised (spline.py:145-155) -- here
    the condition is a device-side counter that is polled lazily (``check_domain()``), so the hot
    path has no host synchronisation.

    Deviation (documented in SURVEY.md section 7): for *mixed* circular masks the number of extra
    slopes is the number of NON-circular dims (the evident intent; the reference's
    ``_n_noncircular`` returns the circular count, spline.py:190-196, which coincides only when
    both counts are equal).  ``enable_identity_init`` is always available (default True, the
    reference's stated intent, spline.py:76-78).
    """

    def __init__(self, params_net, is_circular=False, left=0.0, right=1.0, bottom=0.0, top=1.0):
        super().__init__()
        self._params_net = params_net
        self._is_circular = torch.as_tensor(is_circular, dtype=torch.bool)
        self._left, self._right, self._bottom, self._top = left, right, bottom, top
        self._default_settings = {
            "min_bin_width": DEFAULT_MIN_BIN_WIDTH,
            "min_bin_height": DEFAULT_MIN_BIN_HEIGHT,
            "min_derivative": DEFAULT_MIN_DERIVATIVE,
            "enable_identity_init": True,
        }
        self._nc_cache = {}
        self._oob = {}
        self._fused_cache = {}
        self.allow_fused = True           # set False to force conditioner + bgk_rqs_transform
        self.return_bin_indices = False   # parity hook: stash the bin indices of the last call
        self.last_bin_indices = None

    # -- parameter layout ------------------------------------------------------------------
    def _circular_mask(self, y_dim):
        return np.broadcast_to(self._is_circular.cpu().numpy().astype(bool), (y_dim,)).copy()

    def _n_noncircular(self, y_dim):
        return int((~self._circular_mask(y_dim)).sum())

    def _nc_slot(self, y_dim, device):
        key = (y_dim, str(device))
        if key not in self._nc_cache:
            circ = self._circular_mask(y_dim)
            slots = np.full(y_dim, -1, dtype=np.int32)
            slots[~circ] = np.arange(int((~circ).sum()), dtype=np.int32)
            self._nc_cache[key] = (torch.from_numpy(slots).to(device), slots)
        return self._nc_cache[key]

    def _oob_counter(self, device):
        key = str(device)
        if key not in self._oob:
            self._oob[key] = torch.zeros(1, dtype=torch.int32, device=device)
        return self._oob[key]

    def check_domain(self):
        """Poll the device-side out-of-domain counters (host sync) and raise the reference's
        UserWarning if any input had to be clamped since the last check."""
        total = 0
        for c in self._oob.values():
            total += int(c.item())
            c.zero_()
        if total:
            warnings.warn(f"InputOutsideDomain: {total} inputs outside [{self._left}, {self._right}] were clamped",
                          UserWarning)
        return total

    # -- forward / inverse ------------------------------------------------------------------
    _bgk_acc = True
    _bgk_multi_cond = True

    def _run(self, x, y, inverse, acc=None):
        from .dense import fused_spline_coupling, fused_spline_coupling_train   # late import
        y_dim = y.shape[-1]
        nc_dev, nc_host = self._nc_slot(y_dim, y.device)
        oob = self._oob_counter(y.device)
        grad = torch.is_grad_enabled() and (
            y.requires_grad or x.requires_grad or any(p.requires_grad for p in self._params_net.parameters()))
        if grad:
            acc = None                        # the autograd Functions return fresh tensors
        if not grad and self.allow_fused:
            fused = fused_spline_coupling(self, x, y, nc_host, inverse, oob, want_bin_idx=self.return_bin_indices, acc=acc)
            if fused is not None:
                if self.return_bin_indices:
                    self.last_bin_indices = fused[2]
                return fused[0], fused[1]
        x = as_tensor(x)
        if grad and self.allow_fused and not self.return_bin_indices:
            fused = fused_spline_coupling_train(self, x, y, nc_dev, nc_host, inverse, oob)
            if fused is not None:
                return fused
        params = self._params_net(x)
        n_nc = int((nc_host >= 0).sum())
        P = params.shape[-1]
        n_bins = P // (3 * y_dim)
        if 3 * n_bins * y_dim + n_nc != P:
            raise RuntimeError(
                f"params_net output width {P} does not match 3 * n_bins * {y_dim} + {n_nc} "
                f"(split_with_sizes in the reference, transformer/spline.py:113-117)")
        if grad:           # forward and backward on the kernels (any bin count: bgk_rqs_transform / bgk_rqs_backward)
            return _RQSFn.apply(y, params, nc_dev, n_bins, inverse, self._left, self._right, self._bottom,
                                self._top, self._default_settings, oob)
        res = rqs_transform(y, params, nc_dev, n_bins, inverse, self._left, self._right, self._bottom,
                            self._top, self._default_settings, want_bin_idx=self.return_bin_indices,
                            oob_counter=oob, acc=acc)
        if self.return_bin_indices:
            self.last_bin_indices = res[2]
        return res[0], res[1]

    def _forward(self, x, y, *args, **kwargs):
        return self._run(x, y, False, acc=kwargs.get(ACC_KW))

    def _inverse(self, x, y, *args, **kwargs):
        return self._run(x, y, True, acc=kwargs.get(ACC_KW))


AffineTransformer.invalidate_fused_cache = _invalidate_fused_cache
ConditionalSplineTransformer.invalidate_fused_cache = _invalidate_fused_cache
