"""Domain-mapping layer ``CDFTransform`` (bgflow/nn/flow/cdf.py:13-46).

For the marginals the builder installs (truncated normal, normal, uniform; factory/icmarginals.py:41-77)
and f32 HIP tensors one hand-written kernel (bgk_cdf_transform) replaces the ~10 elementwise aten launches +
row reduction, with bgk_cdf_backward behind a torch.autograd.Function for inputs that need gradients; any other
distribution, or a marginal whose own parameters are being trained, runs the distribution's cdf / icdf / log_prob
(stock PyTorch-ROCm ops, differentiable)."""
import numpy as np
import torch

from . import _lib
from .utils import param_state_key
from .flow import ACC_KW, Flow, InverseFlow, SequentialFlow

__all__ = ["CDFTransform", "DistributionTransferFlow", "ConstrainGaussianFlow"]


def _descriptor(dist, d):
    """[d, 6] float32 descriptor of a supported distribution, or None."""
    name = type(dist).__name__

    def col(v):
        v = torch.as_tensor(v, dtype=torch.float32).detach().cpu().reshape(-1)
        return v.expand(d).clone() if v.numel() == 1 else v
    try:
        desc = torch.zeros(d, 6)
        if name == "TruncatedNormalDistribution":
            desc[:, 0] = 2
            desc[:, 1], desc[:, 2] = col(dist._mu), col(torch.exp(dist._logsigma))
            desc[:, 3] = col(dist._cdf_lower_bound)
            desc[:, 4] = col(dist._cdf_upper_bound) - col(dist._cdf_lower_bound)
        elif name in ("_NormalMarginal", "Normal"):
            desc[:, 0] = 1
            desc[:, 1], desc[:, 2] = col(dist.loc), col(dist.scale)
        elif name in ("SloppyUniform", "_SloppyUniform", "Uniform"):
            desc[:, 0] = 0
            desc[:, 1], desc[:, 2] = col(dist.low), col(dist.high)
            desc[:, 3] = float(getattr(dist, "tol", 0.0))
        else:
            return None
        # slot 5: the element-independent part of -log_prob(y) on the icdf side (used by bgk_icdf_ic2xyz; f64 on the host):
        # uniform log(high - low); normal log(sigma) + log sqrt(2 pi); truncated normal log(Z sigma) + log sqrt(2 pi)
        d64 = desc.double()
        half_log_2pi = 0.9189385332046727
        desc[:, 5] = torch.where(d64[:, 0] == 0, torch.log(d64[:, 2] - d64[:, 1]),
                                 torch.where(d64[:, 0] == 1, torch.log(d64[:, 2]) + half_log_2pi,
                                             torch.log((d64[:, 4] * d64[:, 2]).clamp_min(1e-300)) + half_log_2pi)).float()
        return desc.contiguous()
    except (AttributeError, RuntimeError):
        return None


def _reverted_cdf_series(alpha):
    """c2..c5 of h = s + c2 s^2 + ... + c5 s^5, the solution of Phi(alpha + h) - Phi(alpha) = s pdf(alpha) for small s (reverted
    Taylor series of the standard normal cdf around alpha; derivatives of the pdf: -a phi, (a^2 - 1) phi, -(a^3 - 3a) phi, ...)"""
    a2, a3 = -alpha / 2.0, (alpha ** 2 - 1.0) / 6.0
    a4, a5 = -(alpha ** 3 - 3.0 * alpha) / 24.0, (alpha ** 4 - 6.0 * alpha ** 2 + 3.0) / 120.0
    return (-a2, 2.0 * a2 ** 2 - a3, -5.0 * a2 ** 3 + 5.0 * a2 * a3 - a4,
            14.0 * a2 ** 4 - 21.0 * a2 ** 2 * a3 + 3.0 * a3 ** 2 + 6.0 * a2 * a4 - a5)


TAIL_DESC = 20     # floats per channel of the sampling-tail kernel's descriptor (csrc/bgk_tail.hip)


def _tail_descriptor(dist, d):
    """[d, 20] float32 descriptor of a marginal for bgk_icdf_ic2xyz_reg (layout: csrc/bgk_tail.hip), all constants in f64 on the
    host: the affine maps around erfinv folded into one fma each, the element-independent part of -log_prob, and -- for a
    truncated normal -- the reverted cdf series around each finite bound (the kernel evaluates the DISTANCE to the bound directly
    there instead of mu + sigma z).  None for an unsupported distribution."""
    from scipy import special as sps
    base = _descriptor(dist, d)
    if base is None:
        return None
    b = base.double().numpy()
    out = np.zeros((d, TAIL_DESC), np.float64)
    kinds = np.zeros(d, np.int32)
    half_log_2pi = 0.9189385332046727
    for j in range(d):
        kind = int(b[j, 0])
        kinds[j] = kind
        if kind == 0:
            low, high = b[j, 1], b[j, 2]
            out[j, 1], out[j, 2], out[j, 5] = low, high - low, np.log(high - low)
            continue
        mu, sigma = b[j, 1], b[j, 2]
        clo, Z = (0.0, 1.0) if kind == 1 else (b[j, 3], b[j, 4])
        out[j, 1], out[j, 2], out[j, 3], out[j, 4] = mu, sigma * np.sqrt(2.0), 2.0 * Z, 2.0 * clo - 1.0
        out[j, 5] = np.log(Z * sigma) + half_log_2pi
        out[j, 6], out[j, 13] = 1.0 / (sigma * np.sqrt(2.0)), sigma
        out[j, 7] = out[j, 14] = 1e30                                     # no bound: the series window is never entered
        if kind == 2:
            for lower, off in ((True, 7), (False, 14)):
                c = clo if lower else clo + Z
                if not (1e-300 < c < 1.0 - 1e-16):
                    continue
                x0 = sps.ndtri(c)
                pdf = np.exp(-0.5 * x0 * x0) / np.sqrt(2.0 * np.pi)
                out[j, off] = Z / pdf
                out[j, off + 1:off + 5] = _reverted_cdf_series(x0 if lower else -x0)
                out[j, 12 if lower else 19] = mu + sigma * x0
    out32 = out.astype(np.float32)
    out32[:, 0] = kinds.view(np.float32)                                  # the kind travels as int32 bits
    return torch.from_numpy(out32).contiguous()


def _source_tensors(dist):
    """the tensors a descriptor is built from (parameters / buffers of the marginal), for cache validation"""
    out = []
    for name in ("_mu", "_logsigma", "_cdf_lower_bound", "_cdf_upper_bound", "loc", "scale", "low", "high"):
        v = getattr(dist, name, None)
        if torch.is_tensor(v):
            out.append(v)
    return out


class _CdfFn(torch.autograd.Function):
    """bgk_cdf_transform with the backward kernel bgk_cdf_backward (no aten ops in the KL / NLL step)"""

    @staticmethod
    def forward(ctx, x, desc, inverse, eps):
        out, dlogp = _launch(x, desc, inverse, eps)
        ctx.desc, ctx.inverse, ctx.eps = desc, inverse, eps
        ctx.save_for_backward(x, out)
        return out, dlogp

    @staticmethod
    def backward(ctx, g_y, g_dlogp):
        x, y = ctx.saved_tensors
        return cdf_backward(x, y, ctx.desc, ctx.inverse, ctx.eps, g_y, g_dlogp), None, None, None


def cdf_backward(x, y, desc, inverse, eps, g_y, g_dlogp):
    """bgk_cdf_backward: gradient w.r.t. the map's input x from the cotangents of its output y (contiguous [B, d]) and of its log-det"""
    x2, ldx = _lib.rowmajor(x)
    gy2, ldgy = _lib.rowmajor(g_y.contiguous())
    B, d = x2.shape
    g_x = torch.empty((B, d), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = _lib.lib().bgk_cdf_backward(_lib.ptr(x2), ldx, _lib.ptr(y), d, _lib.ptr(desc), B, d, int(inverse),
                                         int(eps is not None), float(eps or 0.0), _lib.ptr(gy2), ldgy,
                                         _lib.ptr(g_dlogp.reshape(-1).contiguous()), _lib.ptr(g_x), d, _lib.stream_ptr(x.device))
    _lib.check(st, "bgk_cdf_backward")
    return g_x


def _launch(x, desc, inverse, eps, acc=None):
    x2, ldx = _lib.rowmajor(x)
    B, d = x2.shape
    out = torch.empty((B, d), dtype=torch.float32, device=x.device)
    if acc is None:
        dlogp, accumulate = torch.empty((B,), dtype=torch.float32, device=x.device), 0
    else:                      # the pass's running log-det buffer (flow._LogDetAcc): the kernel adds to it
        dlogp, accumulate = acc.peek()
    with torch.cuda.device(x.device):
        st = _lib.lib().bgk_cdf_transform(_lib.ptr(x2), ldx, _lib.ptr(desc), B, d, int(inverse),
                                          int(eps is not None), float(eps or 0.0), _lib.ptr(out), d,
                                          _lib.ptr(dlogp), int(bool(accumulate)), _lib.stream_ptr(x.device))
    _lib.check(st, "bgk_cdf_transform")
    if acc is not None:
        acc.commit()
        return out, acc
    return out, dlogp[:, None]


class CDFTransform(Flow):
    """x -> cdf(x) in [0,1] with log-det = log_prob(x); values are clamped to [eps, 1-eps] and
    log-dets to >= -1/eps exactly like the reference."""

    def __init__(self, distribution, eps=1e-7):
        super().__init__()
        self.distribution = distribution
        self._eps = eps
        self._desc_cache = {}

    _bgk_acc = True

    def _kernel(self, x, inverse, acc=None):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
            return None
        grad = torch.is_grad_enabled()
        # the descriptor is a host snapshot of the marginal's parameters: rebuilt whenever one of them was replaced or written
        # (load_state_dict, optimiser steps, .to()), keyed on storage + version counter; None for a learnable marginal whose
        # parameters need gradients (-> the distribution's own torch ops)
        desc = self.kernel_descriptor(x.shape[-1], x.device)
        if desc is None:
            return None
        if grad and x.requires_grad:
            return _CdfFn.apply(x, desc, inverse, self._eps)
        return _launch(x, desc, inverse, self._eps, acc=acc)

    def kernel_descriptor(self, d, device):
        """the [d, 6] device descriptor of this layer's marginal for the kernels (bgk_cdf_transform, bgk_icdf_ic2xyz), or None
        when the marginal is not one of the supported kinds or its parameters are being trained"""
        src = _source_tensors(self.distribution)
        if torch.is_grad_enabled() and any(t.requires_grad for t in src):
            return None
        key = (d, str(device), tuple((*param_state_key(t), str(t.device)) for t in src))
        if self._desc_cache.get("key") != key:
            desc = _descriptor(self.distribution, d)
            self._desc_cache.update({"key": key, "desc": None if desc is None else desc.to(device)})
        return self._desc_cache["desc"]

    def tail_descriptor(self, d, device):
        """the [d, 20] device descriptor of the sampling-tail kernel (bgk_icdf_ic2xyz_reg), cached like ``kernel_descriptor``"""
        src = _source_tensors(self.distribution)
        if torch.is_grad_enabled() and any(t.requires_grad for t in src):
            return None
        key = (d, str(device), tuple((*param_state_key(t), str(t.device)) for t in src))
        if self._desc_cache.get("key20") != key:
            desc = _tail_descriptor(self.distribution, d)
            self._desc_cache["key20"], self._desc_cache["desc20"] = key, (None if desc is None else desc.to(device))
        return self._desc_cache["desc20"]

    def invalidate_kernel_cache(self):
        """forget the cached descriptor (after in-place edits through ``.data``, which bump no version counter)"""
        self._desc_cache = {}

    def _forward(self, x, *args, **kwargs):
        fast = self._kernel(x, False, acc=kwargs.get(ACC_KW))
        if fast is not None:
            return fast
        y = self.distribution.cdf(x)
        if self._eps is not None:
            y = y.clamp(self._eps, 1.0 - self._eps)
        logdet = self.distribution.log_prob(x)
        if self._eps is not None:
            logdet = logdet.clamp_min(-1 / self._eps)
        return y, logdet.sum(dim=-1, keepdim=True)

    def _inverse(self, x, *args, **kwargs):
        fast = self._kernel(x, True, acc=kwargs.get(ACC_KW))
        if fast is not None:
            return fast
        if self._eps is not None:
            x = x.clamp(self._eps, 1.0 - self._eps)
        y = self.distribution.icdf(x)
        logdet = -self.distribution.log_prob(y)
        if self._eps is not None:
            logdet = logdet.clamp_min(-1 / self._eps)
        return y, logdet.sum(dim=-1, keepdim=True)


class DistributionTransferFlow(SequentialFlow):
    """Samples of ``source_distribution`` -> samples of ``target_distribution``: cdf of the source followed by the icdf of the target
    (bgflow/nn/flow/cdf.py:49-63); two launches of bgk_cdf_transform for the supported marginals."""

    def __init__(self, source_distribution, target_distribution, eps=1e-7):
        super().__init__([CDFTransform(source_distribution, eps=eps), InverseFlow(CDFTransform(target_distribution, eps=eps))])


class ConstrainGaussianFlow(Flow):
    """Squeeze a Gaussian variable N(mu, sigma) into [lower_bound, upper_bound]: Gaussian cdf, then the icdf of the truncated
    Gaussian N(mu_out or mu, sigma_out or sigma) on that interval; the forward output is clamped to the interval against
    round-off (bgflow/nn/flow/cdf.py:66-121; same argument names and defaults)."""
    _bgk_acc = True

    def __init__(self, mu, sigma=torch.tensor(1.0), lower_bound=0.0, upper_bound=np.inf, assert_range=True, mu_out=None,
                 sigma_out=None, eps=1e-7):
        super().__init__()
        from .distributions import TruncatedNormalDistribution
        lo, hi = float(lower_bound), float(upper_bound)
        source = torch.distributions.Normal(mu, sigma.to(mu))
        target = TruncatedNormalDistribution(
            mu=mu if mu_out is None else mu_out.to(mu), sigma=sigma if sigma_out is None else sigma_out.to(mu),
            lower_bound=lo * torch.ones_like(mu), upper_bound=hi * torch.ones_like(mu), assert_range=assert_range)
        self._trafo = DistributionTransferFlow(source, target, eps)
        self._lower_bound, self._upper_bound = lo, hi

    def _forward(self, x, *args, **kwargs):
        y, dlogp = self._trafo.forward(x, *args, **kwargs)
        return y.clamp(self._lower_bound, self._upper_bound), dlogp

    def _inverse(self, x, *args, **kwargs):
        return self._trafo.forward(x, *args, **kwargs, inverse=True)
