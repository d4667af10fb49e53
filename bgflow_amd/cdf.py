"""Domain-mapping layer ``CDFTransform`` (bgflow/nn/flow/cdf.py:13-46).

For the marginals the builder installs (truncated normal, normal, uniform; factory/icmarginals.py:41-77)
and f32 HIP tensors without autograd, one hand-written kernel (bgk_cdf_transform) replaces the ~10
elementwise aten launches + row reduction; any other distribution, or a call that needs gradients,
runs the distribution's own cdf / icdf / log_prob (stock PyTorch-ROCm ops, differentiable)."""
import numpy as np
import torch

from . import _lib
from .flow import Flow

__all__ = ["CDFTransform"]


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
        return desc.contiguous()
    except (AttributeError, RuntimeError):
        return None


class CDFTransform(Flow):
    """x -> cdf(x) in [0,1] with log-det = log_prob(x); values are clamped to [eps, 1-eps] and
    log-dets to >= -1/eps exactly like the reference."""

    def __init__(self, distribution, eps=1e-7):
        super().__init__()
        self.distribution = distribution
        self._eps = eps
        self._desc_cache = {}

    def _kernel(self, x, inverse):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
            return None
        if torch.is_grad_enabled() and x.requires_grad:
            return None
        d = x.shape[-1]
        key = (d, str(x.device))
        if key not in self._desc_cache:
            desc = _descriptor(self.distribution, d)
            self._desc_cache[key] = None if desc is None else desc.to(x.device)
        desc = self._desc_cache[key]
        if desc is None:
            return None
        x2, ldx = _lib.rowmajor(x)
        B = x2.shape[0]
        out = torch.empty((B, d), dtype=torch.float32, device=x.device)
        dlogp = torch.empty((B,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            st = _lib.lib().bgk_cdf_transform(_lib.ptr(x2), ldx, _lib.ptr(desc), B, d, int(inverse),
                                              int(self._eps is not None), float(self._eps or 0.0), _lib.ptr(out), d,
                                              _lib.ptr(dlogp), 0, _lib.stream_ptr(x.device))
        _lib.check(st, "bgk_cdf_transform")
        return out, dlogp[:, None]

    def _forward(self, x, *args, **kwargs):
        fast = self._kernel(x, False)
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
        fast = self._kernel(x, True)
        if fast is not None:
            return fast
        if self._eps is not None:
            x = x.clamp(self._eps, 1.0 - self._eps)
        y = self.distribution.icdf(x)
        logdet = -self.distribution.log_prob(y)
        if self._eps is not None:
            logdet = logdet.clamp_min(-1 / self._eps)
        return y, logdet.sum(dim=-1, keepdim=True)
