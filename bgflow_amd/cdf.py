"""Domain-mapping layer ``CDFTransform`` (bgflow/nn/flow/cdf.py:13-46): stock PyTorch-ROCm ops for
now (SURVEY.md 8(f) row f-1 -- next in line for fusion into the IC kernel prologue)."""
from .flow import Flow

__all__ = ["CDFTransform"]


class CDFTransform(Flow):
    """x -> cdf(x) in [0,1] with log-det = log_prob(x); values are clamped to [eps, 1-eps] and
    log-dets to >= -1/eps exactly like the reference."""

    def __init__(self, distribution, eps=1e-7):
        super().__init__()
        self.distribution = distribution
        self._eps = eps

    def _forward(self, x, *args, **kwargs):
        y = self.distribution.cdf(x)
        if self._eps is not None:
            y = y.clamp(self._eps, 1.0 - self._eps)
        logdet = self.distribution.log_prob(x)
        if self._eps is not None:
            logdet = logdet.clamp_min(-1 / self._eps)
        return y, logdet.sum(dim=-1, keepdim=True)

    def _inverse(self, x, *args, **kwargs):
        if self._eps is not None:
            x = x.clamp(self._eps, 1.0 - self._eps)
        y = self.distribution.icdf(x)
        logdet = -self.distribution.log_prob(y)
        if self._eps is not None:
            logdet = logdet.clamp_min(-1 / self._eps)
        return y, logdet.sum(dim=-1, keepdim=True)
