"""Priors and targets that bracket the flow (stock PyTorch-ROCm ops).

Minimal, API-compatible subset of bgflow/distribution: ``Energy`` / ``Sampler`` protocol
(energy/base.py:124-138, sampling/base.py:32-55), ``NormalDistribution`` (normal.py:17-92),
``TruncatedNormalDistribution`` (normal.py:95-250: cdf / icdf / log_prob only, as used by the
domain-mapping layers), ``UniformDistribution`` / ``SloppyUniform`` (distributions.py:71-117),
``ProductDistribution`` (product.py:13-117) and ``DoubleWellEnergy`` (energy/double_well.py:10-22).
"""
import numpy as np
import torch

from .utils import pack_tensor_in_tuple

__all__ = [
    "Energy", "Sampler", "NormalDistribution", "TruncatedNormalDistribution", "SloppyUniform",
    "UniformDistribution", "ProductDistribution", "DoubleWellEnergy",
]


def _shapes(dim):
    if isinstance(dim, int):
        return [torch.Size([dim])]
    if isinstance(dim, torch.Size) or (len(dim) > 0 and isinstance(dim[0], int)):
        return [torch.Size(dim)]
    return [torch.Size(s) if not isinstance(s, int) else torch.Size([s]) for s in dim]


class Energy(torch.nn.Module):
    """Dimensionless energy u(x) with ``energy(*xs, temperature=1.0) -> [batch, 1]``."""

    def __init__(self, dim):
        super().__init__()
        self._event_shapes = _shapes(dim)

    @property
    def dim(self):
        if len(self._event_shapes) > 1:
            raise ValueError("This energy instance is defined for multiple events.")
        if len(self._event_shapes[0]) > 1:
            raise ValueError("This energy instance is defined on multidimensional events.")
        return self._event_shapes[0][0]

    @property
    def event_shape(self):
        if len(self._event_shapes) > 1:
            raise ValueError("This energy instance is defined for multiple events.")
        return self._event_shapes[0]

    @property
    def event_shapes(self):
        return self._event_shapes

    def _energy(self, *xs, **kwargs):
        raise NotImplementedError()

    def energy(self, *xs, temperature=1.0, **kwargs):
        assert len(xs) == len(self._event_shapes), \
            f"Expected {len(self._event_shapes)} arguments but only received {len(xs)}"
        for x, s in zip(xs, self._event_shapes):
            assert x.shape[-len(s):] == s, f"event shape mismatch: {x.shape} vs {s}"
        return self._energy(*xs, **kwargs) / temperature

    def force(self, *xs, temperature=1.0, **kwargs):
        xs = [x.requires_grad_(True) for x in xs]
        e = self.energy(*xs, temperature=temperature, **kwargs)
        grads = torch.autograd.grad(e.sum(), xs)
        return -grads[0] if len(grads) == 1 else tuple(-g for g in grads)


class Sampler(torch.nn.Module):
    """``sample(n, temperature=1.0)``; subclasses implement ``_sample`` and optionally
    ``_sample_with_temperature``."""

    def _sample(self, n_samples, *args, **kwargs):
        raise NotImplementedError()

    def _sample_with_temperature(self, n_samples, temperature, *args, **kwargs):
        raise NotImplementedError()

    def sample(self, n_samples, temperature=1.0, *args, **kwargs):
        if isinstance(temperature, float) and temperature == 1.0:
            return self._sample(n_samples, *args, **kwargs)
        return self._sample_with_temperature(n_samples, temperature, *args, **kwargs)


class _NormalEnergyFn(torch.autograd.Function):
    """u(x) = 0.5 |x - mean|^2 / T + log Z on bgk_normal_energy; gradient w.r.t. x on bgk_normal_energy_backward"""

    @staticmethod
    def forward(ctx, x, mean, temperature, log_z):
        from . import _lib
        x2, ldx = _lib.rowmajor(x)
        if mean is not None:
            mean = mean.detach().to(device=x.device).contiguous()
        B, d = x2.shape
        u = torch.empty(B, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            st = _lib.lib().bgk_normal_energy(_lib.ptr(x2), ldx, _lib.ptr(mean), d, B, temperature, log_z, _lib.ptr(u),
                                              _lib.stream_ptr(x.device))
        _lib.check(st, "bgk_normal_energy")
        ctx.save_for_backward(x2, mean if mean is not None else x2.new_empty(0))
        ctx.cfg = (ldx, mean is not None, temperature)
        return u[:, None]

    @staticmethod
    def backward(ctx, g_u):
        from . import _lib
        x2, mean = ctx.saved_tensors
        ldx, has_mean, temperature = ctx.cfg
        B, d = x2.shape
        g = g_u.reshape(-1).to(torch.float32).contiguous()
        g_x = torch.empty((B, d), dtype=torch.float32, device=x2.device)
        with torch.cuda.device(x2.device):
            st = _lib.lib().bgk_normal_energy_backward(_lib.ptr(x2), ldx, _lib.ptr(mean) if has_mean else None, d, B, temperature,
                                                       _lib.ptr(g), _lib.ptr(g_x), d, _lib.stream_ptr(x2.device))
        _lib.check(st, "bgk_normal_energy_backward")
        return g_x, None, None, None


class NormalDistribution(Energy, Sampler):
    """Isotropic (optionally shifted) normal; ``cov`` support is limited to diagonalisable
    covariances like the reference (normal.py:17-92)."""

    def __init__(self, dim, mean=None, cov=None):
        super().__init__(dim=dim)
        self._has_mean = mean is not None
        if self._has_mean:
            assert len(mean.shape) == 1 and mean.shape[-1] == self.dim
            self.register_buffer("_mean", mean)
        else:
            self.register_buffer("_mean", torch.zeros(self.dim))
        self._has_cov = False
        if cov is not None:
            self.set_cov(cov)

    def set_cov(self, cov):
        assert cov.shape == (self.dim, self.dim), "`cov` must have dimension `[dim, dim]`"
        diag, rot = torch.linalg.eigh(cov)
        diag = diag + 1e-6
        assert torch.all(diag > 0), "`cov` must be positive definite"
        self._has_cov = True
        self.register_buffer("_log_diag", diag.log().unsqueeze(0))
        self.register_buffer("_rot", rot)

    def _log_Z(self, temperature=1.0):
        t = torch.as_tensor(temperature, dtype=self._mean.dtype, device=self._mean.device)
        log_z = self.dim / 2 * torch.log(2 * np.pi * t)
        if self._has_cov:
            log_z = log_z + 0.5 * self._log_diag.sum()
        return log_z

    def energy(self, x, temperature=1.0):
        if (not self._has_cov and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and self._mean.dtype == torch.float32
                and isinstance(temperature, (int, float)) and temperature > 0 and x.shape[0] > 0):
            # one launch (bgk_normal_energy) instead of sub / div / pow / sum / add, one for the gradient instead of five
            mean = self._mean if self._has_mean else None
            return _NormalEnergyFn.apply(x, mean, float(temperature), float(self.dim / 2 * np.log(2 * np.pi * temperature)))
        if self._has_mean:
            x = x - self._mean
        if self._has_cov:
            x = (x @ self._rot) * torch.exp(-0.5 * self._log_diag)
        x = x / (temperature ** 0.5)
        return 0.5 * x.pow(2).sum(dim=-1, keepdim=True) + self._log_Z(temperature)

    def _sample_with_temperature(self, n_samples, temperature=1.0):
        s = torch.randn(n_samples, self.dim, dtype=self._mean.dtype, device=self._mean.device)
        if self._has_cov:
            s = (s * torch.exp(0.5 * self._log_diag)) @ self._rot.t()
        s = s * (temperature.sqrt() if isinstance(temperature, torch.Tensor) else temperature ** 0.5)
        if self._has_mean:
            s = s + self._mean
        return s

    def _sample(self, n_samples):
        return self._sample_with_temperature(n_samples)


class TruncatedNormalDistribution(Energy, Sampler):
    """Normal restricted to [lower_bound, upper_bound] per dimension (normal.py:95-250)."""

    def __init__(self, mu, sigma=torch.tensor(1.0), lower_bound=torch.tensor(0.0),
                 upper_bound=torch.tensor(np.inf), assert_range=True, sampling_method="icdf",
                 is_learnable=False):
        for t in (mu, sigma, lower_bound, upper_bound):
            assert type(t) is torch.Tensor
            assert t.shape in (torch.Size([]), (1,), mu.shape)
        super().__init__(dim=mu.shape)
        if is_learnable:
            self._mu = torch.nn.Parameter(mu)
            self._logsigma = torch.nn.Parameter(torch.log(sigma.to(mu)))
        else:
            self.register_buffer("_mu", mu)
            self.register_buffer("_logsigma", torch.log(sigma.to(mu)))
        self.register_buffer("_upper_bound", upper_bound.to(mu))
        self.register_buffer("_lower_bound", lower_bound.to(mu))
        self.assert_range = assert_range
        if sampling_method != "icdf":
            raise ValueError(f'Unknown / unsupported sampling method "{sampling_method}"')
        std = torch.distributions.Normal(torch.tensor(0.0).to(mu), torch.tensor(1.0).to(mu))
        alpha = (self._lower_bound - self._mu) / self._sigma
        beta = (self._upper_bound - self._mu) / self._sigma
        self.register_buffer("_cdf_lower_bound", std.cdf(alpha.detach()))
        self.register_buffer("_cdf_upper_bound", std.cdf(beta.detach()))

    @property
    def _sigma(self):
        return torch.exp(self._logsigma)

    @property
    def _standard_normal(self):
        return torch.distributions.Normal(torch.zeros((), dtype=self._mu.dtype, device=self._mu.device),
                                          torch.ones((), dtype=self._mu.dtype, device=self._mu.device))

    Z = property(lambda self: self._cdf_upper_bound - self._cdf_lower_bound)
    upper_bound = property(lambda self: self._upper_bound)
    lower_bound = property(lambda self: self._lower_bound)
    mu = property(lambda self: self._mu)
    sigma = property(lambda self: self._sigma)

    def _sample(self, n_samples):
        return self._sample_with_temperature(n_samples, 1)

    def _sample_with_temperature(self, n_samples, temperature):
        sigma = self._sigma * np.sqrt(temperature)
        u = torch.rand(n_samples, *self.event_shape, dtype=self._mu.dtype, device=self._mu.device)
        r = self.Z * u + self._cdf_lower_bound
        return self._standard_normal.icdf(r) * sigma + self._mu

    def _energy(self, x):
        e = ((x - self._mu) / self._sigma) ** 2
        if self.assert_range:
            if (x < self._lower_bound).any() or (x > self._upper_bound).any():
                raise ValueError("input out of bounds")
        else:
            e = torch.where((x < self._lower_bound) | (x > self._upper_bound), torch.full_like(e, np.inf), e)
        return 0.5 * e.sum(dim=-1, keepdim=True)

    def icdf(self, x):
        return self._standard_normal.icdf(self.Z * x + self._cdf_lower_bound) * self._sigma + self._mu

    def cdf(self, x):
        return (self._standard_normal.cdf((x - self._mu) / self._sigma) - self._cdf_lower_bound) / self.Z

    def log_prob(self, x):
        return self._standard_normal.log_prob((x - self._mu) / self._sigma) - torch.log(self.Z * self._sigma)


class SloppyUniform(torch.nn.Module):
    """Uniform[low, high] whose support check tolerates ``tol`` (distributions.py:71-97)."""

    def __init__(self, low, high, validate_args=None, tol=1e-5):
        super().__init__()
        self.register_buffer("low", low)
        self.register_buffer("high", high)
        self.tol = tol
        self.validate_args = validate_args

    def _uniform(self):
        return torch.distributions.Uniform(self.low, self.high, validate_args=False)

    def cdf(self, x):
        return ((x - self.low) / (self.high - self.low)).clamp(0, 1)

    def icdf(self, u):
        return self.low + u * (self.high - self.low)

    def log_prob(self, x):
        inside = (x >= self.low - self.tol) & (x <= self.high + self.tol)
        lp = -torch.log(self.high - self.low).expand_as(x)
        return torch.where(inside, lp, torch.full_like(lp, -np.inf))

    def sample(self, sample_shape=torch.Size()):
        return self._uniform().sample(sample_shape)


class UniformDistribution(Energy, Sampler):
    """Independent uniform prior (distributions.py:100-117)."""

    def __init__(self, low, high, tol=1e-5, validate_args=None, n_event_dims=1):
        super().__init__(dim=low.shape[-n_event_dims:] if n_event_dims > 0 else low.shape)
        self.uniform = SloppyUniform(low, high, validate_args, tol=tol)

    def _energy(self, x):
        # The reference evaluates -log_prob and, whenever a value lies outside the (tolerant) support, falls back to the energy
        # of a fresh in-support sample for the whole batch (distributions.py:108-114): in either case the result is the
        # constant sum_j log(high_j - low_j) -- finite by construction (never +inf), and no host-side support check is needed.
        const = torch.log(self.uniform.high - self.uniform.low).expand(x.shape[-1:]).sum()
        return const.expand(*x.shape[:-1], 1).to(x.dtype)

    def _sample(self, n_samples):
        return self.uniform.sample(torch.Size([n_samples]))

    def _sample_with_temperature(self, n_samples, temperature):
        return self._sample(n_samples)


class ProductDistribution(Energy, Sampler):
    """Independent product of distributions over several tensors (product.py:13-117)."""

    def __init__(self, components, cat_dim=None):
        shapes = [c.event_shapes[0] if cat_dim is None else c.event_shapes[0] for c in components]
        super().__init__(dim=shapes if cat_dim is None else [sum(s[0] for s in shapes)])
        self._components = torch.nn.ModuleList(components)
        self._cat_dim = cat_dim
        self._lengths = [s[0] for s in shapes]

    def energy(self, *xs, temperature=1.0):
        if self._cat_dim is not None:
            xs = torch.split(xs[0], self._lengths, dim=self._cat_dim)
        # like the reference (product.py:36-44 + energy/base.py:124-146): the components are evaluated at T = 1 and their SUM is
        # divided by the temperature
        es = [c.energy(x) for c, x in zip(self._components, xs)]
        return sum(es[1:], es[0]) / temperature

    def sample(self, n_samples, temperature=1.0):
        parts = tuple(c.sample(n_samples, temperature=temperature) for c in self._components)
        return torch.cat(parts, dim=self._cat_dim) if self._cat_dim is not None else parts


class DoubleWellEnergy(Energy):
    """u(x) = a x0 + b x0^2 + c x0^4 + 0.5 |x_rest|^2 (energy/double_well.py:10-22)."""

    def __init__(self, dim, a=0, b=-4.0, c=1.0):
        super().__init__(dim)
        self._a, self._b, self._c = a, b, c

    def _energy(self, x):
        d = x[..., [0]]
        v = x[..., 1:]
        return self._a * d + self._b * d.pow(2) + self._c * d.pow(4) + 0.5 * v.pow(2).sum(dim=-1, keepdim=True)
