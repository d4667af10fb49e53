"""Priors and targets that bracket the flow (stock PyTorch-ROCm ops).

Minimal, API-compatible subset of bgflow/distribution: ``Energy`` / ``Sampler`` protocol
(energy/base.py:124-138, sampling/base.py:32-55), ``NormalDistribution`` (normal.py:17-92),
``TruncatedNormalDistribution`` (normal.py:95-250: cdf / icdf / log_prob only, as used by the
domain-mapping layers), ``UniformDistribution`` / ``SloppyUniform`` (distributions.py:71-117),
``ProductDistribution`` (product.py:13-117) and ``DoubleWellEnergy`` (energy/double_well.py:10-22).
"""
import numpy as np
import torch

from .utils import pack_tensor_in_tuple  # noqa: F401  (re-exported: bgflow.distribution modules expose it)

__all__ = [
    "Energy", "Sampler", "NormalDistribution", "TruncatedNormalDistribution", "SloppyUniform",
    "UniformDistribution", "ProductDistribution", "DoubleWellEnergy", "kernel_energy", "kl_loss_sums", "philox_sample",
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


# ---- kernel-backed energies (csrc/bgk_energy.hip) ---------------------------------------------------------------------------
# A "field" = one tensor of a sample with an energy of one of the kernel's kinds: (kind, mean tensor or None, (a, b, c), constant
# added INSIDE the temperature division).  A distribution that can describe itself that way implements ``_kernel_fields()``.
def _fields_args(specs, xs):
    """ctypes tables for bgk_energy_fields(_backward): specs = [(kind, param, (a, b, c), c_in)], xs = matching [B, d] f32 HIP tensors"""
    import ctypes
    from . import _lib
    n = len(specs)
    rows = [_lib.rowmajor(x) for x in xs]
    params = [None if sp[1] is None else sp[1].detach().to(device=xs[0].device, dtype=torch.float32).contiguous() for sp in specs]
    X = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in rows])
    LD = (ctypes.c_int64 * n)(*[ld for _, ld in rows])
    D = (ctypes.c_int32 * n)(*[t.shape[1] for t, _ in rows])
    K = (ctypes.c_int32 * n)(*[sp[0] for sp in specs])
    P = (ctypes.c_void_p * n)(*[None if p is None else p.data_ptr() for p in params])
    C = (ctypes.c_float * (3 * n))(*[float(v) for sp in specs for v in sp[2]])
    return (X, LD, D, K, P, C, n), (rows, params)


def _fields_ok(xs, dims):
    return (len(xs) == len(dims) and all(torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                                         and x.shape[1] == d and x.shape[0] == xs[0].shape[0] for x, d in zip(xs, dims))
            and xs[0].shape[0] > 0 and len(xs) <= 8)


class _EnergyFieldsFn(torch.autograd.Function):
    """u = (sum_f e_f(x_f) + c_in) / T + c_out on bgk_energy_fields; gradients of all fields in one launch of
    bgk_energy_fields_backward"""

    @staticmethod
    def forward(ctx, specs, temperature, c_in, c_out, *xs):
        from . import _lib
        args, keep = _fields_args(specs, xs)
        B, dev = xs[0].shape[0], xs[0].device
        u = torch.empty(B, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_energy_fields(*args, B, float(temperature), float(c_in), float(c_out), _lib.ptr(u), None, 0, None, 0, None,
                                              _lib.stream_ptr(dev))
        _lib.check(st, "bgk_energy_fields")
        ctx.save_for_backward(*[t for t, _ in keep[0]])
        ctx.cfg = (specs, float(temperature))
        return u[:, None]

    @staticmethod
    def backward(ctx, g_u):
        import ctypes
        from . import _lib
        specs, temperature = ctx.cfg
        xs = ctx.saved_tensors
        args, keep = _fields_args(specs, xs)
        B, dev, n = xs[0].shape[0], xs[0].device, len(xs)
        g = g_u.reshape(-1).to(torch.float32).contiguous()
        need = ctx.needs_input_grad[4:]
        gx = [torch.empty_like(x, memory_format=torch.contiguous_format) if nd and sp[0] != 2 else None for x, nd, sp in zip(xs, need, specs)]
        G = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in gx])
        LG = (ctypes.c_int64 * n)(*[0 if t is None else t.shape[1] for t in gx])
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_energy_fields_backward(*args, B, temperature, _lib.ptr(g), None, None, None, 0, None, G, LG,
                                                       _lib.stream_ptr(dev))
        _lib.check(st, "bgk_energy_fields_backward")
        return (None, None, None, None, *[t if t is not None else (torch.zeros_like(x) if nd else None) for t, x, nd in zip(gx, xs, need)])


class _KLSumsFn(torch.autograd.Function):
    """[sum_b (u(x_b) - dlogp_b), number of samples kept] (f64 [2]) with the loss partial sums formed by the energy kernel itself;
    backward: one launch for the gradients of every field and of dlogp"""

    @staticmethod
    def forward(ctx, specs, temperature, c_in, c_out, drop_nonfinite, dlogp, *xs):
        from . import _lib
        args, keep = _fields_args(specs, xs)
        B, dev = xs[0].shape[0], xs[0].device
        u = torch.empty(B, dtype=torch.float32, device=dev)
        dl = dlogp.detach().reshape(-1).to(torch.float32).contiguous()
        nblk = 2048
        partial = torch.empty((nblk, 2), dtype=torch.float32, device=dev)
        sums = torch.empty(2, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_energy_fields(*args, B, float(temperature), float(c_in), float(c_out), _lib.ptr(u), _lib.ptr(dl),
                                              int(bool(drop_nonfinite)), _lib.ptr(partial), nblk, _lib.ptr(sums), _lib.stream_ptr(dev))
        _lib.check(st, "bgk_energy_fields")
        ctx.save_for_backward(u, dl, *[t for t, _ in keep[0]])
        ctx.cfg = (specs, float(temperature), bool(drop_nonfinite), dlogp.shape)
        u2 = u[:, None]
        ctx.mark_non_differentiable(u2)            # (the mark must sit on the tensor that is returned, not on its base)
        return sums, u2

    @staticmethod
    def backward(ctx, g_sums, _g_u):
        import ctypes
        from . import _lib
        specs, temperature, drop, dl_shape = ctx.cfg
        u, dl, *xs = ctx.saved_tensors
        args, keep = _fields_args(specs, xs)
        B, dev, n = xs[0].shape[0], xs[0].device, len(xs)
        gs = g_sums[0:1].to(torch.float32).contiguous()
        need = ctx.needs_input_grad[6:]
        gx = [torch.empty_like(x, memory_format=torch.contiguous_format) if nd and sp[0] != 2 else None for x, nd, sp in zip(xs, need, specs)]
        g_dl = torch.empty(B, dtype=torch.float32, device=dev) if ctx.needs_input_grad[5] else None
        G = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in gx])
        LG = (ctypes.c_int64 * n)(*[0 if t is None else t.shape[1] for t in gx])
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_energy_fields_backward(*args, B, temperature, None, _lib.ptr(gs), _lib.ptr(u), _lib.ptr(dl), int(drop),
                                                       _lib.ptr(g_dl), G, LG, _lib.stream_ptr(dev))
        _lib.check(st, "bgk_energy_fields_backward")
        return (None, None, None, None, None, None if g_dl is None else g_dl.reshape(dl_shape),
                *[t if t is not None else (torch.zeros_like(x) if nd else None) for t, x, nd in zip(gx, xs, need)])


def _kernel_plan(dist, temperature):
    """``dist._kernel_fields(temperature)`` -- unless a subclass overrides ``energy`` / ``_energy`` of the class that describes the
    kernel fields: such an override (a custom target, a clipped or regularised energy) must run through its own code"""
    describe = getattr(dist, "_kernel_fields", None)
    if describe is None:
        return None
    owner = next(c for c in type(dist).__mro__ if "_kernel_fields" in c.__dict__)
    for name in ("energy", "_energy"):
        if getattr(type(dist), name, None) is not getattr(owner, name, None):
            return None
    return describe(temperature)


def kernel_energy(dist, xs, temperature=1.0):
    """``dist.energy(*xs, temperature)`` as ONE launch when ``dist`` can describe itself by kernel fields and the inputs are 2-d f32
    HIP tensors, else None"""
    plan = _kernel_plan(dist, temperature)
    if plan is None:
        return None
    specs, dims, c_in, c_out, t_eff = plan
    if not (_fields_ok(xs, dims) and isinstance(temperature, (int, float)) and temperature > 0):
        return None
    return _EnergyFieldsFn.apply(specs, t_eff, c_in, c_out, *xs)


def kl_loss_sums(target, xs, dlogp, temperature=1.0, drop_nonfinite=False):
    """(sums, u): sums = f64 [2] = [sum_b (u_target(x_b) - dlogp_b), samples kept] with autograd to x and dlogp, formed inside the
    target-energy kernel (no per-sample loss tensor, no isfinite / where / sum launches); None if the target has no kernel fields"""
    plan = _kernel_plan(target, temperature)
    if plan is None:
        return None
    specs, dims, c_in, c_out, t_eff = plan
    if not (_fields_ok(xs, dims) and isinstance(temperature, (int, float)) and temperature > 0 and torch.is_tensor(dlogp)
            and dlogp.is_cuda and dlogp.numel() == xs[0].shape[0]):
        return None
    return _KLSumsFn.apply(specs, t_eff, c_in, c_out, bool(drop_nonfinite), dlogp, *xs)


# ---- counter-based prior sampling (csrc/bgk_philox.hip), opt-in -----------------------------------------------------------------
def philox_sample(fields, n_samples, device, seed, offset, row0=0, want_energy=False, c_out=0.0):
    """fields = [(kind, d, p0, p1, scale, e_const)] (kind 0 uniform on [p0, p1], 1 normal p0 + scale n) -> (tensors, energy or None)"""
    import ctypes
    from . import _lib
    n = len(fields)
    outs = [torch.empty((n_samples, f[1]), dtype=torch.float32, device=device) for f in fields]
    if n_samples == 0:
        return outs, (torch.zeros((0, 1), dtype=torch.float32, device=device) if want_energy else None)
    prm = [[None if t is None else t.detach().to(device=device, dtype=torch.float32).contiguous() for t in (f[2], f[3])] for f in fields]
    O = (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs])
    LD = (ctypes.c_int64 * n)(*[t.shape[1] for t in outs])
    D = (ctypes.c_int32 * n)(*[f[1] for f in fields])
    K = (ctypes.c_int32 * n)(*[f[0] for f in fields])
    P0 = (ctypes.c_void_p * n)(*[None if p[0] is None else p[0].data_ptr() for p in prm])
    P1 = (ctypes.c_void_p * n)(*[None if p[1] is None else p[1].data_ptr() for p in prm])
    SC = (ctypes.c_float * n)(*[float(f[4]) for f in fields])
    EC = (ctypes.c_float * n)(*[float(f[5]) for f in fields])
    energy = torch.empty(n_samples, dtype=torch.float32, device=device) if want_energy else None
    with torch.cuda.device(device):
        st = _lib.lib().bgk_philox_fields(int(seed) & (2 ** 64 - 1), int(offset) & 0xffffffff, int(row0), n, O, LD, D, K, P0, P1, SC, EC,
                                          float(c_out), n_samples, _lib.ptr(energy), _lib.stream_ptr(device))
    _lib.check(st, "bgk_philox_fields")
    return outs, (None if energy is None else energy[:, None])


_PHILOX_STREAMS = [0]          # next automatically assigned stream id (construction-order fallback, see _FusedSampling)
_PHILOX_LIVE = {}              # stream id -> weak reference of the live object that draws from it
PHILOX_MAX_WIDTH = 160         # widest field bgk_philox_fields assembles in its LDS tile (4 waves x 64 rows x d floats)


def _philox_claim(obj, stream):
    """register ``obj`` as the owner of Philox stream ``stream``; a stream another LIVE object already draws from is a collision
    (two priors would produce the same numbers): warn, once per pair"""
    import warnings
    import weakref
    holder = _PHILOX_LIVE.get(stream)
    other = holder() if holder is not None else None
    if other is not None and other is not obj:
        warnings.warn(f"Philox stream {stream} is already used by a live {type(other).__name__}: {type(obj).__name__} will draw the SAME "
                      f"numbers under the same seed (give one of them another id with set_philox_stream)", RuntimeWarning, stacklevel=3)
    _PHILOX_LIVE[stream] = weakref.ref(obj)
    for k in [k for k, r in _PHILOX_LIVE.items() if r() is None]:
        del _PHILOX_LIVE[k]


class _FusedSampling(torch.nn.Module):
    """Opt-in (``sample_fused=True``) sampling on bgk_philox_fields.  Key = ``torch.initial_seed()`` (so ``torch.manual_seed`` still
    selects the stream) mixed with the data-parallel rank AND a per-object stream id: two priors of equal shape in one process draw
    independent numbers.  The id is either given (``set_philox_stream(k)``: reproducible whatever else the process samples; code that
    builds several fused-sampling priors and needs run-to-run identical draws calls it itself -- nothing in this package does) or, on
    first use, the smallest id no live object holds (then it depends on which other fused-sampling objects sampled before).  A
    ``copy.deepcopy`` does not inherit the stream: the copy claims its own id at its first sample (an inherited id would draw the SAME
    numbers as the original, silently).  Offset = a per-object call counter.  Stream id and counter travel in ``state_dict`` once
    the object has sampled (key ``_philox_state``; absent otherwise, so reference state_dicts load unchanged): a resumed run
    continues the stream instead of replaying it; loading an id a live object already holds warns.  The prior energy of a sample
    comes out of the same launch and is handed back by ``energy`` when it is asked about exactly these, unmodified tensors."""
    sample_fused = False

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_philox_state":                      # the copy is a new sampler: its own stream, claimed at its first sample
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def set_philox_stream(self, stream, calls=0):
        """draw from Philox stream ``stream`` (a non-negative int), continuing at call ``calls``"""
        stream = int(stream)
        if stream < 0:
            raise ValueError("set_philox_stream: the stream id is a non-negative integer")
        self.__dict__["_philox_state"] = [stream, int(calls)]
        _philox_claim(self, stream)
        return self

    def _philox_ids(self):
        st = self.__dict__.get("_philox_state")
        if st is None:
            k = _PHILOX_STREAMS[0]
            while k in _PHILOX_LIVE and _PHILOX_LIVE[k]() is not None:       # ids given out by hand / loaded from a checkpoint
                k += 1
            _PHILOX_STREAMS[0] = k + 1
            st = self.__dict__["_philox_state"] = [k, 0]      # [stream id, calls]
            _philox_claim(self, k)
        return st

    def _fused_sample(self, fields, n_samples, device, temperature, c_out=0.0):
        import weakref
        from . import dp
        st = self._philox_ids()
        seed = (dp.rank_seed(torch.initial_seed()) + 0x9E3779B97F4A7C15 * (st[0] + 1)) & (2 ** 64 - 1)
        off, st[1] = st[1], st[1] + 1
        outs, energy = philox_sample(fields, n_samples, device, seed, off, want_energy=True, c_out=c_out)
        # the energy is valid for exactly these tensor objects in exactly this state: weak references (no sample batch is pinned)
        # + their version counters (any in-place edit invalidates the cache)
        self.__dict__["_philox_last"] = ([weakref.ref(t) for t in outs], [t._version for t in outs], float(temperature), energy)
        return outs

    def _fused_energy(self, xs, temperature):
        last = self.__dict__.get("_philox_last")
        if last is None or len(last[0]) != len(xs) or last[2] != float(temperature):
            return None
        for ref, ver, x in zip(last[0], last[1], xs):
            if ref() is not x or x._version != ver or (torch.is_grad_enabled() and x.requires_grad):
                return None                       # another tensor, an edited one, or one whose energy must carry a graph
        return last[3].clone()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        st = self.__dict__.get("_philox_state")
        if st is not None and st[1] > 0:
            destination[prefix + "_philox_state"] = torch.tensor(st, dtype=torch.int64)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        st = state_dict.pop(prefix + "_philox_state", None)
        if st is not None:
            self.__dict__["_philox_state"] = [int(st[0]), int(st[1])]
            _philox_claim(self, int(st[0]))
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class NormalDistribution(Energy, Sampler, _FusedSampling):
    """Isotropic (optionally shifted) normal; ``cov`` support is limited to diagonalisable
    covariances like the reference (normal.py:17-92).  ``sample_fused=True`` (not in the reference): samples come from the
    counter-based kernel generator instead of ``torch.randn``."""

    def __init__(self, dim, mean=None, cov=None, sample_fused=False):
        super().__init__(dim=dim)
        self.sample_fused = sample_fused
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

    def _kernel_fields(self, temperature=1.0):
        """(specs, dims, c_in, c_out, T): u = 0.5 |x - mean|^2 / T + d / 2 log(2 pi T)"""
        if self._has_cov or self._mean.dtype != torch.float32 or not isinstance(temperature, (int, float)) or temperature <= 0:
            return None
        return ([(0, self._mean if self._has_mean else None, (0.0, 0.0, 0.0), 0.0)], [self.dim], 0.0,
                float(self.dim / 2 * np.log(2 * np.pi * temperature)), float(temperature))

    def energy(self, x, temperature=1.0):
        cached = self._fused_energy((x,), temperature) if isinstance(temperature, (int, float)) else None
        if cached is not None:
            return cached
        fast = kernel_energy(self, (x,), temperature) if torch.is_tensor(x) and x.dim() == 2 else None
        if fast is not None:           # one launch (bgk_energy_fields) instead of sub / div / pow / sum / add, one for the gradient
            return fast
        if self._has_mean:
            x = x - self._mean
        if self._has_cov:
            x = (x @ self._rot) * torch.exp(-0.5 * self._log_diag)
        x = x / (temperature ** 0.5)
        return 0.5 * x.pow(2).sum(dim=-1, keepdim=True) + self._log_Z(temperature)

    def _philox_field(self, temperature=1.0):
        """(kind, d, p0, p1, scale, e_const) of bgk_philox_fields for this prior at `temperature`, or None (then torch's generator
        samples: covariances, other dtypes, fields wider than the kernel's LDS tile)"""
        if self.dim > PHILOX_MAX_WIDTH:
            return None
        if self._has_cov or self._mean.dtype != torch.float32 or not isinstance(temperature, (int, float)) or temperature <= 0:
            return None
        return (1, self.dim, self._mean if self._has_mean else None, None, float(temperature) ** 0.5,
                float(self.dim / 2 * np.log(2 * np.pi * temperature)))

    def _sample_with_temperature(self, n_samples, temperature=1.0):
        if self.sample_fused and self._mean.is_cuda:
            field = self._philox_field(temperature)
            if field is not None:
                return self._fused_sample([field], n_samples, self._mean.device, temperature)[0]
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


class UniformDistribution(Energy, Sampler, _FusedSampling):
    """Independent uniform prior (distributions.py:100-117).  ``sample_fused=True`` (not in the reference): samples from the
    counter-based kernel generator."""

    def __init__(self, low, high, tol=1e-5, validate_args=None, n_event_dims=1, sample_fused=False):
        super().__init__(dim=low.shape[-n_event_dims:] if n_event_dims > 0 else low.shape)
        self.uniform = SloppyUniform(low, high, validate_args, tol=tol)
        self.sample_fused = sample_fused

    def _const(self, d):
        return torch.log(self.uniform.high - self.uniform.low).expand(d).sum()

    def _energy(self, x):
        # The reference evaluates -log_prob and, whenever a value lies outside the (tolerant) support, falls back to the energy
        # of a fresh in-support sample for the whole batch (distributions.py:108-114): in either case the result is the
        # constant sum_j log(high_j - low_j) -- finite by construction (never +inf), and no host-side support check is needed.
        # A constant needs no kernel: the [B, 1] result is an expanded view of one device scalar (zero bytes moved).
        const = self._const(x.shape[-1:])
        return const.expand(*x.shape[:-1], 1).to(x.dtype)

    def _kernel_fields(self, temperature=1.0):
        if len(self.event_shape) != 1 or not isinstance(temperature, (int, float)) or temperature <= 0:
            return None
        d = self.event_shape[0]
        return ([(2, None, (self._const_host(d), 0.0, 0.0), 0.0)], [d], 0.0, 0.0, float(temperature))

    def _const_host(self, d):
        """sum_j log(high_j - low_j) as a host float, read back ONCE per state of the bounds (a device-to-host sync per energy /
        sample call would sit in the NLL / KL hot path)"""
        lo, hi = self.uniform.low, self.uniform.high
        key = (d, lo.data_ptr(), hi.data_ptr(), lo._version, hi._version)
        hit = self.__dict__.get("_const_cache")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_const_cache"] = (key, float(self._const((d,))))
        return hit[1]

    def _philox_field(self, temperature=1.0):
        if len(self.event_shape) != 1 or self.uniform.low.dtype != torch.float32 or self.event_shape[0] > PHILOX_MAX_WIDTH:
            return None
        d = self.event_shape[0]
        return (0, d, self.uniform.low.expand(d), self.uniform.high.expand(d), 1.0, self._const_host(d) / float(temperature))

    def _sample(self, n_samples):
        if self.sample_fused and self.uniform.low.is_cuda:
            field = self._philox_field()
            if field is not None:
                return self._fused_sample([field], n_samples, self.uniform.low.device, 1.0)[0]
        return self.uniform.sample(torch.Size([n_samples]))

    def _sample_with_temperature(self, n_samples, temperature):
        return self._sample(n_samples)


class ProductDistribution(Energy, Sampler, _FusedSampling):
    """Independent product of distributions over several tensors (product.py:13-117).  With kernel-describable components (normal
    without cov, uniform, double well) the energy of all tensors is ONE launch (bgk_energy_fields); ``sample_fused=True`` (not in
    the reference) draws all tensors and the prior energy in one launch of the counter-based generator."""

    def __init__(self, components, cat_dim=None, sample_fused=False):
        shapes = [c.event_shapes[0] if cat_dim is None else c.event_shapes[0] for c in components]
        super().__init__(dim=shapes if cat_dim is None else [sum(s[0] for s in shapes)])
        self._components = torch.nn.ModuleList(components)
        self._cat_dim = cat_dim
        self._lengths = [s[0] for s in shapes]
        self.sample_fused = sample_fused

    def _kernel_fields(self, temperature=1.0):
        """the components' fields; product.py:36-44 + energy/base.py:124-146: every component at T = 1 (its own log Z included),
        the SUM divided by the temperature"""
        if self._cat_dim is not None or not isinstance(temperature, (int, float)) or temperature <= 0:
            return None
        specs, dims, c_in = [], [], 0.0
        for c in self._components:
            plan = _kernel_plan(c, 1.0)
            if plan is None or len(plan[0]) != 1:
                return None
            specs.append(plan[0][0]); dims.append(plan[1][0])
            c_in += plan[2] + plan[3]                       # the component's constants belong inside the division by T
        return specs, dims, c_in, 0.0, float(temperature)

    def energy(self, *xs, temperature=1.0):
        if self._cat_dim is not None:
            xs = torch.split(xs[0], self._lengths, dim=self._cat_dim)
        cached = self._fused_energy(xs, temperature) if isinstance(temperature, (int, float)) else None
        if cached is not None:
            return cached
        fast = kernel_energy(self, xs, temperature)
        if fast is not None:
            return fast
        # like the reference (product.py:36-44 + energy/base.py:124-146): the components are evaluated at T = 1 and their SUM is
        # divided by the temperature
        es = [c.energy(x) for c, x in zip(self._components, xs)]
        return sum(es[1:], es[0]) / temperature

    def sample(self, n_samples, temperature=1.0):
        if self.sample_fused and self._cat_dim is None and isinstance(temperature, (int, float)):
            fields = [getattr(c, "_philox_field", lambda t: None)(temperature) for c in self._components]
            dev = next((b.device for b in self.buffers()), None)
            if all(f is not None for f in fields) and dev is not None and dev.type == "cuda" and len(fields) <= 8:
                # the energy the launch reports = product energy of the sample: sum_f [0.5 n^2 + log Z_f(1) / T] (uniform: const / T)
                fixed = []
                for f, c in zip(fields, self._components):
                    if f[0] == 1:
                        f = (*f[:5], float(c.dim / 2 * np.log(2 * np.pi)) / float(temperature))
                    fixed.append(f)
                return tuple(self._fused_sample(fixed, n_samples, dev, temperature))
        parts = tuple(c.sample(n_samples, temperature=temperature) for c in self._components)
        return torch.cat(parts, dim=self._cat_dim) if self._cat_dim is not None else parts


class DoubleWellEnergy(Energy):
    """u(x) = a x0 + b x0^2 + c x0^4 + 0.5 |x_rest|^2 (energy/double_well.py:10-22); one launch forward and one backward on
    bgk_energy_fields for 2-d f32 HIP inputs (cfg 2's target: the KL step has no aten energy ops left)."""

    def __init__(self, dim, a=0, b=-4.0, c=1.0):
        super().__init__(dim)
        self._a, self._b, self._c = a, b, c

    def _kernel_fields(self, temperature=1.0):
        if not isinstance(temperature, (int, float)) or temperature <= 0:
            return None
        return ([(1, None, (float(self._a), float(self._b), float(self._c)), 0.0)], [self.dim], 0.0, 0.0, float(temperature))

    def energy(self, *xs, temperature=1.0, **kwargs):
        fast = kernel_energy(self, xs, temperature) if not kwargs else None
        if fast is not None:
            return fast
        return super().energy(*xs, temperature=temperature, **kwargs)

    def _energy(self, x):
        d = x[..., [0]]
        v = x[..., 1:]
        return self._a * d + self._b * d.pow(2) + self._c * d.pow(4) + 0.5 * v.pow(2).sum(dim=-1, keepdim=True)
