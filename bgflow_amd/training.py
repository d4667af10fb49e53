"""Training loop of the accelerated path: ``KLTrainer`` / ``LossReporter`` with the reference's
signatures (bgflow/nn/training/trainers.py:13-205), ``DataSetSampler`` (distribution/sampling/dataset.py) and ``FlatAdam``, an
Adam optimizer over ONE flat parameter / gradient bucket.

What differs from the reference loop, and why:
  * parameters and their gradients live in two contiguous buffers (``FlatAdam`` re-points ``p.data`` / ``p.grad`` at views of
    them): ``zero_grad`` is one memset, the data-parallel gradient all-reduce is ONE collective on the bucket without a
    gather / scatter copy (dp.py), and the optimizer step is one launch (bgk_adam_step);
  * the "found nan in grad; skipping optimization step" rule (trainers.py:198-201) is evaluated on the device
    (bgk_grad_nan_flag -> bgk_adam_step skips itself): no host synchronisation per step; ``skipped_steps()`` polls the count;
  * losses are kept as device scalars and converted lazily by the reporter (the reference's ``assert_numpy`` per iteration
    is a host sync);
  * with an initialised process group the KL / NLL means are global means (one all-reduce of [sum, n], dp.global_mean).
Any other ``torch.optim`` optimizer passed as ``optim`` is used as is (then the NaN check costs a host round trip, like
the reference)."""
import warnings

import numpy as np
import torch

from . import _lib, dp
from .distributions import Sampler

__all__ = ["LossReporter", "KLTrainer", "FlatAdam", "DataSetSampler"]


class DataSetSampler(Sampler, torch.utils.data.Dataset):
    """Sample batches from a data set without replacement, reshuffling when exhausted (distribution/sampling/dataset.py:58-157):
    same constructor (``*data, shuffle, device, dtype``), ``data`` attribute, ``__len__`` / ``__getitem__``, ``reshuffle_`` and
    ``resize_`` as the reference.  The permutation lives on the data's device (the reference indexes with a host numpy
    permutation: one H2D copy of the index batch per call)."""

    def __init__(self, *data, shuffle=True, device=None, dtype=None):
        super().__init__()
        if not all(len(d) == len(data[0]) for d in data):
            raise ValueError("All data items must have the same length.")
        self.data = list(data)
        self._device = data[0].device if device is None else torch.device(device)
        self._dtype = data[0].dtype if dtype is None else dtype
        self._shuffle = shuffle
        self._perm = None
        self._pos = 0

    def __len__(self):
        return self.data[0].shape[0]

    def __getitem__(self, idx):
        return tuple(d[idx] for d in self.data)

    def _new_order(self):
        N, dev = len(self), self.data[0].device
        self._perm = torch.randperm(N, device=dev) if self._shuffle else torch.arange(N, device=dev)
        self._pos = 0

    def _next_indices(self, n):
        N = len(self)
        out = []
        while n > 0:
            if self._perm is None or self._pos >= N:
                self._new_order()
            take = min(n, N - self._pos)
            out.append(self._perm[self._pos:self._pos + take])
            self._pos += take
            n -= take
        return torch.cat(out) if len(out) > 1 else out[0]

    def _sample(self, n_samples, *args, **kwargs):
        idx = self._next_indices(n_samples)
        res = tuple(d[idx].to(device=self._device, dtype=self._dtype) for d in self.data)
        return res[0] if len(res) == 1 else res

    def _sample_with_temperature(self, n_samples, temperature, *args, **kwargs):
        return self._sample(n_samples)

    def reshuffle_(self):
        """new random access order, in place (dataset.py:119-123)"""
        self._new_order()
        return self

    def resize_(self, new_size):
        """Resize the data set to ``new_size`` by drawing rows with replacement and reinitialise the access order; returns the
        row indices used (dataset.py:125-149)"""
        if new_size == len(self):
            return np.arange(len(self))
        indices = np.random.randint(low=0, high=len(self), size=new_size)
        idx_t = torch.as_tensor(indices, device=self.data[0].device)
        self.data = [d[idx_t] for d in self.data]
        self._perm = None
        self._pos = 0
        return indices


class LossReporter:
    """Collects the reported losses (trainers.py:13-45); tensors stay on the device until somebody looks at them."""

    def __init__(self, *labels):
        self._labels = labels
        self._n_reported = len(labels)
        self._raw = [[] for _ in range(self._n_reported)]

    def report(self, *losses):
        assert len(losses) == self._n_reported
        for i in range(self._n_reported):
            v = losses[i]
            self._raw[i].append(v.detach() if torch.is_tensor(v) else v)

    def _np(self, seq):
        return np.array([float(v) for v in seq])

    def print(self, *losses):
        it = len(self._raw[0])
        print(f"{it}\t" + "".join(f"{self._labels[i]}: {float(self._raw[i][-1]):.4f}\t" for i in range(self._n_reported)))

    def losses(self, n_smooth=1):
        x = np.arange(n_smooth, len(self._raw[0]) + 1)
        kernel = np.ones(shape=(n_smooth,)) / n_smooth
        ys = [np.convolve(self._np(raw).reshape(-1), kernel, mode="valid") for raw in self._raw]
        return self._labels, x, ys

    def recent(self, n_recent=1):
        return np.array([self._np(raw[-n_recent:]) for raw in self._raw])


def _bump_versions(params):
    """mark ``params`` as modified in place for torch (a kernel wrote through the flat bucket they are views of)"""
    bump = getattr(torch.autograd.graph, "increment_version", None)
    if bump is None:                      # older torch: the private spelling
        bump = torch._C._increment_version
    for p in params:
        bump(p)


class FlatAdam(torch.optim.Optimizer):
    """Adam (torch.optim.Adam semantics: lr, betas, eps, L2 weight_decay) over one flat f32 bucket on a HIP device."""

    def __init__(self, params, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        assert ps and all(p.is_cuda and p.dtype == torch.float32 for p in ps), "FlatAdam: f32 parameters on a HIP device"
        assert len(self.param_groups) == 1, "FlatAdam: one parameter group"
        dev = ps[0].device
        self._params = ps
        self._param_ids = {id(p) for p in ps}
        n = sum(p.numel() for p in ps)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self._skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        self._step = 0
        self.generation = 0          # number of updates this optimizer has written through the bucket
        off = 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)          # same values, now a view of the bucket
                p.grad = self.grad[off:off + k].view_as(p)
                # kernels that compute parameter gradients themselves (bgk_dense_weight_grad) accumulate straight into the bucket
                p._bgk_grad_dst = self.grad[off:off + k].view_as(p)
                off += k

    def add_param_group(self, param_group):
        if getattr(self, "flat", None) is not None:
            raise ValueError("FlatAdam: the flat bucket is laid out at construction; parameter groups cannot be added afterwards")
        super().add_param_group(param_group)

    def zero_grad(self, set_to_none=False):
        """one memset; the per-parameter .grad views stay attached (autograd accumulates into them in place)"""
        self.grad.zero_()
        off = 0
        for p in self._params:
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + k].view_as(p)
            off += k

    def backward(self, loss, **kwargs):
        """``loss.backward()`` with the fused layers' weight gradients accumulated straight into the bucket (the only place, besides
        KLTrainer.train, where that shortcut is switched on: dense.direct_grad_accumulation)"""
        from .dense import direct_grad_accumulation
        with direct_grad_accumulation():
            loss.backward(**kwargs)

    def allreduce_gradients(self):
        """data-parallel training: ONE all-reduce of the gradient bucket (sum over ranks)"""
        if dp.is_distributed():
            torch.distributed.all_reduce(self.grad, op=torch.distributed.ReduceOp.SUM)

    @torch.no_grad()
    def step(self, closure=None, skip_on_nan=True):
        assert closure is None
        g = self.param_groups[0]
        off = 0
        for p in self._params:       # a gradient tensor autograd swapped in instead of accumulating: bring it home
            k = p.numel()
            if p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                self.grad[off:off + k].copy_(p.grad.reshape(-1))
                p.grad = self.grad[off:off + k].view_as(p)
            off += k
        # `_step` counts the calls; the kernel's Adam time step is `_step - skipped` with the skip count read ON THE DEVICE, so a
        # step skipped for a NaN gradient does not advance the bias corrections (the reference does not call optim.step() then)
        self._step += 1
        lib = _lib.lib()
        dev = self.flat.device
        with torch.cuda.device(dev):
            if skip_on_nan:
                _lib.check(lib.bgk_grad_nan_flag(_lib.ptr(self.grad), self.grad.numel(), _lib.ptr(self._flag), _lib.stream_ptr(dev)),
                           "bgk_grad_nan_flag")
            _lib.check(lib.bgk_adam_step(_lib.ptr(self.flat), _lib.ptr(self.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                         self.flat.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                         float(g["weight_decay"]), self._step, _lib.ptr(self._flag) if skip_on_nan else None,
                                         _lib.ptr(self._skipped), _lib.stream_ptr(dev)), "bgk_adam_step")
        # the kernel wrote through the bucket, which torch's version counters do not see: bump them by hand -- caches keyed on
        # ``_version`` (UniformDistribution._const_host, user code) and autograd's saved-tensor check (a backward through a graph
        # retained across this step raises instead of silently using the new weights) then see the update like any in-place op --
        # and the generation counter the packed-operand caches of the fused kernels are keyed on as well (dense.param_state_key)
        self.generation += 1
        for p in self._params:
            p._bgk_generation = self.generation
        _bump_versions(self._params)
        # ... and re-pack those operands now, for all fused layers at once (three launches instead of five per layer at their next use)
        from . import dense
        dense.repack_training_plans(self._param_ids)
        dense.repack_affine_training_plans(self._param_ids)

    def skipped_steps(self):
        """number of optimizer steps skipped because a gradient was NaN (host sync)"""
        return int(self._skipped.item())

    def state_dict(self):
        """torch's layout (``state`` / ``param_groups``) plus the flat moments and the step counters (they live outside
        ``Optimizer.state``: one tensor per moment, not one per parameter)"""
        sd = super().state_dict()
        sd["flat_adam"] = dict(exp_avg=self.exp_avg.detach().clone(), exp_avg_sq=self.exp_avg_sq.detach().clone(),
                               step=int(self._step), skipped=int(self._skipped.item()))
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        extra = state_dict.pop("flat_adam", None)
        super().load_state_dict(state_dict)
        if extra is None:
            raise ValueError("FlatAdam.load_state_dict: the checkpoint carries no 'flat_adam' entry (moments / step count)")
        if extra["exp_avg"].numel() != self.exp_avg.numel():
            raise ValueError("FlatAdam.load_state_dict: bucket size mismatch")
        self.exp_avg.copy_(extra["exp_avg"].to(self.exp_avg.device))
        self.exp_avg_sq.copy_(extra["exp_avg_sq"].to(self.exp_avg_sq.device))
        self._step = int(extra["step"])
        self._skipped.fill_(int(extra["skipped"]))


class KLTrainer(object):
    """Same constructor and ``train`` signature as the reference (trainers.py:48-205)."""

    def __init__(self, bg, optim=None, train_likelihood=True, train_energy=True, custom_loss=None, test_likelihood=False):
        self.bg = bg
        if optim is None:
            ps = [p for p in bg.parameters() if p.requires_grad]
            optim = FlatAdam(ps, lr=5e-3) if ps and all(p.is_cuda for p in ps) else torch.optim.Adam(bg.parameters(), lr=5e-3)
        self.optim = optim
        loss_names = []
        self.train_likelihood = train_likelihood
        self.w_likelihood = 0.0
        self.train_energy = train_energy
        self.w_energy = 0.0
        self.test_likelihood = test_likelihood
        if train_energy:
            loss_names.append("KLL")
            self.w_energy = 1.0
        if train_likelihood:
            loss_names.append("NLL")
            self.w_likelihood = 1.0
        if test_likelihood:
            loss_names.append("NLL(Test)")
        if custom_loss is not None:
            # deviation: the reference reports the custom loss without registering a label for it, so its own reporter
            # assertion (trainers.py:24) fires as soon as w_custom is used
            loss_names.append("custom")
        self.reporter = LossReporter(*loss_names)
        self.custom_loss = custom_loss

    @staticmethod
    def _mean(per_sample):
        return dp.global_mean(per_sample) if dp.is_distributed() else per_sample.mean()

    # ---- one optimisation step = loss terms (each: evaluate -> report -> weighted backward into the flat gradient bucket) + one
    # parameter update.  The terms and the update are built once per train() call; the loop only runs them.
    def _kl_term(self, batchsize, temperature):
        """reverse-KL term (trainers.py:158-163).  The generator's own ``kldiv`` unless it is the package's implementation: then
        ``kldiv_mean`` forms the loss sums inside the target-energy kernel (same value, no per-sample tensor, one ready 2-vector for the
        data-parallel all-reduce).  A subclass that overrides ``kldiv`` (a regulariser, another target) is evaluated through its code."""
        from .bg import BoltzmannGenerator
        fused = isinstance(self.bg, BoltzmannGenerator) and type(self.bg).kldiv is BoltzmannGenerator.kldiv \
            and type(self.bg).kldiv_mean is BoltzmannGenerator.kldiv_mean
        if fused:
            return lambda: self.bg.kldiv_mean(batchsize, temperature=temperature)
        return lambda: self._mean(self.bg.kldiv(batchsize, temperature=temperature))

    def _nll_term(self, sampler, batchsize, temperature):
        def term():
            batch = sampler.sample(batchsize)
            batch = (batch,) if isinstance(batch, torch.Tensor) else batch
            return self._mean(self.bg.energy(*batch, temperature=temperature))
        return term

    def _update(self, params):
        """gradient exchange + parameter update: FlatAdam = one all-reduce of the bucket and one fused launch that skips itself on the
        device when a gradient is not finite (trainers.py:198-201 without the host sync); any other optimizer: the reference's check"""
        if isinstance(self.optim, FlatAdam):
            self.optim.allreduce_gradients()
            self.optim.step()
            return
        dp.allreduce_gradients_(params)
        if any(torch.any(torch.isnan(p.grad)) for p in params if p.grad is not None):
            print("found nan in grad; skipping optimization step")
        else:
            self.optim.step()

    def train(self, n_iter, data=None, testdata=None, batchsize=128, w_likelihood=None, w_energy=None, w_custom=None,
              custom_loss_kwargs={}, n_print=0, temperature=1.0, schedulers=(), clip_forces=None, progress_bar=lambda x: x):
        import contextlib
        from .dense import direct_grad_accumulation
        w_likelihood = self.w_likelihood if w_likelihood is None else w_likelihood
        w_energy = self.w_energy if w_energy is None else w_energy
        if clip_forces is not None:
            warnings.warn("clip_forces is deprecated and will be ignored. Use GradientClippedEnergy instances instead",
                          DeprecationWarning)
        data = DataSetSampler(data) if isinstance(data, torch.Tensor) else data
        testdata = DataSetSampler(testdata) if isinstance(testdata, torch.Tensor) else testdata
        params = list(self.bg.parameters())
        # weight gradients reduce straight into the optimizer's bucket when it is the flat one
        direct = direct_grad_accumulation if isinstance(self.optim, FlatAdam) else contextlib.nullcontext
        wsum = w_likelihood + w_energy
        terms = []                                     # (evaluate, weight in the total loss or None = report only)
        if self.train_energy:
            terms.append((self._kl_term(batchsize, temperature), w_energy / wsum if w_energy > 0 else None))
        if self.train_likelihood:
            terms.append((self._nll_term(data, batchsize, temperature), w_likelihood / wsum if w_likelihood > 0 else None))

        def backward(loss):
            with direct():
                loss.backward(retain_graph=True)

        for it in progress_bar(range(n_iter)):
            for interval, scheduler in schedulers:
                if it % interval == 0:
                    scheduler.step()
            self.optim.zero_grad()
            reports = []
            for evaluate, weight in terms:
                value = evaluate()
                reports.append(value)
                if weight is not None:
                    backward(weight * value)
            if self.test_likelihood:
                testnll = torch.zeros_like(reports[-1]) if reports else torch.zeros(())
                if testdata is not None:
                    with torch.no_grad():
                        testnll = self._nll_term(testdata, batchsize, temperature)()
                reports.append(testnll)
            if w_custom is not None:
                cl = self.custom_loss(**custom_loss_kwargs)
                (w_custom * cl).backward(retain_graph=True)
                reports.append(cl)
            elif self.custom_loss is not None:
                reports.append(float("nan"))
            self.reporter.report(*reports)
            if n_print > 0 and it % n_print == 0:
                self.reporter.print(*reports)
            self._update(params)

    def losses(self, n_smooth=1):
        return self.reporter.losses(n_smooth=n_smooth)
