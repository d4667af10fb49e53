"""Flow protocol and tuple-plumbing blocks, API-compatible with bgflow.nn.flow.

``Flow.forward(*xs, inverse=False, **kwargs) -> (*ys, dlogp)`` with ``dlogp`` of shape [batch, 1]
(reference: bgflow/nn/flow/base.py:17-33).  The blocks in this file move tensors around and never
touch the data with arithmetic; the arithmetic blocks (transformers, internal coordinates) live in
transformer.py / ic.py and run hand-written HIP kernels.

Reference files mirrored here (names, constructor signatures, error behaviour):
  bgflow/nn/flow/base.py, sequential.py, inverted.py, coupling.py.
"""
import os
import warnings
from collections.abc import Sequence

import numpy as np
import torch

__all__ = [
    "Flow", "SequentialFlow", "InverseFlow", "SplitFlow", "MergeFlow", "SwapFlow", "CouplingFlow",
    "WrapFlow", "SetConstantFlow", "StochasticAugmentation",
]


class Flow(torch.nn.Module):
    """Base class: subclasses implement ``_forward`` / ``_inverse`` (base.py:7-33)."""

    def _forward(self, *xs, **kwargs):
        raise NotImplementedError()

    def _inverse(self, *xs, **kwargs):
        raise NotImplementedError()

    def forward(self, *xs, inverse=False, **kwargs):
        return self._inverse(*xs, **kwargs) if inverse else self._forward(*xs, **kwargs)


def _zero_dlogp(x):
    return torch.zeros(*x.shape[:-1], 1, dtype=x.dtype, device=x.device)


ACC_KW = "_bgk_logdet"      # private kwarg: the running log|det J| buffer of the enclosing SequentialFlow pass


class _LogDetAcc:
    """The running log|det J| of one SequentialFlow pass: ONE [B] f32 buffer that every kernel-backed block adds its log-det to
    inside its own kernel (``accumulate`` of the C ABI) -- the reference's ``dlogp += ddlogp`` (sequential.py:58) without a
    per-block [B, 1] tensor, without the elementwise add launch per block and without the zero tensors of the plumbing blocks.
    A block that received the accumulator through the ``_bgk_logdet`` kwarg and wrote into it returns the accumulator ITSELF in
    place of its dlogp tensor; any other return value is a normal [B, 1] tensor and is added by ``add``.  Only used when no
    gradient is needed (the autograd Functions of the training path return fresh tensors)."""
    __slots__ = ("buf", "started")

    def __init__(self, batch, device):
        self.buf = torch.empty(batch, dtype=torch.float32, device=device)
        self.started = False

    def target(self):
        """(buffer, accumulate flag) for a kernel launch: the first writer overwrites, the others add"""
        acc = self.started
        self.started = True
        return self.buf, acc

    def peek(self):
        """like ``target`` without marking the buffer written; ``commit()`` after the launch succeeded"""
        return self.buf, self.started

    def commit(self):
        self.started = True

    def add(self, t):
        if t is self or t is None:
            return
        t = t.reshape(-1) if torch.is_tensor(t) else t
        if self.started:
            self.buf.add_(t)
        elif torch.is_tensor(t):
            self.buf.copy_(t)
            self.started = True
        else:
            self.buf.fill_(float(t))
            self.started = True

    def result(self):
        if not self.started:
            self.buf.zero_()
            self.started = True
        return self.buf[:, None]


class CatView:
    """Several [B, w_i] f32 HIP tensors that stand for their concatenation along the last axis (``torch.cat(..., -1)`` of
    coupling.py:162-165) without being copied: the fused coupling kernels stage up to three conditioning tensors straight from
    their own rows.  ``cat()`` materialises the concatenation for code that needs one tensor."""
    __slots__ = ("parts", "_cat")

    def __init__(self, parts):
        self.parts = tuple(parts)
        self._cat = None

    @property
    def shape(self):
        return torch.Size([self.parts[0].shape[0], sum(t.shape[1] for t in self.parts)])

    device = property(lambda self: self.parts[0].device)
    dtype = property(lambda self: self.parts[0].dtype)
    is_cuda = property(lambda self: self.parts[0].is_cuda)
    requires_grad = False

    def dim(self):
        return 2

    def cat(self):
        if self._cat is None:
            self._cat = torch.cat(self.parts, dim=-1)
        return self._cat


def as_tensor(x):
    """a CatView's concatenation, any tensor unchanged"""
    return x.cat() if isinstance(x, CatView) else x


def _acc_kwargs(flow, kwargs):
    """kwargs for ``flow``: the accumulator travels only into blocks that declare ``_bgk_acc`` (the classes of this package); a user
    block with a strict signature never sees the private kwarg"""
    if ACC_KW in kwargs and not getattr(flow, "_bgk_acc", False):
        kwargs = {k: v for k, v in kwargs.items() if k != ACC_KW}
    return kwargs


def _acc_eligible(xs):
    """a pass can run on one accumulator when its first input is a 2-d f32 HIP tensor and nothing needs a gradient"""
    x = xs[0] if xs else None
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in xs):
        return False
    return True


class SequentialFlow(Flow):
    """Chain of blocks; log-dets are summed, blocks run reversed for ``inverse=True``
    (sequential.py:26-59)."""

    def __init__(self, blocks):
        super().__init__()
        self._blocks = torch.nn.ModuleList(blocks)

    FUSE_GENERATION_TAIL = True   # icdf domain maps + IC -> xyz as one kernel in the sampling direction (bgk_icdf_ic2xyz)
    FUSE_TRAINING_TAIL = True     # ... also when the inputs need gradients: one-launch forward that keeps the mapped fields for the backward
    FUSE_COUPLING_STACKS = True   # Split -> (affine Coupling | Swap)* -> Merge on ONE [B, D] buffer: no cat / per-layer outputs
    FUSE_TRAINING_CHAINS = os.environ.get("BGK_TRAIN_CHAIN", "1") != "0"   # training: runs of fused spline couplings as ONE autograd node

    _bgk_acc = True
    ACCUMULATE_IN_KERNELS = True   # one running log-det buffer, written by the kernels themselves, when no gradient is needed

    def forward(self, *xs, inverse=False, **kwargs):
        return self.run(xs, inverse=inverse, kwargs=kwargs)

    FUSE_KL_EPILOGUE = os.environ.get("BGK_KL_EPILOGUE", "1") != "0"   # kl_sums: the target energy inside the generation tail's launch

    def kl_sums(self, xs, target, temperature=1.0, drop_nonfinite=False):
        """f64 [sum_b (u_target(x_b) - dlogp_b), samples kept] of ``x, dlogp = self(*xs)`` (BoltzmannGenerator.kldiv, bg.py:140-147, summed)
        in a pass that builds an autograd graph, with the target energy formed INSIDE the launch of the generation tail when the flow
        ends with the builder's [icdf maps, IC -> xyz] and the target is a normal distribution over the Cartesian output; None
        otherwise (the caller evaluates the flow and the energy one after the other)."""
        from .distributions import _kernel_plan
        if not (self.FUSE_KL_EPILOGUE and torch.is_grad_enabled() and isinstance(temperature, (int, float)) and temperature > 0):
            return None
        segs = self.segments(inverse=False, train=True)
        tail = segs[-1][1] if segs else None
        if not isinstance(tail, _FusedGenerationTail):
            return None
        plan = _kernel_plan(target, temperature)
        if plan is None:
            return None
        specs, dims, c_in, c_out, t_eff = plan
        ic = tail._ic
        n_cart = 3 * (getattr(ic, "_rel_ic", ic)._n + getattr(ic, "_rel_ic", ic)._n_fixed)
        if not (len(specs) == 1 and specs[0][0] == 0 and dims == [n_cart]):
            return None
        total = None
        for _label, seg in segs[:-1]:
            *xs, dd = seg(*xs, inverse=False, temperature=temperature)
            total = dd if total is None else total + dd
        if len(xs) != 4:
            res = None
        else:
            res = tail.kl_sums(tuple(xs), total, (specs, t_eff, c_in, c_out), drop_nonfinite)
        if res is not None:
            return res
        # outside the tail's envelope: the tail, then the energy kernel's loss form on its output
        from .distributions import kl_loss_sums
        *x, dd = tail(*xs, inverse=False, temperature=temperature)
        total = dd if total is None else total + dd
        out = kl_loss_sums(target, tuple(x), total, temperature=temperature, drop_nonfinite=drop_nonfinite)
        if out is not None:
            return out[0]
        # the flow HAS run (its segments above built their autograd graph, bumped their out-of-domain counters): finish here with the
        # per-sample form instead of handing None back -- the caller would evaluate the whole flow a second time
        per = target.energy(*x, temperature=temperature) - total
        if drop_nonfinite:
            ok = torch.isfinite(per)
            return torch.stack([torch.where(ok, per, torch.zeros_like(per)).sum().to(torch.float64), ok.sum().to(torch.float64)])
        return torch.stack([per.sum().to(torch.float64), torch.tensor(float(per.numel()), dtype=torch.float64, device=per.device)])

    def run(self, xs, inverse=False, kwargs=None, around=None):
        """The pass itself.  ``around(i, label)``, if given, returns a context manager entered around segment i (bench.py times the
        segments with HIP events that way, on the same code path as ``forward``)."""
        kwargs = dict(kwargs or {})
        outer = kwargs.pop(ACC_KW, None)
        acc = outer
        if acc is None and self.ACCUMULATE_IN_KERNELS and _acc_eligible(xs) and not (
                torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            acc = _LogDetAcc(xs[0].shape[0], xs[0].device)
        if acc is None:
            # the accumulation of the reference (sequential.py:49,58: `dlogp = 0.0; dlogp += ddlogp`) without its first launch: the
            # first segment's log-det IS the running sum (0.0 + t is t), later ones are added out of place
            total = None
            for i, (label, seg) in enumerate(self.segments(inverse=inverse, train=True)):
                if around is None:
                    *xs, ddlogp = seg(*xs, inverse=inverse, **kwargs)
                else:
                    with around(i, label):
                        *xs, ddlogp = seg(*xs, inverse=inverse, **kwargs)
                total = ddlogp if total is None else total + ddlogp
            return (*xs, 0.0 if total is None else total)
        kwargs[ACC_KW] = acc
        for i, (label, seg) in enumerate(self.segments(inverse=inverse)):
            kw = _acc_kwargs(seg, kwargs)
            if around is None:
                *xs, ddlogp = seg(*xs, inverse=inverse, **kw)
                acc.add(ddlogp)
            else:
                with around(i, label):
                    *xs, ddlogp = seg(*xs, inverse=inverse, **kw)
                    acc.add(ddlogp)
        return (*xs, acc if outer is not None else acc.result())

    def segments(self, inverse=False, train=False):
        """[(label, callable)] in execution order: the blocks themselves, except that in the sampling direction a tail of
        builder domain maps ``WrapFlow(InverseFlow(CDFTransform))`` followed by ``WrapFlow(InverseFlow(<IC transform>))``
        (generator_builder.py:443-459 + add_map_to_cartesian) runs as ONE fused kernel when the inputs need no gradients.
        ``train``: the list of a pass that builds an autograd graph -- runs of spline coupling blocks are one segment there
        (_SplineTrainChain: one autograd node, field gradients accumulated inside the backward kernels)."""
        blocks = list(self._blocks)
        # the segment lists (and what the fused segments cache: descriptor tables) are rebuilt only when the blocks or the switches change
        train = bool(train) and self.FUSE_TRAINING_CHAINS
        key = (self.FUSE_GENERATION_TAIL, self.FUSE_COUPLING_STACKS, tuple(id(b) for b in blocks))
        cache = self.__dict__.setdefault("_segment_cache", {})
        if cache.get("key") != key:
            cache.clear()
            cache["key"] = key
        which = (bool(inverse), train)
        if which not in cache:
            segs = self._build_segments(blocks, inverse)
            cache[which] = _with_train_chains(segs) if train else segs
        return cache[which]

    def _build_segments(self, blocks, inverse):
        tail = self._generation_tail() if self.FUSE_GENERATION_TAIL else None
        if inverse:
            if tail is None:
                return self._with_coupling_stacks(list(reversed(blocks)), True)
            # the NLL direction enters through the tail: xyz -> IC + the cdf maps as one fused segment, then the rest reversed
            return [("xyz2ic+cdf", _FusedInferenceHead(self, tail))] + self._with_coupling_stacks(list(reversed(blocks[:tail[0]])), True)
        if tail is None:
            return self._with_coupling_stacks(blocks, False)
        start = tail[0]
        return self._with_coupling_stacks(blocks[:start], False) + [("icdf+ic2xyz", _FusedGenerationTail(self, tail))]

    def _with_coupling_stacks(self, blocks, inverse):
        """[(label, callable)] for ``blocks`` (already in execution order) with every run
        ``split -> (CouplingFlow(AffineTransformer) | SwapFlow)* -> merge`` replaced by one _FusedCouplingStack"""
        out, i = [], 0
        while i < len(blocks):
            j = _coupling_stack_end(blocks, i, inverse) if self.FUSE_COUPLING_STACKS else None
            if j is None:
                out.append((type(blocks[i]).__name__, blocks[i]))
                i += 1
            else:
                out.append(("coupling stack", _FusedCouplingStack(blocks[i:j + 1])))
                i = j + 1
        return out

    def _generation_tail(self):
        """(first block index, {slot: CDFTransform}, ic) if the flow ends with [domain maps..., IC -> xyz], else None"""
        from .cdf import CDFTransform
        blocks = list(self._blocks)
        if len(blocks) < 2:
            return None
        last = blocks[-1]
        ic = getattr(getattr(last, "_flow", None), "_delegate", None)
        if not (type(last) is WrapFlow and type(last._flow) is InverseFlow and hasattr(ic, "_generate_fused")
                and list(last._indices) == [0, 1, 2, 3] and list(last._out_indices) == [0]):
            return None
        maps, others = {}, []
        i = len(blocks) - 1
        while i - 1 >= 0:
            b = blocks[i - 1]
            cdf = getattr(getattr(b, "_flow", None), "_delegate", None)
            if not (type(b) is WrapFlow and type(b._flow) is InverseFlow and type(cdf) is CDFTransform and len(b._indices) == 1
                    and list(b._out_indices) == list(b._indices) and b._indices[0] not in maps):
                break
            if b._indices[0] in (0, 1, 2, 3):
                maps[b._indices[0]] = cdf
            else:
                others.insert(0, b)       # a map on another slot (e.g. auxiliary variables): commutes with the tail, runs as a block
            i -= 1
        if not maps:
            return None
        eps = {c._eps for c in maps.values()}
        if len(eps) != 1:
            return None
        return i, maps, ic, eps.pop(), others

    def _forward(self, *args, **kwargs):
        return self.forward(*args, **kwargs, inverse=False)

    def _inverse(self, *args, **kwargs):
        return self.forward(*args, **kwargs, inverse=True)

    def trigger(self, function_name):
        """Call ``function_name()`` on every block that has it; stack the results (sequential.py:67-80)."""
        results = [getattr(b, function_name)() for b in self._blocks
                   if callable(getattr(b, function_name, None))]
        if results and all(r is not None for r in results):
            return torch.stack(results)
        return torch.zeros(0)

    def __iter__(self):
        return iter(self._blocks)

    def __len__(self):
        return len(self._blocks)

    def __getitem__(self, index):
        if isinstance(index, int):
            return self._blocks[index]
        picked = np.arange(len(self))[index]
        return SequentialFlow([self._blocks[i] for i in picked])


def _with_train_chains(segs):
    """``segs`` with every run of >= 2 consecutive spline coupling blocks replaced by one _SplineTrainChain segment"""
    from .transformer import ConditionalSplineTransformer
    out, run = [], []

    def flush():
        if len(run) >= 2:
            out.append(("spline chain", _SplineTrainChain([b for _, b in run])))
        else:
            out.extend(run)
        run.clear()
    for label, seg in segs:
        if type(seg) is CouplingFlow and type(seg.transformer) is ConditionalSplineTransformer and seg.cat_dim == -1 \
                and len(seg.transformed_indices) == 1 and len(seg.cond_indices) == 1:
            run.append((label, seg))
        else:
            flush()
            out.append((label, seg))
    flush()
    return out


class _SplineTrainChain:
    """callable standing in for a run of spline CouplingFlow blocks in a pass that builds an autograd graph: the whole run is one
    autograd node (dense._SplineChainTrainFn) -- one training-forward launch per layer adding into one log-det buffer; in the
    backward the conditioner-input gradients are accumulated per field inside bgk_dense_backward_dx.  Falls back to the blocks
    themselves when nothing needs a gradient, a block carries extra keyword arguments, or a layer is outside the fused training
    envelope."""

    _bgk_acc = True

    def __init__(self, blocks):
        self._blocks = blocks

    def _blocks_path(self, *xs, inverse=False, **kwargs):
        acc = kwargs.get(ACC_KW)
        total = None
        for block in self._blocks:
            *xs, dd = block(*xs, inverse=inverse, **_acc_kwargs(block, kwargs))
            if acc is not None:
                acc.add(dd)
            else:
                total = dd if total is None else total + dd
        return (*xs, acc if acc is not None else total)

    def __call__(self, *xs, inverse=False, **kwargs):
        from .dense import spline_chain_train
        if kwargs.get(ACC_KW) is None and not set(kwargs) - {"temperature"} and torch.is_grad_enabled() and (
                any(torch.is_tensor(x) and x.requires_grad for x in xs)
                or any(p.requires_grad for b in self._blocks for p in b.parameters())):
            res = spline_chain_train(self._blocks, xs, inverse)
            if res is not None:
                return (*res[0], res[1])
        return self._blocks_path(*xs, inverse=inverse, **kwargs)


def _split_sizes(block, inverse, merging):
    """sizes of a SplitFlow(s0[, s1]) / MergeFlow(s0[, s1]) along the last axis that acts as a split (merging=False) or a
    merge (True) in the given direction, else None"""
    split = block if type(block) is SplitFlow else (block._delegate if type(block) is MergeFlow else None)
    if split is None or split._indices is not None or split._split_dim != -1 or len(split._sizes) not in (1, 2):
        return None
    acts_as_merge = (type(block) is MergeFlow) != bool(inverse)
    return tuple(split._sizes) if acts_as_merge == merging else None


def _coupling_stack_end(blocks, i, inverse):
    """index of the merge closing a run split -> (affine coupling | swap)* -> merge that starts at blocks[i], else None"""
    from .transformer import AffineTransformer
    sizes = _split_sizes(blocks[i], inverse, merging=False)
    if sizes is None:
        return None
    n_couplings, swapped = 0, False
    for j in range(i + 1, len(blocks)):
        b = blocks[j]
        if type(b) is SwapFlow:
            swapped = not swapped
            continue
        if type(b) is CouplingFlow and type(b.transformer) is AffineTransformer and tuple(b.transformed_indices) == (1,) \
                and tuple(b.cond_indices) == (0,) and b.cat_dim == -1:
            n_couplings += 1
            continue
        closing = _split_sizes(b, inverse, merging=True)
        if closing is None or n_couplings < 2:
            return None
        if swapped:                  # the merge sees (part 1, part 0): its first size is the width of part 1 (checked again at run time)
            return j if (len(sizes) < 2 or closing[0] == sizes[1]) else None
        return j if closing[0] == sizes[0] else None
    return None


class _FusedCouplingStack:
    """callable standing in for ``split -> (CouplingFlow(AffineTransformer) | SwapFlow)* -> merge``: the two parts live as column
    ranges of ONE [B, D] workspace, every coupling layer (bgk_coupling_affine_dense_h2) writes its half there -- in place once the
    half has been produced -- and adds its log-det to one accumulator, so that neither the per-layer output tensors, nor the
    dlogp additions, nor the closing concatenation exist.  Falls back to the blocks themselves when gradients are needed, the input
    is not a contiguous f32 HIP matrix, or a layer's conditioners are outside the fused envelope."""

    _bgk_acc = True
    FUSE_TRAINING_STACK = os.environ.get("BGK_TRAIN_STACK", "1") != "0"    # under autograd: the stack as one node (dense._AffineStackTrainFn)

    def __init__(self, blocks):
        self._blocks = blocks

    def _blocks_path(self, *xs, inverse=False, **kwargs):
        acc = kwargs.get(ACC_KW)
        total = 0.0
        for block in self._blocks:
            *xs, dd = block(*xs, inverse=inverse, **_acc_kwargs(block, kwargs))
            if acc is not None:
                acc.add(dd)
            else:
                total = total + dd
        return (*xs, acc if acc is not None else total)

    def __call__(self, *xs, inverse=False, **kwargs):
        from .dense import fused_affine_coupling, _affine_plan, _gemm_mode
        acc = kwargs.get(ACC_KW)
        kwargs_rest = {k: v for k, v in kwargs.items() if k != ACC_KW}
        x = xs[0] if len(xs) == 1 else None
        # a pure `temperature` kwarg does not reach the affine transformers' arithmetic (bg.py:16,21 inject it into every flow call)
        ok = (x is not None and not set(kwargs_rest) - {"temperature"} and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32
              and x.dim() == 2 and x.is_contiguous() and x.shape[0] > 0)
        couplings = [b for b in self._blocks if type(b) is CouplingFlow]
        if ok and torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for b in couplings for p in b.parameters())):
            # a pass that builds an autograd graph: the whole stack as ONE node on the affine training kernels (round 6), else the blocks
            ok = False
            if acc is None and self.FUSE_TRAINING_STACK:
                from .dense import affine_stack_train
                res = affine_stack_train(self._blocks, x, inverse)
                if res is not None:
                    return res
        if ok:
            split = self._blocks[0] if type(self._blocks[0]) is SplitFlow else self._blocks[0]._delegate
            s0, D = split._sizes[0], x.shape[1]
            ok = 0 < s0 < D and (len(split._sizes) == 1 or split._sizes[1] == D - s0)
        if ok:
            widths = (s0, D - s0)
            part = [0, 1]                                   # tuple slot -> column range
            for b in self._blocks[1:-1]:                    # dry run: every layer must be inside the fused envelope
                if type(b) is SwapFlow:
                    part.reverse()
                    continue
                tr = b.transformer
                if not getattr(tr, "allow_fused", False) or _gemm_mode(tr) == "f32" or _affine_plan(tr, widths[part[1]]) is None \
                        or tr._fused_cache.get("d_c") != widths[part[0]]:
                    ok = False
                    break
            if ok:                                          # the closing merge must describe the parts as they arrive
                closing = self._blocks[-1] if type(self._blocks[-1]) is SplitFlow else self._blocks[-1]._delegate
                ok = closing._sizes[0] == widths[part[0]] and (len(closing._sizes) == 1 or closing._sizes[1] == widths[part[1]])
        if not ok:
            return self._blocks_path(*xs, inverse=inverse, **kwargs)
        B = x.shape[0]
        work = torch.empty_like(x)
        if acc is not None:
            dlogp, started = acc.target()
        else:
            dlogp, started = torch.empty(B, dtype=torch.float32, device=x.device), False
        cols = (slice(0, s0), slice(s0, D))
        where = [x, x]                                      # buffer currently holding each column range
        part, first = [0, 1], not started
        for b in self._blocks[1:-1]:
            if type(b) is SwapFlow:
                part.reverse()
                continue
            pc, py = part
            res = fused_affine_coupling(b.transformer, where[pc][:, cols[pc]], where[py][:, cols[py]], inverse,
                                        out=work[:, cols[py]], dlogp=dlogp, accumulate=not first)
            if res is None:                                 # cannot happen after the dry run; keep the semantics anyway
                if acc is not None:
                    raise RuntimeError("fused coupling stack: a layer left the fused envelope after the dry run")
                return self._blocks_path(*xs, inverse=inverse, **kwargs)
            where[py], first = work, False
        for p in (0, 1):
            if where[p] is x:                               # a half no layer transformed
                work[:, cols[p]].copy_(x[:, cols[p]])
        if part != [0, 1]:                                  # odd number of swaps: the merge concatenates (part 1, part 0)
            work = torch.cat([work[:, cols[1]], work[:, cols[0]]], dim=-1)
        return work, (acc if acc is not None else dlogp[:, None])


class _FusedTailTrainFn(torch.autograd.Function):
    """the generation tail [icdf maps..., IC -> xyz] as ONE launch in a training forward (bgk_icdf_ic2xyz_uni_train: it also writes
    the mapped fields); backward on the kernels the block path uses -- bgk_ic_ic2xyz_backward, then bgk_cdf_backward per mapped field"""

    @staticmethod
    def forward(ctx, zb, za, zt, zf, tail, descs, desc20):
        res = tail._ic._generate_fused_train(zb, za, zt, zf, tail._eps, desc20)
        if res is None:
            raise _TailOutsideEnvelope()
        x, dlogp, ys, rel, blacken = res
        ctx.rel, ctx.blacken, ctx.descs, ctx.eps = rel, blacken, descs, tail._eps
        ctx.save_for_backward(zb, za, zt, zf, *ys, x)
        return x, dlogp[:, None]

    @staticmethod
    def backward(ctx, g_x, g_dlogp):
        from .cdf import cdf_backward
        zb, za, zt, zf, yb, ya, yt, yf, x = ctx.saved_tensors
        g_ys = ctx.rel._ic2xyz_backward(yb, ya, yt, x, ctx.blacken, g_x, g_dlogp)
        outs = []
        for z, y, g_y, desc in zip((zb, za, zt, zf), (yb, ya, yt, yf), g_ys, ctx.descs):
            # a field without a domain map passes through (its y is a copy of z, its log-det contribution is 0)
            outs.append(g_y if desc is None else cdf_backward(z.flatten(1), y, desc, True, ctx.eps, g_y, g_dlogp).view_as(z))
        return (*outs, None, None, None)


class _FusedTailKLFn(torch.autograd.Function):
    """the generation tail AND the KL integrand of a normal target as one launch (bgk_icdf_ic2xyz_uni_train_kl): the lanes hold their
    samples' coordinates in registers when the placements are done, so u_target(x) and the tile's share of [sum (u - dlogp), samples
    kept] cost no second pass over x (round 5: what SURVEY f-3 asks of `kldiv`).  Returns the f64 pair; backward = the target-energy
    backward kernel on the saved x (gradient of u and of the mask of kept samples), then the tail's backward kernels."""

    @staticmethod
    def forward(ctx, zb, za, zt, zf, dl_in, tail, descs, desc20, plan, drop):
        specs, t_eff, c_in, c_out = plan
        mean = specs[0][1]
        t_mean = None if mean is None else mean.detach().to(device=zb.device, dtype=torch.float32).contiguous()
        dl = None if dl_in is None else dl_in.detach().reshape(-1).to(torch.float32).contiguous()
        res = tail._ic._generate_fused_train(zb, za, zt, zf, tail._eps, desc20, kl=(t_mean, t_eff, c_in, c_out, drop, dl))
        if res is None:
            raise _TailOutsideEnvelope()
        x, dl_tot, ys, u, sums, rel, blacken = res
        ctx.rel, ctx.blacken, ctx.descs, ctx.eps = rel, blacken, descs, tail._eps
        ctx.cfg = (specs, t_eff, bool(drop), None if dl_in is None else dl_in.shape)
        ctx.save_for_backward(zb, za, zt, zf, *ys, x, u, dl_tot)
        return sums

    @staticmethod
    def backward(ctx, g_sums):
        import ctypes
        from . import _lib
        from .cdf import cdf_backward
        from .distributions import _fields_args
        zb, za, zt, zf, yb, ya, yt, yf, x, u, dl_tot = ctx.saved_tensors
        specs, t_eff, drop, dl_shape = ctx.cfg
        args, _keep = _fields_args(specs, (x,))
        B, dev = x.shape[0], x.device
        gs = g_sums[0:1].to(torch.float32).contiguous()
        g_x = torch.empty_like(x)
        g_dl = torch.empty(B, dtype=torch.float32, device=dev)
        G = (ctypes.c_void_p * 1)(g_x.data_ptr())
        LG = (ctypes.c_int64 * 1)(g_x.shape[1])
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_energy_fields_backward(*args, B, t_eff, None, _lib.ptr(gs), _lib.ptr(u), _lib.ptr(dl_tot), int(drop),
                                                       _lib.ptr(g_dl), G, LG, _lib.stream_ptr(dev))
        _lib.check(st, "bgk_energy_fields_backward")
        g_ys = ctx.rel._ic2xyz_backward(yb, ya, yt, x, ctx.blacken, g_x, g_dl)
        outs = []
        for z, y, g_y, desc in zip((zb, za, zt, zf), (yb, ya, yt, yf), g_ys, ctx.descs):
            outs.append(g_y if desc is None else cdf_backward(z.flatten(1), y, desc, True, ctx.eps, g_y, g_dl).view_as(z))
        return (*outs, None if dl_shape is None else g_dl.reshape(dl_shape), None, None, None, None, None)


class _TailOutsideEnvelope(Exception):
    pass


class _FusedGenerationTail:
    """callable standing in for the tail blocks [icdf maps..., IC -> xyz] of a SequentialFlow (sampling direction).  Falls back
    to the blocks themselves when an input needs gradients, is not an f32 HIP tensor, or a marginal has no kernel descriptor."""

    _bgk_acc = True

    def __init__(self, flow, tail):
        self._flow, (self._start, self._maps, self._ic, self._eps, self._others) = flow, tail

    def _desc20(self, xs):
        """[3 n + keep, 20] descriptor table of the register-resident tail kernel: bonds | angles | torsions rows IN PLACEMENT ORDER
        (row f n + i = the channel of field f that placement i consumes), then the fixed rows; a field without a map gets kind -1
        rows.  Cached on the per-map descriptors' identity; None if a marginal has no descriptor or the transform no placement
        table."""
        from .cdf import TAIL_DESC
        rel = getattr(self._ic, "_rel_ic", self._ic)
        order = getattr(rel, "_placement_zrows", None)
        if order is None:
            return None
        parts, key = [], []
        for slot in range(4):
            d = xs[slot].shape[-1]
            cdf = self._maps.get(slot)
            if cdf is None:
                t = torch.zeros(d, TAIL_DESC, dtype=torch.float32, device=xs[slot].device)
                t[:, 0] = torch.tensor(-1, dtype=torch.int32).view(torch.float32)
            else:
                t = cdf.tail_descriptor(d, xs[slot].device)
                if t is None:
                    return None
            parts.append(t)
            key.append((id(t), d))
        key = tuple(key)
        if getattr(self, "_desc20_key", None) != key:
            idx = torch.as_tensor(order, device=parts[0].device)
            if any(p.shape[0] != len(order) for p in parts[:3]):
                return None
            self._desc20_key = key
            self._desc20_tab = torch.cat([parts[0][idx], parts[1][idx], parts[2][idx], parts[3]], dim=0).contiguous()
            # field-uniform marginals (one distribution per field, the builder's case): a [4, 20] table selects the elementwise kernel
            uniform = all(bool((p == p[:1]).all()) for p in parts)
            self._desc20_tab.uniform4 = torch.cat([p[:1] for p in parts], dim=0).contiguous() if uniform else None
            self._desc20_parts = parts            # keep the per-map tensors alive: their ids are the cache key
        return self._desc20_tab

    def kl_sums(self, xs, dlogp, plan, drop):
        """f64 [sum (u - dlogp), kept] of a normal target with the energy formed inside the tail's training launch, or None when the
        launch is outside its envelope (the caller then runs the tail and the energy kernel one after the other)"""
        ok = (len(xs) == 4 and all(torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 for x in xs)
              and self._flow.FUSE_TRAINING_TAIL and hasattr(self._ic, "_generate_fused_train") and not self._others and len(self._maps) == 4)
        if not ok:
            return None
        descs = [None] * 4
        for slot, cdf in self._maps.items():
            descs[slot] = cdf.kernel_descriptor(xs[slot].shape[-1], xs[slot].device)
            if descs[slot] is None:
                return None
        desc20 = self._desc20(xs)
        if desc20 is None or getattr(desc20, "uniform4", None) is None:
            return None
        try:
            return _FusedTailKLFn.apply(xs[0], xs[1], xs[2], xs[3], dlogp, self, descs, desc20, plan, bool(drop))
        except _TailOutsideEnvelope:
            return None

    def _blocks_path(self, *xs, **kwargs):
        acc = kwargs.get(ACC_KW)
        total = 0.0
        for block in list(self._flow._blocks)[self._start:]:
            *xs, dd = block(*xs, **_acc_kwargs(block, kwargs))
            if acc is not None:
                acc.add(dd)
            else:
                total = total + dd
        return (*xs, acc if acc is not None else total)

    def __call__(self, *xs, inverse=False, **kwargs):
        assert not inverse
        ok = len(xs) >= 4 and all(torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 for x in xs[:4])
        train = ok and torch.is_grad_enabled() and any(x.requires_grad for x in xs[:4])
        if ok and torch.is_grad_enabled() and any(torch.is_tensor(x) and x.requires_grad for x in xs[4:]):
            ok = False
        if train and not (self._flow.FUSE_TRAINING_TAIL and hasattr(self._ic, "_generate_fused_train") and not self._others
                          and len(self._maps) == 4 and kwargs.get(ACC_KW) is None):
            ok = False
        descs = [None] * 4
        if ok:
            for slot, cdf in self._maps.items():
                descs[slot] = cdf.kernel_descriptor(xs[slot].shape[-1], xs[slot].device)
                if descs[slot] is None:
                    ok = False
        if ok and train:
            # training: one launch forward (it also writes the mapped fields), backward on the block path's kernels
            desc20 = self._desc20(xs)
            if desc20 is not None and getattr(desc20, "uniform4", None) is not None:
                try:
                    x, dlogp = _FusedTailTrainFn.apply(xs[0], xs[1], xs[2], xs[3], self, descs, desc20)
                    return (x, *xs[4:], dlogp)
                except _TailOutsideEnvelope:
                    pass
            ok = False
        if not ok:
            return self._blocks_path(*xs, **kwargs)
        acc = kwargs.get(ACC_KW)
        total = 0.0
        for block in self._others:                      # maps on slots the coordinate transform does not touch
            *xs, dd = block(*xs, **_acc_kwargs(block, kwargs))
            if acc is not None:
                acc.add(dd)
            else:
                total = total + dd
        x, dlogp = self._ic._generate_fused(xs[0], xs[1], xs[2], xs[3], descs, self._eps, acc=acc, desc20=self._desc20(xs))
        if acc is not None:
            return (x, *xs[4:], acc)
        return (x, *xs[4:], dlogp + total if self._others else dlogp)


class _FusedInferenceHead:
    """callable standing in for the tail blocks [icdf maps..., IC -> xyz] of a SequentialFlow run in the INVERSE (NLL) direction:
    x -> xyz -> IC + whitening + the four cdf maps in one launch (bgk_xyz2ic_cdf_uni).  Falls back to the blocks themselves (reversed)
    when the input needs gradients, is not a contiguous f32 HIP matrix, or the marginals are not field-uniform."""
    _bgk_acc = True

    def __init__(self, flow, tail):
        self._flow, (self._start, self._maps, self._ic, self._eps, self._others) = flow, tail
        self._tail_fwd = _FusedGenerationTail(flow, tail)       # shares the descriptor cache logic

    def _blocks_path(self, *xs, **kwargs):
        acc = kwargs.get(ACC_KW)
        total = 0.0
        for block in reversed(list(self._flow._blocks)[self._start:]):
            *xs, dd = block(*xs, inverse=True, **_acc_kwargs(block, kwargs))
            if acc is not None:
                acc.add(dd)
            else:
                total = total + dd
        return (*xs, acc if acc is not None else total)

    def _desc4(self, x):
        rel = getattr(self._ic, "_rel_ic", self._ic)
        n, keep = rel._n, self._ic.dim_fixed
        dummy = [torch.empty(0, n, device=x.device)] * 3 + [torch.empty(0, keep, device=x.device)]
        tab = self._tail_fwd._desc20(dummy)
        return None if tab is None else getattr(tab, "uniform4", None)

    def __call__(self, *xs, inverse=True, **kwargs):
        assert inverse
        x = xs[0] if xs else None
        ok = (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
              and not (torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in xs)) and hasattr(self._ic, "_infer_fused"))
        desc4 = self._desc4(x) if ok else None
        if desc4 is None:
            return self._blocks_path(*xs, **kwargs)
        acc = kwargs.get(ACC_KW)
        res = self._ic._infer_fused(x, desc4, self._eps, acc=acc)
        if res is None:
            return self._blocks_path(*xs, **kwargs)
        b, a, t, zf, dlogp = res
        out = (b, a, t, zf, *xs[1:])
        total = 0.0
        for block in reversed(self._others):                # maps on slots the coordinate transform does not touch
            *out, dd = block(*out, inverse=True, **_acc_kwargs(block, kwargs))
            if acc is not None:
                acc.add(dd)
            else:
                total = total + dd
        if acc is not None:
            return (*out, acc)
        return (*out, dlogp + total if self._others else dlogp)


class InverseFlow(Flow):
    """Swap the two directions of ``delegate`` (inverted.py:7-23)."""

    def __init__(self, delegate):
        super().__init__()
        self._delegate = delegate

    _bgk_acc = True

    def _forward(self, *xs, **kwargs):
        return self._delegate._inverse(*xs, **_acc_kwargs(self._delegate, kwargs))

    def _inverse(self, *xs, **kwargs):
        return self._delegate._forward(*xs, **_acc_kwargs(self._delegate, kwargs))


class SplitFlow(Flow):
    """Split one tensor into several along ``dim`` by sizes (views) or by index lists (gathers);
    the inverse concatenates / scatters (coupling.py:13-104).  The size of the last chunk may be
    omitted.  Raises ValueError on a too-short tensor, overlapping or missing indices."""

    def __init__(self, *sizes_or_indices, dim=-1):
        super().__init__()
        first = sizes_or_indices[0]
        by_index = isinstance(first, (Sequence, np.ndarray))
        self._sizes = None if by_index else sizes_or_indices
        self._indices = sizes_or_indices if by_index else None
        self._split_dim = dim

    _bgk_acc = True

    def _forward(self, x, **kwargs):
        parts = self._split_with_sizes(x) if self._indices is None else self._split_with_indices(x)
        acc = kwargs.get(ACC_KW)
        return (*parts, acc if acc is not None else self._dlogp(x))

    def _inverse(self, *xs, **kwargs):
        if self._indices is None:
            y = torch.cat(xs, dim=self._split_dim)
        else:
            y = self._cat_with_indices(*xs)
        acc = kwargs.get(ACC_KW)
        return y, (acc if acc is not None else self._dlogp(xs[0]))

    def _dlogp(self, x):
        return torch.zeros_like(x.narrow(self._split_dim, 0, 1))

    def _split_with_sizes(self, x):
        rest = x.shape[self._split_dim] - sum(self._sizes)
        if rest < 0:
            raise ValueError(f"can't split x [{x.shape}] into sizes {self._sizes} along {self._split_dim}")
        sizes = list(self._sizes) + ([rest] if rest > 0 else [])
        return torch.split(x, sizes, dim=self._split_dim)

    def _check_cover(self, length, verb):
        seen = np.zeros(length, dtype=bool)
        for idx in self._indices:
            idx = np.asarray(idx, dtype=np.int64)
            if seen[idx].any():
                raise ValueError(f"Cannot {verb} tensor. Indices are overlapping.")
            seen[idx] = True
        if not seen.all():
            word = "Split" if verb == "split" else "Merge"
            raise ValueError(f"{word} with indices missed indices {np.arange(length)[~seen]}")

    def _take(self, x, idx):
        index = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=x.device)
        return x.index_select(self._split_dim, index)

    def _split_with_indices(self, x):
        self._check_cover(x.shape[self._split_dim], "split")
        return [self._take(x, idx) for idx in self._indices]

    def _cat_with_indices(self, *xs):
        length = sum(len(idx) for idx in self._indices)
        self._check_cover(length, "merge")
        shape = list(xs[0].shape)
        shape[self._split_dim] = length
        y = torch.empty(*shape, device=xs[0].device, dtype=xs[0].dtype)
        for x, idx in zip(xs, self._indices):
            index = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=x.device)
            y.index_copy_(self._split_dim if self._split_dim >= 0 else y.dim() + self._split_dim, index, x)
        return y


class MergeFlow(InverseFlow):
    """``InverseFlow(SplitFlow(*sizes))`` (coupling.py:107-110)."""

    def __init__(self, *sizes, dim=-1):
        super().__init__(SplitFlow(*sizes, dim=dim))


class SwapFlow(Flow):
    """Exchange the first two tensors (coupling.py:113-130)."""

    _bgk_acc = True

    def _swap(self, *xs, acc=None):
        if len(xs) == 1:
            warnings.warn("applying swapping on a single tensor has no effect")
        return (xs[1], xs[0], *xs[2:], acc if acc is not None else _zero_dlogp(xs[0]))

    def _forward(self, *xs, **kwargs):
        return self._swap(*xs, acc=kwargs.get(ACC_KW))

    def _inverse(self, *xs, **kwargs):
        return self._swap(*xs, acc=kwargs.get(ACC_KW))


class CouplingFlow(Flow):
    """Coupling layer: tensors ``transformed_indices`` are transformed conditioned on tensors
    ``cond_indices`` (coupling.py:133-182).  ValueError if the two index sets intersect."""

    def __init__(self, transformer, transformed_indices=(1,), cond_indices=(0,), cat_dim=-1):
        super().__init__()
        self.transformer = transformer
        self.transformed_indices = transformed_indices
        self.cond_indices = cond_indices
        clash = np.intersect1d(self.transformed_indices, self.cond_indices)
        if len(clash) > 0:
            raise ValueError(f"Indices {clash} cannot be both transformed and conditioned on.")
        self.cat_dim = cat_dim

    _bgk_acc = True
    MULTI_COND_IN_KERNEL = True    # several conditioning tensors go to the fused kernels as they are (False: torch.cat first)

    def _gather(self, x, indices, lazy=False):
        # a single tensor needs no concatenation copy (the kernels take strided rows)
        if len(indices) == 1:
            return x[indices[0]]
        parts = [x[i] for i in indices]
        if lazy and self.MULTI_COND_IN_KERNEL and self.cat_dim in (-1, parts[0].dim() - 1) and all(
                torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 for t in parts) and not (
                torch.is_grad_enabled() and any(t.requires_grad for t in parts)):
            return CatView(parts)         # the fused kernels read up to 3 conditioning tensors in place (no torch.cat launch)
        return torch.cat(parts, dim=self.cat_dim)

    def _couple(self, x, inverse, kwargs):
        lengths = [x[i].shape[self.cat_dim] for i in self.transformed_indices]
        inputs = self._gather(x, self.transformed_indices)
        cond = self._gather(x, self.cond_indices, lazy=getattr(self.transformer, "_bgk_multi_cond", False))
        x = list(x)
        kwargs = _acc_kwargs(self.transformer, kwargs)
        if inverse:
            y, dlogp = self.transformer.forward(cond, inputs, **kwargs, inverse=True)
        else:
            y, dlogp = self.transformer.forward(cond, inputs, **kwargs)
        parts = (y,) if len(lengths) == 1 else torch.split(y, lengths, self.cat_dim)
        for i, yi in zip(self.transformed_indices, parts):
            x[i] = yi
        return (*x, dlogp)

    def _forward(self, *x, **kwargs):
        return self._couple(x, False, kwargs)

    def _inverse(self, *x, **kwargs):
        return self._couple(x, True, kwargs)


class WrapFlow(Flow):
    """Apply ``flow`` to the tensors at ``indices``; its outputs are inserted at ``out_indices``
    (default: the same positions) among the untouched tensors (coupling.py:185-222)."""

    def __init__(self, flow, indices, out_indices=None):
        super().__init__()
        self._flow = flow
        self._indices = indices
        self._argsort_indices = np.argsort(indices)
        self._out_indices = indices if out_indices is None else out_indices
        self._argsort_out_indices = np.argsort(self._out_indices)

    _bgk_acc = True

    @staticmethod
    def _route(flow, xs, take, put, put_order, kwargs):
        rest = [x for i, x in enumerate(xs) if i not in take]
        *ys, dlogp = flow(*(xs[i] for i in take), **_acc_kwargs(flow, kwargs))
        for k in put_order:
            rest.insert(put[k], ys[k])
        return (*rest, dlogp)

    def _forward(self, *xs, **kwargs):
        return self._route(self._flow, xs, self._indices, self._out_indices, self._argsort_out_indices, kwargs)

    def _inverse(self, *xs, **kwargs):
        return self._route(self._flow, xs, self._out_indices, self._indices, self._argsort_indices,
                           dict(kwargs, inverse=True))


class SetConstantFlow(Flow):
    """Forward inserts constant tensors (repeated over the batch) at ``indices``; inverse drops
    them (coupling.py:227-272)."""

    _bgk_acc = True

    def __init__(self, indices, values, n_event_dims0=1):
        super().__init__()
        order = np.argsort(indices)
        self.indices = [indices[i] for i in order]
        for k, i in enumerate(order):
            self.register_buffer(f"_values_{k}", values[i])
        self.n_event_dims0 = n_event_dims0

    @property
    def values(self):
        out, k = [], 0
        while hasattr(self, f"_values_{k}"):
            out.append(getattr(self, f"_values_{k}"))
            k += 1
        return out

    def _forward(self, *xs, **kwargs):
        batch = list(xs[0].shape[:self.n_event_dims0])
        ys = list(xs)
        for i, v in zip(self.indices, self.values):
            ys.insert(i, v.repeat([*batch, *([1] * v.dim())]))
        acc = kwargs.get(ACC_KW)
        dlogp = acc if acc is not None else torch.zeros(batch + [1], device=xs[0].device, dtype=xs[0].dtype)
        return (*ys, dlogp)

    def _inverse(self, *xs, **kwargs):
        ys = tuple(x for i, x in enumerate(xs) if i not in self.indices)
        acc = kwargs.get(ACC_KW)
        if acc is not None:
            return (*ys, acc)
        batch = list(ys[0].shape[:self.n_event_dims0])
        dlogp = torch.zeros(batch + [1], device=ys[0].device, dtype=ys[0].dtype)
        return (*ys, dlogp)


class StochasticAugmentation(Flow):
    """Append auxiliary coordinates sampled from ``distribution`` (forward) / strip them (inverse);
    their energy enters dlogp (bgflow/nn/flow/stochastic/augment.py:27-55).  Pure tuple plumbing +
    the distribution's own sample/energy (stock torch ops)."""

    def __init__(self, distribution):
        super().__init__()
        self.distribution = distribution
        self._cached_momenta_forward = None
        self._cached_momenta_backward = None

    def _forward(self, q, **kwargs):
        temperature = kwargs.get("temperature", 1.0)
        p = kwargs.get("momenta", None)
        if p is None:
            p = self.distribution.sample(q.shape[0], temperature=temperature)
            dlogp = self.distribution.energy(p, temperature=temperature)
        else:
            dlogp = torch.zeros(p.shape[0], 1).to(p)
        if kwargs.get("cache_momenta", False):
            self._cached_momenta_forward = p
        return torch.cat([q, p], dim=1), dlogp

    def _inverse(self, x, **kwargs):
        dim = self.distribution.dim
        p = x[:, dim:]
        if kwargs.get("cache_momenta", False):
            self._cached_momenta_backward = p
        if kwargs.get("return_momenta", False):
            return x, torch.zeros(p.shape[0], 1).to(p)
        return x[:, :dim], -self.distribution.energy(p, temperature=kwargs.get("temperature", 1.0))
