"""Whole-flow CPU oracle: walks a SequentialFlow (bgflow_amd's classes or the reference's -- the
walk is duck-typed on class and attribute names) and evaluates every block with the C restatement
in oracle/bgo_oracle.c (numpy in / numpy out).  TEST INFRASTRUCTURE ONLY: used by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.

Block semantics follow the reference:
  SequentialFlow  nn/flow/sequential.py:49-59      CouplingFlow   nn/flow/coupling.py:162-182
  SplitFlow       nn/flow/coupling.py:46-104       SwapFlow       nn/flow/coupling.py:118-130
  WrapFlow        nn/flow/coupling.py:206-222      InverseFlow    nn/flow/inverted.py:19-23
  CDFTransform    nn/flow/cdf.py:28-46 (+ distribution/normal.py:215-227, torch Normal / Uniform)
  DenseNet        nn/dense.py:47-48                WrapPeriodic   nn/periodic.py:30-37
  transformers / internal coordinates: see oracle/bgo_impl.h
"""
import numpy as np
import scipy.special as sps

from . import oracle as orc


# accumulation order of hidden DenseNet layers: False = k ascending (textbook), True = the order of the
# fused HIP kernel (oracle.mfma_k_order).  Both are valid f32 evaluations of the same sums; the switch
# only matters for BIT-exact comparisons with bgk_coupling_rqs_dense.
MFMA_ORDER = False


def _np(t, dtype):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=dtype)


def _name(m):
    return type(m).__name__


# ---- conditioners -------------------------------------------------------------------------------
def dense_spec(net, dtype):
    """(weights, biases, acts) of a DenseNet (duck-typed: ``_layers`` Sequential of Linear / act)."""
    Ws, bs, acts = [], [], []
    for m in net._layers:
        n = _name(m)
        if n == "Linear":
            Ws.append(_np(m.weight, dtype)); bs.append(_np(m.bias, dtype)); acts.append(None)
        else:
            acts[-1] = {"SiLU": "silu", "ReLU": "relu", "Tanh": "tanh"}[n]
    return Ws, bs, acts


def conditioner(net, x, dtype):
    n = _name(net)
    if n == "DenseNet":
        return orc.dense_net(x, *dense_spec(net, dtype), dtype=dtype, mfma_order=MFMA_ORDER)
    if n == "WrapPeriodic":
        idx = np.arange(x.shape[-1])[net.indices]
        assert len(idx) == x.shape[-1] and net.left == 0.0 and net.right == 1.0, "oracle: all-periodic [0,1] only"
        return conditioner(net.net, orc.wrap_periodic(x, dtype=dtype), dtype)
    if n == "FixedNet":
        return _np(net.out, dtype)
    # arbitrary torch module: evaluate it with torch on the CPU in the flow's dtype
    import torch
    with torch.no_grad():
        return _np(net(torch.as_tensor(x)), dtype)


# ---- transformers -------------------------------------------------------------------------------
def transformer(tr, cond, y, inverse, dtype, details=None):
    n = _name(tr)
    if n == "ConditionalSplineTransformer":
        params = conditioner(tr._params_net, cond, dtype)
        circ = _np(tr._is_circular, bool)
        s = tr._default_settings
        out = orc.rqs(y, params, is_circular=circ, inverse=inverse, left=tr._left, right=tr._right,
                      bottom=tr._bottom, top=tr._top, min_bin_width=s["min_bin_width"],
                      min_bin_height=s["min_bin_height"], min_derivative=s["min_derivative"],
                      identity_init=s.get("enable_identity_init", False), dtype=dtype,
                      want_details=details is not None)
        if details is not None:
            details.append(out[2])
        return out[0], out[1]
    if n == "AffineTransformer":
        mu = conditioner(tr._shift_transformation, cond, dtype) if tr._shift_transformation is not None else None
        s_raw = conditioner(tr._scale_transformation, cond, dtype) if tr._scale_transformation is not None else None
        la = float(_np(tr._log_alpha, np.float64).reshape(-1)[0])
        return orc.affine(y, mu, s_raw, log_alpha=la, preserve_volume=tr._preserve_volume,
                          is_circular=tr._is_circular, inverse=inverse, dtype=dtype)
    raise NotImplementedError(f"oracle: transformer {n}")


# ---- marginal distributions of the CDF layers -------------------------------------------------------
def _dist_fns(dist):
    """(cdf, icdf, log_prob) in float64 numpy for the distributions the builder installs."""
    n = _name(dist)
    if n == "TruncatedNormalDistribution":
        mu = _np(dist._mu, np.float64); sig = np.exp(_np(dist._logsigma, np.float64))
        lo = _np(dist._cdf_lower_bound, np.float64); Z = _np(dist._cdf_upper_bound, np.float64) - lo
        return (lambda x: (sps.ndtr((x - mu) / sig) - lo) / Z,
                lambda u: sps.ndtri(Z * u + lo) * sig + mu,
                lambda x: -0.5 * ((x - mu) / sig) ** 2 - 0.5 * np.log(2 * np.pi) - np.log(Z * sig))
    if n in ("SloppyUniform", "_SloppyUniform", "Uniform"):
        lo = _np(dist.low, np.float64); hi = _np(dist.high, np.float64)
        return (lambda x: np.clip((x - lo) / (hi - lo), 0, 1), lambda u: lo + u * (hi - lo),
                lambda x: np.broadcast_to(-np.log(hi - lo), x.shape))
    if n in ("Normal", "_NormalMarginal"):
        loc = _np(dist.loc, np.float64); sc = _np(dist.scale, np.float64)
        return (lambda x: sps.ndtr((x - loc) / sc), lambda u: loc + sc * sps.ndtri(u),
                lambda x: -0.5 * ((x - loc) / sc) ** 2 - np.log(sc) - 0.5 * np.log(2 * np.pi))
    raise NotImplementedError(f"oracle: distribution {n}")


def cdf_transform(block, x, inverse, dtype):
    cdf, icdf, logp = _dist_fns(block.distribution)
    eps = block._eps
    x64 = np.asarray(x, np.float64)
    if not inverse:
        y = cdf(x64)
        if eps is not None:
            y = np.clip(y, dtype(eps), dtype(1.0) - dtype(eps))
        ld = logp(x64)
    else:
        if eps is not None:
            x64 = np.clip(x64, dtype(eps), dtype(1.0) - dtype(eps)).astype(np.float64)
        y = icdf(x64)
        ld = -logp(y)
    if eps is not None:
        ld = np.maximum(ld, -1.0 / eps)
    return y.astype(dtype), ld.sum(-1, keepdims=True).astype(dtype)


# ---- internal coordinates -------------------------------------------------------------------------
def _ic_tables(ic):
    z = np.asarray(ic.z_matrix if not hasattr(ic.z_matrix, "cpu") else ic.z_matrix.cpu().numpy())
    f = np.asarray(ic.fixed_atoms if not hasattr(ic.fixed_atoms, "cpu") else ic.fixed_atoms.cpu().numpy())
    return z, f


def ic_block(ic, xs, inverse, dtype):
    n = _name(ic)
    if n == "MixedCoordinateTransformation":
        rel, wh = ic._rel_ic, ic._whiten
        jac = float(-np.log(_np(wh.std, np.float64)).sum())
        z, f = _ic_tables(rel)
        kw = dict(normalize_angles=rel._normalize_angles, eps=rel._eps, enforce_boundaries=rel._enforce_boundaries, dtype=dtype)
        if not inverse:
            return orc.ic_xyz2ic(xs[0], z, f, whiten=(_np(wh.X0mean, dtype), _np(wh.Twhiten, dtype), jac), **kw)
        x, dl = orc.ic_ic2xyz(*xs, z, f, blacken=(_np(wh.X0mean, dtype), _np(wh.Tblacken, dtype), jac), **kw)
        return x, dl
    if n == "RelativeInternalCoordinateTransformation":
        z, f = _ic_tables(ic)
        kw = dict(normalize_angles=ic._normalize_angles, eps=ic._eps, enforce_boundaries=ic._enforce_boundaries, dtype=dtype)
        if not inverse:
            return orc.ic_xyz2ic(xs[0], z, f, **kw)
        return orc.ic_ic2xyz(*xs, z, f, **kw)
    raise NotImplementedError(f"oracle: coordinate transform {n}")


# ---- blocks ---------------------------------------------------------------------------------------
def run_block(block, xs, inverse, dtype, trace=None):
    """-> (tuple of arrays, dlogp [B,1])"""
    n = _name(block)
    xs = tuple(xs)
    B = xs[0].shape[0]
    zero = np.zeros((B, 1), dtype)
    if n == "SequentialFlow":
        return run_flow(block, xs, inverse=inverse, dtype=dtype, trace=trace)
    if n in ("InverseFlow", "MergeFlow"):
        return run_block(block._delegate, xs, not inverse, dtype, trace)
    if n == "SplitFlow":
        assert block._indices is None, "oracle: index splits not needed by the configs"
        if not inverse:
            x = xs[0]
            sizes = list(block._sizes)
            rest = x.shape[-1] - sum(sizes)
            if rest > 0:
                sizes.append(rest)
            cuts = np.cumsum(sizes)[:-1]
            return tuple(np.ascontiguousarray(p) for p in np.split(x, cuts, axis=-1)), zero
        return (np.concatenate(xs, axis=-1),), zero
    if n == "SwapFlow":
        return (xs[1], xs[0], *xs[2:]), zero
    if n == "CouplingFlow":
        ti, ci = list(block.transformed_indices), list(block.cond_indices)
        y = np.concatenate([xs[i] for i in ti], axis=-1)
        c = np.concatenate([xs[i] for i in ci], axis=-1)
        det = [] if trace is not None else None
        out, dl = transformer(block.transformer, c, y, inverse, dtype, details=det)
        if trace is not None and det:
            trace.append(det[0])
        outs = list(xs)
        cuts = np.cumsum([xs[i].shape[-1] for i in ti])[:-1]
        for i, part in zip(ti, np.split(out, cuts, axis=-1)):
            outs[i] = np.ascontiguousarray(part)
        return tuple(outs), dl
    if n == "WrapFlow":
        take = list(block._out_indices if inverse else block._indices)
        put = list(block._indices if inverse else block._out_indices)
        rest = [x for i, x in enumerate(xs) if i not in take]
        ys, dl = run_block(block._flow, [xs[i] for i in take], inverse, dtype, trace)
        for k in np.argsort(put):
            rest.insert(put[k], ys[k])
        return tuple(rest), dl
    if n == "CDFTransform":
        y, dl = cdf_transform(block, xs[0], inverse, dtype)
        return (y,), dl
    if n in ("MixedCoordinateTransformation", "RelativeInternalCoordinateTransformation"):
        res = ic_block(block, xs, inverse, dtype)
        return tuple(res[:-1]), res[-1]
    raise NotImplementedError(f"oracle: block {n}")


def run_flow(flow, xs, inverse=False, dtype=np.float32, trace=None, per_block=None):
    """SequentialFlow.forward: returns (tuple of output arrays, dlogp [B,1])."""
    xs = tuple(np.ascontiguousarray(np.asarray(x, dtype=dtype)) for x in xs)
    blocks = list(flow._blocks)
    if inverse:
        blocks = blocks[::-1]
    total = np.zeros((xs[0].shape[0], 1), dtype)
    for block in blocks:
        xs, dl = run_block(block, xs, inverse, dtype, trace)
        total = total + dl
        if per_block is not None:
            per_block.append((tuple(xs), dl))
    return xs, total
