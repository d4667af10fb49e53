"""Vectorised torch-CPU restatement of the reference's op chain for builder flows (spline / affine couplings + CDF domain maps;
the coordinate transform at the end runs the C oracle).  TEST / MEASUREMENT INFRASTRUCTURE ONLY: bench.py's cpu_baseline leg
times it ("the reference's stock-aten op chain on the host cores of the GPU box" -- the reference itself cannot travel), and
tests/test_oracle_golden.py pins it against the golden vectors.  Never imported by the product.

Each function cites the reference code whose operation chain it restates with stock torch ops (same ops, same order, so that
the time it takes is what the reference would spend here):
  rq_spline           nflows.transforms.splines.rational_quadratic_spline as called by nn/flow/transformer/spline.py:133-144,
                      164-175 (public nflows algorithm; in-tree corroboration nn/flow/spline.py:121-188)
  spline_transformer  ConditionalSplineTransformer._compute_params/_forward/_inverse (transformer/spline.py:87-188)
  affine_transformer  AffineTransformer (transformer/affine.py:35-70)
  dense / periodic    DenseNet.forward (nn/dense.py:47-48), WrapPeriodic.forward (nn/periodic.py:30-37)
  cdf_block           CDFTransform (nn/flow/cdf.py:28-46) over TruncatedNormalDistribution (distribution/normal.py:215-227),
                      torch Normal / uniform marginals (factory/icmarginals.py:41-77)
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import flow_oracle as fo


def _name(m):
    return type(m).__name__


def rq_spline(x, uw, uh, ud, inverse, left, right, bottom, top, min_w, min_h, min_d, identity_init):
    K = uw.shape[-1]
    w = F.softmax(uw, dim=-1)
    w = min_w + (1 - min_w * K) * w
    cw = torch.cumsum(w, dim=-1)
    cw = F.pad(cw, pad=(1, 0), mode="constant", value=0.0)
    cw = (right - left) * cw + left
    cw[..., 0] = left
    cw[..., -1] = right
    w = cw[..., 1:] - cw[..., :-1]
    beta = np.log(2) / (1 - min_d) if identity_init else 1.0
    dv = min_d + F.softplus(ud, beta=beta)
    h = F.softmax(uh, dim=-1)
    h = min_h + (1 - min_h * K) * h
    ch = torch.cumsum(h, dim=-1)
    ch = F.pad(ch, pad=(1, 0), mode="constant", value=0.0)
    ch = (top - bottom) * ch + bottom
    ch[..., 0] = bottom
    ch[..., -1] = top
    h = ch[..., 1:] - ch[..., :-1]
    knots = ch if inverse else cw
    knots = knots.clone()
    knots[..., -1] += 1e-6
    idx = (torch.sum(x[..., None] >= knots, dim=-1) - 1)[..., None]
    in_cw = cw.gather(-1, idx)[..., 0]
    in_w = w.gather(-1, idx)[..., 0]
    in_ch = ch.gather(-1, idx)[..., 0]
    delta = h / w
    in_delta = delta.gather(-1, idx)[..., 0]
    in_d = dv.gather(-1, idx)[..., 0]
    in_d1 = dv[..., 1:].gather(-1, idx)[..., 0]
    in_h = h.gather(-1, idx)[..., 0]
    if inverse:
        a = (x - in_ch) * (in_d + in_d1 - 2 * in_delta) + in_h * (in_delta - in_d)
        b = in_h * in_d - (x - in_ch) * (in_d + in_d1 - 2 * in_delta)
        c = -in_delta * (x - in_ch)
        disc = b.pow(2) - 4 * a * c
        root = (2 * c) / (-b - torch.sqrt(disc))
        out = root * in_w + in_cw
        t1mt = root * (1 - root)
        den = in_delta + (in_d + in_d1 - 2 * in_delta) * t1mt
        num = in_delta.pow(2) * (in_d1 * root.pow(2) + 2 * in_delta * t1mt + in_d * (1 - root).pow(2))
        return out, -(torch.log(num) - 2 * torch.log(den))
    theta = (x - in_cw) / in_w
    t1mt = theta * (1 - theta)
    numer = in_h * (in_delta * theta.pow(2) + in_d * t1mt)
    den = in_delta + (in_d + in_d1 - 2 * in_delta) * t1mt
    out = in_ch + numer / den
    num = in_delta.pow(2) * (in_d1 * theta.pow(2) + 2 * in_delta * t1mt + in_d * (1 - theta).pow(2))
    return out, torch.log(num) - 2 * torch.log(den)


# Rounding hook for the GEMM operands of the SPLINE conditioners (None: none).  The reduced-precision mode gemm_mode = "bf16" of the fused
# kernels feeds the matrix cores bf16-rounded weights and layer inputs and accumulates in f32; with ``SPLINE_GEMM_ROUNDING = bf16_round`` an
# f64 run of this restatement is the exact value of THAT arithmetic (up to the accumulation order), which bounds the kernel by an oracle
# instead of by its own f16 twin (tests/test_gpu_round6.py::test_bf16_leg_against_the_rounding_oracle).
SPLINE_GEMM_ROUNDING = None


def bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


def conditioner(net, x, rnd=None):
    n = _name(net)
    if n == "DenseNet":
        first = True
        for m in net._layers:
            if _name(m) == "Linear":
                if rnd is None:
                    x = F.linear(x, m.weight, m.bias)
                else:
                    # the kernels' operand forms: layer 0 carries its bias as the weight column of a constant-1 input feature (one rounded
                    # value); the later layers add it as a two-term sum hi + lo of rounded values (their "bias blocks")
                    b = m.bias
                    b = rnd(b) if first else rnd(b) + rnd(b - rnd(b))
                    x = F.linear(rnd(x), rnd(m.weight), b)
                first = False
            else:
                x = m(x)
        return x
    if n == "WrapPeriodic":
        y = x[..., net.indices]
        cs = torch.cat([torch.cos(2 * np.pi * y), torch.sin(2 * np.pi * y)], dim=-1)      # all inputs periodic on [0, 1]
        return conditioner(net.net, cs, rnd)
    return net(x)


def spline_transformer(tr, cond, y, inverse):
    p = conditioner(tr._params_net, cond, SPLINE_GEMM_ROUNDING)
    d = y.shape[-1]
    circ = torch.as_tensor(np.asarray(tr._is_circular, dtype=bool)) if np.ndim(tr._is_circular) else torch.full((d,), bool(tr._is_circular))
    n_nc = int((~circ).sum())
    K = (p.shape[-1] - n_nc) // (3 * d)
    w, h, s, s_nc = torch.split(p, [d * K, d * K, d * K, n_nc], dim=-1)
    w, h, s = (v.reshape(*v.shape[:-1], d, K) for v in (w, h, s))
    s = torch.cat([s, s[..., [0]]], dim=-1)
    if n_nc:
        s[..., ~circ, -1] = s_nc
    st = tr._default_settings
    # bgflow forward = nflows inverse=True (spline.py:133-144)
    z, ld = rq_spline(y.clamp(tr._left, tr._right), w, h, s, not inverse, tr._left, tr._right, tr._bottom, tr._top,
                      st["min_bin_width"], st["min_bin_height"], st["min_derivative"], st.get("enable_identity_init", False))
    return z, ld.sum(dim=-1, keepdim=True)


def affine_transformer(tr, cond, y, inverse):
    mu = conditioner(tr._shift_transformation, cond) if tr._shift_transformation is not None else torch.zeros_like(y)
    if tr._scale_transformation is not None:
        ls = torch.tanh(conditioner(tr._scale_transformation, cond)) * torch.exp(tr._log_alpha)
        if tr._preserve_volume:
            ls = ls - ls.mean(dim=-1, keepdim=True)
    else:
        ls = torch.zeros_like(y)
    if not inverse:
        out, dl = torch.exp(ls) * y + mu, ls.sum(dim=-1, keepdim=True)
    else:
        out, dl = torch.exp(-ls) * (y - mu), -ls.sum(dim=-1, keepdim=True)
    return (out % 1.0 if tr._is_circular else out), dl


def _marginal(dist):
    n = _name(dist)
    if n == "TruncatedNormalDistribution":
        mu, sig = dist._mu, torch.exp(dist._logsigma)
        lo, Z = dist._cdf_lower_bound, dist._cdf_upper_bound - dist._cdf_lower_bound
        std = torch.distributions.Normal(torch.zeros(()), torch.ones(()))
        return (lambda x: (std.cdf((x - mu) / sig) - lo) / Z, lambda u: std.icdf(Z * u + lo) * sig + mu,
                lambda x: -0.5 * ((x - mu) / sig) ** 2 - 0.5 * np.log(2 * np.pi) - torch.log(Z * sig))
    if n in ("SloppyUniform", "_SloppyUniform", "Uniform"):
        lo, hi = dist.low, dist.high
        return (lambda x: ((x - lo) / (hi - lo)).clamp(0, 1), lambda u: lo + u * (hi - lo),
                lambda x: (-torch.log(hi - lo)).expand_as(x))
    if n in ("Normal", "_NormalMarginal"):
        nd = torch.distributions.Normal(dist.loc, dist.scale)
        return nd.cdf, nd.icdf, nd.log_prob
    raise NotImplementedError(n)


def cdf_block(block, x, inverse):
    cdf, icdf, logp = _marginal(block.distribution)
    eps = block._eps
    if not inverse:
        y = cdf(x)
        if eps is not None:
            y = y.clamp(eps, 1 - eps)
        return y, logp(x).sum(dim=-1, keepdim=True)
    if eps is not None:
        x = x.clamp(eps, 1 - eps)
    y = icdf(x)
    return y, -logp(y).sum(dim=-1, keepdim=True)


def ic2xyz_torch(ic, bonds, angles, torsions, zfix):
    """IC -> xyz with torch ops, differentiable by autograd: the op chain of RelativeInternalCoordinateTransformation._inverse /
    MixedCoordinateTransformation._inverse (nn/flow/crd_transform/ic.py:435-513, 862-884; ic2xyz_deriv ic_helper.py:372-452;
    blackening pca.py:85-93), one vectorised step per dependency block of the Z-matrix like the reference.  Used by the CPU
    KL-step baseline (the C oracle has no autograd)."""
    rel = getattr(ic, "_rel_ic", ic)
    B, n = bonds.shape[0], bonds.shape[1]
    place = torch.as_tensor(np.asarray(rel._tables._host["place"]), dtype=torch.long)
    fixed = torch.as_tensor(np.asarray(rel._tables._host["fixed"]), dtype=torch.long)
    dl = torch.zeros(B, 1, dtype=bonds.dtype)
    if hasattr(ic, "_whiten"):
        w = ic._whiten
        xf = zfix @ w.Tblacken.to(zfix) + w.X0mean.to(zfix)
        dl = dl - w.jacobian_xz.to(zfix)
    else:
        xf = zfix
    eps = float(rel._eps)
    if rel._normalize_angles:
        angles = angles * np.pi
        torsions = torsions * (2 * np.pi) - np.pi
        dl = dl + n * (np.log(np.pi) + np.log(2 * np.pi))
    pos = torch.zeros(B, n + len(fixed), 3, dtype=bonds.dtype)
    pos[:, fixed] = xf.reshape(B, -1, 3)
    start = 0
    for blk in rel._z_blocks:
        rows = place[start:start + len(blk)]
        start += len(blk)
        at, i1, i2, i3, zr = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4]
        p1, p2, p3 = pos[:, i1], pos[:, i2], pos[:, i3]
        d, a, t = bonds[:, zr, None], angles[:, zr, None], torsions[:, zr, None]
        v1, v2 = p1 - p2, p1 - p3
        nv = torch.cross(v1, v2, dim=-1)
        nn = torch.cross(v1, nv, dim=-1)
        nh = nv / nv.norm(dim=-1, keepdim=True).clamp_min(eps)
        nnh = nn / nn.norm(dim=-1, keepdim=True).clamp_min(eps)
        v3 = -torch.sin(t) * nh + torch.cos(t) * nnh
        v3n = v3.norm(dim=-1, keepdim=True).clamp_min(eps)
        v3h = v3 / v3n
        v1h = v1 / v1.norm(dim=-1, keepdim=True).clamp_min(eps)
        new = p1 + v3h * d * torch.sin(a) - v1h * d * torch.cos(a)
        Jd = v3h * torch.sin(a) - v1h * torch.cos(a)
        Ja = v3h * d * torch.cos(a) + v1h * d * torch.sin(a)
        Jt3 = -torch.cos(t) * nh - torch.sin(t) * nnh
        Jt = (d * torch.sin(a) / v3n) * (Jt3 - v3h * (v3h * Jt3).sum(-1, keepdim=True))
        J = torch.stack([Jd, Ja, Jt], dim=-1)
        det = (torch.cross(J[..., 0, :], J[..., 1, :], dim=-1) * J[..., 2, :]).sum(-1)
        dl = dl + torch.log(det.abs()).sum(-1, keepdim=True)
        pos = pos.clone()
        pos[:, at] = new
    return pos.reshape(B, -1), dl


def run_block(block, xs, inverse, grad=False):
    """xs: list of torch CPU tensors -> (list, dlogp [B,1])"""
    if grad and not inverse and _name(block) in ("MixedCoordinateTransformation", "RelativeInternalCoordinateTransformation"):
        raise NotImplementedError("differentiable xyz -> IC is not restated (the KL step only needs IC -> xyz)")
    if grad and inverse and _name(block) in ("MixedCoordinateTransformation", "RelativeInternalCoordinateTransformation"):
        x, dl = ic2xyz_torch(block, *xs)
        return [x], dl
    n = _name(block)
    if n == "SequentialFlow":
        return run_flow(block, xs, inverse, grad=grad)
    if n == "InverseFlow":
        return run_block(block._delegate, xs, not inverse, grad=grad)
    if n == "CouplingFlow":
        cond = torch.cat([xs[i] for i in block.cond_indices], dim=-1)
        y = torch.cat([xs[i] for i in block.transformed_indices], dim=-1)
        tr = block.transformer
        fn = spline_transformer if _name(tr) == "ConditionalSplineTransformer" else affine_transformer
        out, dl = fn(tr, cond, y, inverse)
        outs = list(xs)
        sizes = [xs[i].shape[-1] for i in block.transformed_indices]
        for i, piece in zip(block.transformed_indices, torch.split(out, sizes, dim=-1)):
            outs[i] = piece
        return outs, dl
    if n == "WrapFlow":
        take = list(block._out_indices if inverse else block._indices)
        put = list(block._indices if inverse else block._out_indices)
        rest = [x for i, x in enumerate(xs) if i not in take]
        ys, dl = run_block(block._flow, [xs[i] for i in take], inverse, grad=grad)
        for k in np.argsort(put):
            rest.insert(put[k], ys[k])
        return rest, dl
    if n == "CDFTransform":
        y, dl = cdf_block(block, xs[0], inverse)
        return [y], dl
    # tuple plumbing of a coupling stack with stock torch ops (differentiable: the KL gradient of cfg 2 passes through them)
    if n == "MergeFlow":                                                    # InverseFlow(SplitFlow), nn/flow/coupling.py:107-110
        return run_block(block._delegate, xs, not inverse, grad=grad)
    if n == "SplitFlow" and block._indices is None:                        # coupling.py:46-57 (sizes; the last may be omitted)
        zeros = torch.zeros_like(xs[0].narrow(block._split_dim, 0, 1))
        if inverse:
            return [torch.cat(list(xs), dim=block._split_dim)], zeros
        rest = xs[0].shape[block._split_dim] - sum(block._sizes)
        return list(torch.split(xs[0], list(block._sizes) + ([rest] if rest > 0 else []), dim=block._split_dim)), zeros
    if n == "SwapFlow":                                                    # coupling.py:113-130
        return [xs[1], xs[0], *xs[2:]], torch.zeros_like(xs[0][..., :1])
    # everything else (coordinate transforms, split / merge / swap plumbing): the numpy / C oracle
    dt = np.float32 if xs[0].dtype == torch.float32 else np.float64
    ys, dl = fo.run_block(block, [x.numpy() for x in xs], inverse, dt)
    return [torch.as_tensor(np.asarray(y)) for y in ys], torch.as_tensor(np.asarray(dl))


def run_flow(flow, xs, inverse=False, grad=False):
    """``grad=True``: build the autograd graph (conditioner parameters that require gradients; sampling direction only)"""
    xs = list(xs)
    blocks = list(flow._blocks)[::-1] if inverse else list(flow._blocks)
    total = 0.0
    with torch.set_grad_enabled(bool(grad)):
        for block in blocks:
            xs, dl = run_block(block, xs, inverse, grad=grad)
            total = total + dl
    return xs, total
