"""Conditioner networks: DenseNet and the WrapPeriodic featuriser (bgflow/nn/dense.py,
bgflow/nn/periodic.py) plus the dispatchers of the one-launch coupling kernels.

As stand-alone torch modules the Linear layers of a DenseNet on HIP tensors run on ``bgk_dense_layer``
(one launch per ``Linear (+ SiLU / ReLU / Tanh)``; ``LAYER_KERNEL = False``: ``torch.nn.Linear`` -> hipBLASLt), which is also
what the layer-by-layer path of a coupling outside the fused envelope uses.  When a transformer's conditioner is a ``DenseNet``
(optionally inside a ``WrapPeriodic`` whose inputs are all periodic on [0, 1]) the whole coupling layer is ONE launch -- MLP on the
matrix cores + transformer epilogue:
  spline  (``fused_spline_coupling``): two hidden layers of <= 128 units (``bgk_coupling_rqs_dense_h2``, also the differentiable
          training forward) or <= 256 units (same entry, H0 = H1 = 256, inference); 1, 3, 4 .. 8 hidden layers of <= 128 units
          (``bgk_coupling_rqs_dense_deep``, inference); exact-f32 GEMMs for two hidden layers of 128 (``bgk_coupling_rqs_dense``);
  affine  (``fused_affine_coupling``): two / three hidden layers of <= 128 units (``bgk_coupling_affine_dense_h2`` / ``_h3``), any other
          depth from 1 to 8 (``bgk_coupling_affine_dense_deep``).
state_dict keys match the reference (``_layers.{i}.weight/bias``, ``net._layers...``).
"""
import ctypes
import os
import weakref

import numpy as np
import torch

from . import _lib
from .utils import is_list_or_tuple, param_pitch, param_state_key, row_pitch

__all__ = ["DenseNet", "MeanFreeDenseNet", "WrapPeriodic", "WrapDistances"]


def _matmul_nn(g, w):
    """g [B, n] @ w [n, k] for tall g: hipBLASLt's NN heuristics pick a 32x32x256 tile here (55 TFLOP/s); the same product
    through the TN entry (``linear`` with the small operand transposed once) runs the kernels the forward pass uses."""
    return torch.nn.functional.linear(g, w.t().contiguous())


def _gram_tn(g, h, splits=64):
    """g^T h for tall g [B, n], h [B, k] (weight gradient of a Linear layer): one GEMM with K = B leaves most of the chip
    idle (n k / tile^2 workgroups, hipBLASLt picks no split-K here: 0.5 ms at B = 2^18); as a batched GEMM over row slabs
    + a small sum it runs ~5x faster."""
    B = g.shape[0]
    if B % splits or B // splits < 256:
        return g.t() @ h
    return torch.bmm(g.view(splits, B // splits, -1).transpose(1, 2), h.view(splits, B // splits, -1)).sum(0)


def column_sum(g, nblk=1024):
    """out[c] = sum_r g[r, c] on the HIP kernel bgk_column_sum (f32, 2-D, HIP device)"""
    _lib.require_hip(g)
    g2, ldg = _lib.rowmajor(g)
    B, P = g2.shape
    nblk = int(max(1, min(nblk, (B + 63) // 64)))
    partial = torch.empty((nblk, P), dtype=torch.float32, device=g.device)
    out = torch.empty((P,), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        st = _lib.lib().bgk_column_sum(_lib.ptr(g2), ldg, B, P, _lib.ptr(partial), nblk, _lib.ptr(out), _lib.stream_ptr(g.device))
    _lib.check(st, "bgk_column_sum")
    return out


LAYER_BACKWARD = os.environ.get("BGK_LAYER_BACKWARD", "1") != "0"    # backward of a Linear on the hand-written kernels (0: library GEMMs, the A/B leg)


def linear_weight_grad(g, h, gW=None, gb=None, want_bias=True):
    """(g^T h [n, k], sum_rows g [n]) of 2-d f32 HIP tensors on bgk_linear_weight_grad (any widths); ``gW`` / ``gb``: destinations the
    result is ADDED to (the flat gradient bucket of training.FlatAdam) instead of fresh tensors"""
    _lib.require_hip(g, h)
    g2, ldg = _lib.rowmajor(g)
    h2, ldh = _lib.rowmajor(h)
    B, n = g2.shape
    k = h2.shape[1]
    dev = g.device
    accumulate = int(gW is not None)
    if gW is None:
        gW = torch.empty((n, k), dtype=torch.float32, device=dev)
        gb = torch.empty((n,), dtype=torch.float32, device=dev) if want_bias else None
    lib = _lib.lib()
    ws = torch.empty(int(lib.bgk_linear_weight_grad_workspace(B, n, k)), dtype=torch.float32, device=dev)
    am = absmax_of(g2)
    with torch.cuda.device(dev):
        st = lib.bgk_linear_weight_grad(_lib.ptr(g2), ldg, n, _lib.ptr(h2), ldh, k, B, _lib.ptr(ws), ws.numel(), _lib.ptr(gW), _lib.ptr(gb),
                                        accumulate, _lib.ptr(am), _lib.stream_ptr(dev))
    _lib.check(st, "bgk_linear_weight_grad")
    return gW, gb


class _LinearFn(torch.autograd.Function):
    """A Linear layer under autograd.  On HIP tensors (``lin``: the module): forward on bgk_dense_layer; backward (round 6) on the
    hand-written kernels too -- dX = g W as one more bgk_dense_layer call on the operands of W^T (bgk_refresh_linear_layer,
    transposed), dW = g^T x and db = sum g on bgk_linear_weight_grad, added straight into the FlatAdam bucket inside
    direct_grad_accumulation().  Rounds 1 - 5 ran F.linear + torch.bmm (hipBLASLt) + bgk_column_sum here; that form stays as the
    A/B leg (LAYER_BACKWARD = False) and for tensors the layer kernel does not take (``lin`` None)."""

    @staticmethod
    def forward(ctx, x, weight, bias, lin=None):
        ctx.save_for_backward(x, weight)
        ctx.lin = lin
        ctx.wb = (weight, bias)
        if lin is not None:                 # the module whose parameters these are: forward on bgk_dense_layer
            return dense_layer(x, lin, 0)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        need = ctx.needs_input_grad
        if ctx.lin is not None and LAYER_BACKWARD and g.is_cuda and g.dtype == torch.float32:
            gx = dense_layer(g, ctx.lin, 0, transposed=True) if need[0] else None
            gw = gb = None
            if need[1] or need[2]:
                W, b = ctx.wb
                direct = _DIRECT_GRADS[0] and need[1] and need[2] and all(
                    getattr(p, "_bgk_grad_dst", None) is not None and p.grad is not None and p.grad.data_ptr() == p._bgk_grad_dst.data_ptr()
                    for p in (W, b))
                if direct:
                    linear_weight_grad(g, x.detach(), W._bgk_grad_dst, b._bgk_grad_dst)
                else:
                    gw, gb = linear_weight_grad(g, x.detach(), want_bias=bool(need[2]))
                    gw = gw if need[1] else None
            return gx, gw, gb, None
        gx = _matmul_nn(g, weight) if need[0] else None
        gw = _gram_tn(g, x.contiguous()) if need[1] else None
        gb = column_sum(g) if need[2] else None
        return gx, gw, gb, None


def activation(z, act, g=None):
    """act(z) (``g`` None) or the VJP g * act'(z) on the elementwise kernels bgk_activation / bgk_activation_backward (act: 1 SiLU, 2 ReLU,
    3 Tanh; f32 HIP tensors of any shape, last axis = features)"""
    _lib.require_hip(z, g)
    z2, ldz = _lib.rowmajor(z.reshape(-1, z.shape[-1]))
    B, n = z2.shape
    out = torch.empty((B, n), dtype=torch.float32, device=z.device)
    if B and n:
        with torch.cuda.device(z.device):
            if g is None:
                st = _lib.lib().bgk_activation(_lib.ptr(z2), ldz, B, n, act, _lib.ptr(out), n, _lib.stream_ptr(z.device))
            else:
                g2, ldg = _lib.rowmajor(g.reshape(-1, n))
                st = _lib.lib().bgk_activation_backward(_lib.ptr(z2), ldz, _lib.ptr(g2), ldg, B, n, act, _lib.ptr(out), n, _lib.stream_ptr(z.device))
        _lib.check(st, "bgk_activation")
    return out.reshape(z.shape)


class _ActFn(torch.autograd.Function):
    """hidden activation of a DenseNet under autograd on the elementwise kernels (forward keeps z; backward g * act'(z))"""

    @staticmethod
    def forward(ctx, z, act):
        ctx.save_for_backward(z)
        ctx.act = act
        return activation(z, act)

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return activation(z, ctx.act, g.contiguous()), None


LAYER_KERNEL = True     # Linear layers of a DenseNet on HIP tensors run on bgk_dense_layer; False: torch.nn.Linear (hipBLASLt)

_LAYER_ACTS = {torch.nn.SiLU: 1, torch.nn.ReLU: 2, torch.nn.Tanh: 3}


def pack_linear_layer(weight):
    """Operands of bgk_dense_layer for ``weight`` [n_out, n_in]: a list of passes (A f16 [G * S * 8, 64, 8], S, c, k0, k1), one per
    block of <= 256 input columns [k0, k1): the block scaled by a power of two (largest magnitude into [2^14, 2^15)), zero-padded to
    16 S columns and 128 G rows, per 128-row group in the split-f16 block layout of the coupling kernels (natural k order)."""
    W = weight.detach().float()
    n_out, n_in = W.shape
    G = (n_out + 127) // 128
    passes = []
    for k0 in range(0, n_in, 256):
        k1 = min(n_in, k0 + 256)
        S = int(_lib.lib().bgk_dense_layer_steps(k1 - k0))
        Wb = W[:, k0:k1]
        e = _h2_scale_exp(Wb)
        Wp = torch.zeros(128 * G, 16 * S, dtype=torch.float32, device=W.device)
        Wp[:n_out, :k1 - k0] = Wb * 2.0 ** e
        A = torch.cat([_pack_h2(Wp[g * 128:(g + 1) * 128], None, _h2_k_natural(S)) for g in range(G)], dim=0).contiguous()
        passes.append((A, S, 2.0 ** -e, k0, k1))
    return passes


def pack_linear_layer_device(weight):
    """Device-side twin of pack_linear_layer (bgk_pack_linear_layer: no host synchronisation): passes (A, S, cs, k0, k1) with cs the
    device pair {largest magnitude, unscale factor} of the block"""
    W = weight.detach()
    assert W.is_cuda and W.dtype == torch.float32 and W.dim() == 2 and W.stride(1) == 1
    n_out, n_in = W.shape
    G = (n_out + 127) // 128
    passes = []
    with torch.cuda.device(W.device):
        for k0 in range(0, n_in, 256):
            k1 = min(n_in, k0 + 256)
            S = int(_lib.lib().bgk_dense_layer_steps(k1 - k0))
            A = torch.empty((G * S * 8, 64, 8), dtype=torch.float16, device=W.device)
            cs = torch.empty(2, dtype=torch.float32, device=W.device)
            st = _lib.lib().bgk_pack_linear_layer(W.data_ptr() + 4 * k0, W.stride(0), n_out, k1 - k0, _lib.ptr(A), _lib.ptr(cs),
                                                  _lib.stream_ptr(W.device))
            _lib.check(st, "bgk_pack_linear_layer")
            passes.append((A, S, cs, k0, k1))
    return passes


def _layer_operands(lin, transposed=False):
    """Packed operands of a Linear module that follow its weight ON THE DEVICE (round 6; ``transposed``: those of W^T, for dX = g W): the buffers are allocated once per (shape,
    device); every call launches bgk_refresh_linear_layer per column block, which fingerprints the live weight and re-packs the block
    only when it changed.  No host-side version key: an update torch's version counter does not see (``p.data.mul_(2)``, an old-style
    optimizer, a kernel writing through a view of the parameter) is picked up like any other -- what ``torch.nn.Linear`` does, since it
    reads the parameter itself.  Passes: (A, S, cs, k0, k1)."""
    Wp = lin.weight
    W = Wp.detach() if Wp.stride(1) == 1 else Wp.detach().contiguous()
    assert W.is_cuda and W.dtype == torch.float32 and W.dim() == 2
    n_out, n_in = (W.shape[1], W.shape[0]) if transposed else W.shape        # of the matrix the operands stand for
    key = (n_out, n_in, W.device)
    slot = "_bgk_layer_ops_t" if transposed else "_bgk_layer_ops"
    cached = lin.__dict__.get(slot)
    if cached is None or cached[0] != key:
        G = (n_out + 127) // 128
        passes = []
        for k0 in range(0, n_in, 256):
            k1 = min(n_in, k0 + 256)
            S = int(_lib.lib().bgk_dense_layer_steps(k1 - k0))
            passes.append((torch.empty((G * S * 8, 64, 8), dtype=torch.float16, device=W.device), S,
                           torch.zeros(2, dtype=torch.float32, device=W.device), k0, k1,
                           torch.zeros(2, dtype=torch.int64, device=W.device)))
        cached = (key, passes)
        lin.__dict__[slot] = cached
    with torch.cuda.device(W.device):
        for A, _S, cs, k0, k1, state in cached[1]:
            # column block [k0, k1) of the operand matrix: columns of W, or (transposed) rows k0 .. k1 of W
            base = W.data_ptr() + 4 * (k0 * W.stride(0) if transposed else k0)
            st = _lib.lib().bgk_refresh_linear_layer(base, W.stride(0), n_out, k1 - k0, int(transposed), _lib.ptr(A), _lib.ptr(cs), _lib.ptr(state),
                                                     _lib.stream_ptr(W.device))
            _lib.check(st, "bgk_refresh_linear_layer")
    return [p[:5] for p in cached[1]]


def dense_layer(x, lin, act=0, transposed=False):
    """y = act(x W^T + b) of a Linear module on bgk_dense_layer (x: f32 HIP tensor [..., n_in]; act: 0 none, 1 SiLU, 2 ReLU, 3 Tanh).
    ``transposed``: y = x W (no bias): the layer's input gradient for x = the gradient of its output."""
    n_in_x = lin.out_features if transposed else lin.in_features
    if x.dim() < 1 or x.shape[-1] != n_in_x:
        raise RuntimeError(f"dense_layer: input of shape {tuple(x.shape)} for a Linear layer with {n_in_x} input features "
                           f"(mat1 and mat2 shapes cannot be multiplied)")
    _lib.require_hip(x)
    lead = x.shape[:-1]
    x2, ldx = _lib.rowmajor(x.reshape(-1, x.shape[-1]))
    B, n_out = x2.shape[0], (lin.in_features if transposed else lin.out_features)
    y = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
    if B and n_out:
        passes = _layer_operands(lin, transposed)
        bias = None if (lin.bias is None or transposed) else lin.bias.detach().contiguous()
        with torch.cuda.device(x.device):
            for i, (A, S, c, k0, k1) in enumerate(passes):
                last = i == len(passes) - 1
                c_host, c_dev = (1.0, _lib.ptr(c)) if torch.is_tensor(c) else (c, None)
                st = _lib.lib().bgk_dense_layer(x2.data_ptr() + 4 * k0, ldx, B, k1 - k0, _lib.ptr(A), S, c_host, c_dev,
                                                _lib.ptr(bias) if last else None, n_out, act if last else 0, _lib.ptr(y), n_out, int(i > 0),
                                                _lib.stream_ptr(x.device))
                _lib.check(st, "bgk_dense_layer")
    return y.reshape(*lead, n_out)


def _on_layer_kernel(m, x):
    # (a Linear with forward / pre-forward hooks -- old-style weight_norm, activation capture -- runs as the module itself: m(x))
    return (LAYER_KERNEL and type(m) is torch.nn.Linear and not m._forward_hooks and not m._forward_pre_hooks
            and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 1
            and m.weight.dtype == torch.float32 and m.weight.device == x.device and m.in_features > 0
            and (m.bias is None or m.bias.dtype == torch.float32))


def _run_layers(layers, x):
    """Sequential forward.  Linear layers on HIP tensors run on bgk_dense_layer -- with the following SiLU / ReLU / Tanh in the same
    launch when nothing needs a gradient; 2-D inputs under autograd go through _LinearFn (the kernel forward, GEMM backward)."""
    mods = list(layers)
    grad = torch.is_grad_enabled()
    i = 0
    while i < len(mods):
        m = mods[i]
        if _on_layer_kernel(m, x):
            needs = grad and (x.requires_grad or m.weight.requires_grad or (m.bias is not None and m.bias.requires_grad))
            if not needs:
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                act = _LAYER_ACTS.get(type(nxt), 0) if nxt is not None and not nxt._forward_hooks and not nxt._forward_pre_hooks else 0
                x = dense_layer(x, m, act)
                i += 2 if act else 1
                continue
            if x.dim() == 2 and m.bias is not None:
                x = _LinearFn.apply(x, m.weight, m.bias, m)
                i += 1
                continue
        elif grad and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and type(m) is torch.nn.Linear and m.bias is not None \
                and (m.weight.requires_grad or x.requires_grad):
            x = _LinearFn.apply(x, m.weight, m.bias, None)
            i += 1
            continue
        elif LAYER_KERNEL and LAYER_BACKWARD and grad and type(m) in _LAYER_ACTS and x.is_cuda and x.dtype == torch.float32 and x.requires_grad \
                and x.dim() >= 1 and not m._forward_hooks and not m._forward_pre_hooks:
            x = _ActFn.apply(x, _LAYER_ACTS[type(m)])      # the activation and its VJP on bgk_activation / _backward (round 6; before: aten)
            i += 1
            continue
        x = m(x)
        i += 1
    return x


class DenseNet(torch.nn.Module):
    """Multi-layer perceptron ``n_units[0] -> ... -> n_units[-1]`` with ``activation`` after every
    hidden layer (nn/dense.py:9-48)."""

    def __init__(self, n_units, activation=None, weight_scale=1.0, bias_scale=0.0):
        super().__init__()
        if is_list_or_tuple(activation):
            assert len(activation) == len(n_units) - 2
        layers = []
        n_layers = len(n_units) - 1
        for i in range(n_layers):
            lin = torch.nn.Linear(n_units[i], n_units[i + 1])
            lin.weight.data *= weight_scale
            if bias_scale > 0.0:
                lin.bias.data = torch.rand_like(lin.bias.data) * bias_scale
            layers.append(lin)
            if i < n_layers - 1 and activation is not None:
                layers.append(activation[i] if is_list_or_tuple(activation) else activation)
        self._layers = torch.nn.Sequential(*layers)

    def forward(self, x):
        return _run_layers(self._layers, x)


class MeanFreeDenseNet(DenseNet):
    def forward(self, x):
        y = _run_layers(self._layers, x)
        return y - y.mean(dim=1, keepdim=True)


class WrapPeriodic(torch.nn.Module):
    """Feed ``net`` with (cos, sin) of the periodic inputs followed by the other inputs
    (nn/periodic.py:7-37)."""

    def __init__(self, net, left=0.0, right=1.0, indices=slice(None)):
        super().__init__()
        self.net = net
        self.left = left
        self.right = right
        self.indices = indices

    def forward(self, x):
        n = x.shape[-1]
        per = np.arange(n)[self.indices]
        other = np.setdiff1d(np.arange(n), per)
        if len(other) == 0 and np.array_equal(per, np.arange(n)):
            xp, xo = x, x[..., :0]        # all inputs periodic: no gather (and no scatter-add in backward)
        else:
            xp, xo = x[..., per], x[..., other]
        ang = 2 * np.pi * (xp - self.left) / (self.right - self.left)
        feats = torch.cat([torch.cos(ang), torch.sin(ang), xo], dim=-1)
        return self.net.forward(feats)


class WrapDistances(torch.nn.Module):
    """Feed ``net`` with the other inputs followed by all pairwise distances (i < j, row-major) of the points whose flattened
    xyz coordinates sit at ``indices`` (nn/periodic.py:40-58).  Stock torch ops: a conditioner front end outside the fused
    envelope."""

    def __init__(self, net, left=0.0, right=1.0, indices=slice(None)):
        super().__init__()
        self.net = net
        self.left = left
        self.right = right
        self.indices = indices

    def forward(self, x):
        n = x.shape[-1]
        picked = np.arange(n)[self.indices]
        rest = np.setdiff1d(np.arange(n), picked)
        points = x[..., picked].view(x.shape[0], -1, 3)
        dmat = torch.cdist(points, points)
        upper = torch.triu(torch.ones_like(dmat), diagonal=1).bool()
        dists = dmat[upper].view(x.shape[0], -1)
        return self.net.forward(torch.cat([x[..., rest], dists], dim=-1))


_ACT_CODES = {torch.nn.SiLU: 1, torch.nn.ReLU: 2, torch.nn.Tanh: 3}


def _reject(transformer, reason):
    """A coupling whose conditioner LOOKS fusable (a DenseNet, optionally behind WrapPeriodic) leaves the one-launch kernels'
    envelope: say so once per transformer and reason -- the layer-by-layer path (bgk_dense_layer per Linear + the stand-alone transformer
    kernel) is several times slower (2.2 vs 0.5 ms per cfg-3 layer at 2^20, profiles/r05_w256_layer.txt)."""
    seen = transformer.__dict__.setdefault("_fused_rejections", set())
    if reason not in seen:
        seen.add(reason)
        import warnings
        warnings.warn(f"{type(transformer).__name__}: not running as ONE fused kernel ({reason}); running the conditioner layer by layer "
                      f"(bgk_dense_layer) + the stand-alone transformer kernel", RuntimeWarning, stacklevel=3)
    return None


def _looks_dense(net):
    inner = net.net if type(net) is WrapPeriodic else net
    return type(inner) is DenseNet


def _fusable_dense(net):
    """Return (linears, act_code) if ``net`` is Linear-act-Linear-act-Linear with one supported
    activation type, else None."""
    if type(net) is not DenseNet:
        return None
    mods = list(net._layers)
    if len(mods) != 5:
        return None
    l0, a0, l1, a1, l2 = mods
    if not all(isinstance(m, torch.nn.Linear) for m in (l0, l1, l2)):
        return None
    if type(a0) is not type(a1) or type(a0) not in _ACT_CODES:
        return None
    if any(m.bias is None for m in (l0, l1, l2)):
        return None
    return (l0, l1, l2), _ACT_CODES[type(a0)]


def _fusable_dense_any(net):
    """(linears, act_code) for Linear-(act-Linear) x L, L >= 1 hidden layers, biases, one supported activation type; else None"""
    if type(net) is not DenseNet:
        return None
    mods = list(net._layers)
    if len(mods) < 3 or len(mods) % 2 == 0:
        return None
    lins, acts = mods[0::2], mods[1::2]
    if not all(isinstance(m, torch.nn.Linear) and m.bias is not None for m in lins):
        return None
    if any(type(a) is not type(acts[0]) for a in acts) or type(acts[0]) not in _ACT_CODES:
        return None
    return tuple(lins), _ACT_CODES[type(acts[0])]


def _fusable_dense_deep(net):
    """(linears, act_code) for Linear-(act-Linear) x 2 | x 3 with one supported activation type (the affine coupling kernels take
    two or three hidden layers), else None"""
    if type(net) is not DenseNet:
        return None
    mods = list(net._layers)
    if len(mods) not in (5, 7):
        return None
    lins, acts = mods[0::2], mods[1::2]
    if not all(isinstance(m, torch.nn.Linear) and m.bias is not None for m in lins):
        return None
    if any(type(a) is not type(acts[0]) for a in acts) or type(acts[0]) not in _ACT_CODES:
        return None
    return tuple(lins), _ACT_CODES[type(acts[0])]


# ---- MFMA operand packing for bgk_coupling_rqs_dense ------------------------------------------------
# One k-step of the f32 MFMA (v_mfma_f32_32x32x2_f32, A = weights) over 4 output tiles consumes, per
# lane l (i = l & 31, h = l >> 5), the four values W[32*m + i][k(step, h)], m = 0..3: stored as one
# float4 -> a packed layer is [steps + 1][64 lanes][4] floats; the extra last step carries the bias in
# the lower half-wave (A = bias, B = 1.0) and zeros in the upper one.
#   layer 0 (input from LDS, natural order):            k(t, h) = 2 t + h
#   hidden layers (input = accumulator registers):      k(16 kb + r, h) = 32 kb + (r & 3) + 8 (r >> 2) + 4 h
_LANE_I = torch.arange(64) & 31
_LANE_H = torch.arange(64) >> 5


def _pack_steps(W, bias, k_of_step):
    """W [128, n_in] (rows = output features), k_of_step [S, 2] (k index for half-wave 0 / 1, -1 = none)"""
    S = k_of_step.shape[0]
    dev = W.device
    rows = (32 * torch.arange(4, device=dev)[None, None, :] + _LANE_I.to(dev)[None, :, None]).expand(S, 64, 4)
    kk = k_of_step.to(dev)[:, _LANE_H.to(dev)][:, :, None].expand(S, 64, 4)
    vals = W[rows.reshape(-1), kk.clamp_min(0).reshape(-1)].reshape(S, 64, 4)
    vals = torch.where(kk >= 0, vals, torch.zeros_like(vals))
    bstep = torch.zeros(1, 64, 4, dtype=W.dtype, device=dev)
    bstep[0, :32, :] = bias.reshape(4, 32).t()
    return torch.cat([vals, bstep], dim=0).contiguous()


def _k_natural(n_in):
    T = ((n_in + 1) // 2 + 3) & ~3          # k-steps padded to a multiple of 4 (zero weights)
    k = torch.arange(2 * T).reshape(T, 2)
    return torch.where(k < n_in, k, torch.full_like(k, -1))


def _k_hidden():
    s = torch.arange(64)
    kb, r = s // 16, s % 16
    k0 = 32 * kb + (r & 3) + 8 * (r >> 2)
    return torch.stack([k0, k0 + 4], dim=1)


def pack_dense_for_fused(linears, nc_slot_host, d, n_bins):
    """Pack the three Linear layers of a DenseNet([n_in, 128, 128, P]) for bgk_coupling_rqs_dense.
    Returns (W0p, W1p, W2p) float32 device tensors."""
    l0, l1, l2 = linears
    W0p = _pack_steps(l0.weight.detach().float(), l0.bias.detach().float(), _k_natural(l0.in_features))
    W1p = _pack_steps(l1.weight.detach().float(), l1.bias.detach().float(), _k_hidden())
    # last layer: rows (= reference params columns) regrouped per transformed dim by bgk_pack_rqs_columns
    ncp = _lib.lib().bgk_pack_rqs_columns(d, n_bins, None, None)
    src = np.empty(ncp, dtype=np.int32)
    slots = np.ascontiguousarray(nc_slot_host, dtype=np.int32)
    _lib.lib().bgk_pack_rqs_columns(d, n_bins, slots.ctypes.data, src.ctypes.data)
    src_t = torch.as_tensor(src.astype(np.int64), device=l2.weight.device)
    W2 = l2.weight.detach().float()
    b2 = l2.bias.detach().float()
    W2r = torch.where(src_t[:, None] >= 0, W2[src_t.clamp_min(0)], torch.zeros((), dtype=W2.dtype, device=W2.device))
    b2r = torch.where(src_t >= 0, b2[src_t.clamp_min(0)], torch.zeros((), dtype=b2.dtype, device=b2.device))
    chunks = [_pack_steps(W2r[c * 128:(c + 1) * 128], b2r[c * 128:(c + 1) * 128], _k_hidden()) for c in range(ncp // 128)]
    return W0p, W1p, torch.cat(chunks, dim=0).contiguous()


# ---- split-f16 operand packing for bgk_coupling_rqs_dense_h2 ---------------------------------------
# v = hi + lo with hi = rne_f16(v), lo = rne_f16(v - hi); weights are first scaled by 2^s (exact) so that
# max(|W|, |b|) lands in [2^14, 2^15).  One 1 KiB block = 64 lanes x 8 f16; block(s, m, p) = (s*4 + m)*2 + p
# holds, for lane l = 32 kb + i, W'[32 m + i][k(s, kb, e)], e = 0..7, part p; after the S k16-steps follow 4
# bias blocks (lanes < 32: {b_hi, b_lo, 0, ...}).  See bgk_fused.hip (coupling_rqs_dense_h2_kernel).
# module default of the fused kernel's conditioner GEMMs; a transformer's `gemm_mode` attribute overrides it:
#   "f16x2": split-f16 on the f16 matrix cores (f32-class accuracy, see bgk_fused.hip; ~1.75x the layer throughput)
#   "f32":   f32-input MFMA = exact k-ordered fma chain, bit-identical to the CPU oracle
#   "bf16":  single-bf16 weights and GEMM inputs, f32 accumulate (reduced precision: the "bf16" leg of BASELINE config 5;
#            spline layers only -- affine layers and training run split-f16)
GEMM_MODE = "f16x2"


def _h2_scale_exp(*tensors):
    m = max(float(t.abs().max()) for t in tensors if t.numel())
    if not np.isfinite(m) or m <= 0.0:
        return 0
    return int(np.clip(np.floor(np.log2(32768.0 / m)), -16, 24))


def _h2_split(v):
    hi = v.to(torch.float16)
    lo = (v - hi.to(torch.float32)).to(torch.float16)
    return hi, lo


def _h2_k_natural(S):
    s, kb, e = torch.meshgrid(torch.arange(S), torch.arange(2), torch.arange(8), indexing="ij")
    return 16 * s + 8 * kb + e


def _h2_k_hidden(HT=4):
    s, kb, e = torch.meshgrid(torch.arange(2 * HT), torch.arange(2), torch.arange(8), indexing="ij")
    return 32 * (s >> 1) + (e & 3) + 8 * (2 * (s & 1) + (e >> 2)) + 4 * kb


def _pack_h2(Ws, bs, kidx, NT=4):
    """Ws [32 NT, Kdim] scaled f32 weights, bs [32 NT] scaled bias or None, kidx [S, 2, 8] -> f16 tensor of
    (S * NT * 2 [+ NT]) blocks x 64 lanes x 8."""
    dev = Ws.device
    S = kidx.shape[0]
    lane_i, lane_kb = _LANE_I.to(dev), _LANE_H.to(dev)
    rows = 32 * torch.arange(NT, device=dev)[:, None] + lane_i[None, :]                 # [NT, 64]
    k = kidx.to(dev)[:, lane_kb, :]                                                      # [S, 64, 8]
    vals = Ws[rows[None, :, :, None].expand(S, NT, 64, 8), k[:, None, :, :].expand(S, NT, 64, 8)]
    hi, lo = _h2_split(vals)
    blocks = torch.stack([hi, lo], dim=2).reshape(S * NT * 2, 64, 8)                     # [S, NT, 2, 64, 8]
    if bs is None:
        return blocks.contiguous()
    bb = torch.zeros(NT, 64, 8, dtype=torch.float16, device=dev)
    bhi, blo = _h2_split(bs.reshape(NT, 32))
    bb[:, :32, 0] = bhi
    bb[:, :32, 1] = blo
    return torch.cat([blocks, bb], dim=0).contiguous()


def _pad_rows(W, b, R):
    Wp = torch.zeros(R, W.shape[1], dtype=torch.float32, device=W.device)
    bp = torch.zeros(R, dtype=torch.float32, device=W.device)
    Wp[:W.shape[0]] = W
    bp[:b.shape[0]] = b
    return Wp, bp


def pack_dense_for_affine_h2(linears):
    """Pack DenseNet([n_in, H, H, d]) (H = 64 | 128, d <= 96) for bgk_coupling_affine_dense_h2, or DenseNet([n_in, H, H, H, d])
    for bgk_coupling_affine_dense_h3.  Returns (A0, A1, A2 f16 device tensors, (c0, c1, c2)[, A1b, c1b])."""
    if len(linears) == 4:
        l0, l1, l1b, l2 = linears
        A0, A1, A2, cs = pack_dense_for_affine_h2((l0, l1, l2))
        W, b = l1b.weight.detach().float(), l1b.bias.detach().float()
        e = _h2_scale_exp(W, b)
        return A0, A1, A2, cs, _pack_h2(W * 2.0 ** e, b * 2.0 ** e, _h2_k_hidden(l1b.out_features // 32), NT=l1b.out_features // 32), 2.0 ** -e
    l0, l1, l2 = linears
    W0, b0 = l0.weight.detach().float(), l0.bias.detach().float()
    W1, b1 = l1.weight.detach().float(), l1.bias.detach().float()
    W2, b2 = l2.weight.detach().float(), l2.bias.detach().float()
    n_in, H, d = l0.in_features, l0.out_features, l2.out_features
    HT, OT = H // 32, (d + 31) // 32
    S0 = (n_in + 1 + 15) // 16
    e0, e1, e2 = _h2_scale_exp(W0, b0), _h2_scale_exp(W1, b1), _h2_scale_exp(W2, b2)
    W0e = torch.zeros(H, 16 * S0, dtype=torch.float32, device=W0.device)
    W0e[:, :n_in] = W0
    W0e[:, n_in] = b0
    A0 = _pack_h2(W0e * 2.0 ** e0, None, _h2_k_natural(S0), NT=HT)
    A1 = _pack_h2(W1 * 2.0 ** e1, b1 * 2.0 ** e1, _h2_k_hidden(HT), NT=HT)
    W2p, b2p = _pad_rows(W2 * 2.0 ** e2, b2 * 2.0 ** e2, 32 * OT)
    A2 = _pack_h2(W2p, b2p, _h2_k_hidden(HT), NT=OT)
    return A0, A1, A2, (2.0 ** -e0, 2.0 ** -e1, 2.0 ** -e2)


def pack_dense_for_affine_deep(linears):
    """Pack DenseNet([n_in, H, ..., H, d]) with 1 .. 8 hidden layers (H = 64 | 128, d <= 96) for bgk_coupling_affine_dense_deep: layer 0
    and the output layer as pack_dense_for_affine_h2, the hidden -> hidden layers back to back.  Returns (A0, A1 | None, A2, c0, [c1 per
    hidden -> hidden layer], c2)."""
    l0, l_out, mids = linears[0], linears[-1], linears[1:-1]
    W0, b0 = l0.weight.detach().float(), l0.bias.detach().float()
    W2, b2 = l_out.weight.detach().float(), l_out.bias.detach().float()
    n_in, H, d = l0.in_features, l0.out_features, l_out.out_features
    HT, OT = H // 32, (d + 31) // 32
    S0 = (n_in + 1 + 15) // 16
    e0, e2 = _h2_scale_exp(W0, b0), _h2_scale_exp(W2, b2)
    W0e = torch.zeros(H, 16 * S0, dtype=torch.float32, device=W0.device)
    W0e[:, :n_in] = W0
    W0e[:, n_in] = b0
    A0 = _pack_h2(W0e * 2.0 ** e0, None, _h2_k_natural(S0), NT=HT)
    blocks, c1s = [], []
    for lin in mids:
        W, b = lin.weight.detach().float(), lin.bias.detach().float()
        e = _h2_scale_exp(W, b)
        blocks.append(_pack_h2(W * 2.0 ** e, b * 2.0 ** e, _h2_k_hidden(HT), NT=HT))
        c1s.append(2.0 ** -e)
    A1 = torch.cat(blocks, dim=0).contiguous() if blocks else None
    W2p, b2p = _pad_rows(W2 * 2.0 ** e2, b2 * 2.0 ** e2, 32 * OT)
    A2 = _pack_h2(W2p, b2p, _h2_k_hidden(HT), NT=OT)
    return A0, A1, A2, 2.0 ** -e0, c1s, 2.0 ** -e2


def _affine_plan(transformer, y_dim):
    """Decide (and cache) whether an AffineTransformer's conditioners can run fused; pack their weights."""
    nets = (transformer._shift_transformation, transformer._scale_transformation)
    if all(n is None for n in nets):
        return None
    specs = []
    periodic = None
    for n in nets:
        if n is None:
            specs.append(None)
            continue
        per = type(n) is WrapPeriodic
        if per:
            inner = n.net
            n_raw = getattr(getattr(inner, "_layers", [None])[0], "in_features", 0) // 2
            idx = np.arange(n_raw)[n.indices] if n_raw else np.zeros(0, int)
            if not (n.left == 0.0 and n.right == 1.0) or n_raw == 0 or not np.array_equal(np.asarray(idx), np.arange(n_raw)):
                return None          # only "all conditioner inputs periodic on [0, 1]" is fused
            n = inner
        if periodic is not None and per != periodic:
            return None
        periodic = per
        spec = _fusable_dense_deep(n)
        if spec is None:
            spec = _fusable_dense_any(n)          # any other depth: bgk_coupling_affine_dense_deep
            if spec is not None and len(spec[0]) - 1 > DEEP_MAX_HIDDEN:
                spec = None
        if spec is None:
            if type(n) is DenseNet:
                return _reject(transformer, "the fused affine kernels take DenseNets with 1 .. 8 hidden layers, biases and one of "
                                            "SiLU / ReLU / Tanh")
            return None
        specs.append(spec)
    live = [sp for sp in specs if sp is not None]
    n_in, depth = live[0][0][0].in_features, len(live[0][0])
    anydepth = depth not in (3, 4)                      # other than two / three hidden layers: the loop kernel, hidden widths may differ
    H = max(m.out_features for lins, _ in live for m in lins[:-1])
    for lins, _ in live:
        if len(lins) != depth or lins[0].in_features != n_in or lins[-1].out_features != y_dim:
            return None
        if not anydepth and (any(m.out_features != H for m in lins[:-1]) or any(m.in_features != H for m in lins[1:])):
            return None
    if H > 128 or y_dim > 96 or n_in > 127 or (periodic and n_in % 2):
        return _reject(transformer, f"hidden width {H} / {y_dim} transformed dims / {n_in} input features: fused for widths up to 128, "
                                    f"<= 96 dims, <= 127 input features")
    H_run = 64 if H <= 64 else 128                      # other widths run zero-padded to the kernels' 64 / 128 rows
    params = [p for (ls, _) in live for lin in ls for p in (lin.weight, lin.bias)]
    version = tuple(param_state_key(p) for p in params)
    cache = transformer._fused_cache
    if cache.get("version") != version or cache.get("y_dim") != y_dim:
        cache.clear()
        if H_run != H or (anydepth and any(m.out_features != H_run for lins, _ in live for m in lins[:-1])):
            # (only when the weights changed: the padded copies live in the packed operands)
            specs = [None if sp is None else (_pad_hidden(sp[0], H_run), sp[1]) for sp in specs]
        pack = pack_dense_for_affine_deep if anydepth else pack_dense_for_affine_h2
        cache.update(version=version, y_dim=y_dim, hidden=H_run, d_c=n_in // 2 if periodic else n_in, periodic=bool(periodic), depth=depth,
                     anydepth=anydepth, packed=[None if sp is None else (pack(sp[0]), sp[1]) for sp in specs])
    return cache


class _PaddedLinear:
    """weight / bias of a Linear zero-padded to (out_to, in_to): the stand-in the packers read.  Padded hidden units have weight 0
    and bias 0, so they hold act(0) = 0 (SiLU / ReLU / Tanh) and feed zero columns of the next layer: same function."""

    def __init__(self, lin, out_to, in_to):
        W, b = lin.weight.detach(), lin.bias.detach()
        self.weight = torch.zeros((out_to, in_to), dtype=W.dtype, device=W.device)
        self.weight[:W.shape[0], :W.shape[1]] = W
        self.bias = torch.zeros((out_to,), dtype=b.dtype, device=b.device)
        self.bias[:b.shape[0]] = b
        self.in_features, self.out_features = in_to, out_to


def _pad_hidden(linears, H):
    """the Linear chain with every hidden width zero-padded to H (input and output widths unchanged)"""
    n = len(linears)
    return tuple(_PaddedLinear(lin, H if i < n - 1 else lin.out_features, H if i > 0 else lin.in_features)
                 for i, lin in enumerate(linears))



def _cond_parts(x):
    """the conditioning tensors of a coupling: [x] or the parts of a flow.CatView (<= 3 of them go to the kernels unconcatenated)"""
    from .flow import CatView
    if isinstance(x, CatView):
        return list(x.parts) if len(x.parts) <= 3 else [x.cat()]
    return [x]


def fused_affine_coupling(transformer, x, y, inverse, out=None, dlogp=None, accumulate=False, acc=None):
    """Try the one-launch affine coupling layer (bgk_coupling_affine_dense_h2).  Returns (y', dlogp) or None when
    the conditioners are not fusable DenseNets (the caller then runs the nets + bgk_affine_transform).
    ``out`` ([B, d] rows, any row stride; may alias ``y``: every element is read and written by the same lane) and
    ``dlogp`` ([B] contiguous, ``accumulate``: added to instead of overwritten) let a caller chain layers without copies."""
    if _gemm_mode(transformer) == "f32":
        return _reject(transformer, "gemm_mode 'f32' has no fused affine kernel")     # "f32" selects the generic path ("bf16": split-f16)
    if x.dim() != 2 or y.dim() != 2 or not x.is_cuda or x.dtype != torch.float32:
        return None
    plan = _affine_plan(transformer, y.shape[-1])
    if plan is None or x.shape[-1] != plan["d_c"]:
        return None
    parts = _cond_parts(x)
    if len(parts) > 1 and plan["hidden"] != 128:
        parts = [x.cat()]                 # several conditioning tensors: the width-128 kernel only
    _lib.require_hip(y, *parts)
    y2, ldy = _lib.rowmajor(y)
    B, d = y2.shape
    if out is None:
        out = torch.empty((B, d), dtype=torch.float32, device=y.device)
    if acc is not None:
        dlogp, accumulate = acc.peek()
    if dlogp is None:
        dlogp, accumulate = torch.empty((B,), dtype=torch.float32, device=y.device), False
    ldo = out.stride(0)
    assert out.shape == (B, d) and out.stride(1) == 1 and dlogp.shape == (B,) and dlogp.is_contiguous()
    if plan.get("anydepth"):
        return _fused_affine_anydepth(transformer, plan, parts[0] if len(parts) == 1 else x.cat(), y2, ldy, B, d, out, ldo, dlogp, accumulate,
                                      inverse, acc)
    args = []
    deep = plan["depth"] == 4
    for entry in plan["packed"]:
        if entry is None:
            args += ([None] * 4 + [1.0] * 4 + [0]) if deep else [None, None, None, 1.0, 1.0, 1.0, 0]
        else:
            packed, act = entry
            if packed[0].device != y.device:
                return None
            A0, A1, A2, (c0, c1, c2) = packed[:4]
            if deep:
                args += [_lib.ptr(A0), _lib.ptr(A1), _lib.ptr(packed[4]), _lib.ptr(A2), c0, c1, packed[5], c2, act]
            else:
                args += [_lib.ptr(A0), _lib.ptr(A1), _lib.ptr(A2), c0, c1, c2, act]
    log_alpha = transformer._log_alpha.detach().to(device=y.device, dtype=torch.float32)
    lib = _lib.lib()
    tail = (*args, plan["hidden"], _lib.ptr(log_alpha), int(transformer._preserve_volume), int(transformer._is_circular), int(inverse),
            _lib.ptr(y2), ldy, B, d, _lib.ptr(out), ldo, _lib.ptr(dlogp), int(bool(accumulate)), _lib.stream_ptr(y.device))
    with torch.cuda.device(y.device):
        st = -2
        if len(parts) > 1:
            ptrs, lds, widths, n, keep = _lib.cond_segments(parts)
            st = (lib.bgk_coupling_affine_dense_h3_mc if deep else lib.bgk_coupling_affine_dense_h2_mc)(
                ptrs, lds, widths, n, int(plan["periodic"]), *tail)
            if st == -2:
                parts = [x.cat()]         # a kernel without the segment table: concatenate after all
        if st == -2:
            x2, ldc = _lib.rowmajor(parts[0])
            st = (lib.bgk_coupling_affine_dense_h3 if deep else lib.bgk_coupling_affine_dense_h2)(
                _lib.ptr(x2), ldc, plan["d_c"], int(plan["periodic"]), *tail)
    if st == -2:
        return _reject(transformer, "shape outside the fused affine kernels' envelope: " + _lib.lib().bgk_last_error().decode(errors="replace"))
    _lib.check(st, "bgk_coupling_affine_dense_h3" if deep else "bgk_coupling_affine_dense_h2")
    if acc is not None:
        acc.commit()
        return out, acc
    return out, dlogp[:, None]


def _fused_affine_anydepth(transformer, plan, x, y2, ldy, B, d, out, ldo, dlogp, accumulate, inverse, acc):
    """launch of bgk_coupling_affine_dense_deep (conditioners with 1, 4, 5 .. 8 hidden layers) for fused_affine_coupling"""
    args, keep = [], []
    for entry in plan["packed"]:
        if entry is None:
            args += [None, None, None, 1.0, None, 1.0, 0]
            continue
        (A0, A1, A2, c0, c1s, c2), act = entry
        if A0.device != y2.device:
            return None
        arr = (ctypes.c_float * max(1, len(c1s)))(*c1s)
        keep.append(arr)
        args += [_lib.ptr(A0), _lib.ptr(A1), _lib.ptr(A2), c0, arr, c2, act]
    log_alpha = transformer._log_alpha.detach().to(device=y2.device, dtype=torch.float32)
    x2, ldc = _lib.rowmajor(x)
    with torch.cuda.device(y2.device):
        st = _lib.lib().bgk_coupling_affine_dense_deep(
            _lib.ptr(x2), ldc, plan["d_c"], int(plan["periodic"]), *args, plan["depth"] - 1, plan["hidden"], _lib.ptr(log_alpha),
            int(transformer._preserve_volume), int(transformer._is_circular), int(inverse),
            _lib.ptr(y2), ldy, B, d, _lib.ptr(out), ldo, _lib.ptr(dlogp), int(bool(accumulate)), _lib.stream_ptr(y2.device))
    if st == -2:
        return _reject(transformer, "shape outside the fused affine kernels' envelope: " + _lib.lib().bgk_last_error().decode(errors="replace"))
    _lib.check(st, "bgk_coupling_affine_dense_deep")
    if acc is not None:
        acc.commit()
        return out, acc
    return out, dlogp[:, None]


def pack_dense_for_fused_h2(linears, nc_slot_host, d, n_bins):
    """Pack DenseNet([n_in, 128, 128, P]) for bgk_coupling_rqs_dense_h2.
    Returns (A0, A1, A2 f16 device tensors, (c0, c1, c2) unscale factors)."""
    l0, l1, l2 = linears
    W0, b0 = l0.weight.detach().float(), l0.bias.detach().float()
    W1, b1 = l1.weight.detach().float(), l1.bias.detach().float()
    W2, b2 = l2.weight.detach().float(), l2.bias.detach().float()
    n_in = l0.in_features
    S0 = (n_in + 1 + 15) // 16
    e0, e1, e2 = _h2_scale_exp(W0, b0), _h2_scale_exp(W1, b1), _h2_scale_exp(W2, b2)
    W0e = torch.zeros(128, 16 * S0, dtype=torch.float32, device=W0.device)
    W0e[:, :n_in] = W0
    W0e[:, n_in] = b0                                   # column of the constant-1 feature
    A0 = _pack_h2(W0e * 2.0 ** e0, None, _h2_k_natural(S0))
    A1 = _pack_h2(W1 * 2.0 ** e1, b1 * 2.0 ** e1, _h2_k_hidden())
    ncp = _lib.lib().bgk_pack_rqs_columns(d, n_bins, None, None)
    src = np.empty(ncp, dtype=np.int32)
    slots = np.ascontiguousarray(nc_slot_host, dtype=np.int32)
    _lib.lib().bgk_pack_rqs_columns(d, n_bins, slots.ctypes.data, src.ctypes.data)
    src_t = torch.as_tensor(src.astype(np.int64), device=W2.device)
    W2r = torch.where(src_t[:, None] >= 0, W2[src_t.clamp_min(0)], torch.zeros((), dtype=W2.dtype, device=W2.device)) * 2.0 ** e2
    b2r = torch.where(src_t >= 0, b2[src_t.clamp_min(0)], torch.zeros((), dtype=b2.dtype, device=b2.device)) * 2.0 ** e2
    A2 = torch.cat([_pack_h2(W2r[c * 128:(c + 1) * 128], b2r[c * 128:(c + 1) * 128], _h2_k_hidden())
                    for c in range(ncp // 128)], dim=0).contiguous()
    return A0, A1, A2, (2.0 ** -e0, 2.0 ** -e1, 2.0 ** -e2)


def pack_dense_for_fused_w256(linears, nc_slot_host, d, n_bins):
    """Pack DenseNet([n_in, 256, 256, P]) for the width-256 kernel behind bgk_coupling_rqs_dense_h2 (H0 = H1 = 256;
    bgk_fused.hip::coupling_rqs_dense_w256_kernel): every GEMM produces 128 output rows at a time, so layer 0 and layer 1 are packed as
    two 128-row halves (rows 0..127, then 128..255), each in the width-128 block layout with 16 k16-steps over the 256 hidden inputs
    (k order = the accumulator layout of 8 tiles), the parameter chunks likewise.  Returns (A0, A1, A2 f16 device tensors, (c0, c1, c2))."""
    l0, l1, l2 = linears
    W0, b0 = l0.weight.detach().float(), l0.bias.detach().float()
    W1, b1 = l1.weight.detach().float(), l1.bias.detach().float()
    W2, b2 = l2.weight.detach().float(), l2.bias.detach().float()
    assert W0.shape[0] == 256 and W1.shape == (256, 256) and W2.shape[1] == 256
    n_in = l0.in_features
    S0 = (n_in + 1 + 15) // 16
    e0, e1, e2 = _h2_scale_exp(W0, b0), _h2_scale_exp(W1, b1), _h2_scale_exp(W2, b2)
    W0e = torch.zeros(256, 16 * S0, dtype=torch.float32, device=W0.device)
    W0e[:, :n_in] = W0
    W0e[:, n_in] = b0                                   # column of the constant-1 feature
    kh = _h2_k_hidden(8)
    A0 = torch.cat([_pack_h2(W0e[g * 128:(g + 1) * 128] * 2.0 ** e0, None, _h2_k_natural(S0)) for g in range(2)], dim=0).contiguous()
    A1 = torch.cat([_pack_h2(W1[g * 128:(g + 1) * 128] * 2.0 ** e1, b1[g * 128:(g + 1) * 128] * 2.0 ** e1, kh) for g in range(2)],
                   dim=0).contiguous()
    src_t = _src_col_table(d, n_bins, nc_slot_host, W2.device).to(torch.int64)
    W2r = torch.where(src_t[:, None] >= 0, W2[src_t.clamp_min(0)], torch.zeros((), dtype=W2.dtype, device=W2.device)) * 2.0 ** e2
    b2r = torch.where(src_t >= 0, b2[src_t.clamp_min(0)], torch.zeros((), dtype=b2.dtype, device=b2.device)) * 2.0 ** e2
    A2 = torch.cat([_pack_h2(W2r[c * 128:(c + 1) * 128], b2r[c * 128:(c + 1) * 128], kh)
                    for c in range(src_t.numel() // 128)], dim=0).contiguous()
    return A0, A1, A2, (2.0 ** -e0, 2.0 ** -e1, 2.0 ** -e2)


DEEP_MAX_HIDDEN = 8     # hidden layers bgk_coupling_rqs_dense_deep takes (DEEP_MAX_HH + 1 in bgk_fused.hip)


def pack_dense_for_fused_deep(linears, nc_slot_host, d, n_bins):
    """Pack DenseNet([n_in, 128, ..., 128, P]) with 1 .. 8 hidden layers for bgk_coupling_rqs_dense_deep: layer 0 and the parameter chunks
    as pack_dense_for_fused_h2, the hidden -> hidden layers back to back in A1.  Returns (A0, A1 | None, A2, c0, [c1 per layer], c2)."""
    l0, l_out, mids = linears[0], linears[-1], linears[1:-1]
    W0, b0 = l0.weight.detach().float(), l0.bias.detach().float()
    W2, b2 = l_out.weight.detach().float(), l_out.bias.detach().float()
    n_in = l0.in_features
    S0 = (n_in + 1 + 15) // 16
    e0, e2 = _h2_scale_exp(W0, b0), _h2_scale_exp(W2, b2)
    W0e = torch.zeros(128, 16 * S0, dtype=torch.float32, device=W0.device)
    W0e[:, :n_in] = W0
    W0e[:, n_in] = b0                                   # column of the constant-1 feature
    A0 = _pack_h2(W0e * 2.0 ** e0, None, _h2_k_natural(S0))
    blocks, c1s = [], []
    for lin in mids:
        W, b = lin.weight.detach().float(), lin.bias.detach().float()
        e = _h2_scale_exp(W, b)
        blocks.append(_pack_h2(W * 2.0 ** e, b * 2.0 ** e, _h2_k_hidden()))
        c1s.append(2.0 ** -e)
    A1 = torch.cat(blocks, dim=0).contiguous() if blocks else None
    src_t = _src_col_table(d, n_bins, nc_slot_host, W2.device).to(torch.int64)
    W2r = torch.where(src_t[:, None] >= 0, W2[src_t.clamp_min(0)], torch.zeros((), dtype=W2.dtype, device=W2.device)) * 2.0 ** e2
    b2r = torch.where(src_t >= 0, b2[src_t.clamp_min(0)], torch.zeros((), dtype=b2.dtype, device=b2.device)) * 2.0 ** e2
    A2 = torch.cat([_pack_h2(W2r[c * 128:(c + 1) * 128], b2r[c * 128:(c + 1) * 128], _h2_k_hidden())
                    for c in range(src_t.numel() // 128)], dim=0).contiguous()
    return A0, A1, A2, 2.0 ** -e0, c1s, 2.0 ** -e2


DEVICE_PACK = True     # pack split-f16 operands with bgk_pack_dense_h2 (no host sync); False: the torch reference packer


def pack_dense_for_fused_h2_device(linears, src_col_dev, n_chunks, bufs=None, bf16=False):
    """Device-side twin of pack_dense_for_fused_h2 (bgk_pack_dense_h2): returns (A0, A1, A2, cs) with cs the
    device scale table {2^s, 2^-s} x 3.  ``bufs`` = previous result to overwrite in place."""
    l0, l1, l2 = linears
    dev = l0.weight.device
    n_in = l0.in_features
    S0 = (n_in + 1 + 15) // 16
    if bufs is None:
        A0 = torch.empty((S0 * 8, 64, 8), dtype=torch.float16, device=dev)
        A1 = torch.empty((8 * 8 + 4, 64, 8), dtype=torch.float16, device=dev)
        A2 = torch.empty((n_chunks * (8 * 8 + 4), 64, 8), dtype=torch.float16, device=dev)
        cs = torch.empty(6, dtype=torch.float32, device=dev)
    else:
        A0, A1, A2, cs = bufs
    ws = [t.detach().contiguous() for lin in linears for t in (lin.weight, lin.bias)]
    assert all(w.dtype == torch.float32 for w in ws)
    with torch.cuda.device(dev):
        st = _lib.lib().bgk_pack_dense_h2(
            _lib.ptr(ws[0]), _lib.ptr(ws[1]), n_in, 128, _lib.ptr(ws[2]), _lib.ptr(ws[3]), _lib.ptr(ws[4]), _lib.ptr(ws[5]),
            l2.out_features, _lib.ptr(src_col_dev), n_chunks, 4, int(bf16), _lib.ptr(A0), _lib.ptr(A1), _lib.ptr(A2), _lib.ptr(cs),
            _lib.stream_ptr(dev))
    _lib.check(st, "bgk_pack_dense_h2")
    return A0, A1, A2, cs


T_OPERAND_MAX_IN = 96     # widest conditioner input bgk_pack_dense_h2_t / bgk_dense_backward_dx take (three 32-feature output tiles)

BATCHED_REPACK = True     # training.FlatAdam.step re-packs the operands of every fused training layer in three launches

_TRAIN_PLANS = weakref.WeakSet()      # transformers whose fused plan ran a training forward on device-packed split-f16 operands


def _t_operand_bufs(bufs, P, n_in, dev):
    """transposed-weight operands T0..T2 of bgk_dense_backward_dx for a conditioner [n_in, 128, 128, P] (allocated once per plan)"""
    FT, S2 = (n_in + 31) // 32, ((P + 15) // 16 + 3) // 4 * 4      # T2: whole groups of 4 k-steps (zero blocks behind ceil(P / 16))
    key = (P, n_in, str(dev))
    if bufs.get("key") != key:
        bufs.clear()
        bufs.update(key=key, T0=torch.empty((8 * FT * 2 + FT, 64, 8), dtype=torch.float16, device=dev),
                    T1=torch.empty((8 * 8 + 4, 64, 8), dtype=torch.float16, device=dev),
                    T2=torch.empty((S2 * 8 + 4, 64, 8), dtype=torch.float16, device=dev))
    return bufs["T0"], bufs["T1"], bufs["T2"]


def repack_training_plans(param_ids=None):
    """After an optimizer step: the forward operands (bgk_pack_dense_h2_many) and the transposed backward operands
    (bgk_pack_dense_h2_t_many) of every fused layer that ran a training forward since the last call, in three launches per 16
    layers instead of a memset + four launches per layer -- 0.4 ms of a 17 ms KL step of the 16-layer flow.  ``param_ids``: ids of
    the parameters the caller just updated (layers with other parameters are left to the lazy per-layer path, as are zero-padded
    narrow conditioners and the torch-packed modes).  Returns the number of layers re-packed."""
    if not BATCHED_REPACK:
        return 0
    by_dev = {}
    for tr in list(_TRAIN_PLANS):
        cache = getattr(tr, "_fused_cache", None)
        if not cache or not cache.pop("train_used", False) or cache.get("mode") != "f16x2" or "bufs" not in cache or cache.get("padded"):
            continue
        net = tr._params_net
        inner = net.net if type(net) is WrapPeriodic else net
        spec = _fusable_dense(inner)
        if spec is None:
            continue
        (l0, l1, l2), _ = spec
        params = [p for lin in (l0, l1, l2) for p in (lin.weight, lin.bias)]
        if param_ids is not None and not all(id(p) in param_ids for p in params):
            continue
        if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
            continue
        version = tuple(param_state_key(p) for p in params)
        if cache.get("version") == version:
            continue
        by_dev.setdefault(params[0].device, []).append((cache, l0, l1, l2, params, version))
    done = 0
    for dev, items in by_dev.items():
        n = len(items)
        vps, i32s = (ctypes.c_void_p * n), (ctypes.c_int32 * n)
        col = lambda f: vps(*[f(it) for it in items])       # noqa: E731
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_pack_dense_h2_many(
                n, col(lambda it: it[4][0].data_ptr()), col(lambda it: it[4][1].data_ptr()), i32s(*[it[1].in_features for it in items]),
                col(lambda it: it[4][2].data_ptr()), col(lambda it: it[4][3].data_ptr()),
                col(lambda it: it[4][4].data_ptr()), col(lambda it: it[4][5].data_ptr()), i32s(*[it[3].out_features for it in items]),
                col(lambda it: it[0]["src_col_dev"].data_ptr()), i32s(*[it[0]["src_col_dev"].numel() // 128 for it in items]),
                col(lambda it: it[0]["bufs"][0].data_ptr()), col(lambda it: it[0]["bufs"][1].data_ptr()),
                col(lambda it: it[0]["bufs"][2].data_ptr()), col(lambda it: it[0]["bufs"][3].data_ptr()), _lib.stream_ptr(dev))
            _lib.check(st, "bgk_pack_dense_h2_many")
            # the transposed operands of the input-gradient chain exist for conditioner inputs of <= T_OPERAND_MAX_IN features only
            # (wider layers run that chain as GEMMs, _FusedSplineTrainFn.backward): those layers get the forward operands alone
            t_items = [it for it in items if it[1].in_features <= T_OPERAND_MAX_IN]
            if t_items:
                m = len(t_items)
                tb = [_t_operand_bufs(it[0].setdefault("tbufs", {}), it[3].out_features, it[1].in_features, dev) for it in t_items]
                tcol = lambda f: (ctypes.c_void_p * m)(*[f(it) for it in t_items])       # noqa: E731
                st = _lib.lib().bgk_pack_dense_h2_t_many(
                    m, tcol(lambda it: it[4][0].data_ptr()), (ctypes.c_int32 * m)(*[it[1].in_features for it in t_items]),
                    tcol(lambda it: it[4][2].data_ptr()), tcol(lambda it: it[4][4].data_ptr()),
                    (ctypes.c_int32 * m)(*[it[3].out_features for it in t_items]), tcol(lambda it: it[0]["bufs"][3].data_ptr()),
                    (ctypes.c_void_p * m)(*[t[0].data_ptr() for t in tb]), (ctypes.c_void_p * m)(*[t[1].data_ptr() for t in tb]),
                    (ctypes.c_void_p * m)(*[t[2].data_ptr() for t in tb]), _lib.stream_ptr(dev))
                _lib.check(st, "bgk_pack_dense_h2_t_many")
        for cache, l0, _l1, _l2, _params, version in items:
            cache["version"] = version
            if l0.in_features <= T_OPERAND_MAX_IN:
                cache["tbufs"]["t_version"] = (version[0], version[2], version[4])
        done += n
    return done


def _src_col_table(d, n_bins, nc_slot_host, device):
    ncp = _lib.lib().bgk_pack_rqs_columns(d, n_bins, None, None)
    src = np.empty(ncp, dtype=np.int32)
    slots = np.ascontiguousarray(nc_slot_host, dtype=np.int32)
    _lib.lib().bgk_pack_rqs_columns(d, n_bins, slots.ctypes.data, src.ctypes.data)
    return torch.as_tensor(src, device=device)


def _gemm_mode(transformer):
    mode = getattr(transformer, "gemm_mode", None) or GEMM_MODE
    if mode not in ("f32", "f16x2", "bf16"):
        raise ValueError(f"unknown gemm_mode {mode!r} (expected 'f32', 'f16x2' or 'bf16')")
    return mode


def _fused_plan(transformer, y_dim, nc_slot_host):
    """Decide (and cache) whether the transformer's conditioner can run fused; pack its weights."""
    mode = _gemm_mode(transformer)
    net = transformer._params_net
    periodic = False
    if type(net) is WrapPeriodic:
        if not (net.left == 0.0 and net.right == 1.0):
            return _reject(transformer, "WrapPeriodic on an interval other than [0, 1]") if _looks_dense(net) else None
        inner = net.net
        periodic = True
    else:
        inner = net
    spec = _fusable_dense(inner)
    if spec is None:
        deep = _fusable_dense_any(inner)
        if deep is not None and len(deep[0]) - 1 <= DEEP_MAX_HIDDEN and mode == "f16x2" \
                and all(lin.out_features <= 128 for lin in deep[0][:-1]):
            return _deep_plan(transformer, net, deep, periodic, y_dim, nc_slot_host)
        if type(inner) is DenseNet:
            return _reject(transformer, "the fused spline kernels take a DenseNet with biases and one of SiLU / ReLU / Tanh: two hidden "
                                        "layers of up to 256 units, or (mode 'f16x2') 1 .. 8 hidden layers of up to 128")
        return None
    (l0, l1, l2), act = spec
    H_max = max(l0.out_features, l1.out_features)
    if H_max > 256:
        return _reject(transformer, f"hidden layers ({l0.out_features}, {l1.out_features}): widths up to 256 are fused")
    if H_max > 128 and mode != "f16x2":
        return _reject(transformer, f"hidden layers ({l0.out_features}, {l1.out_features}) in gemm_mode '{mode}': widths above 128 are "
                                    f"fused in mode 'f16x2' only")
    H_run = 128 if H_max <= 128 else 256                # narrower hidden layers run zero-padded to the kernels' 128 / 256 rows
    padded = (l0.out_features, l1.out_features) != (H_run, H_run)
    if y_dim > 64:
        return _reject(transformer, f"{y_dim} transformed dims: at most 64 are fused")
    n_nc = int((nc_slot_host >= 0).sum())
    P = l2.out_features
    n_bins = (P - n_nc) // (3 * y_dim)
    if 3 * n_bins * y_dim + n_nc != P:
        return None
    if n_bins != 8 and not (n_bins in (4, 12, 16, 32) and mode == "f16x2"):
        # K = 4 | 12 | 16 | 32: split-f16 form only (first-generation kernel); anything else: generic path
        return _reject(transformer, f"{n_bins} bins in gemm_mode '{mode}': fused for 8 bins, and for 4 / 12 / 16 / 32 bins in mode 'f16x2'"
                                    + ("; more than 64 bins run on device torch ops" if n_bins > 64 else ""))
    d_c = l0.in_features // 2 if periodic else l0.in_features
    if periodic:
        idx = np.arange(d_c)[net.indices] if not isinstance(net.indices, slice) or net.indices != slice(None) else np.arange(d_c)
        if len(idx) != d_c or 2 * d_c != l0.in_features or not np.array_equal(np.asarray(idx), np.arange(d_c)):
            # only "all conditioner inputs periodic, in natural order" is fused (the kernel featurises columns 0..d_c-1)
            return _reject(transformer, "WrapPeriodic over a subset / permutation of the conditioner inputs")
    params = [p for lin in (l0, l1, l2) for p in (lin.weight, lin.bias)]
    version = tuple(param_state_key(p) for p in params)
    cache = transformer._fused_cache
    dev = l0.weight.device
    stale = cache.get("version") != version
    if cache.get("y_dim") != y_dim or cache.get("mode") != mode or cache.get("device") != dev:
        cache.clear()
        stale = True
    if stale:
        common = dict(version=version, y_dim=y_dim, mode=mode, device=dev, act=act, periodic=periodic, d_c=d_c, n_bins=n_bins,
                      padded=padded, hidden=H_run, circ_mask=int(sum(1 << j for j in range(y_dim) if nc_slot_host[j] < 0)))
        if padded:
            l0, l1, l2 = _pad_hidden((l0, l1, l2), H_run)
        if mode == "bf16" and dev.type != "cuda":
            return None
        if H_run == 256:       # the width-256 kernel (inference): operands from the torch packer
            cache.update(common, packed=pack_dense_for_fused_w256((l0, l1, l2), nc_slot_host, y_dim, n_bins), cs=None)
            cache.pop("bufs", None)
        elif mode == "bf16" or (mode == "f16x2" and DEVICE_PACK and dev.type == "cuda"):
            if "src_col_dev" not in cache:
                cache["src_col_dev"] = _src_col_table(y_dim, n_bins, nc_slot_host, dev)
            n_chunks = cache["src_col_dev"].numel() // 128
            A0, A1, A2, cs = pack_dense_for_fused_h2_device((l0, l1, l2), cache["src_col_dev"], n_chunks, cache.get("bufs"),
                                                            bf16=(mode == "bf16"))
            cache.update(common, bufs=(A0, A1, A2, cs), packed=(A0, A1, A2, (1.0, 1.0, 1.0)), cs=cs)
        else:
            pack = pack_dense_for_fused if mode == "f32" else pack_dense_for_fused_h2
            cache.update(common, packed=pack((l0, l1, l2), nc_slot_host, y_dim, n_bins), cs=None)
            cache.pop("bufs", None)
    return cache


def _deep_plan(transformer, net, spec, periodic, y_dim, nc_slot_host):
    """plan of a conditioner with 1, 3, 4, ... hidden layers (width <= 128) for bgk_coupling_rqs_dense_deep (inference), or None"""
    lins, act = spec
    if y_dim > 64:
        return _reject(transformer, f"{y_dim} transformed dims: at most 64 are fused")
    n_nc = int((nc_slot_host >= 0).sum())
    P = lins[-1].out_features
    n_bins = (P - n_nc) // (3 * y_dim)
    if 3 * n_bins * y_dim + n_nc != P:
        return None
    if n_bins not in (4, 8, 12, 16, 32):
        return _reject(transformer, f"{n_bins} bins: the one-launch kernels take 4 / 8 / 12 / 16 / 32")
    l0 = lins[0]
    d_c = l0.in_features // 2 if periodic else l0.in_features
    if periodic:
        idx = np.arange(d_c)[net.indices] if not isinstance(net.indices, slice) or net.indices != slice(None) else np.arange(d_c)
        if len(idx) != d_c or 2 * d_c != l0.in_features or not np.array_equal(np.asarray(idx), np.arange(d_c)):
            return _reject(transformer, "WrapPeriodic over a subset / permutation of the conditioner inputs")
    version = tuple(param_state_key(p) for lin in lins for p in (lin.weight, lin.bias))
    cache = transformer._fused_cache
    dev = l0.weight.device
    if cache.get("version") != version or cache.get("y_dim") != y_dim or cache.get("mode") != "f16x2" or cache.get("device") != dev \
            or not cache.get("deep"):
        cache.clear()
        padded = any(lin.out_features != 128 for lin in lins[:-1])
        run = _pad_hidden(lins, 128) if padded else lins
        cache.update(version=version, y_dim=y_dim, mode="f16x2", device=dev, act=act, periodic=periodic, d_c=d_c, n_bins=n_bins,
                     padded=padded, hidden=128, deep=len(lins) - 1, cs=None,
                     circ_mask=int(sum(1 << j for j in range(y_dim) if nc_slot_host[j] < 0)),
                     packed=pack_dense_for_fused_deep(run, nc_slot_host, y_dim, n_bins))
    return cache


def fused_spline_coupling(transformer, x, y, nc_slot_host, inverse, oob_counter, want_bin_idx=False, acc=None):
    """Try the one-launch coupling layer (bgk_coupling_rqs_dense).  Returns (y', dlogp[, bin_idx]) or
    None when the conditioner is not a fusable DenseNet (the caller then runs conditioner +
    bgk_rqs_transform)."""
    if x.dim() != 2 or y.dim() != 2 or not x.is_cuda:
        return None
    plan = _fused_plan(transformer, y.shape[-1], nc_slot_host)
    if plan is None or x.shape[-1] != plan["d_c"]:
        return None
    parts = _cond_parts(x)
    if len(parts) > 1 and (plan["mode"] == "f32" or plan["n_bins"] != 8 or plan["hidden"] != 128 or plan.get("deep")):
        parts = [x.cat()]                 # several conditioning tensors: the second-generation kernel only
    _lib.require_hip(y, *parts)
    W0p, W1p, W2p = plan["packed"][:3]
    if W0p.device != y.device:
        return None
    y2, ldy = _lib.rowmajor(y)
    B, d = y2.shape
    out = torch.empty((B, d), dtype=torch.float32, device=y.device)
    if acc is not None:
        dlogp, accumulate = acc.peek()
    else:
        dlogp, accumulate = torch.empty((B,), dtype=torch.float32, device=y.device), False
    bins = torch.empty((B, d), dtype=torch.int32, device=y.device) if want_bin_idx else None
    s = transformer._default_settings
    tail = (plan["hidden"], plan["hidden"], plan["act"], _lib.ptr(y2), ldy, B, d, plan["n_bins"], plan["circ_mask"], int(inverse),
            transformer._left, transformer._right, transformer._bottom, transformer._top,
            s["min_bin_width"], s["min_bin_height"], s["min_derivative"], int(s.get("enable_identity_init", False)),
            _lib.ptr(out), d, _lib.ptr(dlogp), int(bool(accumulate)), _lib.ptr(bins), _lib.ptr(oob_counter), _lib.stream_ptr(y.device))
    with torch.cuda.device(y.device):
        if plan.get("deep"):
            A0, A1, A2, c0, c1s, c2 = plan["packed"]
            x2, ldc = _lib.rowmajor(parts[0])
            c1_arr = (ctypes.c_float * max(1, len(c1s)))(*c1s)
            st = _lib.lib().bgk_coupling_rqs_dense_deep(
                _lib.ptr(x2), ldc, plan["d_c"], int(plan["periodic"]), _lib.ptr(A0), _lib.ptr(A1), _lib.ptr(A2), c0, c1_arr, c2,
                plan["deep"], *tail[2:])
        elif plan["mode"] == "f32":
            x2, ldc = _lib.rowmajor(parts[0])
            st = _lib.lib().bgk_coupling_rqs_dense(
                _lib.ptr(x2), ldc, plan["d_c"], int(plan["periodic"]), _lib.ptr(W0p), _lib.ptr(W1p), _lib.ptr(W2p), *tail)
        else:
            c0, c1, c2 = plan["packed"][3]
            ops = (_lib.ptr(W0p), _lib.ptr(W1p), _lib.ptr(W2p), c0, c1, c2, _lib.ptr(plan.get("cs")), int(plan["mode"] == "bf16"))
            st = -2
            if len(parts) > 1:
                ptrs, lds, widths, n, keep = _lib.cond_segments(parts)
                st = _lib.lib().bgk_coupling_rqs_dense_h2_mc(ptrs, lds, widths, n, int(plan["periodic"]), *ops, *tail)
                if st == -2:
                    parts = [x.cat()]
            if st == -2:
                x2, ldc = _lib.rowmajor(parts[0])
                st = _lib.lib().bgk_coupling_rqs_dense_h2(_lib.ptr(x2), ldc, plan["d_c"], int(plan["periodic"]), *ops, *tail)
    if st == -2:
        return _reject(transformer, "shape outside the fused spline kernels' envelope: " + _lib.lib().bgk_last_error().decode(errors="replace"))
    _lib.check(st, "bgk_coupling_rqs_dense")
    if acc is not None:
        acc.commit()
    res = (out, acc if acc is not None else dlogp[:, None])
    return res + (bins,) if want_bin_idx else res


# ---- training forward of the fused spline coupling layer -------------------------------------------------------
def _act_fwd_bwd(code):
    if code == 1:
        return torch.nn.functional.silu, lambda g, z, h: torch.ops.aten.silu_backward(g, z)
    if code == 2:
        return torch.relu, lambda g, z, h: g * (z > 0).to(g.dtype)
    return torch.tanh, lambda g, z, h: g * (1.0 - h * h)


def _featurise(x, periodic):
    if not periodic:
        return x
    ang = 2 * np.pi * x
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


HALF_PAD_ROWS = int(os.environ.get("BGK_HALF_PAD_ROWS", "0"))      # rows of padding between the [B, 128] halves of one allocation
PACKED_PARAMS = os.environ.get("BGK_PACKED_PARAMS", "1") != "0"   # the fused training forward saves the spline parameters element-major
# ... or not at all (round 5): bgk_coupling_rqs_dense_h2_backward recomputes them from z1 with the forward's operands (bit-identical)
RECOMPUTE_PARAMS = os.environ.get("BGK_RECOMPUTE_PARAMS", "1") != "0"

FUSED_MLP_BACKWARD = True    # input-gradient chain of the conditioner on bgk_dense_backward_dx (False: three GEMMs + torch act ops)


def absmax_of(*tensors):
    """[len(tensors)] device floats: max |t| of each 2-d f32 HIP tensor on bgk_absmax (None entries stay 0) -- the scale source of the
    backward GEMMs for gradient tensors that did not come out of this library's kernels"""
    dev = next(t for t in tensors if t is not None).device
    out = torch.zeros(len(tensors), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        for i, t in enumerate(tensors):
            if t is None or t.numel() == 0:
                continue
            t2, ld = _lib.rowmajor(t.detach())
            _lib.check(_lib.lib().bgk_absmax(_lib.ptr(t2), ld, t2.shape[0], t2.shape[1], _lib.ptr(out[i:]), _lib.stream_ptr(dev)), "bgk_absmax")
    return out


def _dense_backward_dx(g_p, z1, z0, x, W0, W1, W2, cs, act_code, periodic, want_gx, bufs, want_h=True, t_version=None, absmax=None,
                       gx_add=None, gx_out=None):
    """bgk_pack_dense_h2_t + bgk_dense_backward_dx: returns (g_z1, g_z0, h1, h0, g_x or None); ``want_h=False``: the activations
    are not written (h1 = h0 = None: the weight-gradient kernel recomputes them from z1 / z0).  ``t_version``: state key of the three
    weights; when ``bufs`` already holds the transposed operands of that state (repack_training_plans) the pack is skipped.
    ``absmax``: [3] device floats, [0] = max |g_p| on entry (bgk_rqs_backward's), [1] / [2] zero: raised to max |g_z1| / |g_z0|.
    ``gx_add``: a [B, d_c] tensor the kernel adds to the conditioner-input gradient (``gx_out``: where the sum goes; may be ``gx_add``
    itself -- accumulation in place -- default: a fresh tensor)."""
    dev = g_p.device
    B, P = g_p.shape
    n_in = W0.shape[1]
    _t_operand_bufs(bufs, P, n_in, dev)
    T0, T1, T2 = bufs["T0"], bufs["T1"], bufs["T2"]
    g2, ldg = _lib.rowmajor(g_p)
    x2, ldc = _lib.rowmajor(x.detach())
    d_c = x2.shape[1]
    # (the halves of ONE allocation: at B = 2^18 they would sit exactly 2^27 bytes apart -- row r of g_z1 and row r of g_z0, which a
    # wave writes back to back, on the same memory channel and bank.  HALF_PAD_ROWS rows between them break the power-of-two stride.)
    n_out = 4 if want_h else 2
    out = torch.empty((n_out, B + HALF_PAD_ROWS, 128), dtype=torch.float32, device=dev)[:, :B]
    g_x = (gx_out if gx_out is not None else torch.empty((B, d_c), dtype=torch.float32, device=dev)) if want_gx else None
    add2, lda = _lib.rowmajor(gx_add) if (gx_add is not None and want_gx) else (None, 0)
    ws = [w.detach().contiguous() for w in (W0, W1, W2)]
    with torch.cuda.device(dev):
        if t_version is None or bufs.get("t_version") != t_version:
            st = _lib.lib().bgk_pack_dense_h2_t(_lib.ptr(ws[0]), n_in, _lib.ptr(ws[1]), _lib.ptr(ws[2]), P, _lib.ptr(cs),
                                                _lib.ptr(T0), _lib.ptr(T1), _lib.ptr(T2), _lib.stream_ptr(dev))
            _lib.check(st, "bgk_pack_dense_h2_t")
            bufs["t_version"] = t_version
        st = _lib.lib().bgk_dense_backward_dx(_lib.ptr(g2), ldg, P, _lib.ptr(z1), _lib.ptr(z0), _lib.ptr(x2), ldc, d_c, int(periodic),
                                              _lib.ptr(T0), _lib.ptr(T1), _lib.ptr(T2), _lib.ptr(cs), act_code, B,
                                              _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]) if want_h else None,
                                              _lib.ptr(out[3]) if want_h else None, _lib.ptr(g_x), _lib.rowmajor(g_x)[1] if g_x is not None else d_c,
                                              _lib.ptr(add2), lda, _lib.ptr(absmax), _lib.ptr(absmax[1:]) if absmax is not None else None, _lib.stream_ptr(dev))
        _lib.check(st, "bgk_dense_backward_dx")
    return out[0], out[1], (out[2] if want_h else None), (out[3] if want_h else None), g_x


FUSED_WEIGHT_GRAD = True     # weight / bias gradients on bgk_dense_weight_grad (False: split-K bmm + bgk_column_sum)

_DIRECT_GRADS = [False]


DEFERRED_WGRAD_REDUCE = True    # inside direct_grad_accumulation(): one reduction launch for all layers when the context exits

_PENDING_REDUCE = {}            # workspace data_ptr -> (device, B, P, n_in, workspace, the six destinations, H1, H0)


def flush_weight_grad_reductions():
    """reduce the partial weight-gradient sums of every layer whose bgk_dense_weight_grad ran with the reduction deferred
    (bgk_dense_weight_grad_reduce_many: one launch per 16 layers), adding them to the flat gradient bucket"""
    if not _PENDING_REDUCE:
        return
    by_dev = {}
    for dev, *rest in _PENDING_REDUCE.values():
        by_dev.setdefault(dev, []).append(rest)
    _PENDING_REDUCE.clear()
    for dev, items in by_dev.items():
        n = len(items)
        vps, i32s = ctypes.c_void_p * n, ctypes.c_int32 * n
        dst = lambda k: vps(*[it[4][k].data_ptr() for it in items])       # noqa: E731
        with torch.cuda.device(dev):
            # (items: B, P, n_in, workspace, destinations, H1, H0 -- the hidden widths of the layer's weight shapes)
            st = _lib.lib().bgk_mlp_weight_grad_reduce_many(
                n, (ctypes.c_int64 * n)(*[it[0] for it in items]), i32s(*[it[1] for it in items]),
                i32s(*[it[5] for it in items]), i32s(*[it[6] for it in items]),
                i32s(*[it[2] for it in items]), vps(*[it[3].data_ptr() for it in items]),
                dst(4), dst(5), dst(2), dst(3), dst(0), dst(1), 1, _lib.stream_ptr(dev))
        _lib.check(st, "bgk_mlp_weight_grad_reduce_many")


class direct_grad_accumulation:
    """Context manager: inside it, the backward of the fused training layers ADDS its weight / bias gradients straight into the
    flat gradient bucket of a ``training.FlatAdam`` (and returns None to autograd for them) instead of handing them to
    AccumulateGrad.  Only the optimizer's own ``backward`` / ``KLTrainer.train`` switch it on: any other differentiation through
    the flow (``torch.autograd.grad``, ``Energy.force``, gradient penalties) gets ordinary gradients and leaves ``.grad`` alone."""

    def __enter__(self):
        self._prev = _DIRECT_GRADS[0]
        _DIRECT_GRADS[0] = True
        return self

    def __exit__(self, *exc):
        _DIRECT_GRADS[0] = self._prev
        if not self._prev:                      # the outermost context: the deferred reductions of this backward pass
            if exc[0] is None:
                flush_weight_grad_reductions()
            else:
                _PENDING_REDUCE.clear()
        return False


def _dense_weight_grad(g_p, g_z1, g_z0, h1, h0, x, periodic, n_in, need, bufs, params=None, h_act=0, absmax=None):
    """bgk_dense_weight_grad: (gW0, gb0, gW1, gb1, gW2, gb2) of one coupling layer's conditioner.
    ``params`` = (W0, b0, W1, b1, W2, b2): inside ``direct_grad_accumulation()`` (entered by FlatAdam.backward / KLTrainer only),
    when ALL of them carry a flat-bucket gradient destination (``_bgk_grad_dst``, set by training.FlatAdam), the kernel accumulates
    straight into the bucket and None is returned for every gradient -- no per-parameter AccumulateGrad add kernels (96 tiny
    launches per cfg-3 step).  ``h_act`` != 0: ``h1`` / ``h0`` are the saved
    pre-activations and the kernel applies activation ``h_act`` while loading them.  ``absmax``: [3] device floats {max |g_p|,
    max |g_z1|, max |g_z0|} (the kernels that wrote the gradients publish them; None: measured here with bgk_absmax)."""
    dev = g_p.device
    B, P = g_p.shape
    g2, ldg = _lib.rowmajor(g_p)
    x2, ldc = _lib.rowmajor(x)
    lib = _lib.lib()
    need_ws = int(lib.bgk_dense_weight_grad_workspace(B, P, n_in))
    ws = bufs.get("wgrad_ws")
    if ws is None or ws.numel() < need_ws or ws.device != dev:
        ws = bufs["wgrad_ws"] = torch.empty(need_ws, dtype=torch.float32, device=dev)
    direct = _DIRECT_GRADS[0] and params is not None and all(need[2:8]) and all(
        getattr(p, "_bgk_grad_dst", None) is not None and p.grad is not None and p.grad.data_ptr() == p._bgk_grad_dst.data_ptr()
        for p in params)      # only while p.grad IS the bucket view (a zero_grad(set_to_none=True) elsewhere switches this off)
    if direct:
        gW0, gb0, gW1, gb1, gW2, gb2 = (p._bgk_grad_dst for p in params)
    else:
        gW2 = torch.empty((P, 128), dtype=torch.float32, device=dev) if need[6] else None
        gb2 = torch.empty((P,), dtype=torch.float32, device=dev) if need[7] else None
        gW1 = torch.empty((128, 128), dtype=torch.float32, device=dev) if need[4] else None
        gb1 = torch.empty((128,), dtype=torch.float32, device=dev) if need[5] else None
        gW0 = torch.empty((128, n_in), dtype=torch.float32, device=dev) if need[2] else None
        gb0 = torch.empty((128,), dtype=torch.float32, device=dev) if need[3] else None

    def wbuf(w, b, shape):   # a bias gradient without its weight gradient: give the kernel a scratch weight buffer
        return w if (w is not None or b is None) else torch.empty(shape, dtype=torch.float32, device=dev)
    w2, w1, w0 = wbuf(gW2, gb2, (P, 128)), wbuf(gW1, gb1, (128, 128)), wbuf(gW0, gb0, (128, n_in))
    if absmax is None:
        absmax = absmax_of(g_p, g_z1, g_z0)
    mode = int(direct)
    if direct and DEFERRED_WGRAD_REDUCE:
        if ws.data_ptr() in _PENDING_REDUCE:     # the same layer twice in one backward pass: its first partial set goes out first
            flush_weight_grad_reductions()
        _PENDING_REDUCE[ws.data_ptr()] = (dev, B, P, n_in, ws, (gW0, gb0, gW1, gb1, gW2, gb2), 128, 128)
        mode = 2
    with torch.cuda.device(dev):
        st = lib.bgk_dense_weight_grad(_lib.ptr(g2), ldg, P, _lib.ptr(g_z1), _lib.ptr(g_z0), _lib.ptr(h1), _lib.ptr(h0), int(h_act),
                                       _lib.ptr(x2), ldc, x2.shape[1], int(periodic), B, _lib.ptr(ws), ws.numel(),
                                       _lib.ptr(w2), _lib.ptr(gb2), _lib.ptr(w1), _lib.ptr(gb1), _lib.ptr(w0), _lib.ptr(gb0),
                                       mode, _lib.ptr(absmax), _lib.stream_ptr(dev))
    _lib.check(st, "bgk_dense_weight_grad")
    if direct:
        return (None,) * 6
    return gW0, gb0, gW1, gb1, gW2, gb2


def _train_forward_launch(x, y, W2, plan, tcfg, inverse, oob, dlogp=None, accumulate=False):
    """bgk_coupling_rqs_dense_h2_train: (out, dlogp [B], z0, z1, params).  ``dlogp`` / ``accumulate``: the layer's log-det is written
    (or, accumulate, ADDED) into the caller's [B] buffer -- the running log|det J| of a chain of layers"""
    A0, A1, A2, (c0, c1, c2) = plan["packed"]
    x2, ldc = _lib.rowmajor(x)
    y2, ldy = _lib.rowmajor(y)
    B, d = y2.shape
    dev = y.device
    P = W2.shape[0]
    out = torch.empty((B, d), dtype=torch.float32, device=dev)
    if dlogp is None:
        dlogp, accumulate = torch.empty((B,), dtype=torch.float32, device=dev), False
    zz = torch.empty((2, B + HALF_PAD_ROWS, 128), dtype=torch.float32, device=dev)      # (see _dense_backward_dx on the padding)
    z0, z1 = zz[0, :B], zz[1, :B]
    left, right, bottom, top, s = tcfg

    def launch(layout):
        # layout 1: the parameters element-major [B, d (3 K + 1)] -- the kernel's own order, written as 16-byte pieces of contiguous
        # runs (PACKED_PARAMS; bgk_rqs_backward reads that layout); layout 0: the reference's column order [B, P]; layout 2: not
        # written at all (RECOMPUTE_PARAMS; the backward redoes the output layer from z1)
        width = d * (3 * plan["n_bins"] + 1) if layout else P
        ldp = 0 if layout == 2 else (param_pitch(width + 3) if layout else param_pitch(P))
        params = None if layout == 2 else torch.empty((B, ldp), dtype=torch.float32, device=dev)[:, :width]
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_coupling_rqs_dense_h2_train(
                _lib.ptr(x2), ldc, plan["d_c"], int(plan["periodic"]), _lib.ptr(A0), _lib.ptr(A1), _lib.ptr(A2), c0, c1, c2,
                _lib.ptr(plan.get("cs")), 128, 128, plan["act"], _lib.ptr(y2), ldy, B, d, plan["n_bins"], plan["circ_mask"], int(inverse),
                left, right, bottom, top, s["min_bin_width"], s["min_bin_height"], s["min_derivative"],
                int(s.get("enable_identity_init", False)), _lib.ptr(out), d, _lib.ptr(dlogp), int(bool(accumulate)), _lib.ptr(oob),
                _lib.ptr(z0), _lib.ptr(z1), _lib.ptr(params), ldp, _lib.ptr(plan["src_col_dev"]), layout, _lib.stream_ptr(dev))
        return st, params

    st, params = launch(2) if (RECOMPUTE_PARAMS and plan["n_bins"] == 8) else (-2, None)
    plan["params_recompute"] = st == 0
    if st == -2:
        st, params = launch(1) if (PACKED_PARAMS and plan["n_bins"] == 8) else (-2, None)
    plan["params_packed"] = st == 0 and params is not None   # (the layout of the tensor just written: _LayerCtx hands it to the backward)
    if st == -2:                      # (BGK_EUNSUPPORTED: the first-generation kernel runs this layer)
        st, params = launch(0)
    _lib.check(st, "bgk_coupling_rqs_dense_h2_train")
    return out, dlogp, z0, z1, params


class _LayerCtx:
    """what the backward of one fused training layer needs besides its saved tensors"""
    __slots__ = ("params", "cs", "tbufs", "t_version", "act", "periodic", "rcfg", "packed", "recompute", "plan", "pack_version")

    def __init__(self, params, plan, tcfg, inverse, t_version):
        left, right, bottom, top, s = tcfg
        self.params = params                # the nn.Parameters themselves (flat-bucket gradient destinations hang on them)
        self.cs = plan.get("cs")            # scale table of the operands packed for this forward (same weights in backward)
        self.tbufs = plan.setdefault("tbufs", {})
        # state of the three weight PARAMETERS this forward ran on: the key of the transposed operands the backward packs / reuses.
        # (W0..W2 themselves may be temporaries -- the zero-padded views of a narrow conditioner -- whose (data_ptr, version) repeats
        # from step to step once the caching allocator recycles their storage: never key a cache on them.)
        self.t_version = t_version
        self.act, self.periodic = plan["act"], bool(plan["periodic"])
        self.rcfg = (plan["n_bins"], inverse, left, right, bottom, top, dict(s))
        self.packed = bool(plan.get("params_packed"))        # layout of the parameters the forward launch (just before this) saved
        # ... or (A2 operand, c2) of that launch when it saved none: the backward recomputes them.  The device packer REWRITES a plan's
        # operand buffers in place when the weights change (pack_dense_for_fused_h2_device(bufs = ...), repack_training_plans): a
        # backward that runs after such a re-pack -- a retained graph, update-then-second-forward -- would recompute the parameters from the
        # NEW weights.  `plan` / `pack_version` let the backward notice and refuse (_rqs_backward_recompute).
        self.recompute = (plan["packed"][2], plan["packed"][3][2], plan["circ_mask"]) if plan.get("params_recompute") else None
        self.plan, self.pack_version = plan, plan.get("version")


def _train_backward_layer(lc, x, y, W0, W1, W2, z0, z1, params, nc_dev, g_out, g_dlogp, need_gx, need_w, gx_add=None, gx_out=None,
                          absmax=None):
    """Backward of one fused training layer: bgk_rqs_backward, the conditioner's input-gradient chain (bgk_dense_backward_dx; layers
    outside its envelope: library GEMMs) and the weight / bias gradients (bgk_dense_weight_grad).  ``need_w``: the six flags of
    (W0, b0, W1, b1, W2, b2).  ``gx_add`` / ``gx_out``: see _dense_backward_dx (fused chain only; otherwise added here).
    Returns (g_x or None, g_y, (gW0, gb0, gW1, gb1, gW2, gb2))."""
    from .transformer import rqs_backward
    act_code, periodic, rcfg, cs = lc.act, lc.periodic, lc.rcfg, lc.cs
    act, act_bwd = _act_fwd_bwd(act_code)
    need = [need_gx, True] + list(need_w)
    fused_wg = FUSED_WEIGHT_GRAD and y.is_cuda and W0.shape[1] <= 128
    recompute_h = False
    fused_dx = cs is not None and FUSED_MLP_BACKWARD and W0.shape[1] <= T_OPERAND_MAX_IN
    # largest magnitudes of g_params | g_z1 | g_z0, raised by the kernels that write them: the power-of-two scales under which the
    # backward GEMMs split these gradients into f16 hi + lo operand pairs (f32-class products whatever the loss scale)
    if absmax is None:
        absmax = torch.zeros(3, dtype=torch.float32, device=y.device)
    if lc.recompute is not None:
        g_y, g_p = _rqs_backward_recompute(lc, y, z1, W2.shape[0], nc_dev, g_out, g_dlogp, absmax)
    else:
        g_y, g_p = rqs_backward(y, params, nc_dev, rcfg, g_out, g_dlogp, absmax=absmax, packed_width=W2.shape[0] if lc.packed else None)
    if fused_dx:
        # with the fused weight-gradient kernel downstream the activations h1 / h0 are not materialised: it re-applies the
        # activation to the saved pre-activations while loading them (268 MB less written and read per layer at 2^18 samples)
        recompute_h = fused_wg
        g_z1, g_z0, h1, h0, g_x = _dense_backward_dx(g_p, z1, z0, x, W0, W1, W2, cs, act_code, periodic, need_gx, lc.tbufs,
                                                     want_h=not recompute_h, t_version=lc.t_version, absmax=absmax,
                                                     gx_add=gx_add, gx_out=gx_out)
    else:
        h1 = act(z1)
        g_z1 = act_bwd(_matmul_nn(g_p, W2), z1, h1)
        h0 = act(z0)
        g_z0 = act_bwd(_matmul_nn(g_z1, W1), z0, h0)
        g_x = None
        if need_gx and periodic:
            with torch.enable_grad():
                xx = x.detach().requires_grad_(True)
                feats = _featurise(xx, True)
            g_x = torch.autograd.grad(feats, xx, _matmul_nn(g_z0, W0))[0]
        elif need_gx:
            g_x = _matmul_nn(g_z0, W0)
        if g_x is not None and gx_add is not None:
            g_x = g_x + gx_add
        absmax = None            # g_z1 / g_z0 came out of library GEMMs: _dense_weight_grad measures them
    if fused_wg:
        if recompute_h:
            gws = _dense_weight_grad(g_p, g_z1, g_z0, z1, z0, x.detach(), periodic, W0.shape[1], need, lc.tbufs,
                                     params=lc.params, h_act=act_code, absmax=absmax)
        else:
            gws = _dense_weight_grad(g_p, g_z1, g_z0, h1, h0, x.detach(), periodic, W0.shape[1], need, lc.tbufs,
                                     params=lc.params, absmax=absmax)
    else:
        feats = _featurise(x.detach(), periodic)
        gws = (_gram_tn(g_z0, feats.contiguous()) if need[2] else None, column_sum(g_z0) if need[3] else None,
               _gram_tn(g_z1, h0) if need[4] else None, column_sum(g_z1) if need[5] else None,
               _gram_tn(g_p, h1) if need[6] else None, column_sum(g_p) if need[7] else None)
    return g_x, g_y, gws


def _rqs_backward_recompute(lc, y, z1, P, nc_dev, g_out, g_dlogp, absmax):
    """bgk_coupling_rqs_dense_h2_backward: (g_y [B, d], g_params [B, P]) of a layer whose forward saved no parameters"""
    from .transformer import row_pitch
    n_bins, inverse, left, right, bottom, top, s = lc.rcfg
    A2, c2, circ_mask = lc.recompute
    if lc.plan.get("version") != lc.pack_version:
        raise RuntimeError("fused spline coupling: the conditioner's parameters changed between the forward and this backward (the packed "
                           "operands the backward recomputes the spline parameters from were rewritten); run the backward before the "
                           "optimizer step, or call forward again")
    y2, ldy = _lib.rowmajor(y)
    B, d = y2.shape
    dev = y.device
    g_out2 = g_out.reshape(-1, d).contiguous()
    g_dl = g_dlogp.reshape(-1).contiguous()
    g_y = torch.empty((B, d), dtype=torch.float32, device=dev)
    ldgp = row_pitch(P)
    g_p = torch.empty((B, ldgp), dtype=torch.float32, device=dev)[:, :P]
    assert z1.is_contiguous() and z1.shape == (B, 128)
    with torch.cuda.device(dev):
        st = _lib.lib().bgk_coupling_rqs_dense_h2_backward(
            _lib.ptr(z1), _lib.ptr(A2), c2, _lib.ptr(lc.cs), 128, lc.act, _lib.ptr(y2), ldy, B, d, n_bins, P, circ_mask, int(inverse),
            left, right, bottom, top, s["min_bin_width"], s["min_bin_height"], s["min_derivative"],
            int(s.get("enable_identity_init", False)), _lib.ptr(g_out2), d, _lib.ptr(g_dl), _lib.ptr(g_y), d, _lib.ptr(g_p), ldgp,
            _lib.ptr(absmax), _lib.stream_ptr(dev))
    _lib.check(st, "bgk_coupling_rqs_dense_h2_backward")
    return g_y, g_p


class _FusedSplineTrainFn(torch.autograd.Function):
    """Forward = ONE launch of bgk_coupling_rqs_dense_h2_train (conditioner MLP on the f16 matrix cores + spline; the
    pre-activations z0, z1 and the spline parameters are written out for the backward pass).  Backward = _train_backward_layer."""

    @staticmethod
    def forward(ctx, x, y, W0, b0, W1, b1, W2, b2, plan, tcfg, nc_dev, inverse, oob, t_version=None):
        out, dlogp, z0, z1, params = _train_forward_launch(x, y, W2, plan, tcfg, inverse, oob)
        ctx.save_for_backward(x, y, W0, W1, W2, z0, z1, params, nc_dev)
        ctx.lc = _LayerCtx((W0, b0, W1, b1, W2, b2), plan, tcfg, inverse, t_version)
        return out, dlogp[:, None]

    @staticmethod
    def backward(ctx, g_out, g_dlogp):
        x, y, W0, W1, W2, z0, z1, params, nc_dev = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_x, g_y, gws = _train_backward_layer(ctx.lc, x, y, W0, W1, W2, z0, z1, params, nc_dev, g_out, g_dlogp, need[0], need[2:8])
        return (g_x, g_y if need[1] else None, *gws) + (None,) * 6


class _SplineChainTrainFn(torch.autograd.Function):
    """A run of fused spline coupling layers as ONE autograd node.  Forward: one bgk_coupling_rqs_dense_h2_train launch per layer, all of
    them adding their log-det to one [B] buffer.  Backward: the layers in reverse with the field gradients kept in per-slot buffers --
    the conditioner-input gradient of a layer is ADDED to its slot's buffer inside bgk_dense_backward_dx (g_cond_add), so the sums
    autograd forms with one elementwise launch per extra consumer of a field (a field conditions several layers and is transformed by
    others: 18 adds + 16 log-det adds per step of the 16-layer cfg-3 flow) do not exist; neither do the per-layer autograd nodes.

    apply(layers, n_slots, *slot tensors, *per layer (W0, b0, W1, b1, W2, b2)); ``layers[i]`` = (transformed slot, conditioning slot,
    (plan, tcfg, nc_dev, inverse, oob, t_version)).  Returns (*the final tensors of the slots some layer transformed -- in slot order --,
    dlogp [B, 1])."""

    @staticmethod
    def forward(ctx, layers, n_slots, *tensors):
        state = list(tensors[:n_slots])
        weights = tensors[n_slots:]
        saved, lcs, io = [], [], []
        dlogp = None
        for i, (ti, ci, (plan, tcfg, nc_dev, inverse, oob, t_version)) in enumerate(layers):
            W0, b0, W1, b1, W2, b2 = weights[6 * i:6 * i + 6]
            x, y = state[ci], state[ti]
            out, dlogp, z0, z1, params = _train_forward_launch(x, y, W2, plan, tcfg, inverse, oob, dlogp=dlogp, accumulate=i > 0)
            io.append((len(saved), len(saved) + 1))
            saved += [x, y, W0, W1, W2, z0, z1, params, nc_dev]
            lcs.append(_LayerCtx((W0, b0, W1, b1, W2, b2), plan, tcfg, inverse, t_version))
            state[ti] = out
        ctx.save_for_backward(*saved)
        ctx.lcs, ctx.n_slots = lcs, n_slots
        ctx.route = [(ti, ci) for ti, ci, _ in layers]
        ctx.outs = sorted({ti for ti, _, _ in layers})
        # a conditioning slot needs its gradient where the flow's input there does, or where an earlier layer of the chain wrote it
        touched, want = set(), []
        for ti, ci, _ in layers:
            want.append(bool(ctx.needs_input_grad[2 + ci]) or ci in touched)
            touched.add(ti)
        ctx.want_gx = want
        return (*[state[s] for s in ctx.outs], dlogp[:, None])

    @staticmethod
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        *g_outs, g_dlogp = grads
        n_slots, L = ctx.n_slots, len(ctx.lcs)
        G = [None] * n_slots                 # gradient w.r.t. the CURRENT state of every slot, walking the layers backwards
        owned = [False] * n_slots            # buffers of this backward (may be added to in place); autograd's own are never written
        for s, g in zip(ctx.outs, g_outs):
            G[s] = g
        need = ctx.needs_input_grad
        dev = g_dlogp.device
        absmax = torch.zeros((L, 3), dtype=torch.float32, device=dev)       # one fill for the whole chain
        w_grads = [None] * (6 * L)
        for i in range(L - 1, -1, -1):
            ti, ci = ctx.route[i]
            x, y, W0, W1, W2, z0, z1, params, nc_dev = saved[9 * i:9 * i + 9]
            g_out = G[ti]
            if g_out is None:                # the slot's final tensor took no part in the loss
                g_out = torch.zeros_like(y)
            prev = G[ci]
            g_x, g_y, gws = _train_backward_layer(
                ctx.lcs[i], x, y, W0, W1, W2, z0, z1, params, nc_dev, g_out, g_dlogp, ctx.want_gx[i], need[2 + n_slots + 6 * i:2 + n_slots + 6 * i + 6],
                gx_add=prev, gx_out=prev if (prev is not None and owned[ci]) else None, absmax=absmax[i])
            G[ti], owned[ti] = g_y, True
            if g_x is not None:
                G[ci], owned[ci] = g_x, True
            w_grads[6 * i:6 * i + 6] = gws
        slot_grads = [G[s] if need[2 + s] else None for s in range(n_slots)]
        return (None, None, *slot_grads, *w_grads)


def spline_chain_train(blocks, xs, inverse):
    """A run of CouplingFlow blocks (execution order) whose spline transformers all take the fused training path, as ONE autograd node
    (_SplineChainTrainFn).  Returns the new state tuple and dlogp [B, 1], or None when a block is outside the envelope (the caller
    then runs the blocks one by one): one transformed and one conditioning tensor per block, 2-d f32 HIP tensors, fusable
    conditioners in split-f16 mode."""
    xs = list(xs)
    used = sorted({int(b.transformed_indices[0]) for b in blocks} | {int(b.cond_indices[0]) for b in blocks})
    if any(len(b.transformed_indices) != 1 or len(b.cond_indices) != 1 for b in blocks) or used[-1] >= len(xs):
        return None
    if not all(torch.is_tensor(xs[s]) and xs[s].is_cuda and xs[s].dtype == torch.float32 and xs[s].dim() == 2 for s in used):
        return None
    pos = {s: k for k, s in enumerate(used)}
    layers, weights = [], []
    widths = {s: xs[s].shape[-1] for s in used}
    B = xs[used[0]].shape[0]
    if B == 0 or any(xs[s].shape[0] != B for s in used):
        return None
    dev = xs[used[0]].device
    for b in blocks:
        tr = b.transformer
        ti, ci = int(b.transformed_indices[0]), int(b.cond_indices[0])
        if not getattr(tr, "allow_fused", False) or getattr(tr, "return_bin_indices", False):
            return None
        nc_dev, nc_host = tr._nc_slot(widths[ti], dev)
        # shapes only: the prep reads nothing of the tensors' values (layer i's inputs do not exist yet)
        probe_x = torch.empty((0, widths[ci]), dtype=torch.float32, device=dev)
        probe_y = torch.empty((0, widths[ti]), dtype=torch.float32, device=dev)
        prep = _spline_train_prep(tr, probe_x, probe_y, nc_dev, nc_host, inverse, tr._oob_counter(dev))
        if prep is None:
            return None
        layers.append((pos[ti], pos[ci], prep[1]))
        weights += list(prep[0])
    res = _SplineChainTrainFn.apply(layers, len(used), *[xs[s] for s in used], *weights)
    outs = sorted({ti for ti, _, _ in layers})
    for k, o in zip(outs, res[:-1]):
        xs[used[k]] = o
    return tuple(xs), res[-1]


def _spline_train_prep(transformer, x, y, nc_dev, nc_host, inverse, oob_counter):
    """What a differentiable one-launch forward of the spline coupling layer needs (split-f16 mode only): (W0, b0, W1, b1, W2, b2 --
    the parameters, or their zero-padded differentiable views for a narrow conditioner --, plan, tcfg, nc_dev, inverse, oob, t_version),
    or None when the conditioner is not a fusable DenseNet."""
    if x.dim() != 2 or y.dim() != 2 or not x.is_cuda or x.dtype != torch.float32 or _gemm_mode(transformer) != "f16x2":
        return None
    net = transformer._params_net
    spec = _fusable_dense(net.net if type(net) is WrapPeriodic else net)
    if spec is None or max(spec[0][0].out_features, spec[0][1].out_features) > 128:
        return None                 # the training kernels take two hidden layers of up to 128 units; wider / deeper conditioners are fused in
                                    # inference only (and no operands are packed for them per training step)
    plan = _fused_plan(transformer, y.shape[-1], nc_host)
    if plan is None or plan["mode"] != "f16x2" or x.shape[-1] != plan["d_c"] or plan["packed"][0].device != y.device \
            or plan["hidden"] != 128 or plan.get("deep") or plan["n_bins"] not in (4, 8, 12, 16, 32):   # the training variant (saved pre-activations + parameters): K = 8 on the
        return None                                        # second-generation kernel, the others on the first-generation one
    _lib.require_hip(x, y)
    if "src_col_dev" not in plan or plan["src_col_dev"].device != y.device:
        plan["src_col_dev"] = _src_col_table(y.shape[-1], plan["n_bins"], nc_host, y.device)
    plan["train_used"] = True
    _TRAIN_PLANS.add(transformer)
    net = transformer._params_net
    inner = net.net if type(net) is WrapPeriodic else net
    (l0, l1, l2), _ = _fusable_dense(inner)
    tcfg = (transformer._left, transformer._right, transformer._bottom, transformer._top, transformer._default_settings)
    W0, b0, W1, b1, W2, b2 = l0.weight, l0.bias, l1.weight, l1.bias, l2.weight, l2.bias
    t_version = tuple(param_state_key(p) for p in (W0, W1, W2))      # of the parameters, before any padding
    if plan.get("padded"):
        # hidden layers narrower than the kernels' 128 units: the operands packed for the forward are the zero-padded layers
        # (_pad_hidden); the backward kernels get the same padded matrices as differentiable views of the parameters
        # (F.pad: autograd slices the gradients back; padded units hold act(0) = 0 and feed zero columns, so their gradients are 0)
        pad = torch.nn.functional.pad
        h0, h1 = 128 - W0.shape[0], 128 - W1.shape[0]
        W0, b0 = pad(W0, (0, 0, 0, h0)), pad(b0, (0, h0))
        W1, b1 = pad(W1, (0, h0, 0, h1)), pad(b1, (0, h1))
        W2 = pad(W2, (0, h1))
    return (W0, b0, W1, b1, W2, b2), (plan, tcfg, nc_dev, inverse, oob_counter, t_version)


def fused_spline_coupling_train(transformer, x, y, nc_dev, nc_host, inverse, oob_counter):
    """Differentiable one-launch forward of the spline coupling layer (split-f16 mode only).  Returns (y', dlogp) or None
    when the conditioner is not a fusable DenseNet."""
    prep = _spline_train_prep(transformer, x, y, nc_dev, nc_host, inverse, oob_counter)
    if prep is None:
        return None
    return _FusedSplineTrainFn.apply(x, y, *prep[0], *prep[1])


# ---- training path of the fused AFFINE coupling layer (round 6) -----------------------------------------------------------------
# Forward: ONE launch of bgk_coupling_affine_dense_h2_train (both conditioner networks on the matrix cores + the affine tail; it also
# writes the scaled pre-activations of the four hidden layers and the two networks' outputs).  Backward: bgk_affine_backward (tail),
# then per network bgk_dense_backward_dx (input-gradient chain; the second network ADDS its conditioner-input gradient to the
# first's) and bgk_mlp_weight_grad (weight / bias gradients in the parameters' own shapes, straight into the FlatAdam bucket inside
# direct_grad_accumulation()).  Before round 6 an affine coupling under autograd ran its networks layer by layer and their backward
# through _LinearFn (F.linear / bmm -> hipBLASLt) + aten activation kernels.  Reference: nn/flow/transformer/affine.py:35-70,
# nn/dense.py:30-48, nn/flow/coupling.py:152-182, nn/training/trainers.py:156-163.
AFFINE_TRAIN = os.environ.get("BGK_AFFINE_TRAIN", "1") != "0"     # 0: the layer-by-layer path (A/B measurements)

_AFF_TRAIN_TRANSFORMERS = weakref.WeakSet()     # affine transformers whose training plan ran a forward since the last repack


def _affine_train_net(net):
    """((l0, l1, l2), act, periodic) of a shift / scale network inside the training kernels' envelope, else None"""
    per = type(net) is WrapPeriodic
    inner = net.net if per else net
    spec = _fusable_dense(inner)
    if spec is None:
        return None
    (l0, l1, l2), act = spec
    if per:
        n_raw = l0.in_features // 2
        if not (net.left == 0.0 and net.right == 1.0) or n_raw == 0 or 2 * n_raw != l0.in_features:
            return None
        if not np.array_equal(np.asarray(np.arange(n_raw)[net.indices]), np.arange(n_raw)):
            return None
    if l0.out_features > 128 or l1.out_features > 128 or l1.in_features != l0.out_features or l2.in_features != l1.out_features:
        return None
    return (l0, l1, l2), act, per


def _affine_train_plan(transformer, y_dim, dev):
    """Operand buffers of the affine training kernels for a transformer whose networks are two-hidden-layer DenseNets of <= 128 units
    (optionally behind an all-periodic WrapPeriodic): per network the forward operands A0..A2 + device scale table cs (bgk_pack_mlp_h2)
    and the transposed operands T0..T2 (bgk_pack_mlp_h2_t), re-packed when a parameter's state changed.  None outside the envelope."""
    nets = (transformer._shift_transformation, transformer._scale_transformation)
    if all(n is None for n in nets) or _gemm_mode(transformer) == "f32":
        return None
    specs = [None if n is None else _affine_train_net(n) for n in nets]
    if any(n is not None and sp is None for n, sp in zip(nets, specs)):
        return None
    live = [sp for sp in specs if sp is not None]
    n_in, periodic = live[0][0][0].in_features, live[0][2]
    acts = [sp[1] for sp in live]
    if not (acts[0] == acts[-1] or (len(live) == 2 and acts == [2, 3])):       # (shift, scale): both equal, or ReLU / Tanh
        return None
    for lins, _, per in live:
        if lins[0].in_features != n_in or per != periodic or lins[2].out_features != y_dim:
            return None
        if not all(p.is_cuda and p.device == dev and p.dtype == torch.float32 and p.is_contiguous() for lin in lins for p in (lin.weight, lin.bias)):
            return None
    if y_dim > 96 or n_in > T_OPERAND_MAX_IN or (periodic and n_in % 2):
        return None
    cache = transformer.__dict__.setdefault("_train_cache", {})
    key = (y_dim, n_in, bool(periodic), str(dev), FUSED_FWD64, FUSED_BWD64, TAIL_FUSED64, RECOMPUTE64,
           tuple(None if sp is None else (sp[0][0].out_features, sp[0][1].out_features, sp[1]) for sp in specs))
    if cache.get("key") != key:
        cache.clear()
        OT, S0 = (y_dim + 31) // 32, (n_in + 1 + 15) // 16
        # networks of <= 64 hidden units, <= 32 dims / non-periodic inputs: the kernels sized for them (bgk_coupling_affine_dense_fwd64_train,
        # bgk_affine_net_backward64) on operands packed for 64 hidden rows (HT = 2)
        small = (FUSED_FWD64 and y_dim <= 32 and n_in <= 32 and not periodic
                 and all(max(sp[0][0].out_features, sp[0][1].out_features) <= 64 for sp in specs if sp is not None))
        HT = 2 if small else 4
        entries = []
        for sp in specs:
            if sp is None:
                entries.append(None)
                continue
            f16 = lambda nblk: torch.empty((nblk, 64, 8), dtype=torch.float16, device=dev)       # noqa: E731
            tb = {}
            _t_operand_bufs(tb, y_dim, n_in, dev)
            entries.append(dict(lins=sp[0], act=sp[1], H0=sp[0][0].out_features, H1=sp[0][1].out_features,
                                A0=f16(S0 * HT * 2), A1=f16(2 * HT * HT * 2 + HT), A2=f16(2 * HT * OT * 2 + OT),
                                cs=torch.empty(6, dtype=torch.float32, device=dev),
                                tbufs=tb, version=None))
        # row pitch of the saved pre-activations / their gradients: [B, 64] when no hidden layer has more than 64 units (half the bytes)
        ldz = 64 if all(max(e["H0"], e["H1"]) <= 64 for e in entries if e is not None) else 128
        # both networks inside the small kernels' envelope: the backward of a forward-direction layer without volume preservation is ONE
        # library call (bgk_affine_coupling_backward64: the tail's backward inside the scale network's launch); with RECOMPUTE64 the forward
        # saves nothing and that call recomputes the networks from the layer's inputs
        tail_fused = bool(small and FUSED_BWD64 and TAIL_FUSED64 and all(e is not None for e in entries))
        cache.update(key=key, nets=entries, d_c=n_in // 2 if periodic else n_in, periodic=bool(periodic), OT=OT, y_dim=y_dim, ldz=ldz, HT=HT,
                     tail_fused=tail_fused, recompute=bool(tail_fused and RECOMPUTE64))
    lib = _lib.lib()
    for e in cache["nets"]:
        if e is None:
            continue
        l0, l1, l2 = e["lins"]
        version = tuple(param_state_key(p) for lin in e["lins"] for p in (lin.weight, lin.bias))
        if e["version"] == version:
            continue
        ws = [t.detach() for lin in e["lins"] for t in (lin.weight, lin.bias)]
        with torch.cuda.device(dev):
            st = lib.bgk_pack_mlp_h2(_lib.ptr(ws[0]), _lib.ptr(ws[1]), n_in, e["H0"], _lib.ptr(ws[2]), _lib.ptr(ws[3]), e["H1"],
                                     _lib.ptr(ws[4]), _lib.ptr(ws[5]), y_dim, None, 1, cache["OT"], cache["HT"],
                                     _lib.ptr(e["A0"]), _lib.ptr(e["A1"]), _lib.ptr(e["A2"]), _lib.ptr(e["cs"]), _lib.stream_ptr(dev))
            _lib.check(st, "bgk_pack_mlp_h2")
            tb = e["tbufs"]
            st = lib.bgk_pack_mlp_h2_t(_lib.ptr(ws[0]), n_in, e["H0"], _lib.ptr(ws[2]), e["H1"], _lib.ptr(ws[4]), y_dim, _lib.ptr(e["cs"]),
                                       _lib.ptr(tb["T0"]), _lib.ptr(tb["T1"]), _lib.ptr(tb["T2"]), _lib.stream_ptr(dev))
            _lib.check(st, "bgk_pack_mlp_h2_t")
        e["version"] = version
    return cache


def repack_affine_training_plans(param_ids=None):
    """After an optimizer step: the forward and the transposed operands of every affine coupling that ran a training forward since the
    last call, in three launches per 16 networks (bgk_pack_mlp_h2_many + bgk_pack_mlp_h2_t_many).  Returns the number of networks."""
    if not BATCHED_REPACK:
        return 0
    by_dev = {}
    for tr in list(_AFF_TRAIN_TRANSFORMERS):
        cache = getattr(tr, "_train_cache", None)
        if not cache or not cache.pop("train_used", False):
            continue
        for e in cache["nets"]:
            if e is None:
                continue
            params = [p for lin in e["lins"] for p in (lin.weight, lin.bias)]
            if param_ids is not None and not all(id(p) in param_ids for p in params):
                continue
            if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
                continue
            version = tuple(param_state_key(p) for p in params)
            if e["version"] == version:
                continue
            by_dev.setdefault(params[0].device, []).append((cache, e, params, version))
    done = 0
    for dev, items in by_dev.items():
        n = len(items)
        vps, i32s = (ctypes.c_void_p * n), (ctypes.c_int32 * n)
        col = lambda f: vps(*[f(it) for it in items])       # noqa: E731
        num = lambda f: i32s(*[f(it) for it in items])      # noqa: E731
        n_in = num(lambda it: it[1]["lins"][0].in_features)
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_pack_mlp_h2_many(
                n, col(lambda it: it[2][0].data_ptr()), col(lambda it: it[2][1].data_ptr()), n_in, num(lambda it: it[1]["H0"]),
                col(lambda it: it[2][2].data_ptr()), col(lambda it: it[2][3].data_ptr()), num(lambda it: it[1]["H1"]),
                col(lambda it: it[2][4].data_ptr()), col(lambda it: it[2][5].data_ptr()), num(lambda it: it[0]["y_dim"]),
                vps(*[None] * n), num(lambda it: 1), num(lambda it: it[0]["OT"]), num(lambda it: it[0]["HT"]),
                col(lambda it: it[1]["A0"].data_ptr()), col(lambda it: it[1]["A1"].data_ptr()), col(lambda it: it[1]["A2"].data_ptr()),
                col(lambda it: it[1]["cs"].data_ptr()), _lib.stream_ptr(dev))
            _lib.check(st, "bgk_pack_mlp_h2_many")
            st = _lib.lib().bgk_pack_mlp_h2_t_many(
                n, col(lambda it: it[2][0].data_ptr()), n_in, num(lambda it: it[1]["H0"]), col(lambda it: it[2][2].data_ptr()),
                num(lambda it: it[1]["H1"]), col(lambda it: it[2][4].data_ptr()), num(lambda it: it[0]["y_dim"]),
                col(lambda it: it[1]["cs"].data_ptr()), col(lambda it: it[1]["tbufs"]["T0"].data_ptr()),
                col(lambda it: it[1]["tbufs"]["T1"].data_ptr()), col(lambda it: it[1]["tbufs"]["T2"].data_ptr()), _lib.stream_ptr(dev))
            _lib.check(st, "bgk_pack_mlp_h2_t_many")
        for _cache, e, _params, version in items:
            e["version"] = version
        done += n
    return done


def _affine_net_backward(e, plan, g_net, ldg, z1, z0, x2, ldc, absmax, want_gx, gx_buf, gx_add, need_w, version):
    """backward of ONE conditioner network of a fused affine training layer: bgk_dense_backward_dx (g_net [B, d] -> g_z1, g_z0 and the
    conditioner-input gradient, to which ``gx_add`` is added) + bgk_mlp_weight_grad.  Returns the six gradients (W0, b0, W1, b1, W2, b2)
    -- None each when they went straight into the FlatAdam bucket or are not needed."""
    if e["version"] != version:
        raise RuntimeError("fused affine coupling: the conditioner's parameters changed between the forward and this backward (the packed "
                           "operands were rewritten); run the backward before the optimizer step, or call forward again")
    dev = g_net.device
    B, d = g_net.shape[0], plan["y_dim"]
    tb = e["tbufs"]
    d_c, periodic, n_in = plan["d_c"], plan["periodic"], e["lins"][0].in_features
    ldz = plan["ldz"]
    if FUSED_BWD64 and ldz == 64 and d <= 32 and n_in <= 32 and not periodic and (all(need_w) or not any(need_w)):
        return _affine_net_backward64(e, plan, g_net, ldg, z1, z0, x2, ldc, absmax, want_gx, gx_buf, gx_add, need_w)
    gz = torch.empty((2, B + HALF_PAD_ROWS, ldz), dtype=torch.float32, device=dev)[:, :B]
    lib = _lib.lib()
    add2, lda = (gx_add, gx_add.stride(0)) if (gx_add is not None and want_gx) else (None, 0)
    with torch.cuda.device(dev):
        st = lib.bgk_mlp_backward_dx(_lib.ptr(g_net), ldg, d, _lib.ptr(z1), _lib.ptr(z0), ldz, _lib.ptr(x2), ldc, d_c, int(periodic),
                                       _lib.ptr(tb["T0"]), _lib.ptr(tb["T1"]), _lib.ptr(tb["T2"]), _lib.ptr(e["cs"]), e["act"], B,
                                       _lib.ptr(gz[0]), _lib.ptr(gz[1]), None, None, _lib.ptr(gx_buf) if want_gx else None,
                                       gx_buf.stride(0) if want_gx else d_c, _lib.ptr(add2), lda, _lib.ptr(absmax), _lib.ptr(absmax[1:]),
                                       _lib.stream_ptr(dev))
        _lib.check(st, "bgk_mlp_backward_dx")
    if not any(need_w):
        return (None,) * 6
    l0, l1, l2 = e["lins"]
    params = (l0.weight, l0.bias, l1.weight, l1.bias, l2.weight, l2.bias)
    H0, H1 = e["H0"], e["H1"]
    need_ws = int(lib.bgk_mlp_weight_grad_workspace(B, d, H1, H0, n_in))
    ws = tb.get("wgrad_ws")
    if ws is None or ws.numel() < need_ws or ws.device != dev:
        ws = tb["wgrad_ws"] = torch.empty(need_ws, dtype=torch.float32, device=dev)
    direct = _DIRECT_GRADS[0] and all(need_w) and all(
        getattr(p, "_bgk_grad_dst", None) is not None and p.grad is not None and p.grad.data_ptr() == p._bgk_grad_dst.data_ptr()
        for p in params)
    if direct:
        gW0, gb0, gW1, gb1, gW2, gb2 = (p._bgk_grad_dst for p in params)
    else:
        new = lambda shape, need: torch.empty(shape, dtype=torch.float32, device=dev) if need else None      # noqa: E731
        gW0, gb0 = new((H0, n_in), need_w[0] or need_w[1]), new((H0,), need_w[1])
        gW1, gb1 = new((H1, H0), need_w[2] or need_w[3]), new((H1,), need_w[3])
        gW2, gb2 = new((d, H1), need_w[4] or need_w[5]), new((d,), need_w[5])
    mode = int(direct)
    if direct and DEFERRED_WGRAD_REDUCE:
        if ws.data_ptr() in _PENDING_REDUCE:
            flush_weight_grad_reductions()
        _PENDING_REDUCE[ws.data_ptr()] = (dev, B, d, n_in, ws, (gW0, gb0, gW1, gb1, gW2, gb2), H1, H0)
        mode = 2
    with torch.cuda.device(dev):
        st = lib.bgk_mlp_weight_grad(_lib.ptr(g_net), ldg, d, _lib.ptr(gz[0]), _lib.ptr(gz[1]), _lib.ptr(z1), _lib.ptr(z0), ldz, H1, H0,
                                     e["act"], _lib.ptr(x2), ldc, d_c, int(periodic), B, _lib.ptr(ws), ws.numel(),
                                     _lib.ptr(gW2), _lib.ptr(gb2), _lib.ptr(gW1), _lib.ptr(gb1), _lib.ptr(gW0), _lib.ptr(gb0), mode,
                                     _lib.ptr(absmax), _lib.stream_ptr(dev))
    _lib.check(st, "bgk_mlp_weight_grad")
    if direct:
        return (None,) * 6
    return tuple(g if n else None for g, n in zip((gW0, gb0, gW1, gb1, gW2, gb2), need_w))


FUSED_FWD64 = os.environ.get("BGK_FUSED_FWD64", "1") != "0"      # ... and their training forward on the kernel sized for them
FUSED_BWD64 = os.environ.get("BGK_FUSED_BWD64", "1") != "0"      # networks of <= 64 hidden units: chain + weight gradients in one launch
TAIL_FUSED64 = os.environ.get("BGK_TAIL_FUSED64", "1") != "0"    # ... with the affine tail's backward inside the scale network's launch (one call per layer)
# ... and nothing saved by the forward, the backward recomputing both networks: 1.3 KB per sample and layer less memory, 7 % slower than
# the saved form at cfg 2's shapes (profiles/r06_ab_runs.txt) -- opt-in
RECOMPUTE64 = os.environ.get("BGK_RECOMPUTE64", "0") != "0"


def _affine_net_backward64(e, plan, g_net, ldg, z1, z0, x2, ldc, absmax, want_gx, gx_buf, gx_add, need_w):
    """bgk_affine_net_backward64: _affine_net_backward's work for a network of <= 64 hidden units in one launch (g_z1 / g_z0 stay on chip)"""
    dev = g_net.device
    B, d = g_net.shape[0], plan["y_dim"]
    tb = e["tbufs"]
    n_in, H0, H1 = e["lins"][0].in_features, e["H0"], e["H1"]
    lib = _lib.lib()
    need_ws = int(lib.bgk_affine_net_backward64_workspace(B, d, H1, H0, n_in))
    ws = tb.get("bwd64_ws")
    if ws is None or ws.numel() < need_ws or ws.device != dev:
        ws = tb["bwd64_ws"] = torch.empty(need_ws, dtype=torch.float32, device=dev)
    l0, l1, l2 = e["lins"]
    params = (l0.weight, l0.bias, l1.weight, l1.bias, l2.weight, l2.bias)
    want_w = all(need_w)
    direct = want_w and _DIRECT_GRADS[0] and all(
        getattr(p, "_bgk_grad_dst", None) is not None and p.grad is not None and p.grad.data_ptr() == p._bgk_grad_dst.data_ptr() for p in params)
    if direct:
        gW0, gb0, gW1, gb1, gW2, gb2 = (p._bgk_grad_dst for p in params)
    elif want_w:
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)      # noqa: E731
        gW0, gb0, gW1, gb1, gW2, gb2 = new(H0, n_in), new(H0), new(H1, H0), new(H1), new(d, H1), new(d)
    else:
        gW0 = gb0 = gW1 = gb1 = gW2 = gb2 = None
    add2, lda = (gx_add, gx_add.stride(0)) if (gx_add is not None and want_gx) else (None, 0)
    with torch.cuda.device(dev):
        st = lib.bgk_affine_net_backward64(
            _lib.ptr(g_net), ldg, d, _lib.ptr(z1), _lib.ptr(z0), _lib.ptr(x2), ldc, n_in, H1, H0,
            _lib.ptr(tb["T0"]), _lib.ptr(tb["T1"]), _lib.ptr(tb["T2"]), _lib.ptr(e["cs"]), e["act"], B,
            _lib.ptr(gx_buf) if want_gx else None, gx_buf.stride(0) if want_gx else n_in, _lib.ptr(add2), lda, _lib.ptr(absmax),
            _lib.ptr(ws), ws.numel(), _lib.ptr(gW2), _lib.ptr(gb2), _lib.ptr(gW1), _lib.ptr(gb1), _lib.ptr(gW0), _lib.ptr(gb0), int(direct),
            _lib.stream_ptr(dev))
    _lib.check(st, "bgk_affine_net_backward64")
    if direct or not want_w:
        return (None,) * 6
    return gW0, gb0, gW1, gb1, gW2, gb2


def _affine_train_forward(x, y, log_alpha, plan, cfg, dlogp=None, accumulate=False, out=None):
    """one training-forward launch of an affine coupling: (out [B, d], dlogp [B], saved = (x2, y2, zz, ms)).  ``dlogp`` / ``accumulate``:
    the layer's log-det written (or added) into the caller's [B] buffer -- the running log|det J| of a stack of layers; ``out``: where
    the transformed half goes ([B, d] rows of any stride: a column range of the stack's output tensor)"""
    pv, circ, inverse = cfg
    dev = y.device
    x2, ldc = _lib.rowmajor(x.detach())
    y2, ldy = _lib.rowmajor(y.detach())
    B, d = y2.shape
    if out is None:
        out = torch.empty((B, d), dtype=torch.float32, device=dev)
    ldo = out.stride(0) if B > 1 else d
    if dlogp is None:
        dlogp, accumulate = torch.empty((B,), dtype=torch.float32, device=dev), False
    ldms = 32 * plan["OT"]
    fused_bwd = plan.get("tail_fused") and not pv          # (either direction; the backward of an inverse-direction layer reads the OUTPUT)
    if fused_bwd and plan.get("recompute"):
        zz = ms = None                    # nothing saved: bgk_affine_coupling_backward64 recomputes the networks
        zp = [None] * 6
    elif fused_bwd:                       # ... or reads z0 / z1 of both networks and s_raw (ms of ONE array marks this form); mu is not needed
        zz = torch.empty((4, B + HALF_PAD_ROWS, plan["ldz"]), dtype=torch.float32, device=dev)[:, :B]
        ms = torch.empty((1, B, ldms), dtype=torch.float32, device=dev)
        zp = [_lib.ptr(zz[0]), _lib.ptr(zz[1]), _lib.ptr(zz[2]), _lib.ptr(zz[3]), None, _lib.ptr(ms[0])]
    else:
        zz = torch.empty((4, B + HALF_PAD_ROWS, plan["ldz"]), dtype=torch.float32, device=dev)[:, :B]
        ms = torch.empty((2, B, ldms), dtype=torch.float32, device=dev)
        zp = [_lib.ptr(zz[0]), _lib.ptr(zz[1]), _lib.ptr(zz[2]), _lib.ptr(zz[3]), _lib.ptr(ms[0]), _lib.ptr(ms[1])]
    ops = []
    for e in plan["nets"]:
        ops += [None, None, None, None, 0] if e is None else [_lib.ptr(e["A0"]), _lib.ptr(e["A1"]), _lib.ptr(e["A2"]), _lib.ptr(e["cs"]), e["act"]]
    with torch.cuda.device(dev):
        if plan["HT"] == 2:
            what = "bgk_coupling_affine_dense_fwd64_train"
            st = _lib.lib().bgk_coupling_affine_dense_fwd64_train(
                _lib.ptr(x2), ldc, x2.shape[1], *ops, _lib.ptr(log_alpha.detach()), int(pv), int(circ), int(inverse),
                _lib.ptr(y2), ldy, B, d, _lib.ptr(out), ldo, _lib.ptr(dlogp), int(bool(accumulate)),
                *zp, ldms, _lib.stream_ptr(dev))
        else:
            what = "bgk_coupling_affine_dense_h2_train"
            ptrs, lds, widths, n, _keep = _lib.cond_segments([x2])
            st = _lib.lib().bgk_coupling_affine_dense_h2_train(
                ptrs, lds, widths, n, int(plan["periodic"]), *ops, _lib.ptr(log_alpha.detach()), int(pv), int(circ), int(inverse),
                _lib.ptr(y2), ldy, B, d, _lib.ptr(out), ldo, _lib.ptr(dlogp), int(bool(accumulate)),
                *zp[:4], plan["ldz"], *zp[4:], ldms, _lib.stream_ptr(dev))
    _lib.check(st, what)
    return out, dlogp, (x2, out if (fused_bwd and inverse) else y2, zz, ms)


def _affine_train_backward(plan, cfg, versions, x2, y2, log_alpha, zz, ms, g_out, g_dl, need_x, need_la, need_w, gx_add=None, gx_out=None):
    """backward of one fused affine training layer: bgk_affine_backward (tail), then the networks (_affine_net_backward).  ``g_dl``: [B]
    contiguous; ``need_w``: the twelve flags of the networks' parameters; ``gx_add`` / ``gx_out``: a [B, d_c] tensor the conditioner-input
    gradient is added to and where the sum goes (may be the same tensor: accumulation in place; default: a fresh tensor).
    Returns (g_x or None, g_y, g_log_alpha or None, 12 weight gradients -- None where they went straight into the FlatAdam bucket)."""
    pv, circ, inverse = cfg
    es, et = plan["nets"]
    dev = y2.device
    B, d = y2.shape
    if zz is None or ms.shape[0] == 1:
        return _affine_train_backward_fused(plan, versions, x2, y2, zz, ms, log_alpha, g_out, g_dl, need_x, need_la, need_w, gx_add, gx_out,
                                            bool(inverse))
    ldms = ms.shape[2]
    ldc = x2.stride(0) if B > 1 else x2.shape[1]
    g_out2, ldgo = _lib.rowmajor(g_out.reshape(B, d))
    g_y = torch.empty((B, d), dtype=torch.float32, device=dev)
    g_ms = torch.empty((2, B, ldms), dtype=torch.float32, device=dev)
    absmax = torch.zeros((2, 3), dtype=torch.float32, device=dev)
    la_direct = (_DIRECT_GRADS[0] and need_la and getattr(log_alpha, "_bgk_grad_dst", None) is not None and log_alpha.grad is not None
                 and log_alpha.grad.data_ptr() == log_alpha._bgk_grad_dst.data_ptr())
    g_la = log_alpha._bgk_grad_dst if la_direct else (torch.zeros((1,), dtype=torch.float32, device=dev) if et is not None else None)
    with torch.cuda.device(dev):
        st = _lib.lib().bgk_affine_backward(
            _lib.ptr(y2), y2.stride(0) if B > 1 else d, _lib.ptr(ms[0]) if es is not None else None, ldms,
            _lib.ptr(ms[1]) if et is not None else None, ldms, _lib.ptr(log_alpha.detach()),
            int(pv), int(circ), int(inverse), B, d, _lib.ptr(g_out2), ldgo, _lib.ptr(g_dl),
            _lib.ptr(g_y), d, _lib.ptr(g_ms[0]) if es is not None else None, ldms, _lib.ptr(g_ms[1]) if et is not None else None, ldms,
            _lib.ptr(g_la), _lib.ptr(absmax[0]), _lib.ptr(absmax[1]), _lib.stream_ptr(dev))
    _lib.check(st, "bgk_affine_backward")
    g_x = None
    if need_x:
        g_x = gx_out if gx_out is not None else torch.empty((B, plan["d_c"]), dtype=torch.float32, device=dev)
    grads, add = [], gx_add
    for k, e in enumerate((es, et)):
        if e is None:
            grads += [None] * 6
            continue
        grads += list(_affine_net_backward(e, plan, g_ms[k], ldms, zz[2 * k + 1], zz[2 * k], x2, ldc, absmax[k], need_x, g_x, add,
                                           need_w[6 * k:6 * k + 6], versions[k]))
        add = g_x                       # the second network adds to what the first wrote
    return g_x, g_y, (None if (la_direct or et is None or not need_la) else g_la), grads


def _affine_train_backward_fused(plan, versions, x2, y2, zz, ms, log_alpha, g_out, g_dl, need_x, need_la, need_w, gx_add, gx_out, inverse=False):
    """_affine_train_backward for a layer of plan["tail_fused"] (no volume preservation): ONE library call,
    bgk_affine_coupling_backward64 -- tail backward + both networks' backward; from the saved z0 / z1 / s_raw, or (zz is None: the
    forward saved nothing) from x, y, g_out, g_dlogp alone.  ``inverse``: the layer ran in the inverse direction and ``y2`` is its OUTPUT."""
    es, et = plan["nets"]
    for e, v in zip((es, et), versions):
        if e["version"] != v:
            raise RuntimeError("fused affine coupling: the conditioner's parameters changed between the forward and this backward (the packed "
                               "operands were rewritten); run the backward before the optimizer step, or call forward again")
    dev = y2.device
    B, d = y2.shape
    n_in = es["lins"][0].in_features
    lib = _lib.lib()
    ldc = x2.stride(0) if B > 1 else x2.shape[1]
    ldy = y2.stride(0) if B > 1 else d
    g_out2, ldgo = _lib.rowmajor(g_out.reshape(B, d))
    g_y = torch.empty((B, d), dtype=torch.float32, device=dev)
    g_mu = torch.empty((B, d), dtype=torch.float32, device=dev) if inverse else None
    g_x = (gx_out if gx_out is not None else torch.empty((B, plan["d_c"]), dtype=torch.float32, device=dev)) if need_x else None
    add2, lda = (gx_add, gx_add.stride(0)) if (gx_add is not None and need_x) else (None, 0)
    tb = es["tbufs"]
    need_ws = int(lib.bgk_affine_coupling_backward64_workspace(B, d, n_in, es["H1"], es["H0"], et["H1"], et["H0"]))
    ws = tb.get("bwd64r_ws")
    if ws is None or ws.numel() < need_ws or ws.device != dev:
        ws = tb["bwd64r_ws"] = torch.empty(need_ws, dtype=torch.float32, device=dev)
    params = [p for e in (es, et) for lin in e["lins"] for p in (lin.weight, lin.bias)]            # W0, b0, W1, b1, W2, b2 per network
    bucket = lambda p: (getattr(p, "_bgk_grad_dst", None) is not None and p.grad is not None                    # noqa: E731
                        and p.grad.data_ptr() == p._bgk_grad_dst.data_ptr())
    direct = _DIRECT_GRADS[0] and all(need_w) and all(bucket(p) for p in params) and (not need_la or bucket(log_alpha))
    if direct:
        outs = [p._bgk_grad_dst for p in params]
        g_la = log_alpha._bgk_grad_dst if need_la else torch.empty((1,), dtype=torch.float32, device=dev)
    else:
        outs = [torch.empty_like(p, dtype=torch.float32) if nd else None for p, nd in zip(params, need_w)]
        g_la = torch.empty((1,), dtype=torch.float32, device=dev)
    if direct and not need_la:
        g_la = torch.zeros((1,), dtype=torch.float32, device=dev)       # (accumulate = 1 adds into it; the value is dropped)
    saved = [None] * 5 + [0] if zz is None else [_lib.ptr(zz[0]), _lib.ptr(zz[1]), _lib.ptr(zz[2]), _lib.ptr(zz[3]), _lib.ptr(ms[0]), ms.shape[2]]
    arr = lambda six: (ctypes.c_void_p * 6)(*[None if six[i] is None else six[i].data_ptr() for i in (4, 5, 2, 3, 0, 1)])      # noqa: E731  -> gW2, gb2, gW1, gb1, gW0, gb0
    s_arr, t_arr = arr(outs[:6]), arr(outs[6:])
    with torch.cuda.device(dev):
        st = lib.bgk_affine_coupling_backward64(
            _lib.ptr(x2), ldc, n_in, _lib.ptr(y2), ldy, d, _lib.ptr(g_out2), ldgo, _lib.ptr(g_dl), *saved,
            _lib.ptr(es["A0"]), _lib.ptr(es["A1"]), _lib.ptr(es["tbufs"]["T0"]), _lib.ptr(es["tbufs"]["T1"]), _lib.ptr(es["tbufs"]["T2"]),
            _lib.ptr(es["cs"]), es["act"], es["H1"], es["H0"],
            _lib.ptr(et["A0"]), _lib.ptr(et["A1"]), _lib.ptr(et["A2"]), _lib.ptr(et["tbufs"]["T0"]), _lib.ptr(et["tbufs"]["T1"]), _lib.ptr(et["tbufs"]["T2"]),
            _lib.ptr(et["cs"]), et["act"], et["H1"], et["H0"],
            _lib.ptr(log_alpha.detach()), int(inverse), B, _lib.ptr(g_y), d, _lib.ptr(g_mu), _lib.ptr(g_x) if need_x else None, g_x.stride(0) if need_x else n_in, _lib.ptr(add2), lda,
            _lib.ptr(g_la), _lib.ptr(ws), ws.numel(), s_arr, t_arr, int(direct), _lib.stream_ptr(dev))
    _lib.check(st, "bgk_affine_coupling_backward64")
    grads = [None] * 12 if direct else outs
    return g_x, g_y, (None if (direct or not need_la) else g_la), grads


class _FusedAffineTrainFn(torch.autograd.Function):
    """apply(x, y, log_alpha, plan, cfg, *12 network parameters (shift W0, b0, W1, b1, W2, b2, scale ...; None for an absent network))
    -> (y', dlogp [B, 1]).  ``cfg`` = (preserve_volume, is_circular, inverse)."""

    @staticmethod
    def forward(ctx, x, y, log_alpha, plan, cfg, *weights):
        out, dlogp, (x2, y2, zz, ms) = _affine_train_forward(x, y, log_alpha, plan, cfg)
        ctx.save_for_backward(x2, y2, log_alpha, zz, ms)
        ctx.plan, ctx.cfg = plan, cfg
        ctx.versions = [None if e is None else e["version"] for e in plan["nets"]]
        ctx.x_shape = x.shape
        return out, dlogp[:, None]

    @staticmethod
    def backward(ctx, g_out, g_dlogp):
        x2, y2, log_alpha, zz, ms = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_x, g_y, g_la, grads = _affine_train_backward(ctx.plan, ctx.cfg, ctx.versions, x2, y2, log_alpha, zz, ms, g_out,
                                                       g_dlogp.reshape(-1).contiguous(), bool(need[0]), bool(need[2]), need[5:17])
        return (g_x.reshape(ctx.x_shape) if g_x is not None else None, g_y if need[1] else None, g_la, None, None, *grads)


class _AffineStackTrainFn(torch.autograd.Function):
    """``split -> (CouplingFlow(AffineTransformer) | SwapFlow)* -> merge`` under autograd as ONE node (BASELINE cfg 2's whole flow).
    Forward: one training-forward launch per coupling, all of them adding their log-det to one [B] buffer, the halves read and written
    as column views of [B, D] tensors (no per-layer torch.cat, no per-block log-det add).  Backward: the layers in reverse with the two
    halves' gradients kept in two buffers -- a coupling's conditioner-input gradient is ADDED to its half's buffer inside the networks'
    backward kernels (g_cond_add), so the sums autograd would form with one elementwise launch per layer (a half conditions one layer
    and is transformed by the next) do not exist.

    apply(layers, cols, out_cols, x [B, D], *per layer (log_alpha, 12 network parameters)); ``layers[i]`` = (transformed part 0 | 1, plan,
    cfg); ``cols`` / ``out_cols``: the column ranges of the two halves in x / in the result (exchanged after an odd number of swaps).
    Returns (out [B, D] with the parts in slot order, dlogp [B, 1])."""

    @staticmethod
    def forward(ctx, layers, cols, out_cols, x, *tensors):
        B = x.shape[0]
        part = [x[:, cols[0]], x[:, cols[1]]]
        saved, dlogp = [], None
        # the last layer that transforms a half writes it straight into its columns of the result (no torch.cat at the end)
        last = {py: i for i, (py, _, _) in enumerate(layers)}
        final = torch.empty_like(x) if len(last) == 2 else None
        for i, (py, plan, cfg) in enumerate(layers):
            log_alpha = tensors[13 * i]
            dst = final[:, out_cols[py]] if (final is not None and last[py] == i) else None
            out, dlogp, (x2, y2, zz, ms) = _affine_train_forward(part[1 - py], part[py], log_alpha, plan, cfg, dlogp=dlogp, accumulate=i > 0, out=dst)
            saved += [x2, y2, log_alpha, zz, ms]
            part[py] = out
        ctx.save_for_backward(*saved)
        ctx.layers = layers
        ctx.versions = [[None if e is None else e["version"] for e in plan["nets"]] for _, plan, _ in layers]
        ctx.out_cols = out_cols
        if final is None:                            # a half no layer transforms: assemble the result from the parts
            final = torch.empty_like(x)
            final[:, out_cols[0]], final[:, out_cols[1]] = part[0], part[1]
        return final, dlogp[:, None]

    @staticmethod
    def backward(ctx, g_out, g_dlogp):
        saved = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_out = g_out.contiguous()
        G = [g_out[:, ctx.out_cols[0]], g_out[:, ctx.out_cols[1]]]          # gradient w.r.t. the CURRENT state of each half, walking the layers backwards
        owned = [False, False]                       # buffers of this backward (may be added to in place)
        g_dl = g_dlogp.reshape(-1).contiguous()
        L = len(ctx.layers)
        grads = [None] * (13 * L)
        for i in range(L - 1, -1, -1):
            py, plan, cfg = ctx.layers[i]
            pc = 1 - py
            x2, y2, log_alpha, zz, ms = saved[5 * i:5 * i + 5]
            nd = need[4 + 13 * i:4 + 13 * i + 13]
            prev = G[pc]
            if not owned[pc]:                        # autograd's tensor (a view of g_out): never written -- the sum goes to a fresh buffer
                gx_out = None
            else:
                gx_out = prev
            want_gx = i > 0 or bool(need[3])          # (the first layer's conditioner-input gradient only matters to the stack's input)
            g_x, g_y, g_la, gws = _affine_train_backward(plan, cfg, ctx.versions[i], x2, y2, log_alpha, zz, ms, G[py], g_dl, want_gx, bool(nd[0]),
                                                         nd[1:13], gx_add=prev, gx_out=gx_out)
            G[py], owned[py] = g_y, True
            if want_gx:
                G[pc], owned[pc] = g_x, True
            grads[13 * i] = g_la
            grads[13 * i + 1:13 * i + 13] = gws
        g_in = torch.cat(G, dim=1) if need[3] else None
        return (None, None, None, g_in, *grads)


def affine_stack_train(blocks, x, inverse):
    """The blocks ``split, (coupling | swap)*, merge`` (execution order) of a coupling stack on a [B, D] f32 HIP tensor as one autograd node
    (_AffineStackTrainFn); None when a layer is outside the affine training envelope (the caller then runs the blocks one by one)."""
    from .flow import CouplingFlow, SplitFlow, SwapFlow
    if not AFFINE_TRAIN or not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0):
        return None
    split = blocks[0] if type(blocks[0]) is SplitFlow else blocks[0]._delegate
    s0, D = split._sizes[0], x.shape[1]
    if not (0 < s0 < D and (len(split._sizes) == 1 or split._sizes[1] == D - s0)):
        return None
    widths = (s0, D - s0)
    part = [0, 1]                                   # tuple slot -> half of x
    layers, tensors = [], []
    for b in blocks[1:-1]:
        if type(b) is SwapFlow:
            part.reverse()
            continue
        tr = b.transformer
        if not getattr(tr, "allow_fused", False):
            return None
        pc, py = part
        plan = _affine_train_plan(tr, widths[py], x.device)
        if plan is None or plan["d_c"] != widths[pc]:
            return None
        plan["train_used"] = True
        _AFF_TRAIN_TRANSFORMERS.add(tr)
        log_alpha = tr._log_alpha
        if log_alpha.device != x.device or log_alpha.dtype != torch.float32:
            log_alpha = log_alpha.to(device=x.device, dtype=torch.float32)
        layers.append((py, plan, (tr._preserve_volume, tr._is_circular, inverse)))
        tensors.append(log_alpha)
        for e in plan["nets"]:
            tensors += [None] * 6 if e is None else [p for lin in e["lins"] for p in (lin.weight, lin.bias)]
    if not layers:
        return None
    closing = blocks[-1] if type(blocks[-1]) is SplitFlow else blocks[-1]._delegate
    if not (closing._sizes[0] == widths[part[0]] and (len(closing._sizes) == 1 or closing._sizes[1] == widths[part[1]])):
        return None
    cols = (slice(0, s0), slice(s0, D))
    # where the halves sit in the result: an odd number of swaps makes the merge concatenate (half 1, half 0)
    out_cols = cols if part == [0, 1] else (slice(D - s0, D), slice(0, D - s0))
    out, dlogp = _AffineStackTrainFn.apply(layers, cols, out_cols, x, *tensors)
    return out, dlogp


def fused_affine_coupling_train(transformer, x, y, inverse):
    """Differentiable one-launch forward of the affine coupling layer (see _FusedAffineTrainFn), or None outside the envelope: networks
    that are not two-hidden-layer DenseNets of <= 128 units with SiLU / ReLU / Tanh, > 96 input features or transformed dims, gemm_mode
    'f32', inputs that are not 2-d f32 HIP tensors."""
    if not AFFINE_TRAIN or x.dim() != 2 or y.dim() != 2 or not y.is_cuda or x.dtype != torch.float32 or y.dtype != torch.float32 \
            or x.shape[0] != y.shape[0] or y.shape[0] == 0:
        return None
    plan = _affine_train_plan(transformer, y.shape[-1], y.device)
    if plan is None or x.shape[-1] != plan["d_c"]:
        return None
    _lib.require_hip(x, y)
    plan["train_used"] = True
    _AFF_TRAIN_TRANSFORMERS.add(transformer)
    weights = []
    for e in plan["nets"]:
        weights += [None] * 6 if e is None else [p for lin in e["lins"] for p in (lin.weight, lin.bias)]
    log_alpha = transformer._log_alpha
    if log_alpha.device != y.device or log_alpha.dtype != torch.float32:
        log_alpha = log_alpha.to(device=y.device, dtype=torch.float32)
    return _FusedAffineTrainFn.apply(x, y, log_alpha, plan, (transformer._preserve_volume, transformer._is_circular, inverse), *weights)
