"""The coupling layers of tests/golden/g_envelope.npz (conditioners at the wide / deep end of the one-launch kernels' envelope), built
from bgflow_amd classes: the same constructor calls as tests/golden/make_goldens.py::envelope_*_layer makes with the reference's
classes, and bgflow_amd.utils.hash_init_ gives the same weights by parameter name."""
import torch

import bgflow_amd as bg
from bgflow_amd.utils import hash_init_, synth

SPLINE = {"w256": (256, 256), "w200_130": (200, 130), "deep1": (128,), "deep3": (128, 128, 128), "deep4": (64, 128, 32, 100)}
AFFINE = {"readme4": (4,), "deep5": (48,) * 5, "deep4mixed": (128, 64, 32, 100)}
KINDS = {"pc": (True, True), "nn": (False, False)}          # (periodic conditioner input, circular spline)
B = 97


def spline_layer(hidden, periodic, circular, d_c=9, d=7, n_bins=8):
    P = 3 * n_bins * d + (0 if circular else d)
    net = bg.DenseNet([2 * d_c if periodic else d_c, *hidden, P], activation=torch.nn.SiLU())
    if periodic:
        net = bg.WrapPeriodic(net)
    return hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(net, is_circular=circular), transformed_indices=(1,), cond_indices=(0,)))


def affine_layer(hidden, d_c=12, d=20):
    return hash_init_(bg.CouplingFlow(bg.AffineTransformer(bg.DenseNet([d_c, *hidden, d], activation=torch.nn.ReLU()),
                                                           bg.DenseNet([d_c, *hidden, d], activation=torch.nn.Tanh())),
                                      transformed_indices=(1,), cond_indices=(0,)))


def spline_inputs(periodic):
    return synth(61, B, 9, uniform=periodic), synth(62, B, 7, uniform=True)


def affine_inputs():
    return synth(63, B, 12), synth(64, B, 20)
