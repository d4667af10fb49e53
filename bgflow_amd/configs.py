"""The BASELINE.json workloads, assembled from bgflow_amd blocks.

The block sequence, tensor slots, conditioner shapes and parameter NAMES reproduce what bgflow's
``BoltzmannGeneratorBuilder`` emits for the same recipe (factory/generator_builder.py:238-321,
408-459; conditioner_factory.py:23-80,230-241; icmarginals.py:14-77), so name-keyed synthetic
weights (utils.hash_init_) and reference state_dicts line up.

  cfg 1  README flow: dim 2, SplitFlow / one RealNVP coupling / MergeFlow           (README.md:54-96)
  cfg 2  dim 64, 8 x (affine coupling, swap), DenseNet [32,64,64,32] ReLU / Tanh
  cfg 3  alanine-dipeptide shaped: 4 x (T|F, F|T) + 4 x (B|A, A|B) RQ-spline couplings (K = 8,
         hidden 128x128 SiLU) + 4 icdf domain maps + Mixed internal-coordinate transform -> 66 xyz
  cfg 5  cfg-3 coordinate transform + 66 auxiliary dims, 10 spline + 6 affine couplings
"""
import os

import numpy as np
import torch

from .bg import BoltzmannGenerator
from .cdf import CDFTransform
from .dense import DenseNet, WrapPeriodic
from .distributions import (DoubleWellEnergy, NormalDistribution, ProductDistribution, SloppyUniform,
                            TruncatedNormalDistribution, UniformDistribution)
from .flow import CouplingFlow, InverseFlow, MergeFlow, SequentialFlow, SplitFlow, SwapFlow, WrapFlow
from .ic import MixedCoordinateTransformation
from .transformer import AffineTransformer, ConditionalSplineTransformer
from .utils import hash_init_, synth

__all__ = ["ala2_system", "make_readme_generator", "make_affine8_generator", "make_ala2_spline_generator",
           "make_ala2_augmented_generator", "IC_FIELDS"]

IC_FIELDS = ("BONDS", "ANGLES", "TORSIONS", "FIXED")
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ala2_system.npz")


def ala2_system():
    """Topology tables (relative Z-matrix, rigid block) and one geometry of capped alanine: the
    22-atom example the reference's own test-suite uses (tests/nn/flow/crd_transform/test_ic.py:37-118)."""
    d = np.load(_DATA)
    return d["z_matrix"].astype(np.int64), d["rigid_block"].astype(np.int64), d["xyz"].reshape(1, -1)


def ala2_whitening_data(dtype=torch.float32):
    """1000 synthetic frames: reference geometry + closed-form 0.01 nm noise (SURVEY.md 8(d) cfg 3)."""
    _, _, xyz = ala2_system()
    return torch.tensor((xyz + 0.01 * synth(0, 1000, 66, scale=1.0, dtype=np.float64)).astype(np.float32), dtype=dtype)


# ------------------------------------------------------------------------------------------------
def make_readme_generator(device=None):
    dim = 2
    prior = NormalDistribution(dim)
    target = DoubleWellEnergy(dim)
    layers = [
        SplitFlow(dim // 2),
        CouplingFlow(AffineTransformer(
            shift_transformation=DenseNet([dim // 2, 4, dim // 2], activation=torch.nn.ReLU()),
            scale_transformation=DenseNet([dim // 2, 4, dim // 2], activation=torch.nn.Tanh()))),
        InverseFlow(SplitFlow(dim // 2)),
    ]
    flow = hash_init_(SequentialFlow(layers))
    gen = BoltzmannGenerator(prior, flow, target)
    return gen.to(device) if device is not None else gen


def make_affine8_generator(dim=64, n_blocks=8, hidden=(64, 64), device=None):
    half = dim // 2
    layers = [SplitFlow(half)]
    for _ in range(n_blocks):
        layers.append(CouplingFlow(AffineTransformer(
            shift_transformation=DenseNet([half, *hidden, half], activation=torch.nn.ReLU()),
            scale_transformation=DenseNet([half, *hidden, half], activation=torch.nn.Tanh()))))
        layers.append(SwapFlow())
    layers.append(MergeFlow(half))
    flow = hash_init_(SequentialFlow(layers))
    gen = BoltzmannGenerator(NormalDistribution(dim), flow, DoubleWellEnergy(dim))
    return gen.to(device) if device is not None else gen


# ------------------------------------------------------------------------------------------------
def _spline_coupling(what, on, dims, circular, slot, hidden=(128, 128), num_bins=8):
    """One builder-style spline coupling: field ``what`` transformed, field ``on`` conditions."""
    d_what, d_on = dims[what], dims[on]
    what_circ, on_circ = circular[what], circular[on]
    dim_out = 3 * num_bins * d_what + (0 if what_circ else d_what)
    dim_in = 2 * d_on if on_circ else d_on
    net = DenseNet([dim_in, *hidden, dim_out], activation=torch.nn.SiLU())
    if on_circ:
        net = WrapPeriodic(net, indices=np.arange(d_on))
    transformer = ConditionalSplineTransformer(params_net=net, is_circular=torch.full((d_what,), bool(what_circ)))
    return CouplingFlow(transformer, transformed_indices=[slot[what]], cond_indices=[slot[on]])


def _ic_domain_maps(dims, slot, ctx):
    """add_map_to_ic_domains with the default marginals (icmarginals.py:14-77)."""
    one = lambda n, v=1.0: v * torch.ones(n, **ctx)  # noqa: E731
    marginals = {
        "BONDS": TruncatedNormalDistribution(mu=one(dims["BONDS"]), sigma=one(dims["BONDS"]),
                                             lower_bound=torch.as_tensor(1e-5, **ctx),
                                             upper_bound=torch.as_tensor(np.inf, **ctx)),
        "ANGLES": TruncatedNormalDistribution(mu=one(dims["ANGLES"], 0.5), sigma=one(dims["ANGLES"]),
                                              lower_bound=torch.as_tensor(1e-5, **ctx),
                                              upper_bound=torch.as_tensor(1.0, **ctx)),
        "TORSIONS": SloppyUniform(low=one(dims["TORSIONS"], 0.0), high=one(dims["TORSIONS"])),
        "FIXED": _NormalMarginal(torch.zeros(dims["FIXED"], **ctx), one(dims["FIXED"], 20.0)),
    }
    if "AUGMENTED" in dims:
        marginals["AUGMENTED"] = _NormalMarginal(torch.zeros(dims["AUGMENTED"], **ctx), one(dims["AUGMENTED"]))
    return [WrapFlow(InverseFlow(CDFTransform(marginals[f])), (slot[f],)) for f in marginals if f in slot]


class _NormalMarginal(torch.nn.Module):
    """torch.distributions.Normal(loc, scale) as a module (so .to(device) moves it); same
    cdf / icdf / log_prob arithmetic."""

    def __init__(self, loc, scale):
        super().__init__()
        self.register_buffer("loc", loc)
        self.register_buffer("scale", scale)

    def _d(self):
        return torch.distributions.Normal(self.loc, self.scale, validate_args=False)

    def cdf(self, x):
        return self._d().cdf(x)

    def icdf(self, x):
        return self._d().icdf(x)

    def log_prob(self, x):
        return self._d().log_prob(x)


def make_ala2_spline_generator(device=None, dtype=torch.float32, hidden=(128, 128), num_bins=8,
                               n_torsion_blocks=4, n_bond_blocks=4):
    """cfg 3 / 4: 16 RQ-spline couplings + icdf domain maps + Mixed IC -> 66 Cartesian dof."""
    zmat, rigid, xyz = ala2_system()
    ic = MixedCoordinateTransformation(ala2_whitening_data(dtype), zmat, rigid, keepdims=9, raise_warnings=False)
    dims = {"BONDS": ic.dim_bonds, "ANGLES": ic.dim_angles, "TORSIONS": ic.dim_torsions, "FIXED": ic.dim_fixed}
    circular = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(IC_FIELDS)}
    ctx = dict(dtype=dtype)
    layers = []
    for _ in range(n_torsion_blocks):
        layers.append(_spline_coupling("TORSIONS", "FIXED", dims, circular, slot, hidden, num_bins))
        layers.append(_spline_coupling("FIXED", "TORSIONS", dims, circular, slot, hidden, num_bins))
    for _ in range(n_bond_blocks):
        layers.append(_spline_coupling("BONDS", "ANGLES", dims, circular, slot, hidden, num_bins))
        layers.append(_spline_coupling("ANGLES", "BONDS", dims, circular, slot, hidden, num_bins))
    layers += _ic_domain_maps(dims, slot, ctx)
    layers.append(WrapFlow(InverseFlow(ic), indices=[0, 1, 2, 3], out_indices=(0,)))
    flow = hash_init_(SequentialFlow(layers))
    prior = ProductDistribution([UniformDistribution(torch.zeros(dims[f], **ctx), torch.ones(dims[f], **ctx))
                                 for f in IC_FIELDS])
    target = NormalDistribution(66, torch.tensor(xyz[0], dtype=dtype))
    gen = BoltzmannGenerator(prior, flow, target)
    return gen.to(device) if device is not None else gen


def _affine_coupling(what, on, dims, circular, slot, hidden=(128, 128)):
    """One builder-style affine (RealNVP) coupling: ``what`` transformed, fields ``on`` condition
    (conditioner_factory.py:236-241: separate shift and scale DenseNets, SiLU hidden layers)."""
    on = (on,) if isinstance(on, str) else tuple(on)
    d_what = dims[what]
    d_nc = sum(dims[f] for f in on if not circular[f])
    d_c = sum(dims[f] for f in on if circular[f])
    dim_in = d_nc + 2 * d_c

    def net():
        n = DenseNet([dim_in, *hidden, d_what], activation=torch.nn.SiLU())
        if d_c > 0:
            idx = np.concatenate([np.arange(dims[f]) + sum(dims[g] for g in on[:k])
                                  for k, f in enumerate(on) if circular[f]])
            n = WrapPeriodic(n, indices=idx)
        return n
    transformer = AffineTransformer(shift_transformation=net(), scale_transformation=net(), is_circular=False)
    return CouplingFlow(transformer, transformed_indices=[slot[what]], cond_indices=[slot[f] for f in on])


def make_ala2_augmented_generator(device=None, dtype=torch.float32):
    """cfg 5 (augmented normalizing flow): the cfg-3 coordinate transform + 66 auxiliary dims; 16 couplings
    = 10 RQ-spline + 6 affine, 5 icdf maps, Mixed IC.  ``flow(bonds, angles, torsions, fixed, aug) ->
    (x[B,66], aug[B,66], dlogp)``."""
    zmat, rigid, xyz = ala2_system()
    ic = MixedCoordinateTransformation(ala2_whitening_data(dtype), zmat, rigid, keepdims=9, raise_warnings=False)
    fields = IC_FIELDS + ("AUGMENTED",)
    dims = {"BONDS": ic.dim_bonds, "ANGLES": ic.dim_angles, "TORSIONS": ic.dim_torsions, "FIXED": ic.dim_fixed,
            "AUGMENTED": 66}
    circular = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False, "AUGMENTED": False}
    slot = {f: i for i, f in enumerate(fields)}
    ctx = dict(dtype=dtype)
    layers = []
    for _ in range(4):
        layers.append(_spline_coupling("TORSIONS", "AUGMENTED", dims, circular, slot))
        layers.append(_affine_coupling("AUGMENTED", "TORSIONS", dims, circular, slot))
    for _ in range(2):
        layers.append(_spline_coupling("BONDS", "ANGLES", dims, circular, slot))
        layers.append(_spline_coupling("ANGLES", "BONDS", dims, circular, slot))
    for _ in range(2):
        layers.append(_spline_coupling("FIXED", "AUGMENTED", dims, circular, slot))
        layers.append(_affine_coupling("AUGMENTED", ("FIXED", "BONDS", "ANGLES"), dims, circular, slot))
    layers += _ic_domain_maps(dims, slot, ctx)
    layers.append(WrapFlow(InverseFlow(ic), indices=[0, 1, 2, 3], out_indices=(0,)))
    flow = hash_init_(SequentialFlow(layers))
    prior = ProductDistribution([UniformDistribution(torch.zeros(dims[f], **ctx), torch.ones(dims[f], **ctx))
                                 for f in fields])
    target = ProductDistribution([NormalDistribution(66, torch.tensor(xyz[0], dtype=dtype)),
                                  NormalDistribution(66, torch.zeros(66, **ctx))])
    gen = BoltzmannGenerator(prior, flow, target)
    return gen.to(device) if device is not None else gen
