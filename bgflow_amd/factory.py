"""High-level construction API: shape bookkeeping + BoltzmannGeneratorBuilder (bgflow/factory/*).

Pure host-side Python -- no kernels.  It emits exactly the blocks of bgflow_amd.flow / transformer / dense / ic / cdf that the
hot path accelerates, with the reference builder's block order, tensor slots and parameter names
(factory/generator_builder.py:108-459, tensor_info.py:17-371, conditioner_factory.py:23-80,230-251,
transformer_factory.py:15-43, distribution_factory.py:10-57, icmarginals.py:14-77), so a script written against
``bgflow.BoltzmannGeneratorBuilder`` runs unchanged and checkpoints line up.  Covered: dense conditioners, spline and affine
transformers, split / merge / set-constant / arbitrary layers, the default IC marginals, Relative / Mixed / Global coordinate
transforms.  Not covered (out of the hot path's scope, raise NotImplementedError): GNN conditioners, constraint merging,
chirality / torsion-multiplicity helpers.
"""
import warnings
from collections import OrderedDict, namedtuple

import numpy as np
import torch

from .bg import BoltzmannGenerator
from .cdf import CDFTransform
from .dense import DenseNet, WrapPeriodic
from .distributions import (NormalDistribution, ProductDistribution, SloppyUniform, TruncatedNormalDistribution,
                            UniformDistribution)
from .flow import CouplingFlow, Flow, InverseFlow, MergeFlow, SequentialFlow, SetConstantFlow, SplitFlow, WrapFlow
from .ic import GlobalInternalCoordinateTransformation
from .transformer import AffineTransformer, ConditionalSplineTransformer

__all__ = ["TensorInfo", "ShapeDictionary", "BONDS", "ANGLES", "TORSIONS", "FIXED", "ORIGIN", "ROTATION", "AUGMENTED", "TARGET",
           "make_conditioners", "make_transformer", "make_distribution", "InternalCoordinateMarginals",
           "BoltzmannGeneratorBuilder"]


class TensorInfo(namedtuple("TensorInfo", ["name", "is_circular", "is_cartesian"], defaults=(False, False))):
    """Name and support of one tensor of the flow state (tensor_info.py:17-34)."""
    __slots__ = ()


BONDS = TensorInfo("BONDS")
ANGLES = TensorInfo("ANGLES")
TORSIONS = TensorInfo("TORSIONS", is_circular=True)
FIXED = TensorInfo("FIXED", is_cartesian=True)
ORIGIN = TensorInfo("ORIGIN", is_cartesian=True)
ROTATION = TensorInfo("ROTATION")
AUGMENTED = TensorInfo("AUGMENTED")
TARGET = TensorInfo("TARGET", is_cartesian=True)


class ShapeDictionary(OrderedDict):
    """Ordered map TensorInfo -> shape; position = slot of the tensor in the flow's state tuple (tensor_info.py:55-371)."""

    @staticmethod
    def from_coordinate_transform(coordinate_transform, dim_augmented=0, n_constraints=0, remove_origin_and_rotation=True):
        ct = coordinate_transform
        shapes = ShapeDictionary()
        if ct.dim_angles > 0:      # (the reference keys the bond entry on dim_angles too, tensor_info.py:87-90)
            shapes[BONDS] = (ct.dim_bonds - n_constraints,)
            shapes[ANGLES] = (ct.dim_angles,)
        if ct.dim_torsions > 0:
            shapes[TORSIONS] = (ct.dim_torsions,)
        if ct.dim_fixed > 0:
            shapes[FIXED] = (ct.dim_fixed,)
        if dim_augmented > 0:
            shapes[AUGMENTED] = (dim_augmented,)
        if isinstance(ct, GlobalInternalCoordinateTransformation) and not remove_origin_and_rotation:
            shapes[ORIGIN] = (1, 3)
            shapes[ROTATION] = (3,)
        return shapes

    # -- structure edits ---------------------------------------------------------------------------------
    def insert(self, key, index, size):
        assert key not in self
        if index < 0:
            index = len(self) - index      # (sic: the reference's convention, tensor_info.py:192-193)
        items = list(self.items())
        items.insert(min(index, len(items)), (key, tuple(size)))
        self.clear()
        for k, v in items:
            self[k] = v

    def split(self, key, into, sizes, dim=-1):
        shape = list(self[key])
        if sum(sizes) != shape[dim]:
            raise ValueError(f"split sizes {sizes} do not sum up to total ({self[key]})")
        at = self.index(key)
        del self[key]
        for offset, (field, size) in enumerate(zip(into, sizes)):
            assert field not in self
            shape[dim] = size
            self.insert(field, at + offset, tuple(shape))

    def merge(self, keys, to, index=None, dim=-1):
        shape = list(self[keys[0]])
        shape[dim] = sum(self[k][dim] for k in keys)
        first = min(self.index(k) for k in keys)
        for k in keys:
            del self[k]
        assert to not in self
        self.insert(to, first if index is None else index, tuple(shape))

    def replace(self, key, other):
        if isinstance(other, str):
            other = key._replace(name=other)
        at, shape = self.index(key), self[key]
        del self[key]
        self.insert(other, at, shape)
        return other

    def copy(self):
        clone = ShapeDictionary()
        for k, v in self.items():
            clone[k] = v
        return clone

    # -- queries ----------------------------------------------------------------------------------------------
    def _keys(self, keys):
        return list(self) if keys is None else list(keys)

    def index(self, key, keys=None):
        return self._keys(keys).index(key)

    def names(self, keys=None):
        return [k.name for k in self._keys(keys)]

    def dim_all(self, keys=None, dim=-1):
        return sum(self[k][dim] for k in self._keys(keys))

    def dim_circular(self, keys=None, dim=-1):
        return sum(self[k][dim] for k in self._keys(keys) if k.is_circular)

    def dim_noncircular(self, keys=None, dim=-1):
        return sum(self[k][dim] for k in self._keys(keys) if not k.is_circular)

    def dim_cartesian(self, keys=None, dim=-1):
        return sum(self[k][dim] for k in self._keys(keys) if k.is_cartesian)

    def dim_noncartesian(self, keys=None, dim=-1):
        return sum(self[k][dim] for k in self._keys(keys) if not k.is_cartesian)

    def _mask(self, keys, dim, attr):
        parts = [np.full(self[k][dim], bool(getattr(k, attr))) for k in self._keys(keys)]
        return np.concatenate(parts) if parts else np.zeros(0, bool)

    def is_circular(self, keys=None, dim=-1):
        return self._mask(keys, dim, "is_circular")

    def circular_indices(self, keys=None, dim=-1):
        return np.arange(self.dim_all(keys, dim))[self.is_circular(keys, dim)]

    def is_cartesian(self, keys=None, dim=-1):
        return self._mask(keys, dim, "is_cartesian")

    def cartesian_indices(self, keys=None, dim=-1):
        return np.arange(self.dim_all(keys, dim))[self.is_cartesian(keys, dim)]


# ---- factories ---------------------------------------------------------------------------------------------------
def _dense_conditioner(dim_in, dim_out, hidden=(128, 128), activation=None, **_):
    return DenseNet([dim_in, *hidden, dim_out], activation=torch.nn.SiLU() if activation is None else activation)


def _gnn_conditioner(*_, **__):
    raise NotImplementedError("GNN conditioners (factory/GNN_factory.py, nequip) are outside the accelerated hot path")


CONDITIONER_FACTORIES = {"dense": _dense_conditioner, "GNN": _gnn_conditioner}


def _spline_out_dims(what, shape_info, transformer_kwargs=None, num_bins=8, **_):
    return {"params_net": 3 * num_bins * shape_info.dim_all(what) + shape_info.dim_noncircular(what)}


def _affine_out_dims(what, shape_info, transformer_kwargs=None, use_scaling=True, **_):
    d = shape_info.dim_all(what)
    return {"shift_transformation": d, "scale_transformation": d} if use_scaling else {"shift_transformation": d}


CONDITIONER_OUT_DIMS = {ConditionalSplineTransformer: _spline_out_dims, AffineTransformer: _affine_out_dims}


def make_conditioners(transformer_type, what, on, shape_info, transformer_kwargs=None, conditioner_type="dense", **kwargs):
    """One conditioner network per parameter group of ``transformer_type``; circular conditioning inputs are wrapped in the
    cos/sin featuriser (conditioner_factory.py:23-80)."""
    net_factory = CONDITIONER_FACTORIES[conditioner_type]
    out_dims = CONDITIONER_OUT_DIMS[transformer_type](what=what, shape_info=shape_info,
                                                      transformer_kwargs=transformer_kwargs or {}, **kwargs)
    dim_in = shape_info.dim_noncircular(on) + 2 * shape_info.dim_circular(on)
    net_kwargs = {k: v for k, v in kwargs.items() if k not in ("num_bins", "use_scaling")}
    conditioners = {}
    for name, dim_out in out_dims.items():
        net = net_factory(dim_in, dim_out, shape_info=shape_info, on=on, **net_kwargs)
        if shape_info.dim_circular(on) > 0:
            net = WrapPeriodic(net, indices=shape_info.circular_indices(on))
        conditioners[name] = net
    return conditioners


def _spline_transformer(what, shape_info, conditioners, **kwargs):
    return ConditionalSplineTransformer(is_circular=shape_info.is_circular(what), **conditioners, **kwargs)


def _affine_transformer(what, shape_info, conditioners, **kwargs):
    n_circ = shape_info.dim_circular(what)
    if n_circ not in (0, shape_info[what[0]][-1]):
        raise NotImplementedError("Circular affine transformers are currently not supported for partly circular indices.")
    return AffineTransformer(**conditioners, is_circular=n_circ > 0, **kwargs)


TRANSFORMER_FACTORIES = {ConditionalSplineTransformer: _spline_transformer, AffineTransformer: _affine_transformer}


def make_transformer(transformer_type, what, shape_info, conditioners, inverse=False, **kwargs):
    transformer = TRANSFORMER_FACTORIES[transformer_type](what=what, shape_info=shape_info, conditioners=conditioners, **kwargs)
    return InverseFlow(transformer) if inverse else transformer


def _on(value, device, dtype):
    return value.to(device=device, dtype=dtype) if isinstance(value, torch.Tensor) else value


def make_distribution(distribution_type, shape, device=None, dtype=None, **kwargs):
    """Prior factory (distribution_factory.py:10-57): defaults uniform [0, 1], standard normal, truncated normal(0, 1)."""
    if distribution_type is UniformDistribution:
        args = dict(low=torch.zeros(shape), high=torch.ones(shape))
    elif distribution_type is NormalDistribution:
        args = dict(dim=shape, mean=torch.zeros(shape))
    elif distribution_type is TruncatedNormalDistribution:
        args = dict(mu=torch.zeros(shape), sigma=torch.ones(shape))
    else:
        raise KeyError(distribution_type)
    args.update(kwargs)
    return distribution_type(**{k: _on(v, device, dtype) for k, v in args.items()})


class _NormalMarginal(torch.nn.Module):
    """torch.distributions.Normal(loc, scale) as a module (moves with .to(device)); same cdf / icdf / log_prob arithmetic."""

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


class InternalCoordinateMarginals(dict):
    """Default marginal distributions whose inverse CDFs map the unit cube onto the IC domains (icmarginals.py:14-77)."""

    def __init__(self, current_dims, ctx, bond_mu=1.0, bond_sigma=1.0, bond_lower=1e-5, bond_upper=np.inf,
                 angle_mu=0.5, angle_sigma=1.0, angle_lower=1e-5, angle_upper=1.0, torsion_lower=0.0, torsion_upper=1.0,
                 fixed_scale=20.0, bonds=BONDS, angles=ANGLES, torsions=TORSIONS, fixed=FIXED, augmented=AUGMENTED):
        super().__init__()
        self.ctx, self.current_dims = ctx, current_dims
        full = lambda field, v: v * torch.ones(current_dims[field], **ctx)  # noqa: E731
        scalar = lambda v: torch.as_tensor(v, **ctx)                          # noqa: E731
        if bonds in current_dims:
            self[bonds] = TruncatedNormalDistribution(mu=full(bonds, bond_mu), sigma=full(bonds, bond_sigma),
                                                      lower_bound=scalar(bond_lower), upper_bound=scalar(bond_upper))
        if angles in current_dims:
            self[angles] = TruncatedNormalDistribution(mu=full(angles, angle_mu), sigma=full(angles, angle_sigma),
                                                       lower_bound=scalar(angle_lower), upper_bound=scalar(angle_upper))
        if torsions in current_dims:
            self[torsions] = SloppyUniform(low=full(torsions, torsion_lower), high=full(torsions, torsion_upper))
        if fixed in current_dims:
            self[fixed] = _NormalMarginal(torch.zeros(current_dims[fixed], **ctx), full(fixed, fixed_scale))
        if augmented in current_dims:
            self[augmented] = _NormalMarginal(torch.zeros(current_dims[augmented], **ctx), full(augmented, 1.0))


# ---- the builder ----------------------------------------------------------------------------------------------------
def _tuple(thing):
    if isinstance(thing, TensorInfo) or not isinstance(thing, (tuple, list)):
        return (thing,)
    return tuple(thing)


def _common(settings, what, label):
    """per-field settings of all transformed fields must agree (generator_builder.py:274-295)"""
    first = settings[0]
    if any(s != first for s in settings[1:]):
        raise ValueError(f"Fields with different {label} cannot be transformed together.")
    return first


class BoltzmannGeneratorBuilder:
    """Incrementally assemble prior, flow and target of a Boltzmann generator (generator_builder.py:50-537).

    >>> shapes = ShapeDictionary.from_coordinate_transform(coordinate_transform)
    >>> builder = BoltzmannGeneratorBuilder(shapes, target=target, device=device, dtype=torch.float32)
    >>> for _ in range(4):
    ...     builder.add_condition(TORSIONS, on=FIXED)
    ...     builder.add_condition(FIXED, on=TORSIONS)
    >>> builder.add_map_to_ic_domains()
    >>> builder.add_map_to_cartesian(coordinate_transform)
    >>> generator = builder.build_generator()
    """

    def __init__(self, prior_dims, target=None, device=None, dtype=None):
        self.default_transformer_type = ConditionalSplineTransformer
        self.default_conditioner_type = "dense"
        self.default_transformer_kwargs = {}
        self.default_conditioner_kwargs = {}
        self.default_prior_type = UniformDistribution
        self.default_prior_kwargs = {}
        self.ctx = {"device": device, "dtype": dtype}
        self.prior_dims = prior_dims
        self.current_dims = prior_dims.copy()
        self.layers = []
        self.transformer_type, self.transformer_kwargs = {}, {}
        self.conditioner_type, self.conditioner_kwargs = {}, {}
        self.prior_type, self.prior_kwargs = {}, {}
        self.targets = {}
        if target is not None:
            self.targets[TARGET] = target
        if AUGMENTED in prior_dims:
            n = prior_dims[AUGMENTED]
            self.targets[AUGMENTED] = NormalDistribution(n, torch.zeros(n, **self.ctx))
        self.param_groups = {}

    # -- products ---------------------------------------------------------------------------------------------
    def build_generator(self, zero_parameters=False, check_target=True):
        generator = BoltzmannGenerator(prior=self.build_prior(), flow=self.build_flow(zero_parameters=zero_parameters),
                                       target=self.build_target(check_target=check_target))
        self.clear()
        return generator

    def build_flow(self, zero_parameters=False):
        flow = SequentialFlow(self.layers)
        if zero_parameters:
            warnings.warn("Initializing the flow with zeros makes it much less flexible", UserWarning)
            for p in flow.parameters():
                p.data.zero_()
        return flow

    def build_prior(self):
        priors = [make_distribution(self.prior_type.get(f, self.default_prior_type), self.prior_dims[f], **self.ctx,
                                    **self.prior_kwargs.get(f, self.default_prior_kwargs)) for f in self.prior_dims]
        return ProductDistribution(priors) if len(priors) > 1 else priors[0]

    def build_target(self, check_target=False):
        targets = []
        for field in self.current_dims:
            if field in self.targets:
                targets.append(self.targets[field])
            elif check_target:
                warnings.warn(f"No target energy for {field}.", UserWarning)
        if len(targets) > 1:
            return ProductDistribution(targets)
        return targets[0] if targets else None

    def clear(self):
        self.layers = []
        self.current_dims = self.prior_dims.copy()

    # -- layers ------------------------------------------------------------------------------------------------
    def add_condition(self, what, on=tuple(), param_groups=tuple(), conditioner_type=None, transformer_type=None,
                      transformer_kwargs=None, **conditioner_kwargs):
        """Coupling layer: ``what`` is transformed, conditioned on ``on``."""
        on, what = _tuple(on), _tuple(what)
        if len(on) == 0:
            raise ValueError("Need to condition on something.")
        if len(what) == 0:
            raise ValueError("Need to transform something.")
        if transformer_type is None:
            transformer_type = _common([self.transformer_type.get(f, self.default_transformer_type) for f in what],
                                       what, "transformer_type")
        t_kwargs = _common([{**self.transformer_kwargs.get(f, self.default_transformer_kwargs), **(transformer_kwargs or {})}
                            for f in what], what, "transformer_kwargs")
        if conditioner_type is None:
            conditioner_type = _common([self.conditioner_type.get(f, self.default_conditioner_type) for f in what],
                                       what, "conditioner_type")
        c_kwargs = _common([{**self.conditioner_kwargs.get(f, self.default_conditioner_kwargs), **conditioner_kwargs}
                            for f in what], what, "conditioner_kwargs")
        conditioners = make_conditioners(transformer_type=transformer_type, conditioner_type=conditioner_type,
                                         transformer_kwargs=t_kwargs, what=what, on=on, shape_info=self.current_dims.copy(),
                                         **c_kwargs)
        transformer = make_transformer(transformer_type=transformer_type, what=what, shape_info=self.current_dims,
                                       conditioners=conditioners, **t_kwargs)
        coupling = CouplingFlow(transformer=transformer,
                                transformed_indices=[self.current_dims.index(f) for f in what],
                                cond_indices=[self.current_dims.index(f) for f in on]).to(**self.ctx)
        self.add_layer(coupling, param_groups=param_groups)

    def add_set_constant(self, what, tensor):
        if what in self.current_dims:
            if self.current_dims[what] != tuple(tensor.shape):
                raise ValueError(f"Constant tensor {tensor} must have shape {self.current_dims[what]}")
        elif what in self.prior_dims:
            raise ValueError(f"Cannot set {what} constant; field was already deleted or replaced.")
        else:
            self.current_dims[what] = tuple(tensor.shape)
        self.layers.append(SetConstantFlow(indices=[self.current_dims.index(what)], values=[tensor.to(**self.ctx)]))

    def add_layer(self, flow, what=None, inverse=False, param_groups=tuple()):
        """Any Flow that keeps the shapes of the tensors it touches; ``what`` restricts it to some fields."""
        if inverse:
            flow = InverseFlow(flow)
        if what is not None:
            slots = [self.current_dims.index(f) for f in _tuple(what)]
            flow = WrapFlow(flow, slots, slots)
        self._add_to_param_groups(flow.parameters(), param_groups)
        self.layers.append(flow)

    def add_split(self, what, into, sizes_or_indices, dim=-1):
        into = [TensorInfo(name=f, is_circular=what.is_circular) if isinstance(f, str) else f for f in into]
        slot = self.current_dims.index(what)
        split = SplitFlow(*sizes_or_indices, dim=dim)
        sizes = sizes_or_indices if split._sizes is not None else [len(ix) for ix in sizes_or_indices]
        self.current_dims.split(what, into, sizes, dim=dim)
        self.layers.append(WrapFlow(split, indices=(slot,), out_indices=[self.current_dims.index(f) for f in into]))
        return tuple(into)

    def add_merge(self, what, to, dim=-1, output_index=None, sizes_or_indices=None):
        if isinstance(to, str):
            to = TensorInfo(name=to, is_circular=what[0].is_circular)
        if any(f.is_circular != to.is_circular for f in what):
            raise ValueError(f"Merging non-circular with circular tensors is dangerous and therefore disabled. "
                             f"Found discrepancies in f{what} and f{to}.")
        slots = [self.current_dims.index(f) for f in what]
        if sizes_or_indices is None:
            sizes_or_indices = [self.current_dims[f][dim] for f in what]
        merge = MergeFlow(*sizes_or_indices, dim=dim)
        self.current_dims.merge(what, to=to, index=output_index)
        self.layers.append(WrapFlow(merge, indices=slots, out_indices=(self.current_dims.index(to),)))
        return to

    def add_map_to_cartesian(self, coordinate_transform, fixed_origin_and_rotation=True, bonds=BONDS, angles=ANGLES,
                             torsions=TORSIONS, fixed=FIXED, origin=ORIGIN, rotation=ROTATION, out=TARGET):
        fields = [bonds, angles, torsions]
        if isinstance(coordinate_transform, GlobalInternalCoordinateTransformation):
            fields += [origin, rotation]
            if fixed_origin_and_rotation:
                self.add_set_constant(origin, torch.zeros(1, 3, **self.ctx))
                self.add_set_constant(rotation, torch.tensor([0.5, 0.5, 0.5], **self.ctx))
        else:
            fields.append(fixed)
        slots = [self.current_dims.index(f) for f in fields]
        self.layers.append(WrapFlow(InverseFlow(coordinate_transform), indices=slots, out_indices=(min(slots),)))
        self.current_dims.merge(fields, out)

    def add_map_to_ic_domains(self, cdfs=None, return_layers=False):
        if not cdfs:
            cdfs = InternalCoordinateMarginals(self.current_dims, self.ctx)
        added = []
        for field, marginal in cdfs.items():
            if field not in self.current_dims:
                warnings.warn(f"Field {field} not in current dims. CDF is ignored.")
                continue
            icdf = marginal if isinstance(marginal, Flow) else InverseFlow(CDFTransform(marginal))
            self.layers.append(WrapFlow(icdf, (self.current_dims.index(field),)))
            added.append(icdf)
        if return_layers:
            return added

    def add_merge_constraints(self, *args, **kwargs):
        raise NotImplementedError("constraint merging (generator_builder.py:461-498) is outside the accelerated hot path")

    def add_constrain_chirality(self, *args, **kwargs):
        raise NotImplementedError("chirality constraints (generator_builder.py:500-516) are outside the accelerated hot path")

    def add_torsion_multiplicities(self, *args, **kwargs):
        raise NotImplementedError("torsion multiplicities (generator_builder.py:518-521) are outside the accelerated hot path")

    def add_torsion_shifts(self, *args, **kwargs):
        raise NotImplementedError("torsion shifts (generator_builder.py:523-526) are outside the accelerated hot path")

    def _add_to_param_groups(self, parameters, param_groups):
        parameters = list(parameters)
        for group in param_groups:
            self.param_groups.setdefault(group, []).extend(parameters)
