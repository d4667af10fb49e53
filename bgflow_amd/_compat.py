"""Dotted import paths of the reference package, mapped onto this package's flat modules.

bgflow code often imports through sub-packages (``from bgflow.nn.flow.crd_transform.ic import ...``,
``from bgflow.factory.tensor_info import BONDS``).  ``register(root)`` installs lightweight alias modules under
``<root>.nn...``, ``<root>.distribution...``, ``<root>.factory...`` so that such imports resolve to the accelerated classes
when this package stands in for bgflow (``import bgflow_amd as bgflow`` or ``sys.modules["bgflow"] = bgflow_amd``).
Only the modules on the accelerated hot path are mapped; anything else raises ImportError as before.
"""
import sys
import types

# reference module path (relative to the package root) -> name of the flat module here that holds its public names
_MAP = {
    "nn": None,
    "nn.dense": "dense",
    "nn.periodic": "dense",
    "nn.flow": "flow",
    "nn.flow.base": "flow",
    "nn.flow.sequential": "flow",
    "nn.flow.coupling": "flow",
    "nn.flow.inverted": "flow",
    "nn.flow.cdf": "cdf",
    "nn.flow.stochastic": "flow",
    "nn.flow.stochastic.augment": "flow",
    "nn.flow.transformer": "transformer",
    "nn.flow.transformer.base": "transformer",
    "nn.flow.transformer.affine": "transformer",
    "nn.flow.transformer.spline": "transformer",
    "nn.flow.crd_transform": "ic",
    "nn.flow.crd_transform.ic": "ic",
    "nn.flow.crd_transform.pca": "ic",
    "distribution": "distributions",
    "distribution.distributions": "distributions",
    "distribution.normal": "distributions",
    "distribution.product": "distributions",
    "distribution.energy": "distributions",
    "distribution.energy.base": "distributions",
    "distribution.energy.double_well": "distributions",
    "distribution.sampling": "distributions",
    "distribution.sampling.base": "distributions",
    "distribution.sampling.dataset": "training",
    "nn.training": "training",
    "nn.training.trainers": "training",
    "factory.tensor_info": "factory",
    "factory.generator_builder": "factory",
    "factory.conditioner_factory": "factory",
    "factory.transformer_factory": "factory",
    "factory.distribution_factory": "factory",
    "factory.icmarginals": "factory",
    "utils.types": "utils",
}


def register(root, package):
    """Install the alias modules under the dotted name ``root`` (``package`` = the imported bgflow_amd package)."""
    for rel, flat in sorted(_MAP.items(), key=lambda kv: kv[0].count(".")):
        full = f"{root}.{rel}"
        if full in sys.modules:
            continue
        mod = types.ModuleType(full, f"alias of {package.__name__}.{flat}" if flat else "namespace")
        mod.__path__ = []          # behaves like a package: sub-imports go through sys.modules
        if flat is not None:
            src = getattr(package, flat)
            names = getattr(src, "__all__", None) or [n for n in vars(src) if not n.startswith("__")]
            for n in names:
                setattr(mod, n, getattr(src, n))
            for n in vars(src):        # private helpers a few reference call sites reach for (e.g. _tuple)
                if n.startswith("_") and not n.startswith("__") and not hasattr(mod, n):
                    setattr(mod, n, getattr(src, n))
        sys.modules[full] = mod
        parent, _, leaf = full.rpartition(".")
        if parent in sys.modules and not hasattr(sys.modules[parent], leaf):
            setattr(sys.modules[parent], leaf, mod)
