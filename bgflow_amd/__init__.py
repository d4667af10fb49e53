"""bgflow_amd -- MI355X-native coupling-flow engine with bgflow's API for the hot path.

``import bgflow_amd as bg`` gives the bgflow names for the coupling-flow forward/inverse +
log|det J| path (Flow protocol, SequentialFlow, CouplingFlow, Split/Merge/Swap/Wrap/SetConstant,
InverseFlow, AffineTransformer, ConditionalSplineTransformer, DenseNet, WrapPeriodic,
Relative/MixedCoordinateTransformation, WhitenFlow, BoltzmannGenerator) and the priors / targets
needed around it.  The arithmetic of transformers and coordinate transforms runs in hand-written
HIP kernels for gfx950 (libbgflow_amd.so, C ABI in include/bgflow_amd.h); there is no CPU path.
"""
from .flow import *          # noqa: F401,F403
from .transformer import *   # noqa: F401,F403
from .dense import *         # noqa: F401,F403
from .ic import *            # noqa: F401,F403
from .distributions import * # noqa: F401,F403
from .cdf import *           # noqa: F401,F403
from .bg import *            # noqa: F401,F403
from .factory import *       # noqa: F401,F403
from .training import *      # noqa: F401,F403
from . import configs, dp, factory, training, utils      # noqa: F401

__version__ = "0.1.0"

# reference-style dotted import paths (bgflow_amd.nn.flow.crd_transform.ic, bgflow_amd.factory.tensor_info, ...)
from . import _compat as _compat_mod   # noqa: E402
import sys as _sys                     # noqa: E402
_compat_mod.register(__name__, _sys.modules[__name__])

