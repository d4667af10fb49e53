"""Stand-in for the third-party ``nflows`` package (bayesiains/nflows, unpinned in the
reference: .github/workflows/CI.yml:44, docs/requirements.yaml:8) -- TEST INFRASTRUCTURE ONLY.

nflows is not vendored under /root/reference and is not installed in this image, so the
reference's ``ConditionalSplineTransformer`` (bgflow/nn/flow/transformer/spline.py:75,129-144)
cannot import it.  This module restates the *public algorithm* of
``nflows.transforms.splines.rational_quadratic.rational_quadratic_spline`` and
``nflows.utils.torchutils.searchsorted`` in plain torch ops so that the
UNMODIFIED reference classes can run on top of it when golden vectors are generated
(tests/golden/make_goldens.py).  Steps 5-8 of Appendix A (bin search with the in-place
``+= eps``, gathers, root solve / rational evaluate, log-det) are corroborated bit-for-bit by the
reference's own in-tree copy bgflow/nn/flow/spline.py:121-188 (checked in make_goldens.py).

It is only ever installed into ``sys.modules`` by the golden generator; nothing in the product
package imports it.
"""
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3

# side channel so the golden generator can record bin indices / knots of the last call
LAST = {}


class InputOutsideDomain(Exception):
    """Exception to be thrown when the input to a transform is not within its domain."""
    pass


def searchsorted(bin_locations, inputs, eps=1e-6):
    bin_locations[..., -1] += eps
    return torch.sum(inputs[..., None] >= bin_locations, dim=-1) - 1


def rational_quadratic_spline(
    inputs,
    unnormalized_widths,
    unnormalized_heights,
    unnormalized_derivatives,
    inverse=False,
    left=0.0,
    right=1.0,
    bottom=0.0,
    top=1.0,
    min_bin_width=DEFAULT_MIN_BIN_WIDTH,
    min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
    min_derivative=DEFAULT_MIN_DERIVATIVE,
    enable_identity_init=False,
):
    if torch.min(inputs) < left or torch.max(inputs) > right:
        raise InputOutsideDomain()

    num_bins = unnormalized_widths.shape[-1]
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")

    widths = F.softmax(unnormalized_widths, dim=-1)
    widths = min_bin_width + (1 - min_bin_width * num_bins) * widths
    cumwidths = torch.cumsum(widths, dim=-1)
    cumwidths = F.pad(cumwidths, pad=(1, 0), mode="constant", value=0.0)
    cumwidths = (right - left) * cumwidths + left
    cumwidths[..., 0] = left
    cumwidths[..., -1] = right
    widths = cumwidths[..., 1:] - cumwidths[..., :-1]

    if enable_identity_init:
        beta = np.log(2) / (1 - min_derivative)
    else:
        beta = 1
    derivatives = min_derivative + F.softplus(unnormalized_derivatives, beta=beta)

    heights = F.softmax(unnormalized_heights, dim=-1)
    heights = min_bin_height + (1 - min_bin_height * num_bins) * heights
    cumheights = torch.cumsum(heights, dim=-1)
    cumheights = F.pad(cumheights, pad=(1, 0), mode="constant", value=0.0)
    cumheights = (top - bottom) * cumheights + bottom
    cumheights[..., 0] = bottom
    cumheights[..., -1] = top
    heights = cumheights[..., 1:] - cumheights[..., :-1]

    if inverse:
        bin_idx = searchsorted(cumheights, inputs)[..., None]
    else:
        bin_idx = searchsorted(cumwidths, inputs)[..., None]

    LAST.clear()
    LAST.update(
        bin_idx=bin_idx[..., 0].detach().clone(),
        cumwidths=cumwidths.detach().clone(),
        cumheights=cumheights.detach().clone(),
        derivatives=derivatives.detach().clone(),
        widths=widths.detach().clone(),
        heights=heights.detach().clone(),
    )

    input_cumwidths = cumwidths.gather(-1, bin_idx)[..., 0]
    input_bin_widths = widths.gather(-1, bin_idx)[..., 0]

    input_cumheights = cumheights.gather(-1, bin_idx)[..., 0]
    delta = heights / widths
    input_delta = delta.gather(-1, bin_idx)[..., 0]

    input_derivatives = derivatives.gather(-1, bin_idx)[..., 0]
    input_derivatives_plus_one = derivatives[..., 1:].gather(-1, bin_idx)[..., 0]

    input_heights = heights.gather(-1, bin_idx)[..., 0]

    if inverse:
        a = (inputs - input_cumheights) * (
            input_derivatives + input_derivatives_plus_one - 2 * input_delta
        ) + input_heights * (input_delta - input_derivatives)
        b = input_heights * input_derivatives - (inputs - input_cumheights) * (
            input_derivatives + input_derivatives_plus_one - 2 * input_delta
        )
        c = -input_delta * (inputs - input_cumheights)

        discriminant = b.pow(2) - 4 * a * c
        assert (discriminant >= 0).all()

        root = (2 * c) / (-b - torch.sqrt(discriminant))
        outputs = root * input_bin_widths + input_cumwidths

        theta_one_minus_theta = root * (1 - root)
        denominator = input_delta + (
            (input_derivatives + input_derivatives_plus_one - 2 * input_delta)
            * theta_one_minus_theta
        )
        derivative_numerator = input_delta.pow(2) * (
            input_derivatives_plus_one * root.pow(2)
            + 2 * input_delta * theta_one_minus_theta
            + input_derivatives * (1 - root).pow(2)
        )
        logabsdet = torch.log(derivative_numerator) - 2 * torch.log(denominator)
        return outputs, -logabsdet
    else:
        theta = (inputs - input_cumwidths) / input_bin_widths
        theta_one_minus_theta = theta * (1 - theta)

        numerator = input_heights * (
            input_delta * theta.pow(2) + input_derivatives * theta_one_minus_theta
        )
        denominator = input_delta + (
            (input_derivatives + input_derivatives_plus_one - 2 * input_delta)
            * theta_one_minus_theta
        )
        outputs = input_cumheights + numerator / denominator

        derivative_numerator = input_delta.pow(2) * (
            input_derivatives_plus_one * theta.pow(2)
            + 2 * input_delta * theta_one_minus_theta
            + input_derivatives * (1 - theta).pow(2)
        )
        logabsdet = torch.log(derivative_numerator) - 2 * torch.log(denominator)
        return outputs, logabsdet


def install():
    """Register this restatement as ``nflows`` in sys.modules (golden generation only)."""
    me = sys.modules[__name__]
    nflows = types.ModuleType("nflows")
    transforms = types.ModuleType("nflows.transforms")
    splines = types.ModuleType("nflows.transforms.splines")
    base = types.ModuleType("nflows.transforms.base")
    utils = types.ModuleType("nflows.utils")
    torchutils = types.ModuleType("nflows.utils.torchutils")
    splines.rational_quadratic_spline = rational_quadratic_spline
    base.InputOutsideDomain = InputOutsideDomain
    torchutils.searchsorted = searchsorted
    nflows.transforms = transforms
    nflows.utils = utils
    transforms.splines = splines
    transforms.base = base
    utils.torchutils = torchutils
    for name, mod in [
        ("nflows", nflows), ("nflows.transforms", transforms),
        ("nflows.transforms.splines", splines), ("nflows.transforms.base", base),
        ("nflows.utils", utils), ("nflows.utils.torchutils", torchutils),
    ]:
        sys.modules[name] = mod
    return me
