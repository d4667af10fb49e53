"""Round-5 GPU tests (marked gpu): regressions for the advisor's round-4 findings, the any-bin-count spline backward, the f32-class
backward GEMMs, the single-pass KL evaluation and the segment-level training backward."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _spline_layer(dev, what="TORSIONS", on="FIXED", hidden=(128, 128), **kw):
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_
    dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
    circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
    return hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden, **kw)).to(dev)


def _fields(dev, B, seed=5, grad=True):
    return [torch.rand(B, d, device=dev, generator=torch.Generator(device=dev).manual_seed(seed + i)).requires_grad_(grad)
            for i, d in enumerate((17, 17, 17, 9))]


@pytest.mark.parametrize("hidden", [(64, 64), (32, 96)])
def test_narrow_hidden_layers_over_two_optimizer_steps(hip_lib, dev, hidden):
    """advisor (round 4, high): the backward of a zero-padded narrow conditioner keyed its transposed operands on the padded
    TEMPORARIES -- fresh F.pad outputs whose (data_ptr, version) repeats once the caching allocator recycles the storage, so the
    second step could run its input-gradient chain on operands packed from the first step's weights.  Two steps with the graph
    freed in between (the allocator then hands the same addresses back), every step's gradients against the layer-by-layer path on
    the same weights."""
    from bgflow_amd import dense
    layer = _spline_layer(dev, hidden=hidden)
    opt = torch.optim.SGD(layer.parameters(), lr=0.05)
    B = 2048

    def step_grads(fused):
        layer.transformer.allow_fused = fused
        layer.zero_grad()
        xs = _fields(dev, B)
        *out, dl = layer(*xs)
        (sum((o * o).sum() for o in out) + dl.sum()).backward()
        grads = [p.grad.clone() for p in layer.parameters()] + [x.grad.clone() for x in xs if x.grad is not None]
        del out, dl, xs
        layer.transformer.allow_fused = True
        return grads

    for it in range(3):
        g_fused = step_grads(True)
        assert layer.transformer._fused_cache.get("padded"), "the zero-padded fused training path must have run"
        g_ref = step_grads(False)
        for a, b in zip(g_fused, g_ref):
            assert float((a - b).abs().max()) <= 2e-3 * max(float(b.abs().max()), 1e-6), f"step {it}: gradient of shape {tuple(a.shape)}"
        layer.zero_grad()
        for p, g in zip(layer.parameters(), g_ref):
            p.grad = g.clone()
        opt.step()                      # large step: stale operands would be off by far more than the tolerance
        torch.cuda.empty_cache() if it == 1 else None


def test_flat_adam_with_a_conditioner_input_wider_than_the_t_operands(hip_lib, dev):
    """advisor (round 4, medium): FlatAdam.step -> repack_training_plans called bgk_pack_dense_h2_t_many for EVERY fused training
    plan; a conditioner with 97..127 input features (fused forward, GEMM backward) made it fail with BGK_EINVAL on the first update"""
    import bgflow_amd as bg
    from bgflow_amd.training import FlatAdam
    from bgflow_amd.utils import hash_init_
    d_c, d, K = 100, 8, 8
    net = bg.DenseNet([d_c, 128, 128, 3 * K * d + d], activation=torch.nn.SiLU())
    layer = hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(net, is_circular=False))).to(dev)
    opt = FlatAdam(list(layer.parameters()), lr=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        x = torch.rand(512, d_c, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        y = torch.rand(512, d, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
        _, out, dl = layer(x, y)
        loss = (out * out).sum() - dl.sum()
        opt.backward(loss)
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[2] < losses[0]
    assert opt.skipped_steps() == 0


def test_flat_adam_bumps_version_counters(hip_lib, dev):
    """advisor (round 4, low): the fused Adam kernel writes through the flat bucket; torch's version counters must see it (caches
    keyed on ``_version``, autograd's saved-tensor check of a graph retained across the step)"""
    from bgflow_amd.training import FlatAdam
    lin = torch.nn.Linear(8, 8).to(dev)
    opt = FlatAdam(list(lin.parameters()), lr=1e-2)
    v0 = [p._version for p in lin.parameters()]
    x = torch.rand(4, 8, device=dev, requires_grad=True)         # the weight is saved for the input gradient
    loss = (lin(x) ** 2).sum()
    loss.backward(retain_graph=True)
    opt.step()
    assert all(p._version > v for p, v in zip(lin.parameters(), v0))
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()


def test_philox_stream_ids_are_stable_when_given_and_collisions_warn(hip_lib, dev):
    """advisor (round 4, low): with ``set_philox_stream`` the numbers of a prior do not depend on how many other fused-sampling
    objects sampled before; two live objects on one stream warn"""
    import bgflow_amd as bg
    torch.manual_seed(11)
    a = bg.NormalDistribution(9, sample_fused=True).to(dev).set_philox_stream(1000)
    za = a.sample(257)
    for _ in range(3):                                     # other objects come and sample in between
        bg.NormalDistribution(5, sample_fused=True).to(dev).sample(3)
    del a
    torch.manual_seed(11)
    b = bg.NormalDistribution(9, sample_fused=True).to(dev).set_philox_stream(1000)
    assert torch.equal(b.sample(257), za)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        c = bg.NormalDistribution(9, sample_fused=True).to(dev).set_philox_stream(1000)
    assert any("already used by a live" in str(w.message) for w in rec)
    auto = bg.NormalDistribution(9, sample_fused=True).to(dev)
    auto.sample(1)
    assert auto._philox_state[0] != 1000                   # automatic ids skip the ones taken by hand
    del b, c
