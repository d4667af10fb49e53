"""CPU (-m "not gpu"), round 6: host logic around the affine training path, the oracle's new pieces and the advisor fixes."""
import copy

import numpy as np
import torch


def test_affine_training_path_declines_cpu_tensors_without_touching_the_library():
    """the fused affine training forward is a HIP path: on CPU tensors the dispatcher answers None (the caller's generic path then
    raises the package's 'no CPU fallback' error at its first kernel)"""
    import bgflow_amd as bg
    from bgflow_amd import dense
    tr = bg.AffineTransformer(shift_transformation=bg.DenseNet([4, 8, 8, 3], activation=torch.nn.ReLU()),
                              scale_transformation=bg.DenseNet([4, 8, 8, 3], activation=torch.nn.Tanh()))
    x, y = torch.randn(5, 4), torch.randn(5, 3, requires_grad=True)
    assert dense.fused_affine_coupling_train(tr, x, y, False) is None


def test_affine_training_envelope_rules():
    """which conditioner networks the training kernels take (two hidden layers of <= 128 units, biases, one of SiLU / ReLU / Tanh,
    WrapPeriodic over all inputs on [0, 1])"""
    import bgflow_amd as bg
    from bgflow_amd import dense
    ok = dense._affine_train_net(bg.DenseNet([6, 64, 48, 5], activation=torch.nn.Tanh()))
    assert ok is not None and ok[1] == 3 and ok[2] is False
    assert dense._affine_train_net(bg.DenseNet([6, 64, 5], activation=torch.nn.Tanh())) is None              # one hidden layer
    assert dense._affine_train_net(bg.DenseNet([6, 200, 64, 5], activation=torch.nn.Tanh())) is None         # too wide
    assert dense._affine_train_net(bg.DenseNet([6, 64, 64, 5], activation=torch.nn.ELU())) is None           # activation
    per = bg.WrapPeriodic(bg.DenseNet([6, 32, 32, 5], activation=torch.nn.SiLU()), indices=np.arange(3))
    assert dense._affine_train_net(per)[2] is True
    part = bg.WrapPeriodic(bg.DenseNet([5, 32, 32, 5], activation=torch.nn.SiLU()), indices=np.arange(2))
    assert dense._affine_train_net(part) is None                                                             # a subset periodic


def test_torch_oracle_tuple_plumbing_is_differentiable_and_equals_the_numpy_oracle():
    """oracle/torch_flow.py: SplitFlow / SwapFlow / MergeFlow with stock torch ops (round 6: the f64 KL gradient of cfg 2 passes through
    them) -- same values as the numpy oracle, exact inverse, gradients reach every parameter"""
    from bgflow_amd import configs
    from oracle import flow_oracle as fo
    from oracle import torch_flow as tfl
    gen = configs.make_affine8_generator().double()
    z = torch.randn(16, 64, dtype=torch.float64)
    xs, dl = tfl.run_flow(gen.flow, [z], grad=True)
    y64, d64 = fo.run_flow(gen.flow, [z.numpy()], dtype=np.float64)
    assert np.abs(y64[0] - xs[0].detach().numpy()).max() < 1e-13 and np.abs(d64 - dl.detach().numpy()).max() < 1e-13
    (gen._target.energy(xs[0]) - dl).mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in gen.flow.parameters())
    back, dlb = tfl.run_flow(gen.flow, [xs[0].detach()], inverse=True)
    assert float((back[0] - z).abs().max()) < 1e-12 and float((dlb + dl.detach()).abs().max()) < 1e-12


def test_oracle_gemm_rounding_hook_models_bf16_operands():
    """SPLINE_GEMM_ROUNDING: weights and layer inputs through bf16, layer 0's bias as ONE rounded value, later biases as hi + lo"""
    from oracle import torch_flow as tfl
    import bgflow_amd as bg
    net = bg.DenseNet([3, 4, 2], activation=torch.nn.SiLU()).double()
    x = torch.randn(7, 3, dtype=torch.float64)
    r = tfl.bf16_round
    l0, l1 = net._layers[0], net._layers[2]
    h = torch.nn.functional.silu(torch.nn.functional.linear(r(x), r(l0.weight), r(l0.bias)))
    want = torch.nn.functional.linear(r(h), r(l1.weight), r(l1.bias) + r(l1.bias - r(l1.bias)))
    got = tfl.conditioner(net, x, r)
    assert torch.equal(got, want)
    assert not torch.equal(got, tfl.conditioner(net, x))
    assert float((r(x) - x).abs().max()) > 0 and float((r(x) - x).abs().max()) <= 2.0 ** -8 * float(x.abs().max())


def test_deepcopy_of_a_fused_sampling_prior_does_not_inherit_its_philox_stream():
    """advisor finding (round 5): a deep copy carried `_philox_state` along and drew the SAME numbers as the original, silently"""
    import bgflow_amd as bg
    prior = bg.NormalDistribution(5)
    if not hasattr(prior, "set_philox_stream"):
        return
    prior.set_philox_stream(3, calls=7)
    clone = copy.deepcopy(prior)
    assert prior.__dict__.get("_philox_state") == [3, 7]
    assert "_philox_state" not in clone.__dict__
    assert clone.dim == prior.dim


def test_kl_sums_finishes_instead_of_asking_for_a_second_flow_evaluation():
    """advisor finding (round 5): SequentialFlow.kl_sums ran the flow and could still answer None (the caller then ran the whole flow
    again).  Its last exit now finishes with the per-sample form.  Checked on the exit itself with stand-ins (no kernels on CPU)."""
    import bgflow_amd.flow as fl
    src = open(fl.__file__).read()
    body = src[src.index("def kl_sums"):src.index("def run(self")]
    assert "return None if out is None else out[0]" not in body
    assert "per = target.energy(*x, temperature=temperature) - total" in body
