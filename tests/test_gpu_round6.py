"""GPU (-m gpu), round 6: affine couplings under autograd on the hand-written kernels -- the one-launch training forward
(bgk_coupling_affine_dense_h2_train), the tail backward (bgk_affine_backward), the conditioner networks' input-gradient chain
(bgk_dense_backward_dx) and weight gradients (bgk_mlp_weight_grad) -- against f64 autograd of the reference's op chain
(oracle/torch_flow.py; reference: nn/flow/transformer/affine.py:35-70, nn/dense.py:30-48, nn/flow/coupling.py:152-182,
nn/training/trainers.py:156-163).  All kernels are reached through the C ABI (ctypes, bgflow_amd/_lib.py)."""
import copy

import numpy as np
import pytest
import torch

from test_gpu_round3 import _make, _prior
from test_gpu_round4 import _grad_errors

pytestmark = pytest.mark.gpu


def _affine_layer(n_c, hidden, d, acts, periodic=False, seed_name="layer", **kw):
    """CouplingFlow(AffineTransformer(shift, scale)) over two tensors (slot 0 conditions, slot 1 is transformed), hash-initialised"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    n_in = 2 * n_c if periodic else n_c

    def net(act):
        n = bg.DenseNet([n_in, *hidden, d], activation=act())
        return bg.WrapPeriodic(n, indices=np.arange(n_c)) if periodic else n
    tr = bg.AffineTransformer(shift_transformation=net(acts[0]) if acts[0] else None,
                              scale_transformation=net(acts[1]) if acts[1] else None, **kw)
    layer = bg.CouplingFlow(tr, transformed_indices=[1], cond_indices=[0])
    return hash_init_(bg.SequentialFlow([layer]), scale=1.5)


def _f64_layer_grads(flow_cpu, x, y, wy, wl, inverse):
    """loss = sum(wy * y') + sum(wl * dlogp) through the reference's op chain in f64: gradients w.r.t. x, y and every parameter"""
    from oracle import torch_flow as tfl
    for p in flow_cpu.parameters():
        p.grad = None
    xs = [x.clone().requires_grad_(True), y.clone().requires_grad_(True)]
    outs, dl = tfl.run_flow(flow_cpu, xs, inverse=inverse, grad=True)
    loss = (outs[1] * wy).sum() + (dl * wl).sum()
    loss.backward()
    return outs[1].detach(), dl.detach(), xs[0].grad, xs[1].grad, {n: p.grad.clone() for n, p in flow_cpu.named_parameters() if p.grad is not None}


SHAPES = [
    # (n_c, hidden, d, (shift act, scale act), periodic, B)
    (32, (64, 64), 32, (torch.nn.ReLU, torch.nn.Tanh), False, 1000),          # cfg 2's couplings
    (17, (128, 128), 66, (torch.nn.SiLU, torch.nn.SiLU), True, 777),          # cfg 5: AUGMENTED | TORSIONS
    (43, (128, 128), 66, (torch.nn.SiLU, torch.nn.SiLU), False, 4133),        # cfg 5: AUGMENTED | (FIXED, BONDS, ANGLES), concatenated
    (5, (48, 96), 7, (torch.nn.Tanh, torch.nn.Tanh), False, 33),              # narrow, unequal hidden layers
    (9, (128, 128), 40, (None, torch.nn.SiLU), False, 65),                    # scale network only
    (9, (32, 32), 12, (torch.nn.ReLU, None), False, 31),                      # shift network only (NICE)
]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{s[0]}-{'x'.join(map(str, s[1]))}-{s[2]}" + ("-periodic" if s[4] else "") for s in SHAPES])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_coupling_training_layer_against_f64_autograd(hip_lib, dev, shape, inverse):
    """One affine coupling under autograd: forward values equal the inference kernel's; the gradients w.r.t. the conditioning tensor,
    the transformed tensor, log_alpha and all twelve weight / bias tensors are those of f64 autograd through the reference's op chain
    (relative L2 of the flat gradient <= 5e-5: the bound of the spline layers' KL gradient, tests/test_gpu_round4.py)."""
    from bgflow_amd import dense
    n_c, hidden, d, acts, periodic, B = shape
    flow = _affine_layer(n_c, hidden, d, acts, periodic).to(dev)
    flow_cpu = copy.deepcopy(flow).cpu().double()
    g = torch.Generator().manual_seed(B + d)
    x = torch.rand(B, n_c, generator=g, dtype=torch.float64) if periodic else torch.randn(B, n_c, generator=g, dtype=torch.float64)
    y = torch.randn(B, d, generator=g, dtype=torch.float64)
    wy = torch.randn(B, d, generator=g, dtype=torch.float64) / B
    wl = torch.randn(B, 1, generator=g, dtype=torch.float64) / B
    ref_out, ref_dl, ref_gx, ref_gy, ref_gp = _f64_layer_grads(flow_cpu, x, y, wy, wl, inverse)

    xg, yg = x.float().to(dev).requires_grad_(True), y.float().to(dev).requires_grad_(True)
    tr = flow[0].transformer
    _, out, dl = flow(xg, yg, inverse=inverse)
    assert "_train_cache" in tr.__dict__ and tr._train_cache.get("train_used"), "the layer did not take the fused training path"
    with torch.no_grad():
        _, out_inf, dl_inf = flow(xg.detach(), yg.detach(), inverse=inverse)
    if hidden[0] == hidden[1]:       # (unequal hidden widths: inference runs layer by layer, another summation order)
        assert torch.equal(out, out_inf) and torch.equal(dl, dl_inf), "training forward and inference kernel disagree"
    else:
        assert torch.allclose(out, out_inf, rtol=1e-5, atol=1e-5) and torch.allclose(dl, dl_inf, rtol=1e-5, atol=1e-5)
    assert float((out.double().cpu() - ref_out).abs().max()) <= 2e-5 * max(1.0, float(ref_out.abs().max()))
    assert float((dl.double().cpu() - ref_dl).abs().max()) <= 1e-5 * max(1.0, float(ref_dl.abs().max()))
    ((out * wy.float().to(dev)).sum() + (dl * wl.float().to(dev)).sum()).backward()
    got = {n: p.grad.double().cpu() for n, p in flow.named_parameters() if p.grad is not None}
    assert set(got) == set(ref_gp)
    rel, worst = _grad_errors(got, {n: ref_gp[n] for n in got})
    print(f"affine training layer {shape[:3]} inverse={inverse}: flat parameter gradient rel L2 {rel:.2e}, worst {worst[1]} {worst[0]:.2e}")
    assert rel <= 5e-5 and worst[0] <= 3e-4
    for name, a, b in (("g_x", xg.grad, ref_gx), ("g_y", yg.grad, ref_gy)):
        err = float((a.double().cpu() - b).norm() / max(float(b.norm()), 1e-30))
        assert err <= 5e-5, f"{name}: relative L2 {err:.2e}"
    assert dense.AFFINE_TRAIN


def test_affine_training_layer_as_a_library_path_switch(hip_lib, dev):
    """AFFINE_TRAIN = False is the layer-by-layer path (the A/B leg): same gradients within the two paths' tolerances"""
    from bgflow_amd import dense
    flow = _affine_layer(32, (64, 64), 32, (torch.nn.ReLU, torch.nn.Tanh)).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    x, y = torch.randn(512, 32, device=dev, generator=g), torch.randn(512, 32, device=dev, generator=g)

    def grads():
        for p in flow.parameters():
            p.grad = None
        _, out, dl = flow(x, y)
        (out.square().mean() - dl.mean()).backward()
        return {n: p.grad.clone() for n, p in flow.named_parameters()}
    fused = grads()
    try:
        dense.AFFINE_TRAIN = False
        plain = grads()
    finally:
        dense.AFFINE_TRAIN = True
    rel, worst = _grad_errors(fused, plain)
    assert rel <= 1e-4, f"fused vs layer-by-layer gradients: {rel:.2e} ({worst})"


def test_affine_backward_after_a_parameter_update_raises(hip_lib, dev):
    """the packed operands are rewritten in place when the weights change: a backward through a graph built on the old ones refuses"""
    flow = _affine_layer(9, (32, 32), 12, (torch.nn.ReLU, torch.nn.Tanh)).to(dev)
    x, y = torch.randn(64, 9, device=dev), torch.randn(64, 12, device=dev)
    _, out, dl = flow(x, y)
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(1.5)
    _, out2, _ = flow(x, y)           # re-packs the plan's buffers in place
    with pytest.raises(RuntimeError, match="changed between the forward and this backward|modified by an inplace operation"):
        (out.sum() + dl.sum()).backward()
    out2.sum().backward()


def _kl_gradient_cfg2_gpu(gen, z):
    from bgflow_amd import dp
    for p in gen.flow.parameters():
        p.grad = None
    *x, dlogp = gen.flow(*z)
    loss = dp.global_kl_mean(gen._target, x, dlogp)
    loss.backward()
    return {n: p.grad.detach().cpu().double() for n, p in gen.flow.named_parameters()}, float(loss.detach())


def test_kl_gradient_of_cfg2_at_two_to_the_18_against_f64(hip_lib, dev):
    """BASELINE cfg 2 (8 affine couplings, dim 64, DoubleWellEnergy): the flat KL gradient over 2^18 samples in ONE pass == the mean of
    the gradients of its 32 chunks (GPU), and the first / last chunk's gradient against f64 autograd of the reference's op chain:
    relative L2 <= 5e-5 (the verdict's bound), every tensor within 3e-4 of its own norm."""
    from oracle import torch_flow as tfl
    B, n_chunks = 1 << 18, 32
    gen, gen_cpu = _make("cfg2", dev), _make("cfg2").double()
    z = _prior("cfg2", B, dev)
    full, loss_full = _kl_gradient_cfg2_gpu(gen, z)
    assert all(torch.isfinite(v).all() for v in full.values())
    assert all("_train_cache" in b.transformer.__dict__ for b in gen.flow if hasattr(b, "transformer"))
    step = B // n_chunks
    acc, chunk_grads = None, {}
    for c in range(n_chunks):
        gc, _ = _kl_gradient_cfg2_gpu(gen, [z[0][c * step:(c + 1) * step]])
        acc = gc if acc is None else {n: acc[n] + gc[n] for n in gc}
        if c in (0, n_chunks - 1):
            chunk_grads[c] = gc
    report = [("one pass over 2^18 samples vs the mean of 32 chunk gradients", *_grad_errors(full, {n: v / n_chunks for n, v in acc.items()}))]
    target = gen_cpu._target
    for c, gc in chunk_grads.items():
        for p in gen_cpu.flow.parameters():
            p.grad = None
        zc = z[0][c * step:(c + 1) * step].cpu().double()
        xs, dl = tfl.run_flow(gen_cpu.flow, [zc], grad=True)
        (target.energy(xs[0]) - dl).mean().backward()
        ref = {n: p.grad.clone() for n, p in gen_cpu.flow.named_parameters()}
        report.append((f"chunk {c} vs the f64 reference", *_grad_errors(gc, ref)))
    for what, rel, worst in report:
        print(f"cfg 2 KL gradient, {what}: relative L2 {rel:.2e}, worst tensor {worst[1]} {worst[0]:.2e} of its norm")
    for what, rel, worst in report:
        assert rel <= 5e-5 and worst[0] <= 3e-4, f"{what}: rel L2 {rel:.2e}, {worst[1]} {worst[0]:.2e}"


def _device_kernel_names(step):
    from torch.profiler import ProfilerActivity, profile
    step(); step()
    torch.cuda.synchronize()
    try:
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            step()
            torch.cuda.synchronize()
    except Exception as e:          # no kernel tracer on this box
        pytest.skip(f"torch.profiler unavailable: {e!r}")
    names = [e.key for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA for _ in range(e.count)]
    if not names:
        pytest.skip("torch.profiler recorded no device kernels")
    return names


@pytest.mark.parametrize("cfg", ["cfg2", "cfg5"])
def test_kl_step_of_affine_flows_launches_no_library_gemm(hip_lib, dev, cfg):
    """A KL training step of BASELINE cfg 2 (8 affine couplings) and cfg 5 (10 spline + 6 affine): no hipBLASLt / rocBLAS GEMM
    (`Cijk_*`) and no aten activation kernel (silu / tanh / threshold / relu and their backward forms) between the first forward
    launch and the optimizer -- the conditioners' forward, input-gradient chain and weight gradients are the hand-written kernels."""
    from bgflow_amd import dp
    from bgflow_amd.training import FlatAdam
    gen = _make(cfg, dev)
    opt = FlatAdam(list(gen.flow.parameters()), lr=1e-5)
    z = _prior(cfg, 4096, dev)

    def step():
        opt.zero_grad()
        *x, dlogp = gen.flow(*z)
        loss = dp.global_kl_mean(gen._target, x, dlogp, drop_nonfinite=True)
        opt.backward(loss)
        opt.step()
    names = _device_kernel_names(step)
    bad = [n for n in names if "Cijk_" in n or "gemm" in n.lower() or any(k in n.lower() for k in ("silu", "tanh", "threshold", "relu"))]
    # (cfg 2's 64-unit networks: the kernel sized for them; cfg 5's 128-unit ones: the width-128 kernel)
    ours = [n for n in names if "coupling_affine_dense_v2_train_kernel" in n or "coupling_affine_fwd64_train_kernel" in n]
    from collections import Counter
    print(f"{cfg}: {len(names)} device kernels in one KL step;", dict(Counter(n.split("<")[0].split("(")[0][-60:] for n in names)))
    assert not bad, f"library / aten kernels in a {cfg} KL step: {sorted(set(bad))}"
    assert len(ours) == (8 if cfg == "cfg2" else 6), f"{len(ours)} launches of the affine training forward"


def test_kl_training_of_cfg2_reduces_the_loss(hip_lib, dev):
    """KLTrainer on cfg 2 (DoubleWellEnergy, 8 affine couplings): 30 Adam steps on the fused path lower the reverse KL"""
    from bgflow_amd.training import KLTrainer, FlatAdam
    gen = _make("cfg2", dev)
    opt = FlatAdam([p for p in gen.parameters() if p.requires_grad], lr=2e-3)
    trainer = KLTrainer(gen, optim=opt, train_likelihood=False, train_energy=True)
    torch.manual_seed(0)
    trainer.train(30, batchsize=4096)
    _, _, ys = trainer.losses()
    kll = ys[0]
    assert np.isfinite(kll).all() and kll[-5:].mean() < kll[:5].mean() - 0.5, f"KL loss {kll[:5].mean():.3f} -> {kll[-5:].mean():.3f}"
    assert opt.skipped_steps() == 0


def test_bf16_leg_against_the_rounding_oracle(hip_lib, dev):
    """BASELINE cfg 5's reduced-precision leg (gemm_mode "bf16": bf16 weights + GEMM inputs in the 10 spline conditioners, f32 accumulate;
    splines, log-det, affine layers as in the f32-class mode) against an ORACLE of that arithmetic: the reference's op chain in f64 with
    the spline conditioners' weights and layer inputs rounded to bf16 (oracle/torch_flow.py::SPLINE_GEMM_ROUNDING).  Rows of a 2^20-sample
    launch through the 16 couplings (the icdf maps behind them put every f32 evaluation of cfg 5 ~1e-3 from f64: they would hide the
    conditioners' arithmetic).  The kernel sits far closer to the rounding oracle than to the unrounded f64 evaluation -- whose distance
    is the price of the mode -- except where a pre-activation's f32 round-off flips a bf16 rounding."""
    from bgflow_amd import dense
    from oracle import torch_flow as tfl
    from test_gpu_round3 import _rows
    B = 1 << 20
    gen, gen_cpu = _make("cfg5", dev), _make("cfg5").double()
    sub, sub_cpu = gen.flow[:16], gen_cpu.flow[:16]
    z = _prior("cfg5", B, dev)
    rows = _rows(B, n=1024)
    rows_t = torch.as_tensor(rows, device=dev)
    try:
        dense.GEMM_MODE = "bf16"
        with torch.no_grad():
            *x, dl = sub(*z)
    finally:
        dense.GEMM_MODE = "f16x2"
    zr = [v[rows_t].cpu().double() for v in z]
    try:
        tfl.SPLINE_GEMM_ROUNDING = tfl.bf16_round
        xr, dlr = tfl.run_flow(sub_cpu, zr)
    finally:
        tfl.SPLINE_GEMM_ROUNDING = None
    xe, dle = tfl.run_flow(sub_cpu, zr)
    got_dl = dl[rows_t].cpu().double()
    e_round = ((got_dl - dlr).abs() / dlr.abs().clamp(min=1.0)).reshape(-1)
    e_exact = ((got_dl - dle).abs() / dle.abs().clamp(min=1.0)).reshape(-1)
    ex_round = torch.stack([(a[rows_t].cpu().double() - b).abs().max(-1).values for a, b in zip(x, xr)]).max(0).values
    ex_exact = torch.stack([(a[rows_t].cpu().double() - b).abs().max(-1).values for a, b in zip(x, xe)]).max(0).values
    print(f"bf16 leg, 16 couplings, log-det: median error vs the rounding oracle {float(e_round.median()):.2e}, vs unrounded f64 {float(e_exact.median()):.2e}; "
          f"fields: {float(ex_round.median()):.2e} vs {float(ex_exact.median()):.2e}; rows beyond 1e-4 of the rounding oracle: {float((e_round > 1e-4).float().mean()):.3f}")
    assert float(e_round.median()) <= 0.2 * float(e_exact.median()), "the bf16 kernel is no closer to its rounding oracle than to the exact evaluation"
    assert float(ex_round.median()) <= 0.2 * float(ex_exact.median())


def test_densenet_follows_weight_edits_torch_does_not_version(hip_lib, dev):
    """advisor finding (round 5): the packed operands of the per-layer kernel were cached on (data_ptr, _version); `p.data.mul_(2)` (no version
    bump) left them stale.  Now the operands follow the weights on the device (bgk_refresh_linear_layer): the forward changes."""
    import bgflow_amd as bg
    net = bg.DenseNet([7, 300, 19], activation=torch.nn.Tanh()).to(dev)
    x = torch.randn(50, 7, device=dev)
    with torch.no_grad():
        y0 = net(x)
        ref0 = torch.nn.functional.linear(torch.tanh(torch.nn.functional.linear(x, net._layers[0].weight, net._layers[0].bias)),
                                          net._layers[2].weight, net._layers[2].bias)
        assert torch.allclose(y0, ref0, rtol=1e-5, atol=1e-5)
        v0 = net._layers[2].weight._version
        net._layers[2].weight.data.mul_(2.0)
        assert net._layers[2].weight._version == v0          # torch saw nothing
        y1 = net(x)
        ref1 = torch.nn.functional.linear(torch.tanh(torch.nn.functional.linear(x, net._layers[0].weight, net._layers[0].bias)),
                                          net._layers[2].weight, net._layers[2].bias)
        assert torch.allclose(y1, ref1, rtol=1e-5, atol=1e-5) and not torch.allclose(y1, y0)
        y2 = net(x)
        assert torch.equal(y1, y2)                           # unchanged weights: same operands


def test_linear_with_forward_hooks_runs_as_the_module(hip_lib, dev):
    """a Linear (or its activation) carrying forward hooks is not swallowed by the layer kernel"""
    import bgflow_amd as bg
    net = bg.DenseNet([5, 16, 3], activation=torch.nn.ReLU()).to(dev)
    seen = []
    h1 = net._layers[0].register_forward_hook(lambda m, i, o: seen.append(("lin", tuple(o.shape))))
    h2 = net._layers[1].register_forward_hook(lambda m, i, o: seen.append(("act", tuple(o.shape))))
    with torch.no_grad():
        net(torch.randn(4, 5, device=dev))
    h1.remove(); h2.remove()
    assert seen == [("lin", (4, 16)), ("act", (4, 16))]


@pytest.mark.parametrize("units,act", [([17, 256, 256, 425], torch.nn.SiLU), ([9, 32, 64, 32, 40], torch.nn.Tanh), ([300, 520, 7], torch.nn.ReLU)])
def test_linear_backward_kernels_against_f64_autograd(hip_lib, dev, units, act):
    """A DenseNet outside the fused training envelopes (wide: 256 hidden units; deep: three hidden layers; a 520-unit layer fed 300 inputs)
    under autograd: every Linear's forward, input gradient (bgk_dense_layer on the operands of W^T) and weight / bias gradient
    (bgk_linear_weight_grad) and every activation's VJP (bgk_activation_backward) against f64 autograd of the same network; the loss
    scale of a KL step (gradients ~ 1 / B)."""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    net = hash_init_(bg.DenseNet(units, activation=act())).to(dev)
    net64 = copy.deepcopy(net).cpu().double()
    B = 3001
    g = torch.Generator().manual_seed(units[0])
    x = torch.randn(B, units[0], generator=g, dtype=torch.float64)
    w = torch.randn(B, units[-1], generator=g, dtype=torch.float64) / B
    x64 = x.clone().requires_grad_(True)
    y64 = x64
    for m in net64._layers:
        y64 = m(y64)
    (y64 * w).sum().backward()
    xg = x.float().to(dev).requires_grad_(True)
    y = net(xg)
    assert float((y.double().cpu() - y64.detach()).abs().max()) <= 1e-5 * max(1.0, float(y64.abs().max()))
    (y * w.float().to(dev)).sum().backward()
    got = {n: p.grad.double().cpu() for n, p in net.named_parameters()}
    ref = {n: p.grad for n, p in net64.named_parameters()}
    rel, worst = _grad_errors(got, ref)
    ex = float((xg.grad.double().cpu() - x64.grad).norm() / x64.grad.norm())
    print(f"DenseNet {units}: parameter gradient rel L2 {rel:.2e} (worst {worst[1]} {worst[0]:.2e}), input gradient {ex:.2e}")
    assert rel <= 2e-5 and worst[0] <= 1e-4 and ex <= 2e-5


def test_training_a_wide_conditioner_launches_no_library_gemm(hip_lib, dev):
    """a spline coupling whose conditioner has 256-unit hidden layers (fused in inference only) under autograd: layer by
    layer on bgk_dense_layer / bgk_activation forward, bgk_rqs_backward + the per-Linear backward kernels backward -- no `Cijk_*` kernel"""
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_
    dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
    circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
    layer = hash_init_(configs._spline_coupling("BONDS", "ANGLES", dims, circ, slot, hidden=(256, 256))).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.rand(2048, d, device=dev, generator=g) for d in (17, 17, 17, 9)]

    def step():
        for p in layer.parameters():
            p.grad = None
        *ys, dl = layer(*xs)
        (ys[0].square().sum() - dl.sum()).backward()
    names = _device_kernel_names(step)
    bad = [n for n in names if "Cijk_" in n or "gemm" in n.lower() or any(k in n.lower() for k in ("silu", "tanh", "threshold"))]
    assert not bad, sorted(set(bad))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())


def test_small_network_kernels_equal_the_general_path(hip_lib, dev):
    """cfg 2's coupling shape takes the kernels sized for 64-unit networks (bgk_coupling_affine_dense_fwd64_train,
    bgk_affine_net_backward64); with them switched off the same layer runs the width-128 training forward and the three-kernel
    backward: identical forward values (same products, same order), gradients equal to accumulation-order noise; partial tiles and a
    strided conditioning / transformed pair (the two halves of one [B, 64] tensor, as SplitFlow hands them over)."""
    from bgflow_amd import dense
    flow = _affine_layer(32, (64, 64), 32, (torch.nn.ReLU, torch.nn.Tanh)).to(dev)
    g = torch.Generator(device=dev).manual_seed(11)
    full = torch.randn(4133, 64, device=dev, generator=g)

    def run():
        for p in flow.parameters():
            p.grad = None
        z = full.clone().requires_grad_(True)
        x, y = z[:, :32], z[:, 32:]
        _, out, dl = flow(x, y)
        (out.square().mean() - dl.mean() + (out * x).mean()).backward()
        return out.detach(), dl.detach(), z.grad.clone(), {n: p.grad.clone() for n, p in flow.named_parameters()}
    small = run()
    assert flow[0].transformer._train_cache["HT"] == 2
    try:
        dense.FUSED_FWD64 = dense.FUSED_BWD64 = False
        general = run()
        assert flow[0].transformer._train_cache["HT"] == 4
    finally:
        dense.FUSED_FWD64 = dense.FUSED_BWD64 = True
    assert torch.equal(small[0], general[0]) and torch.equal(small[1], general[1])
    assert float((small[2] - general[2]).norm() / general[2].norm()) <= 2e-6
    rel, worst = _grad_errors(small[3], general[3])
    assert rel <= 2e-6, (rel, worst)


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("shape", [(32, 32, (64, 64), ("ReLU", "Tanh")), (24, 20, (48, 64), ("SiLU", "SiLU")), (7, 5, (64, 33), ("Tanh", "Tanh"))])
def test_one_call_backward_variants_of_the_small_couplings(hip_lib, dev, shape, inverse):
    """A coupling of cfg 2's class (either direction: the inverse one reads the layer's OUTPUT in its tail backward) (both networks <= 64 units) has three backward forms: bgk_affine_backward + one
    bgk_affine_net_backward64 per network (dense.TAIL_FUSED64 = False: the form the round started with), ONE bgk_affine_coupling_backward64
    on the saved pre-activations (default: the tail's backward inside the scale network's launch, no g_mu / g_s arrays, no atomics), and
    the same call with NOTHING saved (dense.RECOMPUTE64 = True: every wave recomputes the networks on its tile).  Same forward values
    bit for bit; gradients equal to accumulation-order noise and each within 2e-6 of f64 autograd of the reference's op chain
    (oracle/torch_flow.py; nn/flow/transformer/affine.py:35-70, nn/dense.py:30-48); partial tiles, d and n_in below 32, a frozen
    parameter, a conditioner input that needs no gradient."""
    from bgflow_amd import dense
    n_c, d, hidden, acts = shape
    flow = _affine_layer(n_c, hidden, d, tuple(getattr(torch.nn, a) for a in acts)).to(dev)
    names = [n for n, _ in flow.named_parameters()]
    dict(flow.named_parameters())[[n for n in names if n.endswith("bias")][1]].requires_grad_(False)
    g = torch.Generator(device=dev).manual_seed(5)
    x0, y0 = torch.randn(1000 + d, n_c, device=dev, generator=g), torch.randn(1000 + d, d, device=dev, generator=g)
    w = torch.randn(1000 + d, d, device=dev, generator=g)

    def run(need_x):
        for p in flow.parameters():
            p.grad = None
        x, y = x0.clone().requires_grad_(need_x), y0.clone().requires_grad_(True)
        _, out, dl = flow(x, y, inverse=inverse)
        ((out * w).mean() - 0.3 * dl.mean() + out.square().mean()).backward()
        return (out.detach(), dl.detach(), y.grad.clone(), x.grad.clone() if need_x else None,
                {n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None})
    res = {}
    try:
        for mode in ("separate", "one call", "recompute"):
            dense.TAIL_FUSED64, dense.RECOMPUTE64 = mode != "separate", mode == "recompute"
            res[mode] = [run(True), run(False)]
            cache = flow[0].transformer._train_cache
            assert cache["HT"] == 2 and cache["tail_fused"] == (mode != "separate") and cache["recompute"] == (mode == "recompute")
    finally:
        dense.TAIL_FUSED64, dense.RECOMPUTE64 = True, False
    # f64 reference
    flow64 = _affine_layer(n_c, hidden, d, tuple(getattr(torch.nn, a) for a in acts)).double()
    x, y = x0.cpu().double().requires_grad_(True), y0.cpu().double().requires_grad_(True)
    from oracle import torch_flow as tfl
    outs64, dl64 = tfl.run_flow(flow64, [x, y], inverse=inverse, grad=True)
    out64 = outs64[1]
    ((out64 * w.cpu().double()).mean() - 0.3 * dl64.mean() + out64.square().mean()).backward()
    ref = {n: p.grad for n, p in flow64.named_parameters()}
    for mode, (with_x, without_x) in res.items():
        assert torch.equal(with_x[0], res["separate"][0][0]) and torch.equal(with_x[1], res["separate"][0][1]), mode
        assert with_x[4].keys() == res["separate"][0][4].keys() and len(with_x[4]) == 12
        for got in (with_x, without_x):
            assert float((got[2].cpu().double() - y.grad).norm() / y.grad.norm()) <= 2e-6, mode
            flat_g = torch.cat([got[4][n].reshape(-1).cpu().double() for n in got[4]])
            flat_r = torch.cat([ref[n].reshape(-1) for n in got[4]])
            rel = float((flat_g - flat_r).norm() / flat_r.norm())
            assert rel <= 2e-6, (mode, rel)
        assert float((with_x[3].cpu().double() - x.grad).norm() / x.grad.norm()) <= 2e-6, mode
        assert without_x[3] is None
    print("one-call backward variants agree for", shape, "inverse" if inverse else "forward")


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("n_blocks", [8, 3])
def test_coupling_stack_as_one_autograd_node_equals_the_blocks(hip_lib, dev, inverse, n_blocks):
    """split -> (affine coupling, swap) x n -> merge under autograd as ONE node (dense._AffineStackTrainFn: one log-det buffer, the halves'
    gradients accumulated inside the backward kernels) against the same blocks run one by one: identical outputs, gradients equal to
    accumulation-order noise; an odd number of swaps (the merge then sees the halves exchanged) and the inverse direction included."""
    import bgflow_amd as bg
    from bgflow_amd import configs
    gen = configs.make_affine8_generator(n_blocks=n_blocks, device=dev)
    g = torch.Generator(device=dev).manual_seed(n_blocks)
    z0 = torch.randn(2051, 64, device=dev, generator=g)

    def run():
        for p in gen.flow.parameters():
            p.grad = None
        z = z0.clone().requires_grad_(True)
        x, dl = gen.flow(z, inverse=inverse)
        (x.square().mean() + (x[:, :7] * dl).mean() - dl.mean()).backward()
        return x.detach(), dl.detach(), z.grad.clone(), {n: p.grad.clone() for n, p in gen.flow.named_parameters()}
    one = run()
    try:
        bg.flow._FusedCouplingStack.FUSE_TRAINING_STACK = False
        blocks = run()
    finally:
        bg.flow._FusedCouplingStack.FUSE_TRAINING_STACK = True
    assert torch.equal(one[0], blocks[0])
    assert float((one[1] - blocks[1]).abs().max()) <= 4e-6 * max(1.0, float(blocks[1].abs().max()))
    assert float((one[2] - blocks[2]).norm() / blocks[2].norm()) <= 2e-6
    rel, worst = _grad_errors(one[3], blocks[3])
    assert rel <= 2e-6, (rel, worst)


def test_ic_backward_fixup_with_a_long_list(hip_lib, dev):
    """The fix-up launch of bgk_ic_ic2xyz_backward (one wave per listed sample, round 6) on a batch in which every sample has one or two
    bonds of 5e-5 .. 4e-4 nm at random Z-matrix rows (a placement that refers to such a pair of atoms gets a clamped norm: 40 % of the
    6 000 samples end up listed -- more than the launch has workgroups, which stride over the list)
    against the generic sweep that evaluates the dual numbers in line (no list): bit-identical gradients -- the same placement
    arithmetic per direction, whichever lane carries it (crd_transform/ic.py:386-513 under autograd, ic_helper.py:372-452)."""
    from bgflow_amd import configs
    gen = configs.make_ala2_spline_generator(dev)
    blk = list(gen.flow)[-1]
    B = 6000
    g = torch.Generator().manual_seed(606)
    base = [0.1 + 0.05 * torch.rand(B, 17, generator=g), 0.2 + 0.6 * torch.rand(B, 17, generator=g), torch.rand(B, 17, generator=g),
            torch.randn(B, 9, generator=g)]
    rows = torch.arange(B)
    for _ in range(2):
        col = torch.randint(0, 17, (B,), generator=g)
        base[0][rows, col] = 5e-5 + 3.5e-4 * torch.rand(B, generator=g)
    w = torch.randn(B, 66, generator=g).to(dev)
    res = {}
    rel_ic = [m for m in blk.modules() if hasattr(m, "_fixup_list")]
    assert rel_ic
    try:
        for mode in ("list", "no list"):
            if mode == "no list":
                for m in rel_ic:
                    m._fixup_list = lambda device, n_rows: None
            ins = [t_.to(dev).requires_grad_(True) for t_ in base]
            x, dl = blk(*ins)
            ((x * w).sum() - 0.7 * dl.sum()).backward()
            res[mode] = [t_.grad.clone() for t_ in ins]
            if mode == "list":
                listed = max(int(m._fixup_list(x.device, B)[0]) for m in rel_ic)            # the count the sweep left in the workspace
                print("listed samples:", listed, "of", B)
                assert listed > B // 3
    finally:
        for m in rel_ic:
            m.__dict__.pop("_fixup_list", None)
    for a, b in zip(res["list"], res["no list"]):
        assert bool(torch.isfinite(a).all())
        assert torch.equal(a, b), f"max difference {float((a - b).abs().max()):.2e}"


def test_layer_kernel_scales_every_sample_by_itself(hip_lib, dev):
    """bgk_dense_layer brings its input under a power-of-two scale before the f16 hi / lo split -- per SAMPLE since round 6 (the advisor's
    round-5 finding: one scale per 32-sample tile tied a row's precision to the rows that happened to share its wave).  Rows of
    magnitude 1e-8 .. 1e+8 and a row holding an inf in ONE tile: every finite row within 2e-6 of the f64 product relative to ITS OWN
    scale (|x_row| |W| bound), and bit-identical to the same row evaluated in a batch of its own (nn/dense.py:30-48 on torch.nn.Linear,
    whose rows do not see each other either)."""
    from bgflow_amd import dense
    from bgflow_amd.utils import hash_init_
    lin = hash_init_(torch.nn.Linear(96, 80)).to(dev)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(32, 96, generator=g)
    mags = 10.0 ** torch.linspace(-8, 8, 32)
    x = x * mags[:, None]
    x[7, 5] = float("inf")
    y = dense.dense_layer(x.to(dev), lin).cpu()
    W, b = lin.weight.detach().cpu().double(), lin.bias.detach().cpu().double()
    want = x.double() @ W.T + b
    finite = [r for r in range(32) if r != 7]
    assert not bool(torch.isfinite(y[7]).all())                       # the inf row poisons itself ...
    assert bool(torch.isfinite(y[finite]).all())                      # ... and nothing else
    bound = (x.double().abs() @ W.abs().T + b.abs())[finite]
    err = ((y[finite].double() - want[finite]).abs() / bound).max()
    print("layer kernel, rows of 1e-8 .. 1e+8 in one tile: max error relative to the row's own |x||W| bound", float(err))
    assert float(err) <= 2e-6
    for r in (0, 15, 31):
        alone = dense.dense_layer(x[r:r + 1].to(dev), lin).cpu()
        assert torch.equal(alone[0], y[r]), f"row {r}: its output depends on the rows it shares a tile with"


@pytest.mark.parametrize("B", [1, 37, 4133])
def test_energy_float4_rows_equal_the_staging_kernels(hip_lib, dev, B):
    """bgk_energy_fields(_backward) on rows that are multiples of 4 wide at 16-byte aligned addresses take the float4 kernels (round 6:
    L lanes per row, fixed xor tree; cfg 2's DoubleWellEnergy(64) at 2^20 samples: 0.26 -> see profiles/r06_ab_runs.txt);
    BGK_ENERGY_STAGED=1 keeps the staging kernels.  Same terms in another summation order: energies within 2e-6 of each other and of
    the f64 value, the KL loss sums and the gradients likewise -- DoubleWellEnergy(64) (distribution/energy/double_well.py:17-22),
    NormalDistribution(8) with a mean (distribution/normal.py:61-72), their product with a uniform component (product.py:13-117),
    a row view of a wider tensor."""
    import os
    import bgflow_amd as bg
    g = torch.Generator().manual_seed(B)
    wide = torch.randn(B, 96, generator=g)
    cases = {
        "double well 64": (bg.DoubleWellEnergy(64, a=0.3, b=-2.0, c=0.7), [torch.randn(B, 64, generator=g)]),
        "normal 8": (bg.NormalDistribution(8, mean=torch.randn(8, generator=g)), [torch.randn(B, 8, generator=g)]),
        "row view": (bg.NormalDistribution(32), [wide[:, 32:64]]),
        "product": (bg.ProductDistribution([bg.NormalDistribution(64, mean=torch.randn(64, generator=g)), bg.DoubleWellEnergy(32),
                                            bg.UniformDistribution(torch.zeros(5), 2.0 * torch.ones(5))]),
                    [torch.randn(B, 64, generator=g), torch.randn(B, 32, generator=g), 2.0 * torch.rand(B, 5, generator=g)]),
    }
    for name, (dist, xs) in cases.items():
        dist = dist.to(dev)
        res = {}
        try:
            for mode in ("float4", "staged"):
                if mode == "staged":
                    os.environ["BGK_ENERGY_STAGED"] = "1"
                xg = [x.to(dev).requires_grad_(True) for x in xs]
                u = dist.energy(*xg, temperature=1.3)
                (u * torch.linspace(0.5, 1.5, B, device=dev)[:, None]).sum().backward()
                res[mode] = (u.detach().cpu(), [x.grad.cpu() if x.grad is not None else torch.zeros(x.shape) for x in xg])
        finally:
            os.environ.pop("BGK_ENERGY_STAGED", None)
        x64 = [x.double().requires_grad_(True) for x in xs]
        u64 = dist.cpu().double().energy(*x64, temperature=1.3)
        (u64 * torch.linspace(0.5, 1.5, B, dtype=torch.float64)[:, None]).sum().backward()
        for mode in res:
            scale = u64.abs().clamp_min(1.0)
            assert float(((res[mode][0].double() - u64.detach()).abs() / scale).max()) <= 2e-6, (name, mode)
            for a, b in zip(res[mode][1], x64):
                want = b.grad if b.grad is not None else torch.zeros_like(b)          # (a uniform component's energy does not depend on x)
                assert float((a.double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), (name, mode)


@pytest.mark.parametrize("case", [("B = 1", 1, {}), ("B = 33", 33, {}), ("volume preserving", 200, {"preserve_volume": True})])
def test_small_coupling_training_edge_cases(hip_lib, dev, case):
    """cfg 2's coupling class at the edges of the one-call backward: a single sample, one sample past a tile, and the transformer option
    it does not take (preserve_volume: the log-scales are centred per row, affine.py:46-52 -- that layer keeps bgk_affine_backward +
    bgk_affine_net_backward64; is_circular excludes a scale network in the reference, affine.py:27-28).  Forward equal to the
    inference kernel, every gradient within 5e-5 of f64 autograd of the reference's op chain."""
    label, B, kw = case
    flow = _affine_layer(32, (64, 64), 32, (torch.nn.ReLU, torch.nn.Tanh), **kw).to(dev)
    flow_cpu = _affine_layer(32, (64, 64), 32, (torch.nn.ReLU, torch.nn.Tanh), **kw).double()
    g = torch.Generator().manual_seed(77)
    x, y = torch.randn(B, 32, generator=g).double(), torch.rand(B, 32, generator=g).double()
    wy, wl = torch.randn(B, 32, generator=g).double(), torch.randn(B, 1, generator=g).double()
    ref_out, ref_dl, ref_gx, ref_gy, ref_gp = _f64_layer_grads(flow_cpu, x, y, wy, wl, False)
    xg, yg = x.float().to(dev).requires_grad_(True), y.float().to(dev).requires_grad_(True)
    _, out, dl = flow(xg, yg)
    cache = flow[0].transformer._train_cache
    assert cache["tail_fused"] and cache["HT"] == 2
    with torch.no_grad():
        _, out_inf, dl_inf = flow(xg.detach(), yg.detach())
    assert torch.equal(out, out_inf) and torch.equal(dl, dl_inf), label
    ((out * wy.float().to(dev)).sum() + (dl * wl.float().to(dev)).sum()).backward()
    got = {n: p.grad.double().cpu() for n, p in flow.named_parameters() if p.grad is not None}
    assert set(got) == set(ref_gp)
    rel, worst = _grad_errors(got, {n: ref_gp[n] for n in got})
    assert rel <= 5e-5, (label, rel, worst)
    for name, a, b in (("g_x", xg.grad, ref_gx), ("g_y", yg.grad, ref_gy)):
        err = float((a.double().cpu() - b).norm() / max(float(b.norm()), 1e-30))
        assert err <= 5e-5, f"{label}: {name}: relative L2 {err:.2e}"


def test_wide_whiten_flow_runs_on_the_layer_kernel(hip_lib, dev):
    """A stand-alone WhitenFlow over more than 128 coordinates (a 100-atom molecule's 300 Cartesian dofs, 294 kept: pca.py:37-107) ran
    torch.matmul -> hipBLASLt in rounds 1 - 5 (the verdict's envelope leftover); it is a bias-free Linear on bgk_dense_layer now, in both
    directions and under autograd: no library GEMM launched, values and the input gradient within 2e-5 (relative to the largest entry) of
    the f64 product, whiten o blacken = identity on the kept subspace."""
    import bgflow_amd as bg
    g = torch.Generator().manual_seed(4)
    n, keep, B = 300, 294, 1000
    basis = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))[0]
    data = (torch.randn(4000, n, generator=g, dtype=torch.float64) * torch.logspace(-1.5, 0.5, n, dtype=torch.float64)) @ basis + 3.0
    flow = bg.WhitenFlow(data.float(), keepdims=keep, whiten_inverse=False).to(dev)
    x = data[:B].float().to(dev).requires_grad_(True)
    w = torch.randn(B, keep, generator=g).to(dev)

    def run():
        z, dl = flow(x)
        (z * w).sum().backward()
        with torch.no_grad():
            back, dl2 = flow(z.detach(), inverse=True)
        return z.detach(), back, dl + dl2
    names = _device_kernel_names(run)
    assert not [k for k in names if "Cijk_" in k], "library GEMM in the wide WhitenFlow"
    assert any("dense_layer_kernel" in k for k in names)
    x.grad = None
    z, back, dlsum = run()
    Tw, mean = flow.Twhiten.double().cpu(), flow.X0mean.double().cpu()
    want = (data[:B] .float().double() - mean) @ Tw
    assert float((z.double().cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    want_g = w.double().cpu() @ Tw.T
    assert float((x.grad.double().cpu() - want_g).abs().max()) <= 2e-5 * float(want_g.abs().max())
    assert float(dlsum.abs().max()) == 0.0
    proj = (data[:B].float().double() - mean) @ Tw @ flow.Tblacken.double().cpu() + mean
    assert float((back.double().cpu() - proj).abs().max()) <= 2e-5 * float(proj.abs().max())


@pytest.mark.parametrize("inverse", [False, True])
def test_one_call_backward_is_bit_stable(hip_lib, dev, inverse):
    """bgk_affine_coupling_backward64 sums its partial gradients in a fixed order (no atomics: the log_alpha gradient as one partial per
    workgroup, summed in slab order): the same inputs give the same bits, run after run, in both directions -- the check that caught a
    miscompiled instance of this kernel in round 6 (profiles/r06_ab_runs.txt; tools/r06_dbg_coupling_bwd.py is the long form)."""
    flow = _affine_layer(32, (64, 64), 32, (torch.nn.ReLU, torch.nn.Tanh)).to(dev)
    g = torch.Generator(device=dev).manual_seed(8)
    x0, y0, w = (torch.randn(4133, 32, device=dev, generator=g) for _ in range(3))

    def run():
        for p in flow.parameters():
            p.grad = None
        x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
        _, out, dl = flow(x, y, inverse=inverse)
        ((out * w).sum() + 0.5 * dl.sum()).backward()
        return [x.grad, y.grad] + [p.grad.clone() for p in flow.parameters()]
    first = run()
    assert flow[0].transformer._train_cache["tail_fused"]
    assert all(bool(torch.isfinite(t).all()) for t in first)
    for _ in range(8):
        for a, b in zip(run(), first):
            assert torch.equal(a, b)
