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
    layer = _spline_layer(dev, hidden=hidden)
    opt = torch.optim.SGD(layer.parameters(), lr=0.05)      # (mean loss: gradients of O(1); the weights move by several % per step)
    B = 2048

    def step_grads(fused):
        layer.transformer.allow_fused = fused
        layer.zero_grad()
        xs = _fields(dev, B)
        *out, dl = layer(*xs)
        ((sum((o * o).sum() for o in out) + dl.sum()) / B).backward()
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
        opt.step()
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


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("Kb,d,B", [(5, 7, 300), (6, 17, 1001), (24, 3, 129), (80, 5, 64), (200, 4, 33)])
def test_spline_backward_for_any_bin_count(hip_lib, dev, Kb, d, B, inverse):
    """bgk_rqs_backward for bin counts without a register-resident instance (every K other than 4 / 8 / 12 / 16 / 32, transformer/
    spline.py:87-126 takes any): the direct variant walks the element's parameters in memory.  Gradients w.r.t. the input and every
    parameter against f64 autograd through the torch restatement of the nflows spline in oracle/ -- mixed circular masks, a ragged
    batch, inputs on and beyond the domain's ends (clamped: zero input gradient)."""
    from bgflow_amd.transformer import rqs_backward, rqs_transform
    from oracle import torch_flow as tf
    from test_gpu_parity import synth, t
    circ = np.arange(d) % 3 == 1
    n_nc = int((~circ).sum())
    P = 3 * Kb * d + n_nc
    params, y = synth(700 + Kb, B, P, scale=0.9), synth(800 + Kb, B, d, uniform=True)
    y[0, 0], y[1 % B, d - 1], y[2 % B, 0] = 0.0, 1.0, 1.25          # both ends of the domain and a clamped input
    slots_h = np.full(d, -1, np.int32)
    slots_h[~circ] = np.arange(n_nc, dtype=np.int32)
    slots = torch.as_tensor(slots_h).to(dev)
    st = dict(min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, enable_identity_init=True)
    cfg = (Kb, inverse, 0.0, 1.0, 0.0, 1.0, st)
    g_out, g_dl = synth(900 + Kb, B, d, scale=1.0), synth(901 + Kb, B, 1, scale=1.0)
    gy, gp = rqs_backward(t(y, dev), t(params, dev), slots, cfg, t(g_out, dev), t(g_dl, dev))
    # f64 reference
    p64 = torch.tensor(params, dtype=torch.float64, requires_grad=True)
    y64 = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    w, h, sl, s_nc = torch.split(p64, [d * Kb, d * Kb, d * Kb, n_nc], dim=-1)
    w, h, sl = (v.reshape(B, d, Kb) for v in (w, h, sl))
    cm = torch.tensor(circ)
    nc_full = torch.zeros(B, d, dtype=torch.float64).index_put((torch.arange(B)[:, None], torch.nonzero(~cm).reshape(1, -1)), s_nc)
    last = torch.where(cm[None, :], sl[..., 0], nc_full)
    out, ld = tf.rq_spline(y64.clamp(0.0, 1.0), w, h, torch.cat([sl, last[..., None]], -1), not inverse, 0.0, 1.0, 0.0, 1.0,
                           st["min_bin_width"], st["min_bin_height"], st["min_derivative"], True)
    ((out * torch.tensor(g_out, dtype=torch.float64)).sum() + (ld.sum(-1, keepdim=True) * torch.tensor(g_dl, dtype=torch.float64)).sum()).backward()
    assert float(gy[2 % B, 0]) == 0.0                               # clamped input: no gradient
    for got, want, what in ((gy, y64.grad, "input"), (gp, p64.grad, "parameters")):
        err = (got.cpu().double() - want).abs()
        scale = float(want.abs().max())
        assert bool(torch.isfinite(got).all())
        assert float(err.max()) <= 1e-3 * scale and float(err.quantile(0.99)) <= 1e-4 * scale, \
            f"K = {Kb}: {what} gradient, max error {float(err.max()):.2e} of {scale:.2e}"
    # the forward of the same bin count on the same inputs (bins of forward and backward come from one knot sequence)
    # -- bit-identical to the C oracle's f32 element routine where that holds the knots (K <= 64); against f64 the bound is f32
    # conditioning of the inverse's quadratic: K = 24, element (5, 1) is 5.3e-6 off in the oracle's operation order and 1.1e-6 in
    # torch's f32 op chain
    z, dl = rqs_transform(t(y, dev), t(params, dev), slots, Kb, inverse, 0.0, 1.0, 0.0, 1.0, st)
    assert float((z.cpu().double() - out.detach()).abs().max()) < 1e-5
    if Kb <= 64:
        from oracle import oracle as co
        z_c, dl_c = co.rqs(y, params, is_circular=circ, inverse=inverse, n_bins=Kb, **{k: v for k, v in st.items() if k != "enable_identity_init"})
        assert np.array_equal(z.cpu().numpy(), z_c) and np.array_equal(dl.cpu().numpy().reshape(-1), dl_c.reshape(-1))


def test_hardware_sincos_of_the_generation_tail_on_a_long_chain(hip_lib, oracle, dev):
    """advisor (round 4, low): the generation-tail kernels take the placements' sin / cos from v_sin_f32 / v_cos_f32 (argument in
    revolutions, max abs error 1.24e-7 -- tools/ubench/hw_sincos.hip) instead of the reproducible polynomial: NOT bit-identical to
    the oracle any more (deviations).  What it costs in accuracy, on the worst case the one-launch tail takes: a
    pure 24-atom chain (every atom placed on the three before it, so placement errors travel down the whole chain).  Coordinates
    against the f64 oracle: within 2e-5 nm absolute, and no worse than 4 x the block-by-block path (deterministic polynomial sin / cos)
    on the same inputs."""
    import bgflow_amd as bg
    from bgflow_amd import configs
    from oracle import flow_oracle as fo
    n_atoms = 24
    z = np.array([(i, i - 1, i - 2, i - 3) for i in range(3, n_atoms)], dtype=np.int64)
    n = n_atoms - 3
    ic = bg.RelativeInternalCoordinateTransformation(z, np.array([0, 1, 2]), normalize_angles=True)
    one = lambda v, m: torch.full((m,), float(v))          # noqa: E731
    marginals = {
        0: bg.TruncatedNormalDistribution(mu=one(0.15, n), sigma=one(0.01, n), lower_bound=torch.tensor(1e-5), upper_bound=torch.tensor(np.inf)),
        1: bg.TruncatedNormalDistribution(mu=one(0.5, n), sigma=one(0.05, n), lower_bound=torch.tensor(1e-5), upper_bound=torch.tensor(1.0)),
        2: configs.SloppyUniform(low=one(0.0, n), high=one(1.0, n)),
        3: configs._NormalMarginal(torch.tensor([0, 0, 0, 0.15, 0, 0, 0.2, 0.14, 0.0]), one(0.005, 9)),
    }
    blocks = [bg.WrapFlow(bg.InverseFlow(bg.CDFTransform(marginals[s_])), (s_,)) for s_ in range(4)] \
        + [bg.WrapFlow(bg.InverseFlow(ic), indices=[0, 1, 2, 3], out_indices=(0,))]
    flow_cpu = bg.SequentialFlow(blocks)
    B = 4096
    g = torch.Generator().manual_seed(24)
    xs = [torch.rand(B, w, generator=g).clamp(0.02, 0.98) for w in (n, n, n, 9)]
    x64, dl64 = fo.run_flow(flow_cpu, [v.numpy().astype(np.float64) for v in xs], dtype=np.float64)
    x64 = x64[0] if isinstance(x64, (list, tuple)) else x64
    flow = flow_cpu.to(dev)
    with torch.no_grad():
        x, dl = flow(*[v.to(dev) for v in xs])
        assert any(lbl == "icdf+ic2xyz" for lbl, _ in flow.segments()), "the one-launch tail must have run"
        flow.FUSE_GENERATION_TAIL = False
        x_b, dl_b = flow(*[v.to(dev) for v in xs])
    err = np.abs(x.cpu().numpy().astype(np.float64) - x64).max()
    err_b = np.abs(x_b.cpu().numpy().astype(np.float64) - x64).max()
    assert err <= 2e-5, f"24-atom chain through the one-launch tail: max coordinate error {err:.2e} nm vs f64 (blocks: {err_b:.2e})"
    assert err <= 4 * err_b + 2e-6, f"hardware sin / cos: {err:.2e} against {err_b:.2e} for the polynomial form"
    rel = np.abs(dl.cpu().numpy().reshape(-1) - np.asarray(dl64).reshape(-1)) / np.maximum(np.abs(np.asarray(dl64).reshape(-1)), 1.0)
    assert np.median(rel) <= 2e-6 and rel.max() <= 1e-4


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("B", [777, 32])
def test_spline_chain_as_one_autograd_node(hip_lib, dev, B, inverse):
    """training pass of the cfg-3 couplings as ONE autograd node (flow._SplineTrainChain / dense._SplineChainTrainFn: log-dets added
    into one buffer by the forward launches, conditioner-input gradients accumulated per field inside bgk_dense_backward_dx) against the
    block-by-block path (one node per layer, autograd's own accumulation): same outputs, same log-det, same parameter and input
    gradients -- every field conditions several layers and is transformed by others, inputs that need a gradient and inputs that do not"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    gen = configs.make_ala2_spline_generator(dev)
    flow = bg.SequentialFlow(list(gen.flow)[:16])
    assert [lbl for lbl, _ in flow.segments(train=True)] == ["spline chain"] and len(flow.segments()) == 16
    res = {}
    for chain in (True, False):
        flow.FUSE_TRAINING_CHAINS = chain
        for p in flow.parameters():
            p.grad = None
        xs = [torch.rand(B, d, device=dev, generator=torch.Generator(device=dev).manual_seed(40 + i)) for i, d in enumerate((17, 17, 17, 9))]
        xs[0].requires_grad_(True); xs[2].requires_grad_(True)         # bonds and torsions want gradients, angles / fixed do not
        *out, dl = flow(*xs, inverse=inverse)
        w = torch.linspace(0.5, 1.5, B, device=dev)[:, None]
        (sum((o * o * w).sum() for o in out) - (dl * w).sum()).backward()
        res[chain] = ([o.detach() for o in out], dl.detach(), [p.grad.clone() for p in flow.parameters()], [xs[0].grad.clone(), xs[2].grad.clone()])
        assert xs[1].grad is None and xs[3].grad is None
    (o1, d1, gp1, gx1), (o0, d0, gp0, gx0) = res[True], res[False]
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    assert float((d1 - d0).abs().max()) <= 1e-5 * max(1.0, float(d0.abs().max()))        # (the order of the 16 additions differs)
    for a, b in zip(gp1 + gx1, gp0 + gx0):
        assert a.shape == b.shape and bool(torch.isfinite(a).all())
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6), f"gradient of shape {tuple(a.shape)}"


def test_kl_step_launches_few_aten_kernels(hip_lib, dev):
    """the KL training step of the bench (cfg 3) between its first forward launch and the optimizer: what is left of torch's own
    elementwise / copy kernels once the log-dets and the field gradients are accumulated inside the flow's kernels (round 4: 30 adds +
    26 copies per step)"""
    from torch.profiler import ProfilerActivity, profile
    from bgflow_amd import configs, dp
    from bgflow_amd.training import FlatAdam
    gen = configs.make_ala2_spline_generator(dev)
    opt = FlatAdam(list(gen.flow.parameters()), lr=1e-5)
    z = [torch.rand(4096, d, device=dev) for d in (17, 17, 17, 9)]

    def step():
        opt.zero_grad()
        *x, dlogp = gen.flow(*z)
        loss = dp.global_kl_mean(gen._target, x, dlogp, drop_nonfinite=True)
        opt.backward(loss)
        opt.step()
    step(); step()
    torch.cuda.synchronize()
    try:
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            step()
            torch.cuda.synchronize()
    except Exception as e:          # no kernel tracer on this box
        pytest.skip(f"torch.profiler unavailable: {e!r}")
    names = [e.key for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA for _ in range(e.count)]
    aten = [n for n in names if "at::native" in n or "elementwise" in n or "copyBuffer" in n or "fillBuffer" in n or "Memset" in n or "Memcpy" in n]
    if not names:
        pytest.skip("torch.profiler recorded no device kernels")
    from collections import Counter
    short = Counter(n.split("<")[0].replace("void at::native::", "")[:40] + ("|" + n.split("Functor")[0].split("::")[-1] if "Functor" in n else "") for n in aten)
    print("device kernels of one KL step:", len(names), "of which torch's own:", len(aten), dict(short))
    assert len(aten) <= 16, f"{len(aten)} aten / copy / fill launches in one KL step: {dict(short)}"


def test_ic_backward_where_the_forward_clamped_a_norm(hip_lib, dev):
    """IC -> xyz backward on degenerate geometries.  A bond drawn ~1e-4 from the lower end of its marginal (cfg 3: a normal truncated one
    sigma below its mean -- 0.1 % of uniform prior samples) puts two atoms 1e-4 nm apart; for the placements that refer to them
    |v1 x (v1 x v2)| falls below eps = 1e-7, the reference clamps the norm (ic_helper.py:372-452), the clamped vector is no unit vector
    any more and log|det J| is no longer 2 ln d + ln|sin a|: the reference's autograd differentiates the explicit determinant with
    torch.clamp's derivative.  bgk_ic_ic2xyz_backward evaluates exactly that on dual numbers for such placements (round 5; before,
    the closed-form adjoint was off by 40 - 60 % on these samples, which carry the largest gradients of a KL step: 2e-4 of the flat
    gradient -- oracle/bgo_impl.h's closed-form sweep still is, in f64 too: rel L2 0.6 on the first half of this batch).  Second half:
    angles within 1e-5 of 0 / pi (gl cos a / sin a needs the sine to relative accuracy; the atoms placed next see nearly collinear
    reference atoms).  Yardstick: f64 autograd of the reference's op chain (oracle/torch_flow.py::ic2xyz_torch), and the SAME chain in
    f32 on the host -- what the reference's own f32 autograd gives on these inputs: the kernel must not be further from f64 than
    3 x that (+ 1e-3 of the gradient's norm).  g_fixed of the first half is measured per sample: a fixed atom's adjoint there is the
    O(1) sum of +-1e5 terms (|g_bonds| of the same sample: 1e5), a handful of samples lose it to f32 cancellation in either code."""
    from bgflow_amd import configs
    from oracle import torch_flow as tfl
    gen = configs.make_ala2_spline_generator(dev)
    gen64 = configs.make_ala2_spline_generator().double()
    gen32 = configs.make_ala2_spline_generator()
    blk, blk64, blk32 = list(gen.flow)[-1], list(gen64.flow)[-1], list(gen32.flow)[-1]
    B = 2048
    g = torch.Generator().manual_seed(99)
    bonds = 0.1 + 0.05 * torch.rand(B, 17, generator=g)
    angles = 0.2 + 0.6 * torch.rand(B, 17, generator=g)
    tors = torch.rand(B, 17, generator=g)
    fixed = torch.randn(B, 9, generator=g)
    tiny = torch.rand(B, 17, generator=g) < 0.02
    tiny[B // 2:] = False                                              # second half of the batch: regular bonds
    bonds = torch.where(tiny, 5e-5 + 3.5e-4 * torch.rand(B, 17, generator=g), bonds)
    near = torch.rand(B, 17, generator=g) < 0.01
    near[:B // 2] = False
    angles = torch.where(near, torch.where(torch.rand(B, 17, generator=g) < 0.5, 3e-6 + 1e-5 * torch.rand(B, 17, generator=g),
                                           1.0 - 3e-6 - 1e-5 * torch.rand(B, 17, generator=g)), angles)
    w = torch.randn(B, 66, generator=g)
    v = 0.5 + torch.rand(B, 1, generator=g)
    ref = {}
    for key, b_, dt in (("f64", blk64, torch.float64), ("f32", blk32, torch.float32)):
        ins_ = [t_.detach().clone().to(dt).requires_grad_(True) for t_ in (bonds, angles, tors, fixed)]
        (x_,), dl_ = tfl.run_block(b_, ins_, False, grad=True)
        ((x_ * w.to(dt)).sum() - (dl_ * v.to(dt)).sum()).backward()
        ref[key] = [t_.grad.double() for t_ in ins_]
    ins = [t_.detach().clone().to(dev).requires_grad_(True) for t_ in (bonds, angles, tors, fixed)]
    x, dl = blk(*ins)
    ((x * w.to(dev)).sum() - (dl * v.to(dev)).sum()).backward()
    names = ("bonds", "angles", "torsions", "fixed")
    halves = {"clamped norms": slice(0, B // 2), "angles at 0 / pi": slice(B // 2, B)}
    failures = []
    for what, sl in halves.items():
        for k, nm in enumerate(names):
            got, want, host = ins[k].grad[sl].cpu().double(), ref["f64"][k][sl], ref["f32"][k][sl]
            assert bool(torch.isfinite(got).all())
            per = (got - want).norm(dim=1) / want.norm(dim=1).clamp_min(1e-30)
            rel, rel_host = float((got - want).norm() / want.norm()), float((host - want).norm() / want.norm())
            worst = torch.argsort(per, descending=True)[:3].tolist()
            print(f"IC backward, {what}: g_{nm} rel L2 {rel:.2e} (reference chain in f32: {rel_host:.2e}), median {float(per.median()):.2e}, 99th percentile "
                  f"{float(per.quantile(0.99)):.2e}; worst samples "
                  + "; ".join(f"#{i} ({per[i]:.1e}, |g| {float(want[i].norm()):.1e} of {float(want.norm()):.1e}, tiny bonds at {torch.nonzero(tiny[sl][i]).reshape(-1).tolist()})" for i in worst))
            ok = float(per.median()) <= 1e-5
            if what == "clamped norms" and nm == "fixed":
                ok = ok and float(per.quantile(0.99)) <= 5e-2
            else:
                ok = ok and rel <= 3.0 * rel_host + 1e-3
            if not ok:
                failures.append(f"{what}: g_{nm} rel L2 {rel:.2e} vs {rel_host:.2e} for the f32 chain (median {float(per.median()):.2e})")
    assert not failures, "; ".join(failures)


@pytest.mark.parametrize("B", [1, 63, 64, 2085])
def test_ic_backward_dma_staging_equals_the_row_loop_kernels(hip_lib, dev, B):
    """bgk_ic_ic2xyz_backward on contiguous tensors stages its tiles by DMA and stores the tile images as 16-byte pieces (round 5);
    BGK_IC_BWD_NODMA=1 / BGK_IC_BWD_LDS=1 select the earlier kernels (per-lane row loops; positions in registers / in LDS); all three
    list the samples with a clamped norm for the fix-up launch (dual-number adjoint); without the list (fix_ws = NULL) the generic
    sweep evaluates the dual numbers in line.  The fix-up launch itself is one wave per listed sample, the twelve dual directions on
    twelve lanes (round 6); BGK_IC_FIX_LANES=1 selects round 5's lane-per-sample form (three passes of four directions).  Same
    arithmetic per placement and direction: bit-identical gradients, for whole and partial tiles (row counts that leave 1..3 floats over)."""
    import os
    from bgflow_amd import configs
    gen = configs.make_ala2_spline_generator(dev)
    blk = list(gen.flow)[-1]
    g = torch.Generator().manual_seed(B)
    base = [0.1 + 0.05 * torch.rand(B, 17, generator=g), 0.2 + 0.6 * torch.rand(B, 17, generator=g), torch.rand(B, 17, generator=g),
            torch.randn(B, 9, generator=g)]
    if B > 3:
        base[0][3, 5] = 2e-4                    # one clamped placement: the dual-number path is part of the comparison
    w = torch.randn(B, 66, generator=g).to(dev)
    res = {}
    try:
        rel_ic = [m for m in blk.modules() if hasattr(m, "_fixup_list")]
        assert rel_ic
        for mode in ("dma", "BGK_IC_BWD_NODMA", "BGK_IC_BWD_LDS", "BGK_IC_FIX_LANES", "no list"):
            if mode.startswith("BGK_"):
                os.environ[mode] = "1"
            if mode == "no list":               # without the workspace: the generic sweep with the dual numbers in line
                for m in rel_ic:
                    m._fixup_list = lambda device, n_rows: None
            ins = [t_.to(dev).requires_grad_(True) for t_ in base]
            x, dl = blk(*ins)
            ((x * w).sum() - 0.7 * dl.sum()).backward()
            res[mode] = [t_.grad.clone() for t_ in ins]
            os.environ.pop(mode, None)
    finally:
        os.environ.pop("BGK_IC_BWD_NODMA", None)
        os.environ.pop("BGK_IC_BWD_LDS", None)
        os.environ.pop("BGK_IC_FIX_LANES", None)
        for m in rel_ic:
            m.__dict__.pop("_fixup_list", None)
    for mode in ("BGK_IC_BWD_NODMA", "BGK_IC_BWD_LDS", "BGK_IC_FIX_LANES", "no list"):
        for a, b in zip(res["dma"], res[mode]):
            assert bool(torch.isfinite(a).all())
            assert torch.equal(a, b), f"B = {B}: DMA-staged sweep vs {mode}: max difference {float((a - b).abs().max()):.2e}"


@pytest.mark.parametrize("kind", ["cfg2", "silu64"])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_two_network_pipeline_equals_the_sequential_networks(hip_lib, dev, kind, inverse):
    """cfg 2's weight-resident kernel runs the shift and the scale network as one software pipeline (a GEMM of one network carries the
    activation + f16 split of the other, round 5); BGK_AFFINE_NO_PIPE2=1 runs them one after the other.  Same operations on the same
    operands in the same order per accumulator: bit-identical outputs and log-dets (ReLU / Tanh pair of the RealNVP configs, SiLU / SiLU)."""
    import os
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    acts = (torch.nn.ReLU(), torch.nn.Tanh()) if kind == "cfg2" else (torch.nn.SiLU(), torch.nn.SiLU())
    tr = bg.AffineTransformer(bg.DenseNet([32, 64, 64, 32], acts[0]), bg.DenseNet([32, 64, 64, 32], acts[1]))
    layer = hash_init_(bg.CouplingFlow(tr)).to(dev)
    B = 4133
    g = torch.Generator(device=dev).manual_seed(17)
    x, y = torch.randn(B, 32, device=dev, generator=g), torch.randn(B, 32, device=dev, generator=g)
    res = {}
    try:
        for mode in ("pipeline", "sequential"):
            if mode == "sequential":
                os.environ["BGK_AFFINE_NO_PIPE2"] = "1"
            with torch.no_grad():
                _, out, dl = layer(x, y, inverse=inverse)
            res[mode] = (out.clone(), dl.clone())
    finally:
        os.environ.pop("BGK_AFFINE_NO_PIPE2", None)
    assert bool(torch.isfinite(res["pipeline"][0]).all())
    assert torch.equal(res["pipeline"][0], res["sequential"][0]) and torch.equal(res["pipeline"][1], res["sequential"][1])


@pytest.mark.parametrize("what,on", [("TORSIONS", "FIXED"), ("FIXED", "TORSIONS"), ("BONDS", "ANGLES")])
@pytest.mark.parametrize("B", [77, 4133])
def test_element_major_saved_parameters_equal_the_reference_layout(hip_lib, dev, what, on, B):
    """the fused training forward saves the spline parameters element-major, [B][d][3 K + 1] -- the order the kernel holds them in, 16
    store instructions per chunk instead of 64 (round 5) -- and bgk_rqs_backward reads that layout; dense.PACKED_PARAMS = False keeps
    the reference's column order [w | h | s | s_nc].  Same values either way: bit-identical outputs and gradients -- circular dims
    (no slot), a last chunk of 2 / 4 dims, partial tiles."""
    from bgflow_amd import dense
    layer = _spline_layer(dev, what=what, on=on)
    res = {}
    prev, prev_rc = dense.PACKED_PARAMS, dense.RECOMPUTE_PARAMS
    try:
        dense.RECOMPUTE_PARAMS = False
        for packed in (True, False):
            dense.PACKED_PARAMS = packed
            for p in layer.parameters():
                p.grad = None
            xs = _fields(dev, B)
            *out, dl = layer(*xs)
            assert layer.transformer._fused_cache.get("params_packed") is packed, "the fused training forward must have run in the requested layout"
            w = torch.linspace(0.5, 1.5, B, device=dev)[:, None]
            (sum((o * o * w).sum() for o in out) - (dl * w).sum()).backward()
            res[packed] = ([o.detach().clone() for o in out], dl.detach().clone(), [p.grad.clone() for p in layer.parameters()],
                           [x.grad.clone() for x in xs if x.grad is not None])
    finally:
        dense.PACKED_PARAMS, dense.RECOMPUTE_PARAMS = prev, prev_rc
    for a, b in zip(res[True][0] + [res[True][1]] + res[True][2] + res[True][3], res[False][0] + [res[False][1]] + res[False][2] + res[False][3]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


@pytest.mark.parametrize("what,on", [("TORSIONS", "FIXED"), ("FIXED", "TORSIONS"), ("BONDS", "ANGLES")])
@pytest.mark.parametrize("B", [1, 77, 4133])
def test_spline_backward_with_recomputed_parameters_equals_the_saved_parameters(hip_lib, dev, what, on, B):
    """Round 5: the fused training forward writes NO spline parameters (its params_layout = 2: 446 MB per layer at the bench batch);
    bgk_coupling_rqs_dense_h2_backward redoes the conditioner's output layer from the saved z1 on the matrix cores -- the forward's
    operand blocks, f16 split and MFMA order -- and runs the VJP of bgk_rqs_backward from LDS.  The recomputed parameters are the
    forward's bit for bit, so outputs and every gradient equal the saved-parameter path exactly (circular dims without a slot, last
    chunks of 2 / 4 dims, partial tiles, a single sample).  transformer/spline.py:109-188, nn/dense.py:47-48."""
    from bgflow_amd import dense, _lib
    layer = _spline_layer(dev, what=what, on=on)
    res = {}
    prev = dense.RECOMPUTE_PARAMS
    try:
        for mode in ("recompute-exact", "recompute", "saved"):
            dense.RECOMPUTE_PARAMS = mode != "saved"
            old = _lib.lib().bgk_set_option(3, 1 if mode == "recompute-exact" else 2)
            try:
                for p in layer.parameters():
                    p.grad = None
                xs = _fields(dev, B)
                *out, dl = layer(*xs)
                assert layer.transformer._fused_cache.get("params_recompute") is (mode != "saved"), "the fused training forward must have run in the requested form"
                w = torch.linspace(0.5, 1.5, B, device=dev)[:, None]
                (sum((o * o * w).sum() for o in out) - (dl * w).sum()).backward()
            finally:
                _lib.lib().bgk_set_option(3, old)
            res[mode] = ([o.detach().clone() for o in out], dl.detach().clone(), [p.grad.clone() for p in layer.parameters()],
                         [x.grad.clone() for x in xs if x.grad is not None])
    finally:
        dense.RECOMPUTE_PARAMS = prev
    flat = {m: r[0] + [r[1]] + r[2] + r[3] for m, r in res.items()}
    # the deterministic VJP forms (option 3 = 1): the arithmetic of bgk_rqs_backward on bit-identical parameters
    for a, b in zip(flat["recompute-exact"], flat["saved"]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    # the shipped form (hardware exp2 / rcp in the element's softmax and knots, as in the forward): same outputs; gradients: every
    # tensor to 2e-4 of its scale and 3e-5 in relative L2 (exp2 of a rounded argument: ~5e-7 relative in the bin probabilities, through
    # the divisions by bin widths of ~0.1; an input within ~1e-7 of a knot may be evaluated in the neighbouring bin -- the spline is
    # C1 there)
    n_out = len(res["saved"][0]) + 1
    worst = []
    for k, (a, b) in enumerate(zip(flat["recompute"], flat["saved"])):
        if k < n_out:
            assert torch.equal(a, b)
        else:
            scale = float(b.abs().max()) + 1e-30
            worst.append((float((a - b).abs().max()) / scale, float((a - b).double().norm()) / (float(b.double().norm()) + 1e-30), k))
    assert max(w[0] for w in worst) <= 2e-4 and max(w[1] for w in worst) <= 3e-5, f"(max / scale, relative L2, tensor): {sorted(worst)[-3:]}"


@pytest.mark.parametrize("what,on,dims", [("BONDS", "ANGLES", (3, 17, 12, 9)), ("TORSIONS", "BONDS", (3, 17, 12, 9)), ("ANGLES", "FIXED", (3, 5, 12, 4)),
                                          ("FIXED", "TORSIONS", (3, 5, 12, 1))])
def test_recomputed_parameters_for_other_chunk_shapes(hip_lib, dev, what, on, dims):
    """the recompute backward on transformed widths that end in every slot count: d = 3 (one chunk, two slots), 12 (5 + 5 + 2: a last chunk of
    one slot, circular dims), 5 (exactly one whole chunk), 1 (a single element per sample) -- bit-identical to the saved-parameter path
    with the deterministic VJP forms (bgk_set_option(3, 1))"""
    from bgflow_amd import configs, dense, _lib
    from bgflow_amd.utils import hash_init_
    names = ("BONDS", "ANGLES", "TORSIONS", "FIXED")
    dd = dict(zip(names, dims))
    circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
    layer = hash_init_(configs._spline_coupling(what, on, dd, circ, slot)).to(dev)
    B = 555
    res = {}
    prev = dense.RECOMPUTE_PARAMS
    old = _lib.lib().bgk_set_option(3, 1)
    try:
        for rc in (True, False):
            dense.RECOMPUTE_PARAMS = rc
            for p in layer.parameters():
                p.grad = None
            xs = [torch.rand(B, dd[f], device=dev, generator=torch.Generator(device=dev).manual_seed(7 + i)).requires_grad_(True)
                  for i, f in enumerate(configs.IC_FIELDS)]
            *out, dl = layer(*xs)
            assert layer.transformer._fused_cache.get("params_recompute") is rc
            w = torch.linspace(0.5, 1.5, B, device=dev)[:, None]
            (sum((o * o * w).sum() for o in out) - (dl * w).sum()).backward()
            res[rc] = [o.detach().clone() for o in out] + [dl.detach().clone()] + [p.grad.clone() for p in layer.parameters()] + \
                [x.grad.clone() for x in xs if x.grad is not None]
    finally:
        dense.RECOMPUTE_PARAMS = prev
        _lib.lib().bgk_set_option(3, old)
    for a, b in zip(res[True], res[False]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_recomputed_parameters_with_a_mixed_circular_mask(hip_lib, dev):
    """circular and non-circular dims in one transformer: the recompute backward derives a non-circular dim's slot (its column behind the
    3 K d regular ones) from the mask -- the rank of the dim among the non-circular ones, what ConditionalSplineTransformer._nc_slot
    tabulates for bgk_rqs_backward -- bit-identical gradients with the saved-parameter path (deterministic VJP forms)"""
    import bgflow_amd as bg
    from bgflow_amd import dense, _lib
    from bgflow_amd.utils import hash_init_
    circ = np.array([1, 0, 1, 1, 0, 0, 1, 0, 0, 1, 1, 0], bool)       # d = 12: chunks of 5 + 5 + 2, slots 0 .. 5 spread over all of them
    d, d_on, B = len(circ), 9, 333
    net = bg.DenseNet([d_on, 128, 128, 3 * 8 * d + int((~circ).sum())], activation=torch.nn.SiLU())
    layer = hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(params_net=net, is_circular=torch.tensor(circ)),
                                       transformed_indices=[0], cond_indices=[1])).to(dev)
    res = {}
    prev = dense.RECOMPUTE_PARAMS
    old = _lib.lib().bgk_set_option(3, 1)
    try:
        for rc in (True, False):
            dense.RECOMPUTE_PARAMS = rc
            for p in layer.parameters():
                p.grad = None
            xs = [torch.rand(B, w, device=dev, generator=torch.Generator(device=dev).manual_seed(11 + i)).requires_grad_(True) for i, w in enumerate((d, d_on))]
            *out, dl = layer(*xs)
            assert layer.transformer._fused_cache.get("params_recompute") is rc
            w = torch.linspace(0.5, 1.5, B, device=dev)[:, None]
            (sum((o * o * w).sum() for o in out) - (dl * w).sum()).backward()
            res[rc] = [o.detach().clone() for o in out] + [dl.detach().clone()] + [p.grad.clone() for p in layer.parameters()] + [x.grad.clone() for x in xs]
    finally:
        dense.RECOMPUTE_PARAMS = prev
        _lib.lib().bgk_set_option(3, old)
    for a, b in zip(res[True], res[False]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


@pytest.mark.parametrize("B,drop", [(4096, True), (1000, False)])
def test_kl_integrand_inside_the_generation_tail(hip_lib, dev, B, drop):
    """SURVEY f-3's single-pass `kldiv` (round 5): `BoltzmannGenerator.kldiv_mean` evaluates the target energy and the loss sums inside
    the generation tail's training launch (bgk_icdf_ic2xyz_uni_train_kl -- the lanes hold their samples' coordinates in registers there)
    instead of a second pass over x (bgk_energy_fields).  Same loss and the same gradients as the two-launch form (the energy's 66 terms
    are summed in another order: f32 rounding), whole and partial tiles, with and without the non-finite filter."""
    from bgflow_amd import configs
    gen = configs.make_ala2_spline_generator(dev)
    res = {}
    for fused in (True, False):
        gen.flow.FUSE_KL_EPILOGUE = fused
        for p in gen.flow.parameters():
            p.grad = None
        torch.manual_seed(123)
        loss = gen.kldiv_mean(B, drop_nonfinite=drop)
        if fused:
            node, seen = loss.grad_fn, set()
            stack = [node]
            while stack:                       # the fused node must be part of the graph
                f = stack.pop()
                if f is None or f in seen:
                    continue
                seen.add(f)
                stack.extend(g for g, _ in f.next_functions)
            assert any("_FusedTailKLFn" in type(f).__name__ for f in seen), "kldiv_mean did not take the single-pass form"
        loss.backward()
        res[fused] = (float(loss), [p.grad.clone() for p in gen.flow.parameters()])
    gen.flow.FUSE_KL_EPILOGUE = True
    (l1, g1), (l0, g0) = res[True], res[False]
    assert np.isfinite(l1) and abs(l1 - l0) <= 2e-6 * abs(l0)
    num = sum(float(((a - b).double() ** 2).sum()) for a, b in zip(g1, g0))
    den = sum(float((b.double() ** 2).sum()) for b in g0)
    assert (num / den) ** 0.5 <= 2e-6, f"flat gradient: relative L2 {(num / den) ** 0.5:.2e} between the single-pass and the two-launch form"


# ---- hidden layers of 129 .. 256 units on the one-launch kernel (bgk_fused.hip::coupling_rqs_dense_w256_kernel) ----------------------
@pytest.mark.parametrize("hidden", [(256, 256), (200, 130)])
@pytest.mark.parametrize("inverse", [False, True])
def test_spline_coupling_with_hidden_width_256_runs_fused(hip_lib, dev, hidden, inverse):
    """conditioners with hidden layers wider than 128 (up to 256; in between zero-padded) run as ONE launch in split-f16 mode: same
    function as the conditioner evaluated layer by layer, checked against the f64 oracle (non-periodic and periodic conditioner input,
    non-circular and circular splines, a batch that is not a multiple of the 32-sample tile)"""
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
    circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
    t = lambda v: torch.as_tensor(v, dtype=torch.float32, device=dev)                                   # noqa: E731
    for what, on in (("TORSIONS", "FIXED"), ("BONDS", "TORSIONS"), ("FIXED", "TORSIONS")):
        layer_cpu = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden))
        layer = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden)).to(dev)
        B = 1037
        xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
        with warnings.catch_warnings():
            warnings.simplefilter("error")                     # a rejection (RuntimeWarning) would mean the generic path ran
            with torch.no_grad():
                *outs, dl = layer(*[t(v) for v in xs], inverse=inverse)
        plan = layer.transformer._fused_cache
        assert plan.get("hidden") == 256 and plan.get("padded") == (hidden != (256, 256)), "the width-256 kernel must have run"
        ti = slot[what]
        outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
        np.testing.assert_allclose(outs[ti].cpu().numpy(), outs64[ti], rtol=0, atol=2e-5)
        np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=2e-5, atol=2e-5)
        for k in range(4):
            if k != ti:
                assert torch.equal(outs[k], t(xs[k])), "conditioning fields pass through untouched"
    # training falls back to the layer-by-layer conditioner: gradients flow, no operand packing per step
    xs_t = [t(v) for v in xs]
    *_, dl_t = layer(*xs_t, inverse=inverse)
    dl_t.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters() if p.requires_grad)


@pytest.mark.parametrize("n_bins", [4, 12, 16, 32])
@pytest.mark.parametrize("act", [torch.nn.ReLU, torch.nn.Tanh])
def test_hidden_width_256_other_bin_counts_and_activations(hip_lib, dev, n_bins, act):
    """the width-256 kernel for the other fused bin counts and activations, round trip included; bin indices against the layer-by-layer
    path (ties at a knot aside)"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    d, d_c = 11, 23
    P = 3 * n_bins * d + d
    mk = lambda: hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(                       # noqa: E731
        bg.DenseNet([d_c, 192, 256, P], activation=act()), is_circular=False), transformed_indices=(1,), cond_indices=(0,)))
    layer_cpu, layer = mk(), mk().to(dev)
    B = 4099
    xs = [synth(B, B, d_c), synth(B + 5, B, d, uniform=True)]
    dx = [torch.as_tensor(v, dtype=torch.float32, device=dev) for v in xs]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            _, y, dl = layer(*dx)
            _, back, dl_back = layer(dx[0], y, inverse=True)
    assert layer.transformer._fused_cache.get("hidden") == 256
    outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], False, np.float64, [])
    np.testing.assert_allclose(y.cpu().numpy(), outs64[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(back.cpu().numpy(), xs[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose((dl + dl_back).cpu().numpy(), 0.0, atol=1e-4)


def test_hidden_width_256_full_size_rows_against_the_layerwise_path(hip_lib, dev):
    """2^18 samples through the width-256 kernel: every output against the conditioner run layer by layer in f32 (library GEMMs) + the
    stand-alone spline kernel; the share of samples beyond 1e-5 stays at tie level"""
    from bgflow_amd.utils import synth
    layer = _spline_layer(dev, what="ANGLES", on="BONDS", hidden=(256, 256))
    B = 1 << 18
    xs = [torch.as_tensor(synth(B + 7 * i, B, d, uniform=True), dtype=torch.float32, device=dev) for i, d in enumerate((17, 17, 17, 9))]
    with torch.no_grad():
        *outs, dl = layer(*xs)
        layer.transformer.allow_fused = False
        *outs_ref, dl_ref = layer(*xs)
        layer.transformer.allow_fused = True
    assert layer.transformer._fused_cache.get("hidden") == 256
    err = (outs[1] - outs_ref[1]).abs().max(dim=1).values
    assert float(err.median()) < 2e-6 and float((err > 1e-5).float().mean()) < 2e-3
    derr = (dl - dl_ref).abs().view(-1)
    assert float(derr.median()) < 5e-6 and float((derr > 1e-4).float().mean()) < 2e-3


# ---- bgk_dense_layer: one Linear (+ activation) of a conditioner outside the one-launch kernels' envelope ---------------------------
def _act_ref(v, act):
    if act == 1:
        return v / (1.0 + np.exp(-v))
    if act == 2:
        return np.maximum(v, 0.0)
    if act == 3:
        return np.tanh(v)
    return v


@pytest.mark.parametrize("n_in,n_out", [(1, 4), (4, 1), (17, 128), (60, 425), (255, 256), (256, 300), (300, 77), (700, 130)])
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_dense_layer_kernel_against_f64(hip_lib, dev, n_in, n_out, act):
    """y = act(x W^T + b) on bgk_dense_layer against the f64 product: every kernel instance (1 .. 16 k-steps), accumulating passes beyond
    256 input columns, partial tiles in rows and columns, rows that do not start on 16-byte boundaries"""
    from bgflow_amd import dense
    from bgflow_amd.utils import synth
    lin = torch.nn.Linear(n_in, n_out)
    with torch.no_grad():
        lin.weight.copy_(torch.as_tensor(synth(3 + n_in, n_out, n_in, scale=1.0 / np.sqrt(n_in))))
        lin.bias.copy_(torch.as_tensor(synth(5 + n_out, n_out, scale=0.3)))
    W, b = lin.weight.detach().double().numpy(), lin.bias.detach().double().numpy()
    lin = lin.to(dev)
    for B in (1, 37, 1037):
        x = synth(11 + B, B, n_in + 1, scale=1.5)
        for view in (lambda t: t[:, :n_in], lambda t: t[:, 1:]):       # aligned rows (when n_in + 1 is a multiple of 4) / shifted by 4 bytes
            xv = view(torch.as_tensor(x, device=dev))
            with torch.no_grad():
                y = dense.dense_layer(xv, lin, act)
            ref = _act_ref(view(torch.as_tensor(x)).double().numpy() @ W.T + b, act)
            scale = (np.abs(view(torch.as_tensor(x)).numpy()).astype(np.float64) @ np.abs(W).T + np.abs(b)).max()
            assert y.shape == (B, n_out)
            np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=6e-7 * scale)


def test_dense_layer_kernel_input_range_and_shapes(hip_lib, dev):
    """inputs far outside the f16 range on either side keep f32-class accuracy (per-tile power-of-two scale), leading batch dimensions
    are kept, an empty batch is a no-op"""
    from bgflow_amd import dense
    from bgflow_amd.utils import synth
    lin = torch.nn.Linear(40, 70)
    W, b = lin.weight.detach().double().numpy(), lin.bias.detach().double().numpy()
    lin = lin.to(dev)
    x = synth(2, 3, 50, 40, scale=1.0).astype(np.float64)
    for mag in (1e-12, 1.0, 1e9, 1e30):
        with torch.no_grad():
            y = dense.dense_layer(torch.as_tensor(x * mag, dtype=torch.float32, device=dev), lin, 0)
        assert y.shape == (3, 50, 70)
        xm = (x * mag).astype(np.float32).astype(np.float64)
        ref = xm @ W.T + b
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=0, atol=4e-7 * (np.abs(xm) @ np.abs(W).T + np.abs(b)).max())
    # a non-finite entry poisons its own sample's outputs and nothing else, whatever the other samples of its 32-row tile hold
    xb = (x[0] * 1e5).astype(np.float32)
    xb[3, 7], xb[20, 0] = np.inf, np.nan
    with torch.no_grad():
        yb = dense.dense_layer(torch.as_tensor(xb, device=dev), lin, 0).cpu().numpy()
    good = np.ones(50, bool)
    good[[3, 20]] = False
    assert not np.isfinite(yb[3]).any() and np.isnan(yb[20]).all()
    refb = xb[good].astype(np.float64) @ W.T + b
    np.testing.assert_allclose(yb[good], refb, rtol=0, atol=4e-7 * (np.abs(xb[good].astype(np.float64)) @ np.abs(W).T + np.abs(b)).max())
    with torch.no_grad():
        assert dense.dense_layer(torch.empty(0, 40, device=dev), lin, 1).shape == (0, 70)


def test_densenet_layers_run_on_the_layer_kernel(hip_lib, dev):
    """a DenseNet outside the fused envelope (four hidden layers, one of 300 units, LeakyReLU in between two of them) runs its Linear
    layers on bgk_dense_layer: no library GEMM is launched in inference; forward values and parameter gradients agree with torch's
    own layers on the same weights"""
    import bgflow_amd as bg
    from bgflow_amd import dense
    from bgflow_amd.utils import hash_init_, synth
    acts = [torch.nn.SiLU(), torch.nn.LeakyReLU(0.1), torch.nn.Tanh(), torch.nn.ReLU()]
    net = hash_init_(bg.DenseNet([23, 64, 300, 96, 33, 51], activation=acts)).to(dev)
    x = torch.as_tensor(synth(9, 2051, 23), device=dev)
    names = []
    try:
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
            with torch.no_grad():
                y = net(x)
            torch.cuda.synchronize()
        names = [e.key for e in prof.key_averages()]
    except Exception:               # no kernel tracer on this box: the operand caches below still show which path ran
        with torch.no_grad():
            y = net(x)
    assert all("_bgk_layer_ops" in m.__dict__ for m in net._layers if isinstance(m, torch.nn.Linear)), "every Linear must have run on the kernel"
    assert not any(("Cijk" in n) or ("gemm" in n.lower()) or n.startswith("aten::addmm") or n.startswith("aten::mm") for n in names), names
    dense.LAYER_KERNEL = False
    try:
        with torch.no_grad():
            y_ref = net(x)
        xg = x.clone().requires_grad_(True)
        (net(xg) ** 2).sum().backward()
        g_ref = [p.grad.clone() for p in net.parameters()] + [xg.grad.clone()]
    finally:
        dense.LAYER_KERNEL = True
    assert float((y - y_ref).abs().max()) <= 2e-6 * max(1.0, float(y_ref.abs().max()))
    net.zero_grad()
    xg = x.clone().requires_grad_(True)
    # round 6: the BACKWARD of these layers launches no library GEMM either (dX on bgk_dense_layer with the operands of W^T, dW / db on
    # bgk_linear_weight_grad), and SiLU / Tanh / ReLU and their derivatives run on bgk_activation (the LeakyReLU stays torch's)
    names = []
    try:
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
            (net(xg) ** 2).sum().backward()
            torch.cuda.synchronize()
        names = [e.key for e in prof.key_averages()]
    except Exception:
        net.zero_grad()
        xg.grad = None
        (net(xg) ** 2).sum().backward()
    bad = [n for n in names if ("Cijk" in n) or ("gemm" in n.lower()) or n.startswith("aten::addmm") or n.startswith("aten::mm") or n.startswith("aten::bmm")
           or any(k in n.lower() for k in ("silu", "tanh", "threshold"))]
    assert not bad, bad
    assert all("_bgk_layer_ops_t" in m.__dict__ for m in list(net._layers)[2::2] if isinstance(m, torch.nn.Linear)), "dX of every Linear behind the first on the kernel"
    for a, r in zip([p.grad for p in net.parameters()] + [xg.grad], g_ref):
        assert float((a - r).abs().max()) <= 1e-4 * max(float(r.abs().max()), 1e-6)


def test_readme_flow_launches_no_library_gemm(hip_lib, dev):
    """cfg 1 (README.md:54-96: RealNVP coupling with [1, 4, 1] conditioners): sampling and energy evaluation on the GPU launch no rocBLAS /
    hipBLASLt kernel -- the coupling is ONE launch (conditioners with one hidden layer: bgk_coupling_affine_dense_deep)"""
    from bgflow_amd import configs
    gen = configs.make_readme_generator(dev)
    names = []
    try:
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
            with torch.no_grad():
                x = gen.sample(1000)
                gen.energy(x)
            torch.cuda.synchronize()
        names = [e.key for e in prof.key_averages()]
    except Exception:               # no kernel tracer on this box: the plan below still shows which path ran
        with torch.no_grad():
            gen.energy(gen.sample(1000))
    plan = gen.flow[1].transformer._fused_cache
    assert plan.get("anydepth") and plan["depth"] == 2, "the coupling must have run as one launch (bgk_coupling_affine_dense_deep)"
    assert not any(("Cijk" in n) or ("gemm" in n.lower()) or n.startswith("aten::addmm") or n.startswith("aten::mm") for n in names), names


@pytest.mark.parametrize("n_out,n_in", [(4, 1), (130, 21), (425, 256), (77, 300)])
def test_linear_layer_device_packer_equals_the_host_packer(hip_lib, dev, n_out, n_in):
    """bgk_pack_linear_layer (no host synchronisation) writes the operand blocks and the unscale factor of dense.pack_linear_layer
    (the layout's reference, emulated on the CPU in tests/test_host_logic.py) bit for bit; a column-sliced weight packs like its copy"""
    from bgflow_amd import dense
    from bgflow_amd.utils import synth
    W = torch.as_tensor(synth(17 + n_in, n_out, n_in + 3, scale=0.7), device=dev)[:, 1:1 + n_in]
    ref = dense.pack_linear_layer(W)
    got = dense.pack_linear_layer_device(W)
    assert len(ref) == len(got) == (n_in + 255) // 256
    for (A, S, c, k0, k1), (Ad, Sd, cs, k0d, k1d) in zip(ref, got):
        assert (S, k0, k1) == (Sd, k0d, k1d) and float(cs[1]) == c and float(cs[0]) == float(W[:, k0:k1].abs().max())
        assert torch.equal(A.view(torch.int16), Ad.view(torch.int16))
    Wz = torch.zeros(5, 9, device=dev)
    (Az, _, csz, _, _), = dense.pack_linear_layer_device(Wz)
    assert float(csz[1]) == 1.0 and not Az.view(torch.int16).any()


# ---- conditioners with 1, 3, 4, ... hidden layers on the one-launch kernel (bgk_coupling_rqs_dense_deep) ----------------------------
@pytest.mark.parametrize("hidden", [(128,), (128, 128, 128), (64, 128, 32, 100), (96,) * 8])
@pytest.mark.parametrize("inverse", [False, True])
def test_spline_coupling_with_other_depths_runs_fused(hip_lib, dev, hidden, inverse):
    """spline conditioners with one, three, four and eight hidden layers (up to 128 units, narrower ones zero-padded) run as ONE launch in
    split-f16 mode: same function as the conditioner evaluated layer by layer, against the f64 oracle (non-periodic and periodic input,
    circular and non-circular splines, a batch that is not a multiple of the tile)"""
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
    circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
    t = lambda v: torch.as_tensor(v, dtype=torch.float32, device=dev)                                   # noqa: E731
    for what, on in (("TORSIONS", "FIXED"), ("BONDS", "TORSIONS")):
        layer_cpu = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden))
        layer = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden)).to(dev)
        B = 1037
        xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
        with warnings.catch_warnings():
            warnings.simplefilter("error")                     # a rejection (RuntimeWarning) would mean the layer-by-layer path ran
            with torch.no_grad():
                *outs, dl = layer(*[t(v) for v in xs], inverse=inverse)
        assert layer.transformer._fused_cache.get("deep") == len(hidden), "bgk_coupling_rqs_dense_deep must have run"
        ti = slot[what]
        outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
        np.testing.assert_allclose(outs[ti].cpu().numpy(), outs64[ti], rtol=0, atol=2e-5)
        np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=2e-5, atol=2e-5)
    # training runs the conditioner layer by layer: gradients flow
    xs_t = [t(v) for v in xs]
    *_, dl_t = layer(*xs_t, inverse=inverse)
    dl_t.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters() if p.requires_grad)


@pytest.mark.parametrize("n_bins", [4, 16])
def test_deep_conditioner_other_bin_counts_bins_and_round_trip(hip_lib, dev, n_bins):
    """three hidden layers with Tanh, K = 4 / 16: values against the f64 oracle, the round trip, the bin indices against the layer-by-layer
    path (ties at a knot aside)"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    d, d_c = 11, 23
    P = 3 * n_bins * d + d
    mk = lambda: hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(                       # noqa: E731
        bg.DenseNet([d_c, 128, 96, 128, P], activation=torch.nn.Tanh()), is_circular=False), transformed_indices=(1,), cond_indices=(0,)))
    layer_cpu, layer = mk(), mk().to(dev)
    B = 4099
    xs = [synth(B, B, d_c), synth(B + 5, B, d, uniform=True)]
    dx = [torch.as_tensor(v, dtype=torch.float32, device=dev) for v in xs]
    layer.transformer.return_bin_indices = True
    with torch.no_grad():
        _, y, dl = layer(*dx)
        bins = layer.transformer.last_bin_indices.clone()
        _, back, dl_back = layer(dx[0], y, inverse=True)
        layer.transformer.allow_fused = False
        _, y_ref, _ = layer(*dx)
        bins_ref = layer.transformer.last_bin_indices.clone()
        layer.transformer.allow_fused = True
    assert layer.transformer._fused_cache.get("deep") == 3
    outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], False, np.float64, [])
    np.testing.assert_allclose(y.cpu().numpy(), outs64[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(back.cpu().numpy(), xs[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose((dl + dl_back).cpu().numpy(), 0.0, atol=1e-4)
    assert float((y - y_ref).abs().max()) <= 4e-5
    assert float((bins != bins_ref).float().mean()) < 1e-3 and int((bins - bins_ref).abs().max()) <= 1


# ---- affine couplings whose conditioners have 1, 4, 5, ... hidden layers (bgk_coupling_affine_dense_deep) ----------------------------
@pytest.mark.parametrize("hidden,acts", [((4,), ("ReLU", "Tanh")), ((64,), ("SiLU", "SiLU")), ((128, 64, 32, 100), ("ReLU", "Tanh")),
                                         ((48,) * 5, ("Tanh", "Tanh")), ((128,) * 8, ("SiLU", "ReLU"))])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_coupling_with_other_depths_runs_fused(hip_lib, dev, hidden, acts, inverse):
    """affine couplings whose shift / scale networks have one, four, five or eight hidden layers (<= 128 units, zero-padded to 64 / 128)
    run as ONE launch: same function as the networks evaluated layer by layer, against the f64 oracle; a shift-only coupling too"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    t = lambda v: torch.as_tensor(v, dtype=torch.float32, device=dev)                                   # noqa: E731
    for with_scale in (True, False):
        mk = lambda: hash_init_(bg.CouplingFlow(bg.AffineTransformer(                           # noqa: E731
            bg.DenseNet([12, *hidden, 20], getattr(torch.nn, acts[0])()),
            bg.DenseNet([12, *hidden, 20], getattr(torch.nn, acts[1])()) if with_scale else None),
            transformed_indices=(1,), cond_indices=(0,)))
        layer_cpu, layer = mk(), mk().to(dev)
        B = 2111
        xs = [synth(B + 3 * i, B, d) for i, d in enumerate((12, 20))]
        with warnings.catch_warnings():
            warnings.simplefilter("error")                     # a rejection (RuntimeWarning) would mean the layer-by-layer path ran
            with torch.no_grad():
                _, y, dl = layer(*[t(v) for v in xs], inverse=inverse)
        plan = layer.transformer._fused_cache
        assert plan.get("anydepth") and plan["depth"] == len(hidden) + 1 and plan["hidden"] == (64 if max(hidden) <= 64 else 128)
        outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
        np.testing.assert_allclose(y.cpu().numpy(), outs64[1], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=2e-5, atol=2e-5)
    # under autograd the networks run layer by layer: gradients flow
    xg = [t(v) for v in xs]
    *_, dlg = layer(*xg, inverse=inverse)
    (dlg.sum() if with_scale else layer(*xg)[1].sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters() if p.requires_grad and p.grad is not None)


def test_envelope_layers_vs_reference_goldens(hip_lib, golden, dev):
    """the wide / deep one-launch kernels against outputs of the REFERENCE's CouplingFlow (tests/golden/g_envelope.npz, generated by
    importing bgflow): spline layers with hidden (256, 256), (200, 130), one / three / four hidden layers (periodic + circular and
    plain), affine layers with one / five / four hidden layers; both directions"""
    import envelope_layers as el
    G = golden("g_envelope")
    t = lambda v: torch.as_tensor(np.asarray(v), dtype=torch.float32, device=dev)                      # noqa: E731
    for tag, hidden in el.SPLINE.items():
        for kind, (periodic, circular) in el.KINDS.items():
            c, y = el.spline_inputs(periodic)
            layer = el.spline_layer(hidden, periodic, circular).to(dev)
            key = f"s_{tag}_{kind}_"
            with warnings.catch_warnings():
                warnings.simplefilter("error")                 # a rejection (RuntimeWarning) would mean the layer-by-layer path ran
                with torch.no_grad():
                    _, z, dl = layer(t(c), t(y))
                    _, yb, dli = layer(t(c), t(G[key + "z64"]), inverse=True)
            plan = layer.transformer._fused_cache
            assert (plan.get("hidden") == 256) if tag.startswith("w") else (plan.get("deep") == len(hidden)), (tag, kind)
            np.testing.assert_allclose(z.cpu().numpy(), G[key + "z64"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(dl.cpu().numpy(), G[key + "dlogp64"], rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(yb.cpu().numpy(), G[key + "back64"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(dli.cpu().numpy(), G[key + "dlogp_inv64"], rtol=2e-5, atol=2e-5)
    for tag, hidden in el.AFFINE.items():
        c, y = el.affine_inputs()
        layer = el.affine_layer(hidden).to(dev)
        key = f"a_{tag}_"
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            with torch.no_grad():
                _, z, dl = layer(t(c), t(y))
                _, yb, dli = layer(t(c), t(G[key + "z64"]), inverse=True)
        assert layer.transformer._fused_cache.get("anydepth"), tag
        np.testing.assert_allclose(z.cpu().numpy(), G[key + "z64"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dl.cpu().numpy(), G[key + "dlogp64"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(yb.cpu().numpy(), G[key + "back64"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dli.cpu().numpy(), G[key + "dlogp_inv64"], rtol=2e-5, atol=2e-5)
