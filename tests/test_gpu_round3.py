"""GPU (-m gpu), round 3: parity of LAUNCHES AT THE BASELINE BATCH SIZES against the CPU oracle (rows pulled out of one 2^20 /
2^18 launch, incl. the last tile and rows beyond 2^24 / d elements), the in-kernel running log-det, and coupling layers with several
conditioning tensors read in place (no torch.cat).  All kernels are reached through the C ABI (ctypes, bgflow_amd/_lib.py)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import assert_bin_ties, rel_per_sample, t

pytestmark = pytest.mark.gpu


def _rows(B, n=4096):
    """row sample of a launch: the first and last 64 rows (first / last tiles of every kernel), rows on both sides of the 2^24 / 17
    element boundary (24-bit index arithmetic), and an even spread over the batch"""
    edge = (1 << 24) // 17
    picks = [np.arange(0, 64), np.arange(B - 64, B), np.linspace(0, B - 1, n - 256).astype(np.int64)]
    if B > edge + 64:
        picks.append(np.arange(edge - 64, edge + 64))
    return np.unique(np.concatenate(picks))


def _make(cfg, dev=None):
    from bgflow_amd import configs
    make = {"cfg2": configs.make_affine8_generator, "cfg3": configs.make_ala2_spline_generator,
            "cfg5": configs.make_ala2_augmented_generator}[cfg]
    return make(dev) if cfg != "cfg2" else make(device=dev)


def _prior(cfg, B, dev, seed=1234):
    g = torch.Generator(device=dev).manual_seed(seed)
    if cfg == "cfg2":
        return [torch.randn(B, 64, device=dev, generator=g)]
    dims = (17, 17, 17, 9) + ((66,) if cfg == "cfg5" else ())
    return [torch.rand(B, d, device=dev, generator=g) for d in dims]


@pytest.mark.parametrize("cfg,B", [("cfg3", 1 << 20), ("cfg3", 1 << 18), ("cfg2", 1 << 20), ("cfg5", 1 << 20)])
def test_parity_of_rows_of_a_full_size_launch(hip_lib, oracle, dev, cfg, B):
    """ONE launch per block at the BASELINE batch (shipped split-f16 mode); 4096 rows of it against the oracle evaluated on the
    same rows: every coupling block fed the GPU's own inputs (outputs, bin indices, per-layer log-det), then the whole flow."""
    import bgflow_amd as bg
    from oracle import flow_oracle as fo
    from oracle import torch_flow as tfl
    gen, gen_cpu = _make(cfg, dev), _make(cfg)
    z = _prior(cfg, B, dev)
    rows = _rows(B)
    rows_t = torch.as_tensor(rows, device=dev)
    take = lambda state: [s[rows_t].cpu().numpy() for s in state]      # noqa: E731
    state = tuple(z)
    n_ties = n_el = 0
    worst_out = worst_dl = 0.0
    with torch.no_grad():
        for i, (block, block_cpu) in enumerate(zip(gen.flow, gen_cpu.flow)):
            ins = take(state)
            is_spline = isinstance(block, bg.CouplingFlow) and type(block.transformer).__name__ == "ConditionalSplineTransformer"
            if is_spline:
                block.transformer.return_bin_indices = True
            *state, dl = block(*state)
            if not isinstance(block, bg.CouplingFlow):
                continue
            last_coupling_out = take(state)          # (after the last coupling: what enters the icdf domain maps)
            ti = block.transformed_indices[0]
            got, got_dl = state[ti][rows_t].cpu().numpy(), dl[rows_t].cpu().numpy()
            outs64, dl64 = fo.run_block(block_cpu, [v.astype(np.float64) for v in ins], False, np.float64)
            trace = []
            outs32, dl32 = fo.run_block(block_cpu, ins, False, np.float32, trace)
            scale = max(1.0, float(np.abs(outs64[ti]).max()))
            e_out, e_out32 = np.abs(got - outs64[ti]).max(), np.abs(outs32[ti] - outs64[ti]).max()
            assert e_out <= 3 * e_out32 + 1e-6 * scale, f"block {i}: outputs {e_out:.2e} from the f64 oracle (f32 oracle: {e_out32:.2e})"
            e_dl = rel_per_sample(got_dl, dl64, floor=1.0)
            e_dl32 = rel_per_sample(dl32, dl64, floor=1.0)
            assert e_dl.max() <= max(1e-5, 3 * e_dl32.max() + 2e-6), f"block {i}: log-det {e_dl.max():.2e} (f32 oracle {e_dl32.max():.2e})"
            worst_out, worst_dl = max(worst_out, e_out / scale), max(worst_dl, e_dl.max())
            if is_spline:
                idx = block.transformer.last_bin_indices[rows_t].cpu().numpy()
                block.transformer.return_bin_indices = False
                n_ties += assert_bin_ties(idx, trace[0], ins[ti], f"block {i}")
                n_el += idx.size
        # pass-through tensors of a coupling are returned as they came in
        x_full = gen.flow(*z)
    assert n_ties <= max(2, n_el // 10000), f"{n_ties} knot ties in {n_el} elements"
    # ---- whole flow on the same rows (fused tail / fused coupling stack included), against the f64 oracle of the prior rows
    zr = take(z)
    x64, dlx64 = fo.run_flow(gen_cpu.flow, [v.astype(np.float64) for v in zr], dtype=np.float64)
    x32, dlx32 = fo.run_flow(gen_cpu.flow, zr, dtype=np.float32)
    *xg, dlg = x_full
    r_gpu = rel_per_sample(dlg[rows_t].cpu().numpy(), dlx64, floor=1.0)
    r_f32 = rel_per_sample(dlx32, dlx64, floor=1.0)
    # the reference's own arithmetic: its op chain on stock torch-CPU f32 ops (oracle/torch_flow.py; the C oracle evaluates the
    # icdf maps through double precision and is therefore better than any genuine f32 chain on the icdf tails)
    xt, dlt = tfl.run_flow(gen_cpu.flow, [torch.as_tensor(v) for v in zr])
    r_t32 = rel_per_sample(dlt.numpy(), dlx64, floor=1.0)
    frac_gpu, frac_f32, frac_t32 = float((r_gpu > 1e-5).mean()), float((r_f32 > 1e-5).mean()), float((r_t32 > 1e-5).mean())
    # (cfg 5: the Normal icdf of the 66 auxiliary variables puts every f32 evaluation ~1e-3 away from the f64 one)
    assert np.median(r_gpu) <= 1.5 * max(np.median(r_f32), np.median(r_t32)) + 2e-6, \
        f"log-det median error: GPU {np.median(r_gpu):.2e}, torch f32 chain {np.median(r_t32):.2e}, C f32 oracle {np.median(r_f32):.2e}"
    # icdf tails: an f32 evaluation cannot hold 1e-5 on every random sample (the reference's f32 op chain misses it on ~0.6 % of
    # uniform prior samples, the C oracle on ~0.3 %); the kernels must not miss it more often than 1.5 x the reference's chain
    assert frac_gpu <= 1.5 * max(frac_f32, frac_t32) + 2.0 / len(rows), \
        f"log-det beyond 1e-5: GPU {frac_gpu:.4f} of the rows, torch f32 chain {frac_t32:.4f}, C f32 oracle {frac_f32:.4f}"
    ex = np.abs(xg[0][rows_t].cpu().numpy() - x64[0]).max(-1)
    ex32 = np.maximum(np.abs(x32[0] - x64[0]).max(-1), np.abs(xt[0].numpy() - x64[0]).max(-1))
    assert np.median(ex) <= 3 * np.median(ex32) + 2e-6
    assert float((ex > 1e-4).mean()) <= 1.5 * float((ex32 > 1e-4).mean()) + 2.0 / len(rows)
    if cfg == "cfg3":
        # Next to the statistical bound: EVERY row whose icdf inputs lie away from the tails of the domain maps (|u - 0.5| < 0.49 in every
        # field: where an f32 evaluation of an icdf is well conditioned) holds north_star's 1e-5 on the log-det, per sample.
        core = np.all([np.abs(f - 0.5).max(-1) < 0.49 for f in last_coupling_out], axis=0)
        assert core.sum() > 0.2 * len(rows), f"only {core.sum()} of {len(rows)} rows away from the icdf tails"    # (0.98^60 = 0.30 of uniform rows)
        worst = float(r_gpu[core].max())
        print(f"{cfg} at B = {B}: {int(core.sum())} of {len(rows)} rows away from the icdf tails; worst log-det error on them {worst:.2e} "
              f"(torch f32 chain {float(r_t32[core].max()):.2e}, C f32 oracle {float(r_f32[core].max()):.2e})")
        assert worst <= 1e-5, f"log-det {worst:.2e} on a row away from the icdf tails"


@pytest.mark.parametrize("cfg", ["cfg3", "cfg5", "cfg2"])
@pytest.mark.parametrize("inverse", [False, True])
def test_running_logdet_in_kernels_equals_blockwise_sum(hip_lib, dev, cfg, inverse):
    """SequentialFlow with ONE running log-det buffer written by the kernels (`accumulate`) == the reference's
    `dlogp += ddlogp` over per-block tensors: identical outputs, log-det equal up to the order of the f32 additions"""
    import bgflow_amd as bg
    gen = _make(cfg, dev)
    B = 5000
    z = _prior(cfg, B, dev, seed=3)
    with torch.no_grad():
        *x, dl = gen.flow(*z)
        ins = x if inverse else z
        *o_acc, dl_acc = gen.flow(*ins, inverse=inverse)
        bg.SequentialFlow.ACCUMULATE_IN_KERNELS = False
        try:
            *o_ref, dl_ref = gen.flow(*ins, inverse=inverse)
        finally:
            bg.SequentialFlow.ACCUMULATE_IN_KERNELS = True
    assert dl_acc.shape == dl_ref.shape == (B, 1)
    for a, b in zip(o_acc, o_ref):
        assert torch.equal(a, b)
    scale = float(dl_ref.abs().max())
    assert float((dl_acc - dl_ref).abs().max()) <= 4e-6 * max(1.0, scale)


def test_running_logdet_protocol_details(hip_lib, dev):
    """nested SequentialFlows share the outer buffer; a user block without **kwargs never sees the private kwarg; gradients
    switch the protocol off; a temperature kwarg does not"""
    import bgflow_amd as bg
    from bgflow_amd import configs

    class Doubler(bg.Flow):                              # strict signature on purpose
        def _forward(self, *xs):
            return (*xs, torch.full((xs[0].shape[0], 1), 0.25, device=xs[0].device))

        def _inverse(self, *xs):
            return (*xs, torch.full((xs[0].shape[0], 1), -0.25, device=xs[0].device))

    gen = configs.make_ala2_spline_generator(dev)
    blocks = list(gen.flow)
    nested = bg.SequentialFlow([bg.SequentialFlow(blocks[:4]), Doubler(), bg.SequentialFlow(blocks[4:])])
    z = _prior("cfg3", 777, dev, seed=5)
    with torch.no_grad():
        x, dl = gen.flow(*z)
        xn, dln = nested(*z)
        xt, dlt = gen.flow(*z, temperature=1.0)
    assert torch.equal(x, xn) and torch.equal(x, xt) and torch.equal(dl, dlt)
    assert float((dln - dl - 0.25).abs().max()) <= 4e-6 * float(dl.abs().max())
    zg = [v.clone().requires_grad_(True) for v in z]
    xg, dlg = gen.flow(*zg)
    # (the gradient path runs the blocks one by one: on near-degenerate prior samples its tail differs from the fused one)
    assert dlg.requires_grad and float((dlg.detach() - dl).abs().median()) <= 1e-5 * float(dl.abs().max())


def test_coupling_with_several_conditioning_tensors_reads_them_in_place(hip_lib, dev):
    """CouplingFlow over cond_indices = (a, b, c): the fused kernels stage the tensors from their own rows (bgk_*_mc entry
    points) -- same bits as the torch.cat path, and no CatArrayBatchedCopy launch"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_
    gen = configs.make_ala2_augmented_generator(dev)
    multi = [b for b in gen.flow if isinstance(b, bg.CouplingFlow) and len(b.cond_indices) > 1]
    assert multi, "cfg 5 has AUGMENTED | (FIXED, BONDS, ANGLES) layers"
    z = _prior("cfg5", 3001, dev, seed=9)
    for inverse in (False, True):
        with torch.no_grad():
            ref_state = tuple(z)
            for block in multi:
                *a, dla = block(*ref_state, inverse=inverse)
                bg.CouplingFlow.MULTI_COND_IN_KERNEL = False
                try:
                    *b, dlb = block(*ref_state, inverse=inverse)
                finally:
                    bg.CouplingFlow.MULTI_COND_IN_KERNEL = True
                for u, v in zip(a, b):
                    assert torch.equal(u, v)
                assert torch.equal(dla, dlb)
    # a spline coupling conditioned on two tensors (second-generation kernel, K = 8), periodic and plain conditioner front ends
    for periodic in (False, True):
        n_in = (9 + 17) * (2 if periodic else 1)
        net = bg.DenseNet([n_in, 128, 128, 3 * 8 * 17 + 17], activation=torch.nn.SiLU())
        net = bg.WrapPeriodic(net) if periodic else net
        layer = hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(net, is_circular=False), transformed_indices=(0,),
                                           cond_indices=(3, 1))).to(dev)
        xs = _prior("cfg3", 2049, dev, seed=11)
        with torch.no_grad():
            *a, dla = layer(*xs)
            bg.CouplingFlow.MULTI_COND_IN_KERNEL = False
            try:
                *b, dlb = layer(*xs)
            finally:
                bg.CouplingFlow.MULTI_COND_IN_KERNEL = True
        assert layer.transformer._fused_cache.get("mode") == "f16x2", "the fused path must have run"
        assert torch.equal(a[0], b[0]) and torch.equal(dla, dlb)
    # no concatenation kernel in the timed path: the profiler sees none (first pass: the layers' operands are packed)
    with torch.no_grad():
        gen.flow(*z)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
        with torch.no_grad():
            gen.flow(*z)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert not any("CatArrayBatchedCopy" in n for n in names), "a torch.cat launch is left in the cfg-5 sampling pass"
    assert not any(n.startswith("aten::add") for n in names), "an aten add (log-det accumulation) is left in the cfg-5 sampling pass"


def test_envelope_warnings(hip_lib, dev):
    """a coupling whose DenseNet conditioner leaves the fused kernels' envelope says so, once"""
    import warnings
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    net = bg.DenseNet([9, 384, 384, 3 * 8 * 17 + 17], activation=torch.nn.SiLU())
    layer = hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(net, is_circular=False), transformed_indices=(0,),
                                       cond_indices=(3,))).to(dev)
    xs = _prior("cfg3", 100, dev, seed=13)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            layer(*xs)
            layer(*xs)
    msgs = [str(x.message) for x in w if "fused" in str(x.message)]
    assert len(msgs) == 1 and "(384, 384)" in msgs[0]


@pytest.mark.parametrize("hidden", [(64, 64), (32, 96), (100, 100)])
@pytest.mark.parametrize("mode", ["f16x2", "f32", "bf16"])
@pytest.mark.parametrize("inverse", [False, True])
def test_spline_coupling_with_narrow_hidden_layers_runs_fused(hip_lib, dev, hidden, mode, inverse):
    """hidden widths below 128 run on the one-launch kernels zero-padded to 128 (padded units hold act(0) = 0): same function as the
    conditioner evaluated layer by layer, checked against the f64 oracle"""
    import warnings
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
    circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
    for what, on in (("TORSIONS", "FIXED"), ("BONDS", "TORSIONS")):
        layer_cpu = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden))
        layer = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden)).to(dev)
        layer.transformer.gemm_mode = mode
        B = 1037
        xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
        with warnings.catch_warnings():
            warnings.simplefilter("error")                     # a rejection (RuntimeWarning) would mean the generic path ran
            with torch.no_grad():
                *outs, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
        assert layer.transformer._fused_cache.get("padded"), "the zero-padded fused path must have run"
        ti = slot[what]
        outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
        tol = 1e-2 if mode == "bf16" else 2e-5                 # bf16: the reduced-precision mode (weights and GEMM inputs in bf16)
        np.testing.assert_allclose(outs[ti].cpu().numpy(), outs64[ti], rtol=0, atol=tol)
        np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=tol, atol=tol)
    # training falls back to the layer-by-layer conditioner (the weight-gradient kernels take width 128 only): gradients flow
    layer.transformer.gemm_mode = "f16x2"
    xs_t = [t(v, dev) for v in xs]
    *_, dl_t = layer(*xs_t, inverse=inverse)
    dl_t.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters() if p.requires_grad)


@pytest.mark.parametrize("d,d_c,circ", [(40, 12, False), (64, 30, False), (64, 9, True), (33, 66, False)])
@pytest.mark.parametrize("inverse", [False, True])
def test_spline_coupling_many_transformed_dims(hip_lib, dev, d, d_c, circ, inverse):
    """the upper end of the fused spline kernel's envelope: up to 64 transformed dims (13 parameter chunks, one workgroup per CU) and a
    conditioner input wider than the layer-0 tile of the builder's layers"""
    import warnings
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    P = 3 * 8 * d + (0 if circ else d)
    mk = lambda: hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(                       # noqa: E731
        bg.DenseNet([d_c, 128, 128, P], activation=torch.nn.SiLU()), is_circular=circ), transformed_indices=(1,), cond_indices=(0,)))
    layer_cpu, layer = mk(), mk().to(dev)
    B = 777
    xs = [synth(B, B, d_c), synth(B + 5, B, d, uniform=True)]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            _, y, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
    assert layer.transformer._fused_cache, "the fused path must have run"
    outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
    np.testing.assert_allclose(y.cpu().numpy(), outs64[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("H,acts", [(32, ("ReLU", "Tanh")), (48, ("SiLU", "SiLU")), (96, ("SiLU", "SiLU")), (100, ("ReLU", "Tanh"))])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_coupling_with_other_hidden_widths_runs_fused(hip_lib, dev, H, acts, inverse):
    """hidden widths other than 64 / 128 run on the affine kernels zero-padded (<= 64 on the weight-resident kernel, else width 128)"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    mk = lambda: hash_init_(bg.CouplingFlow(bg.AffineTransformer(                               # noqa: E731
        bg.DenseNet([12, H, H, 20], getattr(torch.nn, acts[0])()), bg.DenseNet([12, H, H, 20], getattr(torch.nn, acts[1])())),
        transformed_indices=(1,), cond_indices=(0,)))
    layer_cpu, layer = mk(), mk().to(dev)
    B = 2111
    xs = [synth(B + 3 * i, B, d) for i, d in enumerate((12, 20))]
    with torch.no_grad():
        _, y, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
    assert layer.transformer._fused_cache and layer.transformer._fused_cache["hidden"] == (64 if H <= 64 else 128)
    outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
    np.testing.assert_allclose(y.cpu().numpy(), outs64[1], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------------------------------------------
# f-3: priors / targets / weights on kernels
# ---------------------------------------------------------------------------------------------------
def test_energy_kernels_vs_reference_goldens(hip_lib, golden, dev):
    """DoubleWellEnergy, NormalDistribution, UniformDistribution, ProductDistribution energies (one launch each, bgk_energy_fields) and
    the double-well force (bgk_energy_fields_backward) against values generated from the reference classes"""
    import bgflow_amd as bg
    G = golden("g_energies")
    x = t(G["dw_x"], dev)
    for tag, kw in (("dw", {}), ("dw_abc", dict(a=0.7, b=-2.5, c=0.4))):
        e = bg.DoubleWellEnergy(64, **kw)
        for key, T in (("_u", 1.0), ("_uT", 2.5)):
            u = e.energy(x, temperature=T)
            assert u.shape == (x.shape[0], 1)
            scale = np.abs(G[tag + key + "64"]).max()
            assert np.abs(u.cpu().numpy() - G[tag + key + "64"]).max() <= 3 * np.abs(G[tag + key + "32"] - G[tag + key + "64"]).max() + 2e-7 * scale
        f = e.force(x.clone(), temperature=2.5)
        np.testing.assert_allclose(f.cpu().numpy(), G[tag + "_force64"], rtol=2e-6, atol=2e-6)
    mean = t(G["norm_mean"], dev)
    comps = [bg.NormalDistribution(66, mean=mean), bg.NormalDistribution(66).to(dev),
             bg.UniformDistribution(t(G["unif_low"], dev), t(G["unif_high"], dev))]
    xs = tuple(t(G[k], dev) for k in ("prod_y", "prod_a", "prod_un"))
    prod = bg.ProductDistribution(comps)
    for got, key in ((comps[0].energy(xs[0], temperature=1.7), "norm_u"), (comps[2].energy(xs[2]), "unif_u"),
                     (prod.energy(*xs), "prod_u"), (prod.energy(*xs, temperature=1.7), "prod_uT")):
        ref64, ref32 = G[key + "64"], G[key + "32"]
        assert np.abs(got.cpu().numpy() - ref64).max() <= 3 * np.abs(ref32 - ref64).max() + 3e-7 * np.abs(ref64).max(), key
    # gradient of the product energy w.r.t. every tensor = autograd of the torch ops
    xr = [v.clone().requires_grad_(True) for v in xs]
    prod.energy(*xr, temperature=1.7).sum().backward()
    xc = [v.detach().cpu().double().requires_grad_(True) for v in xs]
    cpu = bg.ProductDistribution([bg.NormalDistribution(66, mean=mean.cpu().double()), bg.NormalDistribution(66).double(),
                                  bg.UniformDistribution(torch.tensor(G["unif_low"]).double(), torch.tensor(G["unif_high"]).double())])
    cpu.energy(*xc, temperature=1.7).sum().backward()
    for a, b in zip(xr[:2], xc[:2]):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("d,B", [(512, 1000), (4000, 257), (66, 1 << 18)])
def test_normal_energy_any_width(hip_lib, dev, d, B):
    """the energy kernel stages rows in column chunks: widths far beyond one LDS tile (a 1300-atom Cartesian target) work"""
    import bgflow_amd as bg
    g = torch.Generator(device=dev).manual_seed(d)
    x = torch.randn(B, d, device=dev, generator=g)
    mean = torch.randn(d, device=dev, generator=g)
    nd = bg.NormalDistribution(d, mean=mean)
    u = nd.energy(x, temperature=1.3)
    ref = 0.5 * ((x.double() - mean.double()) ** 2).sum(-1, keepdim=True) / 1.3 + d / 2 * np.log(2 * np.pi * 1.3)
    assert float(((u.double() - ref).abs() / ref.abs()).max()) <= 2e-6


def test_kl_loss_sums_in_the_energy_kernel(hip_lib, dev):
    """[sum (u - dlogp), n] formed by the target-energy kernel == the torch formula, with and without dropped samples, and the
    gradients to x and dlogp"""
    import bgflow_amd as bg
    from bgflow_amd import dp
    from bgflow_amd.distributions import kl_loss_sums
    g = torch.Generator(device=dev).manual_seed(3)
    B = 70001
    mean = torch.randn(66, device=dev, generator=g)
    target = bg.NormalDistribution(66, mean=mean)
    x = torch.randn(B, 66, device=dev, generator=g)
    dl = torch.randn(B, 1, device=dev, generator=g) * 3
    dl[5] = float("-inf"); dl[B - 1] = float("nan")
    for drop in (True, False):
        xr, dr = x.clone().requires_grad_(True), dl.clone().requires_grad_(True)
        sums, u = kl_loss_sums(target, (xr,), dr, temperature=1.0, drop_nonfinite=drop)
        loss = (0.5 * ((x.double() - mean.double()) ** 2).sum(-1, keepdim=True) + 33 * np.log(2 * np.pi)) - dl.double()
        ok = torch.isfinite(loss) if drop else torch.ones_like(loss, dtype=torch.bool)
        if drop:
            assert float(sums[1]) == float(ok.sum())
            assert abs(float(sums[0]) - float(loss[ok].sum())) <= 2e-6 * float(loss[ok].abs().sum())
            mean_loss = dp.global_mean_from_sums(sums)
            mean_loss.backward()
            n = float(ok.sum())
            gx_ref = torch.where(ok, (x - mean) / n, torch.zeros_like(x))
            np.testing.assert_allclose(xr.grad.cpu().numpy(), gx_ref.cpu().numpy(), rtol=1e-5, atol=1e-9)
            gd_ref = torch.where(ok, torch.full_like(dl, -1.0 / n), torch.zeros_like(dl))
            np.testing.assert_allclose(dr.grad.cpu().numpy(), gd_ref.cpu().numpy(), rtol=1e-5, atol=1e-12)
        else:
            assert float(sums[1]) == B and not np.isfinite(float(sums[0]))
    # BoltzmannGenerator.kldiv_mean == kldiv(...).mean() on the same samples
    from bgflow_amd import configs
    gen = configs.make_ala2_spline_generator(dev)
    torch.manual_seed(7)
    with torch.no_grad():
        a = float(gen.kldiv_mean(4096, drop_nonfinite=True))
        torch.manual_seed(7)
        k = gen.kldiv(4096)
        b = float(k[torch.isfinite(k)].mean())
    assert abs(a - b) <= 1e-5 * abs(b)


def test_fused_philox_prior_sampling(hip_lib, dev):
    """opt-in `sample_fused=True`: all tensors of a ProductDistribution sample + the prior energy from one launch of the counter-based
    generator; uniforms bit-exact against the numpy Philox restatement, normals to 2e-6; reproducible under torch.manual_seed"""
    import bgflow_amd as bg
    from oracle import philox
    mean = torch.linspace(-1, 1, 66, device=dev)
    low, high = torch.zeros(17, device=dev), torch.linspace(1.0, 3.0, 17, device=dev)
    prior = bg.ProductDistribution([bg.UniformDistribution(low, high), bg.NormalDistribution(66, mean=mean), bg.NormalDistribution(9).to(dev)],
                                   sample_fused=True)
    torch.manual_seed(1234)
    B = 5003
    u, x, z = prior.sample(B, temperature=1.5)
    assert u.shape == (B, 17) and x.shape == (B, 66) and z.shape == (B, 9)
    sid = prior._philox_state[0]                  # the object's stream id (construction order of fused-sampling objects in this process)
    seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (sid + 1)) & (2 ** 64 - 1)
    ru = philox.sample_field(seed, 0, 0, B, 17, 0)
    assert np.array_equal(u.cpu().numpy(), (low.cpu().numpy() + ru * (high - low).cpu().numpy()).astype(np.float32))
    rx = philox.sample_field(seed, 0, 1, B, 66, 1)
    np.testing.assert_allclose(x.cpu().numpy(), mean.cpu().numpy() + np.sqrt(1.5) * rx, rtol=0, atol=4e-6)
    rz = philox.sample_field(seed, 0, 2, B, 9, 1)
    np.testing.assert_allclose(z.cpu().numpy(), np.sqrt(1.5) * rz, rtol=0, atol=4e-6)
    # the energy that came out of the sampling launch == the energy evaluated on the tensors
    e_fused = prior.energy(u, x, z, temperature=1.5)
    prior._philox_last = None
    e_eval = prior.energy(u, x, z, temperature=1.5)
    np.testing.assert_allclose(e_fused.cpu().numpy(), e_eval.cpu().numpy(), rtol=3e-6, atol=1e-4)
    # a second call advances the stream; another object draws from its own stream; the same seed + the same stream state reproduce it
    u2, _, _ = prior.sample(B, temperature=1.5)
    assert not torch.equal(u, u2)
    prior2 = bg.ProductDistribution([bg.UniformDistribution(low, high), bg.NormalDistribution(66, mean=mean), bg.NormalDistribution(9).to(dev)],
                                    sample_fused=True)
    torch.manual_seed(1234)
    u3, x3, z3 = prior2.sample(B, temperature=1.5)
    assert not torch.equal(u, u3)
    prior2._philox_state[:] = [sid, 0]
    torch.manual_seed(1234)
    u3, x3, z3 = prior2.sample(B, temperature=1.5)
    assert torch.equal(u, u3) and torch.equal(x, x3) and torch.equal(z, z3)
    # default priors are untouched: torch's generator
    plain = bg.NormalDistribution(9).to(dev)
    torch.manual_seed(5); a = plain.sample(100)
    torch.manual_seed(5); b = torch.randn(100, 9, device=dev)
    assert torch.equal(a, b)


def test_flat_adam_step_count_checkpoint_and_direct_gradients(hip_lib, dev):
    """advisor items: a step skipped for a NaN gradient does not advance Adam's time step; state_dict / load_state_dict carry the
    moments; add_param_group is rejected; torch.autograd.grad through a fused training layer never touches .grad"""
    from bgflow_amd.training import FlatAdam
    from bgflow_amd import configs
    torch.manual_seed(0)
    pa = [torch.nn.Parameter(torch.randn(7, 5, device=dev)), torch.nn.Parameter(torch.randn(5, device=dev))]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = FlatAdam(pa, lr=1e-2, betas=(0.9, 0.99)), torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99))
    for it in range(5):
        oa.zero_grad(); ob.zero_grad()
        for ps in (pa, pb):
            sum(((p - 0.1 * (i + 1)) ** 2).sum() for i, p in enumerate(ps)).backward()
        if it == 2:                      # poisoned gradient: FlatAdam skips on the device, the reference loop would not call step()
            pa[1].grad[0] = float("nan")
            oa.step()
            continue
        oa.step(); ob.step()
    assert oa.skipped_steps() == 1
    for a, b in zip(pa, pb):             # four effective steps on both sides: same bias corrections
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=3e-6, atol=1e-7)
    sd = oa.state_dict()
    pc = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oc = FlatAdam(pc, lr=1e-2, betas=(0.9, 0.99))
    oc.load_state_dict(sd)
    assert torch.equal(oc.exp_avg, oa.exp_avg) and torch.equal(oc.exp_avg_sq, oa.exp_avg_sq) and oc._step == oa._step and oc.skipped_steps() == 1
    with pytest.raises(ValueError):
        oa.add_param_group(dict(params=[torch.nn.Parameter(torch.zeros(2, device=dev))]))
    # autograd.grad through the fused training forward: gradients are RETURNED, the bucket is not written
    gen = configs.make_ala2_spline_generator(dev)
    opt = FlatAdam([p for p in gen.flow.parameters()], lr=1e-5)
    opt.zero_grad()
    z = _prior("cfg3", 512, dev, seed=21)
    *x, dl = gen.flow(*z)
    w = gen.flow[0].transformer._params_net._layers[0].weight
    (gw,) = torch.autograd.grad(dl.sum(), [w])
    assert gw is not None and float(gw.abs().max()) > 0 and float(opt.grad.abs().max()) == 0.0
    *x, dl = gen.flow(*z)
    opt.backward(dl.sum())
    off = 0
    for p in opt._params:
        if p is w:
            break
        off += p.numel()
    np.testing.assert_allclose(opt.grad[off:off + w.numel()].view_as(w).cpu().numpy(), gw.cpu().numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("cfg", ["cfg3", "cfg5"])
def test_fused_inference_head_equals_blocks_and_oracle(hip_lib, oracle, dev, cfg):
    """NLL direction: xyz -> IC + whitening + 4 cdf maps as ONE launch (bgk_xyz2ic_cdf_uni) against (i) the same blocks one by one
    (bgk_ic_xyz2ic + 4 x bgk_cdf_transform), (ii) the f64 oracle on rows of a 2^18 launch"""
    import bgflow_amd as bg
    from oracle import flow_oracle as fo
    gen, gen_cpu = _make(cfg, dev), _make(cfg)
    assert gen.flow.segments(inverse=True)[0][0] == "xyz2ic+cdf"
    B = 1 << 18
    z = _prior(cfg, B, dev, seed=17)
    with torch.no_grad():
        *x, dl_f = gen.flow(*z)
        *zb, dl_i = gen.flow(*x, inverse=True)
        bg.SequentialFlow.FUSE_GENERATION_TAIL = False
        try:
            *zb_blocks, dl_i_blocks = gen.flow(*x, inverse=True)
        finally:
            bg.SequentialFlow.FUSE_GENERATION_TAIL = True
    # (i) fused head vs block path: same latent up to f32 noise of two different (both valid) evaluations; the bulk tightly
    for a, b in zip(zb, zb_blocks):
        dz = (a - b).abs().max(1).values
        assert float(dz.median()) <= 2e-6 and float(dz.quantile(0.99)) <= 2e-3
    r = rel_per_sample(dl_i.cpu().numpy(), dl_i_blocks.cpu().numpy(), floor=1.0)
    assert np.median(r) <= 2e-6 and np.quantile(r, 0.99) <= 1e-3
    # (no round-trip check here: random prior samples of a random-init flow are mostly near-degenerate geometries, where
    # xyz <-> IC does not round-trip in f32 in the reference either -- SURVEY.md Appendix C; the golden inputs do, see smoke())
    # (ii) rows of the launch vs the f64 oracle of the inverse direction, with the f32 oracle as the yardstick
    rows = _rows(B, 2048)
    rows_t = torch.as_tensor(rows, device=dev)
    xr = [v[rows_t].cpu().numpy() for v in x]
    z64, dl64 = fo.run_flow(gen_cpu.flow, [v.astype(np.float64) for v in xr], inverse=True, dtype=np.float64)
    z32, dl32 = fo.run_flow(gen_cpu.flow, xr, inverse=True, dtype=np.float32)
    r_gpu = rel_per_sample(dl_i[rows_t].cpu().numpy(), dl64, floor=1.0)
    r_f32 = rel_per_sample(dl32, dl64, floor=1.0)
    assert np.median(r_gpu) <= 1.5 * np.median(r_f32) + 2e-6
    assert float((r_gpu > 1e-5).mean()) <= 1.5 * float((r_f32 > 1e-5).mean()) + 4.0 / len(rows), \
        f"beyond 1e-5: GPU {(r_gpu > 1e-5).mean():.4f}, f32 oracle {(r_f32 > 1e-5).mean():.4f}"
    for k in range(4):
        ez = np.abs(zb[k][rows_t].cpu().numpy() - z64[k]).max(-1)
        ez32 = np.abs(z32[k] - z64[k]).max(-1)
        assert np.median(ez) <= 3 * np.median(ez32) + 2e-6


def test_two_rank_rccl_kl_step(hip_lib, dev):
    """2 processes x 2 GPUs over RCCL (skipped on a one-GPU box): the sharded KL step of tests/_two_rank_worker.py"""
    import os
    import socket
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_two_rank_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "TWO_RANK_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg5"])
def test_empty_and_single_sample_batches(hip_lib, dev, cfg):
    """B = 0 and B = 1 through every kernel of a flow, both directions (fused tail / head, running log-det, coupling stacks): an empty
    batch returns empty tensors of the right widths, a single sample equals row 0 of a larger batch"""
    gen = _make(cfg, dev)
    xs8 = _prior(cfg, 8, dev, seed=3)
    with torch.no_grad():
        *ys8, dl8 = gen.flow(*xs8)
        for B in (0, 1):
            xs = tuple(x[:B].contiguous() for x in xs8)
            *ys, dl = gen.flow(*xs)
            assert dl.shape == (B, 1) and all(y.shape == (B, y8.shape[1]) for y, y8 in zip(ys, ys8))
            if B:
                assert torch.allclose(dl, dl8[:1], rtol=1e-6, atol=1e-6) and all(torch.allclose(y, y8[:1], atol=1e-6) for y, y8 in zip(ys, ys8))
            *zs, dli = gen.flow(*[y.clone() for y in ys], inverse=True)
            assert dli.shape == (B, 1) and all(z.shape == (B, x8.shape[1]) for z, x8 in zip(zs, xs8))


@pytest.mark.parametrize("B", [1, 2, 3, 5, 63, 65, 127, 1001, 4133])
@pytest.mark.parametrize("cfg", ["cfg3", "cfg5"])
def test_ragged_batches_through_the_fused_tail_and_head(hip_lib, dev, cfg, B):
    """batch sizes that are no multiple of the 64-sample tiles (nor of the 16-byte DMA pieces): the fused sampling tail and its
    inverse-direction twin against the same blocks run one by one, every row"""
    gen = _make(cfg, dev)
    xs = _prior(cfg, B, dev, seed=B)
    with torch.no_grad():
        *ys, dl = gen.flow(*xs)
        *zs, dli = gen.flow(*[y.clone() for y in ys], inverse=True)
        gen.flow.FUSE_GENERATION_TAIL = False
        try:
            *ys_b, dl_b = gen.flow(*xs)
            *zs_b, dli_b = gen.flow(*[y.clone() for y in ys], inverse=True)
        finally:
            gen.flow.FUSE_GENERATION_TAIL = True
    # The two paths use different (equally accurate) erfinv / sincos / atan2 forms, and a uniform prior sample now and then lands
    # on a near-degenerate geometry (tiny bond, flat angle) that amplifies the last-ulp difference: a few rows may differ visibly.
    # A mis-staged tile edge is wrong by O(1) in the LAST rows of the batch, always: those must agree, and bad rows must stay rare.
    def row_err(a, b):
        a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
        return (np.abs(a - b) / (1e-3 + 1e-4 * np.abs(b))).reshape(a.shape[0], -1).max(-1)
    err_f = np.max([row_err(a, b) for a, b in zip(ys, ys_b)], axis=0)
    err_f = np.maximum(err_f, row_err(dl, dl_b) / 10.0)
    err_i = np.max([row_err(a, b) for a, b in zip(zs, zs_b)], axis=0)
    for name, err in (("forward", err_f), ("inverse", err_i)):
        assert (err > 1.0).mean() <= 0.02, f"{name}: {int((err > 1.0).sum())} of {B} rows differ"
        assert (err[-min(B, 4):] <= 1.0).all(), f"{name}: the last rows of the batch differ: {err[-4:]}"


@pytest.mark.parametrize("B", [1, 3, 65, 1001])
@pytest.mark.parametrize("variant", ["register", "lds_table"])
def test_ragged_batches_through_the_other_tail_kernels(hip_lib, dev, B, variant):
    """the per-channel-descriptor tail kernel (bgk_icdf_ic2xyz_reg) and the round-2 kernel (bgk_icdf_ic2xyz), reached by switching the
    newer ones off, on batch sizes that fill no tile"""
    from bgflow_amd import ic as icmod
    gen = _make("cfg3", dev)
    xs = _prior("cfg3", B, dev, seed=100 + B)
    cls = icmod.RelativeInternalCoordinateTransformation
    saved = (cls.UNIFORM_TAIL, cls.REGISTER_TAIL)
    try:
        cls.UNIFORM_TAIL = False
        cls.REGISTER_TAIL = variant == "register"
        with torch.no_grad():
            *ys, dl = gen.flow(*xs)
            gen.flow.FUSE_GENERATION_TAIL = False
            *ys_b, dl_b = gen.flow(*xs)
    finally:
        cls.UNIFORM_TAIL, cls.REGISTER_TAIL = saved
        gen.flow.FUSE_GENERATION_TAIL = True
    err = (np.abs(ys[0].cpu().numpy() - ys_b[0].cpu().numpy()) / (1e-3 + 1e-4 * np.abs(ys_b[0].cpu().numpy()))).max(-1)
    assert (err > 1.0).mean() <= 0.02 and (err[-min(B, 4):] <= 1.0).all(), f"rows differ: {np.nonzero(err > 1.0)[0][:8]}"
    assert (np.abs(dl.cpu().numpy() - dl_b.cpu().numpy()) / np.maximum(np.abs(dl_b.cpu().numpy()), 1.0) <= 1e-2).mean() >= 0.98


@pytest.mark.parametrize("B", [1, 2, 63, 65, 1000])
def test_energy_kernels_on_ragged_batches_and_strided_rows(hip_lib, dev, B):
    """bgk_energy_fields on row views of wider tensors (row stride != width), small and ragged batches, several fields at once"""
    import bgflow_amd as bg
    g = torch.Generator(device=dev).manual_seed(B)
    wide = torch.randn(B, 200, device=dev, generator=g)
    xa, xb, xc = wide[:, 3:20], wide[:, 20:117], wide[:, 120:121]          # widths 17, 97 (two column chunks), 1
    mean = torch.randn(97, device=dev, generator=g)
    comps = [bg.NormalDistribution(17).to(dev), bg.NormalDistribution(97, mean=mean), bg.NormalDistribution(1).to(dev)]
    prod = bg.ProductDistribution(comps)
    u = prod.energy(xa, xb, xc, temperature=0.7)
    # the reference's ProductDistribution has no temperature-aware components: Energy.energy divides the T = 1 sum (product.py:119-131)
    ref = sum(0.5 * ((x.double() - m) ** 2).sum(-1, keepdim=True) + x.shape[1] / 2 * np.log(2 * np.pi)
              for x, m in ((xa, 0.0), (xb, mean.double()), (xc, 0.0))) / 0.7
    assert u.shape == (B, 1) and float(((u.double() - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= 2e-6
    dw = bg.DoubleWellEnergy(17, a=0.3, b=-2.0, c=0.7)
    ud = dw.energy(xa, temperature=1.9)
    xd = xa.double()
    refd = (0.3 * xd[:, :1] - 2.0 * xd[:, :1] ** 2 + 0.7 * xd[:, :1] ** 4 + 0.5 * (xd[:, 1:] ** 2).sum(-1, keepdim=True)) / 1.9
    assert float(((ud.double() - refd).abs() / refd.abs().clamp_min(1.0)).max()) <= 2e-6


@pytest.mark.parametrize("B", [1, 65])
@pytest.mark.parametrize("inverse", [False, True])
def test_smallest_shapes_through_the_fused_couplings(hip_lib, dev, B, inverse):
    """one transformed dim conditioned on one dim (and 2 | 3): the degenerate end of the fused kernels' envelope, spline and affine,
    against the f64 oracle"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_, synth
    from oracle import flow_oracle as fo
    for d_c, d in ((1, 1), (3, 2)):
        for circ in (False, True):
            mk = lambda: hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(                      # noqa: E731
                bg.DenseNet([d_c, 128, 128, 3 * 8 * d + (0 if circ else d)], activation=torch.nn.SiLU()), is_circular=circ),
                transformed_indices=(1,), cond_indices=(0,)))
            layer_cpu, layer = mk(), mk().to(dev)
            xs = [synth(B, B, d_c), synth(B + 1, B, d, uniform=True)]
            with torch.no_grad():
                _, y, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
            assert layer.transformer._fused_cache, "the fused spline path must have run"
            o64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
            np.testing.assert_allclose(y.cpu().numpy(), o64[1], rtol=0, atol=2e-5)
            np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=2e-5, atol=2e-5)
        for H in (64, 128):
            mk = lambda: hash_init_(bg.CouplingFlow(bg.AffineTransformer(                                   # noqa: E731
                bg.DenseNet([d_c, H, H, d], torch.nn.ReLU()), bg.DenseNet([d_c, H, H, d], torch.nn.Tanh())),
                transformed_indices=(1,), cond_indices=(0,)))
            layer_cpu, layer = mk(), mk().to(dev)
            xs = [synth(B, B, d_c), synth(B + 1, B, d)]
            with torch.no_grad():
                _, y, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
            assert layer.transformer._fused_cache, "the fused affine path must have run"
            o64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
            np.testing.assert_allclose(y.cpu().numpy(), o64[1], rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("B", [3, 65, 1001])
def test_tail_with_per_channel_marginals(hip_lib, dev, B):
    """data-derived marginals (a mean and a width PER CHANNEL, icmarginals.py:41-77 with `InternalCoordinateMarginals` fitted to data):
    the descriptors differ inside a field, so the sampling tail runs on the per-channel kernel (bgk_icdf_ic2xyz_reg) -- against the
    same blocks one by one; the inverse direction has no fused kernel for this case and must agree trivially"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    gen = _make("cfg3", dev)
    blocks = list(gen.flow._blocks)
    rs = np.random.RandomState(5)
    vec = lambda lo, hi, n: torch.as_tensor(lo + (hi - lo) * rs.rand(n), dtype=torch.float32, device=dev)     # noqa: E731
    one = lambda v: torch.as_tensor(v, dtype=torch.float32, device=dev)                                        # noqa: E731
    marginals = {
        0: bg.TruncatedNormalDistribution(mu=vec(0.9, 1.6, 17), sigma=vec(0.02, 0.3, 17), lower_bound=one(1e-5), upper_bound=one(np.inf)),
        1: bg.TruncatedNormalDistribution(mu=vec(0.3, 0.7, 17), sigma=vec(0.05, 0.4, 17), lower_bound=one(1e-5), upper_bound=one(1.0)),
        2: configs.SloppyUniform(low=torch.zeros(17, device=dev), high=torch.ones(17, device=dev)),
        3: configs._NormalMarginal(vec(-1.0, 1.0, 9), vec(5.0, 20.0, 9)).to(dev),
    }
    n_maps = sum(type(getattr(getattr(b, "_flow", None), "_delegate", None)).__name__ == "CDFTransform" for b in blocks)
    assert n_maps == 4
    blocks = blocks[:-1 - n_maps] + [bg.WrapFlow(bg.InverseFlow(bg.CDFTransform(marginals[s_])), (s_,)) for s_ in range(4)] + blocks[-1:]
    flow = bg.SequentialFlow(blocks)                              # fresh segment / descriptor caches
    xs = _prior("cfg3", B, dev, seed=B)
    with torch.no_grad():
        *ys, dl = flow(*xs)
        labels = [lbl for lbl, _ in flow.segments()]
        flow.FUSE_GENERATION_TAIL = False
        *ys_b, dl_b = flow(*xs)
        flow.FUSE_GENERATION_TAIL = True
        *zs, dli = flow(*[y.clone() for y in ys_b], inverse=True)
        flow.FUSE_GENERATION_TAIL = False
        *zs_b, dli_b = flow(*[y.clone() for y in ys_b], inverse=True)
    assert labels[-1] == "icdf+ic2xyz"
    err = (np.abs(ys[0].cpu().numpy() - ys_b[0].cpu().numpy()) / (1e-3 + 1e-4 * np.abs(ys_b[0].cpu().numpy()))).max(-1)
    assert (err > 1.0).mean() <= 0.02 and (err[-min(B, 4):] <= 1.0).all(), f"rows differ: {np.nonzero(err > 1.0)[0][:8]}"
    assert (np.abs(dl.cpu().numpy() - dl_b.cpu().numpy()) / np.maximum(np.abs(dl_b.cpu().numpy()), 1.0) <= 1e-2).mean() >= 0.98
    for a, b in zip(zs, zs_b):
        erri = (np.abs(a.cpu().numpy() - b.cpu().numpy()) / (1e-3 + 1e-4 * np.abs(b.cpu().numpy()))).max(-1)
        assert (erri > 1.0).mean() <= 0.02


def test_one_launch_of_four_million_samples_equals_its_chunks(hip_lib, dev):
    """B = 2^22 + 77 in ONE launch per kernel (cfg 4's global batch on one GPU: 2.8e8 coordinates, 64-bit row offsets everywhere)
    against the same rows run as four chunks: identical bits, both directions"""
    gen = _make("cfg3", dev)
    B = (1 << 22) + 77
    xs = _prior("cfg3", B, dev, seed=4)
    cuts = [0, 1 << 20, (1 << 21) + 13, 3 << 20, B]
    with torch.no_grad():
        *ys, dl = gen.flow(*xs)
        for a, b in zip(cuts[:-1], cuts[1:]):
            *yc, dlc = gen.flow(*[x[a:b].contiguous() for x in xs])
            assert torch.equal(yc[0], ys[0][a:b]) and torch.equal(dlc, dl[a:b]), f"rows {a}:{b} differ"
        x_last = ys[0][-(1 << 20):].contiguous()
        *zs, dli = gen.flow(ys[0], inverse=True)
        *zc, dlic = gen.flow(x_last, inverse=True)
        assert all(torch.equal(u, v[-(1 << 20):]) for u, v in zip(zc, zs)) and torch.equal(dlic, dli[-(1 << 20):])


@pytest.mark.parametrize("widths", [(1, 2, 5), (1, 1, 1), (31, 1), (2, 63)])
@pytest.mark.parametrize("periodic", [False, True])
def test_segment_tables_with_narrow_and_odd_widths(hip_lib, dev, widths, periodic):
    """conditioner input from 2 - 3 tensors of widths down to 1 (segment staging: per-segment magic division, row offsets): same bits as
    the concatenated input, spline and affine couplings, ragged batch"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    B, d = 1031, 7
    n_c = sum(widths)
    g = torch.Generator(device=dev).manual_seed(n_c)
    conds = [torch.rand(B, w, device=dev, generator=g) for w in widths]
    y = torch.rand(B, d, device=dev, generator=g)
    front = (lambda n: bg.WrapPeriodic(n)) if periodic else (lambda n: n)
    n_in = n_c * (2 if periodic else 1)
    layers = [
        bg.CouplingFlow(bg.ConditionalSplineTransformer(front(bg.DenseNet([n_in, 128, 128, 3 * 8 * d + d], activation=torch.nn.SiLU())),
                                                        is_circular=False), transformed_indices=(len(widths),), cond_indices=tuple(range(len(widths)))),
        bg.CouplingFlow(bg.AffineTransformer(front(bg.DenseNet([n_in, 128, 128, d], torch.nn.SiLU())), front(bg.DenseNet([n_in, 128, 128, d], torch.nn.SiLU()))),
                        transformed_indices=(len(widths),), cond_indices=tuple(range(len(widths)))),
    ]
    import warnings
    for li, layer in enumerate(layers):
        layer = hash_init_(layer).to(dev)
        limit = 111 if li == 0 else 127          # features the layer-0 tile of the spline / affine kernels holds: wider inputs run
        beyond = n_in > limit                    # layer by layer, with one warning
        for inverse in (False, True):
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                with torch.no_grad():
                    *a, dla = layer(*conds, y, inverse=inverse)
                    bg.CouplingFlow.MULTI_COND_IN_KERNEL = False
                    try:
                        *b, dlb = layer(*conds, y, inverse=inverse)
                    finally:
                        bg.CouplingFlow.MULTI_COND_IN_KERNEL = True
            said = [str(x.message) for x in w if "fused" in str(x.message)]
            if beyond:
                assert inverse or (len(said) == 1 and str(limit) in said[0]), said      # said once, with the reason; no exception
            else:
                assert not said and layer.transformer._fused_cache, "the fused path must have run"
            assert torch.equal(a[-1], b[-1]) and torch.equal(dla, dlb), f"{type(layer.transformer).__name__} inverse={inverse}"


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg5"])
def test_generator_api_keyword_routing(hip_lib, dev, cfg):
    """BoltzmannGenerator.sample(temperature=..., with_dlogp / with_latent / with_log_weights) and .energy / .kldiv through the fused
    segments (bg.py:105-147): keyword arguments reach the prior, not the kernels; shapes as in the reference"""
    gen = _make(cfg, dev)
    torch.manual_seed(0)
    n = 257
    latent = cfg == "cfg2"        # (like the reference, `with_latent` works for single-tensor priors only: bg.py:120-121 appends *z)
    with torch.no_grad():
        out = gen.sample(n, temperature=1.7, with_latent=latent, with_dlogp=True, with_energy=True, with_log_weights=True, with_weights=True)
    xs_out = out[:len(out) - 4 - int(latent)]          # the flow's outputs (cfg 5: coordinates and the auxiliary variables)
    x = out[0]
    assert x.shape[0] == n and all(o.shape[0] == n for o in out)
    dlogp, energy, logw, w = out[-4], out[-3], out[-2], out[-1]
    assert dlogp.shape == (n, 1) and energy.shape == (n, 1) and logw.shape == (n, 1) and w.shape == (n,)
    assert torch.isfinite(dlogp).all() and abs(float(w.sum()) - 1.0) < 1e-3
    with torch.no_grad():
        e = gen.energy(*xs_out, temperature=1.7)
        kl = gen.kldiv(64, temperature=1.3)
    assert e.shape == (n, 1) and kl.shape == (64, 1)
    # the generator's energy of its own sample = prior energy of the latent - log-det (bg.py:105-123), within the inverse's accuracy
    assert torch.isfinite(e).float().mean() > 0.98


@pytest.mark.parametrize("d_c,periodic", [(100, False), (50, True), (97, False), (111, False)])
def test_training_gradients_with_wide_conditioner_inputs(hip_lib, dev, d_c, periodic):
    """conditioner inputs of 97 .. 111 features: the fused training forward takes them, the input-gradient kernel stops at 96 (the chain
    then runs on GEMMs): every gradient against the layer-by-layer autograd path"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    B, d = 333, 11
    n_in = d_c * (2 if periodic else 1)
    res = {}
    for fused in (True, False):
        net = bg.DenseNet([n_in, 128, 128, 3 * 8 * d + d], activation=torch.nn.SiLU())
        net = bg.WrapPeriodic(net) if periodic else net
        layer = hash_init_(bg.CouplingFlow(bg.ConditionalSplineTransformer(net, is_circular=False), transformed_indices=(1,),
                                           cond_indices=(0,))).to(dev)
        layer.transformer.allow_fused = fused
        g = torch.Generator(device=dev).manual_seed(7)
        c = torch.rand(B, d_c, device=dev, generator=g).requires_grad_(True)
        y = torch.rand(B, d, device=dev, generator=g).requires_grad_(True)
        w = torch.randn(B, d, device=dev, generator=g)
        _, out, dl = layer(c, y)
        if fused:
            assert layer.transformer._fused_cache.get("src_col_dev") is not None, "the fused training forward must have run"
        ((out * w).sum() + dl.sum()).backward()
        res[fused] = [out.detach(), dl.detach(), c.grad, y.grad] + [p.grad for p in layer.parameters()]
    for a, b in zip(res[True], res[False]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=3e-4 * float(b.abs().max()) + 1e-6)


@pytest.mark.parametrize("d", [1, 2, 66, 1500, 3000])
def test_cdf_maps_of_any_width(hip_lib, dev, d):
    """CDFTransform over per-column normal marginals, one column to 3000 (beyond 1800 columns the kernel keeps its column constants in
    global memory), a batch that fills no tile: both directions against torch.distributions in f64"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    B = 77
    g = torch.Generator(device=dev).manual_seed(d)
    loc = torch.randn(d, device=dev, generator=g)
    scale = 0.5 + torch.rand(d, device=dev, generator=g)
    layer = bg.CDFTransform(configs._NormalMarginal(loc, scale)).to(dev)
    u = torch.rand(B, d, device=dev, generator=g).clamp(1e-4, 1 - 1e-4)
    ref = torch.distributions.Normal(loc.double(), scale.double())
    with torch.no_grad():
        y, dl = layer(u, inverse=True)
        assert layer._desc_cache.get("desc") is not None, "the kernel path must have run"
        y_t = ref.icdf(u.double())
        np.testing.assert_allclose(y.cpu().numpy(), y_t.cpu().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dl.cpu().numpy(), (-ref.log_prob(y_t)).sum(-1, keepdim=True).cpu().numpy(), rtol=3e-5, atol=3e-4)
        ub, dlb = layer(y)
        np.testing.assert_allclose(ub.cpu().numpy(), u.cpu().numpy(), rtol=0, atol=3e-6)
        np.testing.assert_allclose(dlb.cpu().numpy(), ref.log_prob(y.double()).sum(-1, keepdim=True).cpu().numpy(), rtol=3e-5, atol=3e-4)


@pytest.mark.parametrize("B,d", [(1, 1), (1, 5), (63, 3), (65, 17), (130, 4), (129, 97)])
def test_philox_fields_small_and_ragged_shapes(hip_lib, dev, B, d):
    """the opt-in counter-based prior on shapes that fill neither a 64-row tile nor a 4-column counter block: uniforms bit-exact
    against the numpy restatement, and the same numbers whatever the batch is split into (the counter is the GLOBAL row)"""
    import bgflow_amd as bg
    from oracle import philox
    low, high = torch.zeros(d, device=dev), torch.ones(d, device=dev) * 2.0
    prior = bg.ProductDistribution([bg.UniformDistribution(low, high), bg.NormalDistribution(d).to(dev)], sample_fused=True)
    torch.manual_seed(99)
    u, z = prior.sample(B)
    seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (prior._philox_state[0] + 1)) & (2 ** 64 - 1)    # key = seed mixed with the object's stream id
    assert np.array_equal(u.cpu().numpy(), (2.0 * philox.sample_field(seed, 0, 0, B, d, 0)).astype(np.float32))
    np.testing.assert_allclose(z.cpu().numpy(), philox.sample_field(seed, 0, 1, B, d, 1), rtol=0, atol=4e-6)
    assert torch.isfinite(prior.energy(u, z)).all()


@pytest.mark.parametrize("n_atoms", [5, 8, 24, 25, 32, 33, 40])
def test_tail_and_head_on_other_molecule_sizes(hip_lib, dev, n_atoms):
    """the fused sampling tail / inference head across the kernels' atom-count instances (<= 24: the elementwise kernels; <= 32: the
    per-channel register kernel; beyond: the round-2 kernel / the blocks) on a synthetic chain, ragged batch: against the blocks"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    z = np.zeros((n_atoms - 3, 4), dtype=np.int64)
    for k, i in enumerate(range(3, n_atoms)):
        z[k] = (i, i - 1, i - 2, i - 3) if (i % 4 or i < 4) else (i, i - 2, i - 3, i - 4)
    n = n_atoms - 3
    ic = bg.RelativeInternalCoordinateTransformation(z, np.array([0, 1, 2]), normalize_angles=True)
    one = lambda v, m: torch.full((m,), float(v))          # noqa: E731
    marginals = {
        0: bg.TruncatedNormalDistribution(mu=one(0.15, n), sigma=one(0.01, n), lower_bound=torch.tensor(1e-5), upper_bound=torch.tensor(np.inf)),
        1: bg.TruncatedNormalDistribution(mu=one(0.5, n), sigma=one(0.05, n), lower_bound=torch.tensor(1e-5), upper_bound=torch.tensor(1.0)),
        2: configs.SloppyUniform(low=one(0.0, n), high=one(1.0, n)),
        3: configs._NormalMarginal(torch.tensor([0, 0, 0, 0.15, 0, 0, 0.2, 0.14, 0.0]), one(0.005, 9)),
    }
    flow = bg.SequentialFlow([bg.WrapFlow(bg.InverseFlow(bg.CDFTransform(marginals[s_])), (s_,)) for s_ in range(4)]
                             + [bg.WrapFlow(bg.InverseFlow(ic), indices=[0, 1, 2, 3], out_indices=(0,))]).to(dev)
    B = 77
    g = torch.Generator(device=dev).manual_seed(n_atoms)
    xs = [torch.rand(B, w, device=dev, generator=g).clamp(0.02, 0.98) for w in (n, n, n, 9)]
    with torch.no_grad():
        x, dl = flow(*xs)
        *zs, dli = flow(x, inverse=True)
        flow.FUSE_GENERATION_TAIL = False
        x_b, dl_b = flow(*xs)
        *zs_b, dli_b = flow(x, inverse=True)
    assert x.shape == (B, 3 * n_atoms)
    np.testing.assert_allclose(x.cpu().numpy(), x_b.cpu().numpy(), rtol=0, atol=2e-5 * n_atoms)
    np.testing.assert_allclose(dl.cpu().numpy(), dl_b.cpu().numpy(), rtol=2e-5, atol=2e-3)
    for a, b in zip(zs, zs_b):
        da = np.abs(a.cpu().numpy() - b.cpu().numpy())
        assert np.minimum(da, 1.0 - da).max() <= 2e-4          # (torsions are periodic)
    np.testing.assert_allclose(dli.cpu().numpy(), dli_b.cpu().numpy(), rtol=1e-4, atol=2e-2)
    # and the round trip closes
    for a, b in zip(zs, xs):
        da = np.abs(a.cpu().numpy() - b.cpu().numpy())
        assert np.minimum(da, 1.0 - da).max() <= 5e-3
