"""GPU (-m gpu): the HIP kernels, called through the C ABI (ctypes), against
 (a) the CPU oracle on the same seeded inputs -- bit-exact bin indices (and, because both sides use
     the same deterministic f32 primitives in the same order, bit-exact outputs for spline/affine),
 (b) the committed golden vectors generated from the reference,
 (c) size-independent properties at BASELINE batch sizes (forward o inverse round trips, dlogp
     cancellation, identity at zero parameters)."""
import numpy as np
import pytest
import torch

from bgflow_amd.utils import synth

pytestmark = pytest.mark.gpu
K = 8


def t(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev)


def test_detmath_device_equals_host(hip_lib, oracle, dev):
    """the deterministic exp/log/softplus/silu/tanh are bit-identical on gfx950 and on the host"""
    from bgflow_amd import _lib
    x = np.concatenate([synth(1, 1 << 16, scale=8.0), synth(2, 4096, scale=40.0), np.array([0.0, -0.0, 1.0, -87.5, 88.5, 20.0, 28.9, -79.5, -80.5, 79.5, 80.5], np.float32)])
    for which, code in (("exp", 0), ("log", 1), ("softplus", 2), ("silu", 3), ("tanh", 4)):
        xin = np.abs(x) + np.float32(1e-30) if which == "log" else x
        ref = oracle.detmath_probe(xin, which)
        xd = t(xin, dev)
        out = torch.empty_like(xd)
        st = hip_lib.bgk_detmath_probe(_lib.ptr(xd), xd.numel(), code, _lib.ptr(out), _lib.stream_ptr(dev))
        assert st == 0
        got = out.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"{which}: {np.sum(got != ref)} of {got.size} differ"


UNIT_CASES = [("nc17", 17, np.zeros(17, bool)), ("c17", 17, np.ones(17, bool)), ("nc9", 9, np.zeros(9, bool)),
              ("mix6", 6, np.array([1, 0, 1, 1, 0, 0], bool))]


def run_rqs_hip(y, params, circ, inverse, dev, **kw):
    from bgflow_amd.transformer import rqs_transform
    from oracle.oracle import nc_slots
    d = y.shape[1]
    slots = t(nc_slots(circ, d), dev)
    P = params.shape[1]
    n_nc = int((~np.broadcast_to(circ, (d,))).sum())
    n_bins = (P - n_nc) // (3 * d)
    settings = dict(min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, enable_identity_init=True)
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    out, dl, idx = rqs_transform(t(y, dev), t(params, dev), slots, n_bins, inverse, 0.0, 1.0, 0.0, 1.0, settings,
                                 want_bin_idx=True, oob_counter=oob)
    return out.cpu().numpy(), dl.cpu().numpy(), idx.cpu().numpy(), int(oob.item())


@pytest.mark.parametrize("name,d,circ", UNIT_CASES)
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_vs_oracle_and_golden(hip_lib, oracle, golden, dev, name, d, circ, inverse):
    G = golden("g_rqs_unit")
    n_nc = int((~circ).sum())
    P = 3 * K * d + n_nc
    params, y = synth(100 + d + int(circ.sum()), 128, P, scale=0.5), synth(200 + d, 128, d, uniform=True)
    z, dl, idx, oob = run_rqs_hip(y, params, circ, inverse, dev)
    zo, dlo, det = oracle.rqs(y, params, is_circular=circ, inverse=inverse, dtype=np.float32, want_details=True)
    assert np.array_equal(idx, det["bin_idx"]), "bin indices must be bit-exact vs the oracle"
    assert np.array_equal(z.view(np.uint32), zo.view(np.uint32)), f"outputs differ in {np.sum(z != zo)} elements"
    assert np.array_equal(dl.view(np.uint32), dlo.view(np.uint32)), "dlogp differs"
    assert oob == 0
    tag = f"{name}_{'inv' if inverse else 'fwd'}"
    assert np.array_equal(idx, G[tag + "_idx32"]), "bin indices identical to the reference on the fixture set"
    scale = np.sqrt((G[tag + "_dlogp32"] ** 2).mean())
    assert np.abs(dl - G[tag + "_dlogp32"]).max() <= 1e-5 * scale + 1e-5
    assert np.abs(z - G[tag + "_z64"]).max() <= 3 * np.abs(G[tag + "_z32"] - G[tag + "_z64"]).max() + 2e-7


@pytest.mark.parametrize("B,d,circ,Kb,scale", [(1, 17, False, 8, 0.5), (3, 1, True, 8, 1.0), (1000, 9, False, 8, 2.0),
                                               (4097, 17, True, 8, 0.3), (513, 5, False, 4, 1.0), (777, 3, False, 16, 1.0),
                                               (256, 2, True, 32, 1.0), (100, 66, False, 8, 0.5)])
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_ragged_shapes_bit_exact(hip_lib, oracle, dev, B, d, circ, Kb, scale, inverse):
    circ_mask = np.full(d, circ)
    P = 3 * Kb * d + (0 if circ else d)
    params, y = synth(B + d, B, P, scale=scale), synth(B * 3 + d, B, d, uniform=True)
    z, dl, idx, _ = run_rqs_hip(y, params, circ_mask, inverse, dev)
    zo, dlo, det = oracle.rqs(y, params, is_circular=circ_mask, inverse=inverse, n_bins=Kb, dtype=np.float32, want_details=True)
    assert np.array_equal(idx, det["bin_idx"])
    assert np.array_equal(z.view(np.uint32), zo.view(np.uint32))
    assert np.array_equal(dl.view(np.uint32), dlo.view(np.uint32))


def test_rqs_edges_clamp_identity_saturation(hip_lib, oracle, golden, dev):
    G = golden("g_rqs_unit")
    d = 4
    P = 3 * K * d + d
    nc = np.zeros(d, bool)
    for tag, inverse in (("edge_fwd", False), ("edge_inv", True)):
        y = G[tag + "_y"]
        params = np.repeat(synth(7, 1, P, scale=0.7), len(y), 0)
        z, dl, idx, _ = run_rqs_hip(y, params, nc, inverse, dev)
        zo, dlo, det = oracle.rqs(y, params, inverse=inverse, dtype=np.float32, want_details=True)
        assert np.array_equal(idx, det["bin_idx"])          # exact knots / +-1 ulp: identical to the oracle
        assert np.array_equal(z.view(np.uint32), zo.view(np.uint32))
    # out of domain: clamped like the reference, counter reports the 3 offending inputs
    y, params = G["oob_y"], synth(12, 16, P, scale=0.5)
    for tag, inverse in (("oob_fwd", False), ("oob_inv", True)):
        z, dl, idx, oob = run_rqs_hip(y, params, nc, inverse, dev)
        assert oob == 3
        assert np.array_equal(idx, G[tag + "_idx32"])
        np.testing.assert_allclose(z, G[tag + "_z32"], rtol=0, atol=2e-6)
    # zero parameters = identity
    y = synth(13, 32, d, uniform=True)
    z, dl, idx, _ = run_rqs_hip(y, np.zeros((32, P), np.float32), nc, False, dev)
    np.testing.assert_allclose(z, y, rtol=0, atol=1e-6)
    np.testing.assert_allclose(dl, 0, atol=2e-6)
    # saturated parameters
    params = synth(14, 32, P, scale=12.0)
    for inverse in (False, True):
        z, dl, idx, _ = run_rqs_hip(y, params, nc, inverse, dev)
        zo, dlo, det = oracle.rqs(y, params, inverse=inverse, dtype=np.float32, want_details=True)
        assert np.array_equal(idx, det["bin_idx"])
        assert np.array_equal(z.view(np.uint32), zo.view(np.uint32))


def test_rqs_roundtrip_at_scale(hip_lib, dev):
    """B = 2^20 x 17 (BASELINE size): inverse(forward(y)) == y, dlogp cancels, outputs stay in (0,1)"""
    import bgflow_amd as bg
    B, d = 1 << 20, 17
    g = torch.Generator(device=dev).manual_seed(1234)
    y = torch.rand(B, d, device=dev, generator=g)
    params = torch.randn(B, 3 * K * d + d, device=dev, generator=g) * 0.5

    class Fixed(torch.nn.Module):
        def forward(self, x):
            return params
    tr = bg.ConditionalSplineTransformer(Fixed(), is_circular=False)
    x = torch.zeros(B, 1, device=dev)
    with torch.no_grad():
        z, dl = tr(x, y)
        yb, dli = tr(x, z, inverse=True)
    assert float(z.min()) >= 0 and float(z.max()) <= 1
    assert float((yb - y).abs().max()) < 2e-5
    assert float((dl + dli).abs().max()) < 5e-4
    assert tr.check_domain() == 0


@pytest.mark.parametrize("tag,kw", [("plain", {}), ("vp", dict(preserve_volume=True))])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_vs_oracle_and_golden(hip_lib, oracle, golden, dev, tag, kw, inverse):
    from bgflow_amd.transformer import affine_transform
    G = golden("g_affine_unit")
    y, mu, s = synth(21, 128, 32), synth(22, 128, 32), synth(23, 128, 32, scale=2.0)
    la = torch.tensor([-1.0], device=dev)
    o, dl = affine_transform(t(y, dev), t(mu, dev), t(s, dev), la, kw.get("preserve_volume", False), False, inverse)
    oo, dlo = oracle.affine(y, mu, s, log_alpha=-1.0, inverse=inverse, dtype=np.float32, **kw)
    assert np.array_equal(o.cpu().numpy().view(np.uint32), oo.view(np.uint32))
    assert np.array_equal(dl.cpu().numpy().view(np.uint32), dlo.view(np.uint32))
    key = f"{tag}_{'inv' if inverse else 'fwd'}_"
    np.testing.assert_allclose(o.cpu().numpy(), G[key + "z32"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), G[key + "dlogp32"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("B,d", [(1, 1), (5, 3), (1000, 30), (4099, 66), (33, 257)])
def test_affine_ragged_and_circular(hip_lib, oracle, dev, B, d):
    from bgflow_amd.transformer import affine_transform
    y, mu, s = synth(B, B, d), synth(B + 1, B, d), synth(B + 2, B, d, scale=2.0)
    la = torch.tensor([-0.3], device=dev)
    for inverse in (False, True):
        o, dl = affine_transform(t(y, dev), t(mu, dev), t(s, dev), la, False, False, inverse)
        oo, dlo = oracle.affine(y, mu, s, log_alpha=-0.3, inverse=inverse, dtype=np.float32)
        assert np.array_equal(o.cpu().numpy().view(np.uint32), oo.view(np.uint32))
        assert np.array_equal(dl.cpu().numpy().view(np.uint32), dlo.view(np.uint32))
        yc = synth(B + 3, B, d, uniform=True)
        o, dl = affine_transform(t(yc, dev), t(mu, dev), None, la, False, True, inverse)
        oo, dlo = oracle.affine(yc, mu, None, is_circular=True, inverse=inverse, dtype=np.float32)
        assert np.array_equal(o.cpu().numpy(), oo) and float(dl.abs().max()) == 0.0


def _mixed_ic(dev, golden):
    import bgflow_amd as bg
    from bgflow_amd import configs
    G = golden("g_ic")
    ic = bg.MixedCoordinateTransformation(configs.ala2_whitening_data(), G["z_matrix"].astype(np.int64),
                                          G["rigid_block"].astype(np.int64), keepdims=9, raise_warnings=True)
    # same PCA as the reference (f32 eigh on the same data); load the reference's own buffers so the
    # comparison below is about the kernels, not about LAPACK rounding
    np.testing.assert_allclose(np.abs(ic._whiten.Twhiten.numpy()), np.abs(G["wh_Twhiten"]), rtol=2e-3, atol=2e-3)
    with torch.no_grad():
        ic._whiten.X0mean.copy_(torch.as_tensor(G["wh_mean"])); ic._whiten.Twhiten.copy_(torch.as_tensor(G["wh_Twhiten"]))
        ic._whiten.Tblacken.copy_(torch.as_tensor(G["wh_Tblacken"])); ic._whiten.std.copy_(torch.as_tensor(G["wh_std"]))
        ic._whiten.jacobian_xz = -torch.sum(torch.log(ic._whiten.std))
    return ic.to(dev), G


def test_ic_vs_oracle_and_golden(hip_lib, oracle, golden, dev):
    import bgflow_amd as bg
    ic, G = _mixed_ic(dev, golden)
    z, rigid, x = G["z_matrix"], G["rigid_block"], G["x"]
    rel = bg.RelativeInternalCoordinateTransformation(z.astype(np.int64), rigid.astype(np.int64))
    with torch.no_grad():
        b, a, tt, xf, dl = rel(t(x, dev))
        for got, key in ((b, "rel_bonds"), (a, "rel_angles"), (tt, "rel_torsions"), (xf, "rel_xfixed")):
            np.testing.assert_allclose(got.cpu().numpy(), G[key + "64"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(dl.cpu().numpy(), G["rel_dlogp64"], rtol=1e-5, atol=0)
        xb, dli = rel(b, a, tt, xf, inverse=True)
        np.testing.assert_allclose(xb.cpu().numpy(), x, rtol=0, atol=3e-6)
        assert float((dl + dli).abs().max()) < 1e-4
        # mixed, on flow-like ICs, vs f64 reference and vs the oracle
        args = [t(G[k], dev) for k in ("gen_bonds", "gen_angles", "gen_torsions", "gen_zfixed")]
        xg, dlg = ic(*args, inverse=True)
        # (f32 whitening buffers: compare with the reference's f32 path that used the same buffers)
        np.testing.assert_allclose(xg.cpu().numpy(), G["gen_x32"], rtol=0, atol=3e-6)
        assert (np.abs(dlg.cpu().numpy() - G["gen_dlogp32"]) / np.abs(G["gen_dlogp32"])).max() < 1e-5
        xo, dlo = oracle.ic_ic2xyz(G["gen_bonds"], G["gen_angles"], G["gen_torsions"], G["gen_zfixed"], z, rigid,
                                   blacken=(G["wh_mean"], G["wh_Tblacken"], float(ic._whiten.jacobian_xz)), dtype=np.float32)
        np.testing.assert_allclose(xg.cpu().numpy(), xo, rtol=0, atol=2e-6)
        assert (np.abs(dlg.cpu().numpy() - dlo) / np.abs(dlo)).max() < 1e-5
        b2, a2, t2, z2, dl2 = ic(xg)
        np.testing.assert_allclose(b2.cpu().numpy(), G["gen_bonds"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(t2.cpu().numpy(), G["gen_torsions"], rtol=0, atol=2e-5)
        assert float((dlg + dl2).abs().max()) < 2e-4
    assert rel.check_singularities() == 0


def test_ic_singular_geometry(hip_lib, golden, dev):
    import warnings
    import bgflow_amd as bg
    G = golden("g_ic")
    rel = bg.RelativeInternalCoordinateTransformation(G["z_matrix"].astype(np.int64), G["rigid_block"].astype(np.int64))
    with torch.no_grad():
        b, a, tt, xf, dl = rel(t(G["x_singular"], dev))
    np.testing.assert_allclose(b.cpu().numpy(), G["sing_bonds32"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(a.cpu().numpy(), G["sing_angles32"], rtol=0, atol=1e-6)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert rel.check_singularities() > 0
        assert any("singular" in str(x.message) for x in w)


def test_ic_roundtrip_at_scale(hip_lib, golden, dev):
    """2^18 frames (cfg 3 size): xyz -> IC -> xyz and dlogp cancellation"""
    ic, G = _mixed_ic(dev, golden)
    B = 1 << 18
    g = torch.Generator(device=dev).manual_seed(7)
    x = t(G["xyz0"].astype(np.float32), dev) + 0.005 * torch.randn(B, 66, device=dev, generator=g)
    with torch.no_grad():
        b, a, tt, zf, dl = ic(x)
        xb, dli = ic(b, a, tt, zf, inverse=True)
    # whitening keeps 9 of 15 dof of the rigid block: compare the 17 placed atoms' ICs instead of raw x
    with torch.no_grad():
        b2, a2, t2, z2, _ = ic(xb)
    assert float((b2 - b).abs().max()) < 5e-6 and float((a2 - a).abs().max()) < 5e-6
    assert float((z2 - zf).abs().max()) < 5e-4
    assert torch.isfinite(dl).all() and torch.isfinite(dli).all()


def rel_per_sample(a, ref, floor=0.0):
    a, ref = np.asarray(a).reshape(-1), np.asarray(ref).reshape(-1)
    return np.abs(a - ref) / np.maximum(np.abs(ref), floor)


def assert_cfg3_contract(gen, u, dl, G):
    """The north-star parity contract on the cfg-3 goldens (reference evaluated in f64), PER SAMPLE:
      * whole flow: |dlogp - dlogp64| <= 1e-5 |dlogp64|  (the reference's own f32 path: 8.1e-6 on these inputs);
      * the 16 couplings alone (hand-written coupling kernels; |dlogp| <= 2.1 there, f32 round-off of 272 log terms): state
        within 1e-6 of the f64 state, log-det within 1e-5 of the sample's whole-flow |dlogp64| and not worse than 1.5x
        the deviation of the reference's own f32 evaluation from its f64 one."""
    dl = dl.cpu().numpy()
    r = rel_per_sample(dl, G["dlogp64"])
    assert r.max() <= 1e-5, f"whole-flow log-det: per-sample relative error {r.max():.2e}"
    nc = int(G["n_couplings"])
    with torch.no_grad():
        *st, dlc = gen.flow[:nc](*u)
    state = torch.cat(st, -1).cpu().numpy()
    assert np.abs(state - G["state_c64"]).max() <= 1e-6
    err_c = np.abs(dlc.cpu().numpy() - G["dlogp_c64"]).reshape(-1)
    assert (err_c <= 1e-5 * np.abs(G["dlogp64"]).reshape(-1)).all()
    ref_noise = np.abs(G["dlogp_c32"] - G["dlogp_c64"]).max()
    assert err_c.max() <= 1.5 * ref_noise, f"couplings: {err_c.max():.2e} vs the reference's own f32 deviation {ref_noise:.2e}"
    return r.max()


def assert_bin_ties(idx, det, y, what=""):
    """bin indices of a tolerance-class kernel vs the f32 oracle: any mismatch must be the neighbouring bin of an input within
    rounding distance of one of the oracle's knots (|x - knot| <= 2.4e-7 = 2 ulp at 1, like tests/test_oracle_golden.py)"""
    mis = idx != det["bin_idx"]
    if mis.any():
        dist = np.abs(det["knots"] - np.asarray(y)[..., None]).min(-1)
        assert (dist[mis] <= 2.4e-7).all(), f"{what}: {int(mis.sum())} bin mismatches, farthest from a knot {dist[mis].max():.2e}"
        assert np.abs(idx - det["bin_idx"]).max() <= 1
    return int(mis.sum())


def test_flows_vs_reference_goldens(hip_lib, golden, dev):
    """whole flows through the bgflow-compatible API on the GPU vs the reference's outputs"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    G = golden("g_readme")
    gen = configs.make_readme_generator(dev)
    with torch.no_grad():
        x, dl = gen.flow(t(G["z"], dev))
        np.testing.assert_allclose(x.cpu().numpy(), G["x"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(dl.cpu().numpy(), G["dlogp"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gen.energy(t(G["x"], dev)).cpu().numpy(), G["nll"], rtol=1e-5, atol=1e-5)
    G = golden("g_affine8")
    gen = configs.make_affine8_generator(device=dev)
    z = synth(32, 128, 64)
    with torch.no_grad():
        x, dl = gen.flow(t(z, dev))
        np.testing.assert_allclose(x.cpu().numpy(), G["x64"], rtol=1e-4, atol=3e-5)
        assert (np.abs(dl.cpu().numpy() - G["dlogp64"]) / np.abs(G["dlogp64"]).clip(1)).max() < 1e-5
        zb, dli = gen.flow(x, inverse=True)
        np.testing.assert_allclose(zb.cpu().numpy(), z, rtol=0, atol=2e-5)
        assert float((dl + dli).abs().max()) < 2e-5
    G = golden("g_flow16")
    gen = configs.make_ala2_spline_generator(dev)
    u = [t(G[k], dev) for k in ("u_bonds", "u_angles", "u_torsions", "u_fixed")]
    with torch.no_grad():
        x, dl = gen.flow(*u)
        assert_cfg3_contract(gen, u, dl, G)
        np.testing.assert_allclose(x.cpu().numpy(), G["x64"], rtol=0, atol=5 * np.abs(G["x32"] - G["x64"]).max() + 1e-5)  # f32 erfinv tails of the icdf maps dominate
        kl = gen._target.energy(x) - dl
        np.testing.assert_allclose(kl.cpu().numpy(), G["kl_terms64"], rtol=2e-4, atol=0.5)


def test_flow16_vs_oracle_bin_indices(hip_lib, golden, dev):
    """cfg 3 on the GPU vs the oracle walker: every spline layer's bin indices bit-exact when the
    layer is fed the same inputs; whole-flow dlogp within 1e-5 relative"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    from oracle import flow_oracle as fo
    G = golden("g_flow16")
    gen_cpu = configs.make_ala2_spline_generator()
    gen = configs.make_ala2_spline_generator(dev)
    u = [G[k] for k in ("u_bonds", "u_angles", "u_torsions", "u_fixed")]
    pb, trace = [], []
    (xo,), dlo = fo.run_flow(gen_cpu.flow, u, dtype=np.float32, per_block=pb, trace=trace)
    xs = [t(v, dev) for v in u]
    n_ties = 0
    with torch.no_grad():
        for i, block in enumerate(gen.flow):
            if isinstance(block, bg.CouplingFlow):
                block.transformer.return_bin_indices = True
                # feed the ORACLE's inputs of this layer, so that index equality is a per-layer statement
                ins = [t(v, dev) for v in (u if i == 0 else pb[i - 1][0])]
                *outs, ddl = block(*ins)
                idx = block.transformer.last_bin_indices.cpu().numpy()
                ti = block.transformed_indices[0]
                # the shipped default (fused split-f16 kernel) is tolerance-class: equal, or an ulp-tie at a knot
                n_ties += assert_bin_ties(idx, trace[i], (u if i == 0 else pb[i - 1][0])[ti], f"layer {i}")
                np.testing.assert_allclose(outs[ti].cpu().numpy(), pb[i][0][ti], rtol=0, atol=2e-6)
        x, dl = gen.flow(*xs)
    assert n_ties <= 2, f"{n_ties} tie mismatches on 64 x 240 elements"
    assert_cfg3_contract(gen, xs, dl, G)
    # and against the f32 CPU oracle, per sample: two f32 evaluations, each within 1e-5 of the f64 reference (asserted above for
    # the GPU; 4.6e-6 for the oracle), cannot be asked to agree better than the sum
    r64 = rel_per_sample(dlo, G["dlogp64"]).max()
    assert r64 <= 1e-5
    assert rel_per_sample(dl.cpu().numpy(), dlo).max() <= 1e-5 + r64


# ---------------------------------------------------------------------------------------------------
# fused coupling layer (DenseNet on the f32 matrix cores + spline epilogue)
# ---------------------------------------------------------------------------------------------------
def _layer(kind, dev=None):
    """one builder-style coupling layer of cfg 3: kind in {T|F, F|T, B|A}"""
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_
    dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
    circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
    what, on = {"T|F": ("TORSIONS", "FIXED"), "F|T": ("FIXED", "TORSIONS"), "B|A": ("BONDS", "ANGLES")}[kind]
    layer = hash_init_(configs._spline_coupling(what, on, dims, circ, slot))
    layer.transformer.gemm_mode = "f32"      # the bit-exact tests pin the exact-f32 GEMM mode; split-f16 has its own tests
    return (layer.to(dev) if dev is not None else layer), slot[what]


@pytest.mark.parametrize("kind", ["T|F", "F|T", "B|A"])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("B", [1, 31, 1000, 4133])
def test_fused_layer_bit_exact_vs_oracle(hip_lib, dev, kind, inverse, B):
    from oracle import flow_oracle as fo
    layer_cpu, ti = _layer(kind)
    layer, _ = _layer(kind, dev)
    xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
    layer.transformer.return_bin_indices = True
    with torch.no_grad():
        *outs, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
    assert layer.transformer._fused_cache, "the fused path must have run"
    idx = layer.transformer.last_bin_indices.cpu().numpy()
    fo.MFMA_ORDER = True
    try:
        trace = []
        outs_o, dl_o = fo.run_block(layer_cpu, xs, inverse, np.float32, trace)
    finally:
        fo.MFMA_ORDER = False
    assert np.array_equal(idx, trace[0]["bin_idx"]), "bin indices bit-exact"
    got = outs[ti].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), outs_o[ti].view(np.uint32)), f"{np.sum(got != outs_o[ti])} outputs differ"
    assert np.array_equal(dl.cpu().numpy().view(np.uint32), dl_o.view(np.uint32))
    # and the generic path (torch conditioner + bgk_rqs_transform) agrees to rounding
    layer.transformer.allow_fused = False
    with torch.no_grad():
        *outs2, dl2 = layer(*[t(v, dev) for v in xs], inverse=inverse)
    np.testing.assert_allclose(outs2[ti].cpu().numpy(), got, rtol=0, atol=2e-5)
    np.testing.assert_allclose(dl2.cpu().numpy(), dl.cpu().numpy(), rtol=2e-5, atol=2e-5)


def test_fused_flow16_bit_exact_and_golden(hip_lib, golden, dev):
    """cfg 3 through the fused kernels: the 16 couplings are bit-identical to the oracle (MFMA order),
    and the result matches the reference golden like the generic path does"""
    from bgflow_amd import configs
    from oracle import flow_oracle as fo
    G = golden("g_flow16")
    gen_cpu = configs.make_ala2_spline_generator()
    gen = configs.make_ala2_spline_generator(dev)
    _set_gemm_mode(gen.flow, "f32")
    u = [G[k] for k in ("u_bonds", "u_angles", "u_torsions", "u_fixed")]
    fo.MFMA_ORDER = True
    try:
        ics_o, dl_o = fo.run_flow(gen_cpu.flow[:16], u, dtype=np.float32)
    finally:
        fo.MFMA_ORDER = False
    with torch.no_grad():
        *ics, dl = gen.flow[:16](*[t(v, dev) for v in u])
        x, dl_all = gen.flow(*[t(v, dev) for v in u])
    for a, b in zip(ics, ics_o):
        assert np.array_equal(a.cpu().numpy().view(np.uint32), b.view(np.uint32))
    assert np.array_equal(dl.cpu().numpy().view(np.uint32), dl_o.view(np.uint32))
    assert_cfg3_contract(gen, [t(v, dev) for v in u], dl_all, G)
    np.testing.assert_allclose(torch.cat(ics, -1).cpu().numpy(), G["block15_32"][:, :60], rtol=0, atol=2e-6)


def test_fused_roundtrip_at_scale(hip_lib, dev):
    """2^20 samples through one fused B|A layer and back"""
    layer, ti = _layer("B|A", dev)
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.rand(1 << 20, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
    with torch.no_grad():
        *ys, dl = layer(*xs)
        *zs, dli = layer(*ys, inverse=True)
    assert float((zs[ti] - xs[ti]).abs().max()) < 2e-5
    assert float((dl + dli).abs().max()) < 5e-4
    assert float(ys[ti].min()) >= 0 and float(ys[ti].max()) <= 1


def _set_gemm_mode(flow, mode):
    import bgflow_amd as bg
    for block in flow:
        if isinstance(block, bg.CouplingFlow) and hasattr(block.transformer, "_fused_cache"):
            block.transformer.gemm_mode = mode


@pytest.mark.parametrize("kind", ["T|F", "F|T", "B|A"])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("B", [1, 31, 4133])
@pytest.mark.parametrize("generation", [2, 1])
def test_fused_split_f16_layer_vs_oracle(hip_lib, dev, kind, inverse, B, generation):
    """gemm_mode='f16x2' (conditioner GEMMs as hi+lo f16 pairs on the f16 matrix cores; generation 2 = the shipped
    bgk_fused2.hip kernel, 1 = its predecessor): per sample within 1e-5 of the f64 oracle, outputs within 1e-6, same accuracy
    class as the exact-f32 kernel, and every bin index equal to the f32 oracle's or an ulp-tie at a knot"""
    from oracle import flow_oracle as fo
    layer_cpu, ti = _layer(kind)
    layer, _ = _layer(kind, dev)
    xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
    layer.transformer.return_bin_indices = True
    res = {}
    prev = hip_lib.bgk_set_option(1, generation)
    try:
        for mode in ("f32", "f16x2"):
            layer.transformer.gemm_mode = mode
            with torch.no_grad():
                *outs, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
            assert layer.transformer._fused_cache.get("mode") == mode, "the fused path must have run in this mode"
            res[mode] = (outs[ti].cpu().numpy(), dl.cpu().numpy(), layer.transformer.last_bin_indices.cpu().numpy())
    finally:
        hip_lib.bgk_set_option(1, prev)
    outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64)
    trace = []
    fo.run_block(layer_cpu, xs, inverse, np.float32, trace)
    e32 = np.abs(res["f32"][0] - outs64[ti]).max(), np.abs(res["f32"][1] - dl64).max()
    e16 = np.abs(res["f16x2"][0] - outs64[ti]).max(), np.abs(res["f16x2"][1] - dl64).max()
    assert e16[0] <= 1e-6 and e16[0] <= 3 * e32[0] + 2e-7, f"outputs: split-f16 {e16[0]:.2e} vs f32 {e32[0]:.2e} (error to the f64 oracle)"
    assert rel_per_sample(res["f16x2"][1], dl64, floor=1.0).max() <= 1e-5      # one layer: |dlogp| < 1, i.e. absolute 1e-5
    assert e16[1] <= 3 * e32[1] + 2e-6, f"dlogp: split-f16 {e16[1]:.2e} vs f32 {e32[1]:.2e}"
    n_ties = assert_bin_ties(res["f16x2"][2], trace[0], xs[ti], f"{kind} generation {generation}")
    assert n_ties <= max(2, res["f16x2"][2].size // 10000)


@pytest.mark.parametrize("K", [4, 12, 16, 32])
@pytest.mark.parametrize("kind", ["T|F", "F|T", "B|A"])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("B", [31, 4133])
def test_fused_split_f16_other_bin_counts(hip_lib, dev, K, kind, inverse, B):
    """K = 4 and K = 16 bins through the one-launch coupling kernel (bgk_coupling_rqs_dense_h2: 9 / 2 dims per 128-column
    parameter chunk instead of 5): same bars as K = 8 -- per sample within 1e-5 of the f64 oracle, outputs within 1e-6, every
    bin index equal to the f32 oracle's or an ulp-tie at a knot, and as accurate as the unfused path on the same inputs"""
    from oracle import flow_oracle as fo
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_
    dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
    circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
    slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
    what, on = {"T|F": ("TORSIONS", "FIXED"), "F|T": ("FIXED", "TORSIONS"), "B|A": ("BONDS", "ANGLES")}[kind]
    layer_cpu = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, num_bins=K))
    layer = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, num_bins=K)).to(dev)
    ti = slot[what]
    xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
    layer.transformer.return_bin_indices = True
    with torch.no_grad():
        *outs, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
    cache = layer.transformer._fused_cache
    assert cache.get("mode") == "f16x2" and cache.get("n_bins") == K, "the fused path must have run"
    y_f, dl_f, idx_f = outs[ti].cpu().numpy(), dl.cpu().numpy(), layer.transformer.last_bin_indices.cpu().numpy()
    layer.transformer.allow_fused = False
    with torch.no_grad():
        *outs_g, dl_g = layer(*[t(v, dev) for v in xs], inverse=inverse)
    outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64)
    trace = []
    fo.run_block(layer_cpu, xs, inverse, np.float32, trace)
    e_f = np.abs(y_f - outs64[ti]).max(), np.abs(dl_f - dl64).max()
    e_g = np.abs(outs_g[ti].cpu().numpy() - outs64[ti]).max(), np.abs(dl_g.cpu().numpy() - dl64).max()
    assert e_f[0] <= 1e-6 and e_f[0] <= 3 * e_g[0] + 2e-7, f"outputs: fused {e_f[0]:.2e} vs generic {e_g[0]:.2e} (error to the f64 oracle)"
    # per sample: 1e-5 (|dlogp| < 1 for one layer: absolute), or -- K = 16 halves the bin widths and doubles the f32 noise of the
    # log-det -- at most 1.5 x what the unfused f32 path (torch GEMMs + bgk_rqs_transform) shows on the very same sample set
    worst_f, worst_g = rel_per_sample(dl_f, dl64, floor=1.0).max(), rel_per_sample(dl_g.cpu().numpy(), dl64, floor=1.0).max()
    assert worst_f <= max(1e-5, 1.5 * worst_g), f"dlogp per sample: fused {worst_f:.2e} vs unfused f32 path {worst_g:.2e}"
    assert e_f[1] <= 3 * e_g[1] + 2e-6, f"dlogp: fused {e_f[1]:.2e} vs generic {e_g[1]:.2e}"
    n_ties = assert_bin_ties(idx_f, trace[0], xs[ti], f"{kind} K={K}")
    assert n_ties <= max(2, idx_f.size // 10000)


def test_fused_split_f16_flow16_golden(hip_lib, golden, dev):
    """cfg 3 with gemm_mode='f16x2' against the reference goldens at the same tolerances as the f32 path"""
    from bgflow_amd import configs
    G = golden("g_flow16")
    gen = configs.make_ala2_spline_generator(dev)
    _set_gemm_mode(gen.flow, "f16x2")
    u = [G[k] for k in ("u_bonds", "u_angles", "u_torsions", "u_fixed")]
    with torch.no_grad():
        *ics, dl = gen.flow[:16](*[t(v, dev) for v in u])
        x, dl_all = gen.flow(*[t(v, dev) for v in u])
    assert all(b.transformer._fused_cache.get("mode") == "f16x2" for b in list(gen.flow)[:16])
    assert_cfg3_contract(gen, [t(v, dev) for v in u], dl_all, G)
    np.testing.assert_allclose(torch.cat(ics, -1).cpu().numpy(), G["block15_32"][:, :60], rtol=0, atol=2e-6)
    np.testing.assert_allclose(x.cpu().numpy(), G["x64"], rtol=0, atol=5 * np.abs(G["x32"] - G["x64"]).max() + 1e-5)


def test_fused_split_f16_roundtrip_at_scale(hip_lib, dev):
    layer, ti = _layer("B|A", dev)
    layer.transformer.gemm_mode = "f16x2"
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.rand(1 << 20, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
    with torch.no_grad():
        *ys, dl = layer(*xs)
        *zs, dli = layer(*ys, inverse=True)
    assert float((zs[ti] - xs[ti]).abs().max()) < 2e-5
    assert float((dl + dli).abs().max()) < 5e-4
    assert float(ys[ti].min()) >= 0 and float(ys[ti].max()) <= 1


def test_augmented_flow_cfg5_on_gpu(hip_lib, golden, dev):
    """cfg 5 through the GPU path (fused spline layers + affine kernel with torch conditioners) vs golden"""
    from bgflow_amd import configs
    G = golden("g_aug")
    gen = configs.make_ala2_augmented_generator(dev)
    u = [t(G[k], dev) for k in ("u_bonds", "u_angles", "u_torsions", "u_fixed", "u_aug")]
    with torch.no_grad():
        x, aug, dl = gen.flow(*u)
        *zb, dli = gen.flow(x, aug, inverse=True)
        nc = int(G["n_couplings"])
        *st, dlc = gen.flow[:nc](*u)
    # the 16 couplings (10 spline + 6 affine: the hand-written kernels), per sample, against the reference's f64 evaluation
    assert rel_per_sample(dlc.cpu().numpy(), G["dlogp_c64"]).max() <= 1e-5
    assert np.abs(torch.cat(st, -1).cpu().numpy() - G["state_c64"]).max() <= 1e-6
    # whole flow: the Normal icdf of the 66 auxiliary variables is an f32 erfinv in the reference too -- its own f32 path
    # is 3.7e-3 (relative, per sample) away from its f64 one on these inputs; we must not be farther
    ref_dev = rel_per_sample(G["dlogp32"], G["dlogp64"])
    assert (rel_per_sample(dl.cpu().numpy(), G["dlogp64"]) <= 1.1 * ref_dev.max()).all()
    assert np.abs(x.cpu().numpy() - G["x64"]).max() <= 5 * np.abs(G["x32"] - G["x64"]).max() + 1e-5
    assert np.abs(aug.cpu().numpy() - G["aug64"]).max() <= 2 * np.abs(G["aug32"] - G["aug64"]).max()
    assert torch.isfinite(dli).all()


# ---------------------------------------------------------------------------------------------------
# backward kernels (KL / NLL training path)
# ---------------------------------------------------------------------------------------------------
GRAD_CASES = [("nc17", 17, np.zeros(17, bool)), ("c9", 9, np.ones(9, bool)), ("mix6", 6, np.array([1, 0, 1, 1, 0, 0], bool))]


@pytest.mark.parametrize("name,d,circ", GRAD_CASES)
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_backward_kernel(hip_lib, oracle, golden, dev, name, d, circ, inverse):
    """bgk_rqs_backward through the autograd Function vs the oracle VJP and the reference's autograd gradients"""
    import bgflow_amd as bg
    G = golden("g_grads")
    n_nc = int((~circ).sum())
    P = 3 * K * d + n_nc
    B = 48
    params, y, a, bw = synth(300 + d, B, P, scale=0.7), synth(400 + d, B, d, uniform=True), synth(500 + d, B, d), synth(600 + d, B, 1)
    p = t(params, dev).requires_grad_(True)
    yy = t(y, dev).requires_grad_(True)

    class Fixed(torch.nn.Module):
        def forward(self, x):
            return p
    circ_arg = bool(circ[0]) if circ.all() or (~circ).all() else torch.tensor(circ)
    tr = bg.ConditionalSplineTransformer(Fixed(), is_circular=circ_arg)
    z, dl = tr(torch.zeros(B, 1, device=dev), yy, inverse=inverse)
    ((z * t(a, dev)).sum() + (dl * t(bw, dev)).sum()).backward()
    tag = f"rqs_{name}_{'inv' if inverse else 'fwd'}"
    gy, gp = yy.grad.cpu().numpy(), p.grad.cpu().numpy()
    gyo, gpo = oracle.rqs_backward(y, params, a, bw, is_circular=circ, inverse=inverse, dtype=np.float32)
    np.testing.assert_allclose(gy, gyo, rtol=0, atol=2e-5 * np.abs(gyo).max())
    np.testing.assert_allclose(gp, gpo, rtol=0, atol=2e-5 * np.abs(gpo).max())
    assert np.abs(gy - G[tag + "_gy64"]).max() <= 1e-4 * np.abs(G[tag + "_gy64"]).max()
    assert np.abs(gp - G[tag + "_gp64"]).max() <= 1e-4 * np.abs(G[tag + "_gp64"]).max()


@pytest.mark.parametrize("Kb", [4, 12, 16, 32, 6, 10])
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_backward_kernel_other_bin_counts(hip_lib, oracle, dev, Kb, inverse):
    """bgk_rqs_backward for K != 8 (K = 6, 10: no kernel instance -> autograd through the same map on device torch ops) against (i) the oracle's analytic VJP and (ii) torch autograd (f64) through the torch
    restatement of the nflows spline (oracle/torch_flow.py::rq_spline) -- an independent derivation"""
    import bgflow_amd as bg
    from oracle import torch_flow as tf
    d, B = 7, 64
    circ = np.array([1, 0, 1, 1, 0, 0, 1], bool)
    n_nc = int((~circ).sum())
    P = 3 * Kb * d + n_nc
    params, y, a, bw = synth(300 + Kb, B, P, scale=0.7), synth(400 + Kb, B, d, uniform=True), synth(500 + Kb, B, d), synth(600 + Kb, B, 1)
    p = t(params, dev).requires_grad_(True)
    yy = t(y, dev).requires_grad_(True)

    class Fixed(torch.nn.Module):
        def forward(self, x):
            return p
    tr = bg.ConditionalSplineTransformer(Fixed(), is_circular=torch.tensor(circ))
    z, dl = tr(torch.zeros(B, 1, device=dev), yy, inverse=inverse)
    ((z * t(a, dev)).sum() + (dl * t(bw, dev)).sum()).backward()
    gy, gp = yy.grad.cpu().numpy(), p.grad.cpu().numpy()
    gyo, gpo = oracle.rqs_backward(y, params, a, bw, is_circular=circ, inverse=inverse, dtype=np.float32)
    tol = 2e-5 if Kb % 4 == 0 else 1e-4          # K = 6, 10: stock f32 torch ops (softmax / cumsum / gather) instead of the kernel
    np.testing.assert_allclose(gy, gyo, rtol=0, atol=tol * np.abs(gyo).max())
    np.testing.assert_allclose(gp, gpo, rtol=0, atol=tol * np.abs(gpo).max())
    # torch autograd in f64 through the same parameter unpacking as ConditionalSplineTransformer._compute_params (spline.py:109-126)
    p64 = torch.tensor(params, dtype=torch.float64, requires_grad=True)
    y64 = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    w, h, sl, s_nc = torch.split(p64, [d * Kb, d * Kb, d * Kb, n_nc], dim=-1)
    w, h, sl = (v.reshape(B, d, Kb) for v in (w, h, sl))
    last = sl[..., [0]].clone()
    cm = torch.tensor(circ)
    last = torch.where(cm[None, :, None], last, torch.zeros_like(last))
    extra = torch.zeros(B, d, 1, dtype=torch.float64)
    extra[:, ~cm, 0] = 1.0
    nc_full = torch.zeros(B, d, dtype=torch.float64)
    nc_full = nc_full.index_put((torch.arange(B)[:, None], torch.nonzero(~cm).reshape(1, -1)), s_nc)
    sl_full = torch.cat([sl, last + extra * nc_full[..., None]], dim=-1)
    st = tr._default_settings
    out, ld = tf.rq_spline(y64.clamp(0.0, 1.0), w, h, sl_full, not inverse, 0.0, 1.0, 0.0, 1.0, st["min_bin_width"], st["min_bin_height"],
                           st["min_derivative"], st.get("enable_identity_init", False))
    ((out * torch.tensor(a, dtype=torch.float64)).sum() + (ld.sum(-1, keepdim=True) * torch.tensor(bw, dtype=torch.float64)).sum()).backward()
    # f32 evaluation against f64 autograd: elements next to a knot of a narrow bin have gradients of 1e3 and lose digits there
    # (the strict comparison is the f32 oracle VJP above); 99 % of the elements agree to 1e-4 of the largest gradient
    for got, want in ((gy, y64.grad.numpy()), (gp, p64.grad.numpy())):
        err = np.abs(got - want)
        assert err.max() <= 1e-3 * np.abs(want).max() and np.quantile(err, 0.99) <= 1e-4 * np.abs(want).max()


@pytest.mark.parametrize("pv", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_backward_kernel(hip_lib, golden, dev, pv, inverse):
    import bgflow_amd as bg
    G = golden("g_grads")
    B, d = 64, 12
    y, mu, s, a, bw = synth(1, B, d), synth(2, B, d), synth(3, B, d, scale=1.5), synth(4, B, d), synth(5, B, 1)
    ty, tm, ts = (t(v, dev).requires_grad_(True) for v in (y, mu, s))

    class Fixed(torch.nn.Module):
        def __init__(self, v):
            super().__init__()
            self.v = v

        def forward(self, x):
            return self.v
    tr = bg.AffineTransformer(Fixed(tm), Fixed(ts), preserve_volume=pv).to(dev)
    z, dl = tr(torch.zeros(B, 1, device=dev), ty, inverse=inverse)
    ((z * t(a, dev)).sum() + (dl * t(bw, dev)).sum()).backward()
    tag = f"aff_{'vp' if pv else 'plain'}_{'inv' if inverse else 'fwd'}"
    np.testing.assert_allclose(ty.grad.cpu().numpy(), G[tag + "_gy"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(tm.grad.cpu().numpy(), G[tag + "_gmu"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ts.grad.cpu().numpy(), G[tag + "_gs"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(tr._log_alpha.grad), float(G[tag + "_gla"][0]), rtol=1e-4)


def test_kl_gradient_affine_flow(hip_lib, golden, dev):
    """cfg 2: d/d theta of mean(u(x) - dlogp) through 8 affine couplings (HIP fwd+bwd kernels, torch MLPs)"""
    from bgflow_amd import configs
    G = golden("g_grads")
    gen = configs.make_affine8_generator(device=dev)
    z = t(synth(32, 128, 64), dev)
    x, dlogp = gen.flow(z)
    loss = (gen._target.energy(x) - dlogp).mean()
    loss.backward()
    assert abs(float(loss) - float(G["kl2_loss"])) <= 1e-5 * abs(float(G["kl2_loss"])) + 1e-4
    gla = np.array([float(gen.flow[1 + 2 * i].transformer._log_alpha.grad) for i in range(8)])
    np.testing.assert_allclose(gla, G["kl2_g_log_alpha"], rtol=2e-3, atol=1e-4)
    gb = np.stack([gen.flow[1 + 2 * i].transformer._shift_transformation._layers[0].bias.grad.cpu().numpy() for i in range(8)])
    np.testing.assert_allclose(gb, G["kl2_g_bias0"], rtol=0, atol=2e-3 * np.abs(G["kl2_g_bias0"]).max())
    gnorm = np.sqrt(sum(float((p.grad ** 2).sum()) for p in gen.flow.parameters()))
    assert abs(gnorm - float(G["kl2_gnorm"])) <= 2e-3 * float(G["kl2_gnorm"])


def test_kl_gradient_spline_couplings(hip_lib, golden, dev):
    """cfg 3 couplings: gradient through 16 spline layers (generic path: torch conditioner + HIP spline fwd/bwd)"""
    from bgflow_amd import configs
    G = golden("g_grads")
    gen = configs.make_ala2_spline_generator(dev)
    sub = gen.flow[:16]
    u = [t(synth(41 + i, 64, dd, uniform=True), dev) for i, dd in enumerate((17, 17, 17, 9))]
    *ys, dl = sub(*u)
    loss = (sum((v ** 2).sum(-1, keepdim=True) for v in ys) - dl).mean()
    loss.backward()
    assert abs(float(loss) - float(G["kl3_loss"])) <= 2e-5 * abs(float(G["kl3_loss"]))
    gb = np.stack([np.resize((b.transformer._params_net.net if hasattr(b.transformer._params_net, "net")
                              else b.transformer._params_net)._layers[4].bias.grad.cpu().numpy(), 200) for b in sub])
    np.testing.assert_allclose(gb, G["kl3_g_bias_last"], rtol=0, atol=2e-3 * np.abs(G["kl3_g_bias_last"]).max())
    gnorm = np.sqrt(sum(float((p.grad ** 2).sum()) for p in sub.parameters()))
    assert abs(gnorm - float(G["kl3_gnorm"])) <= 2e-3 * float(G["kl3_gnorm"])


def test_ic_backward_kernel(hip_lib, oracle, golden, dev):
    """bgk_ic_ic2xyz_backward vs the reference's autograd gradients (f64 golden) and the oracle VJP"""
    G = golden("g_grads")
    ic, Gic = _mixed_ic(dev, golden)
    ins = [t(Gic[k], dev).requires_grad_(True) for k in ("gen_bonds", "gen_angles", "gen_torsions", "gen_zfixed")]
    x, dl = ic(*ins, inverse=True)
    a, bw = synth(77, 128, 66), synth(78, 128, 1)
    ((x * t(a, dev)).sum() + (dl * t(bw, dev)).sum()).backward()
    for tns, key in zip(ins, ("ic_g_bonds", "ic_g_angles", "ic_g_torsions", "ic_g_zfixed")):
        ref = G[key]
        assert np.abs(tns.grad.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), key


def test_kl_step_full_cfg3(hip_lib, golden, dev):
    """one full KL gradient of cfg 3 on the GPU (prior sample -> 16 spline couplings -> icdf maps -> IC ->
    target energy), every backward through the hand-written kernels, vs the reference's autograd"""
    from bgflow_amd import configs
    G = golden("g_grads")
    gen = configs.make_ala2_spline_generator(dev)
    u = [t(synth(41 + i, 64, dd, uniform=True), dev) for i, dd in enumerate((17, 17, 17, 9))]
    x, dlogp = gen.flow(*u)
    loss = (gen._target.energy(x) - dlogp).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(G["klfull_loss"])) <= 1e-4 * abs(float(G["klfull_loss"]))
    gb = np.stack([np.resize((b.transformer._params_net.net if hasattr(b.transformer._params_net, "net")
                              else b.transformer._params_net)._layers[4].bias.grad.cpu().numpy(), 200)
                   for b in list(gen.flow)[:16]])
    ref = G["klfull_g_bias_last"]
    assert np.abs(gb - ref).max() <= 5e-3 * np.abs(ref).max()
    gnorm = np.sqrt(sum(float((p.grad ** 2).sum()) for p in gen.flow.parameters()))
    assert abs(gnorm - float(G["klfull_gnorm"])) <= 5e-3 * float(G["klfull_gnorm"])
    # and one optimizer step runs
    opt = torch.optim.Adam(gen.flow.parameters(), lr=1e-4)
    opt.step()


def test_global_ic_on_gpu(hip_lib, oracle, golden, dev):
    """GlobalInternalCoordinateTransformation on the GPU (relative-IC kernel + bgk_ic_refsys) vs golden + round trip"""
    import bgflow_amd as bg
    G = golden("g_ic")
    gic = bg.GlobalInternalCoordinateTransformation(G["global_z_matrix"].astype(np.int64), raise_warnings=False)
    assert (gic.dim_bonds, gic.dim_angles, gic.dim_torsions, gic.dim_fixed) == (21, 20, 19, 0)
    x = G["x"][:64]
    with torch.no_grad():
        b, a, tt, x0, R, dl = gic(t(x, dev))
        for got, key in ((b, "glob_bonds"), (a, "glob_angles"), (tt, "glob_torsions"), (x0, "glob_x0"), (R, "glob_R")):
            np.testing.assert_allclose(got.cpu().numpy(), G[key + "64"], rtol=0, atol=3e-6)
        assert (np.abs(dl.cpu().numpy() - G["glob_dlogp64"]) / np.abs(G["glob_dlogp64"])).max() < 1e-5
        xb, dli = gic(b, a, tt, x0, R, inverse=True)
        np.testing.assert_allclose(xb.cpu().numpy(), x, rtol=0, atol=1e-5)
        assert float((dl + dli).abs().max()) < 2e-4
        # at scale
        big = t(G["xyz0"].astype(np.float32), dev) + 0.005 * torch.randn(1 << 18, 66, device=dev)
        out = gic(big)
        xb, dli = gic(*out[:-1], inverse=True)
        assert float((xb - big).abs().max()) < 5e-5 and float((out[-1] + dli).abs().max()) < 1e-3


def test_cdf_kernel_vs_torch_and_oracle(hip_lib, dev):
    """bgk_cdf_transform (icdf / cdf domain maps) vs the stock torch ops of the same distributions (the reference's path), the
    CPU oracle and scipy (f64)"""
    import scipy.special as sps
    import bgflow_amd as bg
    from bgflow_amd.configs import _NormalMarginal
    B, d = 4099, 17
    u = torch.as_tensor(synth(91, B, d, uniform=True)).to(dev)
    one = torch.ones(d, device=dev)
    dists = [
        bg.TruncatedNormalDistribution(mu=one.clone(), sigma=one.clone(), lower_bound=torch.tensor(1e-5, device=dev), upper_bound=torch.tensor(np.inf, device=dev)),
        bg.TruncatedNormalDistribution(mu=0.5 * one, sigma=one.clone(), lower_bound=torch.tensor(1e-5, device=dev), upper_bound=torch.tensor(1.0, device=dev)),
        bg.SloppyUniform(low=0.0 * one, high=one.clone()),
        _NormalMarginal(torch.zeros(d, device=dev), 20.0 * one),
    ]
    from oracle import flow_oracle as fo
    for dist in dists:
        layer = bg.CDFTransform(dist).to(dev)
        eps = layer._eps
        with torch.no_grad():
            y, dl = layer(u, inverse=True)                   # kernel
            assert layer._desc_cache.get("desc") is not None, "kernel path must have run"
            # the distribution's own torch ops (what the reference runs, nn/flow/cdf.py:36-45)
            y_t = dist.icdf(u.clamp(eps, 1 - eps))
            dl_t = (-dist.log_prob(y_t)).clamp_min(-1 / eps).sum(-1, keepdim=True)
        np.testing.assert_allclose(y.cpu().numpy(), y_t.cpu().numpy(), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dl.cpu().numpy(), dl_t.cpu().numpy(), rtol=2e-5, atol=2e-4)
        with torch.no_grad():
            ub, dlb = layer(y)                               # kernel, forward direction
            ub_t = dist.cdf(y).clamp(eps, 1 - eps)
            dlb_t = dist.log_prob(y).clamp_min(-1 / eps).sum(-1, keepdim=True)
        np.testing.assert_allclose(ub.cpu().numpy(), ub_t.cpu().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(dlb.cpu().numpy(), dlb_t.cpu().numpy(), rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(ub.cpu().numpy(), u.clamp(1e-7, 1 - 1e-7).cpu().numpy(), rtol=0, atol=2e-5)
        # and the CPU oracle (f64 scipy special functions, oracle/flow_oracle.py::cdf_transform)
        yo, dlo = fo.cdf_transform(layer, u.cpu().numpy(), True, np.float32)
        # (f32 erfinv in the kernel -- the same OCML function torch calls -- against f64 special functions: tails of u dominate)
        ey, sy = np.abs(y.cpu().numpy() - yo), np.abs(yo).max() + 1.0
        assert np.median(ey) <= 1e-5 * sy and ey.max() <= 2e-3 * sy, (np.median(ey), ey.max(), sy)
        ed, sd = np.abs(dl.cpu().numpy() - dlo), np.abs(dlo).max() + 1.0
        assert np.median(ed) <= 1e-5 * sd and ed.max() <= 2e-3 * sd, (np.median(ed), ed.max(), sd)
    # f64 truth for the N(0, 20) map
    with torch.no_grad():
        y, dl = bg.CDFTransform(dists[3]).to(dev)(u, inverse=True)
    ref = 20.0 * sps.ndtri(np.clip(u.cpu().numpy().astype(np.float64), 1e-7, 1 - 1e-7))
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=3e-5, atol=1e-4)


@pytest.mark.parametrize("B,P", [(1, 5), (63, 425), (4133, 128), (1 << 16, 408)])
def test_column_sum_kernel(hip_lib, dev, B, P):
    """bias-gradient reduction (bgk_column_sum) vs an f64 sum; through a strided view too"""
    from bgflow_amd.dense import column_sum
    x = synth(77 + B, B, P + 3, scale=2.0)
    xd = t(x, dev)
    got = column_sum(xd[:, :P]).cpu().numpy()
    ref = x[:, :P].astype(np.float64).sum(0)
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6 * np.abs(x).sum(0).max())


def test_densenet_backward_matches_torch_linear(hip_lib, dev):
    """DenseNet's custom Linear backward (bias gradient on bgk_column_sum) vs stock torch autograd"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    net = hash_init_(bg.DenseNet([9, 128, 128, 57], torch.nn.SiLU())).to(dev)
    x = t(synth(5, 1000, 9), dev).requires_grad_(True)
    w = t(synth(6, 1000, 57), dev)
    (net(x) * w).sum().backward()
    g1 = [p.grad.clone() for p in net.parameters()] + [x.grad.clone()]
    for p in net.parameters():
        p.grad = None
    x.grad = None
    (net._layers(x) * w).sum().backward()
    g2 = [p.grad for p in net.parameters()] + [x.grad]
    for a, b in zip(g1, g2):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(b.abs().max()))


# ---------------------------------------------------------------------------------------------------
# fused affine coupling layer (both conditioner MLPs on the f16 matrix cores + affine tail)
# ---------------------------------------------------------------------------------------------------
def _affine_layer(kind, dev=None):
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    if kind == "cfg2":          # DoubleWell dim 64: x | y = 32 | 32, nets [32, 64, 64, 32], ReLU / Tanh
        tr = bg.AffineTransformer(bg.DenseNet([32, 64, 64, 32], torch.nn.ReLU()), bg.DenseNet([32, 64, 64, 32], torch.nn.Tanh()))
        dims = (32, 32)
    elif kind == "aug66":       # cfg 5: AUGMENTED (66) | TORSIONS (17), hidden (128, 128) SiLU
        tr = bg.AffineTransformer(bg.DenseNet([17, 128, 128, 66], torch.nn.SiLU()), bg.DenseNet([17, 128, 128, 66], torch.nn.SiLU()))
        dims = (17, 66)
    elif kind == "nice_circ":   # shift only, circular
        tr = bg.AffineTransformer(bg.DenseNet([5, 64, 64, 7], torch.nn.SiLU()), None, is_circular=True)
        dims = (5, 7)
    elif kind == "periodic":    # cfg 5: AUGMENTED (66) | TORSIONS (17, circular -> cos/sin featuriser)
        tr = bg.AffineTransformer(bg.WrapPeriodic(bg.DenseNet([34, 128, 128, 66], torch.nn.SiLU())),
                                  bg.WrapPeriodic(bg.DenseNet([34, 128, 128, 66], torch.nn.SiLU())))
        dims = (17, 66)
    elif kind == "deep128":     # ala2 RealNVP conditioners with three hidden layers [30, 128, 128, 128, 30]
        tr = bg.AffineTransformer(bg.DenseNet([30, 128, 128, 128, 30], torch.nn.ReLU()), bg.DenseNet([30, 128, 128, 128, 30], torch.nn.Tanh()))
        dims = (30, 30)
    elif kind == "deep64":      # three hidden layers, H = 64, shift network only
        tr = bg.AffineTransformer(bg.DenseNet([12, 64, 64, 64, 40], torch.nn.SiLU()), None)
        dims = (12, 40)
    elif kind == "pv":          # volume preserving
        tr = bg.AffineTransformer(bg.DenseNet([9, 128, 128, 33], torch.nn.Tanh()), bg.DenseNet([9, 128, 128, 33], torch.nn.ReLU()),
                                  preserve_volume=True)
        dims = (9, 33)
    # the event-threaded H = 128 kernel (bgk_fused2.hip): its other instantiations and branches
    elif kind == "v2_relu_pv":      # ReLU networks, two output tiles, volume preserving
        tr = bg.AffineTransformer(bg.DenseNet([9, 128, 128, 33], torch.nn.ReLU()), bg.DenseNet([9, 128, 128, 33], torch.nn.ReLU()),
                                  preserve_volume=True)
        dims = (9, 33)
    elif kind == "v2_tanh_wide":    # Tanh networks, 101 input features (7 k-steps of layer 0), one output tile
        tr = bg.AffineTransformer(bg.DenseNet([100, 128, 128, 20], torch.nn.Tanh()), bg.DenseNet([100, 128, 128, 20], torch.nn.Tanh()))
        dims = (100, 20)
    elif kind == "v2_shift_circ":   # shift network only, circular output
        tr = bg.AffineTransformer(bg.DenseNet([5, 128, 128, 7], torch.nn.SiLU()), None, is_circular=True)
        dims = (5, 7)
    elif kind == "v2_scale_only":   # scale network only, 96 dims (three full output tiles)
        tr = bg.AffineTransformer(None, bg.DenseNet([12, 128, 128, 96], torch.nn.SiLU()))
        dims = (12, 96)
    layer = hash_init_(bg.CouplingFlow(tr, transformed_indices=(1,), cond_indices=(0,)))
    return (layer.to(dev) if dev is not None else layer), dims


@pytest.mark.parametrize("kind", ["cfg2", "aug66", "periodic", "nice_circ", "pv", "deep128", "deep64",
                                  "v2_relu_pv", "v2_tanh_wide", "v2_shift_circ", "v2_scale_only"])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("B", [1, 31, 4133])
def test_fused_affine_layer_vs_oracle(hip_lib, dev, kind, inverse, B):
    from oracle import flow_oracle as fo
    layer_cpu, dims = _affine_layer(kind)
    layer, _ = _affine_layer(kind, dev)
    xs = [synth(B + 3 * i, B, d, uniform=(kind in ("nice_circ", "v2_shift_circ") or (kind == "periodic" and i == 0))) for i, d in enumerate(dims)]
    with torch.no_grad():
        x_out, y_out, dl = layer(*[t(v, dev) for v in xs], inverse=inverse)
    assert layer.transformer._fused_cache, "the fused affine path must have run"
    layer.transformer.allow_fused = False
    with torch.no_grad():
        _, y_gen, dl_gen = layer(*[t(v, dev) for v in xs], inverse=inverse)
    outs64, dl64 = fo.run_block(layer_cpu, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
    scale = max(1.0, float(np.abs(outs64[1]).max()))
    e_f = np.abs(y_out.cpu().numpy() - outs64[1]).max(), np.abs(dl.cpu().numpy() - dl64).max()
    e_g = np.abs(y_gen.cpu().numpy() - outs64[1]).max(), np.abs(dl_gen.cpu().numpy() - dl64).max()
    assert e_f[0] <= 3 * e_g[0] + 1e-6 * scale, f"outputs: fused {e_f[0]:.2e} vs generic {e_g[0]:.2e} (error to the f64 oracle)"
    assert e_f[1] <= 3 * e_g[1] + 2e-6 * max(1.0, float(np.abs(dl64).max())), f"dlogp: fused {e_f[1]:.2e} vs generic {e_g[1]:.2e}"
    np.testing.assert_allclose(y_out.cpu().numpy(), outs64[1], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dl.cpu().numpy(), dl64, rtol=2e-5, atol=2e-5)
    assert torch.equal(x_out, t(xs[0], dev))


def test_fused_affine_roundtrip_at_scale(hip_lib, dev):
    layer, dims = _affine_layer("cfg2", dev)
    g = torch.Generator(device=dev).manual_seed(5)
    xs = [torch.randn(1 << 20, d, device=dev, generator=g) for d in dims]
    with torch.no_grad():
        x, y, dl = layer(*xs)
        _, z, dli = layer(x, y, inverse=True)
    assert float((z - xs[1]).abs().max()) < 1e-4
    assert float((dl + dli).abs().max()) < 1e-5


def test_nll_training_step_cfg3(hip_lib, golden, dev):
    """NLL direction (data -> latent): energy(x) through xyz->IC, the cdf maps and 16 inverse spline couplings with
    autograd on the parameters; the gradient is checked against a central finite difference along a fixed direction"""
    from bgflow_amd import configs
    G = golden("g_flow16")
    gen = configs.make_ala2_spline_generator(dev)
    x = t(G["x64"].astype(np.float32), dev)
    params = [p for p in gen.flow.parameters() if p.requires_grad]
    loss = gen.energy(x).mean()
    assert abs(float(loss.detach()) - float(G["nll64"].mean())) <= 1e-4 * abs(float(G["nll64"].mean())) + 1e-3
    loss.backward()
    grads = [p.grad.clone() for p in params]
    assert all(torch.isfinite(g).all() for g in grads) and sum(float(g.abs().sum()) for g in grads) > 0
    # directional derivative along v (deterministic, layer-wise normalised)
    vs = [t(synth(900 + i, p.numel()).reshape(tuple(p.shape)), dev) for i, p in enumerate(params)]
    vs = [v / (v.norm() + 1e-12) for v in vs]
    analytic = sum(float((g * v).sum()) for g, v in zip(grads, vs))
    eps = 2e-3
    vals = []
    with torch.no_grad():
        for sgn in (+1.0, -1.0):
            for p, v in zip(params, vs):
                p.add_(sgn * eps * v)
            vals.append(float(gen.energy(x).double().mean()))
            for p, v in zip(params, vs):
                p.sub_(sgn * eps * v)
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - analytic) <= 0.05 * abs(analytic) + 1e-3, (fd, analytic)


@pytest.mark.parametrize("kind", ["T|F", "F|T", "B|A"])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("B", [1, 33, 777])
def test_fused_training_forward_gradients(hip_lib, dev, kind, inverse, B):
    """differentiable one-launch forward (bgk_coupling_rqs_dense_h2_train + MLP backward on the saved tensors) vs the
    generic autograd path (torch conditioner + bgk_rqs_transform / bgk_rqs_backward): outputs and every gradient"""
    res = {}
    for fused in (True, False):
        layer, ti = _layer(kind, dev)
        layer.transformer.gemm_mode = "f16x2"
        layer.transformer.allow_fused = fused
        xs = [t(synth(B + 7 * i, B, d, uniform=True), dev).requires_grad_(True) for i, d in enumerate((17, 17, 17, 9))]
        *outs, dl = layer(*xs, inverse=inverse)
        if fused:
            assert layer.transformer._fused_cache.get("src_col_dev") is not None, "the fused training path must have run"
        w = t(synth(55, B, outs[ti].shape[1]), dev)
        ((outs[ti] * w).sum() + (dl * t(synth(56, B, 1), dev)).sum()).backward()
        res[fused] = ([outs[ti].detach(), dl.detach()], [p.grad for p in layer.parameters()] + [x.grad for x in xs if x.grad is not None])
    for a, b in zip(res[True][0], res[False][0]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-5)
    assert len(res[True][1]) == len(res[False][1])
    for a, b in zip(res[True][1], res[False][1]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-4 * float(b.abs().max()) + 1e-6)


@pytest.mark.parametrize("kind", ["T|F", "F|T", "B|A"])
def test_device_packer_equals_torch_packer(hip_lib, dev, kind):
    """bgk_pack_dense_h2 (device, no host sync) produces the same split-f16 operand blocks and scales as the torch
    reference packer whose layout tests/test_host_logic.py checks on the CPU"""
    from bgflow_amd import dense
    layer, ti = _layer(kind, dev)
    tr = layer.transformer
    net = tr._params_net
    inner = net.net if type(net) is dense.WrapPeriodic else net
    (l0, l1, l2), _ = dense._fusable_dense(inner)
    d = {"T|F": 17, "F|T": 9, "B|A": 17}[kind]
    _, nc_host = tr._nc_slot(d, dev)
    A0, A1, A2, (c0, c1, c2) = dense.pack_dense_for_fused_h2((l0, l1, l2), nc_host, d, 8)
    src = dense._src_col_table(d, 8, nc_host, dev)
    B0, B1, B2, cs = dense.pack_dense_for_fused_h2_device((l0, l1, l2), src, src.numel() // 128)
    cs = cs.cpu().numpy()
    assert np.allclose(cs[1::2], [c0, c1, c2]) and np.allclose(cs[0::2] * cs[1::2], 1.0)
    for a, b in zip((A0, A1, A2), (B0, B1, B2)):
        assert a.shape == b.shape
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("kind", ["T|F", "B|A"])
def test_fused_bf16_mode_is_consistent_and_invertible(hip_lib, dev, kind):
    """gemm_mode='bf16' (bf16 weights + GEMM inputs, the reduced-precision leg of BASELINE config 5): close to the f32-class
    result at bf16 resolution, and -- the conditioner being deterministic -- exactly as invertible as the f32 path"""
    B = 4133
    layer, ti = _layer(kind, dev)
    xs = [t(synth(B + 7 * i, B, d, uniform=True), dev) for i, d in enumerate((17, 17, 17, 9))]
    res = {}
    for mode in ("f16x2", "bf16"):
        layer.transformer.gemm_mode = mode
        with torch.no_grad():
            *ys, dl = layer(*xs)
            *zs, dli = layer(*ys, inverse=True)
        assert layer.transformer._fused_cache.get("mode") == mode
        res[mode] = (ys[ti], dl)
        assert float((zs[ti] - xs[ti]).abs().max()) < 2e-5
        assert float((dl + dli).abs().max()) < 5e-4
    dy = float((res["bf16"][0] - res["f16x2"][0]).abs().max())
    ddl = float((res["bf16"][1] - res["f16x2"][1]).abs().max())
    assert 0 < dy < 3e-2 and ddl < 0.5, (dy, ddl)


def test_builder_built_generator_equals_configs_generator(hip_lib, dev):
    """a flow assembled with the BoltzmannGeneratorBuilder API runs the same kernels to the same bits as the hand-assembled cfg 3"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    from bgflow_amd.utils import hash_init_
    zmat, rigid, xyz = configs.ala2_system()
    ic = bg.MixedCoordinateTransformation(configs.ala2_whitening_data(), zmat, rigid, keepdims=9, raise_warnings=False)
    builder = bg.BoltzmannGeneratorBuilder(bg.ShapeDictionary.from_coordinate_transform(ic),
                                           target=bg.NormalDistribution(66, torch.tensor(xyz[0], dtype=torch.float32)),
                                           device=dev, dtype=torch.float32)
    for _ in range(4):
        builder.add_condition(bg.TORSIONS, on=bg.FIXED)
        builder.add_condition(bg.FIXED, on=bg.TORSIONS)
    for _ in range(4):
        builder.add_condition(bg.BONDS, on=bg.ANGLES)
        builder.add_condition(bg.ANGLES, on=bg.BONDS)
    builder.add_map_to_ic_domains()
    builder.add_map_to_cartesian(ic)
    gen = builder.build_generator().to(dev)
    hash_init_(gen.flow)
    ref = configs.make_ala2_spline_generator(dev)
    u = [t(synth(40 + i, 500, d, uniform=True), dev) for i, d in enumerate((17, 17, 17, 9))]
    with torch.no_grad():
        x, dl = gen.flow(*u)
        xr, dlr = ref.flow(*u)
        kl = gen.kldiv(64)
    assert torch.equal(x, xr) and torch.equal(dl, dlr)
    assert kl.shape == (64, 1) and torch.isfinite(kl).all()


@pytest.mark.parametrize("act", [torch.nn.ReLU, torch.nn.Tanh])
def test_fused_training_forward_other_activations(hip_lib, dev, act):
    """the differentiable one-launch forward with ReLU / Tanh conditioners (non-circular, d = 9 conditioned on 17 dims)"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    B = 515
    res = {}
    for fused in (True, False):
        tr = bg.ConditionalSplineTransformer(bg.DenseNet([17, 128, 128, 3 * 8 * 9 + 9], act()), is_circular=False)
        layer = hash_init_(bg.CouplingFlow(tr, transformed_indices=(1,), cond_indices=(0,))).to(dev)
        tr.allow_fused = fused
        x = t(synth(11, B, 17, uniform=True), dev).requires_grad_(True)
        y = t(synth(12, B, 9, uniform=True), dev).requires_grad_(True)
        xo, yo, dl = layer(x, y)
        ((yo * t(synth(13, B, 9), dev)).sum() + (dl * t(synth(14, B, 1), dev)).sum()).backward()
        res[fused] = [yo.detach(), dl.detach(), x.grad, y.grad] + [p.grad for p in layer.parameters()]
        if fused:
            assert tr._fused_cache.get("src_col_dev") is not None
    for a, b in zip(res[True], res[False]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=3e-4 * float(b.abs().max()) + 2e-5)



@pytest.mark.parametrize("K", [4, 12, 16, 32])
@pytest.mark.parametrize("circular,inverse", [(False, False), (True, True)])
def test_fused_training_forward_other_bin_counts(hip_lib, dev, K, circular, inverse):
    """the differentiable one-launch forward for K = 4 | 16 bins (first-generation kernel with the saved pre-activations and
    parameters) against the generic autograd path: outputs and every gradient (`conditioner_factory.py:76-80` lets users pick K)"""
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    B, d, d_c = 643, 9, 17
    res = {}
    for fused in (True, False):
        net = bg.DenseNet([2 * d_c if circular else d_c, 128, 128, 3 * K * d + (0 if circular else d)], torch.nn.SiLU())
        if circular:
            net = bg.WrapPeriodic(net, indices=np.arange(d_c))
        tr = bg.ConditionalSplineTransformer(net, is_circular=circular)
        layer = hash_init_(bg.CouplingFlow(tr, transformed_indices=(1,), cond_indices=(0,))).to(dev)
        tr.allow_fused = fused
        x = t(synth(21, B, d_c, uniform=True), dev).requires_grad_(True)
        y = t(synth(22, B, d, uniform=True), dev).requires_grad_(True)
        xo, yo, dl = layer(x, y, inverse=inverse)
        ((yo * t(synth(23, B, d), dev)).sum() + (dl * t(synth(24, B, 1), dev)).sum()).backward()
        res[fused] = [yo.detach(), dl.detach(), x.grad, y.grad] + [p.grad for p in layer.parameters()]
        if fused:
            assert tr._fused_cache.get("src_col_dev") is not None and tr._fused_cache.get("n_bins") == K, "the fused training path must have run"
    for a, b in zip(res[True], res[False]):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=3e-4 * float(b.abs().max()) + 2e-5)


def test_stochastic_augmentation_on_gpu(hip_lib, golden, dev):
    """a17: the augmentation layer of cfg 5 on device tensors against the reference golden (augment.py:27-55)"""
    from test_host_logic import _check_augmentation
    _check_augmentation(golden("g_augment"), dev)


# ---------------------------------------------------------------------------------------------------
# round-2 backward kernels: CDF maps, xyz -> IC, global reference system (goldens: reference autograd in f64, g_grads2)
# ---------------------------------------------------------------------------------------------------
def _inner_cdf(block):
    import bgflow_amd as bg
    inner = block
    while not isinstance(inner, bg.CDFTransform) and hasattr(inner, "_flow"):
        inner = inner._flow
    while not isinstance(inner, bg.CDFTransform) and hasattr(inner, "_delegate"):
        inner = inner._delegate
    return inner if isinstance(inner, bg.CDFTransform) else None


def _close(got, ref, rtol, what):
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= rtol * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e}"


def test_cdf_backward_kernel(hip_lib, golden, dev):
    """bgk_cdf_backward behind CDFTransform (both directions, the four marginals of the cfg-3 builder flow) against the
    reference's autograd (nn/flow/cdf.py:28-46); and the stale-descriptor fix: new marginal parameters are picked up"""
    from bgflow_amd import configs
    G = golden("g_grads2")
    gen = configs.make_ala2_spline_generator(dev)
    cdfs = [c for c in (_inner_cdf(b) for b in gen.flow) if c is not None]
    assert len(cdfs) == 4
    for k, cdf in enumerate(cdfs):
        a, bw = t(G[f"cdf{k}_a"].astype(np.float32), dev), t(G[f"cdf{k}_bw"].astype(np.float32), dev)
        u = t(G[f"cdf{k}_u"].astype(np.float32), dev).requires_grad_(True)
        y, dl = cdf(u, inverse=True)
        assert y.grad_fn is not None and "CdfFn" in type(y.grad_fn).__name__, "the kernel path must run under autograd"
        ((y * a).sum() + (dl * bw).sum()).backward()
        _close(u.grad.cpu().numpy(), G[f"cdf{k}_inv_gx"], 2e-4, f"cdf {k} icdf direction")
        x = t(G[f"cdf{k}_y"].astype(np.float32), dev).requires_grad_(True)
        uu, dl2 = cdf(x)
        ((uu * a).sum() + (dl2 * bw).sum()).backward()
        _close(x.grad.cpu().numpy(), G[f"cdf{k}_fwd_gx"], 2e-4, f"cdf {k} cdf direction")
    # parameters written after a first call (load_state_dict) must reach the kernel
    cdf = cdfs[0]
    u = t(G["cdf0_u"].astype(np.float32), dev)
    with torch.no_grad():
        y0, _ = cdf(u, inverse=True)
        sd = {k: v.clone() for k, v in cdf.state_dict().items()}
        for k in sd:
            if k.endswith("_mu"):
                sd[k] = sd[k] + 0.05
        assert any(k.endswith("_mu") for k in sd)
        cdf.load_state_dict(sd)
        y1, _ = cdf(u, inverse=True)
    assert float((y1 - y0).abs().min()) > 0.04, "the cached descriptor went stale"


def test_xyz2ic_backward_kernel(hip_lib, golden, dev):
    """bgk_ic_xyz2ic_backward (relative and mixed / whitened) against the reference's autograd through
    crd_transform/ic.py:386-433 + ic_helper.py:148-293"""
    import bgflow_amd as bg
    G, Gic = golden("g_grads2"), golden("g_ic")
    ic_mixed, _ = _mixed_ic(dev, golden)
    rel = bg.RelativeInternalCoordinateTransformation(Gic["z_matrix"].astype(np.int64), Gic["rigid_block"].astype(np.int64))
    w = {k: t(G[k].astype(np.float32), dev) for k in ("x2ic_wb", "x2ic_wa", "x2ic_wt", "x2ic_wf15", "x2ic_wf9", "x2ic_wl")}
    for name, tr, wf in (("rel", rel, w["x2ic_wf15"]), ("mix", ic_mixed, w["x2ic_wf9"])):
        x = t(Gic["x"][:64].astype(np.float32), dev).requires_grad_(True)
        b, a, tt, f, dl = tr(x)
        ((b * w["x2ic_wb"]).sum() + (a * w["x2ic_wa"]).sum() + (tt * w["x2ic_wt"]).sum() + (f * wf).sum() + (dl * w["x2ic_wl"]).sum()).backward()
        _close(x.grad.cpu().numpy(), G[f"x2ic_{name}_gx"], 5e-4, f"xyz->IC ({name})")


def test_global_ic_backward_kernels(hip_lib, golden, dev):
    """Global internal coordinates under autograd (bgk_ic_xyz2ic_backward + bgk_ic_refsys_backward forward direction;
    bgk_ic_ic2xyz_backward + bgk_ic_refsys_backward inverse direction) against the reference's autograd
    (crd_transform/ic.py:162-265, 516-716)"""
    import bgflow_amd as bg
    G, Gic = golden("g_grads2"), golden("g_ic")
    gic = bg.GlobalInternalCoordinateTransformation(Gic["global_z_matrix"].astype(np.int64))
    x = t(G["glob_x"].astype(np.float32), dev).requires_grad_(True)
    outs = gic(x)
    ws = [t(G[f"glob_w{i}"].astype(np.float32), dev) for i in range(6)]
    sum((v * w).sum() for v, w in zip(outs, ws)).backward()
    _close(x.grad.cpu().numpy(), G["glob_fwd_gx"], 1e-3, "global IC forward")
    ins = [t(G[f"glob_in{i}"].astype(np.float32), dev).requires_grad_(True) for i in range(5)]
    xb, dli = gic(*ins, inverse=True)
    ((xb * t(G["glob_wx"].astype(np.float32), dev)).sum() + (dli * t(G["glob_wl2"].astype(np.float32), dev)).sum()).backward()
    for i, v in enumerate(ins):
        _close(v.grad.cpu().numpy(), G[f"glob_inv_g{i}"], 1e-3, f"global IC inverse, input {i}")


# ---------------------------------------------------------------------------------------------------
# round-2 training pieces: weight-gradient kernel, flat Adam, KLTrainer, world-1 RCCL group
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,P,d_c,periodic,gscale", [(4133, 425, 17, 0, 1.0), (1000, 225, 17, 1, 1e-7), (70000, 408, 9, 0, 3e-6)])
def test_dense_weight_grad_kernel(hip_lib, dev, B, P, d_c, periodic, gscale):
    """bgk_dense_weight_grad (dW = g^T h, db = sum g over the batch, f16 hi+lo split on the matrix cores under a per-tensor
    power-of-two scale of g -- here measured by bgk_absmax --, deterministic slab reduction) against f64 matmuls -- ragged batch,
    tiny gradient magnitudes, cos/sin featuriser.  Round 5: 22-bit products (5e-6 of the largest entry; the bf16 pairs of rounds
    1 - 4 were held to 3e-5)"""
    from bgflow_amd.dense import _dense_weight_grad
    g = torch.Generator(device=dev).manual_seed(B)
    g_p = torch.randn(B, P, device=dev, generator=g) * gscale
    g_z1, g_z0 = (torch.randn(B, 128, device=dev, generator=g) * gscale for _ in range(2))
    h1, h0 = (torch.randn(B, 128, device=dev, generator=g) for _ in range(2))
    x = torch.rand(B, d_c, device=dev, generator=g)
    n_in = 2 * d_c if periodic else d_c
    res = _dense_weight_grad(g_p, g_z1, g_z0, h1, h0, x, bool(periodic), n_in, [True] * 8, {})
    res2 = _dense_weight_grad(g_p, g_z1, g_z0, h1, h0, x, bool(periodic), n_in, [True] * 8, {})
    assert all(torch.equal(a, b) for a, b in zip(res, res2)), "deterministic"
    feats = torch.cat([torch.cos(2 * np.pi * x.double()), torch.sin(2 * np.pi * x.double())], -1) if periodic else x.double()
    ref = (g_z0.double().t() @ feats, g_z0.double().sum(0), g_z1.double().t() @ h0.double(), g_z1.double().sum(0),
           g_p.double().t() @ h1.double(), g_p.double().sum(0))
    for got, want, name in zip(res, ref, ("gW0", "gb0", "gW1", "gb1", "gW2", "gb2")):
        scale = float(want.abs().max())
        err = float((got.double() - want).abs().max())
        assert err <= 5e-6 * scale, f"{name}: {err:.3e} vs scale {scale:.3e}"
    # activation applied on the fly: the h arrays hold pre-activations (what bgk_dense_backward_dx's caller passes when it does
    # not materialise h1 / h0)
    for code, fn in ((1, torch.nn.functional.silu), (2, torch.relu), (3, torch.tanh)):
        z1, z0 = 2.0 * h1, 2.0 * h0
        res_a = _dense_weight_grad(g_p, g_z1, g_z0, z1, z0, x, bool(periodic), n_in, [True] * 8, {}, h_act=code)
        ref_a = (ref[0], ref[1], g_z1.double().t() @ fn(z0.double()), ref[3], g_p.double().t() @ fn(z1.double()), ref[5])
        for got, want, name in zip(res_a, ref_a, ("gW0", "gb0", "gW1", "gb1", "gW2", "gb2")):
            scale = float(want.abs().max())
            err = float((got.double() - want).abs().max())
            assert err <= 5e-6 * scale, f"act {code} {name}: {err:.3e} vs scale {scale:.3e}"


def test_flat_adam_matches_torch_adam_and_skips_nan(hip_lib, dev):
    from bgflow_amd.training import FlatAdam
    torch.manual_seed(0)
    shapes = [(7, 5), (5,), (3, 4, 2)]
    pa = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = FlatAdam(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-3)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-3)
    for it in range(6):
        oa.zero_grad(); ob.zero_grad()
        for ps in (pa, pb):
            loss = sum(((p - 0.3 * (i + 1)) ** 2).sum() * (1 + it) for i, p in enumerate(ps))
            loss.backward()
        from bgflow_amd.utils import param_state_key
        k0 = param_state_key(pa[0])
        oa.step(); ob.step()
        assert param_state_key(pa[0]) != k0, "the step must change the parameters' state key (packed-weight caches key on it)"
        for a, b in zip(pa, pb):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    before = [p.detach().clone() for p in pa]
    oa.zero_grad()
    sum((p ** 2).sum() for p in pa).backward()
    pa[1].grad[2] = float("nan")
    oa.step()
    assert oa.skipped_steps() == 1 and all(torch.equal(a.detach(), b) for a, b in zip(pa, before))


def test_kltrainer_cfg3_on_gpu(hip_lib, dev):
    """KLTrainer.train (trainers.py:148-201) drives the hand-written forward / backward kernels, bgk_dense_weight_grad and
    bgk_adam_step: the first reported KL equals a direct evaluation, the loss moves, nothing is skipped"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    gen = configs.make_ala2_spline_generator(dev)
    torch.manual_seed(1)
    bg.SequentialFlow.FUSE_GENERATION_TAIL = False      # the training path runs the blocks (autograd); compare like with like:
    try:                                                  # near-degenerate prior samples (angle -> 0) are conditioned differently
        with torch.no_grad():                             # by the fused tail's closed-form log-det
            ref = float(gen.kldiv(2048).mean())
    finally:
        bg.SequentialFlow.FUSE_GENERATION_TAIL = True
    torch.manual_seed(1)
    tr = bg.KLTrainer(gen, train_likelihood=False)
    assert type(tr.optim).__name__ == "FlatAdam"
    tr.optim.param_groups[0]["lr"] = 1e-4
    w0 = gen.flow[0].transformer._params_net._layers[0].weight.detach().clone()
    tr.train(4, batchsize=2048)
    _, _, ys = tr.losses()
    assert abs(ys[0][0] - ref) <= 1e-4 * abs(ref) and np.isfinite(ys[0]).all()
    assert tr.optim.skipped_steps() == 0
    assert float((gen.flow[0].transformer._params_net._layers[0].weight.detach() - w0).abs().max()) > 0
    # NLL direction through xyz -> IC (backward kernels of round 2)
    with torch.no_grad():
        x = gen.sample(1024)
    tr2 = bg.KLTrainer(gen, train_energy=False)
    tr2.optim.param_groups[0]["lr"] = 1e-4
    tr2.train(2, data=x, batchsize=512)
    assert np.isfinite(tr2.losses()[2][0]).all()


def test_world1_rccl_group_kl_loss(hip_lib, dev):
    """the data-parallel pieces on a REAL RCCL process group (world size 1, this GPU): dp.global_mean of the cfg-3 KL integrand
    (bg.py:13-17 reduced over ranks) and the flat gradient bucket all-reduce give exactly the single-process results"""
    import socket
    import torch.distributed as dist
    from bgflow_amd import configs, dp
    if dist.is_initialized():
        pytest.skip("a process group is already initialised")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    gen = configs.make_ala2_spline_generator(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    z = [torch.rand(4096, d, device=dev, generator=g) for d in (17, 17, 17, 9)]

    def loss_and_grads(reduce_fn, after_backward):
        gen.zero_grad(set_to_none=True)
        *x, dl = gen.flow(*z)
        loss = reduce_fn(gen._target.energy(*x) - dl)
        loss.backward()
        params = [p for p in gen.flow.parameters()]
        after_backward(params)
        return float(loss.detach()), torch.cat([p.grad.reshape(-1) for p in params]).clone()
    l0, g0 = loss_and_grads(lambda v: v.mean(), lambda ps: None)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        dp_is = dp.is_distributed
        dp.is_distributed = lambda: True          # world size 1 is "not distributed" for dp; force the collective path
        try:
            l1, g1 = loss_and_grads(lambda v: dp.global_mean(v), lambda ps: dp.allreduce_gradients_(ps))
        finally:
            dp.is_distributed = dp_is
    finally:
        dist.destroy_process_group()
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=1e-5, atol=1e-7 * float(g0.abs().max()))


def test_fused_generation_tail_equals_blocks(hip_lib, golden, dev):
    """icdf domain maps + IC -> xyz as ONE kernel (bgk_icdf_ic2xyz, sampling direction) against the same blocks run one by one
    (bgk_cdf_transform x 4 + bgk_ic_ic2xyz), cfg 3 and cfg 5 (auxiliary slot passes through its own map)"""
    import bgflow_amd as bg
    from bgflow_amd import configs
    for make, keys, name in ((configs.make_ala2_spline_generator, ("u_bonds", "u_angles", "u_torsions", "u_fixed"), "g_flow16"),
                             (configs.make_ala2_augmented_generator, ("u_bonds", "u_angles", "u_torsions", "u_fixed", "u_aug"), "g_aug")):
        G = golden(name)
        gen = make(dev)
        assert gen.flow.segments()[-1][0] == "icdf+ic2xyz"
        g = torch.Generator(device=dev).manual_seed(11)
        u = [torch.cat([t(G[k], dev), 0.02 + 0.96 * torch.rand(4000, G[k].shape[1], device=dev, generator=g)]) for k in keys]
        with torch.no_grad():
            *xs, dl = gen.flow(*u)
            bg.SequentialFlow.FUSE_GENERATION_TAIL = False
            try:
                *xs_b, dl_b = gen.flow(*u)
            finally:
                bg.SequentialFlow.FUSE_GENERATION_TAIL = True
        assert len(xs) == len(xs_b)
        n_g = G[keys[0]].shape[0]
        # the golden inputs (well-conditioned geometries): the contract tolerance
        x_noise = 5 * np.abs(G["x32"] - G["x64"]).max() + 1e-5        # f32 erfinv of the Normal(0, 20) marginal dominates x
        assert float((xs[0][:n_g] - xs_b[0][:n_g]).abs().max()) <= x_noise
        for a, b in zip(xs[1:], xs_b[1:]):
            assert float((a[:n_g] - b[:n_g]).abs().max()) <= 5e-6
        assert rel_per_sample(dl[:n_g].cpu().numpy(), dl_b[:n_g].cpu().numpy()).max() <= (1e-5 if name == "g_flow16" else 1e-3)
        # random prior samples include near-collinear reference atoms (normalised angle within 1e-6 of 0 or 1), where any two f32
        # evaluations of the placement chain diverge: compare the bulk
        dx = (xs[0] - xs_b[0]).abs().max(1).values.cpu().numpy()
        r = rel_per_sample(dl.cpu().numpy(), dl_b.cpu().numpy())
        assert np.median(dx) <= 1e-5 and np.quantile(dx, 0.99) <= 2e-3
        assert np.median(r) <= 5e-6 and np.quantile(r, 0.99) <= 2e-4 and r.max() <= 5e-2
        # inputs that need gradients take the block path (autograd through the backward kernels)
        ur = [v[:64].clone().requires_grad_(True) for v in u]
        *xg, dlg = gen.flow(*ur)
        assert xg[0].grad_fn is not None


def _affine_stack(dev, s0, D, n_layers, hidden=(64, 64), swap_every=True, extra_tail=False):
    import bgflow_amd as bg
    from bgflow_amd.utils import hash_init_
    widths = [s0, D - s0]
    layers, part = [bg.SplitFlow(s0)], [0, 1]
    for _ in range(n_layers):
        d_c, d = widths[part[0]], widths[part[1]]
        layers.append(bg.CouplingFlow(bg.AffineTransformer(
            shift_transformation=bg.DenseNet([d_c, *hidden, d], activation=torch.nn.ReLU()),
            scale_transformation=bg.DenseNet([d_c, *hidden, d], activation=torch.nn.Tanh()))))
        if swap_every:
            layers.append(bg.SwapFlow())
            part.reverse()
    if part == [0, 1]:
        layers.append(bg.MergeFlow(s0))
    else:
        layers.append(bg.MergeFlow(widths[1]))         # odd number of swaps: the merge sees (part 1, part 0)
    if extra_tail:
        layers.append(bg.SplitFlow(D // 2)); layers.append(bg.MergeFlow(D // 2))
    return hash_init_(bg.SequentialFlow(layers)).to(dev)


@pytest.mark.parametrize("hidden", [(64, 64), (128, 128), (64, 64, 64)])
@pytest.mark.parametrize("s0,D,n_layers,B", [(32, 64, 8, 4099), (24, 64, 5, 1000), (32, 64, 2, 31), (40, 72, 3, 257)])
def test_fused_coupling_stack_equals_blocks(hip_lib, dev, s0, D, n_layers, B, hidden):
    """Split -> (affine coupling, swap)* -> Merge on one [B, D] buffer (in-place layers, in-kernel dlogp accumulation, no cat)
    against the same blocks run one by one: bit-identical in both directions (same kernels, same summation order)"""
    import bgflow_amd as bg
    flow = _affine_stack(dev, s0, D, n_layers, hidden=hidden)      # (64, 64): weight-resident kernel; the others: streaming kernel
    assert [lbl for lbl, _ in flow.segments()] == ["coupling stack"] and [lbl for lbl, _ in flow.segments(inverse=True)] == ["coupling stack"]
    g = torch.Generator(device=dev).manual_seed(3)
    z = torch.randn(B, D, device=dev, generator=g)
    z_keep = z.clone()
    with torch.no_grad():
        x1, d1 = flow(z)
        zi1, di1 = flow(x1, inverse=True)
        bg.SequentialFlow.FUSE_COUPLING_STACKS = False
        try:
            assert "coupling stack" not in [lbl for lbl, _ in flow.segments()]
            x0, d0 = flow(z)
            zi0, di0 = flow(x1, inverse=True)
        finally:
            bg.SequentialFlow.FUSE_COUPLING_STACKS = True
    assert torch.equal(z, z_keep)                       # the input is never written
    assert torch.equal(x1, x0) and torch.equal(d1, d0) and d1.shape == (B, 1)
    assert torch.equal(zi1, zi0) and torch.equal(di1, di0)
    assert float((zi1 - z).abs().max()) < 1e-3 * max(1.0, float(z.abs().max()))
    # gradients requested: the stack must step aside (autograd needs the per-layer tensors)
    zg = z[:64].clone().requires_grad_(True)
    xg, dg = flow(zg)
    (xg.sum() + dg.sum()).backward()
    assert zg.grad is not None and torch.isfinite(zg.grad).all()


def test_fused_coupling_stack_no_swap_and_neighbours(hip_lib, dev):
    """a half no layer transforms is copied through; blocks after the stack still run"""
    import bgflow_amd as bg
    flow = _affine_stack(dev, 32, 64, 3, swap_every=False, extra_tail=True)
    labels = [lbl for lbl, _ in flow.segments()]
    assert labels[0] == "coupling stack" and len(labels) == 3
    z = torch.randn(777, 64, device=dev)
    with torch.no_grad():
        x1, d1 = flow(z)
        bg.SequentialFlow.FUSE_COUPLING_STACKS = False
        try:
            x0, d0 = flow(z)
        finally:
            bg.SequentialFlow.FUSE_COUPLING_STACKS = True
    assert torch.equal(x1, x0) and torch.equal(d1, d0)
    assert torch.equal(x1[:, :32], z[:, :32])


def test_distribution_transfer_and_constrain_gaussian_on_gpu(hip_lib, dev):
    """DistributionTransferFlow / ConstrainGaussianFlow (cdf.py:49-121) run two bgk_cdf_transform launches: same results as the
    distributions' torch ops (kernel path switched off through a learnable marginal / f64), learnable truncated normal differentiable"""
    import bgflow_amd as bg
    from torch.distributions import Normal
    g = torch.Generator(device=dev).manual_seed(2)
    mu, sig = torch.linspace(0.5, 1.5, 12, device=dev), torch.linspace(0.8, 1.2, 12, device=dev)
    flow = bg.ConstrainGaussianFlow(mu=mu, sigma=sig, lower_bound=0.1, upper_bound=3.0)
    x = mu + sig * torch.randn(5000, 12, device=dev, generator=g)
    with torch.no_grad():
        y, dl = flow.forward(x)
        assert all(c.kernel_descriptor(12, dev) is not None for c in (flow._trafo._blocks[0], flow._trafo._blocks[1]._delegate))
        xb, dlb = flow.forward(y, inverse=True)
        # reference evaluation in f64 on the distributions' torch ops
        f64 = bg.ConstrainGaussianFlow(mu=mu.double(), sigma=sig.double(), lower_bound=0.1, upper_bound=3.0)
        y64, dl64 = f64.forward(x.double())
    assert bool(((y >= 0.1) & (y <= 3.0)).all())
    core = (x - mu).abs() < 3.0 * sig                    # the icdf of f32 tails is ill-conditioned: compare the bulk tightly, all loosely
    assert float((y.double() - y64)[core].abs().max()) < 2e-5 and float((y.double() - y64).abs().max()) < 5e-3
    rows = core.all(dim=1)
    assert float((dl.double() - dl64)[rows].abs().max()) < 2e-4
    assert float((xb - x)[core].abs().max()) < 1e-3 and float((dl + dlb)[rows].abs().max()) < 1e-3
    # the reference's own gradient check (tests/nn/flow/test_cdf.py::test_cdf_transform): learnable marginal, input needs grad
    inp = torch.arange(0.1, 1.0, 0.1, device=dev)[:, None].requires_grad_(True)
    tn = bg.TruncatedNormalDistribution(mu=torch.tensor([0.5], device=dev), upper_bound=torch.tensor([1.0], device=dev), is_learnable=True)
    flow2 = bg.InverseFlow(bg.CDFTransform(tn))
    out, dlogp = flow2.forward(inp)
    assert abs(float(out.mean()) - 0.5) < 1e-5
    out.mean().backward(retain_graph=True)
    dlogp.mean().backward()
    assert inp.grad is not None and tn._mu.grad is not None and torch.isfinite(tn._mu.grad).all()


@pytest.mark.parametrize("n_atoms", [150, 230, 500, 1500])
def test_relative_ic_large_molecules(hip_lib, oracle, dev, n_atoms):
    """molecules whose atoms do not fit the default 64-sample LDS tile (> 213 atoms): the IC kernels shrink the tile instead of
    refusing (150 atoms = default tile, for comparison) -- forward, inverse and both backward kernels against the oracle on a
    synthetic polymer chain with side branches, generated from well-conditioned internal coordinates"""
    import bgflow_amd as bg
    rng = np.random.RandomState(n_atoms)
    z = np.zeros((n_atoms - 3, 4), dtype=np.int64)
    for k, i in enumerate(range(3, n_atoms)):
        z[k] = (i, i - 1, i - 2, i - 3) if i % 4 else (i, i - 2, i - 3, i - 4) if i >= 4 else (i, i - 1, i - 2, i - 3)
    fixed = np.array([0, 1, 2])
    B, n = 37, n_atoms - 3
    bonds = (0.15 + 0.01 * rng.randn(B, n)).astype(np.float32)
    angles = (0.5 + 0.1 * (rng.rand(B, n) - 0.5)).astype(np.float32)           # normalised: 0.45 .. 0.55 of pi
    tors = rng.rand(B, n).astype(np.float32)
    xfix = (np.array([0, 0, 0, 0.15, 0, 0, 0.2, 0.14, 0], dtype=np.float32)[None] + 0.005 * rng.randn(B, 9)).astype(np.float32)
    ic = bg.RelativeInternalCoordinateTransformation(z, fixed, normalize_angles=True).to(dev)
    with torch.no_grad():
        xg, dli = ic(t(bonds, dev), t(angles, dev), t(tors, dev), t(xfix, dev), inverse=True)
    ox, odli = oracle.ic_ic2xyz(bonds, angles, tors, xfix, z, fixed, dtype=np.float64)
    ex = np.abs(xg.cpu().numpy() - ox)
    assert ex.max() < 2e-6 * n_atoms + 1e-5, ex.max()                          # f32 round-off accumulating along the chain
    assert np.abs(dli.cpu().numpy().reshape(-1) - odli.reshape(-1)).max() <= 1e-5 * np.abs(odli).max() + 1e-4
    with torch.no_grad():
        b, a, tt, xf, dl = ic(xg)
    np.testing.assert_allclose(b.cpu().numpy(), bonds, rtol=0, atol=1e-5)
    np.testing.assert_allclose(a.cpu().numpy(), angles, rtol=0, atol=2e-5)
    dt = np.abs(tt.cpu().numpy() - tors); dt = np.minimum(dt, 1.0 - dt)         # torsions are periodic on [0, 1)
    assert dt.max() < 1e-4
    # forward log-det against the oracle on the same xyz (dl = -dli only where no eps clamp fired: at 230+ atoms a few random
    # samples have a clamped norm, in the oracle / reference exactly as here)
    odl = oracle.ic_xyz2ic(xg.cpu().numpy(), z, fixed, dtype=np.float64)[4]
    assert np.abs(dl.cpu().numpy().reshape(-1) - odl.reshape(-1)).max() <= 1e-5 * np.abs(odl).max() + 1e-3
    assert float(np.median((dl + dli).abs().cpu().numpy())) <= 1e-5 * float(dl.abs().max()) + 1e-3
    # backward kernels: IC -> xyz against the oracle's VJP, xyz -> IC against a finite difference along one direction
    bg_, ag_, tg_, fg_ = (t(v, dev).requires_grad_(True) for v in (bonds, angles, tors, xfix))
    xo, dlo = ic(bg_, ag_, tg_, fg_, inverse=True)
    w = t(synth(11, B, 3 * n_atoms), dev)
    ((xo * w).sum() + dlo.sum()).backward()
    gb, ga, gt, gf = oracle.ic_ic2xyz_backward(bonds, angles, tors, ox, w.cpu().numpy(), np.ones(B), z, fixed, dtype=np.float64)
    for got, want in ((bg_.grad, gb), (ag_.grad, ga), (tg_.grad, gt), (fg_.grad, gf)):
        err = np.abs(got.cpu().numpy() - want)
        # f32 reverse sweep over a chain of n placements: lever arms make early gradients 1e3..1e4, round-off grows with n
        assert err.max() <= (2e-4 + 2e-6 * n_atoms) * np.abs(want).max() + 1e-4, (err.max(), np.abs(want).max())
    # xyz -> IC backward: directional derivative against a central difference of the ORACLE's f64 forward (an f32 difference
    # quotient of a sum over 10^4 terms is all noise)
    x0 = xg.detach().clone().requires_grad_(True)
    b2, a2, t2, f2, dl2 = ic(x0)
    ((b2 * 3.0).sum() + a2.sum() + 0.1 * dl2.sum()).backward()
    v = synth(12, B, 3 * n_atoms).astype(np.float64)
    v /= np.linalg.norm(v)

    def f(xx):
        bb, aa, _, _, dd = oracle.ic_xyz2ic(xx, z, fixed, dtype=np.float64)
        return float((bb * 3.0).sum() + aa.sum() + 0.1 * dd.sum())
    x64, h = xg.cpu().numpy().astype(np.float64), 1e-6
    fd = (f(x64 + h * v) - f(x64 - h * v)) / (2 * h)
    an = float((x0.grad.cpu().numpy().astype(np.float64) * v).sum())
    assert abs(fd - an) <= (2e-3 + 2e-6 * n_atoms) * abs(fd) + 1e-3, (fd, an)


@pytest.mark.parametrize("d,B,has_mean,temperature", [(66, 4133, True, 1.0), (64, 1, False, 1.0), (2, 1000, True, 2.5), (9, 257, False, 0.5)])
def test_normal_energy_kernel(hip_lib, dev, d, B, has_mean, temperature):
    """bgk_normal_energy / _backward (target end of the KL integrand, normal.py:61-72) against the torch op chain in f64"""
    import bgflow_amd as bg
    g = torch.Generator(device=dev).manual_seed(d + B)
    mean = torch.randn(d, device=dev, generator=g) if has_mean else None
    dist = bg.NormalDistribution(d, mean).to(dev)
    x = (3.0 * torch.randn(B, d, device=dev, generator=g)).requires_grad_(True)
    u = dist.energy(x, temperature=temperature)
    w = torch.randn(B, 1, device=dev, generator=g)
    (u * w).sum().backward()
    x64 = x.detach().double().requires_grad_(True)
    xc = x64 - mean.double() if has_mean else x64
    u64 = 0.5 * (xc / temperature ** 0.5).pow(2).sum(-1, keepdim=True) + d / 2 * np.log(2 * np.pi * temperature)
    (u64 * w.double()).sum().backward()
    assert u.shape == (B, 1)
    np.testing.assert_allclose(u.detach().cpu().numpy(), u64.detach().cpu().numpy(), rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), x64.grad.cpu().numpy(), rtol=2e-6, atol=1e-6)
    # a strided view (columns of a wider matrix) and the no-grad path
    wide = torch.randn(B, d + 5, device=dev, generator=g)
    with torch.no_grad():
        u2 = dist.energy(wide[:, 2:2 + d], temperature=temperature)
        ref = dist.energy(wide[:, 2:2 + d].double().cpu(), temperature=temperature) if False else None
    v = wide[:, 2:2 + d].double() - (mean.double() if has_mean else 0.0)
    np.testing.assert_allclose(u2.cpu().numpy(), (0.5 * v.pow(2).sum(-1, keepdim=True) / temperature + d / 2 * np.log(2 * np.pi * temperature)).cpu().numpy(),
                               rtol=2e-6, atol=1e-5)


@pytest.mark.parametrize("inverse", [False, True])
def test_spline_beyond_the_register_resident_instances(hip_lib, dev, inverse):
    """n_bins = 80 (> 64: beyond the LDS-staged kernels and the backward kernel's register-resident instances) through
    ConditionalSplineTransformer on the device: forward AND backward on the direct variants of bgk_rqs_transform / bgk_rqs_backward
    (round 5: compensated knot sums beyond 64 bins, no device torch ops); checked against the torch restatement of the nflows spline in
    oracle/ evaluated in f64"""
    import bgflow_amd as bg
    from oracle import torch_flow as tf
    Kb, d, B = 80, 5, 64
    circ = np.array([1, 0, 1, 0, 0], bool)
    n_nc = int((~circ).sum())
    P = 3 * Kb * d + n_nc
    params, y = synth(300 + Kb, B, P, scale=0.7), synth(400 + Kb, B, d, uniform=True)
    p = t(params, dev).requires_grad_(True)
    yy = t(y, dev).requires_grad_(True)

    class Fixed(torch.nn.Module):
        def forward(self, x):
            return p
    tr = bg.ConditionalSplineTransformer(Fixed(), is_circular=torch.tensor(circ))
    z, dl = tr(torch.zeros(B, 1, device=dev), yy, inverse=inverse)
    (z.sum() + dl.sum()).backward()
    p64 = torch.tensor(params, dtype=torch.float64, requires_grad=True)
    y64 = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    w, h, sl, s_nc = torch.split(p64, [d * Kb, d * Kb, d * Kb, n_nc], dim=-1)
    w, h, sl = (v.reshape(B, d, Kb) for v in (w, h, sl))
    cm = torch.tensor(circ)
    nc_full = torch.zeros(B, d, dtype=torch.float64).index_put((torch.arange(B)[:, None], torch.nonzero(~cm).reshape(1, -1)), s_nc)
    last = torch.where(cm[None, :], sl[..., 0], nc_full)
    st = tr._default_settings
    out, ld = tf.rq_spline(y64.clamp(0.0, 1.0), w, h, torch.cat([sl, last[..., None]], -1), not inverse, 0.0, 1.0, 0.0, 1.0,
                           st["min_bin_width"], st["min_bin_height"], st["min_derivative"], st.get("enable_identity_init", False))
    (out.sum() + ld.sum()).backward()
    # (knots by compensated f32 sums: ~1 ulp each, whatever the bin count)
    assert float((z.detach().cpu().double() - out.detach()).abs().max()) < 2e-6
    # (a bin of size ~1 / K between two knots of 1 ulp each: relative error ~K eps per bin, twice that in its log-det term)
    assert float((dl.detach().cpu().double().reshape(-1) - ld.detach().sum(-1)).abs().max()) < 2e-4
    for got, want in ((yy.grad, y64.grad), (p.grad, p64.grad)):
        err = (got.cpu().double() - want).abs()
        assert float(err.max()) <= 1e-3 * float(want.abs().max()) and float(err.quantile(0.99)) <= 1e-4 * float(want.abs().max())
