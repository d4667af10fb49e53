"""CPU: pin the oracle (oracle/bgo_oracle.c) against golden vectors generated from the reference
(tests/golden/make_goldens.py).  f64 flavour: agreement to double rounding.  f32 flavour: as close
to the reference's f64 result as the reference's own f32 path is (both are f32 evaluations of the
same formulas; their mutual distance is bounded by the f32 self-noise the fixtures record)."""
import numpy as np
import pytest

from bgflow_amd.utils import synth

K = 8
UNIT_CASES = [("nc17", 17, np.zeros(17, bool)), ("c17", 17, np.ones(17, bool)), ("nc9", 9, np.zeros(9, bool)),
              ("mix6", 6, np.array([1, 0, 1, 1, 0, 0], bool))]


def unit_inputs(name, d, circ, B=128):
    n_nc = int((~circ).sum())
    P = 3 * K * d + n_nc
    return synth(100 + d + int(circ.sum()), B, P, scale=0.5), synth(200 + d, B, d, uniform=True)


@pytest.mark.parametrize("name,d,circ", UNIT_CASES)
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_unit_f64(oracle, golden, name, d, circ, inverse):
    G = golden("g_rqs_unit")
    params, y = unit_inputs(name, d, circ)
    tag = f"{name}_{'inv' if inverse else 'fwd'}"
    z, dl, det = oracle.rqs(y, params, is_circular=circ, inverse=inverse, dtype=np.float64, want_details=True)
    assert np.array_equal(det["bin_idx"], G[tag + "_idx64"])
    np.testing.assert_allclose(z, G[tag + "_z64"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(dl, G[tag + "_dlogp64"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name,d,circ", UNIT_CASES)
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_unit_f32(oracle, golden, name, d, circ, inverse):
    G = golden("g_rqs_unit")
    params, y = unit_inputs(name, d, circ)
    tag = f"{name}_{'inv' if inverse else 'fwd'}"
    z, dl, det = oracle.rqs(y, params, is_circular=circ, inverse=inverse, dtype=np.float32, want_details=True)
    # bin indices identical to the reference's f32 path on the fixture set
    assert np.array_equal(det["bin_idx"], G[tag + "_idx32"])
    ref_noise_z = np.abs(G[tag + "_z32"] - G[tag + "_z64"]).max()
    ref_noise_dl = np.abs(G[tag + "_dlogp32"] - G[tag + "_dlogp64"]).max()
    assert np.abs(z - G[tag + "_z64"]).max() <= 3 * ref_noise_z + 2e-7
    assert np.abs(dl - G[tag + "_dlogp64"]).max() <= 3 * ref_noise_dl + 1e-6
    # north-star tolerance: log-det-J within 1e-5 relative (of the batch scale of |dlogp|) of the f32 reference
    scale = np.sqrt((G[tag + "_dlogp32"] ** 2).mean())
    assert np.abs(dl - G[tag + "_dlogp32"]).max() <= 1e-5 * scale + 1e-5


@pytest.mark.parametrize("Kb", [4, 6, 12, 16, 32])
@pytest.mark.parametrize("circ", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_other_bin_counts(oracle, golden, Kb, circ, inverse):
    """bin counts other than the default 8 (the transformer infers K from the parameter width, transformer/spline.py:113-126):
    the oracle against the reference in f64 (double rounding) and in f32 (within the reference's own f32 noise; bin indices
    equal, or the sample sits within rounding distance of a knot)"""
    G = golden("g_rqs_bins")
    d, B = 5, 96
    P = 3 * Kb * d + (0 if circ else d)
    params = synth(300 + Kb + 100 * int(circ), B, P, scale=0.5)
    y = synth(400 + Kb, B, d, uniform=True)
    tag = f"K{Kb}_{'c' if circ else 'nc'}_{'inv' if inverse else 'fwd'}"
    is_circ = np.full(d, circ, bool)
    z, dl, det = oracle.rqs(y, params, is_circular=is_circ, inverse=inverse, n_bins=Kb, dtype=np.float64, want_details=True)
    assert np.array_equal(det["bin_idx"], G[tag + "_idx64"])
    np.testing.assert_allclose(z, G[tag + "_z64"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(dl, G[tag + "_dlogp64"], rtol=1e-12, atol=1e-12)
    z, dl, det = oracle.rqs(y, params, is_circular=is_circ, inverse=inverse, n_bins=Kb, dtype=np.float32, want_details=True)
    mism = det["bin_idx"] != G[tag + "_idx32"]
    if mism.any():
        dist = np.abs(det["knots"] - y[..., None]).min(-1)
        assert (dist[mism] <= 2.4e-7).all() and mism.sum() <= 2, f"non-tie bin mismatches: {mism.sum()}"
    ok = ~mism.any(-1)
    ref_noise_z = np.abs(G[tag + "_z32"] - G[tag + "_z64"]).max()
    ref_noise_dl = np.abs(G[tag + "_dlogp32"] - G[tag + "_dlogp64"]).max()
    # outputs live in [0, 1]: a few f32 ulps (96 samples: the reference's own f32 noise on them can be as low as 2e-7)
    assert np.abs(z[ok] - G[tag + "_z64"][ok]).max() <= max(3 * ref_noise_z, 1.5e-6)
    assert np.abs(dl[ok] - G[tag + "_dlogp64"][ok]).max() <= 3 * ref_noise_dl + 1e-6


@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_edges(oracle, golden, inverse):
    """domain ends, exact knots and knots +-1 ulp: indices must agree with the reference except for
    exact ties (|x - knot| <= 1 ulp of the oracle's own knot), which are reported."""
    G = golden("g_rqs_unit")
    tag = "edge_inv" if inverse else "edge_fwd"
    d = 4
    P = 3 * K * d + d
    y = G[tag + "_y"]
    params = np.repeat(synth(7, 1, P, scale=0.7), len(y), 0)
    z, dl, det = oracle.rqs(y, params, inverse=inverse, dtype=np.float32, want_details=True)
    mism = det["bin_idx"] != G[tag + "_idx32"]
    if mism.any():
        # every mismatch must be an ulp-tie between x and one of the oracle's knots
        kn = det["knots"]
        dist = np.abs(kn - y[..., None]).min(-1)
        assert (dist[mism] <= 2.4e-7).all(), f"non-tie bin mismatches: {mism.sum()}"
    ok = ~mism.any(-1)
    np.testing.assert_allclose(z[ok], G[tag + "_z32"][ok], rtol=0, atol=5e-6)
    assert mism.mean() < 0.25


def test_rqs_out_of_domain_clamps(oracle, golden):
    G = golden("g_rqs_unit")
    d = 4
    P = 3 * K * d + d
    params = synth(12, 16, P, scale=0.5)
    y = G["oob_y"]
    for tag, inverse in (("oob_fwd", False), ("oob_inv", True)):
        z, dl, det = oracle.rqs(y, params, inverse=inverse, dtype=np.float32, want_details=True)
        assert det["n_oob"] == 3
        assert np.array_equal(det["bin_idx"], G[tag + "_idx32"])
        np.testing.assert_allclose(z, G[tag + "_z32"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(dl, G[tag + "_dlogp32"], rtol=1e-5, atol=2e-5)


def test_rqs_zero_params_is_identity(oracle, golden):
    G = golden("g_rqs_unit")
    d = 4
    y = synth(13, 32, d, uniform=True)
    z, dl = oracle.rqs(y, np.zeros((32, 3 * K * d + d), np.float32), dtype=np.float32)
    np.testing.assert_allclose(z, y, rtol=0, atol=1e-6)
    np.testing.assert_allclose(dl, 0, atol=2e-6)
    np.testing.assert_allclose(z, G["zero_z32"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_saturated(oracle, golden, inverse):
    G = golden("g_rqs_unit")
    d = 4
    P = 3 * K * d + d
    y = synth(13, 32, d, uniform=True)
    params = synth(14, 32, P, scale=12.0)
    tag = "sat_inv" if inverse else "sat_fwd"
    z, dl, det = oracle.rqs(y, params, inverse=inverse, dtype=np.float64, want_details=True)
    assert np.array_equal(det["bin_idx"], G[tag + "_idx64"])
    np.testing.assert_allclose(z, G[tag + "_z64"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(dl, G[tag + "_dlogp64"], rtol=1e-11, atol=1e-10)
    z, dl, det = oracle.rqs(y, params, inverse=inverse, dtype=np.float32, want_details=True)
    assert (det["bin_idx"] != G[tag + "_idx32"]).mean() < 0.02   # saturated softmax: knots collapse onto each other


@pytest.mark.parametrize("tag,kw", [("plain", {}), ("vp", dict(preserve_volume=True))])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_unit(oracle, golden, tag, kw, inverse):
    G = golden("g_affine_unit")
    y, mu, s = synth(21, 128, 32), synth(22, 128, 32), synth(23, 128, 32, scale=2.0)
    key = f"{tag}_{'inv' if inverse else 'fwd'}_"
    o, dl = oracle.affine(y, mu, s, log_alpha=-1.0, inverse=inverse, dtype=np.float64, **kw)
    np.testing.assert_allclose(o, G[key + "z64"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(dl, G[key + "dlogp64"], rtol=1e-12, atol=1e-13)
    o, dl = oracle.affine(y, mu, s, log_alpha=-1.0, inverse=inverse, dtype=np.float32, **kw)
    np.testing.assert_allclose(o, G[key + "z32"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(dl, G[key + "dlogp32"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("inverse", [False, True])
def test_affine_circular(oracle, golden, inverse):
    G = golden("g_affine_unit")
    y, mu = synth(24, 128, 32, uniform=True), synth(22, 128, 32)
    key = f"circ_{'inv' if inverse else 'fwd'}_"
    o, dl = oracle.affine(y, mu, None, is_circular=True, inverse=inverse, dtype=np.float32)
    assert np.array_equal(o, G[key + "z32"])
    assert (o >= 0).all() and (o <= 1).all()
    assert np.array_equal(dl, G[key + "dlogp32"])


def test_decompose_z_matrix(oracle, golden):
    G = golden("g_ic")
    blocks, i2a, a2i, i2o, table = oracle.decompose_z_matrix(G["z_matrix"], G["rigid_block"])
    assert [len(b) for b in blocks] == list(G["dec_block_sizes"])
    assert np.array_equal(np.concatenate(blocks), G["dec_blocks"])
    assert np.array_equal(i2a, G["dec_index2atom"])
    assert np.array_equal(a2i, G["dec_atom2index"])
    assert np.array_equal(i2o, G["dec_index2order"])


@pytest.mark.parametrize("dtype,sfx,tol_ic,tol_dl", [(np.float64, "64", 1e-13, 1e-12), (np.float32, "32", 5e-7, 4e-5)])
def test_ic_relative_and_mixed(oracle, golden, dtype, sfx, tol_ic, tol_dl):
    G = golden("g_ic")
    z, rigid, x = G["z_matrix"], G["rigid_block"], G["x"]
    b, a, t, xf, dl = oracle.ic_xyz2ic(x, z, rigid, dtype=dtype)
    for got, key in ((b, "rel_bonds"), (a, "rel_angles"), (t, "rel_torsions"), (xf, "rel_xfixed")):
        np.testing.assert_allclose(got, G[key + sfx], rtol=0, atol=tol_ic)
    np.testing.assert_allclose(dl, G["rel_dlogp" + sfx], rtol=0, atol=tol_dl)
    xb, dli = oracle.ic_ic2xyz(G["rel_bonds" + sfx], G["rel_angles" + sfx], G["rel_torsions" + sfx],
                               G["rel_xfixed" + sfx], z, rigid, dtype=dtype)
    np.testing.assert_allclose(xb, G["rel_xback" + sfx], rtol=0, atol=4 * tol_ic)
    np.testing.assert_allclose(dli, G["rel_dlogp_inv" + sfx], rtol=0, atol=tol_dl)
    sf = "" if sfx == "32" else "64"
    jac = -np.log(G["wh_std" + sf].astype(np.float64)).sum()
    b, a, t, zf, dl = oracle.ic_xyz2ic(x, z, rigid, whiten=(G["wh_mean" + sf], G["wh_Twhiten" + sf], jac), dtype=dtype)
    np.testing.assert_allclose(zf, G["mix_zfixed" + sfx], rtol=0, atol=8 * tol_ic)
    np.testing.assert_allclose(dl, G["mix_dlogp" + sfx], rtol=0, atol=tol_dl)
    xg, dlg = oracle.ic_ic2xyz(G["gen_bonds"], G["gen_angles"], G["gen_torsions"], G["gen_zfixed"], z, rigid,
                               blacken=(G["wh_mean" + sf], G["wh_Tblacken" + sf], jac), dtype=dtype)
    np.testing.assert_allclose(xg, G["gen_x" + sfx], rtol=0, atol=4 * tol_ic)
    np.testing.assert_allclose(dlg, G["gen_dlogp" + sfx], rtol=0, atol=tol_dl)
    # log-det within 1e-5 relative (|dlogp| ~ 20..60 here)
    assert (np.abs(dlg - G["gen_dlogp" + sfx]) / np.abs(G["gen_dlogp" + sfx])).max() < 1e-5


def test_ic_singular_geometry_is_clamped_like_the_reference(oracle, golden):
    G = golden("g_ic")
    b, a, t, xf, dl = oracle.ic_xyz2ic(G["x_singular"], G["z_matrix"], G["rigid_block"], dtype=np.float32)
    np.testing.assert_allclose(b, G["sing_bonds32"], rtol=0, atol=1e-6)     # collapsed bond -> eps clamp
    np.testing.assert_allclose(a, G["sing_angles32"], rtol=0, atol=1e-6)    # straight angle -> cos clamp
    fin = np.isfinite(G["sing_dlogp32"]).ravel()
    np.testing.assert_allclose(dl.ravel()[fin], G["sing_dlogp32"].ravel()[fin], rtol=2e-5, atol=2e-4)
    assert not np.isfinite(dl.ravel()[~fin]).any() or (dl.ravel()[~fin] < -80).all()


def test_readme_and_affine8_flows(golden):
    """whole affine flows (cfg 1, cfg 2) through the oracle walker vs reference"""
    import torch
    from bgflow_amd import configs
    from oracle import flow_oracle as fo
    G = golden("g_readme")
    gen = configs.make_readme_generator()
    (x,), dl = fo.run_flow(gen.flow, [G["z"]], dtype=np.float32)
    np.testing.assert_allclose(x, G["x"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(dl, G["dlogp"], rtol=1e-5, atol=1e-6)
    (zb,), dli = fo.run_flow(gen.flow, [G["x"]], inverse=True, dtype=np.float32)
    np.testing.assert_allclose(zb, G["z_back"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dli, G["dlogp_inv"], rtol=1e-5, atol=1e-6)
    G = golden("g_affine8")
    z = synth(32, 128, 64)
    for tdt, dt, sfx, rt, at in ((torch.float64, np.float64, "64", 1e-11, 1e-12), (torch.float32, np.float32, "32", 1e-5, 3e-5)):
        gen = configs.make_affine8_generator().to(tdt)
        (x,), dl = fo.run_flow(gen.flow, [z], dtype=dt)
        np.testing.assert_allclose(x, G["x" + sfx], rtol=rt * 10, atol=at)
        np.testing.assert_allclose(dl, G["dlogp" + sfx], rtol=rt, atol=at)
        (zb,), dli = fo.run_flow(gen.flow, [G["x" + sfx]], inverse=True, dtype=dt)
        np.testing.assert_allclose(dli, G["dlogp_inv" + sfx], rtol=rt, atol=at)


def test_envelope_layers_through_the_oracle_vs_reference(golden):
    """coupling layers with wide / deep conditioners (g_envelope: hidden (256, 256), (200, 130), one / three / four hidden layers; affine
    networks with one / five / four hidden layers) through the oracle walker against the reference's outputs, f64 to rounding and f32
    to f32 noise, both directions"""
    import envelope_layers as el
    from oracle import flow_oracle as fo
    G = golden("g_envelope")
    for tag, hidden in el.SPLINE.items():
        for kind, (periodic, circular) in el.KINDS.items():
            c, y = el.spline_inputs(periodic)
            layer = el.spline_layer(hidden, periodic, circular)
            key = f"s_{tag}_{kind}_"
            for dt, sfx, at in ((np.float64, "64", 1e-11), (np.float32, "32", 2e-5)):
                (_, z), dl = fo.run_block(layer, [c.astype(dt), y.astype(dt)], False, dt, [])
                np.testing.assert_allclose(z, G[key + "z" + sfx], rtol=0, atol=at)
                np.testing.assert_allclose(dl, G[key + "dlogp" + sfx], rtol=at, atol=at * 10)
                (_, yb), dli = fo.run_block(layer, [c.astype(dt), G[key + "z" + sfx]], True, dt, [])
                np.testing.assert_allclose(yb, G[key + "back" + sfx], rtol=0, atol=at)
                np.testing.assert_allclose(dli, G[key + "dlogp_inv" + sfx], rtol=at, atol=at * 10)
    for tag, hidden in el.AFFINE.items():
        c, y = el.affine_inputs()
        layer = el.affine_layer(hidden)
        key = f"a_{tag}_"
        for dt, sfx, at in ((np.float64, "64", 1e-11), (np.float32, "32", 2e-5)):
            (_, z), dl = fo.run_block(layer, [c.astype(dt), y.astype(dt)], False, dt, [])
            np.testing.assert_allclose(z, G[key + "z" + sfx], rtol=at, atol=at)
            np.testing.assert_allclose(dl, G[key + "dlogp" + sfx], rtol=at, atol=at)
            (_, yb), dli = fo.run_block(layer, [c.astype(dt), G[key + "z" + sfx]], True, dt, [])
            np.testing.assert_allclose(yb, G[key + "back" + sfx], rtol=at, atol=at)
            np.testing.assert_allclose(dli, G[key + "dlogp_inv" + sfx], rtol=at, atol=at)


def test_flow16_whole_flow(golden):
    """cfg 3: 16 spline couplings + domain maps + mixed IC, oracle walker vs the reference builder flow"""
    import torch
    from bgflow_amd import configs
    from oracle import flow_oracle as fo
    G = golden("g_flow16")
    u = [G["u_bonds"], G["u_angles"], G["u_torsions"], G["u_fixed"]]
    gen64 = configs.make_ala2_spline_generator(dtype=torch.float64).double()
    assert sum(p.numel() for p in gen64.flow.parameters()) == int(G["n_params"]) and len(gen64.flow) == int(G["n_blocks"])
    (x,), dl = fo.run_flow(gen64.flow, u, dtype=np.float64)
    np.testing.assert_allclose(x, G["x64"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(dl, G["dlogp64"], rtol=1e-12, atol=1e-11)
    zb, dli = fo.run_flow(gen64.flow, [G["x64"]], inverse=True, dtype=np.float64)
    np.testing.assert_allclose(np.concatenate(zb, -1), G["z_back64"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(dli, G["dlogp_inv64"], rtol=1e-9, atol=1e-8)
    # f32: per-block agreement with the reference's f32 path in IC space, and log-det-J within 1e-5
    # relative of the f64 truth up to the reference's own f32 noise
    gen32 = configs.make_ala2_spline_generator()
    pb = []
    (x,), dl = fo.run_flow(gen32.flow, u, dtype=np.float32, per_block=pb)
    for i in range(16):
        got = np.concatenate([*pb[i][0], pb[i][1]], axis=-1)
        np.testing.assert_allclose(got, G[f"block{i:02d}_32"], rtol=0, atol=2e-5)
    noise = np.abs(G["dlogp32"] - G["dlogp64"]).max()
    assert np.abs(dl - G["dlogp64"]).max() <= 1e-5 * np.abs(G["dlogp64"]).max() + noise
    assert (np.abs(dl - G["dlogp32"]) / np.abs(G["dlogp32"])).max() < 2e-5


def test_augmented_flow_cfg5(golden):
    """cfg 5 (10 spline + 6 affine couplings over 5 fields incl. 66 auxiliary dims) vs the reference builder flow"""
    import torch
    from bgflow_amd import configs
    from oracle import flow_oracle as fo
    G = golden("g_aug")
    u = [G[k] for k in ("u_bonds", "u_angles", "u_torsions", "u_fixed", "u_aug")]
    gen64 = configs.make_ala2_augmented_generator(dtype=torch.float64).double()
    assert sum(p.numel() for p in gen64.flow.parameters()) == int(G["n_params"]) and len(gen64.flow) == int(G["n_blocks"])
    assert [type(b).__name__ + ":" + type(getattr(b, "transformer", b)).__name__ for b in gen64.flow] == list(G["block_types"])
    (x, aug), dl = fo.run_flow(gen64.flow, u, dtype=np.float64)
    np.testing.assert_allclose(x, G["x64"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(aug, G["aug64"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(dl, G["dlogp64"], rtol=1e-11, atol=1e-8)
    zb, dli = fo.run_flow(gen64.flow, [G["x64"], G["aug64"]], inverse=True, dtype=np.float64)
    np.testing.assert_allclose(dli, G["dlogp_inv64"], rtol=1e-8, atol=1e-6)
    # f32: this flow is ill-conditioned in f32 (the reference's own f32 path is 0.86 off its f64 result in dlogp):
    # require our f32 evaluation to be as close to the f64 truth as the reference's
    gen32 = configs.make_ala2_augmented_generator()
    (x, aug), dl = fo.run_flow(gen32.flow, u, dtype=np.float32)
    assert np.abs(dl - G["dlogp64"]).max() <= 2 * np.abs(G["dlogp32"] - G["dlogp64"]).max()
    assert np.abs(aug - G["aug64"]).max() <= 2 * np.abs(G["aug32"] - G["aug64"]).max()
    # (clamped icdf inputs, eps = 1e-7 = half an f32 ulp of 1.0, dominate: each costs ~0.17 in log-prob in f32)


# ---- analytic backward (VJP) restatements vs torch autograd through the reference ---------------------
GRAD_CASES = [("nc17", 17, np.zeros(17, bool)), ("c9", 9, np.ones(9, bool)), ("mix6", 6, np.array([1, 0, 1, 1, 0, 0], bool))]


def grad_inputs(d, circ, B=48):
    n_nc = int((~circ).sum())
    P = 3 * K * d + n_nc
    return (synth(300 + d, B, P, scale=0.7), synth(400 + d, B, d, uniform=True), synth(500 + d, B, d), synth(600 + d, B, 1))


@pytest.mark.parametrize("name,d,circ", GRAD_CASES)
@pytest.mark.parametrize("inverse", [False, True])
def test_rqs_backward_vs_reference_autograd(oracle, golden, name, d, circ, inverse):
    G = golden("g_grads")
    params, y, a, bw = grad_inputs(d, circ)
    tag = f"rqs_{name}_{'inv' if inverse else 'fwd'}"
    gy, gp = oracle.rqs_backward(y, params, a, bw, is_circular=circ, inverse=inverse, dtype=np.float64)
    np.testing.assert_allclose(gy, G[tag + "_gy64"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gp, G[tag + "_gp64"], rtol=1e-9, atol=1e-12)
    gy, gp = oracle.rqs_backward(y, params, a, bw, is_circular=circ, inverse=inverse, dtype=np.float32)
    assert np.abs(gy - G[tag + "_gy64"]).max() <= 1e-4 * np.abs(G[tag + "_gy64"]).max()
    assert np.abs(gp - G[tag + "_gp64"]).max() <= 1e-4 * np.abs(G[tag + "_gp64"]).max()


@pytest.mark.parametrize("pv", [False, True])
@pytest.mark.parametrize("inverse", [False, True])
def test_affine_backward_vs_reference_autograd(oracle, golden, pv, inverse):
    G = golden("g_grads")
    B, d = 64, 12
    y, mu, s, a, bw = synth(1, B, d), synth(2, B, d), synth(3, B, d, scale=1.5), synth(4, B, d), synth(5, B, 1)
    tag = f"aff_{'vp' if pv else 'plain'}_{'inv' if inverse else 'fwd'}"
    gy, gm, gs, gla = oracle.affine_backward(y, mu, s, a, bw, log_alpha=-1.0, preserve_volume=pv, inverse=inverse, dtype=np.float64)
    np.testing.assert_allclose(gy, G[tag + "_gy"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(gm, G[tag + "_gmu"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(gs, G[tag + "_gs"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(gla, float(G[tag + "_gla"][0]), rtol=1e-11, atol=1e-12)


def test_ic_backward_vs_reference_autograd(oracle, golden):
    G, Gic = golden("g_grads"), golden("g_ic")
    z, rigid = Gic["z_matrix"], Gic["rigid_block"]
    a, bw = synth(77, 128, 66), synth(78, 128, 1)
    jac = -np.log(Gic["wh_std64"]).sum()
    ins = [Gic[k] for k in ("gen_bonds", "gen_angles", "gen_torsions")]
    gb, ga, gt, gf = oracle.ic_ic2xyz_backward(*ins, G["ic_x"], a, bw, z, rigid,
                                               blacken=(Gic["wh_mean64"], Gic["wh_Tblacken64"], jac), dtype=np.float64)
    for got, key in ((gb, "ic_g_bonds"), (ga, "ic_g_angles"), (gt, "ic_g_torsions"), (gf, "ic_g_zfixed")):
        np.testing.assert_allclose(got, G[key], rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("dtype,sfx,tol,tol_dl", [(np.float64, "64", 1e-13, 1e-11), (np.float32, "32", 2e-6, 1e-4)])
def test_global_ic(oracle, golden, dtype, sfx, tol, tol_dl):
    """GlobalInternalCoordinateTransformation (a15): closed-form 9x9 log-det vs the reference's autograd Jacobian"""
    G = golden("g_ic")
    zg, x = G["global_z_matrix"], G["x"][:64]
    b, a, t, x0, R, dl = oracle.global_ic_forward(x, zg, dtype=dtype)
    assert b.shape == (64, 21) and a.shape == (64, 20) and t.shape == (64, 19)
    for got, key in ((b, "glob_bonds"), (a, "glob_angles"), (t, "glob_torsions"), (x0, "glob_x0"), (R, "glob_R")):
        np.testing.assert_allclose(got, G[key + sfx], rtol=0, atol=tol)
    np.testing.assert_allclose(dl, G["glob_dlogp" + sfx], rtol=0, atol=tol_dl)
    xb, dli = oracle.global_ic_inverse(G["glob_bonds" + sfx], G["glob_angles" + sfx], G["glob_torsions" + sfx],
                                       G["glob_x0" + sfx], G["glob_R" + sfx], zg, dtype=dtype)
    np.testing.assert_allclose(dli, G["glob_dlogp_inv" + sfx], rtol=0, atol=tol_dl)
    # the reference's own inverse is only 3e-8 accurate in f64 (eps clamps); compare with the original x as well
    np.testing.assert_allclose(xb, G["glob_xback" + sfx], rtol=0, atol=max(tol, 1e-7) * 10)
    np.testing.assert_allclose(xb, x, rtol=0, atol=max(tol * 50, 1e-12))


def test_torch_cpu_restatement_matches_reference_goldens(golden):
    """oracle/torch_flow.py (bench.py's second cpu_baseline leg: the reference's op chain on stock torch-CPU ops) reproduces
    the reference's own f32 outputs on the cfg-3 / cfg-5 fixtures -- forward and inverse -- to f32 round-off"""
    import torch
    from bgflow_amd import configs
    from oracle import torch_flow as tfl
    for name, make, keys in (("g_flow16", configs.make_ala2_spline_generator, ("u_bonds", "u_angles", "u_torsions", "u_fixed")),
                             ("g_aug", configs.make_ala2_augmented_generator, ("u_bonds", "u_angles", "u_torsions", "u_fixed", "u_aug"))):
        G = golden(name)
        gen = make()
        xs, dl = tfl.run_flow(gen.flow, [torch.tensor(G[k]) for k in keys])
        scale = np.abs(G["dlogp32"]).max()
        assert np.abs(dl.numpy() - G["dlogp32"]).max() <= 2e-7 * scale + 2e-5
        assert np.abs(xs[0].numpy() - G["x32"]).max() <= 1e-4
        # inverse direction from the reference's own x (the icdf maps are ill-conditioned in x near the domain ends)
        xin = [torch.tensor(G["x32"])] + ([torch.tensor(G["aug32"])] if "aug32" in G else [])
        zs, dli = tfl.run_flow(gen.flow, xin, inverse=True)
        e = np.abs(dli.numpy() - G["dlogp_inv32"]).reshape(-1)
        assert np.median(e) <= 1e-4 and np.quantile(e, 0.9) <= 1e-3 and e.max() <= 1e-3 * scale   # cdf maps: steep near the domain ends
        assert np.abs(torch.cat(list(zs), -1).numpy() - G["z_back32"]).max() <= 1e-4


def test_philox_oracle_known_answers():
    """oracle/philox.py against Random123's known-answer vectors for Philox4x32-10 (kat_vectors: counter / key all zero, all ones,
    digits of pi): the generator of the opt-in fused prior sampler (csrc/bgk_philox.hip) is pinned before it is used as a checker"""
    from oracle import philox
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(ctr, key)
        assert tuple(int(v) for v in got) == want
    u = philox.sample_field(1234, 0, 0, 1000, 17, 0)
    assert u.dtype == np.float32 and u.min() > 0.0 and u.max() < 1.0 and abs(float(u.mean()) - 0.5) < 0.01
    nrm = philox.sample_field(1234, 0, 1, 4000, 66, 1)
    assert abs(float(nrm.mean())) < 0.01 and abs(float(nrm.std()) - 1.0) < 0.01


def test_energies_and_weights_host_path_vs_reference_goldens(golden):
    """f-3 on the host path (torch-CPU ops of bgflow_amd.distributions / bg): DoubleWellEnergy, NormalDistribution, UniformDistribution,
    ProductDistribution energies, forces, log weights and ESS against values generated from the reference classes"""
    import torch
    import bgflow_amd as bg
    from bgflow_amd.bg import log_weights_given_latent, effective_sample_size
    G = golden("g_energies")
    x = torch.tensor(G["dw_x"]).double()
    for tag, kw in (("dw", {}), ("dw_abc", dict(a=0.7, b=-2.5, c=0.4))):
        e = bg.DoubleWellEnergy(64, **kw)
        np.testing.assert_allclose(e.energy(x).numpy(), G[tag + "_u64"], rtol=1e-12)
        np.testing.assert_allclose(e.energy(x, temperature=2.5).numpy(), G[tag + "_uT64"], rtol=1e-12)
        np.testing.assert_allclose(e.force(x.clone(), temperature=2.5).numpy(), G[tag + "_force64"], rtol=1e-12, atol=1e-12)
    mean = torch.tensor(G["norm_mean"]).double()
    comps = [bg.NormalDistribution(66, mean=mean), bg.NormalDistribution(66),       # (f32 buffers like the fixture's: log Z rounds to f32)
             bg.UniformDistribution(torch.tensor(G["unif_low"]).double(), torch.tensor(G["unif_high"]).double())]
    xs = tuple(torch.tensor(G[k]).double() for k in ("prod_y", "prod_a", "prod_un"))
    prod = bg.ProductDistribution(comps)
    np.testing.assert_allclose(comps[0].energy(xs[0], temperature=1.7).numpy(), G["norm_u64"], rtol=1e-10)
    np.testing.assert_allclose(comps[2].energy(xs[2]).numpy(), G["unif_u64"], rtol=1e-10)
    np.testing.assert_allclose(prod.energy(*xs).numpy(), G["prod_u64"], rtol=1e-10)
    np.testing.assert_allclose(prod.energy(*xs, temperature=1.7).numpy(), G["prod_uT64"], rtol=1e-10)
    lw = log_weights_given_latent(xs[0], xs[1], torch.tensor(G["logw_dl"]).double(), bg.NormalDistribution(66),
                                  bg.NormalDistribution(66, mean=mean), temperature=1.3, normalize=True)
    np.testing.assert_allclose(lw.numpy(), G["logw64"], rtol=1e-6, atol=1e-6)      # (the golden's dlogp went through f32 once)
    np.testing.assert_allclose(effective_sample_size(lw).numpy(), G["ess64"], rtol=1e-5)
