#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/*.npz by IMPORTING the reference.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_goldens.py

What runs: the UNMODIFIED reference classes (bgflow.ConditionalSplineTransformer, AffineTransformer,
CouplingFlow, SequentialFlow, the internal-coordinate transforms, BoltzmannGeneratorBuilder, ...)
on the PyTorch-CPU path.  Two shims, neither touching the reference tree:
  * ``numpy.infty = numpy.inf`` (removed in numpy 2; used at import by distribution/normal.py:126);
  * the third-party ``nflows`` package (absent here) is replaced by tests/golden/nflows_stub.py, a
    restatement of its public rational-quadratic-spline algorithm.  Its evaluate/search core is
    cross-checked below against the reference's own in-tree copy bgflow/nn/flow/spline.py.

All network weights come from bgflow_amd.utils.hash_init_ (closed-form, name-keyed), so the tests can
rebuild identical networks without weight files.  The fixtures hold DATA only: inputs, expected
outputs, and the Z-matrix / geometry tables that the reference's own test-suite embeds
(tests/nn/flow/crd_transform/test_ic.py:37-118).
"""
import os
import sys
import warnings

import numpy

numpy.infty = numpy.inf  # numpy-2 shim, see module docstring

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/tests")

import nflows_stub  # noqa: E402

nflows_stub.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bgflow as bg  # noqa: E402
from bgflow.factory.generator_builder import BoltzmannGeneratorBuilder  # noqa: E402
from bgflow.factory.tensor_info import ShapeDictionary, BONDS, ANGLES, TORSIONS, FIXED, AUGMENTED  # noqa: E402
import importlib  # noqa: E402
intree_spline = importlib.import_module("bgflow.nn.flow.spline")  # dead code in the reference; oracle evidence
from bgflow_amd.utils import hash_init_, synth  # noqa: E402

warnings.filterwarnings("ignore")
torch.set_num_threads(4)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


class FixedNet(torch.nn.Module):
    """Conditioner that returns a fixed tensor (so the spline parameters ARE the fixture input)."""

    def __init__(self, out):
        super().__init__()
        self.out = out

    def forward(self, x):
        return self.out


def rng_f32(seed, *shape, scale=1.0, uniform=False):
    """fixture INPUTS are closed-form (bgflow_amd.utils.synth): tests regenerate them, files store outputs"""
    return synth(seed, *shape, scale=scale, uniform=uniform)


# ---------------------------------------------------------------------------------------------
# G-rqs-unit
# ---------------------------------------------------------------------------------------------
def run_spline(params, y, circ, inverse, dtype):
    p = torch.tensor(params, dtype=dtype)
    yy = torch.tensor(y, dtype=dtype)
    tr = bg.ConditionalSplineTransformer(FixedNet(p), is_circular=circ)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        z, dl = tr(torch.zeros(len(y), 1, dtype=dtype), yy, inverse=inverse)
        warned = any("InputOutsideDomain" in str(x.message) for x in w)
    last = {k: v.numpy().copy() for k, v in nflows_stub.LAST.items()}
    return z.numpy(), dl.numpy(), last, warned


def intree_check(params, y, circ_mask, inverse, last):
    """Feed the knots/derivatives the stub produced to the reference's in-tree rq_spline
    (bgflow/nn/flow/spline.py:60-180, offsets set to 0) and require bitwise agreement in fp64
    of the evaluate / root-solve / log-det core."""
    yy = torch.tensor(y, dtype=torch.float64)
    outs, lads = [], []
    cw = torch.tensor(last["cumwidths"]).clone()
    ch = torch.tensor(last["cumheights"]).clone()
    # undo the in-place eps (the in-tree searchsorted adds its own)
    (ch if not inverse else cw)[..., -1] -= 1e-6
    der = torch.tensor(last["derivatives"])
    # rq_spline wants per-distribution (not per-sample) knots: loop samples (small fixture)
    for b in range(min(len(y), 32)):
        o, l = intree_spline.rq_spline(
            yy[b], cw[b].clone(), ch[b].clone(), der[b].clone(), inverse=not inverse,
            min_bin_width=0.0, min_bin_height=0.0, min_derivative=0.0)
        outs.append(o.numpy()); lads.append(l.numpy())
    return np.stack(outs), np.stack(lads)


def g_rqs_unit():
    K = 8
    cases = {}
    intree_err = [0.0, 0.0]
    for name, d, circ in [
        ("nc17", 17, np.zeros(17, bool)),
        ("c17", 17, np.ones(17, bool)),
        ("nc9", 9, np.zeros(9, bool)),
        ("mix6", 6, np.array([1, 0, 1, 1, 0, 0], bool)),  # #circ == #noncirc: the only mixed case the reference supports
    ]:
        n_nc = int((~circ).sum())
        P = 3 * K * d + n_nc
        B = 128
        params = rng_f32(100 + d + int(circ.sum()), B, P, scale=0.5)
        y = rng_f32(200 + d, B, d, uniform=True)
        circ_arg = bool(circ[0]) if circ.all() or (~circ).all() else torch.tensor(circ)
        cases[f"{name}_circ"] = circ   # inputs: synth(100+d+n_circ, B, P, scale=.5), synth(200+d, B, d, uniform=True)
        for inverse in (False, True):
            tag = f"{name}_{'inv' if inverse else 'fwd'}"
            z32, dl32, last32, _ = run_spline(params, y, circ_arg, inverse, torch.float32)
            z64, dl64, last64, _ = run_spline(params, y, circ_arg, inverse, torch.float64)
            o_it, lad_it = intree_check(params, y, circ, inverse, last64)
            sign = -1.0 if not inverse else 1.0
            # in-tree copy returns per-element logdet with the same sign convention as nflows
            # (the in-tree copy re-derives the knots from cumsum(diff(knots)), so agreement is to a few
            #  fp64 ulps rather than bitwise)
            intree_err[0] = max(intree_err[0], np.abs(o_it - z64[: len(o_it)]).max())
            lad_sum = lad_it.sum(-1, keepdims=True)
            intree_err[1] = max(intree_err[1], np.abs(lad_sum - dl64[: len(o_it)]).max())
            assert intree_err[0] < 1e-13 and intree_err[1] < 1e-11, f"in-tree rq_spline mismatch ({tag}): {intree_err}"
            cases[f"{tag}_z32"] = z32
            cases[f"{tag}_dlogp32"] = dl32
            cases[f"{tag}_z64"] = z64
            cases[f"{tag}_dlogp64"] = dl64
            cases[f"{tag}_idx32"] = last32["bin_idx"].astype(np.int32)
            cases[f"{tag}_idx64"] = last64["bin_idx"].astype(np.int32)
    print(f"  in-tree bgflow/nn/flow/spline.py cross-check (fp64): max|dy| {intree_err[0]:.2e}, max|dlogdet| {intree_err[1]:.2e}")

    # ---- edge vectors (non-circular, d=4): domain ends, exact knots, +-1ulp, clamp path, saturation ----
    d = 4
    P = 3 * K * d + d
    base = rng_f32(7, 1, P, scale=0.7)
    _, _, last, _ = run_spline(np.repeat(base, 1, 0), np.full((1, d), 0.5, np.float32), False, False, torch.float32)
    kh = last["cumheights"][0].copy()   # [d, K+1] (fwd searches cumheights); last one carries +1e-6
    _, _, last, _ = run_spline(np.repeat(base, 1, 0), np.full((1, d), 0.5, np.float32), False, True, torch.float32)
    kw = last["cumwidths"][0].copy()
    rows_f, rows_i = [], []
    for k in range(K + 1):
        for off in (-1, 0, 1):
            vf = kh[:, k].copy(); vi = kw[:, k].copy()
            if off:
                vf = np.nextafter(vf, np.float32(2.0 * off)).astype(np.float32)
                vi = np.nextafter(vi, np.float32(2.0 * off)).astype(np.float32)
            rows_f.append(np.clip(vf, 0, 1)); rows_i.append(np.clip(vi, 0, 1))
    for v in (0.0, 1.0, 1e-9, 1 - 1e-7, 0.5):
        rows_f.append(np.full(d, v, np.float32)); rows_i.append(np.full(d, v, np.float32))
    yf = np.stack(rows_f).astype(np.float32); yi = np.stack(rows_i).astype(np.float32)
    pf = np.repeat(base, len(yf), 0)
    for tag, yy, inverse in (("edge_fwd", yf, False), ("edge_inv", yi, True)):
        z32, dl32, last32, _ = run_spline(pf, yy, False, inverse, torch.float32)
        z64, dl64, last64, _ = run_spline(pf, yy, False, inverse, torch.float64)
        cases[f"{tag}_y"] = yy
        cases[f"{tag}_z32"] = z32; cases[f"{tag}_dlogp32"] = dl32
        cases[f"{tag}_z64"] = z64; cases[f"{tag}_dlogp64"] = dl64
        cases[f"{tag}_idx32"] = last32["bin_idx"].astype(np.int32)
        cases[f"{tag}_knots32"] = (last32["cumwidths"] if inverse else last32["cumheights"])
    # out-of-domain (clamp + warning path, spline.py:145-155)
    yo = rng_f32(11, 16, d, uniform=True)
    yo[0, 0] = -0.25; yo[3, 2] = 1.5; yo[7, 1] = 1.0000001
    po = rng_f32(12, 16, P, scale=0.5)
    for tag, inverse in (("oob_fwd", False), ("oob_inv", True)):
        z32, dl32, last32, warned = run_spline(po, yo, False, inverse, torch.float32)
        assert warned
        cases[f"{tag}_z32"] = z32; cases[f"{tag}_dlogp32"] = dl32
        cases[f"{tag}_idx32"] = last32["bin_idx"].astype(np.int32)
    cases["oob_y"] = yo   # params: synth(12, 16, P, scale=.5)
    # zero parameters = identity (enable_identity_init), and saturated parameters
    yz = rng_f32(13, 32, d, uniform=True)
    pz = np.zeros((32, P), np.float32)
    z32, dl32, _, _ = run_spline(pz, yz, False, False, torch.float32)
    cases["zero_z32"] = z32; cases["zero_dlogp32"] = dl32   # y: synth(13, 32, d, uniform=True)
    ps = rng_f32(14, 32, P, scale=12.0)   # softmax saturation / softplus threshold (beta*s > 20)
    for tag, inverse in (("sat_fwd", False), ("sat_inv", True)):
        z32, dl32, last32, _ = run_spline(ps, yz, False, inverse, torch.float32)
        z64, dl64, last64, _ = run_spline(ps, yz, False, inverse, torch.float64)
        cases[f"{tag}_z32"] = z32; cases[f"{tag}_dlogp32"] = dl32
        cases[f"{tag}_z64"] = z64; cases[f"{tag}_dlogp64"] = dl64
        cases[f"{tag}_idx32"] = last32["bin_idx"].astype(np.int32)
        cases[f"{tag}_idx64"] = last64["bin_idx"].astype(np.int32)
    save("g_rqs_unit", **cases)


def g_rqs_bins():
    """other bin counts (the transformer infers K from the parameter width, transformer/spline.py:113-126): the fused kernels
    take K = 4 | 12 | 16 | 32 besides the default 8, the generic kernel any K.  inputs: synth(300 + K + 100 * circular, B, P,
    scale=.5), synth(400 + K, B, d, uniform=True)"""
    cases = {}
    d, B = 5, 96
    for K in (4, 6, 12, 16, 32):
        for circ in (False, True):
            P = 3 * K * d + (0 if circ else d)
            params = rng_f32(300 + K + 100 * int(circ), B, P, scale=0.5)
            y = rng_f32(400 + K, B, d, uniform=True)
            for inverse in (False, True):
                tag = f"K{K}_{'c' if circ else 'nc'}_{'inv' if inverse else 'fwd'}"
                z32, dl32, last32, _ = run_spline(params, y, circ, inverse, torch.float32)
                z64, dl64, last64, _ = run_spline(params, y, circ, inverse, torch.float64)
                assert last64["bin_idx"].max() == K - 1 or last64["bin_idx"].max() < K
                cases[f"{tag}_z32"] = z32; cases[f"{tag}_dlogp32"] = dl32
                cases[f"{tag}_z64"] = z64; cases[f"{tag}_dlogp64"] = dl64
                cases[f"{tag}_idx32"] = last32["bin_idx"].astype(np.int32)
                cases[f"{tag}_idx64"] = last64["bin_idx"].astype(np.int32)
    save("g_rqs_bins", **cases)


# ---------------------------------------------------------------------------------------------
# G-affine (unit + README flow + cfg-2 flow)
# ---------------------------------------------------------------------------------------------
def g_affine():
    out = {}
    B, d = 128, 32
    y = rng_f32(21, B, d)
    mu = rng_f32(22, B, d)
    s = rng_f32(23, B, d, scale=2.0)
    x = torch.zeros(B, 1)
    for tag, kw in [("plain", {}), ("vp", dict(preserve_volume=True))]:
        tr = bg.AffineTransformer(FixedNet(torch.tensor(mu)), FixedNet(torch.tensor(s)), **kw)
        for inverse in (False, True):
            for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
                tr2 = bg.AffineTransformer(FixedNet(torch.tensor(mu, dtype=dt)), FixedNet(torch.tensor(s, dtype=dt)), **kw)
                z, dl = tr2(x.to(dt), torch.tensor(y, dtype=dt), inverse=inverse)
                out[f"{tag}_{'inv' if inverse else 'fwd'}_z{sfx}"] = z.detach().numpy()
                out[f"{tag}_{'inv' if inverse else 'fwd'}_dlogp{sfx}"] = dl.detach().numpy()
    # shift-only circular (NICE on a circle)
    yc = rng_f32(24, B, d, uniform=True)
    tr = bg.AffineTransformer(FixedNet(torch.tensor(mu)), None, is_circular=True)
    for inverse in (False, True):
        z, dl = tr(x, torch.tensor(yc), inverse=inverse)
        out[f"circ_{'inv' if inverse else 'fwd'}_z32"] = z.numpy()
        out[f"circ_{'inv' if inverse else 'fwd'}_dlogp32"] = dl.numpy()
    out.update(log_alpha=np.float32(-1.0))  # inputs: synth(21..24)
    save("g_affine_unit", **out)

    # README flow (cfg 1), README.md:54-96
    dim = 2
    prior = bg.NormalDistribution(dim)
    target = bg.DoubleWellEnergy(dim)
    layers = [bg.SplitFlow(dim // 2),
              bg.CouplingFlow(bg.AffineTransformer(
                  shift_transformation=bg.DenseNet([dim // 2, 4, dim // 2], activation=torch.nn.ReLU()),
                  scale_transformation=bg.DenseNet([dim // 2, 4, dim // 2], activation=torch.nn.Tanh()))),
              bg.InverseFlow(bg.SplitFlow(dim // 2))]
    flow = bg.SequentialFlow(layers)
    hash_init_(flow)
    gen = bg.BoltzmannGenerator(prior, flow, target)
    z = rng_f32(31, 64, dim)
    with torch.no_grad():
        x, dlogp = flow(torch.tensor(z))
        zi, dlogp_inv = flow(x, inverse=True)
        nll = gen.energy(x)
        kl_terms = target.energy(x) - dlogp
    save("g_readme", z=z, x=x.numpy(), dlogp=dlogp.numpy(), z_back=zi.numpy(), dlogp_inv=dlogp_inv.numpy(),
         nll=nll.numpy(), kl_terms=kl_terms.numpy())

    # cfg 2: dim 64, 8 x (affine coupling, swap)
    dim = 64
    layers = [bg.SplitFlow(dim // 2)]
    for _ in range(8):
        layers.append(bg.CouplingFlow(bg.AffineTransformer(
            shift_transformation=bg.DenseNet([32, 64, 64, 32], activation=torch.nn.ReLU()),
            scale_transformation=bg.DenseNet([32, 64, 64, 32], activation=torch.nn.Tanh()))))
        layers.append(bg.SwapFlow())
    layers.append(bg.MergeFlow(dim // 2))
    flow = bg.SequentialFlow(layers)
    hash_init_(flow)
    z = rng_f32(32, 128, dim)
    target = bg.DoubleWellEnergy(dim)
    res = {}
    for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
        f = flow.to(dt)
        with torch.no_grad():
            x, dlogp = f(torch.tensor(z, dtype=dt))
            zb, dli = f(x, inverse=True)
            res[f"x{sfx}"] = x.numpy(); res[f"dlogp{sfx}"] = dlogp.numpy()
            res[f"z_back{sfx}"] = zb.numpy(); res[f"dlogp_inv{sfx}"] = dli.numpy()
            res[f"kl_terms{sfx}"] = (target.energy(x) - dlogp).numpy()
    save("g_affine8", z=z, **res)


# ---------------------------------------------------------------------------------------------
# G-ic
# ---------------------------------------------------------------------------------------------
def ala2_tables():
    from nn.flow.crd_transform.test_ic import alanine_ics
    zrel, zglob, rigid, xyz = alanine_ics.__wrapped__()
    return zrel, zglob, rigid, xyz


def whitening_data(xyz):
    """1000 frames of the reference geometry + closed-form 0.01 nm noise (bgflow_amd.utils.synth)"""
    return (xyz + 0.01 * synth(0, 1000, 66, scale=1.0, dtype=np.float64)).astype(np.float32)


def g_ic():
    zrel, zglob, rigid, xyz = ala2_tables()
    data = whitening_data(xyz)
    g = np.random.default_rng(5)
    x = (xyz + 0.005 * g.standard_normal((128, 66))).astype(np.float32)
    # 8 near-singular frames: collapse a bond / straighten an angle (clamp path of ic_helper)
    xs = x[:8].copy().reshape(8, 22, 3)
    xs[0, 0] = xs[0, 1] + 1e-9                       # bond ~ 0
    xs[1, 2] = xs[1, 1] + (xs[1, 1] - xs[1, 4])       # angle 2-1-4 = pi
    xs[2, 5] = xs[2, 4] + 0.3 * (xs[2, 6] - xs[2, 4])  # collinear 5-4-6
    xs = xs.reshape(8, 66).astype(np.float32)
    out = dict(z_matrix=zrel.astype(np.int32), global_z_matrix=zglob.astype(np.int32),
               rigid_block=rigid.astype(np.int32), xyz0=xyz.astype(np.float64), x=x, x_singular=xs)
    for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
        rel = bg.RelativeInternalCoordinateTransformation(zrel, rigid, raise_warnings=False)
        mix = bg.MixedCoordinateTransformation(torch.tensor(data, dtype=dt), zrel, rigid, keepdims=9,
                                               raise_warnings=False)
        xt = torch.tensor(x, dtype=dt)
        b, a, t, xf, dl = rel(xt)
        xr, dlr = rel(b, a, t, xf, inverse=True)
        out.update({f"rel_bonds{sfx}": b.numpy(), f"rel_angles{sfx}": a.numpy(), f"rel_torsions{sfx}": t.numpy(),
                    f"rel_xfixed{sfx}": xf.numpy(), f"rel_dlogp{sfx}": dl.numpy(),
                    f"rel_xback{sfx}": xr.numpy(), f"rel_dlogp_inv{sfx}": dlr.numpy()})
        b, a, t, zf, dl = mix(xt)
        xr, dlr = mix(b, a, t, zf, inverse=True)
        out.update({f"mix_bonds{sfx}": b.numpy(), f"mix_angles{sfx}": a.numpy(), f"mix_torsions{sfx}": t.numpy(),
                    f"mix_zfixed{sfx}": zf.numpy(), f"mix_dlogp{sfx}": dl.numpy(),
                    f"mix_xback{sfx}": xr.numpy(), f"mix_dlogp_inv{sfx}": dlr.numpy()})
        if sfx == "32":
            out.update(wh_mean=mix._whiten.X0mean.numpy(), wh_Twhiten=mix._whiten.Twhiten.numpy(),
                       wh_Tblacken=mix._whiten.Tblacken.numpy(), wh_std=mix._whiten.std.numpy())
        else:
            out.update(wh_mean64=mix._whiten.X0mean.numpy(), wh_Twhiten64=mix._whiten.Twhiten.numpy(),
                       wh_Tblacken64=mix._whiten.Tblacken.numpy(), wh_std64=mix._whiten.std.numpy())
        xst = torch.tensor(xs, dtype=dt)
        b, a, t, xf, dl = rel(xst)
        out.update({f"sing_bonds{sfx}": b.numpy(), f"sing_angles{sfx}": a.numpy(), f"sing_torsions{sfx}": t.numpy(),
                    f"sing_dlogp{sfx}": dl.numpy()})
    rel = bg.RelativeInternalCoordinateTransformation(zrel, rigid)
    blocks, i2a, a2i, i2o = rel._z_blocks, rel._index2atom, rel._atom2index, rel._index2order
    out.update(dec_block_sizes=np.array([len(bk) for bk in blocks], np.int32), dec_blocks=np.concatenate(blocks).astype(np.int32),
               dec_index2atom=i2a.astype(np.int32), dec_atom2index=a2i.astype(np.int32), dec_index2order=i2o.astype(np.int32))
    # IC -> xyz on ICs that do not come from a forward pass (flow-generated ICs): perturbed ICs
    b32 = out["mix_bonds32"]; a32 = out["mix_angles32"]; t32 = out["mix_torsions32"]; z32 = out["mix_zfixed32"]
    g = np.random.default_rng(6)
    bp = (b32 * (1 + 0.05 * g.standard_normal(b32.shape))).astype(np.float32)
    ap = np.clip(a32 + 0.02 * g.standard_normal(a32.shape), 0.05, 0.95).astype(np.float32)
    tp = ((t32 + 0.1 * g.standard_normal(t32.shape)) % 1.0).astype(np.float32)
    zp = (z32 + 0.3 * g.standard_normal(z32.shape)).astype(np.float32)
    for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
        mix = bg.MixedCoordinateTransformation(torch.tensor(data, dtype=dt), zrel, rigid, keepdims=9, raise_warnings=False)
        xg, dlg = mix(*(torch.tensor(v, dtype=dt) for v in (bp, ap, tp, zp)), inverse=True)
        out.update({f"gen_x{sfx}": xg.numpy(), f"gen_dlogp{sfx}": dlg.numpy()})
    out.update(gen_bonds=bp, gen_angles=ap, gen_torsions=tp, gen_zfixed=zp)
    # Global internal coordinates (a15): forward on x, inverse on the result
    for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
        gic = bg.GlobalInternalCoordinateTransformation(zglob, raise_warnings=False)
        xt = torch.tensor(x[:64], dtype=dt)
        b, a, t, x0, R, dl = gic(xt)
        xb, dli = gic(b.detach(), a.detach(), t.detach(), x0.detach(), R.detach(), inverse=True)
        out.update({f"glob_bonds{sfx}": b.detach().numpy(), f"glob_angles{sfx}": a.detach().numpy(),
                    f"glob_torsions{sfx}": t.detach().numpy(), f"glob_x0{sfx}": x0.detach().numpy(),
                    f"glob_R{sfx}": R.detach().numpy(), f"glob_dlogp{sfx}": dl.detach().numpy(),
                    f"glob_xback{sfx}": xb.detach().numpy(), f"glob_dlogp_inv{sfx}": dli.detach().numpy()})
    save("g_ic", **out)
    # molecule definition used by bgflow_amd.configs (data only: topology tables + one geometry)
    np.savez_compressed(os.path.join(REPO, "bgflow_amd", "data", "ala2_system.npz"),
                        z_matrix=zrel.astype(np.int32), global_z_matrix=zglob.astype(np.int32),
                        rigid_block=rigid.astype(np.int32), xyz=xyz.astype(np.float64))


# ---------------------------------------------------------------------------------------------
# G-flow16 (cfg 3) and G-aug (cfg 5)
# ---------------------------------------------------------------------------------------------
def build_cfg3(dtype=torch.float32):
    zrel, zglob, rigid, xyz = ala2_tables()
    data = torch.tensor(whitening_data(xyz), dtype=dtype)
    ic = bg.MixedCoordinateTransformation(data, zrel, rigid, keepdims=9, raise_warnings=False)
    shape_info = ShapeDictionary.from_coordinate_transform(ic)
    target = bg.NormalDistribution(66, torch.tensor(xyz[0], dtype=dtype))
    builder = BoltzmannGeneratorBuilder(shape_info, target=target, dtype=dtype)
    for _ in range(4):
        builder.add_condition(TORSIONS, on=FIXED)
        builder.add_condition(FIXED, on=TORSIONS)
    for _ in range(4):
        builder.add_condition(BONDS, on=ANGLES)
        builder.add_condition(ANGLES, on=BONDS)
    builder.add_map_to_ic_domains()
    builder.add_map_to_cartesian(ic)
    gen = builder.build_generator()
    hash_init_(gen.flow)
    return gen


def build_cfg5(dtype=torch.float32):
    zrel, zglob, rigid, xyz = ala2_tables()
    data = torch.tensor(whitening_data(xyz), dtype=dtype)
    ic = bg.MixedCoordinateTransformation(data, zrel, rigid, keepdims=9, raise_warnings=False)
    shape_info = ShapeDictionary.from_coordinate_transform(ic, dim_augmented=66)
    target = bg.NormalDistribution(66, torch.tensor(xyz[0], dtype=dtype))
    builder = BoltzmannGeneratorBuilder(shape_info, target=target, dtype=dtype)
    builder.transformer_type[AUGMENTED] = bg.AffineTransformer
    for _ in range(4):
        builder.add_condition(TORSIONS, on=AUGMENTED)
        builder.add_condition(AUGMENTED, on=TORSIONS)
    for _ in range(2):
        builder.add_condition(BONDS, on=ANGLES)
        builder.add_condition(ANGLES, on=BONDS)
    for _ in range(2):
        builder.add_condition(FIXED, on=AUGMENTED)
        builder.add_condition(AUGMENTED, on=(FIXED, BONDS, ANGLES))
    builder.add_map_to_ic_domains()
    builder.add_map_to_cartesian(ic)
    gen = builder.build_generator()
    hash_init_(gen.flow)
    return gen


def run_flow_recording(flow, zs, inverse=False):
    """SequentialFlow.forward (sequential.py:49-59) with per-block outputs recorded."""
    xs = tuple(zs)
    dlogp = 0.0
    per_block = []
    blocks = list(flow._blocks)
    if inverse:
        blocks = blocks[::-1]
    for block in blocks:
        *xs, dd = block(*xs, inverse=inverse)
        dlogp = dlogp + dd
        per_block.append(torch.cat([*xs, dd], dim=-1).numpy())
    return xs, dlogp, per_block


def coupling_prefix(flow, zs):
    """state and accumulated log|det J| after the leading CouplingFlow blocks (i.e. before the icdf domain maps and the
    coordinate transform): the part of the flow the hand-written coupling kernels cover, pinned on its own."""
    xs = tuple(zs)
    dlogp = 0.0
    n = 0
    for block in flow._blocks:
        if not isinstance(block, bg.CouplingFlow):
            break
        *xs, dd = block(*xs)
        dlogp = dlogp + dd
        n += 1
    return torch.cat(list(xs), dim=-1).numpy(), dlogp.numpy(), n


def g_flow16():
    B = 64
    res = {}
    u = [rng_f32(41 + i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
    for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
        gen = build_cfg3(dt)
        with torch.no_grad():
            zs = [torch.tensor(v, dtype=dt) for v in u]
            xs, dlogp, per_block = run_flow_recording(gen.flow, zs)
            x = xs[0]
            res[f"x{sfx}"] = x.numpy(); res[f"dlogp{sfx}"] = dlogp.numpy()
            for i, pb in enumerate(per_block):
                if pb.shape[1] <= 61 and sfx == "32":  # IC-space blocks: cat of the 4 fields + ddlogp
                    res[f"block{i:02d}_{sfx}"] = pb
            zb, dli, _ = run_flow_recording(gen.flow, [x], inverse=True)
            res[f"z_back{sfx}"] = torch.cat(list(zb), dim=-1).numpy(); res[f"dlogp_inv{sfx}"] = dli.numpy()
            res[f"kl_terms{sfx}"] = (gen._target.energy(x) - dlogp).numpy()
            res[f"nll{sfx}"] = gen.energy(x).numpy()
            res[f"state_c{sfx}"], res[f"dlogp_c{sfx}"], n_c = coupling_prefix(gen.flow, zs)
            res["n_couplings"] = np.int64(n_c)
        if sfx == "32":
            res["n_params"] = np.int64(sum(p.numel() for p in gen.flow.parameters()))
            res["n_blocks"] = np.int64(len(gen.flow))
    save("g_flow16", u_bonds=u[0], u_angles=u[1], u_torsions=u[2], u_fixed=u[3], **res)


def g_aug():
    B = 64
    res = {}
    u = [rng_f32(51 + i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9, 66))]
    # auxiliary variables away from the ends of (0, 1): their images go through the Normal icdf (f32 erfinv: the reference's
    # own f32-vs-f64 gap explodes in the tails and would hide every other error)
    u[4] = (np.float32(0.1) + np.float32(0.8) * u[4]).astype(np.float32)
    for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
        gen = build_cfg5(dt)
        with torch.no_grad():
            zs = [torch.tensor(v, dtype=dt) for v in u]
            xs, dlogp, _ = run_flow_recording(gen.flow, zs)
            res[f"state_c{sfx}"], res[f"dlogp_c{sfx}"], n_c = coupling_prefix(gen.flow, zs)
            res["n_couplings"] = np.int64(n_c)
            res[f"x{sfx}"] = xs[0].numpy(); res[f"aug{sfx}"] = xs[1].numpy(); res[f"dlogp{sfx}"] = dlogp.numpy()
            zb, dli, _ = run_flow_recording(gen.flow, list(xs), inverse=True)
            res[f"z_back{sfx}"] = torch.cat(list(zb), dim=-1).numpy(); res[f"dlogp_inv{sfx}"] = dli.numpy()
        if sfx == "32":
            res["n_params"] = np.int64(sum(p.numel() for p in gen.flow.parameters()))
            res["n_blocks"] = np.int64(len(gen.flow))
            res["block_types"] = np.array([type(b).__name__ + ":" + type(getattr(b, "transformer", b)).__name__ for b in gen.flow])
    save("g_aug", u_bonds=u[0], u_angles=u[1], u_torsions=u[2], u_fixed=u[3], u_aug=u[4], **res)


# ---------------------------------------------------------------------------------------------
# G-augment: StochasticAugmentation (nn/flow/stochastic/augment.py:27-55) with pre-sampled momenta (no RNG in the fixture)
# ---------------------------------------------------------------------------------------------
def g_augment():
    from bgflow.nn.flow.stochastic.augment import StochasticAugmentation
    B, dim = 48, 6
    q, p = rng_f32(61, B, dim), rng_f32(62, B, dim, scale=1.3)
    out = dict(q=q, p=p)
    for T, sfx in ((1.0, "T1"), (2.5, "T2p5")):
        layer = StochasticAugmentation(bg.NormalDistribution(dim))
        with torch.no_grad():
            x, dl = layer(torch.tensor(q), momenta=torch.tensor(p), temperature=T, cache_momenta=True)
            qb, dli = layer(x, inverse=True, temperature=T, cache_momenta=True)
            xm, dlm = layer(x, inverse=True, temperature=T, return_momenta=True)
            e = layer.distribution.energy(torch.tensor(p), temperature=T)
        assert torch.equal(layer._cached_momenta_forward, torch.tensor(p)) and torch.equal(layer._cached_momenta_backward, torch.tensor(p))
        out.update({f"x_{sfx}": x.numpy(), f"dlogp_{sfx}": dl.numpy(), f"q_back_{sfx}": qb.numpy(), f"dlogp_inv_{sfx}": dli.numpy(),
                    f"x_mom_{sfx}": xm.numpy(), f"dlogp_mom_{sfx}": dlm.numpy(), f"energy_{sfx}": e.numpy()})
    save("g_augment", **out)


# ---------------------------------------------------------------------------------------------
# G-grads: gradients by torch autograd through the reference (pins the analytic backward kernels)
# ---------------------------------------------------------------------------------------------
def g_grads():
    out = {}
    K = 8
    for name, d, circ in [("nc17", 17, np.zeros(17, bool)), ("c9", 9, np.ones(9, bool)),
                          ("mix6", 6, np.array([1, 0, 1, 1, 0, 0], bool))]:
        n_nc = int((~circ).sum()); P = 3 * K * d + n_nc; B = 48
        params = synth(300 + d, B, P, scale=0.7); y = synth(400 + d, B, d, uniform=True)
        a = synth(500 + d, B, d); bw = synth(600 + d, B, 1)
        circ_arg = bool(circ[0]) if circ.all() or (~circ).all() else torch.tensor(circ)
        for inverse in (False, True):
            for dt, sfx in ((torch.float64, "64"),):
                p = torch.tensor(params, dtype=dt, requires_grad=True)
                yy = torch.tensor(y, dtype=dt, requires_grad=True)
                tr = bg.ConditionalSplineTransformer(FixedNet(p), is_circular=circ_arg)
                z, dl = tr(torch.zeros(B, 1, dtype=dt), yy, inverse=inverse)
                ((z * torch.tensor(a, dtype=dt)).sum() + (dl * torch.tensor(bw, dtype=dt)).sum()).backward()
                tag = f"rqs_{name}_{'inv' if inverse else 'fwd'}"
                out[f"{tag}_gy{sfx}"] = yy.grad.numpy(); out[f"{tag}_gp{sfx}"] = p.grad.numpy()
    B, d = 64, 12
    y = synth(1, B, d); mu = synth(2, B, d); s = synth(3, B, d, scale=1.5); a = synth(4, B, d); bw = synth(5, B, 1)
    for pv in (False, True):
        for inverse in (False, True):
            dt = torch.float64
            ty, tm, ts = (torch.tensor(v, dtype=dt, requires_grad=True) for v in (y, mu, s))
            tr = bg.AffineTransformer(FixedNet(tm), FixedNet(ts), preserve_volume=pv).double()
            z, dl = tr(torch.zeros(B, 1, dtype=dt), ty, inverse=inverse)
            ((z * torch.tensor(a, dtype=dt)).sum() + (dl * torch.tensor(bw, dtype=dt)).sum()).backward()
            tag = f"aff_{'vp' if pv else 'plain'}_{'inv' if inverse else 'fwd'}"
            out.update({f"{tag}_gy": ty.grad.numpy(), f"{tag}_gmu": tm.grad.numpy(), f"{tag}_gs": ts.grad.numpy(),
                        f"{tag}_gla": tr._log_alpha.grad.numpy()})
    # KL gradient of the cfg-2 flow (8 affine blocks): per-block d loss / d log_alpha, first-layer bias gradients
    dim = 64
    layers = [bg.SplitFlow(dim // 2)]
    for _ in range(8):
        layers.append(bg.CouplingFlow(bg.AffineTransformer(
            shift_transformation=bg.DenseNet([32, 64, 64, 32], activation=torch.nn.ReLU()),
            scale_transformation=bg.DenseNet([32, 64, 64, 32], activation=torch.nn.Tanh()))))
        layers.append(bg.SwapFlow())
    layers.append(bg.MergeFlow(dim // 2))
    flow = hash_init_(bg.SequentialFlow(layers)).double()
    target = bg.DoubleWellEnergy(dim)
    z = torch.tensor(synth(32, 128, dim), dtype=torch.float64)
    x, dlogp = flow(z)
    loss = (target.energy(x) - dlogp).mean()
    loss.backward()
    out["kl2_loss"] = loss.detach().numpy()
    out["kl2_g_log_alpha"] = np.array([float(flow[1 + 2 * i].transformer._log_alpha.grad) for i in range(8)])
    out["kl2_g_bias0"] = np.stack([flow[1 + 2 * i].transformer._shift_transformation._layers[0].bias.grad.numpy() for i in range(8)])
    out["kl2_gnorm"] = np.sqrt(sum(float((p.grad ** 2).sum()) for p in flow.parameters()))
    # KL-style gradient through the 16 spline couplings of cfg 3 (IC-space loss: sum of squares of the outputs - dlogp)
    gen = build_cfg3(torch.float64)
    sub = gen.flow[:16]
    u = [torch.tensor(rng_f32(41 + i, 64, dd, uniform=True), dtype=torch.float64) for i, dd in enumerate((17, 17, 17, 9))]
    *ys, dl = sub(*u)
    loss = (sum((v ** 2).sum(-1, keepdim=True) for v in ys) - dl).mean()
    loss.backward()
    out["kl3_loss"] = loss.detach().numpy()
    out["kl3_g_bias_last"] = np.stack([np.resize(b.transformer._params_net.net._layers[4].bias.grad.numpy()
                                                 if hasattr(b.transformer._params_net, "net")
                                                 else b.transformer._params_net._layers[4].bias.grad.numpy(), 200)
                                       for b in sub])
    out["kl3_gnorm"] = np.sqrt(sum(float((p.grad ** 2).sum()) for p in sub.parameters()))
    # IC -> xyz (Mixed) gradients: loss = sum(a * x) + sum(b * dlogp) on the flow-like ICs of g_ic
    Gic = np.load(os.path.join(HERE, "g_ic.npz"))
    zrel, zglob, rigid, xyz = ala2_tables()
    mix = bg.MixedCoordinateTransformation(torch.tensor(whitening_data(xyz), dtype=torch.float64), zrel, rigid,
                                           keepdims=9, raise_warnings=False)
    ins = [torch.tensor(Gic[k], dtype=torch.float64, requires_grad=True)
           for k in ("gen_bonds", "gen_angles", "gen_torsions", "gen_zfixed")]
    x, dl = mix(*ins, inverse=True)
    a = synth(77, 128, 66); bw = synth(78, 128, 1)
    ((x * torch.tensor(a, dtype=torch.float64)).sum() + (dl * torch.tensor(bw, dtype=torch.float64)).sum()).backward()
    out.update(ic_g_bonds=ins[0].grad.numpy(), ic_g_angles=ins[1].grad.numpy(), ic_g_torsions=ins[2].grad.numpy(),
               ic_g_zfixed=ins[3].grad.numpy(), ic_x=x.detach().numpy())
    # full cfg-3 KL gradient (prior sample -> 16 couplings -> icdf maps -> IC -> Normal(66) target)
    gen = build_cfg3(torch.float64)
    u = [torch.tensor(rng_f32(41 + i, 64, dd, uniform=True), dtype=torch.float64) for i, dd in enumerate((17, 17, 17, 9))]
    x, dlogp = gen.flow(*u)
    loss = (gen._target.energy(x) - dlogp).mean()
    loss.backward()
    out["klfull_loss"] = loss.detach().numpy()
    out["klfull_gnorm"] = np.sqrt(sum(float((p.grad ** 2).sum()) for p in gen.flow.parameters()))
    out["klfull_g_bias_last"] = np.stack([np.resize((b.transformer._params_net.net if hasattr(b.transformer._params_net, "net")
                                                      else b.transformer._params_net)._layers[4].bias.grad.numpy(), 200)
                                          for b in list(gen.flow)[:16]])
    save("g_grads", **out)


# ---------------------------------------------------------------------------------------------
# G-grads2: reference autograd gradients (f64) of the layers whose backward kernels came in round 2:
# CDFTransform (nn/flow/cdf.py:28-46), xyz -> IC (crd_transform/ic.py:386-433), global reference system (ic.py:162-265)
# ---------------------------------------------------------------------------------------------
def g_grads2():
    out = {}
    dt = torch.float64
    # --- the four domain maps of the cfg-3 builder flow, both directions
    gen = build_cfg3(dt)
    k = 0
    for block in gen.flow:
        inner = block
        while not isinstance(inner, bg.CDFTransform) and hasattr(inner, "_flow"):
            inner = inner._flow
        while not isinstance(inner, bg.CDFTransform) and hasattr(inner, "_delegate"):
            inner = inner._delegate
        if not isinstance(inner, bg.CDFTransform):
            continue
        d = {0: 17, 1: 17, 2: 17, 3: 9}[k]
        B = 40
        u = (0.02 + 0.96 * rng_f32(700 + k, B, d, uniform=True)).astype(np.float64)
        a, bw = synth(710 + k, B, d).astype(np.float64), synth(720 + k, B, 1).astype(np.float64)
        ut = torch.tensor(u, requires_grad=True)
        y, dl = inner(ut, inverse=True)                      # [0,1] -> physical (the direction the generator runs)
        ((y * torch.tensor(a)).sum() + (dl * torch.tensor(bw)).sum()).backward()
        out.update({f"cdf{k}_u": u, f"cdf{k}_a": a, f"cdf{k}_bw": bw, f"cdf{k}_y": y.detach().numpy(), f"cdf{k}_inv_gx": ut.grad.numpy()})
        xt = torch.tensor(y.detach().numpy(), requires_grad=True)
        uu, dl2 = inner(xt)
        ((uu * torch.tensor(a)).sum() + (dl2 * torch.tensor(bw)).sum()).backward()
        out.update({f"cdf{k}_fwd_gx": xt.grad.numpy(), f"cdf{k}_kind": np.array(type(inner.distribution).__name__)})
        k += 1
    assert k == 4
    # --- xyz -> IC (relative and mixed)
    Gic = np.load(os.path.join(HERE, "g_ic.npz"))
    zrel, zglob, rigid, xyz = ala2_tables()
    x = Gic["x"][:64].astype(np.float64)
    wb, wa, wt = synth(801, 64, 17).astype(np.float64), synth(802, 64, 17).astype(np.float64), synth(803, 64, 17).astype(np.float64)
    wf15, wf9, wl = synth(804, 64, 15).astype(np.float64), synth(805, 64, 9).astype(np.float64), synth(806, 64, 1).astype(np.float64)
    out.update(x2ic_wb=wb, x2ic_wa=wa, x2ic_wt=wt, x2ic_wf15=wf15, x2ic_wf9=wf9, x2ic_wl=wl)
    rel = bg.RelativeInternalCoordinateTransformation(zrel, rigid, raise_warnings=False)
    mix = bg.MixedCoordinateTransformation(torch.tensor(whitening_data(xyz), dtype=dt), zrel, rigid, keepdims=9, raise_warnings=False)
    for name, tr, wf in (("rel", rel, wf15), ("mix", mix, wf9)):
        xt = torch.tensor(x, requires_grad=True)
        b, a, t, f, dl = tr(xt)
        ((b * torch.tensor(wb)).sum() + (a * torch.tensor(wa)).sum() + (t * torch.tensor(wt)).sum() + (f * torch.tensor(wf)).sum()
         + (dl * torch.tensor(wl)).sum()).backward()
        out[f"x2ic_{name}_gx"] = xt.grad.numpy()
    # --- global internal coordinates: forward (x -> ICs) and inverse (ICs -> x)
    gic = bg.GlobalInternalCoordinateTransformation(zglob, raise_warnings=False)
    xg = Gic["x"][:32].astype(np.float64)
    xt = torch.tensor(xg, requires_grad=True)
    b, a, t, x0, R, dl = gic(xt)
    ws = [synth(900 + i, *v.shape).astype(np.float64) for i, v in enumerate((b, a, t, x0, R, dl))]
    sum((v * torch.tensor(w)).sum() for v, w in zip((b, a, t, x0, R, dl), ws)).backward()
    out.update(glob_x=xg, glob_fwd_gx=xt.grad.numpy(), **{f"glob_w{i}": w for i, w in enumerate(ws)})
    ins = [torch.tensor(v.detach().numpy(), requires_grad=True) for v in (b, a, t, x0, R)]
    xb, dli = gic(*ins, inverse=True)
    wx, wl2 = synth(910, 32, 66).astype(np.float64), synth(911, 32, 1).astype(np.float64)
    ((xb * torch.tensor(wx)).sum() + (dli * torch.tensor(wl2)).sum()).backward()
    out.update(glob_wx=wx, glob_wl2=wl2, **{f"glob_inv_g{i}": v.grad.numpy() for i, v in enumerate(ins)},
               **{f"glob_in{i}": v.detach().numpy() for i, v in enumerate(ins)})
    save("g_grads2", **out)


# ---------------------------------------------------------------------------------------------
# DistributionTransferFlow / ConstrainGaussianFlow (nn/flow/cdf.py:49-121), f64
# ---------------------------------------------------------------------------------------------
def g_cdf_flows():
    """inputs: x = synth(501, 64, 6, scale=1.5) + 0.8; mu = synth(502, 6) * 0.3 + 1, sigma = |synth(503, 6)| * 0.4 + 0.6"""
    from bgflow.nn.flow.cdf import DistributionTransferFlow, ConstrainGaussianFlow
    out = {}
    x = torch.tensor(synth(501, 64, 6, scale=1.5).astype(np.float64) + 0.8)
    mu = torch.tensor(synth(502, 6).astype(np.float64) * 0.3 + 1.0)
    sigma = torch.tensor(np.abs(synth(503, 6).astype(np.float64)) * 0.4 + 0.6)
    flow = ConstrainGaussianFlow(mu=mu, sigma=sigma, lower_bound=0.1, upper_bound=3.0)
    y, dl = flow.forward(x)
    xb, dlb = flow.forward(y, inverse=True)
    out.update(cg_y=y.numpy(), cg_dlogp=dl.numpy(), cg_back=xb.numpy(), cg_back_dlogp=dlb.numpy())
    flow = ConstrainGaussianFlow(mu=mu, sigma=sigma, lower_bound=0.0, mu_out=mu + 0.25, sigma_out=0.5 * sigma)
    y, dl = flow.forward(x)
    out.update(cg2_y=y.numpy(), cg2_dlogp=dl.numpy())
    src = torch.distributions.Normal(mu, sigma)
    dst = torch.distributions.Normal(torch.zeros(6, dtype=torch.float64), 2.0 * torch.ones(6, dtype=torch.float64))
    flow = DistributionTransferFlow(src, dst)
    y, dl = flow.forward(x)
    xb, dlb = flow.forward(y, inverse=True)
    out.update(dt_y=y.numpy(), dt_dlogp=dl.numpy(), dt_back=xb.numpy(), dt_back_dlogp=dlb.numpy())
    save("g_cdf_flows", **out)


def g_energies():
    """f-3: energies of the priors / targets around the flow and the importance-weight bookkeeping, from the reference classes:
    DoubleWellEnergy (distribution/energy/double_well.py:10-22) incl. its force, NormalDistribution (normal.py:17-92),
    UniformDistribution (distributions.py:100-117), ProductDistribution / ProductEnergy (product.py:13-117),
    log_weights_given_latent / effective_sample_size (bg.py:54-74)."""
    from bgflow.distribution.energy.double_well import DoubleWellEnergy
    from bgflow.distribution.normal import NormalDistribution
    from bgflow.distribution.distributions import UniformDistribution
    from bgflow.distribution.product import ProductDistribution
    from bgflow.bg import log_weights_given_latent, effective_sample_size
    out = {}
    B = 96
    x64 = torch.tensor(rng_f32(901, B, 64, scale=1.5)).double()
    for tag, kw in (("dw", {}), ("dw_abc", dict(a=0.7, b=-2.5, c=0.4))):
        for dt, suffix in ((torch.float64, "64"), (torch.float32, "32")):
            e = DoubleWellEnergy(64, **kw)
            x = x64.to(dt)
            out[f"{tag}_u{suffix}"] = e.energy(x).detach().numpy()
            out[f"{tag}_uT{suffix}"] = e.energy(x, temperature=2.5).detach().numpy()
            out[f"{tag}_force{suffix}"] = e.force(x.clone(), temperature=2.5).detach().numpy()
    out["dw_x"] = x64.float().numpy()
    mean = torch.tensor(rng_f32(902, 66, scale=0.3)).double()
    y64 = torch.tensor(rng_f32(903, B, 66)).double()
    a64 = torch.tensor(rng_f32(904, B, 66)).double()
    un64 = torch.tensor(rng_f32(905, B, 17, uniform=True)).double()
    low, high = torch.zeros(17).double(), torch.tensor(np.linspace(1.0, 3.0, 17).astype(np.float32)).double()   # f32-representable
    un64 = (un64 * high).float().double()                      # the stored inputs are f32: use exactly those values in both precisions
    for dt, suffix in ((torch.float64, "64"), (torch.float32, "32")):
        comps = [NormalDistribution(66, mean=mean.to(dt)), NormalDistribution(66), UniformDistribution(low.to(dt), high.to(dt))]
        prod = ProductDistribution(comps)
        xs = (y64.to(dt), a64.to(dt), un64.to(dt))
        out[f"norm_u{suffix}"] = comps[0].energy(xs[0], temperature=1.7).numpy()
        out[f"unif_u{suffix}"] = comps[2].energy(xs[2]).numpy()
        out[f"prod_u{suffix}"] = prod.energy(*xs).numpy()
        out[f"prod_uT{suffix}"] = prod.energy(*xs, temperature=1.7).numpy()
        # importance weights of a toy generator: latent z ~ N(0, 1), x = z (identity flow, dlogp = 0) against the shifted normal
        prior, target = NormalDistribution(66), NormalDistribution(66, mean=mean.to(dt))
        dl = torch.tensor(rng_f32(906, B, 1, scale=0.2)).to(dt)
        lw = log_weights_given_latent(xs[0], xs[1], dl, prior, target, temperature=1.3, normalize=True)
        out[f"logw{suffix}"] = lw.numpy()
        out[f"ess{suffix}"] = effective_sample_size(lw).numpy()
    out.update(norm_mean=mean.float().numpy(), prod_y=y64.float().numpy(), prod_a=a64.float().numpy(),
               prod_un=un64.float().numpy(), unif_low=low.float().numpy(), unif_high=high.float().numpy(), logw_dl=dl.float().numpy())
    save("g_energies", **out)


# ---------------------------------------------------------------------------------------------
# G-envelope: coupling layers whose conditioners sit at the wide / deep end of the one-launch kernels' envelope (round 5):
# the reference's CouplingFlow around ConditionalSplineTransformer / AffineTransformer with DenseNets of other widths and depths
# ---------------------------------------------------------------------------------------------
ENVELOPE_SPLINE = {"w256": (256, 256), "w200_130": (200, 130), "deep1": (128,), "deep3": (128, 128, 128), "deep4": (64, 128, 32, 100)}
ENVELOPE_AFFINE = {"readme4": (4,), "deep5": (48,) * 5, "deep4mixed": (128, 64, 32, 100)}


def envelope_spline_layer(lib, hidden, periodic, circular, d_c=9, d=7, n_bins=8):
    """``lib`` = bgflow (the reference) or bgflow_amd: the same constructor calls build the same module tree, hash_init_ gives it the
    same weights"""
    P = 3 * n_bins * d + (0 if circular else d)
    net = lib.DenseNet([2 * d_c if periodic else d_c, *hidden, P], activation=torch.nn.SiLU())
    if periodic:
        net = lib.WrapPeriodic(net)
    return hash_init_(lib.CouplingFlow(lib.ConditionalSplineTransformer(net, is_circular=circular), transformed_indices=(1,),
                                       cond_indices=(0,)))


def envelope_affine_layer(lib, hidden, d_c=12, d=20):
    return hash_init_(lib.CouplingFlow(lib.AffineTransformer(lib.DenseNet([d_c, *hidden, d], activation=torch.nn.ReLU()),
                                                             lib.DenseNet([d_c, *hidden, d], activation=torch.nn.Tanh())),
                                       transformed_indices=(1,), cond_indices=(0,)))


def g_envelope():
    out = {}
    B = 97
    for tag, hidden in ENVELOPE_SPLINE.items():
        for kind, (periodic, circular) in (("pc", (True, True)), ("nn", (False, False))):
            layer = envelope_spline_layer(bg, hidden, periodic, circular)
            c = rng_f32(61, B, 9, uniform=periodic)
            y = rng_f32(62, B, 7, uniform=True)
            for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
                f = layer.to(dt)
                with torch.no_grad():
                    _, z, dl = f(torch.tensor(c, dtype=dt), torch.tensor(y, dtype=dt))
                    _, yb, dli = f(torch.tensor(c, dtype=dt), z, inverse=True)
                out[f"s_{tag}_{kind}_z{sfx}"] = z.numpy(); out[f"s_{tag}_{kind}_dlogp{sfx}"] = dl.numpy()
                out[f"s_{tag}_{kind}_back{sfx}"] = yb.numpy(); out[f"s_{tag}_{kind}_dlogp_inv{sfx}"] = dli.numpy()
    for tag, hidden in ENVELOPE_AFFINE.items():
        layer = envelope_affine_layer(bg, hidden)
        c = rng_f32(63, B, 12)
        y = rng_f32(64, B, 20)
        for dt, sfx in ((torch.float32, "32"), (torch.float64, "64")):
            f = layer.to(dt)
            with torch.no_grad():
                _, z, dl = f(torch.tensor(c, dtype=dt), torch.tensor(y, dtype=dt))
                _, yb, dli = f(torch.tensor(c, dtype=dt), z, inverse=True)
            out[f"a_{tag}_z{sfx}"] = z.numpy(); out[f"a_{tag}_dlogp{sfx}"] = dl.numpy()
            out[f"a_{tag}_back{sfx}"] = yb.numpy(); out[f"a_{tag}_dlogp_inv{sfx}"] = dli.numpy()
    save("g_envelope", **out)      # inputs: synth(61..64)


if __name__ == "__main__":
    which = sys.argv[1:] or ["rqs", "bins", "affine", "ic", "flow16", "aug", "augment", "grads", "grads2", "cdfflows", "energies", "envelope"]
    if "envelope" in which:
        g_envelope()
    if "energies" in which:
        g_energies()
    if "rqs" in which:
        g_rqs_unit()
    if "bins" in which:
        g_rqs_bins()
    if "affine" in which:
        g_affine()
    if "ic" in which:
        g_ic()
    if "flow16" in which:
        g_flow16()
    if "aug" in which:
        g_aug()
    if "augment" in which:
        g_augment()
    if "grads" in which:
        g_grads()
    if "grads2" in which:
        g_grads2()
    if "cdfflows" in which:
        g_cdf_flows()
