"""ctypes/numpy front-end of the CPU oracle (oracle/libbgo_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module -- as the checker, never as the thing measured or shipped.  The product package
``bgflow_amd`` does not import it.

Parity status: pinned against golden vectors generated from the reference by
tests/golden/make_goldens.py (see tests/test_oracle_golden.py).

Reference citations are relative to /root/reference/bgflow/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbgo_oracle.so")
_lib = None

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_dbl = ctypes.c_double
_vp = ctypes.c_void_p


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.bgo_rqs_f32.restype = _c_i64
        _lib.bgo_rqs_f64.restype = _c_i64
        _lib.bgo_num_threads.restype = _c_int
    return _lib


def num_threads():
    return int(lib().bgo_num_threads())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _np(a, dtype):
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", ctypes.c_float
    if dtype == np.float64:
        return "_f64", ctypes.c_double
    raise TypeError(dtype)


def nc_slots(is_circular, d):
    """nc_slot[j] = position of dim j's extra slope in the s_nc block, -1 if circular
    (nn/flow/transformer/spline.py:190-204, evident intent for mixed masks)."""
    circ = np.broadcast_to(np.asarray(is_circular, dtype=bool), (d,))
    slots = np.full(d, -1, dtype=np.int32)
    slots[~circ] = np.arange(int((~circ).sum()), dtype=np.int32)
    return slots


def rqs(y, params, is_circular=False, inverse=False, left=0.0, right=1.0, bottom=0.0, top=1.0,
        min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, identity_init=True,
        n_bins=None, dtype=np.float32, want_details=False):
    """ConditionalSplineTransformer._forward/_inverse (transformer/spline.py:128-188) given the
    conditioner output ``params`` [B, P].  Returns (out [B,d], dlogp [B,1]) and, with
    ``want_details``, a dict with per-element log-dets, bin indices, search knots and the clamp count."""
    sfx, _ = _suffix(dtype)
    y = _np(y, dtype)
    params = _np(params, dtype)
    B, d = y.shape
    slots = nc_slots(is_circular, d)
    n_nc = int((slots >= 0).sum())
    P = params.shape[1]
    K = (P - n_nc) // (3 * d) if n_bins is None else n_bins
    if K > 64:
        raise ValueError("the C oracle holds its knot arrays on the stack (BGO_MAX_BINS = 64)")
    assert 3 * K * d + n_nc == P, f"params width {P} does not match d={d}, K={K}, n_nc={n_nc}"
    out = np.empty((B, d), dtype)
    dlogp = np.empty((B,), dtype)
    elem = np.empty((B, d), dtype) if want_details else None
    idx = np.empty((B, d), np.int32) if want_details else None
    knots = np.empty((B, d, K + 1), dtype) if want_details else None
    fn = getattr(lib(), "bgo_rqs" + sfx)
    n_oob = fn(_ptr(y), _c_i64(d), _ptr(params), _c_i64(P), _ptr(slots), _c_i64(B), _c_int(d), _c_int(K),
               _c_int(int(inverse)), _c_dbl(left), _c_dbl(right), _c_dbl(bottom), _c_dbl(top),
               _c_dbl(min_bin_width), _c_dbl(min_bin_height), _c_dbl(min_derivative),
               _c_int(int(identity_init)), _ptr(out), _c_i64(d), _ptr(dlogp), _ptr(elem), _ptr(idx),
               _ptr(knots))
    if want_details:
        return out, dlogp[:, None], dict(dlogp_elem=elem, bin_idx=idx, knots=knots, n_oob=int(n_oob))
    return out, dlogp[:, None]


def rqs_backward(y, params, g_out, g_dlogp, is_circular=False, inverse=False, left=0.0, right=1.0, bottom=0.0,
                 top=1.0, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, identity_init=True,
                 n_bins=None, dtype=np.float32):
    """Analytic VJP of ``rqs``: returns (g_y [B,d], g_params [B,P]) for upstream g_out [B,d], g_dlogp [B,1]."""
    sfx, _ = _suffix(dtype)
    y, params, g_out = _np(y, dtype), _np(params, dtype), _np(g_out, dtype)
    g_dlogp = _np(np.asarray(g_dlogp).reshape(-1), dtype)
    B, d = y.shape
    slots = nc_slots(is_circular, d)
    n_nc = int((slots >= 0).sum())
    P = params.shape[1]
    K = (P - n_nc) // (3 * d) if n_bins is None else n_bins
    if K > 64:
        raise ValueError("the C oracle holds its knot arrays on the stack (BGO_MAX_BINS = 64)")
    g_y = np.empty((B, d), dtype)
    g_p = np.empty((B, P), dtype)
    getattr(lib(), "bgo_rqs_backward" + sfx)(
        _ptr(y), _c_i64(d), _ptr(params), _c_i64(P), _ptr(slots), _c_i64(B), _c_int(d), _c_int(K), _c_int(int(inverse)),
        _c_dbl(left), _c_dbl(right), _c_dbl(bottom), _c_dbl(top), _c_dbl(min_bin_width), _c_dbl(min_bin_height),
        _c_dbl(min_derivative), _c_int(int(identity_init)), _ptr(g_out), _c_i64(d), _ptr(g_dlogp),
        _ptr(g_y), _c_i64(d), _ptr(g_p), _c_i64(P))
    return g_y, g_p


def affine(y, mu=None, s_raw=None, log_alpha=-1.0, preserve_volume=False, is_circular=False,
           inverse=False, dtype=np.float32):
    """AffineTransformer._forward/_inverse (transformer/affine.py:35-70) given the shift-net output
    ``mu`` and the scale-net output ``s_raw`` (before tanh)."""
    sfx, cr = _suffix(dtype)
    y = _np(y, dtype)
    B, d = y.shape
    mu = None if mu is None else _np(mu, dtype)
    s_raw = None if s_raw is None else _np(s_raw, dtype)
    out = np.empty((B, d), dtype)
    dlogp = np.empty((B,), dtype)
    fn = getattr(lib(), "bgo_affine" + sfx)
    fn(_ptr(y), _c_i64(d), _ptr(mu), _c_i64(d), _ptr(s_raw), _c_i64(d), cr(log_alpha),
       _c_int(int(preserve_volume)), _c_int(int(is_circular)), _c_int(int(inverse)), _c_i64(B), _c_int(d),
       _ptr(out), _c_i64(d), _ptr(dlogp))
    return out, dlogp[:, None]


def affine_backward(y, mu, s_raw, g_out, g_dlogp, log_alpha=-1.0, preserve_volume=False, is_circular=False,
                    inverse=False, dtype=np.float32):
    """Analytic VJP of ``affine``: returns (g_y, g_mu, g_s_raw, g_log_alpha)."""
    sfx, cr = _suffix(dtype)
    y, g_out = _np(y, dtype), _np(g_out, dtype)
    g_dlogp = _np(np.asarray(g_dlogp).reshape(-1), dtype)
    B, d = y.shape
    mu = None if mu is None else _np(mu, dtype)
    s_raw = None if s_raw is None else _np(s_raw, dtype)
    g_y = np.empty((B, d), dtype)
    g_mu = np.empty((B, d), dtype) if mu is not None else None
    g_s = np.empty((B, d), dtype) if s_raw is not None else None
    fn = getattr(lib(), "bgo_affine_backward" + sfx)
    fn.restype = ctypes.c_double
    g_la = fn(_ptr(y), _c_i64(d), _ptr(mu), _c_i64(d), _ptr(s_raw), _c_i64(d), cr(log_alpha),
              _c_int(int(preserve_volume)), _c_int(int(is_circular)), _c_int(int(inverse)), _c_i64(B), _c_int(d),
              _ptr(g_out), _c_i64(d), _ptr(g_dlogp), _ptr(g_y), _ptr(g_mu), _ptr(g_s))
    return g_y, g_mu, g_s, float(g_la)


_ACT = {None: 0, "none": 0, "silu": 1, "relu": 2, "tanh": 3}


def mfma_k_order(n):
    """Accumulation order of the fused HIP kernel for a layer whose INPUT lives in MFMA accumulator
    registers (32-feature blocks visited as 0,4,1,5,2,6,3,7,8,12,...); n must be a multiple of 32."""
    assert n % 32 == 0
    order = []
    for kb in range(n // 32):
        for r in range(16):
            k0 = 32 * kb + (r & 3) + 8 * (r >> 2)
            order += [k0, k0 + 4]
    return np.asarray(order, dtype=np.int32)


def linear(x, W, b=None, act=None, dtype=np.float32, k_order=None):
    """torch.nn.Linear (+activation): one fma chain per output (k ascending unless ``k_order``), bias
    added last (nn/dense.py:30-48)."""
    sfx, _ = _suffix(dtype)
    x = _np(x, dtype)
    W = _np(W, dtype)
    b = None if b is None else _np(b, dtype)
    B, n_in = x.shape
    n_out = W.shape[0]
    assert W.shape[1] == n_in
    out = np.empty((B, n_out), dtype)
    ko = None if k_order is None else _np(k_order, np.int32)
    fn = getattr(lib(), "bgo_linear" + sfx)
    fn(_ptr(x), _c_i64(n_in), _ptr(W), _ptr(b), _c_i64(B), _c_int(n_in), _c_int(n_out),
       _c_int(_ACT[act]), _ptr(ko), _ptr(out), _c_i64(n_out))
    return out


def dense_net(x, weights, biases, acts, dtype=np.float32, mfma_order=False):
    """DenseNet.forward (nn/dense.py:47-48): acts[i] follows layer i (None after the last).
    ``mfma_order``: hidden layers (every layer but the first) accumulate in the fused kernel's order."""
    h = x
    for i, (W, b, a) in enumerate(zip(weights, biases, acts)):
        ko = mfma_k_order(W.shape[1]) if (mfma_order and i > 0 and W.shape[1] % 32 == 0) else None
        h = linear(h, W, b, a, dtype, k_order=ko)
    return h


def wrap_periodic(x, dtype=np.float32):
    """WrapPeriodic featuriser with all inputs periodic on [0,1] (nn/periodic.py:30-37)."""
    sfx, _ = _suffix(dtype)
    x = _np(x, dtype)
    B, d = x.shape
    out = np.empty((B, 2 * d), dtype)
    getattr(lib(), "bgo_wrap_periodic" + sfx)(_ptr(x), _c_i64(d), _c_i64(B), _c_int(d), _ptr(out), _c_i64(2 * d))
    return out


def decompose_z_matrix(z_matrix, fixed):
    """Placement order of a relative Z-matrix (nn/flow/crd_transform/ic.py:25-91): repeatedly take
    every row whose atoms 2-4 are already placed.  Returns (blocks, index2atom, atom2index,
    index2order) like the reference plus the flat placement table [n,5] = (atom, p1, p2, p3, zrow)
    used by the C functions."""
    z = np.asarray(z_matrix, dtype=np.int64)
    fixed = np.asarray(fixed, dtype=np.int64)
    placed = set(int(a) for a in fixed)
    rows = [(i, r) for i, r in enumerate(z) if int(r[0]) not in placed]
    blocks, atoms, order, table = [], [fixed], [], []
    while rows:
        ready = [(i, r) for i, r in rows if all(int(a) in placed for a in r[1:])]
        if not ready:
            raise ValueError("Z-matrix decomposition failed: atoms not reachable from the fixed atoms: "
                             f"{[int(r[0]) for _, r in rows]}")
        blocks.append(np.stack([r for _, r in ready]))
        atoms.append(np.array([r[0] for _, r in ready]))
        order.append(np.array([i for i, _ in ready]))
        for i, r in ready:
            table.append([int(r[0]), int(r[1]), int(r[2]), int(r[3]), i])
        placed.update(int(r[0]) for _, r in ready)
        ready_ids = set(i for i, _ in ready)
        rows = [(i, r) for i, r in rows if i not in ready_ids]
    index2atom = np.concatenate(atoms)
    atom2index = np.argsort(index2atom)
    index2order = np.concatenate(order) if order else np.zeros(0, dtype=np.int64)
    return blocks, index2atom, atom2index, index2order, np.asarray(table, dtype=np.int32).reshape(-1, 5)


def ic_xyz2ic(x, z_matrix, fixed, normalize_angles=True, eps=1e-7, enforce_boundaries=True,
              whiten=None, dtype=np.float32):
    """Relative (whiten=None) or Mixed (whiten=(mean, Twhiten, jac_xz)) xyz -> IC
    (crd_transform/ic.py:386-433, 838-860; pca.py:74-83)."""
    sfx, cr = _suffix(dtype)
    x = _np(x, dtype)
    B = x.shape[0]
    zmat = _np(z_matrix, np.int32)
    fixed = _np(fixed, np.int32)
    n, nf = zmat.shape[0], fixed.shape[0]
    bonds = np.empty((B, n), dtype)
    angles = np.empty((B, n), dtype)
    torsions = np.empty((B, n), dtype)
    dlogp = np.empty((B,), dtype)
    if whiten is None:
        mean = Tw = None
        keep, jac = 3 * nf, 0.0
    else:
        mean, Tw, jac = _np(whiten[0], dtype), _np(whiten[1], dtype), float(whiten[2])
        keep = Tw.shape[1]
    xfix = np.empty((B, keep), dtype)
    getattr(lib(), "bgo_ic_xyz2ic" + sfx)(
        _ptr(x), _c_i64(x.shape[1]), _ptr(zmat), _c_int(n), _ptr(fixed), _c_int(nf),
        _c_int(int(normalize_angles)), cr(eps), _c_int(int(enforce_boundaries)),
        _ptr(mean), _ptr(Tw), _c_int(keep), cr(jac), _c_i64(B),
        _ptr(bonds), _ptr(angles), _ptr(torsions), _ptr(xfix), _ptr(dlogp))
    return bonds, angles, torsions, xfix, dlogp[:, None]


def ic_ic2xyz(bonds, angles, torsions, xfix, z_matrix, fixed, normalize_angles=True, eps=1e-7,
              enforce_boundaries=True, blacken=None, dtype=np.float32):
    """Relative (blacken=None) or Mixed (blacken=(mean, Tblacken, jac_xz)) IC -> xyz
    (crd_transform/ic.py:435-513, 862-884; pca.py:85-93)."""
    sfx, cr = _suffix(dtype)
    bonds, angles, torsions, xfix = (_np(t, dtype) for t in (bonds, angles, torsions, xfix))
    B, n = bonds.shape
    fixed = _np(fixed, np.int32)
    nf = fixed.shape[0]
    place = decompose_z_matrix(z_matrix, fixed)[4]
    n_atoms = n + nf
    x = np.empty((B, 3 * n_atoms), dtype)
    dlogp = np.empty((B,), dtype)
    if blacken is None:
        mean = Tb = None
        keep, jac = 3 * nf, 0.0
    else:
        mean, Tb, jac = _np(blacken[0], dtype), _np(blacken[1], dtype), float(blacken[2])
        keep = Tb.shape[0]
    getattr(lib(), "bgo_ic_ic2xyz" + sfx)(
        _ptr(bonds), _ptr(angles), _ptr(torsions), _ptr(xfix), _ptr(place), _c_int(n), _ptr(fixed),
        _c_int(nf), _c_int(int(normalize_angles)), cr(eps), _c_int(int(enforce_boundaries)),
        _ptr(mean), _ptr(Tb), _c_int(keep), cr(jac), _c_i64(B), _ptr(x), _c_i64(3 * n_atoms), _ptr(dlogp))
    return x, dlogp[:, None]


def ic_ic2xyz_backward(bonds, angles, torsions, x, g_x, g_dlogp, z_matrix, fixed, normalize_angles=True,
                       blacken=None, dtype=np.float32):
    """Analytic VJP of ``ic_ic2xyz`` (x = its forward output): returns (g_bonds, g_angles, g_torsions, g_xfix).  Closed-form
    sweep: valid away from the eps clamps of the placements only (see bgo_impl.h); degenerate geometries are checked against
    autograd of ``torch_flow.ic2xyz_torch``."""
    sfx, _ = _suffix(dtype)
    bonds, angles, torsions, x, g_x = (_np(t, dtype) for t in (bonds, angles, torsions, x, g_x))
    g_dlogp = _np(np.asarray(g_dlogp).reshape(-1), dtype)
    B, n = bonds.shape
    fixed = _np(fixed, np.int32)
    nf = fixed.shape[0]
    place = decompose_z_matrix(z_matrix, fixed)[4]
    Tb = None if blacken is None else _np(blacken[1], dtype)
    keep = 3 * nf if Tb is None else Tb.shape[0]
    gb, ga, gt = (np.empty((B, n), dtype) for _ in range(3))
    gf = np.empty((B, keep), dtype)
    getattr(lib(), "bgo_ic_ic2xyz_backward" + sfx)(
        _ptr(bonds), _ptr(angles), _ptr(torsions), _ptr(x), _c_i64(x.shape[1]), _ptr(place), _c_int(n), _ptr(fixed),
        _c_int(nf), _c_int(int(normalize_angles)), _ptr(Tb), _c_int(keep), _c_i64(B), _ptr(g_x), _c_i64(g_x.shape[1]),
        _ptr(g_dlogp), _ptr(gb), _ptr(ga), _ptr(gt), _ptr(gf))
    return gb, ga, gt, gf


def refsys(v, inverse=False, normalize_angles=True, eps=1e-7, enforce_boundaries=True, dtype=np.float32):
    """ReferenceSystemTransformation (crd_transform/ic.py:128-265) on packed 9-vectors:
    forward (x0,x1,x2) -> (x0, d01, d12, a012, alpha, beta, gamma); inverse the reverse.  Returns (out [B,9], dlogp [B,1])."""
    sfx, cr = _suffix(dtype)
    v = _np(v, dtype)
    B = v.shape[0]
    out = np.empty((B, 9), dtype)
    dl = np.empty((B,), dtype)
    getattr(lib(), "bgo_refsys" + sfx)(_ptr(v), _c_i64(B), _c_int(int(inverse)), _c_int(int(normalize_angles)), cr(eps),
                                       _c_int(int(enforce_boundaries)), _ptr(out), _ptr(dl))
    return out, dl[:, None]


def slice_initial_atoms(z_matrix):
    """first three atoms and the remaining rows of a global Z-matrix (crd_transform/ic.py:94-97)"""
    z = np.asarray(z_matrix)
    s = np.sum(z == -1, axis=-1)
    order = np.argsort(s)[::-1][:3]
    return z[:, 0][order], z[s == 0]


def global_ic_forward(x, z_matrix, normalize_angles=True, eps=1e-7, enforce_boundaries=True, dtype=np.float32):
    """GlobalInternalCoordinateTransformation._forward (ic.py:632-672)"""
    init, zrel = slice_initial_atoms(z_matrix)
    b, a, t, xf, dl_rel = ic_xyz2ic(x, zrel, init, normalize_angles, eps, enforce_boundaries, dtype=dtype)
    ref, dl_ref = refsys(xf, False, normalize_angles, eps, enforce_boundaries, dtype)
    bonds = np.concatenate([ref[:, 3:5], b], axis=1)
    angles = np.concatenate([ref[:, 5:6], a], axis=1)
    return bonds, angles, t, ref[:, None, 0:3], ref[:, 6:9], dl_rel + dl_ref


def global_ic_inverse(bonds, angles, torsions, x0, R, z_matrix, normalize_angles=True, eps=1e-7,
                      enforce_boundaries=True, dtype=np.float32):
    """GlobalInternalCoordinateTransformation._inverse (ic.py:674-716)"""
    init, zrel = slice_initial_atoms(z_matrix)
    bonds, angles = _np(bonds, dtype), _np(angles, dtype)
    v = np.concatenate([_np(x0, dtype).reshape(-1, 3), bonds[:, 0:2], angles[:, 0:1], _np(R, dtype)], axis=1)
    xinit, dl_ref = refsys(v, True, normalize_angles, eps, enforce_boundaries, dtype)
    x, dl_rel = ic_ic2xyz(bonds[:, 2:], angles[:, 1:], torsions, xinit, zrel, init, normalize_angles, eps,
                          enforce_boundaries, dtype=dtype)
    return x, dl_rel + dl_ref


def detmath_probe(x, which):
    names = {"exp": 0, "log": 1, "softplus": 2, "silu": 3, "tanh": 4}
    x = _np(x, np.float32)
    out = np.empty_like(x)
    lib().bgo_detmath_probe(_ptr(x), _c_i64(x.size), _c_int(names[which]), _ptr(out))
    return out
