"""Internal-coordinate (Z-matrix <-> Cartesian) transforms on the HIP kernels bgk_ic_xyz2ic /
bgk_ic_ic2xyz, API-compatible with bgflow/nn/flow/crd_transform/{ic,pca}.py.

``forward`` maps xyz -> (bonds, angles, torsions, fixed) and ``inverse`` back, each with
log|det J| (ic.py:386-513).  ``MixedCoordinateTransformation`` adds PCA whitening of the fixed
atoms (ic.py:719-884, pca.py:37-107); in the kernels the whitening matvec is fused into the same
launch.  Z-matrix decomposition and the PCA are construction-time host code (numpy).
"""
import numpy as np
import torch

from . import _lib
from .flow import ACC_KW, Flow

__all__ = [
    "decompose_z_matrix", "RelativeInternalCoordinateTransformation", "MixedCoordinateTransformation",
    "WhitenFlow", "ReferenceSystemTransformation", "GlobalInternalCoordinateTransformation",
    "slice_initial_atoms", "normalize_torsions", "normalize_angles", "unnormalize_torsions", "unnormalize_angles",
]


# ---- small public helpers of the reference's ic module (crd_transform/ic.py:94-125); the kernels apply the same maps internally
def slice_initial_atoms(z_matrix):
    """(the three atoms of a global Z-matrix with the most ``-1`` placeholders -- origin, axis atom, plane atom --, the complete rows)"""
    n_open = np.sum(z_matrix == -1, axis=-1)
    first_three = np.argsort(n_open)[::-1][:3]
    return z_matrix[:, 0][first_three], z_matrix[n_open == 0]


def normalize_torsions(torsions):
    """[-pi, pi) -> [0, 1): (t + pi) / (2 pi); log-det = -d log(2 pi)"""
    return (torsions + np.pi) / (2 * np.pi), -np.log(2 * np.pi) * torsions.shape[-1]


def normalize_angles(angles):
    """[0, pi] -> [0, 1]: a / pi; log-det = -d log(pi)"""
    return angles / np.pi, -np.log(np.pi) * angles.shape[-1]


def unnormalize_torsions(torsions):
    """[0, 1) -> [-pi, pi): 2 pi t - pi; log-det = d log(2 pi)"""
    return torsions * (2 * np.pi) - np.pi, np.log(2 * np.pi) * torsions.shape[-1]


def unnormalize_angles(angles):
    """[0, 1] -> [0, pi]: pi a; log-det = d log(pi)"""
    return angles * np.pi, np.log(np.pi) * angles.shape[-1]


def decompose_z_matrix(z_matrix, fixed):
    """Group the rows of a relative Z-matrix into placement stages: a row (a, b, c, d) can be
    placed once b, c, d are known (fixed or placed earlier).  Returns ``(blocks, index2atom,
    atom2index, index2order)`` with the meaning of bgflow's function of the same name
    (crd_transform/ic.py:25-91); raises ValueError if some atom is unreachable."""
    z = np.asarray(z_matrix)
    fixed = np.asarray(fixed)
    known = np.zeros(int(max(z.max(), fixed.max())) + 1, dtype=bool)
    known[fixed] = True
    pending = [i for i in range(len(z)) if not known[z[i, 0]]]
    row_of = {i: k for k, i in enumerate(pending)}   # position among the non-fixed rows
    blocks, atoms, order = [], [fixed], []
    while pending:
        ready = [i for i in pending if known[z[i, 1:]].all()]
        if not ready:
            raise ValueError(
                "Z-matrix decomposition failed. The following atoms were not reachable from the fixed atoms: \n"
                f"{z[pending, 0]}")
        blocks.append(z[ready])
        atoms.append(z[ready, 0])
        order.append(np.array([row_of[i] for i in ready]))
        known[z[ready, 0]] = True
        ready_set = set(ready)
        pending = [i for i in pending if i not in ready_set]
    index2atom = np.concatenate(atoms)
    atom2index = np.argsort(index2atom)
    index2order = np.concatenate(order) if order else np.zeros(0, dtype=np.int64)
    return blocks, index2atom, atom2index, index2order


def _placement_table(z_matrix, fixed):
    """[n,5] int32 rows (atom, p1, p2, p3, zrow) in placement order for bgk_ic_ic2xyz."""
    blocks, _, _, index2order = decompose_z_matrix(z_matrix, fixed)
    rows = np.concatenate(blocks) if blocks else np.zeros((0, 4), dtype=np.int64)
    return np.concatenate([rows, index2order[:, None]], axis=1).astype(np.int32)


class _DeviceTables:
    """Small int32 / f32 constant tables, uploaded once per device."""

    def __init__(self):
        self._host = {}
        self._dev = {}

    def set(self, name, array):
        self._host[name] = array
        self._dev = {k: v for k, v in self._dev.items() if k[0] != name}

    def get(self, name, device):
        key = (name, str(device))
        if key not in self._dev:
            self._dev[key] = torch.as_tensor(self._host[name]).to(device)
        return self._dev[key]



def _dl_target(acc, B, dev):
    """([B] log-det buffer, accumulate flag): the pass's running buffer (flow._LogDetAcc) or a fresh tensor"""
    if acc is None:
        return torch.empty((B,), dtype=torch.float32, device=dev), 0
    buf, started = acc.peek()
    assert buf.shape[0] == B and buf.device == dev, "running log-det buffer does not match the batch"
    return buf, int(started)


def _dl_merge(acc, a, b):
    """sum of two launches' log-dets, either of which may already sit in the running buffer"""
    if acc is None or (a is not acc and b is not acc):
        return a + b
    for t in (a, b):
        if t is not acc:
            acc.add(t)
    return acc


def _dl_result(acc, dlogp):
    if acc is None:
        return dlogp[:, None]
    acc.commit()
    return acc


def _contig_rows(*ts):
    """make IC tensors [B, n] row-major with a COMMON row stride (copy only when needed)."""
    outs = []
    ld0 = None
    for t in ts:
        t2, ld = _lib.rowmajor(t)
        if ld0 is None:
            ld0 = ld
        outs.append((t2, ld))
    if any(ld != ld0 for _, ld in outs):
        outs = [(t.contiguous(), t.shape[1]) for t, _ in outs]
        ld0 = outs[0][1]
    return [t for t, _ in outs], ld0


class _IC2XYZFn(torch.autograd.Function):
    """IC -> xyz with the analytic backward kernel bgk_ic_ic2xyz_backward."""

    @staticmethod
    def forward(ctx, rel, bonds, angles, torsions, xfix, blacken):
        x, dlogp = rel._ic2xyz_launch(bonds, angles, torsions, xfix, blacken)
        ctx.rel, ctx.blacken = rel, blacken
        ctx.save_for_backward(bonds, angles, torsions, x)
        return x, dlogp

    @staticmethod
    def backward(ctx, g_x, g_dlogp):
        bonds, angles, torsions, x = ctx.saved_tensors
        g_b, g_a, g_t, g_f = ctx.rel._ic2xyz_backward(bonds, angles, torsions, x, ctx.blacken, g_x, g_dlogp)
        return None, g_b, g_a, g_t, g_f, None


class _XYZ2ICFn(torch.autograd.Function):
    """xyz -> IC with the backward kernel bgk_ic_xyz2ic_backward (dual-number evaluation of the forward formulas)."""

    @staticmethod
    def forward(ctx, rel, x, whiten):
        outs = rel._xyz2ic_launch(x, whiten)
        ctx.rel, ctx.whiten = rel, whiten
        ctx.save_for_backward(x)
        return outs

    @staticmethod
    def backward(ctx, g_b, g_a, g_t, g_f, g_dlogp):
        (x,) = ctx.saved_tensors
        rel, whiten = ctx.rel, ctx.whiten
        dev = x.device
        x2, ldx = _lib.rowmajor(x.flatten(1))
        B, n, nf = x2.shape[0], rel._n, rel._n_fixed
        (gb2, ga2, gt2), ldgic = _contig_rows(g_b, g_a, g_t)
        T = None if whiten is None else whiten[1]
        keep = 3 * nf if T is None else T.shape[1]
        gf2, ldgf = _lib.rowmajor(g_f.flatten(1).contiguous())
        g_dl = g_dlogp.reshape(-1).contiguous()
        g_x = torch.empty((B, 3 * (n + nf)), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_ic_xyz2ic_backward(
                _lib.ptr(x2), ldx, _lib.ptr(rel._tables.get("zmat", dev)), n, _lib.ptr(rel._tables.get("fixed", dev)), nf,
                int(rel._normalize_angles), float(rel._eps), int(rel._enforce_boundaries), _lib.ptr(T), keep, B,
                _lib.ptr(gb2), _lib.ptr(ga2), _lib.ptr(gt2), ldgic, _lib.ptr(gf2), ldgf, _lib.ptr(g_dl),
                _lib.ptr(g_x), g_x.shape[1], _lib.stream_ptr(dev))
        _lib.check(st, "bgk_ic_xyz2ic_backward")
        return None, g_x.reshape(x.shape), None


class _RefSysFn(torch.autograd.Function):
    """global reference system (either direction) with the backward kernel bgk_ic_refsys_backward"""

    @staticmethod
    def forward(ctx, ref, packed, inverse):
        out, dlogp = ref._launch_nograd(packed, inverse)
        ctx.ref, ctx.inverse = ref, inverse
        ctx.save_for_backward(packed)
        return out, dlogp

    @staticmethod
    def backward(ctx, g_out, g_dlogp):
        (packed,) = ctx.saved_tensors
        ref = ctx.ref
        B = packed.shape[0]
        g_in = torch.empty_like(packed)
        with torch.cuda.device(packed.device):
            st = _lib.lib().bgk_ic_refsys_backward(
                _lib.ptr(packed), _lib.ptr(g_out.contiguous()), _lib.ptr(g_dlogp.reshape(-1).contiguous()), B, int(ctx.inverse),
                int(ref._normalize_angles), float(ref._eps), int(ref._enforce_boundaries), _lib.ptr(g_in),
                _lib.stream_ptr(packed.device))
        _lib.check(st, "bgk_ic_refsys_backward")
        return None, g_in, None


class RelativeInternalCoordinateTransformation(Flow):
    """Internal coordinates relative to a set of fixed atoms (crd_transform/ic.py:268-513).

    forward:  x [B, 3*n_atoms] -> bonds, angles, torsions [B, n] each, x_fixed [B, 3*n_fixed], dlogp
    inverse:  the reverse (NeRF-style sequential placement in Z-matrix dependency order).
    Angles / torsions are mapped to [0, 1] when ``normalize_angles``.  Near-singular geometry is
    clamped at ``eps`` exactly like the reference when ``enforce_boundaries``; instead of
    ``warnings.warn`` inside the hot path the kernels count clamp events in a device counter
    (``check_singularities()``)."""

    def __init__(self, z_matrix, fixed_atoms, normalize_angles=True, eps=1e-7, enforce_boundaries=True,
                 raise_warnings=True):
        super().__init__()
        self._z_matrix = z_matrix
        self._fixed_atoms = fixed_atoms
        z = np.asarray(z_matrix if not torch.is_tensor(z_matrix) else z_matrix.cpu().numpy())
        f = np.asarray(fixed_atoms if not torch.is_tensor(fixed_atoms) else fixed_atoms.cpu().numpy())
        (self._z_blocks, self._index2atom, self._atom2index, self._index2order) = decompose_z_matrix(z, f)
        self._bond_indices = z[:, :2]
        self._angle_indices = z[:, :3]
        self._torsion_indices = z[:, :4]
        self._normalize_angles = normalize_angles
        self._eps = eps
        self._enforce_boundaries = enforce_boundaries
        self._raise_warnings = raise_warnings
        self._tables = _DeviceTables()
        self._tables.set("zmat", np.ascontiguousarray(z[:, :4], dtype=np.int32))
        place = _placement_table(z, f)
        self._tables.set("place", place)
        # the register-resident tail kernel reads one 32-byte record per placement (bgk_tail.hip::Rec)
        self._tables.set("place8", np.ascontiguousarray(np.concatenate([place, np.zeros((len(place), 3), np.int32)], axis=1)))
        self._placement_zrows = place[:, 4].astype(np.int64)      # Z row consumed by placement i
        self._tables.set("zmat8", np.ascontiguousarray(np.concatenate([z[:, :4].astype(np.int32), np.zeros((len(z), 4), np.int32)], axis=1)))
        self._tables.set("fixed", np.ascontiguousarray(f, dtype=np.int32))
        self._n, self._n_fixed = len(z), len(f)
        self._warn = {}
        self._fix_ws = {}

    # reference properties (ic.py:315-353)
    z_matrix = property(lambda self: self._z_matrix)
    fixed_atoms = property(lambda self: self._fixed_atoms)
    dim_bonds = property(lambda self: len(self._z_matrix))
    dim_angles = property(lambda self: len(self._z_matrix))
    dim_torsions = property(lambda self: len(self._z_matrix))
    dim_fixed = property(lambda self: 3 * len(self._fixed_atoms))
    bond_indices = property(lambda self: self._bond_indices)
    angle_indices = property(lambda self: self._angle_indices)
    torsion_indices = property(lambda self: self._torsion_indices)
    normalize_angles = property(lambda self: self._normalize_angles)

    def _fixup_list(self, device, B):
        """bgk_ic_ic2xyz_backward's list of samples with a clamped norm (1 + B int32; the library resets its count per call).  One
        buffer per device, grown as needed: backward launches of this object are ordered on the device's current stream."""
        key = str(device)
        buf = self._fix_ws.get(key)
        if buf is None or buf.numel() < B + 1:
            buf = self._fix_ws[key] = torch.empty(B + 1, dtype=torch.int32, device=device)
        return buf

    def _warn_counter(self, device):
        if not self._raise_warnings:
            return None
        key = str(device)
        if key not in self._warn:
            self._warn[key] = torch.zeros(1, dtype=torch.int32, device=device)
        return self._warn[key]

    def check_singularities(self):
        """Poll (host sync) and reset the clamp-event counters; warn like the reference would have."""
        import warnings
        total = 0
        for c in self._warn.values():
            total += int(c.item())
            c.zero_()
        if total:
            warnings.warn(f"singular geometry: {total} norm / division clamps at eps={self._eps}")
        return total

    _bgk_acc = True

    def _xyz2ic(self, x, whiten=None, acc=None):
        _lib.require_hip(x)
        if torch.is_grad_enabled() and x.requires_grad:
            return _XYZ2ICFn.apply(self, x, whiten)
        return self._xyz2ic_launch(x, whiten, acc=acc)

    def _xyz2ic_launch(self, x, whiten=None, acc=None):
        dev = x.device
        x2, ldx = _lib.rowmajor(x.flatten(1))
        B, n, nf = x2.shape[0], self._n, self._n_fixed
        assert x2.shape[1] == 3 * (n + nf), "x must be [batch, 3 * n_atoms]"
        ics = torch.empty((3, B, n), dtype=torch.float32, device=dev)
        if whiten is None:
            mean = T = None
            keep, jac = 3 * nf, 0.0
        else:
            mean, T, jac = whiten
            keep = T.shape[1]
        xfix = torch.empty((B, keep), dtype=torch.float32, device=dev)
        dlogp, accumulate = _dl_target(acc, B, dev)
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_ic_xyz2ic(
                _lib.ptr(x2), ldx, _lib.ptr(self._tables.get("zmat", dev)), n,
                _lib.ptr(self._tables.get("fixed", dev)), nf, int(self._normalize_angles), float(self._eps),
                int(self._enforce_boundaries), _lib.ptr(mean), _lib.ptr(T), keep, float(jac), B,
                _lib.ptr(ics[0]), _lib.ptr(ics[1]), _lib.ptr(ics[2]), n, _lib.ptr(xfix), keep,
                _lib.ptr(dlogp), accumulate, _lib.ptr(self._warn_counter(dev)), _lib.stream_ptr(dev))
        _lib.check(st, "bgk_ic_xyz2ic")
        return ics[0], ics[1], ics[2], xfix, _dl_result(acc, dlogp)

    def _ic2xyz(self, bonds, angles, torsions, xfix, blacken=None, acc=None):
        _lib.require_hip(bonds, angles, torsions, xfix)
        if torch.is_grad_enabled() and any(t.requires_grad for t in (bonds, angles, torsions, xfix)):
            return _IC2XYZFn.apply(self, bonds, angles, torsions, xfix.flatten(1), blacken)
        return self._ic2xyz_launch(bonds, angles, torsions, xfix, blacken, acc=acc)

    def _ic2xyz_launch(self, bonds, angles, torsions, xfix, blacken=None, acc=None):
        dev = bonds.device
        B, n, nf = bonds.shape[0], self._n, self._n_fixed
        assert bonds.shape[-1] == n
        assert angles.shape[-1] == n
        assert torsions.shape[-1] == n
        (b2, a2, t2), ldic = _contig_rows(bonds, angles, torsions)
        f2, ldf = _lib.rowmajor(xfix.flatten(1))
        if blacken is None:
            mean = T = None
            keep, jac = 3 * nf, 0.0
        else:
            mean, T, jac = blacken
            keep = T.shape[0]
        assert f2.shape[1] == keep
        x = torch.empty((B, 3 * (n + nf)), dtype=torch.float32, device=dev)
        dlogp, accumulate = _dl_target(acc, B, dev)
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_ic_ic2xyz(
                _lib.ptr(b2), _lib.ptr(a2), _lib.ptr(t2), ldic, _lib.ptr(f2), ldf,
                _lib.ptr(self._tables.get("place", dev)), n, _lib.ptr(self._tables.get("fixed", dev)), nf,
                int(self._normalize_angles), float(self._eps), int(self._enforce_boundaries),
                _lib.ptr(mean), _lib.ptr(T), keep, float(jac), B, _lib.ptr(x), x.shape[1],
                _lib.ptr(dlogp), accumulate, _lib.ptr(self._warn_counter(dev)), _lib.stream_ptr(dev))
        _lib.check(st, "bgk_ic_ic2xyz")
        return x, _dl_result(acc, dlogp)

    UNIFORM_TAIL = True      # ... and on its elementwise variant when every field has one marginal for all its channels
    REGISTER_TAIL = True     # sampling tail on the register-resident kernel (bgk_icdf_ic2xyz_reg) where its envelope allows

    def _icdf_ic2xyz_train(self, bonds, angles, torsions, xfix, eps, blacken, desc20, kl=None):
        """the fused tail as a TRAINING forward (bgk_icdf_ic2xyz_uni_train): (x, dlogp [B], (y_bonds, y_angles, y_torsions, y_fixed)) with
        y = the mapped fields the backward kernels read, or None outside the elementwise kernel's envelope.  No autograd here
        (flow._FusedTailTrainFn wraps it).  ``kl`` = (t_mean or None, temperature, c_in, c_out, drop_nonfinite, dlogp_in [B] or None):
        the KL integrand of a normal target in the same launch (bgk_icdf_ic2xyz_uni_train_kl); then the result is
        (x, dlogp_total [B], ys, u [B], sums f64 [2])."""
        dev = bonds.device
        B, n, nf = bonds.shape[0], self._n, self._n_fixed
        desc4 = getattr(desc20, "uniform4", None) if desc20 is not None else None
        if desc4 is None or not (self.REGISTER_TAIL and self.UNIFORM_TAIL and self._normalize_angles and n + nf <= 32 and B > 0):
            return None
        (b2, a2, t2), ldic = _contig_rows(bonds.detach(), angles.detach(), torsions.detach())
        f2, ldf = _lib.rowmajor(xfix.detach().flatten(1))
        if blacken is None:
            mean = T = None
            keep, jac = 3 * nf, 0.0
        else:
            mean, T, jac = blacken
            keep = T.shape[0]
        if not (keep <= 16 and f2.shape[1] == keep and ldic == n and ldf == keep and all(v.is_contiguous() for v in (b2, a2, t2, f2))):
            return None
        x = torch.empty((B, 3 * (n + nf)), dtype=torch.float32, device=dev)
        dlogp = torch.empty((B,), dtype=torch.float32, device=dev)
        ys = torch.empty((3, B, n), dtype=torch.float32, device=dev)
        yf = torch.empty((B, keep), dtype=torch.float32, device=dev)
        const_ld = n * (np.log(np.pi) + np.log(2.0 * np.pi)) - (float(jac) if T is not None else 0.0)
        if kl is not None:
            t_mean, temperature, c_in, c_out, drop, dl_in = kl
            u = torch.empty((B,), dtype=torch.float32, device=dev)
            partial = torch.empty(((B + 63) // 64, 2), dtype=torch.float32, device=dev)
            sums = torch.empty(2, dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                st = _lib.lib().bgk_icdf_ic2xyz_uni_train_kl(
                    _lib.ptr(b2), _lib.ptr(a2), _lib.ptr(t2), _lib.ptr(f2), _lib.ptr(desc4), int(eps is not None), float(eps or 0.0),
                    _lib.ptr(self._tables.get("place8", dev)), n, _lib.ptr(self._tables.get("fixed", dev)), nf,
                    float(self._eps), int(self._enforce_boundaries), _lib.ptr(mean), _lib.ptr(T), keep, float(const_ld), B,
                    _lib.ptr(x), x.shape[1], _lib.ptr(dl_in), _lib.ptr(self._warn_counter(dev)),
                    _lib.ptr(ys[0]), _lib.ptr(ys[1]), _lib.ptr(ys[2]), _lib.ptr(yf), _lib.ptr(t_mean), float(temperature), float(c_in), float(c_out),
                    int(bool(drop)), _lib.ptr(u), _lib.ptr(dlogp), _lib.ptr(partial), _lib.ptr(sums), _lib.stream_ptr(dev))
            if st == -2:
                return None
            _lib.check(st, "bgk_icdf_ic2xyz_uni_train_kl")
            return x, dlogp, (ys[0], ys[1], ys[2], yf), u, sums
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_icdf_ic2xyz_uni_train(
                _lib.ptr(b2), _lib.ptr(a2), _lib.ptr(t2), _lib.ptr(f2), _lib.ptr(desc4), int(eps is not None), float(eps or 0.0),
                _lib.ptr(self._tables.get("place8", dev)), n, _lib.ptr(self._tables.get("fixed", dev)), nf,
                float(self._eps), int(self._enforce_boundaries), _lib.ptr(mean), _lib.ptr(T), keep, float(const_ld), B,
                _lib.ptr(x), x.shape[1], _lib.ptr(dlogp), 0, _lib.ptr(self._warn_counter(dev)),
                _lib.ptr(ys[0]), _lib.ptr(ys[1]), _lib.ptr(ys[2]), _lib.ptr(yf), _lib.stream_ptr(dev))
        if st == -2:
            return None
        _lib.check(st, "bgk_icdf_ic2xyz_uni_train")
        return x, dlogp, (ys[0], ys[1], ys[2], yf)

    def _ic2xyz_backward(self, y_bonds, y_angles, y_torsions, x, blacken, g_x, g_dlogp):
        """bgk_ic_ic2xyz_backward: (g_bonds, g_angles, g_torsions, g_fixed) of the IC -> xyz map at the saved point"""
        dev = x.device
        B, n, nf = y_bonds.shape[0], self._n, self._n_fixed
        (b2, a2, t2), ldic = _contig_rows(y_bonds, y_angles, y_torsions)
        g_x2, ldgx = _lib.rowmajor(g_x.contiguous())
        g_dl = g_dlogp.reshape(-1).contiguous()
        T = None if blacken is None else blacken[1]
        keep = 3 * nf if T is None else T.shape[0]
        g_ic = torch.empty((3, B, n), dtype=torch.float32, device=dev)
        g_f = torch.empty((B, keep), dtype=torch.float32, device=dev)
        fix = self._fixup_list(dev, B)
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_ic_ic2xyz_backward(
                _lib.ptr(b2), _lib.ptr(a2), _lib.ptr(t2), ldic, _lib.ptr(x), x.shape[1],
                _lib.ptr(self._tables.get("place", dev)), n, _lib.ptr(self._tables.get("fixed", dev)), nf,
                int(self._normalize_angles), float(self._eps), int(self._enforce_boundaries), _lib.ptr(T), keep, B, _lib.ptr(g_x2), ldgx, _lib.ptr(g_dl),
                _lib.ptr(g_ic[0]), _lib.ptr(g_ic[1]), _lib.ptr(g_ic[2]), n, _lib.ptr(g_f), keep, _lib.ptr(fix), _lib.stream_ptr(dev))
        _lib.check(st, "bgk_ic_ic2xyz_backward")
        return g_ic[0], g_ic[1], g_ic[2], g_f

    def _icdf_ic2xyz(self, bonds, angles, torsions, xfix, descs, eps, blacken=None, acc=None, desc20=None):
        """IC -> xyz with the icdf domain maps of the four inputs fused in (bgk_icdf_ic2xyz); ``descs`` = per-field [d, 6]
        descriptor tensors (cdf.CDFTransform.kernel_descriptor) or None for a field that is used as is.  No autograd."""
        _lib.require_hip(bonds, angles, torsions, xfix)
        dev = bonds.device
        B, n, nf = bonds.shape[0], self._n, self._n_fixed
        (b2, a2, t2), ldic = _contig_rows(bonds, angles, torsions)
        f2, ldf = _lib.rowmajor(xfix.flatten(1))
        if blacken is None:
            mean = T = None
            keep, jac = 3 * nf, 0.0
        else:
            mean, T, jac = blacken
            keep = T.shape[0]
        assert f2.shape[1] == keep
        x = torch.empty((B, 3 * (n + nf)), dtype=torch.float32, device=dev)
        dlogp, accumulate = _dl_target(acc, B, dev)
        if (desc20 is not None and self.REGISTER_TAIL and self._normalize_angles and n + nf <= 32 and keep <= 16 and B > 0
                and ldic == n and ldf == keep and all(v.is_contiguous() for v in (b2, a2, t2, f2))):
            # second-generation tail (csrc/bgk_tail.hip): positions in registers, contiguous field tiles
            const_ld = n * (np.log(np.pi) + np.log(2.0 * np.pi)) - (float(jac) if T is not None else 0.0)
            desc4 = getattr(desc20, "uniform4", None)
            st = -2
            if desc4 is not None and self.UNIFORM_TAIL:
                with torch.cuda.device(dev):
                    st = _lib.lib().bgk_icdf_ic2xyz_uni(
                        _lib.ptr(b2), _lib.ptr(a2), _lib.ptr(t2), _lib.ptr(f2), _lib.ptr(desc4), int(eps is not None), float(eps or 0.0),
                        _lib.ptr(self._tables.get("place8", dev)), n, _lib.ptr(self._tables.get("fixed", dev)), nf,
                        float(self._eps), int(self._enforce_boundaries), _lib.ptr(mean), _lib.ptr(T), keep, float(const_ld), B,
                        _lib.ptr(x), x.shape[1], _lib.ptr(dlogp), accumulate, _lib.ptr(self._warn_counter(dev)), _lib.stream_ptr(dev))
                if st != -2:
                    _lib.check(st, "bgk_icdf_ic2xyz_uni")
                    return x, _dl_result(acc, dlogp)
            with torch.cuda.device(dev):
                st = _lib.lib().bgk_icdf_ic2xyz_reg(
                    _lib.ptr(b2), _lib.ptr(a2), _lib.ptr(t2), _lib.ptr(f2), _lib.ptr(desc20), int(eps is not None), float(eps or 0.0),
                    _lib.ptr(self._tables.get("place8", dev)), n, _lib.ptr(self._tables.get("fixed", dev)), nf,
                    float(self._eps), int(self._enforce_boundaries), _lib.ptr(mean), _lib.ptr(T), keep, float(const_ld), B,
                    _lib.ptr(x), x.shape[1], _lib.ptr(dlogp), accumulate, _lib.ptr(self._warn_counter(dev)), _lib.stream_ptr(dev))
            if st != -2:
                _lib.check(st, "bgk_icdf_ic2xyz_reg")
                return x, _dl_result(acc, dlogp)
        db, da, dt, df = descs
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_icdf_ic2xyz(
                _lib.ptr(b2), _lib.ptr(a2), _lib.ptr(t2), ldic, _lib.ptr(f2), ldf,
                _lib.ptr(db), _lib.ptr(da), _lib.ptr(dt), _lib.ptr(df), int(eps is not None), float(eps or 0.0),
                _lib.ptr(self._tables.get("place", dev)), n, _lib.ptr(self._tables.get("fixed", dev)), nf,
                int(self._normalize_angles), float(self._eps), int(self._enforce_boundaries),
                _lib.ptr(mean), _lib.ptr(T), keep, float(jac), B, _lib.ptr(x), x.shape[1],
                _lib.ptr(dlogp), accumulate, _lib.ptr(self._warn_counter(dev)), _lib.stream_ptr(dev))
        _lib.check(st, "bgk_icdf_ic2xyz")
        return x, _dl_result(acc, dlogp)

    def _generate_fused(self, bonds, angles, torsions, x_fixed, descs, eps, acc=None, desc20=None):
        return self._icdf_ic2xyz(bonds, angles, torsions, x_fixed, descs, eps, acc=acc, desc20=desc20)

    def _generate_fused_train(self, bonds, angles, torsions, x_fixed, eps, desc20, kl=None):
        res = self._icdf_ic2xyz_train(bonds, angles, torsions, x_fixed, eps, None, desc20, kl=kl)
        return None if res is None else (*res, self, None)

    def _xyz2ic_cdf(self, x, desc4, eps, whiten=None, acc=None):
        """xyz -> IC with the four cdf domain maps fused in (bgk_xyz2ic_cdf_uni: the NLL direction of a builder flow's tail); None when
        the launch is outside the kernel's envelope.  No autograd."""
        dev = x.device
        n, nf = self._n, self._n_fixed
        x2 = x.flatten(1)
        B = x2.shape[0]
        if whiten is None:
            mean = T = None
            keep, jac = 3 * nf, 0.0
        else:
            mean, T, jac = whiten
            keep = T.shape[1]
        if not (self._normalize_angles and x2.is_contiguous() and x2.shape[1] == 3 * (n + nf) and B > 0 and n + nf <= 32 and keep <= 16):
            return None
        ics = torch.empty((3, B, n), dtype=torch.float32, device=dev)
        xfix = torch.empty((B, keep), dtype=torch.float32, device=dev)
        dlogp, accumulate = _dl_target(acc, B, dev)
        const_ld = -n * (np.log(np.pi) + np.log(2.0 * np.pi)) + (float(jac) if T is not None else 0.0)
        with torch.cuda.device(dev):
            st = _lib.lib().bgk_xyz2ic_cdf_uni(
                _lib.ptr(x2), _lib.ptr(desc4), int(eps is not None), float(eps or 0.0), _lib.ptr(self._tables.get("zmat8", dev)), n,
                _lib.ptr(self._tables.get("fixed", dev)), nf, float(self._eps), int(self._enforce_boundaries), _lib.ptr(mean), _lib.ptr(T), keep,
                float(const_ld), B, _lib.ptr(ics[0]), _lib.ptr(ics[1]), _lib.ptr(ics[2]), _lib.ptr(xfix), _lib.ptr(dlogp), accumulate,
                _lib.ptr(self._warn_counter(dev)), _lib.stream_ptr(dev))
        if st == -2:
            return None
        _lib.check(st, "bgk_xyz2ic_cdf_uni")
        return ics[0], ics[1], ics[2], xfix, _dl_result(acc, dlogp)

    def _infer_fused(self, x, desc4, eps, acc=None):
        return self._xyz2ic_cdf(x, desc4, eps, acc=acc)

    def _forward(self, x, with_pose=True, *args, **kwargs):
        return self._xyz2ic(x, acc=kwargs.get(ACC_KW))

    def _inverse(self, bonds, angles, torsions, x_fixed, **kwargs):
        return self._ic2xyz(bonds, angles, torsions, x_fixed, acc=kwargs.get(ACC_KW))


def _pca(X0, keepdims=None):
    """PCA of the rows of X0 (numpy, float64 like the input): mean, whitening and blackening
    matrices and the kept standard deviations, eigenvalues descending (pca.py:9-34)."""
    keepdims = X0.shape[1] if keepdims is None else keepdims
    mean = X0.mean(axis=0)
    Xc = X0 - mean
    cov = Xc.T @ Xc / (Xc.shape[0] - 1.0)
    eigval, eigvec = np.linalg.eigh(cov)
    pick = np.argsort(eigval)[::-1][:keepdims]
    std = np.sqrt(eigval[pick])
    V = eigvec[:, pick]
    return mean, V @ np.diag(1.0 / std), np.diag(std) @ V.T, std


class _WhitenFn(torch.autograd.Function):
    """out = (x - pre) T + post on bgk_whiten; backward: g_x = g_out T^T on the same kernel"""

    @staticmethod
    def forward(ctx, x, T, Tt, pre, post):
        ctx.Tt = Tt
        return _whiten_launch(x, T, pre, post)

    @staticmethod
    def backward(ctx, g):
        return _whiten_launch(g.contiguous(), ctx.Tt, None, None), None, None, None, None


class _WideWhitenFn(torch.autograd.Function):
    """out = x W^T on bgk_dense_layer for a fixed (buffer) matrix; backward: g_x = g W on the same kernel (operands of W^T)"""

    @staticmethod
    def forward(ctx, x, lin):
        from . import dense
        ctx.lin = lin
        return dense.dense_layer(x, lin)

    @staticmethod
    def backward(ctx, g):
        from . import dense
        return dense.dense_layer(g.contiguous(), ctx.lin, transposed=True), None


def _whiten_launch(x, T, pre, post):
    x2, ldx = _lib.rowmajor(x)
    B, n_in = x2.shape
    n_out = T.shape[1]
    out = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = _lib.lib().bgk_whiten(_lib.ptr(x2), ldx, _lib.ptr(T), _lib.ptr(pre), _lib.ptr(post), n_in, n_out, B, _lib.ptr(out), n_out,
                                   _lib.stream_ptr(x.device))
    _lib.check(st, "bgk_whiten")
    return out


class WhitenFlow(Flow):
    """Static PCA whitening ``z = (x - mean) @ Twhiten`` with constant log-det (pca.py:37-107).  Stand-alone, blocks of up to 128
    coordinates run on bgk_whiten (mean shift fused, VJP on the same kernel); larger ones on bgk_dense_layer (round 6; a library GEMM before).  Inside
    MixedCoordinateTransformation the product is fused into the IC kernels.  Buffers: X0mean, Twhiten, Tblacken, std."""

    def __init__(self, X0, keepdims=None, whiten_inverse=True):
        super().__init__()
        keepdims = X0.shape[1] if keepdims is None else keepdims
        self.dim = X0.shape[1]
        self.keepdims = keepdims
        self.whiten_inverse = whiten_inverse
        mean, Tw, Tb, std = _pca(X0.detach().cpu().numpy(), keepdims=keepdims)
        self.register_buffer("X0mean", torch.tensor(mean).to(X0))
        self.register_buffer("Twhiten", torch.tensor(Tw).to(X0))
        self.register_buffer("Tblacken", torch.tensor(Tb).to(X0))
        self.register_buffer("std", torch.tensor(std).to(X0))
        if torch.any(self.std <= 0):
            raise ValueError("Cannot construct whiten layer because trying to keep nonpositive eigenvalues.")
        self.jacobian_xz = -torch.sum(torch.log(self.std))

    def _kernel(self, x, which):
        """(x - X0mean) Twhiten | x Tblacken + X0mean on bgk_whiten, or None (not a 2-d f32 HIP tensor, block wider than 128)"""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2) or max(self.dim, self.keepdims) > 128:
            return None
        key = (str(x.device), self.Twhiten.data_ptr(), self.Twhiten._version, self.Tblacken._version, self.X0mean._version)
        cache = self.__dict__.get("_kernel_mats")
        if cache is None or cache[0] != key:
            f = lambda t: t.detach().to(device=x.device, dtype=torch.float32).contiguous()      # noqa: E731
            Tw, Tb, m = f(self.Twhiten), f(self.Tblacken), f(self.X0mean)
            cache = self.__dict__["_kernel_mats"] = (key, dict(whiten=(Tw, Tw.t().contiguous(), m, None),
                                                              blacken=(Tb, Tb.t().contiguous(), None, m)))
        T, Tt, pre, post = cache[1][which]
        if torch.is_grad_enabled() and x.requires_grad:
            return _WhitenFn.apply(x, T, Tt, pre, post)
        return _whiten_launch(x, T, pre, post)

    def _wide_kernel(self, x, which):
        """blocks wider than 128 coordinates on bgk_dense_layer (round 6; before: torch.matmul -> hipBLASLt): the product as a bias-free
        ``Linear`` whose weight is the (transposed) PCA matrix -- split-f16 MFMA GEMM, f32-class, its input gradient on the same kernel
        (dense._LinearFn).  None: not a 2-d f32 HIP tensor (the caller runs torch.matmul)."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
            return None
        from . import dense
        key = (str(x.device), self.Twhiten.data_ptr(), self.Twhiten._version, self.Tblacken._version)
        cache = self.__dict__.get("_wide_lins")
        if cache is None or cache[0] != key:
            lins = {}
            for name, T in (("whiten", self.Twhiten), ("blacken", self.Tblacken)):
                lin = torch.nn.Linear(T.shape[0], T.shape[1], bias=False)
                lin.weight = torch.nn.Parameter(T.detach().to(device=x.device, dtype=torch.float32).t().contiguous(), requires_grad=False)
                lins[name] = lin
            cache = self.__dict__["_wide_lins"] = (key, lins)
        lin = cache[1][which]
        if torch.is_grad_enabled() and x.requires_grad:
            return _WideWhitenFn.apply(x, lin)
        return dense.dense_layer(x, lin)

    def _whiten(self, x):
        z = self._kernel(x, "whiten")
        if z is None:
            z = self._wide_kernel(x - self.X0mean, "whiten")
        if z is None:
            z = torch.matmul(x - self.X0mean, self.Twhiten)
        return z, self.jacobian_xz.to(x) * torch.ones((x.shape[0], 1), dtype=x.dtype, device=x.device)

    def _blacken(self, z):
        x = self._kernel(z, "blacken")
        if x is None:
            x = self._wide_kernel(z, "blacken")
            if x is not None:
                x = x + self.X0mean
        if x is None:
            x = torch.matmul(z, self.Tblacken) + self.X0mean
        return x, -self.jacobian_xz.to(z) * torch.ones((z.shape[0], 1), dtype=z.dtype, device=z.device)

    def _forward(self, x, *args, **kwargs):
        return self._blacken(x) if self.whiten_inverse else self._whiten(x)

    def _inverse(self, x, *args, **kwargs):
        return self._whiten(x) if self.whiten_inverse else self._blacken(x)


class MixedCoordinateTransformation(Flow):
    """Relative internal coordinates + PCA-whitened fixed atoms (crd_transform/ic.py:719-884).
    Sub-modules keep the reference's names (``_whiten``, ``_rel_ic``) so state_dicts load."""

    def __init__(self, data, z_matrix, fixed_atoms, keepdims=None, normalize_angles=True, eps=1e-7,
                 enforce_boundaries=True, raise_warnings=True):
        super().__init__()
        n_data = data.shape[0]
        fa = np.asarray(fixed_atoms if not torch.is_tensor(fixed_atoms) else fixed_atoms.cpu().numpy())
        fixed = data.view(n_data, -1, 3)[:, fa].reshape(n_data, -1)
        self._whiten = WhitenFlow(fixed, keepdims=keepdims, whiten_inverse=False)
        self._rel_ic = RelativeInternalCoordinateTransformation(
            z_matrix=z_matrix, fixed_atoms=fixed_atoms, normalize_angles=normalize_angles, eps=eps,
            enforce_boundaries=enforce_boundaries, raise_warnings=raise_warnings)

    z_matrix = property(lambda self: self._rel_ic.z_matrix)
    fixed_atoms = property(lambda self: self._rel_ic.fixed_atoms)
    dim_bonds = property(lambda self: len(self.z_matrix))
    dim_angles = property(lambda self: len(self.z_matrix))
    dim_torsions = property(lambda self: len(self.z_matrix))
    dim_fixed = property(lambda self: self._whiten.keepdims)
    bond_indices = property(lambda self: self._rel_ic.bond_indices)
    angle_indices = property(lambda self: self._rel_ic.angle_indices)
    torsion_indices = property(lambda self: self._rel_ic.torsion_indices)
    normalize_angles = property(lambda self: self._rel_ic.normalize_angles)

    def _wh(self, which, device):
        w = self._whiten
        T = (w.Twhiten if which == "whiten" else w.Tblacken).to(device=device, dtype=torch.float32).contiguous()
        mean = w.X0mean.to(device=device, dtype=torch.float32).contiguous()
        return mean, T, float(w.jacobian_xz)

    _bgk_acc = True

    def _forward(self, x, *args, **kwargs):
        return self._rel_ic._xyz2ic(x, whiten=self._wh("whiten", x.device), acc=kwargs.get(ACC_KW))

    def _inverse(self, bonds, angles, torsions, z_fixed, *args, **kwargs):
        return self._rel_ic._ic2xyz(bonds, angles, torsions, z_fixed, blacken=self._wh("blacken", bonds.device), acc=kwargs.get(ACC_KW))

    def _infer_fused(self, x, desc4, eps, acc=None):
        return self._rel_ic._xyz2ic_cdf(x, desc4, eps, whiten=self._wh("whiten", x.device), acc=acc)

    def _generate_fused(self, bonds, angles, torsions, z_fixed, descs, eps, acc=None, desc20=None):
        return self._rel_ic._icdf_ic2xyz(bonds, angles, torsions, z_fixed, descs, eps, blacken=self._wh("blacken", bonds.device), acc=acc,
                                         desc20=desc20)

    def _generate_fused_train(self, bonds, angles, torsions, z_fixed, eps, desc20, kl=None):
        blacken = self._wh("blacken", bonds.device)
        res = self._rel_ic._icdf_ic2xyz_train(bonds, angles, torsions, z_fixed, eps, blacken, desc20, kl=kl)
        return None if res is None else (*res, self._rel_ic, blacken)


def slice_initial_atoms(z_matrix):
    """The three rows of a global Z-matrix with -1 entries define the initial atoms (most -1s first);
    the rest is a relative Z-matrix (crd_transform/ic.py:94-97)."""
    z = np.asarray(z_matrix)
    missing = np.sum(z == -1, axis=-1)
    order = np.argsort(missing)[::-1][:3]
    return z[:, 0][order], z[missing == 0]


class ReferenceSystemTransformation(Flow):
    """Origin + Euler orientation + (d01, d12, a012) of the first three atoms (crd_transform/ic.py:128-265)
    on the kernel bgk_ic_refsys; log|det J| in closed form instead of an autograd 9x9 Jacobian."""

    def __init__(self, normalize_angles=True, eps=1e-7, enforce_boundaries=True, raise_warnings=True):
        super().__init__()
        self._normalize_angles = normalize_angles
        self._eps = eps
        self._enforce_boundaries = enforce_boundaries
        self._raise_warnings = raise_warnings

    _bgk_acc = True

    def _launch(self, packed, inverse, acc=None):
        _lib.require_hip(packed)
        packed = packed.contiguous()
        if torch.is_grad_enabled() and packed.requires_grad:
            return _RefSysFn.apply(self, packed, inverse)
        return self._launch_nograd(packed, inverse, acc=acc)

    def _launch_nograd(self, packed, inverse, acc=None):
        B = packed.shape[0]
        out = torch.empty_like(packed)
        dlogp, accumulate = _dl_target(acc, B, packed.device)
        with torch.cuda.device(packed.device):
            st = _lib.lib().bgk_ic_refsys(_lib.ptr(packed), B, int(inverse), int(self._normalize_angles), float(self._eps),
                                          int(self._enforce_boundaries), _lib.ptr(out), _lib.ptr(dlogp), accumulate,
                                          _lib.stream_ptr(packed.device))
        _lib.check(st, "bgk_ic_refsys")
        return out, _dl_result(acc, dlogp)

    def _forward(self, x0, x1, x2, *args, **kwargs):
        B = x0.shape[0]
        out, dlogp = self._launch(torch.cat([x0.reshape(B, 3), x1.reshape(B, 3), x2.reshape(B, 3)], dim=-1), False, acc=kwargs.get(ACC_KW))
        return out[:, None, 0:3], out[:, 6:9], out[:, 3:4], out[:, 4:5], out[:, 5:6], dlogp

    def _inverse(self, x0, orientation, d01, d12, a012, *args, **kwargs):
        B = x0.shape[0]
        out, dlogp = self._launch(torch.cat([x0.reshape(B, 3), d01, d12, a012, orientation], dim=-1), True, acc=kwargs.get(ACC_KW))
        return out[:, None, 0:3], out[:, None, 3:6], out[:, None, 6:9], dlogp


class GlobalInternalCoordinateTransformation(Flow):
    """Full Z-matrix transform: every atom but the origin / orientation of the first three becomes an
    internal coordinate (crd_transform/ic.py:516-716).  forward: x -> bonds [B,n+2], angles [B,n+1],
    torsions [B,n], x0 [B,1,3], R [B,3], dlogp."""

    def __init__(self, z_matrix, normalize_angles=True, eps=1e-7, enforce_boundaries=True, raise_warnings=True):
        super().__init__()
        initial_atoms, z_rel = slice_initial_atoms(z_matrix)
        self._rel_ic = RelativeInternalCoordinateTransformation(
            z_matrix=z_rel, fixed_atoms=initial_atoms, normalize_angles=normalize_angles, eps=eps,
            enforce_boundaries=enforce_boundaries, raise_warnings=raise_warnings)
        self._ref_ic = ReferenceSystemTransformation(normalize_angles=normalize_angles, eps=eps,
                                                     enforce_boundaries=enforce_boundaries, raise_warnings=raise_warnings)

    z_matrix = property(lambda self: self._rel_ic.z_matrix)
    fixed_atoms = property(lambda self: np.array([], dtype=np.int64))
    dim_bonds = property(lambda self: len(self.z_matrix) + 2)
    dim_angles = property(lambda self: len(self.z_matrix) + 1)
    dim_torsions = property(lambda self: len(self.z_matrix))
    dim_fixed = property(lambda self: 0)
    torsion_indices = property(lambda self: self._rel_ic.torsion_indices)
    normalize_angles = property(lambda self: self._rel_ic.normalize_angles)

    @property
    def bond_indices(self):
        fix = self._rel_ic.fixed_atoms
        return np.vstack([np.array([[fix[1], fix[0]], [fix[2], fix[1]]]), self._rel_ic.bond_indices])

    @property
    def angle_indices(self):
        fix = self._rel_ic.fixed_atoms
        return np.vstack([np.array([[fix[2], fix[1], fix[0]]]), self._rel_ic.angle_indices])

    _bgk_acc = True

    def _forward(self, x, *args, **kwargs):
        B = x.shape[0]
        acc = kwargs.get(ACC_KW)
        bonds, angles, torsions, x_fixed, dlogp_rel = self._rel_ic._xyz2ic(x.flatten(1), acc=acc)
        ref, dlogp_ref = self._ref_ic._launch(x_fixed.reshape(B, 9), False, acc=acc)
        bonds = torch.cat([ref[:, 3:5], bonds], dim=-1)
        angles = torch.cat([ref[:, 5:6], angles], dim=-1)
        return bonds, angles, torsions, ref[:, None, 0:3], ref[:, 6:9], _dl_merge(acc, dlogp_rel, dlogp_ref)

    def _inverse(self, bonds, angles, torsions, x0, R, *args, **kwargs):
        B = bonds.shape[0]
        acc = kwargs.get(ACC_KW)
        packed = torch.cat([x0.reshape(B, 3), bonds[:, 0:2], angles[:, 0:1], R], dim=-1)
        x_init, dlogp_ref = self._ref_ic._launch(packed, True, acc=acc)
        x, dlogp_rel = self._rel_ic._ic2xyz(bonds[:, 2:], angles[:, 1:], torsions, x_init, acc=acc)
        return x, _dl_merge(acc, dlogp_rel, dlogp_ref)
