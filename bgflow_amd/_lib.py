"""ctypes binding of libbgflow_amd.so (the C ABI declared in include/bgflow_amd.h).

The product path has NO fallback: if the library is missing, cannot be loaded, or a kernel is
asked to run on a non-HIP tensor, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BGK_LIB") or os.path.join(_HERE, "libbgflow_amd.so")   # BGK_LIB: A/B builds (tools/)
_lib = None

i32, i64, f32, f64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p

_SIGNATURES = {
    "bgk_abi_version": (ctypes.c_int, []),
    "bgk_last_error": (ctypes.c_char_p, []),
    "bgk_set_option": (ctypes.c_int, [i32, i32]),
    "bgk_detmath_probe": (ctypes.c_int, [vp, i64, i32, vp, vp]),
    "bgk_rqs_transform": (ctypes.c_int, [vp, i64, vp, i64, i32, vp, i64, i32, i32, i32,
                                         f64, f64, f64, f64, f64, f64, f64, i32,
                                         vp, i64, vp, i32, vp, vp, vp]),
    "bgk_rqs_backward": (ctypes.c_int, [vp, i64, vp, i64, i32, vp, i64, i32, i32, i32,
                                        f64, f64, f64, f64, f64, f64, f64, i32,
                                        vp, i64, vp, vp, i64, vp, i64, vp, i32, vp]),
    "bgk_affine_transform": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, i64, i32,
                                            vp, i64, vp, i32, vp]),
    "bgk_affine_backward": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, i32, i32, i32, i64, i32,
                                           vp, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp]),
    "bgk_coupling_affine_dense_h2_train": (ctypes.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32,
                                                          vp, i32, i32, i32, vp, i64, i64, i32, vp, i64, vp, i32,
                                                          vp, vp, vp, vp, i64, vp, vp, i64, vp]),
    "bgk_mlp_backward_dx": (ctypes.c_int, [vp, i64, i32, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, i32, i64,
                                           vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp]),
    "bgk_pack_mlp_h2": (ctypes.c_int, [vp, vp, i32, i32, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "bgk_pack_mlp_h2_many": (ctypes.c_int, [i32] + [vp] * 18 + [vp]),
    "bgk_coupling_affine_dense_fwd64_train": (ctypes.c_int, [vp, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32,
                                                             vp, i32, i32, i32, vp, i64, i64, i32, vp, i64, vp, i32,
                                                             vp, vp, vp, vp, vp, vp, i64, vp]),
    "bgk_pack_mlp_h2_t": (ctypes.c_int, [vp, i32, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp]),
    "bgk_pack_mlp_h2_t_many": (ctypes.c_int, [i32] + [vp] * 11 + [vp]),
    "bgk_mlp_weight_grad_workspace": (i64, [i64, i32, i32, i32, i32]),
    "bgk_mlp_weight_grad": (ctypes.c_int, [vp, i64, i32, vp, vp, vp, vp, i64, i32, i32, i32, vp, i64, i32, i32, i64, vp, i64,
                                           vp, vp, vp, vp, vp, vp, i32, vp, vp]),
    "bgk_mlp_weight_grad_reduce_many": (ctypes.c_int, [i32] + [vp] * 12 + [i32, vp]),
    "bgk_ic_xyz2ic": (ctypes.c_int, [vp, i64, vp, i32, vp, i32, i32, f32, i32, vp, vp, i32, f32, i64,
                                     vp, vp, vp, i64, vp, i64, vp, i32, vp, vp]),
    "bgk_ic_ic2xyz": (ctypes.c_int, [vp, vp, vp, i64, vp, i64, vp, i32, vp, i32, i32, f32, i32,
                                     vp, vp, i32, f32, i64, vp, i64, vp, i32, vp, vp]),
    "bgk_ic_refsys": (ctypes.c_int, [vp, i64, i32, i32, f32, i32, vp, vp, i32, vp]),
    "bgk_icdf_ic2xyz": (ctypes.c_int, [vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, i32, f32, vp, i32, vp, i32, i32, f32, i32,
                                       vp, vp, i32, f32, i64, vp, i64, vp, i32, vp, vp]),
    "bgk_icdf_ic2xyz_reg": (ctypes.c_int, [vp, vp, vp, vp, vp, i32, f32, vp, i32, vp, i32, f32, i32, vp, vp, i32, f64, i64,
                                           vp, i64, vp, i32, vp, vp]),
    "bgk_icdf_ic2xyz_uni": (ctypes.c_int, [vp, vp, vp, vp, vp, i32, f32, vp, i32, vp, i32, f32, i32, vp, vp, i32, f64, i64,
                                           vp, i64, vp, i32, vp, vp]),
    "bgk_icdf_ic2xyz_uni_train": (ctypes.c_int, [vp, vp, vp, vp, vp, i32, f32, vp, i32, vp, i32, f32, i32, vp, vp, i32, f64, i64,
                                                 vp, i64, vp, i32, vp, vp, vp, vp, vp, vp]),
    "bgk_icdf_ic2xyz_uni_train_kl": (ctypes.c_int, [vp, vp, vp, vp, vp, i32, f32, vp, i32, vp, i32, f32, i32, vp, vp, i32, f64, i64,
                                                    vp, i64, vp, vp, vp, vp, vp, vp, vp, f64, f64, f64, i32, vp, vp, vp, vp, vp]),
    "bgk_xyz2ic_cdf_uni": (ctypes.c_int, [vp, vp, i32, f32, vp, i32, vp, i32, f32, i32, vp, vp, i32, f64, i64,
                                          vp, vp, vp, vp, vp, i32, vp, vp]),
    "bgk_ic_ic2xyz_backward": (ctypes.c_int, [vp, vp, vp, i64, vp, i64, vp, i32, vp, i32, i32, f32, i32, vp, i32, i64,
                                              vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp]),
    "bgk_cdf_transform": (ctypes.c_int, [vp, i64, vp, i64, i32, i32, i32, f32, vp, i64, vp, i32, vp]),
    "bgk_cdf_backward": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, i32, i32, i32, f32, vp, i64, vp, vp, i64, vp]),
    "bgk_ic_xyz2ic_backward": (ctypes.c_int, [vp, i64, vp, i32, vp, i32, i32, f32, i32, vp, i32, i64,
                                              vp, vp, vp, i64, vp, i64, vp, vp, i64, vp]),
    "bgk_ic_refsys_backward": (ctypes.c_int, [vp, vp, vp, i64, i32, i32, f32, i32, vp, vp]),
    "bgk_coupling_rqs_dense": (ctypes.c_int, [vp, i64, i32, i32, vp, vp, vp, i32, i32, i32,
                                              vp, i64, i64, i32, i32, ctypes.c_uint64, i32,
                                              f64, f64, f64, f64, f64, f64, f64, i32,
                                              vp, i64, vp, i32, vp, vp, vp]),
    "bgk_coupling_rqs_dense_h2": (ctypes.c_int, [vp, i64, i32, i32, vp, vp, vp, f32, f32, f32, vp, i32, i32, i32, i32,
                                                 vp, i64, i64, i32, i32, ctypes.c_uint64, i32,
                                                 f64, f64, f64, f64, f64, f64, f64, i32,
                                                 vp, i64, vp, i32, vp, vp, vp]),
    "bgk_coupling_rqs_dense_deep": (ctypes.c_int, [vp, i64, i32, i32, vp, vp, vp, f32, vp, f32, i32, i32,
                                                   vp, i64, i64, i32, i32, ctypes.c_uint64, i32,
                                                   f64, f64, f64, f64, f64, f64, f64, i32,
                                                   vp, i64, vp, i32, vp, vp, vp]),
    "bgk_coupling_rqs_dense_h2_mc": (ctypes.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, f32, f32, f32, vp, i32, i32, i32, i32,
                                                    vp, i64, i64, i32, i32, ctypes.c_uint64, i32,
                                                    f64, f64, f64, f64, f64, f64, f64, i32,
                                                    vp, i64, vp, i32, vp, vp, vp]),
    "bgk_coupling_affine_dense_h2_mc": (ctypes.c_int, [vp, vp, vp, i32, i32,
                                                       vp, vp, vp, f32, f32, f32, i32, vp, vp, vp, f32, f32, f32, i32,
                                                       i32, vp, i32, i32, i32, vp, i64, i64, i32, vp, i64, vp, i32, vp]),
    "bgk_coupling_affine_dense_h3_mc": (ctypes.c_int, [vp, vp, vp, i32, i32,
                                                       vp, vp, vp, vp, f32, f32, f32, f32, i32, vp, vp, vp, vp, f32, f32, f32, f32, i32,
                                                       i32, vp, i32, i32, i32, vp, i64, i64, i32, vp, i64, vp, i32, vp]),
    "bgk_pack_dense_h2": (ctypes.c_int, [vp, vp, i32, i32, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "bgk_coupling_rqs_dense_h2_train": (ctypes.c_int, [vp, i64, i32, i32, vp, vp, vp, f32, f32, f32, vp, i32, i32, i32,
                                                       vp, i64, i64, i32, i32, ctypes.c_uint64, i32,
                                                       f64, f64, f64, f64, f64, f64, f64, i32,
                                                       vp, i64, vp, i32, vp, vp, vp, vp, i64, vp, i32, vp]),
    "bgk_coupling_rqs_dense_h2_backward": (ctypes.c_int, [vp, vp, f32, vp, i32, i32, vp, i64, i64, i32, i32, i32, ctypes.c_uint64, i32,
                                                          f64, f64, f64, f64, f64, f64, f64, i32,
                                                          vp, i64, vp, vp, i64, vp, i64, vp, vp]),
    "bgk_coupling_affine_dense_h2": (ctypes.c_int, [vp, i64, i32, i32,
                                                    vp, vp, vp, f32, f32, f32, i32, vp, vp, vp, f32, f32, f32, i32,
                                                    i32, vp, i32, i32, i32, vp, i64, i64, i32, vp, i64, vp, i32, vp]),
    "bgk_coupling_affine_dense_h3": (ctypes.c_int, [vp, i64, i32, i32,
                                                    vp, vp, vp, vp, f32, f32, f32, f32, i32, vp, vp, vp, vp, f32, f32, f32, f32, i32,
                                                    i32, vp, i32, i32, i32, vp, i64, i64, i32, vp, i64, vp, i32, vp]),
    "bgk_coupling_affine_dense_deep": (ctypes.c_int, [vp, i64, i32, i32,
                                                      vp, vp, vp, f32, vp, f32, i32, vp, vp, vp, f32, vp, f32, i32,
                                                      i32, i32, vp, i32, i32, i32, vp, i64, i64, i32, vp, i64, vp, i32, vp]),
    "bgk_pack_dense_h2_t": (ctypes.c_int, [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]),
    "bgk_dense_backward_dx": (ctypes.c_int, [vp, i64, i32, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, i32, i64,
                                             vp, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp]),
    "bgk_dense_weight_grad_reduce_many": (ctypes.c_int, [i32] + [vp] * 10 + [i32, vp]),
    "bgk_pack_dense_h2_many": (ctypes.c_int, [i32] + [vp] * 14 + [vp]),
    "bgk_pack_dense_h2_t_many": (ctypes.c_int, [i32] + [vp] * 9 + [vp]),
    "bgk_column_sum": (ctypes.c_int, [vp, i64, i64, i32, vp, i32, vp, vp]),
    "bgk_absmax": (ctypes.c_int, [vp, i64, i64, i32, vp, vp]),
    "bgk_whiten": (ctypes.c_int, [vp, i64, vp, vp, vp, i32, i32, i64, vp, i64, vp]),
    "bgk_normal_energy": (ctypes.c_int, [vp, i64, vp, i32, i64, f64, f64, vp, vp]),
    "bgk_normal_energy_backward": (ctypes.c_int, [vp, i64, vp, i32, i64, f64, vp, vp, i64, vp]),
    "bgk_energy_fields": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, i32, i64, f64, f64, f64, vp, vp, i32, vp, i32, vp, vp]),
    "bgk_energy_fields_backward": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, i32, i64, f64, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "bgk_philox_fields": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint32, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, f64, i64, vp, vp]),
    "bgk_grad_nan_flag": (ctypes.c_int, [vp, i64, vp, vp]),
    "bgk_adam_step": (ctypes.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, vp, vp, vp]),
    "bgk_dense_weight_grad_workspace": (i64, [i64, i32, i32]),
    "bgk_dense_weight_grad": (ctypes.c_int, [vp, i64, i32, vp, vp, vp, vp, i32, vp, i64, i32, i32, i64, vp, i64,
                                             vp, vp, vp, vp, vp, vp, i32, vp, vp]),
    "bgk_pack_rqs_columns": (i32, [i32, i32, vp, vp]),
    "bgk_dense_layer": (ctypes.c_int, [vp, i64, i64, i32, vp, i32, f32, vp, vp, i32, i32, vp, i64, i32, vp]),
    "bgk_pack_linear_layer": (ctypes.c_int, [vp, i64, i32, i32, vp, vp, vp]),
    "bgk_dense_layer_steps": (ctypes.c_int, [i32]),
    "bgk_refresh_linear_layer": (ctypes.c_int, [vp, i64, i32, i32, i32, vp, vp, vp, vp]),
    "bgk_activation": (ctypes.c_int, [vp, i64, i64, i32, i32, vp, i64, vp]),
    "bgk_activation_backward": (ctypes.c_int, [vp, i64, vp, i64, i64, i32, i32, vp, i64, vp]),
    "bgk_affine_net_backward64_workspace": (i64, [i64, i32, i32, i32, i32]),
    "bgk_affine_net_backward64": (ctypes.c_int, [vp, i64, i32, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, i32, i64,
                                                 vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, i32, vp]),
    "bgk_affine_coupling_backward64_workspace": (i64, [i64, i32, i32, i32, i32, i32, i32]),
    "bgk_affine_coupling_backward64": (ctypes.c_int, [vp, i64, i32, vp, i64, i32, vp, i64, vp,
                                                      vp, vp, vp, vp, vp, i64,
                                                      vp, vp, vp, vp, vp, vp, i32, i32, i32,
                                                      vp, vp, vp, vp, vp, vp, vp, i32, i32, i32,
                                                      vp, i32, i64, vp, i64, vp, vp, i64, vp, i64, vp, vp, i64, vp, vp, i32, vp]),
    "bgk_linear_weight_grad_workspace": (i64, [i64, i32, i32]),
    "bgk_linear_weight_grad": (ctypes.c_int, [vp, i64, i32, vp, i64, i32, i64, vp, i64, vp, vp, i32, vp, vp]),
}

ABI_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """Load libbgflow_amd.so (built by ``python -m bgflow_amd.build`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X kernels are not built. Run `python -m bgflow_amd.build` "
                "(hipcc, gfx950). bgflow_amd has no CPU fallback.")
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}. bgflow_amd has no CPU fallback.") from e
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError = ABI mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().bgk_last_error().decode(errors="replace")
        if status == -1 and "Minimal bin" in msg:
            raise ValueError(msg)
        raise RuntimeError(f"{what} failed (status {status}): {msg}")


def require_hip(*tensors):
    """Every operand of a kernel must be an f32 tensor on a HIP device (no CPU path)."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "bgflow_amd kernels run on MI355X (HIP) tensors only; got a tensor on "
                f"'{t.device}'. There is no CPU fallback in this package.")
        if t.dtype not in (torch.float32, torch.int32):
            raise RuntimeError(f"bgflow_amd kernels are float32; got {t.dtype}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def cond_segments(parts):
    """host tables (device pointers, row strides, widths) of 1..3 conditioning tensors for the *_mc entry points; the returned
    tuple keeps the ctypes arrays and the (possibly re-laid-out) tensors alive for the duration of the call"""
    rows = [rowmajor(t) for t in parts]
    n = len(rows)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in rows])
    lds = (ctypes.c_int64 * n)(*[ld for _, ld in rows])
    widths = (ctypes.c_int32 * n)(*[t.shape[1] for t, _ in rows])
    return ptrs, lds, widths, n, rows


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def rowmajor(t):
    """Return a 2-d view with unit column stride (copy only if needed) and its row stride."""
    assert t.dim() == 2, "expected [batch, features]"
    if t.stride(1) != 1 and t.shape[1] > 1:
        t = t.contiguous()
    if t.shape[1] == 1 and t.stride(1) != 1:
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else t.shape[1]
    if t.shape[0] > 1 and ld < t.shape[1]:   # broadcast / expanded rows
        t = t.contiguous()
        ld = t.stride(0)
    return t, ld
