"""Small host-side helpers (no kernels)."""
import zlib
from collections.abc import Iterable

import numpy as np
import torch

__all__ = ["hash_init_", "synth", "param_state_key", "is_list_or_tuple", "pack_tensor_in_tuple", "unpack_tensor_tuple", "pack_tensor_in_list",
           "as_numpy", "assert_numpy"]


def hash_init_(module, scale=1.0):
    """Fill every parameter of ``module`` with closed-form pseudo-random values that depend only on the
    parameter NAME, its shape and the flat index: exact integer arithmetic -> identical on every
    platform / rank, no weight files, no RNG.  Magnitude = ``scale`` / sqrt(fan_in) (torch's default
    Linear range); 1-d parameters use 1/sqrt(len); AffineTransformer's ``_log_alpha`` keeps its
    constructor value.  Used for synthetic benchmark weights and golden-vector generation."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("_log_alpha"):   # learnable scalar of AffineTransformer keeps its ctor value
                continue
            h = zlib.crc32(name.encode()) % 2001
            idx = np.arange(p.numel(), dtype=np.int64)
            v = ((h + 7919 * idx + 104729 * (idx // 97)) % 2001 - 1000).astype(np.float64) / 1000.0
            fan_in = p.shape[-1] if p.dim() > 1 else p.shape[0]
            v = v * (scale / np.sqrt(float(fan_in)))
            # always float32-representable, so float64 copies of a model carry the SAME weights
            v = v.astype(np.float32)
            p.copy_(torch.from_numpy(v.reshape(tuple(p.shape))).to(dtype=p.dtype))
    return module


def synth(seed, *shape, scale=1.0, uniform=False, dtype=np.float32):
    """Closed-form synthetic array (exact integer arithmetic -> bit-identical everywhere, no RNG):
    ``uniform`` -> values in (0,1) on a 20011-level grid; otherwise zero-mean values with standard
    deviation ``scale`` (a rescaled uniform).  Used for fixture inputs so that golden files only
    have to store OUTPUTS."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.int64)
    h = (int(seed) * 1000003 + idx * 7919 + (idx // 97) * 104729 + (idx // 8191) * 31337) % 20011
    u = (h.astype(np.float64) + 0.5) / 20011.0
    v = u if uniform else (u - 0.5) * (scale * 3.4641016151377544)
    return v.reshape(shape).astype(dtype)


def is_list_or_tuple(x):
    return isinstance(x, (list, tuple))


def pack_tensor_in_tuple(seq):
    """bgflow/utils/types.py:46-53"""
    if isinstance(seq, torch.Tensor):
        return (seq,)
    elif isinstance(seq, Iterable):
        return (*seq,)
    return seq


def unpack_tensor_tuple(seq):
    """bgflow/utils/types.py:35-43"""
    if isinstance(seq, torch.Tensor):
        return seq
    if len(seq) == 1:
        return seq[0]
    return (*seq,)


def pack_tensor_in_list(seq):
    """a tensor -> [tensor]; any other iterable -> list of its items (utils/types.py:60-67)"""
    if isinstance(seq, torch.Tensor):
        return [seq]
    try:
        return list(seq)
    except TypeError:
        return seq


def as_numpy(tensor):
    """host numpy copy of anything ``torch.as_tensor`` accepts (utils/types.py:29-31)"""
    return torch.as_tensor(tensor).detach().cpu().numpy()


def assert_numpy(x, arr_type=None):
    """tensor / list / tuple / array -> ndarray (optionally cast); anything else fails the assertion (utils/types.py:16-26)"""
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    if is_list_or_tuple(x):
        x = np.array(x)
    assert isinstance(x, np.ndarray)
    return x if arr_type is None else x.astype(arr_type)


def param_state_key(p):
    """what the packed-operand caches of the fused kernels are keyed on: the parameter's storage, torch's version counter (bumped by
    every in-place update torch knows about) and a generation counter bumped by writers torch does not see (training.FlatAdam's
    fused kernel writes through the flat bucket the parameters are views of)"""
    return (p.data_ptr(), p._version, getattr(p, "_bgk_generation", 0))


ROW_PITCH_FLOATS = int(__import__("os").environ.get("BGK_ROW_PITCH", "32"))
PARAM_PITCH_FLOATS = int(__import__("os").environ.get("BGK_PARAM_PITCH", "4"))


def param_pitch(n_cols):
    """row stride of the saved spline parameters [B, P]: 16-byte aligned rows only -- the streaming spline backward reads 32-byte
    pieces of every row at once, and a 128-byte multiple pitch (1792 B) measured 4 % SLOWER there than the natural 1712 B"""
    q = max(4, PARAM_PITCH_FLOATS)
    return (n_cols + q - 1) // q * q


def row_pitch(n_cols):
    """row stride (floats) of the [B, P] tensors the training kernels exchange (saved spline parameters, their gradients): a multiple
    of 32 floats = 128 bytes, so that the 32- and 64-byte pieces the backward kernels read per lane (one piece of 32 different rows
    per load instruction) never straddle a memory sector -- with the natural 1712-byte pitch of P = 425 + 3 floats
    bgk_dense_backward_dx fetched 885 MB for 717 MB of operands"""
    q = max(4, ROW_PITCH_FLOATS)
    return (n_cols + q - 1) // q * q
