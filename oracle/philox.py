"""Philox4x32-10 and the field sampler of csrc/bgk_philox.hip restated in numpy (integer arithmetic in uint64, exact).
TEST INFRASTRUCTURE ONLY (tests/, never the product).

Algorithm: J. K. Salmon, M. A. Moraes, R. O. Dror, D. E. Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11 (the Random123
library): 10 rounds of  (c0, c1, c2, c3) <- (hi(M1 c2) ^ c1 ^ k0, lo(M1 c2), hi(M0 c0) ^ c3 ^ k1, lo(M0 c0)),  key bumped by the Weyl
constants (0x9E3779B9, 0xBB67AE85) per round, M0 = 0xD2511F53, M1 = 0xCD9E8D57.  Pinned on Random123's known-answer vectors in
tests/test_oracle_golden.py.  The kernel's counter layout: (row low, row high, field << 20 | 4-column block, call offset), key = seed.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: 4 arrays (or ints) of uint32 counter words, key: (k0, k1) -> list of 4 uint32 arrays"""
    c = [np.asarray(v, dtype=np.uint64) for v in ctr]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & _MASK, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & _MASK]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def u01(x):
    """((x >> 8) + 0.5) 2^-24 in f32: exact (24-bit integer + 0.5, scaled by a power of two)"""
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def sample_field(seed, offset, field, n_rows, d, kind, row0=0):
    """raw variates of one field as the kernel draws them: kind 0 -> u in (0, 1) [n_rows, d] (bit-exact), kind 1 -> standard
    normals by Box-Muller on consecutive word pairs (f64 arithmetic here: the kernel's f32 forms agree to ~1e-6)"""
    rows = np.arange(row0, row0 + n_rows, dtype=np.uint64)
    out = np.zeros((n_rows, 4 * ((d + 3) // 4)), np.float64 if kind == 1 else np.float32)
    for cb in range((d + 3) // 4):
        w = philox4x32_10([rows & _MASK, rows >> np.uint64(32), np.full(n_rows, (field << 20) | cb, np.uint64), np.full(n_rows, offset, np.uint64)],
                          (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
        if kind == 0:
            for q in range(4):
                out[:, 4 * cb + q] = u01(w[q])
        else:
            for h in range(2):
                rad = np.sqrt(-2.0 * np.log(u01(w[2 * h]).astype(np.float64)))
                ang = 2.0 * np.pi * u01(w[2 * h + 1]).astype(np.float64)
                out[:, 4 * cb + 2 * h], out[:, 4 * cb + 2 * h + 1] = rad * np.cos(ang), rad * np.sin(ang)
    return out[:, :d]
