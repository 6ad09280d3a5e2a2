"""ORACLE -- test infrastructure.  Host restatement of the engine's fused noise generator
(ccvm_amd/csrc/ccvm_philox.h): Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random
numbers: as easy as 1, 2, 3", SC'11; constants M0 = 0xD2511F53, M1 = 0xCD9E8D57,
W0 = 0x9E3779B9, W1 = 0xBB67AE85) on the counter (column, row_lo, step, row_hi) with the key
(seed_lo, seed_hi), then Box-Muller on 24-bit uniforms.

The integer stage is bit-exact with the device; the float stage is evaluated in float64
here, the device uses v_log_f32 / v_sin_f32 / v_cos_f32, so normals agree to ~1e-6 absolute.
Known-answer check: the Random123 test vector for philox4x32-10 (counter = key = 0 ->
6627e8d5 e169c58d bc57ac4c 9b00dbd8) is asserted in tests/test_philox.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (broadcastable).  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u01(x):
    """((x >> 8) + 0.5) * 2^-24, rounded to float32 like the device computes it."""
    f = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0**-24)
    return f.astype(np.float64)


def normal_pairs(seed, row_offset, step, b, n):
    """(n0, n1) as float32 arrays of shape (B, N): noise of local rows 0..b-1, columns 0..n-1."""
    rows = np.arange(b, dtype=np.int64)[:, None] + np.int64(row_offset)
    cols = np.arange(n, dtype=np.uint32)[None, :]
    urows = rows.astype(np.uint64)
    x0, x1, _, _ = philox4x32_10(
        cols, (urows & MASK).astype(np.uint32), np.uint32(step), (urows >> np.uint64(32)).astype(np.uint32),
        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF,
    )
    u1, u2 = u01(x0), u01(x1)
    r = np.sqrt(-2.0 * np.log(u1))
    theta = 2.0 * np.pi * u2
    return (r * np.cos(theta)).astype(np.float32), (r * np.sin(theta)).astype(np.float32)


class PhiloxNoise:
    """Noise source for oracle.ccvm_oracle loops reproducing the engine's PHILOX mode."""

    def __init__(self, seed, row_offset=0):
        self.seed, self.row_offset = int(seed), int(row_offset)
        self._cache = (None, None)

    def draw(self, step, stream, n, b):
        import torch

        if self._cache[0] != (step, n, b):
            self._cache = ((step, n, b), normal_pairs(self.seed, self.row_offset, step, b, n))
        return torch.from_numpy(self._cache[1][stream].copy())
