"""ORACLE -- test infrastructure.  Host restatement of the engine's fused noise generator
(ccvm_amd/csrc/ccvm_noise.h): Threefry2x32-13 (Salmon, Moraes, Dror, Shaw: "Parallel random
numbers: as easy as 1, 2, 3", SC'11; rotation constants 13,15,26,6,17,29,16,24, key-schedule
parity 0x1BD11BDA and a key injection after every fourth round, as in Random123 v1.14; 13 rounds
is the paper's Crush-resistant configuration) on

    counter = (column, global_row_lo),  key = (K_lo, K_hi ^ global_row_hi),
    K = step_key(seed, step) = SplitMix64 output function of seed + 0x9E3779B97F4A7C15 * (step + 1)

followed by Box-Muller on 24-bit uniforms.  The integer stage is bit-exact with the device
(pinned by the Random123 known-answer vectors in tests/test_noise.py); the float stage is
evaluated in float64 here while the device uses v_log_f32 / v_sin_f32 / v_cos_f32, so normals
agree to ~1e-6 absolute.
"""
import numpy as np

ROT = (13, 15, 26, 6, 17, 29, 16, 24)
PARITY = 0x1BD11BDA
M32 = 0xFFFFFFFF


def _rotl(x, r):
    return ((x << np.uint32(r)) | (x >> np.uint32(32 - r))).astype(np.uint32)


ROUNDS = 13  # what the device runs


def threefry2x32(c0, c1, k0, k1, rounds=ROUNDS):
    """Vectorised over numpy uint32 arrays (broadcastable).  Returns two uint32 arrays."""
    c0, c1, k0, k1 = np.broadcast_arrays(*(np.asarray(x, dtype=np.uint32) for x in (c0, c1, k0, k1)))
    ks = [k0, k1, (np.uint32(PARITY) ^ k0 ^ k1).astype(np.uint32)]
    with np.errstate(over="ignore"):
        x0 = (c0 + ks[0]).astype(np.uint32)
        x1 = (c1 + ks[1]).astype(np.uint32)
        for r in range(rounds):
            x0 = (x0 + x1).astype(np.uint32)
            x1 = _rotl(x1, ROT[r % 8]) ^ x0
            if (r + 1) % 4 == 0:
                s = (r + 1) // 4
                x0 = (x0 + ks[s % 3]).astype(np.uint32)
                x1 = (x1 + ks[(s + 1) % 3] + np.uint32(s)).astype(np.uint32)
    return x0, x1


M64 = 0xFFFFFFFFFFFFFFFF


def step_key(seed, step):
    """The per-step 64-bit Threefry key (ccvm_noise.h step_key): SplitMix64's output function
    (Steele, Lea, Flood 2014) of seed + golden * (step + 1).  step_key(0, 0) is the first output of
    splitmix64 seeded with 0, 0xE220A8397B1DCDAF."""
    z = ((int(seed) & M64) + 0x9E3779B97F4A7C15 * ((int(step) & M32) + 1)) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def u01(x):
    """((x >> 8) + 0.5) * 2^-24, rounded to float32 like the device computes it."""
    f = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0**-24)
    return f.astype(np.float64)


def normal_pairs(seed, row_offset, step, b, n):
    """(n0, n1) as float32 arrays of shape (B, N): noise of local rows 0..b-1, columns 0..n-1."""
    rows = (np.arange(b, dtype=np.int64)[:, None] + np.int64(row_offset)).astype(np.uint64)
    cols = np.arange(n, dtype=np.uint32)[None, :]
    key = step_key(seed, step)
    k0 = np.uint32(key & M32)
    k1 = (np.uint32(key >> 32) ^ (rows >> np.uint64(32)).astype(np.uint32)).astype(np.uint32)
    x0, x1 = threefry2x32(cols, (rows & np.uint64(M32)).astype(np.uint32), k0, k1)
    u1, u2 = u01(x0), u01(x1)
    r = np.sqrt(-2.0 * np.log(u1))
    theta = 2.0 * np.pi * u2
    return (r * np.cos(theta)).astype(np.float32), (r * np.sin(theta)).astype(np.float32)


def normal_singles(seed, row_offset, step, b, n):
    """One-stream solvers (MF, Langevin, pumped Langevin): global rows 2p and 2p+1 share the call
    of "row" p -- row even takes n0, row odd n1.  float32 array of shape (B, N)."""
    first = int(row_offset) >> 1
    last = (int(row_offset) + b - 1) >> 1
    n0, n1 = normal_pairs(seed, first, step, last - first + 1, n)
    rows = np.arange(b, dtype=np.int64) + int(row_offset)
    pair = (rows >> 1) - first
    return np.where((rows & 1)[:, None] == 0, n0[pair], n1[pair]).astype(np.float32)


class FusedNoise:
    """Noise source for oracle.ccvm_oracle loops reproducing the engine's fused generator.
    ``single=True`` for the one-stream solvers (rows share calls pairwise, see normal_singles)."""

    def __init__(self, seed, row_offset=0, single=False):
        self.seed, self.row_offset, self.single = int(seed), int(row_offset), bool(single)
        self._cache = (None, None)

    def draw(self, step, stream, n, b):
        import torch

        if self.single:
            assert stream == 0
            return torch.from_numpy(normal_singles(self.seed, self.row_offset, step, b, n))
        if self._cache[0] != (step, n, b):
            self._cache = ((step, n, b), normal_pairs(self.seed, self.row_offset, step, b, n))
        return torch.from_numpy(self._cache[1][stream].copy())
