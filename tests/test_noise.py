"""The host restatement of the fused noise generator (oracle/noise_ref.py) against the
Random123 known-answer vectors for Threefry2x32 (both the 13-round configuration the device
runs and the 20-round default), plus distributional sanity of the Box-Muller stage."""
import numpy as np

from oracle.noise_ref import ROUNDS, normal_pairs, threefry2x32

KAT = {  # Random123 kat_vectors: threefry2x32 <rounds> <ctr0 ctr1> <key0 key1> -> <out0 out1>
    13: [((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x9D1C5EC6, 0x8BD50731)),
         ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0xFD36D048, 0x2D17272C)),
         ((0x243F6A88, 0x85A308D3), (0x13198A2E, 0x03707344), (0xBA3E4725, 0xF27D669E))],
    20: [((0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6B200159, 0x99BA4EFE)),
         ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
         ((0x243F6A88, 0x85A308D3), (0x13198A2E, 0x03707344), (0xC4923A9C, 0x483DF7A0))],
}


def test_threefry_known_answers():
    assert ROUNDS == 13
    for rounds, vectors in KAT.items():
        for ctr, key, want in vectors:
            got = threefry2x32(ctr[0], ctr[1], key[0], key[1], rounds)
            assert (int(got[0]), int(got[1])) == want, (rounds, ctr, key)


def test_normals_are_standard_and_decorrelated():
    n0, n1 = normal_pairs(0xC0FFEE1234, 0, 5, 1024, 1024)
    for w in (n0, n1):
        x = w.astype(np.float64).ravel()
        assert abs(x.mean()) < 4e-3 and abs(x.var() - 1) < 6e-3
        assert abs((x**3).mean()) < 1.5e-2 and abs((x**4).mean() - 3) < 4e-2
        assert np.isfinite(x).all() and np.abs(x).max() < 6.0
    assert abs((n0.astype(np.float64) * n1).mean()) < 4e-3
    m0, _ = normal_pairs(0xC0FFEE1234, 0, 6, 1024, 1024)  # next step
    assert abs((n0.astype(np.float64) * m0).mean()) < 4e-3
    r0, _ = normal_pairs(0xC0FFEE1234, 1024, 5, 1024, 1024)  # next row block
    assert abs((n0.astype(np.float64) * r0).mean()) < 4e-3
    # sharding: rows [512, 1024) of the block are what row_offset = 512 generates
    h0, h1 = normal_pairs(0xC0FFEE1234, 512, 5, 512, 1024)
    assert np.array_equal(h0, n0[512:]) and np.array_equal(h1, n1[512:])


def test_step_key_known_answers_and_no_seed_step_aliasing():
    """The per-step key is SplitMix64's output function of seed + golden * (step + 1): the first
    outputs of splitmix64 seeded with 0 (Vigna's splitmix64.c) are the keys of steps 0, 1, 2."""
    from oracle.noise_ref import step_key

    assert [step_key(0, i) for i in range(3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    # round 1 keyed the step as seed_lo ^ step: (seed s, step i) and (s ^ d, i ^ d) shared their normals
    a0, a1 = normal_pairs(0, 0, 1, 8, 16)
    b0, b1 = normal_pairs(1, 0, 0, 8, 16)
    assert not np.array_equal(a0, b0) and not np.array_equal(a1, b1)
    c0, _ = normal_pairs(6, 0, 5, 8, 16)
    d0, _ = normal_pairs(5, 0, 6, 8, 16)
    assert not np.array_equal(c0, d0)
    keys = {step_key(s, i) for s in range(64) for i in range(64)}
    assert len(keys) == 64 * 64
    # consecutive small seeds: per-step blocks are uncorrelated
    x, _ = normal_pairs(1, 0, 3, 256, 256)
    y, _ = normal_pairs(2, 0, 3, 256, 256)
    assert abs((x.astype(np.float64) * y).mean()) < 2e-2
