"""GPU tests of the per-step tile kernel's three tile shapes (ccvm_amd/csrc/ccvm_kernels.h: 32 x 128, 32 x 64 split-K
(KS = 2), 32 x 32 split-K (KS = 4)) and of the shapes the default policy now picks: grids of several workgroups per CU
where the finer tiles round up less (N = 1200 ... 1500 at B = 1000) and mid-size batches that leave most of the chip
idle with the wider tiles (B = 129 ... 256 at N = 1000).  The reference runs every shape through the same einsum
(dl_solver.py:145-153, mf_solver.py:214-222, langevin_solver.py:131-139).

Every word of every trajectory against the oracle (fused noise through oracle/noise_ref.py); chunking bit-exact;
sharding bit-exact at equal tile shape (the shape fixes the summation order of a column's contraction)."""
import re

import pytest
import torch

from test_gpu_cluster import _ADAMS, _run_engine, _run_oracle, ATOL_X
from test_gpu_slab import _check_against_oracle, _describe

pytestmark = pytest.mark.gpu


def _ks(kind, b, n, adam=False):
    if "ptile_kernel" in _describe(kind, b, n, adam):  # the persistent form of the 32 x 128 tiles (ccvm_ptile.h)
        return 1
    return int(re.search(r"step_kernel<\d, \w+, 0, (\d)", _describe(kind, b, n, adam)).group(1))


@pytest.mark.parametrize("ks", [1, 2, 4])
@pytest.mark.parametrize("kind,n,b,t,adam", [
    ("dl", 1000, 96, 6, None), ("mf", 1000, 70, 6, "second_moment"), ("langevin", 1030, 33, 8, "add_assign"),
    ("pl", 700, 130, 8, None), ("dl", 300, 40, 10, None), ("mf", 257, 5, 10, None), ("langevin", 31, 3, 12, None),
    ("pl", 2000, 40, 4, "first_moment_only"), ("dl", 129, 1, 12, None),
])
def test_every_tile_shape_matches_oracle(monkeypatch, ks, kind, n, b, t, adam):
    monkeypatch.setenv("CCVM_AMD_KERNEL", "tile")
    monkeypatch.setenv("CCVM_AMD_KS", str(ks))
    assert _ks(kind, b, n, adam is not None) == ks
    _check_against_oracle(kind, n, b, t, adam)


@pytest.mark.parametrize("kind,n,b,t,adam,ks", [
    # 32 x 32 tiles by default: three quarters of the chip idle otherwise (the slab kernel has no cheaper plan here)
    ("dl", 1000, 256, 6, None, 4), ("langevin", 1000, 256, 8, None, 4), ("mf", 1000, 250, 6, "second_moment", 4),
    ("dl", 2000, 128, 3, None, 4), ("pl", 1500, 128, 4, "add_assign", 4), ("dl", 700, 256, 6, None, 4),
    # 32 x 64 tiles in three rounds instead of 32 x 128 tiles in two
    ("langevin", 1500, 1000, 3, None, 2), ("mf", 1300, 1000, 3, "add_assign", 2),
    # several rounds of 32 x 32 tiles (round 5: their later rounds overlap the launch boundary; measured 53.4 us against
    # 58.5 on 32 x 64 tiles at DL N = 1200, B = 1000; 28.8 against 31.2 resident at N = 1000, B = 768; 37.2 against 47.0)
    ("dl", 1200, 1000, 3, None, 4), ("langevin", 1200, 1000, 3, None, 4), ("dl", 1000, 768, 3, None, 4),
    ("dl", 1500, 384, 3, None, 4),
    # unchanged: one workgroup per CU
    ("dl", 1000, 1000, 4, None, 1), ("pl", 2000, 512, 3, None, 1),
    # seven rounds of 32 x 32 tiles in the blocked order against two of 32 x 128 (95.9 vs 104.1 us per step, round 5)
    ("dl", 1700, 1000, 2, None, 4),
])
def test_default_tile_shapes_match_oracle(monkeypatch, kind, n, b, t, adam, ks):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    # (the shapes of a WHOLE batch: some of these batches are cut in two by default since round 4 -- MF N = 1300,
    # B = 1000 into a resident grid of the persistent tile kernel plus 232 rows -- which tests/test_gpu_ptile.py covers)
    monkeypatch.setenv("CCVM_AMD_SPLIT", "0")
    assert _ks(kind, b, n, adam is not None) == ks, _describe(kind, b, n, adam is not None)
    _check_against_oracle(kind, n, b, t, adam)


@pytest.mark.parametrize("ks", [2, 4])
@pytest.mark.parametrize("kind,n,b", [("dl", 1000, 200), ("mf", 600, 70), ("langevin", 1100, 256)])
def test_chunking_and_sharding_are_exact_at_equal_tile_shape(monkeypatch, ks, kind, n, b):
    monkeypatch.setenv("CCVM_AMD_KERNEL", "tile")
    monkeypatch.setenv("CCVM_AMD_KS", str(ks))
    t = 12
    adam = None if kind == "dl" else _ADAMS["second_moment"]
    whole = _run_engine(kind, n, b, t, adam, 4242, 0)
    parts = _run_engine(kind, n, b, t, adam, 4242, 0, chunks=[1, 5, 2, 4])
    for name in whole.state:
        assert torch.equal(whole.compact(name), parts.compact(name)), name
    cut = 37 if b > 37 else 1
    lo = _run_engine(kind, n, cut, t, adam, 4242, 0)
    hi = _run_engine(kind, n, b - cut, t, adam, 4242, cut)
    for name in whole.state:
        w = whole.compact(name)
        assert torch.equal(w[:cut], lo.compact(name)) and torch.equal(w[cut:], hi.compact(name)), name


def test_per_variable_saturation_on_the_finest_tiles(monkeypatch):
    """The VS instantiations of the 32 x 32 tiles (ccvm_tile4_{mf,lv}.hip) against the scalar-S run: S_j = S for
    every column must reproduce it bit for bit (same kernels otherwise)."""
    from ccvm_amd import engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv

    monkeypatch.setenv("CCVM_AMD_KERNEL", "tile")
    monkeypatch.setenv("CCVM_AMD_KS", "4")
    for kind, n, b in (("mf", 600, 50), ("langevin", 1000, 100)):
        q, v, _ = scaled_qv(n, kind)
        prob = engine.DeviceProblem(q, v)
        p = dict(EXAMPLE_PARAMS[kind])
        if kind == "mf":
            p["g"] = 0.01
        outs = []
        for vec in (False, True):
            pp = dict(p)
            if vec:
                pp["S"] = torch.full((n,), float(p["S"]))
            noise = engine.NoiseSpec(mode="philox", seed=31337, row_offset=0)
            traj = engine.Trajectories(prob, b, "mf" if kind == "mf" else "langevin", 10, pp, (0.0, 1.0), noise)
            traj.advance(10)
            traj.check()
            outs.append({k: traj.compact(k).clone() for k in traj.state})
        for k in outs[0]:
            assert float((outs[0][k] - outs[1][k]).abs().max()) <= ATOL_X * (n / 20.0) ** 0.5, (kind, k)
