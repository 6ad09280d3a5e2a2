"""GPU tests of the K split of the row-owner persistent kernel (ccvm_amd/csrc/ccvm_persist.h, KH = 2: 64 < N <= 256,
two or four waves side by side x two K halves; the halves swap partial sums through LDS and each finishes two of the four
MFMA rows).  The reference's loop bodies: dl_solver.py:523-564, mf_solver.py:549-589, langevin_solver.py:411-433.

Every word of every trajectory against the oracle with the split forced on and off, for every count of K chunks
(NCH = 5 ... 8), ragged batches, odd shard starts and the Adam variants; chunking and sharding bit-exact per variant."""
import re

import pytest
import torch

from test_gpu_cluster import _ADAMS, _run_engine
from test_gpu_slab import _check_against_oracle, _describe

pytestmark = pytest.mark.gpu


def _kh(kind, b, n, adam=False):
    return int(re.search(r"persist_kernel<\d, \w+, \d+, \d+, \d+, \d+, (\d)(?:, \d+)*>", _describe(kind, b, n, adam)).group(1))


@pytest.mark.parametrize("kh", [1, 2])
@pytest.mark.parametrize("kind,n,b,t,adam", [
    ("dl", 100, 1000, 30, None), ("dl", 65, 7, 30, None), ("dl", 80, 33, 30, None), ("dl", 96, 130, 30, None),
    ("dl", 113, 1, 30, None), ("dl", 128, 257, 30, None), ("mf", 100, 1000, 30, None), ("mf", 72, 9, 30, "second_moment"),
    ("mf", 128, 100, 30, "add_assign"), ("langevin", 100, 1000, 30, None), ("langevin", 81, 5, 30, "first_moment_only"),
    ("pl", 100, 300, 30, "second_moment"), ("pl", 127, 66, 30, None), ("langevin", 97, 2, 30, None),
    # four waves side by side (128 < N <= 256): one row set of eight waves
    ("dl", 129, 33, 20, None), ("dl", 200, 500, 12, None), ("dl", 256, 64, 12, None), ("dl", 177, 3, 20, None),
    ("mf", 144, 50, 20, "second_moment"), ("mf", 250, 700, 10, None), ("langevin", 256, 1000, 8, "add_assign"),
    ("pl", 193, 129, 16, None), ("langevin", 225, 7, 20, "first_moment_only"), ("dl", 240, 1000, 6, None),
    # three waves side by side (128 < N <= 192, round 6): workgroups of three (whole chains) or six waves, two and four rows in use
    ("dl", 160, 600, 10, None), ("dl", 130, 255, 12, None), ("langevin", 176, 1100, 8, "second_moment"), ("mf", 192, 300, 12, None),
    ("mf", 145, 777, 8, "add_assign"), ("pl", 161, 512, 10, None),
])
def test_k_split_matches_oracle(monkeypatch, kh, kind, n, b, t, adam):
    monkeypatch.setenv("CCVM_AMD_PERSIST_KH", str(kh))
    assert _kh(kind, b, n, adam is not None) == kh
    _check_against_oracle(kind, n, b, t, adam)


@pytest.mark.parametrize("kind,n,b,t,adam", [
    ("dl", 129, 33, 20, None), ("dl", 144, 1000, 8, None), ("dl", 177, 515, 10, None), ("dl", 192, 2001, 6, None),
    ("mf", 144, 50, 20, "second_moment"), ("mf", 160, 1100, 8, None), ("mf", 176, 2000, 6, None), ("mf", 150, 1030, 8, "add_assign"),
    ("langevin", 176, 1100, 8, "second_moment"), ("langevin", 130, 2500, 6, None), ("pl", 161, 512, 10, None),
    ("langevin", 192, 1030, 8, "first_moment_only"), ("pl", 145, 7, 16, None),
])
def test_two_row_sets_per_workgroup_match_oracle(monkeypatch, kind, n, b, t, adam):
    """Round 6: 128 < N <= 192, K split, TWO six-wave row sets per twelve-wave workgroup (ccvm_persist.h: RSWO = 2) -- forced
    here at every batch, ragged ones and a single row included; the default takes it where it costs fewer rounds."""
    monkeypatch.setenv("CCVM_AMD_PERSIST_KH", "2")
    monkeypatch.setenv("CCVM_AMD_PERSIST_RSW", "2")
    assert ", 4, 2, 0, 2> grid" in _describe(kind, b, n, adam is not None) and "x 768 threads" in _describe(kind, b, n, adam is not None)
    _check_against_oracle(kind, n, b, t, adam)


def test_two_row_sets_per_workgroup_are_bit_identical_to_one(monkeypatch):
    """The same K split, one or two row sets per workgroup: the same arithmetic per element, chunked or not."""
    for kind, n, b in (("dl", 160, 700), ("mf", 190, 333), ("langevin", 131, 1027)):
        adam = None if kind == "dl" else _ADAMS["add_assign"]
        monkeypatch.setenv("CCVM_AMD_PERSIST_KH", "2")
        monkeypatch.setenv("CCVM_AMD_PERSIST_RSW", "1")
        one = _run_engine(kind, n, b, 24, adam, 777, 0)
        monkeypatch.setenv("CCVM_AMD_PERSIST_RSW", "2")
        two = _run_engine(kind, n, b, 24, adam, 777, 0, chunks=[1, 9, 3, 11])
        for name in one.state:
            assert torch.equal(one.compact(name), two.compact(name)), (kind, name)


@pytest.mark.parametrize("pw", ["0", "1"])
@pytest.mark.parametrize("ru", ["2", "4"])
@pytest.mark.parametrize("kind,n,b,t,adam", [
    ("dl", 65, 7, 30, None), ("dl", 70, 1000, 12, None), ("dl", 80, 33, 30, None), ("dl", 96, 130, 20, None), ("dl", 81, 1, 30, None),
    ("mf", 72, 9, 30, "second_moment"), ("mf", 96, 1000, 10, None), ("mf", 70, 515, 12, "add_assign"),
    ("langevin", 81, 5, 30, "first_moment_only"), ("langevin", 70, 1027, 10, None), ("pl", 90, 300, 16, "second_moment"),
])
def test_three_narrow_waves_side_by_side_match_oracle(monkeypatch, kind, n, b, t, adam, ru, pw):
    """Round 6: 64 < N <= 96 as THREE 32-column waves side by side, two row groups (eight MFMA rows) each, whole chains, with
    and without noise producer waves, two and four rows in use."""
    monkeypatch.setenv("CCVM_AMD_PERSIST_CW", "32")
    monkeypatch.setenv("CCVM_AMD_PERSIST_RU", ru)
    monkeypatch.setenv("CCVM_AMD_PERSIST_PW", pw)
    d = _describe(kind, b, n, adam is not None)
    assert re.search(r"persist_kernel<\d, \w+, 32, 3, [56], %s, 1%s> grid \d+ x %d threads" % (ru, ", 1" if pw == "1" else "", 384 if pw == "1" else 192), d), d
    _check_against_oracle(kind, n, b, t, adam)


def test_narrow_waves_are_bit_identical_to_wide_ones(monkeypatch):
    """Whole chains either way: a column's arithmetic does not depend on the wave shape (64 columns x 4 rows or 32 x 8)."""
    for kind, n, b in (("dl", 70, 333), ("mf", 96, 200), ("langevin", 65, 1027)):
        adam = None if kind == "dl" else _ADAMS["add_assign"]
        monkeypatch.setenv("CCVM_AMD_PERSIST_KH", "1")
        monkeypatch.setenv("CCVM_AMD_PERSIST_PW", "0")
        monkeypatch.setenv("CCVM_AMD_PERSIST_CW", "64")
        wide = _run_engine(kind, n, b, 24, adam, 777, 0)
        monkeypatch.setenv("CCVM_AMD_PERSIST_CW", "32")
        monkeypatch.setenv("CCVM_AMD_PERSIST_PW", "1")
        narrow = _run_engine(kind, n, b, 24, adam, 777, 0, chunks=[1, 9, 3, 11])
        for name in wide.state:
            assert torch.equal(wide.compact(name), narrow.compact(name)), (kind, name)


@pytest.mark.parametrize("kind,n,b,t", [
    ("dl", 257, 33, 12), ("dl", 300, 1000, 6), ("dl", 320, 515, 6), ("dl", 289, 1, 12), ("dl", 272, 130, 10), ("dl", 305, 64, 10),
    ("langevin", 300, 1000, 6), ("pl", 272, 130, 10), ("langevin", 320, 2000, 4), ("pl", 305, 7, 12), ("langevin", 257, 1, 12),
    ("pl", 288, 1027, 6),
])
def test_five_waves_side_by_side_match_oracle(monkeypatch, kind, n, b, t):
    """Round 6: 256 < N <= 320 on the row-owner kernel -- five waves side by side x two K halves, ten waves of 168 registers,
    the last 32 ... 56 fragments of every wave's K half in LDS (ccvm_persist.h: QL) -- DL and Langevin / pumped Langevin
    without Adam (the default there; the others stay on the column-cluster kernel)."""
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.setenv("CCVM_AMD_PERSIST_WIDE", "1")  # (wherever it applies: the default leaves the smallest batches to the slab kernel)
    d = _describe(kind, b, n)
    assert re.search(r"persist_kernel<[02], false, 64, 5, (17|18|19|20), 4, 2, 0, 0, (32|40|48|56)(, \d+)?> grid \d+ x 640 threads", d), d
    _check_against_oracle(kind, n, b, t, None)


@pytest.mark.parametrize("kind,n,b,t,adam", [
    ("mf", 257, 33, 12, None), ("mf", 272, 1000, 6, None), ("mf", 265, 515, 8, None), ("mf", 260, 1, 12, None),
    ("langevin", 257, 1000, 6, "second_moment"), ("pl", 288, 130, 10, "add_assign"), ("langevin", 280, 2000, 4, "first_moment_only"),
    ("pl", 273, 7, 12, "second_moment"),
])
def test_five_waves_side_by_side_of_mf_and_langevin_with_adam_match_oracle(monkeypatch, kind, n, b, t, adam):
    """... and where their larger working sets leave room: MF up to N = 272 (80 fragments of a wave in registers), Langevin +
    Adam up to N = 288 (88) -- the unequal K split, its short parts all of those registers."""
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.setenv("CCVM_AMD_PERSIST_WIDE", "1")
    d = _describe(kind, b, n, adam is not None)
    assert re.search(r"persist_kernel<[12], \w+, 64, 5, (17|18), 4, 2, 0, 0, (48|56), (80|88)> grid \d+ x 640 threads", d), d
    _check_against_oracle(kind, n, b, t, adam)


@pytest.mark.parametrize("kind,n,b", [("dl", 300, 150), ("langevin", 320, 333), ("pl", 257, 90), ("mf", 270, 120)])
def test_chunking_and_sharding_are_exact_with_five_waves_side_by_side(monkeypatch, kind, n, b):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.setenv("CCVM_AMD_PERSIST_WIDE", "1")
    t = 24
    whole = _run_engine(kind, n, b, t, None, 777, 0)
    parts = _run_engine(kind, n, b, t, None, 777, 0, chunks=[1, 9, 3, 11])
    for name in whole.state:
        assert torch.equal(whole.compact(name), parts.compact(name)), name
    cut = 37
    lo = _run_engine(kind, n, cut, t, None, 777, 0)
    hi = _run_engine(kind, n, b - cut, t, None, 777, cut)
    for name in whole.state:
        w = whole.compact(name)
        assert torch.equal(w[:cut], lo.compact(name)) and torch.equal(w[cut:], hi.compact(name)), name


def test_the_other_variants_stay_on_the_cluster_kernel_between_256_and_320_columns(monkeypatch):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    assert "cluster_kernel" in _describe("mf", 1000, 300) and "cluster_kernel" in _describe("langevin", 1000, 300, adam=True)
    assert "cluster_kernel" in _describe("mf", 1000, 273) and "cluster_kernel" in _describe("langevin", 1000, 289, adam=True)
    assert "cluster_kernel" in _describe("mf", 1000, 260, adam=True)
    assert "cluster_kernel" in _describe("dl", 1000, 321) and "persist_kernel" in _describe("dl", 1000, 320)
    # by the estimates: the slab kernel keeps the smallest batches, the cluster kernel a batch its 48-row clusters hold in ONE
    # round where row sets need two (Langevin, B = 1500: 3.35 us per step against 4.5)
    assert "slab_kernel" in _describe("dl", 32, 300) and "persist_kernel" in _describe("dl", 128, 300)
    assert "slab_kernel" in _describe("langevin", 32, 300) and "persist_kernel" in _describe("langevin", 128, 300)
    assert "cluster_kernel" in _describe("langevin", 1500, 300) and "persist_kernel" in _describe("langevin", 2000, 300)
    assert "estimated 3.80 us per step" in _describe("dl", 1000, 300) and "estimated 2.00 us per step" in _describe("langevin", 1000, 300)
    monkeypatch.setenv("CCVM_AMD_PERSIST_WIDE", "0")
    assert "cluster_kernel" in _describe("dl", 1000, 300) and "slab_kernel" in _describe("dl", 32, 300)
    monkeypatch.delenv("CCVM_AMD_PERSIST_WIDE")
    monkeypatch.setenv("CCVM_AMD_KERNEL", "cluster")
    assert "cluster_kernel" in _describe("dl", 1000, 300)
    monkeypatch.setenv("CCVM_AMD_KERNEL", "nocluster")
    assert "step_kernel" in _describe("dl", 1000, 300)


def test_default_takes_the_split_where_it_costs_fewer_rounds_or_fills_lone_waves(monkeypatch):
    monkeypatch.delenv("CCVM_AMD_PERSIST_KH", raising=False)
    monkeypatch.delenv("CCVM_AMD_PERSIST_RU", raising=False)
    assert _kh("dl", 1000, 100) == 2 and _kh("mf", 1000, 100) == 2 and _kh("dl", 8, 100) == 2
    assert _kh("dl", 2000, 100) == 1 and _kh("langevin", 4000, 100) == 1      # two whole chains = four halves
    assert _kh("dl", 1500, 100) == 2 and _kh("langevin", 3000, 100) == 2      # 1.46 waves per SIMD: three halves < two wholes
    assert _kh("dl", 1000, 64) == 1                                           # one wave per row set: nothing to split
    assert _kh("dl", 1500, 192) == 1 and _kh("dl", 500, 192) == 2 and _kh("langevin", 1000, 256) == 2  # 128 < N <= 256: same rule ...
    assert _kh("dl", 4000, 256) == 2 and _kh("mf", 4000, 176) == 2 and _kh("langevin", 2000, 208) == 1  # ... unless a SIMD holds one unsplit wave
    # small batches, N > 128 (round 6): whole chains over TWO rows while every such row set has a CU of its own
    # (three side by side: MF; DL and Langevin take the six-wave workgroup's unequal K split there, 3 % faster still)
    assert _kh("mf", 512, 144) == 1 and _kh("mf", 513, 144) == 2 and _kh("langevin", 512, 200) == 1 and _kh("langevin", 513, 200) == 2
    assert _kh("dl", 256, 144) == 2 and _kh("dl", 256, 200) == 1
    assert _kh("mf", 512, 224, adam=True) == 1 and _kh("mf", 512, 256) == 2 and _kh("dl", 100, 240) == 2  # (N <= 224: the unsplit kernel's registers)
    assert "persist_kernel<1, false, 64, 3, 9, 2, 1> grid 256 x 192 threads" in _describe("mf", 512, 144)
    assert "persist_kernel<0, false, 64, 3, 9, 4, 2, 0, 0, 0, 16> grid 128 x 384 threads (K split 16 | 128)" in _describe("dl", 256, 144)
    # more row sets than CUs, three waves side by side: two six-wave row sets per workgroup where that is fewer rounds x 1.6
    assert "64, 3, 9, 4, 2, 0, 2> grid 250 x 768 threads" in _describe("dl", 1000, 144) and _kh("dl", 1500, 144) == 1
    assert "0, 2> grid 150 x 768" in _describe("dl", 600, 160) and "0, 2> grid 250 x 768" in _describe("langevin", 2000, 176)
    assert "0, 2>" not in _describe("langevin", 1000, 176) and "0, 2>" not in _describe("mf", 2000, 176, adam=True)


@pytest.mark.parametrize("kh", ["2", "1"])
@pytest.mark.parametrize("kind,n,b", [("dl", 100, 200), ("mf", 120, 77), ("langevin", 90, 300), ("dl", 200, 150),
                                      ("langevin", 256, 90), ("dl", 150, 120), ("mf", 190, 333)])
def test_chunking_and_sharding_are_exact_with_the_split(monkeypatch, kind, n, b, kh):
    monkeypatch.setenv("CCVM_AMD_PERSIST_KH", kh)
    t = 24
    adam = None if kind == "dl" else _ADAMS["add_assign"]
    whole = _run_engine(kind, n, b, t, adam, 777, 0)
    parts = _run_engine(kind, n, b, t, adam, 777, 0, chunks=[1, 9, 3, 11])
    for name in whole.state:
        assert torch.equal(whole.compact(name), parts.compact(name)), name
    cut = 37
    lo = _run_engine(kind, n, cut, t, adam, 777, 0)
    hi = _run_engine(kind, n, b - cut, t, adam, 777, cut)
    for name in whole.state:
        w = whole.compact(name)
        assert torch.equal(w[:cut], lo.compact(name)) and torch.equal(w[cut:], hi.compact(name)), name
