"""GPU tests of the noise producer waves of the row-owner persistent kernel (ccvm_amd/csrc/ccvm_persist.h, PW = 1: a
producer wave per consumer wave makes the NEXT step's normals into an LDS slot, one workgroup barrier per step).  The
reference's loop bodies: dl_solver.py:523-564, mf_solver.py:549-589, langevin_solver.py:411-433,
pumped_langevin_solver.py:286-307.

The producers run the consumers' own generator calls, so a run with them equals the run without them BIT FOR BIT
(fused and replayed noise, every solver, the Adam variants, ragged batches, odd shard starts, chunked runs); both are
checked against the oracle; the default policy takes them where consumers and producers all find a SIMD."""
import re

import pytest
import torch

from test_gpu_cluster import _ADAMS, _run_engine
from test_gpu_slab import _check_against_oracle, _describe

pytestmark = pytest.mark.gpu

_SHAPE = re.compile(r"persist_kernel<\d, \w+, (\d+), (\d+), (\d+), (\d+), (\d+)(, 1)?> grid (\d+) x (\d+) threads")


def _shape(kind, b, n, adam=False):
    m = _SHAPE.search(_describe(kind, b, n, adam))
    assert m, _describe(kind, b, n, adam)
    cw, ncg, nch, ru, kh = (int(m.group(i)) for i in range(1, 6))
    return {"cw": cw, "ncg": ncg, "nch": nch, "ru": ru, "kh": kh, "pw": 1 if m.group(6) else 0, "grid": int(m.group(7)),
            "threads": int(m.group(8))}


CASES = [
    # one wave per row set (N <= 64): the shipped instances' sizes, every K-chunk count, both row fillings
    ("dl", 20, 100, 40, None), ("dl", 20, 1000, 40, None), ("dl", 20, 1, 40, None), ("dl", 16, 37, 30, None),
    ("dl", 7, 5, 30, None), ("dl", 33, 130, 30, None), ("dl", 48, 64, 30, None), ("dl", 64, 1000, 24, None),
    ("dl", 50, 999, 24, None), ("dl", 70, 1000, 24, None), ("dl", 30, 4000, 12, None),
    ("mf", 20, 1000, 40, None), ("mf", 20, 100, 40, "second_moment"), ("mf", 45, 77, 30, "add_assign"),
    ("mf", 64, 300, 24, None), ("mf", 60, 2500, 12, "first_moment_only"),
    ("langevin", 20, 1000, 40, None), ("langevin", 20, 3, 40, "second_moment"), ("langevin", 13, 200, 30, None),
    ("langevin", 64, 513, 24, "add_assign"), ("pl", 20, 1000, 40, None), ("pl", 40, 99, 30, "first_moment_only"),
    ("pl", 63, 2, 30, None),
    # two waves side by side (64 < N <= 128), K split off and on
    ("dl", 100, 1000, 24, None), ("dl", 65, 7, 24, None), ("dl", 128, 257, 16, None), ("mf", 100, 1000, 20, None),
    ("mf", 128, 100, 20, "second_moment"), ("langevin", 100, 1000, 20, None), ("pl", 127, 66, 20, "add_assign"),
]


@pytest.mark.parametrize("kind,n,b,t,adam", CASES)
def test_producer_waves_match_oracle(monkeypatch, kind, n, b, t, adam):
    monkeypatch.setenv("CCVM_AMD_PERSIST_PW", "1")
    if n > 64:
        monkeypatch.setenv("CCVM_AMD_PERSIST_KH", "2")
    assert _shape(kind, b, n, adam is not None)["pw"] == 1
    _check_against_oracle(kind, n, b, t, adam)


@pytest.mark.parametrize("kh", [0, 1, 2])
@pytest.mark.parametrize("kind,n,b,t,adam", CASES)
def test_producer_waves_change_no_bit(monkeypatch, kh, kind, n, b, t, adam):
    if kh and n <= 64:
        pytest.skip("no K split below N = 65")
    if n > 64 and kh == 1:
        pytest.skip("two waves side by side: producers only next to the K split")
    if kh:
        monkeypatch.setenv("CCVM_AMD_PERSIST_KH", str(kh))
    hp = _ADAMS[adam]
    runs = {}
    if n > 64 and not kh:
        monkeypatch.setenv("CCVM_AMD_PERSIST_KH", "2")
    for pw in ("0", "1"):
        monkeypatch.setenv("CCVM_AMD_PERSIST_PW", pw)
        assert _shape(kind, b, n, adam is not None)["pw"] == int(pw)
        runs[pw] = _run_engine(kind, n, b, t, hp, 0xC0FFEE, 3)
    for name in runs["0"].state:
        assert torch.equal(runs["0"].compact(name), runs["1"].compact(name)), name


@pytest.mark.parametrize("kind,n,b", [("dl", 20, 100), ("dl", 20, 1000), ("mf", 33, 77), ("langevin", 64, 300),
                                      ("pl", 20, 129), ("dl", 100, 200)])
def test_chunking_sharding_and_replay_are_exact_with_producers(monkeypatch, kind, n, b):
    monkeypatch.setenv("CCVM_AMD_PERSIST_PW", "1")
    monkeypatch.setenv("CCVM_AMD_PERSIST_KH", "2")  # (N > 64: producers only next to the K split)
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
    # replayed torch noise (the reference's stream): producers load the blocks, consumers never touch them
    rep1 = _run_engine(kind, n, b, t, adam, 99, 0, replay_global_batch=b)
    monkeypatch.setenv("CCVM_AMD_PERSIST_PW", "0")
    rep0 = _run_engine(kind, n, b, t, adam, 99, 0, replay_global_batch=b)
    for name in rep0.state:
        assert torch.equal(rep0.compact(name), rep1.compact(name)), name


def test_long_run_with_producers_is_the_run_without(monkeypatch):
    """4096-step launches back to back (the slot parity runs through thousands of barriers)."""
    outs = {}
    for pw in ("0", "1"):
        monkeypatch.setenv("CCVM_AMD_PERSIST_PW", pw)
        outs[pw] = _run_engine("dl", 20, 1000, 9000, None, 5, 0, chunks=[4096, 4096, 808])
    for name in outs["0"].state:
        assert torch.equal(outs["0"].compact(name), outs["1"].compact(name)), name


def test_default_policy(monkeypatch):
    """(the policy itself is pinned on the host: tests/test_launch_policy.py::test_producer_waves_policy)"""
    for var in ("CCVM_AMD_PERSIST_PW", "CCVM_AMD_PERSIST_RU", "CCVM_AMD_PERSIST_KH"):
        monkeypatch.delenv(var, raising=False)
    for kind in ("dl", "mf", "langevin"):
        for n in (20, 50, 64):
            for b in (1, 100, 1000):
                s = _shape(kind, b, n)
                assert s["pw"] == 1 and s["threads"] == 256, (kind, n, b, s)
    assert _shape("dl", 100, 20)["grid"] == 25 and _shape("dl", 1000, 20)["grid"] == 250  # 4 rows per workgroup at RU = 2
    assert _shape("dl", 1000, 100) == {"cw": 64, "ncg": 2, "nch": 7, "ru": 4, "kh": 2, "pw": 1, "grid": 500, "threads": 512}
    assert _shape("dl", 1500, 100)["pw"] == 0 and _shape("dl", 8000, 64)["pw"] == 0
