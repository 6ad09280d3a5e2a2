"""GPU tests of the whole-run schedule table (include/ccvm_hip.h: `schedule` of ccvm_dl_params / ccvm_langevin_params,
ccvm_dl_schedule, ccvm_mf_schedule, ccvm_langevin_schedule, CCVM_RUN_FORWARD): the per-step scalars (pump and noise ramps,
dl_solver.py:524-527, pumped_langevin_solver.py:279-282; Adam bias corrections, langevin_solver.py:519-540) made once
per run instead of by a small kernel in front of every launch of a persistent path, so that a run call is one launch.

The rows come from the same device code either way: every persistent family must give the same bits with the table
(the engine's default) and with per-call schedule kernels (CCVM_AMD_SCHEDULE=0), however the run is chunked."""
import pytest
import torch

from test_gpu_cluster import _ADAMS, _run_engine
from test_gpu_slab import _describe

pytestmark = pytest.mark.gpu

CASES = [
    # (kind, N, B, steps, adam, family)
    ("dl", 100, 1000, 40, None, "persist_kernel"), ("pl", 64, 300, 30, "second_moment", "persist_kernel"),
    ("dl", 1000, 32, 30, None, "slab_kernel"), ("langevin", 1200, 8, 20, "add_assign", "slab_kernel"),
    ("langevin", 500, 1000, 30, None, "cluster_kernel"), ("dl", 500, 1000, 20, None, "cluster_kernel"),
    ("pl", 640, 1000, 12, "first_moment_only", "cluster_kernel"),
    ("dl", 1000, 1000, 24, None, "ptile_kernel"), ("pl", 2000, 512, 12, "second_moment", "ptile_kernel"),
    ("dl", 1000, 2000, 9, None, "slices"), ("langevin", 1000, 1100, 11, None, "cut in two"),
    ("dl", 1000, 256, 7, None, "step_kernel"),  # (the per-step kernel computes its scalars on the host: no table read)
    # MF: a row's "next step" scalars are the whole run's; the kernels decide "the launch's last step" themselves
    ("mf", 100, 1000, 30, None, "persist_kernel"), ("mf", 500, 32, 20, "second_moment", "slab_kernel"),
    ("mf", 500, 1000, 20, None, "cluster_kernel"), ("mf", 640, 1000, 12, "add_assign", "cluster_kernel"),
    ("mf", 1000, 1000, 16, None, "ptile_kernel"), ("mf", 1000, 2000, 9, "second_moment", "slices"),
]


def _state(traj):
    traj.check()
    assert traj.fallbacks == 0
    return {k: traj.compact(k).clone() for k in traj.state}


@pytest.mark.parametrize("kind,n,b,t,adam,family", CASES)
def test_whole_run_table_equals_per_call_schedules(monkeypatch, kind, n, b, t, adam, family):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    assert family in _describe(kind, b, n, adam is not None)
    hp = _ADAMS[adam]
    chunks = [1, 3, 1, t - 5 - (t - 5) // 2, (t - 5) // 2]
    with_table = _run_engine(kind, n, b, t, hp, 31337, 2, chunks=chunks)
    assert with_table._schedule is not None and with_table.cparams.schedule
    got = _state(with_table)
    whole = _state(_run_engine(kind, n, b, t, hp, 31337, 2))
    monkeypatch.setenv("CCVM_AMD_SCHEDULE", "0")
    per_call = _run_engine(kind, n, b, t, hp, 31337, 2, chunks=chunks)
    assert per_call._schedule is None and not per_call.cparams.schedule
    want = _state(per_call)
    for name in want:
        assert bool(torch.isfinite(want[name]).all()), name
        assert torch.equal(got[name], want[name]), f"{kind} N={n} B={b}: {name} differs with the whole-run table"
        assert torch.equal(whole[name], want[name]), f"{kind} N={n} B={b}: {name} depends on the chunks"


@pytest.mark.parametrize("kind,n,b,adam", [("dl", 1000, 1000, None), ("pl", 2000, 512, "add_assign"), ("dl", 1000, 2000, None)])
def test_flag_lines_are_set_without_the_forward_promise(monkeypatch, kind, n, b, adam):
    """A caller that passes the table but cannot promise that its workspace only ever moved forward (no
    CCVM_RUN_FORWARD): the persistent tile kernel's flag lines are set by a small launch per chunk instead -- same bits."""
    from ccvm_amd import _lib

    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    assert "ptile_kernel" in _describe(kind, b, n, adam is not None)
    hp = _ADAMS[adam]
    want = _state(_run_engine(kind, n, b, 13, hp, 5, 0, chunks=[2, 1, 6, 4]))
    monkeypatch.setattr(_lib, "RUN_FORWARD", 0)
    got = _state(_run_engine(kind, n, b, 13, hp, 5, 0, chunks=[2, 1, 6, 4]))
    for name in want:
        assert torch.equal(got[name], want[name]), name


