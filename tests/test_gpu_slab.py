"""GPU tests of the column-slab small-batch kernel (ccvm_amd/csrc/ccvm_slab.h): N > 256 with few batch rows, Q
resident in the registers of a cluster of workgroups, the GEMM input exchanged as {value, tag} packets, the matvec
rows reduced at wavefront level.  The reference runs any batch_size through the same einsum
(dl_solver.py:145-153, mf_solver.py:214-222, langevin_solver.py:131-139).

Every word of every trajectory is compared with the oracle (fused noise through oracle/noise_ref.py) for every
member width (4 / 8 / 16 / 32 columns), every K (512 ... 2048), clusters inside one XCD and spread over the chip,
one to eight row groups (up to 32 rows per cluster: blocks streamed through the LDS ring, two row pairs per lane),
ragged batches and columns, odd shard starts and the Adam variants."""
import pytest
import torch

from test_gpu_cluster import _ADAMS, _run_engine, _run_oracle, ATOL_X

pytestmark = pytest.mark.gpu


@pytest.fixture
def slab(monkeypatch):
    monkeypatch.setenv("CCVM_AMD_KERNEL", "slab")


def _describe(kind, b, n, adam=False):
    import ctypes

    from ccvm_amd import _lib

    lib = _lib.load()
    buf = ctypes.create_string_buffer(512)
    solver = {"dl": 0, "mf": 1}.get(kind, 2)
    assert lib.ccvm_describe_launch(solver, b, n, 1 if adam else 0, 0, buf, 512) == 0
    return buf.value.decode()


def _check_against_oracle(kind, n, b, t, adam):
    hp = _ADAMS[adam]
    seed, row_offset = 0x51AB_7E12_3456, 64 + (n % 2)
    traj = _run_engine(kind, n, b, t, hp, seed, row_offset)
    gate = (max(n, 20) / 20.0) ** 0.5
    for name, want in _run_oracle(kind, n, b, t, hp, seed, row_offset):
        got = traj.compact(name).cpu()
        scale = max(1.0, float(want.abs().max()))
        err = float((got - want).abs().max())
        assert err <= ATOL_X * gate * scale, f"{kind} N={n} B={b} {name}: {err:.3e}"
    for name, arr in traj.state.items():  # padding stays zero
        assert float(arr[b:].abs().max() if arr.shape[0] > b else 0.0) == 0.0
        assert float(arr[:, n:].abs().max() if arr.shape[1] > n else 0.0) == 0.0


@pytest.mark.parametrize("kind,n,b,t,adam", [
    # the bench workloads of this path
    ("dl", 1000, 1, 40, None), ("dl", 1000, 8, 40, None), ("dl", 1000, 32, 40, None), ("pl", 2000, 32, 16, None),
    # every K (the multiples of 128 from 384 to 2048); ragged columns; B not a multiple of 4; odd shard starts
    ("langevin", 257, 3, 30, None), ("mf", 500, 32, 30, None), ("dl", 512, 5, 24, None),
    ("pl", 513, 9, 24, None), ("mf", 700, 31, 20, "second_moment"), ("dl", 768, 16, 20, None),
    ("langevin", 769, 2, 20, "add_assign"), ("mf", 1000, 32, 20, None), ("pl", 1024, 7, 16, "first_moment_only"),
    ("langevin", 1025, 4, 16, None), ("mf", 1200, 12, 12, None), ("dl", 1280, 6, 12, None),
    ("pl", 1300, 8, 12, "second_moment"), ("dl", 1536, 4, 10, None), ("mf", 1537, 5, 10, "add_assign"),
    ("langevin", 1400, 6, 10, None), ("dl", 1600, 3, 8, None), ("mf", 1900, 8, 8, None),  # K = 1408, 1664, 1920
    ("langevin", 2000, 8, 10, None), ("dl", 2048, 4, 8, None), ("mf", 2048, 3, 8, None),
    # several row groups per cluster (8 ... 32 rows), several clusters per XCD, clusters spread over the XCDs
    ("langevin", 500, 64, 20, None), ("langevin", 500, 256, 12, None), ("dl", 500, 128, 12, None),
    ("mf", 500, 128, 12, "second_moment"), ("dl", 1000, 64, 12, None), ("dl", 1000, 128, 8, None),
    ("langevin", 1000, 128, 10, "second_moment"), ("mf", 1000, 100, 10, None), ("dl", 700, 128, 8, None),
    ("langevin", 1200, 32, 8, None), ("dl", 1500, 16, 8, None), ("pl", 2000, 17, 8, None),
    ("langevin", 300, 128, 16, None), ("dl", 300, 64, 16, None),
    # more than 16 rows per cluster: the blocks stream through the two-slot ring, a lane owns two pairs of rows
    ("langevin", 1000, 256, 8, None), ("dl", 1000, 200, 6, None), ("mf", 500, 400, 8, "second_moment"),
    ("dl", 2000, 64, 5, None), ("pl", 2000, 100, 5, "add_assign"), ("mf", 1000, 250, 6, None), ("dl", 700, 300, 5, None),
])
def test_slab_kernel_matches_oracle(slab, kind, n, b, t, adam):
    assert "slab_kernel" in _describe(kind, b, n, adam is not None)
    _check_against_oracle(kind, n, b, t, adam)


@pytest.mark.parametrize("cgrp", [1, 2, 4, 8])
@pytest.mark.parametrize("kind,n,b", [("dl", 600, 6), ("mf", 1000, 4), ("pl", 1100, 9), ("langevin", 300, 20)])
def test_every_member_width(slab, monkeypatch, cgrp, kind, n, b):
    """4, 8, 16 and 32 columns per member (1, 2, 4, 8 column groups per MFMA: a different split of the sixteen blocks
    between k residues and columns, a different depth of the wavefront reduction), forced."""
    monkeypatch.setenv("CCVM_AMD_SLAB_CGRP", str(cgrp))
    d = _describe(kind, b, n)
    if "slab_kernel" not in d:
        pytest.skip(f"no {4 * cgrp}-column plan for N={n}, B={b}")
    assert f"x {4 * cgrp} columns" in d
    _check_against_oracle(kind, n, b, 16, None)


def test_slab_kernel_is_what_ran_and_is_the_default_for_small_batches(monkeypatch):
    """Default policy: small batches above N = 256 take the slab kernel; it differs from the tile kernel in summation
    order only (close, not bit-identical), and CCVM_AMD_KERNEL=noslab gives the previous paths back."""
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    for kind, n, b in (("dl", 1000, 1), ("dl", 1000, 32), ("pl", 2000, 32), ("mf", 500, 32), ("langevin", 700, 8)):
        assert "slab_kernel" in _describe(kind, b, n), (kind, n, b)
    for kind, n, b in (("dl", 1000, 1000), ("pl", 2000, 512), ("mf", 500, 1000), ("dl", 100, 8), ("langevin", 256, 4)):
        assert "slab_kernel" not in _describe(kind, b, n), (kind, n, b)
    a = _run_engine("langevin", 1000, 8, 20, None, 77, 0).compact("c").cpu()
    monkeypatch.setenv("CCVM_AMD_KERNEL", "noslab")
    assert "slab_kernel" not in _describe("langevin", 8, 1000)
    b_ = _run_engine("langevin", 1000, 8, 20, None, 77, 0).compact("c").cpu()
    monkeypatch.setenv("CCVM_AMD_KERNEL", "tile")
    c = _run_engine("langevin", 1000, 8, 20, None, 77, 0).compact("c").cpu()
    assert torch.equal(b_, c)
    assert not torch.equal(a, b_) and float((a - b_).abs().max()) <= 1e-4


@pytest.mark.parametrize("kind,n,b", [("mf", 1000, 24), ("pl", 2000, 12), ("dl", 1000, 32), ("langevin", 500, 40),
                                      ("langevin", 1000, 200)])
def test_slab_chunking_is_exact_and_sharding_is_exact_at_equal_member_width(slab, monkeypatch, kind, n, b):
    """Chunked launches (evolution sampling, replay staging) reproduce the one-launch run bit for bit; so do batch
    shards whenever they run with the same member width (the summation order of a column's contraction depends on
    the member width and K only, never on the batch or on how rows are grouped into clusters)."""
    import re

    t = 30
    adam = None if kind == "dl" else _ADAMS["add_assign"]
    monkeypatch.setenv("CCVM_AMD_SLAB_CGRP", re.search(r"slab_kernel<\d, (\d)", _describe(kind, b, n)).group(1))
    whole = _run_engine(kind, n, b, t, adam, 99, 0)
    parts = _run_engine(kind, n, b, t, adam, 99, 0, chunks=[1, 7, 2, 20])
    for name in whole.state:
        assert torch.equal(whole.compact(name), parts.compact(name)), name
    cut = 5  # an odd first global row for the second shard
    lo = _run_engine(kind, n, cut, t, adam, 99, 0)
    hi = _run_engine(kind, n, b - cut, t, adam, 99, cut)
    for name in whole.state:
        assert torch.equal(whole.compact(name), torch.cat([lo.compact(name), hi.compact(name)])), name


@pytest.mark.parametrize("kind,post,n,b", [("mf", None, 1000, 10), ("langevin", "adam", 600, 7), ("pl", "grad-descent", 2000, 4),
                                           ("dl", None, 1000, 32), ("dl", "adam", 500, 3)])
def test_slab_replay_mode_through_the_public_api(slab, kind, post, n, b):
    """Replay noise (torch's CPU stream in the reference's order) through Solver.__call__ and the fused finalize,
    against the oracle's solve_* on the same seed."""
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS, synthetic_instance
    from oracle import ccvm_oracle as oracle

    t = 40
    cls = {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver, "pl": PumpedLangevinSolver}[kind]
    solver = cls(device="cpu", batch_size=b)
    solver.noise_mode = "replay"
    inst = synthetic_instance(n)
    inst.optimal_sol = 1.0
    p = dict(EXAMPLE_PARAMS[kind], iterations=t)
    solver.parameter_key = {n: p}
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    torch.manual_seed(31)
    sol = solver(instance=inst, post_processor=post)
    q, v, f = inst.q_matrix, inst.v_vector, float(inst.scaled_by)
    common = dict(scaled_by=f, optimal_value=1.0, post_processor=post)
    torch.manual_seed(31)
    if kind == "dl":
        ref = oracle.solve_dl(q, v, b, t, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"], g=0.05, S=1,
                              **common)
    elif kind == "mf":
        ref = oracle.solve_mf(q, v, b, t, p["pump"], p["dt"], p["j"], p["feedback_scale"], p["S"], g=0.01, **common)
    elif kind == "langevin":
        ref = oracle.solve_langevin(q, v, b, t, p["dt"], p["sigma"], p["feedback_scale"], p["S"], **common)
    else:
        ref = oracle.solve_pl(q, v, b, t, p["pump"], p["dt"], p["sigma"], p["feedback_scale"], p["S"], **common)
    gate = (n / 20.0) ** 0.5
    assert float((sol.variables["problem_variables"] - ref["problem_variables"]).abs().max()) <= 5e-4 * gate
    want = ref["objective_values"]
    assert float((sol.objective_values - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_slab_per_variable_saturation(slab):
    """S as a 1-D tensor of length N (mf_solver.py:834-839, langevin_solver.py:630-635): the row-scaled copy of Q in the
    members' registers, 1 / S_j and the clamp per column in the owners' update -- against the tile kernel."""
    import os

    from ccvm_amd import engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv

    for kind, n, b in (("mf", 600, 6), ("langevin", 1000, 9)):
        q, v, _ = scaled_qv(n, kind)
        g = torch.Generator().manual_seed(5)
        sat = torch.rand(n, generator=g) * 2.0 + 0.25
        out = {}
        for kernel in ("slab", "tile"):
            os.environ["CCVM_AMD_KERNEL"] = kernel
            p = dict(EXAMPLE_PARAMS[kind], S=sat)
            if kind == "mf":
                p["g"] = 0.01
            else:
                p["use_pump"] = False
            traj = engine.Trajectories(engine.DeviceProblem(q, v), b, kind, 24, p, (0.0, 1.0),
                                       engine.NoiseSpec(mode="fused", seed=8, row_offset=3))
            traj.advance(24)
            out[kernel] = {k: traj.compact(k).cpu() for k in traj.state}
        os.environ["CCVM_AMD_KERNEL"] = "slab"
        for k in out["tile"]:
            scale = max(1.0, float(out["tile"][k].abs().max()))
            gate = ATOL_X * (n / 20.0) ** 0.5 * scale  # the stated tolerance: two summation orders of the contraction
            assert float((out["slab"][k] - out["tile"][k]).abs().max()) <= gate, (kind, k)


def test_slab_long_trajectory_under_uneven_load(slab):
    """1000 steps of a chip-wide spread cluster and of XCD-local clusters while a second stream keeps other kernels on
    the chip (uneven load is what exposes a wrong hand-off): still equal to the undisturbed run bit for bit."""
    for kind, n, b in (("pl", 2000, 8), ("dl", 1000, 32)):
        t = 1000
        quiet = _run_engine(kind, n, b, t, None, 5, 0).compact("c").cpu()
        side = torch.cuda.Stream()
        busy = torch.randn((2048, 2048), device="cuda")
        with torch.cuda.stream(side):
            for _ in range(40):
                busy = busy @ busy * 1e-3
        traj = _run_engine(kind, n, b, t, None, 5, 0, chunks=[200] * 5)
        with torch.cuda.stream(side):
            for _ in range(40):
                busy = busy @ busy * 1e-3
        loaded = traj.compact("c").cpu()
        side.synchronize()
        assert torch.equal(quiet, loaded)
        assert bool(torch.isfinite(loaded).all())


@pytest.mark.parametrize("kind,adam,n,b", [("dl", None, 1000, 32), ("pl", None, 2000, 8), ("mf", "second_moment", 500, 64),
                                           ("langevin", None, 1000, 256)])
def test_slab_soak_is_deterministic(slab, kind, adam, n, b):
    """20 000 steps, twice (4096-step launches / ragged chunks): bit-identical and finite.  A single stale or torn
    exchange read anywhere would show here."""
    t = 20000
    first = _run_engine(kind, n, b, t, _ADAMS[adam], 4242, 0)
    second = _run_engine(kind, n, b, t, _ADAMS[adam], 4242, 0, chunks=[4096, 1, 4095, 5000, 6808])
    for name in first.state:
        x, y = first.compact(name), second.compact(name)
        assert bool(torch.isfinite(x).all()), name
        assert torch.equal(x, y), name


@pytest.mark.parametrize("kind,n,b,t", [("dl", 1000, 32, 3), ("pl", 2000, 8, 600)])
def test_slab_time_out_falls_back_to_the_tile_kernel(monkeypatch, kind, n, b, t):
    """Fault injection (CCVM_AMD_FAULT=cluster_drop: the launch omits its last 8 workgroups, so a member of every
    cluster never publishes): the peers' bounded waits give up (~1 s), the launch ends with the status word set, the
    engine restores its snapshot and repeats the steps on the per-step tile kernel with a warning.  Second case: clusters
    over two XCDs in a launch that calibrates its fetch delay."""
    monkeypatch.setenv("CCVM_AMD_KERNEL", "tile")
    want = {k: v for k, v in _state_of(_run_engine(kind, n, b, t, None, 21, 0)).items()}
    monkeypatch.setenv("CCVM_AMD_KERNEL", "slab")
    monkeypatch.setenv("CCVM_AMD_FAULT", "cluster_drop")
    traj = _run_engine(kind, n, b, t, None, 21, 0)
    with pytest.warns(RuntimeWarning, match="timed out waiting for its workgroups"):
        got = _state_of(traj)
    assert traj.fallbacks == 1
    for k in want:
        assert torch.equal(got[k], want[k]), k
    monkeypatch.delenv("CCVM_AMD_FAULT")
    good = _run_engine(kind, n, b, t, None, 21, 0)
    assert bool(torch.isfinite(good.compact("c")).all()) and good.fallbacks == 0


def test_time_out_cool_down(monkeypatch):
    """after a recovered time-out NEW trajectories on the device stay off the cluster / slab kernels for
    $CCVM_AMD_EXCHANGE_COOLDOWN seconds instead of each paying its own ~1 s bounded wait; then they come back."""
    import time

    from ccvm_amd import engine

    monkeypatch.setenv("CCVM_AMD_KERNEL", "slab")
    monkeypatch.setenv("CCVM_AMD_EXCHANGE_COOLDOWN", "4")
    monkeypatch.setenv("CCVM_AMD_FAULT", "cluster_drop")
    first = _run_engine("dl", 1000, 32, 3, None, 21, 0)
    with pytest.warns(RuntimeWarning, match="timed out waiting for its workgroups"):
        first.check()
    t0 = time.monotonic()
    second = _run_engine("dl", 1000, 32, 3, None, 21, 0)   # the fault is still armed, yet no exchange kernel runs
    assert second.no_exchange and second._snap is None
    second.check()
    assert second.fallbacks == 0 and time.monotonic() - t0 < 1.0
    assert torch.equal(second.compact("c"), first.compact("c"))
    monkeypatch.delenv("CCVM_AMD_FAULT")
    time.sleep(max(0.0, 4.2 - (time.monotonic() - t0)))
    third = _run_engine("dl", 1000, 32, 3, None, 21, 0)
    assert not third.no_exchange and third._snap is not None  # back on the slab kernel
    third.check()
    assert third.fallbacks == 0
    engine._exchange_blocked_until.clear()


def _state_of(traj):
    return {k: traj.compact(k).cpu() for k in traj.state}


def _reference_cases():
    from golden_util import golden

    return [(tag, name) for tag in ("synthetic300", "synthetic600") for name in golden(tag).cases]


@pytest.mark.parametrize("tag,case", _reference_cases())
def test_slab_kernel_matches_the_reference_itself(slab, tag, case):
    """The reference's OWN output on dense N = 300 / 600 instances at batch 12 (tests/golden/synthetic300 / 600, every
    solver and Adam variant; made by make_golden.py from the reference in the build container) against the slab
    kernel through the public API in replay mode."""
    import math

    from golden_util import check_noise_checksum, golden
    from test_gpu_parity import ATOL_OBJ, ATOL_X, _run_case

    g = golden(tag)
    meta = g.cases[case]
    n = g.instance["problem_size"]
    check_noise_checksum(meta, n, meta["batch"])
    assert "slab_kernel" in _describe("dl", meta["batch"], n)
    sol = _run_case(g, meta)
    gate = math.sqrt(max(n, 20) / 20.0)
    for field in g.fields(case):
        want = g.out(case, field)
        got = sol.objective_values if field == "objective_values" else sol.variables[field]
        scale = max(1.0, float(want.abs().max()) / (150.0 if field == "objective_values" else 1.0))
        tol = (ATOL_OBJ if field == "objective_values" else ATOL_X) * gate * scale
        err = float((got.cpu() - want).abs().max())
        assert err <= tol, f"{tag}/{case}/{field}: max abs err {err:.3e} > {tol:.1e}"
    assert abs(sol.best_objective_value - meta["best_objective_value"]) <= 1e-5 * abs(meta["best_objective_value"]) + 1e-4
