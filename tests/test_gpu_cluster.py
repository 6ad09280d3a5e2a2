"""GPU tests of the column-cluster persistent kernel (ccvm_amd/csrc/ccvm_cluster.h): 256 < N <= 512,
DL / MF / Langevin / pumped Langevin and the Adam variants, whole chunks of steps in one launch with the
cluster's workgroups exchanging the GEMM input as {value, tag} packets through sc1 stores / loads.

Every word of every trajectory is compared with the oracle (fused noise through oracle/noise_ref.py),
over enough steps that a single stale exchange read would be amplified into a visible difference, on
grids from one cluster to many more clusters than the chip holds at once (one launch per round of resident clusters)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL_X = 5e-4

_ADAMS = {
    None: None,
    "second_moment": {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False},
    "add_assign": {"alpha": 0.01, "beta1": 0.8, "beta2": 0.99, "add_assign": True},
    "first_moment_only": {"alpha": 0.002, "beta1": 0.9, "beta2": 1.0, "add_assign": True},
}


@pytest.fixture
def cluster(monkeypatch):
    monkeypatch.setenv("CCVM_AMD_KERNEL", "cluster")


def _describe(kind, b, n, adam=False):
    import ctypes

    from ccvm_amd import _lib

    buf = ctypes.create_string_buffer(1024)
    assert _lib.load().ccvm_describe_launch({"dl": 0, "mf": 1}.get(kind, 2), b, n, 1 if adam else 0, 0, buf, 1024) == 0
    return buf.value.decode()


def _run_engine(kind, n, b, t, adam, seed, row_offset, chunks=None, replay_global_batch=None):
    """`replay_global_batch`: replay mode instead of the fused generator -- rows [row_offset, row_offset + b) of the
    blocks a run of that many rows draws from a generator seeded with `seed`."""
    from ccvm_amd import engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv

    q, v, _ = scaled_qv(n, kind)
    p = dict(EXAMPLE_PARAMS[kind])
    if replay_global_batch:
        noise = engine.NoiseSpec(mode="replay", row_offset=row_offset, global_batch=replay_global_batch,
                                 generator=torch.Generator().manual_seed(seed))
    else:
        noise = engine.NoiseSpec(mode="fused", seed=seed, row_offset=row_offset)
    prob = engine.DeviceProblem(q, v)
    if kind == "dl":
        traj = engine.Trajectories(prob, b, "dl", t, dict(p, g=0.05), (0.0, 1.0), noise)
    elif kind == "mf":
        traj = engine.Trajectories(prob, b, "mf", t, dict(p, g=0.01), (0.0, 1.0), noise, adam=adam)
    else:
        traj = engine.Trajectories(prob, b, "langevin", t, dict(p, use_pump=kind == "pl"), (0.0, 1.0), noise, adam=adam)
    for k in (chunks or [t]):
        traj.advance(k)
    return traj


def _run_oracle(kind, n, b, t, adam, seed, row_offset):
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv
    from oracle import ccvm_oracle as oracle
    from oracle.noise_ref import FusedNoise

    q, v, _ = scaled_qv(n, kind)
    p = dict(EXAMPLE_PARAMS[kind])
    ref_noise = FusedNoise(seed, row_offset, single=kind != "dl")
    if kind == "dl":
        c, s = oracle.dl_loop(q, v, b, t, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"], 0.05, (0.0, 1.0),
                              True, ref_noise)
        return [("c", c), ("s", s)]
    if kind == "mf":
        mu, mu_tilde, sigma = oracle.mf_loop(q, v, b, t, p["pump"], p["dt"], p["j"], p["feedback_scale"], p["S"], 0.01,
                                             (0.0, 1.0), True, adam, ref_noise)
        return [("mu", mu), ("sigma", sigma), ("mu_tilde", mu_tilde)]
    if kind == "pl":
        c = oracle.pl_loop(q, v, b, t, p["pump"], p["dt"], p["sigma"], p["feedback_scale"], p["S"], (0.0, 1.0), True,
                           adam, ref_noise)
    else:
        c = oracle.langevin_loop(q, v, b, t, p["dt"], p["sigma"], p["feedback_scale"], p["S"], (0.0, 1.0), adam,
                                 ref_noise)
    return [("c", c)]


@pytest.mark.parametrize("kind,n,b,t,adam", [
    # BASELINE config 3 shapes
    ("langevin", 500, 1000, 40, None), ("mf", 500, 1000, 40, None),
    ("mf", 500, 1000, 25, "second_moment"), ("langevin", 500, 1000, 25, "add_assign"),
    # every cluster size G = 5..8 and both K (384, 512), ragged batches and columns, odd shard starts
    ("pl", 257, 33, 30, None), ("mf", 320, 100, 24, None), ("langevin", 321, 31, 24, "first_moment_only"),
    ("pl", 384, 70, 20, "second_moment"), ("mf", 385, 64, 20, "add_assign"), ("langevin", 448, 129, 16, None),
    ("pl", 449, 32, 16, None), ("mf", 512, 256, 12, "first_moment_only"), ("pl", 512, 1, 12, None),
    # more workgroups than the chip holds at once: clusters run in dispatch order
    ("pl", 300, 5000, 8, None), ("mf", 500, 3000, 6, None),
    # K = 640 / 768: three row sets per cluster, Q's k >= 512 in registers
    ("langevin", 513, 100, 20, None), ("pl", 576, 1000, 24, None), ("langevin", 640, 47, 16, None),
    ("pl", 641, 96, 16, None), ("langevin", 700, 1000, 20, None), ("pl", 768, 130, 12, None),
    ("dl", 513, 49, 16, None), ("dl", 640, 1000, 20, None), ("dl", 704, 100, 12, None), ("dl", 768, 1000, 16, None),
    ("mf", 640, 1000, 16, None), ("langevin", 640, 64, 12, "second_moment"), ("mf", 768, 100, 12, "add_assign"),
    # K = 640 in rounds of 24 clusters
    ("pl", 600, 2000, 6, None), ("dl", 640, 2100, 5, None),
    # 11-12 members, more clusters than fit XCD by XCD: spread over the XCDs (B = 1000: 21 clusters x 12 = 252)
    ("langevin", 768, 1000, 24, None), ("dl", 700, 1000, 16, None), ("mf", 768, 1000, 12, "second_moment"),
    ("mf", 513, 33, 12, "first_moment_only"), ("pl", 700, 768, 10, "second_moment"),
    # DL: two exchanged planes (c, s), 2 K / 128 chunks per phase
    ("dl", 500, 1000, 40, None), ("dl", 257, 33, 30, None), ("dl", 320, 100, 24, None), ("dl", 384, 70, 20, None),
    ("dl", 385, 64, 20, None), ("dl", 449, 129, 16, None), ("dl", 512, 1, 12, None), ("dl", 300, 5000, 8, None),
])
def test_cluster_kernel_matches_oracle(cluster, kind, n, b, t, adam):
    hp = _ADAMS[adam]
    seed, row_offset = 0xC1A5_7E12_3456, 128 + (n % 2)
    traj = _run_engine(kind, n, b, t, hp, seed, row_offset)
    gate = (max(n, 20) / 20.0) ** 0.5
    for name, want in _run_oracle(kind, n, b, t, hp, seed, row_offset):
        got = traj.compact(name).cpu()
        scale = max(1.0, float(want.abs().max()))
        err = float((got - want).abs().max())
        assert err <= ATOL_X * gate * scale, f"{kind} N={n} {name}: {err:.3e}"
    for name, arr in traj.state.items():  # padding stays zero
        assert float(arr[b:].abs().max() if arr.shape[0] > b else 0.0) == 0.0
        assert float(arr[:, n:].abs().max() if arr.shape[1] > n else 0.0) == 0.0


@pytest.mark.parametrize("kind,n,b,t,adam", [("pl", 300, 96, 20, None), ("dl", 320, 100, 16, None), ("mf", 448, 70, 16, "add_assign"),
                                              ("mf", 576, 200, 12, None), ("dl", 700, 96, 10, None), ("langevin", 641, 33, 12, "second_moment")])
def test_half_chunk_variant_against_the_full_kernel(cluster, monkeypatch, kind, n, b, t, adam):
    """N mod 128 in 1 .. 64 (an odd number of members) runs cluster_kernel_half, which leaves out the all-padding second
    half of every plane's last chunk; CCVM_AMD_CLUSTER_HALF=0 keeps the full kernel on the same shapes.  Both match the
    oracle, each other up to summation order, and the launch description names the one that runs."""
    import ctypes

    from ccvm_amd import _lib

    def describe():
        buf = ctypes.create_string_buffer(1024)
        assert _lib.load().ccvm_describe_launch({"dl": 0, "mf": 1}.get(kind, 2), b, n, 1 if adam else 0, 0, buf, 1024) == 0
        return buf.value.decode()

    hp = _ADAMS[adam]
    seed, row_offset = 0xAB5_1234, 3
    assert "cluster_kernel_half" in describe()
    half = _run_engine(kind, n, b, t, hp, seed, row_offset)
    monkeypatch.setenv("CCVM_AMD_CLUSTER_HALF", "0")
    assert "cluster_kernel<" in describe() or "cluster_kernel_2sets<" in describe()
    full = _run_engine(kind, n, b, t, hp, seed, row_offset)
    gate = (max(n, 20) / 20.0) ** 0.5
    for name, want in _run_oracle(kind, n, b, t, hp, seed, row_offset):
        scale = max(1.0, float(want.abs().max()))
        for got in (half.compact(name).cpu(), full.compact(name).cpu()):
            assert float((got - want).abs().max()) <= ATOL_X * gate * scale, f"{kind} N={n} {name}"
        assert float((half.compact(name).cpu() - full.compact(name).cpu()).abs().max()) <= 2e-4 * gate * scale


@pytest.mark.parametrize("kind,n,b,t,adam", [("langevin", 640, 500, 16, None), ("pl", 576, 700, 12, "second_moment"), ("dl", 704, 300, 10, None),
                                              ("mf", 768, 512, 10, "add_assign"), ("dl", 768, 672, 8, None), ("mf", 600, 33, 12, None)])
def test_two_and_three_row_sets_above_k_512_are_the_same_numbers(cluster, monkeypatch, kind, n, b, t, adam):
    """K = 640 / 768 runs clusters of 32 rows (two row sets: two phases per step) where they fit the chip and clusters of
    48 (three) where the batch needs them; CCVM_AMD_CLUSTER_SETS forces either.  A row's contraction does not depend on
    how rows are grouped: both give the same bits, and both match the oracle."""
    import ctypes

    from ccvm_amd import _lib

    def describe():
        buf = ctypes.create_string_buffer(1024)
        assert _lib.load().ccvm_describe_launch({"dl": 0, "mf": 1}.get(kind, 2), b, n, 1 if adam else 0, 0, buf, 1024) == 0
        return buf.value.decode()

    hp = _ADAMS[adam]
    seed, row_offset = 0x5E75_0002, 5
    assert "_2sets<" in describe()  # (every case here fits the chip in clusters of 32 rows)
    two = _run_engine(kind, n, b, t, hp, seed, row_offset)
    monkeypatch.setenv("CCVM_AMD_CLUSTER_SETS", "3")
    assert "_2sets<" not in describe() and "cluster_kernel" in describe()
    three = _run_engine(kind, n, b, t, hp, seed, row_offset)
    for name in two.state:
        assert torch.equal(two.compact(name), three.compact(name)), name
    gate = (max(n, 20) / 20.0) ** 0.5
    for name, want in _run_oracle(kind, n, b, t, hp, seed, row_offset):
        scale = max(1.0, float(want.abs().max()))
        assert float((two.compact(name).cpu() - want).abs().max()) <= ATOL_X * gate * scale, f"{kind} N={n} {name}"


def test_cluster_kernel_is_what_ran(cluster):
    """The path under test is the cluster kernel: it differs from the tile kernel in summation order only
    (close, not bit-identical), and CCVM_AMD_KERNEL=nocluster gives the tile kernel's bits back."""
    import os

    a = _run_engine("langevin", 400, 96, 20, None, 77, 0).compact("c").cpu()
    os.environ["CCVM_AMD_KERNEL"] = "nocluster"
    b_ = _run_engine("langevin", 400, 96, 20, None, 77, 0).compact("c").cpu()
    os.environ["CCVM_AMD_KERNEL"] = "tile"
    c = _run_engine("langevin", 400, 96, 20, None, 77, 0).compact("c").cpu()
    os.environ["CCVM_AMD_KERNEL"] = "cluster"
    assert torch.equal(b_, c)
    assert not torch.equal(a, b_) and float((a - b_).abs().max()) <= 1e-4


@pytest.mark.parametrize("kind,n,b", [("mf", 300, 96), ("pl", 300, 96), ("dl", 300, 96), ("pl", 700, 96), ("dl", 600, 96),
                                      ("pl", 768, 1000)])
def test_cluster_chunking_and_sharding_are_exact(cluster, kind, n, b):
    """Chunked launches (evolution sampling, replay staging) and batch shards reproduce the one-launch,
    unsharded run bit for bit (the last case: the whole batch runs spread over the XCDs, its shards XCD by XCD)."""
    t = 30
    adam = None if (kind == "dl" or n > 512) else _ADAMS["add_assign"]
    whole = _run_engine(kind, n, b, t, adam, 99, 0)
    parts = _run_engine(kind, n, b, t, adam, 99, 0, chunks=[1, 7, 2, 20])
    for name in whole.state:
        assert torch.equal(whole.compact(name), parts.compact(name)), name
    lo = _run_engine(kind, n, 40, t, adam, 99, 0)
    hi = _run_engine(kind, n, b - 40, t, adam, 99, 40)
    for name in whole.state:
        assert torch.equal(whole.compact(name), torch.cat([lo.compact(name), hi.compact(name)])), name


@pytest.mark.parametrize("kind,post,n", [("mf", None, 300), ("langevin", "adam", 300), ("pl", "grad-descent", 300),
                                         ("dl", None, 300), ("dl", "adam", 300), ("mf", "adam", 600), ("pl", None, 640),
                                         ("dl", None, 700)])
def test_cluster_replay_mode_through_the_public_api(cluster, kind, post, n):
    """Replay noise (torch's CPU stream in the reference's order) through Solver.__call__ and the fused
    finalize, against the oracle's solve_* on the same seed."""
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS, synthetic_instance
    from oracle import ccvm_oracle as oracle

    b, t = 70, 60
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
    assert float((sol.variables["problem_variables"] - ref["problem_variables"]).abs().max()) <= 5e-4
    want = ref["objective_values"]
    assert float((sol.objective_values - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_cluster_long_trajectory_under_uneven_load(cluster):
    """500 steps at the BASELINE shape while a second stream keeps other kernels on the chip (uneven load
    is what exposes a wrong hand-off): still equal to the undisturbed run bit for bit."""
    n, b, t = 500, 1000, 500
    quiet = _run_engine("pl", n, b, t, None, 5, 0).compact("c").cpu()
    side = torch.cuda.Stream()
    busy = torch.randn((2048, 2048), device="cuda")
    stop = False
    traj = None
    with torch.cuda.stream(side):
        for _ in range(40):
            busy = busy @ busy * 1e-3
    traj = _run_engine("pl", n, b, t, None, 5, 0, chunks=[100] * 5)
    with torch.cuda.stream(side):
        for _ in range(40):
            busy = busy @ busy * 1e-3
    loaded = traj.compact("c").cpu()
    side.synchronize()
    assert torch.equal(quiet, loaded)
    assert bool(torch.isfinite(loaded).all())


def test_status_word_is_checked_at_synchronisation_points(cluster):
    """A kernel-side failure (the cluster path giving up a bounded wait) is reported through the run's status
    word in the workspace (ccvm_status_offset); the engine raises at its next synchronisation point."""
    from ccvm_amd import _lib, engine

    traj = _run_engine("langevin", 300, 64, 5, None, 3, 0)
    traj.check()                                     # a normal run leaves it at 0
    assert traj._status is not None and int(traj._status.cpu().view(torch.int32).item()) == 0
    traj._status.view(torch.int32)[0] = 1            # what a timed-out workgroup stores ...
    traj._runs += 1                                  # ... during a run call (the word is read again only behind one)
    with pytest.raises(_lib.EngineError, match="timed out .* no snapshot"):
        traj.compact("c")
    with pytest.raises(_lib.EngineError):
        traj.score("c", 0.5, 1.0)
    dl = engine.Trajectories(traj.p, 8, "dl", 3, {"pump": 2.0, "dt": 0.001, "noise_ratio": 2.0, "feedback_scale": 1.0,
                                                   "g": 0.05}, (0.0, 1.0), engine.NoiseSpec(mode="fused", seed=1))
    assert dl._status is not None                    # every solver's workspace carries the status word


def test_bounded_waits_give_up_instead_of_hanging(monkeypatch):
    """Fault injection (CCVM_AMD_FAULT=cluster_drop: the launch omits its last 8 workgroups, so the last member of
    the last clusters never publishes): the peers' fetch waves exhaust their bounded retries (~1 s), the workgroups
    leave, the launch ENDS with the status word set.  At its next synchronisation point the engine puts the state
    back to the snapshot it took before the launch, repeats the steps on the per-step tile kernel (a fresh launch in
    the same process) and warns instead of raising: the result is the `nocluster` run bit for bit."""
    monkeypatch.setenv("CCVM_AMD_KERNEL", "nocluster")
    want = _run_engine("langevin", 500, 1000, 3, None, 21, 0).compact("c").cpu()
    monkeypatch.setenv("CCVM_AMD_KERNEL", "cluster")
    monkeypatch.setenv("CCVM_AMD_FAULT", "cluster_drop")
    traj = _run_engine("langevin", 500, 1000, 3, None, 21, 0)
    with pytest.warns(RuntimeWarning, match="timed out waiting for its workgroups"):
        got = traj.compact("c").cpu()
    assert traj.fallbacks == 1 and traj.no_exchange
    assert torch.equal(got, want)
    traj.advance(0)
    assert torch.equal(traj.compact("c").cpu(), want)  # verified state: no second recovery, no warning
    assert traj.fallbacks == 1
    monkeypatch.delenv("CCVM_AMD_FAULT")
    good = _run_engine("langevin", 500, 1000, 3, None, 21, 0)
    assert bool(torch.isfinite(good.compact("c")).all()) and good.fallbacks == 0


@pytest.mark.parametrize("kind,n,b,family", [("dl", 1000, 1000, "ptile"), ("mf", 500, 1000, "cluster"),
                                             ("langevin", 500, 1000, "cluster"), ("dl", 1000, 32, "slab")])
def test_a_dropped_workgroup_costs_milliseconds(monkeypatch, kind, n, b, family):
    """the bound of a cross-workgroup wait is a multiple of the STEP (50 estimated steps, at least 20 ms:
    ccvm_abi.hip spin_ticks), not the 0.5-1.2 s of rounds 2-5.  Fault injection at the BASELINE shapes, end to end --
    launch with 8 workgroups missing, give-up, status word to the host, snapshot restored, the 20 steps again on the
    per-step kernel, verified: the faulty launch ends at its 20 ms bound (HIP events: 20.1-20.7 ms), the repeated steps
    take < 1 ms more.  The host's wall clock is reported, not asserted: on this pool's boxes a blocked synchronisation sometimes
    returns ~85 ms after the GPU is done, whatever the bound (profiles/r06_fault_cost.txt: 5.07 ms on the GPU, 86.9 ms
    wall) -- ROCr's interrupt wake-up, nothing a kernel decides."""
    import time

    monkeypatch.setenv("CCVM_AMD_KERNEL", "nocluster")
    want = _run_engine(kind, n, b, 20, None, 21, 0)      # (also warms the per-step kernels the recovery runs on)
    name = "mu" if kind == "mf" else "c"
    want = want.compact(name).cpu()
    monkeypatch.setenv("CCVM_AMD_KERNEL", family)
    _run_engine(kind, n, b, 20, None, 22, 0).check()      # (and the exchange kernel itself, fault-free)
    monkeypatch.setenv("CCVM_AMD_FAULT", "cluster_drop")
    traj = _run_engine(kind, n, b, 20, None, 21, 0, chunks=[0])  # built, armed by advance below
    traj.arm(force=True)                   # (the snapshot: not part of the fault's cost)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    ev[0].record()
    traj.advance(20)
    ev[1].record()                         # the faulty launch has ended: its waves gave up
    with pytest.warns(RuntimeWarning, match="timed out waiting for its workgroups"):
        assert traj.check() is True       # status word -> restore -> the 20 steps again on the per-step kernel -> verified
    ev[2].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    faulty_ms, total_ms = ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[2])
    assert traj.fallbacks == 1 and torch.equal(traj.compact(name).cpu(), want)
    print(f"{family}: faulty launch {faulty_ms:.2f} ms, to the end of the repeated steps {total_ms:.2f} ms on the GPU; {wall * 1e3:.1f} ms wall")
    assert 15.0 < faulty_ms < 25.0, f"{family}: the faulty launch took {faulty_ms:.2f} ms (bound: 20 ms)"
    # (the second interval holds the host's part -- status word to the host, the restore, the repeated steps' launches --
    # and with it the wake-up artefact above: asserted against the old cliff, 0.5-1.2 s, not against the GPU's 21 ms)
    assert total_ms < 200.0, f"{family}: a dropped workgroup cost {total_ms:.1f} ms from the faulty launch to the repeated steps' end"


def test_a_batch_of_several_rounds_is_one_launch_per_round(monkeypatch):
    """Round 6: a batch of more clusters than the chip holds runs as one launch per round of resident clusters
    (ccvm_cluster.h: launch_cluster, ClusterArgs::cluster0) -- no progress argument rests on dispatch order any more.
    Langevin N = 500, B = 4000 = 125 clusters in four launches of 32: every row equals the same rows run as four
    separate batches of one round each (shards are bit-exact), and the oracle."""
    monkeypatch.setenv("CCVM_AMD_KERNEL", "cluster")
    assert "x 4 launches of at most 32 clusters (125 clusters of 8 workgroups)" in _describe("langevin", 4000, 500)
    whole = _run_engine("langevin", 500, 4000, 12, None, 9, 0)
    got = whole.compact("c")
    for first in range(0, 4000, 1024):
        rows = min(1024, 4000 - first)
        part = _run_engine("langevin", 500, rows, 12, None, 9, first)
        assert torch.equal(got[first:first + rows], part.compact("c")), first
    assert whole.fallbacks == 0


def test_time_out_recovery_in_replay_mode_and_under_sampling(monkeypatch, tmp_path):
    """The same fault through the public API with replayed host noise (the snapshot holds the generator's state) and
    with evolution sampling (every flush of the sample ring is a verification point; the samples since the previous
    flush are taken again): solution and samples equal the `nocluster` run's."""
    from ccvm_amd.solvers import LangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS, synthetic_instance

    def solve():
        solver = LangevinSolver(device="cpu", batch_size=64)
        solver.noise_mode = "replay"
        inst = synthetic_instance(300)
        inst.optimal_sol = 1.0
        solver.parameter_key = {300: dict(EXAMPLE_PARAMS["langevin"], iterations=12)}
        inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
        torch.manual_seed(5)
        sol = solver(instance=inst, evolution_step_size=3, evolution_file=str(tmp_path / "evo.txt"))
        return sol, solver.c_sample.clone()

    monkeypatch.setenv("CCVM_AMD_KERNEL", "nocluster")
    want, want_samples = solve()
    monkeypatch.setenv("CCVM_AMD_KERNEL", "cluster")
    monkeypatch.setenv("CCVM_AMD_FAULT", "cluster_drop")
    with pytest.warns(RuntimeWarning, match="timed out waiting for its workgroups"):
        got, got_samples = solve()
    assert torch.equal(got.variables["problem_variables"], want.variables["problem_variables"])
    assert torch.equal(got.objective_values, want.objective_values)
    assert torch.equal(got_samples, want_samples)


@pytest.mark.parametrize("kind", ["dl", "mf", "langevin"])
def test_time_out_recovery_before_the_state_is_clamped_or_copied(monkeypatch, kind):
    """the solvers verify (and recover) BEFORE they clamp the state or hand copies to the caller.  The
    fault through the public API with device="cpu" and NO evolution sampling: DL's c is clamped after the recovery
    (dl_solver.py:567), and the host copies of every variable are those of the repeated steps -- everything equals
    the `nocluster` run bit for bit."""
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS, synthetic_instance

    def solve():
        cls = {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver}[kind]
        solver = cls(device="cpu", batch_size=64)
        solver.noise_seed = 77
        inst = synthetic_instance(300)
        inst.optimal_sol = 1.0
        solver.parameter_key = {300: dict(EXAMPLE_PARAMS[kind], iterations=40)}
        inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
        return solver(instance=inst)

    monkeypatch.setenv("CCVM_AMD_KERNEL", "nocluster")
    want = solve()
    monkeypatch.setenv("CCVM_AMD_KERNEL", "cluster")
    monkeypatch.setenv("CCVM_AMD_FAULT", "cluster_drop")
    with pytest.warns(RuntimeWarning, match="timed out waiting for its workgroups"):
        got = solve()
    assert set(got.variables) == set(want.variables)
    for name in want.variables:
        assert got.variables[name].device.type == "cpu"
        assert torch.equal(got.variables[name], want.variables[name]), name
    assert torch.equal(got.objective_values, want.objective_values)
    if kind == "dl":  # most amplitudes end ON the clamp: an unclamped c after a recovery would show here
        assert float(got.variables["problem_variables"].abs().max()) <= 1.0


@pytest.mark.parametrize("kind,adam,n,b", [("langevin", None, 500, 1000), ("mf", "second_moment", 500, 1000), ("dl", None, 500, 1000),
                                           ("langevin", None, 768, 1000), ("dl", None, 640, 1000), ("pl", None, 700, 1000),
                                           # K > 512 on two row sets: XCD by XCD, spread over the XCDs, in two rounds; the half-chunk variant
                                           ("mf", None, 640, 512), ("langevin", "add_assign", 768, 672), ("dl", None, 704, 500),
                                           ("pl", None, 640, 1500), ("langevin", None, 300, 1000)])
def test_cluster_soak_is_deterministic(cluster, kind, adam, n, b):
    """20 000 steps at the BASELINE config-3 shape (and at K = 640 / 768), twice (one in 4096-step launches, one in
    ragged chunks): bit-identical and finite.  The exchange is the only cross-workgroup traffic in the engine: a single
    stale or torn read anywhere in 10^10 exchanged words would show here."""
    t = 20000
    first = _run_engine(kind, n, b, t, _ADAMS[adam], 4242, 0)
    second = _run_engine(kind, n, b, t, _ADAMS[adam], 4242, 0, chunks=[4096, 1, 4095, 5000, 6808])
    for name in first.state:
        x, y = first.compact(name), second.compact(name)
        assert bool(torch.isfinite(x).all()), name
        assert torch.equal(x, y), name


def test_trajectories_do_not_interact_at_full_size(cluster):
    """Size-independent property at a BASELINE shape: batch rows are independent, so running rows [0, B) and
    the same rows in two halves (keyed by row_offset) give the same trajectories -- here with the halves on
    DIFFERENT kernel paths' grids (500 rows = 16 clusters vs 1000 rows = 32), bit for bit."""
    n, t = 500, 60
    whole = _run_engine("pl", n, 1000, t, None, 11, 0).compact("c")
    lo = _run_engine("pl", n, 500, t, None, 11, 0).compact("c")
    hi = _run_engine("pl", n, 500, t, None, 11, 500).compact("c")
    assert torch.equal(whole, torch.cat([lo, hi]))


def _reference_cases():
    from golden_util import golden

    return [(tag, name) for tag in ("synthetic300", "synthetic600") for name in golden(tag).cases]


@pytest.mark.parametrize("tag,case", _reference_cases())
def test_cluster_kernel_matches_the_reference_itself(cluster, tag, case):
    """The reference's OWN output (tests/golden/synthetic300 / synthetic600: every solver and Adam variant on dense
    N = 300 / 600 instances, made by make_golden.py --only-cluster-n from the reference in the build container)
    against the cluster kernel through the public API in replay mode: K = 384 with two row sets, K = 640 with three
    and the panel's k >= 512 in registers."""
    import math

    from golden_util import check_noise_checksum, golden
    from test_gpu_parity import ATOL_OBJ, ATOL_X, _run_case

    g = golden(tag)
    meta = g.cases[case]
    n = g.instance["problem_size"]
    check_noise_checksum(meta, n, meta["batch"])
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
