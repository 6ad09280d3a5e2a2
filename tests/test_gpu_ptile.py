"""GPU tests of the persistent streamed-Q tile kernel (ccvm_amd/csrc/ccvm_ptile.h): the 32 x 128 tile grid kept
resident over a chunk of steps, the new state handed over between the workgroups of a row block INSIDE the launch
(drained write-through stores, a flag line per row block, L1-bypassing LDS-DMA loads), the own K range written straight
from the epilogue's registers into the LDS ring.  The reference's loop being replaced: dl_solver.py:523-553.

What can go wrong is visibility (a reader streaming a peer's columns before they landed, or from a stale cache line), so
the central check is BITWISE: a chunk run as one launch must equal the same chunk run one step per launch, where the
kernel boundary guarantees visibility and the arithmetic is the same -- on an idle chip and with another stream keeping
the chip busy.  Then every word against the oracle, chunking and sharding; the reference's own goldens at N = 1000 run through this
family in tests/test_gpu_thick_goldens.py (family "ptile")."""
import pytest
import torch

from test_gpu_cluster import _ADAMS, _run_engine
from test_gpu_slab import _check_against_oracle, _describe

pytestmark = pytest.mark.gpu


@pytest.fixture
def ptile(monkeypatch):
    monkeypatch.setenv("CCVM_AMD_KERNEL", "ptile")
    monkeypatch.setenv("CCVM_AMD_KS", "1")  # (forced on grids the default policy gives to the finer tile shapes too)


_SOLVER = {"dl": 0, "mf": 1, "langevin": 2, "pl": 2}


def _state(traj):
    traj.check()
    assert traj.fallbacks == 0
    return {k: traj.compact(k).clone() for k in traj.state}


def test_ptile_is_the_default_for_full_grids_of_wide_tiles(monkeypatch):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    for kind, n, b, adam in (("dl", 1000, 1000, False), ("pl", 2000, 512, False), ("langevin", 1000, 1000, False),
                             ("dl", 1024, 1024, False), ("mf", 900, 800, False), ("mf", 1000, 1000, False),
                             ("langevin", 1000, 1000, True), ("mf", 1000, 900, True)):
        assert "ptile_kernel" in _describe(kind, b, n, adam), (kind, n, b, _describe(kind, b, n, adam))
    # batches of several rounds that cut into well-filled resident grids: slices of the batch, one after the other
    for kind, n, b, adam, slices in (("dl", 1000, 2000, False, 2), ("dl", 1000, 4000, False, 4), ("pl", 2000, 1000, False, 2),
                                     ("mf", 1000, 2000, True, 2), ("langevin", 1500, 2000, False, 3)):
        assert f"ptile_kernel<{_SOLVER[kind]}, {str(adam).lower()}> {slices} slices" in _describe(kind, b, n, adam), (kind, n, b)
    # not: grids the per-step tile shapes are estimated to serve more than 5 % faster (round 5, measured: DL N = 900,
    # B = 800 on three rounds of 32 x 32 tiles 26.6 us against 31.0 resident; N = 1000, B = 768: 28.8 against 31.2), batches
    # whose slices would leave CUs idle, the cluster kernel's sizes
    for kind, n, b, adam in (("dl", 1000, 1500, False), ("dl", 1000, 1300, False), ("dl", 1000, 512, False),
                             ("dl", 768, 1000, False), ("dl", 1500, 1000, False), ("dl", 1200, 1000, False),
                             ("dl", 900, 800, False), ("dl", 1000, 768, False)):
        assert "ptile_kernel" not in _describe(kind, b, n, adam), (kind, n, b)
    monkeypatch.setenv("CCVM_AMD_KERNEL", "noptile")
    assert "step_kernel" in _describe("dl", 1000, 1000)
    monkeypatch.setenv("CCVM_AMD_KERNEL", "tile")
    assert "step_kernel" in _describe("dl", 1000, 1000)


@pytest.mark.parametrize("busy", [False, True])
@pytest.mark.parametrize("kind,n,b,t,adam", [
    ("dl", 1000, 1000, 60, None), ("pl", 2000, 512, 30, None), ("langevin", 1000, 1000, 40, None),
    ("dl", 1030, 777, 20, None), ("dl", 900, 300, 25, None), ("pl", 3000, 150, 8, None),
    ("mf", 1000, 1000, 40, None), ("mf", 1100, 333, 20, "second_moment"), ("langevin", 1000, 1000, 30, "add_assign"),
    ("pl", 2000, 512, 16, "first_moment_only"), ("mf", 2000, 512, 12, "add_assign")])
def test_one_launch_equals_one_step_per_launch_bit_for_bit(ptile, kind, n, b, t, adam, busy):
    """The hand-over inside the launch against kernel boundaries (same kernel, same arithmetic, one step per launch):
    any stale or early read of a peer's columns shows as a difference.  `busy`: a second stream hammers the memory
    system and the CUs' queues meanwhile (uneven load: the case idle chips hide)."""
    assert "ptile_kernel" in _describe(kind, b, n, adam is not None)
    stepwise = _state(_run_engine(kind, n, b, t, _ADAMS[adam], 9001, 3, chunks=[1] * t))
    noise = None
    if busy:
        side = torch.cuda.Stream()
        scratch = torch.empty((64 * 1024 * 1024,), dtype=torch.float32, device="cuda")
        with torch.cuda.stream(side):
            for _ in range(40):
                scratch.mul_(1.0001)
        noise = (side, scratch)
    whole = _state(_run_engine(kind, n, b, t, _ADAMS[adam], 9001, 3))
    if noise:
        noise[0].synchronize()
    for name in whole:
        assert bool(torch.isfinite(whole[name]).all()), name
        assert torch.equal(whole[name], stepwise[name]), f"{kind} N={n} B={b}: {name} differs"


@pytest.mark.parametrize("kind,n,b,t,adam", [
    ("dl", 1000, 1000, 12, None), ("pl", 2000, 512, 8, None), ("langevin", 1000, 1000, 12, None),
    ("dl", 1001, 999, 6, None), ("pl", 1537, 400, 6, None), ("dl", 800, 290, 10, None), ("langevin", 4000, 200, 3, None),
    ("mf", 1000, 1000, 12, None), ("mf", 1001, 999, 8, "second_moment"), ("mf", 900, 290, 10, "first_moment_only"),
    ("langevin", 1000, 1000, 10, "second_moment"), ("pl", 1537, 400, 6, "add_assign"), ("mf", 2000, 300, 5, "add_assign")])
def test_ptile_matches_oracle(ptile, kind, n, b, t, adam):
    assert "ptile_kernel" in _describe(kind, b, n, adam is not None)
    _check_against_oracle(kind, n, b, t, adam)


@pytest.mark.parametrize("kind,n,b,adam", [("dl", 1000, 1000, None), ("pl", 2000, 512, None), ("langevin", 1100, 500, None),
                                           ("mf", 1000, 1000, None), ("mf", 1000, 1000, "second_moment"),
                                           ("langevin", 1000, 900, "add_assign")])
def test_ptile_chunking_and_sharding_are_exact(ptile, kind, n, b, adam):
    t = 14
    hp = _ADAMS[adam]
    whole = _state(_run_engine(kind, n, b, t, hp, 4242, 0))
    parts = _state(_run_engine(kind, n, b, t, hp, 4242, 0, chunks=[1, 5, 2, 6]))
    odd = _state(_run_engine(kind, n, b, t, hp, 4242, 0, chunks=[3, 3, 3, 5]))
    for name in whole:
        assert torch.equal(whole[name], parts[name]) and torch.equal(whole[name], odd[name]), name
    cut = 357
    lo = _state(_run_engine(kind, n, cut, t, hp, 4242, 0))
    hi = _state(_run_engine(kind, n, b - cut, t, hp, 4242, cut))
    for name in whole:
        assert torch.equal(whole[name][:cut], lo[name]) and torch.equal(whole[name][cut:], hi[name]), name


@pytest.mark.parametrize("kind,n,b,adam,cut", [
    ("dl", 1000, 2000, None, 1024), ("mf", 1000, 2000, "second_moment", 1024), ("pl", 2000, 1000, "add_assign", 512),
    ("langevin", 1500, 2000, None, 672), ("dl", 1001, 2999, None, 1024), ("mf", 1000, 3000, None, 2048)])
@pytest.mark.parametrize("replay", [False, True])
def test_sliced_batches_are_the_rows_of_their_slices(monkeypatch, kind, n, b, adam, cut, replay):
    """Batches of several rounds run as slices of whole row blocks, each a resident grid of its own over the whole
    chunk (default policy).  Rows never meet, so the run must equal, bit for bit: itself one step per launch (every
    slice handed over at kernel boundaries), and separate trajectories of the rows below and above a slice boundary
    (fused generator: the global row index keys the stream; replay: the columns of the unsharded run's blocks)."""
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    assert "slices" in _describe(kind, b, n, adam is not None)
    t, hp = 9, _ADAMS[adam]
    gb = b if replay else None
    whole = _state(_run_engine(kind, n, b, t, hp, 1717, 0, replay_global_batch=gb))
    stepwise = _state(_run_engine(kind, n, b, t, hp, 1717, 0, chunks=[1] * t, replay_global_batch=gb))
    for name in whole:
        assert bool(torch.isfinite(whole[name]).all()), name
        assert torch.equal(whole[name], stepwise[name]), f"{kind} N={n} B={b}: {name} differs from the stepwise run"
    monkeypatch.setenv("CCVM_AMD_KERNEL", "ptile")  # (the parts alone may fall to other shapes by default)
    monkeypatch.setenv("CCVM_AMD_KS", "1")
    lo = _state(_run_engine(kind, n, cut, t, hp, 1717, 0, replay_global_batch=gb))
    hi = _state(_run_engine(kind, n, b - cut, t, hp, 1717, cut, replay_global_batch=gb))
    for name in whole:
        assert torch.equal(whole[name][:cut], lo[name]) and torch.equal(whole[name][cut:], hi[name]), name


@pytest.mark.parametrize("kind,n,b,t,adam", [("dl", 1000, 2000, 5, None), ("mf", 1000, 2000, 5, "second_moment"),
                                             ("pl", 2000, 1000, 4, None), ("dl", 1001, 2999, 3, None)])
def test_sliced_batches_match_oracle(monkeypatch, kind, n, b, t, adam):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    assert "slices" in _describe(kind, b, n, adam is not None)
    _check_against_oracle(kind, n, b, t, adam)


_CUT = [("dl", 1000, 1100, None, 1024), ("mf", 1000, 1500, "second_moment", 1024), ("langevin", 1000, 2500, None, 2048),
        ("pl", 2000, 640, "add_assign", 512), ("dl", 1100, 1000, None, 896), ("mf", 1000, 1200, None, 1024),
        ("dl", 1000, 1030, None, 1024),
        # the cluster kernel's sizes (N <= 512): the rows of the resident clusters, then the rest
        ("langevin", 500, 1100, None, 1024), ("dl", 500, 1100, None, 1024), ("mf", 500, 1100, "second_moment", 1024),
        ("pl", 300, 1600, "add_assign", 1536), ("mf", 500, 2100, None, 2048)]


@pytest.mark.parametrize("kind,n,b,adam,cut", _CUT)
def test_batches_cut_in_two_are_their_parts(monkeypatch, kind, n, b, adam, cut):
    """split_rows (ccvm_abi.hip): a batch that overflows its last resident grid a little runs as two calls, the rows
    of whole resident grids and the rest under its own plan, each with the workspace of its own behind the batch's.
    Bit for bit the two trajectories run separately (the global row index keys the noise), however it is chunked."""
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    d = _describe(kind, b, n, adam is not None)
    assert d.startswith(f"batch cut in two: rows 0-{cut - 1} ccvm::{'ptile' if n > 768 else 'cluster'}_kernel"), d
    t, hp = 11, _ADAMS[adam]
    whole = _state(_run_engine(kind, n, b, t, hp, 606, 5))
    parts = _state(_run_engine(kind, n, b, t, hp, 606, 5, chunks=[1, 4, 1, 5]))
    lo = _state(_run_engine(kind, n, cut, t, hp, 606, 5))
    hi = _state(_run_engine(kind, n, b - cut, t, hp, 606, 5 + cut))
    for name in whole:
        assert bool(torch.isfinite(whole[name]).all()), name
        assert torch.equal(whole[name], parts[name]), f"{kind} N={n} B={b}: {name} depends on the chunks"
        assert torch.equal(whole[name][:cut], lo[name]) and torch.equal(whole[name][cut:], hi[name]), name


@pytest.mark.parametrize("kind,n,b,t,adam", [("dl", 1000, 1100, 6, None), ("mf", 1000, 1500, 4, "second_moment"),
                                             ("pl", 2000, 640, 4, "add_assign"), ("dl", 1100, 1000, 5, None),
                                             ("dl", 500, 1100, 12, None), ("mf", 500, 1100, 12, "second_moment")])
def test_batches_cut_in_two_match_oracle(monkeypatch, kind, n, b, t, adam):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    assert "cut in two" in _describe(kind, b, n, adam is not None)
    _check_against_oracle(kind, n, b, t, adam)


@pytest.mark.parametrize("kind,n,b,adam,cut", [("dl", 1000, 1100, None, 1024), ("mf", 1000, 1200, "second_moment", 1024),
                                               ("langevin", 500, 1100, None, 1024), ("pl", 2000, 640, None, 512)])
def test_cut_batches_in_replay_mode_are_their_parts(monkeypatch, kind, n, b, adam, cut):
    """Replay noise (parity mode) takes the same plan: the parts read their columns of the batch's blocks
    (ccvm_noise::w_ld).  Bit for bit the separately run trajectories of the rows on either side of the cut, fed the
    columns of the unsharded run's blocks."""
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    assert "cut in two" in _describe(kind, b, n, adam is not None)
    t, hp = 7, _ADAMS[adam]
    whole = _state(_run_engine(kind, n, b, t, hp, 12, 0, replay_global_batch=b))
    parts = _state(_run_engine(kind, n, b, t, hp, 12, 0, chunks=[2, 1, 4], replay_global_batch=b))
    lo = _state(_run_engine(kind, n, cut, t, hp, 12, 0, replay_global_batch=b))
    hi = _state(_run_engine(kind, n, b - cut, t, hp, 12, cut, replay_global_batch=b))
    for name in whole:
        assert bool(torch.isfinite(whole[name]).all()), name
        assert torch.equal(whole[name], parts[name]), name
        assert torch.equal(whole[name][:cut], lo[name]) and torch.equal(whole[name][cut:], hi[name]), name


def test_time_out_in_a_part_of_a_cut_batch_recovers_the_whole_batch(monkeypatch):
    """Fault injection in a cut batch: the parts' status words are merged into the batch's, the engine restores its
    snapshot and repeats the steps uncut on the per-step kernel."""
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    monkeypatch.setenv("CCVM_AMD_KERNEL", "nocluster")
    want = _run_engine("dl", 1000, 1100, 5, None, 21, 0).compact("c").cpu()
    monkeypatch.delenv("CCVM_AMD_KERNEL")
    assert "cut in two" in _describe("dl", 1100, 1000)
    monkeypatch.setenv("CCVM_AMD_FAULT", "cluster_drop")
    traj = _run_engine("dl", 1000, 1100, 5, None, 21, 0)
    with pytest.warns(RuntimeWarning, match="timed out waiting for its workgroups"):
        got = traj.compact("c").cpu()
    assert traj.fallbacks == 1 and traj.no_exchange
    assert torch.equal(got, want)
    monkeypatch.delenv("CCVM_AMD_FAULT")
    again = _run_engine("dl", 1000, 1100, 5, None, 21, 0)  # (the parts' status words were cleared by the merge)
    assert again.fallbacks == 0 and not torch.equal(again.compact("c").cpu(), want)


@pytest.mark.parametrize("kind,n,b,adam", [("mf", 1000, 1000, None), ("pl", 2000, 512, "second_moment"),
                                           ("langevin", 1000, 2000, None), ("mf", 1100, 777, "add_assign")])
def test_per_variable_saturation_on_the_persistent_tile_kernel(monkeypatch, kind, n, b, adam):
    """S as a 1-D tensor of length N (mf_solver.py:834-839, langevin_solver.py:630-635): the row-scaled copy of Q
    streamed instead of Q, 1 / S_j and the clamp bound per column in the epilogue -- against the per-step kernel's VS
    instantiation (two summation orders: the stated tolerance), chunking bit-exact, and S_j = S for every column
    against the scalar run (1 / S folded into Q's rows instead of the input map: the stated tolerance)."""
    from ccvm_amd import engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv
    from test_gpu_cluster import ATOL_X

    monkeypatch.setenv("CCVM_AMD_KS", "1")
    q, v, _ = scaled_qv(n, kind)
    prob = engine.DeviceProblem(q, v)
    sat = torch.rand(n, generator=torch.Generator().manual_seed(5)) * 2.0 + 0.25
    hp, t = _ADAMS[adam], 12

    def run(kernel, S, chunks):
        monkeypatch.setenv("CCVM_AMD_KERNEL", kernel)
        p = dict(EXAMPLE_PARAMS[kind], S=S)
        if kind == "mf":
            p["g"] = 0.01
        else:
            p["use_pump"] = kind == "pl"
        traj = engine.Trajectories(prob, b, "mf" if kind == "mf" else "langevin", t, p, (0.0, 1.0),
                                   engine.NoiseSpec(mode="fused", seed=8, row_offset=3), adam=hp)
        assert ("ptile_kernel" in traj_describe(traj)) == (kernel == "ptile")
        for k in chunks:
            traj.advance(k)
        return _state(traj)

    def traj_describe(traj):
        import ctypes
        buf = ctypes.create_string_buffer(1024)
        assert traj.lib.ccvm_describe_launch(traj._SOLVER_ID[traj.kind], traj.b, traj.n, 1 if hp else 0,
                                             1 if traj.s_cols is not None else 0, buf, 1024) == 0
        return buf.value.decode()

    got = run("ptile", sat, [t])
    parts = run("ptile", sat, [1, 4, 1, 6])
    want = run("noptile", sat, [t])
    for name in want:
        assert torch.equal(got[name], parts[name]), name
        scale = max(1.0, float(want[name].abs().max()))
        assert float((got[name] - want[name]).abs().max()) <= ATOL_X * (n / 20.0) ** 0.5 * scale, (kind, name)
    s0 = float(EXAMPLE_PARAMS[kind]["S"])
    flat, scalar = run("ptile", torch.full((n,), s0), [t]), run("ptile", s0, [t])
    for name in scalar:
        assert float((flat[name] - scalar[name]).abs().max()) <= ATOL_X * (n / 20.0) ** 0.5 * max(1.0, float(scalar[name].abs().max())), name


def test_default_policy_is_chunk_invariant_at_the_headline_shape(monkeypatch):
    """A run's result must not depend on how the caller chunks it (evolution sampling, replay-noise staging): one-step
    chunks take the persistent kernel too (the family fixes the summation order)."""
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    whole = _state(_run_engine("dl", 1000, 1000, 21, None, 99, 0))
    parts = _state(_run_engine("dl", 1000, 1000, 21, None, 99, 0, chunks=[1, 1, 6, 1, 12]))
    for name in whole:
        assert torch.equal(whole[name], parts[name]), name


def test_ptile_close_to_the_per_step_kernel_at_the_headline_shape(monkeypatch):
    """Default policy at DL N = 1000, B = 1000 over 300 steps (five launches' worth of hand-overs per row block and
    step): the persistent kernel against the per-step kernel -- same noise, the K order of the contraction rotated."""
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    a = _state(_run_engine("dl", 1000, 1000, 300, None, 77, 0))
    monkeypatch.setenv("CCVM_AMD_KERNEL", "noptile")
    b = _state(_run_engine("dl", 1000, 1000, 300, None, 77, 0))
    for name in a:
        assert not torch.equal(a[name], b[name])
        assert float((a[name] - b[name]).abs().max()) <= 3e-4 * max(1.0, float(b[name].abs().max())), name


def test_ptile_soak_is_deterministic(ptile):
    """6000 steps at the headline shape twice (two launches of 4096 / 1904 steps each, ~48 000 hand-overs per
    workgroup): finite and identical."""
    a = _state(_run_engine("dl", 1000, 1000, 6000, None, 5, 0))
    b = _state(_run_engine("dl", 1000, 1000, 6000, None, 5, 0, chunks=[4096, 1, 1903]))
    for name in a:
        assert bool(torch.isfinite(a[name]).all()) and torch.equal(a[name], b[name]), name


def test_ptile_time_out_falls_back_to_the_per_step_kernel(monkeypatch):
    """Fault injection (CCVM_AMD_FAULT=cluster_drop: the last 8 workgroups leave at once, so their row blocks' peers
    never see their flags): the producer waves' bounded waits give up, the launch ENDS with the status word set, the
    engine restores its snapshot and repeats the steps on the per-step kernel with a warning."""
    monkeypatch.setenv("CCVM_AMD_KERNEL", "noptile")
    want = _run_engine("dl", 1000, 1000, 5, None, 21, 0).compact("c").cpu()
    monkeypatch.setenv("CCVM_AMD_KERNEL", "ptile")
    monkeypatch.setenv("CCVM_AMD_FAULT", "cluster_drop")
    traj = _run_engine("dl", 1000, 1000, 5, None, 21, 0)
    with pytest.warns(RuntimeWarning, match="timed out waiting for its workgroups"):
        got = traj.compact("c").cpu()
    assert traj.fallbacks == 1 and traj.no_exchange
    assert torch.equal(got, want)


def test_chunks_with_library_made_schedule_rows_are_on_the_books(ptile):
    """the library's record of how far a flag area has run must include the chunks whose schedule rows IT made
    (no `schedule` table from the caller).  30 steps that way leave the flag lines at step 30; the same 30 steps again on
    the same workspace WITH the table and CCVM_RUN_FORWARD claimed -- a re-run from step 0 that the promise does not cover
    -- must still get its flag lines set: bit for bit the straight run, no time-out."""
    from ccvm_amd import engine

    want = _state(_run_engine("dl", 1000, 1000, 30, None, 5, 0))
    traj = _run_engine("dl", 1000, 1000, 30, None, 5, 0, chunks=[0])  # (built, not advanced)
    table = traj.cparams.schedule
    traj.cparams.schedule = None
    traj.arm(force=True)
    traj.advance(30)
    assert traj.check(rerun=False, hold=True) is False
    first = {k: engine.unpack(traj.state[k], traj.b, traj.n) for k in traj.state}
    traj.rollback()
    traj.cparams.schedule = table
    traj.advance(30)
    assert traj.check(rerun=False, hold=True) is False and traj.fallbacks == 0
    for name in want:
        assert torch.equal(first[name], want[name]), name
        assert torch.equal(engine.unpack(traj.state[name], traj.b, traj.n), want[name]), name


@pytest.mark.parametrize("kind,n,b", [("dl", 1000, 1000), ("mf", 1000, 1000)])
def test_going_back_on_a_workspace_never_meets_the_flags_of_later_steps(ptile, kind, n, b):
    """Round 5: the flag lines hold absolute step numbers and a reader accepts any number >= the step it waits
    for, so steps RE-RUN on a workspace that already ran later ones would take the old flags for their peers' publications
    -- unless the lines are set to the chunk's first step in front of the launch.  The engine claims CCVM_RUN_FORWARD on
    every call (it saves that launch); the library keeps its own record per flag area and sets the lines whenever a chunk
    does not start behind the last one, whatever the caller claims.  Here: 40 steps, back to the snapshot taken before
    them (Trajectories.rollback: no time-out, the run stays on this kernel), the same 40 steps again in other chunks --
    bit for bit the first pass, and the straight run of a fresh object."""
    want = _state(_run_engine(kind, n, b, 40, None, 5, 0))
    traj = _run_engine(kind, n, b, 40, None, 5, 0, chunks=[0])  # (built, not advanced)
    traj.arm(force=True)
    traj.advance(40)
    assert traj.check(rerun=False, hold=True) is False
    from ccvm_amd import engine

    # (Trajectories.compact verifies, and a verification without `hold` drops the snapshot: unpack the arrays directly)
    logical = lambda: {k: engine.unpack(traj.state[k], traj.b, traj.n) for k in traj.state}
    first = logical()
    for again in ([40], [1, 7, 32], [13, 27]):
        traj.rollback()
        assert traj.step == 0
        for k in again:
            traj.advance(k)
        assert traj.check(rerun=False, hold=True) is False and not traj.no_exchange
        got = logical()
        for name in want:
            assert torch.equal(got[name], first[name]) and torch.equal(got[name], want[name]), (again, name)


def test_sc1_lds_dma_stress(tmp_path):
    """The hand-off form of this kernel -- drained `sc1` stores, `sc1` flag, `sc1` poll, `global_load_lds_dwordx4 ... sc1` --
    under the guide's stress recipe (MI355X_MICROARCH.md: "test every hand-off under UNEVEN load, consumer L1-warm, checking
    every word"; its table of measured forms lists register loads only): tools/ptile_sc1_stress.hip, 256 workgroups each
    producer and consumer of 4 KB payloads, consumers re-reading the slots with plain loads between hand-offs, a bandwidth
    hog on a second stream, pairs inside an XCD and across the fabric.  No wrong word; the control (plain LDS-DMA) must
    show stale words, or the stress would prove nothing.  (profiles/r06_ptile_sc1_stress.txt: the 4096-epoch run.)"""
    import os
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ptile_sc1_stress")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", os.path.join(root, "tools", "ptile_sc1_stress.hip"),
                    "-o", exe], check=True, timeout=300)
    run = subprocess.run([exe, "2048"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "PASSED" in run.stdout, run.stdout[-3000:] + run.stderr[-1000:]
    rows = [ln for ln in run.stdout.splitlines() if "hand-offs" in ln]
    assert len(rows) == 8
    for ln in rows:
        wrong, gave_up = int(re.search(r"(\d+) wrong", ln).group(1)), int(re.search(r"(\d+) give-ups", ln).group(1))
        assert gave_up == 0, ln
        if ln.startswith("CONTROL"):
            assert wrong > 0, ln          # plain LDS-DMA reads this CU's stale L1 lines
        else:
            assert wrong == 0 and "524288 hand-offs" in ln, ln
    assert sum("256 pairs inside an XCD" in ln for ln in rows) == 4 and sum("256 across" in ln for ln in rows) == 4
