"""bench.py's multi-rank plumbing on CPU (world-size-2 gloo): the ranks AGREE on how the one data collective travels
-- RCCL when every rank's set-up succeeded, host copies over gloo in the same processes when any rank's failed
(`"collective": "gloo-fallback: ..."`) -- and the timed region is repeated, by every rank together, when its steps
turn out invalid (a persistent kernel gave up its bounded wait), so the line never times discarded work."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _probe_raises_on_rank_1(dev, world, timeout_s):
    """Stands in for bench.rccl_group: RCCL's transport set-up fails on ONE rank (the others would sit in the warm-up
    all-reduce until its time-out; here they simply succeed -- the agreement is what is tested)."""
    grp = dist.new_group(backend="gloo")  # group creation is collective: every rank gets this far, as in rccl_group
    if dist.get_rank() == 1:
        raise RuntimeError("hipIpcGetMemHandle: invalid argument\n(stub)")
    return grp


def _probe_ok(dev, world, timeout_s):
    return dist.new_group(backend="gloo")  # a group of its own, as the RCCL one would be


def _worker(rank, world, port, which, queue):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import bench

    try:
        probe = {"fail": _probe_raises_on_rank_1, "ok": _probe_ok}[which]
        coll = bench.init_collectives(rank, world, torch.device("cpu"), share=False, probe=probe, timeout_s=30)
        # the data collective of the bench on whatever was agreed
        obj = torch.full((3,), float(rank)).to(coll["device"])
        parts = [torch.empty_like(obj) for _ in range(world)]
        dist.all_gather(parts, obj, group=coll["group"])
        queue.put((rank, coll["collective"], coll["group"] is None, str(coll["device"]), torch.cat(parts).tolist()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _gather_worker(rank, world, port, which, queue):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import bench

    def broken_all_gather(parts, mine, group=None):
        """The agreed group's all-gather fails on ONE rank; the others believe theirs went through (and hold garbage)."""
        if dist.get_rank() == 1:
            raise RuntimeError("NCCL error: unhandled system error\n(stub)")
        for p in parts:
            p.fill_(-1.0)

    try:
        coll = bench.init_collectives(rank, world, torch.device("cpu"), share=False, probe=_probe_ok, timeout_s=30)
        values = torch.full((3,), float(rank))
        parts, label = bench.gather_values(values, coll, world, all_gather=broken_all_gather if which == "broken" else None)
        queue.put((rank, label, torch.cat(parts).tolist()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("which", ["broken", "fine"])
def test_a_failing_data_collective_costs_a_label_not_the_line(which):
    """bench.gather_values: the agreed group's all-gather raising on one rank AFTER the timed region (RCCL's set-up is
    proven by a one-element all-reduce, not by this call) moves every rank to gloo on host copies; the values are the
    right ones on every rank and the label says what happened."""
    world = 3
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, which, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((queue.get(timeout=180) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    labels = {r[1] for r in results}
    assert len(labels) == 1
    label = labels.pop()
    if which == "broken":
        assert label == "gloo-fallback (the all-gather over RCCL failed: rank 1: RuntimeError: NCCL error: unhandled system error (stub))"
    else:
        assert label == "RCCL"
    assert all(r[2] == [float(r2) for r2 in range(world) for _ in range(3)] for r in results)


@pytest.mark.parametrize("which,world", [("fail", 2), ("ok", 2), ("fail", 8), ("ok", 8)])
def test_ranks_agree_on_the_collective(which, world):
    """(world 8: the control plane of the driver's 8-GPU run -- rendez-vous, agreement, 8-way gather -- which this pool's
    boxes cannot put on one GPU: six processes per card at most)"""
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, which, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((queue.get(timeout=180) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    labels = {r[1] for r in results}
    assert len(labels) == 1  # every rank took the same decision
    label = labels.pop()
    if which == "fail":
        # ONE rank's failure moves every rank to gloo, in the same processes, and the line says why
        assert label.startswith("gloo-fallback: rank 1: RuntimeError: hipIpcGetMemHandle: invalid argument")
        assert "\n" not in label
        assert all(r[2] and r[3] == "cpu" for r in results)
    else:
        assert label == "RCCL" and not any(r[2] for r in results)
    assert all(r[4] == [float(r2) for r2 in range(world) for _ in range(3)] for r in results)


class _FakeEvent:
    def __init__(self, enable_timing=False):
        pass

    def record(self):
        pass

    def elapsed_time(self, other):
        return 1.25


class _FakeTraj:
    """Trajectories whose verification points fail as listed (check then reports a recovered time-out and goes back to
    the snapshot)."""

    def __init__(self, invalid_checks):
        self.invalid, self.step, self.calls, self.rollbacks, self.fallbacks = list(invalid_checks), 0, [], 0, 0
        self._start = None

    def arm(self, force=False):
        self._start = self.step

    def advance(self, n):
        self.calls.append((self.step, n))
        self.step += n

    def check(self, rerun=True, hold=False):
        if not hold:
            return False  # (the final check that drops the snapshot)
        if self.invalid and self.invalid.pop(0):
            self.fallbacks += 1
            self.step = self._start
            return True
        return False

    def rollback(self):
        self.rollbacks += 1
        self.step = self._start


def test_an_invalid_run_starts_over(monkeypatch):
    import bench

    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda dev=None: None)
    barriers = []
    # the first attempt's steps (warm-up or timed: one status word) are invalid
    traj = _FakeTraj(invalid_checks=[True])
    walls, stream_ms, attempts = bench.timed_steps(traj, 5, 20, torch.device("cpu"), lambda: barriers.append(1))
    assert attempts == 2 and stream_ms == [1.25] and len(walls) == 1 and walls[0] >= 0
    assert traj.calls == [(0, 5), (5, 20), (0, 5), (5, 20)]  # warm-up and timed steps again, from the snapshot
    assert traj.step == 25 and len(barriers) == 2  # one opening barrier per timed region

    # valid here, invalid on another rank: this rank starts over WITH it (the barriers must pair up)
    traj = _FakeTraj(invalid_checks=[])
    told = iter([True, False])
    _, _, attempts = bench.timed_steps(traj, 5, 20, torch.device("cpu"), lambda: None,
                                       any_rank=lambda flag: flag or next(told))
    assert attempts == 2 and traj.rollbacks == 1 and traj.calls == [(0, 5), (5, 20), (0, 5), (5, 20)]

    traj = _FakeTraj(invalid_checks=[True, True, True])
    with pytest.raises(SystemExit):
        bench.timed_steps(traj, 5, 20, torch.device("cpu"), lambda: None)


def test_repeated_timed_regions(monkeypatch):
    """R regions of exactly K steps, each behind its own barrier and verified; a region that turns out invalid sends
    the whole sequence (warm-up included) back to the snapshot."""
    import bench

    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda dev=None: None)
    barriers = []
    traj = _FakeTraj(invalid_checks=[])
    walls, stream_ms, attempts = bench.timed_steps(traj, 5, 20, torch.device("cpu"), lambda: barriers.append(1), repeats=9)
    assert attempts == 1 and len(walls) == 9 and stream_ms == [1.25] * 9 and len(barriers) == 9
    assert traj.calls == [(0, 5)] + [(5 + 20 * i, 20) for i in range(9)] and traj.step == 185
    # the third region of the first attempt is invalid
    traj = _FakeTraj(invalid_checks=[False, False, True])
    walls, _, attempts = bench.timed_steps(traj, 5, 20, torch.device("cpu"), lambda: None, repeats=4)
    assert attempts == 2 and len(walls) == 4
    assert traj.calls == [(0, 5), (5, 20), (25, 20), (45, 20), (0, 5), (5, 20), (25, 20), (45, 20), (65, 20)]
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 3.0, 2.0]) == 2.5


class _StubEngineTraj:
    """The state Trajectories.check / rollback / arm work on, without a GPU: a status word on the host, a snapshot that
    is just the step counter."""

    kind = "dl"
    fallbacks = 0
    no_exchange = False

    def __init__(self):
        self.device = torch.device("cpu")
        self.step = 0
        self._snap = None
        self._status = torch.zeros(1, dtype=torch.int32)
        self._runs, self._clean_at = 0, -1  # (run calls so far / their count at the last clean read of the status word)
        self.restored = []

    def _exchange_kernel(self):
        return True

    def _snapshot(self):
        return {"step": self.step}

    def _restore(self, snap):
        self.restored.append(snap["step"])
        self.step = snap["step"]

    def advance(self, n):
        # (what engine.Trajectories.advance does around a run call: re-arm unless the run is off the exchange kernels)
        from ccvm_amd import engine

        engine.Trajectories.arm(self)
        self._runs += 1
        self.step += n


def test_recoveries_on_different_ranks_in_successive_attempts(monkeypatch):
    """a rank that recovered a time-out in attempt 1 (no_exchange from then on: advance never re-arms) must
    still be able to roll back when ANOTHER rank times out in attempt 2 -- check(hold=True) keeps the snapshot on the
    recovery path as well."""
    import contextlib
    import warnings

    import bench
    from ccvm_amd import engine

    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda dev=None: None)
    monkeypatch.setattr(torch.cuda, "device", lambda dev: contextlib.nullcontext())
    monkeypatch.setattr(engine, "_exchange_blocked_until", {})
    traj = _StubEngineTraj()
    for name in ("arm", "rollback", "check"):
        setattr(traj, name, getattr(engine.Trajectories, name).__get__(traj))
    fail_next = [True]   # this rank's first timed region times out

    real_advance = traj.advance

    def advance(n):
        real_advance(n)
        if n == 20 and fail_next and fail_next.pop(0):
            traj._status.fill_(1)

    traj.advance = advance
    other_rank = iter([True, False])  # (asked when this rank's steps were valid) attempt 2: the OTHER rank times out; attempt 3: clean
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        walls, _, attempts = bench.timed_steps(traj, 5, 20, torch.device("cpu"), lambda: None,
                                               any_rank=lambda flag: flag or next(other_rank))
    assert attempts == 3 and len(walls) == 1
    assert traj.restored == [0, 0]       # recovery in attempt 1, roll-back with the other rank in attempt 2
    assert traj.no_exchange and traj.fallbacks == 1 and traj.step == 25 and traj._snap is None
