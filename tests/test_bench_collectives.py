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


@pytest.mark.parametrize("which", ["fail", "ok"])
def test_ranks_agree_on_the_collective(which):
    world = 2
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
    assert all(r[4] == [0.0, 0.0, 0.0, 1.0, 1.0, 1.0] for r in results)


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
    elapsed, stream_ms, attempts = bench.timed_steps(traj, 5, 20, torch.device("cpu"), lambda: barriers.append(1))
    assert attempts == 2 and stream_ms == 1.25 and elapsed >= 0
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
