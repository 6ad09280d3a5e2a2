"""Host threads driving the engine concurrently, one stream each, on ONE device.

Kernels whose workgroups wait for each other (persistent tile, column-cluster, column-slab) are deadlock-free only
while their grid is resident; two such grids from two streams can interleave on the CUs, and each then sits out its
bounded wait, falls back to the per-step kernel and puts the device into its cool-down.  The engine chains run calls
that launch such kernels per device across streams (engine._exchange_chain): both runs must come out bit for bit what
they are alone, without a single recovered time-out.  Also: a Trajectories built on one stream and advanced on another
waits for its construction (schedule table, zero-filled arrays)."""
import threading

import pytest
import torch

from test_gpu_cluster import _run_engine

pytestmark = pytest.mark.gpu


def _state(traj):
    return {k: traj.compact(k).clone() for k in traj.state}


@pytest.mark.parametrize("a,b", [(("dl", 1000, 1000), ("langevin", 500, 1000)),     # persistent tile + column-cluster
                                 (("mf", 1000, 1000), ("dl", 1000, 1000))])          # two resident 256-workgroup grids
def test_two_streams_take_turns_on_the_exchange_kernels(monkeypatch, a, b):
    from ccvm_amd import engine

    monkeypatch.setenv("CCVM_AMD_EXCHANGE_COOLDOWN", "0")
    steps, chunks = 120, [7, 33, 20, 60]
    alone = [_state(_run_engine(kind, n, rows, steps, None, 99 + i, 0, chunks=chunks)) for i, (kind, n, rows) in enumerate((a, b))]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    trajs, errors = [None, None], []

    def work(i, spec):
        try:
            kind, n, rows = spec
            with torch.cuda.stream(streams[i]):
                traj = _run_engine(kind, n, rows, steps, None, 99 + i, 0, chunks=[0])
                assert traj._waits is None or traj._waits  # (asked at the first run call)
                for k in chunks:
                    traj.advance(k)
                trajs[i] = traj
        except Exception as exc:  # noqa: BLE001 -- reported by the main thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i, spec)) for i, spec in enumerate((a, b))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            got = _state(trajs[i])
        assert trajs[i].fallbacks == 0 and not trajs[i].no_exchange, i   # nobody sat out a bounded wait
        for name in alone[i]:
            assert torch.equal(got[name], alone[i][name]), (i, name)
    chain = engine._exchange_chain[trajs[0].device.index]
    assert len(chain["streams"]) >= 2 and chain["event"] is not None       # the calls really were chained


def test_a_run_built_on_one_stream_and_advanced_on_another_waits_for_its_construction():
    want = _state(_run_engine("dl", 1000, 1000, 30, None, 5, 0))
    build, run = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(build):
        # (something long in front of the construction on ITS stream: the schedule kernel and the zero fills queue behind it)
        x = torch.randn(4096, 4096, device="cuda")
        for _ in range(20):
            x = x @ x * 1e-3
        traj = _run_engine("dl", 1000, 1000, 30, None, 5, 0, chunks=[0])
    with torch.cuda.stream(run):
        traj.advance(30)
        got = _state(traj)
    for name in want:
        assert torch.equal(got[name], want[name]), name
