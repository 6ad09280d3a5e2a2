"""bench.py's rank launcher (`python bench.py --gpus N` without a torch.distributed environment) with a stub rank
script: environment wiring, return-code propagation with a diagnostic JSON line, time-out kill.  No GPU, no torch
collectives: the launcher must work before anything touches the GPU."""
import json
import os
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STUB = textwrap.dedent("""
    import json, os, sys, time
    out = sys.argv[1]
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
            "HSA_ENABLE_IPC_MODE_LEGACY")
    with open(os.path.join(out, "rank%s.json" % os.environ["RANK"]), "w") as fh:
        json.dump({k: os.environ.get(k) for k in keys} | {"argv": sys.argv[2:]}, fh)
    mode = sys.argv[2]
    if mode == "fail" and os.environ["RANK"] == "1":
        sys.exit(3)
    if mode in ("fail", "hang"):
        time.sleep(60)
    sys.exit(0)
""")


@pytest.fixture
def stub(tmp_path, monkeypatch):
    path = tmp_path / "rank_stub.py"
    path.write_text(STUB)
    monkeypatch.setenv("CCVM_BENCH_SHARE_GPU", "1")  # no GPU-count check: there is no GPU here
    return str(path), str(tmp_path)


def test_launcher_wires_the_distributed_environment(stub, capsys):
    import bench

    script, out = stub
    assert bench.launch_ranks(3, [out, "ok", "--steps", "5"], script=script, timeout=60) == 0
    seen = [json.load(open(os.path.join(out, f"rank{r}.json"))) for r in range(3)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"] and [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2"]
    assert all(s["WORLD_SIZE"] == "3" and s["LOCAL_WORLD_SIZE"] == "3" and s["MASTER_ADDR"] == "127.0.0.1" for s in seen)
    assert len({s["MASTER_PORT"] for s in seen}) == 1 and int(seen[0]["MASTER_PORT"]) > 0
    assert all(s["argv"] == ["ok", "--steps", "5"] for s in seen)
    # dmabuf IPC for RCCL's intra-node transports on this pool (bench.IPC_ENV); an explicit caller value wins
    assert all(s["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") for s in seen)
    assert capsys.readouterr().out.strip() == ""  # nothing but the ranks' own output on success


def test_a_dead_rank_ends_the_run_with_its_return_code_and_a_diagnostic_line(stub, capsys):
    import bench

    script, out = stub
    t0 = time.time()
    rc = bench.launch_ranks(3, [out, "fail"], script=script, timeout=60)
    assert rc != 0 and time.time() - t0 < 30  # the surviving ranks (asleep for 60 s) were killed, not waited for
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert "rank 1 exited with return code 3" in line["error"] and line["return_codes"][1] == 3 and line["n_gpus"] == 3


def test_ranks_that_hang_are_killed_at_the_time_out(stub, capsys):
    import bench

    script, out = stub
    t0 = time.time()
    assert bench.launch_ranks(2, [out, "hang"], script=script, timeout=2) == 124
    assert time.time() - t0 < 30
    assert "time-out" in json.loads(capsys.readouterr().out.strip().splitlines()[-1])["error"]


def test_gpu_count_without_the_hip_runtime(monkeypatch):
    import bench

    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    assert bench.visible_gpu_count() == 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    launcher = src[src.index("def launch_ranks"):src.index("def rccl_group")]
    assert "torch.cuda" not in launcher  # the launcher never asks the HIP runtime anything


def test_strong_scaling_shards_cover_the_batch_and_are_never_empty():
    """--global-batch G over N ranks: G // N rows each, the first G % N ranks one more (ceil-sized shards left
    trailing ranks empty or negative, e.g. 9 rows on 8 ranks: ADVICE r3)."""
    import bench

    for total, world in ((9, 8), (8000, 8), (4096, 8), (4096, 3), (7, 7), (1000, 1), (13, 4)):
        shards = [bench.shard_rows(total, world, r) for r in range(world)]
        assert shards[0][0] == 0 and sum(n for _, n in shards) == total
        assert all(n >= 1 for _, n in shards) and max(n for _, n in shards) - min(n for _, n in shards) <= 1
        assert all(shards[r][0] + shards[r][1] == shards[r + 1][0] for r in range(world - 1))
