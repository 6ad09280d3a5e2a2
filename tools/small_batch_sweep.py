"""Per-step time of small batches above N = 256: default policy (column-slab kernel where it has a plan) against
CCVM_AMD_KERNEL=noslab (what ran before: cluster kernel up to N = 768, else the per-step tile kernel).  Developer tool;
writes gpurun_out/small_batch_sweep.jsonl, one JSON line per case.
   python tools/small_batch_sweep.py [kind:N:B ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cases = sys.argv[1:] or [f"{k}:{n}:{b}" for k in ("dl", "langevin") for n in (300, 500, 700, 1000, 1500, 2000)
                         for b in (1, 4, 8, 16, 32, 64, 128, 256)]
steps = 2048
out = open(os.path.join("gpurun_out", "small_batch_sweep.jsonl"), "w")
for case in cases:
    kind, n, b = case.split(":")
    n, b = int(n), int(b)
    rec = {"kind": kind, "n": n, "b": b}
    for mode in ("auto", "noslab"):
        if mode == "auto":
            os.environ.pop("CCVM_AMD_KERNEL", None)
        else:
            os.environ["CCVM_AMD_KERNEL"] = "noslab"
        traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0)
        rec[mode + "_kernel"] = bench.describe_launch(kind, b, n)
        traj.advance(steps)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            traj.advance(steps)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        traj.check()
        rec[mode + "_us"] = best / steps * 1e6
    os.environ.pop("CCVM_AMD_KERNEL", None)
    print(json.dumps(rec), file=out, flush=True)
    print(f"{case:20s} auto {rec['auto_us']:8.3f}  noslab {rec['noslab_us']:8.3f}  {rec['auto_kernel'][:70]}", flush=True)
