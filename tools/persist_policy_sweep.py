"""Row-owner kernel (N <= 128): per-step time of every variant the launch policy can pick -- rows in use (RU), K split
(KH), noise producer waves (PW) -- over a grid of (solver, N, B), in ONE process at steady clocks (developer tool).
   python tools/persist_policy_sweep.py [--quick] > gpurun_out/r06/persist_policy.jsonl
Each line: {"kind", "n", "b", "ru", "kh", "pw", "us_per_step", "kernel"}; "policy": true marks the default plan's line.
tools/persist_policy_report.py turns the file into the regret table of profiles/r06_persist_policy.md."""
import json
import os
import re
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

SHAPE = re.compile(r"persist_kernel<[^>]*> grid \d+ x \d+")
STEPS = 4096


def describe(kind, b, n):
    return (SHAPE.search(bench.describe_launch(kind, b, n)) or [""])[0]


def measure(kind, n, b):
    traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 22, 0)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.06:  # leave the idle clocks (a 4096-step launch at N = 20 is 1-2 ms)
        traj.advance(STEPS)
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        traj.advance(STEPS)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    traj.check()
    return best / STEPS * 1e6


def main():
    quick = "--quick" in sys.argv
    kinds = ["dl", "langevin", "mf"]
    small_n = [20, 64] if quick else [8, 16, 20, 32, 48, 64]
    mid_n = [100] if quick else [70, 100, 128]
    batches = [100, 1000, 4000] if quick else [100, 500, 1000, 1500, 2000, 3000, 4000, 8000]
    for var in ("CCVM_AMD_PERSIST_RU", "CCVM_AMD_PERSIST_KH", "CCVM_AMD_PERSIST_PW"):
        os.environ.pop(var, None)
    for kind in kinds:
        for n in small_n + mid_n:
            for b in batches:
                variants = [({}, True)]
                if n <= 64:
                    variants += [({"CCVM_AMD_PERSIST_RU": str(ru), "CCVM_AMD_PERSIST_PW": str(pw)}, False)
                                 for ru in (2, 4) for pw in (0, 1)]
                else:
                    variants += [({"CCVM_AMD_PERSIST_KH": "1", "CCVM_AMD_PERSIST_PW": "0"}, False),
                                 ({"CCVM_AMD_PERSIST_KH": "1", "CCVM_AMD_PERSIST_RU": "2", "CCVM_AMD_PERSIST_PW": "0"}, False),
                                 ({"CCVM_AMD_PERSIST_KH": "2", "CCVM_AMD_PERSIST_PW": "0"}, False),
                                 ({"CCVM_AMD_PERSIST_KH": "2", "CCVM_AMD_PERSIST_PW": "1"}, False)]
                for env, is_policy in variants:
                    os.environ.update(env)
                    kernel = describe(kind, b, n)
                    us = measure(kind, n, b)
                    for var in env:
                        os.environ.pop(var, None)
                    m = re.search(r"<\d, \w+, \d+, \d+, \d+, (\d+), (\d+)(, 1)?>", kernel)
                    print(json.dumps({"kind": kind, "n": n, "b": b, "ru": int(m.group(1)), "kh": int(m.group(2)),
                                      "pw": 1 if m.group(3) else 0, "policy": is_policy, "us_per_step": round(us, 4),
                                      "kernel": kernel}), flush=True)


if __name__ == "__main__":
    main()
