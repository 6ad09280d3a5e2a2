"""One-off sweep of the cpu_baseline leg of bench.py over torch thread counts (developer tool): states how
many host cores the reported CPU number uses and that more do not help.
    python tools/cpu_thread_sweep.py [workload] > gpurun_out/cpu_thread_sweep.md"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

workloads = sys.argv[1:] or ["dl_n1000_b1000"]
visible = len(os.sched_getaffinity(0))
print("# CPU baseline (torch-CPU oracle = the reference's op sequence) vs torch threads\n")
print(f"Host: {bench.cpu_model()}, {visible} cores visible to the process (a 1-GPU box is a share of the host).  Each point: "
      "20 warm-up steps, then >= 50 steps and >= 1.5 s (or 8 s) timed -- `bench.cpu_baseline`.\n")
for workload in workloads:
    kind, n, b = bench.WORKLOADS[workload]
    instance = bench.WORKLOAD_EXTRAS.get(workload, {}).get("instance")
    print(f"\n## {workload}\n\n| torch threads | ms/step | row-steps/s | timed steps |\n|---|---|---|---|")
    for t in (1, 2, 4, 8, 16, 32, 64, 128):
        if t > visible:
            break
        out = bench.cpu_baseline(kind, n, b, 100000, budget_s=8.0, threads=t, instance=instance, min_steps=50, min_seconds=1.5)
        print(f"| {t} | {b / out['value'] * 1e3:.3f} | {out['value']:.3e} | {out['timed_steps']} |", flush=True)
