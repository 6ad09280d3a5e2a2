"""One-off sweep of the cpu_baseline leg of bench.py over torch thread counts (developer tool): states how
many host cores the reported CPU number uses and that more do not help.
    python tools/cpu_thread_sweep.py [workload] > gpurun_out/cpu_thread_sweep.md"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

workload = sys.argv[1] if len(sys.argv) > 1 else "dl_n1000_b1000"
kind, n, b = bench.WORKLOADS[workload]
visible = len(os.sched_getaffinity(0))
print(f"# CPU baseline (torch-CPU oracle = the reference's op sequence) vs torch threads: {workload}\n")
print(f"Host: {bench.cpu_model()}, {visible} cores visible to the process (a 1-GPU box is a share of the host).\n")
print("| torch threads | ms/step | row-steps/s |")
print("|---|---|---|")
for t in (1, 2, 4, 8, 16, 32, 64, 128):
    if t > visible:
        break
    out = bench.cpu_baseline(kind, n, b, 100000, budget_s=4.0, threads=t)
    print(f"| {t} | {b / out['value'] * 1e3:.2f} | {out['value']:.3e} |", flush=True)
