"""Where does the time of a cut batch go?  (developer tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def run(kind, n, b, steps, reps=3):
    traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0)
    traj.advance(64); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        traj.advance(steps)
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, t2 - t0)
        print(f"  {kind}:{n}:{b} steps={steps}: host enqueue {1e6*(t1-t0)/steps:7.2f} us/step, wall {1e6*(t2-t0)/steps:7.2f}, events {1e3*e0.elapsed_time(e1)/steps:7.2f}", flush=True)
    traj.check()

for steps in (100, 1000, 4000):
    run("dl", 1000, 1500, steps)
os.environ["CCVM_AMD_SPLIT"] = "0"
run("dl", 1000, 1500, 1000)
run("dl", 1000, 476, 1000)
