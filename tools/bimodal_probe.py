"""Run-to-run spread of a cell's default plan over FRESH trajectories (developer tool): are slow samples of the audit a
property of a launch (placement) or of a run call?   python tools/bimodal_probe.py kind:N:B [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

kind, n, b = sys.argv[1].split(":")
n, b = int(n), int(b)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for rep in range(reps):
    traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0)
    traj.advance(64)
    torch.cuda.synchronize()
    out = []
    for _ in range(4):
        t0 = time.perf_counter()
        traj.advance(512)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 512 * 1e6)
    traj.check()
    del traj
    print(f"{sys.argv[1]} trajectories {rep}: " + " ".join(f"{x:7.2f}" for x in out) + " us/step", flush=True)
