"""Host time of one run call through the engine (developer tool): traj.advance(k) enqueued back to back without a
synchronisation, cProfile of the same -- what a short call pays before its kernel starts."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

kind, n, b = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("dl", 1000, 1000)
traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0)
traj.advance(64)
torch.cuda.synchronize()
for k in (1, 20):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            traj.advance(k)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"advance({k}): {1e6 * (t1 - t0) / 100:.2f} us of host time per call", flush=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    traj.advance(1)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
