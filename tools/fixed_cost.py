"""Fixed cost of a run call (developer tool): wall and HIP-event time of traj.advance(k) for several k at the headline
shape, a + b k fitted -- a = what a call pays regardless of its steps (schedule kernel, launch, first step, sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

kind, n, b = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("dl", 1000, 1000)
traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 22, 0)
for _ in range(40):
    traj.advance(256)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); ev1.record(); torch.cuda.synchronize()
rows = []
for k in (1, 2, 5, 10, 20, 40, 80, 160, 320, 640):
    best_w, best_e = 1e9, 1e9
    for _ in range(7):
        traj.advance(64)  # keep the clocks up, then a synchronise like the bench's
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        traj.advance(k)
        ev1.record()
        torch.cuda.synchronize()
        best_w = min(best_w, time.perf_counter() - t0)
        best_e = min(best_e, ev0.elapsed_time(ev1) * 1e-3)
    rows.append((k, best_w * 1e6, best_e * 1e6))
    print(f"k={k:4d}: wall {best_w * 1e6:9.1f} us ({best_w * 1e6 / k:7.2f} per step)   events {best_e * 1e6:9.1f} us ({best_e * 1e6 / k:7.2f} per step)")
(k1, w1, e1), (k2, w2, e2) = rows[-3], rows[-1]
bw, be = (w2 - w1) / (k2 - k1), (e2 - e1) / (k2 - k1)
print(f"slope: wall {bw:.3f} us/step, events {be:.3f} us/step; fixed cost: wall {w1 - bw * k1:.1f} us, events {e1 - be * k1:.1f} us")
