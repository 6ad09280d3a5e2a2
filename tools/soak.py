"""Long fused-noise runs with odd chunking (developer tool): stays finite, per-step time holds."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
CASES = (("dl", 20, 1000, 100000), ("mf", 64, 500, 50000), ("pl", 300, 512, 20000), ("dl", 200, 256, 30000),
         # the slab kernel: clusters inside XCDs, over two XCDs, streamed row groups
         ("dl", 1000, 32, 100000), ("pl", 2000, 32, 50000), ("langevin", 1000, 128, 30000), ("mf", 500, 64, 100000),
         # the tile kernel's 32 x 32 and multi-round 32 x 64 shapes
         ("dl", 1000, 256, 20000), ("langevin", 1500, 1000, 5000),
         # the row-owner kernel's K split: two and four waves side by side
         ("mf", 100, 1000, 100000), ("dl", 256, 1000, 30000), ("langevin", 200, 500, 50000),
         # round 4: the persistent tile kernel, slices of the batch, batches cut in two
         ("dl", 1000, 1000, 30000), ("mf", 1000, 2000, 10000), ("langevin", 1000, 1100, 20000), ("dl", 500, 1100, 20000),
         ("pl", 2000, 640, 8000))
for kind, n, b, t in CASES:
    traj, q, v = bench.make_trajectories(kind, n, b, t, 0)
    t0 = time.time()
    done = 0
    while done < t:
        k = min(7777, t - done)   # odd chunking on purpose
        traj.advance(k); done += k
    torch.cuda.synchronize()
    dt = time.time() - t0
    ok = all(bool(torch.isfinite(a).all()) for a in traj.state.values())
    name = "mu" if kind == "mf" else "c"
    x = traj.compact(name)
    assert traj.fallbacks == 0
    print(f"{kind} N={n} B={b} T={t}: {dt:.2f} s ({dt/t*1e6:.2f} us/step) finite={ok} |x|max={float(x.abs().max()):.4f} mean={float(x.mean()):.4f}", flush=True)
