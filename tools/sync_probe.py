"""Where the 20-step driver run's wall time goes beyond its kernels (developer tool): bench.py's flow, single sample
per process-like trial, with timestamps around each host call of the timed region."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench

dev = torch.device("cuda", 0)
def trial(pre_events):
    traj, _, _ = bench.make_trajectories("dl", 1000, 1000, 25, 0)
    scratch, _, _ = bench.make_trajectories("dl", 1000, 1000, 1 << 20, 0, seed=2)
    t = time.perf_counter()
    while time.perf_counter() - t < 0.15:
        scratch.advance(256); torch.cuda.synchronize(dev)
    del scratch
    traj.advance(5)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if pre_events:
        ev0.record(); ev1.record(); torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record()
    t1 = time.perf_counter()
    traj.advance(20)
    t2 = time.perf_counter()
    ev1.record()
    t3 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t4 = time.perf_counter()
    return [(b - a) * 1e6 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t0, t4))] + [ev0.elapsed_time(ev1) * 1e3]
for pre in (False, True, False, True, False, True):
    r = trial(pre)
    print("pre-created events" if pre else "fresh events      ", "ev0.record %.0f us, advance(20) host %.0f, ev1.record %.0f, sync %.0f | wall %.1f = %.2f us/step, gpu %.2f us/step"
          % (r[0], r[1], r[2], r[3], r[4], r[4] / 20, r[5] / 20))
