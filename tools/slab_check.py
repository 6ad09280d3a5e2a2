"""Column-slab kernel against the per-step tile kernel on the same fused noise, and its time per step
(developer tool).   python tools/slab_check.py [kind:N:B[:cgrp[:rg]] ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cases = sys.argv[1:] or ["langevin:1000:4", "dl:1000:4", "mf:1000:4"]
steps = int(os.environ.get("SLAB_STEPS", "200"))
tsteps = int(os.environ.get("SLAB_TIME_STEPS", "4096"))


def run(kind, n, b, kernel, cgrp=0, rg=0, nsteps=steps, timing=False):
    os.environ["CCVM_AMD_KERNEL"] = kernel
    for k, v in (("CCVM_AMD_SLAB_CGRP", cgrp), ("CCVM_AMD_SLAB_RG", rg)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)
    traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0)
    desc = bench.describe_launch(kind, b, n)
    traj.advance(nsteps)
    torch.cuda.synchronize()
    traj.check()
    out = {k: traj.compact(k).cpu() for k in traj.state}
    best = None
    if timing:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            traj.advance(tsteps)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        traj.check()
    return out, desc, best


for case in cases:
    parts = case.split(":")
    kind, n, b = parts[0], int(parts[1]), int(parts[2])
    cgrp = int(parts[3]) if len(parts) > 3 else 0
    rg = int(parts[4]) if len(parts) > 4 else 0
    ref, _, _ = run(kind, n, b, "tile")
    other = run(kind, n, b, "noslab", timing=True)[2] if os.environ.get("SLAB_COMPARE") else None
    got, desc, best = run(kind, n, b, "slab", cgrp, rg, timing=True)
    errs = []
    for k in ref:
        scale = max(float(ref[k].abs().max()), 1e-6)
        errs.append(f"{k} {float((ref[k] - got[k]).abs().max()) / scale:.2e}")
    if os.environ.get("SLAB_ROWS"):
        for k in ref:
            print("   ", k, "per-row max err:", [f"{float(x):.1e}" for x in (ref[k] - got[k]).abs().amax(dim=1)[:16]])
    bad = any(not torch.isfinite(got[k]).all() for k in got)
    cmp = f" (noslab {other / tsteps * 1e6:7.3f})" if other else ""
    print(f"{case:24s} {best / tsteps * 1e6:8.3f} us/step{cmp}  rel.err vs tile: {' '.join(errs)}{' NONFINITE' if bad else ''}\n"
          f"    {desc}", flush=True)
