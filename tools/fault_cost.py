"""What a dropped workgroup costs, family by family (developer tool): CCVM_AMD_FAULT=cluster_drop at the BASELINE shapes,
the GPU-side duration of the faulty run call (HIP events) and the wall time of the recovery.
   python tools/fault_cost.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_cluster import _run_engine

os.environ["CCVM_AMD_EXCHANGE_COOLDOWN"] = "0"
for kind, n, b, family in [("dl", 1000, 1000, "ptile"), ("mf", 500, 1000, "cluster"), ("langevin", 500, 1000, "cluster"),
                           ("dl", 1000, 32, "slab"), ("dl", 500, 1000, "cluster")]:
    os.environ["CCVM_AMD_KERNEL"] = "nocluster"
    os.environ.pop("CCVM_AMD_FAULT", None)
    _run_engine(kind, n, b, 20, None, 21, 0).check()
    os.environ["CCVM_AMD_KERNEL"] = family
    _run_engine(kind, n, b, 20, None, 22, 0).check()
    for spin in (None, "5", "50"):
        os.environ["CCVM_AMD_FAULT"] = "cluster_drop"
        if spin: os.environ["CCVM_AMD_SPIN_MS"] = spin
        else: os.environ.pop("CCVM_AMD_SPIN_MS", None)
        traj = _run_engine(kind, n, b, 20, None, 21, 0, chunks=[0])
        traj.arm(force=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); traj.advance(20); e1.record()
        torch.cuda.synchronize()
        t_kernel = time.perf_counter() - t0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rec = traj.check()
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        print(f"{kind} N={n} B={b} {family:8s} bound {spin or 'default'} ms: faulty run call {e0.elapsed_time(e1):7.2f} ms on the GPU "
              f"({t_kernel * 1e3:.1f} ms wall), recovered={rec}, end to end {total * 1e3:.1f} ms", flush=True)
        os.environ.pop("CCVM_AMD_FAULT", None)
