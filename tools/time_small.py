"""Per-step time of the engine for kind:N:B[:adam] cases (developer tool).
   python tools/time_small.py [kind:N:B[:adam] ...]     kinds: dl mf langevin pl; ":adam" = the Adam variant of the
   example scripts (alpha 0.001, beta1 0.9, beta2 0.999)"""
import os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cases = sys.argv[1:] or ["dl:20:1000", "dl:64:1000", "dl:100:1000", "dl:128:1000", "mf:20:1000", "mf:100:1000",
                         "langevin:20:1000", "langevin:100:1000", "pl:100:1000", "dl:100:4000", "dl:20:100"]
steps = 4096
SHAPE = re.compile(r"persist_kernel<[^>]*> grid \d+ x \d+")
for case in cases:
    kind, n, b, *rest = case.split(":")
    n, b = int(n), int(b)
    adam = {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False} if rest == ["adam"] else None
    traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0, adam=adam)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.06:  # leave the idle clocks (a 4096-step launch at N = 20 is 1-2 ms)
        traj.advance(steps)
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        traj.advance(steps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    traj.check()
    # what ran, in the regime map's letters (ccvm_describe_launch under the current tuning environment)
    import ctypes
    from ccvm_amd import _lib
    from tools.regime_map import family
    buf = ctypes.create_string_buffer(1024)
    _lib.load().ccvm_describe_launch(bench.SOLVER_ID[kind], b, n, 1 if adam else 0, 0, buf, 1024)
    print(f"{case:24s} RU={os.environ.get('CCVM_AMD_PERSIST_RU', 'auto'):4s} PW={os.environ.get('CCVM_AMD_PERSIST_PW', 'auto'):4s} "
          f"{best / steps * 1e6:8.3f} us/step "
          f"{steps * b / best:.3e} row-steps/s  [{family(buf.value.decode())}] "
          + (SHAPE.search(buf.value.decode()) or [''])[0], flush=True)
