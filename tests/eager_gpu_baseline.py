"""Context number, not a test: the reference's op sequence (the oracle) run EAGERLY on the MI355X
through torch-ROCm -- i.e. what `device="cuda"` of the reference would do on this box (~55 aten
launches per DL step) -- next to the engine.  Lives under tests/ because it imports the oracle.

    python tests/eager_gpu_baseline.py [N] [B]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv  # noqa: E402
from oracle import ccvm_oracle as oracle  # noqa: E402


class DeviceNoise:
    """Normal(0, 1).sample((N,)) on the device, transposed (dl_solver.py:538-547 with device="cuda")."""

    def draw(self, i, stream, n, b):
        return torch.randn(n, b, device="cuda").T


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    q, v, _ = scaled_qv(n, "dl")
    q, v = q.cuda(), v.cuda()
    p = EXAMPLE_PARAMS["dl"]
    c = torch.zeros((b, n), device="cuda")
    s = torch.zeros((b, n), device="cuda")
    total = 1000
    run = lambda step0, k: oracle.dl_loop(q, v, b, total, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"],
                                          0.05, (0.0, 1.0), True, DeviceNoise(), step0, k, c, s)
    run(0, 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(50, 300)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 300
    print(f"torch-ROCm eager, DL N={n} B={b}: {dt * 1e6:.1f} us/step, {b / dt:.3e} row-steps/s "
          f"(finite: {bool(torch.isfinite(c).all())})")
