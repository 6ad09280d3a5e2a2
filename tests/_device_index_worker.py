"""Worker of tests/test_gpu_device_index.py: one small solve on the device the environment names.
    python _device_index_worker.py <out.pt>
Prints `count=<visible GPUs>`; writes objective values and variables of a DL and a PL solve."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    out = sys.argv[1]
    print(f"count={torch.cuda.device_count()}", flush=True)
    from ccvm_amd import engine
    from ccvm_amd.solvers import DLSolver, PumpedLangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS, synthetic_instance

    dev = engine.gpu_device()
    print(f"device={dev.index}", flush=True)
    res = {"index": dev.index}
    for kind, cls, n, batch in (("dl", DLSolver, 96, 70), ("pl", PumpedLangevinSolver, 300, 100)):
        inst = synthetic_instance(n, seed=5)
        inst.optimal_sol = 1.0
        solver = cls(device="cpu", batch_size=batch)
        solver.parameter_key = {n: dict(EXAMPLE_PARAMS[kind], iterations=30)}
        solver.noise_seed = 0xD15EA5E
        inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
        sol = solver(instance=inst)
        res[kind] = {"obj": sol.objective_values.cpu(), "x": sol.variables["problem_variables"].cpu()}
    # the engine's tensors live where the environment said
    q, v, _ = __import__("ccvm_amd.workloads", fromlist=["scaled_qv"]).scaled_qv(64, "dl")
    res["problem_device"] = engine.DeviceProblem(q, v).q.device.index
    torch.save(res, out)
