"""Where the time of ONE solver call goes at the shipped example's size (developer tool): the reference's `solve_time`
region (dl_solver.py:851-933) against the kernel's own time, and a cProfile of the host side.
   python tools/solve_cost.py [dl|mf|langevin|pl]"""
import cProfile, json, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ccvm_amd.problem_classes.boxqp import ProblemInstance
from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver
from ccvm_amd.workloads import EXAMPLE_PARAMS

kind = sys.argv[1] if len(sys.argv) > 1 else "dl"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
arrays = np.load(os.path.join(root, "tests", "golden", "tuningH020.npz"))
meta = json.load(open(os.path.join(root, "tests", "golden", "tuningH020.json")))["instance"]
inst = ProblemInstance.from_arrays(arrays["q_matrix"], arrays["v_vector"], device="cuda", name=meta["name"],
                                   optimal_sol=meta["optimal_sol"], best_sol=meta["best_sol"])
cls = {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver, "pl": PumpedLangevinSolver}[kind]
solver = cls(device="cuda", batch_size=1000)
solver.parameter_key = {20: dict(EXAMPLE_PARAMS[kind], iterations=1500)}
inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
torch.manual_seed(1)
solver(instance=inst)
walls, solves = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sol = solver(instance=inst)
    torch.cuda.synchronize()
    walls.append(time.perf_counter() - t0)
    solves.append(sol.solve_time * 1000)
walls.sort(); solves.sort()
print(f"{kind}: __call__ wall median {walls[10] * 1e6:.0f} us (min {walls[0] * 1e6:.0f}); solve_time x batch median {solves[10] * 1e6:.0f} us "
      f"(min {solves[0] * 1e6:.0f}) = {solves[10] * 1e6 / 1500:.3f} us per step; the kernel alone: 1500 x 0.35 = 528 us")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    solver(instance=inst)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
