"""Wall time and reported solve_time of repeated solver calls, and a profile of the FIRST call
(developer tool: what one-time cost is still inside the timed region?)."""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, '.')
from ccvm_amd.problem_classes.boxqp import ProblemInstance
from ccvm_amd.solvers import DLSolver

solver = DLSolver(device="cpu", batch_size=1000)
solver.parameter_key = {20: {"pump": 8.0, "feedback_scale": 100, "dt": 0.001, "iterations": 1500, "noise_ratio": 10}}
inst = ProblemInstance(instance_type="test", device="cpu",
                       file_path="examples/benchmarking_instances/single_test_instance/example020-100-29.in")
inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
pr = cProfile.Profile()
for i in range(4):
    t0 = time.time()
    if i == 0:
        pr.enable()
    sol = solver(instance=inst)
    if i == 0:
        pr.disable()
    t1 = time.time()
    print(f"call {i}: wall {1e3 * (t1 - t0):8.2f} ms   solve_time*B {1e3 * sol.solve_time * 1000:8.2f} ms  "
          f"best {sol.best_objective_value:.3f}")
pstats.Stats(pr).sort_stats("cumulative").print_stats("ccvm_amd|built-in|method", 40)
