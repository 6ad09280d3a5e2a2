import sys, time, torch
sys.path.insert(0, '.')
from ccvm_amd.problem_classes.boxqp import ProblemInstance
from ccvm_amd.solvers import DLSolver
solver = DLSolver(device="cpu", batch_size=1000)
solver.parameter_key = {20: {"pump": 8.0, "feedback_scale": 100, "dt": 0.001, "iterations": 1500, "noise_ratio": 10}}
inst = ProblemInstance(instance_type="test", file_path="examples/benchmarking_instances/single_test_instance/synthetic020-100-20.in", device="cpu")
inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
for i in range(4):
    t0 = time.time(); sol = solver(instance=inst); t1 = time.time()
    print(f"call {i}: wall {1e3*(t1-t0):8.2f} ms   solve_time*B {1e3*sol.solve_time*1000:8.2f} ms  best {sol.best_objective_value:.3f}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); solver(instance=inst); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
