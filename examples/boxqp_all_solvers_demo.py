"""All four CCVM solvers (and their Adam variants) on the shipped BoxQP instance, on the MI355X engine.

Run from this directory:   PYTHONPATH=.. python boxqp_all_solvers_demo.py [--batch 1000] [--iterations 1500]
Prints one line per run: best objective, fraction of the batch within 0.1 % / 1 % of the known optimum,
solve time per trajectory and TTS@99 %.
"""
import argparse
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from ccvm_simulators.problem_classes.boxqp import ProblemInstance  # noqa: E402
from ccvm_simulators.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver  # noqa: E402
from ccvm_simulators.solvers.algorithms import AdamParameters  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
INSTANCES = os.path.join(HERE, "benchmarking_instances", "single_test_instance", "*.in")

# per-solver parameters for N = 20 (the values the reference's example scripts ship)
RUNS = [
    ("DL-CCVM", DLSolver, {"pump": 8.0, "feedback_scale": 100, "dt": 0.001, "noise_ratio": 10}, None, None),
    ("MF-CCVM", MFSolver, {"pump": 0.0, "feedback_scale": 4000, "j": 5.0, "S": 20.0, "dt": 0.0025}, None, None),
    ("MF-CCVM + Adam + grad-descent", MFSolver,
     {"pump": 0.0, "feedback_scale": 4000, "j": 5.0, "S": 20.0, "dt": 0.0025},
     AdamParameters(alpha=0.001, beta1=0.9, beta2=0.999, add_assign=False), "grad-descent"),
    ("Langevin", LangevinSolver, {"dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0}, None, None),
    ("Langevin + Adam", LangevinSolver, {"dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0},
     AdamParameters(alpha=0.001, beta1=0.9, beta2=0.999, add_assign=True), None),
    ("pumped Langevin + adam post-processor", PumpedLangevinSolver,
     {"pump": 2.0, "dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0}, None, "adam"),
]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1000)
    ap.add_argument("--iterations", type=int, default=1500)
    args = ap.parse_args()
    for path in sorted(glob.glob(INSTANCES)):
        print(os.path.basename(path))
        for label, cls, params, adam, post in RUNS:
            solver = cls(device="cuda", batch_size=args.batch)
            instance = ProblemInstance(instance_type="test", file_path=path, device=solver.device)
            solver.parameter_key = {instance.problem_size: dict(params, iterations=args.iterations)}
            instance.scale_coefs(solver.get_scaling_factor(instance.q_matrix))
            torch.manual_seed(1234)
            solver(instance=instance, post_processor=post, algorithm_parameters=adam)  # one-time initialisation
            sol = solver(instance=instance, post_processor=post, algorithm_parameters=adam)
            perf = sol.solution_performance
            print(f"  {label:40s} best {sol.best_objective_value:10.4f} (optimum {sol.optimal_value:.4f})  "
                  f"within 0.1%: {perf['optimal']:.3f}  1%: {perf['one_percent']:.3f}  "
                  f"solve {sol.solve_time * 1e6:7.2f} us/trajectory  TTS99 {sol.tts99():.3e} s")
