"""Solve the shipped BoxQP instance(s) with the DL-CCVM solver on the MI355X engine.

Run from this directory:   PYTHONPATH=.. python boxqp_dl_demo.py
(the reference's own examples/ccvm_boxqp_dl.py runs unmodified the same way: it only needs
`ccvm_simulators` to resolve to this repository -- see INTEGRATION.md)
"""
import glob
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from ccvm_simulators.problem_classes.boxqp import ProblemInstance  # noqa: E402
from ccvm_simulators.solvers import DLSolver  # noqa: E402

INSTANCES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "benchmarking_instances",
                         "single_test_instance", "*.in")

if __name__ == "__main__":
    solver = DLSolver(device="cpu", batch_size=1000)  # "cpu": tensors live on the host; compute is on the GPU
    solver.parameter_key = {
        20: {"pump": 8.0, "feedback_scale": 100, "dt": 0.001, "iterations": 1500, "noise_ratio": 10},
    }
    torch.manual_seed(1234)
    for path in sorted(glob.glob(INSTANCES)):
        instance = ProblemInstance(instance_type="test", file_path=path, device=solver.device)
        instance.scale_coefs(solver.get_scaling_factor(instance.q_matrix))
        solution = solver(instance=instance, post_processor=None)
        print(solution)
        print(f"TTS@99% = {solution.tts99():.3e} s  (optimal fraction {solution.solution_performance['optimal']})")
