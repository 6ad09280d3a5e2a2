"""Synthetic BoxQP workloads for N >= 100 (the reference's shipped instances stop at N = 70).

Recipe of SURVEY.md section 8(d): dense symmetric Q with std ~5 and V with std ~17 (the
statistics of the shipped files), taken as already negated (minimisation form); solver
parameters are the reference example scripts' values re-keyed to N.
"""
import torch

from .problem_classes.boxqp import ProblemInstance

# examples/ccvm_boxqp_dl.py:16-24, ccvm_boxqp_mf.py:16-25, langevin_boxqp.py:16-24,
# pumped_langevin_boxqp.py:16-25 of the reference (values only)
EXAMPLE_PARAMS = {
    "dl": {"pump": 8.0, "feedback_scale": 100, "dt": 0.001, "noise_ratio": 10},
    "mf": {"pump": 0.0, "feedback_scale": 4000, "j": 5.0, "S": 20.0, "dt": 0.0025},
    "langevin": {"dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0},
    "pl": {"pump": 2.0, "dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0},
}
SCALING_MULTIPLIER = {"dl": 0.2, "mf": 0.05, "langevin": 0.05, "pl": 0.05}


def synthetic_qv(n, seed=0):
    """(Q, V) float32 host tensors, unscaled."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(n, n, generator=g) * 5
    q = (a + a.T) / 2**0.5
    v = torch.randn(n, generator=g) * 17
    return q, v


def synthetic_instance(n, device="cpu", seed=0, name=None):
    q, v = synthetic_qv(n, seed)
    return ProblemInstance.from_arrays(q, v, device=device, name=name or f"synthetic{n:04d}-100-{seed}")


def scaled_qv(n, kind, seed=0):
    """(Q, V, scaled_by) after scale_coefs(get_scaling_factor(Q)) for a solver kind."""
    q, v = synthetic_qv(n, seed)
    f = torch.sqrt(torch.sum(torch.abs(q))) * SCALING_MULTIPLIER[kind]
    return q / f, v / f, f
