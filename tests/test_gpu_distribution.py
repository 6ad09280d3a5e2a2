"""Solution-level parity of the DEFAULT noise mode with the reference, in distribution.

In replay mode the engine consumes the reference's own normals and is compared value by value
(test_gpu_parity.py).  The default -- normals generated inside the kernels -- cannot be: the reference draws
`Normal.sample` from torch's CPU generator per step (dl_solver.py:538-547).  What must agree is the
distribution of the results ("success fractions / best objective within sampling
error").  Fixture: tests/golden/distribution_anchors.* -- the reference itself on the shipped example
configuration of every solver (tuningH020-100-0, batch 1000, 1500 iterations, the example scripts'
parameters), three seeds each: the objective value of every trajectory (make_golden.py --only-distribution).
The engine runs the same configuration through the public API with the fused generator and five seeds."""
import json
import math
import os

import numpy as np
import pytest
import torch
from scipy import stats

from golden_util import GOLDEN_DIR, golden

pytestmark = pytest.mark.gpu

SEEDS = (11, 22, 33, 44, 55)
THRESHOLDS = {"optimal": 0.1, "one_percent": 1.0, "five_percent": 5.0}


def _anchors():
    arrays = np.load(os.path.join(GOLDEN_DIR, "distribution_anchors.npz"))
    with open(os.path.join(GOLDEN_DIR, "distribution_anchors.json")) as fh:
        return arrays, json.load(fh)["cases"]


def _engine_values(kind, params, batch):
    from ccvm_amd.problem_classes.boxqp import ProblemInstance
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver

    g = golden("tuningH020")
    cls = {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver, "pl": PumpedLangevinSolver}[kind]
    values = []
    for seed in SEEDS:
        inst = ProblemInstance.from_arrays(g.q(), g.v(), device="cuda", name=g.instance["name"],
                                           optimal_sol=g.instance["optimal_sol"], best_sol=g.instance["best_sol"])
        solver = cls(device="cuda", batch_size=batch)
        solver.noise_mode = "fused"
        solver.parameter_key = {20: dict(params)}
        inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
        torch.manual_seed(seed)  # the fused generator's key is drawn from torch's global generator
        sol = solver(instance=inst)
        values.append(sol.objective_values.cpu().numpy())
    return np.concatenate(values), g.instance["optimal_sol"]


def _gaps(objective_values, optimal):
    """Optimality gap in percent as solution.py:87-146 defines it: the objective values are energies of the
    minimisation form, the value found is their negative, the gap is relative to the value found."""
    found = -objective_values
    return (optimal - found) * 100.0 / np.abs(found)


@pytest.mark.parametrize("kind", ["dl", "mf", "langevin", "pl"])
def test_fused_mode_results_are_distributed_like_the_reference(kind):
    arrays, cases = _anchors()
    names = [n for n in cases if cases[n]["kind"] == kind]
    assert len(names) == 3
    meta = cases[names[0]]
    want = np.concatenate([arrays[n] for n in names])
    got, optimal = _engine_values(kind, meta["params"], meta["batch"])
    assert got.shape == (len(SEEDS) * meta["batch"],) and np.isfinite(got).all()

    g_want, g_got = _gaps(want.astype(np.float64), optimal), _gaps(got.astype(np.float64), optimal)
    n_w, n_g = len(g_want), len(g_got)
    for label, gap in THRESHOLDS.items():
        p_w, p_g = float((g_want <= gap).mean()), float((g_got <= gap).mean())
        pooled = (p_w * n_w + p_g * n_g) / (n_w + n_g)
        # two-proportion z-test, 3.5 sigma (12 comparisons in this file: false-alarm rate ~ 0.5 %), with a
        # floor of two trajectories for fractions at 0 or 1
        sigma = math.sqrt(max(pooled * (1 - pooled), 1e-12) * (1 / n_w + 1 / n_g))
        assert abs(p_g - p_w) <= 3.5 * sigma + 2.0 / n_g, f"{kind} {label}: engine {p_g:.4f} vs reference {p_w:.4f}"
    # the whole distribution of the gaps: two-sample Kolmogorov-Smirnov on values rounded to 0.001 % (the optimum
    # is a mass point whose last float bits differ between the two implementations)
    ks = stats.ks_2samp(np.round(g_got, 3), np.round(g_want, 3))
    assert ks.pvalue > 1e-3, f"{kind}: KS statistic {ks.statistic:.4f}, p = {ks.pvalue:.2e}"
    # the best value found over thousands of trajectories is the same optimum
    assert abs(float(got.min()) - float(want.min())) <= 1e-4 * abs(float(want.min()))
    assert 0.0 < float((g_want <= 5.0).mean())  # (the thresholds really discriminate: not everything outside)


def test_dl_example_anchor_success_fraction():
    """SURVEY 8c anchor, the DL example exactly as shipped (B = 1000, T = 1500, seed 1234): `optimal` fraction 0.987,
    best 130.7142.  The engine's fused-mode runs pooled over five seeds land within 3 sigma (binomial, both
    samples' variances) of it."""
    with open(os.path.join(GOLDEN_DIR, "dl_example_anchor.json")) as fh:
        anchor = json.load(fh)
    got, optimal = _engine_values("dl", anchor["params"], anchor["batch"])
    p = float((_gaps(got.astype(np.float64), optimal) <= 0.1).mean())
    p0 = anchor["solution_performance"]["optimal"]
    sigma = math.sqrt(p0 * (1 - p0) * (1 / len(got) + 1 / anchor["batch"]))
    assert abs(p - p0) <= 3 * sigma, f"engine {p:.4f} vs anchor {p0} (sigma {sigma:.4f})"
    assert abs(float((-got).max()) - anchor["best_objective_value"]) <= 1e-4 * anchor["best_objective_value"]
