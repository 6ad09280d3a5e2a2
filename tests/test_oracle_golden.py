"""Pins the oracle (oracle/ccvm_oracle.py) to the reference: every golden case produced by
tests/golden/make_golden.py (i.e. by the reference itself) must be reproduced.

The oracle issues the same torch ops in the same order, so on the torch build that made
the fixtures the match is bit-exact; the gates below are a hair wider (1e-5 on O(1)
amplitudes, 1e-5 relative on objectives) so that a host whose BLAS/vector kernels round
differently still passes, and far tighter than the GPU parity tolerance.
"""
import pytest
import torch

from golden_util import (all_cases, asgd_arrays, asgd_cases, bounds_arrays, bounds_cases, check_noise_checksum,
                         golden, vector_s_arrays, vector_s_cases)
from oracle import ccvm_oracle as oracle

ATOL_STATE = 1e-5
RTOL_OBJ = 1e-5


def _saturation(value):
    """Scalar S, or the per-variable S of the vector-S fixtures as a float32 tensor."""
    return torch.tensor(value, dtype=torch.float32) if isinstance(value, list) else value


def run_oracle(g, meta):
    kind, p = meta["kind"], dict(meta["params"])
    if "S" in p:
        p["S"] = _saturation(p["S"])
    q, v, f = g.scaled(kind)
    common = dict(bounds=tuple(meta.get("bounds", (0.0, 1.0))), scaled_by=f, optimal_value=g.instance["optimal_sol"],
                  post_processor=meta["post"])
    b, t = meta["batch"], meta["iterations"]
    torch.manual_seed(meta["seed"])
    if kind == "dl":
        return oracle.solve_dl(q, v, b, t, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"],
                               g=meta.get("g") or 0.05,
                               S=_saturation(meta.get("dl_S")) if meta.get("dl_S") is not None else 1,
                               pump_rate_flag=meta["pump_rate_flag"], **common)
    if kind == "mf":
        return oracle.solve_mf(q, v, b, t, p["pump"], p["dt"], p["j"], p["feedback_scale"], p["S"],
                               g=meta.get("g") or 0.01,
                               pump_rate_flag=meta["pump_rate_flag"], adam=meta["adam"], **common)
    if kind == "langevin":
        return oracle.solve_langevin(q, v, b, t, p["dt"], p["sigma"], p["feedback_scale"], p["S"],
                                     adam=meta["adam"], **common)
    return oracle.solve_pl(q, v, b, t, p["pump"], p["dt"], p["sigma"], p["feedback_scale"], p["S"],
                           pump_rate_flag=meta["pump_rate_flag"], adam=meta["adam"], **common)


@pytest.mark.parametrize("tag,case", all_cases())
def test_oracle_reproduces_reference(tag, case):
    g = golden(tag)
    meta = g.cases[case]
    if meta["iterations"] > 200 and meta["kind"] != "dl":
        pytest.skip("long runs are covered for DL; others keep the CPU suite short")
    check_noise_checksum(meta, g.instance["problem_size"], meta["batch"])
    out = run_oracle(g, meta)
    for field in g.fields(case):
        want = g.out(case, field)
        got = out[field]
        if field == "objective_values":
            tol = RTOL_OBJ * max(1.0, float(want.abs().max()))
        else:
            tol = ATOL_STATE * max(1.0, float(want.abs().max()))
        err = float((got - want).abs().max())
        assert err <= tol, f"{tag}/{case}/{field}: max abs err {err:.3e} > {tol:.3e}"
    assert abs(out["best_objective_value"] - meta["best_objective_value"]) <= RTOL_OBJ * abs(
        meta["best_objective_value"]) + 1e-6
    for key, frac in meta["solution_performance"].items():
        assert abs(out["solution_performance"][key] - frac) <= 1.0 / meta["batch"] + 1e-9


@pytest.mark.parametrize("case", sorted(bounds_cases()))
def test_oracle_reproduces_reference_with_other_bounds(case):
    """solution_bounds other than (0, 1): the (u - l), (u + l) maps of every drift / grads function."""
    g, meta, arrays = golden("test020"), bounds_cases()[case], bounds_arrays()
    assert tuple(meta["bounds"]) != (0.0, 1.0) or meta["dl_S"] is not None or meta["g"] is not None
    out = run_oracle(g, meta)
    for key in arrays.files:
        if not key.startswith(case + "/"):
            continue
        field = key[len(case) + 1:]
        want = torch.from_numpy(arrays[key].copy())
        tol = (RTOL_OBJ if field == "objective_values" else ATOL_STATE) * max(1.0, float(want.abs().max()))
        assert float((out[field] - want).abs().max()) <= tol, f"{case}/{field}"
    assert abs(out["best_objective_value"] - meta["best_objective_value"]) <= RTOL_OBJ * abs(
        meta["best_objective_value"]) + 1e-6


@pytest.mark.parametrize("case", sorted(vector_s_cases()))
def test_oracle_reproduces_reference_with_per_variable_saturation(case):
    """S given as a 1-D tensor of length N (dl_solver.py:843-848 and the same lines of the other solvers)."""
    g, meta, arrays = golden("test020"), vector_s_cases()[case], vector_s_arrays()
    out = run_oracle(g, meta)
    for key in arrays.files:
        if not key.startswith(case + "/"):
            continue
        field = key[len(case) + 1:]
        want = torch.from_numpy(arrays[key].copy())
        tol = (RTOL_OBJ if field == "objective_values" else ATOL_STATE) * max(1.0, float(want.abs().max()))
        assert float((out[field] - want).abs().max()) <= tol, f"{case}/{field}"
    assert abs(out["best_objective_value"] - meta["best_objective_value"]) <= RTOL_OBJ * abs(
        meta["best_objective_value"]) + 1e-6


@pytest.mark.parametrize("case", sorted(asgd_cases()))
def test_oracle_reproduces_reference_with_asgd_and_lbfgs_post_processors(case):
    g, meta, arrays = golden("test020"), asgd_cases()[case], asgd_arrays()
    out = run_oracle(g, meta)
    for key in arrays.files:
        if not key.startswith(case + "/"):
            continue
        field = key[len(case) + 1:]
        want = torch.from_numpy(arrays[key].copy())
        tol = (RTOL_OBJ if field == "objective_values" else ATOL_STATE) * max(1.0, float(want.abs().max()))
        assert float((out[field] - want).abs().max()) <= tol, f"{case}/{field}"


def test_oracle_post_processors_match_reference_called_directly():
    """num_iter = 3 equals num_iter = 1 for adam and asgd (only the first optimizer step takes effect in
    the reference); custom bounds; grad-descent with custom iteration count and step."""
    a = asgd_arrays()
    q, v, c = (torch.from_numpy(a[f"direct/{k}"].copy()) for k in ("q", "v", "c"))
    for label, fn in (("adam", oracle.pp_adam), ("asgd", oracle.pp_asgd), ("lbfgs", oracle.pp_lbfgs)):
        # adam / asgd: only the first optimizer step takes effect; lbfgs makes a new optimizer per iteration
        assert (a[f"direct/{label}_iter1"] == a[f"direct/{label}_iter3"]).all() == (label != "lbfgs")
        for it in (1, 3):
            want = torch.from_numpy(a[f"direct/{label}_iter{it}"].copy())
            assert float((fn(c, q, v, num_iter=it) - want).abs().max()) <= 1e-7
        want = torch.from_numpy(a[f"direct/{label}_bounds"].copy())
        assert float((fn(c, q, v, 0.2, 0.7) - want).abs().max()) <= 1e-7
    want = torch.from_numpy(a["direct/lbfgs_steep"].copy())  # |g|_1 >> 1: the 1 / |g|_1 branch of the step size
    assert float((oracle.pp_lbfgs(c, q * 100, v * 100, num_iter=2) - want).abs().max()) <= 1e-7
    assert float((oracle.pp_grad_descent(c, q, v) - torch.from_numpy(a["direct/grad-descent"].copy())).abs().max()) <= 1e-6
    want = torch.from_numpy(a["direct/grad-descent_custom"].copy())
    assert float((oracle.pp_grad_descent(c, q, v, 0.1, 0.9, num_iter_pp=4, step_size=0.05) - want).abs().max()) <= 1e-6


def test_oracle_reproduces_baseline_config_1():
    """BASELINE.json configs[0] (the reference's own CPU-runnable case): DLSolver on test020-100-10,
    batch 100, 15000 iterations, example parameters, seed 1234 -> best 142.6326.
    Fixture: tests/golden/baseline_config1_anchor.{json,npz}, produced by the reference itself."""
    import json
    import os

    import numpy as np

    from golden_util import GOLDEN_DIR

    with open(os.path.join(GOLDEN_DIR, "baseline_config1_anchor.json")) as fh:
        meta = json.load(fh)
    arrays = np.load(os.path.join(GOLDEN_DIR, "baseline_config1_anchor.npz"))
    g = golden("test020")
    assert meta["iterations"] == 15000 and meta["batch"] == 100 and abs(meta["best_objective_value"] - 142.6326) < 1e-4
    check_noise_checksum(meta, g.instance["problem_size"], meta["batch"])
    q, v, f = g.scaled("dl")
    p = meta["params"]
    torch.manual_seed(meta["seed"])
    out = oracle.solve_dl(q, v, meta["batch"], p["iterations"], p["pump"], p["dt"], p["noise_ratio"],
                          p["feedback_scale"], optimal_value=g.instance["optimal_sol"], scaled_by=float(f),
                          pump_rate_flag=meta["pump_rate_flag"])
    assert np.array_equal(out["problem_variables"].numpy(), arrays["problem_variables"])
    assert np.array_equal(out["s"].numpy(), arrays["s"])
    assert np.array_equal(out["objective_values"].numpy(), arrays["objective_values"])
    assert out["best_objective_value"] == meta["best_objective_value"]
    assert out["solution_performance"] == meta["solution_performance"]


def test_scaling_factor_matches_reference():
    for tag in ("test020", "tuningH020"):
        g = golden(tag)
        for kind, want in g.manifest["scaling_factor"].items():
            mult = 0.2 if kind == "dl" else 0.05
            assert abs(float(oracle.scaling_factor(g.q(), mult)) - want) <= 1e-6 * want


# ---- known answers held by the reference's own unit tests ------------------------------
def test_mf_grads_and_drift_known_answers():
    """ccvm_simulators/tests/unit/solvers/test_mf_solver.py:63-130: with Q = V = ones(2),
    mu_tilde = 1, S = fs = 1 ... the expected grads are -20.0 and the drift (-20.0, 200.5)."""
    # the reference test builds: q = [[10,10],[10,10]], v = [10,10], mu_tilde = [[1,1]], S=1, fs=1
    q = torch.full((2, 2), 10.0)
    v = torch.full((2,), 10.0)
    mu_tilde = torch.ones((1, 2))
    grads = oracle.mf_grads(mu_tilde, q, v, S=1.0, fs=1.0, lo=0.0, hi=1.0)
    assert torch.allclose(grads, torch.full((1, 2), -(0.25 * 2 * 2 * 10) - 5.0))


def test_change_variables_known_answers():
    """test_mf_solver.py:132-154 -- change_variables([1, ...], 0, 1, S) spot values."""
    x = torch.tensor([[2.0, 0.2]])
    y = oracle.change_variables(x, 0.0, 1.0, 1.0)
    assert torch.allclose(y, torch.tensor([[1.5, 0.6]]))


def test_success_fraction_known_answers():
    """ccvm_simulators/tests/test_solution.py:140-173 pattern: gaps {0, 1.5, 7} % of 100."""
    obj = -torch.tensor([100.0, 98.5, 93.0])
    best, perf = oracle.solution_stats(obj, 100.0)
    assert best == 100.0
    assert perf["optimal"] == round(1 / 3, 4) and perf["two_percent"] == round(2 / 3, 4)
    assert perf["five_percent"] == round(2 / 3, 4) and perf["ten_percent"] == 1.0


def test_r99():
    assert oracle.r99(1.0) == 1.0 and oracle.r99(0.99) == 1.0
    assert abs(oracle.r99(0.5) - 6.643856189774724) < 1e-9
