"""The reference's overridable hooks (ccvm_solver.py:152-170) through the public API on the GPU.

Row (b) of SURVEY section 8: ``calculate_drift`` / ``calculate_grads`` / ``fit_to_constraints`` /
``change_variables`` are attributes a caller may replace, and the reference's own unit tests do
(test_mf_solver.py:264-266).  A hook the selected loop never calls must not matter; a hook it does call
must be honoured -- the run then takes the composed per-step path (ccvm_amd/solvers/composed.py), which
has to agree with the fused kernels when the replacement computes the same thing."""
import warnings
from unittest import mock

import pytest
import torch

from golden_util import golden, reference_unit_vectors

pytestmark = pytest.mark.gpu

_ADAM = dict(alpha=0.001, beta1=0.9, beta2=0.999, add_assign=False)


def _instance(n=96, device="cpu"):
    from ccvm_amd.workloads import synthetic_instance

    inst = synthetic_instance(n, device=device)
    inst.optimal_sol = 1.0
    return inst


def _solver(kind, device="cpu", batch=48, n=96, iterations=25):
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS

    cls = {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver, "pl": PumpedLangevinSolver}[kind]
    solver = cls(device=device, batch_size=batch)
    solver.parameter_key = {n: dict(EXAMPLE_PARAMS[kind], iterations=iterations)}
    solver.noise_seed = 4242
    return solver


def _solve(solver, inst, adam=False):
    from ccvm_amd.solvers.algorithms import AdamParameters

    inst = _instance(inst) if isinstance(inst, int) else inst
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    return solver(instance=inst, algorithm_parameters=AdamParameters(**_ADAM) if adam else None)


def _close(got, want, tol=2e-4):
    assert set(got.variables) == set(want.variables)
    for name in want.variables:
        scale = max(1.0, float(want.variables[name].abs().max()))
        err = float((got.variables[name].cpu() - want.variables[name].cpu()).abs().max())
        assert err <= tol * scale, f"{name}: {err:.3e}"
    rel = float(((got.objective_values.cpu() - want.objective_values.cpu()).abs()
                 / want.objective_values.cpu().abs().clamp_min(1.0)).max())
    assert rel <= 1e-5, f"objective values: {rel:.3e}"


# what each loop calls (solvers/composed.py's table); DL has no reachable Adam variant
_ON_PATH = [("dl", False, "calculate_drift"), ("mf", False, "calculate_drift"), ("mf", False, "fit_to_constraints"),
            ("mf", True, "calculate_grads"), ("mf", True, "fit_to_constraints"),
            ("langevin", False, "calculate_drift"), ("langevin", True, "calculate_grads"),
            ("langevin", True, "fit_to_constraints"), ("pl", False, "calculate_drift"),
            ("pl", True, "calculate_grads"), ("pl", False, "fit_to_constraints")]
_OFF_PATH = [("dl", False, "calculate_grads"), ("mf", False, "calculate_grads"), ("mf", True, "calculate_drift"),
             ("langevin", False, "calculate_grads"), ("langevin", True, "calculate_drift"),
             ("pl", False, "calculate_grads"), ("pl", True, "calculate_drift")]


@pytest.mark.parametrize("kind,adam,hook", _OFF_PATH)
def test_a_replaced_hook_the_loop_never_calls_is_never_looked_at(kind, adam, hook):
    """mf_solver.py:561-572 calls calculate_drift only, :709-716 calculate_grads only (and so on): the fused
    kernels run, without a warning, and the result is the untouched solver's bit for bit."""
    want = _solve(_solver(kind), 96, adam)
    solver = _solver(kind)

    def boom(*args, **kwargs):
        raise AssertionError(f"{hook} is not on this loop's path")

    setattr(solver, hook, boom)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = _solve(solver, 96, adam)
    for name in want.variables:
        assert torch.equal(got.variables[name], want.variables[name]), name
    assert torch.equal(got.objective_values, want.objective_values)


@pytest.mark.parametrize("device", ["cpu", "cuda"])
@pytest.mark.parametrize("kind,adam,hook", _ON_PATH)
def test_a_replaced_hook_on_the_path_is_called_per_step_and_agrees_with_the_fused_kernels(kind, adam, hook, device):
    """The replacement forwards to the built-in, so the composed per-step path computes what the fused kernels
    compute from the same fused-generator normals: equal within the fp32 tolerance; the hook is called where the
    reference calls it (once per step; fit_to_constraints once more after MF's loop, mf_solver.py:591), with its
    tensors on the solver's `device`."""
    t = 25
    want = _solve(_solver(kind, device, iterations=t), _instance(96, device), adam)
    solver = _solver(kind, device, iterations=t)
    builtin, calls = getattr(solver, hook), []

    def forward(*args, **kwargs):
        calls.append([a.device.type for a in args if torch.is_tensor(a)])
        return builtin(*args, **kwargs)

    setattr(solver, hook, forward)
    with pytest.warns(RuntimeWarning, match="composed per-step path"):
        got = _solve(solver, _instance(96, device), adam)
    in_loop = t + (1 if (kind == "mf" and hook == "fit_to_constraints") else 0)
    # (the finalize calls fit_to_constraints / change_variables hooks too: at least the loop's calls)
    assert len(calls) >= in_loop
    assert all(d == device for call in calls for d in call)
    for name in got.variables:
        assert got.variables[name].device.type == device
    _close(got, want)


@pytest.mark.parametrize("case,hook", [
    ("dl_T10", "calculate_drift"), ("mf_T10", "calculate_drift"), ("mf_T10", "fit_to_constraints"),
    ("mf_T60_adamA", "calculate_grads"), ("mf_T60_adamB", "calculate_grads"), ("langevin_T10", "calculate_drift"),
    ("langevin_T60_adamC", "calculate_grads"), ("pl_T10", "calculate_drift"), ("pl_T60_adamA", "calculate_grads"),
    ("pl_T50_noramp", "fit_to_constraints"), ("dl_T50_noramp", "calculate_drift")])
def test_composed_path_reproduces_the_reference_in_replay_mode(case, hook):
    """The composed loops follow the reference's op order, so with the replayed torch CPU stream they land on
    the REFERENCE's own outputs (goldens of test020-100-10) inside the same gates as the fused kernels."""
    from ccvm_amd.problem_classes.boxqp import ProblemInstance
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver
    from ccvm_amd.solvers.algorithms import AdamParameters

    g = golden("test020")
    meta = g.cases[case]
    cls = {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver, "pl": PumpedLangevinSolver}[meta["kind"]]
    inst = ProblemInstance.from_arrays(g.q(), g.v(), name=g.instance["name"], optimal_sol=g.instance["optimal_sol"],
                                       best_sol=g.instance["best_sol"])
    solver = cls(device="cpu", batch_size=meta["batch"])
    solver.noise_mode = "replay"
    solver.parameter_key = {20: dict(meta["params"])}
    builtin = getattr(solver, hook)
    setattr(solver, hook, lambda *a, **k: builtin(*a, **k))
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    kwargs = {"pump_rate_flag": meta["pump_rate_flag"]} if meta["kind"] != "langevin" else {}
    if meta["adam"]:
        kwargs["algorithm_parameters"] = AdamParameters(**meta["adam"])
    torch.manual_seed(meta["seed"])
    with pytest.warns(RuntimeWarning, match="composed per-step path"):
        sol = solver(instance=inst, **kwargs)
    for field in g.fields(case):
        want = g.out(case, field)
        got = sol.objective_values if field == "objective_values" else sol.variables[field]
        tol = 2e-3 if field == "objective_values" else 5e-4 * max(1.0, float(want.abs().max()))
        err = float((got.cpu() - want).abs().max())
        assert err <= tol, f"{case}/{field}: max abs err {err:.3e} > {tol:.1e}"


def test_a_subclass_override_counts_as_a_replaced_hook():
    from ccvm_amd.solvers import LangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS

    seen = []

    class Mine(LangevinSolver):
        def _calculate_drift_boxqp(self, c, lower_limit=0, upper_limit=1, S=1):
            seen.append(tuple(c.shape))
            return torch.zeros_like(c)

    solver = Mine(device="cpu", batch_size=16)
    solver.parameter_key = {96: dict(EXAMPLE_PARAMS["langevin"], iterations=5)}
    solver.noise_seed = 3
    with pytest.warns(RuntimeWarning, match="calculate_drift replaced"):
        sol = _solve(solver, 96)
    assert seen == [(16, 96)] * 5
    assert bool(torch.isfinite(sol.objective_values).all())


def test_dl_fit_to_constraints_hook_is_called_once_after_the_loop():
    """dl_solver.py:567: the only call of the hook in DL's loop function; the fused kernels still run the loop."""
    solver = _solver("dl", iterations=10)
    calls = []

    def halve(c, lower, upper):
        calls.append((tuple(c.shape), lower, upper, c.device.type))
        return torch.clamp(c, lower / 2, upper / 2)

    solver.fit_to_constraints = halve
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # no composed path: calculate_drift is intact
        sol = _solve(solver, 96)
    assert calls == [((48, 96), -1, 1, "cpu")]
    assert float(sol.variables["problem_variables"].abs().max()) <= 0.5


def test_reference_unit_test_solve_success_minimal_inputs():
    """The reference's test_mf_solver.py:249-292 with its inputs verbatim (tests/golden/reference_unit_vectors.json):
    a mocked instance whose V is a (1, N) row vector, fit_to_constraints / change_variables / calculate_grads
    replaced by zero-returning stand-ins, the NON-Adam __call__ (which never calls calculate_grads,
    mf_solver.py:561-572), 15000 iterations at batch 1000."""
    from ccvm_amd.solvers import MFSolver

    vec = reference_unit_vectors()["mf_solver"]
    case, params = vec["solve_minimal_inputs"], vec["parameters"]
    b, n = case["batch_size"], case["problem_size"]
    instance = mock.MagicMock()
    instance.q_matrix = torch.tensor(case["q_matrix"])
    instance.v_vector = torch.tensor(case["v_vector"])
    instance.problem_size = n
    instance.solution_bounds = tuple(case["solution_bounds"])
    instance.compute_energy.return_value = torch.tensor(case["compute_energy_return"])
    instance.optimal_sol = 0.37
    instance.device = case["instance_device"]

    solver = MFSolver(device="cpu", batch_size=b, problem_category="boxqp")
    solver.parameter_key = {n: dict(params)}
    zeros = lambda *args, **kwargs: torch.zeros(b, n)
    solver.fit_to_constraints = zeros
    solver.change_variables = zeros
    solver.calculate_grads = lambda *args, **kwargs: (torch.zeros(b, n), torch.zeros(b, n))
    with pytest.warns(RuntimeWarning, match="fit_to_constraints replaced"):
        solution = solver(instance)

    want = case["expected"]
    assert solution.problem_size == want["problem_size"] and solution.batch_size == want["batch_size"]
    assert solution.objective_values == torch.tensor(want["objective_values"])
    assert solution.iterations == want["iterations"]
    assert solution.solve_time > 0.0 and solution.pp_time == want["pp_time"]
    assert solution.optimal_value == instance.optimal_sol and solution.device == solver.device
    assert set(want["variables_keys"]) <= set(solution.variables)
    assert all(torch.is_tensor(solution.variables[k]) for k in want["variables_keys"])
    # with the measured amplitude forced to zero the mean-field amplitudes only see the V bias
    assert bool(torch.isfinite(solution.variables["mu"]).all())
