"""The known answers the reference's OWN unit tests hold (tests/golden/reference_unit_vectors.json: their
inputs and expected values as data, file:line cited there), asserted on the host mirror of the API.
The arithmetic entry points (hooks on the HIP engine) are checked against the same vectors in
tests/test_gpu_api.py; the oracle in tests/test_oracle_golden.py."""
import csv

import torch

from golden_util import golden, reference_unit_vectors


def _solution(objective_values, optimal_value):
    from ccvm_amd.solution import Solution

    # positional, in the reference's field order (ccvm_simulators/tests/test_solution.py:8-22 setUp values)
    return Solution(10, 5, "test_instance", 1, objective_values, 2.0, 3.0, optimal_value, 3.2, 1,
                    [1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 1.0],
                    {"problem_variables": torch.tensor((10, 30, 50))}, "test", "cpu")


def test_solution_statistics_reference_vectors():
    """test_solution.py:100-138, 140-173, 175-220."""
    vec = reference_unit_vectors()["solution_stats"]
    sol = _solution(torch.tensor(vec["objective_values"]), vec["optimal_value"])
    assert sol.solution_performance == vec["expected_solution_performance"]
    sol = _solution(torch.tensor(vec["out_of_range"]["objective_values"]), vec["optimal_value"])
    sol.get_solution_stats()
    assert sol.solution_performance == vec["out_of_range"]["expected_solution_performance"]
    sol = _solution(torch.tensor(vec["metadata"]["objective_values"]), vec["optimal_value"])
    meta = sol.get_metadata_dict()
    assert meta["solution_performance"] == vec["metadata"]["expected_solution_performance"]
    assert meta["best_objective_value"] == vec["metadata"]["expected_best_objective_value"]
    assert meta == {
        "problem_size": 10, "batch_size": 5, "instance_name": "test_instance", "iterations": 1, "solve_time": 2.0,
        "pp_time": 3.0, "optimal_value": 3.2, "best_value": 3.2, "num_frac_values": 1, "evolution_file": "test",
        "solution_vector": [1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 1.0],
        "solution_performance": vec["metadata"]["expected_solution_performance"], "best_objective_value": 3,
    }


def test_problem_instance_reference_vectors(tmp_path):
    """test_problem_instance.py:98-135 (header values of test020-100-10.in), :137-158 (scale_coefs on
    test002.in with a TENSOR scaling factor, broadcasting included)."""
    from ccvm_amd.problem_classes.boxqp import ProblemInstance

    vec = reference_unit_vectors()["problem_instance"]
    g = golden("test020")  # parsed by the reference from test020-100-10.in
    assert g.instance["optimal_sol"] == vec["test020_header"]["optimal_sol"]
    assert g.instance["best_sol"] == vec["test020_header"]["best_sol"]

    path = tmp_path / "test002.in"
    path.write_text("\n".join(vec["test002_file_lines"]) + "\n")
    for instance_type in ("tuning", "test"):
        inst = ProblemInstance(device="cpu", instance_type=instance_type, file_path=str(path))
        assert inst.problem_size == 2
        assert isinstance(inst.optimal_sol, float) and inst.optimal_sol == vec["test020_header"]["optimal_sol"]
        assert isinstance(inst.best_sol, float) and inst.best_sol == vec["test020_header"]["best_sol"]
        assert inst.sol_time_gb > 0 and inst.sol_time_bfgs > 0
        assert torch.equal(inst.q_matrix, torch.tensor(vec["test002_parsed"]["q_matrix"]))   # negated on load
        assert torch.equal(inst.v_vector, torch.tensor(vec["test002_parsed"]["v_vector"]))
    sc = vec["scale_coefs"]
    factor = torch.FloatTensor(sc["scaling_factor"])
    inst.scale_coefs(factor)
    assert torch.equal(inst.scaled_by, factor * 1)
    assert torch.equal(inst.q_matrix, torch.FloatTensor(sc["expected_q_matrix"]))
    assert torch.equal(inst.v_vector, torch.FloatTensor(sc["expected_v_vector"]))
    # test_scale_coefs_multiple_times (:189-207): three times by 10 stacks to 1000
    inst = ProblemInstance(device="cpu", instance_type="tuning", file_path=str(path))
    for _ in range(3):
        inst.scale_coefs(torch.FloatTensor([[10, 10], [10, 10]]))
    assert torch.equal(inst.scaled_by, torch.FloatTensor([[1000, 1000], [1000, 1000]]))


def test_append_samples_to_file_reference_vector(tmp_path):
    """test_mf_solver.py:206-241: keyword arguments, tab-separated rows, mu block then sigma block."""
    from ccvm_amd.solvers import MFSolver

    vec = reference_unit_vectors()["mf_solver"]["append_samples_to_file"]
    solver = MFSolver(device="cpu", batch_size=1000, problem_category="boxqp")
    path = tmp_path / "test_sample_file.txt"
    with open(path, "a") as out:
        solver._append_samples_to_file(mu_sample=torch.tensor(vec["mu_sample"]),
                                       sigma_sample=torch.tensor(vec["sigma_sample"]), evolution_file_object=out)
    with open(path) as fh:
        assert list(csv.reader(fh, delimiter="\t")) == vec["expected_rows"]


def test_mf_parameter_key_reference_vector():
    """test_mf_solver.py:42-61: the valid key set is accepted, an extra key is a ValueError."""
    import pytest

    from ccvm_amd.solvers import MFSolver

    params = reference_unit_vectors()["mf_solver"]["parameters"]
    solver = MFSolver(device="cpu", batch_size=1000, problem_category="boxqp")
    solver.parameter_key = {2: dict(params)}
    assert solver.parameter_key == {2: params}
    with pytest.raises(ValueError):
        solver.parameter_key = {2: dict(params, invalid_key=1)}
