"""Host-side logic of the drop-in boundary (no GPU): parameter_key validation, error
types/messages, instance parsing, Solution statistics, evolution-sample bookkeeping.
Mirrors what the reference's own unit tests check for these surfaces
(tests/unit/solvers/test_ccvm_solver.py:57-102, test_mf_solver.py:42-61, :206-241, :294-315;
tests/unit/problem_classes/test_problem_instance.py:98-268; tests/test_solution.py:140-220)."""
import io
import os

import pytest
import torch

from golden_util import golden


def solvers():
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver

    return {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver, "pl": PumpedLangevinSolver}


def test_reference_import_paths_resolve():
    from ccvm_simulators.post_processor.factory import PostProcessorFactory  # noqa: F401
    from ccvm_simulators.problem_classes.boxqp import ProblemInstance
    from ccvm_simulators.solution import Solution  # noqa: F401
    from ccvm_simulators.solvers import CCVMSolver, DLSolver, MachineType  # noqa: F401
    from ccvm_simulators.solvers.algorithms import AdamParameters  # noqa: F401

    import ccvm_amd.solvers as s

    assert DLSolver is s.DLSolver and ProblemInstance.__module__.startswith("ccvm_amd")
    # the reference's per-class module paths, in both import spellings
    import ccvm_simulators.post_processor.grad_descent as gd
    import ccvm_simulators.problem_classes.boxqp.problem_instance as pi
    import ccvm_simulators.solvers.dl_solver as dl
    import ccvm_simulators.solvers.pumped_langevin_solver as pl
    from ccvm_simulators.solvers.ccvm_solver import DeviceType  # noqa: F401
    from ccvm_simulators.solvers.langevin_solver import LangevinSolver  # noqa: F401
    from ccvm_simulators.solvers.mf_solver import MFSolver  # noqa: F401

    assert dl.DLSolver is DLSolver and pl.PumpedLangevinSolver is s.PumpedLangevinSolver
    assert pi.ProblemInstance is ProblemInstance and gd.PostProcessorGradDescent is not None


def test_device_and_category_validation():
    for cls in solvers().values():
        with pytest.raises(ValueError, match="Given device is not available"):
            cls(device="tpu")
        with pytest.raises(ValueError, match="not a valid problem category"):
            cls(device="cpu", problem_category="maxcut")
        s = cls(device="cpu")
        assert s.parameter_key is None and s.is_tuned is False and s.batch_size == 1000
        assert callable(s.calculate_drift) and callable(s.change_variables)


@pytest.mark.parametrize("kind,keys", [
    ("dl", {"pump", "dt", "iterations", "noise_ratio", "feedback_scale"}),
    ("mf", {"pump", "feedback_scale", "j", "S", "dt", "iterations"}),
    ("langevin", {"dt", "S", "iterations", "sigma", "feedback_scale"}),
    ("pl", {"pump", "dt", "S", "iterations", "sigma", "feedback_scale"}),
])
def test_parameter_key_requires_the_exact_key_set(kind, keys):
    s = solvers()[kind](device="cpu")
    good = {20: {k: 1.0 for k in keys}}
    s.parameter_key = good
    assert s.parameter_key is good
    for bad in ({20: {k: 1.0 for k in list(keys)[1:]}}, {20: dict(good[20], extra=1)}):
        with pytest.raises(ValueError, match="The parameter key is not valid for this solver"):
            s.parameter_key = bad
    s.tune([])
    assert s.is_tuned
    s.parameter_key = good
    assert not s.is_tuned


def test_scaling_factor_formula():
    q = torch.tensor([[1.0, -2.0], [3.0, -4.0]])
    for kind, mult in (("dl", 0.2), ("mf", 0.05), ("langevin", 0.05), ("pl", 0.05)):
        f = solvers()[kind](device="cpu").get_scaling_factor(q)
        assert f.ndim == 0 and abs(float(f) - (10.0**0.5) * mult) < 1e-6
    g = golden("test020")
    assert abs(float(solvers()["dl"](device="cpu").get_scaling_factor(g.q())) - 8.913559) < 1e-5


def test_call_errors_before_any_compute():
    from ccvm_amd.problem_classes.boxqp import ProblemInstance
    from ccvm_amd.solvers import DLSolver

    inst = ProblemInstance.from_arrays(torch.eye(4), torch.ones(4), device="cpu")
    s = DLSolver(device="cuda", batch_size=4)
    s.parameter_key = {4: {"pump": 2.0, "dt": 0.01, "iterations": 3, "noise_ratio": 2, "feedback_scale": 1}}
    with pytest.raises(ValueError, match=r"The device type of the instance \(cpu\) and the solver \(cuda\) must match"):
        s(instance=inst)
    s = DLSolver(device="cpu", batch_size=4)
    s.parameter_key = {5: {"pump": 2.0, "dt": 0.01, "iterations": 3, "noise_ratio": 2, "feedback_scale": 1}}
    with pytest.raises(KeyError, match="for the given instance size is not defined"):
        s(instance=inst)


def test_adam_parameters_validation():
    from ccvm_amd.solvers.algorithms import AdamParameters

    p = AdamParameters(alpha=0.001, beta1=0.9, beta2=0.999, add_assign=False)
    assert p.to_dict() == {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False}
    assert AdamParameters().to_dict() == {"alpha": 0.1, "beta1": 0.9, "beta2": 0.999, "add_assign": True}
    AdamParameters(beta2=1.0)
    for bad in (dict(alpha=-1), dict(beta1=0), dict(beta1=1), dict(beta2=0), dict(beta2=1.5)):
        with pytest.raises(ValueError, match="AdamAlgorithm"):
            AdamParameters(**bad)


def test_instance_round_trip_and_reference_parse(tmp_path):
    """save_instance -> load_instance is the identity; and (where the reference tree is
    mounted) its shipped files parse to exactly the arrays the reference's parser produced."""
    from ccvm_amd.problem_classes.boxqp import ProblemInstance
    from ccvm_amd.workloads import synthetic_instance

    inst = synthetic_instance(37, seed=5)
    inst.optimal_sol, inst.best_sol, inst.solution_vector = 12.5, 12.25, [0.0, 1.0, 0.5]
    path = str(tmp_path / "syn037-100-5.in")
    inst.save_instance(path)
    back = ProblemInstance(instance_type="test", file_path=path)
    assert back.problem_size == 37 and back.name == "syn037-100-5"
    assert torch.equal(back.q_matrix, inst.q_matrix) and torch.equal(back.v_vector, inst.v_vector)
    assert (back.optimal_sol, back.best_sol, back.solution_vector) == (12.5, 12.25, [0.0, 1.0, 0.5])
    assert back.scaled_by == 1 and back.optimality is False

    for tag in ("test020", "tuningH020"):
        g = golden(tag)
        ref_file = os.path.join("/root/reference", g.instance["source"])
        if not os.path.exists(ref_file):
            continue
        mine = ProblemInstance(instance_type="test", file_path=ref_file)
        assert torch.equal(mine.q_matrix, g.q()) and torch.equal(mine.v_vector, g.v())
        for key in ("problem_size", "optimal_sol", "best_sol", "optimality", "sol_time_gb", "sol_time_bfgs",
                    "num_frac_values", "solution_vector", "name"):
            assert getattr(mine, key) == g.instance[key], key


def test_instance_validation_and_scaling():
    from ccvm_amd.problem_classes.boxqp import ProblemInstance

    with pytest.raises(ValueError, match="instance_type must be tuning or test"):
        ProblemInstance(instance_type="other")
    with pytest.raises(ValueError, match="size 2"):
        ProblemInstance(solution_bounds=(0, 1, 2))
    with pytest.raises(ValueError, match="less than"):
        ProblemInstance(solution_bounds=(1.0, 1.0))
    with pytest.raises(Exception, match="No file path specified"):
        ProblemInstance().load_instance()
    # a missing file is FileNotFoundError, as in the reference (problem_instance.py:154 opens the file outside its
    # try block; its test_problem_instance.py:78-86 asserts the type); a malformed one "Error reading instance file"
    with pytest.raises(FileNotFoundError):
        ProblemInstance(file_path="/test_instances/invalid.in")
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".in") as bad:
        bad.write("not\ta\theader\n")
        bad.flush()
        with pytest.raises(Exception, match="Error reading instance file"):
            ProblemInstance(file_path=bad.name)
    inst = ProblemInstance.from_arrays(torch.full((2, 2), 8.0), torch.tensor([4.0, 2.0]))
    inst.scale_coefs(2.0)
    inst.scale_coefs(torch.tensor(2.0))
    assert torch.equal(inst.q_matrix, torch.full((2, 2), 2.0)) and float(inst.scaled_by) == 4.0
    assert torch.equal(inst.v_vector, torch.tensor([1.0, 0.5]))
    assert ProblemInstance(name="abc").name == "abc"


def test_solution_statistics_known_answers():
    from ccvm_amd.solution import Solution, r99

    sol = Solution(problem_size=2, batch_size=3, instance_name="t", iterations=5,
                   objective_values=-torch.tensor([100.0, 98.5, 93.0]), solve_time=0.5, pp_time=0.0,
                   optimal_value=100.0, best_value=99.0, num_frac_values=0, solution_vector=[],
                   variables={"problem_variables": torch.zeros(3, 2)})
    assert sol.best_objective_value == 100.0
    assert sol.solution_performance == {
        "optimal": 0.3333, "one_percent": 0.3333, "two_percent": 0.6667, "three_percent": 0.6667,
        "four_percent": 0.6667, "five_percent": 0.6667, "ten_percent": 1.0}
    meta = sol.get_metadata_dict()
    assert "objective_values" not in meta and "variables" not in meta and meta["iterations"] == 5
    assert list(meta)[:4] == ["problem_size", "batch_size", "instance_name", "iterations"]
    assert abs(sol.tts99() - 0.5 * r99(0.3333)) < 1e-12 and r99(1.0) == 1.0 and r99(0.0) == float("inf")


def test_evolution_sample_bookkeeping_and_file_format():
    from ccvm_amd.solvers import DLSolver, MFSolver
    from ccvm_amd.solvers.base import num_samples, sample_points

    assert sample_points(10, 3) == [0, 3, 6, 9] and num_samples(10, 3) == 5  # reference over-allocates by one
    assert sample_points(10, 5) == [0, 5, 9] and num_samples(10, 5) == 3
    assert sample_points(4, 1) == [0, 1, 2, 3] and num_samples(4, 1) == 5
    block = torch.tensor([[1.23456, -2.0], [0.00004, 3.5]])
    out = io.StringIO()
    MFSolver(device="cpu")._append_samples_to_file(mu_sample=block, sigma_sample=block, evolution_file_object=out)
    assert out.getvalue() == "1.2346\t-2.0\n0.0\t3.5\n" * 2            # test_mf_solver.py:206-241 format
    out = io.StringIO()
    DLSolver(device="cpu")._append_samples_to_file(block, block, evolution_file_object=out)
    assert out.getvalue() == "1.2346\t-2.0\t\n0.0\t3.5\t\n" * 2        # dl_solver.py:268-272: trailing tab


def test_post_processor_factory_names():
    from ccvm_amd.post_processor.adam import PostProcessorAdam
    from ccvm_amd.post_processor.factory import PostProcessorFactory
    from ccvm_amd.post_processor.grad_descent import PostProcessorGradDescent

    assert isinstance(PostProcessorFactory.create_postprocessor("Adam"), PostProcessorAdam)
    assert isinstance(PostProcessorFactory.create_postprocessor("grad-descent"), PostProcessorGradDescent)
    with pytest.raises(AssertionError, match="Method type is not valid"):
        PostProcessorFactory.create_postprocessor("newton")
    with pytest.raises(TypeError, match="parameter c must be a tensor"):
        PostProcessorAdam().postprocess([1.0], torch.eye(1), torch.ones(1))


def test_wrong_algorithm_parameters_type_is_rejected_before_the_engine_is_touched():
    """`algorithm_parameters` of an unsupported type -> ValueError("Solver option type ... is not
    supported.") as in the reference (dl_solver.py:924-927, mf_solver.py, langevin_solver.py) -- raised
    by the host-side validation, so it is the same with or without a GPU."""
    g = golden("test020")
    from ccvm_amd.problem_classes.boxqp import ProblemInstance

    keys = {"dl": "dl_T1", "mf": "mf_T1", "langevin": "langevin_T1", "pl": "pl_T1"}
    for kind, cls in solvers().items():
        inst = ProblemInstance.from_arrays(g.q(), g.v())
        solver = cls(device="cpu", batch_size=4)
        solver.parameter_key = {20: dict(g.cases[keys[kind]]["params"])}
        with pytest.raises(ValueError, match="Solver option type <class 'str'> is not supported."):
            solver(instance=inst, algorithm_parameters="adam")


def test_design_tables_are_generated_from_the_committed_profiles():
    """DESIGN.md's roofline table is written by tools/make_design_tables.py from profiles/rNN_bench*.json and the PMC
    summaries (numbers not typed by hand): the committed file must be what the tool writes."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "make_design_tables.py"), "--check"])
    assert run.returncode == 0, "run `python tools/make_design_tables.py` and commit DESIGN.md"


def test_schedule_tables_are_shared_only_between_runs_with_the_same_scalars():
    """engine._schedule_key: every scalar of the parameter dictionaries is part of the key (a one-element tensor counts as
    its value); a run with a per-variable / per-element parameter, or one that is no number at all, gets no key -- it
    makes its own table."""
    import numpy as np

    from ccvm_amd.engine import _schedule_key

    base = dict(dt=0.002, sigma=0.5, feedback_scale=1.0, S=1.0, use_pump=False, pump=None)
    key = _schedule_key(0, 7, "langevin", 1500, 0.0, 1.0, base, None)
    assert key is not None and key == _schedule_key(0, 7, "langevin", 1500, 0.0, 1.0, dict(base), None)
    assert key == _schedule_key(0, 7, "langevin", 1500, 0.0, 1.0, dict(base, S=torch.tensor(1.0), dt=np.float32(0.002).astype(np.float64) * 0 + 0.002), None)
    for other in (dict(base, S=2.0), dict(base, S=torch.tensor([2.0])), dict(base, use_pump=True), dict(base, pump=1.5)):
        assert _schedule_key(0, 7, "langevin", 1500, 0.0, 1.0, other, None) != key
    assert _schedule_key(0, 8, "langevin", 1500, 0.0, 1.0, base, None) != key      # another stream
    assert _schedule_key(1, 7, "langevin", 1500, 0.0, 1.0, base, None) != key      # another device
    assert _schedule_key(0, 7, "langevin", 1501, 0.0, 1.0, base, None) != key      # another run length
    assert _schedule_key(0, 7, "langevin", 1500, 0.0, 2.0, base, None) != key      # other bounds
    adam = dict(alpha=0.001, beta1=0.9, beta2=0.999, add_assign=False)
    with_adam = _schedule_key(0, 7, "langevin", 1500, 0.0, 1.0, base, adam)
    assert with_adam is not None and with_adam != key
    assert _schedule_key(0, 7, "langevin", 1500, 0.0, 1.0, base, dict(adam, beta2=1.0)) != with_adam
    for odd in (torch.ones(20), torch.ones(4, 20), np.ones(20), "one"):
        assert _schedule_key(0, 7, "langevin", 1500, 0.0, 1.0, dict(base, S=odd), None) is None
