"""Differential check of the host-side API against the REFERENCE itself (build container only: the
reference tree never travels to the GPU box, and nothing of it is copied here).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/compare_host_api.py

Everything here runs on the CPU and needs no GPU: instance parsing of EVERY .in file the reference
ships (matrices, header fields, both instance types), scaling factors and scale_coefs, Solution success
statistics on random objective values, AdamParameters validation, parameter_key / constructor / call-time
errors.  Prints one line per check; exits non-zero on any difference.  (Set order inside the
"Expected keys: {...}" messages is process-dependent in both packages and is compared as a set.)
"""
import glob
import os
import random
import re
import sys

REFERENCE = os.environ.get("CCVM_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REFERENCE)
sys.path.insert(1, ROOT)
sys.dont_write_bytecode = True

import torch  # noqa: E402

import ccvm_simulators as reference_pkg  # noqa: E402

assert os.path.realpath(reference_pkg.__file__).startswith(os.path.realpath(REFERENCE))
from ccvm_simulators.problem_classes.boxqp import ProblemInstance as RefInstance  # noqa: E402
from ccvm_simulators.solution import Solution as RefSolution  # noqa: E402
from ccvm_simulators.solvers import DLSolver as RDL, LangevinSolver as RL, MFSolver as RMF  # noqa: E402
from ccvm_simulators.solvers import PumpedLangevinSolver as RPL  # noqa: E402
from ccvm_simulators.solvers.algorithms import AdamParameters as RefAdam  # noqa: E402

from ccvm_amd.problem_classes.boxqp import ProblemInstance as OurInstance  # noqa: E402
from ccvm_amd.solution import Solution as OurSolution  # noqa: E402
from ccvm_amd.solvers import DLSolver as ODL, LangevinSolver as OL, MFSolver as OMF  # noqa: E402
from ccvm_amd.solvers import PumpedLangevinSolver as OPL  # noqa: E402
from ccvm_amd.solvers.algorithms import AdamParameters as OurAdam  # noqa: E402

failures = 0


def report(name, ok, detail=""):
    global failures
    failures += 0 if ok else 1
    print(f"{'ok  ' if ok else 'DIFF'} {name} {detail}")


def outcome(fn):
    try:
        fn()
        return ("ok", "")
    except Exception as exc:  # the type and message are what is compared
        msg = re.sub(r"\{[^{}]*\}", lambda m: "{" + ",".join(sorted(m.group(0)[1:-1].replace(" ", "").split(","))) + "}",
                     str(exc))
        return (type(exc).__name__, msg)


files = sorted(glob.glob(os.path.join(REFERENCE, "examples/benchmarking_instances/**/*.in"), recursive=True))
files += sorted(glob.glob(os.path.join(REFERENCE, "ccvm_simulators/tests/data/**/*.in"), recursive=True))
fields = ("problem_size", "optimal_sol", "best_sol", "optimality", "sol_time_gb", "sol_time_bfgs", "num_frac_values",
          "solution_vector", "name", "scaled_by", "solution_bounds", "file_delimiter", "instance_type")
bad = 0
for path in files:
    for kind in ("test", "tuning"):
        a = RefInstance(instance_type=kind, file_path=path, device="cpu")
        b = OurInstance(instance_type=kind, file_path=path, device="cpu")
        same = torch.equal(a.q_matrix, b.q_matrix) and torch.equal(a.v_vector, b.v_vector)
        same = same and all(getattr(a, f, None) == getattr(b, f, None) for f in fields)
        bad += 0 if same else 1
report(f"parsing of {len(files)} instance files x 2 instance types", bad == 0, f"({bad} differ)")

a = RefInstance(instance_type="test", file_path=files[0], device="cpu")
b = OurInstance(instance_type="test", file_path=files[0], device="cpu")
for ref_cls, our_cls in ((RDL, ODL), (RMF, OMF), (RL, OL), (RPL, OPL)):
    fr, fo = ref_cls(device="cpu").get_scaling_factor(a.q_matrix), our_cls(device="cpu").get_scaling_factor(b.q_matrix)
    report(f"get_scaling_factor {ref_cls.__name__}", torch.equal(fr, fo))
a.scale_coefs(fr)
b.scale_coefs(fo)
report("scale_coefs", torch.equal(a.q_matrix, b.q_matrix) and torch.equal(a.v_vector, b.v_vector)
       and bool(a.scaled_by == b.scaled_by))

gen, rng, diff = torch.Generator().manual_seed(0), random.Random(1), 0
for _ in range(200):
    bsz, opt = rng.choice([1, 7, 100, 1000]), rng.choice([130.7, 1.0, 986.0, -5.0])
    obj = -(opt * (1 - torch.rand(bsz, generator=gen) * rng.choice([0.0005, 0.001, 0.02, 0.2])))
    kw = dict(problem_size=20, batch_size=bsz, instance_name="x", iterations=10, objective_values=obj, solve_time=0.1,
              pp_time=0.0, optimal_value=opt, best_value=opt, num_frac_values=0, solution_vector=[],
              variables={"problem_variables": torch.zeros(bsz, 20)})
    clone = lambda d: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
    r, o = RefSolution(**clone(kw)), OurSolution(**clone(kw))
    diff += int(r.solution_performance != o.solution_performance or r.best_objective_value != o.best_objective_value)
report("Solution statistics on 200 random batches", diff == 0, f"({diff} differ)")

for args in (dict(alpha=0.001, beta1=0.9, beta2=0.999, add_assign=False), dict(alpha=-1, beta1=0.9, beta2=0.999, add_assign=False),
             dict(alpha=0.1, beta1=1.5, beta2=0.999, add_assign=False), dict(alpha=0.1, beta1=0.5, beta2=-0.1, add_assign=True),
             dict(alpha="a", beta1=0.5, beta2=0.1, add_assign=True), dict(alpha=0.1, beta1=0.5, beta2=0.1, add_assign="yes")):
    res = []
    for cls in (RefAdam, OurAdam):
        try:
            res.append(("ok", cls(**args).to_dict()))
        except Exception as exc:
            res.append((type(exc).__name__, str(exc)))
    report(f"AdamParameters({args})", res[0] == res[1])

keys = ({"pump": 8.0, "feedback_scale": 100, "dt": 0.001, "iterations": 5, "noise_ratio": 10},
        {"pump": 0.0, "feedback_scale": 4000, "j": 5.0, "S": 20.0, "dt": 0.0025, "iterations": 5},
        {"dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0, "iterations": 5},
        {"pump": 2.0, "dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0, "iterations": 5})
for (ref_cls, our_cls), key in zip(((RDL, ODL), (RMF, OMF), (RL, OL), (RPL, OPL)), keys):
    res = []
    for cls, inst_cls in ((ref_cls, RefInstance), (our_cls, OurInstance)):
        s = cls(device="cpu", batch_size=4)
        inst = inst_cls(instance_type="test", file_path=files[0], device="cpu")
        moved = inst_cls(instance_type="test", file_path=files[0], device="cpu")
        moved.device = "cuda"
        checks = [outcome(lambda: setattr(s, "parameter_key", {20: {"pump": 1.0}})),
                  outcome(lambda: setattr(s, "parameter_key", {20: {"bogus": 1, "dt": 1}})),
                  outcome(lambda: cls(device="tpu")), outcome(lambda: cls(device="cpu", problem_category="maxcut"))]
        s.parameter_key = {999: key}
        checks.append(outcome(lambda: s(instance=inst)))                       # size not in the key
        s.parameter_key = {inst.problem_size: key}
        checks.append(outcome(lambda: s(instance=moved)))                      # device mismatch
        checks.append(outcome(lambda: s(instance=inst, algorithm_parameters="adam")))  # wrong option type
        res.append(checks)
    report(f"{ref_cls.__name__}: setter, constructor and call-time errors", res[0] == res[1],
           "" if res[0] == res[1] else f"\n  ref {res[0]}\n  our {res[1]}")

# a missing instance file: the same exception TYPE (the reference opens the file outside its try block,
# problem_instance.py:154; its test_problem_instance.py:78-86 asserts FileNotFoundError)
res = [outcome(lambda c=c: c(instance_type="test", file_path="/test_instances/invalid.in", device="cpu"))[0]
       for c in (RefInstance, OurInstance)]
report("missing instance file", res[0] == res[1] == "FileNotFoundError", f"({res})")

# which hooks each loop CALLS: read off the reference's source (the `self.<hook>(` call sites inside _solve /
# _solve_adam) against the table this package consults to decide whether a replaced hook matters
# (CCVMSolver._LOOP_HOOKS).  A hook outside the tuple is never looked at, one inside forces the composed path.
import inspect  # noqa: E402

for ref_cls, our_cls in ((RDL, ODL), (RMF, OMF), (RL, OL), (RPL, OPL)):
    for adam, fn in ((False, "_solve"), (True, "_solve_adam")):
        called = set(re.findall(r"self\.(calculate_drift|calculate_grads|fit_to_constraints)\(",
                                inspect.getsource(getattr(ref_cls, fn))))
        ours = set(our_cls._LOOP_HOOKS[adam])
        if ref_cls is RDL:  # DL calls fit_to_constraints once AFTER the loop (dl_solver.py:567): honoured by
            called.discard("fit_to_constraints")  # DLSolver._solve on the fused path too, not a loop hook
        report(f"{ref_cls.__name__}.{fn}: hooks on the loop's path", called == ours, f"(ref {called}, our {ours})")

# a (1, N) row vector as V (the reference's test_mf_solver.py:255): the reference only broadcasts V, so it
# works there; here the staging accepts any tensor of N values (engine.DeviceProblem) -- host-side part of that
# contract: the shape check itself
import ccvm_amd.engine as our_engine  # noqa: E402

src = inspect.getsource(our_engine.DeviceProblem.__init__)
report("DeviceProblem takes V by element count", "v_vector.numel() != self.n" in src and "reshape(-1)" in src)

print("differences:", failures)
sys.exit(1 if failures else 0)
