"""Run the REFERENCE's own in-scope unit tests against this package (build container only: the reference
tree never travels to the GPU box, and nothing of it is copied here).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/run_reference_unit_tests.py

The reference's test modules for the solvers, the problem instance, the Solution and the post-processor
factory (`ccvm_simulators/tests/unit/{solvers,problem_classes}`, `tests/test_solution.py`,
`tests/unit/postprocessor/test_factory.py`) are linked -- symbolic links in a temporary directory, next to a
link to their data directory -- and run with `ccvm_simulators` resolving to THIS repository's alias package.
Every failure is put in one of three bins:

  out of scope   the machine energy / time models (test classes `*MachineEnergy`, `*MachineTime`, methods
                 `*_machine_energy_*` / `*_machine_time_*`: plotting-side bookkeeping, SURVEY section 2 OUT) -- expected;
  needs a GPU    the test reaches the engine (EngineUnavailable: this container has no MI355X); its inputs and
                 expected values are transcribed into tests/golden/reference_unit_vectors.json and asserted on the
                 GPU box by tests/test_gpu_api.py / tests/test_gpu_hooks.py -- expected here;
  boundary       anything else: a host-side mismatch with the reference's API.  The script exits non-zero.

With a GPU present the second bin must be empty too.
"""
import os
import re
import subprocess
import sys
import tempfile
import xml.etree.ElementTree as ET

REFERENCE = os.environ.get("CCVM_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TESTS = os.path.join(REFERENCE, "ccvm_simulators", "tests")
MODULES = [
    "unit/solvers/test_ccvm_solver.py", "unit/solvers/test_dl_solver.py", "unit/solvers/test_mf_solver.py",
    "unit/solvers/test_langevin_solver.py", "unit/solvers/test_pumped_langevin_solver.py",
    "unit/problem_classes/test_problem_instance.py", "unit/postprocessor/test_factory.py", "test_solution.py",
]
# test classes TestCCVMSolverMachineEnergy / TestCCVMSolverMachineTime, and the per-solver
# test_{optics,fpga,...}_machine_{energy,time}_* methods
OUT_OF_SCOPE = re.compile(r"MachineEnergy|MachineTime|_machine_energy|_machine_time")


def main():
    if not os.path.isdir(TESTS):
        print(f"no reference tree at {REFERENCE}: nothing to run (this script is for the build container)")
        return 0
    with tempfile.TemporaryDirectory(prefix="ccvm_reftests_") as tmp:
        base = os.path.join(tmp, "tests")
        for rel in MODULES:
            dst = os.path.join(base, rel)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            os.symlink(os.path.join(TESTS, rel), dst)
        os.symlink(os.path.join(TESTS, "data"), os.path.join(base, "data"))  # their files resolve ../../data
        report = os.path.join(tmp, "report.xml")
        env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
        # -c /dev/null: neither repository's pytest configuration; importlib mode: no sys.path insertion, so
        # `ccvm_simulators` can only come from PYTHONPATH = this repository
        cmd = [sys.executable, "-c",
               "import sys; sys.path[:] = [p for p in sys.path if p not in ('', '.')]; import pytest; "
               f"sys.exit(pytest.main(['-q', '-c', '/dev/null', '--rootdir', {tmp!r}, '--import-mode=importlib', "
               f"'-p', 'no:cacheprovider', '--junitxml', {report!r}, {base!r}]))"]
        run = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True)
        if not os.path.exists(report):
            print(run.stdout[-4000:], run.stderr[-4000:])
            return 2
        cases = ET.parse(report).getroot().iter("testcase")
        bins = {"passed": [], "out of scope": [], "needs a GPU": [], "boundary": []}
        for case in cases:
            name = f"{case.get('classname', '').split('.')[-1]}::{case.get('name')}"
            bad = case.find("failure") if case.find("failure") is not None else case.find("error")
            if bad is None:
                bins["passed"].append(name)
                continue
            text = (bad.get("message") or "") + (bad.text or "")
            if OUT_OF_SCOPE.search(name):
                bins["out of scope"].append(name)
            elif "EngineUnavailable" in text:
                bins["needs a GPU"].append(name)
            else:
                bins["boundary"].append(name + "\n      " + (bad.get("message") or "").splitlines()[0][:200])
    where = __import__("ccvm_simulators")
    assert os.path.realpath(where.__file__).startswith(os.path.realpath(ROOT)), where.__file__
    for key in ("passed", "out of scope", "needs a GPU", "boundary"):
        print(f"{key}: {len(bins[key])}")
        if key != "passed":
            for name in bins[key]:
                print("   ", name)
    return 1 if bins["boundary"] else 0


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.exit(main())
