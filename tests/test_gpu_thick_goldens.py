"""The reference's OWN output at real batch sizes (tests/golden/thick*.npz: N = 300 / 500 / 600 / 768 at batch 100
over 100 steps, N = 1000 at batch 64 over 50; made by make_golden.py --only-thick from the reference in the build
container) against every kernel family of the engine, through the public API in replay mode: the column-cluster
kernel with three clusters and a ragged last one, the slab kernel with several row groups per cluster, both tile
shapes of the per-step kernel and the persistent tile kernel at the headline size."""
import math

import pytest
import torch

from golden_util import check_noise_checksum, compare_with_thick, golden, thick_cases
from test_gpu_parity import ATOL_OBJ, ATOL_X, _run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "cluster", "slab", "ptile", "tile", "tile2"])
def family(request, monkeypatch):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.delenv("CCVM_AMD_KS", raising=False)
    if request.param in ("cluster", "slab", "ptile"):
        monkeypatch.setenv("CCVM_AMD_KERNEL", request.param)
    elif request.param != "auto":
        monkeypatch.setenv("CCVM_AMD_KERNEL", "tile")
        monkeypatch.setenv("CCVM_AMD_KS", "2" if request.param == "tile2" else "1")
    return request.param


@pytest.mark.parametrize("tag,case", thick_cases())
def test_every_kernel_family_matches_the_reference_at_real_batch_sizes(tag, case, family):
    g = golden(tag)
    meta = g.cases[case]
    n = g.instance["problem_size"]
    if family == "cluster" and n > 768:
        pytest.skip("the column-cluster kernel serves N <= 768")
    if family == "ptile" and n <= 768:
        pytest.skip("the persistent tile kernel serves sizes above N = 768")
    check_noise_checksum(meta, n, meta["batch"])
    sol = _run_case(g, meta)
    gate = math.sqrt(max(n, 20) / 20.0)
    compare_with_thick(g, case, lambda f: sol.objective_values if f == "objective_values" else sol.variables[f],
                       ATOL_X * gate, ATOL_OBJ / 150.0 * gate, label=f"[{family}] ")
    assert abs(sol.best_objective_value - meta["best_objective_value"]) <= 1e-5 * abs(meta["best_objective_value"]) + 1e-4
    slack = 1.0 / meta["batch"] + 1e-9
    for key, frac in meta["solution_performance"].items():
        assert abs(sol.solution_performance[key] - frac) <= slack, (key, sol.solution_performance, frac)
