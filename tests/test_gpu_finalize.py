"""GPU tests of the fused device-side finalize (ccvm_finalize / ccvm_objective_stats: SURVEY.md K5,
section 8 f-2): clamp -> change of variables -> [post-processor] -> energy -> success statistics on the
pitched state, against (1) the separate kernels of round 1 (bit for bit) and (2) the reference's
Solution arithmetic (solution.py:65-146) restated on the host."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_stats(obj, optimal):
    """solution.py:65-85 + :87-146 on the host, in torch fp32 like the reference."""
    from ccvm_amd.solution import GAP_THRESHOLDS

    found = -obj
    gap = (optimal - found) * 100 / torch.abs(found)
    return torch.max(found).item(), [int((gap <= thr).sum()) for _, thr in GAP_THRESHOLDS]


@pytest.mark.parametrize("n,b,vector_s", [(20, 100, False), (257, 33, False), (1000, 1000, False), (96, 70, True),
                                          (500, 1000, False), (2000, 512, False)])
def test_finalize_equals_separate_kernels_and_host_statistics(n, b, vector_s):
    from ccvm_amd import _lib, engine
    from ccvm_amd.workloads import scaled_qv

    q, v, f = scaled_qv(n, "dl")
    g = torch.Generator().manual_seed(n + b)
    y = (torch.rand((b, n), generator=g) * 3.0 - 1.5)
    S = (0.6 + torch.rand(n, generator=g)) if vector_s else 1.25
    lo, hi = -0.5, 2.0
    # round-1 path: one pack / kernel / unpack hop per step
    yc = engine.clamp(y, -S, S) if vector_s else engine.clamp(y, -1.0, 1.0)
    x_ref = engine.change_variables(yc, S, lo, hi)
    obj_ref = engine.energy(x_ref, q, v, float(f))
    optimal = float((-obj_ref).max()) * 0.999  # so that the thresholds split the batch
    best_ref, counts_ref = _host_stats(obj_ref, optimal)

    prob = engine.device_problem(q, v)
    dev = prob.device
    with torch.cuda.device(dev):
        state = engine.pack(y.to(dev), engine.rows_of(b), prob.ld)
        x = torch.zeros_like(state)
        scored = engine.finalize_pitched(prob, state, x, b, n, S, lo, hi, float(f), optimal,
                                         clamp=None if vector_s else (-1.0, 1.0))
        if vector_s:  # per-variable clamp goes through the clamp flag with s_cols
            state = engine.pack(y.to(dev), engine.rows_of(b), prob.ld)
            scored = engine.finalize_pitched(prob, state, x, b, n, S, lo, hi, float(f), optimal, clamp=(0.0, 0.0))
    assert torch.equal(scored.objective_values.cpu(), obj_ref)          # bit for bit
    assert torch.equal(scored.variables.cpu(), x_ref)
    assert torch.equal(state[:b, :n].cpu(), yc)                         # clamped in place
    assert float(state[b:].abs().max() if state.shape[0] > b else 0.0) == 0.0   # padding stays zero
    assert float(x[:, n:].abs().max() if x.shape[1] > n else 0.0) == 0.0
    best, within, rows, nonfinite = engine.read_stats(scored.stats)
    assert rows == b and nonfinite == 0
    assert best == best_ref and within == counts_ref
    assert 0 < within[0] < b or b < 50  # the thresholds do discriminate


def test_objective_stats_match_solution_py_on_adversarial_values():
    """Zero, negative, infinite and NaN objective values: the device counters follow the reference's
    fp32 arithmetic (division by |0| -> inf, NaN counts nowhere and makes the best value NaN)."""
    from ccvm_amd import engine

    obj = torch.tensor([-130.7142, -130.5, -129.0, -100.0, 0.0, 5.0, -1e30, float("inf"), -131.0, -130.71],
                       dtype=torch.float32)
    for optimal in (130.714187, 152.602291, 0.0, -3.0):
        best, within, rows, nonfinite = engine.read_stats(engine.objective_stats(obj, optimal))
        want_best, want = _host_stats(obj, optimal)
        assert rows == obj.numel() and nonfinite == 1
        assert within == want and best == want_best, (optimal, within, want)
    with_nan = torch.cat([obj, torch.tensor([float("nan")])])
    best, within, rows, nonfinite = engine.read_stats(engine.objective_stats(with_nan, 130.714187))
    want_best, want = _host_stats(with_nan, 130.714187)
    assert best != best and want_best != want_best and within == want and nonfinite == 2
    big = -(torch.rand(200_003, generator=torch.Generator().manual_seed(1)) * 20 + 120)
    best, within, rows, _ = engine.read_stats(engine.objective_stats(big, 139.9))
    want_best, want = _host_stats(big, 139.9)
    assert rows == 200_003 and within == want and best == want_best


def test_reference_success_fraction_vectors_on_device():
    """The reference's own known answers for Solution.get_solution_stats
    (ccvm_simulators/tests/test_solution.py:140-173), evaluated by ccvm_objective_stats."""
    import json
    import os

    from conftest import ROOT
    from ccvm_amd import engine
    from ccvm_amd.solution import fractions_from_counts

    with open(os.path.join(ROOT, "tests", "golden", "reference_unit_vectors.json")) as fh:
        vec = json.load(fh)["solution_stats"]
    obj = torch.tensor(vec["objective_values"], dtype=torch.float32)
    _, within, rows, _ = engine.read_stats(engine.objective_stats(obj, vec["optimal_value"]))
    assert fractions_from_counts(within, rows) == vec["expected_solution_performance"]


@pytest.mark.parametrize("kind", ["dl", "mf", "langevin", "pl"])
@pytest.mark.parametrize("post", [None, "adam", "grad-descent"])
def test_fused_finalize_equals_the_hook_path(kind, post):
    """Solver __call__ through ccvm_finalize vs the hook-calling path (taken when a hook is replaced,
    here by a pass-through wrapper): same objective values bit for bit, same variables, same
    statistics -- including the DL quirk that applies change_variables before AND after a
    post-processor (dl_solver.py:936-958)."""
    from golden_util import golden
    from test_gpu_parity import _instance, _solver_for

    g = golden("test020")
    meta = g.cases[f"{kind}_T100"]
    out = []
    for wrap in (False, True):
        solver = _solver_for(kind, 64, dl_S=0.8 if kind == "dl" else None)
        inst = _instance(g)
        solver.parameter_key = {20: dict(meta["params"], iterations=40)}
        inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
        if wrap:
            builtin = solver.change_variables
            solver.change_variables = lambda *a, **k: builtin(*a, **k)  # replaced hook: no fused finalize
        torch.manual_seed(5)
        out.append(solver(instance=inst, post_processor=post))
    fused, hooks = out
    assert fused.device_objective_values is not None and hooks.device_objective_values is None
    assert torch.equal(fused.objective_values, hooks.objective_values)
    for key in hooks.variables:
        assert torch.equal(fused.variables[key], hooks.variables[key]), key
    assert fused.best_objective_value == hooks.best_objective_value
    assert fused.solution_performance == hooks.solution_performance


def test_finalize_abi_argument_checks(hip_lib):
    from ccvm_amd import _lib, engine

    dev = engine.gpu_device()
    x = torch.zeros((64, 128), device=dev)
    obj = torch.zeros((10,), device=dev)
    ws = torch.zeros((4096,), dtype=torch.uint8, device=dev)
    fp = _lib.FinalizeParams()
    fp.S, fp.lower, fp.upper, fp.scaled_by, fp.optimal_value, fp.change_variables = 1.0, 0.0, 1.0, 1.0, 1.0, 1
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    call = lambda ld, wsb: hip_lib.ccvm_finalize(P(x), P(x), P(x), P(x), 10, 20, ld, ctypes.byref(fp), P(obj), None,
                                                 P(ws), wsb, None)
    assert call(64, ws.numel()) == -2          # ld != ccvm_ld(20)
    assert call(128, 8) == -3                  # workspace too small
    fp.S = 0.0
    assert call(128, ws.numel()) == -1         # S must be positive
    fp.S = 1.0
    assert call(128, ws.numel()) == 0
    assert hip_lib.ccvm_objective_stats(None, 10, 1.0, None, None) == -1
    torch.cuda.synchronize()
