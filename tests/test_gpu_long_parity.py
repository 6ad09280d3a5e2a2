"""Long-trajectory parity at every BASELINE.json shape ("Next round" #2): the HIP path through
the PUBLIC solver API in replay mode (identical seeded noise: the normals come from torch's CPU stream in
the reference's order) against the oracle -- the reference's op sequence on the host -- at the stated
shapes and hundreds to thousands of steps, not a handful.

A run writes nothing by itself; with $CCVM_PARITY_RECORD naming a file each case appends its measured deviations to it
as one JSON line (tools/make_parity_md.py makes profiles/rNN_parity.md from such a file).  Gates: the tolerance
docs/parity.md states from these measurements, per shape -- no sqrt(N/20) extrapolation.
"""
import json
import os
import time

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

ADAM_A = dict(alpha=0.001, beta1=0.9, beta2=0.999, add_assign=False)  # SURVEY.md 8d config 3 "momentum state"

# (label, solver kind, N, batch, iterations, post-processor, Adam hyper-parameters, gate on |dx|, gate on |dobj| rel)
CASES = [
    ("config2_dl_n100_b1000", "dl", 100, 1000, 1500, None, None, 3e-4, 1e-5),
    ("config3_mf_n500_b1000", "mf", 500, 1000, 500, None, None, 3e-4, 1e-5),
    ("config3_langevin_n500_b1000", "langevin", 500, 1000, 500, None, None, 3e-4, 1e-5),
    ("config3_mf_n500_b1000_adam", "mf", 500, 1000, 300, None, ADAM_A, 3e-4, 1e-5),
    ("config3_langevin_n500_b1000_adam", "langevin", 500, 1000, 300, None, ADAM_A, 3e-4, 1e-5),
    ("config4_dl_n1000_b1000_headline", "dl", 1000, 1000, 1000, None, None, 3e-4, 1e-5),
    ("config5_pl_n2000_b512_adam_pp", "pl", 2000, 512, 200, "adam", None, 3e-4, 1e-5),
    # not a BASELINE configuration: the DL solver at config 3's size (the cluster kernel's two-plane mode)
    ("extra_dl_n500_b1000", "dl", 500, 1000, 500, None, None, 3e-4, 1e-5),
    # the cluster kernel's K = 640 (three row sets, Q's k >= 512 in registers) and K = 768 (spread over the XCDs) variants
    ("extra_mf_n640_b1000_adam", "mf", 640, 1000, 300, None, ADAM_A, 3e-4, 1e-5),
    ("extra_pl_n768_b1000", "pl", 768, 1000, 400, None, None, 3e-4, 1e-5),
    ("extra_dl_n700_b1000", "dl", 700, 1000, 400, None, None, 3e-4, 1e-5),
    # its half-chunk variant (round 5: K = 320 and 448; the DL case above is K = 704)
    ("extra_langevin_n300_b1000", "langevin", 300, 1000, 500, None, None, 3e-4, 1e-5),
    ("extra_mf_n448_b1000_adam", "mf", 448, 1000, 300, None, ADAM_A, 3e-4, 1e-5),
    # the persistent tile kernel's other solvers at the headline size (round 4): MF (mu, sigma in registers, the measured
    # amplitude handed over) and the Adam variants
    ("extra_mf_n1000_b1000", "mf", 1000, 1000, 400, None, None, 3e-4, 1e-5),
    ("extra_mf_n1000_b1000_adam", "mf", 1000, 1000, 200, None, ADAM_A, 3e-4, 1e-5),
    ("extra_langevin_n1000_b1000_adam", "langevin", 1000, 1000, 300, None, ADAM_A, 3e-4, 1e-5),
    # the per-GPU shapes of configs 4 / 5 under strong scaling on 4 GPUs (--global-batch 8000 / 4096): batches of two
    # rounds, run as two slices of the batch on the persistent tile kernel (replay blocks pitched by the whole batch)
    ("scaling_dl_n1000_b2000", "dl", 1000, 2000, 300, None, None, 3e-4, 1e-5),
    ("scaling_pl_n2000_b1024_adam_pp", "pl", 2000, 1024, 100, "adam", None, 3e-4, 1e-5),
]


def _record(entry):
    """The measured deviations as one JSON line per case -- only where $CCVM_PARITY_RECORD names a file
    (tools/profile_round.sh sets it to collect profiles/rNN_parity.md); a plain test run writes nothing."""
    path = os.environ.get("CCVM_PARITY_RECORD")
    if not path:
        return
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "a") as fh:
        fh.write(json.dumps(entry) + "\n")


@pytest.mark.parametrize("label,kind,n,b,t,post,adam,gate_x,gate_obj", CASES, ids=[c[0] for c in CASES])
def test_long_trajectory_matches_oracle_at_baseline_shape(label, kind, n, b, t, post, adam, gate_x, gate_obj):
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver
    from ccvm_amd.solvers.algorithms import AdamParameters
    from ccvm_amd.workloads import EXAMPLE_PARAMS, synthetic_instance
    from oracle import ccvm_oracle as oracle

    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cls = {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver, "pl": PumpedLangevinSolver}[kind]
    solver = cls(device="cpu", batch_size=b)
    solver.noise_mode = "replay"
    inst = synthetic_instance(n)
    inst.optimal_sol = 1.0  # synthetic instance: no known optimum
    p = dict(EXAMPLE_PARAMS[kind], iterations=t)
    solver.parameter_key = {n: p}
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    kwargs = {"algorithm_parameters": AdamParameters(**adam)} if adam else {}

    seed = 20260 + n
    torch.manual_seed(seed)
    t0 = time.time()
    sol = solver(instance=inst, post_processor=post, **kwargs)
    t_engine = time.time() - t0

    q, v, f = inst.q_matrix, inst.v_vector, float(inst.scaled_by)
    common = dict(scaled_by=f, optimal_value=1.0, post_processor=post)
    torch.manual_seed(seed)
    t0 = time.time()
    if kind == "dl":
        ref = oracle.solve_dl(q, v, b, t, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"], g=0.05, S=1,
                              **common)
        fields = ["problem_variables", "s"]
    elif kind == "mf":
        ref = oracle.solve_mf(q, v, b, t, p["pump"], p["dt"], p["j"], p["feedback_scale"], p["S"], g=0.01, adam=adam,
                              **common)
        fields = ["problem_variables", "mu", "sigma"]
    elif kind == "langevin":
        ref = oracle.solve_langevin(q, v, b, t, p["dt"], p["sigma"], p["feedback_scale"], p["S"], adam=adam, **common)
        fields = ["problem_variables"]
    else:
        ref = oracle.solve_pl(q, v, b, t, p["pump"], p["dt"], p["sigma"], p["feedback_scale"], p["S"], adam=adam,
                              **common)
        fields = ["problem_variables"]
    t_oracle = time.time() - t0

    import ctypes

    from ccvm_amd import _lib

    buf = ctypes.create_string_buffer(512)
    _lib.load().ccvm_describe_launch({"dl": 0, "mf": 1}.get(kind, 2), b, n, 1 if adam else 0, 0, buf, 512)
    entry = {"case": label, "solver": kind, "N": n, "batch": b, "iterations": t, "post_processor": post,
             "kernel": buf.value.decode().split(" grid")[0],
             "adam": bool(adam), "noise": "replay (torch CPU stream, reference order)", "fields": {},
             "oracle_s": round(t_oracle, 2), "engine_call_s": round(t_engine, 2)}
    worst_x = 0.0
    for name in fields:
        want, got = ref[name], sol.variables[name].cpu()
        assert bool(torch.isfinite(want).all()) and bool(torch.isfinite(got).all()), name
        err = float((got - want).abs().max())
        scale = max(1.0, float(want.abs().max()))
        entry["fields"][name] = {"max_abs_err": err, "max_abs_value": float(want.abs().max()),
                                 "rows_off_by_more_than_1e-4": int(((got - want).abs().amax(1) > 1e-4).sum())}
        worst_x = max(worst_x, err / scale)
    want, got = ref["objective_values"], sol.objective_values.cpu()
    obj_err = float((got - want).abs().max())
    obj_scale = float(want.abs().max())
    entry["objective_values"] = {"max_abs_err": obj_err, "max_abs_value": obj_scale, "rel": obj_err / obj_scale}
    entry["best_objective_value"] = {"engine": sol.best_objective_value, "oracle": ref["best_objective_value"]}
    entry["gates"] = {"x": gate_x, "obj_rel": gate_obj}
    _record(entry)

    assert worst_x <= gate_x, (label, entry)
    assert obj_err <= gate_obj * obj_scale, (label, entry)
    assert abs(sol.best_objective_value - ref["best_objective_value"]) <= gate_obj * abs(ref["best_objective_value"])
