"""GPU tests of the API surfaces around the loop: compatibility hooks with the reference's
known answers, evolution sampling, the shipped example."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT
from golden_util import golden

pytestmark = pytest.mark.gpu


def test_mf_hooks_known_answers():
    """The reference's own vectors (tests/golden/reference_unit_vectors.json = the inputs and expected
    values of ccvm_simulators/tests/unit/solvers/test_mf_solver.py:63-204, verbatim), through the HIP
    feedback / change-of-variables / clamp kernels: grads -20.0, drift (-20.0, 200.5) with Q = V = ones,
    mu~ = 0, S = 20, fs = 400, pump = 2.5, j = 399, g = 0.1; change_variables(4.0, S=2) = 1.5 and 1.1 with
    bounds (0.2, 0.8); fit_to_constraints with TENSOR bounds, called by keyword like the reference does."""
    from golden_util import reference_unit_vectors

    from ccvm_amd.solvers import MFSolver

    vec = reference_unit_vectors()["mf_solver"]
    p = vec["parameters"]
    solver = MFSolver(device="cpu", batch_size=1000, problem_category="boxqp")
    solver.q_matrix = torch.tensor(vec["q_matrix"])
    solver.v_vector = torch.tensor(vec["v_vector"])
    shape = (vec["batch_size"], vec["problem_size"])
    grads = solver._calculate_grads_boxqp(mu_tilde=torch.full(shape, vec["grads"]["mu_tilde_fill"]), S=p["S"],
                                          fs=p["feedback_scale"])
    assert grads.shape == torch.Size(shape)
    assert torch.equal(grads, torch.full(shape, vec["grads"]["expected_fill"]))
    d = vec["drift"]
    d_mu, d_sigma = solver._calculate_drift_boxqp(
        mu=torch.full(shape, d["mu_fill"]), mu_tilde=torch.full(shape, d["mu_tilde_fill"]),
        sigma=torch.full(shape, d["sigma_fill"]), pump=p["pump"], j=p["j"], g=d["g"], S=p["S"], fs=p["feedback_scale"])
    assert torch.equal(d_mu, torch.full(shape, d["expected_mu_fill"]))
    assert torch.equal(d_sigma, torch.full(shape, d["expected_sigma_fill"]))
    for case in vec["change_variables"]:
        y = solver._change_variables_boxqp(problem_variables=torch.tensor(case["problem_variables"]),
                                           lower_limit=case["lower_limit"], upper_limit=case["upper_limit"], S=case["S"])
        assert torch.equal(y, torch.tensor(case["expected"]))
    for case in vec["fit_to_constraints"]:
        mu_tilde = torch.tensor(case["mu_tilde"])
        z = solver._fit_to_constraints_boxqp(mu_tilde=mu_tilde, lower_clamp=torch.full_like(mu_tilde, case["lower_clamp"]),
                                             upper_clamp=torch.full_like(mu_tilde, case["upper_clamp"]))
        assert torch.equal(z, torch.tensor(case["expected"]))
    z = solver.fit_to_constraints(torch.tensor([[2.0, -3.0, 0.25]]), -1.0, 1.0)
    assert torch.equal(z, torch.tensor([[1.0, -1.0, 0.25]]))


def test_hooks_accept_a_per_variable_saturation():
    """calculate_grads / change_variables / fit_to_constraints with S as a 1-D tensor, against the
    oracle's formulas (which the reference goldens pin for tensor S)."""
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver
    from oracle import ccvm_oracle as oracle

    n, b = 12, 5
    gen = torch.Generator().manual_seed(1)
    q = torch.rand(n, n, generator=gen) - 0.5
    v = torch.rand(n, generator=gen) - 0.5
    x = torch.rand(b, n, generator=gen) - 0.5
    S = 0.5 + torch.rand(n, generator=gen)
    mf = MFSolver(device="cpu", batch_size=b)
    mf.q_matrix, mf.v_vector = q, v
    assert torch.allclose(mf.calculate_grads(x, S, 3.0, -0.5, 2.0), oracle.mf_grads(x, q, v, S, 3.0, -0.5, 2.0),
                          rtol=1e-5, atol=1e-5)
    lv = LangevinSolver(device="cpu", batch_size=b)
    lv.q_matrix, lv.v_vector = q, v
    want = -(torch.einsum("bi,ij->bj", x * 2.5 / (2 * S) + 0.75, q) + v) * 2.5 / (2 * S)   # langevin_solver.py:117-139
    assert torch.allclose(lv.calculate_grads(x, -0.5, 2.0, S), want, rtol=1e-5, atol=1e-5)
    dl = DLSolver(device="cpu", batch_size=b)
    dl.q_matrix, dl.v_vector = q, v
    gc, gs = dl.calculate_grads(x, -x, 0.0, 1.0, S)
    # the hook returns the gradient with its sign (dc = +fsd * grads + ...), the oracle's helper the
    # feedback G of dc = -fsd * G + ...
    assert torch.allclose(gc, -oracle.dl_feedback(x, q, v, 0.0, 1.0, S), rtol=1e-5, atol=1e-5)
    assert torch.allclose(gs, -oracle.dl_feedback(-x, q, v, 0.0, 1.0, S), rtol=1e-5, atol=1e-5)
    assert torch.allclose(mf.change_variables(x, -0.5, 2.0, S), 0.5 * x / S * 2.5 + 0.75, rtol=1e-6, atol=1e-6)
    assert torch.equal(mf.fit_to_constraints(3 * x, -S, S), torch.clamp(3 * x, -S, S))


@pytest.mark.parametrize("kind,step_size", [("dl", 7), ("mf", 10), ("pl", 4)])
def test_evolution_sampling_matches_oracle(tmp_path, kind, step_size):
    """evolution_step_size: samples after steps i % k == 0 and the last one, best row written
    as problem_size lines x num_samples values rounded to 4 d.p. (dl_solver.py:557-564, 252-281,
    961-974; mf_solver.py:285-300)."""
    from test_gpu_parity import _instance, _solver_for
    from oracle import ccvm_oracle as oracle

    g = golden("test020")
    meta = g.cases[f"{kind}_T100"]
    t = 30
    solver = _solver_for(kind, 16)
    inst = _instance(g)
    solver.parameter_key = {20: dict(meta["params"], iterations=t)}
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    path = str(tmp_path / "evo.txt")
    torch.manual_seed(3)
    sol = solver(instance=inst, evolution_step_size=step_size, evolution_file=path)
    assert sol.evolution_file == path

    pts = [i for i in range(t) if i % step_size == 0 or i + 1 >= t]
    depth = int(t / step_size) + 1 + (1 if t % step_size else 0)
    q, v, f = g.scaled(kind)
    p = meta["params"]
    samples = []
    grab = lambda i, *state: samples.append([s.clone() for s in state]) if i in pts else None
    torch.manual_seed(3)
    if kind == "dl":
        oracle.dl_loop(q, v, 16, t, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"], 0.05,
                       (0.0, 1.0), True, None, on_step=grab)
        names = ("c", "s")
    elif kind == "mf":
        oracle.mf_loop(q, v, 16, t, p["pump"], p["dt"], p["j"], p["feedback_scale"], p["S"], 0.01,
                       (0.0, 1.0), True, None, None, on_step=grab)
        names = ("mu", "sigma")
    else:
        oracle.pl_loop(q, v, 16, t, p["pump"], p["dt"], p["sigma"], p["feedback_scale"], p["S"],
                       (0.0, 1.0), True, None, None, on_step=grab)
        names = ("c",)
    assert len(samples) == len(pts)
    for k, name in enumerate(names):
        buf = getattr(solver, f"{name}_sample")
        assert tuple(buf.shape) == (16, 20, depth)
        for idx in range(len(pts)):
            assert float((buf[:, :, idx] - samples[idx][k]).abs().max()) <= 5e-4
        assert float(buf[:, :, len(pts):].abs().max()) == 0.0 if depth > len(pts) else True
    best = int(torch.argmax(-sol.objective_values))
    lines = open(path).read().split("\n")
    assert len(lines) == 20 * len(names) + 1 and lines[-1] == ""
    first = lines[0].rstrip("\t").split("\t")
    assert len(first) == depth
    want = [round(float(x), 4) for x in getattr(solver, f"{names[0]}_sample")[best, 0]]
    assert [float(x) for x in first] == want
    assert lines[0].endswith("\t") == (kind != "mf")


def test_invalid_evolution_step_size():
    from test_gpu_parity import _instance, _solver_for

    g = golden("test020")
    solver = _solver_for("pl", 4)
    solver.parameter_key = {20: dict(g.cases["pl_T1"]["params"])}
    with pytest.raises(ValueError, match="evolution step size"):
        solver(instance=_instance(g), evolution_step_size=-2)


def test_shipped_example_runs():
    """examples/boxqp_dl_demo.py (the reference example's flow on the shipped instance)."""
    out = subprocess.run([sys.executable, "boxqp_dl_demo.py"], cwd=os.path.join(ROOT, "examples"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Solution(problem_size=20, batch_size=1000" in out.stdout
    import re

    best = float(re.search(r"best_objective_value=([0-9.]+)", out.stdout).group(1))
    # known optimum of the shipped instance (tests/golden/make_example_instance.py chose a seed on which
    # the DL example parameters reach it for the whole batch)
    assert abs(best - 986.0) <= 986.0 * 1e-5
    frac = float(re.search(r"optimal fraction ([0-9.]+)", out.stdout).group(1))
    assert frac >= 0.9
    assert "TTS@99%" in out.stdout


def test_all_solvers_demo_runs():
    out = subprocess.run([sys.executable, "boxqp_all_solvers_demo.py", "--batch", "200", "--iterations", "400"],
                         cwd=os.path.join(ROOT, "examples"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if "TTS99" in ln]
    assert len(lines) == 6 and all("best" in ln for ln in lines)


def test_c_abi_from_a_plain_cpp_client(tmp_path):
    """The C ABI without Python or torch in the process (tests/abi_client.cpp): what a cgo / JNI /
    ctypes binding does.  Compiled here with hipcc against include/ccvm_hip.h and the in-tree library."""
    import shutil

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "abi_client"
    lib_dir = os.path.join(ROOT, "ccvm_amd")
    build = subprocess.run(
        [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
         os.path.join(ROOT, "tests", "abi_client.cpp"), "-L", lib_dir, "-lccvm_hip", f"-Wl,-rpath,{lib_dir}",
         "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "ABI_CLIENT_OK" in run.stdout, (run.stdout[-2000:], run.stderr[-2000:])


def test_integration_md_ctypes_stub_runs_and_matches_the_engine():
    """The ctypes stub INTEGRATION.md shows to a reference maintainer (the body that would replace
    DLSolver._solve) is executed as written -- only the library path is made absolute -- and must give
    the engine's own result for the same key."""
    import re
    import types

    from ccvm_amd import engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = next(b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "def _solve(self" in b)
    block = block.replace('ctypes.CDLL("libccvm_hip.so")', f'ctypes.CDLL("{os.path.join(ROOT, "ccvm_amd", "libccvm_hip.so")}")')
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)

    n, b, t = 20, 50, 30
    q, v, _ = scaled_qv(n, "dl")
    p = EXAMPLE_PARAMS["dl"]
    me = types.SimpleNamespace(q_matrix=q, v_vector=v, solution_bounds=(0.0, 1.0))
    torch.manual_seed(4242)
    c, s = ns["_solve"](me, n, b, "cuda", 1.0, p["pump"], p["dt"], t, p["noise_ratio"], p["feedback_scale"], True, 0.05,
                        None, 0)
    torch.cuda.synchronize()
    traj = engine.Trajectories(engine.DeviceProblem(q, v), b, "dl", t, dict(p, g=0.05), (0.0, 1.0),
                               engine.NoiseSpec(mode="philox", seed=4242))
    traj.advance(t)
    traj.clamp("c", -1.0, 1.0)
    assert c.shape == (b, n) and torch.equal(c.cpu(), traj.compact("c").cpu()) and torch.equal(s.cpu(), traj.compact("s").cpu())


def test_integration_md_finalize_stub_runs_and_matches_the_engine():
    """The second stub of INTEGRATION.md (the steps right after the loop through ccvm_finalize) executed as
    written, on the pitched arrays of an engine run: objective values bit-identical to Trajectories.score, best
    value and success fractions equal to Solution's."""
    import re

    from ccvm_amd import engine
    from ccvm_amd.solution import fractions_from_counts, success_fractions
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    first = next(b for b in blocks if "def _solve(self" in b)
    second = next(b for b in blocks if "def _score(" in b)
    lib_path = os.path.join(ROOT, "ccvm_amd", "libccvm_hip.so")
    ns = {}
    exec(compile(first.replace('ctypes.CDLL("libccvm_hip.so")', f'ctypes.CDLL("{lib_path}")'), "INTEGRATION.md", "exec"), ns)
    exec(compile(second, "INTEGRATION.md", "exec"), ns)

    n, b, t, S = 20, 50, 40, 0.8
    q, v, f = scaled_qv(n, "dl")
    p = EXAMPLE_PARAMS["dl"]
    run = lambda: engine.Trajectories(engine.DeviceProblem(q, v), b, "dl", t, dict(p, g=0.05), (0.0, 1.0),
                                      engine.NoiseSpec(mode="fused", seed=77))
    a, ref = run(), run()
    a.advance(t)
    ref.advance(t)
    optimal = 100.0
    obj, best, perf = ns["_score"](a.p.q, a.p.v, a.state["c"], b, n, S, (0.0, 1.0), float(f), optimal)
    torch.cuda.synchronize()
    want = ref.score("c", S, float(f), 0.0, 1.0, optimal_value=optimal, clamp=(-S, S))
    assert torch.equal(obj.cpu(), want.objective_values.cpu())
    assert torch.equal(a.state["c"], ref.state["c"])            # both clamped in place
    wbest, within, rows, _ = engine.read_stats(want.stats)
    assert best == wbest == float(torch.max(-obj).item())
    assert perf == fractions_from_counts(within, rows) == success_fractions(obj.cpu(), optimal)


def test_post_processors_reject_mismatched_shapes():
    """Reference unit tests test_postprocess_error_for_invalid_c_dimension / _invalid_v_vector_shape
    (tests/unit/postprocessor/test_adam.py, test_grad_descent.py): any exception; here a ValueError
    before anything reaches the GPU."""
    from ccvm_amd.post_processor.factory import PostProcessorFactory

    n, m = 12, 5
    q, v, c = torch.rand(n, n), torch.rand(n), torch.rand(m, n)
    for method in ("adam", "grad-descent"):
        pp = PostProcessorFactory.create_postprocessor(method)
        assert tuple(pp.postprocess(c, q, v).shape) == (m, n)
        with pytest.raises(ValueError):
            pp.postprocess(torch.rand(m, 9), q, v)
        with pytest.raises(ValueError):
            pp.postprocess(c, q, torch.rand(n, n))
        with pytest.raises(TypeError, match="parameter c must be a tensor"):
            pp.postprocess("dummy-c", q, v)


def test_diverged_runs_report_nan_like_the_reference():
    """The README snippet's parameters (pump 2.0, dt 0.005) with feedback_scale 100 diverge (SURVEY.md 8d,
    config 1 note): the reference's amplitudes become NaN and torch.clamp keeps them NaN, so the
    objective values are NaN.  The engine must not turn them into bounds and report finite garbage."""
    from ccvm_amd import engine
    from ccvm_amd.workloads import scaled_qv
    from oracle import ccvm_oracle as oracle
    from oracle.noise_ref import FusedNoise

    n, b, t, seed = 20, 64, 400, 99
    q, v, f = scaled_qv(n, "dl")
    p = {"pump": 2.0, "dt": 0.005, "noise_ratio": 10, "feedback_scale": 100}
    c, _ = oracle.dl_loop(q, v, b, t, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"], 0.05, (0.0, 1.0),
                          True, FusedNoise(seed, 0, single=False))
    want = int(torch.isnan(torch.clamp(c, -1, 1)).any(1).sum())
    assert want > 10  # the configuration does diverge
    for path in ("persist", "tile"):
        os.environ["CCVM_AMD_KERNEL"] = path
        try:
            traj = engine.Trajectories(engine.DeviceProblem(q, v), b, "dl", t, dict(p, g=0.05), (0.0, 1.0),
                                       engine.NoiseSpec(mode="philox", seed=seed))
            traj.advance(t)
            traj.clamp("c", -1.0, 1.0)
            x = traj.compact("c")
            got = int(torch.isnan(x).any(1).sum())
            assert abs(got - want) <= 6, (path, got, want)  # which rows blow up is chaotic at the margin
            obj = engine.energy(engine.change_variables(x, 1.0, 0.0, 1.0), q, v, float(f))
            assert int(torch.isnan(obj).sum()) == got
        finally:
            os.environ.pop("CCVM_AMD_KERNEL", None)


def test_concurrent_host_threads_on_their_own_streams():
    """No global mutable state in the library (include/ccvm_hip.h): four host threads, each on its own
    HIP stream, run different solves at the same time; every result equals the same solve run alone."""
    import threading

    from ccvm_amd import engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv

    jobs = [("dl", 40, 96, 60), ("mf", 130, 64, 40), ("langevin", 300, 128, 25), ("dl", 20, 200, 80)]

    def solve(kind, n, b, t, stream=None):
        q, v, _ = scaled_qv(n, "pl" if kind == "langevin" else kind)
        p = dict(EXAMPLE_PARAMS["pl" if kind == "langevin" else kind], g=0.03, use_pump=True)
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            traj = engine.Trajectories(engine.DeviceProblem(q, v), b, kind, t, p, (0.0, 1.0),
                                       engine.NoiseSpec(mode="philox", seed=1000 + n))
            for _ in range(t // 5):
                traj.advance(5)
            out = {k: traj.compact(k) for k in traj.state}
        if stream is not None:
            stream.synchronize()
        else:
            torch.cuda.synchronize()
        return {k: x.cpu() for k, x in out.items()}

    alone = [solve(*job) for job in jobs]
    results, errors = [None] * len(jobs), []

    def worker(i):
        try:
            results[i] = solve(*jobs[i], stream=torch.cuda.Stream())
        except Exception as exc:  # surfaced below
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for want, got in zip(alone, results):
        for key in want:
            assert torch.equal(want[key], got[key]), key


def test_staged_problem_cache_never_serves_stale_data():
    from ccvm_amd import engine

    q, v, x = torch.rand(12, 12), torch.rand(12), torch.rand(7, 12)
    first = engine.device_problem(q, v)
    assert engine.device_problem(q, v) is first                     # same objects, unmodified: reused
    e1 = engine.energy(x, q, v)
    q.mul_(2.0)                                                     # in-place edit: version counter moves
    assert engine.device_problem(q, v) is not first
    e2 = engine.energy(x, q, v)
    want = (0.5 * torch.einsum("bi,ij,bj->b", x, q, x) + x @ v)
    assert torch.allclose(e2, want, rtol=1e-5, atol=1e-5) and not torch.allclose(e1, e2)
    q2 = q.clone()                                                  # equal values, different object
    assert engine.device_problem(q2, v) is not engine.device_problem(q, v)


def test_workspace_padded_flag_skips_the_memset_without_changing_results():
    """CCVM_RUN_WS_PADDED (ccvm_noise.flags): chunked calls that skip re-zeroing the scratch arrays give the same
    trajectories bit for bit as one call, on every kernel path, and the engine sets the flag from its second call on."""
    from ccvm_amd import _lib, engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv

    for kind, n, b in (("dl", 300, 40), ("dl", 1000, 64), ("mf", 600, 1300), ("langevin", 900, 50), ("dl", 100, 30)):
        q, v, _ = scaled_qv(n, kind if kind != "langevin" else "langevin")
        p = dict(EXAMPLE_PARAMS[kind], g=0.05 if kind == "dl" else 0.01)
        out = []
        for chunks in ([12], [1, 5, 6]):
            prob = engine.DeviceProblem(q, v)
            traj = engine.Trajectories(prob, b, kind, 12, p, (0.0, 1.0), engine.NoiseSpec(mode="fused", seed=9))
            assert traj._ws_padded is False
            for k in chunks:
                traj.advance(k)
            assert traj._ws_padded is True
            out.append({name: traj.compact(name).cpu() for name in traj.state})
        for name in out[0]:
            assert torch.equal(out[0][name], out[1][name]), (kind, n, name)
    assert _lib.RUN_WS_PADDED == 1


def test_schedule_tables_are_shared_between_runs_of_the_same_parameters_only():
    """Round 6: a run's schedule table is a function of the solver's scalars and T alone, so runs with the same
    parameters on the same stream share one (engine._schedule_cache: repeated solves of an instance skip the
    allocation and the schedule kernel).  Same parameters -> the same table object and bit-identical trajectories;
    another dt, another T, another Adam setting -> a table of their own; the status word is read once per verified
    state, not once per check."""
    from ccvm_amd import engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv

    q, v, _ = scaled_qv(20, "dl")
    prob = engine.DeviceProblem(q, v)

    def run(params, t, kind="dl", adam=None):
        traj = engine.Trajectories(prob, 64, kind, t, params, (0.0, 1.0), engine.NoiseSpec(mode="fused", seed=5), adam=adam)
        traj.advance(t)
        return traj

    base = dict(EXAMPLE_PARAMS["dl"], g=0.05)
    a1, a2 = run(base, 40), run(base, 40)
    assert a1._schedule.data_ptr() == a2._schedule.data_ptr()
    assert torch.equal(a1.compact("c"), a2.compact("c")) and torch.equal(a1.compact("s"), a2.compact("s"))
    b = run(dict(base, dt=0.002), 40)
    c = run(base, 41)
    assert len({a1._schedule.data_ptr(), b._schedule.data_ptr(), c._schedule.data_ptr()}) == 3
    assert not torch.equal(a1.compact("c"), b.compact("c"))
    lv = dict(EXAMPLE_PARAMS["langevin"], use_pump=False)
    ad = {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False}
    l1, l2 = run(lv, 30, "langevin"), run(lv, 30, "langevin", adam=ad)
    assert l1._schedule.data_ptr() != l2._schedule.data_ptr()  # (the Adam bias corrections are words of the table)
    l3 = run(lv, 30, "langevin", adam=dict(ad))
    assert l2._schedule.data_ptr() == l3._schedule.data_ptr() and torch.equal(l2.compact("c"), l3.compact("c"))
    # one device-to-host read per verified state
    reads = []
    real = torch.Tensor.cpu
    t = run(base, 10)
    try:
        torch.Tensor.cpu = lambda self, *a, **k: (reads.append(1), real(self, *a, **k))[1]
        assert t.check() is False and t.check() is False and t.check(hold=True) is False
        assert len(reads) == 1
        t.advance(0)          # (no run call: nothing to verify)
        t.check()
        assert len(reads) == 1
    finally:
        torch.Tensor.cpu = real
