"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (the reference tree never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

The reference is imported from its read-only tree (/root/reference); nothing of it is
copied -- only numeric inputs (parsed Q, V, header fields, parameters, seeds) and outputs
(final variables, objective values, success fractions) are written, as data.

Recipe: load instance -> scale_coefs(get_scaling_factor(Q)) ->
torch.manual_seed(seed) immediately before solver(instance=...).
"""
import json
import os
import sys

REFERENCE = os.environ.get("CCVM_REFERENCE", "/root/reference")
sys.path.insert(0, REFERENCE)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ccvm_simulators  # noqa: E402

assert os.path.realpath(ccvm_simulators.__file__).startswith(os.path.realpath(REFERENCE)), (
    "make_golden.py must import the reference package, got " + ccvm_simulators.__file__
)
from ccvm_simulators.problem_classes.boxqp import ProblemInstance  # noqa: E402
from ccvm_simulators.solvers import (  # noqa: E402
    DLSolver,
    LangevinSolver,
    MFSolver,
    PumpedLangevinSolver,
)
from ccvm_simulators.solvers.algorithms import AdamParameters  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
INSTANCES = {
    "test020": "ccvm_simulators/tests/data/test_instances/test020-100-10.in",
    "tuningH020": "examples/benchmarking_instances/single_test_instance/tuningH020-100-0.in",
}
B, SEED = 100, 7

# parameter_key values of the reference's example scripts (examples/ccvm_boxqp_dl.py:16-24,
# ccvm_boxqp_mf.py:16-25, langevin_boxqp.py:16-24, pumped_langevin_boxqp.py:16-25)
PARAMS = {
    "dl": {"pump": 8.0, "feedback_scale": 100, "dt": 0.001, "noise_ratio": 10},
    "mf": {"pump": 0.0, "feedback_scale": 4000, "j": 5.0, "S": 20.0, "dt": 0.0025},
    "langevin": {"dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0},
    "pl": {"pump": 2.0, "dt": 0.002, "S": 0.5, "sigma": 0.5, "feedback_scale": 1.0},
}
SOLVERS = {"dl": DLSolver, "mf": MFSolver, "langevin": LangevinSolver, "pl": PumpedLangevinSolver}
ADAMS = {
    "adamA": dict(alpha=0.001, beta1=0.9, beta2=0.999, add_assign=False),
    "adamB": dict(alpha=0.01, beta1=0.8, beta2=1.0, add_assign=True),
    "adamC": dict(alpha=0.05, beta1=0.9, beta2=0.99, add_assign=True),
}


def made_with():
    """Recorded in every manifest: the fixtures are bit-exact only on the configuration that made them (the einsum's
    blocking depends on the thread count: 3e-7 relative at N = 300 between 1 and 8 threads)."""
    return {"torch": torch.__version__, "torch_num_threads": torch.get_num_threads()}


def run_case(kind, path, iterations, adam=None, post=None, flag=True, batch=B, seed=SEED, bounds=(0.0, 1.0),
             dl_S=None, s_vector=None, g=None):
    """``s_vector``: per-variable saturation (1-D tensor of length N) -- DL takes it in the constructor,
    the other solvers in the parameter key, scaled to the magnitude of their scalar default."""
    if kind == "dl" and s_vector is not None:
        dl_S = s_vector  # (also a 2-D tensor: full_s_cases)
    solver = SOLVERS[kind](device="cpu", batch_size=batch, **({"S": dl_S} if dl_S is not None else {}))
    inst = ProblemInstance(instance_type="test", file_path=os.path.join(REFERENCE, path), device="cpu",
                           solution_bounds=bounds)
    key = dict(PARAMS[kind], iterations=iterations)
    if kind != "dl" and s_vector is not None:
        key["S"] = s_vector * PARAMS[kind]["S"]
    solver.parameter_key = {inst.problem_size: key}
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    kwargs = {}
    if kind in ("dl", "mf", "pl"):
        kwargs["pump_rate_flag"] = flag
    if adam:
        kwargs["algorithm_parameters"] = AdamParameters(**ADAMS[adam])
    if g is not None:
        kwargs["g"] = g  # __call__(g=...) of DL (default 0.05) and MF (default 0.01)
    torch.manual_seed(seed)
    first_draw = torch.randn(inst.problem_size, batch)  # checksum of the stream's first block
    torch.manual_seed(seed)
    sol = solver(instance=inst, post_processor=post, **kwargs)
    arrays = {k: v.detach().numpy().copy() for k, v in sol.variables.items()}
    arrays["objective_values"] = sol.objective_values.detach().numpy().copy()
    params_out = {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in key.items()}
    meta = {
        "kind": kind, "iterations": iterations, "adam": ADAMS[adam] if adam else None, "post": post,
        "pump_rate_flag": flag, "batch": batch, "seed": seed, "params": params_out, "bounds": list(bounds),
        "dl_S": dl_S.tolist() if torch.is_tensor(dl_S) else dl_S, "g": g,
        "best_objective_value": sol.best_objective_value,
        "solution_performance": sol.solution_performance,
        "scaled_by": float(inst.scaled_by),
        "noise_checksum": [float(first_draw.double().sum()), float(first_draw.double().abs().sum()),
                           float(first_draw[0, 0]), float(first_draw[-1, -1])],
    }
    return arrays, meta


def anchors():
    # the DL example exactly as shipped: B=1000, T=1500, seed 1234 (SURVEY.md 8c anchor)
    arrays, meta = run_case("dl", INSTANCES["tuningH020"], 1500, batch=1000, seed=1234)
    with open(os.path.join(OUT, "dl_example_anchor.json"), "w") as fh:
        json.dump(dict(meta, made_with=made_with()), fh, indent=1, sort_keys=True)
    print("anchor", meta["best_objective_value"], meta["solution_performance"])
    # BASELINE.json configs[0]: DLSolver on test020-100-10.in, batch 100, 15000 iterations
    # (example parameters, g = 0.05, seed 1234)
    arrays, meta = run_case("dl", INSTANCES["test020"], 15000, batch=100, seed=1234)
    with open(os.path.join(OUT, "baseline_config1_anchor.json"), "w") as fh:
        json.dump(dict(meta, made_with=made_with()), fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "baseline_config1_anchor.npz"), **arrays)
    print("config 1", meta["best_objective_value"], meta["solution_performance"])


def distribution_cases(seeds=(1234, 2345, 3456), batch=1000, iterations=1500):
    """Distributional anchors for the FUSED noise mode ("success fractions / best objective
    within sampling error"): the reference draws Normal.sample per step (dl_solver.py:538-547), the engine's default
    generates its normals in the kernel, so the two can only agree in distribution.  For every solver the shipped
    example configuration on tuningH020-100-0 (the example scripts' instance; seed 1234 of DL is the SURVEY 8c
    anchor: 130.7142 @ 0.987): the objective value of every one of 3 x 1000 trajectories, nothing else."""
    store, manifest = {}, {"cases": {}, "made_with": made_with(), "instance": INSTANCES["tuningH020"]}
    for kind in SOLVERS:
        for seed in seeds:
            arrays, meta = run_case(kind, INSTANCES["tuningH020"], iterations, batch=batch, seed=seed)
            name = f"{kind}_seed{seed}"
            store[name] = arrays["objective_values"]
            manifest["cases"][name] = {k: meta[k] for k in ("kind", "iterations", "batch", "seed", "params",
                                                            "best_objective_value", "solution_performance",
                                                            "scaled_by")}
            print("distribution", name, meta["best_objective_value"], meta["solution_performance"]["optimal"])
    np.savez_compressed(os.path.join(OUT, "distribution_anchors.npz"), **store)
    with open(os.path.join(OUT, "distribution_anchors.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)


def bounds_cases():
    """Non-default solution_bounds (the affine maps (u - l), (u + l) of every drift / grads function)."""
    store, manifest = {}, {"cases": {}}
    for bounds in ((-0.5, 2.0), (1.0, 3.0)):
        for kind in SOLVERS:
            for adam in (None, "adamA") if kind != "dl" else (None,):
                name = f"{kind}_T20_b{bounds[0]}_{bounds[1]}" + (f"_{adam}" if adam else "")
                arrays, meta = run_case(kind, INSTANCES["test020"], 20, adam=adam, batch=40, bounds=bounds)
                for k, v in arrays.items():
                    store[f"{name}/{k}"] = v
                manifest["cases"][name] = meta
                print("bounds", name, meta["best_objective_value"])
    # DLSolver(S=...): the saturation of the constructor only enters the final clamp and the change of
    # variables (dl_solver.py:567, 219-235) -- the drift ignores it
    for dl_S in (2.0, 0.5):
        for post in (None, "grad-descent"):
            name = f"dl_T50_S{dl_S}" + (f"_{post}" if post else "")
            arrays, meta = run_case("dl", INSTANCES["test020"], 50, post=post, batch=40, dl_S=dl_S)
            for k, v in arrays.items():
                store[f"{name}/{k}"] = v
            manifest["cases"][name] = meta
            print("S", name, meta["best_objective_value"])
    # the nonlinearity / measurement strength g of __call__ (dl_solver.py:549, mf_solver.py:141-198)
    for kind, g in (("dl", 0.2), ("dl", 0.005), ("mf", 0.05), ("mf", 0.3)):
        for adam in (None, "adamC") if kind == "mf" else (None,):
            name = f"{kind}_T40_g{g}" + (f"_{adam}" if adam else "")
            arrays, meta = run_case(kind, INSTANCES["test020"], 40, adam=adam, batch=40, g=g)
            for k, v in arrays.items():
                store[f"{name}/{k}"] = v
            manifest["cases"][name] = meta
            print("g", name, meta["best_objective_value"])
    np.savez_compressed(os.path.join(OUT, "test020_bounds.npz"), **store)
    with open(os.path.join(OUT, "test020_bounds.json"), "w") as fh:
        json.dump(dict(manifest, made_with=manifest.get("made_with") or made_with()), fh, indent=1, sort_keys=True)


def vector_s_cases():
    """Per-variable saturation S (a 1-D tensor of length N; dl_solver.py:843-848, mf_solver.py:834-839,
    langevin_solver.py:630-635, pumped_langevin_solver.py:519-524)."""
    store, manifest = {}, {"cases": {}}
    g = torch.Generator().manual_seed(5)
    s_vector = 0.5 + 1.5 * torch.rand(20, generator=g)  # x the solver's scalar default
    for kind in SOLVERS:
        variants = [(None, None), (None, "grad-descent")] + ([("adamA", None), ("adamC", None)] if kind != "dl" else [])
        for adam, post in variants:
            name = f"{kind}_T30_vecS" + (f"_{adam}" if adam else "") + (f"_{post}" if post else "")
            arrays, meta = run_case(kind, INSTANCES["test020"], 30, adam=adam, post=post, batch=40, s_vector=s_vector)
            for k, v in arrays.items():
                store[f"{name}/{k}"] = v
            manifest["cases"][name] = meta
            print("vecS", name, meta["best_objective_value"])
    np.savez_compressed(os.path.join(OUT, "test020_vecS.npz"), **store)
    with open(os.path.join(OUT, "test020_vecS.json"), "w") as fh:
        json.dump(dict(manifest, made_with=manifest.get("made_with") or made_with()), fh, indent=1, sort_keys=True)


def full_s_cases():
    """DLSolver(S=<2-D tensor>): one saturation per trajectory AND variable.  The reference passes any
    non-1-D tensor S straight through (dl_solver.py:843-848) to the final clamp (:567) and the change of
    variables (:956-959); shapes (B, N), (B, 1) and (1, N) all broadcast there."""
    store, manifest = {}, {"cases": {}}
    g = torch.Generator().manual_seed(11)
    batch = 40
    shapes = {"BN": (batch, 20), "B1": (batch, 1), "1N": (1, 20)}
    for label, shape in shapes.items():
        s_full = 0.4 + 1.6 * torch.rand(shape, generator=g)
        for post in (None, "grad-descent", "adam"):
            name = f"dl_T40_fullS_{label}" + (f"_{post}" if post else "")
            arrays, meta = run_case("dl", INSTANCES["test020"], 40, post=post, batch=batch, dl_S=s_full)
            for k, v in arrays.items():
                store[f"{name}/{k}"] = v
            manifest["cases"][name] = meta
            print("fullS", name, meta["best_objective_value"])
    # MF / Langevin / pumped Langevin: the 2-D S of the parameter key acts INSIDE the loop (clamp bound and the
    # 1 / S of the feedback term per element): mf_solver.py:834-839, langevin_solver.py:630-635,
    # pumped_langevin_solver.py:519-524 pass it through
    for kind in ("mf", "langevin", "pl"):
        for label in ("BN", "B1"):
            s_full = 0.5 + 1.5 * torch.rand(shapes[label], generator=g)  # x the solver's scalar default
            for adam, post in ((None, None), ("adamA", None), ("adamC", "grad-descent")):
                name = f"{kind}_T30_fullS_{label}" + (f"_{adam}" if adam else "") + (f"_{post}" if post else "")
                arrays, meta = run_case(kind, INSTANCES["test020"], 30, adam=adam, post=post, batch=batch,
                                        s_vector=s_full)
                for k, v in arrays.items():
                    store[f"{name}/{k}"] = v
                manifest["cases"][name] = meta
                print("fullS", name, meta["best_objective_value"])
    np.savez_compressed(os.path.join(OUT, "test020_fullS.npz"), **store)
    with open(os.path.join(OUT, "test020_fullS.json"), "w") as fh:
        json.dump(dict(manifest, made_with=manifest.get("made_with") or made_with()), fh, indent=1, sort_keys=True)


def asgd_cases():
    """post_processor="asgd" through every solver, and the post-processors called directly with
    num_iter = 1 and 3 (only the first optimizer step of adam / asgd ever takes effect)."""
    from ccvm_simulators.post_processor.adam import PostProcessorAdam
    from ccvm_simulators.post_processor.asgd import PostProcessorASGD
    from ccvm_simulators.post_processor.grad_descent import PostProcessorGradDescent
    from ccvm_simulators.post_processor.lbfgs import PostProcessorLBFGS

    store, manifest = {}, {"cases": {}}
    for kind in SOLVERS:
        for post in ("asgd", "lbfgs"):
            name = f"{kind}_T50_{post}"
            arrays, meta = run_case(kind, INSTANCES["test020"], 50, post=post, batch=40)
            for k, v in arrays.items():
                store[f"{name}/{k}"] = v
            manifest["cases"][name] = meta
            print(post, name, meta["best_objective_value"])
    g = torch.Generator().manual_seed(17)
    n, b = 13, 9
    q, v, c = torch.rand(n, n, generator=g) - 0.5, torch.rand(n, generator=g) - 0.5, torch.rand(b, n, generator=g) * 1.2 - 0.1
    store["direct/q"], store["direct/v"], store["direct/c"] = q.numpy(), v.numpy(), c.numpy()
    for label, cls in (("adam", PostProcessorAdam), ("asgd", PostProcessorASGD), ("lbfgs", PostProcessorLBFGS)):
        for it in (1, 3):
            store[f"direct/{label}_iter{it}"] = cls().postprocess(c.clone(), q, v, num_iter=it).numpy()
        store[f"direct/{label}_bounds"] = cls().postprocess(c.clone(), q, v, lower_clamp=0.2, upper_clamp=0.7).numpy()
    store["direct/lbfgs_steep"] = PostProcessorLBFGS().postprocess(c.clone(), q * 100, v * 100, num_iter=2).numpy()
    store["direct/grad-descent"] = PostProcessorGradDescent().postprocess(c.clone(), q, v).numpy()
    store["direct/grad-descent_custom"] = PostProcessorGradDescent().postprocess(
        c.clone(), q, v, lower_clamp=0.1, upper_clamp=0.9, num_iter_pp=4, step_size=0.05).numpy()
    np.savez_compressed(os.path.join(OUT, "test020_asgd.npz"), **store)
    with open(os.path.join(OUT, "test020_asgd.json"), "w") as fh:
        json.dump(dict(manifest, made_with=manifest.get("made_with") or made_with()), fh, indent=1, sort_keys=True)


def larger_n_cases(n=96, batch=24, iterations=25):
    """The reference on a problem larger than its shipped files (N = 96 / 300 / 600, dense synthetic, written to a
    temporary .in file the reference parses itself): pins the oracle beyond N = 20 -- other einsum
    blocking, several K chunks / column groups of the engine's small-N kernel (96), the column-cluster
    kernel (300: K = 384; 600: K = 640, three row sets)."""
    import tempfile

    g = torch.Generator().manual_seed(n)
    a = torch.randn(n, n, generator=g) * 5
    q_file = ((a + a.T) / 2 ** 0.5).double().numpy()   # maximisation form, as stored in .in files
    v_file = (torch.randn(n, generator=g) * 17).double().numpy()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, f"synthetic{n:03d}-100-{n}.in")
        with open(path, "w") as fh:
            fh.write("\t".join([str(n), "1.0", "1.0", "True", "0.0", "0.0", str(n), "0"]) + "\n")
            fh.write("\t".join(repr(float(x)) for x in v_file) + "\n")
            for row in q_file:
                fh.write("\t".join(repr(float(x)) for x in row) + "\n")
        inst = ProblemInstance(instance_type="test", file_path=path, device="cpu")
        manifest = {"cases": {}, "instance": {"problem_size": n, "optimal_sol": inst.optimal_sol,
                                              "best_sol": inst.best_sol, "name": inst.name}}
        if n <= 96:
            store = {"q_matrix": inst.q_matrix.numpy().copy(), "v_vector": inst.v_vector.numpy().copy()}
        else:  # 1.4 MB at N = 600: not stored, regenerated from the seed by tests/golden_util.py and checked
            store = {}
            qd, vd = inst.q_matrix.double(), inst.v_vector.double()
            assert torch.equal(inst.q_matrix, torch.from_numpy(-q_file).float())
            assert torch.equal(inst.v_vector, torch.from_numpy(-v_file).float())
            manifest["instance"]["generated"] = {
                "seed": n,
                "recipe": "g = torch.Generator().manual_seed(seed); a = torch.randn(n, n, generator=g) * 5; "
                          "q_file = ((a + a.T) / 2 ** 0.5).double(); v_file = (torch.randn(n, generator=g) * 17).double(); "
                          "q_matrix = (-q_file).float(); v_vector = (-v_file).float()   (what the reference's parser "
                          "returns for the .in file make_golden.py wrote; the arrays themselves are not stored: 1.4 MB "
                          "at N = 600)",
                "q_checksum": [float(qd.sum()), float(qd.abs().sum()), float(qd[0, 1]), float(qd[-1, -2])],
                "v_checksum": [float(vd.sum()), float(vd.abs().sum())]}
        rel = os.path.relpath(path, REFERENCE)
        for kind in SOLVERS:
            for adam in (None, "adamA") if kind != "dl" else (None,):
                name = f"{kind}_T{iterations}" + (f"_{adam}" if adam else "")
                arrays, meta = run_case(kind, rel, iterations, adam=adam, batch=batch)
                for k, v in arrays.items():
                    store[f"{name}/{k}"] = v
                manifest["cases"][name] = meta
                print(f"N={n}", name, meta["best_objective_value"])
    np.savez_compressed(os.path.join(OUT, f"synthetic{n:03d}.npz"), **store)
    with open(os.path.join(OUT, f"synthetic{n:03d}.json"), "w") as fh:
        json.dump(dict(manifest, made_with=manifest.get("made_with") or made_with()), fh, indent=1, sort_keys=True)


def thick_cases(n, batch, iterations, kinds=None, seed_offset=7000):
    """The reference at the sizes and batches where the engine's kernels change shape: N = 300 / 500 /
    600 / 768 at batch 100 (column-cluster kernel: three or more clusters, a ragged last one; T = 100), N = 1000 at
    batch 64 (tile kernel's headline shape, and the slab kernel's 8 rows x 8 clusters; T = 50).  To keep the fixtures
    small, the per-row objective values are stored for EVERY row (each is a function of all of that row's variables)
    and the variable arrays for a sample of rows that straddles every cluster / tile boundary.  Instance regenerated
    from the seed, as for synthetic300 / 600."""
    import tempfile

    seed = seed_offset + n
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(n, n, generator=g) * 5
    q_file = ((a + a.T) / 2 ** 0.5).double().numpy()
    v_file = (torch.randn(n, generator=g) * 17).double().numpy()
    keep = sorted({r for r in (0, 1, 2, 3, 4, 15, 16, 31, 32, 33, 47, 48, 49, 63, 64, 95, 96, 97, batch - 2, batch - 1)
                   if 0 <= r < batch})
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, f"synthetic{n:04d}-100-{seed}.in")
        with open(path, "w") as fh:
            fh.write("\t".join([str(n), "1.0", "1.0", "True", "0.0", "0.0", str(n), "0"]) + "\n")
            fh.write("\t".join(repr(float(x)) for x in v_file) + "\n")
            for row in q_file:
                fh.write("\t".join(repr(float(x)) for x in row) + "\n")
        inst = ProblemInstance(instance_type="test", file_path=path, device="cpu")
        qd, vd = inst.q_matrix.double(), inst.v_vector.double()
        manifest = {
            "cases": {},
            "made_with": {"torch": torch.__version__, "torch_num_threads": torch.get_num_threads(),
                          "note": "the einsum's blocking depends on the thread count: another thread count moves the "
                                  "last bits (3e-7 relative observed at N = 300), inside the 1e-5 gate"},
            "rows_kept": keep,
            "instance": {"problem_size": n, "optimal_sol": inst.optimal_sol, "best_sol": inst.best_sol,
                         "name": inst.name,
                         "generated": {"seed": seed,
                                       "recipe": "as synthetic300 / 600 (tests/golden_util.py: Golden._generated)",
                                       "q_checksum": [float(qd.sum()), float(qd.abs().sum()), float(qd[0, 1]),
                                                      float(qd[-1, -2])],
                                       "v_checksum": [float(vd.sum()), float(vd.abs().sum())]}}}
        store = {}
        rel = os.path.relpath(path, REFERENCE)
        for kind in (kinds or SOLVERS):
            for adam in (None, "adamA") if kind in ("mf", "langevin") and kinds is None else (None,):
                name = f"{kind}_T{iterations}" + (f"_{adam}" if adam else "")
                arrays, meta = run_case(kind, rel, iterations, adam=adam, batch=batch)
                for k, v in arrays.items():
                    store[f"{name}/{k}"] = v if k == "objective_values" else v[keep]
                manifest["cases"][name] = meta
                print(f"N={n} B={batch}", name, meta["best_objective_value"])
    np.savez_compressed(os.path.join(OUT, f"thick{n:04d}.npz"), **store)
    with open(os.path.join(OUT, f"thick{n:04d}.json"), "w") as fh:
        json.dump(dict(manifest, made_with=manifest.get("made_with") or made_with()), fh, indent=1, sort_keys=True)


def main():
    torch.set_num_threads(1)  # fixtures independent of intra-op partitioning
    if "--only-thick" in sys.argv:  # where the engine's kernels change shape, at real batch sizes
        for n in (300, 500, 600, 768):
            thick_cases(n, batch=100, iterations=100)
        thick_cases(1000, batch=64, iterations=50, kinds=("dl", "pl", "mf"))
        return
    if "--only-larger-n" in sys.argv:
        larger_n_cases()
        return
    if "--only-cluster-n" in sys.argv:  # the sizes the column-cluster kernel serves
        larger_n_cases(300, batch=12, iterations=20)
        larger_n_cases(600, batch=12, iterations=16)
        return
    if "--only-asgd" in sys.argv:
        asgd_cases()
        return
    if "--only-vector-s" in sys.argv:
        vector_s_cases()
        return
    if "--only-full-s" in sys.argv:
        full_s_cases()
        return
    if "--only-anchors" in sys.argv:
        anchors()
        return
    if "--only-distribution" in sys.argv:
        distribution_cases()
        return
    if "--only-bounds" in sys.argv:
        bounds_cases()
        return
    for tag, path in INSTANCES.items():
        inst = ProblemInstance(instance_type="test", file_path=os.path.join(REFERENCE, path), device="cpu")
        store = {
            "q_matrix": inst.q_matrix.numpy().copy(),  # parsed + negated, unscaled
            "v_vector": inst.v_vector.numpy().copy(),
        }
        manifest = {
            "instance": {
                "source": path, "problem_size": inst.problem_size, "optimal_sol": inst.optimal_sol,
                "best_sol": inst.best_sol, "optimality": inst.optimality, "sol_time_gb": inst.sol_time_gb,
                "sol_time_bfgs": inst.sol_time_bfgs, "num_frac_values": inst.num_frac_values,
                "solution_vector": inst.solution_vector, "name": inst.name,
            },
            "scaling_factor": {},
            "cases": {},
        }
        for kind, cls in SOLVERS.items():
            manifest["scaling_factor"][kind] = float(cls(device="cpu").get_scaling_factor(inst.q_matrix))

        cases = []
        for kind in SOLVERS:
            for t in (1, 2, 10, 100):
                cases.append((kind, t, None, None, True))
            cases.append((kind, 1500, None, None, True))
            if kind != "langevin":
                cases.append((kind, 50, None, None, False))
            if kind != "dl":  # the reference's DL Adam path raises TypeError
                for adam in ADAMS:
                    cases.append((kind, 60, adam, None, True))
            for post in ("adam", "grad-descent"):
                cases.append((kind, 50, None, post, True))
        for kind, t, adam, post, flag in cases:
            name = f"{kind}_T{t}" + (f"_{adam}" if adam else "") + (f"_{post}" if post else "") + (
                "" if flag else "_noramp")
            arrays, meta = run_case(kind, path, t, adam, post, flag)
            for k, v in arrays.items():
                store[f"{name}/{k}"] = v
            manifest["cases"][name] = meta
            print(tag, name, meta["best_objective_value"])

        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **store)
        with open(os.path.join(OUT, f"{tag}.json"), "w") as fh:
            json.dump(dict(manifest, made_with=manifest.get("made_with") or made_with()), fh, indent=1, sort_keys=True)

    anchors()
    distribution_cases()
    bounds_cases()
    vector_s_cases()
    full_s_cases()
    asgd_cases()
    larger_n_cases()


if __name__ == "__main__":
    main()
