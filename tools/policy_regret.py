"""Launch-policy regret audit (developer tool): for every cell of the regime map, time the DEFAULT
plan and every forced kernel family / tile shape that can serve the cell, and list
  * cells where the default is more than 5 % slower than the best forced plan (the policy's regret), and
  * cells where a LARGER batch of the same (solver, N) is faster than a smaller one (non-monotone in B).
The tuning environment (CCVM_AMD_KERNEL / _KS / _SPLIT) is read by the library at every run call, so the variants run in
one process; a variant is identified by what ccvm_describe_launch says runs under it (a forced family that does not
apply falls through to another plan and is deduplicated).

   python tools/policy_regret.py [--kinds dl,langevin,mf] [--ns ...] [--bs ...]   (GPU box; appends to gpurun_out/policy_regret.jsonl)
   python tools/policy_regret.py --md r05                                         (anywhere: prints profiles/<TAG>_policy_regret.md)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NS = (100, 256, 300, 500, 640, 768, 1000, 1500, 2000)
BS = (1, 8, 32, 64, 128, 256, 384, 512, 768, 1000, 1100, 1500, 2000, 4000)
KINDS = ("dl", "langevin", "mf")
PATH = os.path.join("gpurun_out", "policy_regret.jsonl")
TUNING_VARS = ("CCVM_AMD_KERNEL", "CCVM_AMD_KS", "CCVM_AMD_SPLIT")

#: name -> environment of the variant
VARIANTS = {
    "default": {},
    "tile1": {"CCVM_AMD_KERNEL": "tile", "CCVM_AMD_KS": "1"},
    "tile2": {"CCVM_AMD_KERNEL": "tile", "CCVM_AMD_KS": "2"},
    "tile4": {"CCVM_AMD_KERNEL": "tile", "CCVM_AMD_KS": "4"},
    "cluster": {"CCVM_AMD_KERNEL": "cluster"},
    "slab": {"CCVM_AMD_KERNEL": "slab"},
    "ptile": {"CCVM_AMD_KERNEL": "ptile"},
    "nocut": {"CCVM_AMD_SPLIT": "0"},
    "cut": {"CCVM_AMD_SPLIT": "1"},
}


def variants_for(n):
    names = ["default", "tile1", "tile2", "tile4", "nocut", "cut"]
    if 257 <= n <= 768:
        names.append("cluster")
    if n > 256:
        names.append("slab")
    if n > 768:
        names.append("ptile")
    return names


def set_env(env):
    for var in TUNING_VARS:
        os.environ.pop(var, None)
    os.environ.update(env)


ADAM = None  # --adam: the example scripts' Adam variant of MF / Langevin (alpha 0.001, beta1 0.9, beta2 0.999)


def describe(kind, b, n):
    import ctypes

    import bench
    from ccvm_amd import _lib

    buf = ctypes.create_string_buffer(1024)
    _lib.check(_lib.load().ccvm_describe_launch(bench.SOLVER_ID[kind], b, n, 1 if ADAM else 0, 0, buf, 1024), "ccvm_describe_launch")
    return buf.value.decode()


def time_cell(kind, n, b, budget_s=0.05):
    """us per step of the plan the current environment selects; None when a persistent kernel timed out."""
    import torch

    import bench

    traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0, adam=ADAM)
    traj.advance(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    traj.advance(64)
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 64
    steps = int(min(4096, max(64, budget_s / est)))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        traj.advance(steps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    recovered = traj.check(rerun=False)
    del traj
    return None if recovered else best / steps * 1e6


def measure(kinds, ns, bs):
    import bench
    from tools.regime_map import family

    done = set()
    if os.path.exists(PATH):
        for line in open(PATH):
            r = json.loads(line)
            done.add((r["kind"], r["n"], r["b"]))
    out = open(PATH, "a")
    for kind in kinds:
        for n in ns:
            for b in bs:
                if (kind + ("+adam" if ADAM else ""), n, b) in done:
                    continue
                seen, rec = {}, {"kind": kind + ("+adam" if ADAM else ""), "n": n, "b": b, "plans": []}
                for name in variants_for(n):
                    set_env(VARIANTS[name])
                    try:
                        kernel = describe(kind, b, n)
                    except Exception as exc:  # noqa: BLE001 -- a forced plan the library refuses
                        rec["plans"].append({"variant": name, "error": str(exc)[:120]})
                        continue
                    if kernel in seen:
                        seen[kernel]["variants"].append(name)
                        continue
                    try:
                        us = time_cell(kind, n, b)
                    except Exception as exc:  # noqa: BLE001
                        rec["plans"].append({"variant": name, "error": str(exc)[:120]})
                        continue
                    plan = {"variants": [name], "family": family(kernel), "us": us, "kernel": kernel[:200]}
                    seen[kernel] = plan
                    rec["plans"].append(plan)
                set_env({})
                print(json.dumps(rec), file=out, flush=True)
                d = next(p for p in rec["plans"] if "default" in p.get("variants", ()))
                ok = [p for p in rec["plans"] if p.get("us")]
                bestp = min(ok, key=lambda p: p["us"])
                print(f"{kind}:{n}:{b:<5d} default {d['us'] or -1:9.3f} {d['family']:7s} best {bestp['us']:9.3f} "
                      f"{bestp['family']:7s} ({','.join(bestp['variants'])})", flush=True)


def load():
    cells = {}
    for line in open(PATH):
        r = json.loads(line)
        cells[(r["kind"], r["n"], r["b"])] = r  # a re-measured cell replaces the earlier record
    return cells


def markdown(tag):
    cells = load()
    print(f"# Round {tag[1:].lstrip('0')}: launch-policy regret audit (1x MI355X)\n")
    print("`python3 tools/policy_regret.py`: per cell of the regime map (solver, N, B) the default plan and every forced "
          "kernel family / tile shape that serves the cell (`CCVM_AMD_KERNEL=tile|cluster|slab|ptile`, `CCVM_AMD_KS=1|2|4`, "
          "`CCVM_AMD_SPLIT=0|1`; a forced family that does not apply falls through and is deduplicated by what "
          "`ccvm_describe_launch` reports), fused noise, best of 3 run calls of about 50 ms each, no profiler.  Families: "
          "R row-owner, S column-slab, C column-cluster, P persistent tile (Pk: k slices), T1 / T2 / T4 per-step tile kernel "
          "with 32 x 128 / 64 / 32 tiles, X+Y a batch cut in two.  Run-to-run noise of a cell is about 2 %.\n")
    rows, total = [], 0
    for (kind, n, b), r in sorted(cells.items()):
        ok = [p for p in r["plans"] if p.get("us")]
        d = next((p for p in ok if "default" in p["variants"]), None)
        if not d or not ok:
            continue
        total += 1
        best = min(ok, key=lambda p: p["us"])
        if d["us"] > 1.05 * best["us"]:
            rows.append((d["us"] / best["us"], kind, n, b, d, best))
    print(f"## Regret: cells where the default is > 5 % slower than the best forced plan ({len(rows)} of {total})\n")
    print("| solver | N | B | default us/step (family) | best us/step (family, how forced) | default / best |")
    print("|---|---|---|---|---|---|")
    for ratio, kind, n, b, d, best in sorted(rows, reverse=True):
        print(f"| {kind} | {n} | {b} | {d['us']:.2f} ({d['family']}) | {best['us']:.2f} ({best['family']}, "
              f"{' / '.join(best['variants'])}) | {ratio:.2f} |")
    print()
    mono = []
    ns, bs = sorted({n for _, n, _ in cells}), sorted({b for _, _, b in cells})  # (the standard grid + any extra cells)
    kinds = sorted({k for k, _, _ in cells}, key=lambda k: (KINDS + (k,)).index(k))
    for kind in kinds:
        for n in ns:
            series = []
            for b in bs:
                r = cells.get((kind, n, b))
                d = r and next((p for p in r["plans"] if p.get("us") and "default" in p["variants"]), None)
                if d:
                    series.append((b, d))
            for i, (b, d) in enumerate(series):
                larger = [(b2, d2) for b2, d2 in series[i + 1:] if d2["us"] < 0.98 * d["us"]]
                if larger:
                    b2, d2 = min(larger, key=lambda t: t[1]["us"])
                    mono.append((d["us"] / d2["us"], kind, n, b, d, b2, d2))
    print(f"## Non-monotone in B: a larger batch is > 2 % faster than a smaller one under the default policy ({len(mono)})\n")
    print("| solver | N | B | us/step (family) | faster larger batch | its us/step (family) | ratio |")
    print("|---|---|---|---|---|---|---|")
    for ratio, kind, n, b, d, b2, d2 in sorted(mono, reverse=True):
        print(f"| {kind} | {n} | {b} | {d['us']:.2f} ({d['family']}) | {b2} | {d2['us']:.2f} ({d2['family']}) | {ratio:.2f} |")
    print()
    print("## Every cell: default vs forced plans (us per step)\n")
    for kind in kinds:
        print(f"### {kind}\n")
        print("| N | B | default | other plans |")
        print("|---|---|---|---|")
        for n in ns:
            for b in bs:
                r = cells.get((kind, n, b))
                if not r:
                    continue
                ok = [p for p in r["plans"] if p.get("us")]
                d = next((p for p in ok if "default" in p["variants"]), None)
                if not d:
                    continue
                others = ", ".join(f"{p['us']:.2f} {p['family']}" for p in sorted(ok, key=lambda p: p["us"]) if p is not d)
                print(f"| {n} | {b} | {d['us']:.2f} {d['family']} | {others} |")
        print()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default=None)
    ap.add_argument("--regime-md", default=None, help="print profiles/<TAG>_regime_map.md (the default plan of every cell) from the audit's data")
    ap.add_argument("--out", default=None, help="measure: the jsonl file to append to; --md: comma-separated files to read")
    ap.add_argument("--adam", action="store_true", help="the Adam variants (kinds mf, langevin); records carry kind 'mf+adam' / 'langevin+adam'")
    ap.add_argument("--kinds", default=",".join(KINDS))
    ap.add_argument("--ns", default=",".join(map(str, NS)))
    ap.add_argument("--bs", default=",".join(map(str, BS)))
    ap.add_argument("--cells", default=None, help="measure exactly these cells, kind:N:B,... (re-measuring what a policy change moved)")
    args = ap.parse_args()
    if args.out and not (args.md or args.regime_md):
        PATH = args.out
    if (args.md or args.regime_md) and args.out:
        import tempfile
        merged = tempfile.NamedTemporaryFile("w", suffix=".jsonl", delete=False)
        for f in args.out.split(","):
            merged.write(open(f).read())
        merged.close()
        PATH = merged.name
    if args.regime_md:
        from tools.regime_map import markdown as regime_markdown

        rows = []
        for (kind, n, b), r in sorted(load().items()):
            d = next((p for p in r["plans"] if p.get("us") and "default" in p["variants"]), None)
            if d:
                rows.append({"kind": kind, "n": n, "b": b, "us": d["us"], "family": d["family"]})
        regime_markdown(args.regime_md, rows, how="`python3 tools/policy_regret.py` (the default plan of every cell of the "
                        "regret audit, `profiles/" + args.regime_md + "_policy_regret.md`): run calls of the engine, fused noise, "
                        "64-4096 steps of about 50 ms, best of 3, no profiler.")
    elif args.md:
        markdown(args.md)
    else:
        if args.adam:
            ADAM = {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False}
        if args.cells:
            for cell in args.cells.split(","):
                kind, n, b = cell.split(":")
                measure([kind], [int(n)], [int(b)])
        else:
            measure([k for k in args.kinds.split(",") if not (args.adam and k == "dl")], [int(x) for x in args.ns.split(",")],
                    [int(x) for x in args.bs.split(",")])
