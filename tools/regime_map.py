"""The regime map: per-step time of the default launch policy over (solver, N, B), with the kernel family each point
takes (developer tool).  Writes gpurun_out/regime_map.jsonl; `--md TAG` prints profiles/<TAG>_regime_map.md from it.
   python tools/regime_map.py            (GPU box)
   python tools/regime_map.py --md r03   (anywhere)"""
import json, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NS = (100, 256, 300, 500, 640, 768, 1000, 1500, 2000)
BS = (1, 8, 32, 64, 128, 256, 384, 512, 768, 1000, 1100, 1500, 2000, 4000)
KINDS = ("dl", "langevin", "mf")
PATH = os.path.join("gpurun_out", "regime_map.jsonl")


def family(kernel):
    if kernel.startswith("batch cut in two"):
        # "batch cut in two: rows a-b <plan> | rows c-d <plan>"; a part may be cut again, and the library's 512-byte
        # buffers may truncate a nested description: name the families in the order their kernels appear
        parts = re.findall(r"rows \d+-\d+ (?:batch cut in two: )?(ccvm::\w+<[^>]*>[^|]*)", kernel)
        return "+".join(family(part) for part in parts) or "?"
    if "persist_kernel" in kernel:
        return "R"
    if "slab_kernel" in kernel:
        return "S"
    if "cluster_kernel" in kernel:
        return "C"
    if "ptile_kernel" in kernel:
        sliced = re.search(r"(\d+) slices", kernel)
        return "P" + (sliced.group(1) if sliced else "")
    return "T" + re.search(r"step_kernel<\d, \w+, 0, (\d)", kernel).group(1)


def measure():
    import torch
    import bench
    out = open(PATH, "w")
    for kind in KINDS:
        for n in NS:
            for b in BS:
                traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0)
                kernel = bench.describe_launch(kind, b, n)
                traj.advance(128)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                traj.advance(128)
                torch.cuda.synchronize()
                est = (time.perf_counter() - t0) / 128
                steps = int(min(4096, max(128, 0.06 / est)))
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    traj.advance(steps)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                traj.check()
                rec = {"kind": kind, "n": n, "b": b, "us": best / steps * 1e6, "steps": steps, "kernel": kernel}
                print(json.dumps(rec), file=out, flush=True)
                print(f"{kind}:{n}:{b:<6d} {rec['us']:9.3f} us/step  {family(kernel)}  {kernel[:60]}", flush=True)
                del traj


def markdown(tag, rows=None, how=None):
    """`rows`: records {kind, n, b, us, kernel or family} (default: this tool's own measurements)."""
    rows = rows if rows is not None else [json.loads(line) for line in open(PATH)]
    print(f"# Round {tag[1:].lstrip('0')}: the regime map (1x MI355X, default launch policy)\n")
    print((how or "`python3 tools/regime_map.py`: run calls of the engine (fused noise; 128-4096 steps, about 60 ms each), best of 3, "
          "no profiler.") + "  Cell = us per step, kernel family: R row-owner persistent, S column-slab persistent, C "
          "column-cluster persistent, P persistent tile (32 x 128 tiles resident over the chunk; round 4; Pk: k slices of the batch one after the other; X+Y: the batch cut in two, rows of whole resident grids + the rest), T1 / T2 / T4 per-step tile kernel with 32 x 128 / 32 x 64 / 32 x 32 tiles.  Second "
          "table: fraction of the fp32 MFMA peak (157.3 TFLOP/s; DL 4 N^2 B flop per step, the others 2 N^2 B).\n")
    ns, bs = sorted({r["n"] for r in rows}), sorted({r["b"] for r in rows})
    for kind in KINDS:
        print(f"## {kind}\n")
        for what in ("us", "frac"):
            print("| N \\\\ B | " + " | ".join(str(b) for b in bs) + " |")
            print("|---|" + "---|" * len(bs))
            for n in ns:
                cells = []
                for b in bs:
                    r = next((r for r in rows if r["kind"] == kind and r["n"] == n and r["b"] == b), None)
                    if not r:
                        cells.append("")
                    elif what == "us":
                        cells.append(f"{r['us']:.2f} {r.get('family') or family(r['kernel'])}")
                    else:
                        flop = (4.0 if kind == "dl" else 2.0) * n * n * b
                        cells.append(f"{flop / (r['us'] * 1e-6) / 157.3e12:.2f}")
                print(f"| {n} | " + " | ".join(cells) + " |")
            print()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--md":
        markdown(sys.argv[2])
    else:
        measure()
