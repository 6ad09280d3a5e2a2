"""Rewrites the GENERATED blocks of DESIGN.md from the committed bench lines and counter summaries under profiles/
(developer tool; VERDICT r4 item 7: the numbers of the roofline table are not typed by hand).

    python tools/make_design_tables.py [--check]

A block is delimited by `<!-- BEGIN GENERATED: name -->` / `<!-- END GENERATED: name -->`.  Blocks:
  roofline   one row per benchmarked workload: the newest profiles/rNN_bench[_<workload>].json line, the PMC summary of
             the same workload where one is committed (matrix-pipe share, HBM-side bytes per step)
--check: exit 1 when DESIGN.md is not what this tool would write (tests/test_host_logic.py runs it)."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")

# (workload, what it is in BASELINE.json's terms, algorithmic bytes per row-step as a formula string)
ROWS = [
    ("dl_n1000_b1000", "**headline**; config 4 per GPU", "16 N + 4 N²/B"),
    ("dl_n1000_b1000@steps20", "the same as the driver runs it (`--steps 20 --warmup 5`)", "16 N + 4 N²/B"),
    ("pl_n2000_b512_adam", "config 5 per GPU (`--post adam`)", "8 N + 4 N²/B"),
    ("mf_n500_b1000", "config 3", "16 N + 4 N²/B"),
    ("langevin_n500_b1000", "config 3", "8 N + 4 N²/B"),
    ("dl_n100_b1000", "config 2", "16 N + 4 N²/B"),
    ("dl_n20_b100", "**config 1** exactly: `test020-100-10`, batch 100, inside the schedule of a 15000-iteration run", "16 N + 4 N²/B"),
    ("dl_n20_b1000", "the shipped example (`examples/ccvm_boxqp_dl.py:12-24`: `tuningH020-100-0`, batch 1000)", "16 N + 4 N²/B"),
    ("mf_n20_b1000", "the same instance, MF (`examples/ccvm_boxqp_mf.py`)", "16 N + 4 N²/B"),
    ("langevin_n20_b1000", "the same instance, Langevin (`examples/langevin_boxqp.py`)", "8 N + 4 N²/B"),
    ("pl_n20_b1000", "the same instance, pumped Langevin (`examples/pumped_langevin_boxqp.py`)", "8 N + 4 N²/B"),
    ("dl_n70_b1000", "the largest shipped size (synthetic N = 70)", "16 N + 4 N²/B"),
    ("dl_n160_b1000", "128 < N ≤ 192: three waves side by side × two K halves, two row sets per twelve-wave workgroup (round 6; 1.84 µs before)", "16 N + 4 N²/B"),
    ("mf_n257_b1000", "MF just above 256: five waves side by side, unequal K split, part of Q in LDS (round 6; the cluster kernel before: 3.52 µs)", "16 N + 4 N²/B"),
    ("dl_n1000_b2000", "config 4 per GPU on 4 GPUs (strong scaling)", "16 N + 4 N²/B"),
    ("dl_n1000_b4000", "config 4 per GPU on 2 GPUs", "16 N + 4 N²/B"),
    ("pl_n2000_b1024", "config 5 per GPU on 4 GPUs", "8 N + 4 N²/B"),
    ("langevin_n1000_b1000", "one-stream solver at the headline's size", "8 N + 4 N²/B"),
    ("mf_n1000_b1000", "one-stream solver at the headline's size", "16 N + 4 N²/B"),
    ("dl_n500_b1000", "DL at config 3's size (not a BASELINE configuration)", "16 N + 4 N²/B"),
    ("langevin_n300_b1000", "256 < N ≤ 320: the row-owner kernel's five waves side by side since round 6 (the cluster kernel's half-chunk variant before: 3.36 µs)", "8 N + 4 N²/B"),
    ("dl_n300_b1000", "the same, DL", "16 N + 4 N²/B"),
    ("langevin_n640_b512", "K = 640 at half the batch: clusters of 32 rows (two row sets)", "8 N + 4 N²/B"),
    ("dl_n640_b512", "the same, DL", "16 N + 4 N²/B"),
    ("dl_n1000_b256", "mid-size batch", "16 N + 4 N²/B"),
    ("dl_n1000_b32", "small batch (the reference runs any batch_size)", "16 N (Q never moves)"),
    ("dl_n1000_b1", "a single trajectory", "16 N (Q never moves)"),
    ("pl_n2000_b32", "small batch at config 5's size", "8 N (Q never moves)"),
]


def newest(pattern):
    files = sorted(glob.glob(os.path.join(P, pattern)))
    return files[-1] if files else None


def bench_line(workload):
    if workload == "dl_n1000_b1000":
        f = newest("r[0-9][0-9]_bench.json")
    elif workload == "dl_n1000_b1000@steps20":
        f = newest("r[0-9][0-9]_bench_steps20.json")
    else:
        f = newest(f"r[0-9][0-9]_bench_{workload}.json")
    if not f:
        return None, None
    text = open(f).read().strip().splitlines()[-1]
    return json.loads(text), os.path.relpath(f, ROOT)


def pmc(workload):
    stem = "bench" if workload.startswith("dl_n1000_b1000") else workload.replace("_adam", "")
    f = newest(f"r[0-9][0-9]_{stem}_pmc.json")
    if not f:
        return None, None
    return json.load(open(f)), os.path.relpath(f, ROOT)


def short_kernel(k):
    m = re.search(r"(\w+_kernel\w*<[^>]*>)", k)
    name = m.group(1) if m else k[:40]
    extra = re.search(r"(\d+) slices of the batch", k)
    return f"`{name}`" + (f", {extra.group(1)} slices" if extra else "")


def roofline_block():
    out = ["| workload (`python bench.py --workload …`) | what | kernel | µs/step (HIP events) | row-steps/s | hardware roof | achieved / peak | frac (kernel) | frac (wall) | model: bound, frac | matrix pipe busy | HBM-side bytes per step (PMC) vs algorithmic | source |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for workload, what, _ in ROWS:
        d, src = bench_line(workload)
        if d is None:
            continue
        r = d["roofline"]
        c, csrc = pmc(workload)
        busy = traffic = ""
        if c:
            cc = c["counters"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cc and "SQ_BUSY_CYCLES" in cc:
                busy = "%.0f %%" % (100.0 * (cc["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / 1024.0)
                                    / (cc["SQ_BUSY_CYCLES"]["mean_per_dispatch"] / 32.0))
            if "FETCH_SIZE" in cc and "WRITE_SIZE" in cc:
                per_step = 1024.0 * (2.0 * cc["FETCH_SIZE"]["mean_per_dispatch"] + cc["WRITE_SIZE"]["mean_per_dispatch"]) \
                    / c.get("steps_per_dispatch", 1.0)
                alg = r.get("algorithmic_bytes")
                spl = r.get("steps_per_launch") or 1
                traffic = "%.1f MB" % (per_step / 1e6) + (" vs %.1f MB" % (alg / spl / 1e6) if alg else "")
            src = f"{src}, {csrc}"
        unit = r["unit"]
        model = f"{r['model']['bound']} {r['model_frac']:.2f}" if "model" in r else ""
        ach = f"{r['achieved']:.1f} / {r['peak']:.1f} {unit}" if unit == "TFLOP/s" else f"{r['achieved']:.3g} / {r['peak']:.3g} {unit}"
        out.append(f"| `{workload.split('@')[0].replace('_adam', '')}` | {what} | {short_kernel(r['kernel'])} | {r['avg_step_us']:.2f} | "
                   f"{d['value']:.3e} | {r['bound']} | {ach} | {r['frac']:.3f} | {r.get('frac_wall', float('nan')):.3f} | {model} | {busy} | "
                   f"{traffic} | `{src}` |")
    return "\n".join(out)


BLOCKS = {"roofline": roofline_block}


def render(text):
    for name, fn in BLOCKS.items():
        pat = re.compile(rf"(<!-- BEGIN GENERATED: {name} -->\n).*?(<!-- END GENERATED: {name} -->)", re.S)
        if not pat.search(text):
            raise SystemExit(f"DESIGN.md has no GENERATED block {name!r}")
        text = pat.sub(lambda m: m.group(1) + fn() + "\n" + m.group(2), text)
    return text


if __name__ == "__main__":
    path = os.path.join(ROOT, "DESIGN.md")
    old = open(path).read()
    new = render(old)
    if "--check" in sys.argv:
        sys.exit(0 if new == old else 1)
    open(path, "w").write(new)
    print("DESIGN.md: generated blocks rewritten" if new != old else "DESIGN.md: up to date")
