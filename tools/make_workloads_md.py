"""Tables of profiles/rNN_workloads.md from the collected files (developer tool):
   python tools/make_workloads_md.py r02  -> prints the workload table and the mid-size table (markdown)."""
import csv
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = "profiles/"
W = [("dl_n1000_b1000", "bench", "headline, config 4 per GPU", f"{tag}_bench.json", 4),
     ("pl_n2000_b512", "pl_n2000_b512", "config 5 per GPU; bench line with --post adam", f"{tag}_bench_pl_n2000_b512_adam.json", 2),
     ("mf_n500_b1000", "mf_n500_b1000", "config 3", f"{tag}_bench_mf_n500_b1000.json", 2),
     ("langevin_n500_b1000", "langevin_n500_b1000", "config 3", f"{tag}_bench_langevin_n500_b1000.json", 2),
     ("dl_n500_b1000", "dl_n500_b1000", "not a BASELINE configuration: DL at config 3's size", f"{tag}_bench_dl_n500_b1000.json", 4),
     ("dl_n100_b1000", "dl_n100_b1000", "config 2", f"{tag}_bench_dl_n100_b1000.json", 4)]
print("| workload | kernel (as rocprofv3 names it) | calls | avg ns per launch | min ns | steps per launch | bench us/step (HIP events) | row-steps/s | TFLOP/s | frac of 157.3 |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, stem, note, bench, _ in W:
    with open(f"{P}{tag}_{stem}_kernel_stats.csv", newline="") as fh:
        row = next(csv.DictReader(fh))
    d = json.loads(open(P + bench).read().strip().splitlines()[-1])
    r = d["roofline"]
    kern = re.sub(r"^void |\(.*$", "", row["Name"])
    print(f"| {name} ({note}) | `{kern}` | {row['Calls']} | {float(row['AverageNs']):.0f} | {row['MinNs']} | "
          f"{r['steps_per_launch']} | {r['avg_step_us']:.2f} | {d['value']:.3e} | {r['achieved']:.1f} | {r['frac']:.3f} |")
print()
print("| N | Langevin: cluster | Langevin: tile | MF: cluster | MF: tile | DL: cluster | DL: tile |")
print("|---|---|---|---|---|---|---|")
for n in (320, 384, 448, 500):
    t = {}
    for k in ("auto", "nocluster"):
        for line in open(f"{P}{tag}_mid_n{n}_b1000_{k}_unprofiled_timing.txt"):
            m = re.match(r"(\w+):\d+:\d+\s+RU=\w+\s+([\d.]+) us/step", line)
            if m:
                t[(m.group(1), k)] = float(m.group(2))
    print(f"| {n} | " + " | ".join(f"{t[(s, k)]:.2f}" for s in ("langevin", "mf", "dl") for k in ("auto", "nocluster")) + " |")
