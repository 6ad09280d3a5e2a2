"""profiles/rNN_small_batch.md from gpurun_out/small_batch_sweep.jsonl (developer tool; tools/small_batch_sweep.py)."""
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
rows = [json.loads(line) for line in open("gpurun_out/small_batch_sweep.jsonl")]
print(f"# Round {tag[1:].lstrip('0')}: small batches above N = 256 (1x MI355X)\n")
print("`python3 tools/small_batch_sweep.py`: 2048-step run calls of the engine (fused noise), best of 3, no profiler; "
      "default policy against `CCVM_AMD_KERNEL=noslab` (what ran before round 3: the column-cluster kernel up to "
      "N = 768, the per-step tile kernel above).  The reference runs any batch through the same einsum "
      "(`dl_solver.py:145-153`).  `plan` = clusters x members x columns per member, rows per cluster, XCDs a cluster "
      "spans; `-` = no slab plan (the batch's clusters do not fit the chip, or the staged input of a member exceeds "
      "128 KB of LDS): the default is then the previous path.\n")
print("| solver | N | B | default: us/step | row-steps/s | plan | noslab: us/step | speed-up |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    m = re.search(r"slab_kernel<\d, (\d+), (\d+), \w+>.*\((\d+) clusters of (\d+) workgroups x (\d+) columns, (\d+) rows each, K = (\d+)(?:, each over (\d+) XCDs)?",
                  r["auto_kernel"])
    plan = "-"
    if m:
        plan = f"{m.group(3)} x {m.group(4)} x {m.group(5)}, {m.group(6)} rows, {m.group(8) or 1} XCD"
    print(f"| {r['kind']} | {r['n']} | {r['b']} | {r['auto_us']:.2f} | {r['b'] / r['auto_us'] * 1e6:.3g} | {plan} | "
          f"{r['noslab_us']:.2f} | {r['noslab_us'] / r['auto_us']:.1f}x |")
