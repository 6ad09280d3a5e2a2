"""profiles/<tag>_parity.md from the records of `CCVM_PARITY_RECORD=... pytest -m gpu tests/test_gpu_long_parity.py`
(developer tool).   python tools/make_parity_md.py r04 > profiles/r04_parity.md"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
rows = [json.loads(line) for line in open(os.path.join("profiles", f"{tag}_parity.jsonl"))]
last = {}
for r in rows:
    last[r["case"]] = r
print(f"# Round {tag[1:].lstrip('0')}: long-trajectory parity at every BASELINE.json shape (MI355X, `pytest -m gpu tests/test_gpu_long_parity.py`)\n")
print("HIP engine through the public solver API in **replay mode** (the normals come from torch's CPU stream in the "
      "reference's order, i.e. identical seeded noise) against the oracle (`oracle/ccvm_oracle.py`: the reference's op "
      "sequence on the host, bit-identical to the reference on the golden cases), default launch policy (`kernel`: "
      f"`ccvm_describe_launch`).  Raw records: `profiles/{tag}_parity.jsonl`.\n")
print("| case | solver | N | batch | steps | kernel | max abs dx (of max abs x) | max abs d(objective) / max abs objective | best objective: engine vs oracle |")
print("|---|---|---|---|---|---|---|---|---|")
worst_x = worst_o = 0.0
for case, r in last.items():
    fields = "; ".join(f"{k}: {v['max_abs_err']:.2e} (of {v['max_abs_value']:.3g})" for k, v in r["fields"].items())
    for v in r["fields"].values():
        worst_x = max(worst_x, v["max_abs_err"] / max(1.0, v["max_abs_value"]))
    o = r["objective_values"]
    worst_o = max(worst_o, o["rel"])
    solver = r["solver"] + (" (Adam variant)" if r["adam"] else "") + (f" + {r['post_processor']} post-processor" if r["post_processor"] else "")
    b = r["best_objective_value"]
    print(f"| {case} | {solver} | {r['N']} | {r['batch']} | {r['iterations']} | {r.get('kernel', '')} | {fields} | "
          f"{o['max_abs_err']:.2e} / {o['max_abs_value']:.5g} = {o['rel']:.1e} | {b['engine']:.4f} vs {b['oracle']:.4f} |")
print(f"\nWorst case of this run: amplitudes {worst_x:.1e} of the array's range, objective values {worst_o:.1e} relative "
      "(gates: 3e-4 and 1e-5).")
