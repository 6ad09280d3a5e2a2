"""Summarise rocprofv3 --pmc passes into profiles/rNN_<name>_pmc.json (developer tool).

    python tools/pmc_summary.py --kernel 'step_kernel<0' --name 'ccvm::step_kernel<MODE_DL> (N=1000, B=1000)' \
        --command '<what was profiled>' --out profiles/r02_bench_pmc.json  DIR [DIR ...]

Each DIR is the output directory of ONE `rocprofv3 --pmc <counters> --kernel-trace --output-format csv`
pass (counters are collected in passes of their own: FETCH_SIZE and WRITE_SIZE do not fit one pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots").  For every counter the mean over the dispatches whose kernel
name contains --kernel is recorded, with the dispatch count.  gfx950 note carried into the file:
FETCH_SIZE counts half of a wide coalesced read, so HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB.
"""
import argparse
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--kernel", required=True, help="substring of the kernel name")
    ap.add_argument("--name", required=True)
    ap.add_argument("--command", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--skip", type=int, default=0, help="skip the first dispatches of the kernel (warm-up)")
    ap.add_argument("--steps-per-dispatch", type=float, default=1.0,
                    help="time steps a dispatch of this kernel runs, mean over the recorded dispatches (persistent kernels)")
    args = ap.parse_args()
    counters = {}
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = {}
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    if args.kernel not in row["Kernel_Name"]:
                        continue
                    key = (row["Counter_Name"], int(row["Dispatch_Id"]))
                    per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
            by_counter = {}
            for (name, disp), value in sorted(per_dispatch.items(), key=lambda kv: kv[0][1]):
                by_counter.setdefault(name, []).append(value)
            for name, values in by_counter.items():
                values = values[args.skip:]
                if values:
                    counters[name] = {"dispatches": len(values), "mean_per_dispatch": sum(values) / len(values)}
    doc = {
        "kernel": args.name,
        "command": args.command,
        "note": "gfx950: FETCH_SIZE counts half of a wide coalesced read (MI355X_MICROARCH.md, HBM): HBM-side bytes "
                "per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB",
        "steps_per_dispatch": args.steps_per_dispatch,
        "counters": dict(sorted(counters.items())),
    }
    with open(args.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc["counters"], indent=1))


if __name__ == "__main__":
    main()
