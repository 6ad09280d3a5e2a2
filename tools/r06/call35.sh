#!/bin/bash
# round 6, call 35: the driver's commands on the final tree -- GPU suite, smoke, the default bench line and the 20-step form
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite_final.txt 2>&1 || { echo SUITE_FAILED; tail -30 $O/gpu_suite_final.txt; exit 1; }
tail -1 $O/gpu_suite_final.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke_final.txt 2>&1 || { echo SMOKE_FAILED; tail -5 $O/smoke_final.txt; exit 1; }
tail -1 $O/smoke_final.txt
timeout -k 10 200 python bench.py > $O/bench_final.json 2> $O/bench_final.err || { echo BENCH_FAILED; tail -5 $O/bench_final.err; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_final_steps20.json 2> $O/bench_final_steps20.err || { echo BENCH20_FAILED; tail -5 $O/bench_final_steps20.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_final", "bench_final_steps20"):
    d = json.loads(open(f"gpurun_out/r06/{f}.json").read().strip().splitlines()[-1])
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"))
PY
echo CALL35_DONE
