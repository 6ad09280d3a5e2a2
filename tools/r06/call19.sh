#!/bin/bash
# round 6, call 19: the whole GPU suite, the smoke entry and the driver's bench command on the final code
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite_final.txt 2>&1 || { echo SUITE_FAILED; tail -30 $O/gpu_suite_final.txt; exit 1; }
tail -2 $O/gpu_suite_final.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke_final.txt 2>&1 || { echo SMOKE_FAILED; tail -5 $O/smoke_final.txt; exit 1; }
tail -1 $O/smoke_final.txt
timeout -k 10 200 python bench.py > $O/bench_final.json 2> $O/bench_final.err || { echo BENCH_FAILED; tail -5 $O/bench_final.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_final.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
echo CALL19_DONE
