#!/bin/bash
# round 6, call 51: the final build -- the whole GPU suite, smoke, the default bench line, the MF N = 257 bench line
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite_final.txt 2>&1 || { echo SUITE_FAILED; tail -40 $O/gpu_suite_final.txt; exit 1; }
tail -1 $O/gpu_suite_final.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > $O/smoke_final.txt 2>&1 || { echo SMOKE_FAILED; tail -5 $O/smoke_final.txt; exit 1; }
tail -1 $O/smoke_final.txt
timeout -k 10 200 python bench.py > $O/bench_final.json 2> $O/bench_final.err || { echo BENCH_FAILED; tail -5 $O/bench_final.err; exit 1; }
timeout -k 10 200 python bench.py --workload mf_n257_b1000 > gpurun_out/r06_bench_mf_n257_b1000.json 2> $O/bench_mf257.err || { echo BENCH_FAILED mf; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_final.json").read().strip().splitlines()[-1])
print("bench_final value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
echo CALL51_DONE
