#!/bin/bash
# usage: ab_env.sh "ENV=val" workload...   -- alternating bench runs without / with the environment setting
mkdir -p gpurun_out
SETTING=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for mode in base new; do
      if [ $mode = new ]; then export "$SETTING"; else unset "${SETTING%%=*}"; fi
      python bench.py --workload $w --steps 3000 --warmup 300 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w $mode rep$rep: ms/step %.5f  step_us %.3f  frac %.3f  best %.4f %s' % (d['ms_per_step'], r['avg_step_us'], r['frac'], d['check']['best_objective_value'], r['kernel'][:50]))"
    done
  done
done
