#!/bin/bash
# Same-box A/B of an environment switch on the driver-style bench line (developer tool):
#   tools/ab_env_bench.sh VAR=a VAR=b [runs]     -> wall / events us per step of `bench.py --steps 20 --warmup 5`, alternating
A=$1; B=$2; N=${3:-5}
for i in $(seq $N); do
  for e in "$A" "$B"; do
    env "$e" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$e', round(d['ms_per_step']*1e3,2), round(d['roofline']['avg_step_us'],2))"
  done
done
