#!/bin/bash
# usage: ab.sh <base.so> workload...   -- alternating A/B bench runs on the same box
mkdir -p gpurun_out
BASE=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for lib in base new; do
      if [ $lib = base ]; then export CCVM_AMD_LIB=$PWD/$BASE; else unset CCVM_AMD_LIB; fi
      python bench.py --workload $w --steps 3000 --warmup 300 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w $lib rep$rep: ms/step %.5f  step_us %.3f  frac %.3f' % (d['ms_per_step'], r['avg_step_us'], r['frac']))"
    done
  done
done
