#!/bin/bash
# usage: ab_protocol.sh OLD_BENCH.py [reps]  -- the driver's short run (20 timed steps after 5 warm-up steps) with another
# version of bench.py (e.g. `git show <rev>:bench.py > bench_old.py` at the repository root) and with this one, alternating
OLD=$1; REPS=${2:-4}
for rep in $(seq $REPS); do
  for mode in old new; do
    if [ $mode = new ]; then B=bench.py; else B=$OLD; fi
    python $B --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$mode rep$rep: value %.4e  ms/step %.5f  step_us(events) %.3f  frac %.3f frac_wall %.3f' % (d['value'], d['ms_per_step'], r['avg_step_us'], r['frac'], r['frac_wall']))"
  done
done
