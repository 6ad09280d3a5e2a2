#!/bin/bash
# usage: ab_short.sh "ENV=val" [reps]   -- the driver's short run (20 timed steps after 5 warm-up steps) without / with an
# environment setting, alternating: what a 0.7 ms timed region loses to launch and completion latency
SETTING=$1; REPS=${2:-5}
for rep in $(seq $REPS); do
  for mode in base new; do
    if [ $mode = new ]; then export "$SETTING"; else unset "${SETTING%%=*}"; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$mode rep$rep: value %.4e  ms/step %.5f  step_us(events) %.3f  frac %.3f frac_wall %.3f' % (d['value'], d['ms_per_step'], r['avg_step_us'], r['frac'], r['frac_wall']))"
  done
done
