#!/bin/bash
# usage: tools/ab_variant.sh "<extra hipcc flags>" workload...  -- on the GPU box: build a variant of the library with extra
# compiler flags (tools/ab_build.sh -> /tmp/libccvm_variant.so) and alternate bench runs of the in-tree library ("base")
# and the variant ("new") on the same box
FLAGS=$1; shift
bash tools/ab_build.sh "$FLAGS" > /dev/null 2>&1 || { echo "variant build failed: $FLAGS"; exit 1; }
echo "== variant: $FLAGS"
for w in "$@"; do
  for rep in 1 2; do
    for lib in base new; do
      if [ $lib = new ]; then export CCVM_AMD_LIB=/tmp/libccvm_variant.so; else unset CCVM_AMD_LIB; fi
      python bench.py --workload $w --steps 3000 --warmup 300 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w $lib rep$rep: ms/step %.5f  step_us %.3f  frac %.3f' % (d['ms_per_step'], r['avg_step_us'], r['frac']))"
    done
  done
done
