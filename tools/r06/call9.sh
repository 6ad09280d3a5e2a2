#!/bin/bash
# round 6, GPU call 9: the cluster kernel's fetch waves make the next phase's normals -- parity, then a same-box A/B
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_cluster.py tests/test_gpu_thick_goldens.py tests/test_gpu_persist_producers.py -q > $O/call9_pytest.txt 2>&1 || { grep -E "^(FAILED|E  )" $O/call9_pytest.txt | head -20; }
tail -3 $O/call9_pytest.txt
timeout -k 10 600 tools/ab_build.sh "-DCCVM_CL_FETCH_NOISE=0" > $O/ab_build.txt 2>&1 || { tail $O/ab_build.txt; exit 1; }
for rep in 1 2 3; do
  for w in langevin_n500_b1000 mf_n500_b1000 langevin_n300_b1000; do
    for lib in new old; do
      if [ $lib = old ]; then export CCVM_AMD_LIB=/tmp/libccvm_variant.so; else unset CCVM_AMD_LIB; fi
      timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w $lib rep$rep: step_us %.3f  frac %.3f  best %.4f  %s' % (r['avg_step_us'], r['frac'], d['check']['best_objective_value'], r['kernel'][:48]))" >> $O/ab_fetch_noise.txt || exit 1
    done
  done
done
unset CCVM_AMD_LIB
cat $O/ab_fetch_noise.txt
echo CALL9_DONE
