#!/bin/bash
# Same-box sweep (developer tool, round 6): the short part of the six-wave workgroups' unequal K split, shorter / longer than the
# table of persist_xs3 by DELTA k-steps (variant builds: tools/ab_build.sh -DCCVM_PERSIST_XS3_DELTA=...).
#   usage: bash tools/ab_persist_xs3_delta.sh > gpurun_out/r06/ab_persist_xs3_delta.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CASES=""
for n in 144 160 176 192; do CASES="$CASES dl:$n:512 langevin:$n:1000 mf:$n:1000"; done
unset CCVM_AMD_LIB; python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/delta=0 /"
for d in -8 8 16 24; do
  bash tools/ab_build.sh "-DCCVM_PERSIST_XS3_DELTA=$d" > /tmp/ab_build_$d.log 2>&1 || { echo "variant build failed: $d"; tail -3 /tmp/ab_build_$d.log; continue; }
  CCVM_AMD_LIB=/tmp/libccvm_variant.so python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/delta=$d /"
done
echo AB_DONE
