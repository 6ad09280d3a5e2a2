#!/bin/bash
# Same-box sweep (developer tool, round 6): the short part of the five-wave shape's unequal K split, shorter / longer than the
# table of persist_wide_xs by DELTA k-steps (variant builds of the library: tools/ab_build.sh -DCCVM_PERSIST_XS_DELTA=...).
#   usage: bash tools/ab_persist_xs_delta.sh > gpurun_out/r06/ab_persist_xs_delta.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CASES=""
for k in dl langevin; do for n in 257 288 300 320; do CASES="$CASES $k:$n:1000"; done; done
export CCVM_AMD_PERSIST_WIDE=1
unset CCVM_AMD_LIB; python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/delta=0 /"
for d in -16 -8 -4 4 8; do
  bash tools/ab_build.sh "-DCCVM_PERSIST_XS_DELTA=$d" > /dev/null 2>&1 || { echo "variant build failed: $d"; continue; }
  CCVM_AMD_LIB=/tmp/libccvm_variant.so python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/delta=$d /"
done
echo AB_DONE
