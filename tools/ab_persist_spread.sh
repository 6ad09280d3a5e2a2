#!/bin/bash
# Same-box A/B (developer tool, round 6): CCVM_AMD_PERSIST_SPREAD=1 -- LDS padding so that a CU takes no more row-owner
# workgroups than the grid needs -- against the default placement, by shape and batch.
# A RECORD, not a runnable tool: the padding (dynamic LDS of 160 KB / (workgroups per CU + 1) + 1 KB in the row-owner launch)
# changed no cell by 3 % (profiles/r06_ab_persist_spread.txt) and was taken out of the library again; the variable no longer exists.
#   usage: bash tools/ab_persist_spread.sh > gpurun_out/r06/ab_persist_spread.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
CASES=""
for k in dl langevin mf; do
  for n in 20 45 64; do for b in 100 300 500 700 1000 1500 2000; do CASES="$CASES $k:$n:$b"; done; done
  for n in 70 100 128; do for b in 128 256 384 512 768 1000 1500; do CASES="$CASES $k:$n:$b"; done; done
  for n in 144 192 256; do for b in 128 256 384 512 768 1000; do CASES="$CASES $k:$n:$b"; done; done
done
for rep in 1 2; do
for spread in 0 1; do
  export CCVM_AMD_PERSIST_SPREAD=$spread
  for kh in auto 1; do
    if [ $kh = auto ]; then unset CCVM_AMD_PERSIST_KH; else export CCVM_AMD_PERSIST_KH=$kh; fi
    python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/rep$rep spread=$spread KH=$kh /" || exit 1
  done
done
done
echo AB_DONE
