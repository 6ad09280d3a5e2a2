#!/bin/bash
# A/B of the row-owner kernel's noise producer waves (developer tool): tools/time_small.py with CCVM_AMD_PERSIST_PW=0 / 1
# (and the rows in use / K split forced where that changes what the producers compete with), alternating, two rounds.
#   usage: tools/ab_persist_pw.sh > gpurun_out/r06_ab_persist_pw.txt
cd ${GRAFT_REPO_ROOT:-$PWD}
SMALL="dl:20:100 dl:20:1000 dl:20:2000 dl:20:4000 dl:50:1000 dl:64:1000 dl:70:1000 mf:20:1000 mf:64:1000 langevin:20:1000 langevin:64:1000 pl:20:1000 mf:20:1000:adam langevin:20:1000:adam"
MID="dl:100:1000 dl:100:500 dl:128:1000 mf:100:1000 langevin:100:1000 dl:100:2000"
for rep in 1 2; do
  for pw in 0 1; do
    echo "== round $rep  CCVM_AMD_PERSIST_PW=$pw  (rows in use: policy)"
    CCVM_AMD_PERSIST_PW=$pw timeout -k 10 300 python tools/time_small.py $SMALL || exit 1
    for ru in 2 4; do
      echo "== round $rep  CCVM_AMD_PERSIST_PW=$pw CCVM_AMD_PERSIST_RU=$ru"
      CCVM_AMD_PERSIST_PW=$pw CCVM_AMD_PERSIST_RU=$ru timeout -k 10 300 python tools/time_small.py dl:20:1000 dl:50:1000 dl:64:1000 mf:20:1000 langevin:20:1000 dl:20:2000 || exit 1
    done
    for kh in 1 2; do
      echo "== round $rep  CCVM_AMD_PERSIST_PW=$pw CCVM_AMD_PERSIST_KH=$kh  (two waves side by side)"
      CCVM_AMD_PERSIST_PW=$pw CCVM_AMD_PERSIST_KH=$kh timeout -k 10 300 python tools/time_small.py $MID || exit 1
    done
  done
done
echo AB_DONE
