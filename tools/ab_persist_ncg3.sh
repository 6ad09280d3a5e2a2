#!/bin/bash
# Same-box A/B (developer tool, round 6): three waves side by side for 128 < N <= 192 (-DCCVM_PERSIST_NCG3=1, built into
# /tmp/libccvm_variant.so by tools/ab_build.sh) against four, every K-split setting (auto / off / on), B = 256 ... 4000.
#   usage: bash tools/ab_persist_ncg3.sh > gpurun_out/r06/ab_persist_ncg3.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/ab_build.sh "-DCCVM_PERSIST_NCG3=1" > /dev/null 2>&1 || { echo "variant build failed"; exit 1; }
CASES=""
for k in dl langevin mf; do for n in 130 144 160 176 192; do for b in 256 512 768 1000 1500 2000 4000; do CASES="$CASES $k:$n:$b"; done; done; done
for k in langevin mf; do for n in 144 160 192; do for b in 512 1000 2000; do CASES="$CASES $k:$n:$b:adam"; done; done; done
for lib in four three; do
  if [ $lib = three ]; then export CCVM_AMD_LIB=/tmp/libccvm_variant.so; else unset CCVM_AMD_LIB; fi
  for kh in auto 1 2; do
    if [ $kh = auto ]; then unset CCVM_AMD_PERSIST_KH; else export CCVM_AMD_PERSIST_KH=$kh; fi
    python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$lib KH=$kh /" || exit 1
  done
done
echo AB_DONE
