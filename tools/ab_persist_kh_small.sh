#!/bin/bash
# Same-box sweep (developer tool, round 6): the K split at SMALL batches, waves side by side (N > 64) -- KH auto / off / on,
# in-tree library ("four") and the three-column-group variant ("three": -DCCVM_PERSIST_NCG3=1, tools/ab_build.sh).
#   usage: bash tools/ab_persist_kh_small.sh > gpurun_out/r06/ab_persist_kh_small.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/ab_build.sh "-DCCVM_PERSIST_NCG3=1" > /dev/null 2>&1 || { echo "variant build failed"; exit 1; }
CASES=""
for k in dl langevin mf; do
  for n in 144 176 200 224 256; do for b in 64 128 192 256 320 384 512 640; do CASES="$CASES $k:$n:$b"; done; done
  for n in 80 100 128; do for b in 32 64 128 256 512; do CASES="$CASES $k:$n:$b"; done; done
done
for k in langevin mf; do for n in 144 224 256; do for b in 128 256 512; do CASES="$CASES $k:$n:$b:adam"; done; done; done
for lib in four three; do
  if [ $lib = three ]; then export CCVM_AMD_LIB=/tmp/libccvm_variant.so; else unset CCVM_AMD_LIB; fi
  for kh in auto 1 2; do
    if [ $kh = auto ]; then unset CCVM_AMD_PERSIST_KH; else export CCVM_AMD_PERSIST_KH=$kh; fi
    python3 tools/time_small.py $CASES 2>&1 | grep "us/step" | sed "s/^/$lib KH=$kh /" || exit 1
  done
done
echo AB_DONE
