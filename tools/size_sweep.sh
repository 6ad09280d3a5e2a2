#!/bin/bash
# Per-step time across problem sizes at B = 1000 (developer tool): gpurun_out/size_sweep_{auto,tile}.txt
#   usage: tools/size_sweep.sh        then  python tools/make_size_sweep_md.py r03
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
CASES=""
for k in dl langevin mf; do for n in 20 45 64 70 100 128 160 200 256 257 320 384 448 500 513 576 640 700 768 896 1000 1200 1500 2000 2500; do CASES="$CASES $k:$n:1000"; done; done
# the Adam variants (MF / Langevin: second-moment Adam of the example scripts) at the cluster kernel's sizes
for k in langevin mf; do for n in 100 320 500 640 768 1000; do CASES="$CASES $k:$n:1000:adam"; done; done
python3 tools/time_small.py $CASES pl:500:1000 pl:640:1000 pl:768:1000 pl:2000:512 2>&1 | grep "us/step" > gpurun_out/size_sweep_auto.txt || exit 1
CASES=""
for k in dl langevin mf; do for n in 257 320 384 448 500 513 576 640 700 768; do CASES="$CASES $k:$n:1000"; done; done
for k in langevin mf; do for n in 320 500 640 768; do CASES="$CASES $k:$n:1000:adam"; done; done
CCVM_AMD_KERNEL=nocluster python3 tools/time_small.py $CASES 2>&1 | grep "us/step" > gpurun_out/size_sweep_tile.txt || exit 1
echo SWEEP_DONE
