#!/bin/bash
# round 6, GPU call 17: soaks on the final code (producers on / off bit for bit over 10^5 steps; every family's long runs without a recovery)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 500 python tools/soak_producers.py > $O/soak_producers.txt 2>&1 || { tail -20 $O/soak_producers.txt; exit 1; }
grep -v amdgpu.ids $O/soak_producers.txt
timeout -k 10 600 python tools/soak.py > $O/soak.txt 2>&1 || { tail -20 $O/soak.txt; exit 1; }
grep -v amdgpu.ids $O/soak.txt
echo CALL17_DONE
