#!/bin/bash
# round 6, call 26: where the waves of a workgroup land (HW_ID probe), then the size sweep of the final code again
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/simd_probe.hip -o /tmp/simd_probe > $O/simd_probe_build.log 2>&1 || { echo PROBE_BUILD_FAILED; tail $O/simd_probe_build.log; exit 1; }
timeout -k 10 60 /tmp/simd_probe > $O/simd_probe.txt 2>&1 || { echo PROBE_FAILED; tail -5 $O/simd_probe.txt; exit 1; }
cat $O/simd_probe.txt
timeout -k 10 420 bash tools/size_sweep.sh > $O/size_sweep.log 2>&1 || { echo SWEEP_FAILED; tail -5 $O/size_sweep.log; exit 1; }
cp gpurun_out/size_sweep_auto.txt $O/size_sweep_auto.txt; cp gpurun_out/size_sweep_tile.txt $O/size_sweep_tile.txt
echo CALL26_DONE
