#!/bin/bash
# round 6, call 42: the SIMD probe again with ten-wave workgroups and the waves that share a SIMD
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/simd_probe.hip -o /tmp/simd_probe > $O/simd_probe_build.log 2>&1 || { echo PROBE_BUILD_FAILED; tail $O/simd_probe_build.log; exit 1; }
timeout -k 10 60 /tmp/simd_probe > $O/simd_probe.txt 2>&1 || { echo PROBE_FAILED; tail -5 $O/simd_probe.txt; exit 1; }
grep -A3 "640 threads" $O/simd_probe.txt | cut -c1-400
echo CALL42_DONE
