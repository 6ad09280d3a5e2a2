#!/bin/bash
# Same-box A/B of the cluster kernel's half-chunk variant (N mod 128 in 1 .. 64) against the full kernel on the same shapes
# (developer tool; writes gpurun_out/ab_cluster_half.txt).   bash tools/ab_cluster_half.sh
export CCVM_AMD_KERNEL=cluster
CASES="langevin:257:1000 langevin:300:1000 langevin:320:1000 mf:300:1000 dl:300:1000 dl:320:1000 langevin:440:1000 mf:448:1000 dl:448:1000 langevin:576:1000 mf:560:1000 dl:576:1000 langevin:700:1000 dl:704:1000 mf:300:1000:adam langevin:576:1000:adam"
{
echo "== half-chunk variant (default)"
python3 tools/time_small.py $CASES
echo "== CCVM_AMD_CLUSTER_HALF=0 (the full kernel)"
CCVM_AMD_CLUSTER_HALF=0 python3 tools/time_small.py $CASES
echo "== half-chunk variant again"
python3 tools/time_small.py $CASES
} > gpurun_out/ab_cluster_half.txt 2>&1
