#!/bin/bash
# Same-box A/B (developer tool): the cluster kernel above K = 512 in clusters of 48 rows (three row sets, CCVM_AMD_CLUSTER_SETS=3:
# what ran before round 5) against clusters of 32 rows (two), and the per-step tiles at the same shapes.
# Writes gpurun_out/ab_cluster_sets.txt.
CASES=""
for k in langevin mf dl; do for n in 576 640 704 768; do for b in 400 512 640; do CASES="$CASES $k:$n:$b"; done; done; done
CASES="$CASES langevin:576:768 langevin:640:768 langevin:576:896 langevin:640:800 mf:640:768 dl:640:768 dl:576:768 langevin:768:672 mf:704:700 langevin:640:512:adam mf:768:512:adam"
{
echo "== three row sets (forced)"; CCVM_AMD_KERNEL=cluster CCVM_AMD_CLUSTER_SETS=3 python3 tools/time_small.py $CASES
echo "== two row sets (forced)"; CCVM_AMD_KERNEL=cluster CCVM_AMD_CLUSTER_SETS=2 python3 tools/time_small.py $CASES
echo "== per-step tiles (nocluster, noslab)"; CCVM_AMD_KERNEL=nocluster python3 tools/time_small.py $CASES
echo "== default policy"; python3 tools/time_small.py $CASES
} > gpurun_out/ab_cluster_sets.txt 2>&1
