#!/bin/bash
# Same-box A/B (developer tool): per-step grids that HAVE a proper XCD rectangle, as rectangles (default) against the
# blocked order forced on them (CCVM_AMD_XCD_XC=-w: super-columns of w column blocks).  Writes gpurun_out/ab_rect_vs_blocked.txt.
export CCVM_AMD_KERNEL=tile
{
for ks in 1 2 4; do
export CCVM_AMD_KS=$ks
CASES="dl:1500:1000 dl:2000:1000 dl:2500:1000 dl:1536:1024 dl:2048:1024 langevin:1500:1000 langevin:2000:1000 langevin:2048:2048 langevin:1600:1000 mf:2000:768 dl:1200:1000 dl:1000:2000"
echo "== KS=$ks rectangles (default)"; python3 tools/time_small.py $CASES
for w in 2 4 8; do echo "== KS=$ks blocked order forced, width $w"; CCVM_AMD_XCD_XC=-$w python3 tools/time_small.py $CASES; done
done
} > gpurun_out/ab_rect_vs_blocked.txt 2>&1
