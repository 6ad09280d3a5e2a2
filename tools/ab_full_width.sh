#!/bin/bash
# Same-box A/B (developer tool): per-step grids whose only XCD rectangle is full width (47 / 63 column blocks: every XCD
# sweeps the whole Q) in the blocked order of super-columns (default since round 5) against the rectangle
# (CCVM_AMD_XCD_XC=-1).  Writes gpurun_out/ab_full_width.txt.
export CCVM_AMD_KERNEL=tile CCVM_AMD_KS=4
CASES="langevin:2000:256 langevin:2000:512 langevin:2000:768 langevin:2000:1000 langevin:1500:256 langevin:1500:512 langevin:1500:1000 dl:2000:256 dl:2000:512 dl:2000:1000 dl:1500:512 dl:1500:1000 mf:2000:512 mf:1500:768"
{
echo "== blocked order (default)"; python3 tools/time_small.py $CASES
echo "== full-width rectangles (CCVM_AMD_XCD_XC=-1)"; CCVM_AMD_XCD_XC=-1 python3 tools/time_small.py $CASES
echo "== blocked order again"; python3 tools/time_small.py $CASES
} > gpurun_out/ab_full_width.txt 2>&1
