# Fetch delay of slab clusters that span XCDs (developer tool): the library's default (calibrated per launch) against
# fixed delays (CCVM_AMD_SLAB_DELAY, x 64 cycles).   bash tools/delay_sweep.sh [kind:N:B ...]
CASES=${@:-dl:2000:8 dl:2000:32 dl:1500:16 pl:2000:32 pl:2000:8 langevin:1500:32 pl:2000:64 dl:1200:16 langevin:1500:4 langevin:1200:32}
for c in $CASES; do
  echo -n "default  "; python tools/time_small.py $c 2>&1 | grep us/step
  for d in 16 24 32 40 48 56 72; do echo -n "fixed $d "; CCVM_AMD_SLAB_DELAY=$d python tools/time_small.py $c 2>&1 | grep us/step; done
done
