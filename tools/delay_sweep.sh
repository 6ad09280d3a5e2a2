for c in dl:2000:8 dl:2000:32 dl:1500:16 pl:2000:32 pl:2000:8 langevin:1500:32 pl:2000:64 dl:1200:16; do
  for d in 16 24 32 40 48 56 72 96; do echo -n "delay $d "; CCVM_AMD_SLAB_DELAY=$d python tools/time_small.py $c 2>&1 | grep us/step; done
done
