"""Soak of the row-owner kernel's noise producer waves (developer tool): long fused-noise runs in odd chunks with the
producers on and off -- every word of the final state must be equal, no time-out recovery, the per-step time holds.
   python tools/soak_producers.py > gpurun_out/r06/soak_producers.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ADAM = {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False}
CASES = (("dl", 20, 1000, 200000, None), ("dl", 20, 100, 200000, None), ("mf", 20, 1000, 100000, None),
         ("langevin", 20, 1000, 100000, ADAM), ("pl", 45, 333, 100000, None), ("dl", 64, 1000, 100000, None),
         ("mf", 64, 2500, 50000, ADAM), ("dl", 100, 1000, 100000, None), ("langevin", 100, 1000, 100000, None),
         ("mf", 128, 700, 50000, ADAM), ("dl", 70, 1000, 100000, None))
for kind, n, b, t, adam in CASES:
    finals = {}
    for pw in ("0", "1"):
        os.environ["CCVM_AMD_PERSIST_PW"] = pw
        if n > 64:
            os.environ["CCVM_AMD_PERSIST_KH"] = "2"
        else:
            os.environ.pop("CCVM_AMD_PERSIST_KH", None)
        traj, _, _ = bench.make_trajectories(kind, n, b, t, 0, adam=adam)
        torch.cuda.synchronize()
        t0 = time.time()
        done = 0
        while done < t:
            k = min(7777, t - done)   # odd chunking on purpose
            traj.advance(k); done += k
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert traj.check() is False and traj.fallbacks == 0
        finals[pw] = ({k_: v.clone() for k_, v in traj.state.items()}, dt)
    same = all(torch.equal(finals["0"][0][k_], finals["1"][0][k_]) for k_ in finals["0"][0])
    finite = all(bool(torch.isfinite(v).all()) for v in finals["1"][0].values())
    print(f"{kind} N={n} B={b} T={t}{' adam' if adam else ''}: producers off {finals['0'][1] / t * 1e6:.3f} us/step, on {finals['1'][1] / t * 1e6:.3f}; "
          f"final state bit-identical={same} finite={finite}", flush=True)
    assert same and finite
print("SOAK_OK")
