"""Soak of the row-owner kernel's round-6 shapes (developer tool): long fused-noise runs in odd chunks, each shape against
the shape it replaced -- every word of the final state must be equal, no time-out recovery.
   * 64 < N <= 96: three 32-column waves of eight rows (+ producers) against two 64-column waves, whole chains
   * 128 < N <= 192: three waves side by side x two K halves, TWO row sets per twelve-wave workgroup against one
   * 128 < N <= 224: whole chains over two rows in use against four
   python tools/soak_shapes.py > gpurun_out/r06/soak_shapes.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ADAM = {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False}
VARS = ("CCVM_AMD_PERSIST_CW", "CCVM_AMD_PERSIST_KH", "CCVM_AMD_PERSIST_RU", "CCVM_AMD_PERSIST_PW", "CCVM_AMD_PERSIST_RSW",
        "CCVM_AMD_PERSIST_WIDE", "SOAK_CHUNK")
NARROW = ({"CCVM_AMD_PERSIST_CW": "64", "CCVM_AMD_PERSIST_KH": "1", "CCVM_AMD_PERSIST_PW": "0"}, {})   # old (whole chains), default
NARROW_ADAM = NARROW
TWO_SETS = ({"CCVM_AMD_PERSIST_KH": "2", "CCVM_AMD_PERSIST_RSW": "1"}, {})
TWO_ROWS = ({"CCVM_AMD_PERSIST_KH": "1", "CCVM_AMD_PERSIST_RU": "4"}, {})
# 256 < N <= 320: five waves side by side with part of Q in LDS -- nothing to be bit-identical with (the cluster kernel sums in
# another order): the long run twice, whole and in odd chunks ("before" = one call per 7777 steps, "default" = per 1234)
WIDE = ({"CCVM_AMD_PERSIST_WIDE": "1"}, {"CCVM_AMD_PERSIST_WIDE": "1", "SOAK_CHUNK": "1234"})
CASES = (("dl", 70, 1000, 100000, None, NARROW), ("langevin", 96, 2000, 50000, None, NARROW), ("pl", 80, 1000, 100000, ADAM, NARROW_ADAM),
         ("mf", 65, 777, 100000, ADAM, NARROW_ADAM), ("dl", 144, 1000, 60000, None, TWO_SETS), ("dl", 192, 4000, 20000, None, TWO_SETS),
         ("mf", 176, 2000, 40000, None, TWO_SETS), ("langevin", 160, 2000, 40000, ADAM, TWO_SETS), ("mf", 144, 512, 100000, None, TWO_ROWS), ("dl", 200, 256, 60000, None, TWO_ROWS),
         ("mf", 224, 512, 50000, ADAM, TWO_ROWS), ("mf", 130, 500, 100000, ADAM, TWO_ROWS),
         ("dl", 300, 1000, 40000, None, WIDE), ("langevin", 320, 2000, 30000, None, WIDE), ("pl", 257, 333, 60000, None, WIDE))
for kind, n, b, t, adam, (old, new) in CASES:
    finals, shapes = [], []
    for env in (old, new):
        for v in VARS:
            os.environ.pop(v, None)
        os.environ.update(env)
        shapes.append(bench.describe_launch(kind, b, n) if adam is None else "")
        traj, _, _ = bench.make_trajectories(kind, n, b, t, 0, adam=adam)
        torch.cuda.synchronize()
        t0 = time.time()
        done = 0
        while done < t:
            k = min(int(os.environ.get("SOAK_CHUNK", "7777")), t - done)   # odd chunking on purpose
            traj.advance(k); done += k
        torch.cuda.synchronize()
        dt = time.time() - t0
        assert traj.check() is False and traj.fallbacks == 0
        finals.append(({k_: v.clone() for k_, v in traj.state.items()}, dt))
    same = all(torch.equal(finals[0][0][k_], finals[1][0][k_]) for k_ in finals[0][0])
    finite = all(bool(torch.isfinite(v).all()) for v in finals[1][0].values())
    print(f"{kind} N={n} B={b} T={t}{' adam' if adam else ''}: before {finals[0][1] / t * 1e6:.3f} us/step, default {finals[1][1] / t * 1e6:.3f}; "
          f"final state bit-identical={same} finite={finite}   {shapes[0][:70]} -> {shapes[1][:70]}", flush=True)
    assert same and finite
print("SOAK_OK")
