"""Soak of the persistent tile kernel's in-launch hand-over (developer tool): random (solver, Adam variant, N, B, steps,
chunking, row offset) with the family forced; every case runs once as whole launches and once in random chunks (one-step
chunks included: there the kernel boundary publishes the state) and must agree BIT FOR BIT, with a second stream
hammering memory half of the time.   python tools/soak_ptile.py [seconds]
SOAK_MODE=batches: the DEFAULT policy over batches of up to ~4 resident grids instead (slices of the batch, batches cut in
two, the plans of the remainders; the cluster kernel's sizes included)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BATCHES = os.environ.get("SOAK_MODE") == "batches"
if not BATCHES:
    os.environ.setdefault("CCVM_AMD_KERNEL", "ptile")
    os.environ.setdefault("CCVM_AMD_KS", "1")
import torch
import bench

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = random.Random(int(os.environ.get("SOAK_SEED", "4")))
ADAMS = [None, None, {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False},
         {"alpha": 0.01, "beta1": 0.8, "beta2": 1.0, "add_assign": True}]
t_end, cases, steps_total = time.time() + budget, 0, 0
scratch = torch.empty((48 * 1024 * 1024,), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
while time.time() < t_end:
    kind = rng.choice(["dl", "dl", "langevin", "pl", "mf"])
    adam = None if kind == "dl" else rng.choice(ADAMS)
    n = rng.choice([769, 800, 896, 1000, 1000, 1024, 1025, 1300, 1536, 2000, 2048, 3000])
    ncb = (n + 127) // 128
    nrb = rng.randint(1, max(1, 256 // ncb))
    b = max(1, nrb * 32 - rng.choice([0, 0, 1, 7, 24, 31]))
    t = rng.choice([2, 3, 8, 17, 40, 120, 400 if n <= 1100 else 60])
    if BATCHES:
        n = rng.choice([300, 384, 500, 512, 1000, 1000, 1024, 1100, 1200, 1500, 2000])
        fit = 256 // ((n + 127) // 128) * 32 if n > 768 else 8 * (32 // ((n + 63) // 64)) * 32
        b = rng.choice([fit + rng.randint(1, 400), 2 * fit + rng.randint(-40, 300), rng.randint(fit, 4 * fit + 200),
                        3 * fit, 4 * fit + rng.randint(1, 64)])
        t = rng.choice([2, 3, 8, 17, 40])
    chunks, left = [], t
    while left > 0:
        k = min(left, rng.choice([1, 1, 2, 5, 9, 33, t]))
        chunks.append(k); left -= k
    off = rng.choice([0, 0, 3, 64, 1001])
    outs = []
    for plan in ([t], chunks):
        traj, _, _ = bench.make_trajectories(kind, n, b, t, 0, seed=1000 + cases, row_offset=off, adam=adam)
        if rng.random() < 0.5:
            with torch.cuda.stream(side):
                for _ in range(12):
                    scratch.mul_(1.0001)
        for k in plan:
            traj.advance(k)
        traj.check()
        assert traj.fallbacks == 0, (kind, n, b, t)
        outs.append({k: traj.compact(k).clone() for k in traj.state})
        side.synchronize()
    for name in outs[0]:
        assert bool(torch.isfinite(outs[0][name]).all()), (kind, n, b, t, name)
        assert torch.equal(outs[0][name], outs[1][name]), \
            f"DIFFERENCE {kind} adam={adam is not None} N={n} B={b} T={t} chunks={chunks} off={off}: {name}"
    cases += 1; steps_total += t
    if cases % 20 == 0:
        print(f"{cases} cases, {steps_total} steps, last: {kind} adam={adam is not None} N={n} B={b} T={t} "
              f"({len(chunks)} chunks)", flush=True)
print(f"SOAK OK: {cases} cases, {steps_total} steps, no difference", flush=True)
