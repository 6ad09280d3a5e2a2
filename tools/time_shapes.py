"""Per-step time of a list of shapes under one or more launch policies (developer tool; GPU box).
   python tools/time_shapes.py dl:1000:2000 pl:2000:1000 [--env CCVM_AMD_KERNEL=noptile]
Each shape is timed in this process under the current environment and, per `--env K=V`, again with that variable set
(the C library reads its tuning per call).  Best of 3 runs of about 60 ms each, as tools/regime_map.py does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(kind, n, b):
    import torch
    import bench
    traj, _, _ = bench.make_trajectories(kind, n, b, 1 << 20, 0)
    kernel = bench.describe_launch(kind, b, n)
    traj.advance(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    traj.advance(64)
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 64
    steps = int(min(4096, max(64, 0.06 / est)))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        traj.advance(steps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    traj.check()
    return best / steps * 1e6, kernel


def main():
    args = sys.argv[1:]
    envs = [None]
    shapes = []
    while args:
        a = args.pop(0)
        if a == "--env":
            envs.append(args.pop(0))
        else:
            kind, n, b = a.split(":")
            shapes.append((kind, int(n), int(b)))
    for kind, n, b in shapes:
        for env in envs:
            if env:
                k, v = env.split("=", 1)
                old = os.environ.get(k)
                os.environ[k] = v
            us, kernel = one(kind, n, b)
            flop = (4 if kind == "dl" else 2) * n * n * b
            print(f"{kind}:{n}:{b:<6d} {env or 'default':28s} {us:9.2f} us/step  {flop / us / 1e6 / 157.3:5.3f} of peak  {kernel[:72]}",
                  flush=True)
            if env:
                if old is None:
                    del os.environ[k]
                else:
                    os.environ[k] = old


if __name__ == "__main__":
    main()
