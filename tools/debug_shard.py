import sys, torch
sys.path.insert(0, '.')
from ccvm_amd import engine
from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv
n, b, t = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
q, v, _ = scaled_qv(n, "dl")
prob = engine.DeviceProblem(q, v)
p = dict(EXAMPLE_PARAMS["dl"], g=0.05)
def run(bb, off, steps):
    tr = engine.Trajectories(prob, bb, "dl", t, p, (0.0, 1.0), engine.NoiseSpec(mode="philox", seed=5, row_offset=off))
    tr.advance(steps)
    return tr.compact("c").cpu(), tr.compact("s").cpu()
for steps in (1, 2, t):
    full = run(b, 0, steps)
    again = run(b, 0, steps)
    parts = [run(b // 2, r * (b // 2), steps) for r in range(2)]
    cat = torch.cat([x[0] for x in parts])
    d = (full[0] - cat).abs()
    print("steps", steps, "repeat-equal", torch.equal(full[0], again[0]), "max diff", float(d.max()), "n diff", int((d > 0).sum()),
          "rows", sorted(set(torch.nonzero(d > 0)[:, 0].tolist()))[:20], "cols", sorted(set(torch.nonzero(d > 0)[:, 1].tolist()))[:12])
