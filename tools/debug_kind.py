import sys, torch
sys.path.insert(0, '.')
from ccvm_amd import engine
from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv
kind, n, b, t = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
q, v, _ = scaled_qv(n, kind)
prob = engine.DeviceProblem(q, v)
p = dict(EXAMPLE_PARAMS[kind])
noise = engine.NoiseSpec(mode="philox", seed=5, row_offset=4096)
if kind == "dl":
    tr = engine.Trajectories(prob, b, "dl", t, dict(p, g=0.05), (0.0, 1.0), noise)
elif kind == "mf":
    tr = engine.Trajectories(prob, b, "mf", t, dict(p, g=0.01), (0.0, 1.0), noise)
else:
    tr = engine.Trajectories(prob, b, "langevin", t, dict(p, use_pump=kind == "pl"), (0.0, 1.0), noise)
tr.advance(t)
torch.cuda.synchronize()
print(kind, n, b, t, "ok", float(list(tr.state.values())[0].abs().max()))
