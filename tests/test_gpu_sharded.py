"""The multi-GPU path with the REAL engine on the one GPU of the box: 2 and 3 ranks over gloo (all on
cuda:0), and a world-size-1 RCCL ("nccl") group for the device-side collective.  The union of the shards
must equal the unsharded run BIT FOR BIT (noise is keyed on the global row), for uneven shards too."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _unsharded(kind, n, batch):
    from ccvm_amd.solvers import DLSolver, MFSolver, PumpedLangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS, synthetic_instance

    cls = {"dl": DLSolver, "mf": MFSolver, "pl": PumpedLangevinSolver}[kind]
    inst = synthetic_instance(n, seed=11)
    inst.optimal_sol = 1.0
    solver = cls(device="cpu", batch_size=batch)
    solver.parameter_key = {n: dict(EXAMPLE_PARAMS[kind], iterations=20)}
    solver.noise_seed = 0x5EED5EED
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    return solver(instance=inst)


@pytest.mark.parametrize("backend,world,kind,n,batch", [
    ("gloo", 2, "dl", 40, 70),      # persistent kernel, even shards
    ("gloo", 3, "pl", 300, 100),    # tile kernel, uneven shards (34 + 33 + 33): odd row offsets
    ("gloo", 2, "mf", 130, 33),     # uneven shards of an odd batch
    ("nccl", 1, "dl", 64, 50),      # the device-side collective (RCCL)
])
def test_sharded_engine_equals_unsharded(tmp_path, backend, world, kind, n, batch):
    port, out = str(_free_port()), str(tmp_path / "sharded.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_gpu_worker.py"), backend, str(r),
                               str(world), port, kind, str(n), str(batch), out], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
    got = torch.load(out)
    ref = _unsharded(kind, n, batch)
    assert got["batch"] == batch and got["shard"]["world"] == world
    assert torch.equal(got["objective_values"], ref.objective_values.cpu())
    assert torch.equal(got["problem_variables"], ref.variables["problem_variables"].cpu())
    assert got["best"] == ref.best_objective_value
