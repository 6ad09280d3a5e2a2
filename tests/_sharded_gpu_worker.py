"""Worker of tests/test_gpu_sharded.py: one rank of a sharded solve with the REAL engine.
    python _sharded_gpu_worker.py <backend> <rank> <world> <port> <kind> <n> <batch> <out.pt>
All ranks share cuda:0 (a 1-GPU box); "gloo" moves the collective's tensors through the host,
"nccl" (world 1 only here: RCCL refuses two ranks on one device) runs it on the device."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    backend, rank, world, port, kind, n, batch, out = sys.argv[1:9]
    rank, world, n, batch = int(rank), int(world), int(n), int(batch)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccvm_amd.sharded import solve_sharded
    from ccvm_amd.solvers import DLSolver, MFSolver, PumpedLangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS, synthetic_instance

    cls = {"dl": DLSolver, "mf": MFSolver, "pl": PumpedLangevinSolver}[kind]
    inst = synthetic_instance(n, seed=11)
    inst.optimal_sol = 1.0
    solver = cls(device="cpu", batch_size=batch)
    solver.parameter_key = {n: dict(EXAMPLE_PARAMS[kind], iterations=20)}
    solver.noise_seed = 0x5EED5EED  # fixed key: the unsharded run of the test uses the same
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    sol = solve_sharded(solver, inst, gather_variables=True)
    if rank == 0:
        torch.save({"objective_values": sol.objective_values.cpu(),
                    "problem_variables": sol.variables["problem_variables"].cpu(),
                    "best": sol.best_objective_value, "shard": sol.shard, "batch": sol.batch_size}, out)
    dist.barrier()
    dist.destroy_process_group()
