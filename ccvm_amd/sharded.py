"""Batch sharding over the GPUs of one node.

Trajectories never interact, so a batch of B rows is split into contiguous row ranges,
one per rank (one process per GPU, ``torch.distributed``; backend "nccl" is RCCL over
xGMI).  Nothing is exchanged during the T steps.  Each rank runs its rows with the SAME
noise key and its own ``row_offset``, so the union of the shards is bit-identical to the
unsharded run -- in the fused mode because the generator is keyed on the global row, in replay
mode because every rank draws the unsharded run's (N, B) block from an identically seeded torch
generator and keeps its own columns (the ranks must share the seed: ``torch.manual_seed`` with
the same value everywhere, as for any torch.distributed program).  The only collective is one
all-gather of the per-row objective values (B floats in total) after the loop -- from the device
copy ccvm_finalize left on the GPU when the backend is RCCL -- and the global success
statistics are counted on the device from the gathered vector; the reference has no counterpart
(it is single-process).

Evolution sampling under sharding: every rank writes the best row OF ITS SHARD to its own file,
``<evolution_file>.rank<r>`` (rank r; the name is returned in ``Solution.evolution_file``).
"""
import copy

import torch
import torch.distributed as dist

from .solution import Solution


def shard_bounds(batch, world, rank):
    """Contiguous [lo, hi) of ``rank``; the first ``batch % world`` ranks get one extra row."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(int(batch), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _all_gather_rows(local, counts, group):
    """All-gather of 1-D/2-D tensors whose leading sizes differ (pad to the maximum)."""
    world = len(counts)
    width = max(counts)
    pad_shape = (width,) + tuple(local.shape[1:])
    padded = torch.zeros(pad_shape, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([part[:n] for part, n in zip(parts, counts)])


def broadcast_seed(seed, group=None, device="cpu"):
    """Rank 0's Philox key for every rank."""
    t = torch.tensor([int(seed)], dtype=torch.int64, device=device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.item())


def solve_sharded(solver, instance, group=None, gather_variables=False, local_solve=None, **call_kwargs):
    """Solve ``instance`` with ``solver.batch_size`` rows split over the ranks of ``group``.

    Every rank returns the SAME global ``Solution`` (objective values of all rows, global
    success statistics).  ``solve_time`` keeps the reference's definition -- the time of the LOOP
    only, per row (dl_solver.py:851, 933: the timer brackets ``_solve`` and is divided by the batch) --
    for the job as a whole: every rank's local ``Solution.solve_time`` already is its loop's time per
    local row, the ranks run side by side, so the job's loop took ``max_r(solve_time_r * rows_r)`` and
    the global figure is that over the GLOBAL batch; ``pp_time`` likewise.  Priming, the finalize, host
    copies and the gather are outside it, as they are outside the reference's timer.  ``variables``
    hold this rank's rows unless ``gather_variables``.

    ``local_solve(solver, instance, **call_kwargs) -> Solution`` defaults to calling the solver
    (the HIP engine); tests inject an oracle-backed stand-in to exercise the sharding logic on
    CPU with gloo.
    """
    if not dist.is_initialized():
        raise RuntimeError("solve_sharded needs an initialised torch.distributed process group")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    batch = int(solver.batch_size)
    if batch < world:
        raise ValueError(f"batch_size {batch} is smaller than the number of ranks {world}")
    counts = [hi - lo for lo, hi in (shard_bounds(batch, world, r) for r in range(world))]
    lo, hi = shard_bounds(batch, world, rank)
    backend = dist.get_backend(group)
    comm_device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    from . import engine

    local = copy.copy(solver)  # shallow: shares parameter_key, rebinds the per-rank fields
    local.batch_size = hi - lo
    local.row_offset = solver.row_offset + lo
    mode = engine.effective_noise_mode(solver.noise_mode)
    if mode == "replay":
        if solver.row_offset != 0:
            raise ValueError("replay noise under sharding needs row_offset 0 on the unsharded solver")
        local.replay_global_batch = batch  # draw the unsharded block, keep columns [lo, hi)
    elif solver.noise_seed is None:
        local.noise_seed = broadcast_seed(engine.draw_seed() if rank == 0 else 0, group, comm_device)
    if call_kwargs.get("evolution_step_size"):
        base = call_kwargs.get("evolution_file") or f"./{instance.name}_evolution.txt"
        call_kwargs = dict(call_kwargs, evolution_file=f"{base}.rank{rank}")

    sol = (local_solve or (lambda s, inst, **kw: s(instance=inst, **kw)))(local, instance, **call_kwargs)
    # seconds this rank's loop / post-processor took in all (the local Solution reports them per local row)
    wall = torch.tensor([sol.solve_time * (hi - lo), sol.pp_time * (hi - lo)], dtype=torch.float64,
                        device=comm_device)
    dist.all_reduce(wall, op=dist.ReduceOp.MAX, group=group)

    # gather from where the values already are: the device copy of ccvm_finalize under RCCL
    local_obj = sol.objective_values
    dev_obj = getattr(sol, "device_objective_values", None)
    if backend == "nccl" and dev_obj is not None:
        local_obj = dev_obj
    obj = _all_gather_rows(local_obj.to(comm_device), counts, group)
    stats = {}
    if obj.is_cuda and sol.optimal_value is not None:
        # global success statistics on the device (ccvm_objective_stats), 40 bytes to the host
        from .solution import fractions_from_counts

        best, within, rows, _ = engine.read_stats(engine.objective_stats(obj, sol.optimal_value))
        stats = {"best_objective_value": best, "solution_performance": fractions_from_counts(within, rows)}
    variables = dict(sol.variables)
    if gather_variables:
        variables = {
            k: _all_gather_rows(v.to(comm_device), counts, group).to(v.device) if torch.is_tensor(v) else v
            for k, v in sol.variables.items()
        }
    out = Solution(
        problem_size=sol.problem_size,
        batch_size=batch,
        instance_name=sol.instance_name,
        iterations=sol.iterations,
        objective_values=obj.to(sol.objective_values.device),
        solve_time=float(wall[0].item()) / batch,
        pp_time=float(wall[1].item()) / batch,
        optimal_value=sol.optimal_value,
        best_value=sol.best_value,
        num_frac_values=sol.num_frac_values,
        solution_vector=sol.solution_vector,
        variables=variables,
        device=sol.device,
        solution_performance=stats.get("solution_performance"),
        best_objective_value=stats.get("best_objective_value"),
    )
    out.device_objective_values = obj if obj.is_cuda else None
    out.evolution_file = sol.evolution_file
    out.shard = {"rank": rank, "world": world, "rows": (lo, hi)}
    return out
