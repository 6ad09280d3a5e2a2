"""Headline benchmark: SDE row-steps/s of the CCVM inner loop on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload W] [--post adam]

A "step" is one fused Euler-Maruyama update of every batch row.  Default workload (BASELINE.json
`metric`): DL-CCVM, N = 1000 synthetic dense BoxQP, batch 1000 PER GPU (weak scaling: the 8-GPU run
is BASELINE config 4, batch 8000 sharded 1000/GPU), fp32, fused counter-based noise.
`--workload pl_n2000_b512 --post adam` is BASELINE config 5 (pumped Langevin N = 2000, 512 rows per
GPU, on-device Adam post-processor after the loop).

Ranks are independent (batch rows never interact); the only collective is one all-gather of the
final objective values (RCCL), outside the timed region like the reference's own timer
(dl_solver.py:851/933 brackets the loop only).

Launching: `--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a
launcher: it starts N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set)
before anything touches the GPU, waits for them and exits with the worst return code.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the environment is already
there and the process is a rank.  CCVM_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box): every rank uses
cuda:0 and the collectives run over gloo on host copies.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, spec
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (solver kind, N, batch per GPU)
    "dl_n1000_b1000": ("dl", 1000, 1000),
    "dl_n100_b1000": ("dl", 100, 1000),
    "dl_n500_b1000": ("dl", 500, 1000),      # not a BASELINE configuration: the DL solver at config 3's size
    "mf_n500_b1000": ("mf", 500, 1000),
    "langevin_n500_b1000": ("langevin", 500, 1000),
    "pl_n2000_b512": ("pl", 2000, 512),
    # the low end of the column-cluster kernel's range (K = 320: its half-chunk variant, ccvm_cluster.h)
    "langevin_n300_b1000": ("langevin", 300, 1000),
    "dl_n300_b1000": ("dl", 300, 1000),
    # K = 640 in clusters of 32 rows (two row sets where they fit the chip: ccvm_abi.hip, cluster_sets)
    "langevin_n640_b512": ("langevin", 640, 512),
    "dl_n640_b512": ("dl", 640, 512),
    "langevin_n1000_b1000": ("langevin", 1000, 1000),  # the one-stream solvers at the headline's size
    "mf_n1000_b1000": ("mf", 1000, 1000),
    # small batches (the reference runs any batch_size through the same einsum, dl_solver.py:145-153): the
    # column-slab kernel, Q resident in registers chip-wide (ccvm_amd/csrc/ccvm_slab.h)
    "dl_n1000_b1": ("dl", 1000, 1),
    "dl_n1000_b8": ("dl", 1000, 8),
    "dl_n1000_b32": ("dl", 1000, 32),
    "langevin_n1000_b32": ("langevin", 1000, 32),
    "mf_n500_b32": ("mf", 500, 32),
    "pl_n2000_b32": ("pl", 2000, 32),
    # mid-size batches: the per-step tile kernel with 32 x 32 split-K tiles (a quarter of the chip or less with wider ones)
    "dl_n1000_b256": ("dl", 1000, 256),
    "langevin_n1000_b256": ("langevin", 1000, 256),
    # the per-GPU shapes of configs 4 / 5 under strong scaling on 4 and 2 GPUs (--global-batch 8000 / 4096): batches of
    # several rounds, run as slices of the batch on the persistent tile kernel (ccvm_abi.hip: plan_ptile)
    "dl_n1000_b2000": ("dl", 1000, 2000),
    "dl_n1000_b4000": ("dl", 1000, 4000),
    "pl_n2000_b1024": ("pl", 2000, 1024),
    "pl_n2000_b2048": ("pl", 2000, 2048),
}
SOLVER_ID = {"dl": 0, "mf": 1, "langevin": 2, "pl": 2}
SATURATION = {"dl": 1.0, "mf": 20.0, "langevin": 0.5, "pl": 0.5}  # the example scripts' S (workloads.py)


# --------------------------------------------------------------------------- #
# launcher (no GPU call may happen in this process before the ranks are started)
# --------------------------------------------------------------------------- #
#: Multi-process GPU work on this pool's hosts: the kernel driver only supports dmabuf IPC, and with the ROCr default
#: (legacy IPC handles) RCCL's intra-node transport set-up -- hipIpcGetMemHandle of the peers' buffers -- fails with
#: "invalid argument".  The image exports the variable already; the launcher and every rank keep it (setdefault: an
#: explicit value in the caller's environment wins).  tests/test_gpu_sharded.py sets the same for its ranks.
IPC_ENV = ("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def shard_rows(total_rows, world, rank):
    """(first row, rows) of rank `rank` when `total_rows` rows are split over `world` ranks: total // world each, the
    first total % world ranks one more -- every rank gets at least one row whenever total >= world (ceil-sized
    shards left trailing ranks EMPTY, e.g. 9 rows on 8 ranks: ADVICE r3)."""
    base, extra = divmod(int(total_rows), int(world))
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def visible_gpu_count():
    """GPUs this process may use, counted WITHOUT the HIP runtime (the launcher must not initialise the GPU before
    it starts its ranks: torch.cuda.device_count() can fall back to hipGetDeviceCount): the *_VISIBLE_DEVICES
    lists, else the KFD topology in sysfs (nodes with SIMDs are GPUs).  None when neither says anything."""
    counts = []
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None:
            counts.append(len([x for x in val.split(",") if x.strip() != ""]))
    if counts:
        return min(counts)
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(nodes):
            with open(os.path.join(nodes, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        return n
    except (OSError, ValueError):
        return None


def launch_ranks(n, argv, script=None, timeout=None):
    """Start n rank processes of `script` (default: this file) with the torch.distributed environment, wait for
    them, and return the worst return code.  A rank that dies takes the others down with it (they would wait for
    it in a collective until the time-out) and the launcher prints ONE diagnostic JSON line instead of leaving the
    caller without any."""
    import socket

    share = os.environ.get("CCVM_BENCH_SHARE_GPU") in ("1", "try-rccl")
    visible = visible_gpu_count()
    if visible is not None and visible < n and not share:
        raise SystemExit(f"bench.py --gpus {n}: only {visible} GPU(s) visible "
                         "(CCVM_BENCH_SHARE_GPU=1 rehearses the N-rank path on one GPU over gloo)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault(*IPC_ENV)
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__), *argv], env=env))
    deadline = time.time() + float(timeout or os.environ.get("CCVM_BENCH_LAUNCH_TIMEOUT", "1500"))
    codes = [None] * n
    failed = None  # (rank, return code) of the first rank that did not exit cleanly
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = (r, codes[r])
        timed_out = time.time() > deadline
        if failed is not None or timed_out:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.kill()  # exactly the ranks this launcher started
                    p.wait()
                    codes[r] = 124 if timed_out and failed is None else 143
            break
        time.sleep(0.05)
    worst = max(abs(c) for c in codes)
    if worst != 0:
        why = (f"rank {failed[0]} exited with return code {failed[1]}" if failed is not None
               else "time-out: the ranks were killed")
        print(json.dumps({"error": f"bench.py --gpus {n}: {why}", "return_codes": codes, "n_gpus": n}), flush=True)
    return worst


# --------------------------------------------------------------------------- #
# collectives of a multi-rank run
# --------------------------------------------------------------------------- #
def rccl_group(dev, world, timeout_s):
    """An RCCL ("nccl") process group over all ranks, PROVEN by a one-element all-reduce on `dev`: communicator
    creation is lazy, so the transport set-up (xGMI / dmabuf IPC between the ranks' devices) only happens -- and only
    fails -- inside the first collective.  Raises whatever RCCL raises."""
    import datetime

    import torch.distributed as dist

    # a collective that cannot complete raises in THIS thread after the group's time-out instead of leaving the
    # watchdog to abort the process (an explicit setting in the caller's environment wins)
    os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
    grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s), device_id=dev)
    probe = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(probe, group=grp)
    torch.cuda.synchronize(dev)
    if int(probe.item()) != world:
        raise RuntimeError(f"RCCL warm-up all-reduce returned {probe.item()} on {world} ranks")
    return grp


def init_collectives(rank, world, dev, share, probe=None, timeout_s=None):
    """The process groups of a multi-rank run, decided so that the line is never lost to a transport problem.

    The DEFAULT group is gloo (TCP on 127.0.0.1, the rendez-vous the launcher / torch.distributed.run set up): it
    carries the barriers around the timed region and the per-rank clocks (host scalars), and it is the control plane on
    which the ranks AGREE how the one data collective -- the all-gather of the objective values after the loop --
    travels: every rank tries to set up RCCL (`probe`, default rccl_group: group creation + a warm-up all-reduce on
    its device) inside try / except, the outcomes are min-reduced over gloo, and unless EVERY rank succeeded every rank
    gathers host copies over gloo instead, in the same processes -- no re-exec, no relaunch, the measured steps are
    the same either way (no collective sits in the timed region) -- and the line says so
    (`"collective": "gloo-fallback: <first error>"`).  RCCL with more than one rank has never run on this pipeline's
    boxes before the driver's round-end run: its first failure must not cost the scaling curve.

    Returns {"group": data group or None (= default), "device": where the gathered tensors live, "collective": label}."""
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    if share:  # rehearsal on one GPU: RCCL refuses two ranks on one device
        return {"group": None, "device": cpu, "collective": "gloo (rehearsal: the ranks share one GPU)"}
    timeout_s = float(timeout_s or os.environ.get("CCVM_BENCH_RCCL_TIMEOUT", "120"))
    grp, err = None, ""
    try:
        grp = (probe or rccl_group)(dev, world, timeout_s)
    except Exception as exc:  # noqa: BLE001 -- whatever the transport set-up raises
        err = f"{type(exc).__name__}: {exc}".replace("\n", " ")[:300]
    ok = torch.tensor([0 if err else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return {"group": grp, "device": dev, "collective": "RCCL"}
    errs = [None] * world
    dist.all_gather_object(errs, err)
    first = next((f"rank {r}: {e}" for r, e in enumerate(errs) if e), "unknown")
    return {"group": None, "device": cpu, "collective": f"gloo-fallback: {first}"}


def timed_steps(traj, warmup, steps, dev, barrier, any_rank=lambda flag: flag):
    """W untimed warm-up steps, then K steps of `traj` under the contract's clock -- barrier + device synchronisation on
    both sides, HIP events on the launch stream around the same region -- repeated when the steps turn out INVALID: a
    persistent kernel whose workgroups could not all become resident gives up a bounded wait and sets the run's status
    word; its steps are garbage, and `traj.check` puts the trajectories back where the recovery snapshot was taken and
    moves the run to the per-step kernel.  The snapshot is taken ONCE, in front of the warm-up steps (two
    device-to-device copies: right in front of the timed region they would push Q and the state out of the L2s), the
    status word is read ONCE, right after the clock stops (4 bytes; inside the region it would cost ~25 us of a 0.65 ms
    run, between warm-up and timed steps it would leave the chip idle for as long), and a run that was recovered -- the
    word is set by warm-up and timed steps alike and never cleared by a kernel -- starts over from the snapshot on the
    path that then runs, so `value` never describes discarded work.  `any_rank(flag)`: True when the flag is set
    on ANY rank: the ranks start over together (the barriers must pair up).
    Returns (wall seconds, stream ms, attempts)."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()  # torch creates the HIP event at its first record (15-20 us): the measuring apparatus is set up
    ev1.record()  # BEFORE the timed region (tools/sync_probe.py: 36.3 -> 35.5 us per step on a 20-step run)
    traj.arm(force=True)  # (forced: another rank may ask this one to start over although its own steps were valid)
    for attempt in (1, 2, 3):
        traj.advance(warmup)  # (not verified by itself: a status word set here is still set behind the timed steps)
        torch.cuda.synchronize(dev)
        barrier()     # every rank starts its K steps together ...
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record()
        traj.advance(steps)
        ev1.record()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0  # ... and stops ITS OWN clock when its own K steps are done: no collective
        #                                     inside the timed region (a 50-150 us barrier would read as a 7-20 % "scaling
        #                                     loss" on the driver's 0.7 ms); the job's time is the MAX over ranks
        recovered = traj.check(rerun=False, hold=True)  # True: back at the snapshot, on the per-step kernel from here on
        if not any_rank(recovered):
            traj.check()  # (drops the snapshot)
            return elapsed, ev0.elapsed_time(ev1), attempt
        if not recovered:
            traj.rollback()  # another rank starts over: this one does with it
    raise SystemExit("bench.py: the run was invalid three times in a row")


# --------------------------------------------------------------------------- #
def profiled_counters(workload):
    """Counters of the dominant kernel from the COMMITTED rocprofv3 --pmc passes of this same
    command (profiles/r*_<workload>_pmc.json, written by tools/pmc_summary.py).  They are NOT
    measured by this run -- PMC collection needs separate profiler passes -- and are reported in a
    sub-object that says so.  gfx950 correction: FETCH_SIZE counts half of a wide coalesced read, so
    HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md, HBM)."""
    import glob

    stem = "bench" if workload == "dl_n1000_b1000" else workload
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{stem}_pmc.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        doc = json.load(fh)
    c = doc["counters"]
    out = {"source": os.path.relpath(files[-1], ROOT), "measured_by_this_run": False,
           "kernel": doc.get("kernel"), "command": doc.get("command")}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # per time step: a persistent kernel's profiled dispatches run another number of steps than this run's
        out["traffic_bytes_per_step"] = 1024.0 * (2.0 * c["FETCH_SIZE"]["mean_per_dispatch"]
                                                  + c["WRITE_SIZE"]["mean_per_dispatch"]) / doc.get("steps_per_dispatch", 1.0)
        out["dispatches"] = c["FETCH_SIZE"]["dispatches"]
    m = re.search(r"persist_kernel<[^>]*> = .*B=(\d+)", doc.get("kernel") or "")
    if "SQ_INSTS_VALU" in c and m:
        # row-owner kernel: 256-thread workgroups of four waves; rows per workgroup from the kernel's own grid is not in
        # the summary, but SQ_VALU_MFMA_BUSY_CYCLES / 8 cycles = MFMAs issued = waves x steps x 16 NCH
        pk = re.search(r"persist_kernel<\d, \w+, \d+, \d+, (\d+), \d+(?:, (\d+))?>", doc["kernel"])
        nch, kh = int(pk.group(1)), int(pk.group(2) or 1)   # K split: a wave runs 16 NCH / KH MFMAs per step
        wave_steps = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / 8.0 / (16 * nch / kh)
        out["valu_per_wave_step"] = c["SQ_INSTS_VALU"]["mean_per_dispatch"] / wave_steps
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        # MFMA_BUSY is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines (its
        # per-engine value is the kernel's length in cycles)
        out["mfma_busy_frac"] = (c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / 1024.0) / \
                                (c["SQ_BUSY_CYCLES"]["mean_per_dispatch"] / 32.0)
    return out


def workload_params(kind):
    from ccvm_amd.workloads import EXAMPLE_PARAMS

    p = dict(EXAMPLE_PARAMS[kind])
    if kind == "dl":
        p["g"] = 0.05
    elif kind == "mf":
        p["g"] = 0.01
    else:
        p["use_pump"] = kind == "pl"
    return p


def make_trajectories(kind, n, b, total_steps, rank, seed=1, row_offset=None, adam=None):
    from ccvm_amd import engine
    from ccvm_amd.workloads import scaled_qv

    q, v, _ = scaled_qv(n, kind)
    prob = engine.DeviceProblem(q, v)
    noise = engine.NoiseSpec(mode="fused", seed=seed, row_offset=rank * b if row_offset is None else row_offset)
    ekind = {"dl": "dl", "mf": "mf"}.get(kind, "langevin")
    return engine.Trajectories(prob, b, ekind, total_steps, workload_params(kind), (0.0, 1.0), noise, adam=adam), q, v


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            return next(line.split(":", 1)[1].strip() for line in fh if line.startswith("model name"))
    except (OSError, StopIteration):
        return "unknown CPU"


def cpu_baseline(kind, n, b, total_steps, budget_s=12.0, threads=None):
    """The oracle (torch CPU restatement of the reference loop, bit-identical to it on this torch
    build) timed on the host cores on a bounded sample of the same workload.  Threads: a 1-GPU box
    exposes a 16-core CPU share of the host; profiles/r02_cpu_thread_sweep.md holds the 1..64-thread
    sweep behind the default (more threads than the share only oversubscribe)."""
    from ccvm_amd.workloads import scaled_qv
    from oracle import ccvm_oracle as oracle

    visible = len(os.sched_getaffinity(0))
    torch.set_num_threads(max(1, threads or min(16, visible)))
    q, v, _ = scaled_qv(n, kind)
    p = workload_params(kind)
    torch.manual_seed(1)
    bounds = (0.0, 1.0)
    if kind == "dl":
        c = torch.zeros((b, n)); s = torch.zeros((b, n))
        run = lambda step0, k: oracle.dl_loop(q, v, b, total_steps, p["pump"], p["dt"], p["noise_ratio"],
                                              p["feedback_scale"], p["g"], bounds, True, None, step0, k, c, s)
    else:
        # the other loops run whole trajectories: time short runs of the same per-step work
        def run(step0, k):
            if kind == "mf":
                oracle.mf_loop(q, v, b, k, p["pump"], p["dt"], p["j"], p["feedback_scale"], p["S"], p["g"], bounds)
            elif kind == "langevin":
                oracle.langevin_loop(q, v, b, k, p["dt"], p["sigma"], p["feedback_scale"], p["S"], bounds)
            else:
                oracle.pl_loop(q, v, b, k, p["pump"], p["dt"], p["sigma"], p["feedback_scale"], p["S"], bounds)
    run(0, 2)  # warm-up (first-call overheads)
    done, t0 = 2, time.time()
    while time.time() - t0 < budget_s and done + 5 <= max(total_steps, 12):
        run(done, 5)
        done += 5
    dt = time.time() - t0
    steps = done - 2
    return {
        "value": b * steps / dt, "unit": "row-steps/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"{steps} steps of the same workload (N={n}, batch={b}) " if kind == "dl" else
                   f"{steps // 5} whole 5-step runs (N={n}, batch={b}: the same per-step work, each run with its own "
                   f"set-up and a 5-step schedule, not steps {total_steps - steps}.. of the benchmarked run) ")
                  + f"with the torch-CPU oracle "
                  f"(bit-identical to the reference's CPU path), {dt / steps * 1e3:.1f} ms/step, "
                  f"{torch.get_num_threads()} torch threads on {cpu_model()} ({visible} cores visible to this "
                  f"process), torch {torch.__version__}",
    }


def tts99_leg():
    """Secondary metric of BASELINE.json: TTS @ 99 % success = per-row solve time x R99
    (ccvmplotlib/utils/sampleTTSmetric.py:144-153).  Only defined where the optimum is known, so it is
    measured on the reference's shipped tuning instance tuningH020-100-0 (arrays: tests/golden fixture)
    with the shipped example configuration (examples/ccvm_boxqp_dl.py:12-24: B=1000, 1500 iterations),
    through the public solver API with the fused generator."""
    import numpy as np

    from ccvm_amd.problem_classes.boxqp import ProblemInstance
    from ccvm_amd.solvers import DLSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS

    gdir = os.path.join(ROOT, "tests", "golden")
    arrays = np.load(os.path.join(gdir, "tuningH020.npz"))
    with open(os.path.join(gdir, "tuningH020.json")) as fh:
        meta = json.load(fh)["instance"]
    inst = ProblemInstance.from_arrays(arrays["q_matrix"], arrays["v_vector"], device="cuda", name=meta["name"],
                                       optimal_sol=meta["optimal_sol"], best_sol=meta["best_sol"])
    solver = DLSolver(device="cuda", batch_size=1000)
    solver.parameter_key = {20: dict(EXAMPLE_PARAMS["dl"], iterations=1500)}
    inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
    torch.manual_seed(1234)
    solver(instance=inst)  # first call: one-time initialisation
    sol = solver(instance=inst)
    p = sol.solution_performance["optimal"]
    return {
        "value": sol.tts99(), "unit": "s", "instance": meta["name"], "batch": 1000, "iterations": 1500,
        "p_optimal": p, "solve_time_per_row_s": sol.solve_time, "best_objective_value": sol.best_objective_value,
        "optimal_value": meta["optimal_sol"],
    }


def family_roof(launch, kind, n, b, step_us, wall_step_us):
    """Extra roofline fields for the kernel families whose binding roof is not the fp32 MFMA peak.

    slab_kernel (small batches, Q in registers, ccvm_slab.h): no byte of Q moves, the contraction of a member is
    R C K MACs (a few hundred cycles); a step is one hand-off of the GEMM input between the cluster's workgroups
    (own publish -> every peer's packets read: MI355X_MICROARCH.md price list, "handoff-1to1" 0.8-1.1 us idle /
    "allgather" 2.4-2.9 us for 256 CUs) plus the member's matrix time.  bound "latency": peak = steps/s of that
    floor, achieved = steps/s measured."""
    import re

    pm = re.search(r"persist_kernel<(\d), (true|false), (\d+), (\d+), (\d+), (\d+), (\d+)> grid (\d+) x (?:256|512)", launch)
    if pm:
        # Row-owner persistent kernel (N <= 256): a SIMD issues the instructions of its waves' steps one after the other
        # (f32 MFMA and VALU do not overlap on a SIMD: tools/coissue.hip), so the roof is the ISSUE time of a step on the
        # fullest SIMD: per wave 16 NCH / KH v_mfma_f32_4x4x1 x 8 cycles + the other vector instructions x 4 cycles,
        # times the waves that SIMD holds (4 per workgroup over 1024 SIMDs: one, or two with the K split at B = 1000).
        # The instruction count comes from the committed SQ pass of this workload; without one no roof is claimed.
        nch, kh, grid = int(pm.group(5)), int(pm.group(7)), int(pm.group(8))
        prof = profiled_counters(f"{kind}_n{n}_b{b}")
        if not prof or "valu_per_wave_step" not in prof:
            return {}
        waves_per_simd = -(-(8 if int(pm.group(4)) * kh > 4 else 4) * grid // 1024)
        mfma = 16 * nch / kh * waves_per_simd
        issue = 8.0 * mfma + 4.0 * (prof["valu_per_wave_step"] * waves_per_simd - mfma)
        return {
            "bound": "issue", "achieved": 1e6 / step_us, "peak": 2400.0e6 / issue, "unit": "steps/s",
            "frac": issue / 2400.0 / step_us, "frac_wall": issue / 2400.0 / wall_step_us,
            "issue_cycles_per_step": {"mfma": 8.0 * mfma, "other_valu": 4.0 * (prof["valu_per_wave_step"] * waves_per_simd - mfma),
                                      "waves_per_simd": waves_per_simd,
                                      "source": prof["source"] + " (SQ_INSTS_VALU per wave and step, MFMAs included)",
                                      "clock_MHz": 2400},
            "mfma_frac": 2.0 * (2 if kind == "dl" else 1) * n * n * b / (step_us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        }
    m = re.search(r"slab_kernel<\d, (\d+), (\d+), \w+>.*\((\d+) clusters of (\d+) workgroups x (\d+) columns, (\d+) rows each, K = (\d+)(, each over \d+ XCDs)?",
                  launch)
    if not m:
        return {}
    cgrp, nq, clusters, members, cols, rows, k, spread = m.groups()
    planes = 2 if kind == "dl" else 1
    mfma_us = planes * (int(rows) // 4) * int(nq) * 8 / 2400.0  # 4x4x1 MFMAs of a wave x 8 cycles at 2.4 GHz
    handoff_us = 2.4 if spread else 0.8                         # across the fabric / hand-off inside an XCD
    floor_us = handoff_us + mfma_us
    return {
        "bound": "latency", "achieved": 1e6 / step_us, "peak": 1e6 / floor_us, "unit": "steps/s",
        "frac": floor_us / step_us, "frac_wall": floor_us / wall_step_us,
        "latency_floor_us": {"handoff": handoff_us, "member_mfma": mfma_us,
                             "source": "MI355X_MICROARCH.md price list: handoff-1to1 0.8 us (one XCD, idle) / "
                                       "allgather 2.4 us (256 CUs, 8 KB); v_mfma_f32_4x4x1 8 cycles"},
        "mfma_frac": 2.0 * planes * n * n * b / (step_us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "q_bytes_moved_per_step": 0,
        "exchange_bytes_per_member_per_step": planes * int(rows) * int(k) * 8,
    }


def describe_launch(kind, b, n):
    import ctypes

    from ccvm_amd import _lib

    buf = ctypes.create_string_buffer(1024)
    _lib.check(_lib.load().ccvm_describe_launch(SOLVER_ID[kind], b, n, 0, 0, buf, 1024), "ccvm_describe_launch")
    return buf.value.decode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="untimed: run the step kernel on scratch trajectories this long before the warm-up "
                         "steps, so the GPU has left its idle clocks (the DVFS ramp takes ~15 ms: a 300-step "
                         "burst from idle measures the ramp, 42 us/step, not the 36 us/step a real 1500-15000 "
                         "step solve runs at)")
    ap.add_argument("--workload", default="dl_n1000_b1000", choices=sorted(WORKLOADS))
    ap.add_argument("--post", default=None, choices=["adam", "grad-descent"],
                    help="run this on-device post-processor on the final variables (after the timed loop, like "
                         "the reference's pp_time) before they are scored: BASELINE config 5 = "
                         "--workload pl_n2000_b512 --post adam")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="STRONG scaling: this many rows in total, split evenly over the ranks (rank r takes rows "
                         "[r * ceil(G / N), ...)); BASELINE config 5 at every N = --workload pl_n2000_b512 --post adam "
                         "--global-batch 4096, config 4 = --global-batch 8000.  Default: the workload's batch PER GPU "
                         "(weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=None, help="torch threads of the cpu_baseline leg")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # this process never touches the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    # CCVM_BENCH_SHARE_GPU (rehearsals on a 1-GPU box): "1" = every rank on cuda:0, collectives over gloo; "try-rccl" = every
    # rank on cuda:0 and RCCL attempted all the same -- it refuses two ranks on one device, which exercises the REAL
    # fall-back (tests/test_gpu_sharded.py)
    share_mode = os.environ.get("CCVM_BENCH_SHARE_GPU", "")
    share = share_mode == "1"
    if share_mode in ("1", "try-rccl"):
        local = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll = {"group": None, "device": dev, "collective": "single rank"}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault(*IPC_ENV)  # before any process group exists (under torch.distributed.run too)
        coll = init_collectives(rank, world, dev, share)

    kind, n, b = WORKLOADS[args.workload]
    row0 = rank * b
    if args.global_batch is not None:
        if args.global_batch < world:
            raise SystemExit(f"--global-batch {args.global_batch} < {world} ranks")
        row0, b = shard_rows(args.global_batch, world, rank)
    global_rows = args.global_batch if args.global_batch is not None else b * world
    total = args.warmup + args.steps
    traj, q, v = make_trajectories(kind, n, b, total, rank, row_offset=row0)

    def barrier():  # the default (gloo) group: a rendez-vous of the host processes, never on the device's queue
        if world > 1:
            dist.barrier()

    if args.spinup_ms > 0:
        scratch, _, _ = make_trajectories(kind, n, b, 1 << 20, rank, seed=2, row_offset=row0)
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            scratch.advance(256)
            torch.cuda.synchronize(dev)
        del scratch
    # HIP events on the stream the engine launches on (torch's current stream: engine._stream_ptr)
    def any_rank(flag):
        if world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item())

    elapsed, stream_ms, attempts = timed_steps(traj, args.warmup, args.steps, dev, barrier, any_rank)
    gpu_ms_per_step = stream_ms / args.steps  # stream time of the timed region / steps

    per_rank_elapsed = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)  # host scalars over the default (gloo) group
        each = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(each, t)
        per_rank_elapsed = [float(x.item()) for x in each]
    elapsed = max(per_rank_elapsed)

    # the steps right after the loop (device-side finalize: clamp, change of variables, optional
    # post-processor, energy) + the one collective: all-gather of the objective values (RCCL)
    from ccvm_amd.workloads import scaled_qv

    _, _, f = scaled_qv(n, kind)
    pp_seconds = 0.0
    name = "mu_tilde" if kind == "mf" else "c"
    # (DL's final clamp, dl_solver.py:567, inside the finalize: behind score's own verification of the state, never on
    # a state a recovery would still replace)
    obj, pp_seconds = traj.score(name, SATURATION[kind], float(f), post_processor=args.post,
                                 clamp=(-1.0, 1.0) if kind == "dl" else None)
    finite = bool(torch.isfinite(obj).all().item())
    ranks_seen = 1
    if world > 1:
        obj = obj.to(coll["device"])
        if args.global_batch is not None:  # shards differ by a row: gather equal-length, +inf-padded vectors
            per = -(-args.global_batch // world)
            obj = torch.cat([obj, torch.full((per - obj.numel(),), float("inf"), dtype=obj.dtype, device=obj.device)])
        gathered = [torch.empty_like(obj) for _ in range(world)]
        dist.all_gather(gathered, obj, group=coll["group"])  # THE data collective: RCCL (or what the ranks agreed on)
        obj = torch.cat(gathered)
        if args.global_batch is not None:
            obj = obj[torch.isfinite(obj) | torch.isnan(obj)]
        ranks_seen = dist.get_world_size(coll["group"])
    best = float((-obj).max().item())

    if rank == 0:
        na = 2 if kind == "dl" else 1
        launch = describe_launch(kind, b, n)
        if traj.fallbacks:  # the timed steps ran on the per-step kernel (CCVM_RUN_NO_EXCHANGE), not on the default plan
            launch = f"step_kernel<{SOLVER_ID[kind]}, ...> per step, after a recovered time-out of: {launch}"
        persistent = not traj.fallbacks and any(k in launch for k in ("persist_kernel", "cluster_kernel", "slab_kernel", "ptile_kernel"))
        # a persistent launch runs up to 4096 steps (the schedule table of a run call, ccvm_abi.hip: TABLE_STEPS);
        # traj.advance(steps) in fused-noise mode is ONE run call = ceil(steps / 4096) launches
        launches = -(-args.steps // 4096) if persistent else args.steps
        # (a batch of several rounds on the persistent tile kernel: its slices of rows are launches of their own, each
        # over all the steps -- a "step" of the batch below is a step of every slice)
        sliced = re.search(r"(\d+) slices of the batch", launch)
        slices = int(sliced.group(1)) if sliced else 1
        steps_per_launch = args.steps / launches
        flops_per_step = 2.0 * na * n * n * b
        bytes_per_step = (16.0 if kind in ("dl", "mf") else 8.0) * n * b + 4.0 * n * n
        achieved = flops_per_step / (gpu_ms_per_step * 1e-3) / 1e12
        wall_ms_per_step = elapsed / args.steps * 1e3
        metric = "SDE row-steps/s (Euler-Maruyama steps/s x batch)"
        if args.workload == "dl_n1000_b1000":
            metric += ", DL-CCVM N=1000 batch=1000 per GPU"
        roofline = {
            "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            # frac: the kernel (HIP events on the launch stream); frac_wall: the same flops over this line's own
            # ms_per_step (host wall clock incl. the two synchronisations and post-idle launches)
            "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
            "frac_wall": flops_per_step / (wall_ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "traffic": None,  # HBM bytes need separate rocprofv3 --pmc passes: see "profiled"
            "algorithmic_bytes": ((bytes_per_step - 4.0 * n * n) / slices + 4.0 * n * n) * steps_per_launch,
            "algorithmic_flops": flops_per_step * steps_per_launch / slices,
            "kernel": launch,
            "steps_per_launch": steps_per_launch,
            "launches": launches * slices,
            "avg_launch_us": gpu_ms_per_step * 1e3 * steps_per_launch / slices,
            "avg_step_us": gpu_ms_per_step * 1e3,
            "timing": "HIP events on the launch stream around the timed region / launches in it",
            "peak_note": "157.3 TFLOP/s = fp32 MFMA spec (v_mfma_f32_32x32x2_f32); a bare MFMA loop "
                         "sustains ~141 TFLOP/s at steady-state clocks on this chip (tools/ablate.hip)",
            "hbm_algorithmic_GBps": bytes_per_step / (gpu_ms_per_step * 1e-3) / 1e9,
            "hbm_frac": bytes_per_step / (gpu_ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
        }
        if launch.startswith("batch cut in two"):
            # two run plans one after the other (rows of whole resident grids + the rest): the line prices the step of
            # the whole batch against the MFMA peak; there is no single "launch" to quote
            for key in ("algorithmic_bytes", "algorithmic_flops", "steps_per_launch", "launches", "avg_launch_us"):
                roofline[key] = None
        else:
            roofline.update(family_roof(launch, kind, n, b, gpu_ms_per_step * 1e3, wall_ms_per_step * 1e3))
        out = {
            "metric": metric,
            "value": args.steps * global_rows / elapsed,
            "unit": "row-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_ms_per_step,
            "ms_per_step_per_rank": [e / args.steps * 1e3 for e in per_rank_elapsed],
            "higher_is_better": True,
            "scaling": "strong" if args.global_batch is not None else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "spinup_ms": args.spinup_ms,
            "n_ranks_seen": ranks_seen,
            "config": {
                "workload": f"{args.workload}: {kind.upper()} solver, N={n} dense symmetric BoxQP, "
                            + (f"global batch {global_rows} split over {world} GPU ({b} rows on rank 0), "
                               if args.global_batch is not None else f"batch {b} per GPU x {world} GPU, ")
                            + f"fp32 state, fused Threefry noise, schedule of a {total}-step run"
                            + (f", {args.post} post-processor on device after the loop" if args.post else ""),
                "global_batch": global_rows,
                "rows_per_rank": [shard_rows(global_rows, world, r)[1] for r in range(world)],
                "parallelism": f"batch-sharded x{world}, no data-path collective; one all-gather of "
                               f"{global_rows} objective values after the loop "
                               f"({coll['collective']})",
            },
            "collective": coll["collective"],
            "roofline": roofline,
            "check": {"objective_values_finite": finite, "best_objective_value": best,
                      "post_processor": args.post, "pp_seconds": pp_seconds, "time_outs_recovered": traj.fallbacks,
                      # > 1: warm-up or timed steps were invalid (a persistent kernel gave up its bounded wait) and the
                      # run started over on the per-step kernel -- the line's times are the LAST attempt's
                      "timed_attempts": attempts},
        }
        prof = profiled_counters(args.workload)
        if prof is not None:
            out["roofline"]["profiled"] = prof
            per_step = prof.get("traffic_bytes_per_step")
            out["roofline"]["traffic"] = None if per_step is None else per_step * steps_per_launch
            out["roofline"]["traffic_source"] = (f"{prof['source']}: separate rocprofv3 --pmc passes of this command "
                                                 f"({prof.get('dispatches')} dispatches), NOT measured by this run")
        if world == 1 and args.workload == "dl_n1000_b1000":
            out["tts99"] = tts99_leg()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(kind, n, b, total, threads=args.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
