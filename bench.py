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
    # the reference's own shipped regime (N = 20 ... 70: the only sizes with known optima, i.e. where TTS @ 99 % exists):
    # BASELINE config 1 exactly (test020-100-10.in, batch 100, the schedule of a 15000-iteration run), the shipped example
    # (examples/ccvm_boxqp_dl.py:12-24: tuningH020-100-0.in, batch 1000) for every solver, and the largest shipped size
    "dl_n20_b100": ("dl", 20, 100),
    "dl_n20_b1000": ("dl", 20, 1000),
    "mf_n20_b1000": ("mf", 20, 1000),
    "langevin_n20_b1000": ("langevin", 20, 1000),
    "pl_n20_b1000": ("pl", 20, 1000),
    "dl_n70_b1000": ("dl", 70, 1000),
    # between the shipped sizes and the cluster kernel's range (round 6): the row-owner kernel's twelve-wave workgroups
    # (three waves side by side x two K halves x two row sets) and, with the N = 300 workloads above, its five waves side by side
    "dl_n160_b1000": ("dl", 160, 1000),
    "mf_n257_b1000": ("mf", 257, 1000),
}
#: workloads on a shipped instance (arrays: tests/golden/<instance>.npz, made from the reference's .in files) and / or
#: inside the schedule of a longer run than the timed steps (`total`: the iterations of the run the steps belong to)
WORKLOAD_EXTRAS = {
    "dl_n20_b100": {"instance": "test020", "total": 15000},
    "dl_n20_b1000": {"instance": "tuningH020"},
    "mf_n20_b1000": {"instance": "tuningH020"},
    "langevin_n20_b1000": {"instance": "tuningH020"},
    "pl_n20_b1000": {"instance": "tuningH020"},
}
REPEATS = 9  # timed regions of exactly K steps per run; the line reports the median one
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


def gather_values(values, coll, world, all_gather=None):
    """THE data collective of a multi-rank line: every rank's vector of objective values (equal lengths) to every rank,
    over the group the ranks agreed on (init_collectives) -- and, if that group's all-gather fails on ANY rank after
    all (RCCL's set-up was proven by a one-element all-reduce, not by this call), over gloo on host copies, decided by
    the same min-reduce on the control plane, so that a transport problem after the timed region costs a label, not
    the line.  Returns (list of `world` host tensors, the collective's label).  `all_gather` stands in for
    dist.all_gather in tests."""
    import torch.distributed as dist

    label = coll["collective"]
    if coll["group"] is not None:
        parts, err = None, ""
        try:
            mine = values.to(coll["device"])
            parts = [torch.empty_like(mine) for _ in range(world)]
            (all_gather or dist.all_gather)(parts, mine, group=coll["group"])
            parts = [p.cpu() for p in parts]  # (the copy waits for the collective: an asynchronous failure surfaces here)
        except Exception as exc:  # noqa: BLE001 -- whatever the transport raises
            err = f"{type(exc).__name__}: {exc}".replace("\n", " ")[:300]
        ok = torch.tensor([0 if err else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # the default (gloo) group
        if int(ok.item()) == 1:
            return parts, label
        errs = [None] * world
        dist.all_gather_object(errs, err)
        first = next((f"rank {r}: {e}" for r, e in enumerate(errs) if e), "unknown")
        label = f"gloo-fallback (the all-gather over {label} failed: {first})"
    host = values.detach().cpu()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host)
    return parts, label


def timed_steps(traj, warmup, steps, dev, barrier, any_rank=lambda flag: flag, repeats=1):
    """W untimed warm-up steps, then `repeats` timed regions of EXACTLY K steps of `traj` each, every one under the
    contract's clock -- barrier + device synchronisation on both sides, HIP events on the launch stream around the same
    region -- and the whole sequence repeated when steps turn out INVALID: a persistent kernel whose workgroups could not
    all become resident gives up a bounded wait and sets the run's status word; its steps are garbage, and `traj.check`
    puts the trajectories back where the recovery snapshot was taken and moves the run to the per-step kernel.  The
    snapshot is taken ONCE, in front of the warm-up steps (two device-to-device copies: right in front of a timed region
    they would push Q and the state out of the L2s), the status word is read once per region, right after its clock
    stops (4 bytes; inside the region it would cost ~25 us of a 0.65 ms run), and a run that was recovered -- the word is
    set by warm-up and timed steps alike and never cleared by a kernel -- starts over from the snapshot on the path that
    then runs, so no reported time ever describes discarded work.  `any_rank(flag)`: True when the flag is
    set on ANY rank: the ranks start over together (the barriers must pair up).
    Returns ([wall seconds per region], [stream ms per region], attempts)."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(repeats)]
    for ev0, ev1 in evs:
        ev0.record()  # torch creates the HIP event at its first record (15-20 us): the measuring apparatus is set up
        ev1.record()  # BEFORE the timed regions (tools/sync_probe.py: 36.3 -> 35.5 us per step on a 20-step run)
    traj.arm(force=True)  # (forced: another rank may ask this one to start over although its own steps were valid)
    for attempt in (1, 2, 3):
        traj.advance(warmup)  # (not verified by itself: a status word set here is still set behind the timed steps)
        walls, valid = [], True
        for ev0, ev1 in evs:
            torch.cuda.synchronize(dev)
            barrier()     # every rank starts its K steps together ...
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            ev0.record()
            traj.advance(steps)
            ev1.record()
            torch.cuda.synchronize(dev)
            walls.append(time.perf_counter() - t0)  # ... and stops ITS OWN clock when its own K steps are done: no collective
            #                                     inside a timed region (a 50-150 us barrier would read as a 7-20 % "scaling
            #                                     loss" on the driver's 0.7 ms); the job's time is the MAX over ranks
            recovered = traj.check(rerun=False, hold=True)  # True: back at the snapshot, on the per-step kernel from here on
            if any_rank(recovered):
                if not recovered:
                    traj.rollback()  # another rank starts over: this one does with it
                valid = False
                break
        if valid:
            traj.check()  # (drops the snapshot)
            return walls, [ev0.elapsed_time(ev1) for ev0, ev1 in evs], attempt
    raise SystemExit("bench.py: the run was invalid three times in a row")


def median(values):
    v = sorted(values)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


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
    if "SQ_INSTS_VALU" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # executed vector instructions (MFMAs included) and matrix-pipe busy cycles, summed over the chip, per time step
        per = doc.get("steps_per_dispatch", 1.0)
        out["insts_valu_per_step"] = c["SQ_INSTS_VALU"]["mean_per_dispatch"] / per
        out["mfma_busy_cycles_per_step"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / per
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


def problem_qv(kind, n, instance=None):
    """(Q, V, scaled_by) of a workload: the synthetic recipe of SURVEY 8(d), or a shipped instance's arrays (fixtures
    under tests/golden/, made from the reference's .in files) scaled the way the solver scales them
    (ccvm_solver.py get_scaling_factor -> problem_instance.py scale_coefs)."""
    from ccvm_amd.workloads import SCALING_MULTIPLIER, scaled_qv

    if instance is None:
        return scaled_qv(n, kind)
    import numpy as np

    arrays = np.load(os.path.join(ROOT, "tests", "golden", f"{instance}.npz"))
    q = torch.from_numpy(arrays["q_matrix"]).to(torch.float32)
    v = torch.from_numpy(arrays["v_vector"]).to(torch.float32)
    assert q.shape == (n, n), (instance, q.shape, n)
    f = torch.sqrt(torch.sum(torch.abs(q))) * SCALING_MULTIPLIER[kind]
    return q / f, v / f, f


def make_trajectories(kind, n, b, total_steps, rank, seed=1, row_offset=None, adam=None, instance=None):
    from ccvm_amd import engine

    q, v, _ = problem_qv(kind, n, instance)
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


def cpu_baseline(kind, n, b, total_steps, budget_s=30.0, threads=None, instance=None, min_steps=200, min_seconds=2.0,
                 warm_steps=20):
    """The oracle (torch CPU restatement of the reference loop, bit-identical to it on this torch build) timed on the
    host cores on a bounded sample of the same workload: `warm_steps` untimed steps, then steps until BOTH at least
    `min_steps` steps and `min_seconds` seconds are timed (>= 200 timed steps after >= 20
    warm-up) -- whatever --steps says -- or `budget_s` seconds are spent.  Threads: a 1-GPU box exposes a share of the
    host's cores; profiles/r06_cpu_thread_sweep.md holds the 1..64-thread sweep behind the default (more threads than
    the share only oversubscribe).  `cores` = the threads used (the contract's field), `threads` the same,
    `cores_visible` = the cores this process may run on."""
    from oracle import ccvm_oracle as oracle

    visible = len(os.sched_getaffinity(0))
    # profiles/r06_cpu_thread_sweep.md (refreshed on this round's box): 16 threads are the CPU path's best from N = 500
    # (12.7 ms per step at the headline; 32 threads 22 ms), ONE thread below N ~ 200 (N = 20: 0.29 ms against 0.40 on 16;
    # N = 100: 1.25 against 1.75) -- the einsum of a small problem does not amortise a thread team
    default_threads = min(16, visible) if float(b) * n * n >= 1.0e8 else 1
    torch.set_num_threads(max(1, threads or default_threads))
    q, v, _ = problem_qv(kind, n, instance)
    p = workload_params(kind)
    torch.manual_seed(1)
    bounds = (0.0, 1.0)
    chunk = 10
    schedule_total = max(total_steps, 20000)  # (the sample's steps must exist in the schedule of the run)
    if kind == "dl":
        c = torch.zeros((b, n)); s = torch.zeros((b, n))
        run = lambda step0, k: oracle.dl_loop(q, v, b, schedule_total, p["pump"], p["dt"], p["noise_ratio"],
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
    run(0, warm_steps)  # warm-up (first-call overheads, thread pool, caches)
    done, t0 = warm_steps, time.time()
    while True:
        run(done, chunk)
        done += chunk
        dt = time.time() - t0
        if (done - warm_steps >= min_steps and dt >= min_seconds) or dt >= budget_s or done + chunk > schedule_total:
            break
    steps = done - warm_steps
    used = torch.get_num_threads()
    return {
        "value": b * steps / dt, "unit": "row-steps/s", "cores": used, "threads": used, "cores_visible": visible,
        "kind": "port", "timed_steps": steps, "warmup_steps": warm_steps, "seconds": dt,
        "sample": (f"{steps} steps of the same workload (N={n}, batch={b}) after {warm_steps} warm-up steps " if kind == "dl" else
                   f"{steps // chunk} whole {chunk}-step runs after {warm_steps} warm-up steps (N={n}, batch={b}: the same "
                   f"per-step work, each run with its own set-up and a {chunk}-step schedule, not steps of the benchmarked run) ")
                  + f"with the torch-CPU oracle "
                  f"(bit-identical to the reference's CPU path), {dt / steps * 1e3:.2f} ms/step, "
                  f"{used} torch threads on {cpu_model()} ({visible} cores visible to this "
                  f"process), torch {torch.__version__}",
    }


def tts99_leg():
    """Secondary metric of BASELINE.json: TTS @ 99 % success = per-row solve time x R99
    (ccvmplotlib/utils/sampleTTSmetric.py:144-153).  Only defined where the optimum is known, so it is
    measured on the reference's shipped tuning instance tuningH020-100-0 (arrays: tests/golden fixture)
    with the shipped example configurations (examples/ccvm_boxqp_dl.py:12-24 and its three siblings: B=1000, 1500
    iterations), through the public solver API with the fused generator -- DL as `value` (the headline solver), all
    four under `solvers`, each next to the oracle's TTS on this box's host cores (`cpu`: one solve of the same
    configuration with torch's CPU stream, one thread -- the best at this size, profiles/r06_cpu_thread_sweep.md; the
    full protocol with repeats and the second instance: tools/tts_report.py, profiles/r06_tts.md)."""
    import numpy as np

    from ccvm_amd.problem_classes.boxqp import ProblemInstance
    from ccvm_amd.solvers import DLSolver, LangevinSolver, MFSolver, PumpedLangevinSolver
    from ccvm_amd.workloads import EXAMPLE_PARAMS, SCALING_MULTIPLIER
    from oracle import ccvm_oracle as oracle

    gdir = os.path.join(ROOT, "tests", "golden")
    arrays = np.load(os.path.join(gdir, "tuningH020.npz"))
    with open(os.path.join(gdir, "tuningH020.json")) as fh:
        meta = json.load(fh)["instance"]
    batch, iterations, bounds = 1000, 1500, (0.0, 1.0)
    out = {}
    for kind, cls in (("dl", DLSolver), ("mf", MFSolver), ("langevin", LangevinSolver), ("pl", PumpedLangevinSolver)):
        inst = ProblemInstance.from_arrays(arrays["q_matrix"], arrays["v_vector"], device="cuda", name=meta["name"],
                                           optimal_sol=meta["optimal_sol"], best_sol=meta["best_sol"])
        solver = cls(device="cuda", batch_size=batch)
        solver.parameter_key = {20: dict(EXAMPLE_PARAMS[kind], iterations=iterations)}
        inst.scale_coefs(solver.get_scaling_factor(inst.q_matrix))
        torch.manual_seed(1234)
        solver(instance=inst)  # first call: one-time initialisation
        sol = solver(instance=inst)
        # the oracle: the reference's loop on the host (time of the loop only, as solve_time is)
        q = torch.from_numpy(arrays["q_matrix"]).float()
        v = torch.from_numpy(arrays["v_vector"]).float()
        f = oracle.scaling_factor(q, SCALING_MULTIPLIER[kind])
        qs, vs, p = q / f, v / f, EXAMPLE_PARAMS[kind]
        torch.set_num_threads(1)
        torch.manual_seed(1234)
        t0 = time.perf_counter()
        if kind == "dl":
            c, _ = oracle.dl_loop(qs, vs, batch, iterations, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"], 0.05, bounds)
            x = oracle.change_variables(torch.clamp(c, -1, 1), 0.0, 1.0, 1)
        elif kind == "mf":
            _, mt, _ = oracle.mf_loop(qs, vs, batch, iterations, p["pump"], p["dt"], p["j"], p["feedback_scale"], p["S"], 0.01, bounds)
            x = oracle.change_variables(mt, 0.0, 1.0, p["S"])
        else:
            loop = oracle.pl_loop if kind == "pl" else oracle.langevin_loop
            args = (p["pump"],) if kind == "pl" else ()
            c = loop(qs, vs, batch, iterations, *args, p["dt"], p["sigma"], p["feedback_scale"], p["S"], bounds)
            x = (c + p["S"]) / (2 * p["S"])
        cpu_row = (time.perf_counter() - t0) / batch
        _, perf = oracle.solution_stats(oracle.compute_energy(x, qs, vs, float(f)), meta["optimal_sol"])
        out[kind] = {
            "tts99_s": sol.tts99(), "p_optimal": sol.solution_performance["optimal"], "solve_time_per_row_s": sol.solve_time,
            "best_objective_value": sol.best_objective_value,
            "cpu": {"tts99_s": cpu_row * oracle.r99(perf["optimal"]), "p_optimal": perf["optimal"],
                    "solve_time_per_row_s": cpu_row, "threads": 1},
        }
    dl = out["dl"]
    return {
        "value": dl["tts99_s"], "unit": "s", "instance": meta["name"], "batch": batch, "iterations": iterations,
        "p_optimal": dl["p_optimal"], "solve_time_per_row_s": dl["solve_time_per_row_s"],
        "best_objective_value": dl["best_objective_value"], "optimal_value": meta["optimal_sol"], "solvers": out,
    }


def family_model(launch, kind, n, b, step_us, wall_step_us, workload):
    """Performance MODELS of the kernel families whose step is not explained by a hardware roof -- reported beside the
    roofline (`roofline.model`, `roofline.model_frac`), never as `roofline.frac`: a model built from the kernel's own
    instruction counts moves when the kernel changes.

    persist_kernel (row owners, N <= 256): a SIMD issues the instructions of its waves one after the other -- f32 MFMA
    and VALU do not overlap on a SIMD (tools/coissue.hip) -- so a step cannot take less than the chip's executed vector
    instructions of a step, spread evenly over the 1024 SIMDs: matrix-pipe busy cycles + 4 cycles per other VALU
    instruction, both from the committed SQ pass of this workload (no pass, no model).

    slab_kernel (small batches, Q in registers, ccvm_slab.h): no byte of Q moves, the contraction of a member is R C K
    MACs (a few hundred cycles); a step is one hand-off of the GEMM input between the cluster's workgroups (own publish
    -> every peer's packets read: MI355X_MICROARCH.md price list, "handoff-1to1" 0.8-1.1 us idle / "allgather" 2.4-2.9
    us for 256 CUs) plus the member's matrix time."""
    if "persist_kernel" in launch:
        prof = profiled_counters(workload)
        if not prof or "insts_valu_per_step" not in prof:
            return {}
        mfma_cycles = prof["mfma_busy_cycles_per_step"] / 1024.0
        other = 4.0 * (prof["insts_valu_per_step"] - prof["mfma_busy_cycles_per_step"] / 8.0) / 1024.0
        issue_us = (mfma_cycles + other) / 2400.0
        return {
            "model": {"bound": "issue", "floor_us_per_step": issue_us, "clock_MHz": 2400,
                      "issue_cycles_per_simd_and_step": {"mfma": mfma_cycles, "other_valu": other},
                      "source": prof["source"] + " (SQ_VALU_MFMA_BUSY_CYCLES + 4 x the other SQ_INSTS_VALU, chip sums / "
                                                 "1024 SIMDs: an even spread, i.e. a lower bound of the fullest SIMD)",
                      "kernel_profiled": prof.get("kernel")},
            "model_frac": issue_us / step_us, "model_frac_wall": issue_us / wall_step_us,
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
        "model": {"bound": "latency", "floor_us_per_step": floor_us, "handoff_us": handoff_us, "member_mfma_us": mfma_us,
                  "source": "MI355X_MICROARCH.md price list: handoff-1to1 0.8 us (one XCD, idle) / allgather 2.4 us "
                            "(256 CUs, 8 KB); v_mfma_f32_4x4x1 8 cycles",
                  "q_bytes_moved_per_step": 0, "exchange_bytes_per_member_per_step": planes * int(rows) * int(k) * 8},
        "model_frac": floor_us / step_us, "model_frac_wall": floor_us / wall_step_us,
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
    ap.add_argument("--repeats", type=int, default=REPEATS,
                    help="timed regions of exactly --steps steps each (every one verified, every one under the barrier + "
                         "synchronise clock); ms_per_step / value are the MEDIAN region's, ms_per_step_repeats lists all")
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
        # several processes on ONE GPU: a rank's resident grid waits for the other ranks' whole launches, not for a step --
        # the 20 ms bound of a cross-workgroup wait (ccvm_abi.hip: spin_ticks) is for a GPU of one's own
        os.environ.setdefault("CCVM_AMD_SPIN_MS", "2000")
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
    extras = WORKLOAD_EXTRAS.get(args.workload, {})
    instance = extras.get("instance")
    repeats = max(1, args.repeats)
    # the iterations of the run the steps belong to (its schedule: pump ramp, noise-ratio decay): the steps this
    # process runs, or the workload's own run length when that is longer (config 1: 15000)
    total = max(args.warmup + repeats * args.steps, extras.get("total", 0))
    traj, q, v = make_trajectories(kind, n, b, total, rank, row_offset=row0, instance=instance)

    def barrier():  # the default (gloo) group: a rendez-vous of the host processes, never on the device's queue
        if world > 1:
            dist.barrier()

    if args.spinup_ms > 0:
        scratch, _, _ = make_trajectories(kind, n, b, 1 << 20, rank, seed=2, row_offset=row0, instance=instance)
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

    walls, stream_ms_all, attempts = timed_steps(traj, args.warmup, args.steps, dev, barrier, any_rank, repeats)
    gpu_ms_per_step_all = [ms / args.steps for ms in stream_ms_all]  # stream time of each timed region / steps
    gpu_ms_per_step = median(gpu_ms_per_step_all)

    per_rank_walls = [walls]  # [rank][region]
    if world > 1:
        t = torch.tensor(walls, dtype=torch.float64)  # host scalars over the default (gloo) group
        each = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(each, t)
        per_rank_walls = [[float(x) for x in e.tolist()] for e in each]
    # a region's time for the JOB is the slowest rank's; the line reports the median region
    job_walls = [max(per_rank_walls[r][i] for r in range(world)) for i in range(repeats)]
    elapsed = median(job_walls)
    mid = min(range(repeats), key=lambda i: abs(job_walls[i] - elapsed))  # (the region the per-rank list is quoted from)
    per_rank_elapsed = [per_rank_walls[r][mid] for r in range(world)]

    # the steps right after the loop (device-side finalize: clamp, change of variables, optional
    # post-processor, energy) + the one collective: all-gather of the objective values (RCCL)
    _, _, f = problem_qv(kind, n, instance)
    pp_seconds = 0.0
    name = "mu_tilde" if kind == "mf" else "c"
    # (DL's final clamp, dl_solver.py:567, inside the finalize: behind score's own verification of the state, never on
    # a state a recovery would still replace)
    obj, pp_seconds = traj.score(name, SATURATION[kind], float(f), post_processor=args.post,
                                 clamp=(-1.0, 1.0) if kind == "dl" else None)
    finite = bool(torch.isfinite(obj).all().item())
    ranks_seen = 1
    if world > 1:
        if args.global_batch is not None:  # shards differ by a row: gather equal-length, +inf-padded vectors
            per = -(-args.global_batch // world)
            obj = torch.cat([obj, torch.full((per - obj.numel(),), float("inf"), dtype=obj.dtype, device=obj.device)])
        gathered, coll["collective"] = gather_values(obj, coll, world)  # THE data collective: RCCL (or what the ranks agreed on)
        obj = torch.cat(gathered)
        if args.global_batch is not None:
            obj = obj[torch.isfinite(obj) | torch.isnan(obj)]
        ranks_seen = len(gathered)
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
        # The hardware roof that binds: by arithmetic intensity against the ridge of the two peaks (157.3 TFLOP/s fp32
        # MFMA / 8 TB/s HBM = 19.7 flop/B).  `frac` is ALWAYS a fraction of a hardware peak: of the MFMA
        # peak above the ridge, of the HBM peak below it (N < ~85: the shipped instances); both fractions are in the
        # line either way (`mfma_frac`, `hbm_frac`), the families' instruction-count / latency models under `model`.
        hbm_gbs = bytes_per_step / (gpu_ms_per_step * 1e-3) / 1e9
        ai, ridge = flops_per_step / bytes_per_step, PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        wall_scale = gpu_ms_per_step / wall_ms_per_step
        if ai >= ridge:
            head = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_MFMA_TFLOPS}
        else:
            head = {"bound": "hbm", "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS}
        roofline = {
            **head,
            # frac: the kernel (HIP events on the launch stream); frac_wall: the same over this line's own
            # ms_per_step (host wall clock incl. the two synchronisations and post-idle launches)
            "frac_wall": head["frac"] * wall_scale,
            "arithmetic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
            "mfma_frac": achieved / PEAK_FP32_MFMA_TFLOPS, "mfma_TFLOPs": achieved,
            "traffic": None,  # HBM bytes need separate rocprofv3 --pmc passes: see "profiled"
            "algorithmic_bytes": ((bytes_per_step - 4.0 * n * n) / slices + 4.0 * n * n) * steps_per_launch,
            "algorithmic_flops": flops_per_step * steps_per_launch / slices,
            "kernel": launch,
            "steps_per_launch": steps_per_launch,
            "launches": launches * slices,
            "avg_launch_us": gpu_ms_per_step * 1e3 * steps_per_launch / slices,
            "avg_step_us": gpu_ms_per_step * 1e3,
            "avg_step_us_repeats": [ms * 1e3 for ms in gpu_ms_per_step_all],
            "timing": "HIP events on the launch stream around each timed region / launches in it; the median region",
            "peak_note": "157.3 TFLOP/s = fp32 MFMA spec (v_mfma_f32_32x32x2_f32); a bare MFMA loop "
                         "sustains ~141 TFLOP/s at steady-state clocks on this chip (tools/ablate.hip); 8000 GB/s = HBM3E spec",
            "hbm_algorithmic_GBps": hbm_gbs,
            "hbm_frac": hbm_gbs / PEAK_HBM_GBS,
        }
        if launch.startswith("batch cut in two"):
            # two run plans one after the other (rows of whole resident grids + the rest): the line prices the step of
            # the whole batch against the MFMA peak; there is no single "launch" to quote
            for key in ("algorithmic_bytes", "algorithmic_flops", "steps_per_launch", "launches", "avg_launch_us"):
                roofline[key] = None
        else:
            roofline.update(family_model(launch, kind, n, b, gpu_ms_per_step * 1e3, wall_ms_per_step * 1e3, args.workload))
        out = {
            "metric": metric,
            "value": args.steps * global_rows / elapsed,
            "unit": "row-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_ms_per_step,
            "ms_per_step_repeats": [w / args.steps * 1e3 for w in job_walls],
            "repeats": repeats,
            "timing": f"{repeats} timed regions of exactly {args.steps} steps each (barrier + device synchronisation on "
                      "both sides of every one, every one verified); ms_per_step / value = the MEDIAN region (max over "
                      "ranks per region), ms_per_step_repeats = all of them in order",
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
                            + (f"shipped instance {instance} (tests/golden fixture), " if instance else "")
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
            out["cpu_baseline"] = cpu_baseline(kind, n, b, total, threads=args.cpu_threads, instance=instance)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
