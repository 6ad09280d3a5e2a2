"""Host side of the HIP dynamics engine: device buffers, noise sources, chunking.

This module is plumbing around libccvm_hip.so: it owns pitched device buffers
(torch is used for memory and streams only), turns solver parameters into the C
structs of include/ccvm_hip.h and drives ``ccvm_*_run`` in chunks.  All arithmetic of
the SDE loop happens in the HIP kernels; there is no CPU implementation here and
every entry point raises ``EngineUnavailable`` when no MI355X / no built library is
present.

``device`` strings follow the reference API ("cpu" / "cuda") but name where the
CALLER's tensors live: host tensors are staged to the GPU and results are copied
back.  The GPU used is ``cuda:$LOCAL_RANK`` (or ``$CCVM_AMD_DEVICE``, default 0).
"""
import collections
import contextlib
import ctypes
import os
import threading
import time
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import EngineUnavailable

_REPLAY_CHUNK_FLOATS = 32 * 1024 * 1024  # per noise stream, per chunk (128 MiB)


# --------------------------------------------------------------------------- #
# device / library access
# --------------------------------------------------------------------------- #
_gpu_seen = False


def gpu_device():
    """The torch device the engine runs on; raises if there is none."""
    global _gpu_seen
    lib = _lib.load()  # fail on a missing library before touching the GPU
    del lib
    if not _gpu_seen:  # (asked once per process: torch.cuda.is_available() costs milliseconds per call on ROCm)
        if not torch.cuda.is_available():
            raise EngineUnavailable(
                "no MI355X visible (torch.cuda.is_available() is False); the CCVM dynamics engine"
                " has no CPU fallback"
            )
        _gpu_seen = True
    index = int(os.environ.get("CCVM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if index != 0 and not 0 <= index < torch.cuda.device_count():
        # (one process per GPU under a launcher: LOCAL_RANK names the device unless the launcher also narrowed
        # *_VISIBLE_DEVICES to one GPU per rank -- then CCVM_AMD_DEVICE=0 says so)
        raise EngineUnavailable(
            f"device index {index} (from ${'CCVM_AMD_DEVICE' if 'CCVM_AMD_DEVICE' in os.environ else 'LOCAL_RANK'}) "
            f"is outside the {torch.cuda.device_count()} GPU(s) visible to this process; set CCVM_AMD_DEVICE"
        )
    return torch.device("cuda", index)


# Host threads may drive the engine concurrently (one stream each): the module-level caches below -- which devices
# are warm, which configurations are primed, the staged problems -- are only touched under this lock (re-entrant:
# prime() warms up, and both stage problems).
_cache_lock = threading.RLock()
_warm = set()


def warmup():
    with _cache_lock:
        _warmup_locked()


def _warmup_locked():
    """Once per process and device: create the HIP context and load every code object of the library
    (one per translation unit: the ABI + tile kernels, and the five persistent-kernel units), so that
    this one-time cost (~0.2 s) never lands inside a solver's timed region.  One elementwise launch
    plus one step of a 1 x 1 problem through each solver entry point."""
    dev = gpu_device()
    if dev.index in _warm:
        return
    lib = _lib.load()
    with torch.cuda.device(dev):
        x = torch.zeros((64, 128), dtype=torch.float32, device=dev)
        _lib.check(lib.ccvm_clamp(_ptr(x), 1, 1, 128, 0.0, 1.0, _stream_ptr()), "ccvm_clamp (warm-up)")
        prob = DeviceProblem(torch.ones((1, 1)), torch.ones(1))
        adam = {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False}
        dl = {"pump": 2.0, "dt": 0.001, "noise_ratio": 2.0, "feedback_scale": 1.0, "g": 0.05}
        mf = {"pump": 0.0, "dt": 0.001, "j": 1.0, "feedback_scale": 1.0, "S": 1.0, "g": 0.01}
        lv = {"dt": 0.001, "sigma": 0.1, "feedback_scale": 1.0, "S": 1.0, "pump": 0.0, "use_pump": False}
        for kind, params, ad in (("dl", dl, None), ("mf", mf, None), ("mf", mf, adam), ("langevin", lv, None),
                                 ("langevin", lv, adam)):
            Trajectories(prob, 1, kind, 1, params, (0.0, 1.0), NoiseSpec(mode="philox", seed=1), adam=ad).advance(1)
        torch.cuda.synchronize(dev)
    _warm.add(dev.index)  # only after it succeeded: a failed warm-up is retried by the next call


_primed = set()


def prime(kind, n, batch, adam=None):
    """Once per (solver kind, N, batch, Adam variant): one untimed step of a zero problem of the same
    shape, so that first-use costs of THIS configuration (the caching allocator's first hipMalloc of
    each buffer size, the runtime's host staging buffers for the copies, the lazy load of the kernel
    instantiation it selects) do not land inside the
    solve timer of the first real call (8 ms against a 1 ms solve for the shipped example)."""
    warmup()
    if n > 2048:
        return  # a dry run would stage a zero N x N matrix; at these sizes the first-use costs are noise
    with _cache_lock:
        _prime_locked(kind, n, batch, adam)


def _prime_locked(kind, n, batch, adam):
    dev = gpu_device()
    use_v = bool(adam) and float(adam["beta2"]) != 1.0
    key = (dev.index, kind, int(n), int(batch), bool(adam), use_v)
    if key in _primed:
        return
    with torch.cuda.device(dev):
        prob = DeviceProblem(torch.zeros((n, n)), torch.zeros(n))
        params = {
            "dl": {"pump": 2.0, "dt": 0.001, "noise_ratio": 2.0, "feedback_scale": 1.0, "g": 0.05},
            "mf": {"pump": 0.0, "dt": 0.001, "j": 1.0, "feedback_scale": 1.0, "S": 1.0, "g": 0.01},
            "langevin": {"dt": 0.001, "sigma": 0.1, "feedback_scale": 1.0, "S": 1.0, "pump": 0.0, "use_pump": False},
        }[kind]
        traj = Trajectories(prob, batch, kind, 1, params, (0.0, 1.0), NoiseSpec(mode="philox", seed=1),
                            adam=dict(adam) if adam else None)
        traj.advance(1)
        for name in traj.state:  # the runtime's host staging buffers for results of this size, through the very
            traj.compact(name).cpu()          # operations a solver's __call__ uses (the strided view's first copy loads
            traj.view(name).to("cpu")         # torch's copy kernel: 50 ms inside the first solve_time otherwise)
        torch.cuda.synchronize(dev)
    _primed.add(key)  # only after it succeeded


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr():
    """torch's current stream on the current device as a hipStream_t (the raw-handle query where this torch has it:
    0.3 us instead of the 3 us of building a torch.cuda.Stream object -- a run call of 20 steps is 650 us)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def ld_of(n):
    return _lib.load().ccvm_ld(int(n))


def rows_of(b):
    return _lib.load().ccvm_rows(int(b))


def pack(src, rows_pad, ld):
    """Zero-padded pitched copy of a 1-D / 2-D float tensor already on the GPU."""
    lib = _lib.load()
    src2 = src.reshape(1, -1) if src.ndim == 1 else src
    src2 = src2.to(torch.float32).contiguous()
    out = torch.empty((rows_pad, ld), dtype=torch.float32, device=src2.device)
    _lib.check(
        lib.ccvm_pack(_ptr(src2), src2.shape[0], src2.shape[1], src2.shape[1], _ptr(out), rows_pad, ld,
                      _stream_ptr()),
        "ccvm_pack",
    )
    return out


def unpack(src, rows, cols):
    lib = _lib.load()
    out = torch.empty((rows, cols), dtype=torch.float32, device=src.device)
    _lib.check(
        lib.ccvm_unpack(_ptr(src), src.shape[1], _ptr(out), rows, cols, cols, _stream_ptr()), "ccvm_unpack"
    )
    return out


# --------------------------------------------------------------------------- #
# noise sources
# --------------------------------------------------------------------------- #
@dataclass
class NoiseSpec:
    """How the Wiener increments of a run are produced.

    mode "fused" (alias "philox", historical): counter-based generator (Threefry2x32-13 keyed by a
        SplitMix64 hash of (seed, step), then Box-Muller) fused into the step kernel; ``row_offset`` is
        the global index of local row 0 (batch sharding).  Any two seeds -- consecutive small integers
        included -- give unrelated streams.
    mode "replay": standard normals are drawn on the host from ``generator`` (None =
        torch's global CPU generator, i.e. exactly what the reference consumes after
        ``torch.manual_seed``) in the reference's order -- per step one (N, B) block
        per stream, DL drawing the c block before the s block (dl_solver.py:538-547) --
        and uploaded chunk by chunk.  Parity mode; bandwidth is irrelevant.
    """

    mode: str = "philox"
    seed: int = 0
    row_offset: int = 0
    generator: Optional[torch.Generator] = None
    #: replay mode under batch sharding: every rank draws the (N, global_batch) block of the UNSHARDED
    #: run from an identically seeded generator and keeps columns [row_offset, row_offset + B), so the
    #: union of the shards is the unsharded run.  None: the block is (N, B) (unsharded).
    global_batch: Optional[int] = None

    def __post_init__(self):
        if self.mode == "fused":  # clearer name for the in-kernel generator ("philox" is historical)
            self.mode = "philox"
        if self.mode not in ("philox", "replay"):
            raise ValueError(f"unknown noise mode {self.mode!r}; expected 'fused' (alias 'philox') or 'replay'")


def draw_seed():
    """A PHILOX key from torch's global CPU generator (so torch.manual_seed reproduces runs)."""
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


def effective_noise_mode(mode=None):
    """The solver's ``noise_mode`` or, when that is None, $CCVM_AMD_NOISE (default: the fused generator)."""
    mode = mode or os.environ.get("CCVM_AMD_NOISE", "philox")
    return "philox" if mode == "fused" else mode


def default_noise(mode=None, row_offset=0, seed=None, global_batch=None):
    """Noise spec for a solver call.  PHILOX seeds default to a draw from torch's global CPU
    generator so ``torch.manual_seed`` makes runs reproducible in both modes."""
    mode = effective_noise_mode(mode)
    if mode == "replay":
        return NoiseSpec(mode="replay", row_offset=row_offset, global_batch=global_batch)
    return NoiseSpec(mode=mode, seed=draw_seed() if seed is None else int(seed), row_offset=row_offset)


class _NoiseFeeder:
    """Builds the ccvm_noise struct for each chunk of steps."""

    def __init__(self, spec, n, b, streams, device):
        self.spec, self.n, self.b, self.streams, self.device = spec, n, b, streams, device
        self._keep = None

    def steps_per_chunk(self):
        if self.spec.mode == "philox":
            return 1 << 30
        return max(1, _REPLAY_CHUNK_FLOATS // (self.n * self.b))

    def chunk(self, nsteps):
        nz = _lib.Noise()
        nz.row_offset = int(self.spec.row_offset)
        if self.spec.mode == "philox":
            nz.mode = _lib.NOISE_PHILOX
            nz.seed = int(self.spec.seed) & 0xFFFFFFFFFFFFFFFF
            return nz
        nz.mode = _lib.NOISE_REPLAY
        host = torch.empty((nsteps, self.streams, self.n, self.b), dtype=torch.float32)
        gb, lo = self.spec.global_batch, int(self.spec.row_offset)
        if gb is not None and not (0 <= lo and lo + self.b <= gb):
            raise ValueError(f"replay noise: rows [{lo}, {lo + self.b}) outside the global batch of {gb}")
        for t in range(nsteps):
            for k in range(self.streams):
                if gb is None:
                    torch.randn((self.n, self.b), generator=self.spec.generator, out=host[t, k])
                else:  # this shard's columns of the unsharded run's block
                    host[t, k] = torch.randn((self.n, gb), generator=self.spec.generator)[:, lo:lo + self.b]
        dev = host.to(self.device)
        w0 = dev[:, 0].contiguous()
        w1 = dev[:, 1].contiguous() if self.streams == 2 else None
        self._keep = (w0, w1)  # alive until the next chunk replaces it (stream-ordered reuse)
        nz.w0 = w0.data_ptr()
        nz.w1 = w1.data_ptr() if w1 is not None else None
        return nz


# --------------------------------------------------------------------------- #
# problem data on the device
# --------------------------------------------------------------------------- #
class DeviceProblem:
    """Q (ld x ld, zero padded) and V (ld) on the GPU."""

    def __init__(self, q_matrix, v_vector):
        self.device = gpu_device()
        self.n = int(q_matrix.shape[0])
        # V: anything with N elements -- the reference only ever broadcasts it against (batch, N) arrays, so a (1, N)
        # row vector works there too (its test_mf_solver.py:255 passes one)
        if tuple(q_matrix.shape) != (self.n, self.n) or v_vector.numel() != self.n:
            raise ValueError("q_matrix must be (N, N) and v_vector must hold N values")
        v_vector = v_vector.reshape(-1)
        self.ld = ld_of(self.n)
        with torch.cuda.device(self.device):
            self.q = pack(q_matrix.detach().to(self.device), self.ld, self.ld)
            self.v = pack(v_vector.detach().to(self.device), 1, self.ld).reshape(-1)
            # column sums of Q once per staged problem: every run call would otherwise recompute them
            lib = _lib.load()
            self.qsum = torch.empty((self.ld,), dtype=torch.float32, device=self.device)
            ws = torch.empty((max(lib.ccvm_workspace_bytes(_lib.WS_FEEDBACK, 1, self.n), 16),), dtype=torch.uint8,
                             device=self.device)
            _lib.check(lib.ccvm_column_sums(_ptr(self.q), self.n, self.ld, _ptr(self.qsum), _ptr(ws), ws.numel(),
                                            _stream_ptr()), "ccvm_column_sums")


# Kernels whose workgroups wait for each other (column-cluster, column-slab, persistent tile) are deadlock-free only
# while their grid is resident; two such grids launched from different streams of one process can interleave on the
# CUs, and each then sits out its bounded wait (~0.5 s), falls back, and puts the device into its cool-down (ADVICE
# r4).  So run calls that launch one are chained per device ACROSS streams: the enqueue happens under the device's lock
# and, once a second stream has shown up, every such call waits for the event behind the previous one.  A process
# that drives a device from one stream (the normal case) pays a lock and a set lookup per run call.
_exchange_chain = {}  # device index -> {"lock", "streams": raw handles seen, "single": the only one so far (else -1), "event", "last"}


def _exchange_chain_of(device):
    with _cache_lock:
        return _exchange_chain.setdefault(
            device.index, {"lock": threading.Lock(), "streams": set(), "single": -1, "event": None, "last": None})


def _exchange_enter(ch, device, raw):
    """Called under the chain's lock by a run call on stream `raw` that is not the device's only stream so far;
    returns True when the call must leave an event behind it."""
    if raw not in ch["streams"]:
        ch["streams"].add(raw)
        ch["single"] = raw if len(ch["streams"]) == 1 else -1
        if len(ch["streams"]) == 2:
            torch.cuda.synchronize(device)  # the first stream's calls so far left no event behind them
    if len(ch["streams"]) == 1:
        return False
    if ch["event"] is not None and ch["last"] != raw:
        torch.cuda.current_stream(device).wait_event(ch["event"])
    return True


def _exchange_exit(ch, device, raw):
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    ch["event"], ch["last"] = ev, raw


#: device index -> time.monotonic() until which new Trajectories avoid the cluster / slab kernels (set by a time-out
#: recovery, Trajectories.check; $CCVM_AMD_EXCHANGE_COOLDOWN seconds, default 2: a give-up costs ~20 ms since round 6 --
#: ccvm_abi.hip spin_ticks -- so coming back early is cheap; rounds 3-5: 30 s for waits of 0.5-1.2 s)
_exchange_blocked_until = {}

_problem_cache = []  # (weakref(q), weakref(v), q._version, v._version, device index, DeviceProblem, ready event)

#: (kind, batch, N, adam, per-variable S, tuning environment) -> does the plan wait across workgroups (Trajectories._exchange_kernel)
_exchange_kernel_cache = {}
_TUNING_VARS = ("CCVM_AMD_KERNEL", "CCVM_AMD_GEOMETRY", "CCVM_AMD_KS", "CCVM_AMD_SPLIT", "CCVM_AMD_SLAB_CGRP", "CCVM_AMD_SLAB_RG",
                "CCVM_AMD_CLUSTER_SETS", "CCVM_AMD_CLUSTER_HALF", "CCVM_AMD_XCD", "CCVM_AMD_XCD_XC", "CCVM_AMD_PERSIST_WIDE")


def _tuning_env():
    """The environment variables the launch policy reads (ccvm_plan.hip: read_tuning) that can move a shape between
    kernel families, as a hashable snapshot."""
    get = os.environ.get
    return tuple(get(k) for k in _TUNING_VARS)


#: schedule tables of whole runs (ccvm_*_schedule), reused by later runs with the same parameters on the same stream:
#: (device, stream, kind, T, the C parameter struct's bytes, the Adam struct's schedule fields) -> tensor; read-only
_schedule_cache = collections.OrderedDict()


def _schedule_key(device_index, stream, kind, iterations, lo, hi, params, adam):
    """The key of a run's schedule table in ``_schedule_cache``, or None when a parameter is not a plain scalar (such a
    run makes its own table).  Every scalar of the parameter dictionaries is part of the key, whether the table reads it
    or not."""
    def scalars(d):
        out = []
        for k, v in sorted(d.items()):
            if v is None:
                continue
            if torch.is_tensor(v) and v.numel() != 1:
                return None
            try:
                out.append((k, float(v)))
            except (TypeError, ValueError):
                return None
        return tuple(out)

    p, a = scalars(params), (() if adam is None else scalars(adam))
    if p is None or a is None:
        return None
    return (device_index, stream, kind, int(iterations), lo, hi, p, adam is not None, a)


def device_problem(q_matrix, v_vector):
    """``DeviceProblem(q, v)``, reused while the SAME tensor objects are passed unmodified: a solver
    call stages Q for the loop, again for the energy evaluation and again for a post-processor
    (3 x 16 MB at N = 2000), and benchmark loops solve one instance many times.  Identity is the
    Python object (weak reference) plus torch's in-place version counter, so a new tensor that
    happens to reuse the address, or an in-place edit, is never served a stale copy.  (Edits that
    bypass the version counter -- ``q.data.mul_()``, ``set_`` -- are not seen: re-create the tensor.)
    A cached entry served to another stream is recorded on it (``record_stream``), so the caching
    allocator cannot hand its memory out again while kernels of that stream still read it."""
    with _cache_lock:
        return _device_problem_locked(q_matrix, v_vector)


def _device_problem_locked(q_matrix, v_vector):
    import weakref

    dev = gpu_device()
    for entry in _problem_cache:
        wq, wv, qver, vver, index, prob, ready = entry
        if wq() is q_matrix and wv() is v_vector and qver == q_matrix._version and vver == v_vector._version \
                and index == dev.index:
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ready)  # staged on another stream, perhaps
            prob.q.record_stream(cur)
            prob.v.record_stream(cur)
            prob.qsum.record_stream(cur)
            return prob
    prob = DeviceProblem(q_matrix, v_vector)
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))
    _problem_cache.append((weakref.ref(q_matrix), weakref.ref(v_vector), q_matrix._version, v_vector._version,
                           dev.index, prob, ready))
    while len(_problem_cache) > 4:
        _problem_cache.pop(0)
    return prob


def _adam_struct(adam, m, v):
    st = _lib.Adam()
    if adam is None:
        st.enabled = 0
        return st
    st.enabled = 1
    st.add_assign = 1 if adam["add_assign"] else 0
    st.alpha, st.beta1, st.beta2 = float(adam["alpha"]), float(adam["beta1"]), float(adam["beta2"])
    st.m = m.data_ptr()
    st.v = v.data_ptr() if v is not None else None
    return st


class Trajectories:
    """``batch`` independent trajectories of one solver on one GPU.

    kind: "dl" | "mf" | "langevin" (pumped Langevin = langevin with use_pump).
    ``params`` holds the solver's scalars (see the C structs); ``adam`` is an
    ``AdamParameters.to_dict()`` or None.
    """

    _SOLVER_ID = {"dl": _lib.SOLVER_DL, "mf": _lib.SOLVER_MF, "langevin": _lib.SOLVER_LANGEVIN}

    def __init__(self, problem, batch, kind, iterations, params, bounds, noise, adam=None):
        if kind not in self._SOLVER_ID:
            raise ValueError(f"unknown solver kind {kind!r}")
        if kind == "dl" and adam is not None:
            raise ValueError("the DL solver has no Adam variant in the engine")
        self.lib = _lib.load()
        self.p, self.kind, self.b, self.t = problem, kind, int(batch), int(iterations)
        self.n, self.ld, self.device = problem.n, problem.ld, problem.device
        self.rows = rows_of(self.b)
        self.step = 0
        self.s_cols = None  # per-variable saturation (device array), MF / Langevin only
        self.s_full = None  # per-trajectory-and-variable saturation (composed path)
        lo, hi = float(bounds[0]), float(bounds[1])
        with torch.cuda.device(self.device):
            zeros = lambda: torch.zeros((self.rows, self.ld), dtype=torch.float32, device=self.device)
            self.state = {}
            if kind == "dl":
                self.state["c"], self.state["s"] = zeros(), zeros()
                cp = _lib.DlParams()
                cp.pump, cp.dt, cp.noise_ratio = params["pump"], params["dt"], params["noise_ratio"]
                cp.feedback_scale, cp.g = params["feedback_scale"], params["g"]
                cp.pump_rate_flag = 1 if params.get("pump_rate_flag", True) else 0
            elif kind == "mf":
                self.state["mu"], self.state["sigma"], self.state["mu_tilde"] = zeros(), zeros(), zeros()
                self.state["sigma"][: self.b, : self.n] = 0.5  # mf_solver.py:538-540
                cp = _lib.MfParams()
                cp.pump, cp.dt, cp.j = params["pump"], params["dt"], params["j"]
                cp.feedback_scale, cp.g = params["feedback_scale"], params["g"]
                cp.pump_rate_flag = 1 if params.get("pump_rate_flag", True) else 0
                self._set_saturation(cp, params["S"])
            else:
                self.state["c"] = zeros()
                cp = _lib.LangevinParams()
                cp.dt, cp.sigma, cp.feedback_scale = params["dt"], params["sigma"], params["feedback_scale"]
                self._set_saturation(cp, params["S"])
                cp.use_pump = 1 if params.get("use_pump", False) else 0
                cp.pump = float(params.get("pump", 0.0))
                cp.pump_rate_flag = 1 if params.get("pump_rate_flag", True) else 0
            cp.lower, cp.upper = lo, hi
            cp.qsum = problem.qsum.data_ptr()
            self.cparams = cp
            self.adam_m = self.adam_v = None
            if adam is not None:
                self.adam_m = zeros()
                if float(adam["beta2"]) != 1.0:
                    self.adam_v = zeros()
            self.adam = _adam_struct(adam, self.adam_m, self.adam_v)
            # The schedule rows of the whole run, made once (ccvm_hip.h `schedule`):
            # every run call of a persistent path is then one kernel launch instead of a schedule kernel plus the
            # launch (9 of the 657 us of a 20-step call at the headline shape, a quarter of one at N = 100).
            self._schedule = None
            sched_bytes = self.lib.ccvm_schedule_bytes(self._SOLVER_ID[kind], self.t)
            if sched_bytes and os.environ.get("CCVM_AMD_SCHEDULE", "1") != "0":
                # (a table is a function of the scalars below and T alone -- never of Q, the batch or the noise -- and is
                # only ever read: runs with the same parameters on the same stream share one; repeated solves of an
                # instance, the TTS protocol, then skip the allocation and the schedule kernel's launch)
                skey = None
                if self.s_cols is None and self.s_full is None and sched_bytes <= (8 << 20):
                    skey = _schedule_key(self.device.index, _stream_ptr().value, kind, self.t, lo, hi, params, adam)
                with _cache_lock:
                    cached = None if skey is None else _schedule_cache.get(skey)
                    if cached is not None:
                        _schedule_cache.move_to_end(skey)
                if cached is not None:
                    self._schedule = cached
                else:
                    self._schedule = torch.empty((sched_bytes // 4,), dtype=torch.float32, device=self.device)
                    if kind == "dl":
                        rc = self.lib.ccvm_dl_schedule(ctypes.byref(cp), self.t, _ptr(self._schedule), _stream_ptr())
                    else:
                        make = self.lib.ccvm_mf_schedule if kind == "mf" else self.lib.ccvm_langevin_schedule
                        rc = make(ctypes.byref(cp), ctypes.byref(self.adam), self.t, _ptr(self._schedule), _stream_ptr())
                    _lib.check(rc, "ccvm_schedule")
                    if skey is not None:
                        with _cache_lock:
                            _schedule_cache[skey] = self._schedule
                            while len(_schedule_cache) > 16:
                                _schedule_cache.popitem(last=False)
                cp.schedule = self._schedule.data_ptr()
            size_of = self.lib.ccvm_workspace_bytes_cols if self.s_cols is not None else self.lib.ccvm_workspace_bytes
            ws_bytes = size_of(self._SOLVER_ID[kind], self.b, self.n)
            # zeroed: the workspace holds the run's status word (ccvm_status_offset), which run calls never clear
            self.ws = torch.zeros((max(ws_bytes, 16),), dtype=torch.uint8, device=self.device)
            off = self.lib.ccvm_status_offset(self._SOLVER_ID[kind], self.b, self.n)
            self._status = self.ws[off:off + 4] if off != ctypes.c_size_t(-1).value else None
            self._ws_padded = False  # set by the first completed run call (ccvm_hip.h: CCVM_RUN_WS_PADDED)
            # Everything above -- the zero-filled arrays, the schedule kernel -- sits on the stream current NOW; a
            # first run call on another stream waits for this event (nothing else orders that stream's
            # first persistent launch behind the kernel that writes the table it reads)
            self._built = torch.cuda.Event()
            self._built.record()
            self._built_on = _stream_ptr().value
        self.feeder = _NoiseFeeder(noise, self.n, self.b, 2 if kind == "dl" else 1, self.device)
        # Time-out recovery (see check): the state at the last verified point, taken before the first run call that
        # may launch a kernel whose workgroups wait for each other; None while nothing unverified has run.
        self._snap = None
        # True after a time-out: the rest of the run stays on the tile kernel.  A time-out anywhere in this process
        # also keeps NEW trajectories on this device off the exchange kernels for a cool-down period (ADVICE r3:
        # whatever held the GPU -- another process, a CU mask -- would cost each of them its own bounded wait and repeat)
        self.no_exchange = time.monotonic() < _exchange_blocked_until.get(self.device.index, 0.0)
        self.fallbacks = 0         # time-outs recovered so far
        self._runs = 0             # run calls so far; _clean_at: their count at the last clean read of the status word
        self._clean_at = -1
        self._waits = None         # does a run call launch a kernel whose workgroups wait for each other? (asked once)
        self._fixed_args = None    # the run entry point and its constant arguments (made by the first run call)

    def _set_saturation(self, cp, S):
        """Scalar S; a 1-D tensor of length N: per-variable saturation (``s_cols`` of the C structs); a 2-D
        tensor: one saturation per trajectory and variable (``s_full``, composed path)."""
        self.s_full = None
        if is_per_element(S):
            self.s_full = saturation_full(S, self.b, self.n, self.device)
            cp.S, cp.s_cols, cp.s_full = 1.0, None, self.s_full.data_ptr()
        elif is_per_variable(S):
            self.s_cols = saturation_columns(S, self.n, self.device)
            cp.S, cp.s_cols = 1.0, self.s_cols.data_ptr()
        else:
            cp.S, cp.s_cols = float(S), None

    # ------------------------------------------------------------------ #
    def advance(self, nsteps):
        """Run ``nsteps`` more steps (asynchronous on the current stream)."""
        if nsteps < 0 or self.step + nsteps > self.t:
            raise ValueError("step range outside the run")
        per = self.feeder.steps_per_chunk()
        # (the device guard only where another device is current: its set / restore pair costs 2-3 us per call)
        with (contextlib.nullcontext() if torch.cuda.current_device() == self.device.index
              else torch.cuda.device(self.device)):
            if nsteps > 0 and self._snap is None and not self.no_exchange and self._exchange_kernel():
                self._snap = self._snapshot()
            while nsteps > 0:
                k = min(nsteps, per)
                self._run(self.step, k, self.feeder.chunk(k))
                self.step += k
                nsteps -= k

    def arm(self, force=False):
        """Take the recovery snapshot NOW instead of inside the next ``advance`` -- for callers that time that call (two
        device-to-device copies and their allocation are no part of the loop).  ``force``: also when the next run call
        launches no kernel that can time out (``rollback`` then has something to go back to)."""
        if self._snap is None and (force or (not self.no_exchange and self._exchange_kernel())):
            with torch.cuda.device(self.device):
                self._snap = self._snapshot()

    def rollback(self):
        """Back to the snapshot ``check(hold=True)`` kept although the steps since were valid (a caller repeats them:
        bench.py's ranks repeat a timed region TOGETHER when any of them had to).  The kernel policy stays as it is."""
        if self._snap is None:
            raise RuntimeError("rollback needs a snapshot: arm(force=True), then check(hold=True)")
        with torch.cuda.device(self.device):
            self._restore(self._snap)

    def _run(self, step0, k, nz):
        self._runs += 1
        if self._ws_padded:  # this object zero-filled the workspace and only its own run calls have used it since
            nz.flags |= _lib.RUN_WS_PADDED
        if self.no_exchange:
            nz.flags |= _lib.RUN_NO_EXCHANGE
        else:
            # this object's run calls only ever move forward on its workspace (a recovered time-out goes back, and stays
            # off the kernels that care: no_exchange)
            nz.flags |= _lib.RUN_FORWARD
        if self._built is not None:  # the first run call: behind the construction, whatever stream that was on
            if _stream_ptr().value != self._built_on:
                torch.cuda.current_stream(self.device).wait_event(self._built)
            self._built = None
        if self._waits is None:
            self._waits = self._exchange_kernel()
            self._chain = _exchange_chain_of(self.device)
        if self._waits and not self.no_exchange:
            ch = self._chain
            with ch["lock"]:  # (one stream per device, the normal case: a lock and an integer compare per run call)
                raw = _stream_ptr().value or 0
                chained = raw != ch["single"] and _exchange_enter(ch, self.device, raw)
                self._launch(step0, k, nz)
                if chained:
                    _exchange_exit(ch, self.device, raw)
        else:
            self._launch(step0, k, nz)
        self._ws_padded = True

    def _launch(self, step0, k, nz):
        # the arguments that never change over a run -- device pointers, shapes, parameter structs -- are made once
        # (ctypes builds a c_void_p per pointer per call otherwise: a few us of host time inside every caller's timed region)
        fixed = self._fixed_args
        if fixed is None:
            lib, st = self.lib, self.state
            if self.kind == "dl":
                fn, head = lib.ccvm_dl_run, (_ptr(self.p.q), _ptr(self.p.v), _ptr(st["c"]), _ptr(st["s"]))
                mid = (ctypes.byref(self.cparams),)
            elif self.kind == "mf":
                fn, head = lib.ccvm_mf_run, (_ptr(self.p.q), _ptr(self.p.v), _ptr(st["mu"]), _ptr(st["sigma"]), _ptr(st["mu_tilde"]))
                mid = (ctypes.byref(self.cparams), ctypes.byref(self.adam))
            else:
                fn, head = lib.ccvm_langevin_run, (_ptr(self.p.q), _ptr(self.p.v), _ptr(st["c"]))
                mid = (ctypes.byref(self.cparams), ctypes.byref(self.adam))
            fixed = self._fixed_args = (fn, head + (self.b, self.n, self.ld), (self.t,) + mid, (_ptr(self.ws), self.ws.numel()))
        fn, head, mid, ws = fixed
        rc = fn(*head, step0, k, *mid, ctypes.byref(nz), *ws, _stream_ptr())
        _lib.check(rc, f"ccvm_{self.kind}_run")

    # ------------------------------------------------------------------ #
    def clamp(self, name, lo, hi):
        """In-place clamp of a state array; ``hi`` may be a per-variable (1-D) or per-element (batch, N)
        saturation (bounds -hi, +hi)."""
        with torch.cuda.device(self.device):
            if is_per_element(hi):
                sat = saturation_full(hi, self.b, self.n, self.device)
                neg = -sat  # a named tensor: a temporary would be freed (and its memory reused) before the launch
                rc = self.lib.ccvm_clamp_full(_ptr(self.state[name]), self.b, self.n, self.ld, _ptr(neg), _ptr(sat),
                                              _stream_ptr())
            elif is_per_variable(hi):
                cols = saturation_columns(hi, self.n, self.device)
                rc = self.lib.ccvm_clamp_cols(_ptr(self.state[name]), self.b, self.n, self.ld, _ptr(cols),
                                              _stream_ptr())
            else:
                rc = self.lib.ccvm_clamp(_ptr(self.state[name]), self.b, self.n, self.ld, float(lo), float(hi),
                                         _stream_ptr())
            _lib.check(rc, "ccvm_clamp")

    def compact(self, name):
        """Logical (B, N) copy of one state array, on the GPU."""
        self.check()
        with torch.cuda.device(self.device):
            return unpack(self.state[name], self.b, self.n)

    def _exchange_kernel(self):
        """Would a run call of this shape launch a kernel whose workgroups wait for each other (column-cluster,
        column-slab, persistent tile: ccvm_describe_launch under the current tuning environment)?  Asked by every
        advance of a run that has no snapshot yet: memoised per shape and tuning environment (the description is a
        plan evaluation plus string formatting, ~10 us; a solve at the shipped example's size is 530 us of kernel)."""
        key = (self.kind, self.b, self.n, bool(self.adam.enabled), self.s_cols is not None, _tuning_env())
        hit = _exchange_kernel_cache.get(key)
        if hit is not None:
            return hit
        hit = self._exchange_kernel_uncached()
        if len(_exchange_kernel_cache) > 512:
            _exchange_kernel_cache.clear()
        _exchange_kernel_cache[key] = hit
        return hit

    def _exchange_kernel_uncached(self):
        buf = ctypes.create_string_buffer(1024)  # (the description of a cut batch names both parts' plans)
        rc = self.lib.ccvm_describe_launch(self._SOLVER_ID[self.kind], self.b, self.n, 1 if self.adam.enabled else 0,
                                           1 if self.s_cols is not None else 0, buf, 1024)
        return rc == 0 and any(k in buf.value for k in (b"cluster_kernel", b"slab_kernel", b"ptile_kernel"))

    def _snapshot(self):
        """Everything a repeat of the coming steps needs: the state arrays (one device-to-device copy each), the
        step counter and, in replay mode, the host generator's state.  Without an explicit generator that is torch's
        GLOBAL CPU generator (what the reference consumes): a recovery rewinds it to this point, so draws other code
        made from it between the run call and the check are replayed as well -- pass ``NoiseSpec.generator`` to keep
        the run's stream private."""
        gen = self.feeder.spec.generator
        rng = None
        if self.feeder.spec.mode == "replay":
            rng = gen.get_state() if gen is not None else torch.random.get_rng_state()
        return {
            "step": self.step,
            "state": {k: v.clone() for k, v in self.state.items()},
            "adam": [None if t is None else t.clone() for t in (self.adam_m, self.adam_v)],
            "rng": rng,
        }

    def _restore(self, snap):
        for k, v in snap["state"].items():
            self.state[k].copy_(v)
        for dst, src in zip((self.adam_m, self.adam_v), snap["adam"]):
            if dst is not None:
                dst.copy_(src)
        if snap["rng"] is not None:
            gen = self.feeder.spec.generator
            if gen is not None:
                gen.set_state(snap["rng"])
            else:
                torch.random.set_rng_state(snap["rng"])
        self.step = snap["step"]

    def check(self, rerun=True, hold=False):
        """Synchronisation point (4 bytes to the host): did a kernel of this run report a failure through the
        workspace's status word?  The column-cluster and column-slab persistent kernels give up a bounded wait when
        their workgroups cannot all become resident (e.g. another process holding the GPU for a second); the state
        arrays are then invalid.  Recovery: the arrays go back to the last verified point (a snapshot taken before
        the first unverified run call), the rest of the run stays on kernels whose workgroups do not wait for each
        other (CCVM_RUN_NO_EXCHANGE: the per-step tile kernel), and with ``rerun`` the steps since the snapshot are
        repeated there -- same noise, same result up to the summation order -- with a RuntimeWarning instead of an
        error.  Returns True when a time-out was recovered (with ``rerun=False`` the caller repeats the steps:
        ``self.step`` is back at the snapshot).  Without a snapshot (the status word was set by something else) it
        raises.  ``hold``: keep the snapshot of verified steps (see ``rollback``)."""
        if self._status is None:
            if not hold:
                self._snap = None
            return False
        # (no run call since the last clean read: the word cannot have changed -- a solver call verifies at the loop's end,
        # at the timer's stop and in front of the scoring: one 4-byte device-to-host round trip instead of three, ~20 us
        # each against a 530 us solve at the shipped example's size)
        if self._clean_at == self._runs or int(self._status.cpu().view(torch.int32).item()) == 0:
            self._clean_at = self._runs
            if not hold:
                self._snap = None  # verified: the next run call snapshots anew
            return False
        snap = self._snap
        if not hold:
            self._snap = None
        # (hold: the snapshot outlives the recovery too -- the step counter is back at it, the policy is now
        # no_exchange, so advance() never re-arms, and a caller that repeats verified steps, bench.py's ranks when
        # ANOTHER rank times out in a later attempt, still needs something to roll back to: ADVICE r5)
        if snap is None:
            raise _lib.EngineError(
                f"ccvm_{self.kind}_run: a persistent kernel whose workgroups wait for each other (column-cluster, "
                "column-slab or persistent tile kernel) timed out (is another process using this GPU?) and no snapshot "
                "of the trajectories exists; they are invalid -- rerun, or set CCVM_AMD_KERNEL=nocluster (no kernel of "
                "that kind; noptile / noslab switch off one family)"
            )
        import warnings

        reached = self.step
        with torch.cuda.device(self.device):
            self._restore(snap)
            self._status.zero_()
        self.no_exchange = True
        self.fallbacks += 1
        with _cache_lock:
            _exchange_blocked_until[self.device.index] = time.monotonic() + float(
                os.environ.get("CCVM_AMD_EXCHANGE_COOLDOWN", "2"))
        warnings.warn(
            f"ccvm_{self.kind}_run: a persistent kernel timed out waiting for its workgroups (is another process "
            f"using this GPU?); steps {snap['step']}..{reached} are repeated on the per-step tile kernel",
            RuntimeWarning, stacklevel=2)
        if rerun:
            self.advance(reached - snap["step"])
            if int(self._status.cpu().view(torch.int32).item()) != 0:
                raise _lib.EngineError(f"ccvm_{self.kind}_run: status word set by a kernel that never waits")
        return True

    def view(self, name):
        """The logical (B, N) region of one pitched state array as a strided GPU view (no copy)."""
        return self.state[name][: self.b, : self.n]

    def score(self, name, S, scaled_by, lower=0.0, upper=1.0, optimal_value=1.0, clamp=None,
              post_processor=None, rescale_after_pp=False):
        """Everything between the loop and the Solution without leaving the device (ccvm_finalize):
        optional clamp of state ``name`` in place, change of variables with saturation ``S``, optional
        on-device post-processor, energy of every row and the success statistics.  Returns a ``Scored``.

        ``rescale_after_pp``: the DL solver applies the change of variables again AFTER a
        post-processor (dl_solver.py:936-958: the reported variables are the post-processed ones, the
        scored configuration is change_variables of them)."""
        self.check()
        with torch.cuda.device(self.device):
            x = torch.zeros_like(self.state[name])
            return finalize_pitched(self.p, self.state[name], x, self.b, self.n, S, lower, upper, scaled_by,
                                    optimal_value, clamp=clamp, post_processor=post_processor,
                                    rescale_after_pp=rescale_after_pp)


# --------------------------------------------------------------------------- #
# the steps right after the loop
# --------------------------------------------------------------------------- #
def _to_gpu(t):
    dev = gpu_device()
    return t.detach().to(device=dev, dtype=torch.float32), dev


def _rows_of_problem(xg, q_matrix, what):
    """(batch, N) of a (batch, N) variable array that must match the N x N coupling matrix."""
    if xg.ndim != 2 or q_matrix.ndim != 2 or xg.shape[1] != q_matrix.shape[0]:
        raise ValueError(f"{what}: variables of shape {tuple(xg.shape)} do not match a coupling matrix of "
                         f"shape {tuple(q_matrix.shape)}")
    return int(xg.shape[0]), int(xg.shape[1])


def is_per_variable(S):
    """True for a per-variable saturation: the reference accepts S as a 1-D tensor of length N and
    repeats it over the batch (dl_solver.py:843-848, mf_solver.py:834-839, ...)."""
    return torch.is_tensor(S) and S.ndim == 1 and S.numel() > 1


def is_per_element(S):
    """True for a saturation / bound with one value per trajectory AND variable: any 2-D tensor (the
    reference passes a non-1-D tensor S straight through to the elementwise ops, dl_solver.py:843-848)."""
    return torch.is_tensor(S) and S.ndim == 2 and S.numel() > 1


def _full_pitched(t, b, n, dev):
    """A bound broadcast to (b, n) in the pitched device layout."""
    full = torch.as_tensor(t, dtype=torch.float32).detach().to(dev).expand(b, n).contiguous()
    return pack(full, rows_of(b), ld_of(n))


def saturation_full(S, b, n, dev):
    """The pitched device array the ``*_full`` entry points and ``s_full`` take (positive entries)."""
    try:
        fits = tuple(torch.broadcast_shapes(tuple(S.shape), (b, n))) == (b, n)
    except RuntimeError:
        fits = False
    if not fits:
        raise ValueError(f"a 2-D saturation must broadcast to (batch, N) = ({b}, {n}); got {tuple(S.shape)}")
    if not bool((S > 0).all()):
        raise ValueError("every entry of the saturation S must be positive")
    return _full_pitched(S, b, n, dev)


def saturation_columns(S, n, dev):
    """The device array (ld floats, padding 1) the ``*_cols`` entry points and ``s_cols`` fields take."""
    s = S.detach().to(device="cpu", dtype=torch.float32).reshape(-1)
    if s.numel() != n:
        raise ValueError("Tensor S size should be equal to problem size.")
    if not bool((s > 0).all()):
        raise ValueError("every entry of the saturation S must be positive")
    out = torch.ones((ld_of(n),), dtype=torch.float32, device=dev)
    out[:n] = s.to(dev)
    return out


def change_variables(x, S, lower, upper):
    """0.5 * x / S * (upper - lower) + 0.5 * (upper + lower) on the GPU (S scalar or per variable)."""
    lib = _lib.load()
    xg, dev = _to_gpu(x)
    b, n = xg.shape
    with torch.cuda.device(dev):
        xp = pack(xg, rows_of(b), ld_of(n))
        if is_per_element(S):
            sat = saturation_full(S, b, n, dev)  # named: alive until the launch is enqueued
            rc = lib.ccvm_change_variables_full(_ptr(xp), _ptr(xp), b, n, xp.shape[1], _ptr(sat), float(lower),
                                                float(upper), _stream_ptr())
        elif is_per_variable(S):
            cols = saturation_columns(S, n, dev)
            rc = lib.ccvm_change_variables_cols(_ptr(xp), _ptr(xp), b, n, xp.shape[1], _ptr(cols), float(lower),
                                                float(upper), _stream_ptr())
        else:
            rc = lib.ccvm_change_variables(_ptr(xp), _ptr(xp), b, n, xp.shape[1], float(S), float(lower),
                                           float(upper), _stream_ptr())
        _lib.check(rc, "ccvm_change_variables")
        return unpack(xp, b, n).to(x.device)


def clamp(x, lo, hi):
    """clamp(x, lo, hi) on the GPU (fit_to_constraints, torch.clamp semantics).  Scalar bounds; or a
    per-variable saturation ``hi`` (1-D: the bounds are -hi_j, +hi_j); or tensor bounds with one value per
    element (anything else that broadcasts to x's shape, as torch.clamp accepts)."""
    lib = _lib.load()
    xg, dev = _to_gpu(x)
    b, n = xg.shape
    scalar = lambda t: not torch.is_tensor(t) or t.numel() == 1
    with torch.cuda.device(dev):
        xp = pack(xg, rows_of(b), ld_of(n))
        if not is_per_variable(hi) and not (scalar(lo) and scalar(hi)):
            lo_p, hi_p = _full_pitched(lo, b, n, dev), _full_pitched(hi, b, n, dev)  # both alive at the launch
            rc = lib.ccvm_clamp_full(_ptr(xp), b, n, xp.shape[1], _ptr(lo_p), _ptr(hi_p), _stream_ptr())
        elif is_per_variable(hi):
            cols = saturation_columns(hi, n, dev)
            rc = lib.ccvm_clamp_cols(_ptr(xp), b, n, xp.shape[1], _ptr(cols), _stream_ptr())
        else:
            rc = lib.ccvm_clamp(_ptr(xp), b, n, xp.shape[1], float(lo), float(hi), _stream_ptr())
        _lib.check(rc, "ccvm_clamp")
        return unpack(xp, b, n).to(x.device)


def feedback(x, q_matrix, v_vector, in_scale, in_shift, f_q, f_v):
    """f_q * ((x * in_scale + in_shift) @ Q) + f_v * V  -- the bare feedback term."""
    lib = _lib.load()
    xg, dev = _to_gpu(x)
    b, n = _rows_of_problem(xg, q_matrix, "feedback")
    with torch.cuda.device(dev):
        prob = device_problem(q_matrix, v_vector)
        xp = pack(xg, rows_of(b), prob.ld)
        yp = torch.zeros_like(xp)
        ws = torch.empty((max(lib.ccvm_workspace_bytes(_lib.WS_FEEDBACK, b, n), 16),), dtype=torch.uint8, device=dev)
        _lib.check(
            lib.ccvm_feedback(_ptr(prob.q), _ptr(prob.v), _ptr(xp), _ptr(yp), b, n, prob.ld, float(in_scale),
                              float(in_shift), float(f_q), float(f_v), _ptr(ws), ws.numel(), _stream_ptr()),
            "ccvm_feedback",
        )
        return unpack(yp, b, n).to(x.device)


def saturated_feedback(x, q_matrix, v_vector, S, in_scale, in_shift, f_q, f_v):
    """The feedback term of a solver whose input map and output each carry one factor 1 / S:
        (f_q * ((x / S * in_scale + in_shift) @ Q) + f_v * V) / S
    with the scalars given for S = 1.  S: float, or a per-variable 1-D tensor (then x / S and the final
    1 / S are applied per column around the device kernel)."""
    if is_per_variable(S) or is_per_element(S):
        s = S.detach().to(device=x.device, dtype=torch.float32)
        return feedback(x / s, q_matrix, v_vector, in_scale, in_shift, f_q, f_v) / s
    S = float(S)
    return feedback(x, q_matrix, v_vector, in_scale / S, in_shift, f_q / S, f_v / S)


def energy(confs, q_matrix, v_vector, scaled_by=1.0):
    """(1/2 x Q x + V x) * scaled_by for every row of ``confs``."""
    lib = _lib.load()
    xg, dev = _to_gpu(confs)
    b, n = _rows_of_problem(xg, q_matrix, "energy")
    with torch.cuda.device(dev):
        prob = device_problem(q_matrix, v_vector)
        xp = pack(xg, rows_of(b), prob.ld)
        obj = torch.empty((b,), dtype=torch.float32, device=dev)
        ws_bytes = lib.ccvm_workspace_bytes(_lib.WS_ENERGY, b, n)
        ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
        _lib.check(
            lib.ccvm_energy(_ptr(prob.q), _ptr(prob.v), _ptr(xp), b, n, prob.ld, float(scaled_by), _ptr(obj),
                            _ptr(ws), ws.numel(), _stream_ptr()),
            "ccvm_energy",
        )
        return obj.to(confs.device)


@dataclass
class Scored:
    """Result of the device-side finalize: ``variables`` is a strided (B, N) GPU view of the pitched
    array of reported problem variables, ``objective_values`` B floats on the GPU, ``stats`` the
    40-byte ccvm_solution_stats record still on the GPU (``read_stats`` fetches it)."""

    variables: torch.Tensor
    objective_values: torch.Tensor
    stats: torch.Tensor
    pp_seconds: float = 0.0

    def __iter__(self):  # (objective values, post-processing seconds)
        return iter((self.objective_values, self.pp_seconds))


# post-processor defaults of the reference (adam.py:58-66, asgd.py, lbfgs.py, grad_descent.py:58-64)
PP_DEFAULTS = {
    "adam": dict(lr=0.01, eps=1e-8),
    "asgd": dict(lr=0.01, lambd=0.001),
    "lbfgs": dict(iters=1, lr=0.001),
    "grad-descent": dict(iters=10, step=0.1),
}


def read_stats(stats):
    """(best_objective_value, [7 counters], rows, nonfinite) from a device ccvm_solution_stats record."""
    raw = stats.cpu()  # 40 bytes: the only synchronising copy of the finalize
    rec = _lib.SolutionStats.from_buffer_copy(bytes(raw.numpy().tobytes()))
    return float(rec.best_objective_value), [int(c) for c in rec.within], int(rec.rows), int(rec.nonfinite)


def objective_stats(objective_values, optimal_value):
    """ccvm_objective_stats of B objective values that are (or are put) on the GPU; returns the device record."""
    lib = _lib.load()
    dev = gpu_device()
    obj = objective_values.detach().to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        stats = torch.zeros((ctypes.sizeof(_lib.SolutionStats),), dtype=torch.uint8, device=dev)
        _lib.check(lib.ccvm_objective_stats(_ptr(obj), int(obj.numel()), float(optimal_value), _ptr(stats),
                                            _stream_ptr()), "ccvm_objective_stats")
    return stats


def _saturation_args(S, b, n, dev):
    """(scalar S, device s_cols or None, device s_full or None) of a saturation given as a float, a
    per-variable 1-D tensor or a per-element 2-D tensor."""
    if is_per_element(S):
        return 1.0, None, saturation_full(S, b, n, dev)
    if is_per_variable(S):
        return 1.0, saturation_columns(S, n, dev), None
    return float(S.item() if torch.is_tensor(S) else S), None, None


def postprocess_pitched(method, prob, xp, b, n, lower=0.0, upper=1.0, **kw):
    """On-device post-processor in place on the pitched array ``xp``; returns the seconds it took
    (device-synchronised on both sides, like the reference's pp_time brackets the call)."""
    lib = _lib.load()
    dev = prob.device
    if method not in PP_DEFAULTS:
        raise ValueError(f"post-processor {method!r} is not implemented by the HIP engine")
    o = dict(PP_DEFAULTS[method], **{k: v for k, v in kw.items() if v is not None})
    ws_bytes = lib.ccvm_workspace_bytes(_lib.WS_POSTPROCESS, b, n)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    head = (_ptr(prob.q), _ptr(prob.v), _ptr(xp), b, n, prob.ld)
    tail = (float(lower), float(upper), _ptr(ws), ws.numel(), _stream_ptr())
    torch.cuda.synchronize(dev)
    t0 = time.time()
    if method == "grad-descent":
        rc = lib.ccvm_pp_grad_descent(*head, int(o["iters"]), float(o["step"]), *tail)
    elif method == "adam":
        rc = lib.ccvm_pp_adam(*head, float(o["lr"]), float(o["eps"]), *tail)
    elif method == "lbfgs":
        rc = lib.ccvm_pp_lbfgs(*head, int(o["iters"]), float(o["lr"]), *tail)
    else:
        rc = lib.ccvm_pp_asgd(*head, float(o["lr"]), float(o["lambd"]), *tail)
    _lib.check(rc, f"ccvm_pp_{method}")
    torch.cuda.synchronize(dev)
    return time.time() - t0


def finalize_pitched(prob, state, x, b, n, S, lower, upper, scaled_by, optimal_value, clamp=None,
                     post_processor=None, rescale_after_pp=False, pp_options=None):
    """ccvm_finalize on pitched GPU arrays (``state`` in, ``x`` out; both [rows_pad][ld]).

    Without a post-processor this is ONE ABI call: clamp (optional, in place) -> change of variables ->
    energy -> statistics.  With one: change of variables (ccvm_change_variables), the post-processor in
    place on ``x``, then ccvm_finalize on ``x`` (with the second change of variables of the DL quirk when
    ``rescale_after_pp``; the reported variables stay the post-processed ones)."""
    lib = _lib.load()
    dev = prob.device
    with torch.cuda.device(dev):
        s_scalar, s_cols, s_full = _saturation_args(S, b, n, dev)
        obj = torch.empty((b,), dtype=torch.float32, device=dev)
        stats = torch.zeros((ctypes.sizeof(_lib.SolutionStats),), dtype=torch.uint8, device=dev)
        ws = torch.empty((max(lib.ccvm_workspace_bytes(_lib.WS_ENERGY, b, n), 16),), dtype=torch.uint8, device=dev)
        fp = _lib.FinalizeParams()
        fp.S, fp.s_cols = s_scalar, (s_cols.data_ptr() if s_cols is not None else None)
        fp.s_full = s_full.data_ptr() if s_full is not None else None
        fp.lower, fp.upper = float(lower), float(upper)
        fp.scaled_by, fp.optimal_value = float(scaled_by), float(optimal_value)
        fp.clamp, fp.clamp_lo, fp.clamp_hi = 0, 0.0, 0.0
        if clamp is not None:
            fp.clamp, fp.clamp_lo, fp.clamp_hi = 1, float(clamp[0]), float(clamp[1])
        pp_seconds = 0.0
        reported, scored_in, scored_out = x, state, x
        fp.change_variables = 1
        if post_processor:
            if clamp is not None:
                if s_full is not None:
                    neg = -s_full
                    rc = lib.ccvm_clamp_full(_ptr(state), b, n, prob.ld, _ptr(neg), _ptr(s_full), _stream_ptr())
                elif s_cols is not None:
                    rc = lib.ccvm_clamp_cols(_ptr(state), b, n, prob.ld, _ptr(s_cols), _stream_ptr())
                else:
                    rc = lib.ccvm_clamp(_ptr(state), b, n, prob.ld, fp.clamp_lo, fp.clamp_hi, _stream_ptr())
                _lib.check(rc, "ccvm_clamp")
                fp.clamp = 0
            if s_full is not None:
                rc = lib.ccvm_change_variables_full(_ptr(state), _ptr(x), b, n, prob.ld, _ptr(s_full), fp.lower,
                                                    fp.upper, _stream_ptr())
            elif s_cols is None:
                rc = lib.ccvm_change_variables(_ptr(state), _ptr(x), b, n, prob.ld, s_scalar, fp.lower, fp.upper,
                                               _stream_ptr())
            else:
                rc = lib.ccvm_change_variables_cols(_ptr(state), _ptr(x), b, n, prob.ld, _ptr(s_cols), fp.lower,
                                                    fp.upper, _stream_ptr())
            _lib.check(rc, "ccvm_change_variables")
            pp_seconds = postprocess_pitched(post_processor, prob, x, b, n, **(pp_options or {}))
            scored_in = x
            if rescale_after_pp:
                scored_out = torch.zeros_like(x)  # configuration scored; `x` stays the reported variables
            else:
                fp.change_variables = 0
        _lib.check(
            lib.ccvm_finalize(_ptr(prob.q), _ptr(prob.v), _ptr(scored_in), _ptr(scored_out), b, n, prob.ld,
                              ctypes.byref(fp), _ptr(obj), _ptr(stats), _ptr(ws), ws.numel(), _stream_ptr()),
            "ccvm_finalize",
        )
        return Scored(reported[:b, :n], obj, stats, pp_seconds)


def postprocess(method, x, q_matrix, v_vector, lower=0.0, upper=1.0, iters=None, step=None, lr=None,
                eps=None, lambd=None):
    """On-device grad-descent / adam / asgd / lbfgs post-processor; returns (x', seconds)."""
    xg, dev = _to_gpu(x)
    b, n = _rows_of_problem(xg, q_matrix, f"post-processor {method!r}")
    with torch.cuda.device(dev):
        prob = device_problem(q_matrix, v_vector)
        xp = pack(xg, rows_of(b), prob.ld)
        seconds = postprocess_pitched(method, prob, xp, b, n, lower, upper, iters=iters, step=step, lr=lr,
                                      eps=eps, lambd=lambd)
        return unpack(xp, b, n).to(x.device), seconds


def philox_normals(seed, row_offset, step, b, n, two=False):
    """The standard normals PHILOX mode uses at ``step``, as (N, B) tensors."""
    lib = _lib.load()
    dev = gpu_device()
    with torch.cuda.device(dev):
        w0 = torch.empty((n, b), dtype=torch.float32, device=dev)
        w1 = torch.empty((n, b), dtype=torch.float32, device=dev) if two else None
        _lib.check(
            lib.ccvm_philox_normals(int(seed) & 0xFFFFFFFFFFFFFFFF, int(row_offset), int(step), b, n,
                                    _ptr(w0), _ptr(w1), _stream_ptr()),
            "ccvm_philox_normals",
        )
    return (w0, w1) if two else w0
