"""Solver base class: the reference's ``CCVMSolver`` contract over the HIP engine.

API parity targets (reference, relative to ccvm_simulators/solvers/):
  ccvm_solver.py:33-36   device must be "cpu" or "cuda"  -> ValueError
  ccvm_solver.py:65-80   ``parameter_key`` property; subclasses validate an EXACT key set
  ccvm_solver.py:134-150 ``get_scaling_factor`` = sqrt(sum|Q|) * multiplier (0-dim tensor)
  ccvm_solver.py:152-170 ``_method_selector`` binds calculate_drift / calculate_grads /
                         change_variables / fit_to_constraints for "boxqp"
  dl_solver.py:771-999   the ``__call__`` flow every solver shares (device check, parameter
                         lookup, timer, evolution sampling, post-processing, scoring, Solution)

What differs by design: the time loop is not Python.  ``_solve`` / ``_solve_adam`` hand the
whole trajectory to ``ccvm_amd.engine.Trajectories`` (fused HIP kernels); ``device`` says
where the caller's tensors live, the arithmetic always runs on the MI355X.
"""
import enum
import time
from abc import ABC, abstractmethod

import torch

from .. import engine
from ..post_processor.factory import PostProcessorFactory
from ..solution import Solution, fractions_from_counts
from .algorithms import AdamParameters

#: device memory one sampled array's ring may take during evolution sampling
_SAMPLE_RING_BYTES = 256 * 1024 * 1024


def builtin_hook(fn):
    """Marks a ``_<hook>_boxqp`` method as this package's implementation of the hook (see
    ``CCVMSolver._is_builtin``)."""
    fn._ccvm_builtin = True
    return fn


class DeviceType(enum.Enum):
    CPU_DEVICE = "cpu"
    CUDA_DEVICE = "cuda"


class MachineType(enum.Enum):
    """Kept for import compatibility (ccvm_solver.py:15-22); the machine energy/time
    models that consume it are plotting-side bookkeeping and out of scope here."""

    CPU = "cpu"
    GPU = "gpu"
    FPGA = "fpga"
    DL_CCVM = "dl-ccvm"
    MF_CCVM = "mf-ccvm"


def sample_points(iterations, evolution_step_size):
    """Steps after which the reference records a sample: i % k == 0 or the last step
    (dl_solver.py:557-559)."""
    k = int(evolution_step_size)
    pts = [i for i in range(iterations) if i % k == 0 or i + 1 >= iterations]
    return pts


def num_samples(iterations, evolution_step_size):
    """Sample-buffer depth exactly as the reference sizes it (dl_solver.py:866-873)."""
    n = int(iterations / evolution_step_size) + 1
    if iterations % evolution_step_size != 0:
        n += 1
    return n


class CCVMSolver(ABC):
    """Shared machinery of the DL / MF / Langevin / pumped-Langevin solvers."""

    #: exact key set of parameter_key[problem_size]; set by subclasses
    _PARAMETER_KEYS = frozenset()
    #: names of the sampled state arrays (evolution sampling), in file order
    _SAMPLED = ()

    def __init__(self, device):
        if device not in DeviceType._value2member_map_:
            raise ValueError("Given device is not available")
        self.device = device
        self._is_tuned = False
        self._scaling_multiplier = None
        self._parameter_key = None
        self.calculate_drift = None
        self.calculate_grads = None
        self.change_variables = None
        self.fit_to_constraints = None
        #: "fused" (in-kernel counter-based generator, default; "philox" is an alias) or "replay"
        #: (normals drawn from torch's CPU stream exactly as the reference does: parity mode);
        #: None defers to $CCVM_AMD_NOISE.
        self.noise_mode = None
        #: global index of this process's first batch row (multi-GPU sharding)
        self.row_offset = 0
        #: PHILOX key; None draws one from torch's global CPU generator per call
        self.noise_seed = None
        #: replay noise under batch sharding: size of the unsharded batch (solve_sharded sets it)
        self.replay_global_batch = None
        self._traj = None  # device state of the call in flight (set by _new_trajectories)

    # ------------------------------------------------------------------ #
    @property
    def is_tuned(self):
        return self._is_tuned

    @is_tuned.setter
    def is_tuned(self, value):
        self._is_tuned = bool(value)

    @property
    def parameter_key(self):
        return self._parameter_key

    @parameter_key.setter
    def parameter_key(self, parameters):
        expected = set(self._PARAMETER_KEYS)
        for per_size in parameters.values():
            if per_size.keys() != expected:
                raise ValueError(
                    "The parameter key is not valid for this solver. Expected keys: "
                    + str(expected)
                    + " Given keys: "
                    + str(per_size.keys())
                )
        self._parameter_key = parameters
        self._is_tuned = False

    def tune(self, instances=None, post_processor=None, pump_rate_flag=True, g=0.05):
        """Placeholder, as in the reference (dl_solver.py:312-329)."""
        self._is_tuned = True

    def get_scaling_factor(self, q_matrix):
        return torch.sqrt(torch.sum(torch.abs(q_matrix))) * self._scaling_multiplier

    def _method_selector(self, problem_category):
        if problem_category.lower() != "boxqp":
            raise ValueError(
                "The given instance is not a valid problem category."
                f" Given category: {problem_category}"
            )
        self.calculate_drift = self._calculate_drift_boxqp
        self.calculate_grads = self._calculate_grads_boxqp
        self.change_variables = self._change_variables_boxqp
        self.fit_to_constraints = self._fit_to_constraints_boxqp

    # ------------------------------------------------------------------ #
    # hooks: overridable attributes, honoured wherever the reference calls them
    # ------------------------------------------------------------------ #
    #: hooks the time loop calls, by variant (False: _solve, True: _solve_adam); set by subclasses from the
    #: reference's call sites (solvers/composed.py lists them).  A replaced hook that is NOT in the selected
    #: loop's tuple is never looked at, as in the reference (mf_solver.py:561-572 calls calculate_drift only,
    #: :709-716 calculate_grads only).
    _LOOP_HOOKS = {False: (), True: ()}

    def _is_builtin(self, name):
        """True while hook ``name`` is this package's own implementation bound to this solver (the one the
        fused kernels contain).  An attribute assigned by the caller, or a subclass override of
        ``_<name>_boxqp``, is a replaced hook."""
        fn = getattr(self, name, None)
        return getattr(fn, "__self__", None) is self and getattr(fn.__func__, "_ccvm_builtin", False)

    def __copy__(self):
        """A shallow copy (solve_sharded makes one per rank) whose built-in hooks are bound to the COPY: the hook
        attributes are bound methods, and one still bound to the original would read as replaced."""
        new = object.__new__(type(self))
        new.__dict__.update(self.__dict__)
        for name in ("calculate_drift", "calculate_grads", "change_variables", "fit_to_constraints"):
            if self._is_builtin(name):
                setattr(new, name, getattr(new, f"_{name}_boxqp"))
        return new

    def _replaced_on_path(self, adam):
        """Names of the hooks the selected loop calls that are not the built-ins."""
        return [name for name in self._LOOP_HOOKS[bool(adam)] if not self._is_builtin(name)]

    def _call_hook(self, name, *args):
        """Call hook ``name`` from the composed per-step path.  A built-in takes its tensors where they are (the
        GPU) and leaves its result there; a replaced hook gets every tensor argument on the solver's ``device``
        -- where the caller's own tensors live, as in the reference -- and its result is brought back."""
        fn = getattr(self, name)
        if self._is_builtin(name):
            return fn(*args)
        gpu = engine.gpu_device()
        out = fn(*[a.to(self.device) if torch.is_tensor(a) else a for a in args])
        back = lambda t: t.to(device=gpu, dtype=torch.float32) if torch.is_tensor(t) else t
        return tuple(back(o) for o in out) if isinstance(out, (tuple, list)) else back(out)

    def _composed_path(self, adam):
        """The hooks that force the composed per-step path (solvers/composed.py) for this call, after
        warning about it; empty: the fused kernels run."""
        replaced = self._replaced_on_path(adam)
        if replaced:
            import warnings

            warnings.warn(
                f"{type(self).__name__}: {', '.join(replaced)} replaced on this solver; the fused HIP kernels "
                "contain the built-in BoxQP hooks, so this call runs the composed per-step path (Python loop, "
                "hooks called per step): correct but orders of magnitude slower", RuntimeWarning, stacklevel=3)
        return replaced

    @builtin_hook
    def _change_variables_boxqp(self, problem_variables, lower_limit=0, upper_limit=1, S=1):
        return engine.change_variables(problem_variables, S, lower_limit, upper_limit)

    @builtin_hook
    def _fit_to_constraints_boxqp(self, c=None, lower_clamp=None, upper_clamp=None, **kwargs):
        # the reference names the first argument after the solver's state (mf_solver.py:252: mu_tilde);
        # its own tests call it by keyword
        if c is None and len(kwargs) == 1:
            (c,) = kwargs.values()
        return engine.clamp(c, lower_clamp, upper_clamp)

    @abstractmethod
    def _calculate_drift_boxqp(self, *args, **kwargs):
        ...

    @abstractmethod
    def _calculate_grads_boxqp(self, *args, **kwargs):
        ...

    @abstractmethod
    def _solve(self, *args, **kwargs):
        ...

    @abstractmethod
    def _solve_adam(self, *args, **kwargs):
        ...

    # ------------------------------------------------------------------ #
    # shared pieces of __call__
    # ------------------------------------------------------------------ #
    def _bind_instance(self, instance):
        if instance.device != self.device:
            raise ValueError(
                f"The device type of the instance ({instance.device}) and the solver"
                f" ({self.device}) must match."
            )
        self.q_matrix = instance.q_matrix
        self.v_vector = instance.v_vector
        self.solution_bounds = instance.solution_bounds
        self._traj = None
        return instance.problem_size

    def _lookup(self, problem_size, *names):
        try:
            table = self.parameter_key[problem_size]
            return [table[name] for name in names]
        except KeyError as exc:
            raise KeyError(
                f"The parameter '{exc.args[0]}' for the given instance size is not defined."
            ) from exc

    def _broadcast_saturation(self, S, problem_size):
        """The reference repeats a 1-D tensor S of length N over the batch (dl_solver.py:843-848 and the
        same lines of the other solvers).  The engine keeps it as the per-variable vector it is: the
        kernels apply S_j per column (``s_cols`` of the C ABI).  Any other tensor is passed straight through
        by the reference, i.e. a 2-D S is one saturation per trajectory AND variable (``s_full``): elementwise
        kernels after the loop for DL, a composed per-step path (GEMM launch + elementwise launch) for MF /
        Langevin / pumped Langevin, where 1 / S sits inside the loop's GEMM input map per element."""
        if torch.is_tensor(S):
            if S.ndim == 1 and S.size(dim=0) != problem_size and S.numel() != 1:
                raise ValueError("Tensor S size should be equal to problem size.")
            if S.numel() == 1:
                return float(S.item())
            if S.ndim == 1:
                return S.detach().to(device="cpu", dtype=torch.float32)
            if S.ndim == 2:
                return S.detach().to(dtype=torch.float32)
            raise NotImplementedError(
                f"a saturation tensor with {S.ndim} dimensions is not supported; pass a float, a 1-D tensor of "
                "length N or a 2-D tensor that broadcasts to (batch, N)"
            )
        return S

    def _new_trajectories(self, kind, batch_size, iterations, params, adam=None):
        assert not self._replaced_on_path(adam), "the fused kernels contain the built-in hooks"
        problem = engine.device_problem(self.q_matrix, self.v_vector)
        noise = engine.default_noise(self.noise_mode, row_offset=self.row_offset, seed=self.noise_seed,
                                     global_batch=self.replay_global_batch)
        traj = engine.Trajectories(
            problem, batch_size, kind, iterations, params, self.solution_bounds, noise, adam=adam
        )
        self._traj = traj  # the device-side finalize of __call__ scores the state where it lies
        return traj

    def _to_caller(self, traj, name):
        """State array ``name`` for the caller: a strided view of the pitched device array for
        device="cuda" (no copy), one device-to-host copy of the logical (B, N) region for "cpu"."""
        view = traj.view(name)
        return view if self.device == "cuda" else view.to("cpu")

    # ---- the steps right after the loop -------------------------------------------------------- #
    def _device_finalize_ok(self, instance, post_processor):
        """True when clamp / change of variables / post-processor / energy / statistics can run as the
        fused device-side finalize (ccvm_finalize): the hooks and ``compute_energy`` are the built-ins
        and the post-processor is an on-device one.  Otherwise the hook-calling path runs (same kernels,
        one hop per hook)."""
        from ..problem_classes.boxqp.problem_instance import ProblemInstance

        hooks = self._is_builtin("change_variables") and self._is_builtin("fit_to_constraints")
        energy = (
            isinstance(instance, ProblemInstance)
            and type(instance).compute_energy is ProblemInstance.compute_energy
            and "compute_energy" not in vars(instance)
            and instance.q_matrix is getattr(self, "q_matrix", None)
            and instance.v_vector is getattr(self, "v_vector", None)
        )
        pp = not post_processor or (isinstance(post_processor, str) and post_processor.lower() in engine.PP_DEFAULTS)
        return bool(getattr(self, "_traj", None) is not None and hooks and energy and pp
                    and instance.optimal_sol is not None)

    def _score_on_device(self, instance, name, S, lower, upper, post_processor, batch_size,
                         rescale_after_pp=False):
        """ccvm_finalize on state ``name`` of the last trajectories: returns (variables scored or
        post-processed, objective values, pp_time per row, device statistics)."""
        traj = self._traj
        scored = traj.score(
            name, S, float(instance.scaled_by), lower, upper, optimal_value=float(instance.optimal_sol),
            post_processor=post_processor.lower() if post_processor else None, rescale_after_pp=rescale_after_pp,
        )
        best, within, rows, _ = engine.read_stats(scored.stats)
        stats = {"best_objective_value": best, "solution_performance": fractions_from_counts(within, rows),
                 "device_objective_values": scored.objective_values}
        to = (lambda t: t) if self.device == "cuda" else (lambda t: t.to("cpu"))
        return to(scored.variables), to(scored.objective_values), scored.pp_seconds / batch_size, stats

    def _advance_with_samples(self, traj, iterations, evolution_step_size, samples_taken):
        """Run the whole trajectory; copy the sampled state arrays to the host buffers
        ``self.<name>_sample[:, :, k]`` at the reference's sample points."""
        if not evolution_step_size:
            traj.advance(iterations)
            traj.check()  # verify (or recover: see Trajectories.check) BEFORE anything reads or clamps the state
            return
        # Samples are collected in a device-side ring (<= _SAMPLE_RING_BYTES per sampled array, (depth, B, N): a
        # strided device copy per sample point, no synchronisation) and flushed to the host buffers the reference
        # exposes (B, N, depth) whenever the ring is full: one device-to-host copy per flush in ring order, the
        # permutation happens on the host.  (Dense sampling of a large run -- N = B = 1000, every step of 15000 --
        # is 60 GB per array: the reference holds it on the host only, and so does this.)
        # Every flush is a synchronisation point: traj.check verifies the steps since the previous flush (the
        # persistent cluster / slab kernels can time out on a shared GPU); after a time-out the trajectories are back
        # at the previous flush and the samples since then are taken again, on the tile kernel.
        points = sample_points(iterations, evolution_step_size)
        first = samples_taken
        per_sample = traj.b * traj.n * 4
        depth = max(1, min(len(points), _SAMPLE_RING_BYTES // max(per_sample, 1)))
        ring = {
            name: torch.zeros((depth, traj.b, traj.n), dtype=torch.float32, device=traj.device)
            for name in self._SAMPLED
        }
        base = k = 0
        while k < len(points):
            traj.advance(points[k] + 1 - traj.step)
            for name in self._SAMPLED:
                ring[name][k - base].copy_(traj.view(name))
            k += 1
            if k - base == depth or k == len(points):
                if traj.check(rerun=False):  # timed out: traj.step is back at the previous flush
                    k = base
                    continue
                for name in self._SAMPLED:
                    host = ring[name][: k - base].cpu()  # (depth, B, N), contiguous
                    getattr(self, f"{name}_sample")[:, :, first + base:first + k] = host.permute(1, 2, 0)
                base = k
        traj.advance(iterations - traj.step)
        traj.check()

    def _begin_sampling(self, instance, batch_size, problem_size, iterations, evolution_step_size,
                        evolution_file):
        for name in self._SAMPLED:
            setattr(self, f"{name}_sample", None)
        if not evolution_step_size:
            return None, evolution_file
        if evolution_step_size < 1:
            raise ValueError("The evolution step size must be greater than or equal to 1.")
        if evolution_file is None:
            evolution_file = f"./{instance.name}_evolution.txt"
        depth = num_samples(iterations, evolution_step_size)
        for name in self._SAMPLED:
            setattr(
                self,
                f"{name}_sample",
                torch.zeros((batch_size, problem_size, depth), dtype=torch.float, device="cpu"),
            )
        return 0, evolution_file

    #: whether each value in the evolution file is followed by a tab (DL/Langevin) or
    #: values are tab-separated (MF): dl_solver.py:268-272 vs mf_solver.py:285-290
    _TRAILING_TAB = True

    def _append_samples_to_file(self, *samples, evolution_file_object=None, **named):
        """Write (problem_size x num_samples) blocks, one row per line, values rounded to
        4 d.p. (dl_solver.py:252-281)."""
        blocks = list(samples) + [v for k, v in named.items() if k != "evolution_file_object"]
        out = evolution_file_object
        for block in blocks:
            for row in block.tolist():
                cells = [str(round(value, 4)) for value in row]
                if self._TRAILING_TAB:
                    out.write("".join(cell + "\t" for cell in cells))
                else:
                    out.write("\t".join(cells))
                out.write("\n")

    def _write_evolution(self, evolution_file, objval):
        best = torch.argmax(-objval)
        with open(evolution_file, "w") as out:
            self._append_samples_to_file(
                *[getattr(self, f"{name}_sample")[best] for name in self._SAMPLED],
                evolution_file_object=out,
            )

    def _select_algorithm(self, algorithm_parameters):
        if algorithm_parameters is None:
            return None
        if isinstance(algorithm_parameters, AdamParameters):
            return algorithm_parameters.to_dict()
        raise ValueError(f"Solver option type {type(algorithm_parameters)} is not supported.")

    def _sync(self):
        torch.cuda.synchronize(engine.gpu_device())

    def _postprocess(self, post_processor, start_point, batch_size):
        if not post_processor:
            return start_point, 0.0
        pp = PostProcessorFactory.create_postprocessor(post_processor)
        out = pp.postprocess(start_point, self.q_matrix, self.v_vector)
        return out, pp.pp_time / batch_size

    def _solution(self, instance, batch_size, iterations, objval, solve_time, pp_time, variables,
                  evolution_step_size, evolution_file, stats=None):
        if evolution_step_size:
            self._write_evolution(evolution_file, objval)
        stats = stats or {}
        solution = Solution(
            problem_size=instance.problem_size,
            batch_size=batch_size,
            instance_name=instance.name,
            iterations=iterations,
            objective_values=objval,
            solve_time=solve_time,
            pp_time=pp_time,
            optimal_value=instance.optimal_sol,
            best_value=instance.best_sol,
            num_frac_values=instance.num_frac_values,
            solution_vector=instance.solution_vector,
            variables=variables,
            device=self.device,
            solution_performance=stats.get("solution_performance"),
            best_objective_value=stats.get("best_objective_value"),
        )
        #: the objective values where ccvm_finalize left them (GPU): solve_sharded gathers from here
        solution.device_objective_values = stats.get("device_objective_values")
        self._traj = None  # release the device state of this call
        if evolution_step_size:
            solution.evolution_file = evolution_file
        return solution

    def _timer_start(self, kind, problem_size, algorithm_parameters=None):
        """Start of the timed region (dl_solver.py:851).  One-time costs -- context creation,
        code-object load, the first allocation and kernel load of this configuration -- are paid
        before it (engine.prime), so `solve_time` is the loop's time on the first call too."""
        adam = None
        if isinstance(algorithm_parameters, AdamParameters):
            adam = algorithm_parameters.to_dict()
        engine.prime(kind, problem_size, self.batch_size, adam)
        self._sync()
        return time.time()

    def _timer_stop(self, start, batch_size):
        """Per-instance solve time (dl_solver.py:933) -- with the device sync the
        reference forgets."""
        if getattr(self, "_traj", None) is not None:
            # a kernel-side failure of the run (status word) is recovered or raised here at the latest, inside the
            # timed region: the steps a recovery repeats are part of the solve (the loops already verified
            # before they clamped / copied the state, `_advance_with_samples`)
            self._traj.check()
        self._sync()
        elapsed = time.time() - start
        return elapsed / batch_size
