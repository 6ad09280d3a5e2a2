"""DL-CCVM solver (delay-line coherent continuous-variable machine).

API: reference ``ccvm_simulators/solvers/dl_solver.py`` -- constructor :21-27,
``parameter_key`` :96-115, ``_solve`` :468-483, ``__call__`` :771-780.  Per step the
engine (``ccvm_dl_run``) evaluates, for amplitudes c (in-phase) and s (quadrature):

    G(y)  = 1/4 ((y a + (u+l)) @ Q) a + V (u-l)/(2 S_d),  a = (u-l)/S_d
    dc    = -fs (1/2 + r) G(c) + (-1 + p r - c^2 - s^2) c
    ds    = -fs (1/2 + r) G(s) + (-1 - p r - c^2 - s^2) s
    c    += dt dc + 2g sqrt(c^2 + s^2 + 1/2) sqrt(dt) rho_i  W_c
    s    += dt ds + 2g sqrt(c^2 + s^2 + 1/2) sqrt(dt) / rho_i W_s

with r = (i+1)/T (or 1), rho_i = (noise_ratio - 1) e^{-3(i+1)/T} + 1 and
S_d = sqrt(pump - 1) if pump > 1 else 1.  Note the reference's drift never sees
``self.S`` (dl_solver.py:140-141, call at :529-537); ``self.S`` only enters the final
clamp and the change of variables.  That quirk is reproduced.
"""
import numpy as np
import torch

from .. import engine
from . import composed
from .base import CCVMSolver, builtin_hook

DL_SCALING_MULTIPLIER = 0.2


class DLSolver(CCVMSolver):
    _PARAMETER_KEYS = frozenset(["pump", "dt", "iterations", "noise_ratio", "feedback_scale"])
    _SAMPLED = ("c", "s")
    _LOOP_HOOKS = {False: ("calculate_drift",), True: ("calculate_grads",)}  # dl_solver.py:529, :695

    def __init__(self, device, problem_category="boxqp", batch_size=1000, S=1):
        super().__init__(device)
        self.batch_size = batch_size
        self.S = S
        self._scaling_multiplier = DL_SCALING_MULTIPLIER
        self._method_selector(problem_category)

    # ---- the built-in hooks (the fused kernels contain them; the composed path calls them) ---- #
    @builtin_hook
    def _calculate_grads_boxqp(self, c, s, lower_limit=0, upper_limit=1, S=1):
        ul, up = upper_limit - lower_limit, upper_limit + lower_limit
        kw = dict(in_scale=ul, in_shift=up, f_q=-0.25 * ul, f_v=-ul / 2)
        return (
            engine.saturated_feedback(c, self.q_matrix, self.v_vector, S, **kw),
            engine.saturated_feedback(s, self.q_matrix, self.v_vector, S, **kw),
        )

    @builtin_hook
    def _calculate_drift_boxqp(
        self, c, s, pump, rate, feedback_scale=100, lower_limit=0, upper_limit=1, S=1
    ):
        if pump > 1:
            S = np.sqrt(pump - 1)
        c_grads, s_grads = self._calculate_grads_boxqp(c, s, lower_limit, upper_limit, S)
        fsd = feedback_scale * (0.5 + rate)
        r2 = c * c + s * s
        return (
            fsd * c_grads + (-1 + pump * rate - r2) * c,
            fsd * s_grads + (-1 - pump * rate - r2) * s,
        )

    # ---- the loop ------------------------------------------------------------------ #
    def _solve(
        self,
        problem_size,
        batch_size,
        device,
        S,
        pump,
        dt,
        iterations,
        noise_ratio,
        feedback_scale,
        pump_rate_flag,
        g,
        evolution_step_size,
        samples_taken,
    ):
        params = dict(
            pump=pump, dt=dt, noise_ratio=noise_ratio, feedback_scale=feedback_scale, g=g,
            pump_rate_flag=pump_rate_flag,
        )
        if self._composed_path(adam=False):  # calculate_drift was replaced: the hook is called per step
            c, s = composed.dl_loop(
                self, problem_size, batch_size, S, pump, dt, iterations, noise_ratio, feedback_scale,
                pump_rate_flag, g, composed.Sampler(self, iterations, evolution_step_size, samples_taken))
            return c.to(self.device), s.to(self.device)
        traj = self._new_trajectories("dl", batch_size, iterations, params)
        self._advance_with_samples(traj, iterations, evolution_step_size, samples_taken)
        if not self._is_builtin("fit_to_constraints"):  # dl_solver.py:567 through the caller's hook
            self._traj = None  # the scored state is whatever the hook returns, not the device arrays
            return self.fit_to_constraints(self._to_caller(traj, "c"), -S, S), self._to_caller(traj, "s")
        traj.clamp("c", -S, S)  # dl_solver.py:567 -- fit_to_constraints with self.S
        return self._to_caller(traj, "c"), self._to_caller(traj, "s")

    def _solve_adam(self, *args, **kwargs):
        # The reference's DL Adam path cannot be reached: __call__ passes `feedback_scale`
        # to a signature without it (dl_solver.py:908-923 vs :571-586) -> TypeError.
        raise TypeError(
            "DLSolver._solve_adam() is unreachable in the reference (argument mismatch at"
            " dl_solver.py:908-923); the engine does not define a DL Adam variant"
        )

    def __call__(
        self,
        instance,
        post_processor=None,
        pump_rate_flag=True,
        g=0.05,
        evolution_step_size=None,
        evolution_file=None,
        algorithm_parameters=None,
    ):
        problem_size = self._bind_instance(instance)
        batch_size, device = self.batch_size, self.device
        pump, dt, iterations, noise_ratio, feedback_scale = self._lookup(
            problem_size, "pump", "dt", "iterations", "noise_ratio", "feedback_scale"
        )
        S = self._broadcast_saturation(self.S, problem_size)
        lo, hi = self.solution_bounds

        self._select_algorithm(algorithm_parameters)  # validates the type before anything touches the GPU
        start = self._timer_start("dl", problem_size)
        samples_taken, evolution_file = self._begin_sampling(
            instance, batch_size, problem_size, iterations, evolution_step_size, evolution_file
        )
        if self._select_algorithm(algorithm_parameters) is None:
            c, s = self._solve(
                problem_size, batch_size, device, S, pump, dt, iterations, noise_ratio,
                feedback_scale, pump_rate_flag, g, evolution_step_size, samples_taken,
            )
        else:
            c, s = self._solve_adam()
        solve_time = self._timer_stop(start, batch_size)

        # Reference quirk kept: without a post-processor the reported variables are the
        # raw clamped c; with one, change_variables is applied before AND after it
        # (dl_solver.py:936-958).
        stats = None
        if self._device_finalize_ok(instance, post_processor):
            # fused on the device (ccvm_finalize): change of variables [+ post-processor + change of
            # variables again] + energy + success statistics on the pitched state, no host hop
            scored_x, objval, pp_time, stats = self._score_on_device(
                instance, "c", S, lo, hi, post_processor, batch_size, rescale_after_pp=True
            )
            problem_variables = scored_x if post_processor else c
        else:
            if post_processor:
                problem_variables, pp_time = self._postprocess(
                    post_processor, self.change_variables(c, lo, hi, S), batch_size
                )
            else:
                problem_variables, pp_time = c, 0.0
            confs = self.change_variables(problem_variables, lo, hi, S)
            objval = instance.compute_energy(confs)
        return self._solution(
            instance, batch_size, iterations, objval, solve_time, pp_time,
            {"problem_variables": problem_variables, "s": s}, evolution_step_size, evolution_file, stats,
        )
