"""Langevin and pumped-Langevin solvers (one amplitude per variable, clamp every step).

API: reference ``ccvm_simulators/solvers/langevin_solver.py`` (constructor :20-25,
``parameter_key`` :96-115, ``_solve`` :368-380, ``_solve_adam`` :437-450, ``__call__``
:563-570) and ``pumped_langevin_solver.py`` (:18-23, :74-93, :232-246, :311-326,
:451-459).  Per step the engine (``ccvm_langevin_run``) evaluates

    G   = -((c (u-l)/(2S) + (u+l)/2) @ Q + V) (u-l)/(2S)
    c  += dt fs G'                         + sigma sqrt(dt) W        (Langevin)
    c  += dt [(-1 + p_i - c^2) c + fs G']  + sigma sqrt(dt) W        (pumped; p_i = pump (i+1)/T)
    c   = clamp(c, -S, S)

where G' = G, or its Adam-preconditioned value in the ``_solve_adam`` variants
(langevin_solver.py:519-540).  Both score (c + S)/(2S).
"""
from .. import engine
from . import composed
from .base import CCVMSolver, builtin_hook

LANGEVIN_SCALING_MULTIPLIER = 0.05


class _LangevinFamily(CCVMSolver):
    _SAMPLED = ("c",)
    _USE_PUMP = False
    # langevin_solver.py:412 + :423, :515 + :549; pumped_langevin_solver.py:287 + :297, :397 + :437
    _LOOP_HOOKS = {False: ("calculate_drift", "fit_to_constraints"), True: ("calculate_grads", "fit_to_constraints")}

    def __init__(self, device, problem_category="boxqp", batch_size=1000):
        super().__init__(device)
        self.batch_size = batch_size
        self._scaling_multiplier = LANGEVIN_SCALING_MULTIPLIER
        self._method_selector(problem_category)

    @builtin_hook
    def _calculate_grads_boxqp(self, c, lower_limit=0, upper_limit=1, S=1):
        ul, up = upper_limit - lower_limit, upper_limit + lower_limit
        k = ul / 2
        return engine.saturated_feedback(
            c, self.q_matrix, self.v_vector, S, in_scale=k, in_shift=up / 2, f_q=-k, f_v=-k
        )

    def _run(self, problem_size, batch_size, device, S, pump, dt, iterations, sigma, pump_rate_flag,
             feedback_scale, evolution_step_size, samples_taken, adam):
        params = dict(
            dt=dt, sigma=sigma, feedback_scale=feedback_scale, S=S, pump=pump,
            use_pump=self._USE_PUMP, pump_rate_flag=pump_rate_flag,
        )
        if self._composed_path(adam):  # a hook this loop calls was replaced: called per step, from Python
            c = composed.langevin_loop(
                self, problem_size, batch_size, S, pump, dt, iterations, sigma, pump_rate_flag, feedback_scale,
                self._USE_PUMP, adam, composed.Sampler(self, iterations, evolution_step_size, samples_taken))
            return c.to(self.device)
        traj = self._new_trajectories("langevin", batch_size, iterations, params, adam=adam)
        self._advance_with_samples(traj, iterations, evolution_step_size, samples_taken)
        return self._to_caller(traj, "c")

    def _finish(self, instance, c, S, iterations, batch_size, solve_time, post_processor,
                evolution_step_size, evolution_file):
        # (c + S) / (2S): langevin_solver.py:722, pumped_langevin_solver.py:604
        stats = None
        if self._device_finalize_ok(instance, post_processor):
            problem_variables, objval, pp_time, stats = self._score_on_device(
                instance, "c", S, 0.0, 1.0, post_processor, batch_size
            )
        else:
            calibrated = engine.change_variables(c, S, 0.0, 1.0)
            problem_variables, pp_time = self._postprocess(post_processor, calibrated, batch_size)
            objval = instance.compute_energy(problem_variables)
        return self._solution(
            instance, batch_size, iterations, objval, solve_time, pp_time,
            {"problem_variables": problem_variables}, evolution_step_size, evolution_file, stats,
        )


class LangevinSolver(_LangevinFamily):
    _PARAMETER_KEYS = frozenset(["dt", "S", "iterations", "sigma", "feedback_scale"])

    @builtin_hook
    def _calculate_drift_boxqp(self, c, lower_limit=0, upper_limit=1, S=1):
        return self._calculate_grads_boxqp(c, lower_limit, upper_limit, S)

    def _solve(
        self,
        problem_size,
        batch_size,
        device,
        S,
        dt,
        iterations,
        sigma,
        feedback_scale,
        evolution_step_size,
        samples_taken,
    ):
        return self._run(problem_size, batch_size, device, S, 0.0, dt, iterations, sigma, False, feedback_scale,
                         evolution_step_size, samples_taken, None)

    def _solve_adam(
        self,
        problem_size,
        batch_size,
        device,
        S,
        dt,
        iterations,
        sigma,
        feedback_scale,
        evolution_step_size,
        samples_taken,
        hyperparameters,
    ):
        return self._run(problem_size, batch_size, device, S, 0.0, dt, iterations, sigma, False, feedback_scale,
                         evolution_step_size, samples_taken, hyperparameters)

    def __call__(
        self,
        instance,
        post_processor=None,
        evolution_step_size=None,
        evolution_file=None,
        algorithm_parameters=None,
    ):
        problem_size = self._bind_instance(instance)
        batch_size, device = self.batch_size, self.device
        dt, S, iterations, sigma, feedback_scale = self._lookup(
            problem_size, "dt", "S", "iterations", "sigma", "feedback_scale"
        )
        S = self._broadcast_saturation(S, problem_size)

        self._select_algorithm(algorithm_parameters)  # validates the type before anything touches the GPU
        start = self._timer_start("langevin", problem_size, algorithm_parameters)
        samples_taken, evolution_file = self._begin_sampling(
            instance, batch_size, problem_size, iterations, evolution_step_size, evolution_file
        )
        adam = self._select_algorithm(algorithm_parameters)
        args = (problem_size, batch_size, device, S, dt, iterations, sigma, feedback_scale,
                evolution_step_size, samples_taken)
        c = self._solve(*args) if adam is None else self._solve_adam(*args, adam)
        solve_time = self._timer_stop(start, batch_size)
        return self._finish(instance, c, S, iterations, batch_size, solve_time, post_processor,
                            evolution_step_size, evolution_file)


class PumpedLangevinSolver(_LangevinFamily):
    _PARAMETER_KEYS = frozenset(["pump", "dt", "S", "iterations", "sigma", "feedback_scale"])
    _USE_PUMP = True

    @builtin_hook
    def _calculate_drift_boxqp(self, c, p, S, feedback_scale):
        lo, hi = self.solution_bounds
        return (-1 + p - c * c) * c + feedback_scale * self._calculate_grads_boxqp(c, lo, hi, S)

    def _solve(
        self,
        problem_size,
        batch_size,
        device,
        S,
        pump,
        dt,
        iterations,
        sigma,
        pump_rate_flag,
        feedback_scale,
        evolution_step_size,
        samples_taken,
    ):
        return self._run(problem_size, batch_size, device, S, pump, dt, iterations, sigma, pump_rate_flag,
                         feedback_scale, evolution_step_size, samples_taken, None)

    def _solve_adam(
        self,
        problem_size,
        batch_size,
        device,
        S,
        pump,
        dt,
        iterations,
        sigma,
        pump_rate_flag,
        feedback_scale,
        evolution_step_size,
        samples_taken,
        hyperparameters,
    ):
        return self._run(problem_size, batch_size, device, S, pump, dt, iterations, sigma, pump_rate_flag,
                         feedback_scale, evolution_step_size, samples_taken, hyperparameters)

    def __call__(
        self,
        instance,
        post_processor=None,
        pump_rate_flag=True,
        evolution_step_size=None,
        evolution_file=None,
        algorithm_parameters=None,
    ):
        problem_size = self._bind_instance(instance)
        batch_size, device = self.batch_size, self.device
        pump, dt, S, iterations, sigma, feedback_scale = self._lookup(
            problem_size, "pump", "dt", "S", "iterations", "sigma", "feedback_scale"
        )
        S = self._broadcast_saturation(S, problem_size)

        self._select_algorithm(algorithm_parameters)  # validates the type before anything touches the GPU
        start = self._timer_start("langevin", problem_size, algorithm_parameters)
        samples_taken, evolution_file = self._begin_sampling(
            instance, batch_size, problem_size, iterations, evolution_step_size, evolution_file
        )
        adam = self._select_algorithm(algorithm_parameters)
        args = (problem_size, batch_size, device, S, pump, dt, iterations, sigma, pump_rate_flag,
                feedback_scale, evolution_step_size, samples_taken)
        c = self._solve(*args) if adam is None else self._solve_adam(*args, adam)
        solve_time = self._timer_stop(start, batch_size)
        return self._finish(instance, c, S, iterations, batch_size, solve_time, post_processor,
                            evolution_step_size, evolution_file)
