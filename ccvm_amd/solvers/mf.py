"""MF-CCVM solver (mean-field measurement-feedback machine).

API: reference ``ccvm_simulators/solvers/mf_solver.py`` -- constructor :20-25,
``parameter_key`` :120-139, ``_solve`` :493-508, ``_solve_adam`` :595-611,
``__call__`` :766-775.  Per step i the engine (``ccvm_mf_run``) evaluates

    j_i   = j e^{-3(i+1)/T},   Wdot = W / sqrt(dt)
    mu~   = clamp(mu + sqrt(1/(4 j_i)) Wdot, -S, S)            (measured amplitude)
    p_i   = pump r + 1 + j_i
    F     = fs (-1/4 ((mu~ a + (u+l)) @ Q) a - V (u-l)/(2S)),  a = (u-l)/S
    mu   += dt [(-(1+j_i) + p_i - g^2 mu^2) mu + F + sqrt(j_i) (sigma - 1/2) Wdot]
    sigma+= dt [2(-(1+j_i) + p_i - 3 g^2 mu^2) sigma - 2 j_i (sigma - 1/2)^2
                + (1+j_i) + 2 g^2 mu^2]

(the Adam variant replaces F by its Adam-preconditioned value, :717-738) and returns
(mu, mu~ of the LAST step, sigma): the scored amplitude is the measured one from before
the last update (:591-593), a reference quirk that is reproduced.
"""
from .. import engine
from . import composed
from .base import CCVMSolver, builtin_hook

MF_SCALING_MULTIPLIER = 0.05


class MFSolver(CCVMSolver):
    _PARAMETER_KEYS = frozenset(["pump", "feedback_scale", "j", "S", "dt", "iterations"])
    _SAMPLED = ("mu", "sigma")
    _TRAILING_TAB = False
    # mf_solver.py:554 + :561 (_solve), :703 + :709 (_solve_adam)
    _LOOP_HOOKS = {False: ("fit_to_constraints", "calculate_drift"), True: ("fit_to_constraints", "calculate_grads")}

    def __init__(self, device, problem_category="boxqp", batch_size=1000):
        super().__init__(device)
        self.batch_size = batch_size
        self._scaling_multiplier = MF_SCALING_MULTIPLIER
        self._method_selector(problem_category)

    # ---- the built-in hooks (the fused kernels contain them; the composed path calls them) ---- #
    @builtin_hook
    def _calculate_grads_boxqp(self, mu_tilde, S, fs, lower_limit=0, upper_limit=1):
        ul, up = upper_limit - lower_limit, upper_limit + lower_limit
        return engine.saturated_feedback(
            mu_tilde, self.q_matrix, self.v_vector, S,
            in_scale=ul, in_shift=up, f_q=-fs * 0.25 * ul, f_v=-fs * ul / 2,
        )

    @builtin_hook
    def _calculate_drift_boxqp(
        self, mu, mu_tilde, sigma, pump, j, g, S, fs, lower_limit=0, upper_limit=1
    ):
        mu_pow = mu * mu
        a0 = -(1 + j) + pump
        drift_mu = (a0 - g**2 * mu_pow) * mu + self._calculate_grads_boxqp(
            mu_tilde, S, fs, lower_limit, upper_limit
        )
        drift_sigma = (
            2 * (a0 - 3 * g**2 * mu_pow) * sigma
            - 2 * j * (sigma - 0.5) ** 2
            + ((1 + j) + 2 * g**2 * mu_pow)
        )
        return drift_mu, drift_sigma

    # ---- the loop ------------------------------------------------------------------ #
    def _run(self, problem_size, batch_size, device, S, pump, dt, iterations, j, feedback_scale,
             pump_rate_flag, g, evolution_step_size, samples_taken, adam):
        params = dict(
            pump=pump, dt=dt, j=j, feedback_scale=feedback_scale, g=g, S=S,
            pump_rate_flag=pump_rate_flag,
        )
        if self._composed_path(adam):  # a hook this loop calls was replaced: called per step, from Python
            out = composed.mf_loop(
                self, problem_size, batch_size, S, pump, dt, iterations, j, feedback_scale, pump_rate_flag, g,
                adam, composed.Sampler(self, iterations, evolution_step_size, samples_taken))
            return tuple(t.to(self.device) for t in out)
        traj = self._new_trajectories("mf", batch_size, iterations, params, adam=adam)
        self._advance_with_samples(traj, iterations, evolution_step_size, samples_taken)
        return tuple(self._to_caller(traj, name) for name in ("mu", "mu_tilde", "sigma"))

    def _solve(
        self,
        problem_size,
        batch_size,
        device,
        S,
        pump,
        dt,
        iterations,
        j,
        feedback_scale,
        pump_rate_flag,
        g,
        evolution_step_size,
        samples_taken,
    ):
        return self._run(problem_size, batch_size, device, S, pump, dt, iterations, j, feedback_scale,
                         pump_rate_flag, g, evolution_step_size, samples_taken, None)

    def _solve_adam(
        self,
        problem_size,
        batch_size,
        device,
        S,
        pump,
        dt,
        iterations,
        j,
        feedback_scale,
        pump_rate_flag,
        g,
        evolution_step_size,
        samples_taken,
        hyperparameters,
    ):
        return self._run(problem_size, batch_size, device, S, pump, dt, iterations, j, feedback_scale,
                         pump_rate_flag, g, evolution_step_size, samples_taken, hyperparameters)

    def __call__(
        self,
        instance,
        post_processor=None,
        g=0.01,
        pump_rate_flag=True,
        evolution_step_size=None,
        evolution_file=None,
        algorithm_parameters=None,
    ):
        problem_size = self._bind_instance(instance)
        batch_size, device = self.batch_size, self.device
        pump, dt, iterations, j, feedback_scale, S = self._lookup(
            problem_size, "pump", "dt", "iterations", "j", "feedback_scale", "S"
        )
        S = self._broadcast_saturation(S, problem_size)
        lo, hi = self.solution_bounds

        self._select_algorithm(algorithm_parameters)  # validates the type before anything touches the GPU
        start = self._timer_start("mf", problem_size, algorithm_parameters)
        samples_taken, evolution_file = self._begin_sampling(
            instance, batch_size, problem_size, iterations, evolution_step_size, evolution_file
        )
        adam = self._select_algorithm(algorithm_parameters)
        args = (problem_size, batch_size, device, S, pump, dt, iterations, j, feedback_scale,
                pump_rate_flag, g, evolution_step_size, samples_taken)
        if adam is None:
            mu, mu_tilde, sigma = self._solve(*args)
        else:
            mu, mu_tilde, sigma = self._solve_adam(*args, adam)
        solve_time = self._timer_stop(start, batch_size)

        stats = None
        if self._device_finalize_ok(instance, post_processor):
            problem_variables, objval, pp_time, stats = self._score_on_device(
                instance, "mu_tilde", S, lo, hi, post_processor, batch_size
            )
        else:
            problem_variables, pp_time = self._postprocess(
                post_processor, self.change_variables(mu_tilde, lo, hi, S), batch_size
            )
            objval = instance.compute_energy(problem_variables)
        return self._solution(
            instance, batch_size, iterations, objval, solve_time, pp_time,
            {"problem_variables": problem_variables, "mu": mu, "sigma": sigma},
            evolution_step_size, evolution_file, stats,
        )
