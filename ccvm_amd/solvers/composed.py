"""Composed per-step path: the time loop with the solver's hooks called from Python.

The fused kernels contain the built-in BoxQP drift / gradient and the per-step clamp.  The reference
keeps ``calculate_drift``, ``calculate_grads``, ``fit_to_constraints`` and ``change_variables`` as
attributes a caller may replace (ccvm_solver.py:152-170; its own tests do, test_mf_solver.py:264-266).
When a hook that the selected loop CALLS has been replaced, the run takes this path instead: one Python
iteration per Euler-Maruyama step, the hooks called at the reference's call sites with the reference's
arguments

    DL  _solve        calculate_drift (dl_solver.py:529), fit_to_constraints after the loop (:567)
    MF  _solve        fit_to_constraints (mf_solver.py:554, :591), calculate_drift (:561)
    MF  _solve_adam   fit_to_constraints (:703, :762), calculate_grads (:709)
    L   _solve        calculate_drift (langevin_solver.py:412), fit_to_constraints (:423)
    L   _solve_adam   calculate_grads (:515), fit_to_constraints (:549)
    PL  _solve        calculate_drift (pumped_langevin_solver.py:287), fit_to_constraints (:297)
    PL  _solve_adam   calculate_grads (:397), fit_to_constraints (:437)

and everything else -- state, noise, the update arithmetic -- on the MI355X: the state arrays are GPU
tensors, the noise is the run's NoiseSpec (the fused generator's normals through ``ccvm_philox_normals``,
i.e. the numbers the fused kernels would have used, or the replayed torch CPU stream), a hook that is
still the built-in runs on the GPU through ``ccvm_feedback`` / ``ccvm_clamp``; a replaced hook receives
its tensors on the solver's ``device`` (where the caller's own tensors live) and its result is brought
back.  Slow by construction (tens of launches per step, a host hop per replaced hook for device="cpu");
the solvers warn (RuntimeWarning) when they take it.  No CPU arithmetic of the engine's own: without
the GPU / library every call below raises ``EngineUnavailable``.
"""
import numpy as np
import torch

from .. import engine


class StepNoise:
    """Standard normals of step i as (batch, N) GPU tensors, from the run's NoiseSpec."""

    def __init__(self, spec, batch, n, streams, device):
        self.spec, self.b, self.n, self.streams, self.device = spec, int(batch), int(n), streams, device

    def draw(self, step):
        spec = self.spec
        if spec.mode == "philox":
            out = engine.philox_normals(spec.seed, spec.row_offset, step, self.b, self.n, two=self.streams == 2)
            out = out if self.streams == 2 else (out,)
            return tuple(w.T for w in out)
        gb, lo = spec.global_batch, int(spec.row_offset)
        blocks = []
        for _ in range(self.streams):  # the reference's order: an (N, B) block per stream, c before s
            if gb is None:
                w = torch.randn((self.n, self.b), generator=spec.generator)
            else:
                w = torch.randn((self.n, gb), generator=spec.generator)[:, lo:lo + self.b]
            blocks.append(w.to(self.device).T)
        return tuple(blocks)


class AdamMoments:
    """The bias-corrected moment recurrences of the ``_solve_adam`` loops (mf_solver.py:717-738)."""

    def __init__(self, hyperparameters, like):
        self.alpha, self.beta1, self.beta2 = (hyperparameters[k] for k in ("alpha", "beta1", "beta2"))
        self.add_assign = bool(hyperparameters["add_assign"])
        self.m = torch.zeros_like(like)
        self.v = torch.zeros_like(like) if self.beta2 != 1.0 else None

    def __call__(self, grads, i):
        self.m = self.beta1 * self.m + (1.0 - self.beta1) * grads
        mhat = self.m / (1.0 - self.beta1 ** (i + 1))
        if self.v is not None:
            self.v = self.beta2 * self.v + (1.0 - self.beta2) * torch.pow(grads, 2)
            vhat = self.v / (1.0 - self.beta2 ** (i + 1))
            step = self.alpha * torch.div(mhat, torch.sqrt(vhat) + 1e-8)
        else:
            step = self.alpha * mhat
        return grads + step if self.add_assign else step


class Sampler:
    """Evolution sampling at the reference's sample points (dl_solver.py:557-564) into the host buffers
    ``solver.<name>_sample`` that ``_begin_sampling`` allocated."""

    def __init__(self, solver, iterations, evolution_step_size, samples_taken):
        self.solver, self.t, self.k, self.taken = solver, iterations, evolution_step_size, samples_taken

    def __call__(self, i, **arrays):
        if self.k and (i % self.k == 0 or i + 1 >= self.t):
            for name, value in arrays.items():
                getattr(self.solver, f"{name}_sample")[:, :, self.taken] = value.to("cpu")
            self.taken += 1


def _setup(solver, batch_size, problem_size, streams):
    device = engine.gpu_device()
    spec = engine.default_noise(solver.noise_mode, row_offset=solver.row_offset, seed=solver.noise_seed,
                                global_batch=solver.replay_global_batch)
    zeros = lambda: torch.zeros((batch_size, problem_size), dtype=torch.float32, device=device)
    return device, StepNoise(spec, batch_size, problem_size, streams, device), zeros


def _on_gpu(S, device):
    return S.to(device) if torch.is_tensor(S) else S


def dl_loop(solver, problem_size, batch_size, S, pump, dt, iterations, noise_ratio, feedback_scale, pump_rate_flag,
            g, sampler):
    device, noise, zeros = _setup(solver, batch_size, problem_size, 2)
    c, s = zeros(), zeros()
    lo, hi = solver.solution_bounds
    rate = 1
    with torch.cuda.device(device):
        for i in range(iterations):
            if pump_rate_flag:
                rate = (i + 1) / iterations
            ratio_i = (noise_ratio - 1) * np.exp(-(i + 1) / iterations * 3) + 1
            dc, ds = solver._call_hook("calculate_drift", c, s, pump, rate, feedback_scale, lo, hi)
            w_c, w_s = noise.draw(i)
            w_c = w_c * np.sqrt(dt) * ratio_i
            w_s = w_s * np.sqrt(dt) / ratio_i
            diffusion = 2 * g * torch.sqrt(c**2 + s**2 + 0.5)
            c += dt * dc + diffusion * w_c
            s += dt * ds + diffusion * w_s
            sampler(i, c=c, s=s)
        c = solver._call_hook("fit_to_constraints", c, -_on_gpu(S, device), _on_gpu(S, device))
    return c, s


def mf_loop(solver, problem_size, batch_size, S, pump, dt, iterations, j, feedback_scale, pump_rate_flag, g,
            hyperparameters, sampler):
    device, noise, zeros = _setup(solver, batch_size, problem_size, 1)
    mu, sigma = zeros(), zeros() + 0.5
    S = _on_gpu(S, device)
    lo, hi = solver.solution_bounds
    adam = AdamMoments(hyperparameters, mu) if hyperparameters is not None else None
    rate, mu_tilde = 1, mu
    with torch.cuda.device(device):
        for i in range(iterations):
            j_i = j * np.exp(-(i + 1) / iterations * 3.0)
            w_dot = noise.draw(i)[0] / np.sqrt(dt)
            mu_tilde = mu + np.sqrt(1 / (4 * j_i)) * w_dot
            measured = solver._call_hook("fit_to_constraints", mu_tilde, -S, S)
            if pump_rate_flag:
                rate = (i + 1) / iterations
            pump_i = pump * rate + 1 + j_i
            if adam is None:
                d_mu, d_sigma = solver._call_hook("calculate_drift", mu, measured, sigma, pump_i, j_i, g, S,
                                                  feedback_scale, lo, hi)
                mu += dt * (d_mu + np.sqrt(j_i) * (sigma - 0.5) * w_dot)
                sigma += dt * d_sigma
            else:
                grads = adam(solver._call_hook("calculate_grads", measured, S, feedback_scale, lo, hi), i)
                mu_pow = torch.pow(mu, 2)
                d_mu = (-(1 + j_i) + pump_i - g**2 * mu_pow) * mu
                d_mu += np.sqrt(j_i) * (sigma - 0.5) * w_dot
                mu += dt * (grads + d_mu)
                d_sigma = 2 * (-(1 + j_i) + pump_i - 3 * g**2 * mu_pow) * sigma
                d_sigma += -2 * j_i * (sigma - 0.5).pow(2)
                d_sigma += (1 + j_i) + 2 * g**2 * mu_pow
                sigma += dt * d_sigma
            sampler(i, mu=mu, sigma=sigma)
        mu_tilde = solver._call_hook("fit_to_constraints", mu_tilde, -S, S)
    return mu, mu_tilde, sigma


def langevin_loop(solver, problem_size, batch_size, S, pump, dt, iterations, sigma, pump_rate_flag, feedback_scale,
                  use_pump, hyperparameters, sampler):
    """Langevin (langevin_solver.py:411-433, :513-559) and pumped Langevin
    (pumped_langevin_solver.py:286-307, :395-447)."""
    device, noise, zeros = _setup(solver, batch_size, problem_size, 1)
    c = zeros()
    S = _on_gpu(S, device)
    lo, hi = solver.solution_bounds
    adam = AdamMoments(hyperparameters, c) if hyperparameters is not None else None
    pump_field = (lambda i: pump * (i + 1) / iterations) if pump_rate_flag else (lambda i: pump)
    with torch.cuda.device(device):
        for i in range(iterations):
            if adam is None and use_pump:
                drift = solver._call_hook("calculate_drift", c, pump_field(i), S, feedback_scale)
            elif adam is None:
                drift = solver._call_hook("calculate_drift", c, lo, hi, S)
            else:
                drift = adam(solver._call_hook("calculate_grads", c, lo, hi, S), i)
            w = noise.draw(i)[0] * np.sqrt(dt)
            if not use_pump:
                c += dt * feedback_scale * drift + sigma * w
            elif adam is None:
                c += dt * drift + sigma * w
            else:
                pumped = torch.einsum("cj,cj -> cj", -1 + pump_field(i) - torch.pow(c, 2), c)
                c += dt * (pumped + feedback_scale * drift) + sigma * w
            c = solver._call_hook("fit_to_constraints", c, -S, S)
            sampler(i, c=c)
    return c
