"""ORACLE -- test infrastructure, not product code.

CPU restatement (torch fp32 on the host, written from the equations) of the reference's
SDE hot path and of the steps right after it, used ONLY by tests/, by
``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg as the checker /
the timed CPU baseline.  Nothing under ``ccvm_amd/`` imports this module.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here
against golden vectors produced by importing the reference itself in the build
container (``tests/golden/make_golden.py``; the reference cannot travel to the GPU box),
and against the known answers in the reference's own unit tests
(``ccvm_simulators/tests/unit/solvers/test_mf_solver.py:63-154``,
``tests/test_solution.py:140-173``, ``tests/unit/problem_classes/test_problem_instance.py``).

Why torch and not numpy/C: the reference *is* a sequence of torch CPU ops; restating it
with the same ops in the same order makes the oracle bit-identical to the reference on
the same torch build (so the pin is exact, not a tolerance), and makes its wall time the
honest "reference CPU path" figure for the bench's cpu_baseline.

Each function cites the reference lines it restates (paths relative to
``ccvm_simulators/`` in the reference tree).

Noise: a source object with ``draw(step, stream, n, b) -> (b, n) tensor``.
``TorchStreamNoise`` consumes a torch CPU generator exactly as the reference does;
``oracle.noise_ref.FusedNoise`` reproduces the engine's fused generator.
"""
import math

import numpy as np
import torch


# --------------------------------------------------------------------------- #
# noise sources
# --------------------------------------------------------------------------- #
class TorchStreamNoise:
    """Standard normals in the reference's stream order: per step and per stream one
    (N, B) draw, used transposed (solvers/dl_solver.py:538-547; Normal(0,1).sample((N,))
    over a batch-shaped distribution consumes the generator like ``torch.randn(N, B)``)."""

    def __init__(self, generator=None):
        self.generator = generator

    def draw(self, step, stream, n, b):
        return torch.randn((n, b), generator=self.generator).transpose(0, 1)


class RecordedNoise:
    """Replays blocks captured earlier: ``blocks[step][stream]`` is an (N, B) tensor."""

    def __init__(self, blocks):
        self.blocks = blocks

    def draw(self, step, stream, n, b):
        return self.blocks[step][stream].transpose(0, 1)


# --------------------------------------------------------------------------- #
# pieces shared by the solvers
# --------------------------------------------------------------------------- #
def scaling_factor(q, multiplier):
    """solvers/ccvm_solver.py:147-149."""
    return torch.sqrt(torch.sum(torch.abs(q))) * multiplier


def change_variables(x, lo, hi, S):
    """solvers/dl_solver.py:233-235 (identical in all four solvers)."""
    return 0.5 * x / S * (hi - lo) + 0.5 * (hi + lo)


def compute_energy(x, q, v, scaled_by=1):
    """problem_classes/boxqp/problem_instance.py:236-241."""
    e1 = torch.einsum("bi, ij, bj -> b", x, q, x) * scaled_by
    e2 = torch.einsum("bi, i -> b", x, v) * scaled_by
    return 0.5 * e1 + e2


def solution_stats(objective_values, optimal_value):
    """solution.py:79-146: best = max(-E); fractions of rows with percentage gap
    (opt - (-E)) * 100 / |-E| within {0.1, 1, 2, 3, 4, 5, 10}, rounded to 4 d.p."""
    found = -objective_values
    best = torch.max(found).item()
    gap = (optimal_value - found) * 100 / torch.abs(found)
    names = ["optimal", "one_percent", "two_percent", "three_percent", "four_percent",
             "five_percent", "ten_percent"]
    thresholds = [0.1, 1, 2, 3, 4, 5, 10]
    ones, zeros = torch.ones(found.size()), torch.zeros(found.size())
    perf = {}
    for name, thr in zip(names, thresholds):
        count = torch.where(gap <= thr, ones, zeros).sum().item()
        perf[name] = round(count / found.size()[0], 4)
    return best, perf


def r99(p):
    """ccvmplotlib/utils/sampleTTSmetric.py:144-153."""
    if p <= 0:
        return math.inf
    if p >= 1:
        return 1.0
    return max(1.0, math.log(1 - 0.99) / math.log(1 - p))


def pp_grad_descent(x, q, v, lo=0.0, hi=1.0, num_iter_main=1000, num_iter_pp=None, step_size=0.1):
    """post_processor/grad_descent.py:58-64."""
    if num_iter_pp is None:
        num_iter_pp = int(num_iter_main * 0.01)
    x = x.clone()
    for _ in range(num_iter_pp):
        grads = torch.einsum("bi,ij -> bj", x, q) + v
        x += -step_size * grads
        x = torch.clamp(x, lo, hi)
    return x


def _pp_gradient(x, q, v):
    """d/dx (1/2 xQx + Vx) = 1/2 (Q + Q') x + V  (autograd of box_qp_model.py:72-74)."""
    return 0.5 * (torch.einsum("bi,ij -> bj", x, q) + torch.einsum("bj,ij -> bi", x, q)) + v


def pp_adam(x, q, v, lo=0.0, hi=1.0, num_iter=1, lr=0.01, eps=1e-8):
    """post_processor/adam.py:58-66 in closed form.  The reference rebuilds the Parameter after every
    step while the optimizer keeps the original one (whose grad stays None afterwards): only the FIRST
    torch.optim.Adam step ever takes effect -- num_iter = 3 equals num_iter = 1 bit for bit (probed).
    From zero moments: m^ = g, v^ = g^2  ->  x <- clamp(x - lr g/(sqrt(g^2) + eps))."""
    if num_iter < 1:
        return x.clone()
    g = _pp_gradient(x, q, v)
    return torch.clamp(x - lr * g / (g.abs() + eps), lo, hi)


def pp_asgd(x, q, v, lo=0.0, hi=1.0, num_iter=1, lr=0.01, lambd=0.001):
    """post_processor/asgd.py in closed form: the first step of torch.optim.ASGD(lr, lambd)
    (eta = lr: param *= 1 - lambd eta; param -= eta grad), then clamp; later iterations are
    no-ops for the same reason as in pp_adam (probed)."""
    if num_iter < 1:
        return x.clone()
    return torch.clamp(x * (1.0 - lambd * lr) - lr * _pp_gradient(x, q, v), lo, hi)


def pp_lbfgs(x, q, v, lo=0.0, hi=1.0, num_iter=1, lr=0.001):
    """post_processor/lbfgs.py in closed form: per row and iteration a NEW torch.optim.LBFGS(lr,
    max_iter=1) -- no curvature history, so its one iteration is the steepest-descent step
    t = lr * min(1, 1/|g|_1) (torch's first-iteration step), skipped below LBFGS's tolerances
    (max|g| <= 1e-7 or g.g < 1e-9); then clamp.  Identical to the reference for num_iter 1 and 3 (probed)."""
    x = x.clone()
    for _ in range(num_iter):
        g = _pp_gradient(x, q, v)
        t = torch.clamp(1.0 / g.abs().sum(1, keepdim=True), max=1.0) * lr
        moves = (g.abs().amax(1, keepdim=True) > 1e-7) & ((g * g).sum(1, keepdim=True) >= 1e-9)
        x = torch.clamp(x - torch.where(moves, t, torch.zeros_like(t)) * g, lo, hi)
    return x


_POST = {"grad-descent": pp_grad_descent, "adam": pp_adam, "asgd": pp_asgd, "lbfgs": pp_lbfgs}


def _adam_update(grads, m, v, i, hp):
    """Shared Adam recurrences (solvers/mf_solver.py:717-738, langevin_solver.py:519-540,
    pumped_langevin_solver.py:401-422).  Returns (preconditioned grads, m, v)."""
    alpha, beta1, beta2 = hp["alpha"], hp["beta1"], hp["beta2"]
    eps = 1e-8
    m = beta1 * m + (1.0 - beta1) * grads
    mhat = m / (1.0 - beta1 ** (i + 1))
    if not beta2 == 1.0:
        v = beta2 * v + (1.0 - beta2) * torch.pow(grads, 2)
        vhat = v / (1.0 - beta2 ** (i + 1))
        upd = alpha * torch.div(mhat, torch.sqrt(vhat) + eps)
    else:
        upd = alpha * mhat
    return (grads + upd if hp["add_assign"] else upd), m, v


def _finish(out, x, q, v, scaled_by, optimal_value, post_processor):
    if post_processor:
        x = _POST[post_processor](x, q, v)
    out["problem_variables"] = x
    out["objective_values"] = compute_energy(x, q, v, scaled_by)
    if optimal_value is not None:
        out["best_objective_value"], out["solution_performance"] = solution_stats(
            out["objective_values"], optimal_value
        )
    return out


# --------------------------------------------------------------------------- #
# DL-CCVM
# --------------------------------------------------------------------------- #
def dl_feedback(y, q, v, lo, hi, Sd):
    """grad_1 + grad_3 of solvers/dl_solver.py:143-154."""
    g1 = 0.25 * torch.einsum("bi,ij -> bj", y * (hi - lo) / Sd + (hi + lo), q) * (hi - lo) / Sd
    g3 = v * (hi - lo) / (2 * Sd)
    return g1 + g3


def dl_drift(c, s, q, v, pump, rate, fs, lo, hi, S=1):
    """solvers/dl_solver.py:117-172.  ``S`` is overridden whenever pump > 1 (:140-141);
    ``_solve`` never passes it, so the default 1 applies otherwise."""
    c2, s2 = torch.pow(c, 2), torch.pow(s, 2)
    if pump > 1:
        S = np.sqrt(pump - 1)
    fsd = fs * (0.5 + rate)
    dc = -fsd * dl_feedback(c, q, v, lo, hi, S) + (-1 + (pump * rate) - c2 - s2) * c
    ds = -fsd * dl_feedback(s, q, v, lo, hi, S) + (-1 - (pump * rate) - c2 - s2) * s
    return dc, ds


def dl_loop(q, v, b, t, pump, dt, noise_ratio, fs, g, bounds, pump_rate_flag=True, noise=None,
            step0=0, nsteps=None, c=None, s=None, on_step=None):
    """The loop body of DLSolver._solve, solvers/dl_solver.py:523-564 (no final clamp)."""
    n = q.shape[0]
    noise = noise or TorchStreamNoise()
    lo, hi = bounds
    c = torch.zeros((b, n), dtype=torch.float) if c is None else c
    s = torch.zeros((b, n), dtype=torch.float) if s is None else s
    rate = 1
    for i in range(step0, step0 + (t - step0 if nsteps is None else nsteps)):
        if pump_rate_flag:
            rate = (i + 1) / t
        ratio = (noise_ratio - 1) * np.exp(-(i + 1) / t * 3) + 1
        dc, ds = dl_drift(c, s, q, v, pump, rate, fs, lo, hi)
        wc = noise.draw(i, 0, n, b) * np.sqrt(dt) * ratio
        ws = noise.draw(i, 1, n, b) * np.sqrt(dt) / ratio
        diff = 2 * g * torch.sqrt(c**2 + s**2 + 0.5)
        c += dt * dc + diff * wc
        s += dt * ds + diff * ws
        if on_step:
            on_step(i, c, s)
    return c, s


def solve_dl(q, v, b, t, pump, dt, noise_ratio, feedback_scale, g=0.05, S=1, bounds=(0.0, 1.0),
             scaled_by=1, optimal_value=None, pump_rate_flag=True, post_processor=None, noise=None):
    """DLSolver.__call__ without the bookkeeping (solvers/dl_solver.py:889-959): loop,
    clamp with the constructor's S (:567), then score.  With a post-processor the change of
    variables is applied before AND after it (:941-958) -- reference quirk, kept."""
    lo, hi = bounds
    c, s = dl_loop(q, v, b, t, pump, dt, noise_ratio, feedback_scale, g, bounds, pump_rate_flag, noise)
    c = torch.clamp(c, -S, S)
    out = {"c": c, "s": s}
    if post_processor:
        x = _POST[post_processor](change_variables(c, lo, hi, S), q, v)
    else:
        x = c
    out["problem_variables"] = x
    out["objective_values"] = compute_energy(change_variables(x, lo, hi, S), q, v, scaled_by)
    if optimal_value is not None:
        out["best_objective_value"], out["solution_performance"] = solution_stats(
            out["objective_values"], optimal_value
        )
    return out


# --------------------------------------------------------------------------- #
# MF-CCVM
# --------------------------------------------------------------------------- #
def mf_grads(mu_tilde, q, v, S, fs, lo, hi):
    """solvers/mf_solver.py:200-233."""
    t1 = -(1 / 4) * torch.einsum("bi,ij -> bj", mu_tilde * (hi - lo) / S + (hi + lo), q) * (hi - lo) / S
    t2 = -v * (hi - lo) / (2 * S)
    return fs * (t1 + t2)


def mf_drift(mu, mu_tilde, sigma, q, v, pump, j, g, S, fs, lo, hi):
    """solvers/mf_solver.py:141-198."""
    mu2 = torch.pow(mu, 2)
    drift_mu = (-(1 + j) + pump - g**2 * mu2) * mu + mf_grads(mu_tilde, q, v, S, fs, lo, hi)
    s1 = 2 * (-(1 + j) + pump - 3 * g**2 * mu2) * sigma
    s2 = -2 * j * (sigma - 0.5).pow(2)
    s3 = (1 + j) + 2 * g**2 * mu2
    return drift_mu, s1 + s2 + s3


def mf_loop(q, v, b, t, pump, dt, j, fs, S, g, bounds, pump_rate_flag=True, adam=None, noise=None,
            on_step=None):
    """MFSolver._solve (solvers/mf_solver.py:549-593) and _solve_adam (:698-764).
    Returns (mu, clamp(mu_tilde of the LAST iteration), sigma)."""
    n = q.shape[0]
    noise = noise or TorchStreamNoise()
    lo, hi = bounds
    mu = torch.zeros((b, n), dtype=torch.float)
    sigma = torch.ones((b, n), dtype=torch.float) * (1 / 2)
    m = torch.zeros((b, n), dtype=torch.float)
    vv = torch.zeros((b, n), dtype=torch.float)
    rate = 1
    mu_tilde = None
    for i in range(t):
        j_i = j * np.exp(-(i + 1) / t * 3.0)
        wdot = noise.draw(i, 0, n, b) / np.sqrt(dt)
        mu_tilde = mu + np.sqrt(1 / (4 * j_i)) * wdot
        mu_tilde_c = torch.clamp(mu_tilde, -S, S)
        if pump_rate_flag:
            rate = (i + 1) / t
        p_i = pump * rate + 1 + j_i
        if adam is None:
            d_mu, d_sigma = mf_drift(mu, mu_tilde_c, sigma, q, v, p_i, j_i, g, S, fs, lo, hi)
            diffusion = np.sqrt(j_i) * (sigma - 0.5) * wdot
            mu += dt * (d_mu + diffusion)
            sigma += dt * d_sigma
        else:
            grads, m, vv = _adam_update(mf_grads(mu_tilde_c, q, v, S, fs, lo, hi), m, vv, i, adam)
            mu2 = torch.pow(mu, 2)
            mu_drift = (-(1 + j_i) + p_i - g**2 * mu2) * mu
            mu_drift += np.sqrt(j_i) * (sigma - 0.5) * wdot
            mu += dt * (grads + mu_drift)
            sd = 2 * (-(1 + j_i) + p_i - 3 * g**2 * mu2) * sigma
            sd += -2 * j_i * (sigma - 0.5).pow(2)
            sd += (1 + j_i) + 2 * g**2 * mu2
            sigma += dt * sd
        if on_step:
            on_step(i, mu, sigma)
    return mu, torch.clamp(mu_tilde, -S, S), sigma


def solve_mf(q, v, b, t, pump, dt, j, feedback_scale, S, g=0.01, bounds=(0.0, 1.0), scaled_by=1,
             optimal_value=None, pump_rate_flag=True, adam=None, post_processor=None, noise=None):
    """MFSolver.__call__ scoring (solvers/mf_solver.py:928-948)."""
    lo, hi = bounds
    mu, mu_tilde, sigma = mf_loop(q, v, b, t, pump, dt, j, feedback_scale, S, g, bounds,
                                  pump_rate_flag, adam, noise)
    out = {"mu": mu, "mu_tilde": mu_tilde, "sigma": sigma}
    return _finish(out, change_variables(mu_tilde, lo, hi, S), q, v, scaled_by, optimal_value,
                   post_processor)


# --------------------------------------------------------------------------- #
# Langevin / pumped Langevin
# --------------------------------------------------------------------------- #
def langevin_grads(c, q, v, lo, hi, S):
    """solvers/langevin_solver.py:131-139 (drift) == :157-166 (grads)."""
    d1 = torch.einsum("bi,ij -> bj", c * (hi - lo) / (2 * S) + (hi + lo) / 2, q)
    return -(d1 + v) * (hi - lo) / (2 * S)


def pl_grads(c, q, v, lo, hi, S):
    """solvers/pumped_langevin_solver.py:133-147."""
    g1 = torch.einsum("bi,ij -> bj", c * (hi - lo) / (2 * S) + (hi + lo) / 2, q) * (hi - lo) / (2 * S)
    g2 = v * (hi - lo) / (2 * S)
    return -g1 - g2


def langevin_loop(q, v, b, t, dt, sigma, fs, S, bounds, adam=None, noise=None, on_step=None):
    """LangevinSolver._solve (solvers/langevin_solver.py:411-433) / _solve_adam (:513-559)."""
    n = q.shape[0]
    noise = noise or TorchStreamNoise()
    lo, hi = bounds
    c = torch.zeros((b, n), dtype=torch.float)
    m = torch.zeros((b, n), dtype=torch.float)
    vv = torch.zeros((b, n), dtype=torch.float)
    for i in range(t):
        grads = langevin_grads(c, q, v, lo, hi, S)
        if adam is not None:
            grads, m, vv = _adam_update(grads, m, vv, i, adam)
        w = noise.draw(i, 0, n, b) * np.sqrt(dt)
        c += dt * fs * grads + sigma * w
        c = torch.clamp(c, -S, S)
        if on_step:
            on_step(i, c)
    return c


def pl_loop(q, v, b, t, pump, dt, sigma, fs, S, bounds, pump_rate_flag=True, adam=None, noise=None,
            on_step=None):
    """PumpedLangevinSolver._solve (solvers/pumped_langevin_solver.py:286-307) /
    _solve_adam (:395-447)."""
    n = q.shape[0]
    noise = noise or TorchStreamNoise()
    lo, hi = bounds
    c = torch.zeros((b, n), dtype=torch.float)
    m = torch.zeros((b, n), dtype=torch.float)
    vv = torch.zeros((b, n), dtype=torch.float)
    for i in range(t):
        p_i = pump * (i + 1) / t if pump_rate_flag else pump
        if adam is None:
            drift = (-1 + p_i - torch.pow(c, 2)) * c + fs * pl_grads(c, q, v, lo, hi, S)
            w = noise.draw(i, 0, n, b) * np.sqrt(dt)
            c += dt * drift + sigma * w
        else:
            grads, m, vv = _adam_update(pl_grads(c, q, v, lo, hi, S), m, vv, i, adam)
            w = noise.draw(i, 0, n, b) * np.sqrt(dt)
            c_pump = (-1 + p_i - torch.pow(c, 2)) * c
            c += dt * (c_pump + fs * grads) + sigma * w
        c = torch.clamp(c, -S, S)
        if on_step:
            on_step(i, c)
    return c


def solve_langevin(q, v, b, t, dt, sigma, feedback_scale, S, bounds=(0.0, 1.0), scaled_by=1,
                   optimal_value=None, adam=None, post_processor=None, noise=None):
    """LangevinSolver.__call__ scoring (solvers/langevin_solver.py:711-726)."""
    c = langevin_loop(q, v, b, t, dt, sigma, feedback_scale, S, bounds, adam, noise)
    return _finish({"c": c}, (c + S) / (2 * S), q, v, scaled_by, optimal_value, post_processor)


def solve_pl(q, v, b, t, pump, dt, sigma, feedback_scale, S, bounds=(0.0, 1.0), scaled_by=1,
             optimal_value=None, pump_rate_flag=True, adam=None, post_processor=None, noise=None):
    """PumpedLangevinSolver.__call__ scoring (solvers/pumped_langevin_solver.py:603-622)."""
    c = pl_loop(q, v, b, t, pump, dt, sigma, feedback_scale, S, bounds, pump_rate_flag, adam, noise)
    return _finish({"c": c}, (c + S) / (2 * S), q, v, scaled_by, optimal_value, post_processor)
