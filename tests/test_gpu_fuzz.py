"""Seeded random sweep of the engine against the oracle (fused generator): awkward problem sizes
(around every tile / shape boundary), tiny and ragged batches, odd shard offsets, random chunking,
every solver and Adam variant, fused and replayed noise, non-default bounds / g / pump ramp.  One
process, 128 cases by default (CCVM_FUZZ_SEED / CCVM_FUZZ_COUNT select another sweep), each finishes
in well under a second."""
import math
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 96, 100, 127, 128, 129, 144, 200, 255, 256,
         257, 300, 383, 385, 449, 511, 513, 576, 640, 641, 700, 768]
BATCHES = [1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 100, 257, 1000]
ADAMS = [None,
         {"alpha": 0.001, "beta1": 0.9, "beta2": 0.999, "add_assign": False},
         {"alpha": 0.01, "beta1": 0.8, "beta2": 0.99, "add_assign": True},
         {"alpha": 0.002, "beta1": 0.9, "beta2": 1.0, "add_assign": True}]
ATOL_X = 5e-4


def _cases(count=int(os.environ.get("CCVM_FUZZ_COUNT", "128")), seed=int(os.environ.get("CCVM_FUZZ_SEED", "20240607"))):
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        kind = rng.choice(["dl", "mf", "langevin", "pl"])
        n, b, t = rng.choice(SIZES), rng.choice(BATCHES), rng.choice([1, 2, 5, 9])
        if n * n * b > 4.2e8:  # keep the oracle side in seconds
            b = max(1, int(4.2e8 / (n * n)))
        adam = None if kind == "dl" else rng.choice(ADAMS)
        offset = rng.choice([0, 1, 64, 4097, 123456])
        cuts = sorted(rng.sample(range(1, t), min(t - 1, rng.choice([0, 1, 2])))) if t > 1 else []
        replay = rng.random() < 0.3  # parity mode: torch's CPU stream in the reference's order
        bounds = rng.choice([(0.0, 1.0), (0.0, 1.0), (-1.0, 1.0), (-0.5, 2.0), (1.0, 3.0)])  # solution_bounds
        g = rng.choice([None, None, 0.1, 0.002])      # __call__(g=...) of DL / MF
        ramp = rng.random() < 0.7                     # pump_rate_flag
        vec_s = kind != "dl" and rng.random() < 0.35  # per-variable saturation (1-D tensor S)
        force_cluster = rng.random() < 0.4           # CCVM_AMD_KERNEL=cluster: the cluster kernel wherever it exists
        out.append((kind, n, b, t, ADAMS.index(adam), 0 if replay else offset, tuple(cuts), replay, bounds, g, ramp,
                    vec_s, force_cluster))
    return out


def _check_configuration(kind, n, b, t, adam_i, offset, cuts, replay, bounds, g, ramp, vec_s, force_cluster, monkeypatch):
    from ccvm_amd import engine
    from ccvm_amd.workloads import EXAMPLE_PARAMS, scaled_qv
    from oracle import ccvm_oracle as oracle
    from oracle.noise_ref import FusedNoise

    adam = ADAMS[adam_i]
    if force_cluster:
        monkeypatch.setenv("CCVM_AMD_KERNEL", "cluster")
    else:
        monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    q, v, _ = scaled_qv(n, kind)
    p = dict(EXAMPLE_PARAMS[kind])
    if vec_s:
        gen = torch.Generator().manual_seed(n * 131 + b)
        p["S"] = p["S"] * (0.5 + 1.5 * torch.rand(n, generator=gen))
    seed = 0xABCDEF12345 + 7919 * n + b
    if replay:
        noise = engine.NoiseSpec(mode="replay")
        ref_noise = None  # the oracle's default: torch's global CPU stream, as the reference draws it
        torch.manual_seed(seed)
    else:
        noise = engine.NoiseSpec(mode="philox", seed=seed, row_offset=offset)
        ref_noise = FusedNoise(seed, offset, single=kind != "dl")
    prob = engine.DeviceProblem(q, v)
    if kind == "dl":
        g = 0.05 if g is None else g
        traj = engine.Trajectories(prob, b, "dl", t, dict(p, g=g, pump_rate_flag=ramp), bounds, noise)
        c, s = oracle.dl_loop(q, v, b, t, p["pump"], p["dt"], p["noise_ratio"], p["feedback_scale"], g,
                              bounds, ramp, ref_noise)
        want = {"c": c, "s": s}
    elif kind == "mf":
        g = 0.01 if g is None else g
        traj = engine.Trajectories(prob, b, "mf", t, dict(p, g=g, pump_rate_flag=ramp), bounds, noise, adam=adam)
        mu, mu_tilde, sigma = oracle.mf_loop(q, v, b, t, p["pump"], p["dt"], p["j"], p["feedback_scale"], p["S"],
                                             g, bounds, ramp, adam, ref_noise)
        want = {"mu": mu, "sigma": sigma, "mu_tilde": mu_tilde}
    else:
        traj = engine.Trajectories(prob, b, "langevin", t, dict(p, use_pump=kind == "pl", pump_rate_flag=ramp),
                                   bounds, noise, adam=adam)
        if kind == "pl":
            c = oracle.pl_loop(q, v, b, t, p["pump"], p["dt"], p["sigma"], p["feedback_scale"], p["S"], bounds,
                               ramp, adam, ref_noise)
        else:
            c = oracle.langevin_loop(q, v, b, t, p["dt"], p["sigma"], p["feedback_scale"], p["S"], bounds, adam,
                                     ref_noise)
        want = {"c": c}
    if replay:
        torch.manual_seed(seed)  # the engine's feeder consumes the same stream from the same point
    done = 0
    for cut in list(cuts) + [t]:
        traj.advance(cut - done)
        done = cut
    gate = ATOL_X * math.sqrt(max(n, 20) / 20.0)
    for name, ref in want.items():
        got = traj.compact(name).cpu()
        scale = max(1.0, float(ref.abs().max()))
        err = float((got - ref).abs().max())
        assert err <= gate * scale, (f"{kind} N={n} B={b} T={t} adam={adam_i} offset={offset} cuts={cuts} "
                                     f"replay={replay} bounds={bounds} g={g} ramp={ramp} vec_s={vec_s} {name}: {err:.3e}")
    for name, arr in traj.state.items():  # padding stays zero
        assert float(arr[b:].abs().max() if arr.shape[0] > b else 0.0) == 0.0
        assert float(arr[:, n:].abs().max() if arr.shape[1] > n else 0.0) == 0.0


@pytest.mark.parametrize("kind,n,b,t,adam_i,offset,cuts,replay,bounds,g,ramp,vec_s,force_cluster", _cases())
def test_random_configuration_matches_oracle(kind, n, b, t, adam_i, offset, cuts, replay, bounds, g, ramp, vec_s,
                                             force_cluster, monkeypatch):
    _check_configuration(kind, n, b, t, adam_i, offset, cuts, replay, bounds, g, ramp, vec_s, force_cluster, monkeypatch)


# ---- the column-slab small-batch kernel: sizes up to 2048, small batches, forced member widths ------------------
SLAB_SIZES = [257, 300, 511, 512, 513, 700, 767, 769, 1000, 1023, 1025, 1100, 1279, 1300, 1537, 1900, 2047, 2048]
SLAB_BATCHES = [1, 2, 3, 4, 5, 7, 8, 9, 16, 31, 33, 64, 100, 130, 200, 256]


def _slab_cases(count=int(os.environ.get("CCVM_FUZZ_SLAB_COUNT", "64")), seed=int(os.environ.get("CCVM_FUZZ_SEED", "20240607"))):
    rng = random.Random(seed + 1)
    out = []
    for _ in range(count):
        kind = rng.choice(["dl", "mf", "langevin", "pl"])
        n, b, t = rng.choice(SLAB_SIZES), rng.choice(SLAB_BATCHES), rng.choice([1, 2, 5, 9])
        if n * n * b > 4.2e8:
            b = max(1, int(4.2e8 / (n * n)))
        adam = None if kind == "dl" else rng.choice(ADAMS)
        offset = rng.choice([0, 1, 64, 4097, 123456])
        cuts = sorted(rng.sample(range(1, t), min(t - 1, rng.choice([0, 1, 2])))) if t > 1 else []
        replay = rng.random() < 0.3
        vec_s = kind != "dl" and rng.random() < 0.3
        cgrp = rng.choice([0, 0, 1, 2, 4, 8])  # 0: the plan's own choice
        out.append((kind, n, b, t, ADAMS.index(adam), 0 if replay else offset, tuple(cuts), replay, vec_s, cgrp))
    return out


@pytest.mark.parametrize("kind,n,b,t,adam_i,offset,cuts,replay,vec_s,cgrp", _slab_cases())
def test_random_slab_configuration_matches_oracle(kind, n, b, t, adam_i, offset, cuts, replay, vec_s, cgrp, monkeypatch):
    """The same sweep for the slab kernel (CCVM_AMD_KERNEL=slab: wherever a plan exists, also where the default would
    take another path; CCVM_AMD_SLAB_CGRP forces the member width when that width has a plan)."""
    import ctypes

    from ccvm_amd import _lib

    monkeypatch.setenv("CCVM_AMD_KERNEL", "slab")
    if cgrp:
        monkeypatch.setenv("CCVM_AMD_SLAB_CGRP", str(cgrp))
        buf = ctypes.create_string_buffer(512)
        _lib.load().ccvm_describe_launch({"dl": 0, "mf": 1}.get(kind, 2), b, n, 0, 0, buf, 512)
        if b"slab_kernel" not in buf.value:
            monkeypatch.delenv("CCVM_AMD_SLAB_CGRP")  # no plan at that width: the plan's own choice
    _check_configuration(kind, n, b, t, adam_i, offset, cuts, replay, (0.0, 1.0), None, True, vec_s, False,
                         _KeepEnv(monkeypatch))


class _KeepEnv:
    """A monkeypatch stand-in for the shared body: keeps the kernel selection this test made."""

    def __init__(self, mp):
        self.mp = mp

    def setenv(self, *a, **k):
        pass

    def delenv(self, *a, **k):
        pass


# ---- the per-step tile kernel with a forced tile shape (32 x 128, 32 x 64 split-K, 32 x 32 split-K) ---------------
def _tile_cases(count=int(os.environ.get("CCVM_FUZZ_TILE_COUNT", "48")), seed=int(os.environ.get("CCVM_FUZZ_SEED", "20240607"))):
    rng = random.Random(seed + 2)
    out = []
    for _ in range(count):
        kind = rng.choice(["dl", "mf", "langevin", "pl"])
        n = rng.choice([1, 31, 32, 33, 64, 100, 129, 257, 300, 511, 640, 769, 1000, 1025, 1300])
        b = rng.choice([1, 2, 31, 32, 33, 64, 100, 130, 257, 600])
        t = rng.choice([1, 2, 5, 9])
        if n * n * b > 4.2e8:
            b = max(1, int(4.2e8 / (n * n)))
        adam = None if kind == "dl" else rng.choice(ADAMS)
        cuts = sorted(rng.sample(range(1, t), min(t - 1, rng.choice([0, 1, 2])))) if t > 1 else []
        replay = rng.random() < 0.3
        out.append((kind, n, b, t, ADAMS.index(adam), 0 if replay else rng.choice([0, 1, 64, 4097]), tuple(cuts), replay,
                    kind != "dl" and rng.random() < 0.3, rng.choice([1, 2, 4])))
    return out


@pytest.mark.parametrize("kind,n,b,t,adam_i,offset,cuts,replay,vec_s,ks", _tile_cases())
def test_random_tile_shape_configuration_matches_oracle(kind, n, b, t, adam_i, offset, cuts, replay, vec_s, ks, monkeypatch):
    monkeypatch.setenv("CCVM_AMD_KERNEL", "tile")
    monkeypatch.setenv("CCVM_AMD_KS", str(ks))
    _check_configuration(kind, n, b, t, adam_i, offset, cuts, replay, (0.0, 1.0), None, True, vec_s, False,
                         _KeepEnv(monkeypatch))


# ---- the row-owner persistent kernel with its shape knobs forced (rows in use per group, K split) ------------------
def _persist_cases(count=int(os.environ.get("CCVM_FUZZ_PERSIST_COUNT", "48")), seed=int(os.environ.get("CCVM_FUZZ_SEED", "20240607"))):
    rng = random.Random(seed + 3)
    out = []
    for _ in range(count):
        kind = rng.choice(["dl", "mf", "langevin", "pl"])
        n = rng.choice([1, 16, 17, 33, 64, 65, 66, 80, 81, 96, 97, 100, 112, 113, 127, 128, 129, 200, 256])
        b = rng.choice([1, 2, 3, 5, 31, 64, 100, 257, 1000, 3000])
        t = rng.choice([1, 2, 5, 9])
        adam = None if kind == "dl" else rng.choice(ADAMS)
        cuts = sorted(rng.sample(range(1, t), min(t - 1, rng.choice([0, 1, 2])))) if t > 1 else []
        replay = rng.random() < 0.3
        out.append((kind, n, b, t, ADAMS.index(adam), 0 if replay else rng.choice([0, 1, 64, 4097]), tuple(cuts), replay,
                    kind != "dl" and rng.random() < 0.3, rng.choice([1, 2]), rng.choice([0, 0, 2, 4])))
    return out


@pytest.mark.parametrize("kind,n,b,t,adam_i,offset,cuts,replay,vec_s,kh,ru", _persist_cases())
def test_random_row_owner_shape_matches_oracle(kind, n, b, t, adam_i, offset, cuts, replay, vec_s, kh, ru, monkeypatch):
    monkeypatch.delenv("CCVM_AMD_KERNEL", raising=False)
    monkeypatch.setenv("CCVM_AMD_PERSIST_KH", str(kh))
    if ru:
        monkeypatch.setenv("CCVM_AMD_PERSIST_RU", str(ru))
    _check_configuration(kind, n, b, t, adam_i, offset, cuts, replay, (0.0, 1.0), None, True, vec_s, False,
                         _KeepEnv(monkeypatch))
