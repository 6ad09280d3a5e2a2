"""Fits behind the launch policy's time estimates of the per-step tile kernel (developer tool, round 5).

    python tools/fit_tile_model.py [--write]      (anywhere: reads profiles/r05_policy_regret*.jsonl)

Data: every timing of a per-step plan (families T1 / T2 / T4 = 32 x 128 / 64 / 32 tiles) in the regret audits -- the
first pass under round 4's policy and the final one: a forced plan's time does not depend on what the default was.
Model per (solver, tile shape), N >= 300:
    one round  (tiles <= CUs):  t = l0 + l1 N + tiles / CUs * (m0 + m1 N)
    several rounds:             t = ceil(tiles / CUs) * (a N + b + q 1e-6 N^2) + e
weighted least squares on the relative error.  Prints the coefficient table in the form of the generated header's
TILE_FIT (ccvm_amd/csrc/ccvm_plan_model.h), the fit errors, and the errors of the coefficients the LIBRARY carries (parsed
from that header) on the same data.  `--write` puts the refit INTO the header (a re-fit on another box: tools/policy_regret.py
there for the data, then this)."""
import collections
import json
import math
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = [os.path.join(ROOT, "profiles", f) for f in ("r05_policy_regret_first_pass.jsonl", "r05_policy_regret.jsonl")]
CUS = 256


def load():
    data = collections.defaultdict(list)
    for path in FILES:
        for line in open(path):
            r = json.loads(line)
            if r["n"] < 300 or "+" in r["kind"]:
                continue
            if "first_pass" in path and r["n"] >= 1400:
                continue  # (measured before the blocked tile order of grids no XCD rectangle divides: ccvm_abi.hip, set_grid)
            for p in r["plans"]:
                if p.get("us") and p["family"] in ("T1", "T2", "T4"):
                    ks = int(p["family"][1])
                    tiles = -(-r["b"] // 32) * -(-r["n"] // (128 // ks))
                    data[(r["kind"], ks)].append((r["n"], tiles, p["us"]))
    return data


def features(n, tiles):
    if tiles <= CUS:
        return "one", [1.0, n, tiles / CUS, tiles / CUS * n]
    r = math.ceil(tiles / CUS)
    return "many", [r * n, r, 1.0, r * n * n * 1e-6]


def predict(c, n, tiles):
    l0, l1, m0, m1, a, b, e, q = c
    if tiles <= CUS:
        return l0 + l1 * n + tiles / CUS * (m0 + m1 * n)
    return math.ceil(tiles / CUS) * (a * n + b + q * 1e-6 * n * n) + e


def fit(rows):
    coef = {}
    for regime in ("one", "many"):
        sel = [(features(n, t)[1], us) for n, t, us in rows if features(n, t)[0] == regime]
        A, y = np.array([f for f, _ in sel]), np.array([us for _, us in sel])
        w = 1.0 / y
        coef[regime] = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]
    l0, l1, m0, m1 = coef["one"]
    a, b, e, q = coef["many"]
    return (l0, l1, m0, m1, a, b, e, q)


def errors(c, rows):
    err = np.array([(predict(c, n, t) - us) / us for n, t, us in rows])
    return float(np.sqrt((err ** 2).mean())), float(np.abs(err).max())


MODEL = os.path.join(ROOT, "ccvm_amd", "csrc", "ccvm_plan_model.h")


def library_table():
    src = open(MODEL).read()
    block = src[src.index("constexpr TileFit TILE_FIT[3][3]"):]
    block = block[:block.index("};") + 2]
    nums = [float(x) for x in re.findall(r"-?\d+\.\d+", block)]
    assert len(nums) == 72, len(nums)
    return {(kind, ks): tuple(nums[(i * 3 + j) * 8:(i * 3 + j) * 8 + 8])
            for i, kind in enumerate(("dl", "mf", "langevin")) for j, ks in enumerate((1, 2, 4))}


def table_text(data):
    rows = ["    {" + ", ".join("{%.3f, %.5f, %.3f, %.5f, %.5f, %.3f, %.3f, %.3f}" % fit(data[(kind, ks)]) for ks in (1, 2, 4)) + "},"
            for kind in ("dl", "mf", "langevin")]
    return ("constexpr TileFit TILE_FIT[3][3] = {  // [DL, MF, Langevin / pumped Langevin][32 x 128, 32 x 64, 32 x 32]\n"
            + "\n".join(rows) + "\n};")


if __name__ == "__main__":
    import sys

    data, lib = load(), library_table()
    if "--write" in sys.argv:
        # the generated header's TILE_FIT block <- this refit (the measured tables beside it are carried as they are)
        src = open(MODEL).read()
        start = src.index("constexpr TileFit TILE_FIT[3][3]")
        end = src.index("};", start) + 2
        open(MODEL, "w").write(src[:start] + table_text(data) + src[end:])
        print(f"{os.path.relpath(MODEL, ROOT)}: TILE_FIT rewritten from {', '.join(os.path.basename(f) for f in FILES)}")
        lib = library_table()
    print(table_text(data).replace("// [DL", "// refit from the committed audit data  [DL") + "\n")
    print("| solver | tiles | timings | refit: rms / max relative error | the library's coefficients: rms / max |")
    print("|---|---|---|---|---|")
    for kind in ("dl", "mf", "langevin"):
        for ks in (1, 2, 4):
            rows = data[(kind, ks)]
            r1, m1 = errors(fit(rows), rows)
            r2, m2 = errors(lib[(kind, ks)], rows)
            print(f"| {kind} | 32 x {128 // ks} | {len(rows)} | {100 * r1:.1f} % / {100 * m1:.0f} % | {100 * r2:.1f} % / {100 * m2:.0f} % |")
