"""Access to the committed golden vectors (tests/golden/*.npz + *.json)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# synthetic096 / 300 / 600: the reference on dense synthetic instances (300, 600: the column-cluster kernel's sizes; their
# Q and V are regenerated from the recorded seed instead of stored, and checked against the recorded checksums)
TAGS = ("test020", "tuningH020", "synthetic096", "synthetic300", "synthetic600")


class Golden:
    def __init__(self, tag):
        self.tag = tag
        self.arrays = np.load(os.path.join(GOLDEN_DIR, f"{tag}.npz"))
        with open(os.path.join(GOLDEN_DIR, f"{tag}.json")) as fh:
            self.manifest = json.load(fh)
        self.instance = self.manifest["instance"]
        self.cases = self.manifest["cases"]

    def _generated(self):
        """(q_matrix, v_vector) of a synthetic instance from its recorded seed (see the manifest's recipe)."""
        gen = self.instance["generated"]
        n = self.instance["problem_size"]
        g = torch.Generator().manual_seed(gen["seed"])
        a = torch.randn(n, n, generator=g) * 5
        q = (-((a + a.T) / 2 ** 0.5).double()).float()
        v = (-(torch.randn(n, generator=g) * 17).double()).float()
        qd, vd = q.double(), v.double()
        got = [float(qd.sum()), float(qd.abs().sum()), float(qd[0, 1]), float(qd[-1, -2]), float(vd.sum()),
               float(vd.abs().sum())]
        want = gen["q_checksum"] + gen["v_checksum"]  # (sums: up to the summation order)
        assert all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(got, want)), \
            f"{self.tag}: torch's CPU generator no longer reproduces the instance the fixture was made from"
        return q, v

    def q(self):
        if "q_matrix" not in self.arrays.files:
            return self._generated()[0]
        return torch.from_numpy(self.arrays["q_matrix"].copy())

    def v(self):
        if "v_vector" not in self.arrays.files:
            return self._generated()[1]
        return torch.from_numpy(self.arrays["v_vector"].copy())

    def scaled(self, kind):
        """(Q, V, scaled_by) after scale_coefs(get_scaling_factor(Q)) for a solver kind."""
        mult = 0.2 if kind == "dl" else 0.05
        q, v = self.q(), self.v()
        f = torch.sqrt(torch.sum(torch.abs(q))) * mult
        return q / f, v / f, f

    def out(self, case, field):
        return torch.from_numpy(self.arrays[f"{case}/{field}"].copy())

    def fields(self, case):
        prefix = case + "/"
        return [k[len(prefix):] for k in self.arrays.files if k.startswith(prefix)]


def bounds_cases():
    """Cases of tests/golden/test020_bounds.{npz,json}: test020's instance with non-default
    solution_bounds (made by make_golden.py --only-bounds from the reference)."""
    with open(os.path.join(GOLDEN_DIR, "test020_bounds.json")) as fh:
        return json.load(fh)["cases"]


def bounds_arrays():
    return np.load(os.path.join(GOLDEN_DIR, "test020_bounds.npz"))


def vector_s_cases():
    """Cases of tests/golden/test020_vecS.{npz,json}: per-variable saturation S (1-D tensor of length N),
    made by make_golden.py --only-vector-s from the reference."""
    with open(os.path.join(GOLDEN_DIR, "test020_vecS.json")) as fh:
        return json.load(fh)["cases"]


def vector_s_arrays():
    return np.load(os.path.join(GOLDEN_DIR, "test020_vecS.npz"))


def full_s_cases():
    """tests/golden/test020_fullS.{npz,json}: DLSolver(S=<2-D tensor>), one saturation per trajectory AND
    variable (shapes (B, N), (B, 1), (1, N)), made by make_golden.py --only-full-s from the reference."""
    with open(os.path.join(GOLDEN_DIR, "test020_fullS.json")) as fh:
        return json.load(fh)["cases"]


def full_s_arrays():
    return np.load(os.path.join(GOLDEN_DIR, "test020_fullS.npz"))


def reference_unit_vectors():
    """tests/golden/reference_unit_vectors.json: inputs and expected values of the reference's own
    unit tests (file:line cited inside)."""
    with open(os.path.join(GOLDEN_DIR, "reference_unit_vectors.json")) as fh:
        return json.load(fh)


def asgd_cases():
    """tests/golden/test020_asgd.{npz,json}: post_processor="asgd" through every solver, and under the
    "direct/" prefix the post-processors called on their own (num_iter 1 and 3, custom bounds)."""
    with open(os.path.join(GOLDEN_DIR, "test020_asgd.json")) as fh:
        return json.load(fh)["cases"]


def asgd_arrays():
    return np.load(os.path.join(GOLDEN_DIR, "test020_asgd.npz"))


# thick0300 ... thick1000: the reference at real batch sizes where the engine's kernels change shape (make_golden.py
# --only-thick): objective values of EVERY row, variable arrays for the rows listed in the manifest ("rows_kept")
THICK_TAGS = ("thick0300", "thick0500", "thick0600", "thick0768", "thick1000")


def thick_cases():
    return [(tag, name) for tag in THICK_TAGS for name in golden(tag).cases]


def compare_with_thick(g, case, fields_of, atol_x, atol_obj, label=""):
    """Compare a result (``fields_of(field)`` -> (batch, ...) tensor) with a thick fixture: objective values of every
    row, variables on the kept rows.  Gates: atol_x * max(1, |want|max) on variables, atol_obj relative to the
    magnitude of the objective values."""
    rows = torch.tensor(g.manifest["rows_kept"])
    for field in g.fields(case):
        want = g.out(case, field)
        got = fields_of(field).cpu()
        if field == "objective_values":
            tol = atol_obj * max(1.0, float(want.abs().max()))
        else:
            got = got[rows]
            tol = atol_x * max(1.0, float(want.abs().max()))
        err = float((got - want).abs().max())
        assert err <= tol, f"{label}{g.tag}/{case}/{field}: max abs err {err:.3e} > {tol:.3e}"


_cache = {}


def golden(tag):
    if tag not in _cache:
        _cache[tag] = Golden(tag)
    return _cache[tag]


def all_cases():
    return [(tag, name) for tag in TAGS for name in golden(tag).cases]


def check_noise_checksum(meta, n, b):
    """The fixtures regenerate the reference's noise from the seed; make a drift of the
    torch CPU generator (or of its vectorised normal kernel) visible instead of letting it
    silently break parity."""
    torch.manual_seed(meta["seed"])
    first = torch.randn(n, b)
    s, a, f0, f1 = meta["noise_checksum"]
    assert abs(float(first.double().sum()) - s) <= 1e-3 * max(1.0, abs(s)), "torch CPU randn stream drifted"
    assert abs(float(first.double().abs().sum()) - a) <= 1e-3 * a, "torch CPU randn stream drifted"
    assert abs(float(first[0, 0]) - f0) <= 1e-5 and abs(float(first[-1, -1]) - f1) <= 1e-5
