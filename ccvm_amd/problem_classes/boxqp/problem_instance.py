"""BoxQP problem instance: the data boundary of the hot path.

Mirrors the public surface of the reference's
``ccvm_simulators/problem_classes/boxqp/problem_instance.py`` (constructor
at :23-94, ``load_instance`` :116-224, ``compute_energy`` :226-241,
``scale_coefs`` :243-255): same attribute names, same sign convention
(instance files describe a *maximisation* problem, so Q and V are negated
on load, :183/:188), same error types.  The implementation is new: the
file is parsed in bulk with numpy instead of one tensor write per element,
which is what makes N >= 1000 instances loadable in milliseconds.

``compute_energy`` is the step right after the SDE loop; it runs on the HIP
engine (``ccvm_energy`` in include/ccvm_hip.h) wherever the tensors live.
"""
import enum

import numpy as np
import torch


class DeviceType(enum.Enum):
    """Where the caller's tensors live (reference: problem_instance.py:5-9)."""

    CPU_DEVICE = "cpu"
    CUDA_DEVICE = "cuda"


class InstanceType(enum.Enum):
    """Reference: problem_instance.py:12-16."""

    TUNING = "tuning"
    TEST = "test"


_INSTANCE_TYPES = frozenset(member.value for member in InstanceType)


def _split(line, delimiter):
    return line.rstrip("\n").split(delimiter)


class ProblemInstance:
    """A box-constrained quadratic program  min 1/2 x'Qx + V'x,  l <= x <= u."""

    def __init__(
        self,
        device="cpu",
        instance_type="tuning",
        file_path=None,
        file_delimiter="\t",
        name=None,
        solution_bounds=(0.0, 1.0),
    ):
        if instance_type not in _INSTANCE_TYPES:
            raise ValueError("instance_type must be tuning or test")
        self.instance_type = instance_type
        self.device = device
        self.file_delimiter = file_delimiter

        self.problem_size = None
        self.optimal_sol = None
        self.best_sol = None
        self.optimality = None
        self.sol_time_gb = None
        self.sol_time_bfgs = None
        self.num_frac_values = None
        self.q_matrix = None
        self.v_vector = None
        self.solution_vector = None
        self.scaled_by = 1

        self._custom_name = bool(name)
        if name:
            self.name = name
        if file_path:
            self.file_path = file_path
            self.load_instance(
                device=device,
                instance_type=instance_type,
                file_path=file_path,
                file_delimiter=file_delimiter,
            )
        self.problem_category = "boxqp"
        self.solution_bounds = solution_bounds

    # ------------------------------------------------------------------ #
    @property
    def solution_bounds(self):
        """(lower, upper) inclusive bounds of every variable."""
        return self._solution_bounds

    @solution_bounds.setter
    def solution_bounds(self, bounds):
        if len(bounds) != 2:
            raise ValueError(
                "solution_bounds must be a tuple of size 2, containing the minimum"
                " and maximum bounds (inclusive)"
            )
        if bounds[0] >= bounds[1]:
            raise ValueError(
                "Minimum solution bound must be less than maximum solution bound"
            )
        self._solution_bounds = bounds

    # ------------------------------------------------------------------ #
    def load_instance(
        self, device="cpu", instance_type="tuning", file_path=None, file_delimiter=None
    ):
        """Read a ``.in`` file (format: reference examples/README.md:26-40).

        line 0: size, optimum, best, optimal?, t_gurobi, t_bfgs, seed, n_frac
        line 1: V;  lines 2..N+1: rows of Q;  line N+2 (optional): solution.
        """
        if not file_path and not getattr(self, "file_path", None):
            raise Exception("No file path specified, cannot load instance.")
        if file_path:
            self.file_path = file_path
        if file_delimiter:
            self.file_delimiter = file_delimiter
        path, delim = self.file_path, self.file_delimiter

        # opening the file is outside the guarded block, as in the reference (problem_instance.py:154):
        # a missing file raises FileNotFoundError (its test_problem_instance.py:78-86), only the
        # parsing errors are re-raised as "Error reading instance file"
        stream = open(path, "r")
        try:
            with stream:
                lines = stream.readlines()
            header = _split(lines[0], delim)
            n = int(header[0])
            optimal_sol = float(header[1])
            best_sol = float(header[2])
            optimality = header[3].lower() == "true"
            sol_time_gb = float(header[4])
            sol_time_bfgs = float(header[5])
            num_frac_values = int(header[7])  # header[6] is the generator seed

            v_host = np.array(_split(lines[1], delim)[:n], dtype=np.float64)
            if v_host.shape[0] != n:
                raise ValueError("V has %d entries, expected %d" % (v_host.shape[0], n))
            q_host = np.zeros((n, n), dtype=np.float64)
            for row, line in enumerate(lines[2 : n + 2]):
                fields = _split(line, delim)[:n]
                q_host[row, : len(fields)] = np.array(fields, dtype=np.float64)

            solution_vector = []
            if len(lines) > n + 2:
                solution_vector = [
                    float(tok) for tok in _split(lines[n + 2], delim) if tok != ""
                ]
        except Exception as exc:
            raise Exception("Error reading instance file: " + str(exc))

        # float64 text -> float32, then negate: the file holds a max problem.
        self.q_matrix = (-torch.from_numpy(q_host.astype(np.float32))).to(device)
        self.v_vector = (-torch.from_numpy(v_host.astype(np.float32))).to(device)
        self.device = device
        self.instance_type = instance_type
        self.problem_size = n
        self.optimal_sol = optimal_sol
        self.best_sol = best_sol
        self.optimality = optimality
        self.sol_time_gb = sol_time_gb
        self.sol_time_bfgs = sol_time_bfgs
        self.num_frac_values = num_frac_values
        self.solution_vector = solution_vector
        self.scaled_by = 1
        if not self._custom_name:
            self.name = path.split("/")[-1].split(".")[0]

    @classmethod
    def from_arrays(
        cls,
        q_matrix,
        v_vector,
        device="cpu",
        name="synthetic",
        optimal_sol=1.0,
        best_sol=1.0,
        instance_type="test",
        solution_bounds=(0.0, 1.0),
    ):
        """Build an instance from an already *negated* (minimisation-form) Q, V.

        Used for the synthetic N >= 100 workloads; the
        shipped reference files stop at N = 70.
        """
        inst = cls(
            device=device,
            instance_type=instance_type,
            name=name,
            solution_bounds=solution_bounds,
        )
        q = torch.as_tensor(q_matrix, dtype=torch.float32)
        v = torch.as_tensor(v_vector, dtype=torch.float32)
        if q.ndim != 2 or q.shape[0] != q.shape[1] or v.shape != (q.shape[0],):
            raise ValueError("q_matrix must be (N, N) and v_vector (N,)")
        inst.q_matrix = q.to(device)
        inst.v_vector = v.to(device)
        inst.problem_size = int(q.shape[0])
        inst.optimal_sol = optimal_sol
        inst.best_sol = best_sol
        inst.optimality = False
        inst.sol_time_gb = 0.0
        inst.sol_time_bfgs = 0.0
        inst.num_frac_values = 0
        inst.solution_vector = []
        return inst

    def save_instance(self, file_path, file_delimiter=None, seed=0):
        """Write this instance in the ``.in`` format (undoing the negation and
        any scaling), so synthetic workloads are reproducible as files."""
        delim = file_delimiter or self.file_delimiter
        scale = float(self.scaled_by)
        q = (-self.q_matrix.detach().cpu().double() * scale).numpy()
        v = (-self.v_vector.detach().cpu().double() * scale).numpy()
        with open(file_path, "w") as out:
            header = [
                str(self.problem_size),
                repr(float(self.optimal_sol)),
                repr(float(self.best_sol)),
                str(bool(self.optimality)),
                repr(float(self.sol_time_gb or 0.0)),
                repr(float(self.sol_time_bfgs or 0.0)),
                str(int(seed)),
                str(int(self.num_frac_values or 0)),
            ]
            out.write(delim.join(header) + "\n")
            out.write(delim.join(repr(float(x)) for x in v) + "\n")
            for row in q:
                out.write(delim.join(repr(float(x)) for x in row) + "\n")
            if self.solution_vector:
                out.write(delim.join(repr(float(x)) for x in self.solution_vector) + "\n")

    # ------------------------------------------------------------------ #
    def compute_energy(self, confs):
        """E_b = (1/2 x_b' Q x_b + V' x_b) * scaled_by   (reference :226-241).

        Always evaluated by the HIP engine (``ccvm_energy``); host-resident
        tensors are staged to the GPU and the result comes back on ``confs``'
        device.  There is no CPU implementation in the product.
        """
        from ccvm_amd import engine

        return engine.energy(confs, self.q_matrix, self.v_vector, self.scaled_by)

    def scale_coefs(self, scaling_factor):
        """Divide Q and V by ``scaling_factor``; calls stack (reference :243-255)."""
        self.q_matrix = self.q_matrix / scaling_factor
        self.v_vector = self.v_vector / scaling_factor
        self.scaled_by *= scaling_factor
