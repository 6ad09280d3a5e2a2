"""Result record returned by every solver call.

Field order, names and derived quantities follow the reference's
``ccvm_simulators/solution.py`` (:48-63 fields, :65-85 post-init, :87-146
success fractions) so that downstream consumers (metadata JSON, TTS plots)
keep working.  The success fractions feed the "TTS @ 99 %" metric
(``r99`` below restates ``ccvmplotlib/utils/sampleTTSmetric.py:144-153``).
"""
import math
import os
from dataclasses import asdict, dataclass, field

import torch

# (key, gap threshold in percent) in the reference's order (solution.py:128-146)
GAP_THRESHOLDS = (
    ("optimal", 0.1),
    ("one_percent", 1),
    ("two_percent", 2),
    ("three_percent", 3),
    ("four_percent", 4),
    ("five_percent", 5),
    ("ten_percent", 10),
)


def success_fractions(objective_values, optimal_value):
    """Fraction of rows whose percentage gap to ``optimal_value`` is within each
    threshold, rounded to 4 d.p.  ``objective_values`` are minimisation-form
    energies; the reference scores ``-E`` (solution.py:91, 118-122)."""
    found = -objective_values
    gap = (optimal_value - found) * 100 / torch.abs(found)
    count = found.shape[0]
    return {
        key: round(int((gap <= thr).sum().item()) / count, 4)
        for key, thr in GAP_THRESHOLDS
    }


def fractions_from_counts(within, rows):
    """``solution_performance`` from the device counters of ccvm_objective_stats: round(count / rows, 4)
    per threshold, keys in the reference's order."""
    return {key: round(int(cnt) / int(rows), 4) for (key, _), cnt in zip(GAP_THRESHOLDS, within)}


def r99(p_success):
    """Runs needed for 99 % success probability: max(1, ln(0.01)/ln(1-p))."""
    if p_success <= 0.0:
        return math.inf
    if p_success >= 1.0:
        return 1.0
    return max(1.0, math.log(1.0 - 0.99) / math.log(1.0 - p_success))


@dataclass
class Solution:
    problem_size: int
    batch_size: int
    instance_name: str
    iterations: int
    objective_values: torch.Tensor = field(repr=False)
    solve_time: float
    pp_time: float
    optimal_value: float
    best_value: float
    num_frac_values: int
    solution_vector: list
    variables: dict = field(repr=False)
    evolution_file: str = None
    device: str = field(default="cpu", repr=False)
    solution_performance: dict = None
    best_objective_value: float = None

    def __post_init__(self):
        target = torch.device(self.device)
        for key, value in self.variables.items():
            if torch.is_tensor(value) and value.device.type != target.type:
                self.variables[key] = value.to(self.device)
        if (
            torch.is_tensor(self.objective_values)
            and self.objective_values.device.type != target.type
        ):
            self.objective_values = self.objective_values.to(self.device)
        # The solvers pass both, computed on the device by ccvm_finalize (ccvm_objective_stats: the same
        # fp32 arithmetic as below); anyone else constructing a Solution gets them computed here.
        if self.best_objective_value is None:
            self.best_objective_value = torch.max(-self.objective_values).item()
        if self.solution_performance is None:
            self.get_solution_stats()

    def get_solution_stats(self):
        """(Re)compute ``solution_performance`` from ``objective_values``."""
        self.solution_performance = success_fractions(
            self.objective_values, self.optimal_value
        )

    def tts99(self):
        """Time-to-solution at 99 % success: per-instance solve time x R99
        (reference: ccvmplotlib/problem_metadata/boxqp_metadata.py:117-132)."""
        return self.solve_time * r99(self.solution_performance["optimal"])

    def get_metadata_dict(self) -> dict:
        """All non-tensor fields (those with ``repr=True``)."""
        return {
            name: value
            for name, value in asdict(self).items()
            if self.__dataclass_fields__[name].repr
        }

    def save_tensor_to_file(self, tensor_name, file_dir=".", file_name=None):
        """Dump one entry of ``variables`` with ``torch.save``."""
        if tensor_name not in self.variables:
            raise Exception(f"Cannot find the {tensor_name} in the variables dictionary.")
        value = self.variables[tensor_name]
        if not torch.is_tensor(value):
            raise Exception(
                f"A tensor object cannot be obtained by the given tensor_name: {tensor_name}"
            )
        try:
            if file_dir != "." and not os.path.isdir(file_dir):
                os.makedirs(file_dir)
        except Exception as exc:
            raise Exception(f"Failed to create the folder path: {exc}")
        torch.save(value, f"{file_dir}/{file_name or tensor_name}.pt")
