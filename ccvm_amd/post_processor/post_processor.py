"""Post-processor interface (reference: ccvm_simulators/post_processor/post_processor.py)."""
from abc import ABC, abstractmethod
from enum import Enum


class MethodType(str, Enum):
    BFGS = "bfgs"
    LBFGS = "lbfgs"
    Adam = "adam"
    ASGD = "asgd"
    GradDescent = "grad-descent"


class PostProcessor(ABC):
    """Local refinement of the solver's final variables; ``pp_time`` holds the seconds
    the last ``postprocess`` call took."""

    pp_time = 0

    @abstractmethod
    def postprocess(self, c, q_matrix, v_vector, *args, **kwargs):
        ...


def require_tensors(c, q_matrix, v_vector):
    """Same TypeErrors as the reference post-processors (adam.py:46-52)."""
    import torch

    for name, value in (("c", c), ("q_matrix", q_matrix), ("v_vector", v_vector)):
        if not torch.is_tensor(value):
            raise TypeError(f"parameter {name} must be a tensor")
