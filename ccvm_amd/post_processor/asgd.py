"""ASGD post-processor on the HIP engine (``ccvm_pp_asgd``).

Reference ``post_processor/asgd.py``: ``num_iter`` (default 1) steps of
``torch.optim.ASGD(lr=0.01, lambd=0.001)`` on the loss 1/2 xQx + Vx, clamping after each step and
REBUILDING the parameter.  The optimizer keeps the original Parameter, whose gradient stays ``None``
from the second iteration on, so only the FIRST step ever takes effect (probed: ``num_iter=3`` equals
``num_iter=1`` bit for bit).  That step is  x <- clamp(x (1 - lambd lr) - lr g, lo, hi),
g = 1/2 (Q + Q')x + V.
"""
from .. import engine
from .post_processor import MethodType, PostProcessor, require_tensors


class PostProcessorASGD(PostProcessor):
    def __init__(self):
        self.pp_time = 0
        self.method_type = MethodType.ASGD

    def postprocess(self, c, q_matrix, v_vector, lower_clamp=0.0, upper_clamp=1.0, num_iter=1,
                    device="cpu"):
        require_tensors(c, q_matrix, v_vector)
        (batch_size, _) = c.size()
        self.pp_time = 0
        if num_iter < 1:
            return c
        c, self.pp_time = engine.postprocess(
            "asgd", c, q_matrix, v_vector, lower=lower_clamp, upper=upper_clamp, lr=0.01, lambd=0.001
        )
        return c
