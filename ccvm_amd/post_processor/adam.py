"""Adam post-processor on the HIP engine (``ccvm_pp_adam``).

Reference ``post_processor/adam.py:58-66`` runs ``num_iter`` (default 1) steps of
``torch.optim.Adam(lr=0.01, betas=(0.9, 0.99))`` on the loss 1/2 xQx + Vx, clamping after
each step and REBUILDING the parameter.  The optimizer keeps the original Parameter, whose
gradient stays ``None`` from the second iteration on, so only the FIRST step ever takes effect
(probed: ``num_iter=3`` equals ``num_iter=1`` bit for bit).  That step starts from zero moments:
x <- clamp(x - lr g/(|g| + eps), lo, hi),  g = 1/2 (Q + Q')x + V.
"""
from .. import engine
from .post_processor import MethodType, PostProcessor, require_tensors


class PostProcessorAdam(PostProcessor):
    def __init__(self):
        self.pp_time = 0
        self.method_type = MethodType.Adam

    def postprocess(self, c, q_matrix, v_vector, lower_clamp=0.0, upper_clamp=1.0, num_iter=1,
                    device="cpu"):
        require_tensors(c, q_matrix, v_vector)
        (batch_size, _) = c.size()
        self.pp_time = 0
        if num_iter < 1:
            return c
        c, self.pp_time = engine.postprocess(
            "adam", c, q_matrix, v_vector, lower=lower_clamp, upper=upper_clamp, lr=0.01, eps=1e-8
        )
        return c
