"""Projected gradient descent post-processor on the HIP engine (``ccvm_pp_grad_descent``).

Reference ``post_processor/grad_descent.py:58-64``: ``int(num_iter_main * 0.01)`` (= 10 by
default) steps of  x <- clamp(x - step_size (xQ + V), lo, hi).
"""
from .. import engine
from .post_processor import PostProcessor, require_tensors


class PostProcessorGradDescent(PostProcessor):
    def __init__(self):
        self.pp_time = 0

    def postprocess(self, c, q_matrix, v_vector, lower_clamp=0.0, upper_clamp=1.0, num_iter_main=1000,
                    num_iter_pp=None, step_size=0.1):
        require_tensors(c, q_matrix, v_vector)
        if num_iter_pp is None:
            num_iter_pp = int(num_iter_main * 0.01)
        out, self.pp_time = engine.postprocess(
            "grad-descent", c, q_matrix, v_vector, lower=lower_clamp, upper=upper_clamp,
            iters=num_iter_pp, step=step_size,
        )
        return out
