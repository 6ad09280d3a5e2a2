"""LBFGS post-processor on the HIP engine (``ccvm_pp_lbfgs``).

Reference ``post_processor/lbfgs.py``: for every row, ``num_iter`` (default 1) times a NEW
``torch.optim.LBFGS(lr=0.001, max_iter=1)`` step on 1/2 xQx + Vx followed by a clamp.  A fresh LBFGS
with ``max_iter=1`` has no curvature history: its one iteration is a steepest-descent step of length
``lr * min(1, 1/|g|_1)`` (torch's first-iteration step size), skipped when the gradient is already
below LBFGS's tolerances.  So each iteration is
x <- clamp(x - lr min(1, 1/|g|_1) g, lo, hi),  g = 1/2 (Q + Q')x + V  (probed: identical to the
reference for num_iter 1 and 3) -- a row-parallel update that runs as one GEMM-shaped launch and
one row kernel per iteration instead of a Python loop over the batch.
"""
from .. import engine
from .post_processor import MethodType, PostProcessor, require_tensors


class PostProcessorLBFGS(PostProcessor):
    def __init__(self):
        self.pp_time = 0
        self.method_type = MethodType.LBFGS

    def postprocess(self, c, q_matrix, v_vector, lower_clamp=0.0, upper_clamp=1.0, num_iter=1):
        require_tensors(c, q_matrix, v_vector)
        (batch_size, size) = c.size()
        self.pp_time = 0
        if num_iter < 1:
            return c
        c, self.pp_time = engine.postprocess(
            "lbfgs", c, q_matrix, v_vector, lower=lower_clamp, upper=upper_clamp, iters=num_iter, lr=0.001
        )
        return c
