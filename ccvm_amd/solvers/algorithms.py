"""Algorithm option types accepted by ``Solver.__call__(algorithm_parameters=...)``.

Same constructor, validation rules and ``to_dict`` keys as the reference's
``ccvm_simulators/solvers/algorithms.py:1-45``.
"""


class AdamParameters:
    """Hyper-parameters of the Adam-preconditioned solver variants."""

    _FIELDS = ("alpha", "beta1", "beta2", "add_assign")

    def __init__(self, alpha=0.1, beta1=0.9, beta2=0.999, add_assign=True):
        if alpha < 0.0:
            raise ValueError(f"AdamAlgorithm: Invalid `alpha` value: {alpha}")
        if not 0 < beta1 < 1:
            raise ValueError(f"AdamAlgorithm: Invalid `beta1` value: {beta1}")
        if not 0 < beta2 <= 1:  # beta2 == 1 disables the second moment
            raise ValueError(f"AdamAlgorithm: Invalid `beta2` value: {beta2}")
        self.alpha = alpha
        self.beta1 = beta1
        self.beta2 = beta2
        self.add_assign = bool(add_assign)

    def to_dict(self):
        return {name: getattr(self, name) for name in self._FIELDS}
