"""ccvm_amd -- MI355X-native dynamics engine for the CCVM BoxQP solvers.

The package mirrors the import surface of the reference ``ccvm_simulators`` for the
solver hot path (solvers, problem_classes.boxqp, solution, post_processor) and binds
hand-written gfx950 HIP kernels (``csrc/``, C ABI in ``include/ccvm_hip.h``).
"""
__version__ = "0.1.0"
