"""Drop-in import alias: ``ccvm_simulators`` -> ``ccvm_amd``.

Scripts written against the reference package (e.g. its examples/ccvm_boxqp_dl.py:2-4)
import ``ccvm_simulators.solvers``, ``ccvm_simulators.solvers.algorithms``,
``ccvm_simulators.problem_classes.boxqp``, ``ccvm_simulators.solution`` and
``ccvm_simulators.post_processor.factory``.  With this repository on PYTHONPATH those
imports resolve to the MI355X engine's modules; nothing is re-implemented here.
"""
import importlib
import sys

_ALIASES = {
    "solvers": "ccvm_amd.solvers",
    "solvers.algorithms": "ccvm_amd.solvers.algorithms",
    "solvers.ccvm_solver": "ccvm_amd.solvers.base",
    "solvers.dl_solver": "ccvm_amd.solvers.dl",
    "solvers.mf_solver": "ccvm_amd.solvers.mf",
    "solvers.langevin_solver": "ccvm_amd.solvers.langevin",
    "solvers.pumped_langevin_solver": "ccvm_amd.solvers.langevin",
    "problem_classes": "ccvm_amd.problem_classes",
    "problem_classes.boxqp": "ccvm_amd.problem_classes.boxqp",
    "problem_classes.boxqp.problem_instance": "ccvm_amd.problem_classes.boxqp.problem_instance",
    "solution": "ccvm_amd.solution",
    "post_processor": "ccvm_amd.post_processor",
    "post_processor.factory": "ccvm_amd.post_processor.factory",
    "post_processor.post_processor": "ccvm_amd.post_processor.post_processor",
    "post_processor.adam": "ccvm_amd.post_processor.adam",
    "post_processor.asgd": "ccvm_amd.post_processor.asgd",
    "post_processor.lbfgs": "ccvm_amd.post_processor.lbfgs",
    "post_processor.grad_descent": "ccvm_amd.post_processor.grad_descent",
}

for _name, _target in _ALIASES.items():
    _module = importlib.import_module(_target)
    sys.modules[f"{__name__}.{_name}"] = _module
    # `import ccvm_simulators.a.b as m` binds through getattr(parent, "b"): set the attribute too
    _parent, _, _child = _name.rpartition(".")
    setattr(sys.modules[f"{__name__}.{_parent}"] if _parent else sys.modules[__name__], _child, _module)

from ccvm_amd import __version__  # noqa: E402,F401
