"""Post-processor factory (reference: ccvm_simulators/post_processor/factory.py:13-35).

"adam", "asgd", "lbfgs" and "grad-descent" run on the HIP engine (closed forms of what the reference's
torch.optim calls compute).  "bfgs" is a per-row scipy L-BFGS-B minimisation to convergence in the
reference -- a host-side Fortran line-search loop, not data-parallel and out of this engine's scope
(SURVEY.md section 2, row 9); asking for it fails loudly.
"""
from .adam import PostProcessorAdam
from .asgd import PostProcessorASGD
from .grad_descent import PostProcessorGradDescent
from .lbfgs import PostProcessorLBFGS
from .post_processor import MethodType

_ON_DEVICE = {
    MethodType.Adam.value: PostProcessorAdam,
    MethodType.ASGD.value: PostProcessorASGD,
    MethodType.GradDescent.value: PostProcessorGradDescent,
    MethodType.LBFGS.value: PostProcessorLBFGS,
}
_HOST_ONLY = {MethodType.BFGS.value}


class PostProcessorFactory:
    @staticmethod
    def create_postprocessor(method):
        key = method.lower()
        if key in _ON_DEVICE:
            return _ON_DEVICE[key]()
        if key in _HOST_ONLY:
            raise NotImplementedError(
                f"post-processor '{key}' is a host-side scipy minimisation loop in the reference and"
                " is not provided by the MI355X engine; use 'adam', 'asgd', 'lbfgs' or 'grad-descent'"
            )
        raise AssertionError(f"Method type is not valid. Provided: {method}")
