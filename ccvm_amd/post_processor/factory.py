"""Post-processor factory (reference: ccvm_simulators/post_processor/factory.py:13-35).

"adam", "asgd" and "grad-descent" run on the HIP engine.  "bfgs" and "lbfgs" are per-row scipy /
torch.optim.LBFGS host loops with line searches in the reference -- not data-parallel and out of
this engine's scope; asking for them fails loudly.
"""
from .adam import PostProcessorAdam
from .asgd import PostProcessorASGD
from .grad_descent import PostProcessorGradDescent
from .post_processor import MethodType

_ON_DEVICE = {
    MethodType.Adam.value: PostProcessorAdam,
    MethodType.ASGD.value: PostProcessorASGD,
    MethodType.GradDescent.value: PostProcessorGradDescent,
}
_HOST_ONLY = {MethodType.BFGS.value, MethodType.LBFGS.value}


class PostProcessorFactory:
    @staticmethod
    def create_postprocessor(method):
        key = method.lower()
        if key in _ON_DEVICE:
            return _ON_DEVICE[key]()
        if key in _HOST_ONLY:
            raise NotImplementedError(
                f"post-processor '{key}' is a host-side scipy/torch.optim loop in the reference and"
                " is not provided by the MI355X engine; use 'adam', 'asgd' or 'grad-descent'"
            )
        raise AssertionError(f"Method type is not valid. Provided: {method}")
