from .factory import PostProcessorFactory
from .post_processor import MethodType, PostProcessor

__all__ = ["PostProcessorFactory", "MethodType", "PostProcessor"]
