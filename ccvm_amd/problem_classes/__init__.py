from .boxqp import ProblemInstance, InstanceType, DeviceType

__all__ = ["ProblemInstance", "InstanceType", "DeviceType"]
