from .problem_instance import ProblemInstance, InstanceType, DeviceType

__all__ = ["ProblemInstance", "InstanceType", "DeviceType"]
