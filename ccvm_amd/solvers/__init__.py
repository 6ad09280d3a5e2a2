from .base import CCVMSolver, DeviceType, MachineType
from .dl import DLSolver
from .mf import MFSolver
from .langevin import LangevinSolver, PumpedLangevinSolver
from .algorithms import AdamParameters

__all__ = [
    "CCVMSolver", "DeviceType", "MachineType", "DLSolver", "MFSolver", "LangevinSolver",
    "PumpedLangevinSolver", "AdamParameters",
]
