"""ctypes binding of libccvm_hip.so (include/ccvm_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (or
``python -m ccvm_amd.build``).  Loading never falls back to anything: a missing
library, a missing symbol or an ABI mismatch raises ``EngineUnavailable``.
"""
import ctypes
import os
from ctypes import (
    POINTER,
    Structure,
    c_char_p,
    c_double,
    c_float,
    c_int,
    c_int32,
    c_int64,
    c_size_t,
    c_uint64,
    c_void_p,
)

LIB_NAME = "libccvm_hip.so"
# CCVM_AMD_LIB: another build of the same library (same-box A/B of kernel variants); default: the in-tree one
LIB_PATH = os.environ.get("CCVM_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)
ABI_VERSION = 10

NOISE_PHILOX = 0
NOISE_REPLAY = 1

SOLVER_DL = 0
SOLVER_MF = 1
SOLVER_LANGEVIN = 2
WS_ENERGY = 3
WS_POSTPROCESS = 4
WS_FEEDBACK = 5


class EngineUnavailable(RuntimeError):
    """The HIP engine cannot be used (library not built / no MI355X visible)."""


RUN_WS_PADDED = 1  # ccvm_noise.flags: the workspace's scratch arrays still have zero padding (see ccvm_hip.h)
RUN_NO_EXCHANGE = 2  # ccvm_noise.flags: no kernel whose workgroups wait for each other (cluster / slab)
RUN_FORWARD = 4  # ccvm_noise.flags: this workspace has never run a step behind the call's first one (see ccvm_hip.h)


class Noise(Structure):
    _fields_ = [
        ("mode", c_int32),
        ("flags", c_int32),
        ("seed", c_uint64),
        ("row_offset", c_int64),
        ("w0", c_void_p),
        ("w1", c_void_p),
        ("w_ld", c_int64),
    ]


class Adam(Structure):
    _fields_ = [
        ("enabled", c_int32),
        ("add_assign", c_int32),
        ("alpha", c_double),
        ("beta1", c_double),
        ("beta2", c_double),
        ("m", c_void_p),
        ("v", c_void_p),
    ]


class DlParams(Structure):
    _fields_ = [
        ("pump", c_double),
        ("dt", c_double),
        ("noise_ratio", c_double),
        ("feedback_scale", c_double),
        ("g", c_double),
        ("lower", c_double),
        ("upper", c_double),
        ("pump_rate_flag", c_int32),
        ("reserved", c_int32),
        ("qsum", c_void_p),
        ("schedule", c_void_p),
    ]


class MfParams(Structure):
    _fields_ = [
        ("pump", c_double),
        ("dt", c_double),
        ("j", c_double),
        ("feedback_scale", c_double),
        ("g", c_double),
        ("S", c_double),
        ("lower", c_double),
        ("upper", c_double),
        ("pump_rate_flag", c_int32),
        ("reserved", c_int32),
        ("s_cols", c_void_p),
        ("qsum", c_void_p),
        ("s_full", c_void_p),
        ("schedule", c_void_p),
    ]


class LangevinParams(Structure):
    _fields_ = [
        ("dt", c_double),
        ("sigma", c_double),
        ("feedback_scale", c_double),
        ("S", c_double),
        ("pump", c_double),
        ("lower", c_double),
        ("upper", c_double),
        ("use_pump", c_int32),
        ("pump_rate_flag", c_int32),
        ("s_cols", c_void_p),
        ("qsum", c_void_p),
        ("s_full", c_void_p),
        ("schedule", c_void_p),
    ]


class FinalizeParams(Structure):
    _fields_ = [
        ("S", c_double),
        ("s_cols", c_void_p),
        ("s_full", c_void_p),
        ("lower", c_double),
        ("upper", c_double),
        ("clamp_lo", c_double),
        ("clamp_hi", c_double),
        ("scaled_by", c_double),
        ("optimal_value", c_double),
        ("clamp", c_int32),
        ("change_variables", c_int32),
    ]


class SolutionStats(Structure):
    _fields_ = [
        ("best_objective_value", c_float),
        ("within", c_int32 * 7),
        ("rows", c_int32),
        ("nonfinite", c_int32),
    ]


_P = c_void_p  # device pointers travel as integers (tensor.data_ptr())

# name -> (restype, argtypes); the single source of truth checked against the header
# by tests/test_abi.py.
SIGNATURES = {
    "ccvm_abi_version": (c_int, []),
    "ccvm_last_error": (c_char_p, []),
    "ccvm_ld": (c_int, [c_int]),
    "ccvm_rows": (c_int, [c_int]),
    "ccvm_pack": (c_int, [_P, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "ccvm_unpack": (c_int, [_P, c_int, _P, c_int, c_int, c_int, _P]),
    "ccvm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ccvm_workspace_bytes_cols": (c_size_t, [c_int, c_int, c_int]),
    "ccvm_status_offset": (c_size_t, [c_int, c_int, c_int]),
    "ccvm_column_sums": (c_int, [_P, c_int, c_int, _P, _P, c_size_t, _P]),
    "ccvm_schedule_bytes": (c_size_t, [c_int, c_int]),
    "ccvm_dl_schedule": (c_int, [POINTER(DlParams), c_int, _P, _P]),
    "ccvm_mf_schedule": (c_int, [POINTER(MfParams), POINTER(Adam), c_int, _P, _P]),
    "ccvm_langevin_schedule": (c_int, [POINTER(LangevinParams), POINTER(Adam), c_int, _P, _P]),
    "ccvm_describe_launch": (c_int, [c_int, c_int, c_int, c_int, c_int, c_char_p, c_size_t]),
    "ccvm_dl_run": (
        c_int,
        [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
         POINTER(DlParams), POINTER(Noise), _P, c_size_t, _P],
    ),
    "ccvm_mf_run": (
        c_int,
        [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
         POINTER(MfParams), POINTER(Adam), POINTER(Noise), _P, c_size_t, _P],
    ),
    "ccvm_langevin_run": (
        c_int,
        [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
         POINTER(LangevinParams), POINTER(Adam), POINTER(Noise), _P, c_size_t, _P],
    ),
    "ccvm_clamp": (c_int, [_P, c_int, c_int, c_int, c_float, c_float, _P]),
    "ccvm_change_variables": (c_int, [_P, _P, c_int, c_int, c_int, c_double, c_double, c_double, _P]),
    "ccvm_clamp_cols": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "ccvm_change_variables_cols": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_double, c_double, _P]),
    "ccvm_clamp_full": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "ccvm_change_variables_full": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_double, c_double, _P]),
    "ccvm_energy": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_double, _P, _P, c_size_t, _P]),
    "ccvm_objective_stats": (c_int, [_P, c_int, c_double, _P, _P]),
    "ccvm_finalize": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_int, POINTER(FinalizeParams), _P, _P, _P, c_size_t, _P]
    ),
    "ccvm_feedback": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_double, c_double, c_double, c_double, _P, c_size_t, _P]
    ),
    "ccvm_pp_grad_descent": (
        c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, c_double, _P, c_size_t, _P]
    ),
    "ccvm_pp_adam": (
        c_int, [_P, _P, _P, c_int, c_int, c_int, c_double, c_double, c_double, c_double, _P, c_size_t, _P]
    ),
    "ccvm_pp_lbfgs": (
        c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_double, c_double, c_double, _P, c_size_t, _P]
    ),
    "ccvm_pp_asgd": (
        c_int, [_P, _P, _P, c_int, c_int, c_int, c_double, c_double, c_double, c_double, _P, c_size_t, _P]
    ),
    "ccvm_philox_normals": (c_int, [c_uint64, c_int64, c_int, c_int, c_int, _P, _P, _P]),
}

_lib = None


def load():
    """Return the loaded library with every declared entry point bound."""
    global _lib
    if _lib is not None:
        return _lib
    # torch owns the device memory and streams this library works on: load torch (and with it
    # the HIP runtime it was built against) FIRST, so that libccvm_hip.so's libamdhip64 dependency
    # resolves to that same already-loaded runtime instead of a second copy from /opt/rocm.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise EngineUnavailable(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise EngineUnavailable(f"{LIB_NAME} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.ccvm_abi_version() != ABI_VERSION:
        raise EngineUnavailable(
            f"{LIB_NAME} has ABI {lib.ccvm_abi_version()}, this package expects {ABI_VERSION}: rebuild"
        )
    _lib = lib
    return lib


class EngineError(RuntimeError):
    """A C-ABI call returned a negative status."""


def check(status, what):
    if status != 0:
        msg = load().ccvm_last_error()
        raise EngineError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
