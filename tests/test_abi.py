"""The C-ABI shared library loads without a GPU and exports exactly what
include/ccvm_hip.h declares; host-only entry points behave (no compute calls here)."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ccvm_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ccvm_[a-z0-9_]+)\s*\(", text)))


def test_header_binding_and_library_agree(hip_lib):
    from ccvm_amd import _lib

    declared = declared_functions()
    assert declared, "no declarations parsed from the header"
    assert sorted(_lib.SIGNATURES) == declared
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True)
    exported = {line.split()[-1] for line in nm.stdout.splitlines() if " T " in line}
    for name in declared:
        assert name in exported, f"{name} declared in ccvm_hip.h but not exported"
        assert getattr(hip_lib, name) is not None


def test_library_is_gfx950_only(hip_lib):
    from ccvm_amd import _lib

    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _lib.LIB_PATH],
                         capture_output=True, text=True)
    if out.returncode == 0 and "amdgcn" in out.stdout:
        assert "gfx950" in out.stdout
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_layout_helpers_and_version(hip_lib):
    assert hip_lib.ccvm_abi_version() == 10
    assert [hip_lib.ccvm_ld(n) for n in (1, 20, 128, 129, 1000, 2000)] == [128, 128, 128, 256, 1024, 2048]
    assert [hip_lib.ccvm_rows(b) for b in (1, 64, 65, 1000, 4096)] == [64, 64, 128, 1024, 4096]
    assert hip_lib.ccvm_ld(0) == 0 and hip_lib.ccvm_rows(-3) == 0
    state = 1024 * 1024 * 4
    qs = 33 * 1024 * 4  # column sums of Q + 32 slice partials
    table = 4096 * 16 * 4  # schedule table of the persistent small-N path
    sync = 128  # the cluster path's launch status word (its own 128-byte line), last in the workspace
    flags = 32 * 128  # the persistent tile kernel's flag line per row block of 32 (N > 768)
    assert hip_lib.ccvm_workspace_bytes(0, 1000, 1000) == 2 * state + qs + table + flags + sync
    assert hip_lib.ccvm_workspace_bytes(1, 1000, 1000) == 3 * state + qs + table + flags + sync
    assert hip_lib.ccvm_workspace_bytes(2, 1000, 1000) == 2 * state + qs + table + flags + sync
    assert hip_lib.ccvm_status_offset(1, 1000, 1000) == 3 * state + qs + table + flags
    assert hip_lib.ccvm_status_offset(2, 1000, 1000) == 2 * state + qs + table + flags
    assert hip_lib.ccvm_status_offset(0, 1000, 1000) == 2 * state + qs + table + flags
    assert hip_lib.ccvm_status_offset(3, 1000, 1000) == ctypes.c_size_t(-1).value
    # 256 < N <= 512: + the cluster path's two exchange buffers of 8-byte {value, tag} packets
    state5, qs5 = 1024 * 512 * 4, 33 * 512 * 4
    xchg = 2 * 32 * 32 * 512 * 8
    assert hip_lib.ccvm_workspace_bytes(2, 1000, 500) == 2 * state5 + qs5 + table + xchg + sync
    assert hip_lib.ccvm_workspace_bytes(1, 1000, 500) == 3 * state5 + qs5 + table + xchg + sync
    assert hip_lib.ccvm_status_offset(2, 1000, 500) == 2 * state5 + qs5 + table + xchg
    assert hip_lib.ccvm_workspace_bytes(0, 1000, 500) == 2 * state5 + qs5 + table + 2 * xchg + sync  # DL: c and s planes
    assert hip_lib.ccvm_workspace_bytes(3, 1000, 1000) == 32 * 1024 * 4
    assert hip_lib.ccvm_workspace_bytes(4, 1000, 1000) == 2 * state
    assert hip_lib.ccvm_workspace_bytes(5, 1000, 1000) == qs
    assert hip_lib.ccvm_workspace_bytes(9, 1000, 1000) == 0
    # per-variable saturation: room for the row-scaled copy of Q (MF, Langevin only)
    assert hip_lib.ccvm_workspace_bytes_cols(1, 1000, 1000) == 3 * state + qs + table + flags + sync + 1024 * 1024 * 4
    assert hip_lib.ccvm_workspace_bytes_cols(2, 1000, 1000) == 2 * state + qs + table + flags + sync + 1024 * 1024 * 4
    assert hip_lib.ccvm_workspace_bytes_cols(0, 1000, 1000) == hip_lib.ccvm_workspace_bytes(0, 1000, 1000)


def test_struct_layouts_match_the_header():
    """ctypes mirrors of the C structs: sizes follow from the header's field lists."""
    from ccvm_amd import _lib

    assert ctypes.sizeof(_lib.Noise) == 4 + 4 + 8 + 8 + 8 + 8 + 8   # ... + w_ld
    assert ctypes.sizeof(_lib.Adam) == 4 + 4 + 3 * 8 + 2 * 8
    assert ctypes.sizeof(_lib.DlParams) == 7 * 8 + 8 + 8 + 8          # ... + qsum + schedule
    assert ctypes.sizeof(_lib.MfParams) == 8 * 8 + 8 + 8 + 8 + 8 + 8      # ... + s_cols + qsum + s_full + schedule
    assert ctypes.sizeof(_lib.LangevinParams) == 7 * 8 + 8 + 8 + 8 + 8 + 8  # ... + s_cols + qsum + s_full + schedule
    assert ctypes.sizeof(_lib.FinalizeParams) == 9 * 8 + 2 * 4
    assert ctypes.sizeof(_lib.SolutionStats) == 4 + 7 * 4 + 4 + 4


def test_schedule_table_sizes(hip_lib):
    """The schedule rows of a whole run (ccvm_dl_schedule / ccvm_mf_schedule / ccvm_langevin_schedule): 16 words per step."""
    assert hip_lib.ccvm_schedule_bytes(0, 1500) == 1500 * 64 == hip_lib.ccvm_schedule_bytes(2, 1500) == hip_lib.ccvm_schedule_bytes(1, 1500)
    assert hip_lib.ccvm_schedule_bytes(3, 1500) == 0 and hip_lib.ccvm_schedule_bytes(0, 0) == 0


def test_describe_launch_names_the_instantiation(hip_lib):
    """Host-only: which kernel a run of this shape launches (what rocprofv3 prints for it)."""
    buf = ctypes.create_string_buffer(256)
    want = {
        (0, 1000, 1000, 0): "ccvm::ptile_kernel<0, false> grid 256 x 512 threads (32 row blocks x 8 column blocks resident",
        (2, 1000, 1000, 1): "ccvm::ptile_kernel<2, true> grid 256 x 512",
        (1, 1000, 1000, 0): "ccvm::ptile_kernel<1, false> grid 256 x 512",
        (0, 1000, 100, 0): "ccvm::persist_kernel<0, false, 64, 2, 7, 4, 2, 1> grid 500 x 512 threads (noise producer waves)",   # K split + producers: one row set of 4 + 4 waves per workgroup
        (0, 1500, 100, 0): "ccvm::persist_kernel<0, false, 64, 2, 7, 4, 2> grid 750 x 256",   # three half-chain consumers per SIMD: no room for producers
        (0, 1000, 20, 0): "ccvm::persist_kernel<0, false, 32, 1, 2, 2, 1, 1> grid 250 x 256 threads (noise producer waves)",  # the shipped example
        (0, 100, 20, 0): "ccvm::persist_kernel<0, false, 32, 1, 2, 2, 1, 1> grid 25 x 256 threads (noise producer waves)",   # BASELINE config 1
        (1, 1000, 500, 0): "ccvm::cluster_kernel<1, false, 4, false> grid 256 x 512 threads (32 clusters of 8 workgroups)",
        (2, 1000, 500, 1): "ccvm::cluster_kernel<2, true, 4, false> grid 256 x 512 threads (32 clusters of 8 workgroups)",
        (0, 1000, 500, 0): "ccvm::cluster_kernel<0, false, 4, false> grid 256 x 512 threads (32 clusters of 8 workgroups)",
        (0, 1000, 300, 0): "ccvm::persist_kernel<0, false, 64, 5, 19, 4, 2, 0, 0, 48, 104> grid 500 x 640 threads (five waves side by side, K split 104 | 200, the long parts' last 96 fragments in LDS)",
        (1, 1000, 300, 0): "ccvm::cluster_kernel_half<1, false, 3, false> grid 160 x 512 threads (32 clusters of 5 workgroups)",
        (2, 4000, 500, 0): "ccvm::cluster_kernel<2, false, 4, false> grid 256 x 512 threads x 4 launches of at most 32 clusters (125 clusters of 8 workgroups)",
        (0, 4000, 500, 0): "ccvm::step_kernel<0, false, 0, 1, false, 0> grid 500 x 512",
        (1, 1000, 600, 0): "ccvm::cluster_kernel<1, false, 5, false> grid 240 x 512 threads (21 clusters of 10 workgroups)",
        (2, 1000, 600, 0): "ccvm::cluster_kernel<2, false, 5, false> grid 240 x 512 threads (21 clusters of 10 workgroups)",
        (2, 1000, 600, 1): "ccvm::cluster_kernel<2, true, 5, false> grid 240 x 512 threads (21 clusters of 10 workgroups)",
        (1, 1200, 600, 0): "ccvm::cluster_kernel<1, false, 5, false> grid 250 x 512 threads (25 clusters of 10 workgroups, spread over the XCDs)",
        # two rounds of 32-row clusters (MF N = 640, B = 1500: 13.1 us per step against 15.1 on the per-step tiles)
        (1, 1300, 600, 0): "ccvm::cluster_kernel_2sets<1, false, 5, false> grid 240 x 512 threads x 2 launches of at most 24 clusters (41 clusters of 10 workgroups)",
        (0, 1000, 640, 0): "ccvm::cluster_kernel<0, false, 5, false> grid 240 x 512 threads (21 clusters of 10 workgroups)",
        (0, 768, 768, 0): "ccvm::cluster_kernel<0, false, 6, false> grid 192 x 512 threads (16 clusters of 12 workgroups)",
        (0, 1000, 768, 0): "ccvm::cluster_kernel<0, false, 6, false> grid 252 x 512 threads (21 clusters of 12 workgroups, spread over the XCDs)",
        (2, 1000, 700, 1): "ccvm::cluster_kernel_half<2, true, 6, false> grid 231 x 512 threads (21 clusters of 11 workgroups, spread over the XCDs)",
        (0, 1100, 768, 0): "ccvm::step_kernel<0, false, 0, 1, false, 0> grid 210 x 512",
        (2, 2000, 640, 0): "ccvm::cluster_kernel<2, false, 5, false> grid 240 x 512 threads x 2 launches of at most 24 clusters (42 clusters of 10 workgroups)",
        (2, 1500, 640, 0): "ccvm::cluster_kernel_2sets<2, false, 5, false> grid 240 x 512 threads x 2 launches of at most 24 clusters (47 clusters of 10 workgroups)",  # 12.9 us vs 13.2
        (2, 2000, 768, 0): "ccvm::step_kernel<2, false, 0, 2, false, 0> grid 756 x 512",   # 3 rounds of 32 x 64 tiles < 2 of 32 x 128
        (0, 256, 1000, 0): "ccvm::step_kernel<0, false, 0, 4, false, 0> grid 256 x 512",   # 32 x 32 tiles fill the chip
        # K = 768 in clusters of 32 rows (two row sets) where they fit the chip: 7.3 us per step against 9.6 on 32 x 64 tiles
        (2, 512, 768, 0): "ccvm::cluster_kernel_2sets<2, false, 6, false> grid 192 x 512 threads (16 clusters of 12 workgroups)",
        (0, 768, 640, 0): "ccvm::cluster_kernel_2sets<0, false, 5, false> grid 240 x 512 threads (24 clusters of 10 workgroups)",
        (1, 600, 700, 0): "ccvm::cluster_kernel_half_2sets<1, false, 6, false> grid 209 x 512 threads (19 clusters of 11 workgroups, spread over the XCDs)",
        (2, 512, 2000, 0): "ccvm::ptile_kernel<2, false> grid 256 x 512 threads (16 row blocks x 16 column blocks resident",
    }
    for (solver, b, n, adam), text in want.items():
        assert hip_lib.ccvm_describe_launch(solver, b, n, adam, 0, buf, 256) == 0
        assert buf.value.decode().startswith(text), buf.value
    assert hip_lib.ccvm_describe_launch(7, 10, 10, 0, 0, buf, 256) == -1


def test_host_side_argument_checks_need_no_gpu(hip_lib):
    """Validation happens before any HIP call, so these run in the CPU container."""
    assert hip_lib.ccvm_clamp(None, 4, 4, 128, 0.0, 1.0, None) == -1
    assert b"NULL" in hip_lib.ccvm_last_error()
    assert hip_lib.ccvm_pack(None, 1, 1, 1, None, 1, 1, None) == -1
    buf = ctypes.create_string_buffer(64)
    addr = ctypes.c_void_p(ctypes.addressof(buf))
    assert hip_lib.ccvm_clamp(addr, 4, 20, 64, 0.0, 1.0, None) == -2  # ld != ccvm_ld(20)
    # whole-run schedule tables: NULL arguments, T <= 0, parameters no run call would accept
    from ccvm_amd import _lib

    dl = _lib.DlParams(pump=2.5, dt=0.005, noise_ratio=10.0, feedback_scale=100.0, g=0.05, lower=0.0, upper=1.0, pump_rate_flag=1)
    assert hip_lib.ccvm_dl_schedule(None, 10, addr, None) == -1 and hip_lib.ccvm_dl_schedule(ctypes.byref(dl), 0, addr, None) == -1
    assert hip_lib.ccvm_dl_schedule(ctypes.byref(dl), 10, None, None) == -1
    dl.upper = 0.0
    assert hip_lib.ccvm_dl_schedule(ctypes.byref(dl), 10, addr, None) == -1 and b"upper > lower" in hip_lib.ccvm_last_error()
    mf = _lib.MfParams(pump=0.5, dt=0.0025, j=399.0, feedback_scale=20.0, g=0.01, S=0.0, lower=0.0, upper=1.0)
    assert hip_lib.ccvm_mf_schedule(ctypes.byref(mf), None, 10, addr, None) == -1 and b"S > 0" in hip_lib.ccvm_last_error()
    lv = _lib.LangevinParams(dt=0.002, sigma=0.5, feedback_scale=1.0, S=0.5, lower=0.0, upper=1.0)
    assert hip_lib.ccvm_langevin_schedule(ctypes.byref(lv), None, -3, addr, None) == -1


def test_product_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    import torch

    from ccvm_amd import _lib, engine

    if not torch.cuda.is_available():
        import pytest

        with pytest.raises(_lib.EngineUnavailable):
            engine.gpu_device()
        from ccvm_amd.problem_classes.boxqp import ProblemInstance

        inst = ProblemInstance.from_arrays(torch.eye(3), torch.ones(3))
        with pytest.raises(_lib.EngineUnavailable):
            inst.compute_energy(torch.ones((2, 3)))
    # a missing library is an error, never a fallback
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libccvm_hip.so"))
    import pytest

    with pytest.raises(_lib.EngineUnavailable):
        _lib.load()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under ccvm_amd/ or ccvm_simulators/ may
    import it (and nothing there reads /root/reference)."""
    bad = []
    for pkg in ("ccvm_amd", "ccvm_simulators"):
        for base, _, files in os.walk(os.path.join(ROOT, pkg)):
            for name in files:
                if name.endswith((".py", ".hip", ".h")):
                    text = open(os.path.join(base, name)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "/root/reference" in text:
                        bad.append(os.path.join(base, name))
    assert not bad, bad
