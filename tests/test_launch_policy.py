"""The launch policy of the C ABI on the host (ccvm_describe_launch; no GPU): which kernel family a shape takes, for
the real chip (256 CUs in 8 XCDs) and for other geometries (CCVM_AMD_GEOMETRY=cus,xcds: CPX / DPX partitions, CU
masks, other parts).  The column-cluster kernel's placement is written for 8 XCDs x 32 CUs: anything else must take
the per-step tile kernel; the slab kernel plans for whatever geometry it is given."""
import ctypes
import re

import pytest


def _describe(hip_lib, solver, b, n, adam=0, per_variable_s=0):
    buf = ctypes.create_string_buffer(1024)
    assert hip_lib.ccvm_describe_launch(solver, b, n, adam, per_variable_s, buf, 1024) == 0
    return buf.value.decode()


@pytest.fixture
def clean_env(monkeypatch):
    for var in ("CCVM_AMD_KERNEL", "CCVM_AMD_GEOMETRY", "CCVM_AMD_SLAB_CGRP", "CCVM_AMD_SLAB_RG", "CCVM_AMD_KS", "CCVM_AMD_SPLIT"):
        monkeypatch.delenv(var, raising=False)
    return monkeypatch


def test_families_on_the_nominal_chip(hip_lib, clean_env):
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    assert "persist_kernel<0" in _describe(hip_lib, 0, 1000, 100)
    assert "persist_kernel<2" in _describe(hip_lib, 2, 4, 256)            # N <= 256: row owners, any batch
    assert "cluster_kernel<1" in _describe(hip_lib, 1, 1000, 500)
    assert "cluster_kernel<0" in _describe(hip_lib, 0, 1000, 768) and "spread" in _describe(hip_lib, 0, 1000, 768)
    assert "ptile_kernel<0, false>" in _describe(hip_lib, 0, 1000, 1000)    # one full round of 32 x 128 tiles: resident
    assert "ptile_kernel<2, false>" in _describe(hip_lib, 2, 512, 2000)
    assert "ptile_kernel<1, false>" in _describe(hip_lib, 1, 1000, 1000)    # every solver
    assert "ptile_kernel<0, false> 2 slices" in _describe(hip_lib, 0, 2000, 1000)  # two rounds: two resident slices of the batch
    assert "step_kernel<0, false, 0, 2" in _describe(hip_lib, 0, 1500, 1000)  # 1.5 rounds: 32 x 64 tiles, one launch per step
    d = _describe(hip_lib, 0, 32, 1000)                                    # small batch: slab, one cluster per XCD
    assert "slab_kernel<0, 8, 128, false>" in d and "8 clusters of 32 workgroups x 32 columns, 4 rows each" in d
    assert "XCDs" not in d
    d = _describe(hip_lib, 2, 32, 2000)                                    # N > 1024: 63 members of 32 columns = two XCDs
    assert "slab_kernel<2, 8, 256, true>" in d and "each over 2 XCDs" in d
    assert "16 rows each" in _describe(hip_lib, 2, 64, 2000)               # four clusters of 16 rows
    assert "step_kernel" in _describe(hip_lib, 0, 128, 2000)               # a plan exists (32 rows x 32 columns per member)
    clean_env.setenv("CCVM_AMD_KERNEL", "slab")                            # but is priced above the tile kernel
    assert "32 rows each" in _describe(hip_lib, 0, 128, 2000)
    clean_env.delenv("CCVM_AMD_KERNEL")
    assert "step_kernel" in _describe(hip_lib, 2, 256, 2000)               # no plan at all
    assert "cluster_kernel" in _describe(hip_lib, 2, 512, 500)             # N <= 512: large batches stay with the cluster kernel
    assert "slab_kernel" in _describe(hip_lib, 2, 256, 500) and "slab_kernel" in _describe(hip_lib, 2, 128, 1000)
    assert "step_kernel<2, false, 0, 4" in _describe(hip_lib, 2, 256, 1000)  # 32 rows per cluster: 32 x 32 tiles win (8.0 vs 9.1 us)


def test_tile_shape_follows_the_estimates(hip_lib, clean_env):
    """Solver steps on the per-step kernel take the tile shape with the smallest estimate (round 5: fits of the regret
    audit, ccvm_abi.hip: tile_us -- one round: a lone workgroup's time plus the grid's share of the chip; several rounds:
    rounds x a round + what a launch pays once): 32 x 64 split-K tiles (KS = 2) where the 32 x 128 grid would leave half
    the chip idle, 32 x 32 (KS = 4) where three quarters -- and on grids of SEVERAL rounds where the finer tiles round up
    less: their later rounds overlap the launch boundary (9.0 us per round at DL N = 1000 where a lone round takes 11.7),
    which round 3's model in relative rounds ("0.37 of a 32 x 128 workgroup") did not know.  Every shape below that moved
    was measured (profiles/r05_policy_regret.md and its extra cells at N = 900 / 1200)."""
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    clean_env.setenv("CCVM_AMD_KERNEL", "tile")
    ks = lambda solver, b, n: int(re.search(r"step_kernel<\d, \w+, 0, (\d)", _describe(hip_lib, solver, b, n)).group(1))
    assert ks(0, 1000, 1000) == 1 and ks(0, 1000, 896) == 1 and ks(2, 512, 2000) == 1   # one workgroup per CU
    assert ks(2, 1000, 500) == 2 and ks(0, 384, 1000) == 2 and ks(0, 256, 2000) == 2     # half the chip or less
    assert ks(0, 256, 1000) == 4 and ks(2, 129, 1000) == 4 and ks(0, 128, 2000) == 4     # a quarter or less: 32 x 32 tiles
    assert ks(2, 1000, 1500) == 2 and ks(0, 1000, 2500) == 2                             # 3 rounds of 32 x 64 against 2 of 32 x 128
    assert ks(0, 1000, 2000) == 1 and ks(0, 1000, 3000) == 1
    # 7 rounds of 32 x 32 tiles in the blocked order against 2 of 32 x 128 (measured 95.9 vs 104.1 us per step; the 32 x 64
    # tiles 104.3); at N = 1800 the 32 x 128 and 32 x 64 tiles are ahead again (110.3 / 109.8 vs 113.3)
    assert ks(0, 1000, 1700) == 4 and ks(0, 1000, 1800) in (1, 2)
    assert ks(0, 2000, 1000) == 1
    # several rounds of 32 x 32 tiles (measured, us per step, 32 x 32 against the shape round 3 took): DL N = 1200,
    # B = 1000: 53.4 vs 58.5; N = 1000, B = 768: 28.8 vs 34.1; N = 1500, B = 384: 37.2 vs 47.0; N = 900, B = 800: 26.6 vs 31.3;
    # Langevin N = 1200, B = 1000: 29.4 vs 33.2 (MF N = 1200, B = 1000: 37.0 vs 35.6 -- within the model's error, not pinned)
    assert ks(0, 1000, 1200) == 4 and ks(0, 768, 1000) == 4 and ks(0, 384, 1500) == 4 and ks(0, 800, 900) == 4
    assert ks(2, 1000, 1200) == 4 and ks(0, 500, 1500) == 4
    clean_env.setenv("CCVM_AMD_GEOMETRY", "128,4")                                      # half a chip: N = 1000 is two rounds
    assert ks(0, 1000, 1000) == 1 and ks(0, 1000, 700) == 2


def test_batches_cut_in_two(hip_lib, clean_env):
    """split_rows (N > 768): a batch that overflows its last resident grid a little runs as the rows of whole resident
    grids plus the rest under its own plan, where the plans' estimates say so; never with a forced family, a
    per-variable saturation, or where slices of the batch already fill the chip."""
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    d = _describe(hip_lib, 0, 1100, 1000)
    assert d.startswith("batch cut in two: rows 0-1023 ccvm::ptile_kernel<0, false> grid 256 x") and "| rows 1024-1099 ccvm::slab_kernel<0" in d
    assert "rows 1024-1499 ccvm::step_kernel<1, true, 0, 2" in _describe(hip_lib, 1, 1500, 1000, adam=1)
    d = _describe(hip_lib, 2, 2500, 1000)
    assert "rows 0-2047 ccvm::ptile_kernel<2, false> 2 slices" in d and "| rows 2048-2499 ccvm::step_kernel<2" in d
    assert "rows 0-511 ccvm::ptile_kernel<0, false>" in _describe(hip_lib, 0, 640, 2000)
    # the cluster kernel's sizes up to N = 512: 8 x floor(32 / G) resident clusters of 32 rows, then the rest
    d = _describe(hip_lib, 2, 1100, 500)
    assert d.startswith("batch cut in two: rows 0-1023 ccvm::cluster_kernel<2, false, 4, false>") and "| rows 1024-1099 ccvm::slab_kernel<2" in d
    assert "rows 0-1535 ccvm::cluster_kernel_half<1, true, 3" in _describe(hip_lib, 1, 1600, 300, adam=1)
    for solver, b, n in ((2, 1500, 500), (2, 2000, 500), (2, 1000, 500), (0, 1100, 640), (2, 1100, 256)):
        assert "cut in two" not in _describe(hip_lib, solver, b, n), (solver, b, n)
    for solver, b, n in ((0, 2000, 1000), (0, 1800, 1000), (0, 1000, 1000), (0, 1000, 1500), (0, 900, 1000), (2, 3000, 1000)):
        assert "cut in two" not in _describe(hip_lib, solver, b, n), (solver, b, n)
    assert "cut in two" not in _describe(hip_lib, 2, 1100, 1000, per_variable_s=1)
    for env in ({"CCVM_AMD_KERNEL": "noptile"}, {"CCVM_AMD_KERNEL": "tile"}, {"CCVM_AMD_KERNEL": "ptile"}, {"CCVM_AMD_KS": "2"},
                {"CCVM_AMD_SPLIT": "0"}):
        for k, v in env.items():
            clean_env.setenv(k, v)
        assert "cut in two" not in _describe(hip_lib, 0, 1100, 1000), env
        for k in env:
            clean_env.delenv(k)
    # the workspace carries the parts' workspaces behind its own, whose layout (status word included) is that of the
    # uncut batch -- with room for the cut whatever the estimates will say
    plain = hip_lib.ccvm_status_offset(0, 1100, 1000) + 128
    assert hip_lib.ccvm_workspace_bytes(0, 1100, 1000) >= plain + hip_lib.ccvm_workspace_bytes(0, 1024, 1000) + hip_lib.ccvm_workspace_bytes(0, 76, 1000)
    assert hip_lib.ccvm_workspace_bytes(0, 1000, 1000) == hip_lib.ccvm_status_offset(0, 1000, 1000) + 128
    assert hip_lib.ccvm_workspace_bytes(0, 1100, 640) == hip_lib.ccvm_status_offset(0, 1100, 640) + 128
    assert hip_lib.ccvm_workspace_bytes(0, 1100, 500) > hip_lib.ccvm_status_offset(0, 1100, 500) + 128


@pytest.mark.parametrize("geometry", ["64,2", "128,4", "256,1", "240,8", "32,1", "304,8"])
def test_other_geometries_never_take_the_cluster_kernel_unless_it_is_8_xcds(hip_lib, clean_env, geometry):
    clean_env.setenv("CCVM_AMD_GEOMETRY", geometry)
    cus, xcds = map(int, geometry.split(","))
    for solver, b, n in ((1, 1000, 500), (1, 1000, 300), (0, 1000, 640), (2, 1000, 768)):  # (DL / Langevin at N <= 320: row owners, on any chip)
        d = _describe(hip_lib, solver, b, n)
        if xcds != 8:
            assert "step_kernel" in d, (geometry, d)
        elif "cluster_kernel" in d:
            grid = int(re.search(r"grid (\d+) x 512", d).group(1))
            members = int(re.search(r"clusters of (\d+) workgroups", d).group(1))
            assert members <= cus // xcds or grid <= cus, (geometry, d)
    clean_env.setenv("CCVM_AMD_KERNEL", "cluster")                         # forcing it does not override the geometry
    if xcds != 8:
        assert "step_kernel" in _describe(hip_lib, 2, 1000, 500)


@pytest.mark.parametrize("geometry,b,n", [("64,2", 4, 1000), ("64,2", 8, 500), ("32,1", 4, 1000), ("128,4", 16, 1000),
                                          ("256,8", 128, 1000), ("256,8", 3, 2048), ("304,8", 8, 1200)])
def test_slab_plans_fit_the_geometry_they_are_given(hip_lib, clean_env, geometry, b, n):
    clean_env.setenv("CCVM_AMD_GEOMETRY", geometry)
    cus, xcds = map(int, geometry.split(","))
    d = _describe(hip_lib, 2, b, n)
    assert "slab_kernel" in d, d
    clusters, members, cols, rows, k = map(int, re.search(
        r"\((\d+) clusters of (\d+) workgroups x (\d+) columns, (\d+) rows each, K = (\d+)", d).groups())
    assert clusters * rows >= b and members * cols >= n and k >= n
    span = int(re.search(r"each over (\d+) XCDs", d).group(1)) if "each over" in d else 1
    groups = xcds // span
    assert -(-clusters // groups) * members <= span * (cus // xcds)
    assert int(re.search(r"grid (\d+) x 256", d).group(1)) <= cus


def test_no_plan_means_the_tile_kernel(hip_lib, clean_env):
    clean_env.setenv("CCVM_AMD_GEOMETRY", "16,1")     # 16 CUs cannot hold N = 1000 in 32-column members
    assert "step_kernel" in _describe(hip_lib, 2, 4, 1000)
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    clean_env.setenv("CCVM_AMD_KERNEL", "noslab")
    assert "step_kernel" in _describe(hip_lib, 0, 4, 1000)
    assert "cluster_kernel" in _describe(hip_lib, 2, 32, 500)  # what ran before the slab kernel existed
    clean_env.setenv("CCVM_AMD_KERNEL", "nocluster")
    assert "step_kernel" in _describe(hip_lib, 2, 32, 500)
    clean_env.setenv("CCVM_AMD_KERNEL", "tile")
    assert "step_kernel" in _describe(hip_lib, 0, 1000, 100)


def test_kernels_do_not_spill_and_the_k_split_thresholds_match_the_register_counts(hip_lib, clean_env):
    """Code-object metadata of the built library (tools/kernel_resources.py, no GPU): no kernel uses scratch, and the
    row-owner kernel's "one unsplit wave per SIMD" thresholds (ccvm_persist_launch.h: lone_from) are the K chunk counts
    from which the unsplit kernels of three (N <= 192) / four waves side by side need more than 256 VGPRs -- a compiler
    change that moves them fails here."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources

    ks = kernel_resources.kernels()
    assert len(ks) > 300
    bad = [k["name"] for k in ks if k["scratch"] or k["spill"]]
    assert not bad, bad
    regs = {}
    for k in ks:
        m = re.search(r"persist_kernel<(\d), (true|false), 64, [34], (\d+), 4, 1(?:, 0)*>", k["name"])
        if m:
            regs[(int(m.group(1)), m.group(2) == "true", int(m.group(3)))] = k["vgpr"]
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    clean_env.setenv("CCVM_AMD_PERSIST_RSW", "1")  # (not the twelve-wave workgroups of N <= 192: their own rule, tested below)
    for solver, adam in ((0, False), (1, False), (1, True), (2, False), (2, True)):
        for nch in range(9, 17):
            # B = 4096 rows per ... : whole rounds either way, so the split is taken only for the register reason
            d = _describe(hip_lib, solver, 8192 if solver else 4096, 16 * nch, 1 if adam else 0)
            split = int(re.search(r"persist_kernel<\d, \w+, 64, %d, \d+, 4, (\d)>" % (3 if nch <= 12 else 4), d).group(1)) == 2
            assert split == (regs[(solver, adam, nch)] > 256), (solver, adam, nch, regs[(solver, adam, nch)], d)


def test_row_owner_shapes_above_128_columns(hip_lib, clean_env):
    """Round 6: 128 < N <= 192 runs THREE waves side by side (with four, the fourth owned no real column), and a batch small
    enough for every two-row set to have a CU of its own runs whole chains over two rows in use instead of the K split
    (profiles/r06_ab_persist_ncg3.txt, r06_ab_persist_kh_small.txt) -- up to N = 224: from 15 K chunks on the unsplit kernel's
    registers turn the comparison around."""
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    shape = re.compile(r"persist_kernel<\d, \w+, 64, (\d), (\d+), (\d), (\d)(?:, \d+)*> grid (\d+) x (\d+) threads")

    def plan(solver, b, n, adam=0):
        m = shape.search(_describe(hip_lib, solver, b, n, adam))
        return tuple(int(g) for g in m.groups())  # (ncg, nch, ru, kh, grid, threads)

    for n, ncg in ((129, 3), (160, 3), (192, 3), (193, 4), (256, 4)):
        assert plan(0, 1000, n)[0] == ncg and plan(2, 4000, n)[0] == ncg
    # workgroups: three waves (whole chains over two rows: MF; DL and Langevin take the six-wave workgroup's unequal K split
    # instead, test_unequal_k_split_of_six_wave_workgroups), six (K split); four / eight above N = 192
    assert plan(1, 512, 144) == (3, 9, 2, 1, 256, 192) and plan(1, 513, 144) == (3, 9, 4, 2, 129, 384)
    assert plan(0, 256, 144) == (3, 9, 4, 2, 128, 384) and plan(0, 257, 144) == (3, 9, 4, 2, 129, 384)
    assert plan(2, 512, 200) == (4, 13, 2, 1, 256, 256) and plan(2, 513, 200) == (4, 13, 4, 2, 129, 512)
    assert plan(1, 512, 224, 1)[2:4] == (2, 1) and plan(1, 512, 240)[2:4] == (4, 2) and plan(0, 64, 256)[2:4] == (4, 2)
    # half the chip: half the batch
    clean_env.setenv("CCVM_AMD_GEOMETRY", "128,4")
    assert plan(1, 256, 144)[2:4] == (2, 1) and plan(1, 257, 144)[2:4] == (4, 2) and plan(0, 128, 200)[2:4] == (2, 1)
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    # the overrides still pin their dimension
    clean_env.setenv("CCVM_AMD_PERSIST_KH", "2")
    assert plan(0, 100, 144)[2:4] == (4, 2)
    clean_env.delenv("CCVM_AMD_PERSIST_KH")
    clean_env.setenv("CCVM_AMD_PERSIST_RU", "4")
    assert plan(0, 100, 144)[2] == 4
    clean_env.delenv("CCVM_AMD_PERSIST_RU")
    # three waves side by side: the split where every wave has a SIMD of its own or the unsplit kernel holds one wave per
    # SIMD, NOT for "three halves instead of two wholes" in six-wave workgroups (DL N = 176, B = 1000: 2.26 us per step, 1.96 whole)
    assert plan(0, 1500, 176)[3] == 1 and plan(0, 1000, 100)[3] == 2 and plan(2, 1000, 176)[3] == 2 and plan(1, 3000, 176)[3] == 2
    # ... but TWO six-wave row sets in a twelve-wave workgroup (three half chains on every SIMD) wherever rounds of those
    # x 1.6 are fewer than rounds of single row sets (profiles/r06_ab_persist_rsw.txt)
    two = lambda solver, b, n, adam=0: "0, 2> grid" in _describe(hip_lib, solver, b, n, adam)
    assert plan(0, 1000, 176) == (3, 11, 4, 2, 250, 768) and two(0, 1000, 176) and two(0, 513, 144) and not two(0, 512, 144)
    assert two(0, 2000, 160) and two(0, 4000, 192) and not two(0, 1500, 160)          # 4 / 8 rounds against 2 / 4; 3 against 2
    assert two(2, 2000, 130) and not two(2, 1000, 130) and not two(2, 3000, 130) and two(1, 1500, 192)
    assert two(1, 2000, 160, 1) and not two(1, 2000, 176, 1)                          # MF + Adam from 11 K chunks: > 168 VGPRs
    assert not two(0, 1000, 200) and not two(0, 1000, 128)                            # three side by side only
    clean_env.setenv("CCVM_AMD_PERSIST_RSW", "1")
    assert not two(0, 1000, 176)
    clean_env.setenv("CCVM_AMD_PERSIST_RSW", "2")
    clean_env.setenv("CCVM_AMD_PERSIST_KH", "2")
    assert two(0, 64, 176) and plan(0, 64, 176)[4:] == (16, 768)
    clean_env.delenv("CCVM_AMD_PERSIST_RSW")
    clean_env.delenv("CCVM_AMD_PERSIST_KH")
    # the twelve-wave kernels fit three waves per SIMD
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources

    twelve = [k for k in kernel_resources.kernels() if re.search(r"persist_kernel<\d, \w+, 64, 3, \d+, 4, 2, 0, 2(?:, 0)*>", k["name"])]
    assert len(twelve) == 18 and all(k["vgpr"] + k["agpr"] <= 168 and not k["spill"] for k in twelve), twelve


def test_row_owners_between_256_and_320_columns(hip_lib, clean_env):
    """Round 6: DL and Langevin / pumped Langevin without Adam run the row-owner kernel up to N = 320 -- five waves side by
    side x two K halves, ten waves of at most 168 registers, the last 8 NCH - 104 fragments of a wave's K half in LDS -- where
    rounds x the measured round (ccvm_plan_model.h: PERSIST_WIDE_ROUND_US) is less than the plan that would run otherwise
    (profiles/r06_ab_persist_wide.txt).  MF and the Adam variants do not fit (registers, then LDS) and keep the cluster kernel."""
    import os
    import sys

    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    wide = lambda solver, b, n, adam=0: re.search(
        r"persist_kernel<[012], \w+, 64, 5, (\d+), 4, 2, 0, 0, (\d+)(?:, \d+)?> grid (\d+) x 640 threads .*estimated ([\d.]+) us per step",
        _describe(hip_lib, solver, b, n, adam))
    for n, nch, ql in ((257, 17, 32), (272, 17, 32), (288, 18, 40), (300, 19, 48), (304, 19, 48), (320, 20, 56)):
        m = wide(0, 1000, n)
        assert m and (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (nch, ql, 500), (n, m and m.groups())
        m = wide(2, 1000, n)
        assert m and (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (nch, ql, 250), (n, m and m.groups())
        # MF keeps 84 fragments of a wave in registers and reaches N = 272, Langevin + Adam 92 and N = 288, MF + Adam nothing
        assert (wide(1, 1000, n) is not None) == (n <= 272) and (wide(2, 1000, n, 1) is not None) == (n <= 288) and not wide(1, 1000, n, 1)
    assert wide(1, 1000, 272).groups()[:3] == ("17", "56", "250") and wide(2, 1000, 288, 1).groups()[:3] == ("18", "56", "250")
    assert float(wide(1, 1000, 260).group(4)) == 2.27 and abs(float(wide(2, 1000, 260, 1).group(4)) - 2.29) < 0.011
    assert not wide(0, 1000, 256) and not wide(0, 1000, 321)
    # by estimate: rounds of one row set (two DL rows / four rows) per CU x the round -- against slab, cluster, tiles
    assert float(wide(0, 512, 300).group(4)) == 1.90 and float(wide(0, 513, 300).group(4)) == 3.80 and float(wide(2, 1024, 320).group(4)) == 2.17
    # DL and Langevin without Adam: the UNEQUAL K split (the waves that share a SIMD in threes take the short part)
    assert ", 0, 0, 48, 104> grid" in _describe(hip_lib, 0, 1000, 300) and "K split 104 | 200" in _describe(hip_lib, 0, 1000, 300)
    assert ", 0, 0, 32, 96> grid" in _describe(hip_lib, 2, 1000, 257) and ", 0, 0, 56, 80> grid" in _describe(hip_lib, 1, 1000, 257)
    assert "slab_kernel" in _describe(hip_lib, 0, 32, 300) and wide(0, 128, 300)            # DL: the slab kernel up to a few dozen rows
    assert "slab_kernel" in _describe(hip_lib, 2, 32, 300) and wide(2, 128, 300)           # Langevin: up to ~100
    assert "cluster_kernel" in _describe(hip_lib, 2, 1500, 300) and wide(2, 2000, 300)     # 48-row clusters: 1536 rows in ONE round
    # forced families keep what they meant; CCVM_AMD_PERSIST_WIDE pins the choice
    for forced, family in (("cluster", "cluster_kernel"), ("nocluster", "step_kernel"), ("tile", "step_kernel"), ("slab", "slab_kernel")):
        clean_env.setenv("CCVM_AMD_KERNEL", forced)
        assert family in _describe(hip_lib, 0, 256 if forced == "slab" else 1000, 300), forced
    clean_env.delenv("CCVM_AMD_KERNEL")
    clean_env.setenv("CCVM_AMD_PERSIST_WIDE", "0")
    assert "cluster_kernel" in _describe(hip_lib, 0, 1000, 300)
    clean_env.setenv("CCVM_AMD_PERSIST_WIDE", "1")
    assert wide(0, 1, 300) and wide(2, 1500, 300) and not wide(1, 1000, 300) and wide(1, 1, 272)
    clean_env.delenv("CCVM_AMD_PERSIST_WIDE")
    # a smaller chip: more rounds
    clean_env.setenv("CCVM_AMD_GEOMETRY", "128,4")
    assert float(wide(0, 512, 300).group(4)) == 3.80
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    clean_env.setenv("CCVM_AMD_PERSIST_XS", "0")  # equal halves (tuning)
    assert ", 0, 0, 48> grid" in _describe(hip_lib, 0, 1000, 300)
    clean_env.delenv("CCVM_AMD_PERSIST_XS")
    # the ten-wave kernels: at most 168 registers, no spill, within the 160 KB of LDS
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources

    ten = [k for k in kernel_resources.kernels() if re.search(r"persist_kernel<[012], \w+, 64, 5, ", k["name"])]
    assert len(ten) == 22 and all(k["vgpr"] + k["agpr"] <= 168 and not k["spill"] and k["lds"] <= 160 * 1024 for k in ten), ten


def test_unequal_k_split_of_six_wave_workgroups(hip_lib, clean_env):
    """Round 6: three waves side by side x two K halves in ONE six-wave workgroup per CU land {0, 4} {1, 5} {2} {3} on the SIMDs
    (tools/simd_probe.hip); the waves alone on a SIMD take the long parts of an UNEQUAL K split (profiles/r06_ab_persist_xs3.txt:
    -12 ... -17 %).  Only while every row set has a CU of its own: the long parts' 180-250 registers let no second workgroup in."""
    import os
    import sys

    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    xs = lambda solver, b, n, adam=0: re.search(r"persist_kernel<\d, \w+, 64, 3, \d+, 4, 2, 0, 0, 0, (\d+)> grid (\d+) x 384 threads \(K split (\d+) \| (\d+)\)",
                                                _describe(hip_lib, solver, b, n, adam))
    assert xs(0, 512, 144).groups() == ("16", "256", "16", "128") and xs(0, 300, 192).groups() == ("28", "150", "28", "164")
    assert xs(2, 1000, 144).groups() == ("24", "250", "24", "120") and xs(1, 1024, 176).groups() == ("24", "256", "24", "152")
    assert xs(1, 1000, 160, 1) and xs(2, 600, 130, 1)
    assert xs(0, 256, 144) and xs(2, 512, 144) and xs(0, 1, 130)   # DL / Langevin: from one row on (3 % ahead of whole chains over two rows)
    assert not xs(1, 512, 144) and xs(1, 513, 144)                 # MF: whole chains over two rows while every such set has a CU
    assert not xs(0, 513, 144) and not xs(2, 1025, 144)     # more row sets than CUs: twelve-wave workgroups / whole chains
    assert not xs(0, 512, 200) and not xs(0, 512, 128)      # three side by side only
    clean_env.setenv("CCVM_AMD_PERSIST_XS", "0")
    assert not xs(0, 512, 144) and "64, 3, 9, 4, 2> grid 256 x 384" in _describe(hip_lib, 0, 512, 144)
    clean_env.delenv("CCVM_AMD_PERSIST_XS")
    clean_env.setenv("CCVM_AMD_PERSIST_KH", "2")            # forced at a small batch: the same shape
    assert xs(0, 33, 129)
    clean_env.delenv("CCVM_AMD_PERSIST_KH")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources

    six = [k for k in kernel_resources.kernels() if re.search(r"persist_kernel<\d, \w+, 64, 3, \d+, 4, 2, 0, 0, 0, [1-9]\d*>", k["name"])]
    assert len(six) == 20 and all(k["vgpr"] + k["agpr"] <= 256 and not k["spill"] for k in six), six


def test_every_row_owner_shape_the_policy_picks_is_in_the_library(hip_lib, clean_env):
    """The description names the instantiation that runs (one persist_shape for launcher and description); every name the
    policy can produce over N = 1 ... 320, the batches around its thresholds and all five solver variants must be a kernel of
    the built library -- a shape picked but not instantiated would be a launch of something else."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources

    in_library = set()
    for k in kernel_resources.kernels():
        m = re.search(r"persist_kernel<([^>]*)>", k["name"])
        if m:
            in_library.add(tuple(x.strip() for x in m.group(1).split(",")))
    assert len(in_library) > 300
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    picked = set()
    for solver, adam in ((0, 0), (1, 0), (1, 1), (2, 0), (2, 1)):
        for n in list(range(1, 130, 3)) + list(range(129, 321, 5)) + [64, 65, 96, 97, 128, 192, 193, 224, 225, 256, 257, 272, 273, 288, 289, 320]:
            for b in (1, 33, 256, 257, 512, 513, 1000, 1024, 1025, 1500, 2048, 2049, 4000):
                m = re.search(r"persist_kernel<([^>]*)>", _describe(hip_lib, solver, b, n, adam))
                if m:
                    args = [x.strip() for x in m.group(1).split(",")]
                    args += ["1", "0", "0", "0", "0"][len(args) - 6:]  # the defaults: KH = 1, PW = 0, RSWO = 0, QL = 0, XS = 0
                    picked.add(tuple(args))
    assert len(picked) > 150 and picked <= in_library, sorted(picked - in_library)[:5]


def test_narrow_waves_between_64_and_96_columns(hip_lib, clean_env):
    """Round 6: 64 < N <= 96 can run THREE 32-column waves of eight rows side by side instead of two 64-column waves of four
    (ccvm_persist_launch.h: narrow).  Mostly a wash (profiles/r06_ab_persist_cw32.txt); taken where (1) the wide shape needs two
    eight-wave workgroups on a CU and this one six-wave workgroup (DL, Langevin without Adam), (2) the Adam variants while
    every two-row set has a CU of its own."""
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    narrow = lambda solver, b, n, adam=0: re.search(r"persist_kernel<\d, \w+, 32, 3, [56], (\d), 1, 1> grid (\d+) x 384 threads", _describe(hip_lib, solver, b, n, adam))
    for n in (65, 70, 80, 96):
        assert narrow(0, 1000, n).groups() == ("4", "250") and narrow(0, 513, n) and narrow(0, 1024, n)
        assert not narrow(0, 512, n) and not narrow(0, 1025, n) and not narrow(0, 100, n)
        assert narrow(2, 2000, n).groups() == ("4", "250") and narrow(2, 1025, n) and not narrow(2, 1000, n) and not narrow(2, 2049, n)
        assert not narrow(1, 1500, n)                                        # MF without Adam: +-1 %, left alone
        for solver in (1, 2):                                                # the Adam variants: two rows in use, B <= 4 x CUs
            assert narrow(solver, 1000, n, 1).groups() == ("2", "250") and narrow(solver, 1, n, 1) and narrow(solver, 1024, n, 1)
            assert (narrow(solver, 1025, n, 1) is not None) == (solver == 2) and not narrow(solver, 4000, n, 1)
        assert narrow(2, 2048, n, 1).groups() == ("4", "256")               # Langevin + Adam: rule (1) as without Adam
    assert not narrow(0, 1000, 64) and not narrow(0, 1000, 97) and not narrow(0, 1000, 100)
    # any override of the wide shape's dimensions keeps the wide shape; CCVM_AMD_PERSIST_CW pins the choice
    clean_env.setenv("CCVM_AMD_PERSIST_KH", "2")
    assert not narrow(0, 1000, 70)
    clean_env.delenv("CCVM_AMD_PERSIST_KH")
    clean_env.setenv("CCVM_AMD_PERSIST_CW", "64")
    assert not narrow(0, 1000, 70) and not narrow(2, 1000, 70, 1)
    clean_env.setenv("CCVM_AMD_PERSIST_CW", "32")
    assert "persist_kernel<0, false, 32, 3, 5, 2, 1> grid 50 x 192 threads" in _describe(hip_lib, 0, 100, 70)  # (two rows in use x two row groups: two DL rows per wave set)
    clean_env.delenv("CCVM_AMD_PERSIST_CW")
    clean_env.setenv("CCVM_AMD_GEOMETRY", "128,4")                          # half the chip: half the batch
    assert narrow(0, 500, 70) and not narrow(0, 1000, 70)


def test_producer_waves_policy(hip_lib, clean_env):
    """Round 6: the row-owner kernel's noise producer waves (ccvm_persist.h, PW).  N <= 64: the variant (rows in use x
    producers) with the smallest estimate of the fitted model (ccvm_persist_model.h, generated); 64 < N <= 128: next to
    the K split while a SIMD holds at most two half-chain consumers and two eight-wave workgroups still fit a CU where
    the batch needs them -- the NCH thresholds in persist_shape must be the code objects' register counts."""
    import os
    import sys

    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    shape = re.compile(r"persist_kernel<\d, \w+, (\d+), (\d+), (\d+), (\d+), (\d+)(, 1)?> grid (\d+) x (\d+) threads")

    def plan(solver, b, n, adam=0):
        m = shape.search(_describe(hip_lib, solver, b, n, adam))
        return {"ru": int(m.group(4)), "kh": int(m.group(5)), "pw": 1 if m.group(6) else 0, "grid": int(m.group(7)), "threads": int(m.group(8))}

    # the shipped instances at the example scripts' batch sizes: producers, one four-wave workgroup = two row sets; two rows in
    # use at N = 20, and wherever consumers + producers of four-row sets would not be fewer rounds of waves
    for solver in (0, 1, 2):
        for n in (20, 50, 64):
            for b in (1, 100, 1000):
                p = plan(solver, b, n)
                assert p["pw"] == 1 and p["threads"] == 256, (solver, n, b, p)
                assert p["ru"] == 2 or (solver == 0 and n > 32 and b == 1000), (solver, n, b, p)
    assert plan(0, 100, 20)["grid"] == 25 and plan(0, 1000, 20)["grid"] == 250
    # DL N = 64, B = 1000: 1000 one-row consumers + 1000 producers are two rounds of waves, 500 + 500 of the two-row kind one
    # (measured 0.595 against 0.483 us per step: profiles/r06_persist_policy.md)
    assert plan(0, 1000, 64) == {"ru": 4, "kh": 1, "pw": 1, "grid": 250, "threads": 256}
    # more rounds of waves: four rows in use + producers, then (consumers alone fill the SIMDs twice over) no producers
    assert plan(0, 1500, 64) == {"ru": 4, "kh": 1, "pw": 1, "grid": 375, "threads": 256}
    assert plan(0, 8000, 64)["pw"] == 0 and plan(0, 8000, 64)["ru"] == 4
    assert plan(2, 3000, 64)["pw"] == 1 and plan(2, 3000, 64)["ru"] == 4
    # overrides pin their dimension only
    clean_env.setenv("CCVM_AMD_PERSIST_PW", "0")
    assert plan(0, 1000, 20)["pw"] == 0
    clean_env.setenv("CCVM_AMD_PERSIST_PW", "1")
    clean_env.setenv("CCVM_AMD_PERSIST_RU", "4")
    assert plan(0, 1000, 20) == {"ru": 4, "kh": 1, "pw": 1, "grid": 125, "threads": 256}
    clean_env.delenv("CCVM_AMD_PERSIST_PW")
    clean_env.delenv("CCVM_AMD_PERSIST_RU")
    # two waves side by side
    assert plan(0, 1000, 100) == {"ru": 4, "kh": 2, "pw": 1, "grid": 500, "threads": 512}
    assert plan(0, 1500, 100)["pw"] == 0 and plan(0, 1500, 100)["kh"] == 2
    assert plan(2, 2000, 100)["pw"] == 1 and plan(2, 3000, 100)["pw"] == 0
    assert plan(0, 2000, 100) == {"ru": 4, "kh": 1, "pw": 0, "grid": 500, "threads": 256}
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources

    regs = {}
    for k in kernel_resources.kernels():
        m = re.search(r"persist_kernel<(\d), (true|false), 64, 2, (\d+), 4, 2, 1(?:, 0)*>", k["name"])
        if m:
            regs[(int(m.group(1)), m.group(2) == "true", int(m.group(3)))] = k["vgpr"]
    assert len(regs) == 20
    clean_env.setenv("CCVM_AMD_PERSIST_CW", "64")  # (the 64-column shape's own rule; where 64 < N <= 96 goes narrow: the test above)
    for (solver, adam, nch), vgpr in regs.items():
        # B = 1500 one-stream rows (375 row sets: two workgroups on some CUs) / 700 DL rows (350 row sets)
        p = plan(solver, 1500 if solver else 700, 16 * nch, 1 if adam else 0)
        assert p["kh"] == 2 and (p["pw"] == 1) == (vgpr <= 128), (solver, adam, nch, vgpr, p)
        assert plan(solver, 1000 if solver else 500, 16 * nch, 1 if adam else 0)["pw"] == 1  # one workgroup per CU: always


def test_description_of_a_resident_tile_grid_is_the_grid_that_runs(hip_lib, clean_env):
    """a single-slice persistent tile launch is 32 x 128 tiles whatever shape the per-step plan has; the
    description printed the per-step grid (KS = 2 / 4: up to four times the workgroups).  Every single-slice description
    over a scan of shapes: at most one workgroup per CU, row blocks x column blocks = the grid, and the rectangle a
    resident grid keeps (no "0 x w" blocked order: ptile_kernel has none)."""
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    seen = 0
    for solver in (0, 1, 2):
        for n in (800, 1000, 1400, 1600, 1800, 2000, 2048):
            for b in (160, 256, 512, 672, 1000, 1024):
                d = _describe(hip_lib, solver, b, n)
                m = re.match(r"ccvm::ptile_kernel<\d, \w+> grid (\d+) x 512 threads \((\d+) row blocks x (\d+) column blocks resident, "
                             r"XCD rectangle (\d+) x (\d+)\)", d)
                if not m:
                    continue
                grid, nrb, ncb, xr, xc = (int(g) for g in m.groups())
                seen += 1
                assert grid == nrb * ncb <= 256, d
                assert nrb == -(-b // 32) and ncb == -(-n // 128), d
                assert (xr == 0) == (xc == 0), d
    assert seen >= 10
    assert "grid 147 x 512 threads (21 row blocks x 7 column blocks resident" in _describe(hip_lib, 1, 672, 800)  # (printed 525 = 21 x 25)


def test_persistent_tile_kernel_needs_the_whole_grid_resident(hip_lib, clean_env):
    """ccvm_ptile.h: its workgroups wait for each other, so the grid must fit the chip the policy plans for (CU masks,
    partitions: CCVM_AMD_GEOMETRY) and be estimated no more than 5 % behind the best per-step tile shape (round 5: the
    estimates of tile_us / ptile_round_us; before: "at least three quarters of the chip"), and CCVM_AMD_KERNEL=tile /
    nocluster / noptile switch it off.  A batch of several rounds is cut into slices of whole row blocks, each a resident
    grid of its own, under the same rule."""
    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    assert "ptile_kernel" in _describe(hip_lib, 0, 1000, 1000) and "ptile_kernel" in _describe(hip_lib, 1, 800, 900)
    # (DL N = 900, B = 800 -- 200 of 256 CUs -- went to three rounds of 32 x 32 tiles in round 5: 26.6 against 31.0 us)
    assert "step_kernel<0, false, 0, 4" in _describe(hip_lib, 0, 800, 900)
    d = _describe(hip_lib, 0, 1000, 1100)                                  # 32 x 9 = 288 tiles: more than the chip holds
    assert d.startswith("batch cut in two: rows 0-895 ccvm::ptile_kernel<0, false> grid 252 x") and "| rows 896-999 ccvm::step_kernel<0, false, 0, 4" in d
    clean_env.setenv("CCVM_AMD_SPLIT", "0")
    assert _describe(hip_lib, 0, 1000, 1100).startswith("ccvm::step_kernel<0, false, 0, 4")  # (32 x 64 tiles until round 5)
    clean_env.delenv("CCVM_AMD_SPLIT")
    assert "ptile_kernel" not in _describe(hip_lib, 0, 512, 1000)          # half the chip: 32 x 64 tiles
    # per-variable saturation (MF, Langevin): the kernel's VS instantiation
    assert "ptile_kernel<1, false, false, true> grid 256" in _describe(hip_lib, 1, 1000, 1000, per_variable_s=1)
    assert "ptile_kernel<2, true, false, true> 2 slices" in _describe(hip_lib, 2, 2000, 1000, adam=1, per_variable_s=1)
    for off in ("tile", "nocluster", "noptile"):
        clean_env.setenv("CCVM_AMD_KERNEL", off)
        assert "step_kernel" in _describe(hip_lib, 0, 1000, 1000), off
    clean_env.delenv("CCVM_AMD_KERNEL")
    clean_env.setenv("CCVM_AMD_GEOMETRY", "240,8")                         # a CU-masked chip: 256 workgroups do not fit
    d = _describe(hip_lib, 0, 1000, 1000)                                  # 30 row blocks resident, the last 40 rows apart
    assert d.startswith("batch cut in two: rows 0-959 ccvm::ptile_kernel<0, false> grid 240 x") and "| rows 960-999 ccvm::slab_kernel" in d
    clean_env.setenv("CCVM_AMD_SPLIT", "0")
    assert "step_kernel" in _describe(hip_lib, 0, 1000, 1000)
    clean_env.delenv("CCVM_AMD_SPLIT")
    clean_env.setenv("CCVM_AMD_GEOMETRY", "128,4")
    assert "ptile_kernel<0, false> 2 slices" in _describe(hip_lib, 0, 1000, 1000)  # two resident grids of 16 x 8
    assert "ptile_kernel" in _describe(hip_lib, 0, 500, 1000)              # 16 x 8 = 128 workgroups fill that chip once


def test_default_policy_against_the_regret_audit(hip_lib, clean_env):
    """Round 5: profiles/r05_policy_regret.jsonl holds, for every (solver, N, B) cell of the regime
    map, the measured time per step of every plan that can serve the cell (tools/policy_regret.py on an MI355X: the
    default and every forced family / tile shape).  Whatever the policy functions become, the plan they pick for a cell
    must not be measured more than 9 % behind the best plan of that cell (the audit itself lists what is beyond 5 %:
    seven cells of 773 with the Adam variants, the largest without Adam 8 %: Langevin N = 640, B = 2500, four rounds of
    32-row clusters 25.6 us against 32 x 32 tiles 23.7; run-to-run noise of a cell is
    about 2 %), and a larger batch must never be faster than a smaller one by more than 8 % under the picked plans.
    (The cells of N = 300, 448, 576 and 700 were measured after the cluster kernel's half-chunk variant went in, those of
    N = 1500 and 2000 after the blocked order of the 32 x 32 tiles.)"""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tools.regime_map import family

    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    solver_id = {"dl": 0, "mf": 1, "langevin": 2}
    cells = {}
    with open(os.path.join(root, "profiles", "r05_policy_regret.jsonl")) as fh:
        for line in fh:
            r = json.loads(line)
            cells[(r["kind"], r["n"], r["b"])] = r
    assert len(cells) >= 530
    # cells exempt from the bound, each with its reason:
    known = {
        # two resident slices of 16 x 16 tiles: measured 69.3 in the audit but 60.5 for the same two slices as a cut, in
        # round 4's map (60.5) and per slice at every other batch (B = 768 60.5, 1500 91.0 = 3 x 30.3): one bad sample
        ("mf", 2000, 1000),
    }
    picked, regrets, unmeasured = {}, [], []
    for (kind, n, b), r in sorted(cells.items()):
        plans = [p for p in r["plans"] if p.get("us")]
        fam = family(_describe(hip_lib, solver_id[kind], b, n))
        if fam == "R" and n > 256:
            continue  # (round 6: DL / Langevin at 256 < N <= 320 run the row-owner kernel's five waves side by side, which this
                      # audit predates -- measured against every other plan in profiles/r06_policy_regret.jsonl, the test below)
        mine = [p for p in plans if p["family"] == fam]
        if not mine:
            unmeasured.append((kind, n, b, fam))
            continue
        best = min(p["us"] for p in plans)
        picked[(kind, n, b)] = mine[0]["us"]
        if mine[0]["us"] > 1.09 * best and (kind, n, b) not in known:
            regrets.append((kind, n, b, fam, round(mine[0]["us"], 2), round(best, 2)))
    assert not regrets, regrets
    assert len(unmeasured) <= 6, unmeasured  # (a plan the audit did not time: re-run tools/policy_regret.py)
    # the Adam variants of MF / Langevin (84 cells; their estimates carry a factor per family: ccvm_abi.hip, tile_us)
    adam_regrets = []
    with open(os.path.join(root, "profiles", "r05_policy_regret_adam.jsonl")) as fh:
        adam_cells = {(r["kind"], r["n"], r["b"]): r for r in map(json.loads, fh)}
    for (kind, n, b), r in sorted(adam_cells.items()):
        plans = {p["family"]: p["us"] for p in r["plans"] if p.get("us")}
        fam = family(_describe(hip_lib, solver_id[kind.split("+")[0]], b, n, adam=1))
        # (Langevin + Adam N = 640, B = 256: the slab plan -- 6 row groups over two XCDs -- measures 6.32 us, its model says
        # 6.9 and the 32 x 32 tiles are taken at 6.97; the slab model's largest error on a plan it loses with)
        # (10 % here: the Adam variants' estimates are the plain ones times ONE factor per family, and the measured factors
        # scatter -- 32 x 64 tiles 1.05 ... 1.31 around the 1.15 used: MF + Adam N = 640, B = 4000 runs them at 47.7 us where
        # four rounds of clusters take 43.5)
        if fam in plans and plans[fam] > 1.10 * min(plans.values()) and (kind, n, b) != ("langevin+adam", 640, 256):
            adam_regrets.append((kind, n, b, fam, round(plans[fam], 2), round(min(plans.values()), 2)))
    assert len(adam_cells) >= 120 and not adam_regrets, adam_regrets
    upside_down = []
    for (kind, n, b), us in picked.items():
        for (k2, n2, b2), us2 in picked.items():
            # (8 %: DL N = 300 on ONE round of 32 x 32 tiles takes 6.5 us at B = 512 and 6.1 at B = 768, the same kernel)
            if k2 == kind and n2 == n and b2 > b and us2 < 0.92 * us:
                upside_down.append((kind, n, b, round(us, 2), b2, round(us2, 2)))
    assert not upside_down, upside_down


def test_default_policy_against_a_second_boxes_audit(hip_lib, clean_env):
    """Round 6 ("pins the policy to that data, not to a second box"): the same audit run again on the
    box of round 6's call 18 -- another machine of the pool, the final code of round 6 (producer waves, one launch per
    cluster round, time-bounded waits) -- over 162 cells (N = 100 ... 2000, B = 32 ... 2000, the three solvers), and the
    cells of N = 257 / 300 / 320 again on the box of call 33, when DL / Langevin there moved to the row-owner kernel's five
    waves side by side (216 cells in all): profiles/r06_policy_regret.{jsonl,md}.  The policy's constants were NOT refitted to it.  The plan picked for a cell must
    be within 7 % of the best plan THAT box measured for the cell (the audit's own list beyond 5 %: one cell, Langevin
    N = 2000, B = 768 at 6 %)."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tools.regime_map import family

    clean_env.setenv("CCVM_AMD_GEOMETRY", "256,8")
    solver_id = {"dl": 0, "mf": 1, "langevin": 2}
    with open(os.path.join(root, "profiles", "r06_policy_regret.jsonl")) as fh:
        cells = {(r["kind"], r["n"], r["b"]): r for r in map(json.loads, fh)}
    assert len(cells) >= 216
    regrets, unmeasured = [], []
    for (kind, n, b), r in sorted(cells.items()):
        plans = [p for p in r["plans"] if p.get("us")]
        fam = family(_describe(hip_lib, solver_id[kind], b, n))
        mine = [p for p in plans if p["family"] == fam]
        if not mine:
            unmeasured.append((kind, n, b, fam))
        elif mine[0]["us"] > 1.07 * min(p["us"] for p in plans):
            regrets.append((kind, n, b, fam, round(mine[0]["us"], 2), round(min(p["us"] for p in plans), 2)))
    assert not regrets, regrets
    assert not unmeasured, unmeasured


def test_tile_time_model_fits_the_audit_data():
    """The coefficients ccvm_abi.hip carries for the per-step tile kernel's time (TILE_FIT) against every timing of such a
    plan in the committed audits (tools/fit_tile_model.py): rms relative error below 4 % for the 32 x 128 and 32 x 64
    tiles, 7 % for the 32 x 32 ones, worst cell below 25 % -- a refit that drifts away from the data fails here."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fit_tile_model as ftm

    data, lib = ftm.load(), ftm.library_table()
    for (kind, ks), rows in data.items():
        rms, worst = ftm.errors(lib[(kind, ks)], rows)
        assert len(rows) > 100 and rms < (0.07 if ks == 4 else 0.04) and worst < 0.25, (kind, ks, rms, worst)
