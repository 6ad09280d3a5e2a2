// The launch policy of libccvm_hip.so: see ccvm_plan.h.  Host code only (no kernel is instantiated here).
#define CCVM_STEP_KERNEL_ONLY  // (ccvm_kernels.h: no elementwise kernels in this unit)
#include "ccvm_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "ccvm_plan_model.h"

namespace ccvm {

static int plan_ld(int N) { return N <= 0 ? 0 : round_up(N, 128); }  // (= ccvm_ld of the C ABI)

Tuning read_tuning() {
    Tuning t;
    if (const char* e = std::getenv("CCVM_AMD_KERNEL")) {
        t.force_tile = !std::strcmp(e, "tile");
        if (!std::strcmp(e, "cluster")) { t.cluster = 1; t.slab = 0; }
        if (!std::strcmp(e, "nocluster") || t.force_tile) t.cluster = t.slab = t.ptile = 0;  // no cross-workgroup kernel at all
        if (!std::strcmp(e, "ptile")) { t.ptile = 1; t.slab = 0; }  // (the slab path is asked first: a forced family stands alone)
        if (!std::strcmp(e, "noptile")) t.ptile = 0;
        if (!std::strcmp(e, "slab")) t.slab = 1;
        if (!std::strcmp(e, "noslab")) t.slab = 0;
    }
    if (const char* e = std::getenv("CCVM_AMD_SPLIT")) t.split = e[0] == '1' ? 1 : 0;
    if (const char* e = std::getenv("CCVM_AMD_SLAB_CGRP")) t.slab_cgrp = std::atoi(e);
    if (const char* e = std::getenv("CCVM_AMD_SLAB_RG")) t.slab_rg = std::atoi(e);
    if (const char* e = std::getenv("CCVM_AMD_SLAB_DELAY")) t.slab_delay = std::atoi(e);
    // CCVM_AMD_GEOMETRY=cus,xcds: the launch policy plans for this chip instead of the device's (tests of the policy
    // functions, and a way to keep a solve inside a CU-masked or partitioned share of the chip)
    if (const char* e = std::getenv("CCVM_AMD_GEOMETRY")) {
        int cus = 0, xcds = 0;
        if (std::sscanf(e, "%d,%d", &cus, &xcds) == 2 && cus > 0 && xcds > 0) t.chip = ChipGeometry{cus, xcds};
    }
    if (const char* e = std::getenv("CCVM_AMD_KS"))
        if (e[0] == '1' || e[0] == '2' || e[0] == '4') t.ks = e[0] - '0';
    if (const char* e = std::getenv("CCVM_AMD_XCD")) t.xcd = e[0] != '0';
    if (const char* e = std::getenv("CCVM_AMD_XCD_XC")) t.xcd_xc = std::atoi(e);
    if (const char* e = std::getenv("CCVM_AMD_CLUSTER_HALF")) t.cluster_half = e[0] != '0';
    if (const char* e = std::getenv("CCVM_AMD_CLUSTER_SETS")) t.cluster_sets = (e[0] == '2' || e[0] == '3') ? e[0] - '0' : 0;
    if (const char* e = std::getenv("CCVM_AMD_PERSIST_RU"))
        if (e[0] == '2' || e[0] == '4') t.persist_ru = e[0] - '0';
    if (const char* e = std::getenv("CCVM_AMD_PERSIST_KH"))
        if (e[0] == '1' || e[0] == '2') t.persist_kh = e[0] - '0';
    if (const char* e = std::getenv("CCVM_AMD_PERSIST_PW"))
        if (e[0] == '0' || e[0] == '1') t.persist_pw = e[0] - '0' + 1;
    if (const char* e = std::getenv("CCVM_AMD_PERSIST_RSW"))
        if (e[0] == '1' || e[0] == '2') t.persist_rsw = e[0] - '0';
    if (const char* e = std::getenv("CCVM_AMD_PERSIST_WIDE")) t.persist_wide = e[0] != '0';
    if (const char* e = std::getenv("CCVM_AMD_PERSIST_XS"))
        if (e[0] == '0' || e[0] == '1') t.persist_xs = e[0] - '0' + 1;
    if (const char* e = std::getenv("CCVM_AMD_PERSIST_CW"))
        if (!std::strcmp(e, "32") || !std::strcmp(e, "64")) t.persist_cw = std::atoi(e);
    // CCVM_AMD_FAULT=cluster_drop: the cluster path launches without its last 8 workgroups, so the last member of
    // up to 8 clusters never runs and their peers' bounded waits must give up (status word, ~1 s): the error path
    // of tests/test_gpu_cluster.py -- never set in production
    if (const char* e = std::getenv("CCVM_AMD_FAULT")) t.cluster_drop = !std::strcmp(e, "cluster_drop") ? 8 : 0;
    if (const char* e = std::getenv("CCVM_AMD_SPIN_MS")) t.spin_ms = std::atof(e);
    return t;
}

// How long a wave of a persistent exchange kernel polls for another workgroup's data before it gives up (status word ->
// the engine repeats the steps on the per-step kernel), in ticks of the 100 MHz reference clock the kernels read on
// their slow path (s_memrealtime; a COUNT of polls bounds nothing: a retry round is 0.3 us on an idle chip, several us
// when every wave of a cluster retries -- the first version of this bound, 16 667 retries, took 95 ms).  Every launch's grid is resident
// (ptile: one round of tiles, cluster: one launch per round of clusters, slab: the plan fits the chip), so a peer is never
// more than a step or two behind unless its workgroup is not running at all -- another process holding CUs, a fault.
// The bound is therefore a multiple of the STEP, not of the launch: 50 estimated steps, at least 20 ms -- rounds 2-5
// waited 0.5-1.2 s, a 10^4 x cliff.  (The floor was 5 ms for one GPU run of the suite: one of ~290 tests --
// MF + Adam N = 1100, B = 777, the fourth launch of that shape in a row -- gave up a wait that the old bound had always
// seen through, i.e. a resident peer was once more than 5 ms late; what held it is not known, so the floor keeps a
// margin and a dropped workgroup still costs < 50 ms end to end: tests/test_gpu_cluster.py.)
unsigned spin_ticks(double est_step_us, const Tuning& tun) {
    const double us = tun.spin_ms > 0.0 ? 1e3 * tun.spin_ms : std::max(20000.0, 50.0 * est_step_us);
    const double ticks = 100.0 * us;
    return ticks > 4.0e9 ? 4000000000u : ticks < 100.0 ? 100u : (unsigned)ticks;
}

// The chip as the launch policies see it: CU and XCD counts, asked ONCE per device (hipDeviceAttributeMultiprocessorCount,
// hipDeviceAttributeNumberOfXccs; immutable device properties behind a mutex, no other global state).  Without a
// device (ccvm_describe_launch on a host without a GPU) the nominal MI355X in SPX mode.  CCVM_AMD_GEOMETRY overrides.
ChipGeometry device_geometry() {
    static std::mutex mu;
    static ChipGeometry cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return NOMINAL_CHIP;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (cache[dev].cus == 0) {
        int cus = 0, xcds = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            return NOMINAL_CHIP;
        }
        if (hipDeviceGetAttribute(&xcds, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || xcds <= 0) {
            (void)hipGetLastError();
            xcds = 1;  // unknown: one L2 domain (placement is a speed matter only)
        }
        cache[dev] = ChipGeometry{cus, xcds};
    }
    return cache[dev];
}
ChipGeometry chip_of(const Tuning& tun) { return tun.chip.cus > 0 ? tun.chip : device_geometry(); }

// ---- what a step costs on the per-step tile kernel (us), per solver and tile shape ---------------------------------
// Fits of the round-5 regret audit (tools/policy_regret.py: every (solver, N, B) cell of the regime map timed under the
// default plan and under every forced family / tile shape; profiles/r05_policy_regret.md): a launch of a solver step
// runs ROUNDS of workgroups, ceil(tiles / CUs); relative error of the fits 2-3 % rms for the 32 x 128 and 32 x 64 tiles,
// 3-7 % (worst cells 15-20 %: the one-stream solvers at N = 2000) for the 32 x 32 tiles, over 300 <= N <= 2000,
// 1 <= B <= 4000 (tools/fit_tile_model.py refits and checks them: profiles/r05_tile_model_fit.md).  The 32 x 32 tiles' later rounds overlap the launch boundary: 9.0 us
// per round at DL N = 1000 where a lone round takes 11.7 -- round 3's "0.37 of a 32 x 128 workgroup" priced every round
// as the first and kept these tiles off every multi-round grid.
// (TileFit, TILE_FIT: ccvm_plan_model.h, generated)
int fit_row(int mode) { return mode == MODE_MF ? 1 : mode == MODE_LANGEVIN ? 2 : 0; }
// The Adam variants (MF / Langevin: moments read and written every step by the per-step kernel, kept in registers by the
// persistent ones) cost more by family -- medians over the 84 audited cells (profiles/r05_policy_regret_adam.md): per-step
// tiles x 1.15 (32 x 32: 1.12), cluster x 1.175, resident tile x 1.10, slab x 1.05.
double tile_us(int mode, int ks, int B, int N, int cus, bool adam) {
    const TileFit& f = TILE_FIT[fit_row(mode)][ks == 1 ? 0 : ks == 2 ? 1 : 2];
    const int tiles = ((B + BM - 1) / BM) * ((N + BN / ks - 1) / (BN / ks));
    const double up = adam ? (ks == 4 ? ADAM_TILE32 : ADAM_TILE) : 1.0;
    // one round: what a lone workgroup takes, plus what the chip's share of the grid adds (more workgroups stream more
    // through the L2s); several rounds: rounds x a round, plus what a launch pays once
    if (tiles <= cus) return up * (f.l0 + f.l1 * N + (double)tiles / cus * (f.m0 + f.m1 * N));
    return up * (((tiles + cus - 1) / cus) * (f.a * N + f.b + f.q * 1e-6 * N * N) + f.e);
}
bool solver_mode(int mode) { return mode == MODE_DL || mode == MODE_MF || mode == MODE_LANGEVIN; }

// Tile shape of a per-step launch.  Solver steps (`mode`): the shape with the smallest estimate above, a finer one
// only where it is estimated 3 % ahead (ties to the larger tile).  Other kernels (energy, post-processors: 32 x 128 or
// 32 x 64 tiles only, once per solve): 32 x 64 where the 32 x 128 grid would leave half the chip idle or round up more
// (a 32 x 64 workgroup = 0.54 of a 32 x 128 one, round 3).
int choose_ks(int B, int N, const Tuning& tun, int max_ks, int mode) {
    if (tun.ks) return tun.ks < max_ks ? tun.ks : max_ks;
    const int cus = chip_of(tun).cus;
    const int nrb = (B + BM - 1) / BM;
    int best_ks = 1;
    double best = 0.0;
    static const double rel[3] = {1.0, 0.54, 0.37};
    for (int i = 0, ks = 1; ks <= max_ks && i < 3; ++i, ks *= 2) {
        const int tiles = nrb * ((N + BN / ks - 1) / (BN / ks));
        const double cost = solver_mode(mode) ? tile_us(mode, ks, B, N, cus, tun.adam) : rel[i] * ((tiles + cus - 1) / cus);
        if (i == 0 || cost < 0.97 * best) { best = cost; best_ks = ks; }
    }
    return best_ks;
}
// the cheapest per-step plan's estimate (solver steps)
double best_tile_us(int mode, int B, int N, const Tuning& tun) {
    return tile_us(mode, choose_ks(B, N, tun, 4, mode), B, N, chip_of(tun).cus, tun.adam);
}

// Grid of 32 x (128 / ks) tiles and the XCD rectangles: xr * xc = tiles / 8, xr | nrb, xc | ncb,
// (nrb/xr) * (ncb/xc) = 8, minimising the L2 footprint  xr * (bytes of an A row block) + xc * (bytes
// of a Q column panel).
// `resident`: the grid of the persistent tile kernel (ptile_kernel has no blocked order: with xr == 0 it falls back to the
// row-major map, so its grids keep whatever rectangle exists -- ADVICE r5: the demotions below were measured on
// step_kernel only and silently cost the resident grids of 11 / 13 / 15 column blocks their rectangles)
void set_grid(StepArgs& a, const Tuning& tun, bool resident) {
    const int ks = a.ks;
    a.nrb = (a.B + BM - 1) / BM;
    a.ncb = (a.N + BN / ks - 1) / (BN / ks);
    a.xr = a.xc = 0;
    const int total = a.nrb * a.ncb;
    if (total % 8 == 0 && tun.xcd && chip_of(tun).xcds == 8) {  // (a bijection either way: speed only)
        const int per = total / 8;
        long best = -1;
        for (int xc = 1; xc <= a.ncb; ++xc) {
            if (per % xc || a.ncb % xc) continue;
            const int xr = per / xc;
            if (xr > a.nrb || a.nrb % xr || (a.nrb / xr) * (a.ncb / xc) != 8) continue;
            long cost = 2L * xr + (4L / ks) * xc;
            if (tun.xcd_xc == xc) cost = 0;  // tuning: force the rectangle's width
            if (best < 0 || cost < best) { best = cost; a.xr = xr; a.xc = xc; }
        }
    }
    // No rectangle (43 % of the audited cells on this kernel: 47 or 63 column blocks, ragged batches) and a Q that an XCD's
    // L2 cannot hold next to the state (4 N^2 bytes > 7 MB: N >= 1400): super-columns of 256 output columns -- 2 / 4 / 8
    // column blocks -- for the blocked order of step_kernel, so the workgroups an XCD runs at a time share a few Q panels
    // instead of pulling a whole row of them (same-box A/B, us per step, row-major -> blocked: DL N = 2000, B = 384
    // 62.9 -> 59.3; MF N = 1500, B = 128 13.3 -> 12.1; N = 1500, B = 384 37.3 -> 37.3).  Below that size row-major runs
    // are better -- Q stays in L2 anyway and one A row block serves the whole run (DL N = 500, B = 2000 19.7 -> 22.2 blocked;
    // N = 300, B = 2000 13.2 -> 14.7): profiles/r05_ab_blocked_order.txt.  CCVM_AMD_XCD_XC forces the width at any size.
    // The same for a FULL-WIDTH rectangle (xc == ncb: 47 or 63 column blocks have no other divisor, so each XCD would
    // sweep the whole Q -- 9 / 16 MB -- through its 4 MB L2; round 5's second audit: 32 x 32 tiles at N = 2000, B = 256 in
    // rectangles 1 x 63: 25.3 us for 504 tiles where B = 384 in the blocked order takes 27.2 for 756; same-box A/B
    // profiles/r05_ab_full_width.txt: Langevin N = 2000 -11 ... -21 %, N = 1500 -3 ... -10 %, DL -2 ... -6 %), and for every
    // grid of 32 x 32 tiles at these sizes, proper rectangle or not (profiles/r05_ab_rect_vs_blocked.txt: 2048 x 2048
    // -12 %, 1024 x 1536 -6 %, else +-0; the 32 x 128 and 32 x 64 grids are no faster blocked, some 10-14 % slower: they
    // keep their rectangles).  CCVM_AMD_XCD_XC=-1 keeps the rectangle, -w forces the blocked order (A/B runs).
    if (resident) return;
    if (a.xr > 0 && (a.xc == a.ncb || ks == 4) && a.N >= 1400 && tun.xcd_xc == 0 && a.ncb > 2 * ks) a.xr = a.xc = 0;
    if (tun.xcd_xc < -1) a.xr = a.xc = 0;  // tuning: -w = the blocked order with super-columns of w blocks whatever rectangle exists
    if (a.xr == 0 && tun.xcd && total > 8 && (tun.xcd_xc > 0 || tun.xcd_xc < -1 || a.N >= 1400)) {
        const int w = tun.xcd_xc > 0 ? tun.xcd_xc : tun.xcd_xc < -1 ? -tun.xcd_xc : 2 * ks;
        if (w < a.ncb) a.xc = w;
    }
}

// Everything of a launch that does not change from step to step: operands, tile shape and grid.
void base_args(StepArgs& a, const float* Q, const float* V, int B, int N, int ld, const Tuning& tun, int max_ks, int mode) {
    std::memset(&a, 0, sizeof(a));
    a.Q = Q;
    a.V = V;
    a.B = B;
    a.N = N;
    a.ld = ld;
    a.in_scale = 1.0f;
    a.in_shift = 0.0f;
    a.qsum = V;  // any valid array while in_shift == 0
    a.ks = choose_ks(B, N, tun, max_ks, mode);
    set_grid(a, tun);
}

// ---- persistent small-N path -----------------------------------------------------------------
// (TABLE_STEPS: ccvm_plan.h)
// (table_bytes: ccvm_plan.h)

// The persistent row-owner kernel applies up to PERSIST_MAX_N columns (CCVM_AMD_KERNEL=tile forces
// the per-step kernel; read once per ABI call so a test can flip it between calls).
// (256 < N <= 320, round 6: five waves side by side for the solver variants whose working set leaves room -- persist_wide_ok --
// where its rounds cost less than the plan that would run otherwise: the slab kernel keeps the smallest batches, the cluster
// kernel a batch its 48-row clusters hold in one round where row sets need two (Langevin B = 1500: 3.35 against 4.5 us per
// step); unless a kernel family is forced: CCVM_AMD_KERNEL=cluster / nocluster / slab / ptile keep what they meant before.
// CCVM_AMD_PERSIST_WIDE=1: wherever it applies.)
bool want_persist(int N, const Tuning& tun, int solver, int B) {
    if (tun.force_tile) return false;
    if (N <= PERSIST_MAX_N) return true;
    if (N > PERSIST_WIDE_MAX_N || !persist_wide_ok(solver, tun.adam, N) || tun.persist_wide == 0 || tun.cluster != CLUSTER_DEFAULT ||
        tun.slab == 1 || tun.ptile == 1)
        return false;
    return tun.persist_wide > 0 || persist_wide_us(solver, tun.adam, B, N, chip_of(tun).cus) < plan_us(solver, B, N, tun);
}

// ---- column-cluster persistent path (ccvm_cluster.h): 256 < N <= 768, every solver and Adam variant ---------
// (CLUSTER_TWO_SETS, CLUSTER_TWO_SETS_SPREAD -- a step on two row sets relative to three, XCD by XCD / spread --, CLUSTER_ROUND_US,
// CLUSTER_MARGIN: ccvm_plan_model.h)
// Row sets of 16 per cluster (ccvm_cluster.h: SETS).  K <= 512: two.  K = 640 / 768 (9-12 members): three where the batch
// needs 48-row clusters to be on the chip at once (B = 1000: 21 clusters x 10 members), two where 32-row clusters are --
// XCD by XCD or spread -- or run in no more rounds than the 48-row ones, since a step is then two phases instead of three
// (round 5; measured, us per step, three sets -> two: profiles/r05_ab_cluster_sets.txt).  `force` (CCVM_AMD_CLUSTER_SETS=2|3, tuning): that many above K = 512.
int cluster_sets(int B, int N, const ChipGeometry& chip, int force) {
    if (round_up(N, 128) <= CL_LDS_K) return 2;
    if (force == 2 || force == 3) return force;
    const int G = (N + CL_COLS - 1) / CL_COLS, xcds = chip.xcds > 0 ? chip.xcds : 1;
    const int per_round = xcds * std::max(1, chip.cus / xcds / G);
    // rounds of resident clusters x the time of a round relative to three row sets (ccvm_abi.hip: cluster_us); batches of
    // several rounds count too: N = 640, B = 1500 = 32 clusters of 48 rows or 47 of 32, two rounds of 24 either way --
    // MF 19.0 us per step on three sets, 13.2 on two (the per-step tiles: 15.4)
    auto cost = [&](int sets) {
        const int count = (B + sets * CL_ROWS - 1) / (sets * CL_ROWS);
        const bool pinned = (count + xcds - 1) / xcds * G <= chip.cus / xcds, spread = !pinned && count * G <= chip.cus;
        const int rounds = (pinned || spread) ? 1 : (count + per_round - 1) / per_round;
        return rounds * (sets == 3 ? 1.0 : spread ? CLUSTER_TWO_SETS_SPREAD : CLUSTER_TWO_SETS);
    };
    return cost(2) < cost(3) ? 2 : 3;
}
int cluster_rows(int B, int N, const ChipGeometry& chip, int force) { return cluster_sets(B, N, chip, force) * CL_ROWS; }
int cluster_count(int B, int N, const ChipGeometry& chip, int force) {
    const int rows = cluster_rows(B, N, chip, force);
    return (B + rows - 1) / rows;
}
// the two exchange buffers of the cluster path: 8-byte {value, tag} packets (ccvm_cluster.h), one per element of
// the clusters' rows; nothing at the sizes the cluster kernel does not serve
// (planes: DL exchanges c and s, the one-stream solvers one array)
// (sized for either number of row sets: a workspace does not know which one a call will take)
size_t cluster_exchange_bytes(int B, int N, int planes) {
    if (N < CL_MIN_N || N > CL_MAX_N) return 0;
    const int rows2 = round_up(B, 2 * CL_ROWS), rows3 = round_up(N, 128) > CL_LDS_K ? round_up(B, 3 * CL_ROWS) : 0;
    return 2 * (size_t)std::max(rows2, rows3) * planes * round_up(N, 128) * CL_XE;
}
// every cluster inside one XCD and all of them on the chip at once: ceil(clusters / XCDs) x members <= CUs per XCD
bool cluster_resident_pinned(int B, int N, const ChipGeometry& chip, int force) {
    return (cluster_count(B, N, chip, force) + chip.xcds - 1) / chip.xcds * ((N + CL_COLS - 1) / CL_COLS) <= chip.cus / chip.xcds;
}
// K > 512 and the XCD-pinned placement does not fit the chip at once while the plain one does: spread
bool cluster_spread(int B, int N, const ChipGeometry& chip, int force) {
    return round_up(N, 128) > CL_LDS_K && !cluster_resident_pinned(B, N, chip, force) &&
           cluster_count(B, N, chip, force) * ((N + CL_COLS - 1) / CL_COLS) <= chip.cus;
}
// what a step costs on the cluster kernel (us): rounds of resident clusters, a round by K = 320 / 384 / ... / 768 in
// steps of 64 (measured at B = 1000: docs/kernel-cluster.md; the audit's cluster cells lie within 3 % of it; the odd
// multiples of 64 are the half-chunk variant, profiles/r05_ab_cluster_half.txt)
double cluster_us(int mode, int B, int N, const ChipGeometry& chip, bool adam, bool half, int force_sets) {
    const int G = (N + CL_COLS - 1) / CL_COLS, count = cluster_count(B, N, chip, force_sets);
    const int per_round = cluster_spread(B, N, chip, force_sets) ? count : chip.xcds * std::max(1, chip.cus / chip.xcds / G);
    const int k = (cluster_half(N, !half) ? G : round_up(N, 128) / 64) - 5;
    // (the table's K = 576 ... 768 rows are three row sets; two sets: two phases of the three -- 0.70-0.72 of the time
    // measured XCD by XCD, 0.76-0.82 spread over the XCDs: two sets leave an input one phase to cross the fabric, three two)
    const double sets = (round_up(N, 128) > CL_LDS_K && cluster_sets(B, N, chip, force_sets) == 2)
                            ? (cluster_spread(B, N, chip, force_sets) ? CLUSTER_TWO_SETS_SPREAD : CLUSTER_TWO_SETS) : 1.0;
    return sets * (adam ? ADAM_CLUSTER : 1.0) * ((count + per_round - 1) / per_round) * CLUSTER_ROUND_US[k < 0 ? 0 : k > 7 ? 7 : k][mode == MODE_DL ? 0 : mode == MODE_MF ? 1 : 2];
}
// (CLUSTER_MARGIN: ccvm_plan_model.h)
// mode: MODE_DL / MODE_MF / MODE_LANGEVIN of the run
bool want_cluster(int B, int N, const Tuning& tun, int mode, bool adam) {
    if (!tun.cluster || N < CL_MIN_N || N > CL_MAX_N) return false;
    // The kernel's block -> (cluster, member) map is written for 8 XCDs (blocks b and b + 8 share one), and its
    // measured policy below for 32 CUs in each: any other geometry (CPX / DPX partitions, CU masks: CCVM_AMD_GEOMETRY,
    // a different part) takes the per-step tile kernel, which assumes nothing about the chip.
    const ChipGeometry chip = chip_of(tun);
    if (chip.xcds != 8 || chip.cus < 8 * (CL_MIN_N / CL_COLS + 1)) return false;
    const int cus_per_xcd = chip.cus / chip.xcds;
    const int planes = mode == MODE_DL ? 2 : 1;
    if (round_up(N, 128) > CL_LDS_K && !cluster_wide_ok(mode, adam)) return false;
    const int G = (N + CL_COLS - 1) / CL_COLS;
    // Default policy (us per step, N = 500, cluster vs tile kernel): B = 1000: Langevin 4.95 vs 8.15, MF 5.39 vs 8.31,
    // DL 10.1 vs 13.1.  With more clusters than CUs they run in rounds of 256 workgroups: B = 2000 / 4000: Langevin
    // 9.97 / 20.0 vs 11.5 / 21.1, MF 10.8 vs 13.0 -- still ahead; DL 20.2 / 40.5 vs 19.8 / 38.8 -- the tile kernel's
    // larger tiles win, so DL takes the cluster path only while every cluster is resident at once.
    // Above K = 512 (three row sets of 16 per cluster, Q's k >= 512 in registers; Langevin and DL here): N = 576 / 640,
    // B = 1000: 9.5 / 9.6 and 18.8 / 18.7 vs 12.3 / 13.0 and 22 / 23.2; N = 768 (12 members: two clusters per XCD, so
    // B <= 768 is what fits at once): B = 768: 10.6 / 21.8 vs 15.5 / 27.8, but B = 512: 10.6 / 21.8 vs 9.6 / 16.0 --
    // a cluster's time per step does not shrink with a smaller batch, the tile grid's does.  So: only while the grid
    // is resident, and from B = 640 up.
    // With 11-12 members only two clusters fit an XCD's 32 CUs; spread over the XCDs (ClusterArgs::spread: the exchange
    // crosses the fabric, +10-14 % per step) 21 clusters x 12 = 252 workgroups still fit the chip: N = 768, B = 1000:
    // 10.8 / 21.8 vs 15.6 / 28.0.
    // Larger batches above K = 512 run in rounds of 8 x floor(32 / G) clusters against the tile kernel's waves of 256
    // workgroups (N = 640, Langevin: 8.9 us per round vs 12.1 per wave; B = 1500: 2 rounds 18.2 vs 1 wave 13.1; B = 2000:
    // 2 rounds 18.0 vs 2 waves 24.3, MF 19.2 vs 28.5, DL 36.3 vs 44.9; B = 4000: 4 rounds 36.0 vs 3 waves 36.3; N = 768,
    // B = 2000: 3 rounds 31.8 vs 2 waves 27.9): the cluster path is taken when rounds < 1.3 x waves.
    const bool wide = round_up(N, 128) > CL_LDS_K;
    const int count = cluster_count(B, N, chip, tun.cluster_sets);
    const bool spread_fits = wide && count * G <= chip.cus;
    if (G > cus_per_xcd && !spread_fits) return false;  // a cluster must fit an XCD, or the whole grid the chip
    const bool resident = cluster_resident_pinned(B, N, chip, tun.cluster_sets) || spread_fits;
    if (tun.cluster < 0 && planes == 2 && !wide && !resident) return false;
    if (tun.cluster < 0 && wide) {
        // (clusters of 32 rows -- two phases per step -- compete by their estimate alone, below)
        const bool three = cluster_sets(B, N, chip, tun.cluster_sets) == 3;
        if (B < 640 && three) return false;
        if (!resident && three) {
            const int per_round = chip.xcds * (cus_per_xcd / G);
            const int rounds = (count + per_round - 1) / per_round;
            const int waves = (((B + BM - 1) / BM) * ((N + BN - 1) / BN) + chip.cus - 1) / chip.cus;
            if (10 * rounds >= 13 * waves) return false;
        }
    }
    // Round 5 (regret audit): a cluster's time per step does not shrink with the batch, the per-step kernel's rounds of
    // 32 x 32 tiles do -- DL N = 640, B = 768: 18.3 us against 14.6; N = 300, B = 768: 7.9 against 6.1; N = 768, B = 1500
    // (two rounds of clusters): 43.2 against 37.0 -- so by default the cluster path must not be estimated more than 2 %
    // behind the best per-step shape
    if (tun.cluster < 0 && best_tile_us(mode, B, N, tun) < CLUSTER_MARGIN * cluster_us(mode, B, N, chip, tun.adam, tun.cluster_half != 0, tun.cluster_sets)) return false;
    return cluster_exchange_bytes(B, N, planes) / 2 < ((size_t)1 << 31);  // 32-bit buffer offsets
}
// ---- column-slab persistent path (ccvm_slab.h): small batches above N = 256 --------------------------------
// Default policy: the slab path wherever its plan exists (slab_plan: the batch's clusters fit the chip with at most
// 512 row pairs per member), except large batches at N <= 512 (below).  Measured, us per step, slab vs what ran before (gpurun_out/slab8.txt,
// profiles/r03_small_batch.md): DL N = 1000: B <= 32 3.05 vs 18.5, B = 64 5.0 vs 18.8, B = 128 9.5 vs 19.4; Langevin
// N = 1000: B = 32 2.1 vs 10.7, B = 128 5.7 vs 11.0; N = 500: Langevin B = 32 / 256 1.55 / 3.7 vs 4.8 (cluster kernel),
// DL B = 128 3.6 vs 9.9; clusters spread over the XCDs (N > 1024): PL N = 2000 B = 8 / 32 5.6 / 13.9 vs 17.5, DL
// N = 1500 B = 16 10.8 vs 25.2.  CCVM_AMD_KERNEL=slab / noslab force either.
SlabPlan want_slab(int B, int N, const Tuning& tun, int mode) {
    SlabPlan none{};
    if (!tun.slab) return none;
    const int planes = mode == MODE_DL ? 2 : 1;
    const SlabPlan p = slab_plan(B, N, planes, chip_of(tun), tun.slab_cgrp, tun.slab_rg);
    if (!p.ok) return none;
    if ((size_t)p.nclusters * planes * p.rg * p.K * 4 * SL_XE >= ((size_t)1 << 31)) return none;
    // up to N = 512 the alternatives are the cluster kernel, whose time per step does not grow with the batch, and one
    // round of 32 x 32 tiles: beyond 20 rows per cluster they win (N = 500, us per step, slab vs cluster: Langevin
    // B = 256 3.4 vs 4.9, B = 512 6.2 vs 4.9; N = 300 B = 512: 4.9 vs 3.8; DL in 24-row clusters, B = 384, vs 32 x 32 tiles:
    // N = 448 8.4 vs 7.4, N = 500 8.4 vs 8.0 -- the audit of round 5; DL N = 300, B = 512 8.4 vs 6.5)
    if (tun.slab < 0 && N <= CL_LDS_K && p.rg > 5) return none;
    // above that the alternative is the per-step tile kernel, whose time at these batches depends on N and the tile
    // shape only (measured, us per step: 32 x 64 tiles Langevin 8.5 / 10.8 / 14.2 / 17.6 at N = 700 / 1000 / 1500 / 2000,
    // DL 13.9 / 18.5 / 25 / 32; 32 x 32 tiles Langevin 5.9 / 7.8 / 10.0 / 12.5 at N = 600 / 1000 / 1500 / 2000, DL 7.9 /
    // 11.4 / 14.3 / 17.9): the plan's own estimate must beat it by 10 % (Langevin N = 700, B = 256, 28 rows per
    // cluster over the chip: 9.2 vs 8.6 measured)
    if (tun.slab < 0 && N > CL_LDS_K && p.est_us > 0.95 * best_tile_us(mode, B, N, tun)) return none;
    return p;
}
// ---- persistent streamed-Q tile kernel (ccvm_ptile.h): the 32 x 128 tile grid kept resident over a chunk --------
// Applies where the per-step kernel would run 32 x 128 tiles (`a` = the launch plan of base_args) as ONE round of
// workgroups that fills at least three quarters of the chip (every workgroup resident: its workgroups wait for each
// other), every solver and Adam variant, scalar or per-variable saturation -- chunks of any
// length, one step included: the kernel family fixes the summation order of a column's contraction, and a run's
// result must not depend on how the caller chunks it.  The headline (DL N = 1000, B = 1000: 32 x 8 = 256) and config 5 per GPU (PL N = 2000, B = 512:
// 16 x 16 = 256) are such shapes.
// Batches of several rounds: the rows of a batch never meet, so the batch is cut into SLICES of whole row blocks, each
// a resident grid of its own, run one after the other over all the steps of the chunk (`slices` launches per chunk
// instead of one per step).  Priced like choose_ks prices the per-step shapes, in per-step rounds of 32 x 128 tiles of
// the same solver: a resident round costs 0.91 (DL: 30.9 against 34.0 us at N = 1000), 0.76 (MF: 16.3 / 21.4) or 0.82
// (Langevin: 15.9 / 19.5) of one, a round of 32 x 64 tiles 0.54 / 0.49 / 0.48; taken where that beats the per-step
// plan (DL N = 1000: B = 2000 two slices, 67.5 -> 62.6 us per step; MF 41.7 -> 32.7; profiles/r04_sliced_batches.txt:
// every measured point on the side the model puts it, ties included).
// (PtilePlan: ccvm_plan.h)
// a resident round of 32 x 128 tiles (us per step; fits of the regime map: DL 30.9 at N = 1000, 59 at N = 2000)
double ptile_round_us(int mode, int N, bool adam) {
    return (adam ? ADAM_PTILE : 1.0) * (mode == MODE_DL ? PTILE_DL_PER_N * N + PTILE_DL_0 : PTILE_ONE_PER_N * N + PTILE_ONE_0 + (mode == MODE_MF ? PTILE_MF_EXTRA : 0.0));
}
PtilePlan plan_ptile(const StepArgs& a, const Tuning& tun, bool vs, int mode) {
    PtilePlan p;
    (void)vs;  // per-variable saturation is the kernel's VS template parameter (ptile_launch_*): every variant exists, the plan is the same
    if (!tun.ptile || a.N <= CL_MAX_N) return p;
    const ChipGeometry chip = chip_of(tun);
    const int nrb = (a.B + BM - 1) / BM, ncb = (a.N + BN - 1) / BN;
    if (ncb > PT_FLAG_WORDS || ncb > chip.cus) return p;
    const int fit = chip.cus / ncb;  // row blocks one resident grid holds
    const int slices = (nrb + fit - 1) / fit;
    // By default: where the resident slices are estimated no more than 5 % behind the best per-step shape (round 5; before:
    // one slice only on grids that fill three quarters of the chip, several by a model in relative rounds that priced
    // every round of 32 x 32 tiles as a lone one -- DL N = 1500, B = 512: resident 45.4 us against 39.3 on three rounds of
    // 32 x 32 tiles; N = 1000, B = 768: 31.2 against 28.8; and N = 1500, B = 384, half the chip: resident 45.2 against
    // 47.0 on 32 x 128 tiles per step, but 37.2 on 32 x 32)
    // (MF keeps the 5 %: its 32 x 32 tiles cost a round more than the model says on grids of just under three rounds --
    // N = 2000, B = 384: 36.7 us measured, 29.6 estimated -- Langevin compares as estimated: N = 900, B = 800 resident
    // 16.0 against 14.9 on 32 x 32 tiles, N = 1200, B = 512 19.3 against 18.2; DL in between: N = 1200, B = 1500 two
    // resident slices 76.0 against 80.2)
    const double margin = mode == MODE_MF ? 1.05 : mode == MODE_DL ? 1.03 : 1.0;
    if (tun.ptile < 0 && slices * ptile_round_us(mode, a.N, tun.adam) > margin * best_tile_us(mode, a.B, a.N, tun)) return p;
    p.slices = slices;
    p.rbs = (nrb + slices - 1) / slices;
    return p;
}
bool want_ptile(const StepArgs& a, const Tuning& tun, int mode, bool vs) {
    return plan_ptile(a, tun, vs, mode).slices > 0;  // (every solver and Adam variant has an instantiation)
}
// ---- batches cut in two (N > 768, and the cluster kernel's N <= 512) -------------------------------------------------------------------------------
// The rows of a batch never meet, so a batch that overflows its last resident grid a little -- B = 1100 at N = 1000:
// 35 row blocks, three rounds of 32 x 64 tiles per step, 50.9 us -- runs as two calls on the same stream: the rows that
// fill whole resident grids (1024: one launch per chunk, 30.9 us per step) and the rest under its own plan (76 rows:
// the column-slab kernel).  Decided on the per-step estimates of the plans involved (us; fits of
// the round-5 regret audit: tile_us / cluster_us / ptile_round_us above), the cut must be estimated 3 % ahead (round 4: 7 %; the
// audit's cuts came out as estimated, and MF N = 2000, B = 768 lost 7 % to a cut not taken).  Replay
// noise: the parts read their columns of the batch's blocks (ccvm_noise::w_ld).  Not with saturation arrays.
double plan_us(int mode, int B, int N, const Tuning& tun) {
    if (const SlabPlan sp = want_slab(B, N, tun, mode); sp.ok) return (tun.adam ? 1.05 : 1.0) * sp.est_us;
    if (want_cluster(B, N, tun, mode, tun.adam)) return cluster_us(mode, B, N, chip_of(tun), tun.adam, tun.cluster_half != 0, tun.cluster_sets);
    StepArgs a;
    base_args(a, nullptr, nullptr, B, N, plan_ld(N), tun, 4, mode);
    if (const PtilePlan pp = plan_ptile(a, tun, false, mode); pp.slices) return pp.slices * ptile_round_us(mode, N, tun.adam);
    return tile_us(mode, a.ks, B, N, chip_of(tun).cus, tun.adam);  // per-step kernel, the shape base_args chose
}
// rows of the first part (a multiple of 64: the parts' pitched arrays and workspaces tile the batch's), 0: no cut
int split_rows(int mode, int B, int N, const Tuning& tun) {
    if (!tun.split || tun.force_tile) return 0;
    if (tun.split < 0 && (tun.ptile > 0 || tun.ks || tun.slab > 0 || tun.cluster > 0)) return 0;  // a forced family: one plan per batch
    if (want_persist(N, tun, mode, B)) return 0;  // (row owners never interact: nothing to cut)
    const ChipGeometry chip = chip_of(tun);
    int rows_fit;  // rows of one resident grid
    if (N > CL_MAX_N) {
        const int ncb = (N + BN - 1) / BN;
        if (!tun.ptile || ncb > PT_FLAG_WORDS || ncb > chip.cus) return 0;
        rows_fit = chip.cus / ncb * BM;
    } else if (N >= CL_MIN_N && round_up(N, 128) <= CL_LDS_K) {
        // the cluster kernel's 32-row clusters, each inside an XCD (N = 500, B = 1100: 35 clusters of 8 run in two
        // rounds, 9.8 us per step for Langevin; 32 resident clusters + 76 rows on the slab kernel: 6.7)
        const int G = (N + CL_COLS - 1) / CL_COLS;
        if (!tun.cluster || chip.xcds != 8 || chip.cus / chip.xcds < G) return 0;
        rows_fit = chip.xcds * (chip.cus / chip.xcds / G) * 2 * CL_ROWS;
    } else {
        return 0;
    }
    const int cut = (B / rows_fit) * rows_fit / 64 * 64;
    if (cut <= 0 || cut >= B) return 0;
    if (tun.split > 0) return cut;
    return plan_us(mode, cut, N, tun) + plan_us(mode, B - cut, N, tun) < 0.97 * plan_us(mode, B, N, tun) ? cut : 0;
}
}  // namespace ccvm
