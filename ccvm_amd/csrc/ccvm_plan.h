// The launch policy of libccvm_hip.so (split out of ccvm_abi.hip): the tuning environment, the chip's
// geometry, the cost models of the kernel families and every want_* / plan_* function -- which kernel a (solver, B, N)
// takes, in which shape.  Definitions: ccvm_plan.hip; the fitted constants: ccvm_plan_model.h, a GENERATED header
// (tools/fit_tile_model.py --write refits the per-step tile model from profiles/*_policy_regret*.jsonl and rewrites it;
// the row-owner kernel's variant model is ccvm_persist_model.h, tools/fit_persist_model.py).  ccvm_describe_launch names
// the plan and prints the estimate it rests on; tests/test_launch_policy.py is the gate.
#pragma once
#include <cstddef>

#include "ccvm_cluster.h"
#include "ccvm_kernels.h"
#include "ccvm_persist_launch.h"
#include "ccvm_ptile.h"
#include "ccvm_slab.h"

namespace ccvm {

// Tuning knobs from the environment (tests / profiling only), read ONCE per ABI call: the per-step
// launch loop never touches environ (getenv is a linear scan, and not safe against a concurrent
// setenv from another host thread).
//   CCVM_AMD_KERNEL=tile      force the per-step tile kernel where a persistent kernel would apply
//   CCVM_AMD_KERNEL=cluster   the column-cluster persistent kernel wherever it applies (256 < N <= 768), also for
//                             batches whose clusters run in several rounds, and never the slab kernel;
//                             =nocluster: neither of the two kernels whose workgroups exchange data (cluster, slab):
//                             every size above 256 on the per-step tile kernel
//   CCVM_AMD_KERNEL=slab      the column-slab small-batch kernel wherever it has a plan; =noslab: never
//   CCVM_AMD_KERNEL=ptile     the persistent streamed-Q tile kernel wherever it applies (ccvm_ptile.h); =noptile: never
//                             (=tile and =nocluster switch it off too: its workgroups wait for each other)
//   CCVM_AMD_SLAB_CGRP=1|2|4|8, CCVM_AMD_SLAB_RG=n   force its member width (4 CGRP columns) / row groups per cluster
//   CCVM_AMD_SLAB_DELAY=n     fetch delay of its clusters that span XCDs, x 64 cycles (timing only)
//   CCVM_AMD_GEOMETRY=cus,xcds  plan for this chip instead of the device's
//   CCVM_AMD_KS=1|2           force the tile shape (32 x 128 / 32 x 64 split-K)
//   CCVM_AMD_XCD=0            linear block -> tile map instead of the XCD rectangles
//   CCVM_AMD_XCD_XC=n         force the XCD rectangle's width
//   CCVM_AMD_PERSIST_RU=2|4   rows in use per 4-row group of the persistent kernel
//   CCVM_AMD_PERSIST_PW=0|1   its noise producer waves off / on (N <= 128; default: by shape and batch size)
//   CCVM_AMD_PERSIST_RSW=1|2  row sets per workgroup of its six-wave row sets (128 < N <= 192 with the K split; default: by batch size)
//   CCVM_AMD_PERSIST_CW=32|64 64 < N <= 96: three 32-column waves side by side (eight rows each) / two 64-column waves
//   CCVM_AMD_PERSIST_XS=0|1   five waves side by side: equal K halves / the unequal split that balances the SIMDs (default)
//   CCVM_AMD_PERSIST_WIDE=0   256 < N <= 320 stays on the column-cluster / slab / tile kernels (default: DL and Langevin without
//                             Adam run the row-owner kernel's five-waves-side-by-side shape there)
//   CCVM_AMD_SPIN_MS=x        bound of a wait for another workgroup, milliseconds (default: spin_ticks below -- 20 ms or 50
//                             estimated steps; rehearsals that put several processes on ONE GPU raise it)
constexpr int CLUSTER_DEFAULT = -1;  // -1: where it applies AND the whole grid is resident at once

struct Tuning {
    bool force_tile = false;
    int cluster = CLUSTER_DEFAULT;  // 1: the cluster kernel wherever it applies, 0: never, -1: see want_cluster
    int ks = 0;          // 0: choose by grid size
    bool xcd = true;
    int xcd_xc = 0;      // 0: choose by L2 footprint
    int persist_ru = 0;  // 0: choose by batch size
    int persist_kh = 0;  // K split of the two-wave shapes (64 < N <= 128): 1 off, 2 on, 0: by batch size
    int persist_pw = 0;  // noise producer waves of the row-owner kernel: 1 off, 2 on, 0: by shape and batch size
    int persist_rsw = 0; // row sets per workgroup where a row set is six waves (128 < N <= 192, K split): 1 / 2, 0: by batch size
    int persist_cw = 0;  // 64 < N <= 96: 32 = three 32-column waves side by side (eight rows each), 64 = two 64-column waves, 0: by policy
    int persist_xs = 0;  // five waves side by side: 1 equal K halves, 2 the unequal split, 0: default (unequal)
    int persist_wide = -1;  // 256 < N <= 320 on the row-owner kernel (five waves side by side, DL / Langevin without Adam): 0 never, else where it applies
    int cluster_drop = 0;  // fault injection (tests): workgroups left out of a cluster launch
    double spin_ms = 0.0;  // CCVM_AMD_SPIN_MS: > 0 replaces the bound of the cross-workgroup waits
    int cluster_sets = 0;  // CCVM_AMD_CLUSTER_SETS=2|3 (tuning): that many row sets per cluster above K = 512; 0: cluster_sets()
    int cluster_half = 1;  // 0 (CCVM_AMD_CLUSTER_HALF=0, tuning): the full-chunk cluster kernel also where N mod 128 is in 1 .. 64
    int slab = CLUSTER_DEFAULT;  // column-slab small-batch kernel: 1 wherever it applies, 0 never, -1: see want_slab
    int slab_cgrp = 0, slab_rg = 0;  // 0: choose (ccvm_slab.h: slab_plan)
    int slab_delay = -1;             // >= 0: the fetch delay of clusters that span XCDs (x 64 cycles), else slab_fabric_delay
    int ptile = CLUSTER_DEFAULT;     // persistent streamed-Q tile kernel: 1 wherever it applies, 0 never, -1: see want_ptile
    int split = CLUSTER_DEFAULT;     // batches cut into a part of whole resident grids and the rest: 1 wherever a cut exists, 0 never, -1: see split_rows
    ChipGeometry chip{0, 0};  // 0: ask the device
    bool adam = false;        // the run's Adam variant (set by the entry points, not by the environment): the estimates below price it
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

Tuning read_tuning();
unsigned spin_ticks(double est_step_us, const Tuning& tun);

constexpr ChipGeometry NOMINAL_CHIP{256, 8};
ChipGeometry device_geometry();
ChipGeometry chip_of(const Tuning& tun);

constexpr int TABLE_STEPS = 4096;  // steps per persistent launch (schedule table rows in the workspace)
inline size_t table_bytes() { return (size_t)TABLE_STEPS * TABLE_WORDS * sizeof(float); }

// ---- the per-step tile kernel ----
int fit_row(int mode);
double tile_us(int mode, int ks, int B, int N, int cus, bool adam = false);
bool solver_mode(int mode);
int choose_ks(int B, int N, const Tuning& tun, int max_ks, int mode = -1);
double best_tile_us(int mode, int B, int N, const Tuning& tun);
void set_grid(StepArgs& a, const Tuning& tun, bool resident = false);
void base_args(StepArgs& a, const float* Q, const float* V, int B, int N, int ld, const Tuning& tun, int max_ks = 2, int mode = -1);

// ---- the row-owner persistent kernel (N <= 256, DL / Langevin without Adam N <= 320; its shape: ccvm_persist_launch.h, persist_shape) ----
bool want_persist(int N, const Tuning& tun, int solver, int B);

// ---- the column-cluster persistent kernel ----
int cluster_sets(int B, int N, const ChipGeometry& chip, int force = 0);
int cluster_rows(int B, int N, const ChipGeometry& chip, int force = 0);
int cluster_count(int B, int N, const ChipGeometry& chip, int force = 0);
bool cluster_resident_pinned(int B, int N, const ChipGeometry& chip, int force = 0);
bool cluster_spread(int B, int N, const ChipGeometry& chip, int force = 0);
double cluster_us(int mode, int B, int N, const ChipGeometry& chip, bool adam = false, bool half = true, int force_sets = 0);
size_t cluster_exchange_bytes(int B, int N, int planes);
bool want_cluster(int B, int N, const Tuning& tun, int mode, bool adam);

// ---- the column-slab persistent kernel ----
SlabPlan want_slab(int B, int N, const Tuning& tun, int mode);

// ---- the persistent tile kernel, batches cut in two ----
struct PtilePlan {
    int slices = 0;  // 0: not this kernel
    int rbs = 0;     // row blocks per slice (the last one may hold fewer)
};
double ptile_round_us(int mode, int N, bool adam = false);
PtilePlan plan_ptile(const StepArgs& a, const Tuning& tun, bool vs, int mode);
bool want_ptile(const StepArgs& a, const Tuning& tun, int mode, bool vs);
double plan_us(int mode, int B, int N, const Tuning& tun);
int split_rows(int mode, int B, int N, const Tuning& tun);

}  // namespace ccvm
