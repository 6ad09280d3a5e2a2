// The fitted / measured constants of the launch policy (ccvm_plan.hip) -- a file of numbers only, WRITTEN BY
// tools/fit_tile_model.py --write: it refits TILE_FIT from profiles/r05_policy_regret_first_pass.jsonl +
// profiles/r05_policy_regret.jsonl and replaces the block below (a re-fit on another box: tools/policy_regret.py there for
// the data, then the tool; tests/test_launch_policy.py is the gate).  The block as committed is round 5's fit of the same
// data -- the policy every audit of that round ran under; today's refit moves no coefficient by more than 3 % and has the
// same error statistics (the tool prints both) -- so splitting the planner out of ccvm_abi.hip changed no plan.  The
// other tables are measurements, each with its source.
#pragma once
namespace ccvm {
// one round (tiles <= CUs): t = l0 + l1 N + tiles / CUs (m0 + m1 N); several: ceil(tiles / CUs) (a N + b + q 1e-6 N^2) + e
struct TileFit { double l0, l1, m0, m1, a, b, e, q; };
constexpr TileFit TILE_FIT[3][3] = {  // [DL, MF, Langevin / pumped Langevin][32 x 128, 32 x 64, 32 x 32]
    {{5.989, 0.02665, 0.864, 0.00062, 0.02783, 5.196, 0.046, -0.243}, {4.411, 0.01362, 0.862, 0.00064, 0.01387, 2.241, 2.017, -0.050}, {3.822, 0.00674, 0.626, 0.00050, 0.00701, 1.544, 2.199, 0.086}},
    {{5.299, 0.01336, 1.111, 0.00080, 0.01419, 5.922, -0.338, -0.167}, {3.931, 0.00667, 0.751, 0.00058, 0.00772, 2.558, 0.295, -0.220}, {3.962, 0.00355, -0.466, 0.00183, 0.00413, 1.572, 1.712, -0.135}},
    {{4.220, 0.01350, 0.356, 0.00080, 0.01390, 3.206, 0.717, 0.002}, {3.919, 0.00678, 0.035, 0.00078, 0.00695, 1.594, 1.753, 0.094}, {3.563, 0.00380, -0.294, 0.00072, 0.00361, 0.807, 2.761, 0.068}},
};
// the Adam variants relative to the plain ones, medians over the 84 audited cells (profiles/r05_policy_regret_adam.md)
constexpr double ADAM_TILE = 1.15, ADAM_TILE32 = 1.12, ADAM_CLUSTER = 1.175, ADAM_PTILE = 1.10;
// cluster kernel: us per step of a round of resident clusters by K = 320, 384, ... 768 in steps of 64 [DL, MF, Langevin],
// measured at B = 1000 (docs/kernel-cluster.md; profiles/r05_ab_cluster_half.txt for the odd multiples of 64)
constexpr double CLUSTER_ROUND_US[8][3] = {{7.16, 3.49, 3.37}, {7.9, 4.05, 3.77}, {9.57, 4.60, 4.40}, {10.1, 5.30, 4.85},
                                           {16.8, 8.69, 8.32}, {18.1, 9.5, 8.9},  {20.2, 10.6, 10.1}, {21.8, 11.2, 10.7}};
// a step on two row sets relative to three above K = 512: XCD by XCD / spread over the XCDs (profiles/r05_ab_cluster_sets.txt)
constexpr double CLUSTER_TWO_SETS = 0.70, CLUSTER_TWO_SETS_SPREAD = 0.80;
constexpr double CLUSTER_MARGIN = 0.98;  // (the audit: 0.95 kept six cells on the cluster path that one round of 32 x 64 tiles beats by 5-7 %; 1.0 loses Langevin + Adam N = 640, B = 2000 by 9 %)
// the row-owner kernel's five waves side by side (256 < N <= 320): us per step of a round of one row set per CU, by K chunks
// 17 ... 20 [DL, Langevin / pumped Langevin], flat in the batch up to a row set per CU (profiles/r06_ab_persist_xs.txt,
// r06_ab_persist_xs_delta.txt: the unequal K split; with equal halves 2.03 ... 2.24 / 2.10 ... 2.31: r06_ab_persist_wide.txt)
constexpr double PERSIST_WIDE_ROUND_US[2][4] = {{1.75, 1.84, 1.90, 2.01}, {1.86, 1.94, 2.00, 2.17}};
// ... MF (17 chunks only: 2.27; 2.30-2.34 with equal halves), and Langevin + Adam relative to Langevin (17 / 18 chunks: 2.27 / 2.39;
// 2.50 / 2.56 with equal halves) -- profiles/r06_ab_persist_wide2.txt, r06_ab_persist_xs2.txt
constexpr double PERSIST_WIDE_ROUND_MF_US = 2.27, PERSIST_WIDE_ADAM = 1.23;
// a resident round of 32 x 128 tiles, us per step (fits of the regime map: DL 30.9 at N = 1000, 59 at N = 2000)
constexpr double PTILE_DL_PER_N = 0.0281, PTILE_DL_0 = 2.8, PTILE_ONE_PER_N = 0.0145, PTILE_ONE_0 = 1.4, PTILE_MF_EXTRA = 0.4;
}  // namespace ccvm
