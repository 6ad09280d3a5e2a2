/*
 * ccvm_hip.h -- C ABI of libccvm_hip.so, the MI355X (gfx950) dynamics engine for
 * the CCVM BoxQP solvers.
 *
 * The reference (1QB-Information-Technologies/ccvm) has no FFI: its hot path is
 * the Python loop body of  DLSolver._solve / MFSolver._solve[_adam] /
 * LangevinSolver._solve[_adam] / PumpedLangevinSolver._solve[_adam].  Each entry
 * point below replaces one of those loop bodies (file:line cited per function,
 * relative to the reference tree) with `nsteps` fused Euler-Maruyama steps on the
 * GPU.  A binding is a ctypes/cffi stub that passes `tensor.data_ptr()` values;
 * see INTEGRATION.md.
 *
 * Conventions
 *  - Plain pointers and sizes only; every pointer is a DEVICE pointer unless a
 *    parameter is documented as host.  The caller owns every buffer.
 *  - Return value: 0 on success, a negative ccvm_status otherwise; the message is
 *    available from ccvm_last_error() (thread-local).  Nothing throws.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 *    the default stream).  No call synchronises, allocates or frees device memory.
 *  - No global mutable state: N host threads / processes can drive N GPUs.
 *
 * Padded layout ("pitched" arrays)
 *  - A batch x N state array is [rows_pad][ld] floats, row-major, with
 *    ld = ccvm_ld(N) (N rounded up to 128) and rows_pad = ccvm_rows(B) (B rounded
 *    up to 64).  The coupling matrix Q is [ld][ld] and V is [ld].  All padding
 *    MUST be zero on entry (ccvm_pack does that) and is kept zero by the kernels.
 *  - Q is used as  x @ Q  (reference einsum "bi,ij->bj"), i.e. Q[i][j] couples
 *    input column i to output column j; symmetry is never assumed.
 */
#ifndef CCVM_HIP_H
#define CCVM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCVM_ABI_VERSION 10  /* 8: workspaces above N = 768 carry the persistent tile kernel's flag lines */

typedef enum ccvm_status {
    CCVM_OK = 0,
    CCVM_E_INVALID = -1,     /* bad argument (NULL pointer, negative size, step range) */
    CCVM_E_LAYOUT = -2,      /* ld / rows do not match ccvm_ld / ccvm_rows, or misaligned */
    CCVM_E_WORKSPACE = -3,   /* workspace too small */
    CCVM_E_HIP = -4,         /* a HIP runtime call failed */
    CCVM_E_UNSUPPORTED = -5  /* valid request the engine does not implement */
} ccvm_status;

/* Wiener-noise source for one call. */
typedef enum ccvm_noise_mode {
    CCVM_NOISE_FUSED = 0,  /* counter-based generator fused into the step kernel: Threefry2x32-13 on
                              counter (column, global row), key = SplitMix64 hash of (seed, step),
                              then Box-Muller (ccvm_amd/csrc/ccvm_noise.h)                     */
    CCVM_NOISE_PHILOX = 0, /* historical alias of CCVM_NOISE_FUSED (the generator is Threefry)    */
    CCVM_NOISE_REPLAY = 1  /* read standard normals the caller generated (parity mode) */
} ccvm_noise_mode;

/* Run flags (ccvm_noise::flags; the struct every run call takes).
 * CCVM_RUN_WS_PADDED: the caller guarantees that this workspace's scratch state arrays still have zero padding,
 * i.e. the workspace was zero-filled, or a run call of the same entry point, B and N completed on it, and nothing
 * else wrote to it since.  The call then skips re-zeroing them (8 MB and ~6 us per call at N = 1000, B = 1000:
 * 2 % of a 20-step call).  Without the flag every call zeroes them itself. */
#define CCVM_RUN_WS_PADDED 1
/* CCVM_RUN_NO_EXCHANGE: run this call on kernels whose workgroups do not wait for each other (the per-step tile kernel,
 * or the row-owner persistent kernel for N <= 256), never on the column-cluster / column-slab / persistent-tile kernels.  What
 * a caller sets to repeat a chunk whose status word reported a time-out (ccvm_status_offset): same noise, same
 * result up to the summation order of the contraction. */
#define CCVM_RUN_NO_EXCHANGE 2
/* CCVM_RUN_FORWARD: the caller guarantees that this workspace has never run a step behind this call's first one (a
 * trajectory advanced chunk by chunk on a workspace that started zeroed; NOT a re-run of earlier steps).  Only read
 * together with a caller-made schedule table (`schedule` in the parameters): the persistent tile kernel's flag lines
 * then need no initialising launch, and the run call of a persistent path is a single kernel launch.  A broken promise
 * would be a silent read-before-publish race, so the library does not rely on it alone: it records, per flag area (host
 * memory, keyed by the area's address and size), the step behind the last chunk it launched there -- chunks whose
 * schedule rows it made itself included -- and a chunk that does not start exactly at that step -- or a first chunk at
 * step > 0 on an area it has never seen -- gets its flag lines set whatever this bit says (one more launch, never a wrong
 * result). */
#define CCVM_RUN_FORWARD 4

typedef struct ccvm_noise {
    int32_t mode;        /* ccvm_noise_mode */
    int32_t flags;       /* CCVM_RUN_* bits (0: none) */
    uint64_t seed;       /* FUSED: 64-bit seed; any two seeds give unrelated streams at every step */
    int64_t row_offset;  /* FUSED: global index of local row 0 (batch sharding over GPUs:
                            a row's noise depends on its GLOBAL index only) */
    /* REPLAY: standard normals for steps step0 .. step0+nsteps-1, laid out as the
     * reference draws them -- per step an (N, B) block, batch-contiguous:
     *   W[step][b][n] = w[((step - step0) * N + n) * B + b]
     * (reference: Normal.sample((N,)).transpose(0,1), dl_solver.py:538-547).
     * w1 is the second stream of the DL solver (s quadrature); NULL otherwise. */
    const float* w0;
    const float* w1;
    /* REPLAY: pitch of the blocks in elements, W[step][b][n] = w[((step - step0) * N + n) * w_ld + b], for a call that
     * runs rows [r, r + B) of a larger batch out of that batch's blocks (pass w0 + r, w1 + r and the larger batch as
     * w_ld: what the library does itself for the two parts of a cut batch); 0 = B. */
    int64_t w_ld;
} ccvm_noise;

/* Optional Adam preconditioning of the feedback term (reference
 * solvers/algorithms.py:1-45 and e.g. mf_solver.py:717-738).  enabled = 0 selects
 * the original solver.  m (and v when beta2 != 1) are pitched B x N state arrays
 * owned by the caller, zero before step 0. */
typedef struct ccvm_adam {
    int32_t enabled;
    int32_t add_assign;
    double alpha, beta1, beta2;
    float* m;
    float* v; /* may be NULL when beta2 == 1.0 */
} ccvm_adam;

/* DL-CCVM: reference dl_solver.py:468-569 (_solve) + :117-172 (drift). */
typedef struct ccvm_dl_params {
    double pump, dt, noise_ratio, feedback_scale, g;
    double lower, upper;     /* solution_bounds */
    int32_t pump_rate_flag;  /* rate = (i+1)/T when set, else 1 */
    int32_t reserved;
    const float* qsum;       /* column sums of Q from ccvm_column_sums (ld floats, device), or NULL: computed by
                                every run call (two small kernels, ~15 us: worth passing for short chunks) */
    const float* schedule;   /* the schedule rows of the WHOLE run from ccvm_dl_schedule (device, ccvm_schedule_bytes),
                                made once with these parameters and the run calls' T, or NULL: the rows of a call's
                                steps are made by a small kernel in front of every launch of a persistent path
                                (~9 us per call at the headline shape: worth passing for short chunks) */
} ccvm_dl_params;

/* MF-CCVM: reference mf_solver.py:493-593 (_solve), :595-764 (_solve_adam),
 * :141-233 (drift / grads). */
typedef struct ccvm_mf_params {
    double pump, dt, j, feedback_scale, g, S;
    double lower, upper;
    int32_t pump_rate_flag;
    int32_t reserved;
    /* Per-variable saturation (the reference accepts S as a 1-D tensor of length N,
     * mf_solver.py:834-839): device array of ld floats, S_j > 0 for j < N; NULL selects the
     * scalar S above (which is ignored otherwise).  Needs the larger workspace of
     * ccvm_workspace_bytes_cols. */
    const float* s_cols;
    const float* qsum;       /* as in ccvm_dl_params (always the sums of the ORIGINAL Q, also with s_cols) */
    /* One saturation per trajectory AND variable (the reference passes a 2-D tensor S straight through,
     * mf_solver.py:834-839): pitched [rows][ld] device array, S > 0 on the logical B x N region; exclusive with
     * s_cols.  1 / S_bk then sits inside the GEMM's input map per element: this rare input runs on a composed
     * path (one GEMM launch + one elementwise launch per step) instead of the fused kernels. */
    const float* s_full;
    const float* schedule;   /* as in ccvm_dl_params, from ccvm_mf_schedule (made with the SAME ccvm_adam settings the
                                run calls pass), or NULL */
} ccvm_mf_params;

/* Langevin (use_pump = 0): reference langevin_solver.py:368-435, :437-561, :117-166.
 * Pumped Langevin (use_pump = 1): pumped_langevin_solver.py:232-309, :311-449,
 * :95-147. */
typedef struct ccvm_langevin_params {
    double dt, sigma, feedback_scale, S;
    double pump;             /* pumped Langevin only */
    double lower, upper;
    int32_t use_pump;
    int32_t pump_rate_flag;  /* p_i = pump*(i+1)/T when set, else pump */
    const float* s_cols;     /* per-variable saturation, as in ccvm_mf_params (langevin_solver.py:630-635,
                                pumped_langevin_solver.py:519-524); NULL = scalar S */
    const float* qsum;       /* as in ccvm_dl_params */
    const float* s_full;     /* as in ccvm_mf_params (langevin_solver.py:630-635, pumped_langevin_solver.py:519-524) */
    const float* schedule;   /* as in ccvm_dl_params, from ccvm_langevin_schedule (made with the SAME ccvm_adam
                                settings the run calls pass), or NULL */
} ccvm_langevin_params;

/* ---- library / layout ------------------------------------------------------- */
int ccvm_abi_version(void);
const char* ccvm_last_error(void);
int ccvm_ld(int N);    /* leading dimension for problem size N   */
int ccvm_rows(int B);  /* padded row count for batch size B      */

/* Zero-padded copy of a compact row-major [rows][cols] array (row stride src_ld)
 * into a pitched [dst_rows][dst_ld] array, and the inverse. */
int ccvm_pack(const float* src, int rows, int cols, int src_ld,
              float* dst, int dst_rows, int dst_ld, void* stream);
int ccvm_unpack(const float* src, int src_ld,
                float* dst, int rows, int cols, int dst_ld, void* stream);

/* ---- the hot path: nsteps fused Euler-Maruyama steps ------------------------- */
/* Bytes of caller-provided scratch a call needs (ping-pong state, the MF measured-amplitude
 * buffers, column sums of Q, schedule table; for 256 < N <= 768 the column-cluster path's exchange
 * buffers of 8-byte {value, tag} packets, and the column-slab path's for small batches above N = 256; a status
 * line; above N = 768, for a batch larger than one resident grid of the persistent tile kernel, behind all that the
 * workspaces of the two parts a run call may cut the batch into -- the rows of whole resident grids and the rest, run
 * as two calls on the same stream; a call given less than this but enough for the uncut batch runs
 * uncut).  `what`: 0 ccvm_dl_run, 1 ccvm_mf_run,
 * 2 ccvm_langevin_run, 3 ccvm_energy, 4 ccvm_pp_*, 5 ccvm_feedback. */
size_t ccvm_workspace_bytes(int solver, int B, int N);
/* Byte offset, inside the workspace of ccvm_dl_run (solver 0) / ccvm_mf_run (1) / ccvm_langevin_run (2),
 * of a 128-byte line whose first 4 bytes are the run's status word; (size_t)-1 for the other entries.  The caller
 * zeroes the WHOLE workspace once when it is fresh (the parts of a cut batch keep lines of their own behind the
 * batch's workspace) and may read the status word after
 * synchronising the stream: 0 = ok; 1 = a bounded in-kernel wait of a persistent path whose workgroups exchange data
 * (column-cluster kernel, 256 < N <= 768; column-slab kernel, small batches above N = 256; persistent tile kernel, full
 * grids of 32 x 128 tiles above N = 768) gave up because its
 * workgroups could not become resident (another process holding the GPU for ~1 s; in a cut batch: in either part) -- the
 * state arrays are then
 * invalid: restore them and repeat the steps with CCVM_RUN_NO_EXCHANGE.  Run calls never clear the status word.  The
 * rest of the line is the library's: it records what the exchange area in front of it holds, so that a later call
 * on the same workspace with later steps does not clear the area again. */
size_t ccvm_status_offset(int solver, int B, int N);
/* The same plus room for the row-scaled copy of Q a run with per-variable saturation (s_cols) makes. */
size_t ccvm_workspace_bytes_cols(int solver, int B, int N);

/* The per-step schedule scalars of a whole run (T steps: pump and noise ramps, dl_solver.py:524-527;
 * mf_solver.py:550-559, pumped_langevin_solver.py:279-282; Adam bias corrections, langevin_solver.py:519-540), made once on the device in
 * fp64 exactly as the run calls make them per chunk: pass `table` as `schedule` in the parameters of every run call of
 * that run (same parameters, same T, same Adam settings).  ccvm_schedule_bytes: bytes of `table` for solver 0 (DL),
 * 1 (MF) and 2 (Langevin / pumped Langevin); 0 for the other entries. */
size_t ccvm_schedule_bytes(int solver, int T);
int ccvm_dl_schedule(const ccvm_dl_params* p, int T, float* table, void* stream);
int ccvm_mf_schedule(const ccvm_mf_params* p, const ccvm_adam* adam, int T, float* table, void* stream);
int ccvm_langevin_schedule(const ccvm_langevin_params* p, const ccvm_adam* adam, int T, float* table, void* stream);

/* Column sums of Q (the constant term of the folded affine input map: (x a + b) @ Q = a (x @ Q) + b colsum(Q)),
 * deterministic two-pass reduction; qsum: ld floats.  Optional: pass the result in the run parameters' `qsum`
 * field to save the per-call recomputation.  workspace: ccvm_workspace_bytes(5, 1, N). */
int ccvm_column_sums(const float* Q, int N, int ld, float* qsum,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Which kernel instantiation and grid ccvm_dl_run (solver 0) / ccvm_mf_run (1) / ccvm_langevin_run (2)
 * launch for this shape under the current tuning environment, as text (the name is what rocprofv3
 * prints for the kernel): benchmark lines and profiles name the kernel that actually ran.  Host only. */
int ccvm_describe_launch(int solver, int B, int N, int adam, int per_variable_s, char* buf, size_t buf_len);

/* Steps step0 .. step0+nsteps-1 of a T-step DL-CCVM run, in place on c and s.
 * Chunking a run into several calls does not change the result.  The final clamp
 * (dl_solver.py:567) is NOT applied here; see ccvm_clamp. */
int ccvm_dl_run(const float* Q, const float* V, float* c, float* s,
                int B, int N, int ld, int step0, int nsteps, int T,
                const ccvm_dl_params* params, const ccvm_noise* noise,
                void* workspace, size_t workspace_bytes, void* stream);

/* MF-CCVM steps, in place on mu and sigma.  mu_tilde_out (pitched, may be NULL)
 * receives the clamped measured amplitude of the LAST executed step, which is what
 * the reference returns and scores (mf_solver.py:591-593). */
int ccvm_mf_run(const float* Q, const float* V, float* mu, float* sigma,
                float* mu_tilde_out,
                int B, int N, int ld, int step0, int nsteps, int T,
                const ccvm_mf_params* params, const ccvm_adam* adam,
                const ccvm_noise* noise,
                void* workspace, size_t workspace_bytes, void* stream);

/* Langevin / pumped-Langevin steps, in place on c (clamped to [-S, S] every step). */
int ccvm_langevin_run(const float* Q, const float* V, float* c,
                      int B, int N, int ld, int step0, int nsteps, int T,
                      const ccvm_langevin_params* params, const ccvm_adam* adam,
                      const ccvm_noise* noise,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- the steps right after the loop ------------------------------------------ */
/* x = clamp(x, lo, hi) on the logical B x N region (fit_to_constraints,
 * dl_solver.py:237-250). */
int ccvm_clamp(float* x, int B, int N, int ld, float lo, float hi, void* stream);

/* The two above with a per-variable saturation S_j (s_cols: device array of ld floats):
 * x = clamp(x, -S_j, S_j) and y = 0.5 * x / S_j * (upper - lower) + 0.5 * (upper + lower)
 * (the reference repeats a 1-D S to (B, N), dl_solver.py:843-848). */
int ccvm_clamp_cols(float* x, int B, int N, int ld, const float* s_cols, void* stream);
int ccvm_change_variables_cols(const float* x, float* y, int B, int N, int ld,
                               const float* s_cols, double lower, double upper, void* stream);

/* The same with one bound / saturation per trajectory AND variable (pitched [rows][ld] arrays): the
 * reference passes a non-1-D tensor S straight to the elementwise ops (dl_solver.py:843-848) and
 * torch.clamp takes tensor bounds (dl_solver.py:237-250). */
int ccvm_clamp_full(float* x, int B, int N, int ld, const float* lo, const float* hi, void* stream);
int ccvm_change_variables_full(const float* x, float* y, int B, int N, int ld,
                               const float* s_full, double lower, double upper, void* stream);

/* y = 0.5 * x / S * (upper - lower) + 0.5 * (upper + lower)   (change_variables,
 * dl_solver.py:219-235).  y may alias x. */
int ccvm_change_variables(const float* x, float* y, int B, int N, int ld,
                          double S, double lower, double upper, void* stream);

/* obj[b] = (0.5 * x_b Q x_b + V . x_b) * scaled_by   (problem_instance.py:226-241).
 * obj is a compact array of B floats.  workspace: ccvm_workspace_bytes(3, B, N). */
int ccvm_energy(const float* Q, const float* V, const float* x,
                int B, int N, int ld, double scaled_by, float* obj,
                void* workspace, size_t workspace_bytes, void* stream);

/* Success statistics of a batch, computed on the device (solution.py:65-85 best_objective_value,
 * :87-146 the seven gap thresholds 0.1, 1, 2, 3, 4, 5, 10 percent): found_b = -obj_b,
 * gap_b = (optimal_value - found_b) * 100 / |found_b| in fp32, within[k] = #{b: gap_b <= thr_k}.
 * The reference's fractions are round(within[k] / rows, 4).  A NaN objective value counts in no
 * threshold and makes best_objective_value NaN (torch.max). */
typedef struct ccvm_solution_stats {
    float best_objective_value;
    int32_t within[7];
    int32_t rows;
    int32_t nonfinite;  /* rows with a NaN / infinite objective value */
} ccvm_solution_stats;

/* stats: DEVICE pointer (8-byte aligned), written on `stream`.  obj: B floats (device). */
int ccvm_objective_stats(const float* obj, int B, double optimal_value,
                         ccvm_solution_stats* stats, void* stream);

/* Everything between the loop and the Solution, on the pitched state in place, without leaving the
 * device (dl_solver.py:567 clamp, :956-959 change_variables + compute_energy, problem_instance.py:226-241,
 * solution.py:65-146):
 *   1. clamp != 0:             state = clamp(state, clamp_lo, clamp_hi)  (with s_cols / s_full: to -S .. S), in place;
 *   2. change_variables != 0:  x = 0.5 * state / S * (upper - lower) + 0.5 * (upper + lower)
 *                              (S scalar, per-variable s_cols or per-element s_full); else x = state.
 *                              x may alias state;
 *   3. obj[b] = (0.5 * x_b Q x_b + V . x_b) * scaled_by        (same kernels and summation order as ccvm_energy);
 *   4. stats (device pointer, may be NULL): ccvm_objective_stats(obj, optimal_value).
 * Only obj (B floats) and stats (40 bytes) need to leave the GPU.  workspace: ccvm_workspace_bytes(3, B, N). */
typedef struct ccvm_finalize_params {
    double S;              /* saturation of the change of variables (ignored with s_cols) */
    const float* s_cols;   /* per-variable saturation (ld floats, device) or NULL */
    const float* s_full;   /* per-trajectory-and-variable saturation (pitched, device) or NULL */
    double lower, upper;   /* solution bounds of the change of variables */
    double clamp_lo, clamp_hi;
    double scaled_by;      /* ProblemInstance.scaled_by */
    double optimal_value;  /* ProblemInstance.optimal_sol */
    int32_t clamp;
    int32_t change_variables;
} ccvm_finalize_params;

int ccvm_finalize(const float* Q, const float* V, float* state, float* x,
                  int B, int N, int ld, const ccvm_finalize_params* params,
                  float* obj, ccvm_solution_stats* stats,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The bare feedback term of every solver ("Qx matvec + V bias"):
 *   y = f_q * ((x * in_scale + in_shift) @ Q) + f_v * V
 * e.g. MFSolver._calculate_grads_boxqp (mf_solver.py:200-233) is in_scale = (u-l)/S,
 * in_shift = u+l, f_q = -fs (u-l)/(4S), f_v = -fs (u-l)/(2S).  y must not alias x and
 * must have zero padding on entry (it is only written on the logical B x N region).
 * workspace: ccvm_workspace_bytes(5, B, N). */
int ccvm_feedback(const float* Q, const float* V, const float* x, float* y,
                  int B, int N, int ld, double in_scale, double in_shift,
                  double f_q, double f_v,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Post-processors.  x is updated in place.
 * grad-descent: `iters` times  x <- clamp(x - step * (x Q + V), lo, hi)
 *   (post_processor/grad_descent.py:58-64).
 * adam: one torch.optim.Adam(lr, betas) step from zero moments on 1/2 xQx + Vx,
 *   then clamp (post_processor/adam.py:58-66); closed form
 *   x <- clamp(x - lr * g / (|g| + eps), lo, hi),  g = 1/2 (Q + Q') x + V.
 * workspace: ccvm_workspace_bytes(4, B, N). */
int ccvm_pp_grad_descent(const float* Q, const float* V, float* x,
                         int B, int N, int ld, int iters, double step,
                         double lo, double hi,
                         void* workspace, size_t workspace_bytes, void* stream);
int ccvm_pp_adam(const float* Q, const float* V, float* x,
                 int B, int N, int ld, double lr, double eps,
                 double lo, double hi,
                 void* workspace, size_t workspace_bytes, void* stream);
/* asgd: the first step of torch.optim.ASGD(lr, lambd) on 1/2 xQx + Vx, then clamp
 *   (post_processor/asgd.py): x <- clamp(x * (1 - lambd * lr) - lr * g, lo, hi), g as for adam.
 *   (In the reference only the first optimizer step of adam / asgd ever takes effect: the Parameter is
 *   replaced after every step while the optimizer keeps the original, so num_iter > 1 changes nothing.) */
int ccvm_pp_asgd(const float* Q, const float* V, float* x,
                 int B, int N, int ld, double lr, double lambd,
                 double lo, double hi,
                 void* workspace, size_t workspace_bytes, void* stream);
/* lbfgs: `iters` times a fresh torch.optim.LBFGS(lr, max_iter=1) step per row, then clamp
 *   (post_processor/lbfgs.py): one steepest-descent step of length lr * min(1, 1 / |g|_1),
 *   x <- clamp(x - lr * min(1, 1 / |g|_1) * g, lo, hi), g as for adam; a row whose gradient is below
 *   LBFGS's tolerances (max|g| <= 1e-7 or g.g < 1e-9) does not move. */
int ccvm_pp_lbfgs(const float* Q, const float* V, float* x,
                  int B, int N, int ld, int iters, double lr,
                  double lo, double hi,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---- noise generator, exposed for tests -------------------------------------- */
/* Fill w0 (and w1 if not NULL) with the standard normals the fused PHILOX mode uses
 * at `step`, in the REPLAY layout [N][B] (so the two modes can be cross-checked). */
int ccvm_philox_normals(uint64_t seed, int64_t row_offset, int step,
                        int B, int N, float* w0, float* w1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCVM_HIP_H */
