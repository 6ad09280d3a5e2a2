// C ABI of libccvm_hip.so (declared in include/ccvm_hip.h): argument validation,
// per-step schedule scalars (fp64 on the host, as the reference computes them with
// Python/numpy doubles), and kernel launches.  No allocation, no synchronisation.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <map>
#include <unordered_map>

#include "../../include/ccvm_hip.h"
#include "ccvm_cluster.h"
#include "ccvm_kernels.h"
#include "ccvm_persist_launch.h"
#include "ccvm_ptile.h"
#include "ccvm_schedule.h"
#include "ccvm_slab.h"
#include "ccvm_plan.h"

using namespace ccvm;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define CCVM_CHECK_LAUNCH(name)                                                         \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) return fail(CCVM_E_HIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

inline int ew_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_layout(const char* fn, int B, int N, int ld) {
    if (B <= 0 || N <= 0) return fail(CCVM_E_INVALID, "%s: B=%d N=%d must be positive", fn, B, N);
    if (ld != ccvm_ld(N)) return fail(CCVM_E_LAYOUT, "%s: ld=%d but ccvm_ld(%d)=%d", fn, ld, N, ccvm_ld(N));
    return CCVM_OK;
}

int check_steps(const char* fn, int step0, int nsteps, int T) {
    if (step0 < 0 || nsteps < 0 || T <= 0 || step0 + nsteps > T)
        return fail(CCVM_E_INVALID, "%s: step range [%d, %d) outside a %d-step run", fn, step0, step0 + nsteps, T);
    return CCVM_OK;
}

// REPLAY: the pitch of the noise blocks, element (step, j, b) at w[(step N + j) w_ld + b] (0: the batch itself)
size_t noise_pitch(const ccvm_noise* nz, int B) { return nz->w_ld > 0 ? (size_t)nz->w_ld : (size_t)B; }
int check_noise(const char* fn, const ccvm_noise* nz, bool two_streams, int B) {
    if (!nz) return fail(CCVM_E_INVALID, "%s: noise is NULL", fn);
    if (nz->mode == CCVM_NOISE_REPLAY) {
        if (!nz->w0 || (two_streams && !nz->w1))
            return fail(CCVM_E_INVALID, "%s: REPLAY noise needs w0%s", fn, two_streams ? " and w1" : "");
        if (nz->w_ld != 0 && (nz->w_ld < B || nz->w_ld > 0x7FFFFFFF))
            return fail(CCVM_E_INVALID, "%s: REPLAY noise pitch w_ld must be 0 or at least the batch", fn);
    } else if (nz->mode != CCVM_NOISE_PHILOX) {
        return fail(CCVM_E_INVALID, "%s: unknown noise mode %d", fn, nz->mode);
    }
    return CCVM_OK;
}

int check_adam(const char* fn, const ccvm_adam* ad) {
    if (!ad || !ad->enabled) return CCVM_OK;
    if (!ad->m) return fail(CCVM_E_INVALID, "%s: Adam enabled but m is NULL", fn);
    if (ad->beta2 != 1.0 && !ad->v) return fail(CCVM_E_INVALID, "%s: Adam with beta2 != 1 needs v", fn);
    return CCVM_OK;
}

void fill_adam(AdamScalars& s, const ccvm_adam* ad, int i) {
    s.beta1 = (float)ad->beta1;
    s.one_m_beta1 = (float)(1.0 - ad->beta1);
    s.inv_bc1 = (float)(1.0 / (1.0 - std::pow(ad->beta1, (double)(i + 1))));
    s.use_v = ad->beta2 != 1.0;
    s.beta2 = (float)ad->beta2;
    s.one_m_beta2 = (float)(1.0 - ad->beta2);
    s.inv_bc2 = s.use_v ? (float)(1.0 / (1.0 - std::pow(ad->beta2, (double)(i + 1)))) : 1.0f;
    s.alpha = (float)ad->alpha;
    s.eps = 1e-8f;
    s.add_assign = ad->add_assign;
}

// (the launch policy -- tuning environment, chip geometry, the cost models and every want_* / plan_* function -- lives in
// ccvm_plan.hip, its fitted constants in the generated ccvm_plan_model.h: VERDICT r5 item 8)
// Column sums of Q into `area` ((QSUM_SLICES + 1) * ld floats); returns the qsum pointer.
size_t qsum_area_bytes(int N) { return (size_t)(QSUM_SLICES + 1) * ccvm_ld(N) * sizeof(float); }

// `given`: column sums the caller computed once for this Q (ccvm_column_sums): nothing to launch then.
int compute_qsum(const float* Q, int N, int ld, float* area, hipStream_t st, const float** out,
                 const float* given = nullptr) {
    if (given) {
        *out = given;
        return CCVM_OK;
    }
    float* part = area + ld;
    hipLaunchKernelGGL(qsum_partial_kernel, dim3((ld + 127) / 128, QSUM_SLICES), dim3(128), 0, st, Q, N, ld, part);
    hipLaunchKernelGGL(qsum_final_kernel, dim3((ld + 127) / 128), dim3(128), 0, st, part, ld, area);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CCVM_E_HIP, "column sums of Q: %s", hipGetErrorString(e));
    *out = area;
    return CCVM_OK;
}

// Per-variable saturation: the row-scaled copy Qs[k][j] = Q[k][j] / S_k at the end of the workspace.
const float* scaled_rows(const float* Q, const float* s_cols, int N, int ld, void* ws, size_t base_bytes,
                         hipStream_t st) {
    float* qs = reinterpret_cast<float*>(static_cast<char*>(ws) + base_bytes);
    const size_t total = (size_t)ld * ld;
    size_t g = (total + 255) / 256;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, st, Q, s_cols, qs, N, ld);
    return qs;
}

void set_noise(StepArgs& a, const ccvm_noise* nz, int i, int step0, int B, int N, bool two, bool next) {
    a.step = i;
    a.seed = nz->seed;
    a.row_offset = nz->row_offset;
    a.replay = nz->mode == CCVM_NOISE_REPLAY;
    if (a.replay) {
        const size_t blk = (size_t)N * noise_pitch(nz, B);
        a.wld = (int)noise_pitch(nz, B);
        a.w0 = nz->w0 + (size_t)(i - step0) * blk;
        a.w1 = two ? nz->w1 + (size_t)(i - step0) * blk : nullptr;
        a.w0n = next ? nz->w0 + (size_t)(i + 1 - step0) * blk : nullptr;
    }
}

template <int MODE, bool ADAM>
int launch_step(const StepArgs& a, hipStream_t st, const char* name) {
    const int ks = a.ks;
    const int grid = a.nrb * a.ncb;
    constexpr bool CAN_VS = (MODE == MODE_MF || MODE == MODE_LANGEVIN);
    if (ks == 4) {  // solver steps only (base_args(..., max_ks = 4))
        if constexpr (MODE == MODE_DL) tile4_launch_dl(a, grid, st);
        else if constexpr (MODE == MODE_MF) tile4_launch_mf(a, grid, ADAM, a.s_cols != nullptr, st);
        else if constexpr (MODE == MODE_LANGEVIN) tile4_launch_lv(a, grid, ADAM, a.s_cols != nullptr, st);
        else return fail(CCVM_E_INVALID, "%s: no 32 x 32 tiles for this kernel", name);
        CCVM_CHECK_LAUNCH(name);
        return CCVM_OK;
    }
    if (CAN_VS && a.s_cols) {  // per-variable saturation
        if constexpr (CAN_VS) {
            if (ks == 2)
                hipLaunchKernelGGL((step_kernel<MODE, ADAM, 0, 2, true>), dim3(grid), dim3(WG_THREADS), 0, st, a);
            else
                hipLaunchKernelGGL((step_kernel<MODE, ADAM, 0, 1, true>), dim3(grid), dim3(WG_THREADS), 0, st, a);
        }
    } else if (ks == 2) {
        hipLaunchKernelGGL((step_kernel<MODE, ADAM, 0, 2>), dim3(grid), dim3(WG_THREADS), 0, st, a);
    } else {
        hipLaunchKernelGGL((step_kernel<MODE, ADAM, 0, 1>), dim3(grid), dim3(WG_THREADS), 0, st, a);
    }
    CCVM_CHECK_LAUNCH(name);
    return CCVM_OK;
}

// the launch status word (its own 128-byte line), last in the workspace
size_t cluster_sync_bytes(int) { return 128; }
// (cluster_exchange_bytes: ccvm_plan.hip -- want_cluster needs it for its 32-bit offset check)
// the persistent tile kernel's flag lines (ccvm_ptile.h): one 128-byte line of step counters per row block
size_t ptile_flag_bytes(int B, int N) {
    return N > CL_MAX_N ? (size_t)((B + BM - 1) / BM) * PT_FLAG_WORDS * sizeof(unsigned) : 0;
}
// the exchange area of a workspace serves whichever persistent path a call takes
size_t exchange_bytes(int B, int N, int planes) {
    return std::max(std::max(cluster_exchange_bytes(B, N, planes), slab_exchange_bytes(B, N, planes)), ptile_flag_bytes(B, N));
}
// ---- the exchange area of the cluster / slab paths -----------------------------------------------------------
// Before a call both exchange buffers must hold no tag the call will await: every packet {0, 0}; for the slab path the
// packets of the columns nobody owns (k >= Kx = G C: that kernel fetches all K columns of a block) {0, 0xFFFFFFFF}, a
// tag that no step awaits and the minimum over a unit's tags ignores.  Tags are global step numbers, so an area left
// by earlier run calls of the SAME layout whose steps all came before step0 is as good as a fresh one: the 128-byte
// line behind the area (word 0: the status word) records in words 8 / 9 what the area holds -- a layout id and the
// largest tag ever stored -- and the init kernel returns at once when that matches (8-25 MB of memset per call
// otherwise: as much as several steps of a short chunk).  Zero the whole line once (fresh workspace).
constexpr int XHDR_ID = 8, XHDR_MAXTAG = 9;
__global__ void exchange_init_kernel(uint2* xb, size_t packets, int K, int Kx, const unsigned* hdr, unsigned id,
                                     unsigned step0) {
    if (hdr[XHDR_ID] == id && hdr[XHDR_MAXTAG] <= step0) return;  // (read-only here: the commit kernel writes it)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < packets; i += (size_t)gridDim.x * blockDim.x) {
        const bool pad = K > 0 && (int)((i >> 2) % (size_t)K) >= Kx;
        xb[i] = make_uint2(0u, pad ? 0xFFFFFFFFu : 0u);
    }
}
// Between the init kernel and the commit kernel of a call the record is INVALID (id 0 matches no layout): a call that
// returns early -- a failed launch in the middle of its chunk loop -- leaves tags of its own steps in the area, and a
// retry of the same steps must clear them instead of taking them for the peers' packets.
__global__ void exchange_open_kernel(unsigned* hdr) { hdr[XHDR_ID] = 0u; }
__global__ void exchange_commit_kernel(unsigned* hdr, unsigned id, unsigned max_tag) {
    hdr[XHDR_ID] = id;
    hdr[XHDR_MAXTAG] = max_tag;
}
unsigned exchange_layout_id(std::initializer_list<int> what) {
    unsigned h = 2166136261u;  // FNV-1a over the layout's parameters; never 0 (a zeroed line matches nothing)
    for (int w : what) h = (h ^ (unsigned)w) * 16777619u;
    return h ? h : 1u;
}
int exchange_prepare(void* area, size_t bytes, int K, int Kx, unsigned* hdr, unsigned id, int step0, hipStream_t st) {
    const size_t packets = bytes / 8;
    hipLaunchKernelGGL(exchange_init_kernel, dim3(ew_grid(packets)), dim3(256), 0, st, static_cast<uint2*>(area), packets,
                       K, Kx, hdr, id, (unsigned)step0);
    hipLaunchKernelGGL(exchange_open_kernel, dim3(1), dim3(1), 0, st, hdr);  // stream order: after every init thread's read
    return hipGetLastError() == hipSuccess ? CCVM_OK : CCVM_E_HIP;
}
int exchange_commit(unsigned* hdr, unsigned id, int step_end, hipStream_t st) {
    hipLaunchKernelGGL(exchange_commit_kernel, dim3(1), dim3(1), 0, st, hdr, id, (unsigned)step_end);
    return hipGetLastError() == hipSuccess ? CCVM_OK : CCVM_E_HIP;
}

// the part of ClusterArgs every solver shares; `area` = what follows the schedule table in the workspace:
// [exchange buffer 0][exchange buffer 1][status word].  Zeroes the exchange buffers (once per call: the tags are
// global step numbers, unique across the launches of a call).
int cluster_base(ClusterArgs& ca, unsigned& xid, const float* Q, const float* V, const float* qsum, int B, int N, int ld,
                 const ccvm_noise* nz, float* table, void* area, int step0, hipStream_t st, const Tuning& tun,
                 int planes = 1) {
    std::memset(&ca, 0, sizeof(ca));
    ca.drop = tun.cluster_drop;
    ca.cus = chip_of(tun).cus; ca.xcds = chip_of(tun).xcds;
    // (a round of clusters: <= 22 us per step, docs/kernel-cluster.md)
    ca.spin_limit = spin_ticks(25.0, tun);
    ca.half_off = !tun.cluster_half;
    ca.Q = Q; ca.V = V; ca.qsum = qsum; ca.table = table;
    const size_t xb = cluster_exchange_bytes(B, N, planes);
    ca.xb0 = static_cast<float*>(area);
    ca.xb1 = reinterpret_cast<float*>(static_cast<char*>(area) + xb / 2);
    ca.status = reinterpret_cast<unsigned*>(static_cast<char*>(area) + exchange_bytes(B, N, planes));
    const ChipGeometry chip = chip_of(tun);
    ca.sets = cluster_sets(B, N, chip, tun.cluster_sets);
    xid = exchange_layout_id({1, B, N, planes, cluster_half(N, !tun.cluster_half) ? 1 : 0, ca.sets});
    if (exchange_prepare(area, xb, 0, 0, ca.status, xid, step0, st)) return CCVM_E_HIP;
    ca.seed = nz->seed; ca.row_offset = nz->row_offset; ca.replay = nz->mode == CCVM_NOISE_REPLAY;
    ca.B = B; ca.N = N; ca.ld = ld; ca.wld = (int)noise_pitch(nz, B);
    ca.nclusters = cluster_count(B, N, chip, tun.cluster_sets);
    ca.G = (N + CL_COLS - 1) / CL_COLS;
    ca.spread = cluster_spread(B, N, chip, tun.cluster_sets);
    return CCVM_OK;
}

// the part of SlabArgs every solver shares; `area` as in cluster_base: [exchange buffer 0][exchange buffer 1][status word]
int slab_base(SlabArgs& sa, unsigned& xid, const SlabPlan& p, const float* Q, const float* V, const float* qsum, int B, int N,
              int ld, const ccvm_noise* nz, float* table, void* area, int step0, hipStream_t st, const Tuning& tun,
              int planes) {
    std::memset(&sa, 0, sizeof(sa));
    sa.drop = tun.cluster_drop;
    sa.spin_limit = spin_ticks(p.est_us, tun);
    sa.Q = Q; sa.V = V; sa.qsum = qsum; sa.table = table;
    const size_t half = (size_t)p.nclusters * planes * p.rg * p.K * 4 * SL_XE;  // <= exchange_bytes / 2
    sa.xb0 = static_cast<float*>(area);
    sa.xb1 = reinterpret_cast<float*>(static_cast<char*>(area) + half);
    sa.status = reinterpret_cast<unsigned*>(static_cast<char*>(area) + exchange_bytes(B, N, planes));
    xid = exchange_layout_id({2, B, N, planes, p.cgrp, p.rg, p.K, p.nclusters, p.G});
    if (exchange_prepare(area, 2 * half, p.K, p.G * 4 * p.cgrp, sa.status, xid, step0, st)) return CCVM_E_HIP;
    sa.seed = nz->seed; sa.row_offset = nz->row_offset; sa.replay = nz->mode == CCVM_NOISE_REPLAY;
    sa.B = B; sa.N = N; sa.ld = ld; sa.wld = (int)noise_pitch(nz, B);
    sa.nclusters = p.nclusters; sa.G = p.G; sa.RG = p.rg; sa.span = p.span;
    sa.nxcd = chip_of(tun).xcds;
    sa.delay_fabric = tun.slab_delay >= 0 ? tun.slab_delay : slab_fabric_delay(planes, p.rg, p.K);
    sa.delay_fixed = tun.slab_delay >= 0;
    return CCVM_OK;
}

__global__ void status_merge_kernel(unsigned* whole, unsigned* part) {
    if (*part) { *whole = *part; *part = 0u; }
}
// ---- schedule rows made once per run by the caller (ccvm_dl_schedule / ccvm_langevin_schedule) ------------------
// A run call of a persistent path is then ONE launch: no schedule kernel in front of it (2 us + the 4-8 us a short
// kernel cannot hide of the next one's dispatch: 9 of the 657 us of a 20-step call at the headline shape).  The
// persistent tile kernel's flag lines must hold no step number beyond the call's first step: set here unless the
// caller says the workspace has only ever run earlier steps (CCVM_RUN_FORWARD).
__global__ void flags_init_kernel(unsigned* flags, int words, unsigned step0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) flags[i] = step0;
}
// CCVM_RUN_FORWARD is a promise, and a broken one is a silent read-before-publish race (a reader takes the flag a LATER
// step left behind for its peer's publication), so the library keeps its own books on the host: per flag area the
// step behind the last chunk launched on it.  A chunk that does not start at or behind that step gets its flag lines
// set whatever the caller says; a workspace the library has never seen is covered by the contract that
// fresh workspaces are zeroed.  (Host memory behind a mutex, keyed by the device pointer AND the area's size -- the flag
// words are a function of the batch -- so that another run's workspace at a reused address is a new area unless it has
// the same shape; no device traffic, nothing the 8 host threads of 8 GPUs share but the lock.)  EVERY chunk that touches
// a flag area is recorded, those with library-made schedule rows too (their schedule kernel sets the lines: ADVICE r5 --
// a no-schedule run of steps 0..k followed by a schedule + FORWARD re-run from step 0 was an "unknown area, first chunk":
// init skipped over flags that stood at k), and only a chunk that starts EXACTLY where the last one ended skips the
// init (the chunks of a run are contiguous; a stale entry of a freed workspace then matches a new run only if that run
// resumes at the very step the old one stopped at, on a workspace of the same shape at the same address -- and what it
// finds there are zeroed lines: its first step's readers would wait for publications that never come, give up their
// bounded wait and the steps are repeated on the per-step kernel: a cost, never a wrong result).
bool forward_holds(const unsigned* flags, int words, int first, int k) {
    static std::mutex mu;
    static std::map<std::pair<const unsigned*, int>, int> reached;
    std::lock_guard<std::mutex> lock(mu);
    if (reached.size() > 4096) reached.clear();  // (bounded: forgetting an area costs its next call one launch)
    const auto key = std::make_pair(flags, words);
    auto it = reached.find(key);
    const bool ok = it != reached.end() ? first == it->second : first == 0;  // unknown area: only a run's first chunk may skip the init
    reached[key] = first + k;
    return ok;
}
const float* given_rows(const float* schedule, const ccvm_noise* nz, int first, int k, unsigned* flags, int words,
                        hipStream_t st) {
    if (!schedule) {
        if (flags) (void)forward_holds(flags, words, first, k);  // (the chunk's schedule kernel sets the lines to `first`)
        return nullptr;
    }
    if (flags) {
        const bool forward = forward_holds(flags, words, first, k) && (nz->flags & CCVM_RUN_FORWARD);
        if (!forward)
            hipLaunchKernelGGL(flags_init_kernel, dim3((words + 255) / 256), dim3(256), 0, st, flags, words, (unsigned)first);
    }
    return schedule + (size_t)first * TABLE_WORDS;
}
// blocks of a schedule-kernel launch that also initialises the flag lines of `nrb` row blocks
int ptile_sched_grid(int k, int nrb) { return (std::max(k, nrb * PT_FLAG_WORDS) + 255) / 256; }
// `area`: what follows the schedule table in the workspace ([flag lines][...][status line])
void persist_adam(PersistArgs& pa, AdamSched& sc, const ccvm_adam* adam, bool use_adam);
// `par`: which of the two input buffers the launch's first step reads.  MF: `st0` / `st1` = mu / sigma, `carry` = the
// normals of the launch's first step (mf_prepare_kernel).
template <int MODE>
int run_ptile(const StepArgs& a, float* const (&x0)[2], float* const (&x1)[2], const ccvm_noise* nz, const float* table,
              void* area, unsigned* status, int step0, int done, int k, const Tuning& tun, hipStream_t st,
              const char* fn, int par, const ccvm_adam* adam = nullptr, float* st0 = nullptr, float* st1 = nullptr,
              const float* carry = nullptr) {
    const PtilePlan plan = plan_ptile(a, tun, false, MODE);
    const bool use_adam = adam && adam->enabled;
    const int nrb_all = (a.B + BM - 1) / BM;
    for (int rb0 = 0; rb0 < nrb_all; rb0 += plan.rbs) {
        // one slice of whole row blocks: rows [r0, r0 + rows), a resident grid of its own over the k steps
        const int r0 = rb0 * BM, rows = std::min(a.B - r0, plan.rbs * BM);
        const size_t off = (size_t)r0 * a.ld;
        StepArgs g = a;
        g.B = rows;
        g.ks = 1;
        set_grid(g, tun, true);
        PtileArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.st0 = st0 ? st0 + off : nullptr;
        pa.st1 = st1 ? st1 + off : nullptr;
        pa.carry = carry ? carry + off : nullptr;
        if (use_adam) {
            PersistArgs tmp;
            std::memset(&tmp, 0, sizeof(tmp));
            AdamSched unused;
            persist_adam(tmp, unused, adam, true);
            pa.ad = tmp.ad;
            pa.am = tmp.am ? tmp.am + off : nullptr;
            pa.av = tmp.av ? tmp.av + off : nullptr;
        }
        pa.Q = a.Q; pa.V = a.V; pa.qsum = a.qsum; pa.table = table;
        pa.s_cols = a.s_cols;
        pa.x0[0] = x0[0] + off; pa.x0[1] = x0[1] + off;
        pa.x1[0] = x1[0] ? x1[0] + off : nullptr; pa.x1[1] = x1[1] ? x1[1] + off : nullptr;
        // (the flag lines of ALL row blocks were set to step0 by this chunk's schedule kernel: ptile_sched_grid)
        pa.flags = static_cast<unsigned*>(area) + (size_t)rb0 * PT_FLAG_WORDS;
        pa.status = status;
        pa.seed = nz->seed; pa.row_offset = nz->row_offset + r0; pa.replay = nz->mode == CCVM_NOISE_REPLAY;
        pa.wld = (int)noise_pitch(nz, a.B);
        if (pa.replay) {
            pa.w0 = nz->w0 + (size_t)done * a.N * pa.wld + r0;
            pa.w1 = nz->w1 ? nz->w1 + (size_t)done * a.N * pa.wld + r0 : nullptr;
        }
        pa.B = rows; pa.N = a.N; pa.ld = a.ld; pa.nrb = g.nrb; pa.ncb = g.ncb; pa.xr = g.xr; pa.xc = g.xc;
        pa.par = par;
        pa.step0 = step0 + done; pa.nsteps = k;
        pa.in_scale = a.in_scale; pa.in_shift = a.in_shift;
        // (a step of the resident grid: its flops at ~100 TFLOP/s)
        pa.spin_limit = spin_ticks(2.0 * (MODE == MODE_DL ? 2 : 1) * (double)a.N * a.N * rows / 1.0e8, tun);
        pa.drop = tun.cluster_drop;
        if constexpr (MODE == MODE_DL) ptile_launch_dl(pa, st);
        else if constexpr (MODE == MODE_MF) ptile_launch_mf(pa, use_adam, st);
        else ptile_launch_lv(pa, use_adam, st);
        CCVM_CHECK_LAUNCH(fn);
    }
    return CCVM_OK;
}

template <int MODE, bool ADAM>
int launch_persist(const PersistArgs& a, hipStream_t st, const char* name) {
    if constexpr (MODE == MODE_DL) persist_launch_dl(a, st);
    else if constexpr (MODE == MODE_MF) ADAM ? persist_launch_mf_adam(a, st) : persist_launch_mf(a, st);
    else ADAM ? persist_launch_lv_adam(a, st) : persist_launch_lv(a, st);
    CCVM_CHECK_LAUNCH(name);
    return CCVM_OK;
}

void persist_adam(PersistArgs& pa, AdamSched& sc, const ccvm_adam* adam, bool use_adam) {
    std::memset(&sc, 0, sizeof(sc));
    if (!use_adam) return;
    sc.enabled = 1;
    sc.beta1 = adam->beta1;
    sc.beta2 = adam->beta2;
    sc.use_v = adam->beta2 != 1.0;
    AdamScalars s;
    fill_adam(s, adam, 0);
    pa.ad = AdamConsts{s.beta1, s.one_m_beta1, s.beta2, s.one_m_beta2, s.alpha, s.eps, s.use_v, s.add_assign};
    pa.am = adam->m;
    pa.av = adam->v;
}


}  // namespace

extern "C" {

int ccvm_abi_version(void) { return CCVM_ABI_VERSION; }
const char* ccvm_last_error(void) { return g_err; }
int ccvm_ld(int N) { return N <= 0 ? 0 : round_up(N, 128); }
int ccvm_rows(int B) { return B <= 0 ? 0 : round_up(B, 64); }

}  // extern "C"
namespace {
// a run's workspace without the parts of a cut batch (below)
size_t workspace_plain(int solver, int B, int N) {
    const size_t ld = (size_t)ccvm_ld(N), rows = (size_t)ccvm_rows(B);
    const size_t state = rows * ld * sizeof(float);
    const size_t qs = qsum_area_bytes(N);  // column sums of Q (+ their slice partials)
    switch (solver) {
        // DL: c', s', the schedule table of the persistent paths, the cluster path's exchange buffers and status word
        case 0: return 2 * state + qs + table_bytes() + exchange_bytes(B, N, 2) + cluster_sync_bytes(B);
        // MF: measured-amplitude ping-pong + noise carry; Langevin: c' (+ one spare state); both: the cluster
        // path's exchange buffers, status word and counters
        case 1: return 3 * state + qs + table_bytes() + exchange_bytes(B, N, 1) + cluster_sync_bytes(B);
        case 2: return 2 * state + qs + table_bytes() + exchange_bytes(B, N, 1) + cluster_sync_bytes(B);
        case 3: return (ld / 32) * rows * sizeof(float); // energy: column-strip partials
        case 4: return state + ld * ld * sizeof(float);  // post-processors: x' + 1/2(Q+Q')
        case 5: return qs;                               // ccvm_feedback
        default: return 0;
    }
}

size_t part_offset(size_t bytes) { return (bytes + 255) / 256 * 256; }
}  // namespace
extern "C" {

// A batch that split_rows cuts in two carries the workspaces of its parts behind its own (whose layout, status word
// included, stays what it is without the cut: a call in replay mode or after a time-out runs uncut).
size_t ccvm_workspace_bytes(int solver, int B, int N) {
    const size_t plain = workspace_plain(solver, B, N);
    if (solver < 0 || solver > 2 || B <= 0 || N <= 0) return plain;
    Tuning tun = read_tuning();
    if (tun.split < 0) tun.split = 1;  // (whatever the estimates say at run time: room for the cut)
    tun.persist_wide = 0;              // (... and whichever solver variant runs: the Adam variants of 256 < N <= 320 are cut)
    tun.ptile = tun.ptile ? -1 : 0;
    tun.ks = 0;
    tun.slab = tun.slab ? -1 : 0;
    tun.cluster = tun.cluster ? -1 : 0;
    const int cut = split_rows(solver, B, N, tun);
    return cut ? part_offset(plain) + part_offset(workspace_plain(solver, cut, N)) + workspace_plain(solver, B - cut, N) : plain;
}

size_t ccvm_status_offset(int solver, int B, int N) {
    if (solver < 0 || solver > 2) return (size_t)-1;
    return workspace_plain(solver, B, N) - cluster_sync_bytes(B);
}

size_t ccvm_workspace_bytes_cols(int solver, int B, int N) {
    const size_t ld = (size_t)ccvm_ld(N);
    const size_t base = ccvm_workspace_bytes(solver, B, N);
    return (solver == 1 || solver == 2) ? base + ld * ld * sizeof(float) : base;
}

size_t ccvm_schedule_bytes(int solver, int T) {
    return solver >= 0 && solver <= 2 && T > 0 ? (size_t)T * TABLE_WORDS * sizeof(float) : 0;
}

// The rows of a whole run, by the kernels that make them per chunk inside the run calls (same device code, same bits).
int ccvm_dl_schedule(const ccvm_dl_params* p, int T, float* table, void* stream) {
    const char* fn = "ccvm_dl_schedule";
    if (!p || !table || T <= 0) return fail(CCVM_E_INVALID, "%s: NULL argument or T <= 0", fn);
    if (!(p->upper > p->lower) || !(p->dt > 0)) return fail(CCVM_E_INVALID, "%s: need upper > lower, dt > 0", fn);
    const double ul = p->upper - p->lower;
    const double Sd = p->pump > 1.0 ? std::sqrt(p->pump - 1.0) : 1.0;  // dl_solver.py:140-141
    DlSched sc{p->pump, p->dt, p->noise_ratio, p->feedback_scale, p->g, ul, Sd, p->pump_rate_flag, T, 0, T};
    hipLaunchKernelGGL(dl_schedule_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, sc, table);
    CCVM_CHECK_LAUNCH(fn);
    return CCVM_OK;
}

int ccvm_mf_schedule(const ccvm_mf_params* p, const ccvm_adam* adam, int T, float* table, void* stream) {
    const char* fn = "ccvm_mf_schedule";
    if (!p || !table || T <= 0) return fail(CCVM_E_INVALID, "%s: NULL argument or T <= 0", fn);
    if (!(p->upper > p->lower) || !(p->dt > 0) || !(p->s_cols || p->s_full || p->S > 0) || !(p->j > 0))
        return fail(CCVM_E_INVALID, "%s: need upper > lower, dt > 0, S > 0, j > 0", fn);
    const double S_eff = (p->s_cols || p->s_full) ? 1.0 : p->S;  // see ccvm_mf_run
    const bool use_adam = adam && adam->enabled;
    AdamSched asc;
    std::memset(&asc, 0, sizeof(asc));
    if (use_adam) {
        asc.enabled = 1;
        asc.beta1 = adam->beta1;
        asc.beta2 = adam->beta2;
        asc.use_v = adam->beta2 != 1.0;
    }
    MfSched sc{p->pump, p->dt, p->j, p->feedback_scale, p->g, S_eff, p->upper - p->lower, p->pump_rate_flag, T, 0, T, asc};
    hipLaunchKernelGGL(mf_schedule_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, sc, table);
    CCVM_CHECK_LAUNCH(fn);
    return CCVM_OK;
}

int ccvm_langevin_schedule(const ccvm_langevin_params* p, const ccvm_adam* adam, int T, float* table, void* stream) {
    const char* fn = "ccvm_langevin_schedule";
    if (!p || !table || T <= 0) return fail(CCVM_E_INVALID, "%s: NULL argument or T <= 0", fn);
    if (!(p->upper > p->lower) || !(p->dt > 0) || !(p->s_cols || p->s_full || p->S > 0))
        return fail(CCVM_E_INVALID, "%s: need upper > lower, dt > 0, S > 0", fn);
    const double S_eff = (p->s_cols || p->s_full) ? 1.0 : p->S;  // see ccvm_mf_run
    const double ul = p->upper - p->lower;
    const bool use_adam = adam && adam->enabled;
    AdamSched asc;
    std::memset(&asc, 0, sizeof(asc));
    if (use_adam) {
        asc.enabled = 1;
        asc.beta1 = adam->beta1;
        asc.beta2 = adam->beta2;
        asc.use_v = adam->beta2 != 1.0;
    }
    LvSched sc{p->dt, p->sigma, p->feedback_scale, S_eff, p->pump, ul, p->use_pump, p->pump_rate_flag, T, 0, T, asc};
    hipLaunchKernelGGL(lv_schedule_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, sc, table);
    CCVM_CHECK_LAUNCH(fn);
    return CCVM_OK;
}

int ccvm_column_sums(const float* Q, int N, int ld, float* qsum, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_column_sums";
    int rc;
    if (!Q || !qsum || !ws) return fail(CCVM_E_INVALID, "%s: NULL argument", fn);
    if ((rc = check_layout(fn, 1, N, ld))) return rc;
    if (ws_bytes < ccvm_workspace_bytes(5, 1, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    const float* out;
    if ((rc = compute_qsum(Q, N, ld, static_cast<float*>(ws), (hipStream_t)stream, &out))) return rc;
    if (hipMemcpyAsync(qsum, out, (size_t)ld * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: copy failed", fn);
    return CCVM_OK;
}

}  // extern "C"
namespace {
int describe_plan(int solver, int B, int N, int adam, int per_variable_s, char* buf, size_t buf_len);
}
extern "C" {
int ccvm_describe_launch(int solver, int B, int N, int adam, int per_variable_s, char* buf, size_t buf_len) {
    if (!buf || buf_len == 0 || solver < 0 || solver > 2 || B <= 0 || N <= 0)
        return fail(CCVM_E_INVALID, "ccvm_describe_launch: bad argument");
    const int rc = describe_plan(solver, B, N, adam, per_variable_s, buf, buf_len);
    if (rc != CCVM_OK || !std::strncmp(buf, "batch cut in two", 16)) return rc;  // (a cut batch: its parts carry theirs)
    // ... and the estimate the plan rests on (ccvm_plan.hip; the row-owner kernel: its fitted variant model, N <= 64 only)
    Tuning tun = read_tuning();
    tun.adam = adam && solver != 0;
    double est = 0.0;
    if (want_persist(N, tun, solver, B))
        est = persist_shape(solver, tun.adam, B, N, tun.persist_ru, tun.persist_kh, 4 * chip_of(tun).cus, tun.persist_pw,
                            tun.persist_rsw, tun.persist_cw).est_us;
    else
        est = plan_us(solver, B, N, tun);
    const size_t used = std::strlen(buf);
    if (est > 0.0 && used + 1 < buf_len) std::snprintf(buf + used, buf_len - used, "; estimated %.2f us per step", est);
    return CCVM_OK;
}
}  // extern "C"
namespace {
int describe_plan(int solver, int B, int N, int adam, int per_variable_s, char* buf, size_t buf_len) {
    Tuning tun = read_tuning();
    const bool ad = adam && solver != 0;
    tun.adam = ad;
    if (const int cut = per_variable_s ? 0 : split_rows(solver, B, N, tun)) {
        char first[512], rest[512];
        int rc;
        if ((rc = ccvm_describe_launch(solver, cut, N, adam, 0, first, sizeof(first)))) return rc;
        if ((rc = ccvm_describe_launch(solver, B - cut, N, adam, 0, rest, sizeof(rest)))) return rc;
        std::snprintf(buf, buf_len, "batch cut in two: rows 0-%d %s | rows %d-%d %s", cut - 1, first, cut, B - 1, rest);
        return CCVM_OK;
    }
    if (const SlabPlan sp = want_persist(N, tun, solver, B) ? SlabPlan{} : want_slab(B, N, tun, solver); sp.ok) {
        // the fourth template argument: launches of 512 steps or more of clusters that span XCDs calibrate their
        // fetch delay (ccvm_slab.h: slab_calibrates); shorter launches of the same shape run the `false` variant
        char where[48] = "";
        if (sp.span > 1) std::snprintf(where, sizeof(where), ", each over %d XCDs", sp.span);
        const bool cal = CCVM_SL_CALIBRATE && sp.span > 1 && tun.slab_delay < 0;
        std::snprintf(buf, buf_len, "ccvm::slab_kernel<%d, %d, %d, %s> grid %d x 256 threads (%d clusters of %d workgroups x %d columns, %d rows each, K = %d%s), up to %d steps per launch",
                      solver, sp.cgrp, sp.nq, cal ? "true" : "false", sp.grid, sp.nclusters, sp.G, 4 * sp.cgrp, 4 * sp.rg, sp.K,
                      where, TABLE_STEPS);
        return CCVM_OK;
    }
    if (!want_persist(N, tun, solver, B) && want_cluster(B, N, tun, solver, ad)) {
        const ChipGeometry chip = chip_of(tun);
        const int G = (N + CL_COLS - 1) / CL_COLS, count = cluster_count(B, N, chip, tun.cluster_sets);
        const bool spread = cluster_spread(B, N, chip, tun.cluster_sets);
        const bool two_wide = ccvm_ld(N) > CL_LDS_K && cluster_sets(B, N, chip, tun.cluster_sets) == 2;
        // a batch of more clusters than the chip holds runs as one launch per round of resident clusters (launch_cluster)
        const int per_xcd = chip.cus / chip.xcds / G;
        const int per_round = spread ? count : chip.xcds * std::max(1, per_xcd), rounds = (count + per_round - 1) / per_round;
        const int first = std::min(count, per_round);
        char how[96] = "";
        if (rounds > 1) std::snprintf(how, sizeof(how), " x %d launches of at most %d clusters", rounds, per_round);
        std::snprintf(buf, buf_len, "ccvm::cluster_kernel%s%s<%d, %s, %d, false> grid %d x 512 threads%s (%d clusters of %d workgroups%s), up to %d steps per launch",
                      cluster_half(N, !tun.cluster_half) ? "_half" : "", two_wide ? "_2sets" : "", solver, ad ? "true" : "false",
                      ccvm_ld(N) / CL_KC, spread ? first * G : (first + 7) / 8 * 8 * G, how, count, G,
                      spread ? ", spread over the XCDs" : "", TABLE_STEPS);
        return CCVM_OK;
    }
    if (want_persist(N, tun, solver, B)) {
        const PersistShape sh = persist_shape(solver, ad, B, N, tun.persist_ru, tun.persist_kh, 4 * chip_of(tun).cus,
                                              tun.persist_pw, tun.persist_rsw, tun.persist_cw);
        if (sh.ncg == 5)
            if (tun.persist_xs == 1 || persist_wide_xs(solver, ad, sh.nch) == 0)
                std::snprintf(buf, buf_len, "ccvm::persist_kernel<%d, %s, %d, %d, %d, %d, %d, 0, 0, %d> grid %d x %d threads (five waves side by side, %d of a wave's %d fragments in LDS), up to %d steps per launch",
                              solver, ad ? "true" : "false", sh.cw, sh.ncg, sh.nch, sh.ru, sh.kh, 8 * sh.nch - persist_wide_kr(solver, ad), sh.grid, sh.threads,
                              8 * sh.nch - persist_wide_kr(solver, ad), 8 * sh.nch, TABLE_STEPS);
            else
                std::snprintf(buf, buf_len, "ccvm::persist_kernel<%d, %s, %d, %d, %d, %d, %d, 0, 0, %d, %d> grid %d x %d threads (five waves side by side, K split %d | %d, the long parts' last %d fragments in LDS), up to %d steps per launch",
                              solver, ad ? "true" : "false", sh.cw, sh.ncg, sh.nch, sh.ru, sh.kh, 8 * sh.nch - persist_wide_kr_xs(solver, ad),
                              persist_wide_xs(solver, ad, sh.nch), sh.grid, sh.threads, persist_wide_xs(solver, ad, sh.nch),
                              16 * sh.nch - persist_wide_xs(solver, ad, sh.nch),
                              16 * sh.nch - persist_wide_xs(solver, ad, sh.nch) - persist_wide_kr_xs(solver, ad), TABLE_STEPS);
        else if (sh.xs > 0 && tun.persist_xs != 1)
            std::snprintf(buf, buf_len, "ccvm::persist_kernel<%d, %s, %d, %d, %d, %d, %d, 0, 0, 0, %d> grid %d x %d threads (K split %d | %d), up to %d steps per launch",
                          solver, ad ? "true" : "false", sh.cw, sh.ncg, sh.nch, sh.ru, sh.kh, sh.xs, sh.grid, sh.threads, sh.xs, 16 * sh.nch - sh.xs, TABLE_STEPS);
        else if (sh.rsw == 2)
            std::snprintf(buf, buf_len, "ccvm::persist_kernel<%d, %s, %d, %d, %d, %d, %d, 0, 2> grid %d x %d threads (two row sets per workgroup), up to %d steps per launch",
                          solver, ad ? "true" : "false", sh.cw, sh.ncg, sh.nch, sh.ru, sh.kh, sh.grid, sh.threads, TABLE_STEPS);
        else if (sh.pw)
            std::snprintf(buf, buf_len, "ccvm::persist_kernel<%d, %s, %d, %d, %d, %d, %d, 1> grid %d x %d threads (noise producer waves), up to %d steps per launch",
                          solver, ad ? "true" : "false", sh.cw, sh.ncg, sh.nch, sh.ru, sh.kh, sh.grid, sh.threads, TABLE_STEPS);
        else
            std::snprintf(buf, buf_len, "ccvm::persist_kernel<%d, %s, %d, %d, %d, %d, %d> grid %d x %d threads, up to %d steps per launch",
                          solver, ad ? "true" : "false", sh.cw, sh.ncg, sh.nch, sh.ru, sh.kh, sh.grid, sh.threads, TABLE_STEPS);
    } else {
        StepArgs a;
        base_args(a, nullptr, nullptr, B, N, ccvm_ld(N), tun, 4, solver);
        if (const PtilePlan plan = plan_ptile(a, tun, per_variable_s && solver != 0, solver); plan.slices == 1) {
            // (the grid run_ptile launches: 32 x 128 tiles whatever shape the per-step plan `a` has -- ADVICE r5: with KS = 2 / 4
            // in `a` this line printed the per-step grid, up to four times the workgroups that run)
            StepArgs g = a;
            g.ks = 1;
            set_grid(g, tun, true);
            std::snprintf(buf, buf_len, "ccvm::ptile_kernel<%d, %s%s> grid %d x %d threads (%d row blocks x %d column blocks resident, XCD rectangle %d x %d), up to %d steps per launch",
                          solver, ad ? "true" : "false", (per_variable_s && solver != 0) ? ", false, true" : "", g.nrb * g.ncb,
                          WG_THREADS, g.nrb, g.ncb, g.xr, g.xc, TABLE_STEPS);
            return CCVM_OK;
        } else if (plan.slices > 1) {
            StepArgs g = a;
            g.B = plan.rbs * BM;
            g.ks = 1;
            set_grid(g, tun, true);
            std::snprintf(buf, buf_len, "ccvm::ptile_kernel<%d, %s%s> %d slices of the batch one after the other, grid %d x %d threads each (up to %d row blocks x %d column blocks resident), up to %d steps per launch",
                          solver, ad ? "true" : "false", (per_variable_s && solver != 0) ? ", false, true" : "", plan.slices,
                          g.nrb * g.ncb, WG_THREADS, g.nrb, g.ncb, TABLE_STEPS);
            return CCVM_OK;
        }
        std::snprintf(buf, buf_len, "ccvm::step_kernel<%d, %s, 0, %d, %s, 0> grid %d x %d threads, XCD rectangle %d x %d, 1 step per launch",
                      solver, ad ? "true" : "false", a.ks, (per_variable_s && solver != 0) ? "true" : "false",
                      a.nrb * a.ncb, WG_THREADS, a.xr, a.xc);
    }
    return CCVM_OK;
}

}  // namespace
extern "C" {
int ccvm_pack(const float* src, int rows, int cols, int src_ld, float* dst, int dst_rows, int dst_ld,
              void* stream) {
    if (!src || !dst || rows < 0 || cols < 0 || src_ld < cols || dst_rows < rows || dst_ld < cols)
        return fail(CCVM_E_INVALID, "ccvm_pack: bad arguments");
    const size_t total = (size_t)dst_rows * dst_ld;
    if (total == 0) return CCVM_OK;
    hipLaunchKernelGGL(pack_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, rows, cols,
                       src_ld, dst, dst_rows, dst_ld);
    CCVM_CHECK_LAUNCH("ccvm_pack");
    return CCVM_OK;
}

int ccvm_unpack(const float* src, int src_ld, float* dst, int rows, int cols, int dst_ld, void* stream) {
    if (!src || !dst || rows < 0 || cols < 0 || src_ld < cols || dst_ld < cols)
        return fail(CCVM_E_INVALID, "ccvm_unpack: bad arguments");
    const size_t total = (size_t)rows * cols;
    if (total == 0) return CCVM_OK;
    hipLaunchKernelGGL(unpack_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, src_ld, dst,
                       rows, cols, dst_ld);
    CCVM_CHECK_LAUNCH("ccvm_unpack");
    return CCVM_OK;
}

int ccvm_dl_run(const float* Q, const float* V, float* c, float* s, int B, int N, int ld, int step0,
                int nsteps, int T, const ccvm_dl_params* p, const ccvm_noise* nz, void* ws, size_t ws_bytes,
                void* stream) {
    const char* fn = "ccvm_dl_run";
    Tuning tun = read_tuning();
    if (nz && (nz->flags & CCVM_RUN_NO_EXCHANGE)) tun.cluster = tun.slab = tun.ptile = 0;
    if (!Q || !V || !c || !s || !p) return fail(CCVM_E_INVALID, "%s: NULL argument", fn);
    int rc;
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if ((rc = check_steps(fn, step0, nsteps, T))) return rc;
    if ((rc = check_noise(fn, nz, true, B))) return rc;
    if (!aligned16(Q) || !aligned16(c) || !aligned16(s) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, c, s and workspace must be 16-byte aligned", fn);
    if (ws_bytes < workspace_plain(0, B, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    if (!(p->upper > p->lower) || !(p->dt > 0)) return fail(CCVM_E_INVALID, "%s: need upper > lower, dt > 0", fn);
    hipStream_t st = (hipStream_t)stream;

    if (const int cut = nsteps > 0 ? split_rows(MODE_DL, B, N, tun) : 0;
        cut && ws_bytes >= ccvm_workspace_bytes(0, B, N)) {
        // two calls on this stream: the rows of whole resident grids, then the rest (split_rows)
        char* part_ws = static_cast<char*>(ws) + part_offset(workspace_plain(0, B, N));
        unsigned* status = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + ccvm_status_offset(0, B, N));
        for (int r0 = 0; r0 < B; r0 = r0 ? B : cut) {
            const int rows = r0 ? B - cut : cut;
            ccvm_noise n = *nz;  // (replay: the part's columns of the batch's blocks)
            n.row_offset += r0;
            n.w_ld = (int64_t)noise_pitch(nz, B);
            if (n.w0) n.w0 += r0;
            if (n.w1) n.w1 += r0;
            const size_t off = (size_t)r0 * ld, bytes = workspace_plain(0, rows, N);
            if ((rc = ccvm_dl_run(Q, V, c + off, s + off, rows, N, ld, step0, nsteps, T, p, &n, part_ws, bytes, stream))) return rc;
            hipLaunchKernelGGL(status_merge_kernel, dim3(1), dim3(1), 0, st, status,
                               reinterpret_cast<unsigned*>(part_ws + ccvm_status_offset(0, rows, N)));
            CCVM_CHECK_LAUNCH(fn);
            part_ws += part_offset(bytes);
        }
        return CCVM_OK;
    }

    const size_t state = (size_t)ccvm_rows(B) * ld;
    float* bufc[2] = {c, static_cast<float*>(ws)};
    float* bufs[2] = {s, static_cast<float*>(ws) + state};
    if (nsteps > 0 && !(nz->flags & CCVM_RUN_WS_PADDED)) {
        // keep the padding of the scratch buffers zero (rows >= B, cols >= N are never written)
        if (hipMemsetAsync(ws, 0, 2 * state * sizeof(float), st) != hipSuccess)
            return fail(CCVM_E_HIP, "%s: memset failed", fn);
    }

    const double ul = p->upper - p->lower, up = p->upper + p->lower;
    const double Sd = p->pump > 1.0 ? std::sqrt(p->pump - 1.0) : 1.0;  // dl_solver.py:140-141
    StepArgs a;
    base_args(a, Q, V, B, N, ld, tun, 4, MODE_DL);
    a.in_scale = (float)(ul / Sd);
    a.in_shift = (float)up;
    if (nsteps > 0 && (rc = compute_qsum(Q, N, ld, static_cast<float*>(ws) + 2 * state, st, &a.qsum, p->qsum))) return rc;
    if (nsteps > 0 && want_persist(N, tun, MODE_DL, B)) {
        // whole chunks of the trajectory in one launch each (ccvm_persist.h)
        float* table = reinterpret_cast<float*>(static_cast<char*>(ws) + 2 * state * sizeof(float) + qsum_area_bytes(N));
        PersistArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.Q = Q; pa.V = V; pa.qsum = a.qsum; pa.x0 = c; pa.x1 = s; pa.table = table;
        pa.seed = nz->seed; pa.row_offset = nz->row_offset; pa.replay = nz->mode == CCVM_NOISE_REPLAY;
        pa.B = B; pa.N = N; pa.ld = ld; pa.wld = (int)noise_pitch(nz, B); pa.in_scale = a.in_scale; pa.in_shift = a.in_shift;
        pa.ru_override = tun.persist_ru;
        pa.kh_override = tun.persist_kh;
        pa.pw_override = tun.persist_pw;
        pa.rsw_override = tun.persist_rsw;
        pa.cw_override = tun.persist_cw;
        pa.xs_override = tun.persist_xs;
        pa.simds = 4 * chip_of(tun).cus;
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            DlSched sc{p->pump, p->dt, p->noise_ratio, p->feedback_scale, p->g, ul, Sd, p->pump_rate_flag, T,
                       step0 + done, k};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(dl_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            pa.step0 = step0 + done;
            pa.table = rows;
            pa.nsteps = k;
            if (pa.replay) {
                pa.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
                pa.w1 = nz->w1 + (size_t)done * N * noise_pitch(nz, B);
            }
            if ((rc = launch_persist<MODE_DL, false>(pa, st, fn))) return rc;
        }
        return CCVM_OK;
    }
    if (const SlabPlan sp = (nsteps > 0 && !want_persist(N, tun, MODE_DL, B)) ? want_slab(B, N, tun, MODE_DL) : SlabPlan{}; sp.ok) {
        // small batch: whole chunks in one launch each, Q resident in the members' registers (ccvm_slab.h)
        char* after = static_cast<char*>(ws) + 2 * state * sizeof(float) + qsum_area_bytes(N);
        float* table = reinterpret_cast<float*>(after);
        SlabArgs sa;
        unsigned xid;
        if (slab_base(sa, xid, sp, Q, V, a.qsum, B, N, ld, nz, table, after + table_bytes(), step0, st, tun, 2))
            return fail(CCVM_E_HIP, "%s: memset failed", fn);
        sa.x0 = c; sa.x1 = s;
        sa.in_scale = a.in_scale; sa.in_shift = a.in_shift;
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            DlSched sc{p->pump, p->dt, p->noise_ratio, p->feedback_scale, p->g, ul, Sd, p->pump_rate_flag, T,
                       step0 + done, k};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(dl_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            sa.step0 = step0 + done;
            sa.table = rows;
            sa.nsteps = k;
            if (sa.replay) {
                sa.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
                sa.w1 = nz->w1 + (size_t)done * N * noise_pitch(nz, B);
            }
            slab_launch_dl(sa, sp, st);
            CCVM_CHECK_LAUNCH(fn);
        }
        if (exchange_commit(sa.status, xid, step0 + nsteps, st)) return fail(CCVM_E_HIP, "%s: launch failed", fn);
        return CCVM_OK;
    }
    if (nsteps > 0 && want_cluster(B, N, tun, MODE_DL, false)) {
        // whole chunks of the trajectory in one launch each, Q panels resident in LDS (ccvm_cluster.h)
        char* after = static_cast<char*>(ws) + 2 * state * sizeof(float) + qsum_area_bytes(N);
        float* table = reinterpret_cast<float*>(after);
        ClusterArgs ca;
        unsigned xid;
        if (cluster_base(ca, xid, Q, V, a.qsum, B, N, ld, nz, table, after + table_bytes(), step0, st, tun, 2))
            return fail(CCVM_E_HIP, "%s: memset failed", fn);
        ca.x0 = c; ca.x1 = s;
        ca.in_scale = a.in_scale; ca.in_shift = a.in_shift;
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            DlSched sc{p->pump, p->dt, p->noise_ratio, p->feedback_scale, p->g, ul, Sd, p->pump_rate_flag, T,
                       step0 + done, k};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(dl_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            ca.step0 = step0 + done;
            ca.table = rows;
            ca.nsteps = k;
            if (ca.replay) {
                ca.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
                ca.w1 = nz->w1 + (size_t)done * N * noise_pitch(nz, B);
            }
            cluster_launch_dl(ca, st);
            CCVM_CHECK_LAUNCH(fn);
        }
        if (exchange_commit(ca.status, xid, step0 + nsteps, st)) return fail(CCVM_E_HIP, "%s: launch failed", fn);
        return CCVM_OK;
    }
    if (nsteps > 0 && want_ptile(a, tun, MODE_DL, false)) {
        // whole chunks in one launch each, the tile grid resident, the state handed over between the workgroups of a
        // row block inside the launch (ccvm_ptile.h)
        char* after = static_cast<char*>(ws) + 2 * state * sizeof(float) + qsum_area_bytes(N);
        float* table = reinterpret_cast<float*>(after);
        unsigned* status = reinterpret_cast<unsigned*>(after + table_bytes() + exchange_bytes(B, N, 2));
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            DlSched sc{p->pump, p->dt, p->noise_ratio, p->feedback_scale, p->g, ul, Sd, p->pump_rate_flag, T,
                       step0 + done, k, reinterpret_cast<unsigned*>(after + table_bytes()), a.nrb * PT_FLAG_WORDS};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(dl_schedule_kernel, dim3(ptile_sched_grid(k, a.nrb)), dim3(256), 0, st, sc, table);
                rows = table;
            }
            if ((rc = run_ptile<MODE_DL>(a, bufc, bufs, nz, rows, after + table_bytes(), status, step0, done, k, tun, st, fn,
                                         done & 1)))  // launches of a call alternate the buffers like its steps
                return rc;
        }
        if (nsteps & 1) {
            if (hipMemcpyAsync(c, bufc[1], state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
                hipMemcpyAsync(s, bufs[1], state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return fail(CCVM_E_HIP, "%s: copy-back failed", fn);
        }
        return CCVM_OK;
    }
    int cur = 0;
    for (int i = step0; i < step0 + nsteps; ++i) {
        const double frac = (double)(i + 1) / (double)T;
        const double rate = p->pump_rate_flag ? frac : 1.0;                           // :524-525
        const double ratio = (p->noise_ratio - 1.0) * std::exp(-frac * 3.0) + 1.0;    // :527
        const double fsd = p->feedback_scale * (0.5 + rate);                          // :169
        DlScalars& k = a.s.dl;
        k.a_q = (float)(-p->dt * fsd * 0.25 * ul / Sd);
        k.a_v = (float)(-p->dt * fsd * ul / (2.0 * Sd));
        k.pm_c = (float)(-1.0 + p->pump * rate);
        k.pm_s = (float)(-1.0 - p->pump * rate);
        k.dt = (float)p->dt;
        k.g2 = (float)(2.0 * p->g);
        k.w_c = (float)(std::sqrt(p->dt) * ratio);
        k.w_s = (float)(std::sqrt(p->dt) / ratio);
        a.a0 = bufc[cur];
        a.a1 = bufs[cur];
        a.o0 = bufc[cur ^ 1];
        a.o1 = bufs[cur ^ 1];
        set_noise(a, nz, i, step0, B, N, true, false);
        if ((rc = launch_step<MODE_DL, false>(a, st, fn))) return rc;
        cur ^= 1;
    }
    if (cur == 1) {
        if (hipMemcpyAsync(c, bufc[1], state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(s, bufs[1], state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(CCVM_E_HIP, "%s: copy-back failed", fn);
    }
    return CCVM_OK;
}

int ccvm_mf_run(const float* Q, const float* V, float* mu, float* sigma, float* mu_tilde_out, int B, int N,
                int ld, int step0, int nsteps, int T, const ccvm_mf_params* p, const ccvm_adam* adam,
                const ccvm_noise* nz, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_mf_run";
    Tuning tun = read_tuning();
    tun.adam = adam && adam->enabled;
    if (nz && (nz->flags & CCVM_RUN_NO_EXCHANGE)) tun.cluster = tun.slab = tun.ptile = 0;
    if (!Q || !V || !mu || !sigma || !p) return fail(CCVM_E_INVALID, "%s: NULL argument", fn);
    int rc;
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if ((rc = check_steps(fn, step0, nsteps, T))) return rc;
    if ((rc = check_noise(fn, nz, false, B))) return rc;
    if ((rc = check_adam(fn, adam))) return rc;
    if (!aligned16(Q) || !aligned16(mu) || !aligned16(sigma) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, mu, sigma and workspace must be 16-byte aligned", fn);
    const float* s_cols = p->s_cols;
    if (ws_bytes < workspace_plain(1, B, N) + (s_cols ? (size_t)ld * ld * sizeof(float) : 0))
        return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    const float* s_full = p->s_full;
    if (!(p->upper > p->lower) || !(p->dt > 0) || !(s_cols || s_full || p->S > 0) || !(p->j > 0))
        return fail(CCVM_E_INVALID, "%s: need upper > lower, dt > 0, S > 0, j > 0", fn);
    if (s_cols && s_full) return fail(CCVM_E_INVALID, "%s: s_cols and s_full are exclusive", fn);
    if (s_full && nz->mode == CCVM_NOISE_REPLAY && nz->w_ld != 0 && nz->w_ld != B)
        return fail(CCVM_E_INVALID, "%s: s_full needs replay blocks pitched by the batch (w_ld = 0)", fn);
    if (s_cols && !aligned16(s_cols)) return fail(CCVM_E_LAYOUT, "%s: s_cols must be 16-byte aligned", fn);
    // per-variable saturation: every 1 / S factor of the scalars is left out (S_eff = 1) and applied per
    // column -- 1 / S_k of the input map through the row-scaled copy Qs, 1 / S_j in the epilogue
    const double S_eff = (s_cols || s_full) ? 1.0 : p->S;
    if (nsteps == 0) return CCVM_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool use_adam = adam && adam->enabled;

    if (const int cut = (!s_cols && !s_full) ? split_rows(MODE_MF, B, N, tun) : 0;
        cut && ws_bytes >= ccvm_workspace_bytes(1, B, N)) {
        // two calls on this stream: the rows of whole resident grids, then the rest (split_rows)
        char* part_ws = static_cast<char*>(ws) + part_offset(workspace_plain(1, B, N));
        unsigned* status = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + ccvm_status_offset(1, B, N));
        for (int r0 = 0; r0 < B; r0 = r0 ? B : cut) {
            const int rows = r0 ? B - cut : cut;
            const size_t off = (size_t)r0 * ld, bytes = workspace_plain(1, rows, N);
            ccvm_noise n = *nz;  // (replay: the part's columns of the batch's blocks)
            n.row_offset += r0;
            n.w_ld = (int64_t)noise_pitch(nz, B);
            if (n.w0) n.w0 += r0;
            if (n.w1) n.w1 += r0;
            ccvm_adam ad;
            if (adam) {
                ad = *adam;
                if (ad.m) ad.m += off;
                if (ad.v) ad.v += off;
            }
            if ((rc = ccvm_mf_run(Q, V, mu + off, sigma + off, mu_tilde_out ? mu_tilde_out + off : nullptr, rows, N, ld, step0,
                                  nsteps, T, p, adam ? &ad : nullptr, &n, part_ws, bytes, stream)))
                return rc;
            hipLaunchKernelGGL(status_merge_kernel, dim3(1), dim3(1), 0, st, status,
                               reinterpret_cast<unsigned*>(part_ws + ccvm_status_offset(1, rows, N)));
            CCVM_CHECK_LAUNCH(fn);
            part_ws += part_offset(bytes);
        }
        return CCVM_OK;
    }

    const size_t state = (size_t)ccvm_rows(B) * ld;
    const double ul = p->upper - p->lower, up = p->upper + p->lower;
    const double sdt = std::sqrt(p->dt);
    auto j_at = [&](int i) { return p->j * std::exp(-(double)(i + 1) / (double)T * 3.0); };  // :550
    const bool replay = nz->mode == CCVM_NOISE_REPLAY;

    if (s_full) {
        // one saturation per trajectory AND variable: composed path (ccvm_kernels.h, fulls_* kernels):
        // workspace = [xs = mt / S][y = AFFINE result][mt]
        float* xs = static_cast<float*>(ws);
        float* y = xs + state;
        float* mtb = xs + 2 * state;
        if (hipMemsetAsync(ws, 0, 3 * state * sizeof(float), st) != hipSuccess)
            return fail(CCVM_E_HIP, "%s: memset failed", fn);
        StepArgs a;
        base_args(a, Q, V, B, N, ld, tun);
        a.in_scale = (float)ul;
        a.in_shift = (float)up;
        if ((rc = compute_qsum(Q, N, ld, static_cast<float*>(ws) + 3 * state, st, &a.qsum, p->qsum))) return rc;
        a.a0 = xs;
        a.o0 = y;
        a.s.pp.step = (float)(-p->feedback_scale * 0.25 * ul);  // f_q for S = 1; 1 / S_bj in the update kernel
        a.s.pp.eps = (float)(-p->feedback_scale * ul / 2.0);    // f_v
        FullSArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.x0 = mu; fa.x1 = sigma; fa.mt = mtb; fa.xs = xs; fa.y = y; fa.s_full = s_full;
        fa.seed = nz->seed; fa.row_offset = nz->row_offset; fa.B = B; fa.N = N; fa.ld = ld;
        fa.adam = use_adam;
        if (use_adam) { fa.am = adam->m; fa.av = adam->v; }
        const size_t blk = (size_t)N * B;
        fa.step = step0;
        fa.w0 = replay ? nz->w0 : nullptr;
        hipLaunchKernelGGL(fulls_mf_prepare_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, st, fa,
                           (float)(std::sqrt(1.0 / (4.0 * j_at(step0))) / sdt));
        CCVM_CHECK_LAUNCH(fn);
        for (int i = step0; i < step0 + nsteps; ++i) {
            const bool has_next = (i + 1 < step0 + nsteps);
            if ((rc = launch_step<MODE_AFFINE, false>(a, st, fn))) return rc;
            const double j_i = j_at(i);
            const double rate = p->pump_rate_flag ? (double)(i + 1) / (double)T : 1.0;
            const double p_i = p->pump * rate + 1.0 + j_i;
            MfScalars k;
            k.a0 = (float)(-(1.0 + j_i) + p_i);
            k.g2 = (float)(p->g * p->g);
            k.f_q = a.s.pp.step;
            k.f_v = a.s.pp.eps;
            k.j_i = (float)j_i;
            k.one_j = (float)(1.0 + j_i);
            k.sqrt_j = (float)std::sqrt(j_i);
            k.inv_sdt = (float)(1.0 / sdt);
            k.dt = (float)p->dt;
            k.k_next = has_next ? (float)(std::sqrt(1.0 / (4.0 * j_at(i + 1))) / sdt) : 0.0f;
            k.S = 1.0f;
            k.has_next = has_next;
            if (use_adam) fill_adam(fa.ad, adam, i);
            fa.step = i;
            fa.w0 = replay ? nz->w0 + (size_t)(i - step0) * blk : nullptr;
            fa.w0n = (replay && has_next) ? nz->w0 + (size_t)(i + 1 - step0) * blk : nullptr;
            hipLaunchKernelGGL(fulls_mf_update_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, st, fa, k);
            CCVM_CHECK_LAUNCH(fn);
        }
        if (mu_tilde_out &&
            hipMemcpyAsync(mu_tilde_out, mtb, state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(CCVM_E_HIP, "%s: copy of mu_tilde failed", fn);
        return CCVM_OK;
    }

    if (want_persist(N, tun, MODE_MF, B)) {
        // whole chunks of the trajectory in one launch each (ccvm_persist.h)
        const float* qsum;
        if ((rc = compute_qsum(Q, N, ld, static_cast<float*>(ws) + 3 * state, st, &qsum, p->qsum))) return rc;
        float* table = reinterpret_cast<float*>(static_cast<char*>(ws) + 3 * state * sizeof(float) + qsum_area_bytes(N));
        PersistArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.Q = s_cols ? scaled_rows(Q, s_cols, N, ld, ws, workspace_plain(1, B, N), st) : Q;
        pa.V = V; pa.qsum = qsum; pa.x0 = mu; pa.x1 = sigma; pa.xt = mu_tilde_out; pa.table = table;
        pa.seed = nz->seed; pa.row_offset = nz->row_offset; pa.replay = replay;
        pa.B = B; pa.N = N; pa.ld = ld; pa.wld = (int)noise_pitch(nz, B); pa.in_scale = (float)(ul / S_eff); pa.in_shift = (float)up;
        pa.S = (float)S_eff;
        pa.ru_override = tun.persist_ru;
        pa.kh_override = tun.persist_kh;
        pa.pw_override = tun.persist_pw;
        pa.rsw_override = tun.persist_rsw;
        pa.cw_override = tun.persist_cw;
        pa.xs_override = tun.persist_xs;
        pa.simds = 4 * chip_of(tun).cus;
        pa.s_cols = s_cols;
        AdamSched asc;
        persist_adam(pa, asc, adam, use_adam);
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            MfSched sc{p->pump, p->dt, p->j, p->feedback_scale, p->g, S_eff, ul, p->pump_rate_flag, T, step0 + done, k, asc};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(mf_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            pa.step0 = step0 + done;
            pa.table = rows;
            pa.nsteps = k;
            pa.k_first = (float)(std::sqrt(1.0 / (4.0 * j_at(step0 + done))) / sdt);
            if (replay) pa.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
            rc = use_adam ? launch_persist<MODE_MF, true>(pa, st, fn) : launch_persist<MODE_MF, false>(pa, st, fn);
            if (rc) return rc;
        }
        return CCVM_OK;
    }

    if (const SlabPlan sp = want_slab(B, N, tun, MODE_MF); sp.ok) {
        // small batch: whole chunks in one launch each, Q resident in the members' registers (ccvm_slab.h)
        const float* qsum;
        if ((rc = compute_qsum(Q, N, ld, static_cast<float*>(ws) + 3 * state, st, &qsum, p->qsum))) return rc;
        char* after = static_cast<char*>(ws) + 3 * state * sizeof(float) + qsum_area_bytes(N);
        float* table = reinterpret_cast<float*>(after);
        SlabArgs sa;
        unsigned xid;
        const float* q_used = s_cols ? scaled_rows(Q, s_cols, N, ld, ws, workspace_plain(1, B, N), st) : Q;
        if (slab_base(sa, xid, sp, q_used, V, qsum, B, N, ld, nz, table, after + table_bytes(), step0, st, tun, 1))
            return fail(CCVM_E_HIP, "%s: memset failed", fn);
        sa.x0 = mu; sa.x1 = sigma; sa.xt = mu_tilde_out;
        sa.in_scale = (float)(ul / S_eff); sa.in_shift = (float)up; sa.S = (float)S_eff; sa.s_cols = s_cols;
        PersistArgs pa_ad;  // the Adam constants in the persistent kernels' form
        std::memset(&pa_ad, 0, sizeof(pa_ad));
        AdamSched asc;
        persist_adam(pa_ad, asc, adam, use_adam);
        sa.adam = use_adam; sa.ad = pa_ad.ad; sa.am = pa_ad.am; sa.av = pa_ad.av;
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            MfSched sc{p->pump, p->dt, p->j, p->feedback_scale, p->g, S_eff, ul, p->pump_rate_flag, T, step0 + done, k, asc};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(mf_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            sa.step0 = step0 + done;
            sa.table = rows;
            sa.nsteps = k;
            sa.k_first = (float)(std::sqrt(1.0 / (4.0 * j_at(step0 + done))) / sdt);
            if (replay) sa.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
            slab_launch_mf(sa, sp, st);
            CCVM_CHECK_LAUNCH(fn);
        }
        if (exchange_commit(sa.status, xid, step0 + nsteps, st)) return fail(CCVM_E_HIP, "%s: launch failed", fn);
        return CCVM_OK;
    }

    if (want_cluster(B, N, tun, MODE_MF, use_adam)) {
        // whole chunks of the trajectory in one launch each, Q panels resident in LDS (ccvm_cluster.h)
        const float* qsum;
        if ((rc = compute_qsum(Q, N, ld, static_cast<float*>(ws) + 3 * state, st, &qsum, p->qsum))) return rc;
        char* after = static_cast<char*>(ws) + 3 * state * sizeof(float) + qsum_area_bytes(N);
        float* table = reinterpret_cast<float*>(after);
        ClusterArgs ca;
        unsigned xid;
        const float* q_used = s_cols ? scaled_rows(Q, s_cols, N, ld, ws, workspace_plain(1, B, N), st) : Q;
        if (cluster_base(ca, xid, q_used, V, qsum, B, N, ld, nz, table, after + table_bytes(), step0, st, tun))
            return fail(CCVM_E_HIP, "%s: memset failed", fn);
        ca.x0 = mu; ca.x1 = sigma; ca.xt = mu_tilde_out;
        ca.in_scale = (float)(ul / S_eff); ca.in_shift = (float)up; ca.S = (float)S_eff; ca.s_cols = s_cols;
        PersistArgs pa_ad;  // the Adam constants in the persistent kernels' form
        std::memset(&pa_ad, 0, sizeof(pa_ad));
        AdamSched asc;
        persist_adam(pa_ad, asc, adam, use_adam);
        ca.ad = pa_ad.ad; ca.am = pa_ad.am; ca.av = pa_ad.av;
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            MfSched sc{p->pump, p->dt, p->j, p->feedback_scale, p->g, S_eff, ul, p->pump_rate_flag, T, step0 + done, k, asc};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(mf_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            ca.step0 = step0 + done;
            ca.table = rows;
            ca.nsteps = k;
            ca.k_first = (float)(std::sqrt(1.0 / (4.0 * j_at(step0 + done))) / sdt);
            if (replay) ca.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
            cluster_launch_mf(ca, use_adam, st);
            CCVM_CHECK_LAUNCH(fn);
        }
        if (exchange_commit(ca.status, xid, step0 + nsteps, st)) return fail(CCVM_E_HIP, "%s: launch failed", fn);
        return CCVM_OK;
    }

    float* mt[2] = {static_cast<float*>(ws), static_cast<float*>(ws) + state};
    float* carry = static_cast<float*>(ws) + 2 * state;  // this step's normals (fused mode)
    if (!(nz->flags & CCVM_RUN_WS_PADDED) && hipMemsetAsync(ws, 0, 3 * state * sizeof(float), st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: memset failed", fn);

    StepArgs a;
    base_args(a, Q, V, B, N, ld, tun, 4, MODE_MF);
    a.in_scale = (float)(ul / S_eff);
    a.in_shift = (float)up;
    if ((rc = compute_qsum(Q, N, ld, static_cast<float*>(ws) + 3 * state, st, &a.qsum, p->qsum))) return rc;
    if (s_cols) {
        a.Q = scaled_rows(Q, s_cols, N, ld, ws, workspace_plain(1, B, N), st);
        a.s_cols = s_cols;
    }
    // measured amplitude of the first step of a chunk (mf_solver.py:551-554), and that step's normals
    auto prepare = [&](int first_step, int done) {
        const float k0 = (float)(std::sqrt(1.0 / (4.0 * j_at(first_step))) / sdt);
        hipLaunchKernelGGL(mf_prepare_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, st, mu, mt[0], carry, B, N,
                           ld, k0, (float)S_eff, s_cols, nz->seed, nz->row_offset, first_step,
                           replay ? nz->w0 + (size_t)done * N * noise_pitch(nz, B) : nullptr, (int)noise_pitch(nz, B));
    };
    if (want_ptile(a, tun, MODE_MF, s_cols != nullptr)) {
        // whole chunks in one launch each, the tile grid resident (ccvm_ptile.h): the exchanged plane is the measured
        // amplitude; mu, sigma and the Adam moments stay in the workgroups' registers for the launch
        char* after = static_cast<char*>(ws) + 3 * state * sizeof(float) + qsum_area_bytes(N);
        float* table = reinterpret_cast<float*>(after);
        unsigned* status = reinterpret_cast<unsigned*>(after + table_bytes() + exchange_bytes(B, N, 1));
        PersistArgs pa_ad;
        std::memset(&pa_ad, 0, sizeof(pa_ad));
        AdamSched asc;
        persist_adam(pa_ad, asc, adam, use_adam);
        float* const none[2] = {nullptr, nullptr};
        int last_k = 0;
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            prepare(step0 + done, done);
            MfSched sc{p->pump, p->dt, p->j, p->feedback_scale, p->g, S_eff, ul, p->pump_rate_flag, T, step0 + done, k, asc,
                       reinterpret_cast<unsigned*>(after + table_bytes()), a.nrb * PT_FLAG_WORDS};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(mf_schedule_kernel, dim3(ptile_sched_grid(k, a.nrb)), dim3(256), 0, st, sc, table);
                rows = table;
            }
            if ((rc = run_ptile<MODE_MF>(a, mt, none, nz, rows, after + table_bytes(), status, step0, done, k, tun, st, fn, 0,
                                         adam, mu, sigma, carry)))
                return rc;
            last_k = k;
        }
        // the measured amplitude fed to the LAST step (mf_solver.py:591-593): the buffer that step read
        if (mu_tilde_out && hipMemcpyAsync(mu_tilde_out, mt[(last_k - 1) & 1], state * sizeof(float),
                                           hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(CCVM_E_HIP, "%s: copy of mu_tilde failed", fn);
        return CCVM_OK;
    }
    prepare(step0, 0);
    CCVM_CHECK_LAUNCH(fn);

    a.carry = carry;
    a.st0 = mu;
    a.st1 = sigma;
    if (use_adam) {
        a.am = adam->m;
        a.av = adam->v;
    }
    int cur = 0;
    for (int i = step0; i < step0 + nsteps; ++i) {
        const bool has_next = (i + 1 < step0 + nsteps);
        const double j_i = j_at(i);
        const double rate = p->pump_rate_flag ? (double)(i + 1) / (double)T : 1.0;  // :556-557
        const double p_i = p->pump * rate + 1.0 + j_i;                              // :559
        MfScalars& k = a.s.mf;
        k.a0 = (float)(-(1.0 + j_i) + p_i);
        k.g2 = (float)(p->g * p->g);
        k.f_q = (float)(-p->feedback_scale * 0.25 * ul / S_eff);
        k.f_v = (float)(-p->feedback_scale * ul / (2.0 * S_eff));
        k.j_i = (float)j_i;
        k.one_j = (float)(1.0 + j_i);
        k.sqrt_j = (float)std::sqrt(j_i);
        k.inv_sdt = (float)(1.0 / sdt);
        k.dt = (float)p->dt;
        k.k_next = has_next ? (float)(std::sqrt(1.0 / (4.0 * j_at(i + 1))) / sdt) : 0.0f;
        k.S = (float)S_eff;
        k.has_next = has_next;
        if (use_adam) fill_adam(a.ad, adam, i);
        a.a0 = mt[cur];
        a.o0 = mt[cur ^ 1];
        set_noise(a, nz, i, step0, B, N, false, has_next);
        rc = use_adam ? launch_step<MODE_MF, true>(a, st, fn) : launch_step<MODE_MF, false>(a, st, fn);
        if (rc) return rc;
        if (has_next) cur ^= 1;
    }
    if (mu_tilde_out) {
        if (hipMemcpyAsync(mu_tilde_out, mt[cur], state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(CCVM_E_HIP, "%s: copy of mu_tilde failed", fn);
    }
    return CCVM_OK;
}

int ccvm_langevin_run(const float* Q, const float* V, float* c, int B, int N, int ld, int step0, int nsteps,
                      int T, const ccvm_langevin_params* p, const ccvm_adam* adam, const ccvm_noise* nz,
                      void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_langevin_run";
    Tuning tun = read_tuning();
    tun.adam = adam && adam->enabled;
    if (nz && (nz->flags & CCVM_RUN_NO_EXCHANGE)) tun.cluster = tun.slab = tun.ptile = 0;
    if (!Q || !V || !c || !p) return fail(CCVM_E_INVALID, "%s: NULL argument", fn);
    int rc;
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if ((rc = check_steps(fn, step0, nsteps, T))) return rc;
    if ((rc = check_noise(fn, nz, false, B))) return rc;
    if ((rc = check_adam(fn, adam))) return rc;
    if (!aligned16(Q) || !aligned16(c) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, c and workspace must be 16-byte aligned", fn);
    const float* s_cols = p->s_cols;
    if (ws_bytes < workspace_plain(2, B, N) + (s_cols ? (size_t)ld * ld * sizeof(float) : 0))
        return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    const float* s_full = p->s_full;
    if (!(p->upper > p->lower) || !(p->dt > 0) || !(s_cols || s_full || p->S > 0))
        return fail(CCVM_E_INVALID, "%s: need upper > lower, dt > 0, S > 0", fn);
    if (s_cols && s_full) return fail(CCVM_E_INVALID, "%s: s_cols and s_full are exclusive", fn);
    if (s_full && nz->mode == CCVM_NOISE_REPLAY && nz->w_ld != 0 && nz->w_ld != B)
        return fail(CCVM_E_INVALID, "%s: s_full needs replay blocks pitched by the batch (w_ld = 0)", fn);
    if (s_cols && !aligned16(s_cols)) return fail(CCVM_E_LAYOUT, "%s: s_cols must be 16-byte aligned", fn);
    const double S_eff = (s_cols || s_full) ? 1.0 : p->S;  // see ccvm_mf_run
    if (nsteps == 0) return CCVM_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool use_adam = adam && adam->enabled;

    if (const int cut = (!s_cols && !s_full) ? split_rows(MODE_LANGEVIN, B, N, tun) : 0;
        cut && ws_bytes >= ccvm_workspace_bytes(2, B, N)) {
        // two calls on this stream: the rows of whole resident grids, then the rest (split_rows)
        char* part_ws = static_cast<char*>(ws) + part_offset(workspace_plain(2, B, N));
        unsigned* status = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + ccvm_status_offset(2, B, N));
        for (int r0 = 0; r0 < B; r0 = r0 ? B : cut) {
            const int rows = r0 ? B - cut : cut;
            const size_t off = (size_t)r0 * ld, bytes = workspace_plain(2, rows, N);
            ccvm_noise n = *nz;  // (replay: the part's columns of the batch's blocks)
            n.row_offset += r0;
            n.w_ld = (int64_t)noise_pitch(nz, B);
            if (n.w0) n.w0 += r0;
            if (n.w1) n.w1 += r0;
            ccvm_adam ad;
            if (adam) {
                ad = *adam;
                if (ad.m) ad.m += off;
                if (ad.v) ad.v += off;
            }
            if ((rc = ccvm_langevin_run(Q, V, c + off, rows, N, ld, step0, nsteps, T, p, adam ? &ad : nullptr, &n, part_ws, bytes,
                                        stream)))
                return rc;
            hipLaunchKernelGGL(status_merge_kernel, dim3(1), dim3(1), 0, st, status,
                               reinterpret_cast<unsigned*>(part_ws + ccvm_status_offset(2, rows, N)));
            CCVM_CHECK_LAUNCH(fn);
            part_ws += part_offset(bytes);
        }
        return CCVM_OK;
    }

    // workspace: [c' = exchange buffer 0][exchange buffer 1][column sums of Q][schedule table][cluster sync]
    const size_t state = (size_t)ccvm_rows(B) * ld;
    float* buf[2] = {c, static_cast<float*>(ws)};
    if (!(nz->flags & CCVM_RUN_WS_PADDED) && hipMemsetAsync(ws, 0, 2 * state * sizeof(float), st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: memset failed", fn);

    const double ul = p->upper - p->lower, up = p->upper + p->lower;
    StepArgs a;
    base_args(a, Q, V, B, N, ld, tun, s_full ? 2 : 4, s_full ? -1 : MODE_LANGEVIN);  // (the composed per-element-saturation path below runs MODE_AFFINE)
    a.in_scale = (float)(ul / (2.0 * S_eff));  // langevin_solver.py:133
    a.in_shift = (float)(up / 2.0);
    if ((rc = compute_qsum(Q, N, ld, static_cast<float*>(ws) + 2 * state, st, &a.qsum, p->qsum))) return rc;
    if (s_cols) {
        a.Q = scaled_rows(Q, s_cols, N, ld, ws, workspace_plain(2, B, N), st);
        a.s_cols = s_cols;
    }
    char* after = static_cast<char*>(ws) + 2 * state * sizeof(float) + qsum_area_bytes(N);
    float* table = reinterpret_cast<float*>(after);
    if (s_full) {
        // one saturation per trajectory AND variable: composed path (see ccvm_mf_run): workspace = [xs = c / S][y]
        float* xs = static_cast<float*>(ws);
        float* y = xs + state;
        hipLaunchKernelGGL(fulls_scale_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, st, c, s_full, xs, B, N, ld);
        CCVM_CHECK_LAUNCH(fn);
        a.a0 = xs;
        a.o0 = y;
        a.s.pp.step = (float)(-ul / 2.0);  // g_q for S = 1; 1 / S_bj in the update kernel
        a.s.pp.eps = (float)(-ul / 2.0);   // g_v
        const float f_q = a.s.pp.step, f_v = a.s.pp.eps;
        FullSArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.x0 = c; fa.xs = xs; fa.y = y; fa.s_full = s_full;
        fa.seed = nz->seed; fa.row_offset = nz->row_offset; fa.B = B; fa.N = N; fa.ld = ld;
        fa.adam = use_adam;
        if (use_adam) { fa.am = adam->m; fa.av = adam->v; }
        const bool replay = nz->mode == CCVM_NOISE_REPLAY;
        for (int i = step0; i < step0 + nsteps; ++i) {
            a.s.pp.step = f_q;  // (the scalar union is shared with the solver scalars below)
            a.s.pp.eps = f_v;
            if ((rc = launch_step<MODE_AFFINE, false>(a, st, fn))) return rc;
            LvScalars k;
            k.g_q = f_q;
            k.g_v = f_v;
            const double p_i = p->pump_rate_flag ? p->pump * (double)(i + 1) / (double)T : p->pump;
            k.pm = (float)(-1.0 + p_i);
            k.dt = (float)p->dt;
            k.dt_fs = (float)(p->dt * p->feedback_scale);
            k.w = (float)(p->sigma * std::sqrt(p->dt));
            k.S = 1.0f;
            k.use_pump = p->use_pump;
            if (use_adam) fill_adam(fa.ad, adam, i);
            fa.step = i;
            fa.w0 = replay ? nz->w0 + (size_t)(i - step0) * N * B : nullptr;
            hipLaunchKernelGGL(fulls_langevin_update_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, st, fa, k);
            CCVM_CHECK_LAUNCH(fn);
        }
        return CCVM_OK;
    }
    if (want_persist(N, tun, MODE_LANGEVIN, B)) {
        PersistArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.Q = a.Q; pa.V = V; pa.qsum = a.qsum; pa.x0 = c; pa.table = table; pa.s_cols = s_cols;
        pa.seed = nz->seed; pa.row_offset = nz->row_offset; pa.replay = nz->mode == CCVM_NOISE_REPLAY;
        pa.B = B; pa.N = N; pa.ld = ld; pa.wld = (int)noise_pitch(nz, B); pa.in_scale = a.in_scale; pa.in_shift = a.in_shift;
        pa.ru_override = tun.persist_ru;
        pa.kh_override = tun.persist_kh;
        pa.pw_override = tun.persist_pw;
        pa.rsw_override = tun.persist_rsw;
        pa.cw_override = tun.persist_cw;
        pa.xs_override = tun.persist_xs;
        pa.simds = 4 * chip_of(tun).cus;
        AdamSched asc;
        persist_adam(pa, asc, adam, use_adam);
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            LvSched sc{p->dt, p->sigma, p->feedback_scale, S_eff, p->pump, ul, p->use_pump, p->pump_rate_flag, T,
                       step0 + done, k, asc};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(lv_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            pa.step0 = step0 + done;
            pa.table = rows;
            pa.nsteps = k;
            if (pa.replay) pa.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
            rc = use_adam ? launch_persist<MODE_LANGEVIN, true>(pa, st, fn)
                          : launch_persist<MODE_LANGEVIN, false>(pa, st, fn);
            if (rc) return rc;
        }
        return CCVM_OK;
    }
    if (const SlabPlan sp = want_slab(B, N, tun, MODE_LANGEVIN); sp.ok) {
        // small batch: whole chunks in one launch each, Q resident in the members' registers (ccvm_slab.h)
        SlabArgs sa;
        unsigned xid;
        if (slab_base(sa, xid, sp, a.Q, V, a.qsum, B, N, ld, nz, table, after + table_bytes(), step0, st, tun, 1))
            return fail(CCVM_E_HIP, "%s: memset failed", fn);
        sa.x0 = c;
        sa.in_scale = a.in_scale; sa.in_shift = a.in_shift; sa.s_cols = s_cols;
        PersistArgs pa_ad;  // the Adam constants in the persistent kernels' form
        std::memset(&pa_ad, 0, sizeof(pa_ad));
        AdamSched asc;
        persist_adam(pa_ad, asc, adam, use_adam);
        sa.adam = use_adam; sa.ad = pa_ad.ad; sa.am = pa_ad.am; sa.av = pa_ad.av;
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            LvSched sc{p->dt, p->sigma, p->feedback_scale, S_eff, p->pump, ul, p->use_pump, p->pump_rate_flag, T,
                       step0 + done, k, asc};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(lv_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            sa.step0 = step0 + done;
            sa.table = rows;
            sa.nsteps = k;
            if (sa.replay) sa.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
            slab_launch_lv(sa, sp, st);
            CCVM_CHECK_LAUNCH(fn);
        }
        if (exchange_commit(sa.status, xid, step0 + nsteps, st)) return fail(CCVM_E_HIP, "%s: launch failed", fn);
        return CCVM_OK;
    }
    if (want_cluster(B, N, tun, MODE_LANGEVIN, use_adam)) {
        // whole chunks of the trajectory in one launch each, Q panels resident in LDS (ccvm_cluster.h)
        ClusterArgs ca;
        unsigned xid;
        if (cluster_base(ca, xid, a.Q, V, a.qsum, B, N, ld, nz, table, after + table_bytes(), step0, st, tun))
            return fail(CCVM_E_HIP, "%s: memset failed", fn);
        ca.x0 = c;
        ca.in_scale = a.in_scale; ca.in_shift = a.in_shift; ca.s_cols = s_cols;
        PersistArgs pa_ad;  // the Adam constants in the persistent kernels' form
        std::memset(&pa_ad, 0, sizeof(pa_ad));
        AdamSched asc;
        persist_adam(pa_ad, asc, adam, use_adam);
        ca.ad = pa_ad.ad; ca.am = pa_ad.am; ca.av = pa_ad.av;
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            LvSched sc{p->dt, p->sigma, p->feedback_scale, S_eff, p->pump, ul, p->use_pump, p->pump_rate_flag, T,
                       step0 + done, k, asc};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(lv_schedule_kernel, dim3((k + 255) / 256), dim3(256), 0, st, sc, table);
                rows = table;
            }
            ca.step0 = step0 + done;
            ca.table = rows;
            ca.nsteps = k;
            if (ca.replay) ca.w0 = nz->w0 + (size_t)done * N * noise_pitch(nz, B);
            cluster_launch_lv(ca, use_adam, st);
            CCVM_CHECK_LAUNCH(fn);
        }
        if (exchange_commit(ca.status, xid, step0 + nsteps, st)) return fail(CCVM_E_HIP, "%s: launch failed", fn);
        return CCVM_OK;
    }
    if (want_ptile(a, tun, MODE_LANGEVIN, s_cols != nullptr)) {
        // whole chunks in one launch each, the tile grid resident (ccvm_ptile.h)
        unsigned* status = reinterpret_cast<unsigned*>(after + table_bytes() + exchange_bytes(B, N, 1));
        PersistArgs pa_ad;  // (the Adam schedule constants in the persistent kernels' form)
        std::memset(&pa_ad, 0, sizeof(pa_ad));
        AdamSched asc;
        persist_adam(pa_ad, asc, adam, use_adam);
        float* const none[2] = {nullptr, nullptr};
        for (int done = 0; done < nsteps; done += TABLE_STEPS) {
            const int k = std::min(TABLE_STEPS, nsteps - done);
            LvSched sc{p->dt, p->sigma, p->feedback_scale, S_eff, p->pump, ul, p->use_pump, p->pump_rate_flag, T,
                       step0 + done, k, asc, reinterpret_cast<unsigned*>(after + table_bytes()), a.nrb * PT_FLAG_WORDS};
            const float* rows = given_rows(p->schedule, nz, step0 + done, k, sc.flags, sc.flag_words, st);
            if (!rows) {
                hipLaunchKernelGGL(lv_schedule_kernel, dim3(ptile_sched_grid(k, a.nrb)), dim3(256), 0, st, sc, table);
                rows = table;
            }
            if ((rc = run_ptile<MODE_LANGEVIN>(a, buf, none, nz, rows, after + table_bytes(), status, step0, done, k, tun, st,
                                               fn, done & 1, adam)))
                return rc;
        }
        if ((nsteps & 1) && hipMemcpyAsync(c, buf[1], state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(CCVM_E_HIP, "%s: copy-back failed", fn);
        return CCVM_OK;
    }
    if (use_adam) {
        a.am = adam->m;
        a.av = adam->v;
    }
    int cur = 0;
    for (int i = step0; i < step0 + nsteps; ++i) {
        LvScalars& k = a.s.lv;
        k.g_q = (float)(-ul / (2.0 * S_eff));
        k.g_v = k.g_q;
        const double p_i = p->pump_rate_flag ? p->pump * (double)(i + 1) / (double)T : p->pump;  // pl:279-282
        k.pm = (float)(-1.0 + p_i);
        k.dt = (float)p->dt;
        k.dt_fs = (float)(p->dt * p->feedback_scale);
        k.w = (float)(p->sigma * std::sqrt(p->dt));
        k.S = (float)S_eff;
        k.use_pump = p->use_pump;
        if (use_adam) fill_adam(a.ad, adam, i);
        a.a0 = buf[cur];
        a.o0 = buf[cur ^ 1];
        set_noise(a, nz, i, step0, B, N, false, false);
        rc = use_adam ? launch_step<MODE_LANGEVIN, true>(a, st, fn) : launch_step<MODE_LANGEVIN, false>(a, st, fn);
        if (rc) return rc;
        cur ^= 1;
    }
    if (cur == 1) {
        if (hipMemcpyAsync(c, buf[1], state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(CCVM_E_HIP, "%s: copy-back failed", fn);
    }
    return CCVM_OK;
}

int ccvm_clamp(float* x, int B, int N, int ld, float lo, float hi, void* stream) {
    int rc;
    if (!x) return fail(CCVM_E_INVALID, "ccvm_clamp: NULL argument");
    if ((rc = check_layout("ccvm_clamp", B, N, ld))) return rc;
    hipLaunchKernelGGL(clamp_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, (hipStream_t)stream, x, B, N, ld,
                       lo, hi);
    CCVM_CHECK_LAUNCH("ccvm_clamp");
    return CCVM_OK;
}

int ccvm_clamp_cols(float* x, int B, int N, int ld, const float* s_cols, void* stream) {
    int rc;
    if (!x || !s_cols) return fail(CCVM_E_INVALID, "ccvm_clamp_cols: NULL argument");
    if ((rc = check_layout("ccvm_clamp_cols", B, N, ld))) return rc;
    hipLaunchKernelGGL(clamp_cols_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, (hipStream_t)stream, x, B, N,
                       ld, s_cols);
    CCVM_CHECK_LAUNCH("ccvm_clamp_cols");
    return CCVM_OK;
}

int ccvm_change_variables_cols(const float* x, float* y, int B, int N, int ld, const float* s_cols, double lower,
                               double upper, void* stream) {
    int rc;
    if (!x || !y || !s_cols) return fail(CCVM_E_INVALID, "ccvm_change_variables_cols: NULL argument");
    if ((rc = check_layout("ccvm_change_variables_cols", B, N, ld))) return rc;
    hipLaunchKernelGGL(change_variables_cols_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, (hipStream_t)stream,
                       x, y, B, N, ld, s_cols, (float)(upper - lower), (float)(0.5 * (upper + lower)));
    CCVM_CHECK_LAUNCH("ccvm_change_variables_cols");
    return CCVM_OK;
}

int ccvm_clamp_full(float* x, int B, int N, int ld, const float* lo, const float* hi, void* stream) {
    int rc;
    if (!x || !lo || !hi) return fail(CCVM_E_INVALID, "ccvm_clamp_full: NULL argument");
    if ((rc = check_layout("ccvm_clamp_full", B, N, ld))) return rc;
    hipLaunchKernelGGL(clamp_full_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, (hipStream_t)stream, x, B, N,
                       ld, lo, hi);
    CCVM_CHECK_LAUNCH("ccvm_clamp_full");
    return CCVM_OK;
}

int ccvm_change_variables_full(const float* x, float* y, int B, int N, int ld, const float* s_full, double lower,
                               double upper, void* stream) {
    int rc;
    if (!x || !y || !s_full) return fail(CCVM_E_INVALID, "ccvm_change_variables_full: NULL argument");
    if ((rc = check_layout("ccvm_change_variables_full", B, N, ld))) return rc;
    hipLaunchKernelGGL(change_variables_full_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, (hipStream_t)stream,
                       x, y, B, N, ld, s_full, (float)(upper - lower), (float)(0.5 * (upper + lower)));
    CCVM_CHECK_LAUNCH("ccvm_change_variables_full");
    return CCVM_OK;
}

int ccvm_change_variables(const float* x, float* y, int B, int N, int ld, double S, double lower, double upper,
                          void* stream) {
    int rc;
    if (!x || !y) return fail(CCVM_E_INVALID, "ccvm_change_variables: NULL argument");
    if ((rc = check_layout("ccvm_change_variables", B, N, ld))) return rc;
    if (!(S > 0)) return fail(CCVM_E_INVALID, "ccvm_change_variables: S must be positive");
    hipLaunchKernelGGL(change_variables_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, (hipStream_t)stream,
                       x, y, B, N, ld, (float)S, (float)(upper - lower), (float)(0.5 * (upper + lower)));
    CCVM_CHECK_LAUNCH("ccvm_change_variables");
    return CCVM_OK;
}

int ccvm_energy(const float* Q, const float* V, const float* x, int B, int N, int ld, double scaled_by,
                float* obj, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_energy";
    const Tuning tun = read_tuning();
    int rc;
    if (!Q || !V || !x || !obj) return fail(CCVM_E_INVALID, "%s: NULL argument", fn);
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if (!aligned16(Q) || !aligned16(x) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, x and workspace must be 16-byte aligned", fn);
    if (ws_bytes < ccvm_workspace_bytes(3, B, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    hipStream_t st = (hipStream_t)stream;
    StepArgs a;
    base_args(a, Q, V, B, N, ld, tun);
    a.a0 = x;
    a.o0 = static_cast<float*>(ws);
    if ((rc = launch_step<MODE_ENERGY, false>(a, st, fn))) return rc;
    hipLaunchKernelGGL(energy_reduce_kernel, dim3((B + 255) / 256), dim3(256), 0, st,
                       static_cast<const float*>(ws), (N + 31) / 32, a.nrb * BM, B, (float)scaled_by, obj);
    CCVM_CHECK_LAUNCH(fn);
    return CCVM_OK;
}

int ccvm_objective_stats(const float* obj, int B, double optimal_value, ccvm_solution_stats* stats, void* stream) {
    static_assert(sizeof(ccvm_solution_stats) == sizeof(ObjectiveStats), "ccvm_solution_stats layout");
    if (!obj || !stats || B <= 0) return fail(CCVM_E_INVALID, "ccvm_objective_stats: bad argument");
    hipLaunchKernelGGL(objective_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, obj, B,
                       (float)optimal_value, reinterpret_cast<ObjectiveStats*>(stats));
    CCVM_CHECK_LAUNCH("ccvm_objective_stats");
    return CCVM_OK;
}

int ccvm_finalize(const float* Q, const float* V, float* state, float* x, int B, int N, int ld,
                  const ccvm_finalize_params* p, float* obj, ccvm_solution_stats* stats, void* ws, size_t ws_bytes,
                  void* stream) {
    const char* fn = "ccvm_finalize";
    const Tuning tun = read_tuning();
    int rc;
    if (!Q || !V || !state || !x || !p || !obj) return fail(CCVM_E_INVALID, "%s: NULL argument", fn);
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if (!aligned16(Q) || !aligned16(state) || !aligned16(x) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, state, x and workspace must be 16-byte aligned", fn);
    if (ws_bytes < ccvm_workspace_bytes(3, B, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    if (p->change_variables && !(p->s_cols || p->s_full || p->S > 0))
        return fail(CCVM_E_INVALID, "%s: S must be positive", fn);
    if (p->s_cols && p->s_full) return fail(CCVM_E_INVALID, "%s: s_cols and s_full are exclusive", fn);
    if (p->clamp && !p->s_cols && !p->s_full && !(p->clamp_hi >= p->clamp_lo))
        return fail(CCVM_E_INVALID, "%s: clamp bounds out of order", fn);
    hipStream_t st = (hipStream_t)stream;
    // 1. clamp (in place) + change of variables: state -> x
    if (p->clamp || p->change_variables || x != state) {
        hipLaunchKernelGGL(finalize_prepare_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, st, state, x, B, N, ld,
                           p->clamp, (float)p->clamp_lo, (float)p->clamp_hi, p->change_variables, (float)p->S,
                           p->s_cols, p->s_full, (float)(p->upper - p->lower), (float)(0.5 * (p->upper + p->lower)));
        CCVM_CHECK_LAUNCH(fn);
    }
    // 2. x @ Q with the row-dot epilogue (column-strip partials), 3. fixed-order strip sum * scaled_by
    StepArgs a;
    base_args(a, Q, V, B, N, ld, tun);
    a.a0 = x;
    a.o0 = static_cast<float*>(ws);
    if ((rc = launch_step<MODE_ENERGY, false>(a, st, fn))) return rc;
    hipLaunchKernelGGL(energy_reduce_kernel, dim3((B + 255) / 256), dim3(256), 0, st,
                       static_cast<const float*>(ws), (N + 31) / 32, a.nrb * BM, B, (float)p->scaled_by, obj);
    CCVM_CHECK_LAUNCH(fn);
    // 4. best objective value + the seven gap counters
    if (stats) return ccvm_objective_stats(obj, B, p->optimal_value, stats, stream);
    return CCVM_OK;
}

int ccvm_feedback(const float* Q, const float* V, const float* x, float* y, int B, int N, int ld, double in_scale,
                  double in_shift, double f_q, double f_v, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_feedback";
    const Tuning tun = read_tuning();
    int rc;
    if (!Q || !V || !x || !y || x == y) return fail(CCVM_E_INVALID, "%s: NULL or aliased argument", fn);
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if (!aligned16(Q) || !aligned16(x) || !aligned16(y) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, x, y and workspace must be 16-byte aligned", fn);
    if (ws_bytes < ccvm_workspace_bytes(5, B, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    StepArgs a;
    base_args(a, Q, V, B, N, ld, tun);
    a.in_scale = (float)in_scale;
    a.in_shift = (float)in_shift;
    if ((rc = compute_qsum(Q, N, ld, static_cast<float*>(ws), (hipStream_t)stream, &a.qsum))) return rc;
    a.a0 = x;
    a.o0 = y;
    a.s.pp.step = (float)f_q;
    a.s.pp.eps = (float)f_v;
    return launch_step<MODE_AFFINE, false>(a, (hipStream_t)stream, fn);
}

int ccvm_pp_grad_descent(const float* Q, const float* V, float* x, int B, int N, int ld, int iters, double step,
                         double lo, double hi, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_pp_grad_descent";
    const Tuning tun = read_tuning();
    int rc;
    if (!Q || !V || !x || iters < 0) return fail(CCVM_E_INVALID, "%s: bad argument", fn);
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if (!aligned16(Q) || !aligned16(x) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, x and workspace must be 16-byte aligned", fn);
    if (ws_bytes < ccvm_workspace_bytes(4, B, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    if (iters == 0) return CCVM_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t state = (size_t)ccvm_rows(B) * ld;
    float* buf[2] = {x, static_cast<float*>(ws)};
    if (hipMemsetAsync(ws, 0, state * sizeof(float), st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: memset failed", fn);
    StepArgs a;
    base_args(a, Q, V, B, N, ld, tun);
    a.s.pp.step = (float)step;
    a.s.pp.lo = (float)lo;
    a.s.pp.hi = (float)hi;
    int cur = 0;
    for (int i = 0; i < iters; ++i) {
        a.a0 = buf[cur];
        a.o0 = buf[cur ^ 1];
        if ((rc = launch_step<MODE_GD, false>(a, st, fn))) return rc;
        cur ^= 1;
    }
    if (cur == 1 &&
        hipMemcpyAsync(x, buf[1], state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: copy-back failed", fn);
    return CCVM_OK;
}

int ccvm_pp_adam(const float* Q, const float* V, float* x, int B, int N, int ld, double lr, double eps, double lo,
                 double hi, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_pp_adam";
    const Tuning tun = read_tuning();
    int rc;
    if (!Q || !V || !x) return fail(CCVM_E_INVALID, "%s: NULL argument", fn);
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if (!aligned16(Q) || !aligned16(x) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, x and workspace must be 16-byte aligned", fn);
    if (ws_bytes < ccvm_workspace_bytes(4, B, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    hipStream_t st = (hipStream_t)stream;
    const size_t state = (size_t)ccvm_rows(B) * ld;
    float* xn = static_cast<float*>(ws);
    float* qs = xn + state;
    if (hipMemsetAsync(xn, 0, state * sizeof(float), st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: memset failed", fn);
    hipLaunchKernelGGL(symmetrize_kernel, dim3(ew_grid((size_t)ld * ld)), dim3(256), 0, st, Q, qs, ld);
    CCVM_CHECK_LAUNCH(fn);
    StepArgs a;
    base_args(a, qs, V, B, N, ld, tun);
    a.a0 = x;
    a.o0 = xn;
    a.s.pp.step = (float)lr;
    a.s.pp.eps = (float)eps;
    a.s.pp.lo = (float)lo;
    a.s.pp.hi = (float)hi;
    if ((rc = launch_step<MODE_ADAMPP, false>(a, st, fn))) return rc;
    if (hipMemcpyAsync(x, xn, state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: copy-back failed", fn);
    return CCVM_OK;
}

int ccvm_pp_asgd(const float* Q, const float* V, float* x, int B, int N, int ld, double lr, double lambd, double lo,
                 double hi, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_pp_asgd";
    const Tuning tun = read_tuning();
    int rc;
    if (!Q || !V || !x) return fail(CCVM_E_INVALID, "%s: NULL argument", fn);
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if (!aligned16(Q) || !aligned16(x) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, x and workspace must be 16-byte aligned", fn);
    if (ws_bytes < ccvm_workspace_bytes(4, B, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    hipStream_t st = (hipStream_t)stream;
    const size_t state = (size_t)ccvm_rows(B) * ld;
    float* xn = static_cast<float*>(ws);
    float* qs = xn + state;
    if (hipMemsetAsync(xn, 0, state * sizeof(float), st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: memset failed", fn);
    hipLaunchKernelGGL(symmetrize_kernel, dim3(ew_grid((size_t)ld * ld)), dim3(256), 0, st, Q, qs, ld);
    CCVM_CHECK_LAUNCH(fn);
    StepArgs a;
    base_args(a, qs, V, B, N, ld, tun);
    a.a0 = x;
    a.o0 = xn;
    a.s.pp.step = (float)lr;
    a.s.pp.eps = (float)(1.0 - lambd * lr);  // first step of torch.optim.ASGD: eta = lr
    a.s.pp.lo = (float)lo;
    a.s.pp.hi = (float)hi;
    if ((rc = launch_step<MODE_ASGDPP, false>(a, st, fn))) return rc;
    if (hipMemcpyAsync(x, xn, state * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: copy-back failed", fn);
    return CCVM_OK;
}

int ccvm_pp_lbfgs(const float* Q, const float* V, float* x, int B, int N, int ld, int iters, double lr, double lo,
                  double hi, void* ws, size_t ws_bytes, void* stream) {
    const char* fn = "ccvm_pp_lbfgs";
    const Tuning tun = read_tuning();
    int rc;
    if (!Q || !V || !x || iters < 0) return fail(CCVM_E_INVALID, "%s: bad argument", fn);
    if ((rc = check_layout(fn, B, N, ld))) return rc;
    if (!aligned16(Q) || !aligned16(x) || !aligned16(ws))
        return fail(CCVM_E_LAYOUT, "%s: Q, x and workspace must be 16-byte aligned", fn);
    if (ws_bytes < ccvm_workspace_bytes(4, B, N)) return fail(CCVM_E_WORKSPACE, "%s: workspace too small", fn);
    if (iters == 0) return CCVM_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t state = (size_t)ccvm_rows(B) * ld;
    float* grad = static_cast<float*>(ws);
    float* qs = grad + state;
    if (hipMemsetAsync(grad, 0, state * sizeof(float), st) != hipSuccess)
        return fail(CCVM_E_HIP, "%s: memset failed", fn);
    hipLaunchKernelGGL(symmetrize_kernel, dim3(ew_grid((size_t)ld * ld)), dim3(256), 0, st, Q, qs, ld);
    CCVM_CHECK_LAUNCH(fn);
    StepArgs a;
    base_args(a, qs, V, B, N, ld, tun);  // g = x @ 1/2 (Q + Q') + V
    a.a0 = x;
    a.o0 = grad;
    a.s.pp.step = 1.0f;
    a.s.pp.eps = 1.0f;
    for (int i = 0; i < iters; ++i) {
        if ((rc = launch_step<MODE_AFFINE, false>(a, st, fn))) return rc;
        hipLaunchKernelGGL(lbfgs_row_kernel, dim3(B), dim3(256), 0, st, x, grad, N, ld, (float)lr, (float)lo, (float)hi);
        CCVM_CHECK_LAUNCH(fn);
    }
    return CCVM_OK;
}

int ccvm_philox_normals(uint64_t seed, int64_t row_offset, int step, int B, int N, float* w0, float* w1,
                        void* stream) {
    if (!w0 || B <= 0 || N <= 0 || step < 0) return fail(CCVM_E_INVALID, "ccvm_philox_normals: bad argument");
    hipLaunchKernelGGL(philox_fill_kernel, dim3(ew_grid((size_t)B * N)), dim3(256), 0, (hipStream_t)stream, seed,
                       row_offset, step, B, N, w0, w1);
    CCVM_CHECK_LAUNCH("ccvm_philox_normals");
    return CCVM_OK;
}

}  // extern "C"
