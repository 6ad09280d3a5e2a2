// A client of the C ABI with no Python and no torch in the process (test infrastructure): what a
// cgo / JNI / N-API / ctypes binding of include/ccvm_hip.h does, in C++.  Compiled and run by
// tests/test_gpu_api.py::test_c_abi_from_a_plain_cpp_client.  Checks, through the ABI only:
//   * layout helpers, pack / unpack round trip;
//   * every solver entry point runs (tile and persistent sizes), results are finite, padding stays 0;
//   * chunked runs equal single runs bit for bit (step0 / nsteps contract);
//   * two halves of a batch with row_offset equal the full batch bit for bit (sharding contract);
//   * ccvm_finalize (clamp -> change of variables -> energy -> success statistics in one call) equals the
//     separate calls bit for bit, its counters equal a host recount, the run's status word stays 0;
//   * errors come back as codes + ccvm_last_error(), never as exceptions.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ccvm_hip.h"

#define REQUIRE(cond)                                                          \
    do {                                                                       \
        if (!(cond)) {                                                         \
            printf("FAILED %s:%d: %s   [%s]\n", __FILE__, __LINE__, #cond, ccvm_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static unsigned g_rng = 12345;
static float rnd() {
    g_rng = g_rng * 1664525u + 1013904223u;
    return (g_rng >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

struct Dev {
    float* p = nullptr;
    size_t n = 0;
    explicit Dev(size_t count) : n(count) { hipMalloc(&p, count * 4); hipMemset(p, 0, count * 4); }
    ~Dev() { hipFree(p); }
    std::vector<float> host() const {
        std::vector<float> h(n);
        hipMemcpy(h.data(), p, n * 4, hipMemcpyDeviceToHost);
        return h;
    }
};

static bool all_finite(const std::vector<float>& v) {
    for (float x : v) if (!std::isfinite(x)) return false;
    return true;
}

static int run_case(int N, int B) {
    const int ld = ccvm_ld(N), rows = ccvm_rows(B);
    REQUIRE(ld >= N && ld % 128 == 0 && rows >= B && rows % 64 == 0);
    // a symmetric coupling matrix of the scale the solvers' scaling produces
    std::vector<float> q((size_t)N * N), v(N);
    for (int i = 0; i < N; ++i)
        for (int j = i; j < N; ++j) q[(size_t)i * N + j] = q[(size_t)j * N + i] = rnd() * 0.2f / std::sqrt((float)N);
    for (auto& x : v) x = rnd() * 0.2f;
    Dev qc((size_t)N * N), vc(N), Q((size_t)ld * ld), V(ld);
    hipMemcpy(qc.p, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(vc.p, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    REQUIRE(ccvm_pack(qc.p, N, N, N, Q.p, ld, ld, nullptr) == CCVM_OK);
    REQUIRE(ccvm_pack(vc.p, 1, N, N, V.p, 1, ld, nullptr) == CCVM_OK);
    {   // round trip
        Dev back((size_t)N * N);
        REQUIRE(ccvm_unpack(Q.p, ld, back.p, N, N, N, nullptr) == CCVM_OK);
        REQUIRE(back.host() == q);
    }
    const size_t state = (size_t)rows * ld;
    const int T = 24;
    ccvm_noise nz;
    memset(&nz, 0, sizeof(nz));
    nz.mode = CCVM_NOISE_PHILOX;
    nz.seed = 0xFEEDFACEULL;

    // ---- DL: one call == chunks == two half batches with row_offset ---------------------------
    ccvm_dl_params dl = {8.0, 0.001, 10.0, 100.0, 0.05, 0.0, 1.0, 1, 0};
    Dev ws(ccvm_workspace_bytes(0, B, N) / 4 + 1);
    Dev c1(state), s1(state), c2(state), s2(state);
    REQUIRE(ccvm_dl_run(Q.p, V.p, c1.p, s1.p, B, N, ld, 0, T, T, &dl, &nz, ws.p, ws.n * 4, nullptr) == CCVM_OK);
    // the chunked run passes column sums computed once (ccvm_column_sums) instead of letting every call recompute them
    Dev qsum(ld), wq(ccvm_workspace_bytes(5, 1, N) / 4 + 1);
    REQUIRE(ccvm_column_sums(Q.p, N, ld, qsum.p, wq.p, wq.n * 4, nullptr) == CCVM_OK);
    ccvm_dl_params dlq = dl;
    dlq.qsum = qsum.p;
    REQUIRE(ccvm_dl_run(Q.p, V.p, c2.p, s2.p, B, N, ld, 0, 5, T, &dlq, &nz, ws.p, ws.n * 4, nullptr) == CCVM_OK);
    REQUIRE(ccvm_dl_run(Q.p, V.p, c2.p, s2.p, B, N, ld, 5, T - 5, T, &dlq, &nz, ws.p, ws.n * 4, nullptr) == CCVM_OK);
    REQUIRE(hipDeviceSynchronize() == hipSuccess);
    const std::vector<float> hc1 = c1.host(), hs1 = s1.host();
    REQUIRE(all_finite(hc1) && all_finite(hs1));
    REQUIRE(hc1 == c2.host() && hs1 == s2.host());
    float moved = 0.f;
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < N; ++j) moved = std::fmax(moved, std::fabs(hc1[(size_t)b * ld + j]));
    REQUIRE(moved > 1e-3f);  // the trajectories did move
    for (int b = 0; b < rows; ++b)
        for (int j = 0; j < ld; ++j)
            if (b >= B || j >= N) REQUIRE(hc1[(size_t)b * ld + j] == 0.0f);  // padding stays zero
    if (B % 2 == 0) {
        const int hb = B / 2, hrows = ccvm_rows(hb);
        Dev ca((size_t)hrows * ld), sa((size_t)hrows * ld), cb((size_t)hrows * ld), sb((size_t)hrows * ld);
        Dev wsh(ccvm_workspace_bytes(0, hb, N) / 4 + 1);
        ccvm_noise nb = nz;
        REQUIRE(ccvm_dl_run(Q.p, V.p, ca.p, sa.p, hb, N, ld, 0, T, T, &dl, &nb, wsh.p, wsh.n * 4, nullptr) == CCVM_OK);
        nb.row_offset = hb;
        REQUIRE(ccvm_dl_run(Q.p, V.p, cb.p, sb.p, hb, N, ld, 0, T, T, &dl, &nb, wsh.p, wsh.n * 4, nullptr) == CCVM_OK);
        const std::vector<float> ha = ca.host(), hbv = cb.host();
        for (int b = 0; b < hb; ++b)
            for (int j = 0; j < N; ++j) {
                REQUIRE(ha[(size_t)b * ld + j] == hc1[(size_t)b * ld + j]);
                REQUIRE(hbv[(size_t)b * ld + j] == hc1[(size_t)(b + hb) * ld + j]);
            }
    }

    // ---- MF with Adam, Langevin, pumped Langevin: run, finite, chunk invariance ---------------------
    ccvm_mf_params mf = {0.0, 0.0025, 5.0, 4000.0, 0.01, 20.0, 0.0, 1.0, 1, 0};
    Dev wsm(ccvm_workspace_bytes(1, B, N) / 4 + 1);
    for (int pass = 0; pass < 2; ++pass) {
        Dev mu(state), sg(state), mt(state), am(state), av(state);
        std::vector<float> half((size_t)rows * ld, 0.f);
        for (int b = 0; b < B; ++b)
            for (int j = 0; j < N; ++j) half[(size_t)b * ld + j] = 0.5f;  // sigma starts at 1/2 (mf_solver.py:538-540)
        hipMemcpy(sg.p, half.data(), half.size() * 4, hipMemcpyHostToDevice);
        ccvm_adam ad = {1, 0, 0.001, 0.9, 0.999, am.p, av.p};
        static std::vector<float> first_mu;
        if (pass == 0) {
            REQUIRE(ccvm_mf_run(Q.p, V.p, mu.p, sg.p, mt.p, B, N, ld, 0, T, T, &mf, &ad, &nz, wsm.p, wsm.n * 4, nullptr) == CCVM_OK);
            first_mu = mu.host();
            REQUIRE(all_finite(first_mu) && all_finite(sg.host()) && all_finite(mt.host()));
        } else {
            REQUIRE(ccvm_mf_run(Q.p, V.p, mu.p, sg.p, mt.p, B, N, ld, 0, 7, T, &mf, &ad, &nz, wsm.p, wsm.n * 4, nullptr) == CCVM_OK);
            REQUIRE(ccvm_mf_run(Q.p, V.p, mu.p, sg.p, mt.p, B, N, ld, 7, T - 7, T, &mf, &ad, &nz, wsm.p, wsm.n * 4, nullptr) == CCVM_OK);
            REQUIRE(mu.host() == first_mu);
        }
    }
    for (int pumped = 0; pumped < 2; ++pumped) {
        ccvm_langevin_params lv = {0.002, 0.5, 1.0, 0.5, 2.0, 0.0, 1.0, pumped, 1};
        Dev wl(ccvm_workspace_bytes(2, B, N) / 4 + 1), x(state), obj(B), we(ccvm_workspace_bytes(3, B, N) / 4 + 1);
        REQUIRE(ccvm_langevin_run(Q.p, V.p, x.p, B, N, ld, 0, T, T, &lv, nullptr, &nz, wl.p, wl.n * 4, nullptr) == CCVM_OK);
        // the steps right after the loop in ONE call (on a copy), against the separate calls below
        Dev xcopy(state), xf(state), objf(B), stats(16);
        hipMemcpy(xcopy.p, x.p, state * 4, hipMemcpyDeviceToDevice);
        ccvm_finalize_params fp;
        memset(&fp, 0, sizeof(fp));
        fp.S = 0.5; fp.lower = 0.0; fp.upper = 1.0; fp.clamp = 1; fp.clamp_lo = -0.5; fp.clamp_hi = 0.5;
        fp.change_variables = 1; fp.scaled_by = 1.0; fp.optimal_value = 0.05;
        REQUIRE(ccvm_finalize(Q.p, V.p, xcopy.p, xf.p, B, N, ld, &fp, objf.p,
                              reinterpret_cast<ccvm_solution_stats*>(stats.p), we.p, we.n * 4, nullptr) == CCVM_OK);
        REQUIRE(ccvm_change_variables(x.p, x.p, B, N, ld, 0.5, 0.0, 1.0, nullptr) == CCVM_OK);
        REQUIRE(ccvm_energy(Q.p, V.p, x.p, B, N, ld, 1.0, obj.p, we.p, we.n * 4, nullptr) == CCVM_OK);
        const std::vector<float> hx = x.host(), ho = obj.host();
        REQUIRE(ho == objf.host() && hx == xf.host());
        {
            ccvm_solution_stats st;
            hipMemcpy(&st, stats.p, sizeof(st), hipMemcpyDeviceToHost);
            const float thr[7] = {0.1f, 1.f, 2.f, 3.f, 4.f, 5.f, 10.f};
            int want[7] = {0, 0, 0, 0, 0, 0, 0};
            float best = -INFINITY;
            for (int b = 0; b < B; ++b) {
                const float found = -ho[b], gap = ((float)fp.optimal_value - found) * 100.0f / std::fabs(found);
                best = std::fmax(best, found);
                for (int k = 0; k < 7; ++k) want[k] += gap <= thr[k];
            }
            REQUIRE(st.rows == B && st.nonfinite == 0 && st.best_objective_value == best);
            for (int k = 0; k < 7; ++k) REQUIRE(st.within[k] == want[k]);
            // the run's status word (solver workspaces): zeroed by the caller, still 0 after the run
            const size_t off = ccvm_status_offset(2, B, N);
            REQUIRE(off != (size_t)-1 && off + 4 <= wl.n * 4 && ccvm_status_offset(0, B, N) != (size_t)-1 &&
                    ccvm_status_offset(3, B, N) == (size_t)-1);
            unsigned status = 7;
            hipMemcpy(&status, reinterpret_cast<char*>(wl.p) + off, 4, hipMemcpyDeviceToHost);
            REQUIRE(status == 0);
            char what[256];
            REQUIRE(ccvm_describe_launch(2, B, N, 0, 0, what, sizeof(what)) == CCVM_OK && strstr(what, "_kernel<2, ") != nullptr);
        }
        REQUIRE(all_finite(hx) && all_finite(ho));
        for (int b = 0; b < B; ++b)
            for (int j = 0; j < N; ++j) REQUIRE(hx[(size_t)b * ld + j] >= 0.0f && hx[(size_t)b * ld + j] <= 1.0f);
    }

    // ---- error paths: codes + message, no exceptions ------------------------------------------------
    REQUIRE(ccvm_dl_run(Q.p, V.p, c1.p, s1.p, B, N, ld + 1, 0, 1, T, &dl, &nz, ws.p, ws.n * 4, nullptr) == CCVM_E_LAYOUT);
    REQUIRE(strlen(ccvm_last_error()) > 0);
    REQUIRE(ccvm_dl_run(Q.p, V.p, c1.p, s1.p, B, N, ld, 0, 1, T, &dl, &nz, ws.p, 16, nullptr) == CCVM_E_WORKSPACE);
    REQUIRE(ccvm_dl_run(nullptr, V.p, c1.p, s1.p, B, N, ld, 0, 1, T, &dl, &nz, ws.p, ws.n * 4, nullptr) == CCVM_E_INVALID);
    REQUIRE(ccvm_dl_run(Q.p, V.p, c1.p, s1.p, B, N, ld, T, 1, T, &dl, &nz, ws.p, ws.n * 4, nullptr) == CCVM_E_INVALID);
    printf("case N=%d B=%d ok\n", N, B);
    return 0;
}

int main() {
    if (ccvm_abi_version() != CCVM_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }
    if (run_case(37, 50)) return 1;    // persistent kernel (one wave per row set)
    if (run_case(100, 70)) return 1;   // persistent kernel (two waves side by side)
    if (run_case(300, 96)) return 1;   // per-step tile kernel (DL), column-cluster persistent kernel (MF, Langevin)
    if (run_case(600, 64)) return 1;   // per-step tile kernel for every solver
    printf("ABI_CLIENT_OK\n");
    return 0;
}
