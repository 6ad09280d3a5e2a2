// Where do the waves of a workgroup land?  (developer tool, round 6)   hipcc --offload-arch=gfx950 -O2 tools/simd_probe.hip -o /tmp/simd_probe
// Every wave records HW_ID (wave slot, SIMD, CU, SE) and XCC_ID; the host prints, per workgroup size, which SIMD wave w of a
// workgroup ran on and how many waves the fullest SIMD of a CU held while the grid was resident.  Behind the row-owner
// kernel's row sets per workgroup (ccvm_persist_launch.h: a step costs what the fullest SIMD issues).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
    const int wave = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay resident for a while so that the whole grid is placed together
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        const int waves = blockDim.x >> 6;
        out[(blockIdx.x * waves + wave) * 2 + 0] = hw;
        out[(blockIdx.x * waves + wave) * 2 + 1] = xcc & 0xF;
    }
}

int main() {
    const int sizes[] = {192, 256, 384, 512, 640, 768};
    const int grids[] = {128, 256, 500};
    for (int threads : sizes) {
        for (int grid : grids) {
            const int waves = threads / 64;
            unsigned* d;
            hipMalloc(&d, (size_t)grid * waves * 8);
            hipMemset(d, 0, (size_t)grid * waves * 8);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), 0, 0, d, 20000);  // 200 us
            hipDeviceSynchronize();
            std::vector<unsigned> h((size_t)grid * waves * 2);
            hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
            hipFree(d);
            // SIMD of wave w, over all workgroups
            std::vector<std::vector<int>> hist(waves, std::vector<int>(4, 0));
            std::map<unsigned, std::vector<int>> per_cu;  // (xcc, se, sh, cu) -> waves per SIMD
            std::map<unsigned, int> wgs_on_cu;
            for (int b = 0; b < grid; ++b)
                for (int w = 0; w < waves; ++w) {
                    const unsigned hw = h[(b * waves + w) * 2], xcc = h[(b * waves + w) * 2 + 1];
                    const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    hist[w][simd]++;
                    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
                    auto& v = per_cu[key];
                    if (v.empty()) v.assign(4, 0);
                    v[simd]++;
                    if (w == 0) wgs_on_cu[key]++;
                }
            int fullest = 0, cus = (int)per_cu.size(), max_wg = 0;
            std::map<int, int> fullest_hist;
            for (auto& kv : per_cu) {
                int m = 0;
                for (int s = 0; s < 4; ++s) m = kv.second[s] > m ? kv.second[s] : m;
                fullest_hist[m]++;
                fullest = m > fullest ? m : fullest;
            }
            for (auto& kv : wgs_on_cu) max_wg = kv.second > max_wg ? kv.second : max_wg;
            std::printf("%4d threads (%2d waves) x %3d workgroups: %3d CUs used, at most %d workgroups on a CU; fullest SIMD of a CU holds",
                        threads, waves, grid, cus, max_wg);
            for (auto& kv : fullest_hist) std::printf(" %d waves on %d CUs,", kv.first, kv.second);
            std::printf("\n    SIMD of wave w (count over the workgroups, SIMD 0..3):");
            for (int w = 0; w < waves; ++w) std::printf("  w%d: %d/%d/%d/%d", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
            std::printf("\n");
            // which waves of a workgroup share a SIMD: the pattern (wave indices per SIMD, SIMDs in the order of their first wave)
            std::map<std::string, int> patterns;
            for (int b = 0; b < grid; ++b) {
                std::vector<std::string> per(4);
                std::vector<int> first(4, 99);
                for (int w = 0; w < waves; ++w) {
                    const int simd = (h[(b * waves + w) * 2] >> 4) & 3;
                    per[simd] += (per[simd].empty() ? "" : "+") + std::to_string(w);
                    if (first[simd] == 99) first[simd] = w;
                }
                std::string key;
                for (int f = 0; f < waves; ++f)
                    for (int sd = 0; sd < 4; ++sd)
                        if (first[sd] == f) key += "{" + per[sd] + "} ";
                patterns[key]++;
            }
            std::printf("    waves sharing a SIMD:");
            for (auto& kv : patterns) std::printf("  %s x %d;", kv.first.c_str(), kv.second);
            std::printf("\n");
        }
    }
    return 0;
}
