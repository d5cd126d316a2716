// Which workgroups of a grid share a compute unit, and do two of them really run there at the same time?
// (tuning helper for k_strip_jacobi_pair_walk: 512 threads, ~79 KB of LDS, <= 128 VGPRs -- two per CU by the occupancy
// rules; the hand-off between a workgroup's load phase and its neighbour's sweeps only works if the pairs are known)
//   hipcc -O3 --offload-arch=gfx950 tools/occupancy_probe.hip -o tools/occupancy_probe && tools/occupancy_probe [wgs=512] [threads=512] [lds_kb=78]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Rec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };

__global__ void __launch_bounds__(1024) k_probe(Rec* out, int spin_ticks) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        Rec r;
        r.hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, all 32 bits
        r.xcc_id = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
        r.t0 = t0;
        r.t1 = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x] = r;
    }
    if (lds[(threadIdx.x + 1) % blockDim.x] < 0) out[0].t0 = 0;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 512, threads = argc > 2 ? atoi(argv[2]) : 512;
    const int lds_kb = argc > 3 ? atoi(argv[3]) : 78;
    Rec* d = nullptr;
    CHECK(hipMalloc((void**)&d, sizeof(Rec) * wgs));
    CHECK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(wgs), dim3(threads), (size_t)lds_kb * 1024, 0, d, 2000);   // 20 us
        CHECK(hipDeviceSynchronize());
    }
    std::vector<Rec> h(wgs);
    CHECK(hipMemcpy(h.data(), d, sizeof(Rec) * wgs, hipMemcpyDeviceToHost));
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se 3 bits + xcc from XCC_ID)
    std::map<unsigned, std::vector<int>> by_cu;
    unsigned long long tmin = ~0ull;
    for (auto& r : h) tmin = std::min(tmin, r.t0);
    for (int b = 0; b < wgs; ++b) {
        const unsigned cu = (h[b].hw_id >> 8) & 0xf, sh = (h[b].hw_id >> 12) & 1, se = (h[b].hw_id >> 13) & 7, xcc = h[b].xcc_id & 0xf;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    int concurrent = 0, cus = 0;
    std::map<int, int> hist;
    for (auto& kv : by_cu) {
        ++cus;
        hist[(int)kv.second.size()]++;
        // overlapping in time?
        for (size_t i = 0; i < kv.second.size(); ++i)
            for (size_t j = i + 1; j < kv.second.size(); ++j) {
                const Rec &a = h[kv.second[i]], &b = h[kv.second[j]];
                if (a.t0 < b.t1 && b.t0 < a.t1) ++concurrent;
            }
    }
    printf("%d workgroups of %d threads, %d KB LDS: %d distinct CUs;", wgs, threads, lds_kb, cus);
    for (auto& kv : hist) printf(" %d CUs hold %d workgroups;", kv.second, kv.first);
    printf(" %d pairs on one CU overlap in time\n", concurrent);
    int shown = 0;
    for (auto& kv : by_cu) {
        if (shown++ >= 6) break;
        printf("  cu %05x:", kv.first);
        for (int b : kv.second) printf(" wg %d [%llu..%llu]", b, h[b].t0 - tmin, h[b].t1 - tmin);
        printf("\n");
    }
    return 0;
}
