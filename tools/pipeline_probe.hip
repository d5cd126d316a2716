// Can a one-workgroup-per-CU patch kernel overlap its sweep phase with the HBM loads of its NEXT patch?
// (tuning helper: de-risks a persistent, LDS-DMA-pipelined k_strip4_jacobi)
//
// 256 persistent workgroups x 1024 threads, P patches each.  Per patch a thread needs 25 doubles (the ~200 KB a
// masked 64 x 64 Jacobi patch loads) and then runs K register-resident sweeps (same code as sweep_probe).
//   A  serial      : 25 plain loads -> wait -> K sweeps                         (today's kernel)
//   B  pipelined   : 13 plain loads -> wait; 12 doubles come from LDS, where LDS-DMA (global_load_lds, 16 B/lane)
//                    put them while the PREVIOUS patch was sweeping; raw s_barrier in the sweeps so that nothing
//                    drains the DMAs; vmcnt(0) + barrier before the staged data is read
//   C  loads only, D  sweeps only   (the two phases alone)
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/pipeline_probe.hip -o tools/pipeline_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ double dpp_from_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_prev(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int NDIRECT_A = 25, NDIRECT_B = 13, NSTAGED = 12;   // doubles per thread and patch
constexpr int64_t PATCH_DOUBLES = 1024 * 25;                  // 200 KB

// MODE 0: A serial, 1: B pipelined, 2: loads only, 3: sweeps only
template <int MODE>
__global__ void __launch_bounds__(1024) k_patches(const double* __restrict__ src, double* __restrict__ out, int P, int K) {
    __shared__ double top[2][16][64], bot[2][16][64];
    extern __shared__ __attribute__((aligned(16))) double stage_flat[];    // NSTAGED x 1024 doubles = 96 KB (dynamic)
    double (*stage)[1024] = reinterpret_cast<double (*)[1024]>(stage_flat);
    const int lx = threadIdx.x & 63, st = threadIdx.x >> 6;
    double lv[4][6], bv[4], x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = 0.0; bv[r] = 0.1; for (int s = 0; s < 6; ++s) lv[r][s] = -0.1; }
    double acc_out = 0.0;
    auto patch_base = [&](int p) { return src + ((int64_t)(p * gridDim.x + blockIdx.x)) * PATCH_DOUBLES; };
    auto issue_stage = [&](int p) {   // 12 KB per wave = 6 LDS-DMA of 1 KB; array a, rows of 1024 doubles
        const double* base = patch_base(p) + (int64_t)NDIRECT_B * 1024;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            // wave st, instruction j covers stage[(st*6 + j) / 8 ...]: simply a linear 96 KB copy, 1 KB per instruction
            const int64_t off = ((int64_t)(st * 6 + j) * 64 + lx) * 2;   // doubles
            // inline asm, not __builtin_amdgcn_global_load_lds: hipcc orders every later LDS access behind a DMA it knows
            // of with s_waitcnt vmcnt(0), which would drain the prefetch before the first sweep
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(stage_flat + (st * 6 + j) * 128));
            const double* gsrc = base + off;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        }
    };
    if (MODE == 1) {
        issue_stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int p = 0; p < P; ++p) {
        const double* base = patch_base(p);
        if (MODE == 0 || MODE == 2) {
            double v[NDIRECT_A];
#pragma unroll
            for (int j = 0; j < NDIRECT_A; ++j) v[j] = base[(int64_t)j * 1024 + threadIdx.x];
#pragma unroll
            for (int j = 0; j < NDIRECT_A; ++j) { lv[j & 3][j % 6] += 1e-9 * v[j]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = v[r];
        } else if (MODE == 1) {
            double v[NDIRECT_B];
#pragma unroll
            for (int j = 0; j < NDIRECT_B; ++j) v[j] = base[(int64_t)j * 1024 + threadIdx.x];
            double w[NSTAGED];
#pragma unroll
            for (int j = 0; j < NSTAGED; ++j) w[j] = stage[j][threadIdx.x];
#pragma unroll
            for (int j = 0; j < NDIRECT_B; ++j) { lv[j & 3][j % 6] += 1e-9 * v[j]; }
#pragma unroll
            for (int j = 0; j < NSTAGED; ++j) { lv[j & 3][(j + 1) % 6] += 1e-9 * w[j]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = v[r];
            // pin: every value that came from a plain load is in its register BEFORE the DMAs are issued -- hipcc does
            // not count asm DMAs, so a vmcnt(0) it emits later for one of its own loads would drain them as well
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                asm volatile("" : "+v"(lv[r][0]), "+v"(lv[r][1]), "+v"(lv[r][2]), "+v"(lv[r][3]), "+v"(lv[r][4]), "+v"(lv[r][5]),
                             "+v"(x[r]));
            }
            // everyone has copied the staged values out: the buffers may be refilled for the next patch
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (p + 1 < P) issue_stage(p + 1);
        }
        if (MODE != 2) {
            for (int k = 0; k < K; ++k) {
                const int par = k & 1;
                bot[par][st][lx] = x[0];
                top[par][st][lx] = x[3];
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // raw: must not drain the DMAs
                const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;
                const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;
                double e_[4], w_[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { e_[r] = dpp_from_next(x[r]); w_[r] = dpp_from_prev(x[r]); }
                const double ea_ = dpp_from_next(above), wb_ = dpp_from_prev(below);
                double xn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double acc = bv[r];
                    acc = fma(-lv[r][0], e_[r], acc);
                    acc = fma(-lv[r][1], r < 3 ? e_[(r + 1) & 3] : ea_, acc);
                    acc = fma(-lv[r][2], r < 3 ? x[(r + 1) & 3] : above, acc);
                    acc = fma(-lv[r][3], w_[r], acc);
                    acc = fma(-lv[r][4], r > 0 ? w_[(r + 3) & 3] : wb_, acc);
                    acc = fma(-lv[r][5], r > 0 ? x[(r + 3) & 3] : below, acc);
                    xn[r] = acc;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = xn[r];
            }
        }
        acc_out += x[0] + x[1] + x[2] + x[3];
        if (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next patch's staged data has landed
            __builtin_amdgcn_s_barrier();
        }
    }
    out[(int64_t)blockIdx.x * 1024 + threadIdx.x] = acc_out;
}

template <int MODE>
static float run(const char* name, const double* src, double* out, int P, int K) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipFuncSetAttribute((const void*)k_patches<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGED * 1024 * 8));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_patches<MODE>), dim3(256), dim3(1024), NSTAGED * 1024 * 8, 0, src, out, P, K);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double bytes = 256.0 * P * PATCH_DOUBLES * 8;
    printf("%-34s P=%d K=%d: %8.1f us  = %6.2f us per patch;  %.2f TB/s of patch data\n", name, P, K, best * 1e3, best * 1e3 / P,
           MODE == 3 ? 0.0 : bytes / (best * 1e-3) / 1e12);
    return best;
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 8, K = argc > 2 ? atoi(argv[2]) : 9;
    const size_t n = (size_t)256 * P * PATCH_DOUBLES;
    double *src, *out;
    CHECK(hipMalloc(&src, n * 8));
    CHECK(hipMalloc(&out, 256 * 1024 * 8));
    CHECK(hipMemset(src, 0, n * 8));
    run<2>("C loads only (25 per thread)", src, out, P, K);
    run<3>("D sweeps only", src, out, P, K);
    run<0>("A serial: loads then sweeps", src, out, P, K);
    run<1>("B pipelined: 13 direct + 12 staged", src, out, P, K);
    return 0;
}
