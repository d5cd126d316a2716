// Where do the cycles of one register-resident Jacobi sweep go?  (tuning helper for k_strip4_jacobi)
// Same structure as the product kernel: 1024 threads, a wave owns a 64-wide 4-row strip, 6 coefficients per
// node in registers, E/W neighbours by wave-wide DPP shifts, strip edge rows through LDS, one barrier per
// sweep.  Variants switch the parts off one by one; the output is the time of one sweep of one workgroup
// (all CUs busy, WG/CU = 1) in shader cycles per wave at the measured duration.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/sweep_probe.hip -o /tmp/sweep_probe && /tmp/sweep_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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
// row-local shifts (16-lane rows) -- wrong at row boundaries, timing only
__device__ __forceinline__ double row_from_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true);   // row_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_from_prev(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);   // row_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// SHIFT: 0 none (uses own value), 1 wave_shl/shr DPP, 2 row_shl/shr DPP, 3 ds_bpermute
// LDSX: 0 no LDS/barrier, 1 LDS rows + barrier (product), 2 barrier only
// PH: patch height (64: the product's patch; 32 with ROWS = 4: 512-thread workgroups, two of which fit a CU)
template <int SHIFT, int LDSX, int ROWS, int ORDER = 0, int PH = 64>
__global__ void __launch_bounds__(64 * (PH / ROWS)) k_sweeps(const double* __restrict__ in, double* __restrict__ out, int K) {
    constexpr int NS = PH / ROWS;   // strips (waves) per workgroup
    __shared__ double top[2][NS][64], bot[2][NS][64];
    const int lx = threadIdx.x & 63, st = threadIdx.x >> 6;
    double lv[ROWS][6], bv[ROWS], x[ROWS];
    const int64_t base = ((int64_t)blockIdx.x * (64 * NS) + threadIdx.x) * ROWS;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        x[r] = in[(base + r) & 0xfffff];
        bv[r] = 0.25 * x[r];
#pragma unroll
        for (int s = 0; s < 6; ++s) lv[r][s] = -0.11 - 0.001 * (s + r) + 1e-6 * x[r];
    }
    for (int k = 0; k < K; ++k) {
        const int par = k & 1;
        double above = 0.0, below = 0.0;
        if (LDSX == 1) {
            bot[par][st][lx] = x[0];
            top[par][st][lx] = x[ROWS - 1];
        }
        if (LDSX >= 1) __syncthreads();
        if (LDSX == 1) {
            above = (st < NS - 1) ? bot[par][st + 1][lx] : 0.0;
            below = (st > 0) ? top[par][st - 1][lx] : 0.0;
        }
        double e_[ROWS], w_[ROWS], ea_, wb_;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (SHIFT == 1) { e_[r] = dpp_from_next(x[r]); w_[r] = dpp_from_prev(x[r]); }
            else if (SHIFT == 2) { e_[r] = row_from_next(x[r]); w_[r] = row_from_prev(x[r]); }
            else if (SHIFT == 3) { e_[r] = __shfl_down(x[r], 1); w_[r] = __shfl_up(x[r], 1); }
            else { e_[r] = x[r] * 1.0000001; w_[r] = x[r] * 0.9999999; }
        }
        if (SHIFT == 1) { ea_ = dpp_from_next(above); wb_ = dpp_from_prev(below); }
        else if (SHIFT == 2) { ea_ = row_from_next(above); wb_ = row_from_prev(below); }
        else if (SHIFT == 3) { ea_ = __shfl_down(above, 1); wb_ = __shfl_up(below, 1); }
        else { ea_ = above; wb_ = below; }
        double xn[ROWS];
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            // ORDER 1: the strip's inner rows first (they need nothing from LDS), the edge rows last
            const int r = ORDER ? (rr < ROWS - 2 ? rr + 1 : (rr == ROWS - 2 ? 0 : ROWS - 1)) : rr;
            double acc = bv[r];
            acc = fma(-lv[r][0], e_[r], acc);
            acc = fma(-lv[r][1], r < ROWS - 1 ? e_[r + 1 < ROWS ? r + 1 : r] : ea_, acc);
            acc = fma(-lv[r][2], r < ROWS - 1 ? x[r + 1 < ROWS ? r + 1 : r] : above, acc);
            acc = fma(-lv[r][3], w_[r], acc);
            acc = fma(-lv[r][4], r > 0 ? w_[r > 0 ? r - 1 : 0] : wb_, acc);
            acc = fma(-lv[r][5], r > 0 ? x[r > 0 ? r - 1 : 0] : below, acc);
            xn[r] = acc;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) x[r] = xn[r];
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) out[base + r] = x[r];
}

template <int SHIFT, int LDSX, int ROWS, int ORDER = 0, int PH = 64>
static void run(const char* name, const double* in, double* out, int wgs, int K) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f, best0 = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        for (int pass = 0; pass < 2; ++pass) {
            const int kk = pass ? K : 0;
            CHECK(hipEventRecord(a, 0));
            hipLaunchKernelGGL((k_sweeps<SHIFT, LDSX, ROWS, ORDER, PH>), dim3(wgs * (64 / PH)), dim3(64 * (PH / ROWS)), 0, 0, in, out, kk);
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (pass) best = ms < best ? ms : best; else best0 = ms < best0 ? ms : best0;
        }
    }
    const double rounds = wgs / 256.0;
    const double us_per_wg_sweep = (best - best0) * 1e3 / K / rounds;
    printf("%-44s K=%d: %8.3f ms (K=0: %.3f) -> %.3f us per workgroup-sweep (64x64 patch) = %.0f cycles @2.4GHz per 4 rows of one SIMD lane-set\n",
           name, K, best, best0, us_per_wg_sweep, us_per_wg_sweep * 2400.0 / 4.0);
}

// Chebyshev-on-the-mass-matrix sweep (k_strip4_cheb_mass, interior rows): six-neighbour sum, two FMAs, the three-term
// update with the previous iterate.  CHAIN 0: the product's expression order; 1: the sum as two independent halves and
// the update split so that no dependent chain is longer than four operations
template <int CHAIN>
__global__ void __launch_bounds__(1024) k_cheb_sweeps(const double* __restrict__ in, double* __restrict__ out, int K) {
    __shared__ double top[2][16][64], bot[2][16][64];
    const int lx = threadIdx.x & 63, st = threadIdx.x >> 6;
    double bv[4], cw[4], ym[4], yo[4];
    const int64_t base = ((int64_t)blockIdx.x * 1024 + threadIdx.x) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ym[r] = in[(base + r) & 0xfffff]; yo[r] = 0.5 * ym[r]; bv[r] = 0.25 * ym[r]; cw[r] = 0.066 + 1e-6 * ym[r]; }
    const double inv_scale = 0.8;
    for (int k = 0; k < K; ++k) {
        const int par = k & 1;
        bot[par][st][lx] = ym[0];
        top[par][st][lx] = ym[3];
        __syncthreads();
        const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;
        const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;
        double e_[4], w_[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { e_[r] = dpp_from_next(ym[r]); w_[r] = dpp_from_prev(ym[r]); }
        const double ea_ = dpp_from_next(above), wb_ = dpp_from_prev(below);
        const double wk = 1.1 + 1e-3 * k;
        double yn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double n0 = e_[r], n1 = r < 3 ? e_[(r + 1) & 3] : ea_, n2 = r < 3 ? ym[(r + 1) & 3] : above;
            const double n3 = w_[r], n4 = r > 0 ? w_[(r + 3) & 3] : wb_, n5 = r > 0 ? ym[(r + 3) & 3] : below;
            if (CHAIN == 0) {
                const double sum = ((n0 + n1) + (n2 + n3)) + (n4 + n5);
                const double z = fma(-cw[r], sum, fma(-inv_scale, ym[r], bv[r]));
                yn[r] = wk * (z + ym[r] - yo[r]) + yo[r];
            } else {
                const double sa = (n0 + n1) + n2, sb = (n3 + n4) + n5;
                const double t = fma(-inv_scale, ym[r], bv[r]) + (ym[r] - yo[r]);
                const double z = fma(-cw[r], sa + sb, t);
                yn[r] = fma(wk, z, yo[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { yo[r] = ym[r]; ym[r] = yn[r]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[base + r] = ym[r] + yo[r];
}

template <int CHAIN>
static void run_cheb(const char* name, const double* in, double* out, int wgs, int K) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e30f, best0 = 1e30f;
    for (int rep = 0; rep < 4; ++rep)
        for (int pass = 0; pass < 2; ++pass) {
            CHECK(hipEventRecord(a, 0));
            hipLaunchKernelGGL((k_cheb_sweeps<CHAIN>), dim3(wgs), dim3(1024), 0, 0, in, out, pass ? K : 0);
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (pass) best = ms < best ? ms : best; else best0 = ms < best0 ? ms : best0;
        }
    const double us = (best - best0) * 1e3 / K / (wgs / 256.0);
    printf("%-44s K=%d: %8.3f ms -> %.3f us per workgroup-sweep = %.0f cycles @2.4GHz per wave-sweep\n", name, K, best, us, us * 600.0);
}

int main(int argc, char** argv) {
    const int wgs = 2048, K = argc > 1 ? atoi(argv[1]) : 64;
    double *in, *out;
    CHECK(hipMalloc(&in, (1 << 20) * 8));
    CHECK(hipMalloc(&out, (size_t)wgs * 4096 * 8));
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5 + 1e-3 * (i % 977);
    CHECK(hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    run<1, 1, 4>("product: wave DPP + LDS rows + barrier", in, out, wgs, K);
    run<0, 1, 4>("no shifts, LDS rows + barrier", in, out, wgs, K);
    run<1, 0, 4>("wave DPP, no LDS / barrier", in, out, wgs, K);
    run<1, 2, 4>("wave DPP, barrier only", in, out, wgs, K);
    run<0, 0, 4>("FMA only", in, out, wgs, K);
    run<2, 1, 4>("row DPP (16-lane rows) + LDS rows + barrier", in, out, wgs, K);
    run<2, 0, 4>("row DPP, no LDS / barrier", in, out, wgs, K);
    run<3, 1, 4>("ds_bpermute shuffles + LDS rows + barrier", in, out, wgs, K);
    run<1, 1, 4, 1>("product, inner rows first", in, out, wgs, K);
    run<1, 1, 8, 0>("8 rows per thread, 8 waves", in, out, wgs, K);
    run<1, 1, 8, 1>("8 rows per thread, 8 waves, inner rows first", in, out, wgs, K);
    run<1, 0, 8, 0>("8 rows per thread, no LDS / barrier", in, out, wgs, K);
    run<1, 1, 16, 1>("16 rows per thread, 4 waves, inner first", in, out, wgs, K);
    run<1, 1, 4, 0, 32>("64 x 32 patches, 8 waves, 2 workgroups per CU (time per PAIR)", in, out, wgs, K);
    run<1, 1, 4, 1, 32>("64 x 32 patches, 8 waves, inner rows first (per pair)", in, out, wgs, K);
    run_cheb<0>("Chebyshev sweep, product expression order", in, out, wgs, K);
    run_cheb<1>("Chebyshev sweep, short dependent chains", in, out, wgs, K);
    return 0;
}
