// Semantics check of the wave-wide DPP shifts used by the strip kernels (tuning helper).
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ double dpp_from_next(double v) {   // lane i <- lane i+1 (lane 63 <- 0.0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_prev(double v) {   // lane i <- lane i-1 (lane 0 <- 0.0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
    double v = 100.0 + threadIdx.x;
    out[threadIdx.x] = dpp_from_next(v);
    out[64 + threadIdx.x] = dpp_from_prev(v);
}
int main() {
    double* d; hipMalloc(&d, 128 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        double en = i < 63 ? 101.0 + i : 0.0, ep = i > 0 ? 99.0 + i : 0.0;
        if (h[i] != en || h[64 + i] != ep) { ++bad; printf("lane %d: next %g (want %g) prev %g (want %g)\n", i, h[i], en, h[64 + i], ep); }
    }
    printf(bad ? "DPP MISMATCH\n" : "DPP OK: wave_shl:1 = from lane+1, wave_shr:1 = from lane-1, out-of-range lanes read 0\n");
    return bad != 0;
}
