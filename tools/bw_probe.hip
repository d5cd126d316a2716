// Bandwidth probe (tuning aid, not part of the library): how fast can gfx950 stream
// 8 B/lane vs 16 B/lane, 1 stream vs 8 concurrent streams (the ELL slot-major access shape)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void copy8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
__global__ void copy16(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
// 8 read streams (slot-major ELL shape) + 1 write, 8 B per lane
__global__ void streams8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a[k * n + i];
        b[i] = s;
    }
}
__global__ void streams16(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += st) {
        double2 s = make_double2(0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) { double2 v = a[k * n2 + i]; s.x += v.x; s.y += v.y; }
        b[i] = s;
    }
}
// contiguous chunk per block (like block_rows) instead of grid-stride
__global__ void streams8_chunk(const double* __restrict__ a, double* __restrict__ b, size_t n) {
    size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    size_t beg = blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a[k * n + i];
        b[i] = s;
    }
}
// SpMV shape: 7 value streams + 3 vector streams + 6 gathers from x (vertex-order stencil), chunked
__global__ void spmv_like(const double* __restrict__ a, const double* __restrict__ x, const double* __restrict__ bb,
                          const double* __restrict__ yo, double* __restrict__ y, size_t n, int N) {
    size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + 63) & ~(size_t)63;
    size_t beg = blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    const long off[7] = {0, 1, (long)N + 1, (long)N, -1, -(long)N - 1, -(long)N};
    for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            long j = (long)i + off[k];
            j = (j < 0 || j >= (long)n) ? (long)i : j;
            acc += a[k * n + i] * x[j];
        }
        double r = bb[i] - acc;
        y[i] = 1.1 * (r * 0.8 + x[i] - yo[i]) + yo[i];
    }
}
__device__ __forceinline__ int xcd_remap(int b, int G) {
    int q = G >> 3, r = G & 7; int xcd = b & 7, idx = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}
// spmv_like with the XCD-contiguous chunk mapping used by the library
__global__ void spmv_like_xcd(const double* __restrict__ a, const double* __restrict__ x, const double* __restrict__ bb,
                          const double* __restrict__ yo, double* __restrict__ y, size_t n, int N) {
    size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + 63) & ~(size_t)63;
    size_t beg = (size_t)xcd_remap(blockIdx.x, gridDim.x) * chunk, end = beg + chunk < n ? beg + chunk : n;
    if (beg > n) beg = n;
    const long off[7] = {0, 1, (long)N + 1, (long)N, -1, -(long)N - 1, -(long)N};
    for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            long j = (long)i + off[k];
            j = (j < 0 || j >= (long)n) ? (long)i : j;
            acc += a[k * n + i] * x[j];
        }
        double r = bb[i] - acc;
        y[i] = 1.1 * (r * 0.8 + x[i] - yo[i]) + yo[i];
    }
}
// same, two rows per thread (independent chains, more loads in flight)
__global__ void spmv_like2(const double* __restrict__ a, const double* __restrict__ x, const double* __restrict__ bb,
                           const double* __restrict__ yo, double* __restrict__ y, size_t n, int N) {
    size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + 127) & ~(size_t)127;
    size_t beg = blockIdx.x * chunk, end = beg + chunk < n ? beg + chunk : n;
    const long off[7] = {0, 1, (long)N + 1, (long)N, -1, -(long)N - 1, -(long)N};
    for (size_t i0 = beg + threadIdx.x; i0 < end; i0 += 2 * blockDim.x) {
        size_t i1 = i0 + blockDim.x;
        bool v1 = i1 < end;
        double acc0 = 0, acc1 = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            long j0 = (long)i0 + off[k]; j0 = (j0 < 0 || j0 >= (long)n) ? (long)i0 : j0;
            acc0 += a[k * n + i0] * x[j0];
            if (v1) { long j1 = (long)i1 + off[k]; j1 = (j1 < 0 || j1 >= (long)n) ? (long)i1 : j1; acc1 += a[k * n + i1] * x[j1]; }
        }
        y[i0] = 1.1 * ((bb[i0] - acc0) * 0.8 + x[i0] - yo[i0]) + yo[i0];
        if (v1) y[i1] = 1.1 * ((bb[i1] - acc1) * 0.8 + x[i1] - yo[i1]) + yo[i1];
    }
}
template <class F> double timeit(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}
int main() {
    size_t n = 4198400;  // ~ 2049^2, even
    double *a, *b; CK(hipMalloc(&a, 8 * n * 8)); CK(hipMalloc(&b, 8 * n * 8));
    CK(hipMemset(a, 0, 8 * n * 8)); CK(hipMemset(b, 0, 8 * n * 8));
    {
        size_t nn = 4198401; int N = 2049;
        double *A, *x, *bb, *yo, *y;
        CK(hipMalloc(&A, 7 * nn * 8)); CK(hipMalloc(&x, nn * 8)); CK(hipMalloc(&bb, nn * 8)); CK(hipMalloc(&yo, nn * 8)); CK(hipMalloc(&y, nn * 8));
        CK(hipMemset(A, 0, 7 * nn * 8)); CK(hipMemset(x, 0, nn * 8)); CK(hipMemset(bb, 0, nn * 8)); CK(hipMemset(yo, 0, nn * 8));
        for (int g : {2048, 8192}) for (int bs : {256}) {
            double t = timeit([&] { hipLaunchKernelGGL(spmv_like, dim3(g), dim3(bs), 0, 0, A, x, bb, yo, y, nn, N); }, 20);
            printf("grid %5d bs %3d spmv_like   %7.1f GB/s (88 B/row) %.1f us\n", g, bs, 88.0 * nn / t / 1e6, t * 1e3);
            t = timeit([&] { hipLaunchKernelGGL(spmv_like_xcd, dim3(g), dim3(bs), 0, 0, A, x, bb, yo, y, nn, N); }, 20);
            printf("grid %5d bs %3d spmv_likeXCD %7.1f GB/s (88 B/row) %.1f us\n", g, bs, 88.0 * nn / t / 1e6, t * 1e3);
            t = timeit([&] { hipLaunchKernelGGL(spmv_like2, dim3(g), dim3(bs), 0, 0, A, x, bb, yo, y, nn, N); }, 20);
            printf("grid %5d bs %3d spmv_like2  %7.1f GB/s (88 B/row) %.1f us\n", g, bs, 88.0 * nn / t / 1e6, t * 1e3);
        }
    }
    for (int g : {2048, 8192}) {
        double t;
        t = timeit([&] { hipLaunchKernelGGL(copy8, dim3(g), dim3(256), 0, 0, a, b, 8 * n); }, 20);
        printf("grid %5d copy8      %7.1f GB/s\n", g, 2.0 * 8 * n * 8 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(copy16, dim3(g), dim3(256), 0, 0, (double2*)a, (double2*)b, 4 * n); }, 20);
        printf("grid %5d copy16     %7.1f GB/s\n", g, 2.0 * 8 * n * 8 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(streams8, dim3(g), dim3(256), 0, 0, a, b, n); }, 20);
        printf("grid %5d streams8   %7.1f GB/s\n", g, 9.0 * n * 8 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(streams16, dim3(g), dim3(256), 0, 0, (double2*)a, (double2*)b, n / 2); }, 20);
        printf("grid %5d streams16  %7.1f GB/s\n", g, 9.0 * n * 8 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(streams8_chunk, dim3(g), dim3(256), 0, 0, a, b, n); }, 20);
        printf("grid %5d streams8ch %7.1f GB/s\n", g, 9.0 * n * 8 / t / 1e6);
    }
    return 0;
}
