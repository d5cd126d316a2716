// Optimisation-layer reductions and updates kept on the device so that a projected
// gradient iteration moves only scalars across PCIe:
//   L2_norm_sq_Q / L2_norm_sq_Omega   /root/reference/helpers.py:330-381
//   cost_functional                   /root/reference/helpers.py:383-441
//   update_control (clip)             /root/reference/helpers.py:1666-1667
//
// Quadratic forms phi^T M phi are summed with a fixed reduction tree (per-thread row
// sums -> wave64 butterflies -> LDS -> per-block partials -> one block), no atomics:
// bitwise reproducible.
#include "femfct_internal.h"
#include "device_utils.h"

namespace {

// partial[level*G + block] = sum_{i in block rows} phi_i (M phi)_i,  phi = a - b
__global__ void k_quadform(int n, int W, const int32_t* __restrict__ cols, const double* __restrict__ M,
                           const double* __restrict__ a_, const double* __restrict__ b_, int64_t a_lstride,
                           int64_t b_lstride, double* __restrict__ partial) {
    __shared__ double smem[32];
    const int lvl = blockIdx.y;
    const double* a = a_ + (int64_t)lvl * a_lstride;
    const double* b = b_ ? b_ + (int64_t)lvl * b_lstride : nullptr;
    RowRange rr = block_rows(n);
    double s = 0.0;
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        double pi = a[i] - (b ? b[i] : 0.0);
        double acc = M[i] * pi;
        for (int k = 1; k < W; ++k) {
            int64_t idx = (int64_t)k * n + i;
            int j = cols[idx];
            acc += M[idx] * (a[j] - (b ? b[j] : 0.0));
        }
        s += pi * acc;
    }
    s = block_reduce(s, OpSum(), 0.0, smem);
    if (threadIdx.x == 0) partial[(int64_t)lvl * gridDim.x + blockIdx.x] = s;
}

// out[b] (+)= scale * sum_l w_l * sum_blocks partial[(b*levels + l)*G + block]
// w_l = 1 except 1/2 at the first and last level when trapezoid != 0.
__global__ void k_reduce_levels(int levels, int G, const double* __restrict__ partial, int trapezoid, double scale,
                                int accumulate, double* __restrict__ out) {
    __shared__ double smem[32];
    const int b = blockIdx.x;
    const double* p = partial + (int64_t)b * levels * G;
    double s = 0.0;
    for (int64_t k = threadIdx.x; k < (int64_t)levels * G; k += blockDim.x) {
        int l = (int)(k / G);
        double w = (trapezoid && (l == 0 || l == levels - 1)) ? 0.5 : 1.0;
        s += w * p[k];
    }
    s = block_reduce(s, OpSum(), 0.0, smem);
    if (threadIdx.x == 0) out[b] = (accumulate ? out[b] : 0.0) + scale * s;
}

__global__ void k_clip_axpy(int64_t count, const double* __restrict__ c, double s, const double* __restrict__ d,
                            double lo, double hi, double* __restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < count; k += stride) out[k] = fmin(fmax(c[k] + s * d[k], lo), hi);
}

// d = -(beta*c - t),  t = x*y/divisor (y given) or scale*x: the pointwise gradient expressions of the
// refactored drivers, evaluated in the reference's operation order (no contraction: -ffp-contract=off)
//   nonlinear_FCT_PDECO_refactored.py:148   dk = -(beta*ck - pk)
//   Schnak_FCT_PDECO_refactored.py:167      dk = -(beta*ck - gamma/rescaling*pk)
//   chemotaxis_FCT_PDECO_AT_refactored.py:158   dk = -(beta*ck - qk*uk/rescaling)
__global__ void k_descent_pointwise(int64_t count, double beta, const double* __restrict__ c, double scale,
                                    const double* __restrict__ x, const double* __restrict__ y, double divisor,
                                    double* __restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < count; k += stride) {
        const double t = y ? (x[k] * y[k]) / divisor : scale * x[k];
        out[k] = -(beta * c[k] - t);
    }
}

int ensure_scratch(femfct_ctx* ctx, size_t doubles) {
    if (doubles <= ctx->scratch_count) return FEMFCT_OK;
    if (ctx->d_scratch) hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr;
    ctx->scratch_count = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_scratch, sizeof(double) * doubles));
    ctx->scratch_count = doubles;
    return FEMFCT_OK;
}

// enqueue: out_dev[b] (+)= scale * sum_levels w phi^T M phi   (phi = a - b)
int enqueue_norm(femfct_ctx* ctx, const double* a, const double* b, int64_t a_lstride, int64_t b_lstride, int levels,
                 int trapezoid, double scale, int accumulate, int32_t batch, double* out_dev, size_t scratch_off) {
    LaunchGeom g = femfct_geom(ctx, 1);
    const int G = g.grid.x;
    g.grid.y = (unsigned)(levels * batch);
    double* part = ctx->d_scratch + scratch_off;
    hipLaunchKernelGGL(k_quadform, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->W, ctx->d_cols, ctx->d_M, a, b,
                       a_lstride, b_lstride, part);
    hipLaunchKernelGGL(k_reduce_levels, dim3(batch), dim3(256), 0, ctx->stream, levels, G, part, trapezoid, scale,
                       accumulate, out_dev);
    return FEMFCT_OK;
}

}  // namespace

extern "C" {

int femfct_l2_norm_sq_Q(femfct_ctx* ctx, const double* a_dev, const double* b_dev, int32_t num_steps, double dt,
                        double* out_host, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0 && ctx->have_mass, "mass matrix not set");
    ARG_TRY(ctx, a_dev && out_host && num_steps >= 0 && batch >= 1, "bad argument");
    const int levels = num_steps + 1;
    ARG_TRY(ctx, (int64_t)levels * batch <= 65535, "too many levels*batch for one launch");
    LaunchGeom g = femfct_geom(ctx, 1);
    int rc = ensure_scratch(ctx, (size_t)levels * batch * g.grid.x + batch);
    if (rc != FEMFCT_OK) return rc;
    double* out_dev = ctx->d_scratch + (size_t)levels * batch * g.grid.x;
    // a batch member's levels are contiguous, so (batch, level) flattens to one level index
    enqueue_norm(ctx, a_dev, b_dev, ctx->n, ctx->n, levels, 1, dt, 0, batch, out_dev, 0);
    HIP_TRY(ctx, hipMemcpyAsync(out_host, out_dev, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FEMFCT_OK;
}

int femfct_l2_norm_sq_Omega(femfct_ctx* ctx, const double* a_dev, const double* b_dev, double* out_host,
                            int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0 && ctx->have_mass, "mass matrix not set");
    ARG_TRY(ctx, a_dev && out_host && batch >= 1 && batch <= 65535, "bad argument");
    LaunchGeom g = femfct_geom(ctx, 1);
    int rc = ensure_scratch(ctx, (size_t)batch * g.grid.x + batch);
    if (rc != FEMFCT_OK) return rc;
    double* out_dev = ctx->d_scratch + (size_t)batch * g.grid.x;
    enqueue_norm(ctx, a_dev, b_dev, ctx->n, ctx->n, 1, 0, 1.0, 0, batch, out_dev, 0);
    HIP_TRY(ctx, hipMemcpyAsync(out_host, out_dev, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FEMFCT_OK;
}

int femfct_cost_functional(femfct_ctx* ctx, const double* var1, const double* var1_target, const double* control,
                           int32_t control_shared, int32_t num_steps, double dt, double beta, int32_t finaltime,
                           const double* var2, const double* var2_target, double* J_host, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0 && ctx->have_mass, "mass matrix not set");
    ARG_TRY(ctx, var1 && var1_target && control && J_host && num_steps >= 1 && batch >= 1, "bad argument");
    ARG_TRY(ctx, (var2 == nullptr) == (var2_target == nullptr), "var2 and var2_target must be given together");
    const int levels = num_steps + 1;
    ARG_TRY(ctx, (int64_t)levels * batch <= 65535, "too many levels*batch for one launch");
    LaunchGeom g = femfct_geom(ctx, 1);
    const size_t psz = (size_t)levels * batch * g.grid.x;
    int rc = ensure_scratch(ctx, psz + batch);
    if (rc != FEMFCT_OK) return rc;
    double* J = ctx->d_scratch + psz;
    const int64_t n = ctx->n, tstride = (int64_t)levels * n;
    if (!finaltime) {
        // 0.5 * ||var - target||^2_{L2(Q)}  (helpers.py:422-426)
        enqueue_norm(ctx, var1, var1_target, n, n, levels, 1, 0.5 * dt, 0, batch, J, 0);
        if (var2) enqueue_norm(ctx, var2, var2_target, n, n, levels, 1, 0.5 * dt, 1, batch, J, 0);
    } else {
        // 0.5 * ||var(T) - target||^2_{L2(Omega)}  (helpers.py:428-434): target is n doubles per batch member
        enqueue_norm(ctx, var1 + (int64_t)num_steps * n, var1_target, tstride, n, 1, 0, 0.5, 0, batch, J, 0);
        if (var2) enqueue_norm(ctx, var2 + (int64_t)num_steps * n, var2_target, tstride, n, 1, 0, 0.5, 1, batch, J, 0);
    }
    // + beta/2 ||c||^2_{L2(Q)}  (helpers.py:440)
    if (control_shared) {
        // one control for the whole batch: norm once into a spare slot, then add to every member
        ARG_TRY(ctx, batch == 1, "control_shared requires batch == 1 in femfct_cost_functional");
    }
    enqueue_norm(ctx, control, nullptr, n, n, levels, 1, 0.5 * beta * dt, 1, batch, J, 0);
    HIP_TRY(ctx, hipMemcpyAsync(J_host, J, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FEMFCT_OK;
}

int femfct_descent_pointwise(femfct_ctx* ctx, int64_t count, double beta, const double* c_dev, double scale,
                             const double* x_dev, const double* y_dev, double divisor, double* out_dev) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && c_dev && x_dev && out_dev && count >= 0 && divisor != 0.0, "bad argument");
    int bs = 256;
    int64_t g = (count + bs - 1) / bs;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_descent_pointwise, dim3((unsigned)g), dim3(bs), 0, ctx->stream, count, beta, c_dev, scale, x_dev,
                       y_dev, divisor, out_dev);
    return FEMFCT_OK;
}

int femfct_project_control(femfct_ctx* ctx, const double* c_dev, double s, const double* d_dev, double c_lower,
                           double c_upper, double* out_dev, int64_t count) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && c_dev && d_dev && out_dev && count >= 0, "bad argument");
    int bs = 256;
    int64_t g = (count + bs - 1) / bs;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_clip_axpy, dim3((unsigned)g), dim3(bs), 0, ctx->stream, count, c_dev, s, d_dev, c_lower,
                       c_upper, out_dev);
    return FEMFCT_OK;
}

}  // extern "C"
