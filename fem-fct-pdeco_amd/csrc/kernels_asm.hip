// On-device P1 assembly on the structured mesh (what the reference obtains from
// dolfin.assemble through helpers.py:87-141).  Row-gather formulation: the thread
// that owns row P visits the <= 6 triangles around P and adds their element-matrix
// rows straight into its 7 ELL slots -- no scatter, no atomics, deterministic.
#include "femfct_internal.h"
#include "device_utils.h"
#include "stencil.h"
#include "solidbody_op.h"

namespace {

// ---------------------------------------------------------------------------
// M = u*v*dx, Ad = dot(grad(u),grad(v))*dx, ml = row_lump(M)   (helpers.py:553-555)
// ---------------------------------------------------------------------------
__global__ void k_mesh_constants(int n, int N, int nc, double h, const int32_t* __restrict__ d2v,
                                 double* __restrict__ M, double* __restrict__ Ad, double* __restrict__ ml) {
    RowRange rr = block_rows(n);
    const double area = 0.5 * h * h;
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        NodeXY p = node_xy(i, d2v, N);
        double m[STENCIL_W] = {0, 0, 0, 0, 0, 0, 0}, a[STENCIL_W] = {0, 0, 0, 0, 0, 0, 0};
        for_each_tri(p, nc, [&](const TriInfo& T, int, int) {
            double gpx = tri_gx(T.type, T.pl), gpy = tri_gy(T.type, T.pl);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                m[T.slot[k]] += (area / 12.0) * (k == T.pl ? 2.0 : 1.0);
                a[T.slot[k]] += 0.5 * (gpx * tri_gx(T.type, k) + gpy * tri_gy(T.type, k));  // |K|/h^2 = 1/2
            }
        });
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < STENCIL_W; ++k) {
            M[(int64_t)k * n + i] = m[k];
            Ad[(int64_t)k * n + i] = a[k];
            s += m[k];
        }
        ml[i] = s;
    }
}

// ---------------------------------------------------------------------------
// dot(wind, grad(v))*u*dx with the wind tabulated at the 6 quadrature points of
// every triangle: A[P,j] = sum_q w_q |K| (wind(x_q).grad(lambda_P)) lambda_j(x_q)
// (helpers.py:581,933,1015; advection_solidbody_FCT_PDECO_finaltime.py:122).
// Exact for winds that are polynomials of degree <= 3 (all winds of the reference).
// wind layout: [((cy*nc+cx)*2 + type)*6 + q][2]
// ---------------------------------------------------------------------------
__global__ void k_convection_tab(int n, int N, int nc, double h, const int32_t* __restrict__ d2v,
                                 const double* __restrict__ wind, double scale, double* __restrict__ A) {
    RowRange rr = block_rows(n);
    const double area = 0.5 * h * h;
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        NodeXY p = node_xy(i, d2v, N);
        double acc[STENCIL_W] = {0, 0, 0, 0, 0, 0, 0};
        for_each_tri(p, nc, [&](const TriInfo& T, int cx, int cy) {
            const double* w = wind + ((((int64_t)cy * nc + cx) * 2 + T.type) * 6) * 2;
            double gpx = tri_gx(T.type, T.pl) / h, gpy = tri_gy(T.type, T.pl) / h;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                double wg = quad6_w(q) * area * (w[2 * q] * gpx + w[2 * q + 1] * gpy);
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[T.slot[k]] += wg * quad6_l(q, k);
            }
        });
#pragma unroll
        for (int k = 0; k < STENCIL_W; ++k) A[(int64_t)k * n + i] = scale * acc[k];
    }
}

// the same operator for the wind om * (-y, x) in closed form (solidbody_op.h: sb_rot_row); check != null: compare
// a stored operator with it bit for bit instead (*check |= 1 on the first difference)
__global__ void k_rotation_op(int n, int N, int nc, double h, double a1, double om, const int32_t* __restrict__ d2v,
                              double* __restrict__ A, const double* __restrict__ stored, int* __restrict__ check) {
    RowRange rr = block_rows(n);
    bool differ = false;
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        NodeXY p = node_xy(i, d2v, N);
        double rot[STENCIL_W], rotT[STENCIL_W];
        sb_rot_row<false>(p, nc, h, a1, om, rot, rotT);
#pragma unroll
        for (int k = 0; k < STENCIL_W; ++k) {
            const int64_t idx = (int64_t)k * n + i;
            if (check) differ |= __double_as_longlong(stored[idx]) != __double_as_longlong(rot[k]);
            else A[idx] = rot[k];
        }
    }
    if (check && differ) atomicOr(check, 1);
}

__global__ void k_quad_points(int nc, double a1, double h, double* __restrict__ xq, double* __restrict__ yq) {
    int64_t tri = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t ntri = (int64_t)nc * nc * 2;
    if (tri >= ntri) return;
    int type = (int)(tri & 1);
    int64_t cell = tri >> 1;
    int cx = (int)(cell % nc), cy = (int)(cell / nc);
    for (int q = 0; q < 6; ++q) {
        double x = 0.0, y = 0.0;
        for (int k = 0; k < 3; ++k) {
            x += quad6_l(q, k) * (a1 + (cx + tri_nx(type, k)) * h);
            y += quad6_l(q, k) * (a1 + (cy + tri_ny(type, k)) * h);
        }
        xq[tri * 6 + q] = x;
        yq[tri * 6 + q] = y;
    }
}

// ---------------------------------------------------------------------------
// Solid-body rotation + drift control: the per-step flux matrix handed to the FCT
// step (advection_solidbody_FCT_PDECO_finaltime.py:187-193, 215-218):
//   A_ref = eps*Ad + sigma * (rot_scale*Arot + Adrift1(c) + Adrift2(c))
// sigma = -1 forward (FCT_alg(A_u) == FCT_alg_ref(-A_u)), +1 adjoint.
//   Adrift1[P,j] = (b.grad c_h)|_K M_K[P,j]            dot(drift,grad(c))*u*v*dx
//   Adrift2[P,j] = (b.grad lambda_P)(M_K c_K)_j        dot(drift,grad(v))*c*u*dx
// ---------------------------------------------------------------------------
__global__ void k_ops_solidbody(int n, int N, int nc, double h, const int32_t* __restrict__ d2v,
                                const int32_t* __restrict__ cols, const double* __restrict__ Ad,
                                const double* __restrict__ Arot, VecRef c_ref, int64_t c_bstride, double eps,
                                double sigma, double rot_scale, double bx, double by, double* __restrict__ A_,
                                int levels) {
    // levels = 1: the operator of the current level (c_ref is level-indirected).  levels > 1: every level
    // at once, blockIdx.y = member * levels + k uses control level c_ref.level_off + k (c_ref.level unused).
    const int bz = blockIdx.y / levels, lk = blockIdx.y - bz * levels;
    const double* c = (levels > 1 ? c_ref.base + (int64_t)(c_ref.level_off + lk) * c_ref.stride : vec_ptr(c_ref)) +
                      bz * c_bstride;
    double* A = A_ + (int64_t)blockIdx.y * STENCIL_W * n;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        NodeXY p = node_xy(i, d2v, N);
        double cv[STENCIL_W];
        cv[0] = c[i];
#pragma unroll
        for (int s = 1; s < STENCIL_W; ++s) cv[s] = c[cols[(int64_t)s * n + i]];
        double acc[STENCIL_W], accT[STENCIL_W];
        sb_drift_row<false>(p, nc, h, cv, bx, by, acc, accT);
#pragma unroll
        for (int k = 0; k < STENCIL_W; ++k) {
            int64_t idx = (int64_t)k * n + i;
            A[idx] = eps * Ad[idx] + sigma * (rot_scale * Arot[idx] + acc[k]);
        }
    }
}

// rhs = M (a - b)  ==  assemble((a_h - b_h) * v * dx)   (advection_solidbody_FCT_PDECO_alltime.py:257)
__global__ void k_mass_diff(int n, int W, const int32_t* __restrict__ cols, const double* __restrict__ M,
                            VecRef a_ref, int64_t a_bstride, VecRef b_ref, int64_t b_bstride,
                            double* __restrict__ out_) {
    const int bz = blockIdx.y;
    const double* a = vec_ptr(a_ref) + bz * a_bstride;
    const double* b = vec_ptr(b_ref);           // may be absent: out = M a
    if (b) b += bz * b_bstride;
    double* out = out_ + (int64_t)bz * n;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        double acc = M[i] * (a[i] - (b ? b[i] : 0.0));
        for (int s = 1; s < W; ++s) {
            int64_t idx = (int64_t)s * n + i;
            int j = cols[idx];
            acc += M[idx] * (a[j] - (b ? b[j] : 0.0));
        }
        out[i] = acc;
    }
}

// Descent-direction right-hand side for the drift control, every time level at once:
//   rhs = -(beta * M c + assemble(p_h * dot(drift, grad(u_h)) * v * dx))
// (advection_solidbody_FCT_PDECO_finaltime.py:233-236);  blockIdx.y = time level.
__global__ void k_drift_gradient_rhs(int n, int N, int nc, double h, const int32_t* __restrict__ d2v,
                                     const int32_t* __restrict__ cols, const double* __restrict__ M,
                                     const double* __restrict__ c_, const double* __restrict__ u_,
                                     const double* __restrict__ p_, double beta, double bx, double by,
                                     double* __restrict__ out_) {
    const int64_t voff = (int64_t)blockIdx.y * n;
    const double *c = c_ + voff, *u = u_ + voff, *pv = p_ + voff;
    double* out = out_ + voff;
    RowRange rr = block_rows(n);
    const double m12 = 0.5 * h * h / 12.0;
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        NodeXY p = node_xy(i, d2v, N);
        double uv[STENCIL_W], pp[STENCIL_W];
        uv[0] = u[i]; pp[0] = pv[i];
        double mc = M[i] * c[i];
#pragma unroll
        for (int s = 1; s < STENCIL_W; ++s) {
            int64_t idx = (int64_t)s * n + i;
            int j = cols[idx];
            uv[s] = u[j];
            pp[s] = pv[j];
            mc += M[idx] * c[j];
        }
        double g = 0.0;
        for_each_tri(p, nc, [&](const TriInfo& T, int, int) {
            double u0 = uv[T.slot[0]], u1 = uv[T.slot[1]], u2 = uv[T.slot[2]];
            double gux = (u0 * tri_gx(T.type, 0) + u1 * tri_gx(T.type, 1) + u2 * tri_gx(T.type, 2)) / h;
            double guy = (u0 * tri_gy(T.type, 0) + u1 * tri_gy(T.type, 1) + u2 * tri_gy(T.type, 2)) / h;
            double psum = pp[T.slot[0]] + pp[T.slot[1]] + pp[T.slot[2]];
            g += (bx * gux + by * guy) * m12 * (pp[0] + psum);   // (M_K p_K)_P, P is slot 0
        });
        out[i] = -(beta * mc + g);
    }
}

// out = a - b elementwise (terminal condition p(T) = uhat_T - u(T), helpers.py:1020)
__global__ void k_axpby(int64_t count, double alpha, const double* __restrict__ a, double beta,
                        const double* __restrict__ b, double* __restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < count; k += stride) out[k] = alpha * a[k] + (b ? beta * b[k] : 0.0);
}

}  // namespace

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
int femfct_enqueue_mesh_constants(femfct_ctx* ctx) {
    LaunchGeom g = femfct_geom(ctx, 1);
    hipLaunchKernelGGL(k_mesh_constants, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->N, ctx->n_cells, ctx->h,
                       ctx->d_d2v, ctx->d_M, ctx->d_Ad, ctx->d_ml);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return femfct_fail(ctx, FEMFCT_ERR_HIP, "k_mesh_constants: %s", hipGetErrorString(e));
    return FEMFCT_OK;
}

int femfct_enqueue_ops_solidbody(femfct_ctx* ctx, const double* Arot, VecRef c_ref, int64_t c_bstride, double eps,
                                 double sigma, double rot_scale, double bx, double by, double* A, int32_t batch,
                                 int32_t levels) {
    LaunchGeom g = femfct_geom(ctx, batch * levels);
    femfct_prof_begin(ctx, KC_ASSEMBLE);
    hipLaunchKernelGGL(k_ops_solidbody, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->N, ctx->n_cells, ctx->h,
                       ctx->d_d2v, ctx->d_cols, ctx->d_Ad, Arot, c_ref, c_bstride, eps, sigma, rot_scale, bx, by, A,
                       levels);
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

int femfct_enqueue_mass_diff(femfct_ctx* ctx, VecRef a, int64_t a_bstride, VecRef b, int64_t b_bstride, double* out,
                             int32_t batch) {
    LaunchGeom g = femfct_geom(ctx, batch);
    hipLaunchKernelGGL(k_mass_diff, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->W, ctx->d_cols, ctx->d_M, a,
                       a_bstride, b, b_bstride, out);
    return FEMFCT_OK;
}

int femfct_enqueue_axpby(femfct_ctx* ctx, int64_t count, double alpha, const double* a, double beta, const double* b,
                         double* out) {
    int bs = 256;
    int64_t g = (count + bs - 1) / bs;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_axpby, dim3((unsigned)g), dim3(bs), 0, ctx->stream, count, alpha, a, beta, b, out);
    return FEMFCT_OK;
}

extern "C" {

int femfct_mesh_quad_points(femfct_ctx* ctx, double* xq_host, double* yq_host) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->structured, "structured mesh not set");
    ARG_TRY(ctx, xq_host && yq_host, "null argument");
    int64_t ntri = (int64_t)ctx->n_cells * ctx->n_cells * 2;
    double *dx = nullptr, *dy = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&dx, sizeof(double) * ntri * 6));
    HIP_TRY(ctx, hipMalloc((void**)&dy, sizeof(double) * ntri * 6));
    hipLaunchKernelGGL(k_quad_points, dim3((unsigned)((ntri + 255) / 256)), dim3(256), 0, ctx->stream, ctx->n_cells,
                       ctx->a1, ctx->h, dx, dy);
    HIP_TRY(ctx, hipMemcpyAsync(xq_host, dx, sizeof(double) * ntri * 6, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(yq_host, dy, sizeof(double) * ntri * 6, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(dx);
    hipFree(dy);
    return FEMFCT_OK;
}

int femfct_assemble_convection(femfct_ctx* ctx, const double* wind_q_host, double scale, double* A_ell_dev) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->structured, "structured mesh not set");
    ARG_TRY(ctx, wind_q_host && A_ell_dev, "null argument");
    int64_t cnt = (int64_t)ctx->n_cells * ctx->n_cells * 2 * 6 * 2;
    double* dw = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&dw, sizeof(double) * cnt));
    HIP_TRY(ctx, hipMemcpyAsync(dw, wind_q_host, sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream));
    LaunchGeom g = femfct_geom(ctx, 1);
    hipLaunchKernelGGL(k_convection_tab, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->N, ctx->n_cells, ctx->h,
                       ctx->d_d2v, dw, scale, A_ell_dev);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(dw);
    return FEMFCT_OK;
}

int femfct_assemble_rotation(femfct_ctx* ctx, double om, double* A_ell_dev) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->structured, "structured mesh not set");
    ARG_TRY(ctx, A_ell_dev, "null argument");
    LaunchGeom g = femfct_geom(ctx, 1);
    hipLaunchKernelGGL(k_rotation_op, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->N, ctx->n_cells, ctx->h, ctx->a1, om,
                       ctx->d_d2v, A_ell_dev, (const double*)nullptr, (int*)nullptr);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->rot_om = om;          // (a hint for femfct_rotation_is_geometric: the array itself is checked at use)
    ctx->rot_om_set = true;
    return FEMFCT_OK;
}

int femfct_drift_gradient_rhs(femfct_ctx* ctx, const double* c_dev, const double* u_dev, const double* p_dev,
                              double beta, double bx, double by, double* out_dev, int32_t levels) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->structured, "structured mesh not set");
    ARG_TRY(ctx, c_dev && u_dev && p_dev && out_dev && levels >= 1, "bad argument");
    LaunchGeom g = femfct_geom(ctx, levels);
    hipLaunchKernelGGL(k_drift_gradient_rhs, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->N, ctx->n_cells, ctx->h,
                       ctx->d_d2v, ctx->d_cols, ctx->d_M, c_dev, u_dev, p_dev, beta, bx, by, out_dev);
    return FEMFCT_OK;
}

}  // extern "C"

// Is the stored operator Arot, bit for bit, what femfct_assemble_rotation(om) writes for the angular velocity this
// context last assembled?  Then the bandwidth-regime step kernels derive its rows instead of loading them.  One pass
// over the array + one 4-byte read-back per trajectory sweep (the sweep itself is >= hundreds of such passes).
bool femfct_rotation_is_geometric(femfct_ctx* ctx, const double* Arot, double* om_out) {
    if (!ctx->geom_rot || !ctx->rot_om_set || !ctx->structured || !Arot) return false;
    if (!ctx->d_rot_check && hipMalloc((void**)&ctx->d_rot_check, sizeof(int)) != hipSuccess) return false;
    if (hipMemsetAsync(ctx->d_rot_check, 0, sizeof(int), ctx->stream) != hipSuccess) return false;
    LaunchGeom g = femfct_geom(ctx, 1);
    hipLaunchKernelGGL(k_rotation_op, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->N, ctx->n_cells, ctx->h, ctx->a1,
                       ctx->rot_om, ctx->d_d2v, (double*)nullptr, Arot, ctx->d_rot_check);
    int differ = 1;
    if (hipMemcpyAsync(&differ, ctx->d_rot_check, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    if (differ) return false;
    *om_out = ctx->rot_om;
    return true;
}
