// Per-step FEM forms with P1 coefficient functions, assembled on the device by quadrature
// (the dolfin.assemble calls inside the reference's time loops):
//   M_u2 = u_h^2*u*v*dx, M_uv = u_h*v_h*u*v*dx          helpers.py:591,683,692,953,1032
//   load vectors  (g/r*c + g*u^2*v)*v*dx, g*p*u^2*w*dx, -2g*u*v*q*w*dx, c*v*dx,
//                 v_n*v*dx + dt*c*u/r*v*dx, c*q/r*w*dx   helpers.py:584-585,594,684,693,956,1339-1340,1505
//   chemotaxis exp-forms                                 helpers.py:1350-1351,1499-1500,1531-1532
// Row-gather: the thread owning row P visits the <= 6 triangles around P (stencil.h); field values
// at triangle nodes are gathered through the ELL column table, so either DoF ordering works.
// Polynomial integrands have degree <= 4: the 6-point degree-4 rule integrates them exactly, as
// FEniCS does.  No atomics; deterministic.
#include "femfct_internal.h"
#include "device_utils.h"
#include "stencil.h"
#include "forms.h"

#include <math.h>
#include <algorithm>

namespace {

struct NodeXY { int ix, iy; };

__device__ __forceinline__ NodeXY node_xy(int i, const int32_t* __restrict__ d2v, int N) {
    int v = d2v ? d2v[i] : i;
    return NodeXY{v % N, v / N};
}

template <class F>
__device__ __forceinline__ void for_each_tri(NodeXY p, int nc, F&& f) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const TriInfo T = tri_info(t);
        int cx = p.ix + T.cdx, cy = p.iy + T.cdy;
        if (cx < 0 || cy < 0 || cx >= nc || cy >= nc) continue;
        f(T);
    }
}

// gather the 7 stencil values of a P1 field around row i
__device__ __forceinline__ void gather7(const double* __restrict__ f, const int32_t* __restrict__ cols, int n, int i,
                                        double (&v)[STENCIL_W]) {
    v[0] = f[i];
#pragma unroll
    for (int s = 1; s < STENCIL_W; ++s) v[s] = f[cols[(int64_t)s * n + i]];
}

__device__ __forceinline__ const double* bptr(const VecRef& r, int64_t bstride, int bz) {
    const double* p = vec_ptr(r);
    return p ? p + bz * bstride : nullptr;
}

// ---------------------------------------------------------------------------
// out = alpha*M + gamma*Base + beta * int f1_h f2_h phi_i phi_j
// ---------------------------------------------------------------------------
__device__ __forceinline__ void form_weighted_mass(const MeshArgs& m, const WMassSpec& sp, double* __restrict__ out_, int bz) {
    const int n = m.n;
    const double* f1 = bptr(sp.f1, sp.f1_bs, bz);
    const double* f2 = bptr(sp.f2, sp.f2_bs, bz);
    double* out = out_ + (int64_t)bz * STENCIL_W * n;
    const double area = 0.5 * m.h * m.h;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        double acc[STENCIL_W] = {0, 0, 0, 0, 0, 0, 0};
        if (sp.beta != 0.0) {
            NodeXY p = node_xy(i, m.d2v, m.N);
            double a[STENCIL_W], b[STENCIL_W];
            gather7(f1, m.cols, n, i, a);
            gather7(f2, m.cols, n, i, b);
            for_each_tri(p, m.nc, [&](const TriInfo& T) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    double l0 = quad6_l(q, 0), l1 = quad6_l(q, 1), l2 = quad6_l(q, 2);
                    double fa = l0 * a[T.slot[0]] + l1 * a[T.slot[1]] + l2 * a[T.slot[2]];
                    double fb = l0 * b[T.slot[0]] + l1 * b[T.slot[1]] + l2 * b[T.slot[2]];
                    double w = quad6_w(q) * area * fa * fb * quad6_l(q, T.pl);
                    acc[T.slot[0]] += w * l0;
                    acc[T.slot[1]] += w * l1;
                    acc[T.slot[2]] += w * l2;
                }
            });
        }
#pragma unroll
        for (int k = 0; k < STENCIL_W; ++k) {
            int64_t idx = (int64_t)k * n + i;
            double v = sp.beta * acc[k];
            if (sp.alpha != 0.0) v += sp.alpha * m.M[idx];
            if (sp.base) v += sp.gamma * sp.base[idx];
            out[idx] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// out_i = s0*(M x)_i + s1 * int (k0 + k1*p1 + k2*q1*q2*q3) phi_i + s2*(da_i - db_i) + s3*(M (ea - eb))_i
// (q3 may be absent = 1).  The raw nodal difference reproduces helpers.py:1506-1507,1533-1534; the
// mass-weighted one is assemble((ea_h - eb_h)*w*dx) (Schnak_FCT_PDECO_alltime.py:268,278).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void form_load(const MeshArgs& m, const LoadSpec& sp, double* __restrict__ out_, int bz) {
    const int n = m.n;
    const double* mx = bptr(sp.mx, sp.mx_bs, bz);
    const double* p1 = bptr(sp.p1, sp.p1_bs, bz);
    const double* q1 = bptr(sp.q1, sp.q1_bs, bz);
    const double* q2 = bptr(sp.q2, sp.q2_bs, bz);
    const double* q3 = bptr(sp.q3, sp.q3_bs, bz);
    const double* da = bptr(sp.da, sp.da_bs, bz);
    const double* db = bptr(sp.db, sp.db_bs, bz);
    const double* ea = bptr(sp.ea, sp.ea_bs, bz);
    const double* eb = bptr(sp.eb, sp.eb_bs, bz);
    double* out = out_ + (int64_t)bz * n;
    const double area = 0.5 * m.h * m.h;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        double res = 0.0;
        if (mx && sp.s0 != 0.0) {
            double acc = m.M[i] * mx[i];
#pragma unroll
            for (int s = 1; s < STENCIL_W; ++s) {
                int64_t idx = (int64_t)s * n + i;
                acc += m.M[idx] * mx[m.cols[idx]];
            }
            res += sp.s0 * acc;
        }
        if (sp.s1 != 0.0) {
            NodeXY p = node_xy(i, m.d2v, m.N);
            double a[STENCIL_W], b1[STENCIL_W], b2[STENCIL_W], b3[STENCIL_W];
            if (p1) gather7(p1, m.cols, n, i, a);
            if (q1) gather7(q1, m.cols, n, i, b1);
            if (q2) gather7(q2, m.cols, n, i, b2);
            if (q3) gather7(q3, m.cols, n, i, b3);
            double ld = 0.0;
            for_each_tri(p, m.nc, [&](const TriInfo& T) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    double l0 = quad6_l(q, 0), l1 = quad6_l(q, 1), l2 = quad6_l(q, 2);
                    double f = sp.k0;
                    if (p1) f += sp.k1 * (l0 * a[T.slot[0]] + l1 * a[T.slot[1]] + l2 * a[T.slot[2]]);
                    if (q1) {
                        double g = sp.k2 * (l0 * b1[T.slot[0]] + l1 * b1[T.slot[1]] + l2 * b1[T.slot[2]]);
                        if (q2) g *= (l0 * b2[T.slot[0]] + l1 * b2[T.slot[1]] + l2 * b2[T.slot[2]]);
                        if (q3) g *= (l0 * b3[T.slot[0]] + l1 * b3[T.slot[1]] + l2 * b3[T.slot[2]]);
                        f += g;
                    }
                    ld += quad6_w(q) * area * f * quad6_l(q, T.pl);
                }
            });
            res += sp.s1 * ld;
        }
        if (da && sp.s2 != 0.0) res += sp.s2 * (da[i] - (db ? db[i] : 0.0));
        if (ea && sp.s3 != 0.0) {
            double acc = m.M[i] * (ea[i] - (eb ? eb[i] : 0.0));
#pragma unroll
            for (int s = 1; s < STENCIL_W; ++s) {
                int64_t idx = (int64_t)s * n + i;
                int j = m.cols[idx];
                acc += m.M[idx] * (ea[j] - (eb ? eb[j] : 0.0));
            }
            res += sp.s3 * acc;
        }
        out[i] = res;
    }
}

// ---------------------------------------------------------------------------
// chemotaxis forward flux matrix (helpers.py:1350-1352):
//   A = Dm*Ad - chi * int exp(-eta u_h) (grad v_h . grad phi_i) phi_j      (6-point rule)
// chemotaxis adjoint flux matrix (helpers.py:1499-1503), adjoint != 0:
//   A = Dm*Ad - chi * int (1 - eta u_h) exp(-eta u_h) (grad phi_j . grad v_h) phi_i   (7-point rule)
// ---------------------------------------------------------------------------
template <int ADJ>
__device__ __forceinline__ void form_chtxs_matrix(const MeshArgs& m, VecRef u_ref, int64_t u_bs, VecRef v_ref, int64_t v_bs, double Dm,
                                                  double chi, double eta, double* __restrict__ out_, int bz) {
    const int n = m.n;
    const double* u = bptr(u_ref, u_bs, bz);
    const double* v = bptr(v_ref, v_bs, bz);
    double* out = out_ + (int64_t)bz * STENCIL_W * n;
    const double area = 0.5 * m.h * m.h, ih = 1.0 / m.h;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        NodeXY p = node_xy(i, m.d2v, m.N);
        double uu[STENCIL_W], vv[STENCIL_W];
        gather7(u, m.cols, n, i, uu);
        gather7(v, m.cols, n, i, vv);
        double acc[STENCIL_W] = {0, 0, 0, 0, 0, 0, 0};
        for_each_tri(p, m.nc, [&](const TriInfo& T) {
            double v0 = vv[T.slot[0]], v1 = vv[T.slot[1]], v2 = vv[T.slot[2]];
            double gvx = (v0 * tri_gx(T.type, 0) + v1 * tri_gx(T.type, 1) + v2 * tri_gx(T.type, 2)) * ih;
            double gvy = (v0 * tri_gy(T.type, 0) + v1 * tri_gy(T.type, 1) + v2 * tri_gy(T.type, 2)) * ih;
            double u0 = uu[T.slot[0]], u1 = uu[T.slot[1]], u2 = uu[T.slot[2]];
            if (!ADJ) {
                double gvp = (gvx * tri_gx(T.type, T.pl) + gvy * tri_gy(T.type, T.pl)) * ih;  // grad v . grad phi_P
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    double l0 = quad6_l(q, 0), l1 = quad6_l(q, 1), l2 = quad6_l(q, 2);
                    double e = exp(-eta * (l0 * u0 + l1 * u1 + l2 * u2));
                    double w = quad6_w(q) * area * e * gvp;
                    acc[T.slot[0]] += w * l0;
                    acc[T.slot[1]] += w * l1;
                    acc[T.slot[2]] += w * l2;
                }
            } else {
                double t = 0.0;  // int (1 - eta u) exp(-eta u) phi_P
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    double uq = quad7_l(q, 0) * u0 + quad7_l(q, 1) * u1 + quad7_l(q, 2) * u2;
                    t += quad7_w(q) * area * (1.0 - eta * uq) * exp(-eta * uq) * quad7_l(q, T.pl);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc[T.slot[k]] += t * (gvx * tri_gx(T.type, k) + gvy * tri_gy(T.type, k)) * ih;  // grad phi_j . grad v
            }
        });
#pragma unroll
        for (int k = 0; k < STENCIL_W; ++k) {
            int64_t idx = (int64_t)k * n + i;
            out[idx] = Dm * m.Ad[idx] - chi * acc[k];
        }
    }
}

// chemotaxis adjoint rhs for q (helpers.py:1531-1534):
//   out_i = int chi u_h exp(-eta u_h) (grad p_h . grad phi_i)  [+ (da_i - db_i)]    (6-point rule)
// mx_ref != null: the species right-hand side in one pass, out_i = s0 * (M mx)_i + s2 * rhs_q_i (helpers.py:1538) -- the
// expressions of k_load applied to this row's own value, so the bits of the two-launch sequence
__device__ __forceinline__ void form_chtxs_rhs_q(const MeshArgs& m, VecRef u_ref, int64_t u_bs, VecRef p_ref, int64_t p_bs, double chi,
                                                 double eta, VecRef da_ref, int64_t da_bs, VecRef db_ref, int64_t db_bs,
                                                 VecRef mx_ref, int64_t mx_bs, double s0, double s2, double* __restrict__ out_, int bz) {
    const int n = m.n;
    const double* mx = bptr(mx_ref, mx_bs, bz);
    const double* u = bptr(u_ref, u_bs, bz);
    const double* pp = bptr(p_ref, p_bs, bz);
    const double* da = bptr(da_ref, da_bs, bz);
    const double* db = bptr(db_ref, db_bs, bz);
    double* out = out_ + (int64_t)bz * n;
    const double area = 0.5 * m.h * m.h, ih = 1.0 / m.h;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        NodeXY p = node_xy(i, m.d2v, m.N);
        double uu[STENCIL_W], pv[STENCIL_W];
        gather7(u, m.cols, n, i, uu);
        gather7(pp, m.cols, n, i, pv);
        double res = 0.0;
        for_each_tri(p, m.nc, [&](const TriInfo& T) {
            double p0 = pv[T.slot[0]], p1 = pv[T.slot[1]], p2 = pv[T.slot[2]];
            double gpx = (p0 * tri_gx(T.type, 0) + p1 * tri_gx(T.type, 1) + p2 * tri_gx(T.type, 2)) * ih;
            double gpy = (p0 * tri_gy(T.type, 0) + p1 * tri_gy(T.type, 1) + p2 * tri_gy(T.type, 2)) * ih;
            double gpp = (gpx * tri_gx(T.type, T.pl) + gpy * tri_gy(T.type, T.pl)) * ih;
            double u0 = uu[T.slot[0]], u1 = uu[T.slot[1]], u2 = uu[T.slot[2]];
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                double uq = quad6_l(q, 0) * u0 + quad6_l(q, 1) * u1 + quad6_l(q, 2) * u2;
                t += quad6_w(q) * area * chi * uq * exp(-eta * uq);
            }
            res += t * gpp;
        });
        if (da) res += da[i] - (db ? db[i] : 0.0);
        if (mx) {
            double fin = 0.0;
            if (s0 != 0.0) {
                double acc = m.M[i] * mx[i];
#pragma unroll
                for (int s = 1; s < STENCIL_W; ++s) {
                    int64_t idx = (int64_t)s * n + i;
                    acc += m.M[idx] * mx[m.cols[idx]];
                }
                fin += s0 * acc;
            }
            if (s2 != 0.0) fin += s2 * (res - 0.0);
            res = fin;
        }
        out[i] = res;
    }
}

// ---------------------------------------------------------------------------
// the kernels: one form per launch, or up to three independent forms in ONE launch (blockIdx.z picks the job): on the
// config meshes a form is ~1 us of work inside ~5 us of launch latency, and the forms of a time step that depend only
// on earlier levels can share it (FormGroup, forms.h)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void run_form(const MeshArgs& m, const FormJob& j, int bz) {
    if (bz >= j.batch) return;
    switch (j.type) {
        case FORM_WMASS: form_weighted_mass(m, j.w, j.out, bz); break;
        case FORM_LOAD: form_load(m, j.l, j.out, bz); break;
        case FORM_CHTXS_MAT0: form_chtxs_matrix<0>(m, j.c.u, j.c.u_bs, j.c.v, j.c.v_bs, j.c.p0, j.c.p1, j.c.p2, j.out, bz); break;
        case FORM_CHTXS_MAT1: form_chtxs_matrix<1>(m, j.c.u, j.c.u_bs, j.c.v, j.c.v_bs, j.c.p0, j.c.p1, j.c.p2, j.out, bz); break;
        default: break;
    }
}

__global__ void k_weighted_mass(MeshArgs m, WMassSpec sp, double* __restrict__ out_) { form_weighted_mass(m, sp, out_, blockIdx.y); }
__global__ void k_load(MeshArgs m, LoadSpec sp, double* __restrict__ out_) { form_load(m, sp, out_, blockIdx.y); }
template <int ADJ>
__global__ void k_chtxs_matrix(MeshArgs m, VecRef u_ref, int64_t u_bs, VecRef v_ref, int64_t v_bs, double Dm, double chi,
                               double eta, double* __restrict__ out_) {
    form_chtxs_matrix<ADJ>(m, u_ref, u_bs, v_ref, v_bs, Dm, chi, eta, out_, blockIdx.y);
}
__global__ void k_chtxs_rhs_q(MeshArgs m, VecRef u_ref, int64_t u_bs, VecRef p_ref, int64_t p_bs, double chi, double eta,
                              VecRef da_ref, int64_t da_bs, VecRef db_ref, int64_t db_bs, VecRef mx_ref, int64_t mx_bs,
                              double s0, double s2, double* __restrict__ out_) {
    form_chtxs_rhs_q(m, u_ref, u_bs, p_ref, p_bs, chi, eta, da_ref, da_bs, db_ref, db_bs, mx_ref, mx_bs, s0, s2, out_, blockIdx.y);
}
__global__ void k_forms2(MeshArgs m, FormJob j0, FormJob j1) { run_form(m, blockIdx.z == 0 ? j0 : j1, blockIdx.y); }
__global__ void k_forms3(MeshArgs m, FormJob j0, FormJob j1, FormJob j2) {
    run_form(m, blockIdx.z == 0 ? j0 : (blockIdx.z == 1 ? j1 : j2), blockIdx.y);
}

}  // namespace

MeshArgs femfct_mesh_args(const femfct_ctx* ctx) {
    MeshArgs m;
    m.n = ctx->n; m.N = ctx->N; m.nc = ctx->n_cells; m.h = ctx->h;
    m.d2v = ctx->d_d2v; m.cols = ctx->d_cols; m.M = ctx->d_M; m.Ad = ctx->d_Ad;
    return m;
}

int femfct_enqueue_weighted_mass(femfct_ctx* ctx, const WMassSpec& sp, double* out, int32_t batch) {
    LaunchGeom g = femfct_geom(ctx, batch);
    femfct_prof_begin(ctx, KC_ASSEMBLE);
    hipLaunchKernelGGL(k_weighted_mass, g.grid, g.block, 0, ctx->stream, femfct_mesh_args(ctx), sp, out);
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

int femfct_enqueue_load(femfct_ctx* ctx, const LoadSpec& sp, double* out, int32_t batch) {
    LaunchGeom g = femfct_geom(ctx, batch);
    femfct_prof_begin(ctx, KC_ASSEMBLE);
    hipLaunchKernelGGL(k_load, g.grid, g.block, 0, ctx->stream, femfct_mesh_args(ctx), sp, out);
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

int femfct_enqueue_chtxs_matrix(femfct_ctx* ctx, int adjoint, VecRef u, int64_t u_bs, VecRef v, int64_t v_bs,
                                double Dm, double chi, double eta, double* out, int32_t batch) {
    LaunchGeom g = femfct_geom(ctx, batch);
    femfct_prof_begin(ctx, KC_ASSEMBLE);
    if (adjoint)
        hipLaunchKernelGGL(k_chtxs_matrix<1>, g.grid, g.block, 0, ctx->stream, femfct_mesh_args(ctx), u, u_bs, v, v_bs,
                           Dm, chi, eta, out);
    else
        hipLaunchKernelGGL(k_chtxs_matrix<0>, g.grid, g.block, 0, ctx->stream, femfct_mesh_args(ctx), u, u_bs, v, v_bs,
                           Dm, chi, eta, out);
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

int femfct_enqueue_chtxs_rhs_q(femfct_ctx* ctx, VecRef u, int64_t u_bs, VecRef p, int64_t p_bs, double chi, double eta,
                               VecRef da, int64_t da_bs, VecRef db, int64_t db_bs, double* out, int32_t batch,
                               VecRef mx, int64_t mx_bs, double s0, double s2) {
    LaunchGeom g = femfct_geom(ctx, batch);
    femfct_prof_begin(ctx, KC_ASSEMBLE);
    hipLaunchKernelGGL(k_chtxs_rhs_q, g.grid, g.block, 0, ctx->stream, femfct_mesh_args(ctx), u, u_bs, p, p_bs, chi, eta,
                       da, da_bs, db, db_bs, mx, mx_bs, s0, s2, out);
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

// ------------------------------------------------------------------ FormGroup
void FormGroup::weighted_mass(const WMassSpec& sp, double* out, int32_t batch) {
    if (n == 3) launch();
    FormJob& j = jobs[n++];
    j = FormJob{};
    j.type = FORM_WMASS; j.batch = batch; j.out = out; j.w = sp;
}

void FormGroup::load(const LoadSpec& sp, double* out, int32_t batch) {
    if (n == 3) launch();
    FormJob& j = jobs[n++];
    j = FormJob{};
    j.type = FORM_LOAD; j.batch = batch; j.out = out; j.l = sp;
}

void FormGroup::chtxs_matrix(int adjoint, VecRef u, int64_t u_bs, VecRef v, int64_t v_bs, double Dm, double chi, double eta,
                             double* out, int32_t batch) {
    if (n == 3) launch();
    FormJob& j = jobs[n++];
    j = FormJob{};
    j.type = adjoint ? FORM_CHTXS_MAT1 : FORM_CHTXS_MAT0; j.batch = batch; j.out = out;
    j.c.u = u; j.c.u_bs = u_bs; j.c.v = v; j.c.v_bs = v_bs; j.c.p0 = Dm; j.c.p1 = chi; j.c.p2 = eta;
}

int FormGroup::launch() {
    const int cnt = n;
    n = 0;
    if (cnt == 0) return FEMFCT_OK;
    if (cnt == 1 || !ctx->form_groups) {          // one form, or FEMFCT_FORM_GROUPS=0: the forms' own kernels, in order
        for (int k = 0; k < cnt; ++k) {
            const FormJob& j = jobs[k];
            int r = FEMFCT_OK;
            if (j.type == FORM_WMASS) r = femfct_enqueue_weighted_mass(ctx, j.w, j.out, j.batch);
            else if (j.type == FORM_LOAD) r = femfct_enqueue_load(ctx, j.l, j.out, j.batch);
            else r = femfct_enqueue_chtxs_matrix(ctx, j.type == FORM_CHTXS_MAT1, j.c.u, j.c.u_bs, j.c.v, j.c.v_bs, j.c.p0, j.c.p1,
                                                 j.c.p2, j.out, j.batch);
            if (r != FEMFCT_OK) return r;
        }
        return FEMFCT_OK;
    }
    int32_t bmax = 1;
    for (int k = 0; k < cnt; ++k) bmax = std::max(bmax, jobs[k].batch);
    LaunchGeom g = femfct_geom(ctx, bmax);
    g.grid.z = cnt;
    femfct_prof_begin(ctx, KC_ASSEMBLE);
    if (cnt == 2) hipLaunchKernelGGL(k_forms2, g.grid, g.block, 0, ctx->stream, femfct_mesh_args(ctx), jobs[0], jobs[1]);
    else hipLaunchKernelGGL(k_forms3, g.grid, g.block, 0, ctx->stream, femfct_mesh_args(ctx), jobs[0], jobs[1], jobs[2]);
    femfct_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return femfct_fail(ctx, FEMFCT_ERR_HIP, "form group launch failed: %s", hipGetErrorString(e));
    return FEMFCT_OK;
}
