// The flux operator of the drift-control problem, row by row, shared by the kernel that stores it
// (k_ops_solidbody) and the step kernels that derive it on the fly in the bandwidth regime
// (k_build_low_sb, k_dudt_rhs_sb): one definition => the same bits either way.
//
//   A = eps*Ad + sigma*(rot_scale*Arot + Adrift1(c) + Adrift2(c))
//   Adrift1[P,j] = (b.grad c_h)|_K M_K[P,j]          dot(drift, grad(c))*u*v*dx
//   Adrift2[P,j] = (b.grad lambda_P)(M_K c_K)_j      dot(drift, grad(v))*c*u*dx
//   /root/reference/advection_solidbody_FCT_PDECO_finaltime.py:187-193 (state, sigma = -1 through FCT_alg),
//   :215-218 (adjoint, sigma = +1)
#pragma once

#include "device_utils.h"
#include "stencil.h"

struct SbOpArgs {
    const double* Arot;     // ELL, constant
    const double* Ad;       // ELL stiffness (read only when eps != 0)
    VecRef c;               // control of this step (level-indirected), batch stride c_bstride
    int64_t c_bstride;
    double eps, sigma, rot_scale, bx, by;
    // rot_geom: Arot is the rotation operator of the wind om * (-y, x) as femfct_assemble_rotation stores it, bit for
    // bit (checked by the caller): the step kernels evaluate its rows from (ix, iy) instead of reading 56 B per row
    int rot_geom;
    double om, a1;          // angular velocity, lower-left corner of the mesh
};

#ifdef __HIPCC__

struct NodeXY { int ix, iy; };

__device__ __forceinline__ NodeXY node_xy(int i, const int32_t* __restrict__ d2v, int N) {
    int v = d2v ? d2v[i] : i;
    return NodeXY{v % N, v / N};
}

// the <= 6 triangles around node p, counter-clockwise from the cell (ix, iy)
template <class F>
__device__ __forceinline__ void for_each_tri(NodeXY p, int nc, F&& f) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const TriInfo T = tri_info(t);
        int cx = p.ix + T.cdx, cy = p.iy + T.cdy;
        if (cx < 0 || cy < 0 || cx >= nc || cy >= nc) continue;
        f(T, cx, cy);
    }
}

// Drift part of row P: acc[s] = (Adrift1 + Adrift2)[P, col(s)] from the control values cv[s] of the node's 1-ring.
// WITH_T: also accT[s] = (Adrift1 + Adrift2)[col(s), P] (s >= 1) -- the transposed entries the artificial diffusion
// needs.  Both triangles of the edge (P, j) lie in P's ring, and each term is the expression row j itself
// evaluates (two-term sums commute), so accT[s] has the bits of row j's stored entry.
template <bool WITH_T>
__device__ __forceinline__ void sb_drift_row(NodeXY p, int nc, double h, const double (&cv)[STENCIL_W], double bx,
                                             double by, double (&acc)[STENCIL_W], double (&accT)[STENCIL_W]) {
    const double m12 = 0.5 * h * h / 12.0;
    const double rh = 1.0 / h;          // one division per row instead of two to four per triangle (f64 divides are ~10 x an FMA)
#pragma unroll
    for (int s = 0; s < STENCIL_W; ++s) { acc[s] = 0.0; accT[s] = 0.0; }
    for_each_tri(p, nc, [&](const TriInfo& T, int, int) {
        double c0 = cv[T.slot[0]], c1 = cv[T.slot[1]], c2 = cv[T.slot[2]];
        // b . grad c_h  (constant on the triangle)
        double gcx = (c0 * tri_gx(T.type, 0) + c1 * tri_gx(T.type, 1) + c2 * tri_gx(T.type, 2)) * rh;
        double gcy = (c0 * tri_gy(T.type, 0) + c1 * tri_gy(T.type, 1) + c2 * tri_gy(T.type, 2)) * rh;
        double bgc = bx * gcx + by * gcy;
        double bgp = (bx * tri_gx(T.type, T.pl) + by * tri_gy(T.type, T.pl)) * rh;
        double csum = c0 + c1 + c2;
        double ck[3] = {c0, c1, c2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double d1 = bgc * m12 * (k == T.pl ? 2.0 : 1.0);
            double d2 = bgp * m12 * (ck[k] + csum);
            acc[T.slot[k]] += d1 + d2;
            if (WITH_T && k != T.pl) {
                // row = local node k, column = P: what row k's own loop adds for its column P
                double bgk = (bx * tri_gx(T.type, k) + by * tri_gy(T.type, k)) * rh;
                double t1 = bgc * m12 * 1.0;
                double t2 = bgk * m12 * (ck[T.pl] + csum);
                accT[T.slot[k]] += t1 + t2;
            }
        }
    });
}

// Rotation part in closed form.  The wind w = om * (-y, x) is linear, so on a triangle K
//   int_K (w . grad lambda_P) lambda_k dx = grad lambda_P . |K|/12 (w_0 + w_1 + w_2 + w_k)        (w_k: wind at local node k)
// -- the same integral the quadrature of k_convection_tab evaluates (exactly, up to rounding).  rot[s] = Arot[P, col(s)];
// WITH_T: rotT[s] = Arot[col(s), P], the term row col(s) itself evaluates for its column P (same triangle, same local
// numbering, the roles of the two local nodes swapped; two-term sums commute) => the bits of row col(s)'s entry.
// No contraction: the expression tree below is what every kernel that inlines this evaluates.
template <bool WITH_T>
__device__ __forceinline__ void sb_rot_row(NodeXY p, int nc, double h, double a1, double om, double (&rot)[STENCIL_W],
                                           double (&rotT)[STENCIL_W]) {
#pragma clang fp contract(off)
    const double c = h / 24.0;          // |K| / 12 / h
#pragma unroll
    for (int s = 0; s < STENCIL_W; ++s) { rot[s] = 0.0; rotT[s] = 0.0; }
    for_each_tri(p, nc, [&](const TriInfo& T, int cx, int cy) {
        double wx[3], wy[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double xk = a1 + (double)(cx + tri_nx(T.type, k)) * h;
            const double yk = a1 + (double)(cy + tri_ny(T.type, k)) * h;
            wx[k] = -om * yk;
            wy[k] = om * xk;
        }
        const double Wx = (wx[0] + wx[1]) + wx[2], Wy = (wy[0] + wy[1]) + wy[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rot[T.slot[k]] += c * (tri_gx(T.type, T.pl) * (Wx + wx[k]) + tri_gy(T.type, T.pl) * (Wy + wy[k]));
            if (WITH_T && k != T.pl)
                rotT[T.slot[k]] += c * (tri_gx(T.type, k) * (Wx + wx[T.pl]) + tri_gy(T.type, k) * (Wy + wy[T.pl]));
        }
    });
}

#endif  // __HIPCC__
