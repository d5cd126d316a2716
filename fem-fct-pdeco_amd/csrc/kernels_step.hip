// FEM-FCT step operator on gfx950: hand-written HIP kernels for
//   FCT_alg_ref                 /root/reference/helpers.py:1715-1872
//   ChebSI                      /root/reference/helpers.py:143-185
//   artificial_diffusion_mat    /root/reference/helpers.py:206-242
//
// One thread owns one matrix row (ELL, slot-major => every matrix/index load of
// a wave is one coalesced 512-byte line per slot); neighbour values are gathered
// through L1/L2.  The flux F_ij is recomputed per row from the row's own data
// (F_ji = -F_ij), so nothing is scattered and no floating-point atomics exist:
// results are bitwise reproducible run to run.  The only cross-thread
// reductions are max/min (order independent), done with wave64 shuffles, LDS,
// and per-block partials re-reduced by every block of the consuming kernel.
//
// Kernel sequence of one step (all bandwidth bound; bytes per row, W = 7):
//   k_build_low   A -> D, L, b, x0, row-sum / ||b|| partials        W*37+32  B
//   k_jacobi  xI  x <- D^-1 (b - offdiag(L) x), residual partials   W*12+32  B
//   k_dudt_rhs    r = rhs - A u_L ; first Chebyshev iterate         W*12+40  B
//   k_cheb   x19  Chebyshev semi-iteration on M                     W*12+40  B
//   k_flux        F_ij, P+-, Q+-, R+-                               W*28+40  B
//   k_limit       alpha_ij, Fbar, u^{n+1}                           W*12+40  B
#include "femfct_internal.h"
#include "device_utils.h"
#include "solve_ctl.h"
#include "forms.h"
#include "solidbody_op.h"

#include <math.h>
#include <stdlib.h>

namespace {



// Column index of slot s in row i.  IMP = 0: explicit ELL column table (any pattern / ordering).
// IMP = 1: structured mesh in vertex order -- the neighbour is i + const offset, so no index bytes are
// read and the x-gather does not wait for an index load; slots that fall outside the grid resolve
// to the row itself, exactly like the padded entries of the explicit table.
template <int IMP>
__device__ __forceinline__ unsigned nb_mask(int i, int Nw) {
    if (!IMP) return 0u;
    const int iy = i / Nw, ix = i - iy * Nw;
    const unsigned e = ix < Nw - 1, w = ix > 0, nn = iy < Nw - 1, ss = iy > 0;
    return (e << 1) | ((e & nn) << 2) | (nn << 3) | (w << 4) | ((w & ss) << 5) | (ss << 6);
}

template <int IMP>
__device__ __forceinline__ int col_of(const int32_t* __restrict__ cols, int n, int Nw, unsigned mask, int s, int i) {
    if (IMP) {
        const int off = (s == 1) ? 1 : (s == 2) ? Nw + 1 : (s == 3) ? Nw : (s == 4) ? -1 : (s == 5) ? -Nw - 1 : -Nw;
        return ((mask >> s) & 1u) ? i + off : i;
    }
    return cols[(int64_t)s * n + i];
}

template <int IMP>
__device__ __forceinline__ int tslot_of(const uint8_t* __restrict__ tslot, int n, unsigned mask, int s, int i) {
    if (IMP) {
        const int opp = (s < 4) ? s + 3 : s - 3;
        return ((mask >> s) & 1u) ? opp : s;
    }
    return tslot[(int64_t)s * n + i];
}

// Where a step kernel gets the entries of the flux matrix A from.  MatOp: a stored ELL matrix (any caller).
// SbOp: the drift-control operator derived on the fly from Arot and the control's 1-ring (bandwidth regime of the
// solid-body sweeps: saves writing A once and reading it twice per step; solidbody_op.h keeps the bits identical).
struct MatOp {
    const double* A;
    int n;
    __device__ __forceinline__ void row(int, unsigned) {}
    __device__ __forceinline__ double a(int s, int i) const { return A[(int64_t)s * n + i]; }
    __device__ __forceinline__ double at(int, int, int j, int ts) const { return A[(int64_t)ts * n + j]; }
};

// ROTG: the rotation part from the node's lattice position (sb_rot_row) instead of the stored Arot.
template <bool WITH_T, bool ROTG>
struct SbOp {
    SbOpArgs p;
    const double* c;
    int n, Nw;
    double h;
    double acc[STENCIL_W], accT[STENCIL_W];      // ROTG: rot_scale * Arot + drift part; else the drift part
    __device__ __forceinline__ void row(int i, unsigned mask) {
        double cv[STENCIL_W];
        cv[0] = c[i];
#pragma unroll
        for (int s = 1; s < STENCIL_W; ++s) cv[s] = c[col_of<1>(nullptr, n, Nw, mask, s, i)];
        const int iy = i / Nw;
        const NodeXY xy{i - iy * Nw, iy};
        sb_drift_row<WITH_T>(xy, Nw - 1, h, cv, p.bx, p.by, acc, accT);
        if (ROTG) {
            double rot[STENCIL_W], rotT[STENCIL_W];
            sb_rot_row<WITH_T>(xy, Nw - 1, h, p.a1, p.om, rot, rotT);
#pragma unroll
            for (int s = 0; s < STENCIL_W; ++s) {
                acc[s] = p.rot_scale * rot[s] + acc[s];
                if (WITH_T) accT[s] = p.rot_scale * rotT[s] + accT[s];
            }
        }
    }
    __device__ __forceinline__ double a(int s, int i) const {
        const int64_t idx = (int64_t)s * n + i;
        if (ROTG) return p.eps * (p.eps != 0.0 ? p.Ad[idx] : 0.0) + p.sigma * acc[s];
        return p.eps * (p.eps != 0.0 ? p.Ad[idx] : 0.0) + p.sigma * (p.rot_scale * p.Arot[idx] + acc[s]);
    }
    // a_ji for the neighbour j in slot s (its slot towards i is ts); Ad is symmetric to the bit
    __device__ __forceinline__ double at(int s, int i, int j, int ts) const {
        if (ROTG) return p.eps * (p.eps != 0.0 ? p.Ad[(int64_t)s * n + i] : 0.0) + p.sigma * accT[s];
        return p.eps * (p.eps != 0.0 ? p.Ad[(int64_t)s * n + i] : 0.0) +
               p.sigma * (p.rot_scale * p.Arot[(int64_t)ts * n + j] + accT[s]);
    }
};

// ---------------------------------------------------------------------------
// k_build_low: artificial diffusion + low-order operator + rhs (helpers.py:1769-1780)
// ---------------------------------------------------------------------------
template <int WT, int BS, int IMP, class AOp>
__device__ __forceinline__ void build_low_body(int n, int Wrt, int Nw, const int32_t* __restrict__ cols,
                            const uint8_t* __restrict__ tslot, AOp op,
                            const double* __restrict__ N_, int nshared, VecRef rhs_ref, VecRef u_ref,
                            int64_t rhs_bstride, int64_t u_bstride,
                            const double* __restrict__ ml, double dt, double* __restrict__ L_,
                            double* __restrict__ D_, double* __restrict__ b_, double* __restrict__ x0_,
                            double* __restrict__ part, StepCtl* __restrict__ ctl_,
                            uint8_t* __restrict__ lmask, int half_d) {
    // half_d: d_ij = d_ji to the bit (max is commutative), so only the three "forward" slots E, NE, N are stored;
    // the limiter of the bandwidth regime (k_tile_flux_limit<.., HALFD = 1>) takes the other three from the neighbours
    __shared__ double smem[32];
    const int W = WT ? WT : Wrt;
    const int bz = blockIdx.y;
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* Nm = N_ ? N_ + (nshared ? 0 : moff) : nullptr;
    const double* rhs = vec_ptr(rhs_ref);
    if (rhs) rhs += bz * rhs_bstride;
    const double* u = vec_ptr(u_ref) + bz * u_bstride;
    double* L = L_ + moff;
    double* D = D_ + moff;
    double* b = b_ + voff;
    double* x0 = x0_ + voff;

    if (blockIdx.x == 0 && threadIdx.x == 0) {
        StepCtl* c = ctl_ + bz;
        c->flags = 0; c->iters = 0; c->done = 0; c->parity = 0; c->resid = 0.0; c->bnorm = 0.0;
        c->min_rowsum = 0.0;
    }

    RowRange rr = block_rows(n);
    double bmax = 0.0, rsmin = INFINITY;
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        const unsigned mask = nb_mask<IMP>(i, Nw);
        op.row(i, mask);
        double dsum = 0.0, rs = 0.0;
        unsigned nzrow = 0;
        if constexpr (WT != 0) {
            // Every load of the row first, the stores last: the operator's arrays are not known to be distinct from L / D
            // (plain pointers inside the functor), so a store inside the slot loop pins the next slot's loads behind it --
            // six dependent round trips per row instead of one (k_build_low_sb at 2049^2: 167 us, latency- not
            // bandwidth-bound).
            double lrow[WT - 1], drow[WT - 1];
#pragma unroll
            for (int s = 1; s < WT; ++s) {
                const int64_t idx = (int64_t)s * n + i;
                const int j = col_of<IMP>(cols, n, Nw, mask, s, i);
                const int ts = tslot_of<IMP>(tslot, n, mask, s, i);
                const double a = op.a(s, i);
                const double at = op.at(s, i, j, ts);
                const double d = (j != i) ? fmax(0.0, fmax(a, at)) : 0.0;   // d_ij = max(0, a_ij, a_ji)
                dsum += d;
                double l = dt * (a - d);
                if (Nm) l += dt * Nm[idx];
                rs += l;
                lrow[s - 1] = l;
                drow[s - 1] = d;
            }
#pragma unroll
            for (int s = 1; s < WT; ++s) {
                const int64_t idx = (int64_t)s * n + i;
                if (!half_d || s <= 3) D[idx] = drow[s - 1];
                // which entries of this row are exactly zero (the low-order operator is an upwind stencil: about half of
                // its off-diagonals vanish): six bits per row, one byte per node.  With the mask in force its only reader
                // (k_strip4_jacobi[_walk]) never touches a vanishing entry, so those are not stored either: a 128-byte
                // line of zeros is neither written here nor read there.
                if (lmask) {
                    if (lrow[s - 1] != 0.0) { L[idx] = lrow[s - 1]; nzrow |= 1u << (s - 1); }
                } else {
                    L[idx] = lrow[s - 1];
                }
            }
        } else {
            for (int s = 1; s < W; ++s) {
                int64_t idx = (int64_t)s * n + i;
                int j = col_of<IMP>(cols, n, Nw, mask, s, i);
                int ts = tslot_of<IMP>(tslot, n, mask, s, i);
                double a = op.a(s, i);
                double at = op.at(s, i, j, ts);
                double d = (j != i) ? fmax(0.0, fmax(a, at)) : 0.0;
                dsum += d;
                double l = dt * (a - d);
                if (Nm) l += dt * Nm[idx];
                if (!half_d || s <= 3) D[idx] = d;
                rs += l;
                if (lmask) {
                    if (l != 0.0) { L[idx] = l; nzrow |= 1u << (s - 1); }
                } else {
                    L[idx] = l;
                }
            }
        }
        double mli = ml[i];
        double ld = mli + dt * (op.a(0, i) + dsum);                // d_ii = -sum_j d_ij
        if (Nm) ld += dt * Nm[i];
        L[i] = ld;                  // (the diagonal of D is -dsum; the limiter reads off-diagonals only: not stored)
        if (lmask) lmask[voff + i] = (uint8_t)nzrow;
        rs += ld;
        double ui = u[i];
        double bi = mli * ui + (rhs ? dt * rhs[i] : 0.0);
        b[i] = bi;
        x0[i] = ui;
        bmax = fmax(bmax, fabs(bi));
        rsmin = fmin(rsmin, rs);
    }
    bmax = block_reduce(bmax, OpMax(), 0.0, smem);
    rsmin = block_reduce(rsmin, OpMin(), INFINITY, smem);
    if (threadIdx.x == 0) {
        double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
        p[2 * FEMFCT_MAX_PARTIALS + blockIdx.x] = bmax;
        p[3 * FEMFCT_MAX_PARTIALS + blockIdx.x] = rsmin;
    }
}

template <int WT, int BS, int IMP>
__global__ void __launch_bounds__(BS) k_build_low(int n, int Wrt, int Nw, const int32_t* __restrict__ cols,
                            const uint8_t* __restrict__ tslot, MatRef A_ref,
                            const double* __restrict__ N_, int nshared, VecRef rhs_ref, VecRef u_ref,
                            int64_t rhs_bstride, int64_t u_bstride,
                            const double* __restrict__ ml, double dt, double* __restrict__ L_,
                            double* __restrict__ D_, double* __restrict__ b_, double* __restrict__ x0_,
                            double* __restrict__ part, StepCtl* __restrict__ ctl_,
                            uint8_t* __restrict__ lmask, int half_d) {
    build_low_body<WT, BS, IMP>(n, Wrt, Nw, cols, tslot, MatOp{mat_ptr(A_ref, blockIdx.y), n}, N_, nshared, rhs_ref, u_ref,
                                rhs_bstride, u_bstride, ml, dt, L_, D_, b_, x0_, part, ctl_, lmask, half_d);
}

// the same with the solid-body operator derived on the fly (structured mesh, vertex order)
template <int BS, bool ROTG>
__global__ void __launch_bounds__(BS) k_build_low_sb(int n, int Nw, double h, SbOpArgs sb, VecRef rhs_ref, VecRef u_ref,
                            int64_t rhs_bstride, int64_t u_bstride,
                            const double* __restrict__ ml, double dt, double* __restrict__ L_,
                            double* __restrict__ D_, double* __restrict__ b_, double* __restrict__ x0_,
                            double* __restrict__ part, StepCtl* __restrict__ ctl_,
                            uint8_t* __restrict__ lmask, int half_d) {
    SbOp<true, ROTG> op;
    op.p = sb; op.c = vec_ptr(sb.c) + blockIdx.y * sb.c_bstride; op.n = n; op.Nw = Nw; op.h = h;
    build_low_body<7, BS, 1>(n, 7, Nw, nullptr, nullptr, op, nullptr, 0, rhs_ref, u_ref, rhs_bstride, u_bstride, ml, dt,
                             L_, D_, b_, x0_, part, ctl_, lmask, half_d);
}

// ---------------------------------------------------------------------------
// k_jacobi: one sweep of x <- Dg^-1 (b - O x) for the M-matrix L = Dg + O
// (replaces the per-step SuperLU factorisation, helpers.py:1782).
// Since x_new - x_old = Dg^-1 r(x_old), every sweep also yields the true residual
// of its input iterate; convergence is decided on the device from those.
// ---------------------------------------------------------------------------
template <int WT, int BS, int IMP>
__global__ void __launch_bounds__(BS) k_jacobi(int n, int Wrt, int Nw, const int32_t* __restrict__ cols, const double* __restrict__ L_,
                         const double* __restrict__ b_, double* __restrict__ xa_, double* __restrict__ xb_,
                         double* __restrict__ part, StepCtl* __restrict__ ctl_, int sweep, double rel_tol) {
    __shared__ double smem[32];
    const int W = WT ? WT : Wrt;
    const int bz = blockIdx.y;
    StepCtl* ctl = ctl_ + bz;
    if (ctl->done) return;
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    const int G = gridDim.x;
    double bnorm;
    if (sweep == 0) {
        bnorm = reduce_partials(p + 2 * FEMFCT_MAX_PARTIALS, G, OpMax(), 0.0, smem);
        double rsmin = reduce_partials(p + 3 * FEMFCT_MAX_PARTIALS, G, OpMin(), INFINITY, smem);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctl->bnorm = bnorm;
            ctl->min_rowsum = rsmin;
            if (!(rsmin > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        }
    } else {
        bnorm = ctl->bnorm;
        double rmax = reduce_partials(p + ((sweep - 1) & 1) * FEMFCT_MAX_PARTIALS, G, OpMax(), 0.0, smem);
        if (rmax <= rel_tol * bnorm) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                ctl->done = 1; ctl->parity = sweep & 1; ctl->iters = sweep;
                ctl->resid = bnorm > 0.0 ? rmax / bnorm : 0.0;
            }
            return;
        }
    }
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* L = L_ + moff;
    const double* b = b_ + voff;
    const double* xin = ((sweep & 1) ? xb_ : xa_) + voff;
    double* xout = ((sweep & 1) ? xa_ : xb_) + voff;

    RowRange rr = block_rows(n);
    double rmax = 0.0;
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        const unsigned mask = nb_mask<IMP>(i, Nw);
        double acc = b[i];
#pragma unroll
        for (int s = 1; s < W; ++s) {
            int64_t idx = (int64_t)s * n + i;
            acc = fma(-L[idx], xin[col_of<IMP>(cols, n, Nw, mask, s, i)], acc);
        }
        double ld = L[i];
        double xi = xin[i];
        double xn = acc / ld;
        xout[i] = xn;
        rmax = fmax(rmax, fabs(acc - ld * xi));   // |r_i(x_in)|
    }
    rmax = block_reduce(rmax, OpMax(), 0.0, smem);
    if (threadIdx.x == 0) p[(sweep & 1) * FEMFCT_MAX_PARTIALS + blockIdx.x] = rmax;
}

// ---------------------------------------------------------------------------
// k_dudt_rhs: r = rhs - A u_L (helpers.py:1814) fused with Chebyshev iterate 1
// (y_1 = omega_1 * r / (1.25 diag M), helpers.py:175-182 with y_0 = y_-1 = 0).
// ---------------------------------------------------------------------------
template <int WT, int BS, int IMP, class AOp>
__device__ __forceinline__ void dudt_rhs_body(int n, int Wrt, int Nw, const int32_t* __restrict__ cols, AOp op,
                           VecRef rhs_ref, int64_t rhs_bstride, const double* __restrict__ M,
                           const double* __restrict__ xa_, const double* __restrict__ xb_,
                           double* __restrict__ ulow_, double* __restrict__ rdu_, double* __restrict__ y1_,
                           double* __restrict__ part, StepCtl* __restrict__ ctl_, int budget, int part_count,
                           int iters_per_unit, double rel_tol, double md_scale, double omega1,
                           const double* __restrict__ partk, int exact_k) {
    __shared__ double smem[32];
    const int W = WT ? WT : Wrt;
    const int bz = blockIdx.y;
    StepCtl* ctl = ctl_ + bz;
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    // solution buffer: decided by the sweep that detected convergence, else by the budget parity
    const int parity = ctl->done ? ctl->parity : (budget & 1);
    finalize_solve(ctl, p, part_count ? part_count : (int)gridDim.x, budget, iters_per_unit, rel_tol, smem,
                   partk ? partk + (int64_t)bz * 16 * FEMFCT_MAX_PARTIALS : nullptr, exact_k, blockIdx.x == 0);
    const int64_t voff = (int64_t)bz * n;
    const double* x = (parity ? xb_ : xa_) + voff;
    const double* rhs = vec_ptr(rhs_ref);
    if (rhs) rhs += bz * rhs_bstride;
    double* ulow = ulow_ + voff;
    double* rdu = rdu_ + voff;
    double* y1 = y1_ + voff;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        const unsigned mask = nb_mask<IMP>(i, Nw);
        op.row(i, mask);
        double xi = x[i];
        // (requested before the stores below: the compiler cannot tell M / rhs from the output arrays, and a load
        // behind a store waits for it)
        const double mdi = M[i], rhi = rhs ? rhs[i] : 0.0;
        double acc = op.a(0, i) * xi;
#pragma unroll
        for (int s = 1; s < W; ++s) acc = fma(op.a(s, i), x[col_of<IMP>(cols, n, Nw, mask, s, i)], acc);
        double r = -acc + rhi;
        rdu[i] = r;
        ulow[i] = xi;               // stable home of u_L for the flux/limit kernels
        y1[i] = omega1 * (r / (md_scale * mdi));
    }
}

template <int WT, int BS, int IMP>
__global__ void __launch_bounds__(BS) k_dudt_rhs(int n, int Wrt, int Nw, const int32_t* __restrict__ cols, MatRef A_ref,
                           VecRef rhs_ref, int64_t rhs_bstride, const double* __restrict__ M,
                           const double* __restrict__ xa_, const double* __restrict__ xb_,
                           double* __restrict__ ulow_, double* __restrict__ rdu_, double* __restrict__ y1_,
                           double* __restrict__ part, StepCtl* __restrict__ ctl_, int budget, int part_count,
                           int iters_per_unit, double rel_tol, double md_scale, double omega1,
                           const double* __restrict__ partk, int exact_k) {
    dudt_rhs_body<WT, BS, IMP>(n, Wrt, Nw, cols, MatOp{mat_ptr(A_ref, blockIdx.y), n}, rhs_ref, rhs_bstride, M, xa_, xb_,
                               ulow_, rdu_, y1_, part, ctl_, budget, part_count, iters_per_unit, rel_tol, md_scale, omega1,
                               partk, exact_k);
}

template <int BS, bool ROTG>
__global__ void __launch_bounds__(BS) k_dudt_rhs_sb(int n, int Nw, double h, SbOpArgs sb,
                           VecRef rhs_ref, int64_t rhs_bstride, const double* __restrict__ M,
                           const double* __restrict__ xa_, const double* __restrict__ xb_,
                           double* __restrict__ ulow_, double* __restrict__ rdu_, double* __restrict__ y1_,
                           double* __restrict__ part, StepCtl* __restrict__ ctl_, int budget, int part_count,
                           int iters_per_unit, double rel_tol, double md_scale, double omega1,
                           const double* __restrict__ partk, int exact_k) {
    SbOp<false, ROTG> op;
    op.p = sb; op.c = vec_ptr(sb.c) + blockIdx.y * sb.c_bstride; op.n = n; op.Nw = Nw; op.h = h;
    dudt_rhs_body<7, BS, 1>(n, 7, Nw, nullptr, op, rhs_ref, rhs_bstride, M, xa_, xb_, ulow_, rdu_, y1_, part, ctl_, budget,
                            part_count, iters_per_unit, rel_tol, md_scale, omega1, partk, exact_k);
}

// ---------------------------------------------------------------------------
// k_cheb: one Chebyshev semi-iteration step (helpers.py:175-184)
//   r = b - M y_mid ; z = r / Md ; y_new = omega (z + y_mid - y_old) + y_old
// ---------------------------------------------------------------------------
template <int WT, int BS, int IMP>
__global__ void __launch_bounds__(BS) k_cheb(int n, int Wrt, int Nw, const int32_t* __restrict__ cols, const double* __restrict__ M,
                       const double* __restrict__ b_, const double* __restrict__ ymid_,
                       const double* __restrict__ yold_, double* __restrict__ ynew_, double omega,
                       double md_scale, const double* __restrict__ mdv) {
    // mdv: the caller's own preconditioner diagonal Md (ChebSI's third argument) when it is not diag(M)
    const int W = WT ? WT : Wrt;
    const int64_t voff = (int64_t)blockIdx.y * n;
    const double* b = b_ + voff;
    const double* ymid = ymid_ ? ymid_ + voff : nullptr;   // null: y_mid = 0 (first iterate)
    const double* yold = yold_ ? yold_ + voff : nullptr;   // null: y_old = 0
    double* ynew = ynew_ + voff;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        const unsigned mask = nb_mask<IMP>(i, Nw);
        double md = M[i];
        double ym = ymid ? ymid[i] : 0.0;
        double acc = md * ym;
        if (ymid) {
#pragma unroll
            for (int s = 1; s < W; ++s) {
                int64_t idx = (int64_t)s * n + i;
                acc = fma(M[idx], ymid[col_of<IMP>(cols, n, Nw, mask, s, i)], acc);
            }
        }
        double r = b[i] - acc;
        double z = r / (md_scale * (mdv ? mdv[i] : md));
        double yo = yold ? yold[i] : 0.0;
        ynew[i] = omega * (z + ym - yo) + yo;
    }
}

// ---------------------------------------------------------------------------
// k_flux: raw antidiffusive fluxes + Zalesak P/Q/R (helpers.py:1818-1851)
// ---------------------------------------------------------------------------
template <int WT, int BS, int IMP>
__global__ void __launch_bounds__(BS) k_flux(int n, int Wrt, int Nw, const int32_t* __restrict__ cols, const double* __restrict__ M,
                       const double* __restrict__ D_, const double* __restrict__ ulow_,
                       const double* __restrict__ du_, const double* __restrict__ ml, double dt,
                       double* __restrict__ F_, double* __restrict__ rp_, double* __restrict__ rm_) {
    const int W = WT ? WT : Wrt;
    const int bz = blockIdx.y;
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* D = D_ + moff;
    const double* ulow = ulow_ + voff;
    const double* du = du_ + voff;
    double* F = F_ + moff;
    double* rp = rp_ + voff;
    double* rm = rm_ + voff;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        const unsigned mask = nb_mask<IMP>(i, Nw);
        double ui = ulow[i], dui = du[i];
        double pp = 0.0, pm = 0.0, umax = ui, umin = ui;
#pragma unroll
        for (int s = 1; s < W; ++s) {
            int64_t idx = (int64_t)s * n + i;
            int j = col_of<IMP>(cols, n, Nw, mask, s, i);
            double uj = ulow[j];
            double f = M[idx] * (dui - du[j]) + D[idx] * (ui - uj);
            F[idx] = f;
            pp += fmax(f, 0.0);
            pm += fmin(f, 0.0);
            umax = fmax(umax, uj);
            umin = fmin(umin, uj);
        }
        double mli = ml[i];
        double qp = umax - ui, qm = umin - ui;
        rp[i] = (pp != 0.0) ? fmin(1.0, mli * qp / (dt * pp)) : 1.0;
        rm[i] = (pm != 0.0) ? fmin(1.0, mli * qm / (dt * pm)) : 1.0;
    }
}

// ---------------------------------------------------------------------------
// k_limit: alpha_ij, limited flux sum and explicit correction (helpers.py:1860-1870)
// ---------------------------------------------------------------------------
template <int WT, int BS, int IMP>
__global__ void __launch_bounds__(BS) k_limit(int n, int Wrt, int Nw, const int32_t* __restrict__ cols, const double* __restrict__ F_,
                        const double* __restrict__ rp_, const double* __restrict__ rm_,
                        const double* __restrict__ ulow_, const double* __restrict__ ml, double dt,
                        VecRef out_ref, int64_t out_bstride) {
    const int W = WT ? WT : Wrt;
    const int bz = blockIdx.y;
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* F = F_ + moff;
    const double* rp = rp_ + voff;
    const double* rm = rm_ + voff;
    const double* ulow = ulow_ + voff;
    double* out = const_cast<double*>(vec_ptr(out_ref)) + bz * out_bstride;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        const unsigned mask = nb_mask<IMP>(i, Nw);
        double rpi = rp[i], rmi = rm[i];
        double fbar = 0.0;
#pragma unroll
        for (int s = 1; s < W; ++s) {
            int64_t idx = (int64_t)s * n + i;
            int j = col_of<IMP>(cols, n, Nw, mask, s, i);
            double f = F[idx];
            double a = (f > 0.0) ? fmin(rpi, rm[j]) : fmin(rmi, rp[j]);
            fbar += a * f;
        }
        out[i] = ulow[i] + dt * fbar / ml[i];
    }
}

// ---------------------------------------------------------------------------
// stand-alone helpers: artificial diffusion only, SpMV
// ---------------------------------------------------------------------------
__global__ void k_artdiff(int n, int W, const int32_t* __restrict__ cols, const uint8_t* __restrict__ tslot,
                          const double* __restrict__ K_, double* __restrict__ D_) {
    const int64_t moff = (int64_t)blockIdx.y * W * n;
    const double* K = K_ + moff;
    double* D = D_ + moff;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        double dsum = 0.0;
        for (int s = 1; s < W; ++s) {
            int64_t idx = (int64_t)s * n + i;
            int j = cols[idx];
            double k = K[idx], kt = K[(int64_t)tslot[idx] * n + j];
            double d = (j != i) ? fmax(0.0, fmax(-k, -kt)) : 0.0;
            D[idx] = d;
            dsum += d;
        }
        D[i] = -dsum;
    }
}

__global__ void k_spmv(int n, int W, const int32_t* __restrict__ cols, const double* __restrict__ A_,
                       const double* __restrict__ x_, double alpha, double beta, double* __restrict__ y_) {
    const int64_t moff = (int64_t)blockIdx.y * W * n, voff = (int64_t)blockIdx.y * n;
    const double* A = A_ + moff;
    const double* x = x_ + voff;
    double* y = y_ + voff;
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        double acc = A[i] * x[i];
        for (int s = 1; s < W; ++s) {
            int64_t idx = (int64_t)s * n + i;
            acc += A[idx] * x[cols[idx]];
        }
        y[i] = (beta != 0.0) ? alpha * acc + beta * y[i] : alpha * acc;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
LaunchGeom femfct_geom(const femfct_ctx* ctx, int32_t batch) {
    int bs = (ctx->n <= 65536) ? 64 : 256;
    if (const char* e = getenv("FEMFCT_BLOCK")) { int v = atoi(e); if (v == 64 || v == 256) bs = v; }  // tuning knob
    int64_t g = ((int64_t)ctx->n + bs - 1) / bs;
    if (g > FEMFCT_MAX_PARTIALS) g = FEMFCT_MAX_PARTIALS;
    if (g < 1) g = 1;
    return LaunchGeom{dim3((unsigned)g, (unsigned)batch, 1), dim3((unsigned)bs, 1, 1)};
}

// Chebyshev weights of helpers.py:170-179 (omega_1 from the else-branch with omega = 0).
static void cheb_omegas(int iters, double lmin, double lmax, std::vector<double>& om) {
    double rho = (lmax - lmin) / (lmax + lmin);
    double w = 0.0;
    om.resize(iters);
    for (int k = 1; k <= iters; ++k) {
        if (k == 2) w = 1.0 / (1.0 - rho * rho / 2.0);
        else w = 1.0 / (1.0 - (w * rho * rho) / 4.0);
        om[k - 1] = w;
    }
}

// BS is a template parameter (64 for small meshes, 256 otherwise) so that profiles list the
// latency-regime and the bandwidth-regime instances of a kernel under different names.
#define LAUNCH_W(cls, kern, geom, stream, ...)                                                          \
    do {                                                                                                \
        femfct_prof_begin(ctx, cls);                                                                    \
        if (ctx->W == 7 && ctx->implicit_cols) {                                                        \
            if (geom.block.x == 64) hipLaunchKernelGGL((kern<7, 64, 1>), geom.grid, geom.block, 0, stream, __VA_ARGS__);  \
            else hipLaunchKernelGGL((kern<7, 256, 1>), geom.grid, geom.block, 0, stream, __VA_ARGS__);   \
        } else if (ctx->W == 7) {                                                                       \
            if (geom.block.x == 64) hipLaunchKernelGGL((kern<7, 64, 0>), geom.grid, geom.block, 0, stream, __VA_ARGS__);  \
            else hipLaunchKernelGGL((kern<7, 256, 0>), geom.grid, geom.block, 0, stream, __VA_ARGS__);   \
        } else {                                                                                        \
            if (geom.block.x == 64) hipLaunchKernelGGL((kern<0, 64, 0>), geom.grid, geom.block, 0, stream, __VA_ARGS__);  \
            else hipLaunchKernelGGL((kern<0, 256, 0>), geom.grid, geom.block, 0, stream, __VA_ARGS__);   \
        }                                                                                               \
        femfct_prof_end(ctx);                                                                           \
    } while (0)

int femfct_enqueue_cheb(femfct_ctx* ctx, const double* b, double* y_out, int iters, double lmin, double lmax,
                        int32_t batch, bool first_done_in_y1, const double* mdv = nullptr) {
    // rotating buffers y0,y1,y2; iterate k reads mid=buf[(k-1)%3], old=buf[(k-2)%3], writes buf[k%3]
    LaunchGeom g = femfct_geom(ctx, batch);
    std::vector<double> om;
    cheb_omegas(iters, lmin, lmax, om);
    const double md_scale = (lmin + lmax) / 2.0;
    if (mdv) {   // a preconditioner diagonal of the caller's: the one-sweep row kernels take it as a vector
        double* bufv[3] = {ctx->d_y0, ctx->d_y1, ctx->d_y2};
        for (int k = 1; k <= iters; ++k) {
            const double* mid = (k >= 2) ? bufv[(k - 1) % 3] : nullptr;
            const double* old = (k >= 3) ? bufv[(k - 2) % 3] : nullptr;
            double* out = (k == iters) ? y_out : bufv[k % 3];
            LAUNCH_W(KC_CHEB, k_cheb, g, ctx->stream, ctx->n, ctx->W, ctx->N, ctx->d_cols, ctx->d_M, b, mid, old, out, om[k - 1],
                     md_scale, mdv);
        }
        return FEMFCT_OK;
    }
    if (femfct_tile4_wanted(ctx, batch)) {
        if (first_done_in_y1)
            return femfct_enqueue_tile4_cheb(ctx, b, ctx->d_y1, nullptr, y_out, 2, iters, om.data(), md_scale, ctx->d_y0,
                                             ctx->d_y2, ctx->d_y1, ctx->d_rp, batch);
        return femfct_enqueue_tile4_cheb(ctx, b, nullptr, nullptr, y_out, 1, iters, om.data(), md_scale, ctx->d_y0,
                                         ctx->d_y2, ctx->d_y1, ctx->d_rp, batch);
    }
    TilePlan tp;
    if (femfct_tile_plan(ctx, &tp, false)) {
        if (first_done_in_y1)
            return femfct_enqueue_tile_cheb(ctx, tp, b, ctx->d_y1, nullptr, y_out, 2, iters, om.data(), md_scale,
                                            ctx->d_y0, ctx->d_y2, ctx->d_y1, ctx->d_rp, batch);
        return femfct_enqueue_tile_cheb(ctx, tp, b, nullptr, nullptr, y_out, 1, iters, om.data(), md_scale, ctx->d_y0,
                                        ctx->d_y2, ctx->d_y1, ctx->d_rp, batch);
    }
    StripPlan pl;
    if (femfct_strip_plan(ctx, &pl)) {
        // y_1 (if already produced by k_dudt_rhs) lives in d_y1; pairs (y0,y2) / (y1,rp) alternate as scratch
        if (first_done_in_y1)
            return femfct_enqueue_strip_cheb(ctx, pl, b, ctx->d_y1, nullptr, y_out, 2, iters, om.data(), md_scale,
                                             ctx->d_y0, ctx->d_y2, ctx->d_y1, ctx->d_rp, batch);
        return femfct_enqueue_strip_cheb(ctx, pl, b, nullptr, nullptr, y_out, 1, iters, om.data(), md_scale, ctx->d_y0,
                                         ctx->d_y2, ctx->d_y1, ctx->d_rp, batch);
    }
    double* buf[3] = {ctx->d_y0, ctx->d_y1, ctx->d_y2};
    int n = ctx->n, W = ctx->W;
    for (int k = 1; k <= iters; ++k) {
        if (k == 1 && first_done_in_y1) continue;  // produced by k_dudt_rhs into buf[1]
        const double* mid = (k >= 2) ? buf[(k - 1) % 3] : nullptr;
        const double* old = (k >= 3) ? buf[(k - 2) % 3] : nullptr;
        double* out = (k == iters) ? y_out : buf[k % 3];
        LAUNCH_W(KC_CHEB, k_cheb, g, ctx->stream, n, W, ctx->N, ctx->d_cols, ctx->d_M, b, mid, old, out, om[k - 1], md_scale,
                 (const double*)nullptr);
    }
    return FEMFCT_OK;
}

// Enqueue the whole step with level-indirected in/out vectors (trajectory drivers) --------
int femfct_enqueue_step_mat(femfct_ctx* ctx, MatRef A, const double* N, int32_t nshared, VecRef rhs, int64_t rhs_bstride,
                            VecRef u_n, int64_t u_bstride, double dt, VecRef u_out, int64_t out_bstride, int32_t batch,
                            int32_t budget);

int femfct_enqueue_step_ref(femfct_ctx* ctx, const double* A, const double* N, int32_t nshared, VecRef rhs,
                            int64_t rhs_bstride, VecRef u_n, int64_t u_bstride, double dt, VecRef u_out,
                            int64_t out_bstride, int32_t batch, int32_t budget) {
    return femfct_enqueue_step_mat(ctx, MatRef{A, nullptr, 0, 0, (int64_t)ctx->W * ctx->n}, N, nshared, rhs, rhs_bstride, u_n,
                                   u_bstride, dt, u_out, out_bstride, batch, budget);
}

int femfct_enqueue_step_op(femfct_ctx* ctx, MatRef A, const SbOpArgs* sb, const double* N, int32_t nshared, VecRef rhs,
                           int64_t rhs_bstride, VecRef u_n, int64_t u_bstride, double dt, VecRef u_out,
                           int64_t out_bstride, int32_t batch, int32_t budget);

// the drift-control operator may be derived inside the step kernels instead of being passed as a matrix
bool femfct_inline_ops_wanted(const femfct_ctx* ctx, int32_t batch) {
    if (femfct_mesh_step_wanted(ctx, batch)) return false;   // the one-workgroup step takes a stored operator
    return ctx->inline_ops && ctx->structured && ctx->implicit_cols && ctx->W == 7 && femfct_tile4_wanted(ctx, batch);
}

int femfct_enqueue_step_mat(femfct_ctx* ctx, MatRef A, const double* N, int32_t nshared, VecRef rhs,
                            int64_t rhs_bstride, VecRef u_n, int64_t u_bstride, double dt, VecRef u_out,
                            int64_t out_bstride, int32_t batch, int32_t budget) {
    return femfct_enqueue_step_op(ctx, A, nullptr, N, nshared, rhs, rhs_bstride, u_n, u_bstride, dt, u_out, out_bstride, batch,
                                  budget);
}

// A: the flux matrix of this step, possibly one of a pre-assembled per-level sequence (MatRef); or sb != null
// (only with femfct_inline_ops_wanted): the solid-body operator, derived in k_build_low_sb / k_dudt_rhs_sb
int femfct_enqueue_step_op(femfct_ctx* ctx, MatRef A, const SbOpArgs* sb, const double* N, int32_t nshared, VecRef rhs,
                           int64_t rhs_bstride, VecRef u_n, int64_t u_bstride, double dt, VecRef u_out,
                           int64_t out_bstride, int32_t batch, int32_t budget) {
    if (sb && (!femfct_inline_ops_wanted(ctx, batch) || N))
        return femfct_fail(ctx, FEMFCT_ERR_INVALID, "inline solid-body operator outside its regime");
    if (!sb && femfct_mesh_step_wanted(ctx, batch, N != nullptr)) {
        // config-sized mesh: the whole step (and its end) in one launch, one workgroup per trajectory
        const bool fuse_end = ctx->end_req_delta != 0 && ctx->d_ticket && ctx->d_level && ctx->d_log && !ctx->prof_on;
        int r = femfct_enqueue_mesh_step(ctx, A, N, nshared, rhs, rhs_bstride, u_n, u_bstride, dt, u_out, out_bstride, batch,
                                         budget, fuse_end);
        if (r == FEMFCT_OK && fuse_end) ctx->end_fused = true;
        return r;
    }
    LaunchGeom g = femfct_geom(ctx, batch);
    hipStream_t st = ctx->stream;
    int n = ctx->n, W = ctx->W;
    StripPlan pl;
    TilePlan tp;
    // Jacobi tiles: small grids reduce the residual partials in the consumer; large grids use the
    // extra reduce kernel (needs the big partial buffer, allocated by femfct_ensure_workspace)
    const bool tiles = femfct_tile_plan(ctx, &tp, false, budget, batch) && (!femfct_tile_big(ctx, tp) || ctx->d_bigpart);
    const bool strips = !tiles && femfct_strip_plan(ctx, &pl);
    int units = budget, part_count = 0, ipu = 1, exact_k = 0;
    const bool tile4 = tiles && femfct_tile4_wanted(ctx, batch);
    // latency regime: the operator construction rides in the first tile-Jacobi launch
    const bool fused_build = ctx->fuse_build && tiles && !tile4 && !femfct_tile_big(ctx, tp) &&
                             ctx->solver != FEMFCT_SOLVER_BICGSTAB && (budget + tp.K - 1) / tp.K >= 2;
    uint8_t* lmask = (tile4 && ctx->l_mask && ctx->t4_dpp && W == 7 && ctx->solver == FEMFCT_SOLVER_JACOBI)
                         ? reinterpret_cast<uint8_t*>(ctx->d_Lmask + 1) : nullptr;
    // symmetric storage of D between k_build_low and the 2-D tile limiter (both sides of this step or neither)
    const int half_d = (tile4 && ctx->half_d && ctx->structured && ctx->implicit_cols && W == 7) ? 1 : 0;
    if (sb) {
        femfct_prof_begin(ctx, KC_BUILD_LOW);
        if (sb->rot_geom)
            hipLaunchKernelGGL((k_build_low_sb<256, true>), g.grid, dim3(256), 0, st, n, ctx->N, ctx->h, *sb, rhs, u_n, rhs_bstride,
                               u_bstride, ctx->d_ml, dt, ctx->d_L, ctx->d_D, ctx->d_b, ctx->d_xa, ctx->d_part, ctx->d_ctl, lmask, half_d);
        else
            hipLaunchKernelGGL((k_build_low_sb<256, false>), g.grid, dim3(256), 0, st, n, ctx->N, ctx->h, *sb, rhs, u_n, rhs_bstride,
                               u_bstride, ctx->d_ml, dt, ctx->d_L, ctx->d_D, ctx->d_b, ctx->d_xa, ctx->d_part, ctx->d_ctl, lmask, half_d);
        femfct_prof_end(ctx);
    } else if (!fused_build)
        LAUNCH_W(KC_BUILD_LOW, k_build_low, g, st, n, W, ctx->N, ctx->d_cols, ctx->d_tslot, A, N, nshared, rhs, u_n,
                 rhs_bstride, u_bstride, ctx->d_ml, dt, ctx->d_L, ctx->d_D, ctx->d_b, ctx->d_xa, ctx->d_part, ctx->d_ctl,
                 lmask, half_d);
    if (ctx->solver == FEMFCT_SOLVER_BICGSTAB) {
        // robust alternative for operators far from diagonal dominance: Jacobi-preconditioned BiCGStab
        // from x0 = u^n into d_xa; outcome mirrored into StepCtl (done, parity 0)
        int r = femfct_enqueue_bicgstab(ctx, ctx->d_L, 0, ctx->d_b, make_ref(ctx->d_xa), n, make_ref(ctx->d_xa), n, batch,
                                        budget, true);
        if (r != FEMFCT_OK) return r;
        femfct_enqueue_kry_to_stepctl(ctx, (int)g.grid.x, batch);
        units = 0;
    } else if (tile4) {
        const bool single = femfct_single_patch(ctx, batch);    // (reached only with FEMFCT_TILE4=2 on such meshes)
        const int h4 = femfct_tile4_halo(ctx, budget);
        const int t4 = femfct_tile4_tiles(ctx, h4);
        const bool big4 = (int64_t)t4 * t4 > FEMFCT_MAX_PARTIALS;
        // sweeps per launch: the halo depth; a single patch runs the whole budget in one launch and stops by itself
        const int k4 = single ? budget : std::min(ctx->t4_k == 8 ? h4 : ctx->t4_k, h4);
        units = (budget + k4 - 1) / k4;
        const bool pair4 = !single && femfct_jacobi_pair_wanted(ctx, h4, batch, lmask != nullptr);
        const int walkers = single ? 0 : femfct_tile4_walkers(ctx, h4, batch, pair4);
        part_count = walkers > 0 ? walkers : big4 ? -1 : t4 * t4;
        ipu = k4;
        for (int s = 0; s < units; ++s)
            femfct_enqueue_tile4_jacobi(ctx, ctx->d_L, ctx->d_b, ctx->d_xa, ctx->d_xb, s, (int)g.grid.x, batch, h4, k4,
                                        single ? 2 : 0, lmask);
    } else if (tiles) {
        units = (budget + tp.K - 1) / tp.K;
        part_count = femfct_tile_big(ctx, tp) ? -1 : tp.tiles * tp.tiles;
        ipu = tp.K;
        exact_k = (femfct_tile_big(ctx, tp) || !ctx->exact_iters) ? 0 : tp.K;
        // two to four launches, the first of which builds the operator, and the fused du/dt kernel behind them: the
        // residual tests move out of the later launches into that kernel (exact_k = -1; solve_ctl.h: deferred_test_*)
        const bool defer = ctx->defer_check && fused_build && units >= 2 && units <= 4 && exact_k == 0 && ctx->fuse_dudt;
        if (defer) exact_k = -units;
        if (fused_build)
            femfct_enqueue_tile_build_jacobi(ctx, tp, A, N, nshared, rhs, rhs_bstride, u_n, u_bstride, dt, batch);
        for (int s = fused_build ? 1 : 0; s < units; ++s)
            femfct_enqueue_tile_jacobi(ctx, tp, ctx->d_L, ctx->d_b, ctx->d_xa, ctx->d_xb, s,
                                       fused_build ? tp.tiles * tp.tiles : (int)g.grid.x, batch,
                                       exact_k > 0 && s == units - 1, fused_build ? 1 : 0, defer ? 1 : 0);
    } else if (strips) {
        units = (budget + pl.K - 1) / pl.K;
        part_count = pl.S;
        ipu = pl.K;
        for (int s = 0; s < units; ++s)
            femfct_enqueue_strip_jacobi(ctx, pl, ctx->d_L, ctx->d_b, ctx->d_xa, ctx->d_xb, s, (int)g.grid.x, batch);
    } else {
        for (int s = 0; s < budget; ++s)
            LAUNCH_W(KC_JACOBI, k_jacobi, g, st, n, W, ctx->N, ctx->d_cols, ctx->d_L, ctx->d_b, ctx->d_xa, ctx->d_xb,
                     ctx->d_part, ctx->d_ctl, s, ctx->rel_tol);
    }
    // stable home of u_L for the flux/limit kernels: d_b (the low-order rhs is dead after the solve)
    double* ulow = ctx->d_b;
    bool step_done = false;
    if (tiles && !tile4 && !femfct_tile_big(ctx, tp) && ctx->fuse_dudt) {
        std::vector<double> om;
        cheb_omegas(20, 0.5, 2.0, om);
        int tail = 0;
        const bool fuse_tail = femfct_cheb_flux_fusable(ctx, batch);
        femfct_enqueue_tile_dudt_cheb(ctx, A, rhs, rhs_bstride, ulow, units, part_count, ipu, exact_k, 20, om.data(), 1.25,
                                      batch, fuse_tail ? &tail : nullptr);
        if (fuse_tail && tail > 0) {
            // iterations tail..20 of du/dt + flux + limiter (+ step end) in one launch
            const bool fuse_end = ctx->end_req_delta != 0 && ctx->d_ticket && ctx->d_level && ctx->d_log && !ctx->prof_on;
            int r = femfct_enqueue_tile_cheb_flux_limit(ctx, ctx->d_rdu, ctx->d_y0, ctx->d_y2, tail, 20, om.data(), 1.25,
                                                        ctx->d_D, ulow, dt, u_out, out_bstride, batch, fuse_end);
            if (r != FEMFCT_OK) return r;
            if (fuse_end) ctx->end_fused = true;
            step_done = true;
        } else if (fuse_tail) {
            return femfct_fail(ctx, FEMFCT_ERR_INVALID, "fused Chebyshev tail: nothing left to fuse");
        }
    } else {
        if (sb) {
            femfct_prof_begin(ctx, KC_DUDT_RHS);
            if (sb->rot_geom)
                hipLaunchKernelGGL((k_dudt_rhs_sb<256, true>), g.grid, dim3(256), 0, st, n, ctx->N, ctx->h, *sb, rhs, rhs_bstride,
                                   ctx->d_M, ctx->d_xa, ctx->d_xb, ulow, ctx->d_rdu, ctx->d_y1, ctx->d_part, ctx->d_ctl, units,
                                   part_count, ipu, ctx->rel_tol, 1.25, 1.0, (const double*)nullptr, 0);
            else
                hipLaunchKernelGGL((k_dudt_rhs_sb<256, false>), g.grid, dim3(256), 0, st, n, ctx->N, ctx->h, *sb, rhs, rhs_bstride,
                                   ctx->d_M, ctx->d_xa, ctx->d_xb, ulow, ctx->d_rdu, ctx->d_y1, ctx->d_part, ctx->d_ctl, units,
                                   part_count, ipu, ctx->rel_tol, 1.25, 1.0, (const double*)nullptr, 0);
            femfct_prof_end(ctx);
        } else {
            LAUNCH_W(KC_DUDT_RHS, k_dudt_rhs, g, st, n, W, ctx->N, ctx->d_cols, A, rhs, rhs_bstride, ctx->d_M, ctx->d_xa,
                     ctx->d_xb, ulow, ctx->d_rdu, ctx->d_y1, ctx->d_part, ctx->d_ctl, units, part_count, ipu, ctx->rel_tol,
                     1.25, 1.0, exact_k ? ctx->d_partk : nullptr, exact_k);
        }
        femfct_enqueue_cheb(ctx, ctx->d_rdu, ctx->d_du, 20, 0.5, 2.0, batch, true);
    }
    if (step_done) {
        // limiter already ran inside the fused tail
    } else if (tiles) {
        bool fuse_end = ctx->end_req_delta != 0 && ctx->d_ticket && ctx->d_level && ctx->d_log && !ctx->prof_on;
        femfct_enqueue_tile_flux_limit(ctx, ctx->d_D, ulow, ctx->d_du, dt, u_out, out_bstride, batch, &fuse_end, half_d);
        if (fuse_end) ctx->end_fused = true;
    } else {
        LAUNCH_W(KC_FLUX, k_flux, g, st, n, W, ctx->N, ctx->d_cols, ctx->d_M, ctx->d_D, ulow, ctx->d_du, ctx->d_ml, dt,
                 ctx->d_F, ctx->d_rp, ctx->d_rm);
        LAUNCH_W(KC_LIMIT, k_limit, g, st, n, W, ctx->N, ctx->d_cols, ctx->d_F, ctx->d_rp, ctx->d_rm, ulow, ctx->d_ml, dt,
                 u_out, out_bstride);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return femfct_fail(ctx, FEMFCT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return FEMFCT_OK;
}

bool femfct_jacobi_plan(const femfct_ctx* ctx, int budget, int batch, int* K, int* launches) {
    TilePlan tp;
    if (!femfct_tile_plan(ctx, &tp, false, budget, batch)) return false;
    if (femfct_tile4_wanted(ctx, batch) && femfct_single_patch(ctx, batch)) {
        *K = budget;
    } else if (femfct_tile4_wanted(ctx, batch)) {
        const int h4 = femfct_tile4_halo(ctx, budget);
        *K = std::min(ctx->t4_k == 8 ? h4 : ctx->t4_k, h4);
    } else {
        *K = tp.K;
    }
    *launches = (budget + *K - 1) / *K;
    return true;
}

int femfct_enqueue_step(femfct_ctx* ctx, const double* A, const double* N, int32_t nshared, const double* rhs,
                        const double* u_n, double dt, double* u_out, int32_t batch, int32_t budget) {
    return femfct_enqueue_step_ref(ctx, A, N, nshared, make_ref(rhs), ctx->n, make_ref(u_n), ctx->n, dt,
                                   make_ref(u_out), ctx->n, batch, budget);
}

int femfct_enqueue_artdiff(femfct_ctx* ctx, const double* K, double* D, int32_t batch) {
    LaunchGeom g = femfct_geom(ctx, batch);
    hipLaunchKernelGGL(k_artdiff, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->W, ctx->d_cols, ctx->d_tslot, K, D);
    return FEMFCT_OK;
}

int femfct_enqueue_spmv(femfct_ctx* ctx, const double* A, const double* x, double alpha, double beta, double* y,
                        int32_t batch) {
    LaunchGeom g = femfct_geom(ctx, batch);
    hipLaunchKernelGGL(k_spmv, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->W, ctx->d_cols, A, x, alpha, beta, y);
    return FEMFCT_OK;
}
