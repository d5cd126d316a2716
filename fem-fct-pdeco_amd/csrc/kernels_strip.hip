// Strip-fused multi-sweep kernels (temporal blocking) for the latency regime.
//
// On the config meshes (n = 1681 / 6561) a Jacobi sweep or a Chebyshev step moves < 1 MB and costs
// one dependent kernel boundary (~2.7 us measured) -- the step is launch-latency bound, not
// bandwidth bound.  These kernels run K sweeps per launch: a 1024-thread workgroup owns R
// consecutive rows, stages the iterate for its rows plus a halo of K*bw rows (bw = matrix
// bandwidth, N+1 on the structured mesh in either DoF order) in LDS, keeps its matrix rows
// (values + LDS-local column offsets) in registers, and sweeps K times with __syncthreads()
// between sweeps; the region of valid rows shrinks by bw per sweep and still covers the owned rows
// at the end.  Arithmetic per row is identical to the one-sweep kernels (same operation order),
// nothing is exchanged between workgroups inside a launch, no atomics: deterministic.
// Works for any ELL pattern whose bandwidth admits K >= 2 within the LDS/register budget.
#include "femfct_internal.h"
#include "device_utils.h"
#include "solve_ctl.h"
#include "forms.h"
#include "step_end.h"

#include <math.h>
#include <type_traits>

#define STRIP_T 1024

namespace {

struct CheOmegas { double w[24]; };   // omegas of one launch (up to TILE_HMAX iterations; all 19 of ChebSI on a single patch)

template <int RPT>
__global__ void __launch_bounds__(STRIP_T)
k_strip_jacobi(int n, const int32_t* __restrict__ cols, const double* __restrict__ L_, const double* __restrict__ b_,
               double* __restrict__ xa_, double* __restrict__ xb_, double* __restrict__ part,
               StepCtl* __restrict__ ctl_, int launch, int K, int bw, int R, int g_build, double rel_tol) {
    constexpr int W = 7, EXT = RPT * STRIP_T;
    extern __shared__ double lds[];
    __shared__ double smem[32];
    const int bz = blockIdx.y;
    StepCtl* ctl = ctl_ + bz;
    if (ctl->done) return;
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    double bnorm;
    if (launch == 0) {
        bnorm = reduce_partials(p + 2 * FEMFCT_MAX_PARTIALS, g_build, OpMax(), 0.0, smem);
        double rsmin = reduce_partials(p + 3 * FEMFCT_MAX_PARTIALS, g_build, OpMin(), INFINITY, smem);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ctl->bnorm = bnorm;
            ctl->min_rowsum = rsmin;
            if (!(rsmin > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        }
    } else {
        bnorm = ctl->bnorm;
        double rmax = reduce_partials(p + ((launch - 1) & 1) * FEMFCT_MAX_PARTIALS, gridDim.x, OpMax(), 0.0, smem);
        if (rmax <= rel_tol * bnorm) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                ctl->done = 1; ctl->parity = launch & 1; ctl->iters = launch * K; ctl->flags |= FEMFCT_FLAG_COARSE_ITERS;
                ctl->resid = bnorm > 0.0 ? rmax / bnorm : 0.0;
            }
            return;
        }
    }
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* L = L_ + moff;
    const double* b = b_ + voff;
    const double* xin = ((launch & 1) ? xb_ : xa_) + voff;
    double* xout = ((launch & 1) ? xa_ : xb_) + voff;

    const int r0 = blockIdx.x * R, r1 = min(n, r0 + R);
    const int e0 = max(0, r0 - K * bw), e1 = min(n, r1 + K * bw), ext = e1 - e0;
    double lv[RPT][W];       // lv[r][0] holds 1 / L_ii after loading
    double dg[RPT];
    int lc[RPT][W - 1];
    double bv[RPT];
    double* cur = lds;
    double* nxt = lds + EXT;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int li = threadIdx.x + r * STRIP_T;
        if (li < ext) {
            const int i = e0 + li;
            dg[r] = L[i];
            lv[r][0] = 1.0 / dg[r];
#pragma unroll
            for (int s = 1; s < W; ++s) {
                int64_t idx = (int64_t)s * n + i;
                lv[r][s] = L[idx];
                int c = cols[idx] - e0;
                lc[r][s - 1] = (c >= 0 && c < ext) ? c : li;
            }
            bv[r] = b[i];
            cur[li] = xin[i];
        }
    }
    __syncthreads();
    double rmax = 0.0;
    for (int k = 0; k < K; ++k) {
        const int lo = (e0 == 0) ? 0 : (k + 1) * bw;
        const int hi = (e1 == n) ? ext : ext - (k + 1) * bw;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int li = threadIdx.x + r * STRIP_T;
            if (li < ext) {
                const double xi = cur[li];
                double xn = xi;
                if (li >= lo && li < hi) {
                    double acc = bv[r];
#pragma unroll
                    for (int s = 1; s < W; ++s) acc = fma(-lv[r][s], cur[lc[r][s - 1]], acc);
                    xn = acc * lv[r][0];          // one reciprocal per row and launch instead of K divisions
                    const int i = e0 + li;
                    if (k == K - 1 && i >= r0 && i < r1) rmax = fmax(rmax, fabs(acc - dg[r] * xi));
                }
                nxt[li] = xn;
            }
        }
        __syncthreads();
        double* t = cur; cur = nxt; nxt = t;
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int li = threadIdx.x + r * STRIP_T;
        const int i = e0 + li;
        if (li < ext && i >= r0 && i < r1) xout[i] = cur[li];
    }
    rmax = block_reduce(rmax, OpMax(), 0.0, smem);
    if (threadIdx.x == 0) p[(launch & 1) * FEMFCT_MAX_PARTIALS + blockIdx.x] = rmax;
}

// Chebyshev semi-iteration steps k0..k1-1 (1-based iteration numbers as in helpers.py:175):
//   y_k = w_k ( (b - M y_{k-1}) / Md + y_{k-1} - y_{k-2} ) + y_{k-2}
// in: y_{k0-1} (mid, null = 0), y_{k0-2} (old, null = 0); out: y_{k1-1} (mid) and y_{k1-2} (old, optional)
template <int RPT>
__global__ void __launch_bounds__(STRIP_T)
k_strip_cheb(int n, const int32_t* __restrict__ cols, const double* __restrict__ M, const double* __restrict__ b_,
             const double* __restrict__ ymid_, const double* __restrict__ yold_, double* __restrict__ omid_,
             double* __restrict__ oold_, int k0, int k1, CheOmegas om, double md_scale, int bw, int R) {
    constexpr int W = 7, EXT = RPT * STRIP_T;
    extern __shared__ double lds[];
    const int64_t voff = (int64_t)blockIdx.y * n;
    const double* b = b_ + voff;
    const double* ymid = ymid_ ? ymid_ + voff : nullptr;
    const double* yold = yold_ ? yold_ + voff : nullptr;
    double* omid = omid_ + voff;
    double* oold = oold_ ? oold_ + voff : nullptr;
    const int K = k1 - k0;
    const int r0 = blockIdx.x * R, r1 = min(n, r0 + R);
    const int e0 = max(0, r0 - K * bw), e1 = min(n, r1 + K * bw), ext = e1 - e0;
    double mv[RPT][W];
    double rmd[RPT];         // 1 / (md_scale * M_ii)
    int lc[RPT][W - 1];
    double bv[RPT];
    double* y_old = lds;
    double* y_mid = lds + EXT;
    double* y_new = lds + 2 * EXT;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int li = threadIdx.x + r * STRIP_T;
        if (li < ext) {
            const int i = e0 + li;
            mv[r][0] = M[i];
#pragma unroll
            for (int s = 1; s < W; ++s) {
                int64_t idx = (int64_t)s * n + i;
                mv[r][s] = M[idx];
                int c = cols[idx] - e0;
                lc[r][s - 1] = (c >= 0 && c < ext) ? c : li;
            }
            bv[r] = b[i];
            rmd[r] = 1.0 / (md_scale * mv[r][0]);
            y_mid[li] = ymid ? ymid[i] : 0.0;
            y_old[li] = yold ? yold[i] : 0.0;
        }
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const int lo = (e0 == 0) ? 0 : (k + 1) * bw;
        const int hi = (e1 == n) ? ext : ext - (k + 1) * bw;
        const double omega = om.w[k];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int li = threadIdx.x + r * STRIP_T;
            if (li < ext) {
                const double ym = y_mid[li];
                double yn = ym;
                if (li >= lo && li < hi) {
                    double acc = mv[r][0] * ym;
#pragma unroll
                    for (int s = 1; s < W; ++s) acc = fma(mv[r][s], y_mid[lc[r][s - 1]], acc);
                    const double rr = bv[r] - acc;
                    const double z = rr * rmd[r];
                    const double yo = y_old[li];
                    yn = omega * (z + ym - yo) + yo;
                }
                y_new[li] = yn;
            }
        }
        __syncthreads();
        double* t = y_old; y_old = y_mid; y_mid = y_new; y_new = t;
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int li = threadIdx.x + r * STRIP_T;
        const int i = e0 + li;
        if (li < ext && i >= r0 && i < r1) {
            omid[i] = y_mid[li];
            if (oold) oold[i] = y_old[li];
        }
    }
}

template <int RPT>
void launch_jacobi(femfct_ctx* ctx, dim3 grid, size_t lds, const double* L, const double* b, double* xa, double* xb,
                   int launch, int K, int bw, int R, int g_build) {
    hipLaunchKernelGGL((k_strip_jacobi<RPT>), grid, dim3(STRIP_T), lds, ctx->stream, ctx->n, ctx->d_cols, L, b, xa, xb,
                       ctx->d_part, ctx->d_ctl, launch, K, bw, R, g_build, ctx->rel_tol);
}

template <int RPT>
void launch_cheb(femfct_ctx* ctx, dim3 grid, size_t lds, const double* b, const double* ymid, const double* yold,
                 double* omid, double* oold, int k0, int k1, const CheOmegas& om, double md_scale, int bw, int R) {
    hipLaunchKernelGGL((k_strip_cheb<RPT>), grid, dim3(STRIP_T), lds, ctx->stream, ctx->n, ctx->d_cols, ctx->d_M, b, ymid,
                       yold, omid, oold, k0, k1, om, md_scale, bw, R);
}

}  // namespace

// Plan: K sweeps per launch, R owned rows per workgroup, RPT rows per thread.  Returns false when
// the pattern's bandwidth leaves no room for K >= 2 (large meshes: the bandwidth-bound kernels win).
bool femfct_strip_plan(const femfct_ctx* ctx, StripPlan* pl) {
    if (!ctx->use_strips || ctx->W != 7 || ctx->bandwidth <= 0) return false;
    const int bw = ctx->bandwidth;
    int K = 1300 / bw;
    if (K > 8) K = 8;
    if (ctx->strip_k > 0) K = std::min(ctx->strip_k, 1800 / bw);   // tuning knob (FEMFCT_STRIP_K)
    if (K > 8) K = 8;
    if (K < 2) return false;
    const int halo = K * bw;
    // fewest rows per thread that still leave >= 256 owned rows, then the largest R for that
    int rpt = (2 * halo + 256 + STRIP_T - 1) / STRIP_T;
    if (rpt < 2) rpt = 2;
    if (rpt > 4) return false;
    int R = rpt * STRIP_T - 2 * halo;
    if (R > ctx->n) R = ctx->n;
    pl->K = K; pl->R = R; pl->bw = bw; pl->rpt = rpt;
    pl->S = (ctx->n + R - 1) / R;
    return true;
}

int femfct_strip_init(femfct_ctx* ctx) {
    // > 64 KB of dynamic LDS needs the attribute
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_strip_cheb<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * STRIP_T * 8));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_strip_cheb<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4 * STRIP_T * 8));
    return femfct_tile4_init(ctx);
}

int femfct_enqueue_strip_jacobi(femfct_ctx* ctx, const StripPlan& pl, const double* L, const double* b, double* xa,
                                double* xb, int launch, int g_build, int32_t batch) {
    dim3 grid(pl.S, batch, 1);
    size_t lds = (size_t)2 * pl.rpt * STRIP_T * 8;
    femfct_prof_begin(ctx, KC_JACOBI);
    switch (pl.rpt) {
        case 2: launch_jacobi<2>(ctx, grid, lds, L, b, xa, xb, launch, pl.K, pl.bw, pl.R, g_build); break;
        case 3: launch_jacobi<3>(ctx, grid, lds, L, b, xa, xb, launch, pl.K, pl.bw, pl.R, g_build); break;
        default: launch_jacobi<4>(ctx, grid, lds, L, b, xa, xb, launch, pl.K, pl.bw, pl.R, g_build); break;
    }
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

// Chebyshev iterations k_first..k_last (1-based, inclusive) in ceil(count/K) launches.
// in_mid = y_{k_first-1} (null = 0), in_old = y_{k_first-2} (null = 0); result y_{k_last} -> y_out.
// pairs (bufA0,bufA1)/(bufB0,bufB1) are alternated as intermediate (mid, old) storage.
int femfct_enqueue_strip_cheb(femfct_ctx* ctx, const StripPlan& pl, const double* b, const double* in_mid,
                              const double* in_old, double* y_out, int k_first, int k_last, const double* omegas,
                              double md_scale, double* bufA0, double* bufA1, double* bufB0, double* bufB1,
                              int32_t batch) {
    dim3 grid(pl.S, batch, 1);
    size_t lds = (size_t)3 * pl.rpt * STRIP_T * 8;
    const double* mid = in_mid;
    const double* old = in_old;
    int which = 0;
    for (int k0 = k_first; k0 <= k_last; k0 += pl.K) {
        int k1 = std::min(k_last + 1, k0 + pl.K);
        CheOmegas om;
        for (int k = k0; k < k1; ++k) om.w[k - k0] = omegas ? omegas[k - 1] : 0.0;
        const bool last = (k1 == k_last + 1);
        double* omid = last ? y_out : (which ? bufB0 : bufA0);
        double* oold = last ? nullptr : (which ? bufB1 : bufA1);
        femfct_prof_begin(ctx, KC_CHEB);
        switch (pl.rpt) {
            case 2: launch_cheb<2>(ctx, grid, lds, b, mid, old, omid, oold, k0, k1, om, md_scale, pl.bw, pl.R); break;
            case 3: launch_cheb<3>(ctx, grid, lds, b, mid, old, omid, oold, k0, k1, om, md_scale, pl.bw, pl.R); break;
            default: launch_cheb<4>(ctx, grid, lds, b, mid, old, omid, oold, k0, k1, om, md_scale, pl.bw, pl.R); break;
        }
        femfct_prof_end(ctx);
        mid = omid;
        old = oold;
        which ^= 1;
    }
    return FEMFCT_OK;
}

// ===========================================================================================
// 2-D tile variants for the structured mesh in vertex order: a 1024-thread workgroup stages a
// 32 x 32 patch = (32 - 2H)^2 tile + halo H on every side -- one node per thread, the node's
// matrix row in registers, the iterate in LDS -- and runs up to H sweeps per launch (the halo
// shrinks by one ring per sweep).  H = 8 on large grids (least re-reading), 8..13 in the latency
// regime (fewest launches: see femfct_tile_plan).  Compared with row strips there is no column
// table to load and the work spreads over (N / (32 - 2H))^2 workgroups.
// ===========================================================================================
#define TILE_T 16
#define TILE_H 8
#define TILE_L 32
#define TILE_LD 33   // padded LDS row
#define TILE_HMAX 13 // deepest halo instantiated (Jacobi, latency regime)

namespace {

struct TileGeom {
    int lx, ly, gx, gy, i;
    bool inside, owned;
    int kvalid;          // the node's value is exact for sweeps k < kvalid
    int nb[6];           // LDS offsets of the six neighbours (clamped into the patch)
    int self;
};

template <int T = TILE_T, int H = TILE_H, int PL = TILE_L>
__device__ __forceinline__ TileGeom tile_geom(int N) {
    static_assert(T + 2 * H == PL, "patch edge = tile + 2 halos");
    constexpr int PLD = PL + 1;
    TileGeom g;
    g.lx = threadIdx.x % PL;
    g.ly = threadIdx.x / PL;
    const int x0 = blockIdx.x * T - H, y0 = blockIdx.y * T - H;
    g.gx = x0 + g.lx;
    g.gy = y0 + g.ly;
    g.inside = g.gx >= 0 && g.gx < N && g.gy >= 0 && g.gy < N;
    g.i = g.inside ? g.gy * N + g.gx : 0;
    g.owned = g.inside && g.lx >= H && g.lx < H + T && g.ly >= H && g.ly < H + T;
    int kv = 1 << 20;
    if (x0 > 0) kv = min(kv, g.lx);
    if (x0 + PL - 1 < N - 1) kv = min(kv, PL - 1 - g.lx);
    if (y0 > 0) kv = min(kv, g.ly);
    if (y0 + PL - 1 < N - 1) kv = min(kv, PL - 1 - g.ly);
    g.kvalid = g.inside ? kv : 0;
    const int dx[6] = {1, 1, 0, -1, -1, 0}, dy[6] = {0, 1, 1, 0, -1, -1};
    g.self = g.ly * PLD + g.lx;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        int nx = min(max(g.lx + dx[s], 0), PL - 1), ny = min(max(g.ly + dy[s], 0), PL - 1);
        g.nb[s] = ny * PLD + nx;
    }
    return g;
}

// BIG = 1: grids with more workgroups than in-kernel partials (bandwidth regime); the residual maxima
// go through k_reduce_resid.  A template parameter so that profiles list the two regimes separately.
template <int H, int EXACT, int BIG>
__global__ void __launch_bounds__(STRIP_T)
k_tile_jacobi(int n, int N, const double* __restrict__ L_, const double* __restrict__ b_, double* __restrict__ xa_,
              double* __restrict__ xb_, double* __restrict__ part, StepCtl* __restrict__ ctl_, int launch, int K,
              int g_build, double rel_tol, double* __restrict__ bigpart, double* __restrict__ partk, int bn_launch,
              int defer) {
    constexpr int W = 7;
    __shared__ double xs[2][TILE_L * TILE_LD];
    __shared__ double smem[96];
    const int bz = blockIdx.z;
    StepCtl* ctl = ctl_ + bz;
    if (!defer && ctl->done) return;       // (defer: nothing sets `done` before this launch -- one dependent round trip less)
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
    double bnorm;
    // ||b||, min row sum: reduced from the partials of the kernel that built L (k_build_low before launch 0,
    // or the fused k_tile_build_jacobi = launch 0 itself, then bn_launch = 1); launch >= 1 also tests the
    // residual the previous launch left.  In the fused case all three come out of one reduction pass.
    double rmax_prev = 0.0;
    bool have_rmax = false;
    // defer: a later launch of a solve of <= 4 launches whose first launch built the operator -- nothing is reduced and nothing
    // tested here (the test of launch 0's residual practically never passes; waiting for its partials costs this launch
    // ~2 us of its ~10); workgroup 0 of k_tile_dudt_cheb reduces everything at once (solve_ctl.h, deferred_test_*)
    if (defer) {
        bnorm = 0.0;
    } else if (launch == bn_launch) {
        double rsmin = INFINITY;
        bnorm = 0.0;
        if (!BIG && launch > 0 && g_build == nwg) {
            const double* pr = p + ((launch - 1) & 1) * FEMFCT_MAX_PARTIALS;
            for (int k = threadIdx.x; k < nwg; k += blockDim.x) {
                bnorm = fmax(bnorm, p[2 * FEMFCT_MAX_PARTIALS + k]);
                rmax_prev = fmax(rmax_prev, pr[k]);
                rsmin = fmin(rsmin, p[3 * FEMFCT_MAX_PARTIALS + k]);
            }
            block_reduce_max_max_min(bnorm, rmax_prev, rsmin, smem);
            have_rmax = true;
        } else {
            bnorm = reduce_partials(p + 2 * FEMFCT_MAX_PARTIALS, g_build, OpMax(), 0.0, smem);
            rsmin = reduce_partials(p + 3 * FEMFCT_MAX_PARTIALS, g_build, OpMin(), INFINITY, smem);
        }
        if (wg == 0 && threadIdx.x == 0) {
            ctl->bnorm = bnorm;
            ctl->min_rowsum = rsmin;
            if (!(rsmin > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        }
    } else {
        bnorm = ctl->bnorm;
    }
    if (launch > 0 && !defer) {
        double rmax = have_rmax ? rmax_prev
                      : BIG   ? ctl->rs[(launch - 1) & 1]
                              : reduce_partials(p + ((launch - 1) & 1) * FEMFCT_MAX_PARTIALS, nwg, OpMax(), 0.0, smem);
        if (rmax <= rel_tol * bnorm) {
            if (wg == 0 && threadIdx.x == 0) {
                ctl->done = 1; ctl->parity = launch & 1; ctl->iters = launch * K; ctl->flags |= FEMFCT_FLAG_COARSE_ITERS;
                ctl->resid = bnorm > 0.0 ? rmax / bnorm : 0.0;
            }
            return;
        }
    }
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* L = L_ + moff;
    const double* xin = ((launch & 1) ? xb_ : xa_) + voff;
    double* xout = ((launch & 1) ? xa_ : xb_) + voff;
    const TileGeom g = tile_geom<TILE_L - 2 * H, H>(N);
    double lv[W - 1], dg = 1.0, rdg = 1.0, bv = 0.0, xi = 0.0;
#pragma unroll
    for (int s = 0; s < W - 1; ++s) lv[s] = 0.0;
    if (g.inside) {
        dg = L[g.i];
        rdg = 1.0 / dg;
#pragma unroll
        for (int s = 1; s < W; ++s) lv[s - 1] = L[(int64_t)s * n + g.i];
        bv = b_[voff + g.i];
        xi = xin[g.i];
    }
    xs[0][g.self] = xi;
    __syncthreads();
    double rmax = 0.0;
    int cur = 0;
    if (!EXACT) {
        for (int k = 0; k < K; ++k) {
            const double* c = xs[cur];
            double xn = c[g.self];
            if (k < g.kvalid) {
                double acc = bv;
#pragma unroll
                for (int s = 0; s < W - 1; ++s) acc = fma(-lv[s], c[g.nb[s]], acc);
                if (k == K - 1 && g.owned) rmax = fmax(rmax, fabs(acc - dg * xn));
                xn = acc * rdg;
            }
            xs[cur ^ 1][g.self] = xn;
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // last launch of the budget, optional: log the residual of every sweep's input so that the
        // host learns the exact sweep count (fully unrolled: per-sweep values stay in registers)
        double rk[H];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            rk[k] = 0.0;
            if (k < K) {
                const double* c = xs[cur];
                double xn = c[g.self];
                if (k < g.kvalid) {
                    double acc = bv;
#pragma unroll
                    for (int s = 0; s < W - 1; ++s) acc = fma(-lv[s], c[g.nb[s]], acc);
                    if (g.owned) rk[k] = fabs(acc - dg * xn);
                    if (k == K - 1) rmax = fmax(rmax, rk[k]);
                    xn = acc * rdg;
                }
                xs[cur ^ 1][g.self] = xn;
                __syncthreads();
                cur ^= 1;
            }
        }
        __shared__ double sk[H][STRIP_T / WAVE];
        const int wid = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
        for (int k = 0; k < H; ++k) {
            double v = wave_reduce(rk[k], OpMax());
            if (lane == 0) sk[k][wid] = v;
        }
        __syncthreads();
        if (threadIdx.x < H) {
            double v = 0.0;
            for (int w = 0; w < STRIP_T / WAVE; ++w) v = fmax(v, sk[threadIdx.x][w]);
            partk[((int64_t)bz * 16 + threadIdx.x) * FEMFCT_MAX_PARTIALS + wg] = v;
        }
    }
    if (g.owned) xout[g.i] = xs[cur][g.self];
    rmax = block_reduce(rmax, OpMax(), 0.0, smem);
    if (threadIdx.x == 0) {
        if (BIG) bigpart[(int64_t)bz * nwg + wg] = rmax;
        else if (defer) partk[((int64_t)bz * 16 + launch) * FEMFCT_MAX_PARTIALS + wg] = rmax;   // one slot per launch
        else p[(launch & 1) * FEMFCT_MAX_PARTIALS + wg] = rmax;
    }
}

// Launch 0 of the low-order solve with the construction of the low-order operator folded in
// (what k_build_low does, helpers.py:1769-1780, same expressions in the same order => bitwise the same
// L, D, b): every thread builds the row of its patch node from A (a_ji comes from the neighbour thread
// through LDS, three slots at a time), the tile's owned rows are stored for the later launches / the
// limiter, then K Jacobi sweeps run as in k_tile_jacobi.  Saves one dependent launch per time step.
template <int H>
__global__ void __launch_bounds__(STRIP_T)
k_tile_build_jacobi(int n, int N, MatRef A_ref, const double* __restrict__ N_, int nshared,
                    VecRef rhs_ref, int64_t rhs_bstride, VecRef u_ref, int64_t u_bstride,
                    const double* __restrict__ ml, double dt, double* __restrict__ L_, double* __restrict__ D_,
                    double* __restrict__ b_, double* __restrict__ xb_, double* __restrict__ part,
                    StepCtl* __restrict__ ctl_, int K) {
    constexpr int W = 7;
    __shared__ double xs[2][TILE_L * TILE_LD];
    __shared__ double as[3][TILE_L * TILE_LD];
    __shared__ double smem[96];
    const int bz = blockIdx.z;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    if (wg == 0 && threadIdx.x == 0) {
        StepCtl* c = ctl_ + bz;
        c->flags = 0; c->iters = 0; c->done = 0; c->parity = 0; c->resid = 0.0; c->bnorm = 0.0;
        c->min_rowsum = 0.0;
    }
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* A = mat_ptr(A_ref, bz);
    const double* Nm = N_ ? N_ + (nshared ? 0 : moff) : nullptr;
    const double* rhs = vec_ptr(rhs_ref);
    if (rhs) rhs += bz * rhs_bstride;
    const double* u = vec_ptr(u_ref) + bz * u_bstride;
    const TileGeom g = tile_geom<TILE_L - 2 * H, H>(N);
    // neighbour s exists in the grid (otherwise the ELL slot is padding: column = row, value 0)
    const int dx[6] = {1, 1, 0, -1, -1, 0}, dy[6] = {0, 1, 1, 0, -1, -1};
    bool ex[W - 1];
#pragma unroll
    for (int s = 0; s < W - 1; ++s) {
        const int nx = g.gx + dx[s], ny = g.gy + dy[s];
        ex[s] = g.inside && nx >= 0 && nx < N && ny >= 0 && ny < N;
    }
    double av[W - 1], at[W - 1], a0 = 0.0;
#pragma unroll
    for (int s = 0; s < W - 1; ++s) av[s] = 0.0;
    if (g.inside) {
        a0 = A[g.i];
#pragma unroll
        for (int s = 1; s < W; ++s) av[s - 1] = A[(int64_t)s * n + g.i];
    }
    // a_ji of slot s lives in row j at the opposite slot: slots E,NE,N <-> W,SW,S
#pragma unroll
    for (int s = 0; s < 3; ++s) as[s][g.self] = av[s + 3];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 3; ++s) at[s] = as[s][g.nb[s]];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 3; ++s) as[s][g.self] = av[s];
    __syncthreads();
#pragma unroll
    for (int s = 3; s < 6; ++s) at[s] = as[s - 3][g.nb[s]];
    double lv[W - 1], dv[W - 1], dsum = 0.0, rs = 0.0;
#pragma unroll
    for (int s = 0; s < W - 1; ++s) {
        const double a = av[s];
        const double d = ex[s] ? fmax(0.0, fmax(a, at[s])) : 0.0;   // d_ij = max(0, a_ij, a_ji)
        dsum += d;
        double l = dt * (a - d);
        if (Nm && g.inside) l += dt * Nm[(int64_t)(s + 1) * n + g.i];
        lv[s] = l;
        dv[s] = d;
        rs += l;
    }
    double dg = 1.0, rdg = 1.0, bv = 0.0, xi = 0.0;
    if (g.inside) {
        const double mli = ml[g.i];
        double ld = mli + dt * (a0 + dsum);                         // d_ii = -sum_j d_ij
        if (Nm) ld += dt * Nm[g.i];
        rs += ld;
        dg = ld;
        rdg = 1.0 / dg;
        xi = u[g.i];
        bv = mli * xi + (rhs ? dt * rhs[g.i] : 0.0);
    }
    double bmax = 0.0, rsmin = INFINITY;
    if (g.owned) {
        double* L = L_ + moff;
        double* D = D_ + moff;
        L[g.i] = dg;
        D[g.i] = -dsum;
#pragma unroll
        for (int s = 1; s < W; ++s) {
            L[(int64_t)s * n + g.i] = lv[s - 1];
            D[(int64_t)s * n + g.i] = dv[s - 1];
        }
        b_[voff + g.i] = bv;
        bmax = fabs(bv);
        rsmin = rs;
    }
    xs[0][g.self] = xi;
    __syncthreads();
    double rmax = 0.0;
    int cur = 0;
    for (int k = 0; k < K; ++k) {
        const double* c = xs[cur];
        double xn = c[g.self];
        if (k < g.kvalid) {
            double acc = bv;
#pragma unroll
            for (int s = 0; s < W - 1; ++s) acc = fma(-lv[s], c[g.nb[s]], acc);
            if (k == K - 1 && g.owned) rmax = fmax(rmax, fabs(acc - dg * xn));
            xn = acc * rdg;
        }
        xs[cur ^ 1][g.self] = xn;
        __syncthreads();
        cur ^= 1;
    }
    if (g.owned) xb_[voff + g.i] = xs[cur][g.self];
    block_reduce_max_max_min(rmax, bmax, rsmin, smem);
    if (threadIdx.x == 0) {
        p[wg] = rmax;
        p[2 * FEMFCT_MAX_PARTIALS + wg] = bmax;
        p[3 * FEMFCT_MAX_PARTIALS + wg] = rsmin;
    }
}

// one block per batch member: max over the residual partials of a fused launch on a large grid
__global__ void __launch_bounds__(STRIP_T)
k_reduce_resid(const double* __restrict__ bigpart, int64_t count, StepCtl* __restrict__ ctl_, int launch) {
    __shared__ double smem[32];
    const int bz = blockIdx.x;
    if (ctl_[bz].done) return;
    const double* q = bigpart + (int64_t)bz * count;
    double v = 0.0;
    for (int64_t k = threadIdx.x; k < count; k += blockDim.x) v = fmax(v, q[k]);
    v = block_reduce(v, OpMax(), 0.0, smem);
    if (threadIdx.x == 0) ctl_[bz].rs[launch & 1] = v;
}

template <int H>
__global__ void __launch_bounds__(STRIP_T)
k_tile_cheb(int n, int N, const double* __restrict__ M, const double* __restrict__ b_, const double* __restrict__ ymid_,
            const double* __restrict__ yold_, double* __restrict__ omid_, double* __restrict__ oold_, int K,
            CheOmegas om, double md_scale, ChebIO cio) {
    constexpr int W = 7;
    __shared__ double ys[3][TILE_L * TILE_LD];
    const int64_t voff = (int64_t)blockIdx.z * n;
    if (cio.mat) M = cio.mat + (int64_t)blockIdx.z * cio.mat_bs;
    if (cio.scale_dev) md_scale = cio.scale_dev[blockIdx.z];
    const double* omd = cio.om_dev ? cio.om_dev + (int64_t)blockIdx.z * cio.om_bs + cio.k0 : nullptr;
    if (cio.mid_ref.base) ymid_ = vec_ptr(cio.mid_ref) + (int64_t)blockIdx.z * cio.mid_bs - voff;
    if (cio.out_ref.base) omid_ = const_cast<double*>(vec_ptr(cio.out_ref)) + (int64_t)blockIdx.z * cio.out_bs - voff;
    const TileGeom g = tile_geom<TILE_L - 2 * H, H>(N);
    double mv[W - 1], md = 1.0, rmd = 1.0, bv = 0.0, ym = 0.0, yo = 0.0;
#pragma unroll
    for (int s = 0; s < W - 1; ++s) mv[s] = 0.0;
    if (g.inside) {
        md = M[g.i];
        rmd = 1.0 / (md_scale * md);
#pragma unroll
        for (int s = 1; s < W; ++s) mv[s - 1] = M[(int64_t)s * n + g.i];
        bv = b_[voff + g.i];
        if (ymid_) ym = ymid_[voff + g.i];
        if (yold_) yo = yold_[voff + g.i];
    }
    ys[0][g.self] = yo;
    ys[1][g.self] = ym;
    __syncthreads();
    int io = 0, im = 1, in_ = 2;
    for (int k = 0; k < K; ++k) {
        const double* ymd = ys[im];
        const double ymv = ymd[g.self];
        double yn = ymv;
        if (k < g.kvalid) {
            double acc = md * ymv;
#pragma unroll
            for (int s = 0; s < W - 1; ++s) acc = fma(mv[s], ymd[g.nb[s]], acc);
            const double z = (bv - acc) * rmd;
            const double yov = ys[io][g.self];
            const double wk = omd ? omd[k] : om.w[k];
            yn = wk * (z + ymv - yov) + yov;
        }
        ys[in_][g.self] = yn;
        __syncthreads();
        int t = io; io = im; im = in_; in_ = t;
    }
    if (g.owned) {
        omid_[voff + g.i] = ys[im][g.self];
        if (oold_) oold_[voff + g.i] = ys[io][g.self];
    }
}


// flux + Zalesak limiter + explicit correction in one launch (helpers.py:1818-1870): a 12 x 12
// tile with a halo of two rings (16 x 16 patch, 256 threads) (R+- of the first ring needs u_L, du/dt of the second);
// F_ij stays in registers, R+- goes through LDS.
#define FL_H 2
// (two 1024-thread workgroups per CU need <= 64 VGPRs: the second launch-bound argument is waves per SIMD)
// LDS slot of neighbour s of this thread's node (clamped into the patch, as TileGeom::nb), recomputed from the thread
// index at every use: kept in registers across the limiter's three phases the six slots are what the 32-patch variant
// (64-VGPR limit) spills -- 12 bytes per lane, 66 MB of scratch traffic per launch at 2049^2
template <int PL>
__device__ __forceinline__ int fl_nb(int s) {
    const int lx = threadIdx.x % PL, ly = threadIdx.x / PL;
    const int dx = (s == 0 || s == 1) ? 1 : (s == 3 || s == 4) ? -1 : 0;
    const int dy = (s == 1 || s == 2) ? 1 : (s == 4 || s == 5) ? -1 : 0;
    return min(max(ly + dy, 0), PL - 1) * (PL + 1) + min(max(lx + dx, 0), PL - 1);
}

template <int FL_L, int GEOM, int HALFD>   // patch edge: 16 (256 threads, many workgroups: small meshes) or 32 (less halo re-reading)
__global__ void __launch_bounds__(FL_L * FL_L, FL_L == 32 ? 8 : 1)
k_tile_flux_limit(int n, int N, double h, const double* __restrict__ M, const double* __restrict__ D_,
                  const double* __restrict__ ulow_, const double* __restrict__ du_, const double* __restrict__ ml,
                  double dt, VecRef out_ref, int64_t out_bstride, EndArgs e) {
    constexpr int W = 7;
    constexpr int FL_LD = FL_L + 1, FL_T = FL_L - 2 * FL_H;
    __shared__ double su[FL_L * FL_LD], sd[FL_L * FL_LD], srp[FL_L * FL_LD], srm[FL_L * FL_LD];
    __shared__ double sdf[HALFD ? 3 : 1][HALFD ? FL_L * FL_LD : 1];   // HALFD: the forward slots (E, NE, N) of D
    // The step end is folded in for the 16-patch only (meshes up to 512^2, where a launch counts); the 32-patch runs at
    // its 64-VGPR limit (two 1024-thread workgroups per CU) and every spilled dword there is 44 MB of scratch traffic
    // at 2049^2 -- its step end stays a separate tiny launch (femfct_enqueue_tile_flux_limit reports fuse_end back).
    if (FL_L == 16) step_log_early(e);
    const int bz = blockIdx.z;
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const TileGeom g = tile_geom<FL_T, FL_H, FL_L>(N);
    double ui = 0.0, dui = 0.0, mli = 1.0, dfw[3] = {0.0, 0.0, 0.0};
    if (g.inside) { ui = ulow_[voff + g.i]; dui = du_[voff + g.i]; mli = ml[g.i]; }
    if (HALFD) {
        // d_ij is stored once per edge, in the row whose slot towards the neighbour is E, NE or N (k_build_low, half_d)
        if (g.inside) {
#pragma unroll
            for (int s = 0; s < 3; ++s) dfw[s] = D_[moff + (int64_t)(s + 1) * n + g.i];
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) sdf[s][g.self] = dfw[s];
    }
    su[g.self] = ui;
    sd[g.self] = dui;
    srp[g.self] = 1.0;
    srm[g.self] = 1.0;
    __syncthreads();
    double f[W - 1];
    // every inside node whose six neighbours are in the patch (or outside the grid) gets its fluxes
    const bool have = g.inside && g.kvalid >= 1;
    if (have) {
        double pp = 0.0, pm = 0.0, umax = ui, umin = ui;
        const int pc = (GEOM || HALFD) ? mass_edge_counts(g.gx, g.gy, N - 1) : 0;
        const double mq = (0.5 * h * h) / 12.0;
#pragma unroll
        for (int s = 1; s < W; ++s) {
            const int64_t idx = (int64_t)s * n + g.i;
            const int nbs = fl_nb<FL_L>(s - 1);
            const double uj = su[nbs];
            const double mij = GEOM ? (double)((pc >> (2 * (s - 1))) & 3) * mq : M[idx];
            double dij;
            if (HALFD) {
                // backward slots W, SW, S: the neighbour's forward entry (a neighbour outside the grid has none:
                // its clamped LDS slot aliases a patch node)
                const bool exists = ((pc >> (2 * (s - 1))) & 3) != 0;
                dij = s <= 3 ? sdf[s - 1][g.self] : (exists ? sdf[s - 4][nbs] : 0.0);   // (own entries re-read from LDS: six registers less across the barrier)
            } else {
                dij = D_[moff + idx];
            }
            const double fs = mij * (dui - sd[nbs]) + dij * (ui - uj);
            f[s - 1] = fs;
            pp += fmax(fs, 0.0);
            pm += fmin(fs, 0.0);
            // a clamped neighbour (outside the grid) has M = D = 0 and must not enter the bounds:
            // its LDS slot then aliases a patch node, so take it only when the coefficient is live
            const bool live = (mij != 0.0) || (dij != 0.0);
            umax = live ? fmax(umax, uj) : umax;
            umin = live ? fmin(umin, uj) : umin;
        }
        const double qp = umax - ui, qm = umin - ui;
        srp[g.self] = (pp != 0.0) ? fmin(1.0, mli * qp / (dt * pp)) : 1.0;
        srm[g.self] = (pm != 0.0) ? fmin(1.0, mli * qm / (dt * pm)) : 1.0;
    }
    __syncthreads();
    if (g.owned) {
        const double rpi = srp[g.self], rmi = srm[g.self];
        double fbar = 0.0;
#pragma unroll
        for (int s = 0; s < W - 1; ++s) {
            const double fs = f[s];
            const int nbs = fl_nb<FL_L>(s);
            const double a = (fs > 0.0) ? fmin(rpi, srm[nbs]) : fmin(rmi, srp[nbs]);
            fbar += a * fs;
        }
        double* out = const_cast<double*>(vec_ptr(out_ref)) + bz * out_bstride;
        out[g.i] = ui + dt * fbar / mli;
    }
    if (FL_L == 16) step_end_by_last_workgroup(e);
}

}  // namespace

namespace {

// The last <= 10 Chebyshev iterations of du/dt (helpers.py:175-184) and the whole limiter
// (helpers.py:1818-1870) in one launch: 8 x 8 tile + halo 12 (ten rings for the iterations, two for
// the limiter: R+- of the ring-1 neighbours).  du never goes to memory.  Same expressions in the same
// order as k_tile_cheb + k_tile_flux_limit => bitwise the same step.  Latency regime only.
template <int GEOM>
__global__ void __launch_bounds__(STRIP_T)
k_tile_cheb_flux_limit(int n, int N, double h, const double* __restrict__ M, const double* __restrict__ b_,
                       const double* __restrict__ ymid_, const double* __restrict__ yold_, int K, CheOmegas om,
                       double md_scale, const double* __restrict__ D_, const double* __restrict__ ulow_,
                       const double* __restrict__ ml, double dt, VecRef out_ref, int64_t out_bstride, EndArgs e) {
    constexpr int W = 7, HH = 12;
    __shared__ double ys[3][TILE_L * TILE_LD];
    __shared__ double su[TILE_L * TILE_LD], srp[TILE_L * TILE_LD], srm[TILE_L * TILE_LD];
    const int bz = blockIdx.z;
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const TileGeom g = tile_geom<TILE_L - 2 * HH, HH>(N);
    double mv[W - 1], dv[W - 1], md = 1.0, rmd = 1.0, bv = 0.0, ym = 0.0, yo = 0.0, ui = 0.0, mli = 1.0;
#pragma unroll
    for (int s = 0; s < W - 1; ++s) { mv[s] = 0.0; dv[s] = 0.0; }
    if (g.inside) {
        md = M[g.i];
        rmd = 1.0 / (md_scale * md);
        if (GEOM) {
            const int pc = mass_edge_counts(g.gx, g.gy, N - 1);
            const double mq = (0.5 * h * h) / 12.0;
#pragma unroll
            for (int s = 1; s < W; ++s) mv[s - 1] = (double)((pc >> (2 * (s - 1))) & 3) * mq;
        } else {
#pragma unroll
            for (int s = 1; s < W; ++s) mv[s - 1] = M[(int64_t)s * n + g.i];
        }
        bv = b_[voff + g.i];
        ym = ymid_[voff + g.i];
        yo = yold_[voff + g.i];
#pragma unroll
        for (int s = 1; s < W; ++s) dv[s - 1] = D_[moff + (int64_t)s * n + g.i];
        ui = ulow_[voff + g.i];
        mli = ml[g.i];
    }
    ys[0][g.self] = yo;
    ys[1][g.self] = ym;
    su[g.self] = ui;
    srp[g.self] = 1.0;
    srm[g.self] = 1.0;
    __syncthreads();
    step_log_early(e);        // (behind this kernel's own loads; its round trip is covered by the Chebyshev iterations)
    int io = 0, im = 1, in_ = 2;
    for (int k = 0; k < K; ++k) {
        const double* ymd = ys[im];
        const double ymv = ymd[g.self];
        double yn = ymv;
        if (k < g.kvalid) {
            double acc = md * ymv;
#pragma unroll
            for (int s = 0; s < W - 1; ++s) acc = fma(mv[s], ymd[g.nb[s]], acc);
            const double z = (bv - acc) * rmd;
            const double yov = ys[io][g.self];
            yn = om.w[k] * (z + ymv - yov) + yov;
        }
        ys[in_][g.self] = yn;
        __syncthreads();
        int t = io; io = im; im = in_; in_ = t;
    }
    const double* sd = ys[im];
    const double dui = sd[g.self];
    double f[W - 1];
    // fluxes where all six neighbours carry the final du (one ring inside the iterations' validity)
    const bool have = g.inside && g.kvalid >= K + 1;
    if (have) {
        double pp = 0.0, pm = 0.0, umax = ui, umin = ui;
#pragma unroll
        for (int s = 0; s < W - 1; ++s) {
            const double uj = su[g.nb[s]];
            const double mij = mv[s], dij = dv[s];
            const double fs = mij * (dui - sd[g.nb[s]]) + dij * (ui - uj);
            f[s] = fs;
            pp += fmax(fs, 0.0);
            pm += fmin(fs, 0.0);
            const bool live = (mij != 0.0) || (dij != 0.0);
            umax = live ? fmax(umax, uj) : umax;
            umin = live ? fmin(umin, uj) : umin;
        }
        const double qp = umax - ui, qm = umin - ui;
        srp[g.self] = (pp != 0.0) ? fmin(1.0, mli * qp / (dt * pp)) : 1.0;
        srm[g.self] = (pm != 0.0) ? fmin(1.0, mli * qm / (dt * pm)) : 1.0;
    }
    __syncthreads();
    if (g.owned) {
        const double rpi = srp[g.self], rmi = srm[g.self];
        double fbar = 0.0;
#pragma unroll
        for (int s = 0; s < W - 1; ++s) {
            const double fs = f[s];
            const double a = (fs > 0.0) ? fmin(rpi, srm[g.nb[s]]) : fmin(rmi, srp[g.nb[s]]);
            fbar += a * fs;
        }
        double* out = const_cast<double*>(vec_ptr(out_ref)) + bz * out_bstride;
        out[g.i] = ui + dt * fbar / mli;
    }
    step_end_by_last_workgroup(e);
}

}  // namespace

// the registered mass matrix is the structured mesh's own and may be derived from the cell geometry (FEMFCT_GEOM_MASS)
bool femfct_geom_mass(const femfct_ctx* ctx) {
    return ctx->geom_mass && ctx->structured && ctx->mass_is_mesh && ctx->implicit_cols;
}

// one workgroup per CU is the regime where the fused tail pays (see femfct_tile_plan)
bool femfct_cheb_flux_fusable(const femfct_ctx* ctx, int32_t batch) {
    if (!ctx->fuse_flux || ctx->N > 512) return false;
    const int t = (ctx->N + 7) / 8;
    return (int64_t)t * t * batch <= ctx->wg_slots;
}

int femfct_enqueue_tile_cheb_flux_limit(femfct_ctx* ctx, const double* b, const double* in_mid, const double* in_old,
                                        int k_first, int k_last, const double* omegas, double md_scale, const double* D,
                                        const double* ulow, double dt, VecRef out, int64_t out_bstride, int32_t batch,
                                        bool fuse_end) {
    const int K = k_last - k_first + 1;
    if (K < 1 || K > 10) return femfct_fail(ctx, FEMFCT_ERR_INVALID, "fused Chebyshev tail: %d iterations", K);
    EndArgs e{};
    e.level = nullptr;
    if (fuse_end) {
        e.level = ctx->d_level; e.delta = ctx->rep_last ? ctx->end_req_delta * ctx->rep_total : 0;
        e.ord_adv = ctx->rep_last ? ctx->rep_total : 0; e.ord_off = ctx->ord_bias; e.ctl = ctx->d_ctl; e.log = ctx->d_log;
        e.kctl = ctx->end_req_krylov ? (const KrylovCtl*)ctx->d_kry_ctl : nullptr; e.klog = (KrylovCtl*)ctx->d_klog;
        e.batch = batch; e.ticket = ctx->d_ticket;
    }
    CheOmegas om;
    for (int k = k_first; k <= k_last; ++k) om.w[k - k_first] = omegas[k - 1];
    const int t = (ctx->N + 7) / 8;
    femfct_prof_begin(ctx, KC_FLUX);
    if (femfct_geom_mass(ctx))
        hipLaunchKernelGGL(k_tile_cheb_flux_limit<1>, dim3(t, t, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, ctx->h,
                           ctx->d_M, b, in_mid, in_old, K, om, md_scale, D, ulow, ctx->d_ml, dt, out, out_bstride, e);
    else
        hipLaunchKernelGGL(k_tile_cheb_flux_limit<0>, dim3(t, t, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, ctx->h,
                           ctx->d_M, b, in_mid, in_old, K, om, md_scale, D, ulow, ctx->d_ml, dt, out, out_bstride, e);
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

// *fuse_end (in/out): whether this launch also does the step end (only the 16-patch variant can: see the kernel)
int femfct_enqueue_tile_flux_limit(femfct_ctx* ctx, const double* D, const double* ulow, const double* du, double dt,
                                   VecRef out, int64_t out_bstride, int32_t batch, bool* fuse_end_io, int half_d) {
    const bool fuse_end = fuse_end_io && *fuse_end_io && ctx->N <= 512;
    if (fuse_end_io) *fuse_end_io = fuse_end;
    EndArgs e{};
    e.level = nullptr;
    if (fuse_end) {
        e.level = ctx->d_level; e.delta = ctx->rep_last ? ctx->end_req_delta * ctx->rep_total : 0;
        e.ord_adv = ctx->rep_last ? ctx->rep_total : 0; e.ord_off = ctx->ord_bias; e.ctl = ctx->d_ctl; e.log = ctx->d_log;
        e.kctl = ctx->end_req_krylov ? (const KrylovCtl*)ctx->d_kry_ctl : nullptr; e.klog = (KrylovCtl*)ctx->d_klog;
        e.batch = batch; e.ticket = ctx->d_ticket;
    }
    femfct_prof_begin(ctx, KC_FLUX);
    const bool geom = femfct_geom_mass(ctx);
#define FL_(PL, G, HD, T_) hipLaunchKernelGGL((k_tile_flux_limit<PL, G, HD>), dim3(T_, T_, batch), dim3(PL * PL), 0, ctx->stream, \
                                              ctx->n, ctx->N, ctx->h, ctx->d_M, D, ulow, du, ctx->d_ml, dt, out, out_bstride, e)
    if (ctx->N <= 512) {
        const int t = (ctx->N + 11) / 12;
        if (half_d) { if (geom) FL_(16, 1, 1, t); else FL_(16, 0, 1, t); }
        else { if (geom) FL_(16, 1, 0, t); else FL_(16, 0, 0, t); }
    } else {
        const int t = (ctx->N + 27) / 28;
        if (half_d) { if (geom) FL_(32, 1, 1, t); else FL_(32, 0, 1, t); }
        else { if (geom) FL_(32, 1, 0, t); else FL_(32, 0, 0, t); }
    }
#undef FL_
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

// The 32 x 32 patch is split as tile + 2 halos with halo H in {8, 9, 10}: H sweeps fit in one launch.
// Latency regime (small grids): pick the H that needs the fewest launches for the sweep budget
// (Chebyshev: 19 remaining iterations = 10 + 9 with H = 10).  Bandwidth regime (large grids): H = 8
// keeps the halo re-reading lowest.  need_partials: the Jacobi variant publishes one residual partial
// per workgroup, consumed in-kernel up to FEMFCT_MAX_PARTIALS workgroups (else a reduce kernel).
bool femfct_tile_plan(const femfct_ctx* ctx, TilePlan* pl, bool need_partials, int budget, int batch) {
    if (!ctx->use_strips || !ctx->use_tiles || !ctx->implicit_cols || ctx->W != 7) return false;
    int H = 8;
    const bool small = ctx->N <= 512;
    if (small) {
        if (budget <= 0) H = 10;
        else {
            // fewest launches first, then the smallest halo.  Deep halos (11..13: tiles of 10..6 nodes per
            // side) only while every workgroup still gets its own CU -- there a launch costs ~4.4 us fixed
            // + ~0.3 us per sweep whatever the tile size (tools/lat_probe.hip), so 2 x 13 beats 3 x 9.
            int best = 1 << 30;
            for (int h = 8; h <= TILE_HMAX; ++h) {
                const int T = TILE_L - 2 * h, t = (ctx->N + T - 1) / T;
                if (h > 10 && (!ctx->deep_halo || (int64_t)t * t * batch > ctx->wg_slots)) break;
                int launches = (budget + h - 1) / h;
                if (launches < best) { best = launches; H = h; }
            }
        }
    }
    if (small && ctx->strip_k >= 8 && ctx->strip_k <= TILE_HMAX) H = ctx->strip_k;   // tuning knob (latency regime only)
    const int T = TILE_L - 2 * H;
    const int t = (ctx->N + T - 1) / T;
    if (need_partials && (int64_t)t * t > FEMFCT_MAX_PARTIALS) return false;
    if (t > 65535) return false;
    pl->tiles = t;
    pl->K = H;
    pl->H = H;
    return true;
}

bool femfct_tile_big(const femfct_ctx* ctx, const TilePlan& pl) {
    return (int64_t)pl.tiles * pl.tiles > FEMFCT_MAX_PARTIALS;
}

// launch 0 with the operator construction fused in (latency regime; needs >= 2 launches in total because
// ||b|| is reduced by launch 1).  Later launches: femfct_enqueue_tile_jacobi(..., bn_launch = 1, g_build = tiles^2).
int femfct_enqueue_tile_build_jacobi(femfct_ctx* ctx, const TilePlan& pl, MatRef A, const double* Nm, int32_t nshared,
                                     VecRef rhs, int64_t rhs_bstride, VecRef u_n, int64_t u_bstride, double dt,
                                     int32_t batch) {
    dim3 grid(pl.tiles, pl.tiles, batch);
    femfct_prof_begin(ctx, KC_JACOBI);
#define TB(HH) hipLaunchKernelGGL((k_tile_build_jacobi<HH>), grid, dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, A, Nm,  \
                                  nshared, rhs, rhs_bstride, u_n, u_bstride, ctx->d_ml, dt, ctx->d_L, ctx->d_D, ctx->d_b, \
                                  ctx->d_xb, ctx->d_part, ctx->d_ctl, pl.K)
    switch (pl.H) {
        case 8: TB(8); break; case 9: TB(9); break; case 10: TB(10); break;
        case 11: TB(11); break; case 12: TB(12); break; default: TB(13); break;
    }
#undef TB
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

int femfct_enqueue_tile_jacobi(femfct_ctx* ctx, const TilePlan& pl, const double* L, const double* b, double* xa,
                               double* xb, int launch, int g_build, int32_t batch, bool last, int bn_launch, int defer) {
    const bool big = femfct_tile_big(ctx, pl);
    double* bigp = big ? ctx->d_bigpart : nullptr;
    double* pk = ((last || defer) && !big) ? ctx->d_partk : nullptr;
    dim3 grid(pl.tiles, pl.tiles, batch);
    femfct_prof_begin(ctx, KC_JACOBI);
#define TJ(HH)                                                                                                          \
    do {                                                                                                                \
        if (big) hipLaunchKernelGGL((k_tile_jacobi<8, 0, 1>), grid, dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, L, b,  \
                                    xa, xb, ctx->d_part, ctx->d_ctl, launch, pl.K, g_build, ctx->rel_tol, bigp, pk, bn_launch, 0); \
        else if (pk && !defer) hipLaunchKernelGGL((k_tile_jacobi<HH, 1, 0>), grid, dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, L, \
                                        b, xa, xb, ctx->d_part, ctx->d_ctl, launch, pl.K, g_build, ctx->rel_tol, bigp, pk, bn_launch, 0); \
        else hipLaunchKernelGGL((k_tile_jacobi<HH, 0, 0>), grid, dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, L, b, xa, \
                                xb, ctx->d_part, ctx->d_ctl, launch, pl.K, g_build, ctx->rel_tol, bigp, pk, bn_launch, defer); \
    } while (0)
    switch (pl.H) {
        case 8: TJ(8); break; case 9: TJ(9); break; case 10: TJ(10); break;
        case 11: TJ(11); break; case 12: TJ(12); break; default: TJ(13); break;
    }
#undef TJ
    if (big)
        hipLaunchKernelGGL(k_reduce_resid, dim3(batch), dim3(STRIP_T), 0, ctx->stream, ctx->d_bigpart,
                           (int64_t)pl.tiles * pl.tiles, ctx->d_ctl, launch);
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

int femfct_enqueue_tile_cheb(femfct_ctx* ctx, const TilePlan& pl, const double* b, const double* in_mid,
                             const double* in_old, double* y_out, int k_first, int k_last, const double* omegas,
                             double md_scale, double* bufA0, double* bufA1, double* bufB0, double* bufB1, int32_t batch,
                             const ChebIO* io_in) {
    ChebIO io0{};
    if (io_in) io0 = *io_in;
    io0.mid_ref = make_ref(nullptr); io0.mid_bs = 0; io0.out_ref = make_ref(nullptr); io0.out_bs = 0;
    const double* mid = in_mid;
    const double* old = in_old;
    int which = 0;
    for (int k0 = k_first; k0 <= k_last; k0 += pl.K) {
        int k1 = std::min(k_last + 1, k0 + pl.K);
        CheOmegas om;
        for (int k = k0; k < k1; ++k) om.w[k - k0] = omegas ? omegas[k - 1] : 0.0;
        const bool last = (k1 == k_last + 1);
        double* omid = last ? y_out : (which ? bufB0 : bufA0);
        double* oold = last ? nullptr : (which ? bufB1 : bufA1);
        femfct_prof_begin(ctx, KC_CHEB);
        ChebIO io = io0;
        io.k0 = k0 - 1;
        if (io_in && k0 == k_first) { io.mid_ref = io_in->mid_ref; io.mid_bs = io_in->mid_bs; }
        if (io_in && last) { io.out_ref = io_in->out_ref; io.out_bs = io_in->out_bs; }
#define TC(HH) hipLaunchKernelGGL((k_tile_cheb<HH>), dim3(pl.tiles, pl.tiles, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, \
                                  ctx->N, ctx->d_M, b, mid, old, omid, oold, k1 - k0, om, md_scale, io)
        switch (pl.H) {
            case 8: TC(8); break; case 9: TC(9); break; case 10: TC(10); break;
            case 11: TC(11); break; case 12: TC(12); break; default: TC(13); break;
        }
#undef TC
        femfct_prof_end(ctx);
        mid = omid;
        old = oold;
        which ^= 1;
    }
    return FEMFCT_OK;
}

// ===========================================================================================
// du/dt right-hand side + the first Chebyshev iterations in one launch (latency regime):
//   r = rhs - A u_L (helpers.py:1814), y_1 = w_1 r / Md, then iterations 2..K+1 (helpers.py:175-184).
// 12 x 12 tile + halo 10: one ring is spent on A u_L, nine on Chebyshev iterations 2..10.
// Also finalises the low-order solve's bookkeeping (what k_dudt_rhs does in the unfused sequence).
// ===========================================================================================
namespace {

__global__ void __launch_bounds__(STRIP_T)
k_tile_dudt_cheb(int n, int N, MatRef A_ref, VecRef rhs_ref, int64_t rhs_bstride,
                 const double* __restrict__ M, const double* __restrict__ xa_, const double* __restrict__ xb_,
                 double* __restrict__ ulow_, double* __restrict__ rdu_, double* __restrict__ omid_,
                 double* __restrict__ oold_, double* __restrict__ part, StepCtl* __restrict__ ctl_, int budget,
                 int part_count, int iters_per_unit, double rel_tol, const double* __restrict__ partk, int exact_k,
                 int K, CheOmegas om, double md_scale, double omega1) {
    constexpr int W = 7, H = 10;
    __shared__ double ys[3][TILE_L * TILE_LD];
    __shared__ double smem[64];
    const int bz = blockIdx.z;
    StepCtl* ctl = ctl_ + bz;
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (exact_k < 0 && blockIdx.y == gridDim.x) {
        // deferred test: the launch carries one extra row of workgroups; its first one does nothing but reduce the solve's
        // partials and write the step record (no tile behind it: no workgroup of the kernel ends later for it)
        if (blockIdx.x == 0)
            deferred_test_publish(ctl, deferred_test_load(p, partk + (int64_t)bz * 16 * FEMFCT_MAX_PARTIALS, part_count, -exact_k),
                                  -exact_k, iters_per_unit, rel_tol);
        return;
    }
    // deferred test (exact_k < 0): no launch of the solve set `done`, the iterate is the budget-parity one; nothing below
    // depends on the test's outcome, only the step log does -- workgroup 0 alone reduces the partials, requested here
    // and consumed after its own tile, so that nobody waits for them (nor for a look at the control block)
    const int parity = exact_k < 0 ? (budget & 1) : ctl->done ? ctl->parity : (budget & 1);
    if (exact_k >= 0)
        finalize_solve(ctl, p, part_count, budget, iters_per_unit, rel_tol, smem,
                       partk ? partk + (int64_t)bz * 16 * FEMFCT_MAX_PARTIALS : nullptr, exact_k, wg == 0);
    const int64_t voff = (int64_t)bz * n;
    const double* A = mat_ptr(A_ref, bz);
    const double* x = (parity ? xb_ : xa_) + voff;
    const double* rhs = vec_ptr(rhs_ref);
    if (rhs) rhs += bz * rhs_bstride;
    const TileGeom g = tile_geom<TILE_L - 2 * H, H>(N);
    double av[W], mv[W - 1], md = 1.0, rmd = 1.0, ui = 0.0, ri = 0.0;
#pragma unroll
    for (int s = 0; s < W; ++s) av[s] = 0.0;
#pragma unroll
    for (int s = 0; s < W - 1; ++s) mv[s] = 0.0;
    if (g.inside) {
        md = M[g.i];
        rmd = 1.0 / (md_scale * md);
#pragma unroll
        for (int s = 0; s < W; ++s) av[s] = A[(int64_t)s * n + g.i];
#pragma unroll
        for (int s = 1; s < W; ++s) mv[s - 1] = M[(int64_t)s * n + g.i];
        ui = x[g.i];
        ri = rhs ? rhs[g.i] : 0.0;
    }
    ys[2][g.self] = ui;
    __syncthreads();
    // r = rhs - A u_L on every node whose neighbours are in the patch
    double r = 0.0, y1 = 0.0;
    if (g.kvalid >= 1) {
        double acc = av[0] * ui;
#pragma unroll
        for (int s = 1; s < W; ++s) acc = fma(av[s], ys[2][g.nb[s - 1]], acc);
        r = -acc + ri;
        y1 = omega1 * (r / (md_scale * md));
    }
    __syncthreads();
    ys[0][g.self] = 0.0;
    ys[1][g.self] = y1;
    __syncthreads();
    int io = 0, im = 1, in_ = 2;
    for (int k = 0; k < K; ++k) {
        const double* ymd = ys[im];
        const double ymv = ymd[g.self];
        double yn = ymv;
        if (k + 1 < g.kvalid) {      // one ring already spent on A u_L
            double acc = md * ymv;
#pragma unroll
            for (int s = 0; s < W - 1; ++s) acc = fma(mv[s], ymd[g.nb[s]], acc);
            const double z = (r - acc) * rmd;
            const double yov = ys[io][g.self];
            yn = om.w[k] * (z + ymv - yov) + yov;
        }
        ys[in_][g.self] = yn;
        __syncthreads();
        int t = io; io = im; im = in_; in_ = t;
    }
    if (g.owned) {
        ulow_[voff + g.i] = ui;
        rdu_[voff + g.i] = r;
        omid_[voff + g.i] = ys[im][g.self];
        if (oold_) oold_[voff + g.i] = ys[io][g.self];
    }
}

}  // namespace

// r, y_1 and Chebyshev iterations 2..(K+1) in one launch, the rest in ceil(.../10) tile launches.
// tail_first (optional): the caller runs the remaining iterations *tail_first .. iters itself (inputs
// mid = d_y0, old = d_y2), e.g. fused with the limiter; 0 is stored when nothing remains.
int femfct_enqueue_tile_dudt_cheb(femfct_ctx* ctx, MatRef A, VecRef rhs, int64_t rhs_bstride, double* ulow,
                                  int budget_units, int part_count, int iters_per_unit, int exact_k, int iters,
                                  const double* omegas, double md_scale, int32_t batch, int* tail_first) {
    constexpr int H = 10;
    const int T = TILE_L - 2 * H, t = (ctx->N + T - 1) / T;
    const int K = std::min(iters - 1, H - 1);            // iterations 2..K+1 here
    CheOmegas om;
    for (int k = 0; k < K; ++k) om.w[k] = omegas[k + 1];
    const bool last = (K + 1 == iters);
    double* omid = last ? ctx->d_du : ctx->d_y0;
    double* oold = last ? nullptr : ctx->d_y2;
    femfct_prof_begin(ctx, KC_DUDT_RHS);
    hipLaunchKernelGGL(k_tile_dudt_cheb, dim3(t, exact_k < 0 ? t + 1 : t, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, A, rhs,
                       rhs_bstride, ctx->d_M, ctx->d_xa, ctx->d_xb, ulow, ctx->d_rdu, omid, oold, ctx->d_part, ctx->d_ctl,
                       budget_units, part_count, iters_per_unit, ctx->rel_tol, exact_k ? ctx->d_partk : nullptr, exact_k,
                       K, om, md_scale, omegas[0]);
    femfct_prof_end(ctx);
    if (tail_first) *tail_first = last ? 0 : K + 2;
    if (last || tail_first) return FEMFCT_OK;
    TilePlan tp;
    tp.H = H; tp.K = H; tp.tiles = t;
    // remaining iterations K+2 .. iters; inputs (mid, old) = (y0, y2); scratch pair (y1, rp) then (y0, y2)
    return femfct_enqueue_tile_cheb(ctx, tp, ctx->d_rdu, ctx->d_y0, ctx->d_y2, ctx->d_du, K + 2, iters, omegas, md_scale,
                                    ctx->d_y1, ctx->d_rp, ctx->d_y0, ctx->d_y2, batch, nullptr);
}

// ===========================================================================================
// Bandwidth-regime tiles: 64 x 64 patch, (64 - 2H)^2 tile + halo H (8..10), four nodes per thread
// (1024 threads).  Each matrix row loaded from HBM is used for H sweeps and the halo re-read drops
// from 4x (32-patch) to 1.78x: ~17 B of HBM traffic per row and sweep instead of ~35 (one-sweep
// kernels: 104).  Used for large grids and for batches of small trajectories (femfct_tile4_wanted).
// Two implementations: k_tile4_* keep the iterate as an LDS image (fixed halo 8; FEMFCT_T4_DPP=0),
// k_strip4_* keep it in registers and exchange by DPP lane shifts (default, below).
// ===========================================================================================
#define T4_L 64
#define T4_LD 65
#define T4_H 8
#define T4_T 48
#define T4_BUF ((T4_L + 2) * T4_LD)   // one zero row above and below: neighbour offsets never leave the buffer

namespace {

// Per-node state kept small (four nodes per thread): neighbours are addressed as self + constant
// offset.  A neighbour outside the grid has a zero matrix entry and lands on a zero pad element or
// on another patch node (finite), so no clamping is needed.
struct Tile4Node {
    int i, self, kvalid;
    bool inside, owned;
};

__device__ __forceinline__ Tile4Node tile4_node(int N, int q) {
    Tile4Node g;
    const int lx = (threadIdx.x & 31) + 32 * (q & 1), ly = (threadIdx.x >> 5) + 32 * (q >> 1);
    const int x0 = blockIdx.x * T4_T - T4_H, y0 = blockIdx.y * T4_T - T4_H;
    const int gx = x0 + lx, gy = y0 + ly;
    g.inside = gx >= 0 && gx < N && gy >= 0 && gy < N;
    g.i = g.inside ? gy * N + gx : 0;
    g.owned = g.inside && lx >= T4_H && lx < T4_H + T4_T && ly >= T4_H && ly < T4_H + T4_T;
    int kv = 1 << 20;
    if (x0 > 0) kv = min(kv, lx);
    if (x0 + T4_L - 1 < N - 1) kv = min(kv, T4_L - 1 - lx);
    if (y0 > 0) kv = min(kv, ly);
    if (y0 + T4_L - 1 < N - 1) kv = min(kv, T4_L - 1 - ly);
    g.kvalid = g.inside ? kv : 0;
    g.self = (ly + 1) * T4_LD + lx;
    return g;
}

__device__ __forceinline__ int t4_off(int s) {   // slots E, NE, N, W, SW, S
    return (s == 0) ? 1 : (s == 1) ? T4_LD + 1 : (s == 2) ? T4_LD : (s == 3) ? -1 : (s == 4) ? -T4_LD - 1 : -T4_LD;
}

__device__ __forceinline__ void t4_zero(double* buf, int count) {
    for (int k = threadIdx.x; k < count; k += blockDim.x) buf[k] = 0.0;
}

// BIG as in k_tile_jacobi: residual partials via k_reduce_resid when the grid has more workgroups
// than in-kernel partials
template <int BIG>
__global__ void __launch_bounds__(STRIP_T)
k_tile4_jacobi(int n, int N, const double* __restrict__ L_, const double* __restrict__ b_, double* __restrict__ xa_,
               double* __restrict__ xb_, double* __restrict__ part, StepCtl* __restrict__ ctl_, int launch, int K,
               int g_build, double rel_tol, double* __restrict__ bigpart) {
    constexpr int W = 7;
    extern __shared__ double lds[];
    __shared__ double smem[32];
    const int bz = blockIdx.z;
    StepCtl* ctl = ctl_ + bz;
    if (ctl->done) return;
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
    double bnorm;
    if (launch == 0) {
        bnorm = reduce_partials(p + 2 * FEMFCT_MAX_PARTIALS, g_build, OpMax(), 0.0, smem);
        double rsmin = reduce_partials(p + 3 * FEMFCT_MAX_PARTIALS, g_build, OpMin(), INFINITY, smem);
        if (wg == 0 && threadIdx.x == 0) {
            ctl->bnorm = bnorm;
            ctl->min_rowsum = rsmin;
            if (!(rsmin > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        }
    } else {
        bnorm = ctl->bnorm;
        double rmax = BIG ? ctl->rs[(launch - 1) & 1]
                                  : reduce_partials(p + ((launch - 1) & 1) * FEMFCT_MAX_PARTIALS, nwg, OpMax(), 0.0, smem);
        if (rmax <= rel_tol * bnorm) {
            if (wg == 0 && threadIdx.x == 0) {
                ctl->done = 1; ctl->parity = launch & 1; ctl->iters = launch * K; ctl->flags |= FEMFCT_FLAG_COARSE_ITERS;
                ctl->resid = bnorm > 0.0 ? rmax / bnorm : 0.0;
            }
            return;
        }
    }
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* L = L_ + moff;
    const double* xin = ((launch & 1) ? xb_ : xa_) + voff;
    double* xout = ((launch & 1) ? xa_ : xb_) + voff;
    double* cur = lds;
    double* nxt = lds + T4_BUF;
    t4_zero(lds, 2 * T4_BUF);
    __syncthreads();
    Tile4Node g[4];
    double lv[4][W - 1], dg[4], rdg[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        g[q] = tile4_node(N, q);
        dg[q] = 1.0; rdg[q] = 1.0; bv[q] = 0.0;
#pragma unroll
        for (int s = 0; s < W - 1; ++s) lv[q][s] = 0.0;
        if (g[q].inside) {
            dg[q] = L[g[q].i];
            rdg[q] = 1.0 / dg[q];
#pragma unroll
            for (int s = 1; s < W; ++s) lv[q][s - 1] = L[(int64_t)s * n + g[q].i];
            bv[q] = b_[voff + g[q].i];
            cur[g[q].self] = xin[g[q].i];
        }
    }
    __syncthreads();
    double rmax = 0.0;
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double xn = cur[g[q].self];
            if (k < g[q].kvalid) {
                double acc = bv[q];
#pragma unroll
                for (int s = 0; s < W - 1; ++s) acc = fma(-lv[q][s], cur[g[q].self + t4_off(s)], acc);
                if (k == K - 1 && g[q].owned) rmax = fmax(rmax, fabs(acc - dg[q] * xn));
                xn = acc * rdg[q];
            }
            if (g[q].inside) nxt[g[q].self] = xn;
        }
        __syncthreads();
        double* t = cur; cur = nxt; nxt = t;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (g[q].owned) xout[g[q].i] = cur[g[q].self];
    rmax = block_reduce(rmax, OpMax(), 0.0, smem);
    if (threadIdx.x == 0) {
        if (BIG) bigpart[(int64_t)bz * nwg + wg] = rmax;
        else p[(launch & 1) * FEMFCT_MAX_PARTIALS + wg] = rmax;
    }
}

__global__ void __launch_bounds__(STRIP_T)
k_tile4_cheb(int n, int N, const double* __restrict__ M, const double* __restrict__ b_, const double* __restrict__ ymid_,
             const double* __restrict__ yold_, double* __restrict__ omid_, double* __restrict__ oold_, int K,
             CheOmegas om, double md_scale, ChebIO cio) {
    constexpr int W = 7;
    extern __shared__ double lds[];
    if (cio.mat) M = cio.mat + (int64_t)blockIdx.z * cio.mat_bs;
    if (cio.scale_dev) md_scale = cio.scale_dev[blockIdx.z];
    const double* omd = cio.om_dev ? cio.om_dev + (int64_t)blockIdx.z * cio.om_bs + cio.k0 : nullptr;
    if (cio.mid_ref.base) ymid_ = vec_ptr(cio.mid_ref) + (int64_t)blockIdx.z * cio.mid_bs - (int64_t)blockIdx.z * n;
    if (cio.out_ref.base) omid_ = const_cast<double*>(vec_ptr(cio.out_ref)) + (int64_t)blockIdx.z * cio.out_bs - (int64_t)blockIdx.z * n;
    double* y_old = lds;
    double* y_mid = lds + T4_BUF;
    double* y_new = lds + 2 * T4_BUF;
    const int64_t voff = (int64_t)blockIdx.z * n;
    t4_zero(lds, 3 * T4_BUF);
    __syncthreads();
    Tile4Node g[4];
    double mv[4][W - 1], md[4], rmd[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        g[q] = tile4_node(N, q);
        md[q] = 1.0; rmd[q] = 1.0; bv[q] = 0.0;
#pragma unroll
        for (int s = 0; s < W - 1; ++s) mv[q][s] = 0.0;
        if (g[q].inside) {
            md[q] = M[g[q].i];
            rmd[q] = 1.0 / (md_scale * md[q]);
#pragma unroll
            for (int s = 1; s < W; ++s) mv[q][s - 1] = M[(int64_t)s * n + g[q].i];
            bv[q] = b_[voff + g[q].i];
            if (ymid_) y_mid[g[q].self] = ymid_[voff + g[q].i];
            if (yold_) y_old[g[q].self] = yold_[voff + g[q].i];
        }
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double ymv = y_mid[g[q].self];
            double yn = ymv;
            if (k < g[q].kvalid) {
                double acc = md[q] * ymv;
#pragma unroll
                for (int s = 0; s < W - 1; ++s) acc = fma(mv[q][s], y_mid[g[q].self + t4_off(s)], acc);
                const double z = (bv[q] - acc) * rmd[q];
                const double yov = y_old[g[q].self];
                const double wk = omd ? omd[k] : om.w[k];
                yn = wk * (z + ymv - yov) + yov;
            }
            if (g[q].inside) y_new[g[q].self] = yn;
        }
        __syncthreads();
        double* t = y_old; y_old = y_mid; y_mid = y_new; y_new = t;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (g[q].owned) {
            omid_[voff + g[q].i] = y_mid[g[q].self];
            if (oold_) oold_[voff + g[q].i] = y_old[g[q].self];
        }
}

// -------------------------------------------------------------------------------------------
// Register-resident variant of the 64 x 64 patch kernels.  A wave owns a 64-wide, 4-row strip; a
// thread keeps its four (vertically adjacent) iterate values in registers.  Vertical neighbours are the
// thread's own registers, E/NE/W/SW come from the neighbouring lanes by wave-wide DPP shifts
// (wave_shl:1 / wave_shr:1, tools/dpp_probe.hip) and only the strip's first and last row go through LDS
// (2 writes + 2 reads per thread and sweep instead of 24 reads + 4 writes): the sweeps of the LDS
// variant cost ~10 us each over a 2049^2 mesh and were not overlapped with the matrix load
// (1 workgroup per CU at 104 VGPRs).  Same expressions in the same order => bitwise the same iterates.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ double dpp_from_next(double v) {   // lane i <- lane i+1 (lane 63 <- 0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_prev(double v) {   // lane i <- lane i-1 (lane 0 <- 0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

struct Strip4Node {
    int i, kvalid;
    bool inside, owned;
};

// Patch of this workgroup.  remap: consecutive patches (row-major) go to workgroups of the SAME XCD -- dispatch deals
// workgroups round-robin over the 8 XCDs, so without it two x-neighbouring patches, which share 2H of their 64
// columns, always sit under different L2s and each XCD fetches the shared halo lines for itself.
__device__ __forceinline__ unsigned strip4_patch(int remap) {
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (remap && gridDim.z == 1) {
        const int G = gridDim.x * gridDim.y;
        const int p = xcd_remap((int)(by * gridDim.x + bx), G);
        by = p / gridDim.x;
        bx = p - by * gridDim.x;
    }
    return bx | (by << 16);
}

__device__ __forceinline__ Strip4Node strip4_node(int N, int r, int H, unsigned pxy, int lx, int st) {
    Strip4Node g;
    const int T = T4_L - 2 * H;          // halo depth is a launch parameter (8..10): K <= H sweeps per launch
    const int ly = 4 * st + r;
    const int x0 = (int)(pxy & 0xffffu) * T - H, y0 = (int)(pxy >> 16) * T - H;
    const int gx = x0 + lx, gy = y0 + ly;
    g.inside = gx >= 0 && gx < N && gy >= 0 && gy < N;
    g.i = g.inside ? gy * N + gx : 0;
    g.owned = g.inside && lx >= H && lx < H + T && ly >= H && ly < H + T;
    g.kvalid = 0;                        // (unused: see the note on validity guards in k_strip4_jacobi)
    return g;
}
__device__ __forceinline__ Strip4Node strip4_node(int N, int r, int H, unsigned pxy) {
    return strip4_node(N, r, H, pxy, threadIdx.x & 63, threadIdx.x >> 6);
}

// the six neighbour values of the thread's node r from registers / lane shifts / the two LDS rows
#define STRIP4_NEIGHBOURS(X, ABOVE, BELOW)                                                          \
    double e_[4], w_[4];                                                                            \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) { e_[r] = dpp_from_next(X[r]); w_[r] = dpp_from_prev(X[r]); } \
    const double ea_ = dpp_from_next(ABOVE), wb_ = dpp_from_prev(BELOW)
#define STRIP4_NB(X, ABOVE, BELOW, r, s)                                                            \
    ((s) == 0 ? e_[r] : (s) == 1 ? ((r) < 3 ? e_[((r) + 1) & 3] : ea_) : (s) == 2 ? ((r) < 3 ? X[((r) + 1) & 3] : ABOVE) \
     : (s) == 3 ? w_[r] : (s) == 4 ? ((r) > 0 ? w_[((r) + 3) & 3] : wb_) : ((r) > 0 ? X[((r) + 3) & 3] : BELOW))
// Accumulation order of the Jacobi rows in the 64-patch family: by opposing pairs -- (E, W), (NE, SW), (N, S).  The
// upwind low-order operator has at most one non-zero entry per pair (d_ij = max(0, a_ij, a_ji) cancels the other),
// so k_strip_jacobi_pair_walk, which keeps ONE value per pair and selects the neighbour, adds the same non-zero
// terms in the same order: the skipped terms are fma(-0, x, acc) = acc.  Same bits from every kernel of the family.
#define STRIP4_ROW_FMAS(ACC, LV, X, ABOVE, BELOW, r)                                                \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                              \
        ACC = fma(-LV[p_], STRIP4_NB(X, ABOVE, BELOW, r, p_), ACC);                                 \
        ACC = fma(-LV[p_ + 3], STRIP4_NB(X, ABOVE, BELOW, r, p_ + 3), ACC);                         \
    }

// MODE 0: residual partials reduced by the next launch's workgroups; 1: more workgroups than in-kernel partials
// (separate k_reduce_resid); 2: ONE workgroup covers the whole mesh and stops by itself (check_every > 0) -- its own
// instantiation so that the multi-patch bandwidth kernels carry neither the branch nor its block reduction.
// A launch of the 64-patch kernels is a matrix-load phase (HBM bound, ~2/3 of the time) followed by a register-resident
// sweep phase (VALU bound) with one workgroup per CU, so the two phases of a CU never overlap -- and since every CU
// starts together, all CUs load together and sweep together: HBM idles during the sweeps, the VALUs during the loads.
// Holding back HALF of the first round's workgroups by about one load phase puts the two halves of the chip in
// opposite phases for the rest of the launch (every workgroup takes the same time, so the offset persists): one half
// streams at up to twice its share of HBM while the other computes.  `stagger` = (ticks of the 100 MHz constant
// clock) | (bit << 24): the workgroups of the first round (linear index < 256) whose index has `bit` set are held
// (dispatch order deals consecutive workgroups round-robin over the 8 XCDs, so bits 0-2 select XCDs and bits 3-7
// step through an XCD's 32 CUs; bit 7 = the second half of every XCD).
// Placement only affects speed, never results; the wait is bounded (every wave leaves after `ticks`).
__device__ __forceinline__ void stagger_first_round(int stagger) {
    if (stagger <= 0) return;
    const unsigned wgl = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wgl >= 256u) return;
    const int pat = stagger >> 24, ticks = stagger & 0xffffff;
    // pat < 8: two groups by one index bit; pat >= 8: (pat - 6) groups of CU quads, group g held g * ticks
    const int mult = pat < 8 ? (int)((wgl >> pat) & 1u) : (int)((wgl >> 5) % (unsigned)(pat - 6));
    if (mult == 0) return;
    const int64_t wait = (int64_t)ticks * mult;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) < wait) __builtin_amdgcn_s_sleep(16);
}

template <int MODE>
__global__ void __launch_bounds__(STRIP_T)
k_strip4_jacobi(int n, int N, const double* __restrict__ L_, const double* __restrict__ b_, double* __restrict__ xa_,
                double* __restrict__ xb_, double* __restrict__ part, StepCtl* __restrict__ ctl_, int launch, int K,
                int g_build, double rel_tol, double* __restrict__ bigpart, int H, int check_every, int stagger,
                const uint8_t* __restrict__ lmask, int remap) {
    constexpr int W = 7;
    __shared__ double top[2][16][64], bot[2][16][64];
    __shared__ double smem[32];
    const int bz = blockIdx.z;
    StepCtl* ctl = ctl_ + bz;
    if (ctl->done) return;
    stagger_first_round(stagger);
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
    double bnorm;
    if (launch == 0) {
        bnorm = reduce_partials(p + 2 * FEMFCT_MAX_PARTIALS, g_build, OpMax(), 0.0, smem);
        double rsmin = reduce_partials(p + 3 * FEMFCT_MAX_PARTIALS, g_build, OpMin(), INFINITY, smem);
        if (wg == 0 && threadIdx.x == 0) {
            ctl->bnorm = bnorm;
            ctl->min_rowsum = rsmin;
            if (!(rsmin > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        }
    } else {
        bnorm = ctl->bnorm;
        double rmax = (MODE == 1) ? ctl->rs[(launch - 1) & 1]
                                  : reduce_partials(p + ((launch - 1) & 1) * FEMFCT_MAX_PARTIALS, nwg, OpMax(), 0.0, smem);
        if (rmax <= rel_tol * bnorm) {
            if (wg == 0 && threadIdx.x == 0) {
                ctl->done = 1; ctl->parity = launch & 1; ctl->iters = launch * K; ctl->flags |= FEMFCT_FLAG_COARSE_ITERS;
                ctl->resid = bnorm > 0.0 ? rmax / bnorm : 0.0;
            }
            return;
        }
    }
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* L = L_ + moff;
    const double* xin = ((launch & 1) ? xb_ : xa_) + voff;
    double* xout = ((launch & 1) ? xa_ : xb_) + voff;
    const int lx = threadIdx.x & 63, st = threadIdx.x >> 6;
    Strip4Node g[4];
    // rows pre-scaled by 1 / l_ii: x_new = bs - sum ls * x_nb;  residual of the input iterate = l_ii |x_new - x|
    double lv[4][W - 1], dg[4], bv[4], x[4];
    // Zero masks (k_build_low: one byte per node, bit s-1 = l_(i,s) != 0): a lane whose entry is exactly zero reads
    // one shared zero word instead of its own, so a 128-byte line of L whose 16 entries all vanish is never requested.
    // Each lane fetches the mask bytes of its four nodes (64 contiguous bytes per wave and row), then issues its row
    // loads; nothing waits on a value of L until all four rows are in flight.
    const double* zero = reinterpret_cast<const double*>(lmask) - 1;   // the word in front of the masks holds 0
    const int64_t zoff = zero - L;                                     // (flat global address space)
    unsigned nzbits[4];
    const unsigned pxy = strip4_patch(remap);
#pragma unroll
    for (int r = 0; r < 4; ++r) { g[r] = strip4_node(N, r, H, pxy); nzbits[r] = 0x3fu; }
    // order of issue: the mask words, then the 12 loads that do not depend on them (diagonal, b, x), and only then
    // -- after waiting for the (older) mask loads alone -- the 24 masked loads of L: the mask round trip is covered
    if (lmask) {
#pragma unroll
        for (int r = 0; r < 4; ++r) nzbits[r] = lmask[voff + g[r].i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = g[r].i;              // lanes outside the mesh read node 0's (finite) data and are zeroed below
        dg[r] = L[i];
        bv[r] = b_[voff + i];
        x[r] = xin[i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = g[r].i;
#pragma unroll
        for (int s = 1; s < W; ++s) {
            const int64_t off = ((nzbits[r] >> (s - 1)) & 1u) ? (int64_t)s * n + i : zoff;   // one load either way
            lv[r][s - 1] = L[off];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double rdg = 1.0 / dg[r];
#pragma unroll
        for (int s = 0; s < W - 1; ++s) lv[r][s] = g[r].inside ? lv[r][s] * rdg : 0.0;
        bv[r] = g[r].inside ? bv[r] * rdg : 0.0;
        x[r] = g[r].inside ? x[r] : 0.0;
        dg[r] = g[r].inside ? dg[r] : 1.0;
    }
    double rmax = 0.0;
    for (int k = 0; k < K; ++k) {
        const int par = k & 1;
        bot[par][st][lx] = x[0];
        top[par][st][lx] = x[3];
        __syncthreads();
        const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;
        const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;
        STRIP4_NEIGHBOURS(x, above, below);
        double xn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xn[r] = x[r];
            double acc = bv[r];
            STRIP4_ROW_FMAS(acc, lv[r], x, above, below, r);
            // no validity guard: a node at distance d from the patch border is exact after k <= d sweeps whatever
            // the nodes further out hold (they stay bounded: rows are diagonally dominant, outside rows are zero)
            if (k == K - 1 && g[r].owned) rmax = fmax(rmax, dg[r] * fabs(acc - x[r]));
            xn[r] = acc;
        }
        // one workgroup = the whole mesh (check_every > 0): the residual of this sweep's input iterate is known
        // to the workgroup, so it stops by itself -- exact sweep counts, the budget is only an upper bound
        if (MODE == 2 && (k % check_every) == check_every - 1 && k < K - 1) {
            double rk = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (g[r].owned) rk = fmax(rk, dg[r] * fabs(xn[r] - x[r]));
            rk = block_reduce(rk, OpMax(), 0.0, smem);
            if (rk <= rel_tol * bnorm) {                 // x (the input of this sweep) already meets the tolerance
                if (threadIdx.x == 0) {
                    ctl->done = 1; ctl->parity = (launch + 1) & 1; ctl->iters = launch * K + k;
                    ctl->resid = bnorm > 0.0 ? rk / bnorm : 0.0;
                }
                break;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = xn[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (g[r].owned) xout[g[r].i] = x[r];
    rmax = block_reduce(rmax, OpMax(), 0.0, smem);
    if (threadIdx.x == 0) {
        if (MODE == 1) bigpart[(int64_t)bz * nwg + wg] = rmax;
        else p[(launch & 1) * FEMFCT_MAX_PARTIALS + wg] = rmax;
    }
}

// The bandwidth-regime Jacobi launch as a WALK: `gridDim.x` persistent workgroups per batch member, each taking a
// contiguous run of the patches in column-major order (down a column of patches, then the next column).  Vertically
// adjacent patches share 2H of their 64 rows; the walker keeps the scaled rows (six off-diagonals, diagonal, b, input
// iterate) of those rows in LDS when it steps down, so that they are fetched from memory once instead of twice: the
// halo re-read remains in x only, (64/T) instead of (64/T)^2 of the matrix per launch.  The carried values are the
// registers the previous patch computed from the same addresses with the same expressions: results are those of
// k_strip4_jacobi<0> bit for bit.  Residual partials: one per walker (max over its patches).
constexpr int WALK_CARRY_ROWS = 20;     // 2 * H, H <= 10
__global__ void __launch_bounds__(STRIP_T)
k_strip4_jacobi_walk(int n, int N, const double* __restrict__ L_, const double* __restrict__ b_, double* __restrict__ xa_,
                     double* __restrict__ xb_, double* __restrict__ part, StepCtl* __restrict__ ctl_, int launch, int K,
                     int g_build, double rel_tol, int H, int stagger, const uint8_t* __restrict__ lmask,
                     int npy, int npatch, int carry_on, int up) {
    constexpr int W = 7;
    __shared__ double top[2][16][64], bot[2][16][64];
    __shared__ double carry[WALK_CARRY_ROWS][W + 2][64];
    __shared__ double smem[32];
    const int bz = blockIdx.z;
    StepCtl* ctl = ctl_ + bz;
    if (ctl->done) return;
    stagger_first_round(stagger);
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    const int nwg = gridDim.x, wg = blockIdx.x;
    double bnorm;
    if (launch == 0) {
        bnorm = reduce_partials(p + 2 * FEMFCT_MAX_PARTIALS, g_build, OpMax(), 0.0, smem);
        double rsmin = reduce_partials(p + 3 * FEMFCT_MAX_PARTIALS, g_build, OpMin(), INFINITY, smem);
        if (wg == 0 && threadIdx.x == 0) {
            ctl->bnorm = bnorm;
            ctl->min_rowsum = rsmin;
            if (!(rsmin > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        }
    } else {
        bnorm = ctl->bnorm;
        double rprev = reduce_partials(p + ((launch - 1) & 1) * FEMFCT_MAX_PARTIALS, nwg, OpMax(), 0.0, smem);
        if (rprev <= rel_tol * bnorm) {
            if (wg == 0 && threadIdx.x == 0) {
                ctl->done = 1; ctl->parity = launch & 1; ctl->iters = launch * K; ctl->flags |= FEMFCT_FLAG_COARSE_ITERS;
                ctl->resid = bnorm > 0.0 ? rprev / bnorm : 0.0;
            }
            return;
        }
    }
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* L = L_ + moff;
    const double* xin = ((launch & 1) ? xb_ : xa_) + voff;
    double* xout = ((launch & 1) ? xa_ : xb_) + voff;
    const int st = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double* zero = reinterpret_cast<const double*>(lmask) - 1;   // the word in front of the masks holds 0
    const int64_t zoff = zero - L;
    const int T = T4_L - 2 * H;
    const int q0 = (int)((int64_t)wg * npatch / nwg), q1 = (int)((int64_t)(wg + 1) * npatch / nwg);
    double rmax = 0.0;
    unsigned nznext = 0;
    const int lx0 = threadIdx.x & 63;
    // up: the run is walked backwards (up the columns).  Successive launches of a solve alternate the direction, so that a
    // launch starts on the rows the previous one touched last: what of the matrix is still in the Infinity Cache is hit
    // before it is evicted (a cyclic sweep over more than the cache holds would otherwise miss every time).
    for (int j = 0; j < q1 - q0; ++j) {
        const int q = up ? q1 - 1 - j : q0 + j;
        // the lane index is made opaque per patch: otherwise the ~20 LDS / global addresses derived from it are hoisted
        // out of the walk as loop invariants and spilled -- and a spill in the load phase waits on a load (vmcnt is one
        // in-order queue), which serialises the mask and row round trips
        int lx = lx0;
        asm volatile("" : "+v"(lx));
        const int px = q / npy, py = q - px * npy;
        // the previous / next patch of this walker is the vertical neighbour (not across a column end)
        const bool cin = carry_on && j > 0 && (up ? py + 1 < npy : py > 0);
        const bool cout = carry_on && j + 1 < q1 - q0 && (up ? py > 0 : py + 1 < npy);
        const unsigned pxy = (unsigned)px | ((unsigned)py << 16);
        Strip4Node g[4];
        double lv[4][W - 1], dg[4], bv[4], x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = strip4_node(N, r, H, pxy, lx, st);
        // a wave takes its four rows either all from the carry or all from memory (for H = 9 the wave that straddles
        // row 2H re-reads two rows: 3 % of the patch)
        if (cin && (up ? 4 * st >= T : 4 * st + 3 < 2 * H)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * st + r - (up ? T : 0);
#pragma unroll
                for (int s = 0; s < W - 1; ++s) lv[r][s] = carry[c][s][lx];
                dg[r] = carry[c][W - 1][lx];
                bv[r] = carry[c][W][lx];
                x[r] = carry[c][W + 1][lx];
            }
        } else {
            // the four mask bytes first, then all 36 loads of the four rows at once
            unsigned nzbits[4] = {0x3fu, 0x3fu, 0x3fu, 0x3fu};
            if (lmask) {
                if (j > 0) {           // fetched during the previous patch's sweeps
#pragma unroll
                    for (int r = 0; r < 4; ++r) nzbits[r] = (nznext >> (8 * r)) & 0xffu;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) nzbits[r] = lmask[voff + g[r].i];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = g[r].i;
                dg[r] = L[i];
                bv[r] = b_[voff + i];
                x[r] = xin[i];
#pragma unroll
                for (int s = 1; s < W; ++s) {
                    const int64_t off = ((nzbits[r] >> (s - 1)) & 1u) ? (int64_t)s * n + i : zoff;
                    lv[r][s - 1] = L[off];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double rdg = 1.0 / dg[r];
#pragma unroll
                for (int s = 0; s < W - 1; ++s) lv[r][s] = g[r].inside ? lv[r][s] * rdg : 0.0;
                bv[r] = g[r].inside ? bv[r] * rdg : 0.0;
                x[r] = g[r].inside ? x[r] : 0.0;
                dg[r] = g[r].inside ? dg[r] : 1.0;
            }
        }
        __syncthreads();                   // carry consumed; the previous patch's last sweep has left top / bot
        if (cout) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * st + r - (up ? 0 : T);       // walking down the top 2H rows are kept, walking up the bottom ones
                if (c < 0 || c >= 2 * H) continue;
#pragma unroll
                for (int s = 0; s < W - 1; ++s) carry[c][s][lx] = lv[r][s];
                carry[c][W - 1][lx] = dg[r];
                carry[c][W][lx] = bv[r];
                carry[c][W + 1][lx] = x[r];
            }
        }
        // The next patch's mask bytes are requested before the sweeps: its 36 row loads can then go out right behind
        // this patch's stores instead of waiting a round trip for the masks (which, vmcnt being one in-order queue,
        // would itself wait for the stores to be acknowledged).
        unsigned nzn[4] = {0, 0, 0, 0};
        if (lmask && j + 1 < q1 - q0) {
            const int qn = up ? q - 1 : q + 1, pxn = qn / npy, pyn = qn - pxn * npy;
            const unsigned pxyn = (unsigned)pxn | ((unsigned)pyn << 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) nzn[r] = lmask[voff + strip4_node(N, r, H, pxyn, lx, st).i];
        }
        for (int k = 0; k < K; ++k) {
            const int par = k & 1;
            bot[par][st][lx] = x[0];
            top[par][st][lx] = x[3];
            __syncthreads();
            const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;
            const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;
            STRIP4_NEIGHBOURS(x, above, below);
            double xn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = bv[r];
                STRIP4_ROW_FMAS(acc, lv[r], x, above, below, r);
                if (k == K - 1 && g[r].owned) rmax = fmax(rmax, dg[r] * fabs(acc - x[r]));
                xn[r] = acc;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = xn[r];
        }
        nznext = nzn[0] | (nzn[1] << 8) | (nzn[2] << 16) | (nzn[3] << 24);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (g[r].owned) xout[g[r].i] = x[r];
    }
    rmax = block_reduce(rmax, OpMax(), 0.0, smem);
    if (threadIdx.x == 0) p[(launch & 1) * FEMFCT_MAX_PARTIALS + wg] = rmax;
}

// The walking Jacobi launch with ONE VALUE PER OPPOSING PAIR -- two workgroups to a compute unit.
// A launch of k_strip4_jacobi_walk is a load phase at memory speed followed by register-resident sweeps, and with 115 VGPRs
// only one 1024-thread workgroup fits a CU, so the two phases never overlap there.  For an upwind operator the six
// off-diagonals of a row are three values: of each opposing pair (E, W), (NE, SW), (N, S) at most one entry is non-zero
// (the zero mask k_build_low publishes says which).  This kernel keeps the non-zero value of each pair plus three select
// bits: 12 instead of 18 registers per node.  A workgroup of NST waves owns a 64 x (R NST) patch, each thread R vertically
// adjacent nodes (fewer strips: fewer LDS edge rows and barrier participants per node), sized so that TWO workgroups fit
// a CU -- the product shape is 6 rows x 8 waves (512 threads, a 64 x 48 patch) at <= 128 VGPRs (four waves per SIMD) and
// ~81.5 KB of LDS -- each walking its own run of patches, one's loads under the other's sweeps.  The FMAs run in the
// family's pair order (STRIP4_ROW_FMAS) with the neighbour selected per lane: bit-identical to k_strip4_jacobi<0>.
// Patches advance by T = 64 - 2H columns and TY = R NST - 2H rows.
// Rows with BOTH entries of a pair (where a wind component changes sign) go to a patch-wide pool in LDS (below); only a
// patch with more of them than the pool holds (a diffusive or otherwise non-upwind operator) raises FEMFCT_FLAG_ROW_PAIRS
// in the step record, and the host repeats the sweep with the full-row kernels (femfct_run_sweep); outside the
// trajectory sweeps the kernel is not used.
// The measurement knobs of round 3 (start-up stagger, phase timestamps, priority scheme, work split between the two
// workgroups of a CU: all measured neutral) and the other patch shapes exist in the -DFEMFCT_TUNING build only.
constexpr int PAIR_CARRY_ROWS = 18;       // 2 * H, H <= 9 (the Jacobi launches: 36 sweeps = 4 x 9)
constexpr int PAIR_POOL = 160;            // nodes of a patch with both entries of some pair (see below): 6 doubles each
#ifdef FEMFCT_TUNING
#define PAIR_TUNING_PARAMS , int stagger_ticks, unsigned long long* __restrict__ trace, int prio_mode, int first_half_pct
#define PAIR_TUNING_ARGS , ctx->pair_stagger, ctx->d_pair_trace, ctx->pair_prio, ctx->pair_split
#else
#define PAIR_TUNING_PARAMS
#define PAIR_TUNING_ARGS
#endif
template <int R, int NST>
__global__ void __launch_bounds__(64 * NST, (2 * NST + 3) / 4)
k_strip_jacobi_pair_walk(int n, int N, const double* __restrict__ L_, const double* __restrict__ b_, double* __restrict__ xa_,
                         double* __restrict__ xb_, double* __restrict__ part, StepCtl* __restrict__ ctl_, int launch, int K,
                         int g_build, double rel_tol, int H, const uint8_t* __restrict__ lmask, int npy, int npatch,
                         int up, int64_t zero_index PAIR_TUNING_PARAMS) {
#ifndef FEMFCT_TUNING
    constexpr int stagger_ticks = 0, prio_mode = 0, first_half_pct = 50;
    constexpr unsigned long long* trace = nullptr;
#endif
    constexpr int W = 7, PR = R * NST;
    __shared__ double top[2][NST][64], bot[2][NST][64];
    __shared__ double carry[PAIR_CARRY_ROWS][6][64];        // three pair values, diagonal, b, input iterate
    __shared__ uint8_t carry_nz[PAIR_CARRY_ROWS][64];
    // Rows with BOTH entries of some pair (along the curves where a wind component changes sign: ~0.1 % of the rows for a
    // smooth control) do not fit three values.  Their six scaled off-diagonals live here -- a patch-wide pool, each
    // thread's nodes contiguous from `xstart` on in row order, found again by counting bits (no search, one register) --
    // and the generic row update recomputes such a node's chain from them (all six FMAs, in the family's order).
    __shared__ double pool_val[PAIR_POOL][6];
    __shared__ unsigned pool_node[PAIR_POOL];       // node index | mask byte << 24 ... (index < 2^24 is not assumed: two words)
    __shared__ uint8_t pool_mask[PAIR_POOL];
    __shared__ int pool_cnt;
    __shared__ double smem[32];
    // two workgroups per CU is the point of this kernel: growing PAIR_POOL or PAIR_CARRY_ROWS must not cost that silently
    static_assert(2 * (sizeof(top) + sizeof(bot) + sizeof(carry) + sizeof(carry_nz) + sizeof(pool_val) + sizeof(pool_node) +
                       sizeof(pool_mask) + sizeof(smem) + 64) <= 160 * 1024 || NST != 8,
                  "k_strip_jacobi_pair_walk: two workgroups no longer fit the LDS of a compute unit");
    const int bz = blockIdx.z;
    StepCtl* ctl = ctl_ + bz;
    if (ctl->done) return;
    double* p = part + (int64_t)bz * 4 * FEMFCT_MAX_PARTIALS;
    const int nwg = gridDim.x, wg = blockIdx.x;
    double bnorm;
    if (launch == 0) {
        bnorm = reduce_partials(p + 2 * FEMFCT_MAX_PARTIALS, g_build, OpMax(), 0.0, smem);
        double rsmin = reduce_partials(p + 3 * FEMFCT_MAX_PARTIALS, g_build, OpMin(), INFINITY, smem);
        if (wg == 0 && threadIdx.x == 0) {
            ctl->bnorm = bnorm;
            ctl->min_rowsum = rsmin;
            if (!(rsmin > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        }
    } else {
        bnorm = ctl->bnorm;
        double rprev = reduce_partials(p + ((launch - 1) & 1) * FEMFCT_MAX_PARTIALS, nwg, OpMax(), 0.0, smem);
        if (rprev <= rel_tol * bnorm) {
            if (wg == 0 && threadIdx.x == 0) {
                ctl->done = 1; ctl->parity = launch & 1; ctl->iters = launch * K; ctl->flags |= FEMFCT_FLAG_COARSE_ITERS;
                ctl->resid = bnorm > 0.0 ? rprev / bnorm : 0.0;
            }
            return;
        }
    }
    // Two workgroups share a CU.  Started together they load together and sweep together: nothing overlaps.  The second
    // half of the grid (dispatched onto CUs that already hold one workgroup) is held back by about half a patch period,
    // after which the two stay in opposite phases (both take the same time per patch).  Bounded wait; speed only.
    if (stagger_ticks > 0 && wg >= nwg / 2) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) < (int64_t)stagger_ticks) __builtin_amdgcn_s_sleep(16);
    }
    const int64_t moff = (int64_t)bz * W * n, voff = (int64_t)bz * n;
    const double* L = L_ + moff;
    const double* bvec = b_ + voff;
    const uint8_t* msk = lmask + voff;
    const double* xin = ((launch & 1) ? xb_ : xa_) + voff;
    double* xout = ((launch & 1) ? xa_ : xb_) + voff;
    const unsigned zo = (unsigned)(zero_index - moff);      // L[zo] == 0: read in place of a pair without an entry
    const unsigned un = (unsigned)n;
    const int st = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = T4_L - 2 * H, TY = PR - 2 * H;
    // The two workgroups of a CU do not run at the same speed: issue arbitration prefers the older wave, so the workgroup
    // dispatched second (the grid's second half) falls behind and ends up sweeping alone -- latency bound -- at the end.
    // first_half_pct: share of the patches given to the first half of the walkers (50 = even); prio_mode 1: the two
    // alternate in raised priority from patch to patch, 2: the younger one raised throughout.
    int q0, q1;
    {
        const int half = nwg / 2;
        const int np0 = (nwg >= 2 && first_half_pct != 50) ? (int)((int64_t)npatch * first_half_pct / 100) : (int)((int64_t)half * npatch / nwg);
        if (wg < half) { q0 = (int)((int64_t)wg * np0 / half); q1 = (int)((int64_t)(wg + 1) * np0 / half); }
        else { q0 = np0 + (int)((int64_t)(wg - half) * (npatch - np0) / (nwg - half)); q1 = np0 + (int)((int64_t)(wg - half + 1) * (npatch - np0) / (nwg - half)); }
    }
    double rmax = 0.0;
    unsigned viol = 0;
    const int lx0 = threadIdx.x & 63;
    // first row index (patch-local) of this wave's strip; a wave takes its R rows either all from the carry or all from
    // memory (the strip that straddles row 2H re-reads its carried rows)
    const int ly0 = R * st;
    const bool strip_in_carry = up ? ly0 >= TY : ly0 + R <= 2 * H;
    // mask bytes of a patch's rows, four to a register (rows outside the mesh: node 0's, zeroed on use)
    constexpr int NZW = (R + 3) / 4;
    auto patch_masks = [&](int qq, unsigned (&out)[NZW], int lane) {
        const int pxn = qq / npy, pyn = qq - pxn * npy;
        const int gxn = pxn * T - H + lane, y0n = pyn * TY - H + ly0;
        const int gxnc = (gxn >= 0 && gxn < N) ? gxn : 0;
#pragma unroll
        for (int w = 0; w < NZW; ++w) out[w] = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = y0n + r;
            const int i = (gy >= 0 && gy < N) ? gy * N + gxnc : 0;
            out[r >> 2] |= (unsigned)msk[i] << (8 * (r & 3));
        }
    };
    unsigned nznext[NZW];
    if (q1 > q0) patch_masks(up ? q1 - 1 : q0, nznext, lx0);
    for (int j = 0; j < q1 - q0; ++j) {
        const int q = up ? q1 - 1 - j : q0 + j;
        int lx = lx0;
        asm volatile("" : "+v"(lx));       // per-patch addresses are not loop invariants (see k_strip4_jacobi_walk)
        const int px = q / npy, py = q - px * npy;
        const bool cin = j > 0 && (up ? py + 1 < npy : py > 0);
        const bool cout = j + 1 < q1 - q0 && (up ? py > 0 : py + 1 < npy);
        if (prio_mode == 1) { if ((j & 1) ^ (wg >= nwg / 2 ? 1 : 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        else if (prio_mode == 2 && j == 0 && wg >= nwg / 2) __builtin_amdgcn_s_setprio(1);
        // FEMFCT_PAIR_TRACE (tuning): 100 MHz timestamps of workgroup wg at patch start / rows in registers / after the
        // carry barrier / after the sweeps / after the stores were issued -- 5 words per patch, 16 patches per walker
        unsigned long long* tr = (trace && threadIdx.x == 0 && j < 16 && launch == 1) ? trace + ((int64_t)wg * 16 + j) * 5 : nullptr;
        if (tr) tr[0] = __builtin_amdgcn_s_memrealtime();
        const int x0 = px * T - H, y0 = py * TY - H + ly0;             // y0: mesh row of the strip's first row (uniform)
        const int gx = x0 + lx;
        const bool xin_mesh = gx >= 0 && gx < N;
        const bool xowned = xin_mesh && lx >= H && lx < H + T;
        const int gxc = xin_mesh ? gx : 0;
        double v[R][3], dg[R], bv[R], x[R];
        unsigned nzp[NZW];
        if (cin && strip_in_carry) {
            const int c0 = ly0 - (up ? TY : 0);
#pragma unroll
            for (int w = 0; w < NZW; ++w) nzp[w] = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) v[r][pr] = carry[c0 + r][pr][lx];
                dg[r] = carry[c0 + r][3][lx];
                bv[r] = carry[c0 + r][4][lx];
                x[r] = carry[c0 + r][5][lx];
                nzp[r >> 2] |= (unsigned)carry_nz[c0 + r][lx] << (8 * (r & 3));
            }
        } else {
            // straight-line: 6 R loads in flight together, the entry of each pair chosen by the (prefetched) mask bits
            unsigned ia[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int gy = y0 + r;
                const bool row_in = gy >= 0 && gy < N;                 // uniform
                ia[r] = row_in ? (unsigned)(gy * N + gxc) : 0u;
                if (!(row_in && xin_mesh)) nznext[r >> 2] &= ~(0xffu << (8 * (r & 3)));
            }
#pragma unroll
            for (int w = 0; w < NZW; ++w) nzp[w] = nznext[w];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned m = nzp[r >> 2] >> (8 * (r & 3));
                dg[r] = L[ia[r]];
                bv[r] = bvec[ia[r]];
                x[r] = xin[ia[r]];
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
                    const unsigned lo = (m >> pr) & 1u, hi = (m >> (pr + 3)) & 1u;
                    const unsigned slot = lo ? (unsigned)(pr + 1) : (unsigned)(pr + 4);
                    const unsigned off = (lo | hi) ? slot * un + ia[r] : zo;
                    v[r][pr] = L[off];
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int gy = y0 + r;
                const bool inside = xin_mesh && gy >= 0 && gy < N;
                const double rdg = 1.0 / dg[r];
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) v[r][pr] = inside ? v[r][pr] * rdg : 0.0;
                bv[r] = inside ? bv[r] * rdg : 0.0;
                x[r] = inside ? x[r] : 0.0;
                dg[r] = inside ? dg[r] : 1.0;
            }
        }
        if (tr) tr[1] = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) pool_cnt = 0;        // (the previous patch's sweeps no longer read the counter, only their entries)
        __syncthreads();                   // carry consumed; the previous patch's last sweep has left top / bot and the pool
        if (tr) tr[2] = __builtin_amdgcn_s_memrealtime();
        // this thread's rows with a double pair (one bit per row) and their records in the pool (rare: a dynamic loop that
        // re-reads the row from memory, so that it needs no register array -- same operands, same bits as lv * rdg)
        unsigned dm = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned m = nzp[r >> 2] >> (8 * (r & 3));
            dm |= (((m & (m >> 3)) & 7u) ? 1u : 0u) << r;
        }
        // each such node gets a pool record: the thread only enters (node, mask) -- after the next barrier the whole
        // workgroup fills the records' 6 scaled coefficients, one coefficient per thread, all loads in flight together
        // (a thread filling its own records would wait two dependent memory round trips per node, up to R nodes in a row)
        int xstart = 0;
        if (dm) {
            const int cnt = __popc(dm);
            xstart = atomicAdd(&pool_cnt, cnt);
            if (xstart + cnt > PAIR_POOL) { viol = 1; dm = 0; }
            int e = xstart;
            for (unsigned rest = dm; rest; rest &= rest - 1, ++e) {
                const int r = __ffs(rest) - 1;
                pool_node[e] = (unsigned)((y0 + r) * N + gxc);          // a double pair only exists inside the mesh
                unsigned word = nzp[0];
#pragma unroll
                for (int w = 1; w < NZW; ++w) word = (r >> 2) == w ? nzp[w] : word;
                pool_mask[e] = (uint8_t)(word >> (8 * (r & 3)));
            }
        }
        if (cout) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int c = ly0 + r - (up ? 0 : TY);       // walking down the top 2H rows are kept, walking up the bottom ones
                if (c < 0 || c >= 2 * H) continue;
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) carry[c][pr][lx] = v[r][pr];
                carry[c][3][lx] = dg[r];
                carry[c][4][lx] = bv[r];
                carry[c][5][lx] = x[r];
                carry_nz[c][lx] = (uint8_t)(nzp[r >> 2] >> (8 * (r & 3)));
            }
        }
        // the next patch's mask bytes are requested before the sweeps: its row loads can then go out right behind this
        // patch's stores instead of waiting a round trip for the masks
        if (j + 1 < q1 - q0 && !(cout && strip_in_carry)) patch_masks(up ? q - 1 : q + 1, nznext, lx);
        bot[0][st][lx] = x[0];             // edge rows of the input iterate for the first sweep
        top[0][st][lx] = x[R - 1];
        __syncthreads();
        {
            const int nrec = pool_cnt <= PAIR_POOL ? pool_cnt : 0;   // uniform: patches without a double pair (most) skip this; an overflowed pool (flagged: the sweep is void) holds stale records, which are not touched
            if (nrec > 0) {
                for (int t = threadIdx.x; t < nrec * 6; t += 64 * NST) {
                    const int e = t / 6, sl = t - 6 * e;
                    const unsigned i = pool_node[e], m = pool_mask[e];
                    const double lvv = L[((m >> sl) & 1u) ? (unsigned)(sl + 1) * un + i : zo], dgv = L[i];
                    pool_val[e][sl] = lvv * (1.0 / dgv);
                }
                __syncthreads();
            }
        }
        // Which side each pair takes is a property of the wind direction, i.e. the same for whole regions: a wave whose R x 64
        // nodes all take the same side of every pair (nodes without an entry in a pair do not care: their value is 0) runs a
        // sweep loop specialised for that pattern P (bit p: pair p takes W / SW / S) -- no selects, and only the lane shifts
        // the pattern needs (one per row for E + NE and for W + SW, two otherwise).  Other waves run the generic loop (P = 8).
        unsigned fold = 0;
#pragma unroll
        for (int w = 0; w < NZW; ++w) fold |= nzp[w];
        fold |= fold >> 16;
        fold |= fold >> 8;
        int pat = 0;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            const bool lo_any = __ballot((fold >> pr) & 1u) != 0, hi_any = __ballot((fold >> (pr + 3)) & 1u) != 0;
            pat |= (lo_any && hi_any) ? 8 : (hi_any ? (1 << pr) : 0);
        }
        pat = __builtin_amdgcn_readfirstlane(pat > 7 ? 8 : pat);
        if (trace && launch == 1 && (threadIdx.x & 63) == 0) atomicAdd(trace + (int64_t)FEMFCT_MAX_PARTIALS * 16 * 5 + pat, 1ull);   // FEMFCT_PAIR_TRACE: waves per pattern
        auto sweeps = [&](auto PC) {
            constexpr int P = decltype(PC)::value;
            bool hsel[R][3];
            if constexpr (P == 8) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr) {
                        const unsigned m = nzp[r >> 2] >> (8 * (r & 3));
                        hsel[r][pr] = ((m >> (3 + pr)) & 1u) != 0 && ((m >> pr) & 1u) == 0;
                    }
            }
            // One barrier per sweep, placed so that it is covered: a sweep first updates the strip's two EDGE rows (they need
            // the neighbouring strips' edge rows of the previous sweep, read from LDS), stores the new edge rows for the
            // next sweep, and only then updates its interior rows (own registers only) -- the LDS stores and the arrival of
            // the other waves at the barrier happen under the interior rows' arithmetic.
            auto row = [&](int r, double xr, double xs, double xn) {     // xs / xn: OLD values of the S / N neighbour
                double acc = bv[r];
                if constexpr (P == 8) {
                    // generic row: the neighbour of each pair's entry selected per lane.  A node with a double pair (rare;
                    // lanes without one skip the block) takes its whole chain from its pool record instead.  The index
                    // passes through an empty asm so that these loop-invariant LDS reads stay inside the sweeps.
                    const double ev = dpp_from_next(xr), wv = dpp_from_prev(xr), ne = dpp_from_next(xn), sw = dpp_from_prev(xs);
                    acc = fma(-v[r][0], hsel[r][0] ? wv : ev, acc);      // (E, W)
                    acc = fma(-v[r][1], hsel[r][1] ? sw : ne, acc);      // (NE, SW)
                    acc = fma(-v[r][2], hsel[r][2] ? xs : xn, acc);      // (N, S)
                    if ((dm >> r) & 1u) {
                        int idx = xstart + __popc(dm & ((1u << r) - 1u));
                        asm volatile("" : "+v"(idx));
                        const double* c = pool_val[idx];
                        double a2 = bv[r];
                        a2 = fma(-c[0], ev, a2); a2 = fma(-c[3], wv, a2);
                        a2 = fma(-c[1], ne, a2); a2 = fma(-c[4], sw, a2);
                        a2 = fma(-c[2], xn, a2); a2 = fma(-c[5], xs, a2);
                        acc = a2;
                    }
                } else {
                    acc = fma(-v[r][0], (P & 1) ? dpp_from_prev(xr) : dpp_from_next(xr), acc);
                    acc = fma(-v[r][1], (P & 2) ? dpp_from_prev(xs) : dpp_from_next(xn), acc);
                    acc = fma(-v[r][2], (P & 4) ? xs : xn, acc);
                }
                return acc;
            };
            for (int k = 0; k < K; ++k) {
                const int par = k & 1;
                const bool last = k == K - 1;
                const double above = (st < NST - 1) ? bot[par][st + 1][lx] : 0.0;
                const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;
                double xo[R];
#pragma unroll
                for (int r = 0; r < R; ++r) xo[r] = x[r];
                x[0] = row(0, xo[0], below, xo[1]);
                x[R - 1] = row(R - 1, xo[R - 1], xo[R - 2], above);
                if (!last) {
                    bot[par ^ 1][st][lx] = x[0];
                    top[par ^ 1][st][lx] = x[R - 1];
                }
#pragma unroll
                for (int r = 1; r < R - 1; ++r) {
                    x[r] = row(r, xo[r], xo[r - 1], xo[r + 1]);
                    if (P == 8 && (r & 1)) __builtin_amdgcn_sched_barrier(0);   // generic loop: keep the live lane shifts few
                }
                if (last) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int ly = ly0 + r, gy = y0 + r;
                        if (xowned && ly >= H && ly < H + TY && gy >= 0 && gy < N) rmax = fmax(rmax, dg[r] * fabs(x[r] - xo[r]));
                    }
                } else {
                    __syncthreads();
                }
            }
        };
        switch (pat) {
            case 0: sweeps(std::integral_constant<int, 0>{}); break;
            case 1: sweeps(std::integral_constant<int, 1>{}); break;
            case 2: sweeps(std::integral_constant<int, 2>{}); break;
            case 3: sweeps(std::integral_constant<int, 3>{}); break;
            case 4: sweeps(std::integral_constant<int, 4>{}); break;
            case 5: sweeps(std::integral_constant<int, 5>{}); break;
            case 6: sweeps(std::integral_constant<int, 6>{}); break;
            case 7: sweeps(std::integral_constant<int, 7>{}); break;
            default: sweeps(std::integral_constant<int, 8>{}); break;
        }
        if (tr) tr[3] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int ly = ly0 + r, gy = y0 + r;
            if (xowned && ly >= H && ly < H + TY && gy >= 0 && gy < N) xout[gy * N + gx] = x[r];
        }
        if (tr) tr[4] = __builtin_amdgcn_s_memrealtime();
    }
    if (viol) atomicOr(&ctl->flags, FEMFCT_FLAG_ROW_PAIRS);     // a patch with more double pairs than the pool holds
    rmax = block_reduce(rmax, OpMax(), 0.0, smem);
    if (threadIdx.x == 0) p[(launch & 1) * FEMFCT_MAX_PARTIALS + wg] = rmax;
}

__global__ void __launch_bounds__(STRIP_T)
k_strip4_cheb(int n, int N, const double* __restrict__ M, const double* __restrict__ b_, const double* __restrict__ ymid_,
              const double* __restrict__ yold_, double* __restrict__ omid_, double* __restrict__ oold_, int K,
              CheOmegas om, double md_scale, ChebIO cio, int H, int remap) {
    constexpr int W = 7;
    __shared__ double top[2][16][64], bot[2][16][64];
    if (cio.mat) M = cio.mat + (int64_t)blockIdx.z * cio.mat_bs;
    if (cio.scale_dev) md_scale = cio.scale_dev[blockIdx.z];
    const double* omd = cio.om_dev ? cio.om_dev + (int64_t)blockIdx.z * cio.om_bs + cio.k0 : nullptr;
    if (cio.mid_ref.base) ymid_ = vec_ptr(cio.mid_ref) + (int64_t)blockIdx.z * cio.mid_bs - (int64_t)blockIdx.z * n;
    if (cio.out_ref.base) omid_ = const_cast<double*>(vec_ptr(cio.out_ref)) + (int64_t)blockIdx.z * cio.out_bs - (int64_t)blockIdx.z * n;
    const int64_t voff = (int64_t)blockIdx.z * n;
    const int lx = threadIdx.x & 63, st = threadIdx.x >> 6;
    Strip4Node g[4];
    // rows pre-scaled by 1 / (md_scale * m_ii): z = bs - ym / md_scale - sum ms * y_nb  (two registers per node
    // fewer than keeping m_ii and its reciprocal: no scratch spills at 128 VGPRs)
    const double inv_scale = 1.0 / md_scale;
    double mv[4][W - 1], bv[4], ym[4], yo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        g[r] = strip4_node(N, r, H, strip4_patch(remap));
        bv[r] = 0.0; ym[r] = 0.0; yo[r] = 0.0;
#pragma unroll
        for (int s = 0; s < W - 1; ++s) mv[r][s] = 0.0;
        if (g[r].inside) {
            const double rmd = 1.0 / (md_scale * M[g[r].i]);
#pragma unroll
            for (int s = 1; s < W; ++s) mv[r][s - 1] = M[(int64_t)s * n + g[r].i] * rmd;
            bv[r] = b_[voff + g[r].i] * rmd;
            if (ymid_) ym[r] = ymid_[voff + g[r].i];
            if (yold_) yo[r] = yold_[voff + g[r].i];
        }
    }
    for (int k = 0; k < K; ++k) {
        const int par = k & 1;
        bot[par][st][lx] = ym[0];
        top[par][st][lx] = ym[3];
        __syncthreads();
        const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;
        const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;
        STRIP4_NEIGHBOURS(ym, above, below);
        const double wk = omd ? omd[k] : om.w[k];
        double yn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            yn[r] = ym[r];
            double z = fma(-inv_scale, ym[r], bv[r]);
#pragma unroll
            for (int s = 0; s < W - 1; ++s) z = fma(-mv[r][s], STRIP4_NB(ym, above, below, r, s), z);
            yn[r] = wk * (z + ym[r] - yo[r]) + yo[r];     // no validity guard (see k_strip4_jacobi)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { yo[r] = ym[r]; ym[r] = yn[r]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (g[r].owned) {
            omid_[voff + g[r].i] = ym[r];
            if (oold_) oold_[voff + g[r].i] = yo[r];
        }
}

// Chebyshev iterations on the mesh's own consistent mass matrix without reading it: on the right-diagonal P1 mesh
// m_ii = ntri |K| / 6 and m_ij = cnt_ij |K| / 12 (ntri triangles around the node, cnt_ij in {0,1,2} triangles on the
// edge), so the row scaled by 1 / (md_scale m_ii) is cnt_ij / (2 md_scale ntri) -- small-integer ratios decided by
// which of the four cells around the node exist.  HBM traffic per node and launch drops from 9 doubles (7 matrix
// entries, rhs, iterate) to 2-3, and the interior update is one multiply of the neighbour sum.
template <int RING>    // 1: the boundary-ring launch next to k_strip4_cheb_mass_int (its own instantiation: 48 VGPRs of weights)
__global__ void __launch_bounds__(STRIP_T)
k_strip4_cheb_mass(int n, int N, double h, const double* __restrict__ b_, const double* __restrict__ ymid_,
                   const double* __restrict__ yold_, double* __restrict__ omid_, double* __restrict__ oold_, int K,
                   CheOmegas om, double md_scale, int H, int remap, int ring) {
    __shared__ double top[2][16][64], bot[2][16][64];
    // ring > 0: only the patches outside the interior block [1, ring]^2 (k_strip4_cheb_mass_int's) are launched, as a
    // 1-D grid: the row below the block, the rows above it, then the columns left and right of it
    unsigned pxy = strip4_patch(remap);
    if (RING) {
        const int t = (N + (T4_L - 2 * H) - 1) / (T4_L - 2 * H), side = t - ring;   // patches per row, per row beside the block
        int id = blockIdx.x, bx, by;
        if (id < t) { bx = id; by = 0; }
        else if (id < t * side) { id -= t; by = ring + 1 + id / t; bx = id % t; }
        else { id -= t * side; by = 1 + id / side; const int c = id % side; bx = c == 0 ? 0 : ring + c; }
        pxy = (unsigned)bx | ((unsigned)by << 16);
    }
    const int64_t voff = (int64_t)blockIdx.z * n;
    const int lx = threadIdx.x & 63, st = threadIdx.x >> 6;
    const int nc = N - 1;
    const double inv_scale = 1.0 / md_scale;
    Strip4Node g[4];
    double bv[4], cw[4], ym[4], yo[4];
    int pc[4];            // six 2-bit edge counts, slots E, NE, N, W, SW, S
    // (interior nodes -- six triangles -- share their two quotients: two f64 divisions per thread instead of eight)
    const double cw6 = 1.0 / (2.0 * md_scale * 6), bs6 = 12.0 / (md_scale * 6 * h * h);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        g[r] = strip4_node(N, r, H, pxy);
        bv[r] = 0.0; cw[r] = 0.0; ym[r] = 0.0; yo[r] = 0.0; pc[r] = 0;
        if (g[r].inside) {
            const int gx = (int)(pxy & 0xffffu) * (T4_L - 2 * H) - H + lx, gy = (int)(pxy >> 16) * (T4_L - 2 * H) - H + 4 * st + r;
            const int c00 = (gx < nc && gy < nc), c10 = (gx > 0 && gy < nc), c01 = (gx < nc && gy > 0), c11 = (gx > 0 && gy > 0);
            const int ntri = 2 * c00 + c10 + c01 + 2 * c11;
            pc[r] = (c00 + c01) | ((2 * c00) << 2) | ((c00 + c10) << 4) | ((c10 + c11) << 6) | ((2 * c11) << 8) | ((c11 + c01) << 10);
            const double bi = b_[voff + g[r].i];
            if (ntri == 6) { cw[r] = cw6; bv[r] = bi * bs6; }
            else { cw[r] = 1.0 / (2.0 * md_scale * ntri); bv[r] = bi * (12.0 / (md_scale * ntri * h * h)); }
            // interior rows weight every neighbour by two: (2 sum) cw == sum (2 cw) to the bit (scaling by two is exact),
            // so the factor moves out of the sweeps
            if (pc[r] == 0xAAA) cw[r] = 2.0 * cw[r];
            if (ymid_) ym[r] = ymid_[voff + g[r].i];
            if (yold_) yo[r] = yold_[voff + g[r].i];
        }
    }
    // Two copies of the sweep loop, chosen per WAVE: ~95 % of the waves hold interior nodes only and run a loop with no
    // stencil-shape test at all; the others keep the general weights -- decoded from pc inside the loop (an empty asm
    // hides pc from loop-invariant code motion: hoisted, the 24 weights cost 48 VGPRs in BOTH loops).
    const bool all_interior = __all(pc[0] == 0xAAA && pc[1] == 0xAAA && pc[2] == 0xAAA && pc[3] == 0xAAA);
#define CHEB_MASS_SWEEPS(INTERIOR)                                                                              \
    for (int k = 0; k < K; ++k) {                                                                               \
        const int par = k & 1;                                                                                  \
        bot[par][st][lx] = ym[0];                                                                               \
        top[par][st][lx] = ym[3];                                                                               \
        __syncthreads();                                                                                        \
        const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;                                            \
        const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;                                             \
        STRIP4_NEIGHBOURS(ym, above, below);                                                                    \
        const double wk = om.w[k];                                                                              \
        double yn[4];                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                         \
            double sum;                                                                                         \
            int pcr = pc[r];                                                                                    \
            if (!(INTERIOR)) asm volatile("" : "+v"(pcr));                                                      \
            if ((INTERIOR) || pcr == 0xAAA) {  /* interior node: all six edges carry two triangles (2 in cw) */ \
                sum = ((STRIP4_NB(ym, above, below, r, 0) + STRIP4_NB(ym, above, below, r, 1)) +                \
                       (STRIP4_NB(ym, above, below, r, 2) + STRIP4_NB(ym, above, below, r, 3))) +               \
                      (STRIP4_NB(ym, above, below, r, 4) + STRIP4_NB(ym, above, below, r, 5));                  \
            } else {                                                                                            \
                sum = 0.0;                                                                                      \
                _Pragma("unroll") for (int s = 0; s < 6; ++s)                                                   \
                    sum = fma((double)((pcr >> (2 * s)) & 3), STRIP4_NB(ym, above, below, r, s), sum);          \
            }                                                                                                   \
            const double z = fma(-cw[r], sum, fma(-inv_scale, ym[r], bv[r]));                                   \
            yn[r] = wk * (z + ym[r] - yo[r]) + yo[r];                                                           \
        }                                                                                                       \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) { yo[r] = ym[r]; ym[r] = yn[r]; }                         \
    }
    if (RING) {
        // boundary-ring launch: one workgroup per CU anyway and most patches hold some boundary nodes -- every wave runs
        // the general stencil, with the 24 edge counts decoded ONCE (48 VGPRs this launch can afford): six FMAs per row
        // and sweep instead of six decode-and-FMA groups.  (double)cnt is the same value either way: same bits.
        double wq[4][6];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int s = 0; s < 6; ++s) wq[r][s] = (double)((pc[r] >> (2 * s)) & 3);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (pc[r] == 0xAAA) {          // interior rows carry the factor two in cw (see above): unit weights
#pragma unroll
                for (int s = 0; s < 6; ++s) wq[r][s] = 1.0;
            }
        for (int k = 0; k < K; ++k) {
            const int par = k & 1;
            bot[par][st][lx] = ym[0];
            top[par][st][lx] = ym[3];
            __syncthreads();
            const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;
            const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;
            STRIP4_NEIGHBOURS(ym, above, below);
            const double wk = om.w[k];
            double yn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double sum;
                if (pc[r] == 0xAAA) {
                    sum = ((STRIP4_NB(ym, above, below, r, 0) + STRIP4_NB(ym, above, below, r, 1)) +
                           (STRIP4_NB(ym, above, below, r, 2) + STRIP4_NB(ym, above, below, r, 3))) +
                          (STRIP4_NB(ym, above, below, r, 4) + STRIP4_NB(ym, above, below, r, 5));
                } else {
                    sum = 0.0;
#pragma unroll
                    for (int s = 0; s < 6; ++s) sum = fma(wq[r][s], STRIP4_NB(ym, above, below, r, s), sum);
                }
                const double z = fma(-cw[r], sum, fma(-inv_scale, ym[r], bv[r]));
                yn[r] = wk * (z + ym[r] - yo[r]) + yo[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { yo[r] = ym[r]; ym[r] = yn[r]; }
        }
    } else if (all_interior) { CHEB_MASS_SWEEPS(true) } else { CHEB_MASS_SWEEPS(false) }
#undef CHEB_MASS_SWEEPS
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (g[r].owned) {
            omid_[voff + g[r].i] = ym[r];
            if (oold_) oold_[voff + g[r].i] = yo[r];
        }
}

// The same iterations for patches that lie entirely in the mesh interior (every node has six triangles): no stencil
// shapes, no per-row weights, no validity flags -- 60-odd VGPRs, so TWO 1024-thread workgroups fit a CU and each one's
// barrier and LDS round trips are covered by the other's arithmetic (tools/sweep_probe.hip: 0.58 instead of 1.0 us per
// patch and sweep).  One workgroup per patch, blockIdx + (px0, py0); the patches that touch the mesh boundary (the outer
// ring of the patch grid) are left to k_strip4_cheb_mass with skip_interior = 1.  Same expressions as the interior branch
// there: same bits.
__global__ void __launch_bounds__(STRIP_T, 8)
k_strip4_cheb_mass_int(int n, int N, double h, const double* __restrict__ b_, const double* __restrict__ ymid_,
                       const double* __restrict__ yold_, double* __restrict__ omid_, double* __restrict__ oold_, int K,
                       CheOmegas om, double md_scale, int H, int px0, int py0) {
    __shared__ double top[2][16][64], bot[2][16][64];
    const int lx = threadIdx.x & 63, st = threadIdx.x >> 6;
    const int T = T4_L - 2 * H;
    const double inv_scale = 1.0 / md_scale;
    // (interior rows weight every neighbour by two: the factor sits in cw, as in k_strip4_cheb_mass)
    const double cw = 2.0 * (1.0 / (2.0 * md_scale * 6)), bs6 = 12.0 / (md_scale * 6 * h * h);
    const int gx = ((int)blockIdx.x + px0) * T - H + lx, gy0 = ((int)blockIdx.y + py0) * T - H + 4 * st;
    const int64_t i0 = (int64_t)blockIdx.z * n + (int64_t)gy0 * N + gx;          // node of row 0; rows are N apart
    double bv[4], ym[4], yo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bv[r] = b_[i0 + (int64_t)r * N] * bs6;
        ym[r] = ymid_ ? ymid_[i0 + (int64_t)r * N] : 0.0;
        yo[r] = yold_ ? yold_[i0 + (int64_t)r * N] : 0.0;
    }
    for (int k = 0; k < K; ++k) {
        const int par = k & 1;
        bot[par][st][lx] = ym[0];
        top[par][st][lx] = ym[3];
        __syncthreads();
        const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;
        const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;
        STRIP4_NEIGHBOURS(ym, above, below);
        const double wk = om.w[k];
        double yn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double sum = ((STRIP4_NB(ym, above, below, r, 0) + STRIP4_NB(ym, above, below, r, 1)) +
                                (STRIP4_NB(ym, above, below, r, 2) + STRIP4_NB(ym, above, below, r, 3))) +
                               (STRIP4_NB(ym, above, below, r, 4) + STRIP4_NB(ym, above, below, r, 5));
            const double z = fma(-cw, sum, fma(-inv_scale, ym[r], bv[r]));
            yn[r] = wk * (z + ym[r] - yo[r]) + yo[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { yo[r] = ym[r]; ym[r] = yn[r]; }
    }
    if (lx >= H && lx < H + T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ly = 4 * st + r;
            if (ly >= H && ly < H + T) {
                omid_[i0 + (int64_t)r * N] = ym[r];
                if (oold_) oold_[i0 + (int64_t)r * N] = yo[r];
            }
        }
    }
}

// k_strip4_cheb_mass with persistent workgroups (see k_strip4_jacobi_walk), each taking every gridDim.x-th patch.
// No rows are carried (the inputs are three vectors, 24 bytes per node); what the walk buys here is that the next
// patch's inputs are requested BEFORE the sweeps of the current one, so the ~10 sweeps cover their latency, and that no
// workgroup start-up sits between two patches on a CU.
__global__ void __launch_bounds__(STRIP_T)
k_strip4_cheb_mass_walk(int n, int N, double h, const double* __restrict__ b_, const double* __restrict__ ymid_,
                        const double* __restrict__ yold_, double* __restrict__ omid_, double* __restrict__ oold_, int K,
                        CheOmegas om, double md_scale, int H, int npy, int npatch) {
    __shared__ double top[2][16][64], bot[2][16][64];
    const int64_t voff = (int64_t)blockIdx.z * n;
    const int st = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nc = N - 1;
    const double inv_scale = 1.0 / md_scale;
    const int nwg = gridDim.x, wg = blockIdx.x;
    // patches wg, wg + nwg, ...: patches on the mesh boundary (general stencil weights: slower sweeps) spread evenly
    const int q0 = wg, q1 = npatch;
    const int lx0 = threadIdx.x & 63;
    // The two quotients of an interior node (six triangles: all but the mesh-boundary rows), once per kernel: eight f64
    // divisions per thread and patch are ~2 us of a 13 us patch.  Same expressions, same operands: same bits.
    const double cw6 = 1.0 / (2.0 * md_scale * 6), bs6 = 12.0 / (md_scale * 6 * h * h);
    double pb[4], pym[4], pyo[4];          // raw inputs of the patch about to be processed
#pragma unroll
    for (int r = 0; r < 4; ++r) { pb[r] = 0.0; pym[r] = 0.0; pyo[r] = 0.0; }
    if (q0 < q1) {
        const int px = q0 / npy, py = q0 - px * npy;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = strip4_node(N, r, H, (unsigned)px | ((unsigned)py << 16), lx0, st).i;   // (outside lanes: node 0, discarded)
            pb[r] = b_[voff + i];
            if (ymid_) pym[r] = ymid_[voff + i];
            if (yold_) pyo[r] = yold_[voff + i];
        }
    }
    for (int q = q0; q < q1; q += nwg) {
        int lx = lx0;
        asm volatile("" : "+v"(lx));       // (keeps the addresses derived from the lane index out of the walk's invariants)
        const int px = q / npy, py = q - px * npy;
        Strip4Node g[4];
        double bv[4], cw[4], ym[4], yo[4];
        int pc[4];            // six 2-bit edge counts, slots E, NE, N, W, SW, S
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            g[r] = strip4_node(N, r, H, (unsigned)px | ((unsigned)py << 16), lx, st);
            bv[r] = 0.0; cw[r] = 0.0; ym[r] = 0.0; yo[r] = 0.0; pc[r] = 0;
            if (g[r].inside) {
                const int gx = px * (T4_L - 2 * H) - H + lx, gy = py * (T4_L - 2 * H) - H + 4 * st + r;   // (no i / N)
                const int c00 = (gx < nc && gy < nc), c10 = (gx > 0 && gy < nc), c01 = (gx < nc && gy > 0), c11 = (gx > 0 && gy > 0);
                const int ntri = 2 * c00 + c10 + c01 + 2 * c11;
                pc[r] = (c00 + c01) | ((2 * c00) << 2) | ((c00 + c10) << 4) | ((c10 + c11) << 6) | ((2 * c11) << 8) | ((c11 + c01) << 10);
                if (ntri == 6) {
                    cw[r] = cw6;
                    bv[r] = pb[r] * bs6;
                } else {
                    cw[r] = 1.0 / (2.0 * md_scale * ntri);
                    bv[r] = pb[r] * (12.0 / (md_scale * ntri * h * h));
                }
                if (pc[r] == 0xAAA) cw[r] = 2.0 * cw[r];
                ym[r] = pym[r];
                yo[r] = pyo[r];
            }
        }
        if (q + nwg < q1) {
            const int qn = q + nwg, pxn = qn / npy, pyn = qn - pxn * npy;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = strip4_node(N, r, H, (unsigned)pxn | ((unsigned)pyn << 16), lx, st).i;
                pb[r] = b_[voff + i];
                if (ymid_) pym[r] = ymid_[voff + i];
                if (yold_) pyo[r] = yold_[voff + i];
            }
        }
        const bool all_interior = __all(pc[0] == 0xAAA && pc[1] == 0xAAA && pc[2] == 0xAAA && pc[3] == 0xAAA);
        __syncthreads();                   // the previous patch's last sweep has left top / bot
#define CHEB_MASS_SWEEPS(INTERIOR)                                                                              \
        for (int k = 0; k < K; ++k) {                                                                           \
            const int par = k & 1;                                                                              \
            bot[par][st][lx] = ym[0];                                                                           \
            top[par][st][lx] = ym[3];                                                                           \
            __syncthreads();                                                                                    \
            const double above = (st < 15) ? bot[par][st + 1][lx] : 0.0;                                        \
            const double below = (st > 0) ? top[par][st - 1][lx] : 0.0;                                         \
            STRIP4_NEIGHBOURS(ym, above, below);                                                                \
            const double wk = om.w[k];                                                                          \
            double yn[4];                                                                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                     \
                double sum;                                                                                     \
                int pcr = pc[r];                                                                                \
                if (!(INTERIOR)) asm volatile("" : "+v"(pcr));                                                  \
                if ((INTERIOR) || pcr == 0xAAA) {                                                               \
                    sum = ((STRIP4_NB(ym, above, below, r, 0) + STRIP4_NB(ym, above, below, r, 1)) +            \
                           (STRIP4_NB(ym, above, below, r, 2) + STRIP4_NB(ym, above, below, r, 3))) +           \
                          (STRIP4_NB(ym, above, below, r, 4) + STRIP4_NB(ym, above, below, r, 5));              \
                } else {                                                                                        \
                    sum = 0.0;                                                                                  \
                    _Pragma("unroll") for (int s = 0; s < 6; ++s)                                               \
                        sum = fma((double)((pcr >> (2 * s)) & 3), STRIP4_NB(ym, above, below, r, s), sum);      \
                }                                                                                               \
                const double z = fma(-cw[r], sum, fma(-inv_scale, ym[r], bv[r]));                               \
                yn[r] = wk * (z + ym[r] - yo[r]) + yo[r];                                                       \
            }                                                                                                   \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) { yo[r] = ym[r]; ym[r] = yn[r]; }                     \
        }
        if (all_interior) { CHEB_MASS_SWEEPS(true) } else { CHEB_MASS_SWEEPS(false) }
#undef CHEB_MASS_SWEEPS
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (g[r].owned) {
                omid_[voff + g[r].i] = ym[r];
                if (oold_) oold_[voff + g[r].i] = yo[r];
            }
    }
}

}  // namespace

int femfct_tile4_init(femfct_ctx* ctx) {
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_tile4_jacobi<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T4_BUF * 8));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_tile4_jacobi<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T4_BUF * 8));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_tile4_cheb, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * T4_BUF * 8));
    return FEMFCT_OK;
}

// bandwidth regime: the mesh (times the batch) is large enough that traffic, not launch latency, rules
// One 64 x 64 patch covers the whole mesh (N <= 48 with halo 8): no halo limits the iterations per launch and one
// workgroup advances one system without any grid-wide dependency.  Used for the long species solves of batched
// small trajectories (Armijo trials / beta sweeps of the 41 x 41 configs: 221 Chebyshev iterations in one launch,
// B workgroups in the time of one; measured at n = 1681: slower than 17 tile launches for B = 4, 7 % faster for
// B = 10, 30 % for B = 20).  The FCT step itself stays on the fused 32-patch kernels unless FEMFCT_TILE4=2.
bool femfct_single_patch(const femfct_ctx* ctx, int32_t batch) {
    if (!ctx->use_strips || !ctx->use_tiles || !ctx->implicit_cols || ctx->W != 7) return false;
    if (!ctx->t4_dpp || ctx->t4_k != 8 || ctx->tile4_mode == 0) return false;
    return ctx->N <= T4_L - 2 * T4_H && batch >= ctx->single_patch_min_batch;
}

bool femfct_tile4_wanted(const femfct_ctx* ctx, int32_t batch) {
    if (!ctx->use_strips || !ctx->use_tiles || !ctx->implicit_cols || ctx->W != 7) return false;
    if (ctx->tile4_mode == 0) return false;
    if (ctx->tile4_mode == 2) return true;
    // measured crossover (register-resident strip kernels): n = 6561 between batch 12 and 16, single meshes
    // between 257^2 and 321^2 nodes
    return (int64_t)ctx->n * batch >= 90000;
}

int femfct_tile4_tiles(const femfct_ctx* ctx, int H) {
    const int T = T4_L - 2 * H;
    return (ctx->N + T - 1) / T;
}

// Halo depth (= sweeps per launch) of the 64-patch strip kernels for `sweeps` sweeps in a row.  Measured launch
// model at n = 2049^2: ~106 us of matrix load + ~5.6 us per sweep with halo 8; both scale with the patch
// redundancy (48 / (64 - 2H))^2.  E.g. 36 sweeps = 4 x 9, 19 Chebyshev iterations = 10 + 9.
int femfct_tile4_halo(const femfct_ctx* ctx, int sweeps) {
    if (!ctx->t4_dpp || ctx->t4_k != 8) return T4_H;   // LDS-image variant / measurement knob: fixed geometry
    if (ctx->N <= T4_L - 2 * T4_H) return T4_H;        // one patch covers the mesh: nothing to trade
    int best_h = T4_H;
    double best = 1e300;
    for (int h = T4_H; h <= 10; ++h) {
        const double af = (48.0 / (T4_L - 2 * h)) * (48.0 / (T4_L - 2 * h));
        const int launches = (sweeps + h - 1) / h;
        const double cost = af * (106.0 * launches + 5.6 * sweeps);
        if (cost < best - 1e-9) { best = cost; best_h = h; }
    }
    return best_h;
}

// Walkers per batch member of the walking Jacobi launch (0: one workgroup per patch).  One 1024-thread workgroup fits
// a CU, so a launch keeps num_cus of them busy: with at least two patches per walker the carried rows pay, below that
// hardware dispatch of one workgroup per patch fills the chip better.
// pair: the launch is k_strip_jacobi_pair_walk<PAIR_R, PAIR_NST> -- two such workgroups fit a CU, so twice the walkers
// (but still at least two patches per walker, and the same regime boundary as the 1024-thread walk).
// shape = rows per thread x waves per workgroup: 6 x 8.  (-DFEMFCT_TUNING builds: FEMFCT_PAIR_SHAPE 3 = 8 x 8, 4 = 7 x 8,
// 6 = 12 x 4, the shapes DESIGN.md section 8 reports on)
static void pair_shape(const femfct_ctx* ctx, int* R, int* NST) {
    *R = 6; *NST = 8;
#ifdef FEMFCT_TUNING
    switch (ctx->pair_shape) {
        case 6: *R = 12; *NST = 4; break;
        case 3: *R = 8; *NST = 8; break;
        case 4: *R = 7; *NST = 8; break;
        default: break;
    }
#endif
}
static int pair_tiles_y(const femfct_ctx* ctx, int H) {
    int R, NST;
    pair_shape(ctx, &R, &NST);
    const int TY = R * NST - 2 * H;
    return (ctx->N + TY - 1) / TY;
}
int femfct_tile4_walkers(const femfct_ctx* ctx, int H, int32_t batch, bool pair) {
    if (!ctx->t4_walk || !ctx->t4_dpp || 2 * H > WALK_CARRY_ROWS) return 0;
    const int t = femfct_tile4_tiles(ctx, H);
    const int w = std::min(ctx->num_cus / std::max(1, (int)batch), FEMFCT_MAX_PARTIALS);
    if (w < 1 || (int64_t)t * t < 2 * (int64_t)w) return 0;
    if (pair) return (int)std::min<int64_t>(std::min(2 * w, FEMFCT_MAX_PARTIALS), (int64_t)t * pair_tiles_y(ctx, H) / 2);
    return w;
}

// the pair-compact walking launch applies: asked for by the sweep driver (ctx->pair_rows: a kind of sweep whose rows
// have so far all been upwind rows), zero masks in use, walking regime
bool femfct_jacobi_pair_wanted(const femfct_ctx* ctx, int H, int32_t batch, bool have_lmask, bool assume_upwind_rows) {
    if (!(ctx->t4_pair && (ctx->pair_rows || assume_upwind_rows) && have_lmask && ctx->t4_walk == 1 && 2 * H <= PAIR_CARRY_ROWS && ctx->t4_k == 8)) return false;
    if ((uint64_t)ctx->ws_batch * ctx->W * (uint64_t)ctx->n + 1 >= (1ull << 32)) return false;   // 32-bit element offsets into L
    return femfct_tile4_walkers(ctx, H, batch, false) > 0;
}

int femfct_enqueue_tile4_jacobi(femfct_ctx* ctx, const double* L, const double* b, double* xa, double* xb, int launch,
                                int g_build, int32_t batch, int H, int K, int check_every, const uint8_t* lmask) {
    const int t = femfct_tile4_tiles(ctx, H);
    const bool big = (int64_t)t * t > FEMFCT_MAX_PARTIALS;
    dim3 grid(t, t, batch);
    const size_t lds = (size_t)2 * T4_BUF * 8;
    femfct_prof_begin(ctx, KC_JACOBI);
    const bool pair = check_every <= 0 && femfct_jacobi_pair_wanted(ctx, H, batch, lmask != nullptr);
    const int walkers = check_every > 0 ? 0 : femfct_tile4_walkers(ctx, H, batch, pair);
    ctx->last_launch[0] = pair ? 3 : walkers > 0 ? 2 : ctx->t4_dpp ? 1 : 0;
    ctx->last_launch[1] = walkers;
    ctx->last_launch[3] = H;
    if (pair) {
        const int npy = pair_tiles_y(ctx, H);
#define PAIR_LAUNCH(RR, NN)                                                                                              \
        hipLaunchKernelGGL((k_strip_jacobi_pair_walk<RR, NN>), dim3(walkers, 1, batch), dim3(64 * NN), 0, ctx->stream,    \
                           ctx->n, ctx->N, L, b, xa, xb, ctx->d_part, ctx->d_ctl, launch, K, g_build, ctx->rel_tol, H, lmask, npy, \
                           t * npy, ctx->t4_snake ? (launch & 1) : 0, (int64_t)ctx->ws_batch * ctx->W * ctx->n PAIR_TUNING_ARGS)
#ifdef FEMFCT_TUNING
        switch (ctx->pair_shape) {
            case 6: PAIR_LAUNCH(12, 4); break;
            case 3: PAIR_LAUNCH(8, 8); break;
            case 4: PAIR_LAUNCH(7, 8); break;
            default: PAIR_LAUNCH(6, 8); break;
        }
#else
        PAIR_LAUNCH(6, 8);
#endif
#undef PAIR_LAUNCH
    } else if (walkers > 0) {
        hipLaunchKernelGGL(k_strip4_jacobi_walk, dim3(walkers, 1, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, L, b, xa,
                           xb, ctx->d_part, ctx->d_ctl, launch, K, g_build, ctx->rel_tol, H, ctx->t4_stagger, lmask, t, t * t, ctx->t4_walk == 1 ? 1 : 0, ctx->t4_snake ? (launch & 1) : 0);
    } else if (ctx->t4_dpp) {
        if (big) {
            hipLaunchKernelGGL(k_strip4_jacobi<1>, grid, dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, L, b, xa, xb, ctx->d_part,
                               ctx->d_ctl, launch, K, g_build, ctx->rel_tol, ctx->d_bigpart, H, check_every, (int64_t)t * t * batch >= 1024 ? ctx->t4_stagger : 0, lmask, ctx->t4_xcd);
            hipLaunchKernelGGL(k_reduce_resid, dim3(batch), dim3(STRIP_T), 0, ctx->stream, ctx->d_bigpart, (int64_t)t * t,
                               ctx->d_ctl, launch);
        } else if (check_every > 0) {
            hipLaunchKernelGGL(k_strip4_jacobi<2>, grid, dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, L, b, xa, xb, ctx->d_part,
                               ctx->d_ctl, launch, K, g_build, ctx->rel_tol, (double*)nullptr, H, check_every, 0, lmask, 0);
        } else {
            hipLaunchKernelGGL(k_strip4_jacobi<0>, grid, dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, L, b, xa, xb, ctx->d_part,
                               ctx->d_ctl, launch, K, g_build, ctx->rel_tol, (double*)nullptr, H, 0,
                               (int64_t)t * t * batch >= 1024 ? ctx->t4_stagger : 0, lmask, ctx->t4_xcd);
        }
    } else if (big) {
        hipLaunchKernelGGL(k_tile4_jacobi<1>, grid, dim3(STRIP_T), lds, ctx->stream, ctx->n, ctx->N, L, b, xa, xb, ctx->d_part,
                           ctx->d_ctl, launch, K, g_build, ctx->rel_tol, ctx->d_bigpart);
        hipLaunchKernelGGL(k_reduce_resid, dim3(batch), dim3(STRIP_T), 0, ctx->stream, ctx->d_bigpart, (int64_t)t * t,
                           ctx->d_ctl, launch);
    } else {
        hipLaunchKernelGGL(k_tile4_jacobi<0>, grid, dim3(STRIP_T), lds, ctx->stream, ctx->n, ctx->N, L, b, xa, xb, ctx->d_part,
                           ctx->d_ctl, launch, K, g_build, ctx->rel_tol, (double*)nullptr);
    }
    femfct_prof_end(ctx);
    return FEMFCT_OK;
}

int femfct_enqueue_tile4_cheb(femfct_ctx* ctx, const double* b, const double* in_mid, const double* in_old, double* y_out,
                              int k_first, int k_last, const double* omegas, double md_scale, double* bufA0, double* bufA1,
                              double* bufB0, double* bufB1, int32_t batch, const ChebIO* io_in) {
    ChebIO io0{};
    if (io_in) io0 = *io_in;
    io0.mid_ref = make_ref(nullptr); io0.mid_bs = 0; io0.out_ref = make_ref(nullptr); io0.out_bs = 0;
    const bool single = femfct_single_patch(ctx, batch);
    const int H = single ? T4_H : femfct_tile4_halo(ctx, k_last - k_first + 1);
    const int per_launch = single ? ((io_in && io_in->om_dev) ? k_last - k_first + 1 : 24) : H;   // by-value omega table: 24
    const int t = femfct_tile4_tiles(ctx, H);
    const int walkers = single ? 0 : femfct_tile4_walkers(ctx, H, batch, false);
    // patches 1 .. n_int (per direction) lie wholly in the mesh interior: (p + 1) T + H <= N - 1
    const int n_int = (single || !ctx->t4_int) ? 0 : std::max(0, (ctx->N - 1 - H) / (T4_L - 2 * H) - 1);
    const size_t lds = (size_t)3 * T4_BUF * 8;
    const double* mid = in_mid;
    const double* old = in_old;
    int which = 0;
    for (int k0 = k_first; k0 <= k_last; k0 += per_launch) {
        int k1 = std::min(k_last + 1, k0 + per_launch);
        CheOmegas om;
        for (int k = k0; k < k1 && k - k0 < 24; ++k) om.w[k - k0] = omegas ? omegas[k - 1] : 0.0;
        const bool last = (k1 == k_last + 1);
        double* omid = last ? y_out : (which ? bufB0 : bufA0);
        double* oold = last ? nullptr : (which ? bufB1 : bufA1);
        ChebIO io = io0;
        io.k0 = k0 - 1;
        if (io_in && k0 == k_first) { io.mid_ref = io_in->mid_ref; io.mid_bs = io_in->mid_bs; }
        if (io_in && last) { io.out_ref = io_in->out_ref; io.out_bs = io_in->out_bs; }
        femfct_prof_begin(ctx, KC_CHEB);
        ctx->last_launch[2] = (ctx->t4_dpp && femfct_geom_mass(ctx) && !io_in && n_int > 0 && (int64_t)t * t * batch > ctx->num_cus) ? n_int : 0;
        if (ctx->t4_dpp && femfct_geom_mass(ctx) && !io_in && n_int > 0 && (int64_t)t * t * batch > ctx->num_cus) {
            // more patches than compute units: the interior ones two workgroups to a CU, the boundary ring by the general kernel
            hipLaunchKernelGGL(k_strip4_cheb_mass_int, dim3(n_int, n_int, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N,
                               ctx->h, b, mid, old, omid, oold, k1 - k0, om, md_scale, H, 1, 1);
            hipLaunchKernelGGL(k_strip4_cheb_mass<1>, dim3(t * t - n_int * n_int, 1, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n,
                               ctx->N, ctx->h, b, mid, old, omid, oold, k1 - k0, om, md_scale, H, 0, n_int);
        } else if (ctx->t4_dpp && femfct_geom_mass(ctx) && !io_in && walkers > 0)
            hipLaunchKernelGGL(k_strip4_cheb_mass_walk, dim3(walkers, 1, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N,
                               ctx->h, b, mid, old, omid, oold, k1 - k0, om, md_scale, H, t, t * t);
        else if (ctx->t4_dpp && femfct_geom_mass(ctx) && !io_in)
            hipLaunchKernelGGL(k_strip4_cheb_mass<0>, dim3(t, t, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, ctx->h, b,
                               mid, old, omid, oold, k1 - k0, om, md_scale, H, ctx->t4_xcd, 0);
        else if (ctx->t4_dpp)
            hipLaunchKernelGGL(k_strip4_cheb, dim3(t, t, batch), dim3(STRIP_T), 0, ctx->stream, ctx->n, ctx->N, ctx->d_M, b, mid,
                               old, omid, oold, k1 - k0, om, md_scale, io, H, ctx->t4_xcd);
        else
            hipLaunchKernelGGL(k_tile4_cheb, dim3(t, t, batch), dim3(STRIP_T), lds, ctx->stream, ctx->n, ctx->N, ctx->d_M, b, mid,
                               old, omid, oold, k1 - k0, om, md_scale, io);
        femfct_prof_end(ctx);
        mid = omid;
        old = oold;
        which ^= 1;
    }
    return FEMFCT_OK;
}
