// Device-side bookkeeping of the low-order solve shared by the step kernels.
#pragma once

#include "femfct_internal.h"
#include "device_utils.h"

// Deferred test of a solve of U <= 4 launches (launch 0 = k_tile_build_jacobi, launches 1.. = k_tile_jacobi with
// defer = 1): no Jacobi launch looked at a partial, so ||b||, the minimal row sum and the U residual maxima are all
// reduced by the kernel behind them, in one pass, and the verdict the in-launch tests would have reached is
// reconstructed (the first launch whose residual passed gives the sweep count the host budgets from).  Split in two so
// that the caller (wave 0 of workgroup 0 of k_tile_dudt_cheb) can request the partials before its tile work and
// publish after it: no LDS, no barrier -- the publish is a handful of lane shifts at the kernel's end.  What differs
// from the in-launch test: had an earlier launch already converged, the later ones ran nevertheless (their sweeps only
// lower the residual further; the iterate returned is the budget-parity one).
// Residual maxima: launch 0 at p[0 .. G), launch s >= 1 at partk[s * FEMFCT_MAX_PARTIALS ..).
constexpr int FEMFCT_DEFER_MAX_UNITS = 4;
struct DeferredPartials { double r[FEMFCT_DEFER_MAX_UNITS], bn, rs; };

__device__ __forceinline__ DeferredPartials deferred_test_load(const double* p, const double* partk, int G, int units) {
    DeferredPartials d;
#pragma unroll
    for (int s = 0; s < FEMFCT_DEFER_MAX_UNITS; ++s) d.r[s] = 0.0;
    d.bn = 0.0; d.rs = INFINITY;
    if (threadIdx.x < WAVE) {
#pragma unroll 4                           // (up to 256 workgroups: all loads of the wave in flight together)
        for (int k = threadIdx.x; k < G; k += WAVE) {
            d.r[0] = fmax(d.r[0], p[k]);
#pragma unroll
            for (int s = 1; s < FEMFCT_DEFER_MAX_UNITS; ++s)
                if (s < units) d.r[s] = fmax(d.r[s], partk[(int64_t)s * FEMFCT_MAX_PARTIALS + k]);
            d.bn = fmax(d.bn, p[2 * FEMFCT_MAX_PARTIALS + k]);
            d.rs = fmin(d.rs, p[3 * FEMFCT_MAX_PARTIALS + k]);
        }
    }
    return d;
}

__device__ __forceinline__ void deferred_test_publish(StepCtl* ctl, DeferredPartials d, int units, int iters_per_unit,
                                                      double rel_tol) {
    if (threadIdx.x >= WAVE) return;
    double r[FEMFCT_DEFER_MAX_UNITS];
#pragma unroll
    for (int s = 0; s < FEMFCT_DEFER_MAX_UNITS; ++s) r[s] = wave_reduce(d.r[s], OpMax());
    const double bn = wave_reduce(d.bn, OpMax()), rs = wave_reduce(d.rs, OpMin());
    if (threadIdx.x == 0) {
        ctl->bnorm = bn;
        ctl->min_rowsum = rs;
        if (!(rs > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        const double tolb = rel_tol * bn;
        int first = -1;
        double rlast = 0.0;
#pragma unroll
        for (int s = 0; s < FEMFCT_DEFER_MAX_UNITS; ++s)
            if (s < units) {
                if (first < 0 && r[s] <= tolb) first = s;
                rlast = r[s];
            }
        ctl->iters = (first >= 0 ? first + 1 : units) * iters_per_unit;
        if (first >= 0 && first < units - 1) ctl->flags |= FEMFCT_FLAG_COARSE_ITERS;
        if (first < 0) ctl->flags |= FEMFCT_FLAG_SOLVER_BUDGET;
        ctl->resid = bn > 0.0 ? rlast / bn : 0.0;
    }
}

// Finalise the solve bookkeeping when the sweep budget ran out before `done`.
__device__ __forceinline__ void finalize_solve(StepCtl* ctl, double* p, int G, int budget, int iters_per_unit,
                                               double rel_tol, double* smem, const double* partk, int exact_k,
                                               bool leader) {
    if (ctl->done) return;
    if (partk && exact_k > 0) {
        // the last fused launch logged the residual of every sweep's input: exact sweep count
        const double tolb = rel_tol * ctl->bnorm;
        int first = -1;
        double rlast = 0.0;
        for (int k = 0; k < exact_k; ++k) {
            rlast = reduce_partials(partk + (int64_t)k * FEMFCT_MAX_PARTIALS, G, OpMax(), 0.0, smem);
            if (first < 0 && rlast <= tolb) { first = k; break; }
        }
        if (leader && threadIdx.x == 0) {
            const double bn = ctl->bnorm;
            ctl->iters = (first >= 0) ? (budget - 1) * iters_per_unit + first + 1 : budget * iters_per_unit;
            ctl->resid = bn > 0.0 ? rlast / bn : 0.0;
            if (first < 0) ctl->flags |= FEMFCT_FLAG_SOLVER_BUDGET;
        }
        return;
    }
    double rmax = (G < 0) ? ctl->rs[(budget - 1) & 1]
                          : reduce_partials(p + ((budget - 1) & 1) * FEMFCT_MAX_PARTIALS, G, OpMax(), 0.0, smem);
    // every block computes the same values; block 0 publishes them for later kernels' diagnostics.
    // parity is derived locally below (ctl->parity is only written here by block 0 and read by
    // later kernels, never by other blocks of this kernel).
    if (leader && threadIdx.x == 0) {
        double bn = ctl->bnorm;
        ctl->iters = budget * iters_per_unit;
        ctl->resid = bn > 0.0 ? rmax / bn : 0.0;
        if (!(rmax <= rel_tol * bn)) ctl->flags |= FEMFCT_FLAG_SOLVER_BUDGET;
    }
}

