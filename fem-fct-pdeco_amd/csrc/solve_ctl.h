// Device-side bookkeeping of the low-order solve shared by the step kernels.
#pragma once

#include "femfct_internal.h"
#include "device_utils.h"

// Deferred test of a two-launch solve (launch 0 = k_tile_build_jacobi, launch 1 = k_tile_jacobi with defer = 1): neither
// Jacobi launch looked at a partial, so ||b||, the minimal row sum and the two residual maxima are all reduced by the
// kernel behind them, in one pass, and the verdict a launch-1 test would have
// reached is reconstructed.  Split in two so that the caller (workgroup 0 of k_tile_dudt_cheb) can request the partials
// before its tile work and publish after it.  What differs from the in-launch test: had launch 0 already converged,
// launch 1 ran nevertheless (its sweeps only lower the residual further; the iterate returned is the budget-parity one).
struct DeferredPartials { double r0, r1, bn, rs; };

// (wave 0 of the calling workgroup only: no LDS, no barrier -- the publish is a handful of lane shifts at the kernel's end)
__device__ __forceinline__ DeferredPartials deferred_test_load(const double* p, int G) {
    DeferredPartials d{0.0, 0.0, 0.0, INFINITY};
    if (threadIdx.x < WAVE) {
        for (int k = threadIdx.x; k < G; k += WAVE) {
            d.r0 = fmax(d.r0, p[k]);
            d.r1 = fmax(d.r1, p[FEMFCT_MAX_PARTIALS + k]);
            d.bn = fmax(d.bn, p[2 * FEMFCT_MAX_PARTIALS + k]);
            d.rs = fmin(d.rs, p[3 * FEMFCT_MAX_PARTIALS + k]);
        }
    }
    return d;
}

__device__ __forceinline__ void deferred_test_publish(StepCtl* ctl, DeferredPartials d, int iters_per_unit, double rel_tol) {
    if (threadIdx.x >= WAVE) return;
    const double r0 = wave_reduce(d.r0, OpMax()), r1 = wave_reduce(d.r1, OpMax());
    const double bn = wave_reduce(d.bn, OpMax()), rs = wave_reduce(d.rs, OpMin());
    if (threadIdx.x == 0) {
        ctl->bnorm = bn;
        ctl->min_rowsum = rs;
        if (!(rs > 0.0)) ctl->flags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        const double tolb = rel_tol * bn;
        if (r0 <= tolb) { ctl->iters = iters_per_unit; ctl->flags |= FEMFCT_FLAG_COARSE_ITERS; }
        else ctl->iters = 2 * iters_per_unit;
        ctl->resid = bn > 0.0 ? r1 / bn : 0.0;
        if (!(r1 <= tolb) && !(r0 <= tolb)) ctl->flags |= FEMFCT_FLAG_SOLVER_BUDGET;
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

