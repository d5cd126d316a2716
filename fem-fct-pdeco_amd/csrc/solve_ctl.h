// Device-side bookkeeping of the low-order solve shared by the step kernels.
#pragma once

#include "femfct_internal.h"
#include "device_utils.h"

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

