// Shared machinery of the trajectory drivers (traj.hip, traj_systems.hip).
#pragma once

#include <chrono>

#include "femfct_internal.h"
#include "device_utils.h"
#include "forms.h"

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

int femfct_ensure_traj_ws(femfct_ctx* ctx, int32_t batch, int32_t steps);
int femfct_ensure_krylov_ws(femfct_ctx* ctx, int32_t batch);
int femfct_enqueue_step_end(femfct_ctx* ctx, int delta, int32_t batch, bool with_krylov);
void femfct_request_fused_end(femfct_ctx* ctx, int delta, bool with_krylov);
int femfct_enqueue_step_ref(femfct_ctx* ctx, const double* A, const double* N, int32_t nshared, VecRef rhs,
                            int64_t rhs_bstride, VecRef u_n, int64_t u_bstride, double dt, VecRef u_out,
                            int64_t out_bstride, int32_t batch, int32_t budget);
int femfct_enqueue_step_mat(femfct_ctx* ctx, MatRef A, const double* N, int32_t nshared, VecRef rhs, int64_t rhs_bstride,
                            VecRef u_n, int64_t u_bstride, double dt, VecRef u_out, int64_t out_bstride, int32_t batch,
                            int32_t budget);
int femfct_enqueue_axpby(femfct_ctx* ctx, int64_t count, double alpha, const double* a, double beta, const double* b,
                         double* out);
struct SbOpArgs;
bool femfct_inline_ops_wanted(const femfct_ctx* ctx, int32_t batch);
int femfct_enqueue_step_op(femfct_ctx* ctx, MatRef A, const SbOpArgs* sb, const double* N, int32_t nshared, VecRef rhs,
                           int64_t rhs_bstride, VecRef u_n, int64_t u_bstride, double dt, VecRef u_out, int64_t out_bstride,
                           int32_t batch, int32_t budget);

static inline int femfct_round_kry_budget(const femfct_ctx* ctx, int b) {
    b = (b + 3) & ~3;
    if (b < 4) b = 4;
    if (b > ctx->kry_max_iters) b = ctx->kry_max_iters;
    return b;
}

// Captured graph holding `reps` consecutive, identical time steps (the time level is a device counter).
template <class F>
int femfct_run_graph_reps(femfct_ctx* ctx, femfct_ctx::GraphKey key, int reps, int delta, F&& enqueue_one) {
    key.push_back(key_bits((int32_t)reps));
    key.push_back(key_bits((int32_t)ctx->pair_rows));     // a different Jacobi kernel is captured
    key.push_back(key_bits((int32_t)ctx->solver));
    // delta: the time-level step of this kind of sweep (+1 forward, -1 adjoint).  Step r is enqueued with its level
    // offset baked into every level-indirected reference (lref, MatRef::level_off); the device counters move once, in
    // the last step of the graph -- no per-step ticket / counter update on the critical path of the other R - 1 steps.
    struct Restore {
        femfct_ctx* c;
        ~Restore() { c->level_bias = 0; c->ord_bias = 0; c->rep_total = 1; c->rep_last = true; }
    } restore{ctx};
    return femfct_run_graph(ctx, key, [&]() {
        for (int r = 0; r < reps; ++r) {
            ctx->level_bias = r * delta; ctx->ord_bias = r; ctx->rep_total = reps; ctx->rep_last = (r == reps - 1);
            int rc = enqueue_one();
            if (rc != FEMFCT_OK) return rc;
        }
        return (int)FEMFCT_OK;
    });
}

// level-indirected reference of the step being enqueued (see femfct_run_graph_reps)
static inline VecRef lref(const femfct_ctx* ctx, const double* p, const int32_t* level, int64_t stride, int32_t off) {
    return make_ref(p, level, stride, off + (level ? ctx->level_bias : 0));
}

// Replays `step(jacobi_budget, krylov_budget, reps)` until num_steps are done, then inspects the per-step solver
// logs; if a sweep/iteration budget was too small anywhere the whole sweep is repeated with a
// larger one (the sweep's inputs are never overwritten, so a repeat is exact).
template <class Begin, class Step>
int femfct_run_sweep(femfct_ctx* ctx, int kind, int32_t num_steps, int32_t batch, int level0, bool krylov,
                     Begin&& begin, Step&& step) {
    // budgets are remembered per kind of sweep: the forward and the adjoint operator of a problem
    // need different sweep counts, and a shared budget would make them evict each other
    if (!ctx->kind_budget.count(kind)) ctx->kind_budget[kind] = 48;
    struct RestoreSolver {       // the effective solver is per kind of sweep; the user's choice comes back on every exit
        femfct_ctx* c;
        ~RestoreSolver() { c->solver = c->solver_user; c->pair_rows = false; }
    } restore_solver{ctx};
    for (;;) {
        // bandwidth regime: rows of an upwind operator as one value per opposing pair (k_strip_jacobi_pair_walk), until a
        // sweep of this kind shows a row that is not one (FEMFCT_FLAG_ROW_PAIRS below)
        ctx->pair_rows = !ctx->kind_fullrows.count(kind);
        // a kind whose operator lies outside the scheme's dt restriction (Jacobi does not contract: the reference's
        // spsolve does not care, helpers.py:1782) is solved with Jacobi-preconditioned BiCGStab from then on
        ctx->solver = ctx->kind_low_bicg.count(kind) ? FEMFCT_SOLVER_BICGSTAB : ctx->solver_user;
        if (ctx->solver == FEMFCT_SOLVER_BICGSTAB) {
            int rk = femfct_ensure_krylov_ws(ctx, batch);
            if (rk != FEMFCT_OK) return rk;
        }
        // species solves: Chebyshev (structured mesh) and BiCGStab keep separate iteration budgets
        const bool cheb = krylov && femfct_species_cheb(ctx, kind);
        const int kkey = cheb ? kind : kind + 1000;
        if (!ctx->kind_kbudget.count(kkey)) ctx->kind_kbudget[kkey] = 40;
        // the one-workgroup step (kernels_mesh.hip) stops by itself: its budget is only a cap, kept fixed so that every sweep
        // of the kind replays the same captured graphs (a budget that follows the iteration counts would re-capture them)
        const bool meshp = femfct_mesh_step_wanted(ctx, batch);
        if (meshp && !ctx->kind_mesh_budget.count(kind)) ctx->kind_mesh_budget[kind] = std::min(ctx->max_iters, 96);
        const int budget = meshp ? ctx->kind_mesh_budget[kind] : femfct_round_budget(ctx, ctx->kind_budget[kind]);
        const int kbudget = femfct_round_kry_budget(ctx, ctx->kind_kbudget[kkey]);
        const bool dbg_t = getenv("FEMFCT_DEBUG_TIMES") != nullptr;
        auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tt0 = dbg_t ? now() : 0.0;
        int rc = begin();
        if (rc != FEMFCT_OK) return rc;
        const double tt1 = dbg_t ? now() : 0.0;
        int32_t init[2] = {level0, 0};
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_level, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
        // several identical time steps per captured graph: fewer graph launches, no inter-graph gaps
        const int32_t per_graph = std::max(1, std::min(ctx->steps_per_graph, num_steps));
        for (int32_t k = 0; k < num_steps; k += per_graph) {
            rc = step(budget, kbudget, std::min(per_graph, num_steps - k));
            if (rc != FEMFCT_OK) return rc;
        }
        const double tt2 = dbg_t ? now() : 0.0;
        ctx->log_steps = ctx->log_batch = 0;      // no matching log while the copies are in flight / after a failure
        ctx->h_log.resize((size_t)num_steps * batch);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_log.data(), ctx->d_log, sizeof(StepCtl) * ctx->h_log.size(),
                                    hipMemcpyDeviceToHost, ctx->stream));
        if (krylov) {
            ctx->h_klog.resize(sizeof(KrylovCtl) * (size_t)num_steps * batch);
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_klog.data(), ctx->d_klog, ctx->h_klog.size(), hipMemcpyDeviceToHost,
                                        ctx->stream));
        } else {
            ctx->h_klog.clear();
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (dbg_t) {
            const double tt3 = now();
            if (tt3 - tt0 > atof(getenv("FEMFCT_DEBUG_TIMES")))      // FEMFCT_DEBUG_TIMES=<ms>: report sweeps that take longer
                fprintf(stderr, "[femfct] sweep kind %d took %.1f ms: begin %.2f, enqueue %.2f, wait %.2f\n", kind, tt3 - tt0, tt1 - tt0, tt2 - tt1, tt3 - tt2);
        }
        ctx->log_steps = num_steps;
        ctx->log_batch = batch;
        int worst = 0, kworst = 0;
        bool short_budget = false, kshort = false, coarse = false, not_pairs = false;
        double worst_res = 0.0, kworst_res = 0.0;
        for (const StepCtl& c : ctx->h_log) not_pairs = not_pairs || (c.flags & FEMFCT_FLAG_ROW_PAIRS);
        if (not_pairs) {             // what the pair-compact launches computed is void: repeat with the full-row kernels
            if (getenv("FEMFCT_DEBUG")) fprintf(stderr, "[femfct] sweep kind %d: rows with both entries of a pair -> full-row Jacobi launches\n", kind);
            ctx->kind_fullrows.insert(kind);
            ctx->log_steps = ctx->log_batch = 0;
            continue;
        }
        for (const StepCtl& c : ctx->h_log) {
            if (c.iters > worst) coarse = (c.flags & FEMFCT_FLAG_COARSE_ITERS) != 0;
            worst = std::max(worst, c.iters);
            if (c.flags & FEMFCT_FLAG_SOLVER_BUDGET) {
                short_budget = true;
                worst_res = (c.resid == c.resid) ? std::max(worst_res, c.resid) : INFINITY;   // NaN: diverged
            }
        }
        if (krylov) {
            const KrylovCtl* kl = (const KrylovCtl*)ctx->h_klog.data();
            for (size_t k = 0; k < (size_t)num_steps * batch; ++k) {
                kworst = std::max(kworst, kl[k].iters);
                if ((kl[k].flags & FEMFCT_FLAG_SOLVER_BUDGET) || !(kl[k].resid == kl[k].resid)) {
                    kshort = true;
                    kworst_res = std::max(kworst_res, kl[k].resid == kl[k].resid ? kl[k].resid : 1.0);
                }
            }
        }
        if (getenv("FEMFCT_DEBUG"))
            fprintf(stderr, "[femfct] sweep kind %d: budget %d (krylov %d) worst %d kworst %d short %d/%d, %d graphs captured so far (%zu cached)\n",
                    kind, budget, kbudget, worst, kworst, (int)short_budget, (int)kshort, ctx->graph_captures, ctx->graphs.size());
        if (!short_budget && !kshort && meshp) {
            if (krylov)
                ctx->kind_kbudget[kkey] = std::min(ctx->kry_max_iters, cheb ? std::max(8, kworst + 1)
                                                                             : std::max(8, kworst + kworst / 4 + 2));
            return FEMFCT_OK;
        }
        if (short_budget && meshp && budget < ctx->max_iters && worst_res < 1.0) {
            ctx->kind_mesh_budget[kind] = std::min(ctx->max_iters, 2 * budget);
            short_budget = false;
            if (!kshort) continue;
        }
        if (!short_budget && !kshort) {
            ctx->kind_good[kind] = budget;
            ctx->kind_budget[kind] = femfct_next_budget(ctx, worst, coarse);
            // whole-mesh workgroups stop by themselves: the budget is only an upper bound, keep a margin
            const bool single = femfct_tile4_wanted(ctx, batch) && femfct_single_patch(ctx, batch);
            if (single) ctx->kind_budget[kind] = std::min(ctx->max_iters, worst + 6);
            // Tiles report whole launches: `worst` = U launches of K sweeps, and every step was still above
            // the tolerance after (U-1)*K.  If one launch fewer of the deepest halo could cover that, try it
            // once (a failure is remembered per kind and costs one repeated sweep).
            int K = 0, U = 0, K2 = 0, U2 = 0;
            if (!single && femfct_jacobi_plan(ctx, budget, batch, &K, &U) && worst > 0) {
                U = (worst + K - 1) / K;
                const int lb = (U - 1) * K;
                int top = 0;
                for (int b_try = lb + 1; U >= 2 && b_try < worst; ++b_try)
                    if (femfct_jacobi_plan(ctx, b_try, batch, &K2, &U2) && U2 <= U - 1) top = b_try;
                const int known_fail = ctx->kind_fail.count(kind) ? ctx->kind_fail[kind] : 0;
                if (top > lb && top > known_fail && 10 * top >= 7 * worst) ctx->kind_budget[kind] = top;
            }
            // (Chebyshev reports the count that meets tol/10 at its asymptotic rate: no extra margin)
            if (krylov)
                ctx->kind_kbudget[kkey] = std::min(ctx->kry_max_iters, cheb ? std::max(8, kworst + 1)
                                                                             : std::max(8, kworst + kworst / 4 + 2));
            return FEMFCT_OK;
        }
        if (short_budget) {
            ctx->kind_fail[kind] = std::max(ctx->kind_fail.count(kind) ? ctx->kind_fail[kind] : 0, budget);
            // no contraction at all (residual not below ||b|| after a whole budget, or not a number), or the sweep cap
            // reached: hand this kind of sweep to BiCGStab and repeat it
            const bool hopeless = !(worst_res < 1.0) || budget >= ctx->max_iters;
            if (hopeless && ctx->solver == FEMFCT_SOLVER_JACOBI) {
                if (getenv("FEMFCT_DEBUG"))
                    fprintf(stderr, "[femfct] sweep kind %d: Jacobi residual %.3e after %d sweeps -> BiCGStab\n", kind,
                            worst_res, budget);
                ctx->kind_low_bicg.insert(kind);
                ctx->kind_budget[kind] = 40;
                ctx->kind_fail.erase(kind);
                ctx->kind_good.erase(kind);
                femfct_drop_graphs(ctx);
                continue;
            }
            if (budget >= ctx->max_iters)
                return femfct_fail(ctx, FEMFCT_ERR_NOT_CONVERGED,
                                   "low-order solve: residual %.3e after %d %s (tol %.1e)", worst_res, budget,
                                   ctx->solver == FEMFCT_SOLVER_BICGSTAB ? "BiCGStab iterations" : "Jacobi sweeps",
                                   ctx->rel_tol);
            // a failed attempt at fewer launches goes back to the budget that worked
            const int good = ctx->kind_good.count(kind) ? ctx->kind_good[kind] : 0;
            ctx->kind_budget[kind] = good > budget ? good : femfct_grow_budget(ctx, budget);
        }
        if (kshort && cheb) {
            // not contracting (complex spectrum outside the assumed interval) or out of budget: BiCGStab
            if (!(kworst_res < 10.0) || kbudget >= ctx->kry_max_iters) ctx->kind_cheb_off.insert(kind);
            else ctx->kind_kbudget[kkey] = std::min(ctx->kry_max_iters, std::max(kbudget + 10, kworst + kworst / 10 + 5));
        } else if (kshort) {
            if (kbudget >= ctx->kry_max_iters)
                return femfct_fail(ctx, FEMFCT_ERR_NOT_CONVERGED,
                                   "BiCGStab: residual %.3e after %d iterations (tol %.1e)", kworst_res, kbudget,
                                   ctx->kry_tol);
            ctx->kind_kbudget[kkey] = std::min(ctx->kry_max_iters, kbudget * 2);
        }
    }
}
