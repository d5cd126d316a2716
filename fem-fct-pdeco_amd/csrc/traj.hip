// Forward / adjoint trajectory sweeps resident on one GPU.
//
// One time step = [assemble the step's flux matrix] -> [FCT step] -> [advance level];
// the whole sequence is captured once into a hipGraph and replayed num_steps times.
// The current time level lives in a device-side counter that the captured kernels
// read (VecRef), so the replayed graph needs no per-step host patching.
//
//   solid-body rotation + drift control
//     forward  /root/reference/advection_solidbody_FCT_PDECO_finaltime.py:175-193
//     adjoint  /root/reference/advection_solidbody_FCT_PDECO_finaltime.py:200-221
//              /root/reference/advection_solidbody_FCT_PDECO_alltime.py:232-259
#include "femfct_internal.h"
#include "device_utils.h"
#include "forms.h"
#include "traj_common.h"
#include "solidbody_op.h"

#include <algorithm>

int femfct_enqueue_step_ref(femfct_ctx* ctx, const double* A, const double* N, int32_t nshared, VecRef rhs,
                            int64_t rhs_bstride, VecRef u_n, int64_t u_bstride, double dt, VecRef u_out,
                            int64_t out_bstride, int32_t batch, int32_t budget);
int femfct_enqueue_ops_solidbody(femfct_ctx* ctx, const double* Arot, VecRef c_ref, int64_t c_bstride, double eps,
                                 double sigma, double rot_scale, double bx, double by, double* A, int32_t batch,
                                 int32_t levels = 1);

// The solid-body operator depends on the control only, which is given for the whole sweep: assemble the
// matrices of all time levels in one launch before the sweep (HBM is large: Nt * 7n doubles per control
// trajectory, 92 MB at C2) instead of one small dependent launch per step.  Returns false (per-step
// assembly) when the sequence would not fit the configured cap or the launch grid.
static bool solidbody_preassemble(femfct_ctx* ctx, const double* Arot, const double* c_traj, int32_t c_shared,
                                  int64_t tstride, int32_t c_level0, double eps, double sigma, double rot_scale, double bx,
                                  double by, int32_t num_steps, int32_t batch, MatRef* out) {
    const int32_t members = c_shared ? 1 : batch;
    const size_t count = (size_t)members * num_steps * ctx->W * ctx->n;
    if (!ctx->preassemble || (double)count * 8.0 > ctx->preassemble_max_bytes) return false;
    if ((int64_t)members * num_steps > 65535) return false;
    if (count > ctx->trAall_count) {
        femfct_drop_graphs(ctx);
        if (ctx->d_trAall) hipFree(ctx->d_trAall);
        ctx->d_trAall = nullptr; ctx->trAall_count = 0;
        if (hipMalloc((void**)&ctx->d_trAall, sizeof(double) * count) != hipSuccess) { (void)hipGetLastError(); return false; }
        ctx->trAall_count = count;
    }
    const int64_t n = ctx->n;
    femfct_enqueue_ops_solidbody(ctx, Arot, make_ref(c_traj, nullptr, n, c_level0), c_shared ? 0 : tstride, eps, sigma,
                                 rot_scale, bx, by, ctx->d_trAall, members, num_steps);
    const int64_t wn = (int64_t)ctx->W * n;
    *out = MatRef{ctx->d_trAall, ctx->d_level, wn, 0, c_shared ? 0 : wn * num_steps};
    return true;
}
int femfct_enqueue_mass_diff(femfct_ctx* ctx, VecRef a, int64_t a_bstride, VecRef b, int64_t b_bstride, double* out,
                             int32_t batch);
int femfct_enqueue_axpby(femfct_ctx* ctx, int64_t count, double alpha, const double* a, double beta, const double* b,
                         double* out);

namespace {

// end of a step: log the solver control blocks, then move the level counter
__global__ void k_step_end(int32_t* level, int delta, int ord_adv, int ord_off, const StepCtl* __restrict__ ctl,
                           StepCtl* __restrict__ log, const KrylovCtl* __restrict__ kctl, KrylovCtl* __restrict__ klog,
                           int batch) {
    const int ord = level[1];
    for (int b = threadIdx.x; b < batch; b += blockDim.x) {
        log[(int64_t)(ord + ord_off) * batch + b] = ctl[b];
        if (kctl) klog[(int64_t)(ord + ord_off) * batch + b] = kctl[b];
    }
    __syncthreads();
    if (threadIdx.x == 0 && ord_adv) {      // the last step of a graph moves the counters for all of its steps
        level[0] += delta;
        level[1] = ord + ord_adv;
    }
}

}  // namespace

// The FCT step is the last operation of this kind of time step: let its final kernel log and advance.
void femfct_request_fused_end(femfct_ctx* ctx, int delta, bool with_krylov) {
    ctx->end_req_delta = ctx->fuse_end ? delta : 0;
    ctx->end_req_krylov = with_krylov;
    ctx->end_fused = false;
}

int femfct_enqueue_step_end(femfct_ctx* ctx, int delta, int32_t batch, bool with_krylov) {
    ctx->end_req_delta = 0;
    if (ctx->end_fused) {   // already done by the step's last kernel
        ctx->end_fused = false;
        return FEMFCT_OK;
    }
    hipLaunchKernelGGL(k_step_end, dim3(1), dim3(64), 0, ctx->stream, ctx->d_level, ctx->rep_last ? delta * ctx->rep_total : 0,
                       ctx->rep_last ? ctx->rep_total : 0, ctx->ord_bias, ctx->d_ctl, ctx->d_log,
                       with_krylov ? (const KrylovCtl*)ctx->d_kry_ctl : nullptr, (KrylovCtl*)ctx->d_klog, batch);
    return FEMFCT_OK;
}

int femfct_ensure_traj_ws(femfct_ctx* ctx, int32_t batch, int32_t steps) {
    int rc = femfct_ensure_workspace(ctx, batch);
    if (rc != FEMFCT_OK) return rc;
    if (batch > ctx->tr_batch) {
        femfct_drop_graphs(ctx);
        if (ctx->d_trA) hipFree(ctx->d_trA);
        if (ctx->d_trN) hipFree(ctx->d_trN);
        if (ctx->d_trRhs) hipFree(ctx->d_trRhs);
        ctx->d_trA = ctx->d_trN = ctx->d_trRhs = nullptr;
        size_t nv = (size_t)batch * ctx->n, nm = nv * ctx->W;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trA, sizeof(double) * nm));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trN, sizeof(double) * nm));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trRhs, sizeof(double) * nv));
        for (double** q : {&ctx->d_trMat, &ctx->d_trBase, &ctx->d_trBase2, &ctx->d_trRhs2, &ctx->d_trTmp}) {
            if (*q) hipFree(*q);
            *q = nullptr;
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trMat, sizeof(double) * nm));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trBase, sizeof(double) * ctx->W * ctx->n));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trBase2, sizeof(double) * ctx->W * ctx->n));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trRhs2, sizeof(double) * nv));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trTmp, sizeof(double) * nv));
        if (ctx->d_log) hipFree(ctx->d_log);
        ctx->d_log = nullptr;
        ctx->tr_steps = 0;
        ctx->tr_batch = batch;
    }
    if (!ctx->d_level) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_level, sizeof(int32_t) * 2));
    if (!ctx->d_ticket) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ticket, sizeof(unsigned)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_ticket, 0, sizeof(unsigned), ctx->stream));
    }
    if (steps > ctx->tr_steps || !ctx->d_log) {
        femfct_drop_graphs(ctx);
        if (ctx->d_log) hipFree(ctx->d_log);
        ctx->d_log = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_log, sizeof(StepCtl) * (size_t)steps * ctx->tr_batch));
        if (ctx->d_klog) hipFree(ctx->d_klog);
        ctx->d_klog = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_klog, sizeof(KrylovCtl) * (size_t)steps * ctx->tr_batch));
        ctx->tr_steps = steps;
    }
    return FEMFCT_OK;
}

extern "C" {

int femfct_solidbody_forward(femfct_ctx* ctx, const double* Arot_ell, const double* c_traj, int32_t c_shared,
                             double* u_traj, int32_t num_steps, double dt, double eps, double rot_scale, double bx,
                             double by, int32_t batch) {
    return femfct_solidbody_forward_src(ctx, Arot_ell, c_traj, c_shared, nullptr, u_traj, num_steps, dt, eps, rot_scale, bx,
                                        by, batch);
}

// the same sweep with a source trajectory: rhs_{n+1} = assemble(src_{n+1} * v * dx) = M src_{n+1}
// (advection_FCT_PDECO_alltime_exact.py:249-253: u_rhs = assemble((g_np1 + c_np1)*v*dx), A_u = A - eps*Ad)
int femfct_solidbody_forward_src(femfct_ctx* ctx, const double* Arot_ell, const double* c_traj, int32_t c_shared,
                                 const double* src_traj, double* u_traj, int32_t num_steps, double dt, double eps,
                                 double rot_scale, double bx, double by, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->structured, "structured mesh not set (femfct_set_mesh_square)");
    ARG_TRY(ctx, c_traj && u_traj && num_steps >= 1 && dt > 0 && batch >= 1, "bad argument");
    ARG_TRY(ctx, Arot_ell || rot_scale == 0.0, "Arot_ell is required when rot_scale != 0");
    int rc = femfct_ensure_traj_ws(ctx, batch, num_steps);
    if (rc != FEMFCT_OK) return rc;
    const int64_t n = ctx->n, tstride = (int64_t)(num_steps + 1) * n;
    const double* Arot = Arot_ell ? Arot_ell : ctx->d_Ad;  // any valid ELL array; multiplied by 0
    int32_t* lv = ctx->d_level;
    MatRef Aall{};
    bool pre = false;
    // step k (level counter k) uses the control of level k+1 (finaltime.py:185): sequence entry k
    // bandwidth regime: the step kernels derive the operator from Arot and the control themselves (no stored A)
    bool inl = false;
    int rotg = 0;
    double rot_om = 0.0;
    auto begin = [&]() {
        inl = femfct_inline_ops_wanted(ctx, batch);     // (inside the sweep driver: depends on the kind's Jacobi kernel)
        rotg = inl && Arot_ell && rot_scale != 0.0 && femfct_rotation_is_geometric(ctx, Arot, &rot_om);
        ctx->last_rot_geom = rotg != 0;
        if (!inl)
            pre = solidbody_preassemble(ctx, Arot, c_traj, c_shared, tstride, 1, eps, -1.0, rot_scale, bx, by, num_steps,
                                        batch, &Aall);
        return FEMFCT_OK;
    };
    auto step = [&](int budget, int, int reps) {
        femfct_ctx::GraphKey key{(uint64_t)2, key_bits(Arot), key_bits(c_traj), key_bits(c_shared), key_bits(u_traj),
                                 key_bits(num_steps), key_bits(dt), key_bits(eps), key_bits(rot_scale), key_bits(bx),
                                 key_bits(by), key_bits(batch), key_bits((int32_t)budget), key_bits(ctx->rel_tol),
                                 key_bits(pre ? Aall.base : nullptr), key_bits(src_traj), key_bits((int32_t)inl),
                                 key_bits((int32_t)rotg), key_bits(rot_om)};
        return femfct_run_graph_reps(ctx, key, reps, +1, [&]() {
            // control at level n+1 (finaltime.py:185), state from level n into level n+1
            MatRef A = Aall;
            if (A.level) A.level_off += ctx->level_bias;
            const SbOpArgs sb{Arot, ctx->d_Ad, lref(ctx, c_traj, lv, n, 1), c_shared ? 0 : tstride, eps, -1.0, rot_scale, bx, by,
                              rotg, rot_om, ctx->a1};
            if (!pre && !inl) {
                femfct_enqueue_ops_solidbody(ctx, Arot, lref(ctx, c_traj, lv, n, 1), c_shared ? 0 : tstride, eps, -1.0,
                                             rot_scale, bx, by, ctx->d_trA, batch);
                A = MatRef{ctx->d_trA, nullptr, 0, 0, (int64_t)ctx->W * n};
            }
            VecRef rhs = make_ref(nullptr);
            if (src_traj) {
                femfct_enqueue_mass_diff(ctx, lref(ctx, src_traj, lv, n, 1), tstride, make_ref(nullptr), 0, ctx->d_trRhs, batch);
                rhs = make_ref(ctx->d_trRhs);
            }
            femfct_request_fused_end(ctx, 1, false);
            int r = femfct_enqueue_step_op(ctx, A, inl ? &sb : nullptr, nullptr, 0, rhs, n,
                                           lref(ctx, u_traj, lv, n, 0), tstride, dt, lref(ctx, u_traj, lv, n, 1),
                                           tstride, batch, budget);
            if (r != FEMFCT_OK) return r;
            femfct_enqueue_step_end(ctx, 1, batch, false);
            return FEMFCT_OK;
        });
    };
    // (a diffusive operator has no upwind rows: no point in finding that out from a whole sweep with the pair-compact launch)
    if (eps != 0.0) ctx->kind_fullrows.insert(2);
    return femfct_run_sweep(ctx, 2, num_steps, batch, 0, false, begin, step);
}

int femfct_solidbody_adjoint(femfct_ctx* ctx, const double* Arot_ell, const double* c_traj, int32_t c_shared,
                             const double* u_traj, const double* uhat, double* p_traj, int32_t num_steps, double dt,
                             double eps, double rot_scale, double bx, double by, int32_t alltime, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->structured, "structured mesh not set (femfct_set_mesh_square)");
    ARG_TRY(ctx, c_traj && u_traj && uhat && p_traj && num_steps >= 1 && dt > 0 && batch >= 1, "bad argument");
    ARG_TRY(ctx, Arot_ell || rot_scale == 0.0, "Arot_ell is required when rot_scale != 0");
    int rc = femfct_ensure_traj_ws(ctx, batch, num_steps);
    if (rc != FEMFCT_OK) return rc;
    const int64_t n = ctx->n, tstride = (int64_t)(num_steps + 1) * n;
    const double* Arot = Arot_ell ? Arot_ell : ctx->d_Ad;
    int32_t* lv = ctx->d_level;
    MatRef Aall{};
    bool pre = false;
    bool inl = false;
    int rotg = 0;
    double rot_om = 0.0;
    auto begin = [&]() {
        inl = femfct_inline_ops_wanted(ctx, batch);
        rotg = inl && Arot_ell && rot_scale != 0.0 && femfct_rotation_is_geometric(ctx, Arot, &rot_om);
        ctx->last_rot_geom = rotg != 0;
        // level counter n uses the control of level n (finaltime.py:213): sequence entry n
        if (!inl)
            pre = solidbody_preassemble(ctx, Arot, c_traj, c_shared, tstride, 0, eps, +1.0, rot_scale, bx, by, num_steps,
                                        batch, &Aall);
        // terminal condition: p(T) = uhat_T - u(T) (finaltime.py:201) or 0 (alltime.py:232)
        for (int32_t b = 0; b < batch; ++b) {
            double* pT = p_traj + b * tstride + (int64_t)num_steps * n;
            if (alltime) HIP_TRY(ctx, hipMemsetAsync(pT, 0, sizeof(double) * n, ctx->stream));
            else femfct_enqueue_axpby(ctx, n, 1.0, uhat + (int64_t)b * n, -1.0, u_traj + b * tstride + (int64_t)num_steps * n, pT);
        }
        return FEMFCT_OK;
    };
    auto step = [&](int budget, int, int reps) {
        femfct_ctx::GraphKey key{(uint64_t)3, key_bits(Arot), key_bits(c_traj), key_bits(c_shared), key_bits(u_traj),
                                 key_bits(uhat), key_bits(p_traj), key_bits(num_steps), key_bits(dt), key_bits(eps),
                                 key_bits(rot_scale), key_bits(bx), key_bits(by), key_bits(alltime), key_bits(batch),
                                 key_bits((int32_t)budget), key_bits(ctx->rel_tol), key_bits(pre ? Aall.base : nullptr),
                                 key_bits((int32_t)inl), key_bits((int32_t)rotg), key_bits(rot_om)};
        return femfct_run_graph_reps(ctx, key, reps, -1, [&]() {
            // level counter = n: control c_n (finaltime.py:213), p_{n+1} -> p_n
            MatRef A = Aall;
            if (A.level) A.level_off += ctx->level_bias;
            const SbOpArgs sb{Arot, ctx->d_Ad, lref(ctx, c_traj, lv, n, 0), c_shared ? 0 : tstride, eps, +1.0, rot_scale, bx, by,
                              rotg, rot_om, ctx->a1};
            if (!pre && !inl) {
                femfct_enqueue_ops_solidbody(ctx, Arot, lref(ctx, c_traj, lv, n, 0), c_shared ? 0 : tstride, eps, +1.0,
                                             rot_scale, bx, by, ctx->d_trA, batch);
                A = MatRef{ctx->d_trA, nullptr, 0, 0, (int64_t)ctx->W * n};
            }
            VecRef rhs = make_ref(nullptr);
            if (alltime) {  // rhs = assemble((uhat_n - u_n) v dx)  (alltime.py:257)
                femfct_enqueue_mass_diff(ctx, lref(ctx, uhat, lv, n, 0), tstride, lref(ctx, u_traj, lv, n, 0), tstride,
                                         ctx->d_trRhs, batch);
                rhs = make_ref(ctx->d_trRhs);
            }
            femfct_request_fused_end(ctx, -1, false);
            int r = femfct_enqueue_step_op(ctx, A, inl ? &sb : nullptr, nullptr, 0, rhs, n, lref(ctx, p_traj, lv, n, 1), tstride,
                                           dt, lref(ctx, p_traj, lv, n, 0), tstride, batch, budget);
            if (r != FEMFCT_OK) return r;
            femfct_enqueue_step_end(ctx, -1, batch, false);
            return FEMFCT_OK;
        });
    };
    if (eps != 0.0) ctx->kind_fullrows.insert(3);
    return femfct_run_sweep(ctx, 3, num_steps, batch, num_steps - 1, false, begin, step);
}

// per-step solver diagnostics of the most recent trajectory sweep: info[step*batch + b]
int femfct_traj_info(femfct_ctx* ctx, femfct_step_info* info_host, int32_t num_steps, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && info_host, "null argument");
    ARG_TRY(ctx, num_steps == ctx->log_steps && batch == ctx->log_batch, "no matching trajectory log");
    for (size_t k = 0; k < ctx->h_log.size(); ++k) {
        info_host[k].flags = ctx->h_log[k].flags;
        info_host[k].solver_iters = ctx->h_log[k].iters;
        info_host[k].solver_resid = ctx->h_log[k].resid;
        info_host[k].min_rowsum = ctx->h_log[k].min_rowsum;
    }
    return FEMFCT_OK;
}

}  // extern "C"
