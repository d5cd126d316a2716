// Forward / adjoint trajectory sweeps of the reference's three PDE systems, resident on one GPU:
//   solve_nonlinear_equation / solve_adjoint_nonlinear_equation   /root/reference/helpers.py:881-1038
//   solve_schnak_system      / solve_adjoint_schnak_system        /root/reference/helpers.py:511-698
//   solve_chtxs_system       / solve_adjoint_chtxs_system         /root/reference/helpers.py:1250-1581
// One time step = [assemble the step's operators] -> [FCT step] and/or [BiCGStab for the non-FCT
// species] -> [log + advance level]; captured once as a hipGraph and replayed per step.
#include "traj_common.h"


namespace {

__global__ void k_ell_transpose(int n, int W, const int32_t* __restrict__ cols, const uint8_t* __restrict__ tslot,
                                const double* __restrict__ in, double* __restrict__ out) {
    RowRange rr = block_rows(n);
    for (int i = rr.begin + threadIdx.x; i < rr.end; i += blockDim.x) {
        out[i] = in[i];
        for (int s = 1; s < W; ++s) {
            int64_t idx = (int64_t)s * n + i;
            out[idx] = in[(int64_t)tslot[idx] * n + cols[idx]];
        }
    }
}

// out = alpha*a + (beta * scale[*level + off]) * b: an operator whose convection part carries a time-dependent scalar
// factor (separable wind w(x, t) = s(t) w0(x): A(t_n) = s(t_n) * A0), refreshed per step inside the captured graph
__global__ void k_axpby_level(int64_t count, double alpha, const double* __restrict__ a, double beta,
                              const double* __restrict__ b, const double* __restrict__ scale,
                              const int32_t* __restrict__ level, int off, double* __restrict__ out) {
    const double bs = beta * scale[*level + off];
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < count; k += stride) out[k] = alpha * a[k] + bs * b[k];
}

int enqueue_axpby_level(femfct_ctx* ctx, int64_t count, double alpha, const double* a, double beta, const double* b,
                        const double* scale, int off, double* out) {
    int64_t g = (count + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_axpby_level, dim3((unsigned)g), dim3(256), 0, ctx->stream, count, alpha, a, beta, b, scale,
                       ctx->d_level, off + ctx->level_bias, out);
    return FEMFCT_OK;
}

// per-level wind factors s(t_0..t_Nt) of a sweep, uploaded into the ctx-owned buffer (null: stationary wind)
int upload_wind_scale(femfct_ctx* ctx, const double* scale_host, int32_t num_steps, const double** dev) {
    *dev = nullptr;
    if (!scale_host) return FEMFCT_OK;
    const size_t cnt = (size_t)num_steps + 1;
    if (cnt > ctx->wscale_count) {
        femfct_drop_graphs(ctx);
        if (ctx->d_wscale) hipFree(ctx->d_wscale);
        ctx->d_wscale = nullptr; ctx->wscale_count = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_wscale, sizeof(double) * cnt));
        ctx->wscale_count = cnt;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_wscale, scale_host, sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // scale_host may be a temporary of the caller
    *dev = ctx->d_wscale;
    return FEMFCT_OK;
}

struct Lv {  // VecRef factory bound to the ctx level counter (and to the step of the graph being enqueued)
    const femfct_ctx* ctx;
    const int32_t* lv;
    int64_t n;
    VecRef operator()(const double* base, int off) const { return lref(ctx, base, lv, n, off); }
};

#define KEY(...) femfct_ctx::GraphKey { __VA_ARGS__ }

int check_common(femfct_ctx* ctx, int32_t num_steps, double dt, int32_t batch) {
    ARG_TRY(ctx, ctx && ctx->structured, "structured mesh not set (femfct_set_mesh_square)");
    ARG_TRY(ctx, num_steps >= 1 && dt > 0 && batch >= 1, "need num_steps >= 1, dt > 0, batch >= 1");
    return FEMFCT_OK;
}

// terminal condition p(T) = target - state(T) for every batch member (helpers.py:660-661,1020,1476-1477)
int terminal_diff(femfct_ctx* ctx, const double* target, const double* state, double* adj, int32_t num_steps,
                  int32_t batch) {
    const int64_t n = ctx->n, ts = (int64_t)(num_steps + 1) * n;
    for (int32_t b = 0; b < batch; ++b) {
        int rc = femfct_enqueue_axpby(ctx, n, 1.0, target + (int64_t)b * n, -1.0, state + b * ts + (int64_t)num_steps * n,
                                      adj + b * ts + (int64_t)num_steps * n);
        if (rc != FEMFCT_OK) return rc;
    }
    return FEMFCT_OK;
}

}  // namespace

extern "C" {

// out = in^T on the registered (structurally symmetric) pattern: assemble_sparse(dot(wind,grad(u))*w*dx)
// is the transpose of assemble_sparse(dot(wind,grad(w))*u*dx)  (helpers.py:581 vs 681)
int femfct_ell_transpose(femfct_ctx* ctx, const double* in_ell, double* out_ell) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0 && in_ell && out_ell && in_ell != out_ell, "bad argument");
    LaunchGeom g = femfct_geom(ctx, 1);
    hipLaunchKernelGGL(k_ell_transpose, g.grid, g.block, 0, ctx->stream, ctx->n, ctx->W, ctx->d_cols, ctx->d_tslot,
                       in_ell, out_ell);
    return FEMFCT_OK;
}

// out = alpha*a + beta*b over `count` doubles (b may be NULL); matrix/vector combinations such as
// Du*Ad - omega1*A  (helpers.py:583)
int femfct_axpby(femfct_ctx* ctx, int64_t count, double alpha, const double* a_dev, double beta, const double* b_dev,
                 double* out_dev) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && a_dev && out_dev && count >= 0, "bad argument");
    return femfct_enqueue_axpby(ctx, count, alpha, a_dev, beta, b_dev, out_dev);
}

int femfct_set_krylov(femfct_ctx* ctx, double rel_tol, int32_t max_iters) {
    ARG_TRY(ctx, ctx && rel_tol > 0 && rel_tol < 1 && max_iters >= 1, "bad tolerance / iteration cap");
    ctx->kry_tol = rel_tol;
    ctx->kry_max_iters = max_iters;
    if (ctx->kry_budget > max_iters) ctx->kry_budget = max_iters;
    femfct_drop_graphs(ctx);
    return FEMFCT_OK;
}

int femfct_set_species_solver(femfct_ctx* ctx, int32_t mode) {
    ARG_TRY(ctx, ctx && (mode == FEMFCT_SPECIES_AUTO || mode == FEMFCT_SPECIES_BICGSTAB), "unknown species solver");
    ctx->species_solver = mode;
    ctx->kind_cheb_off.clear();
    return FEMFCT_OK;
}

// spsolve(Mat, b) replacement for the non-FCT implicit solves (helpers.py:596,686,1342,1538):
// Jacobi-preconditioned BiCGStab, x0 = initial guess, synchronises, FEMFCT_ERR_NOT_CONVERGED on failure.
int femfct_bicgstab(femfct_ctx* ctx, const double* mat_ell, int32_t mat_shared, const double* b_dev,
                    const double* x0_dev, double* x_dev, int32_t batch, femfct_step_info* info_host) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0 && mat_ell && b_dev && x0_dev && x_dev && batch >= 1, "bad argument");
    int rc = femfct_ensure_krylov_ws(ctx, batch);
    if (rc != FEMFCT_OK) return rc;
    for (;;) {
        const int budget = femfct_round_kry_budget(ctx, ctx->kry_budget);
        rc = femfct_enqueue_bicgstab(ctx, mat_ell, mat_shared, b_dev, make_ref(x0_dev), ctx->n, make_ref(x_dev), ctx->n,
                                     batch, budget);
        if (rc != FEMFCT_OK) return rc;
        std::vector<KrylovCtl> h(batch);
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_kry_ctl, sizeof(KrylovCtl) * batch, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        bool bad = false;
        int worst = 0;
        double wres = 0.0;
        for (int b = 0; b < batch; ++b) {
            worst = std::max(worst, h[b].iters);
            bool nan = !(h[b].resid == h[b].resid);
            if ((h[b].flags & FEMFCT_FLAG_SOLVER_BUDGET) || nan) { bad = true; wres = std::max(wres, nan ? 1.0 : h[b].resid); }
            if (info_host) {
                info_host[b].flags = h[b].flags;
                info_host[b].solver_iters = h[b].iters;
                info_host[b].solver_resid = h[b].resid;
                info_host[b].min_rowsum = 0.0;
            }
        }
        if (!bad) {
            ctx->kry_budget = std::min(ctx->kry_max_iters, std::max(8, worst + worst / 4 + 2));
            return FEMFCT_OK;
        }
        if (budget >= ctx->kry_max_iters)
            return femfct_fail(ctx, FEMFCT_ERR_NOT_CONVERGED, "BiCGStab: residual %.3e after %d iterations (tol %.1e)",
                               wres, budget, ctx->kry_tol);
        ctx->kry_budget = std::min(ctx->kry_max_iters, budget * 2);
    }
}

// ------------------------------------------------------------------ nonlinear equation
// du/dt + div(-eps grad u + w u) - u + u^3/3 = c      helpers.py:881-966
// Aw_ell = assemble_sparse(dot(wind, grad(v))*u*dx); c_level = the control level the reference
// freezes for the whole sweep (level 1: helpers.py:950-951) as n doubles per batch member.
int femfct_nonlinear_forward(femfct_ctx* ctx, const double* Aw_ell, const double* c_level, double* u_traj,
                             int32_t num_steps, double dt, double eps, int32_t batch) {
    FEMFCT_ENTER(ctx);
    int rc = check_common(ctx, num_steps, dt, batch);
    if (rc != FEMFCT_OK) return rc;
    ARG_TRY(ctx, Aw_ell && c_level && u_traj, "null argument");
    if ((rc = femfct_ensure_traj_ws(ctx, batch, num_steps)) != FEMFCT_OK) return rc;
    const int64_t n = ctx->n, wn = (int64_t)ctx->W * n, ts = (int64_t)(num_steps + 1) * n;
    Lv L{ctx, ctx->d_level, n};
    auto begin = [&]() {
        // FCT_alg_ref(-Mat_var1, ...): A = eps*Ad - Aw (helpers.py:935,957); rhs = assemble(c*v*dx) (:956)
        for (int32_t b = 0; b < batch; ++b) femfct_enqueue_axpby(ctx, wn, eps, ctx->d_Ad, -1.0, Aw_ell, ctx->d_trA + b * wn);
        LoadSpec ls;
        ls.s1 = 1.0; ls.k1 = 1.0; ls.p1 = make_ref(c_level); ls.p1_bs = n;
        return femfct_enqueue_load(ctx, ls, ctx->d_trRhs, batch);
    };
    auto step = [&](int budget, int, int reps) {
        auto key = KEY((uint64_t)10, key_bits(Aw_ell), key_bits(c_level), key_bits(u_traj), key_bits(num_steps),
                       key_bits(dt), key_bits(eps), key_bits(batch), key_bits((int32_t)budget), key_bits(ctx->rel_tol));
        return femfct_run_graph_reps(ctx, key, reps, +1, [&]() {
            WMassSpec ws;  // Mat_rhs = -M + M_u2/3 (helpers.py:953-955)
            ws.alpha = -1.0; ws.beta = 1.0 / 3.0; ws.f1 = L(u_traj, 0); ws.f2 = L(u_traj, 0); ws.f1_bs = ws.f2_bs = ts;
            femfct_enqueue_weighted_mass(ctx, ws, ctx->d_trN, batch);
            femfct_request_fused_end(ctx, 1, false);
            int r = femfct_enqueue_step_ref(ctx, ctx->d_trA, ctx->d_trN, 0, make_ref(ctx->d_trRhs), n, L(u_traj, 0), ts,
                                            dt, L(u_traj, 1), ts, batch, budget);
            if (r != FEMFCT_OK) return r;
            return femfct_enqueue_step_end(ctx, 1, batch, false);
        });
    };
    ctx->kind_fullrows.insert(10);      // (diffusion / reaction terms: rows with both entries of a pair from the start)
    return femfct_run_sweep(ctx, 10, num_steps, batch, 0, false, begin, step);
}

// helpers.py:968-1038: p(T) = uhat_T - u(T); FCT_alg_ref(-Mat_p, 0, p_{n+1}, non_flux_mat = M_u2(u_n) - M)
int femfct_nonlinear_adjoint(femfct_ctx* ctx, const double* Aw_ell, const double* u_traj, const double* uhat_T,
                             double* p_traj, int32_t num_steps, double dt, double eps, int32_t batch) {
    FEMFCT_ENTER(ctx);
    int rc = check_common(ctx, num_steps, dt, batch);
    if (rc != FEMFCT_OK) return rc;
    ARG_TRY(ctx, Aw_ell && u_traj && uhat_T && p_traj, "null argument");
    if ((rc = femfct_ensure_traj_ws(ctx, batch, num_steps)) != FEMFCT_OK) return rc;
    const int64_t n = ctx->n, wn = (int64_t)ctx->W * n, ts = (int64_t)(num_steps + 1) * n;
    Lv L{ctx, ctx->d_level, n};
    auto begin = [&]() {
        for (int32_t b = 0; b < batch; ++b) femfct_enqueue_axpby(ctx, wn, eps, ctx->d_Ad, 1.0, Aw_ell, ctx->d_trA + b * wn);
        return terminal_diff(ctx, uhat_T, u_traj, p_traj, num_steps, batch);
    };
    auto step = [&](int budget, int, int reps) {
        auto key = KEY((uint64_t)11, key_bits(Aw_ell), key_bits(u_traj), key_bits(uhat_T), key_bits(p_traj),
                       key_bits(num_steps), key_bits(dt), key_bits(eps), key_bits(batch), key_bits((int32_t)budget),
                       key_bits(ctx->rel_tol));
        return femfct_run_graph_reps(ctx, key, reps, -1, [&]() {
            WMassSpec ws;  // Mat_rhs = M_u2(u_n) - M (helpers.py:1032-1034)
            ws.alpha = -1.0; ws.beta = 1.0; ws.f1 = L(u_traj, 0); ws.f2 = L(u_traj, 0); ws.f1_bs = ws.f2_bs = ts;
            femfct_enqueue_weighted_mass(ctx, ws, ctx->d_trN, batch);
            femfct_request_fused_end(ctx, -1, false);
            int r = femfct_enqueue_step_ref(ctx, ctx->d_trA, ctx->d_trN, 0, make_ref(nullptr), 0, L(p_traj, 1), ts, dt,
                                            L(p_traj, 0), ts, batch, budget);
            if (r != FEMFCT_OK) return r;
            return femfct_enqueue_step_end(ctx, -1, batch, false);
        });
    };
    ctx->kind_fullrows.insert(11);      // (diffusion / reaction terms: rows with both entries of a pair from the start)
    return femfct_run_sweep(ctx, 11, num_steps, batch, num_steps - 1, false, begin, step);
}

// ------------------------------------------------------------------ advective Schnakenberg
// helpers.py:511-597.  par = {Du, Dv, c_b, gamma, omega1, omega2}
int femfct_schnak_forward(femfct_ctx* ctx, const double* Aw_ell, const double* c_level, double* u_traj,
                          double* v_traj, int32_t num_steps, double dt, const double* par, double rescaling,
                          int32_t batch) {
    return femfct_schnak_forward_tw(ctx, Aw_ell, nullptr, c_level, u_traj, v_traj, num_steps, dt, par, rescaling, batch);
}

// The same with a separable time-dependent wind w(x, t) = s(t) w0(x) (Schnak_FCT_PDECO_alltime.py:55,174-175:
// sin(2 pi t) * rotation, re-assembled every step; helpers.py:566 sets wind.t = t_{n+1} before assembling):
// Aw_ell = assemble(dot(w0, grad(v))*u*dx), wind_scale_host[k] = s(t_k), k = 0..num_steps; the step to level n+1
// uses s(t_{n+1}).  NULL: stationary wind.
int femfct_schnak_forward_tw(femfct_ctx* ctx, const double* Aw_ell, const double* wind_scale_host, const double* c_level,
                             double* u_traj, double* v_traj, int32_t num_steps, double dt, const double* par,
                             double rescaling, int32_t batch) {
    FEMFCT_ENTER(ctx);
    int rc = check_common(ctx, num_steps, dt, batch);
    if (rc != FEMFCT_OK) return rc;
    ARG_TRY(ctx, Aw_ell && c_level && u_traj && v_traj && par && rescaling != 0.0, "bad argument");
    if ((rc = femfct_ensure_traj_ws(ctx, batch, num_steps)) != FEMFCT_OK) return rc;
    if ((rc = femfct_ensure_krylov_ws(ctx, batch)) != FEMFCT_OK) return rc;
    const double Du = par[0], Dv = par[1], c_b = par[2], gam = par[3], om1 = par[4], om2 = par[5];
    const int64_t n = ctx->n, wn = (int64_t)ctx->W * n, ts = (int64_t)(num_steps + 1) * n;
    Lv L{ctx, ctx->d_level, n};
    const double* wsc = nullptr;
    if ((rc = upload_wind_scale(ctx, wind_scale_host, num_steps, &wsc)) != FEMFCT_OK) return rc;
    // the wind-dependent operators: once per sweep for a stationary wind, once per step (level + lvoff) otherwise
    auto wind_ops = [&](int lvoff) {
        // Mat_var1 = Du*Ad - omega1*A (helpers.py:583)
        for (int32_t b = 0; b < batch; ++b) {
            if (wsc) enqueue_axpby_level(ctx, wn, Du, ctx->d_Ad, -om1, Aw_ell, wsc, lvoff, ctx->d_trA + b * wn);
            else femfct_enqueue_axpby(ctx, wn, Du, ctx->d_Ad, -om1, Aw_ell, ctx->d_trA + b * wn);
        }
        // Base = M + dt*(Dv*Ad - omega2*A)  (helpers.py:595 without the u-dependent part)
        if (wsc) enqueue_axpby_level(ctx, wn, Dv, ctx->d_Ad, -om2, Aw_ell, wsc, lvoff, ctx->d_trBase2);
        else femfct_enqueue_axpby(ctx, wn, Dv, ctx->d_Ad, -om2, Aw_ell, ctx->d_trBase2);
        WMassSpec ws;
        ws.alpha = 1.0; ws.gamma = dt; ws.base = ctx->d_trBase2;
        return femfct_enqueue_weighted_mass(ctx, ws, ctx->d_trBase, 1);
    };
    auto begin = [&]() {
        int rn = femfct_enqueue_axpby(ctx, wn, gam, ctx->d_M, 0.0, nullptr, ctx->d_trN);   // non_flux_mat = gamma*M (:588, shared)
        if (rn != FEMFCT_OK) return rn;
        return wsc ? (int)FEMFCT_OK : wind_ops(0);
    };
    auto step = [&](int budget, int kbudget, int reps) {
        auto key = KEY((uint64_t)12, key_bits(Aw_ell), key_bits(c_level), key_bits(u_traj), key_bits(v_traj),
                       key_bits(num_steps), key_bits(dt), key_bits(Du), key_bits(Dv), key_bits(c_b), key_bits(gam),
                       key_bits(om1), key_bits(om2), key_bits(rescaling), key_bits(batch), key_bits((int32_t)budget),
                       key_bits((int32_t)kbudget), key_bits(ctx->rel_tol), key_bits(ctx->kry_tol),
                       key_bits((int32_t)femfct_species_cheb(ctx, 12)), key_bits(wsc));
        return femfct_run_graph_reps(ctx, key, reps, +1, [&]() {
            if (wsc) wind_ops(1);       // wind.t = t_{n+1} (helpers.py:565-566)
            LoadSpec l1;  // (gamma/r*c + gamma*u_n^2*v_n)*v*dx  (helpers.py:584-585)
            l1.s1 = 1.0; l1.k1 = gam / rescaling; l1.p1 = make_ref(c_level); l1.p1_bs = n;
            l1.k2 = gam; l1.q1 = L(u_traj, 0); l1.q2 = L(u_traj, 0); l1.q3 = L(v_traj, 0);
            l1.q1_bs = l1.q2_bs = l1.q3_bs = ts;
            LoadSpec l2;  // M@v_n + dt*assemble(gamma*c_b*v*dx)  (helpers.py:594,596): level n only, so it rides in l1's launch
            l2.s0 = 1.0; l2.mx = L(v_traj, 0); l2.mx_bs = ts; l2.s1 = dt; l2.k0 = gam * c_b;
            FormGroup fg(ctx);
            fg.load(l1, ctx->d_trRhs, batch);
            fg.load(l2, ctx->d_trRhs2, batch);
            int r = fg.launch();
            if (r != FEMFCT_OK) return r;
            r = femfct_enqueue_step_ref(ctx, ctx->d_trA, ctx->d_trN, 1, make_ref(ctx->d_trRhs), n, L(u_traj, 0), ts,
                                        dt, L(u_traj, 1), ts, batch, budget);
            if (r != FEMFCT_OK) return r;
            WMassSpec ws;  // Mat_var2 = Base2 + dt*gamma*M_u2(u_{n+1})  (helpers.py:591,595)
            ws.gamma = 1.0; ws.base = ctx->d_trBase; ws.beta = dt * gam;
            ws.f1 = L(u_traj, 1); ws.f2 = L(u_traj, 1); ws.f1_bs = ws.f2_bs = ts;
            femfct_enqueue_weighted_mass(ctx, ws, ctx->d_trMat, batch);
            femfct_request_fused_end(ctx, 1, true);       // (the one-launch species solve logs and advances itself)
            r = femfct_enqueue_species_solve(ctx, 12, ctx->d_trMat, 0, ctx->d_trRhs2, L(v_traj, 0), ts, L(v_traj, 1), ts, batch, kbudget, dt * Dv);
            if (r != FEMFCT_OK) return r;
            return femfct_enqueue_step_end(ctx, 1, batch, true);
        });
    };
    ctx->kind_fullrows.insert(12);      // (diffusion / reaction terms: rows with both entries of a pair from the start)
    return femfct_run_sweep(ctx, 12, num_steps, batch, 0, true, begin, step);
}

// helpers.py:599-698.  AwT_ell = assemble_sparse(dot(wind,grad(u))*w*dx) = transpose of Aw.
// alltime != 0 (no HEAD counterpart in helpers.py; structure of the inline loop Schnak_FCT_PDECO_alltime.py:204-284
// with the HEAD operators): uhat/vhat are trajectories, p(T) = q(T) = 0, the q right-hand side gains
// dt*assemble((vhat_n - v_n)*w*dx) (:268) and the p right-hand side assemble((uhat_n - u_n)*w*dx) (:278).
int femfct_schnak_adjoint(femfct_ctx* ctx, const double* AwT_ell, const double* u_traj, const double* v_traj,
                          const double* uhat_T, const double* vhat_T, double* p_traj, double* q_traj,
                          int32_t num_steps, double dt, const double* par, int32_t alltime, int32_t batch) {
    return femfct_schnak_adjoint_tw(ctx, AwT_ell, nullptr, u_traj, v_traj, uhat_T, vhat_T, p_traj, q_traj, num_steps, dt, par,
                                    alltime, batch);
}

// with the separable time-dependent wind of femfct_schnak_forward_tw: the step that produces level n uses s(t_n)
// (helpers.py:664-679: t -= dt; wind.t = t)
int femfct_schnak_adjoint_tw(femfct_ctx* ctx, const double* AwT_ell, const double* wind_scale_host, const double* u_traj,
                             const double* v_traj, const double* uhat_T, const double* vhat_T, double* p_traj,
                             double* q_traj, int32_t num_steps, double dt, const double* par, int32_t alltime,
                             int32_t batch) {
    FEMFCT_ENTER(ctx);
    int rc = check_common(ctx, num_steps, dt, batch);
    if (rc != FEMFCT_OK) return rc;
    ARG_TRY(ctx, AwT_ell && u_traj && v_traj && uhat_T && vhat_T && p_traj && q_traj && par, "null argument");
    if ((rc = femfct_ensure_traj_ws(ctx, batch, num_steps)) != FEMFCT_OK) return rc;
    if ((rc = femfct_ensure_krylov_ws(ctx, batch)) != FEMFCT_OK) return rc;
    const double Du = par[0], Dv = par[1], gam = par[3], om1 = par[4], om2 = par[5];
    const int64_t n = ctx->n, wn = (int64_t)ctx->W * n, ts = (int64_t)(num_steps + 1) * n;
    Lv L{ctx, ctx->d_level, n};
    const double* wsc = nullptr;
    if ((rc = upload_wind_scale(ctx, wind_scale_host, num_steps, &wsc)) != FEMFCT_OK) return rc;
    auto wind_ops = [&](int lvoff) {
        for (int32_t b = 0; b < batch; ++b) {
            if (wsc) enqueue_axpby_level(ctx, wn, Du, ctx->d_Ad, -om1, AwT_ell, wsc, lvoff, ctx->d_trA + b * wn);
            else femfct_enqueue_axpby(ctx, wn, Du, ctx->d_Ad, -om1, AwT_ell, ctx->d_trA + b * wn);
        }
        if (wsc) enqueue_axpby_level(ctx, wn, Dv, ctx->d_Ad, -om2, AwT_ell, wsc, lvoff, ctx->d_trBase2);
        else femfct_enqueue_axpby(ctx, wn, Dv, ctx->d_Ad, -om2, AwT_ell, ctx->d_trBase2);
        WMassSpec ws;
        ws.alpha = 1.0; ws.gamma = dt; ws.base = ctx->d_trBase2;
        return femfct_enqueue_weighted_mass(ctx, ws, ctx->d_trBase, 1);
    };
    auto begin = [&]() {
        if (!wsc) { int rw = wind_ops(0); if (rw != FEMFCT_OK) return rw; }
        if (alltime) {
            for (int32_t b = 0; b < batch; ++b) {
                HIP_TRY(ctx, hipMemsetAsync(p_traj + b * ts + (int64_t)num_steps * n, 0, sizeof(double) * n, ctx->stream));
                HIP_TRY(ctx, hipMemsetAsync(q_traj + b * ts + (int64_t)num_steps * n, 0, sizeof(double) * n, ctx->stream));
            }
            return FEMFCT_OK;
        }
        int rt = terminal_diff(ctx, uhat_T, u_traj, p_traj, num_steps, batch);
        if (rt != FEMFCT_OK) return rt;
        return terminal_diff(ctx, vhat_T, v_traj, q_traj, num_steps, batch);
    };
    auto step = [&](int budget, int kbudget, int reps) {
        auto key = KEY((uint64_t)13, key_bits(AwT_ell), key_bits(u_traj), key_bits(v_traj), key_bits(uhat_T),
                       key_bits(vhat_T), key_bits(p_traj), key_bits(q_traj), key_bits(num_steps), key_bits(dt),
                       key_bits(Du), key_bits(Dv), key_bits(gam), key_bits(om1), key_bits(om2), key_bits(batch),
                       key_bits(alltime), key_bits((int32_t)budget), key_bits((int32_t)kbudget), key_bits(ctx->rel_tol), key_bits(ctx->kry_tol),
                       key_bits((int32_t)femfct_species_cheb(ctx, 13)), key_bits(wsc));
        return femfct_run_graph_reps(ctx, key, reps, -1, [&]() {
            if (wsc) wind_ops(0);       // level counter = n: wind.t = t_n (helpers.py:664,679)
            // q first (helpers.py:683-686): Mat_q = M + dt*(Dv*Ad - omega2*A' + gamma*M_u2(u_n))
            WMassSpec wq;
            wq.gamma = 1.0; wq.base = ctx->d_trBase; wq.beta = dt * gam;
            wq.f1 = L(u_traj, 0); wq.f2 = L(u_traj, 0); wq.f1_bs = wq.f2_bs = ts;
            LoadSpec lq;  // M@q_{n+1} + dt*assemble(gamma*p_{n+1}*u_n^2*w*dx)
            lq.s0 = 1.0; lq.mx = L(q_traj, 1); lq.mx_bs = ts; lq.s1 = dt; lq.k2 = gam;
            lq.q1 = L(p_traj, 1); lq.q2 = L(u_traj, 0); lq.q3 = L(u_traj, 0); lq.q1_bs = lq.q2_bs = lq.q3_bs = ts;
            if (alltime) { lq.s3 = dt; lq.ea = L(vhat_T, 0); lq.eb = L(v_traj, 0); lq.ea_bs = lq.eb_bs = ts; }
            // N = gamma*M - 2*gamma*M_uv of the p step below (helpers.py:690-692) needs the states only: one launch for the three
            WMassSpec wn_;
            wn_.alpha = gam; wn_.beta = -2.0 * gam; wn_.f1 = L(u_traj, 0); wn_.f2 = L(v_traj, 0); wn_.f1_bs = wn_.f2_bs = ts;
            FormGroup fg(ctx);
            fg.weighted_mass(wq, ctx->d_trMat, batch);
            fg.load(lq, ctx->d_trRhs2, batch);
            fg.weighted_mass(wn_, ctx->d_trN, batch);
            int r = fg.launch();
            if (r != FEMFCT_OK) return r;
            r = femfct_enqueue_species_solve(ctx, 13, ctx->d_trMat, 0, ctx->d_trRhs2, L(q_traj, 1), ts, L(q_traj, 0), ts, batch, kbudget, dt * Dv);
            if (r != FEMFCT_OK) return r;
            // then p by FCT (helpers.py:690-697): N = gamma*M - 2*gamma*M_uv (above), rhs = -2*gamma*u_n*v_n*q_n
            LoadSpec lp;
            lp.s1 = 1.0; lp.k2 = -2.0 * gam; lp.q1 = L(u_traj, 0); lp.q2 = L(v_traj, 0); lp.q3 = L(q_traj, 0);
            lp.q1_bs = lp.q2_bs = lp.q3_bs = ts;
            if (alltime) { lp.s3 = 1.0; lp.ea = L(uhat_T, 0); lp.eb = L(u_traj, 0); lp.ea_bs = lp.eb_bs = ts; }
            femfct_enqueue_load(ctx, lp, ctx->d_trRhs, batch);
            femfct_request_fused_end(ctx, -1, true);
            r = femfct_enqueue_step_ref(ctx, ctx->d_trA, ctx->d_trN, 0, make_ref(ctx->d_trRhs), n, L(p_traj, 1), ts, dt,
                                        L(p_traj, 0), ts, batch, budget);
            if (r != FEMFCT_OK) return r;
            return femfct_enqueue_step_end(ctx, -1, batch, true);
        });
    };
    ctx->kind_fullrows.insert(13);      // (diffusion / reaction terms: rows with both entries of a pair from the start)
    return femfct_run_sweep(ctx, 13, num_steps, batch, num_steps - 1, true, begin, step);
}

// ------------------------------------------------------------------ chemotaxis
// helpers.py:1250-1385.  par = {delta, Dm, Df, chi, eta}
int femfct_chtxs_forward(femfct_ctx* ctx, const double* c_level, double* u_traj, double* v_traj, int32_t num_steps,
                         double dt, const double* par, double rescaling, int32_t batch) {
    FEMFCT_ENTER(ctx);
    int rc = check_common(ctx, num_steps, dt, batch);
    if (rc != FEMFCT_OK) return rc;
    ARG_TRY(ctx, c_level && u_traj && v_traj && par && rescaling != 0.0, "bad argument");
    if ((rc = femfct_ensure_traj_ws(ctx, batch, num_steps)) != FEMFCT_OK) return rc;
    if ((rc = femfct_ensure_krylov_ws(ctx, batch)) != FEMFCT_OK) return rc;
    const double delta = par[0], Dm = par[1], Df = par[2], chi = par[3], eta = par[4];
    const int64_t n = ctx->n, ts = (int64_t)(num_steps + 1) * n;
    Lv L{ctx, ctx->d_level, n};
    auto begin = [&]() {
        WMassSpec ws;  // Mat_var2 = M + dt*(Df*Ad + delta*M)  (helpers.py:1308), constant SPD, shared
        ws.alpha = 1.0 + dt * delta; ws.gamma = dt * Df; ws.base = ctx->d_Ad;
        return femfct_enqueue_weighted_mass(ctx, ws, ctx->d_trBase, 1);
    };
    auto step = [&](int budget, int kbudget, int reps) {
        auto key = KEY((uint64_t)14, key_bits(c_level), key_bits(u_traj), key_bits(v_traj), key_bits(num_steps),
                       key_bits(dt), key_bits(delta), key_bits(Dm), key_bits(Df), key_bits(chi), key_bits(eta),
                       key_bits(rescaling), key_bits(batch), key_bits((int32_t)budget), key_bits((int32_t)kbudget),
                       key_bits(ctx->rel_tol), key_bits(ctx->kry_tol),
                       key_bits((int32_t)femfct_species_cheb(ctx, 14)));
        return femfct_run_graph_reps(ctx, key, reps, +1, [&]() {
            LoadSpec l2;  // assemble(v_n*v*dx + dt*c*u_n/r*v*dx)  (helpers.py:1339-1340)
            l2.s0 = 1.0; l2.mx = L(v_traj, 0); l2.mx_bs = ts; l2.s1 = dt / rescaling; l2.k2 = 1.0;
            l2.q1 = make_ref(c_level); l2.q1_bs = n; l2.q2 = L(u_traj, 0); l2.q2_bs = ts;
            femfct_enqueue_load(ctx, l2, ctx->d_trRhs2, batch);
            int r = femfct_enqueue_species_solve(ctx, 14, ctx->d_trBase, 1, ctx->d_trRhs2, L(v_traj, 0), ts, L(v_traj, 1), ts, batch, kbudget, dt * Df);
            if (r != FEMFCT_OK) return r;
            // A_var1 = Dm*Ad - chi*Aa(u_n, v_{n+1})  (helpers.py:1350-1352)
            femfct_enqueue_chtxs_matrix(ctx, 0, L(u_traj, 0), ts, L(v_traj, 1), ts, Dm, chi, eta, ctx->d_trA, batch);
            femfct_request_fused_end(ctx, 1, true);
            r = femfct_enqueue_step_ref(ctx, ctx->d_trA, nullptr, 0, make_ref(nullptr), 0, L(u_traj, 0), ts, dt,
                                        L(u_traj, 1), ts, batch, budget);
            if (r != FEMFCT_OK) return r;
            return femfct_enqueue_step_end(ctx, 1, batch, true);
        });
    };
    ctx->kind_fullrows.insert(14);      // (diffusion / reaction terms: rows with both entries of a pair from the start)
    return femfct_run_sweep(ctx, 14, num_steps, batch, 0, true, begin, step);
}

// helpers.py:1387-1581.  alltime = 0: optim == "finaltime" (uhat/vhat: n doubles per member, terminal
// conditions set); alltime = 1: optim == "alltime" (uhat/vhat trajectories, level num_steps of p/q left
// as passed, raw nodal misfits added to the load vectors: helpers.py:1506-1507,1533-1534).
int femfct_chtxs_adjoint(femfct_ctx* ctx, const double* u_traj, const double* v_traj, const double* uhat,
                         const double* vhat, double* p_traj, double* q_traj, const double* c_traj, int32_t num_steps,
                         double dt, const double* par, double rescaling, int32_t alltime, int32_t batch) {
    FEMFCT_ENTER(ctx);
    int rc = check_common(ctx, num_steps, dt, batch);
    if (rc != FEMFCT_OK) return rc;
    ARG_TRY(ctx, u_traj && v_traj && uhat && vhat && p_traj && q_traj && c_traj && par && rescaling != 0.0, "bad argument");
    if ((rc = femfct_ensure_traj_ws(ctx, batch, num_steps)) != FEMFCT_OK) return rc;
    if ((rc = femfct_ensure_krylov_ws(ctx, batch)) != FEMFCT_OK) return rc;
    const double delta = par[0], Dm = par[1], Df = par[2], chi = par[3], eta = par[4];
    const int64_t n = ctx->n, ts = (int64_t)(num_steps + 1) * n;
    Lv L{ctx, ctx->d_level, n};
    auto begin = [&]() {
        WMassSpec ws;  // Mat_q = M + dt*(Df*Ad + delta*M)  (helpers.py:1536)
        ws.alpha = 1.0 + dt * delta; ws.gamma = dt * Df; ws.base = ctx->d_Ad;
        femfct_enqueue_weighted_mass(ctx, ws, ctx->d_trBase, 1);
        if (!alltime) {
            terminal_diff(ctx, uhat, u_traj, p_traj, num_steps, batch);
            terminal_diff(ctx, vhat, v_traj, q_traj, num_steps, batch);
        }
        return FEMFCT_OK;
    };
    auto step = [&](int budget, int kbudget, int reps) {
        auto key = KEY((uint64_t)15, key_bits(u_traj), key_bits(v_traj), key_bits(uhat), key_bits(vhat), key_bits(p_traj),
                       key_bits(q_traj), key_bits(c_traj), key_bits(num_steps), key_bits(dt), key_bits(delta),
                       key_bits(Dm), key_bits(Df), key_bits(chi), key_bits(eta), key_bits(rescaling), key_bits(alltime),
                       key_bits(batch), key_bits((int32_t)budget), key_bits((int32_t)kbudget), key_bits(ctx->rel_tol),
                       key_bits(ctx->kry_tol), key_bits((int32_t)femfct_species_cheb(ctx, 15)));
        return femfct_run_graph_reps(ctx, key, reps, -1, [&]() {
            // Mat_p = Dm*Ad - chi*Aa'(u_n, v_n)  (helpers.py:1499-1503)
            LoadSpec lp;  // assemble(c_n*q_{n+1}/r*w*dx) [+ uhat_n - u_n]  (helpers.py:1505-1507)
            lp.s1 = 1.0 / rescaling; lp.k2 = 1.0; lp.q1 = L(c_traj, 0); lp.q2 = L(q_traj, 1); lp.q1_bs = lp.q2_bs = ts;
            if (alltime) { lp.s2 = 1.0; lp.da = L(uhat, 0); lp.db = L(u_traj, 0); lp.da_bs = lp.db_bs = ts; }
            FormGroup fg(ctx);
            fg.chtxs_matrix(1, L(u_traj, 0), ts, L(v_traj, 0), ts, Dm, chi, eta, ctx->d_trA, batch);
            fg.load(lp, ctx->d_trRhs, batch);
            int r = fg.launch();
            if (r != FEMFCT_OK) return r;
            r = femfct_enqueue_step_ref(ctx, ctx->d_trA, nullptr, 0, make_ref(ctx->d_trRhs), n, L(p_traj, 1), ts, dt,
                                            L(p_traj, 0), ts, batch, budget);
            if (r != FEMFCT_OK) return r;
            // rhs_q = assemble(chi*u_n*exp(-eta*u_n)*dot(grad(p_n),grad(w))*dx) [+ vhat_n - v_n]  (helpers.py:1531-1534)
            VecRef none = make_ref(nullptr);
            // ... and M@q_{n+1} + dt*rhs_q (helpers.py:1538) in the same pass
            femfct_enqueue_chtxs_rhs_q(ctx, L(u_traj, 0), ts, L(p_traj, 0), ts, chi, eta, alltime ? L(vhat, 0) : none, ts,
                                       alltime ? L(v_traj, 0) : none, ts, ctx->d_trRhs2, batch, L(q_traj, 1), ts, 1.0, dt);
            femfct_request_fused_end(ctx, -1, true);
            r = femfct_enqueue_species_solve(ctx, 15, ctx->d_trBase, 1, ctx->d_trRhs2, L(q_traj, 1), ts, L(q_traj, 0), ts, batch, kbudget, dt * Df);
            if (r != FEMFCT_OK) return r;
            return femfct_enqueue_step_end(ctx, -1, batch, true);
        });
    };
    ctx->kind_fullrows.insert(15);      // (diffusion / reaction terms: rows with both entries of a pair from the start)
    return femfct_run_sweep(ctx, 15, num_steps, batch, num_steps - 1, true, begin, step);
}

// BiCGStab diagnostics of the most recent sweep that used it: info_host[step*batch + b]
int femfct_traj_krylov_info(femfct_ctx* ctx, femfct_step_info* info_host, int32_t num_steps, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && info_host, "null argument");
    ARG_TRY(ctx, num_steps == ctx->log_steps && batch == ctx->log_batch &&
                     ctx->h_klog.size() == sizeof(KrylovCtl) * (size_t)num_steps * batch, "no matching Krylov log");
    const KrylovCtl* kl = (const KrylovCtl*)ctx->h_klog.data();
    for (size_t k = 0; k < (size_t)num_steps * batch; ++k) {
        info_host[k].flags = kl[k].flags;
        info_host[k].solver_iters = kl[k].iters;
        info_host[k].solver_resid = kl[k].resid;
        info_host[k].min_rowsum = 0.0;
    }
    return FEMFCT_OK;
}

}  // extern "C"
