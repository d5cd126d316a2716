// Argument structs of the quadrature-form kernels (kernels_forms.hip) and the Krylov solver.
#pragma once

#include "device_utils.h"

struct femfct_ctx;

struct MeshArgs {
    int n, N, nc;
    double h;
    const int32_t* d2v;
    const int32_t* cols;
    const double* M;
    const double* Ad;
};

// out = alpha*M + gamma*base + beta * int f1 f2 phi_i phi_j
struct WMassSpec {
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
    const double* base = nullptr;   // constant ELL matrix (shared by the batch)
    VecRef f1{nullptr, nullptr, 0, 0}, f2{nullptr, nullptr, 0, 0};
    int64_t f1_bs = 0, f2_bs = 0;
};

// out_i = s0*(M mx)_i + s1 * int (k0 + k1*p1 + k2*q1*q2*q3) phi_i + s2*(da_i - db_i) + s3*(M (ea - eb))_i
struct LoadSpec {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, k0 = 0.0, k1 = 0.0, k2 = 0.0;
    VecRef mx{nullptr, nullptr, 0, 0}, p1{nullptr, nullptr, 0, 0}, q1{nullptr, nullptr, 0, 0},
        q2{nullptr, nullptr, 0, 0}, q3{nullptr, nullptr, 0, 0}, da{nullptr, nullptr, 0, 0},
        db{nullptr, nullptr, 0, 0}, ea{nullptr, nullptr, 0, 0}, eb{nullptr, nullptr, 0, 0};
    int64_t mx_bs = 0, p1_bs = 0, q1_bs = 0, q2_bs = 0, q3_bs = 0, da_bs = 0, db_bs = 0, ea_bs = 0, eb_bs = 0;
};

// up to three independent forms of a time step in one launch (kernels_forms.hip: k_forms2 / k_forms3)
enum { FORM_NONE = 0, FORM_WMASS, FORM_LOAD, FORM_CHTXS_MAT0, FORM_CHTXS_MAT1 };
struct ChtxsMatSpec {
    VecRef u{nullptr, nullptr, 0, 0}, v{nullptr, nullptr, 0, 0};
    int64_t u_bs = 0, v_bs = 0;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;      // Dm, chi, eta
};
struct FormJob {
    int type = FORM_NONE;
    int batch = 0;
    double* out = nullptr;
    WMassSpec w;
    LoadSpec l;
    ChtxsMatSpec c;
};
// Collects forms and launches them together; forms of one group must not depend on each other's output.
// (FEMFCT_FORM_GROUPS=0: every form in its own launch, in the order given -- the same bits.)
struct FormGroup {
    femfct_ctx* ctx;
    int n = 0;
    FormJob jobs[3];
    explicit FormGroup(femfct_ctx* c) : ctx(c) {}
    void weighted_mass(const WMassSpec& sp, double* out, int32_t batch);
    void load(const LoadSpec& sp, double* out, int32_t batch);
    void chtxs_matrix(int adjoint, VecRef u, int64_t u_bs, VecRef v, int64_t v_bs, double Dm, double chi, double eta,
                      double* out, int32_t batch);
    int launch();
};

MeshArgs femfct_mesh_args(const femfct_ctx* ctx);
int femfct_enqueue_weighted_mass(femfct_ctx* ctx, const WMassSpec& sp, double* out, int32_t batch);
int femfct_enqueue_load(femfct_ctx* ctx, const LoadSpec& sp, double* out, int32_t batch);
int femfct_enqueue_chtxs_matrix(femfct_ctx* ctx, int adjoint, VecRef u, int64_t u_bs, VecRef v, int64_t v_bs,
                                double Dm, double chi, double eta, double* out, int32_t batch);
// mx.base != null: out = s0 * M mx + s2 * rhs_q in the same pass (the species right-hand side, helpers.py:1538)
int femfct_enqueue_chtxs_rhs_q(femfct_ctx* ctx, VecRef u, int64_t u_bs, VecRef p, int64_t p_bs, double chi, double eta,
                               VecRef da, int64_t da_bs, VecRef db, int64_t db_bs, double* out, int32_t batch,
                               VecRef mx = VecRef{nullptr, nullptr, 0, 0}, int64_t mx_bs = 0, double s0 = 0.0, double s2 = 0.0);

// Jacobi-preconditioned BiCGStab for the non-FCT implicit solves (kernels_krylov.hip)
struct KrylovCtl {
    int32_t flags;       // 1: breakdown, FEMFCT_FLAG_SOLVER_BUDGET: budget exhausted
    int32_t iters;
    int32_t done;
    int32_t pad;
    double resid;        // ||r||_inf / ||b||_inf
    double bnorm;
    double rho2[2];      // rho of iteration it at [it & 1] (two slots: no same-kernel read/write)
    double alpha, omega;
};
// lowsolve = true: the low-order solve of the FCT step (own control blocks, tolerance rel_tol; the
// result is reported through StepCtl by femfct_enqueue_kry_to_stepctl)
int femfct_enqueue_bicgstab(femfct_ctx* ctx, const double* mat, int32_t mat_shared, const double* b, VecRef x0,
                            int64_t x0_bs, VecRef x_out, int64_t out_bs, int32_t batch, int32_t budget,
                            bool lowsolve = false);
// Chebyshev variant (structured vertex-order mesh) and the per-sweep-kind choice between the two
bool femfct_species_cheb(const femfct_ctx* ctx, int kind);
bool femfct_mesh_solve_fits(const femfct_ctx* ctx);
int femfct_enqueue_cheb_solve(femfct_ctx* ctx, const double* mat, int32_t mat_shared, const double* b, VecRef x0,
                              int64_t x0_bs, VecRef x_out, int64_t out_bs, int32_t batch, int32_t budget, double tau);
int femfct_enqueue_species_solve(femfct_ctx* ctx, int kind, const double* mat, int32_t mat_shared, const double* b,
                                 VecRef x0, int64_t x0_bs, VecRef x_out, int64_t out_bs, int32_t batch, int32_t budget,
                                 double tau);   // tau: dt * diffusion coefficient of the system matrix (< 0: unknown)
int femfct_enqueue_kry_to_stepctl(femfct_ctx* ctx, int g_build, int32_t batch);
