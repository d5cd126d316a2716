// Drives libfemfct's HOST code through the C ABI under AddressSanitizer, on top of the fake HIP runtime (fake_hip.cpp):
// context life cycle, mesh registration (both DoF orders, small and patch-regime sizes), every trajectory sweep with
// growing / shrinking batch and step counts (the log buffers, workspaces and graph keys are resized on the way), the
// info calls with matching and mismatching sizes, error paths with null arguments, knob setters.  Kernels do not run:
// this checks the host side only -- sizes, lifetimes, indices.
#include "../../include/femfct.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define OK(x) do { int rc_ = (x); if (rc_ != FEMFCT_OK) { printf("line %d: rc %d (%s)\n", __LINE__, rc_, femfct_last_error(ctx)); fails++; } } while (0)
#define BAD(x) do { int rc_ = (x); if (rc_ == FEMFCT_OK) { printf("line %d: expected an error\n", __LINE__); fails++; } } while (0)

static double* dmalloc(femfct_ctx* ctx, size_t count) {
    void* p = nullptr;
    if (femfct_malloc(ctx, &p, count * sizeof(double)) != FEMFCT_OK) abort();
    femfct_memset0(ctx, p, count * sizeof(double));
    return (double*)p;
}

int main() {
    int fails = 0;
    for (int order = 0; order < 2; ++order)
        for (int nc : {8, 40, 330}) {                  // 330: the bandwidth regime's launch planning (walkers, pair launch)
            femfct_ctx* ctx = nullptr;
            if (femfct_create(&ctx, 0) != FEMFCT_OK) { printf("create failed\n"); return 1; }
            OK(femfct_set_mesh_square(ctx, -1.0, 1.0, nc, order));
            const int n = (nc + 1) * (nc + 1), W = 7;
            int active = -1; int32_t li[4];
            OK(femfct_graph_replay_active(ctx, &active));
            OK(femfct_launch_info(ctx, li));
            double* Arot = dmalloc(ctx, (size_t)W * n);
            const double par6[6] = {0.01, 8.6676, 0.9, 230.82, 100.0, 0.6}, par5[5] = {100, 0.05, 0.05, 0.25, 0.5};
            for (int pass = 0; pass < 3; ++pass) {
                const int Nt = pass == 1 ? 7 : 3, B = pass == 1 ? 3 : (pass == 2 ? 1 : 2);     // grow, then shrink
                const size_t tl = (size_t)(Nt + 1) * n;
                double *c = dmalloc(ctx, tl * B), *u = dmalloc(ctx, tl * B), *v = dmalloc(ctx, tl * B), *p = dmalloc(ctx, tl * B),
                       *q = dmalloc(ctx, tl * B), *uh = dmalloc(ctx, tl * B), *vh = dmalloc(ctx, tl * B), *cl = dmalloc(ctx, (size_t)n * B);
                std::vector<femfct_step_info> info((size_t)Nt * B), small(1);
                OK(femfct_solidbody_forward(ctx, Arot, c, 0, u, Nt, 1e-3, 0.0, 1.0, 1.0, 1.0, B));
                OK(femfct_traj_info(ctx, info.data(), Nt, B));
                BAD(femfct_traj_info(ctx, small.data(), Nt + 1, B));             // no matching log: must refuse, not copy
                OK(femfct_solidbody_adjoint(ctx, Arot, c, 0, u, uh, p, Nt, 1e-3, 0.0, 1.0, 1.0, 1.0, 0, B));
                OK(femfct_solidbody_adjoint(ctx, Arot, c, 1, u, uh, p, Nt, 1e-3, 1e-3, 0.0, 1.0, 1.0, 1, B));
                OK(femfct_solidbody_forward_src(ctx, Arot, c, 1, uh, u, Nt, 1e-3, 1e-3, 1.0, 0.0, 0.0, B));
                if (nc <= 40) {
                    std::vector<double> ws((size_t)Nt + 1, 0.5);
                    OK(femfct_nonlinear_forward(ctx, Arot, cl, u, Nt, 1e-3, 1e-3, B));
                    OK(femfct_nonlinear_adjoint(ctx, Arot, u, uh, p, Nt, 1e-3, 1e-3, B));
                    OK(femfct_schnak_forward(ctx, Arot, cl, u, v, Nt, 5e-4, par6, 1.0, B));
                    OK(femfct_traj_krylov_info(ctx, info.data(), Nt, B));
                    OK(femfct_schnak_adjoint(ctx, Arot, u, v, uh, vh, p, q, Nt, 5e-4, par6, 0, B));
                    OK(femfct_schnak_adjoint(ctx, Arot, u, v, uh, vh, p, q, Nt, 5e-4, par6, 1, B));
                    OK(femfct_schnak_forward_tw(ctx, Arot, ws.data(), cl, u, v, Nt, 5e-4, par6, 1.0, B));
                    OK(femfct_schnak_adjoint_tw(ctx, Arot, ws.data(), u, v, uh, vh, p, q, Nt, 5e-4, par6, 1, B));
                    OK(femfct_chtxs_forward(ctx, cl, u, v, Nt, 5e-4, par5, 0.1, B));
                    OK(femfct_chtxs_adjoint(ctx, u, v, uh, vh, p, q, c, Nt, 5e-4, par5, 0.1, 1, B));
                    OK(femfct_chtxs_adjoint(ctx, u, v, uh, vh, p, q, c, Nt, 5e-4, par5, 0.1, 0, B));
                    BAD(femfct_schnak_adjoint(ctx, Arot, u, v, nullptr, vh, p, q, Nt, 5e-4, par6, 0, B));
                    BAD(femfct_chtxs_forward(ctx, cl, u, v, 0, 5e-4, par5, 0.1, B));
                }
                double J[4] = {0, 0, 0, 0};
                OK(femfct_cost_functional(ctx, u, uh, c, 0, Nt, 1e-3, 0.1, 1, nullptr, nullptr, J, B));
                OK(femfct_l2_norm_sq_Q(ctx, u, p, Nt, 1e-3, J, B));
                OK(femfct_fct_step(ctx, Arot, nullptr, 0, nullptr, u, 1e-3, p, B));
                OK(femfct_last_step_info(ctx, info.data(), B));
                for (double* a : {c, u, v, p, q, uh, vh, cl}) OK(femfct_free(ctx, a));
            }
            OK(femfct_set_graphs(ctx, 0));
            OK(femfct_set_fusion(ctx, 0, 0));
            OK(femfct_set_solver(ctx, FEMFCT_SOLVER_BICGSTAB, 1e-12, 200));
            {
                double *c = dmalloc(ctx, 4 * (size_t)n), *u = dmalloc(ctx, 4 * (size_t)n);
                OK(femfct_solidbody_forward(ctx, Arot, c, 0, u, 3, 1e-3, 0.0, 1.0, 1.0, 1.0, 1));
                OK(femfct_free(ctx, c)); OK(femfct_free(ctx, u));
            }
            BAD(femfct_set_solver(ctx, 7, 1e-12, 200));
            BAD(femfct_solidbody_forward(ctx, Arot, nullptr, 0, nullptr, 3, 1e-3, 0.0, 1.0, 1.0, 1.0, 1));
            OK(femfct_free(ctx, Arot));
            OK(femfct_destroy(ctx));
        }
    printf("host_asan_driver: %d unexpected return codes\n", fails);
    return fails ? 1 : 0;
}
