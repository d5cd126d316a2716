// Structured UnitSquare/RectangleMesh P1 pattern, DoF numbering and constant operators.
//   RectangleMesh + FunctionSpace(mesh,'CG',1) + vertex_to_dof_map
//        /root/reference/advection_solidbody_FCT_PDECO_finaltime.py:63-64,101
//   find_node_neighbours          /root/reference/helpers.py:271-307
//   M, row_lump(M), Ad            /root/reference/helpers.py:553-555
#include "femfct_internal.h"
#include "device_utils.h"
#include "stencil.h"

#include <algorithm>
#include <stdlib.h>

int femfct_install_pattern(femfct_ctx* ctx, int32_t n, int32_t W, const std::vector<int32_t>& cols,
                           const std::vector<uint8_t>& tslot);
int femfct_enqueue_mesh_constants(femfct_ctx* ctx);

// FEniCS CG1 dof of vertex (ix,iy): rank of (ix-iy, iy) in lexicographic order
// (pinned by tests/golden/chtxs_fenics_traj.npz).
static inline int64_t fenics_dof(int64_t ix, int64_t iy, int64_t N) {
    int64_t d = ix - iy;  // -(N-1) .. N-1
    // offset(d) = sum_{d' < d} (N - |d'|)
    int64_t off;
    if (d <= 0) {
        int64_t k = d + (N - 1);        // number of diagonals before d: lengths 1..k
        off = k * (k + 1) / 2;
        return off + (iy - (-d));       // iy runs from -d
    }
    int64_t k = N - 1;                  // diagonals -(N-1)..-1 : 1..N-1
    off = k * (k + 1) / 2 + N;          // plus diagonal 0
    // diagonals 1..d-1 have lengths N-1 .. N-d+1
    off += (d - 1) * N - (d - 1) * d / 2;
    return off + iy;
}

int femfct_mesh_release(femfct_ctx* ctx) {
    ctx->structured = false;
    return FEMFCT_OK;
}

extern "C" int femfct_set_mesh_square(femfct_ctx* ctx, double a1, double a2, int32_t n_cells, int32_t order) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx, "null ctx");
    ARG_TRY(ctx, n_cells >= 1 && a2 > a1, "need n_cells >= 1 and a2 > a1");
    ARG_TRY(ctx, order == FEMFCT_ORDER_VERTEX || order == FEMFCT_ORDER_FENICS, "unknown dof order");
    const int64_t N = (int64_t)n_cells + 1;
    ARG_TRY(ctx, N * N * STENCIL_W < 2147483647LL, "mesh too large for int32 ELL indexing");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    femfct_release_pattern(ctx);
    const int32_t n = (int32_t)(N * N);
    std::vector<int32_t> cols((size_t)STENCIL_W * n);
    std::vector<uint8_t> tslot((size_t)STENCIL_W * n);
    std::vector<int32_t> d2v;
    if (order == FEMFCT_ORDER_FENICS) d2v.resize(n);
    for (int64_t iy = 0; iy < N; ++iy)
        for (int64_t ix = 0; ix < N; ++ix) {
            int64_t i = (order == FEMFCT_ORDER_FENICS) ? fenics_dof(ix, iy, N) : iy * N + ix;
            if (order == FEMFCT_ORDER_FENICS) d2v[i] = (int32_t)(iy * N + ix);
            for (int s = 0; s < STENCIL_W; ++s) {
                int64_t jx = ix + stencil_dx(s), jy = iy + stencil_dy(s);
                bool in = jx >= 0 && jx < N && jy >= 0 && jy < N;
                int64_t j = !in ? i : ((order == FEMFCT_ORDER_FENICS) ? fenics_dof(jx, jy, N) : jy * N + jx);
                cols[(size_t)s * n + i] = (int32_t)j;
                tslot[(size_t)s * n + i] = (uint8_t)(in ? stencil_opp(s) : s);
            }
        }
    int rc = femfct_install_pattern(ctx, n, STENCIL_W, cols, tslot);
    if (rc != FEMFCT_OK) return rc;
    ctx->structured = true;
    ctx->mass_is_mesh = true;       // d_M is the P1 mass matrix of this mesh (femfct_set_mass would clear this)
    ctx->implicit_cols = (order == FEMFCT_ORDER_VERTEX);
    if (const char* e = getenv("FEMFCT_IMPLICIT")) ctx->implicit_cols = ctx->implicit_cols && atoi(e) != 0;
    ctx->a1 = a1; ctx->a2 = a2; ctx->n_cells = n_cells; ctx->N = (int32_t)N; ctx->order = order;
    ctx->h = (a2 - a1) / n_cells;
    if (order == FEMFCT_ORDER_FENICS) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_d2v, sizeof(int32_t) * n));
        HIP_TRY(ctx, hipMemcpy(ctx->d_d2v, d2v.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_Ad, sizeof(double) * STENCIL_W * n));
    rc = femfct_enqueue_mesh_constants(ctx);
    if (rc != FEMFCT_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_mass = true;
    return FEMFCT_OK;
}

// CSR pattern (sorted columns) of the structured mesh, built lazily for csr<->ell conversion.
int femfct_build_structured_csr(femfct_ctx* ctx) {
    const int32_t n = ctx->n, W = ctx->W;
    ctx->h_indptr.assign((size_t)n + 1, 0);
    ctx->h_indices.clear();
    ctx->h_indices.reserve((size_t)n * W);
    int32_t tmp[STENCIL_W];
    for (int32_t i = 0; i < n; ++i) {
        int cnt = 0;
        tmp[cnt++] = i;
        for (int s = 1; s < W; ++s) {
            int32_t j = ctx->h_cols[(size_t)s * n + i];
            if (j != i) tmp[cnt++] = j;
        }
        std::sort(tmp, tmp + cnt);
        for (int k = 0; k < cnt; ++k) ctx->h_indices.push_back(tmp[k]);
        ctx->h_indptr[i + 1] = (int32_t)ctx->h_indices.size();
    }
    return FEMFCT_OK;
}
