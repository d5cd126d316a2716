// libfemfct context, memory, sparsity pattern and the generic step entry points.
#include "femfct_internal.h"
#include "device_utils.h"

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <tuple>

int femfct_enqueue_artdiff(femfct_ctx* ctx, const double* K, double* D, int32_t batch);
int femfct_enqueue_spmv(femfct_ctx* ctx, const double* A, const double* x, double alpha, double beta, double* y,
                        int32_t batch);
int femfct_enqueue_cheb(femfct_ctx* ctx, const double* b, double* y_out, int iters, double lmin, double lmax,
                        int32_t batch, bool first_done_in_y1, const double* mdv = nullptr);
int femfct_mesh_release(femfct_ctx* ctx);
int femfct_build_structured_csr(femfct_ctx* ctx);  // mesh.hip

int femfct_fail(femfct_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

namespace {

__global__ void k_scatter_csr_to_ell(int64_t nnz, const int32_t* __restrict__ map, const double* __restrict__ v,
                                     double* __restrict__ ell) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) ell[map[k]] = v[k];
}

__global__ void k_gather_ell_to_csr(int64_t nnz, const int32_t* __restrict__ map, const double* __restrict__ ell,
                                    double* __restrict__ v) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) v[k] = ell[map[k]];
}

// bits set in the zero mask of the low-order operator (one workgroup; diagnostic, not on the hot path)
__global__ void k_mask_popcount(int64_t nbytes, const unsigned long long* __restrict__ mask, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long s[256];
    unsigned long long c = 0;
    const int64_t words = nbytes >> 3;
    for (int64_t k = threadIdx.x; k < words; k += blockDim.x) c += (unsigned long long)__popcll(mask[k]);
    if (threadIdx.x == 0)
        for (int64_t k = words * 8; k < nbytes; ++k) c += (unsigned long long)__popc(reinterpret_cast<const uint8_t*>(mask)[k]);
    s[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s[0];
}

template <class T>
int dev_alloc(femfct_ctx* ctx, T** p, size_t count) {
    if (*p) { hipFree(*p); *p = nullptr; }
    if (count == 0) return FEMFCT_OK;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        return femfct_fail(ctx, FEMFCT_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T),
                           hipGetErrorString(e));
    }
    return FEMFCT_OK;
}

template <class T>
void dev_free(T** p) {
    if (*p) hipFree(*p);
    *p = nullptr;
}

}  // namespace

void femfct_release_pattern(femfct_ctx* ctx) {
    femfct_drop_graphs(ctx);
    dev_free(&ctx->d_cols); dev_free(&ctx->d_tslot); dev_free(&ctx->d_csr2ell); dev_free(&ctx->d_d2v);
    dev_free(&ctx->d_M); dev_free(&ctx->d_Ad); dev_free(&ctx->d_ml);
    dev_free(&ctx->d_L); dev_free(&ctx->d_D); dev_free(&ctx->d_F);
    dev_free(&ctx->d_b); dev_free(&ctx->d_xa); dev_free(&ctx->d_xb);
    dev_free(&ctx->d_du); dev_free(&ctx->d_y0); dev_free(&ctx->d_y1); dev_free(&ctx->d_y2); dev_free(&ctx->d_rdu);
    dev_free(&ctx->d_rp); dev_free(&ctx->d_rm); dev_free(&ctx->d_part); dev_free(&ctx->d_ctl);
    dev_free(&ctx->d_bigpart); ctx->bigpart_count = 0;
    dev_free(&ctx->d_Lmask);
    dev_free(&ctx->d_partk);
    dev_free(&ctx->d_hA); dev_free(&ctx->d_hN); dev_free(&ctx->d_hrhs); dev_free(&ctx->d_hu);
    dev_free(&ctx->d_hout); dev_free(&ctx->d_hcsr);
    femfct_mesh_release(ctx);
    ctx->h_indptr.clear(); ctx->h_indices.clear(); ctx->h_csr2ell.clear(); ctx->h_cols.clear();
    dev_free(&ctx->d_trAall); ctx->trAall_count = 0;
    dev_free(&ctx->d_trA); dev_free(&ctx->d_trN); dev_free(&ctx->d_trRhs); dev_free(&ctx->d_level); dev_free(&ctx->d_log);
    dev_free(&ctx->d_ticket);
    ctx->tr_batch = 0; ctx->tr_steps = 0;
    dev_free(&ctx->d_scratch); ctx->scratch_count = 0;
    dev_free(&ctx->d_kry); dev_free(&ctx->d_kry_part); dev_free(&ctx->d_chs_om); dev_free(&ctx->d_chs_scale);
    ctx->chs_om_cap = 0;
    if (ctx->d_kry_ctl) { hipFree(ctx->d_kry_ctl); ctx->d_kry_ctl = nullptr; }
    if (ctx->d_kry_ctl2) { hipFree(ctx->d_kry_ctl2); ctx->d_kry_ctl2 = nullptr; }
    if (ctx->d_klog) { hipFree(ctx->d_klog); ctx->d_klog = nullptr; }
    ctx->kry_batch = 0;
    dev_free(&ctx->d_wscale); ctx->wscale_count = 0;
    dev_free(&ctx->d_trMat); dev_free(&ctx->d_trBase); dev_free(&ctx->d_trBase2); dev_free(&ctx->d_trRhs2); dev_free(&ctx->d_trTmp);
    ctx->implicit_cols = false;
    ctx->n = 0; ctx->W = 0; ctx->nnz_csr = 0; ctx->ws_batch = 0; ctx->have_mass = false; ctx->structured = false; ctx->mass_is_mesh = false;
}

void femfct_prof_begin(femfct_ctx* ctx, int cls) {
    if (!ctx->prof_on) return;
    femfct_ctx::ProfRec r;
    r.cls = cls;
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    hipEventRecord(r.a, ctx->stream);
    ctx->prof.push_back(r);
}

void femfct_prof_end(femfct_ctx* ctx) {
    if (!ctx->prof_on || ctx->prof.empty()) return;
    hipEventRecord(ctx->prof.back().b, ctx->stream);
}

void femfct_drop_graphs(femfct_ctx* ctx) {
    for (auto& kv : ctx->graphs) hipGraphExecDestroy(kv.second);
    ctx->graphs.clear();
}

int femfct_ensure_workspace(femfct_ctx* ctx, int32_t batch) {
    if (ctx->solver == FEMFCT_SOLVER_BICGSTAB) {
        int rk = femfct_ensure_krylov_ws(ctx, batch);
        if (rk != FEMFCT_OK) return rk;
    }
#ifdef FEMFCT_TUNING
    if (!ctx->d_mesh_trace && getenv("FEMFCT_MESH_TRACE"))
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_mesh_trace, 16 * sizeof(unsigned long long)));
#endif
    if (batch <= ctx->ws_batch) return FEMFCT_OK;
    femfct_drop_graphs(ctx);
    size_t nv = (size_t)batch * ctx->n, nm = nv * ctx->W;
    int rc;
#define A_(p, c) if ((rc = dev_alloc(ctx, &ctx->p, (c))) != FEMFCT_OK) return rc
    A_(d_L, nm + 1); A_(d_D, nm); A_(d_F, nm);     // d_L[nm] stays zero: read by k_strip_jacobi_pair_walk for a pair without an entry
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_L + nm, 0, sizeof(double), ctx->stream));
    A_(d_b, nv); A_(d_xa, nv); A_(d_xb, nv); A_(d_du, nv); A_(d_y0, nv); A_(d_y1, nv); A_(d_y2, nv);
    A_(d_rdu, nv); A_(d_rp, nv); A_(d_rm, nv);
    A_(d_part, (size_t)batch * 4 * FEMFCT_MAX_PARTIALS);
    A_(d_ctl, (size_t)batch);
    A_(d_partk, (size_t)batch * 16 * FEMFCT_MAX_PARTIALS);
    {   // word 0 stays zero (read in place of vanishing entries); one mask byte per node from word 1
        const size_t words = ((size_t)batch * (size_t)ctx->n + 7) / 8 + 1;
        A_(d_Lmask, words);
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_Lmask, 0, words * 8, ctx->stream));
    }
    {
        TilePlan tp;
        ctx->bigpart_count = 0;
        if (femfct_tile_plan(ctx, &tp, false) && femfct_tile_big(ctx, tp)) {
            // covers the 32-patch grid and the (coarser) 64-patch grid
            ctx->bigpart_count = (int64_t)tp.tiles * tp.tiles;
            A_(d_bigpart, (size_t)batch * ctx->bigpart_count);
        }
    }
#undef A_
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctl, 0, sizeof(StepCtl) * batch, ctx->stream));
    ctx->ws_batch = batch;
    return FEMFCT_OK;
}

// 0: one sweep per launch; 1: row strips (K sweeps per launch, coarse sweep count);
// 2: 2-D tiles on a small grid (exact sweep count logged by the last launch)
static int fusion_mode(const femfct_ctx* ctx, int* K) {
    TilePlan tp;
    if (femfct_tile_plan(ctx, &tp, false)) {
        *K = tp.K;
        return femfct_tile_big(ctx, tp) ? 1 : 2;
    }
    StripPlan pl;
    if (femfct_strip_plan(ctx, &pl)) { *K = pl.K; return 1; }
    *K = 1;
    return 0;
}

int femfct_fused_k(const femfct_ctx* ctx) {
    int K;
    fusion_mode(ctx, &K);
    return K;
}

int femfct_next_budget(const femfct_ctx* ctx, int worst, bool coarse) {
    int K;
    const int mode = fusion_mode(ctx, &K);
    int b;
    // tiles: counts are whole launches (an upper bound) unless the exact-count variant ran, which gets one
    // sweep of margin.  No margin otherwise: the halo depth is re-chosen from the budget, and a budget that
    // creeps up by one per sweep walks through 2 x 12, 2 x 13, 3 x 9, 3 x 10, ... launches for nothing.
    if (mode == 2) b = std::max(worst, 1) + ((coarse || !ctx->exact_iters) ? 0 : 1);
    else if (mode == 1) b = std::max(worst, 1);   // whole launches are formed from the budget by the launch plan
    else b = std::max(8, worst + worst / 8 + 2);
    return std::min(ctx->max_iters, b);
}

int femfct_grow_budget(const femfct_ctx* ctx, int budget) {
    int K;
    const int mode = fusion_mode(ctx, &K);
    if (mode == 2) return std::min(ctx->max_iters, budget + 2);
    if (mode == 1) return std::min(ctx->max_iters, budget + K);
    return std::min(ctx->max_iters, budget * 2);
}

int femfct_round_budget(const femfct_ctx* ctx, int b) {
    int K;
    if (fusion_mode(ctx, &K) == 0) b = (b + 3) & ~3;    // fewer distinct graphs for the one-sweep kernels
    if (b < 1) b = 1;
    if (b > ctx->max_iters) b = ctx->max_iters;
    return b;
}

// Shared tail of both pattern builders: uploads cols/tslot (host ELL arrays).
int femfct_install_pattern(femfct_ctx* ctx, int32_t n, int32_t W, const std::vector<int32_t>& cols,
                           const std::vector<uint8_t>& tslot) {
    int rc;
    ctx->n = n;
    ctx->W = W;
    if ((rc = dev_alloc(ctx, &ctx->d_cols, (size_t)W * n)) != FEMFCT_OK) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_tslot, (size_t)W * n)) != FEMFCT_OK) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->d_cols, cols.data(), sizeof(int32_t) * W * n, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_tslot, tslot.data(), sizeof(uint8_t) * W * n, hipMemcpyHostToDevice));
    if ((rc = dev_alloc(ctx, &ctx->d_M, (size_t)W * n)) != FEMFCT_OK) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_ml, (size_t)n)) != FEMFCT_OK) return rc;
    ctx->h_cols = cols;
    int32_t bwmax = 0;
    for (int32_t s = 1; s < W; ++s)
        for (int32_t i = 0; i < n; ++i) {
            int32_t d = cols[(size_t)s * n + i] - i;
            if (d < 0) d = -d;
            if (d > bwmax) bwmax = d;
        }
    ctx->bandwidth = bwmax;
    if (const char* e = getenv("FEMFCT_STRIPS")) ctx->use_strips = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_STRIP_K")) ctx->strip_k = atoi(e);
    if (const char* e = getenv("FEMFCT_TILES")) ctx->use_tiles = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_EXACT")) ctx->exact_iters = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_FUSE_BUILD")) ctx->fuse_build = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_PREASSEMBLE")) ctx->preassemble = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_PREASSEMBLE_MAX_GB")) ctx->preassemble_max_bytes = atof(e) * 1024.0 * 1024.0 * 1024.0;
    if (const char* e = getenv("FEMFCT_WG_SLOTS")) ctx->wg_slots = std::max(1, atoi(e));
    if (const char* e = getenv("FEMFCT_DEEP_HALO")) ctx->deep_halo = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_FUSE_FLUX")) ctx->fuse_flux = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_FUSE_DUDT")) ctx->fuse_dudt = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_FUSE_END")) ctx->fuse_end = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_TILE4")) ctx->tile4_mode = atoi(e);
    if (const char* e = getenv("FEMFCT_LMASK")) ctx->l_mask = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_T4_XCD")) ctx->t4_xcd = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_T4_WALK")) ctx->t4_walk = atoi(e);   // 2: walk without the carry (measurement)
    if (const char* e = getenv("FEMFCT_T4_SNAKE")) ctx->t4_snake = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_T4_PAIR")) ctx->t4_pair = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_PAIR_SHAPE")) ctx->pair_shape = atoi(e);
    if (const char* e = getenv("FEMFCT_PAIR_PRIO")) ctx->pair_prio = atoi(e);
    if (const char* e = getenv("FEMFCT_PAIR_SPLIT")) ctx->pair_split = std::min(90, std::max(10, atoi(e)));
    if (getenv("FEMFCT_PAIR_TRACE") && !ctx->d_pair_trace) {
        if (hipMalloc((void**)&ctx->d_pair_trace, sizeof(unsigned long long) * (FEMFCT_MAX_PARTIALS * 16 * 5 + 16)) == hipSuccess)
            hipMemset(ctx->d_pair_trace, 0, sizeof(unsigned long long) * (FEMFCT_MAX_PARTIALS * 16 * 5 + 16));
    }
    if (const char* e = getenv("FEMFCT_PAIR_STAGGER_US")) ctx->pair_stagger = (int)(atof(e) * 100.0);
    if (const char* e = getenv("FEMFCT_DEFER_CHECK")) ctx->defer_check = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_T4_INT")) ctx->t4_int = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_T4_WALKERS")) { int v = atoi(e); if (v > 0) ctx->num_cus = v; }   // tests: walks on small meshes
    if (const char* e = getenv("FEMFCT_INLINE_OPS")) ctx->inline_ops = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_HALF_D")) ctx->half_d = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_T4_STAGGER_US")) {
        const double us = atof(e);
        int pat = 7;
        if (const char* p = getenv("FEMFCT_T4_STAGGER_PAT")) pat = atoi(p);
        ctx->t4_stagger = us > 0 ? ((int)(us * 100.0) & 0xffffff) | ((pat & 15) << 24) : 0;
    }
    if (const char* e = getenv("FEMFCT_SINGLE_PATCH_BATCH")) { int v = atoi(e); ctx->single_patch_min_batch = v > 0 ? v : (1 << 30); }
    if (const char* e = getenv("FEMFCT_MESH_SOLVE")) ctx->mesh_solve = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_GEOM_MASS")) ctx->geom_mass = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_MESH_STEP")) ctx->mesh_step = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_MESH_STEP_BATCH")) ctx->mesh_step_min_batch = std::max(1, atoi(e));
    if (const char* e = getenv("FEMFCT_T4_DPP")) ctx->t4_dpp = atoi(e) != 0;
    if (const char* e = getenv("FEMFCT_T4_K")) ctx->t4_k = std::min(8, std::max(1, atoi(e)));
    if (const char* e = getenv("FEMFCT_SPECIES_SOLVER")) ctx->species_solver = atoi(e);
    if (const char* e = getenv("FEMFCT_STEPS_PER_GRAPH")) ctx->steps_per_graph = std::max(1, atoi(e));
    return femfct_strip_init(ctx);
}

// host CSR <-> ELL map for a pattern whose ELL cols are known (h_cols) and CSR given
static int build_csr_map(femfct_ctx* ctx) {
    const int32_t n = ctx->n, W = ctx->W;
    ctx->nnz_csr = ctx->h_indptr[n];
    ctx->h_csr2ell.assign(ctx->nnz_csr, 0);
    for (int32_t i = 0; i < n; ++i) {
        for (int32_t k = ctx->h_indptr[i]; k < ctx->h_indptr[i + 1]; ++k) {
            int32_t j = ctx->h_indices[k];
            int32_t s_found = -1;
            if (j == i) s_found = 0;
            else
                for (int32_t s = 1; s < W; ++s)
                    if (ctx->h_cols[(size_t)s * n + i] == j) { s_found = s; break; }
            if (s_found < 0) return femfct_fail(ctx, FEMFCT_ERR_INVALID, "CSR entry (%d,%d) not in the ELL pattern", i, j);
            ctx->h_csr2ell[k] = (int32_t)((size_t)s_found * n + i);
        }
    }
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->d_csr2ell, (size_t)ctx->nnz_csr)) != FEMFCT_OK) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->d_csr2ell, ctx->h_csr2ell.data(), sizeof(int32_t) * ctx->nnz_csr, hipMemcpyHostToDevice));
    if ((rc = dev_alloc(ctx, &ctx->d_hcsr, (size_t)ctx->nnz_csr)) != FEMFCT_OK) return rc;
    return FEMFCT_OK;
}

// ============================================================================ C ABI
extern "C" {

int femfct_abi_version(void) { return FEMFCT_ABI_VERSION; }

int femfct_create(femfct_ctx** out, int device_id) {
    if (!out) return FEMFCT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FEMFCT_ERR_HIP;
    if (device_id < 0 || device_id >= count) return FEMFCT_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return FEMFCT_ERR_HIP;
    femfct_ctx* ctx = new (std::nothrow) femfct_ctx();
    if (!ctx) return FEMFCT_ERR_NOMEM;
    ctx->device = device_id;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) ctx->num_cus = cus;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return FEMFCT_ERR_HIP;
    }
    // a rocprofiler-sdk tool in this process (rocprofv3 preloads it) intercepts every HSA queue: no graph replay then
    // (femfct_ctx::graphs_blocked); RTLD_NOLOAD only asks whether the library is already resident
    const char* pg = getenv("FEMFCT_PROFILER_GRAPHS");
    if (!(pg && atoi(pg) != 0)) {
        for (const char* name : {"librocprofiler-sdk.so.1", "librocprofiler-sdk.so"}) {
            if (void* h = dlopen(name, RTLD_LAZY | RTLD_NOLOAD)) {
                ctx->graphs_blocked = true;
                dlclose(h);
                break;
            }
        }
    }
    *out = ctx;
    return FEMFCT_OK;
}

// 1 while sweeps are replayed as hipGraphs, 0 while they are enqueued kernel by kernel (femfct_set_graphs(0), the
// per-class profiling of bench.py, or a rocprofiler-sdk tool attached to the process)
int femfct_graph_replay_active(const femfct_ctx* ctx, int* active_host) {
    if (!ctx || !active_host) return FEMFCT_ERR_INVALID;
    *active_host = (ctx->use_graphs && !ctx->prof_on && !ctx->graphs_blocked) ? 1 : 0;
    return FEMFCT_OK;
}

int femfct_destroy(femfct_ctx* ctx) {
    if (!ctx) return FEMFCT_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->d_pair_trace) {
        if (const char* path = getenv("FEMFCT_PAIR_TRACE")) {
            std::vector<unsigned long long> h((size_t)FEMFCT_MAX_PARTIALS * 16 * 5 + 16);
            if (hipMemcpy(h.data(), ctx->d_pair_trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
        }
        hipFree(ctx->d_pair_trace);
        ctx->d_pair_trace = nullptr;
    }
    femfct_release_pattern(ctx);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return FEMFCT_OK;
}

const char* femfct_last_error(const femfct_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int femfct_synchronize(femfct_ctx* ctx) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx, "null ctx");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FEMFCT_OK;
}

void* femfct_stream(femfct_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int femfct_set_solver(femfct_ctx* ctx, int solver, double rel_tol, int max_iters) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx, "null ctx");
    ARG_TRY(ctx, solver == FEMFCT_SOLVER_JACOBI || solver == FEMFCT_SOLVER_BICGSTAB, "unknown solver");
    ARG_TRY(ctx, rel_tol > 0 && rel_tol < 1 && max_iters >= 1, "bad tolerance / iteration cap");
    ctx->solver = solver;
    ctx->solver_user = solver;
    ctx->kind_low_bicg.clear();
    ctx->rel_tol = rel_tol;
    ctx->max_iters = max_iters;
    if (ctx->sweep_budget > max_iters) ctx->sweep_budget = max_iters;
    ctx->kind_budget.clear();
    femfct_drop_graphs(ctx);
    return FEMFCT_OK;
}

int femfct_set_fusion(femfct_ctx* ctx, int strips, int tiles) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx, "null ctx");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->use_strips = strips != 0;
    ctx->use_tiles = tiles != 0;
    femfct_drop_graphs(ctx);
    ctx->kind_budget.clear();
    ctx->ws_batch = 0;   // re-plan the workspace (big partial buffer) on the next call
    return FEMFCT_OK;
}

// which family of Jacobi / Chebyshev kernels a step with `batch` members runs (tests assert that a size class really
// exercises the kernels it is meant to cover; bench.py labels its roofline record with it)
int femfct_kernel_regime(const femfct_ctx* ctx, int32_t batch) {
    if (!ctx || ctx->n <= 0) return -1;
    if (femfct_tile4_wanted(ctx, batch)) return FEMFCT_REGIME_PATCH64;
    TilePlan tp;
    if (ctx->use_strips && femfct_tile_plan(ctx, &tp, true, 0, batch)) return FEMFCT_REGIME_TILE32;
    StripPlan sp;
    if (femfct_strip_plan(ctx, &sp)) return FEMFCT_REGIME_STRIPS;
    return FEMFCT_REGIME_ROWS;
}

int femfct_patch_walkers(const femfct_ctx* ctx, int32_t batch, int32_t sweeps) {
    if (!ctx || ctx->n <= 0 || batch < 1 || !femfct_tile4_wanted(ctx, batch) || femfct_single_patch(ctx, batch)) return 0;
    return femfct_tile4_walkers(ctx, femfct_tile4_halo(ctx, sweeps > 0 ? sweeps : 36), batch, false);
}

// Which bandwidth-regime kernels the most recent step / sweep enqueued: out[0] Jacobi launch (0 none or another regime,
// 1 one workgroup per patch, 2 k_strip4_jacobi_walk, 3 k_strip_jacobi_pair_walk), out[1] its walkers, out[2] interior
// patches per side of the split Chebyshev launch (0: not split), out[3] halo depth of the Jacobi launch.  Diagnostic.
int femfct_launch_info(const femfct_ctx* ctx, int32_t* out4_host) {
    if (!ctx || !out4_host) return FEMFCT_ERR_INVALID;
    for (int k = 0; k < 4; ++k) out4_host[k] = ctx->last_launch[k];
    return FEMFCT_OK;
}

// Share of the off-diagonal entries of the most recent low-order operator (batch member 0) that are non-zero, i.e.
// that the 64-patch Jacobi launches actually load; 1.0 when the zero mask is not in use.  Diagnostic for bench.py:
// the compulsory bytes of a launch "as executed" must not count entries the kernel never touches.  Synchronises.
int femfct_lowop_nonzero_fraction(femfct_ctx* ctx, double* fraction_host) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, fraction_host, "null argument");
    *fraction_host = 1.0;
    if (!ctx->d_Lmask || ctx->n <= 0 || !ctx->l_mask || !femfct_tile4_wanted(ctx, 1) || !ctx->t4_dpp ||
        ctx->solver != FEMFCT_SOLVER_JACOBI)
        return FEMFCT_OK;
    unsigned long long* d_cnt = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_cnt, sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_mask_popcount, dim3(1), dim3(256), 0, ctx->stream, (int64_t)ctx->n, ctx->d_Lmask + 1, d_cnt);   // batch member 0
    unsigned long long cnt = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(d_cnt);
    *fraction_host = (double)cnt / (6.0 * (double)ctx->n);
    return FEMFCT_OK;
}

int femfct_set_graphs(femfct_ctx* ctx, int enable) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx, "null ctx");
    ctx->use_graphs = enable != 0;
    if (!enable) femfct_drop_graphs(ctx);
    return FEMFCT_OK;
}

// per-kernel timing with HIP events on the ctx stream.  Profiling forces eager launches
// (graphs off) while enabled.
int femfct_set_profiling(femfct_ctx* ctx, int enable) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx, "null ctx");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    ctx->prof.clear();
    ctx->prof_on = enable != 0;
    return FEMFCT_OK;
}

int femfct_profile_report(femfct_ctx* ctx, double* total_ms_host, int32_t* launches_host, int32_t n_classes) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && total_ms_host && launches_host && n_classes >= KC_COUNT, "need >= 8 classes");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < n_classes; ++k) { total_ms_host[k] = 0.0; launches_host[k] = 0; }
    for (auto& r : ctx->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            total_ms_host[r.cls] += ms;
            launches_host[r.cls] += 1;
        }
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    ctx->prof.clear();
    return FEMFCT_OK;
}

// ----------------------------------------------------------------- memory
int femfct_malloc(femfct_ctx* ctx, void** dev_ptr, size_t bytes) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && dev_ptr, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dev_ptr, bytes ? bytes : 8);
    if (e != hipSuccess) return femfct_fail(ctx, FEMFCT_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return FEMFCT_OK;
}
int femfct_free(femfct_ctx* ctx, void* p) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx, "null ctx");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(p));
    return FEMFCT_OK;
}
int femfct_memcpy_h2d(femfct_ctx* ctx, void* d, const void* h, size_t bytes) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && (bytes == 0 || (d && h)), "null argument");
    HIP_TRY(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FEMFCT_OK;
}
int femfct_memcpy_d2h(femfct_ctx* ctx, void* h, const void* d, size_t bytes) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && (bytes == 0 || (d && h)), "null argument");
    HIP_TRY(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FEMFCT_OK;
}
int femfct_memcpy_d2d(femfct_ctx* ctx, void* d, const void* s, size_t bytes) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && (bytes == 0 || (d && s)), "null argument");
    HIP_TRY(ctx, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return FEMFCT_OK;
}
int femfct_memset0(femfct_ctx* ctx, void* d, size_t bytes) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && (bytes == 0 || d), "null argument");
    HIP_TRY(ctx, hipMemsetAsync(d, 0, bytes, ctx->stream));
    return FEMFCT_OK;
}

// ---------------------------------------------------------------- pattern
int femfct_set_pattern_csr(femfct_ctx* ctx, int32_t n, const int32_t* indptr, const int32_t* indices) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && indptr && indices && n > 0, "null/empty pattern");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    femfct_release_pattern(ctx);
    ARG_TRY(ctx, indptr[0] == 0, "indptr[0] must be 0");
    int32_t W = 1;
    for (int32_t i = 0; i < n; ++i) {
        ARG_TRY(ctx, indptr[i + 1] >= indptr[i], "indptr not monotone");
        int32_t len = 0;
        bool diag = false;
        for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            int32_t j = indices[k];
            ARG_TRY(ctx, j >= 0 && j < n, "column index out of range");
            if (j == i) diag = true; else ++len;
        }
        ARG_TRY(ctx, diag, "pattern must contain the full diagonal");
        W = std::max(W, len + 1);
    }
    if (W > FEMFCT_MAX_W) return femfct_fail(ctx, FEMFCT_ERR_INVALID, "row with %d entries exceeds FEMFCT_MAX_W=%d", W, FEMFCT_MAX_W);
    std::vector<int32_t> cols((size_t)W * n);
    std::vector<uint8_t> tslot((size_t)W * n);
    for (int32_t s = 0; s < W; ++s)
        for (int32_t i = 0; i < n; ++i) { cols[(size_t)s * n + i] = i; tslot[(size_t)s * n + i] = (uint8_t)s; }
    for (int32_t i = 0; i < n; ++i) {
        int32_t s = 1;
        for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k) {
            int32_t j = indices[k];
            if (j == i) continue;
            for (int32_t q = 1; q < s; ++q)
                ARG_TRY(ctx, cols[(size_t)q * n + i] != j, "duplicate column in a row");
            cols[(size_t)s * n + i] = j;
            ++s;
        }
    }
    // transposed-entry slots (structural symmetry required)
    for (int32_t i = 0; i < n; ++i)
        for (int32_t s = 1; s < W; ++s) {
            int32_t j = cols[(size_t)s * n + i];
            if (j == i) continue;  // padding
            int32_t ts = -1;
            for (int32_t q = 1; q < W; ++q)
                if (cols[(size_t)q * n + j] == i) { ts = q; break; }
            if (ts < 0) return femfct_fail(ctx, FEMFCT_ERR_INVALID, "pattern not symmetric: (%d,%d) has no transpose", i, j);
            tslot[(size_t)s * n + i] = (uint8_t)ts;
        }
    int rc = femfct_install_pattern(ctx, n, W, cols, tslot);
    if (rc != FEMFCT_OK) return rc;
    ctx->h_indptr.assign(indptr, indptr + n + 1);
    ctx->h_indices.assign(indices, indices + indptr[n]);
    return build_csr_map(ctx);
}

int32_t femfct_n(const femfct_ctx* ctx) { return ctx ? ctx->n : 0; }
int32_t femfct_ell_width(const femfct_ctx* ctx) { return ctx ? ctx->W : 0; }

int femfct_get_ell_cols(femfct_ctx* ctx, int32_t* cols_host) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && cols_host && ctx->n > 0, "no pattern");
    memcpy(cols_host, ctx->h_cols.data(), sizeof(int32_t) * ctx->h_cols.size());
    return FEMFCT_OK;
}


static int need_csr_map(femfct_ctx* ctx) {
    ARG_TRY(ctx, ctx && ctx->n > 0, "no pattern registered");
    if (ctx->d_csr2ell) return FEMFCT_OK;
    if (ctx->structured) {
        int rc = femfct_build_structured_csr(ctx);
        if (rc != FEMFCT_OK) return rc;
        return build_csr_map(ctx);
    }
    return femfct_fail(ctx, FEMFCT_ERR_INVALID, "no CSR pattern");
}

int femfct_csr_to_ell(femfct_ctx* ctx, const double* csr_vals_host, double* ell_dev) {
    FEMFCT_ENTER(ctx);
    int rc = need_csr_map(ctx);
    if (rc != FEMFCT_OK) return rc;
    ARG_TRY(ctx, csr_vals_host && ell_dev, "null argument");
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_hcsr, csr_vals_host, sizeof(double) * ctx->nnz_csr, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ell_dev, 0, sizeof(double) * ctx->W * ctx->n, ctx->stream));
    int bs = 256;
    hipLaunchKernelGGL(k_scatter_csr_to_ell, dim3((unsigned)((ctx->nnz_csr + bs - 1) / bs)), dim3(bs), 0, ctx->stream,
                       ctx->nnz_csr, ctx->d_csr2ell, ctx->d_hcsr, ell_dev);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // host buffer may be reused by the caller
    return FEMFCT_OK;
}

int femfct_ell_to_csr(femfct_ctx* ctx, const double* ell_dev, double* csr_vals_host) {
    FEMFCT_ENTER(ctx);
    int rc = need_csr_map(ctx);
    if (rc != FEMFCT_OK) return rc;
    ARG_TRY(ctx, csr_vals_host && ell_dev, "null argument");
    int bs = 256;
    hipLaunchKernelGGL(k_gather_ell_to_csr, dim3((unsigned)((ctx->nnz_csr + bs - 1) / bs)), dim3(bs), 0, ctx->stream,
                       ctx->nnz_csr, ctx->d_csr2ell, ell_dev, ctx->d_hcsr);
    HIP_TRY(ctx, hipMemcpyAsync(csr_vals_host, ctx->d_hcsr, sizeof(double) * ctx->nnz_csr, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FEMFCT_OK;
}

int femfct_set_mass(femfct_ctx* ctx, const double* M_csr_vals_host, const double* ml_host) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0, "no pattern registered");
    ARG_TRY(ctx, M_csr_vals_host && ml_host, "null argument");
    int rc = femfct_csr_to_ell(ctx, M_csr_vals_host, ctx->d_M);
    if (rc != FEMFCT_OK) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->d_ml, ml_host, sizeof(double) * ctx->n, hipMemcpyHostToDevice));
    ctx->have_mass = true;
    ctx->mass_is_mesh = false;      // a user matrix: the geometry-defined mass kernels do not apply
    return FEMFCT_OK;
}

const double* femfct_mass_ell(const femfct_ctx* ctx) { return ctx && ctx->have_mass ? ctx->d_M : nullptr; }
const double* femfct_stiffness_ell(const femfct_ctx* ctx) { return ctx ? ctx->d_Ad : nullptr; }
const double* femfct_lumped_mass(const femfct_ctx* ctx) { return ctx && ctx->have_mass ? ctx->d_ml : nullptr; }

// ------------------------------------------------------------------- step
static int launch_step(femfct_ctx* ctx, const double* A, const double* N, int32_t nshared, const double* rhs,
                       const double* u_n, double dt, double* u_out, int32_t batch) {
    const int budget = femfct_round_budget(ctx, ctx->sweep_budget);
    femfct_ctx::GraphKey key{(uint64_t)1, key_bits(A), key_bits(N), key_bits(rhs), key_bits(u_n), key_bits(u_out),
                             key_bits(nshared), key_bits(batch), key_bits((int32_t)budget), key_bits(dt),
                             key_bits(ctx->rel_tol)};
    return femfct_run_graph(ctx, key, [&]() {
        return femfct_enqueue_step(ctx, A, N, nshared, rhs, u_n, dt, u_out, batch, budget);
    });
}

int femfct_fct_step(femfct_ctx* ctx, const double* A_ell, const double* N_ell, int32_t N_shared, const double* rhs,
                    const double* u_n, double dt, double* u_out, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0, "no pattern registered");
    ARG_TRY(ctx, ctx->have_mass, "mass matrix not set (femfct_set_mass / femfct_set_mesh_square)");
    ARG_TRY(ctx, A_ell && u_n && u_out, "null argument");
    ARG_TRY(ctx, batch >= 1 && dt > 0, "batch must be >= 1 and dt > 0");
    int rc = femfct_ensure_workspace(ctx, batch);
    if (rc != FEMFCT_OK) return rc;
    return launch_step(ctx, A_ell, N_ell, N_shared, rhs, u_n, dt, u_out, batch);
}

int femfct_last_step_info(femfct_ctx* ctx, femfct_step_info* info, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && info && batch >= 1 && batch <= ctx->ws_batch, "bad argument");
    std::vector<StepCtl> h(batch);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->d_ctl, sizeof(StepCtl) * batch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int worst = 0;
    for (int b = 0; b < batch; ++b) {
        info[b].flags = h[b].flags;
        info[b].solver_iters = h[b].iters;
        info[b].solver_resid = h[b].resid;
        info[b].min_rowsum = h[b].min_rowsum;
        worst = std::max(worst, h[b].iters);
    }
    // adapt the number of sweeps enqueued per step to what the operator needs
    if (!(h[0].flags & FEMFCT_FLAG_SOLVER_BUDGET)) ctx->sweep_budget = femfct_next_budget(ctx, worst);
    return FEMFCT_OK;
}

int femfct_fct_step_host(femfct_ctx* ctx, const double* A_csr, const double* N_csr, const double* rhs,
                         const double* u_n, double dt, double* u_out, femfct_step_info* info) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0, "no pattern registered");
    ARG_TRY(ctx, A_csr && u_n && u_out, "null argument");
    int rc;
    const size_t n = ctx->n, nm = n * ctx->W;
    if (!ctx->d_hA) {
        if ((rc = dev_alloc(ctx, &ctx->d_hA, nm)) != FEMFCT_OK) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_hN, nm)) != FEMFCT_OK) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_hrhs, n)) != FEMFCT_OK) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_hu, n)) != FEMFCT_OK) return rc;
        if ((rc = dev_alloc(ctx, &ctx->d_hout, n)) != FEMFCT_OK) return rc;
    }
    if ((rc = femfct_csr_to_ell(ctx, A_csr, ctx->d_hA)) != FEMFCT_OK) return rc;
    if (N_csr && (rc = femfct_csr_to_ell(ctx, N_csr, ctx->d_hN)) != FEMFCT_OK) return rc;
    if (rhs) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_hrhs, rhs, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_hu, u_n, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    femfct_step_info local;
    for (;;) {
        rc = femfct_fct_step(ctx, ctx->d_hA, N_csr ? ctx->d_hN : nullptr, 0, rhs ? ctx->d_hrhs : nullptr, ctx->d_hu,
                             dt, ctx->d_hout, 1);
        if (rc != FEMFCT_OK) return rc;
        const int used_budget = femfct_round_budget(ctx, ctx->sweep_budget);
        if ((rc = femfct_last_step_info(ctx, &local, 1)) != FEMFCT_OK) return rc;
        if (!(local.flags & FEMFCT_FLAG_SOLVER_BUDGET)) break;
        if (used_budget >= ctx->max_iters)
            return femfct_fail(ctx, FEMFCT_ERR_NOT_CONVERGED,
                               "low-order solve: residual %.3e after %d sweeps (tol %.1e)", local.solver_resid,
                               local.solver_iters, ctx->rel_tol);
        ctx->sweep_budget = femfct_grow_budget(ctx, used_budget);
    }
    if (info) *info = local;
    HIP_TRY(ctx, hipMemcpyAsync(u_out, ctx->d_hout, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FEMFCT_OK;
}

int femfct_chebsi(femfct_ctx* ctx, const double* b_dev, double* y_dev, int32_t cheb_iter, double lmin, double lmax,
                  int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0 && ctx->have_mass, "mass matrix not set");
    ARG_TRY(ctx, b_dev && y_dev && cheb_iter >= 1 && batch >= 1 && lmin + lmax != 0, "bad argument");
    int rc = femfct_ensure_workspace(ctx, batch);
    if (rc != FEMFCT_OK) return rc;
    return femfct_enqueue_cheb(ctx, b_dev, y_dev, cheb_iter, lmin, lmax, batch, false);
}

int femfct_chebsi_md(femfct_ctx* ctx, const double* b_dev, double* y_dev, const double* md_dev, int32_t cheb_iter,
                     double lmin, double lmax, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0 && ctx->have_mass, "mass matrix not set");
    ARG_TRY(ctx, b_dev && y_dev && md_dev && cheb_iter >= 1 && batch == 1 && lmin + lmax != 0, "bad argument");
    int rc = femfct_ensure_workspace(ctx, batch);
    if (rc != FEMFCT_OK) return rc;
    return femfct_enqueue_cheb(ctx, b_dev, y_dev, cheb_iter, lmin, lmax, batch, false, md_dev);
}

int femfct_artificial_diffusion(femfct_ctx* ctx, const double* K_ell, double* D_ell, int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0, "no pattern registered");
    ARG_TRY(ctx, K_ell && D_ell && batch >= 1, "bad argument");
    return femfct_enqueue_artdiff(ctx, K_ell, D_ell, batch);
}

int femfct_spmv(femfct_ctx* ctx, const double* mat_ell, const double* x_dev, double alpha, double beta, double* y_dev,
                int32_t batch) {
    FEMFCT_ENTER(ctx);
    ARG_TRY(ctx, ctx && ctx->n > 0, "no pattern registered");
    ARG_TRY(ctx, mat_ell && x_dev && y_dev && batch >= 1, "bad argument");
    return femfct_enqueue_spmv(ctx, mat_ell, x_dev, alpha, beta, y_dev, batch);
}

}  // extern "C"
