// Internal declarations of libfemfct (gfx950).  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>
#include <set>
#include <tuple>

#include "../../include/femfct.h"

#define FEMFCT_MAX_W 16          // widest ELL row supported (P1 structured mesh: 7)
#ifndef FEMFCT_MAX_PARTIALS
#define FEMFCT_MAX_PARTIALS 2048 // cap on per-kernel block partials (row chunks beyond)
#endif

// ---------------------------------------------------------------------------
// device-side control block of one low-order solve / step (one per batch member)
// ---------------------------------------------------------------------------
struct StepCtl {
    int32_t flags;        // FEMFCT_FLAG_*
    int32_t iters;        // Jacobi sweeps that did work
    int32_t done;         // set when the residual test passed
    int32_t parity;       // which ping-pong buffer holds the solution (0: xa, 1: xb)
    double  resid;        // ||r||_inf / ||b||_inf seen by the last active sweep
    double  bnorm;        // ||b||_inf
    double  min_rowsum;   // min_i rowsum(L)
    double  rs[2];        // residual max of fused launch j at [j & 1] when a separate reduce kernel is used
    double  pad;
};

struct femfct_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // pattern
    int32_t n = 0;            // rows
    int32_t W = 0;            // ELL width
    int64_t nnz_csr = 0;
    int32_t* d_cols = nullptr;     // [W*n] column of (slot,row); padded entries = row
    uint8_t* d_tslot = nullptr;    // [W*n] slot of the transposed entry in row cols[s,i]
    int32_t* d_csr2ell = nullptr;  // [nnz_csr] linear ELL index of every CSR entry
    std::vector<int32_t> h_indptr, h_indices, h_csr2ell, h_cols;
    bool structured = false;
    bool mass_is_mesh = false;     // d_M = the structured mesh's own mass matrix
    double a1 = 0, a2 = 0, h = 0;
    int32_t n_cells = 0, N = 0, order = 0;
    int32_t* d_d2v = nullptr;      // [n] dof -> vertex (structured, FEniCS order only)

    // constant operators
    double* d_M = nullptr;    // ELL [W*n]
    double* d_Ad = nullptr;   // ELL [W*n] (structured only)
    double* d_ml = nullptr;   // [n]
    bool have_mass = false;

    // solver settings
    int solver = FEMFCT_SOLVER_JACOBI;       // low-order solver of the sweep in progress
    int solver_user = FEMFCT_SOLVER_JACOBI;  // what femfct_set_solver asked for
    std::set<int> kind_low_bicg;             // sweep kinds whose Jacobi iteration did not contract: BiCGStab from then on
    double rel_tol = 1e-13;
    int max_iters = 400;
    int sweep_budget = 48;      // adaptive: sweeps enqueued per step (stand-alone femfct_fct_step)
    std::map<int, int> kind_good;   // last Jacobi budget that sufficed, per kind
    std::map<int, int> kind_fail;   // largest Jacobi budget known to be too small, per kind
    std::map<int, int> kind_mesh_budget;   // sweep cap of the one-workgroup step, per kind (fixed unless it proved too small)
    std::map<int, int> kind_budget, kind_kbudget;   // per trajectory kind (forward/adjoint of each system)
    bool use_graphs = true;
    // hipGraph replay is held back while a rocprofiler-sdk tool intercepts the HSA queues (rocprofv3 --kernel-trace /
    // --pmc): ROCm 7.2.0's interceptor walks a graph launch's AQL packet batch past the end of the 16384-packet ring when
    // the batch straddles the wrap (host SIGSEGV inside librocprofiler-sdk.so; DESIGN.md section 9,
    // tools/graph_intercept_probe.hip reproduces it without this library).  FEMFCT_PROFILER_GRAPHS=1 overrides.
    bool graphs_blocked = false;
    int graph_captures = 0;            // (FEMFCT_DEBUG: graphs captured + instantiated so far)
    bool profiler_graphs_ok = false;   // FEMFCT_PROFILER_GRAPHS=1
    int32_t steps_per_graph = 50;   // time steps captured per hipGraph in the trajectory drivers (only the last one moves the counters)
    bool use_strips = true;     // strip-fused multi-sweep kernels when the bandwidth allows
    bool use_tiles = true;      // 2-D tile variant (structured mesh, vertex order)
    bool mesh_solve = true;         // species solves of meshes with n <= 4096 as one workgroup per system (FEMFCT_MESH_SOLVE)
    bool mesh_solve_attr[8] = {false, false, false, false, false, false, false, false};
    int single_patch_min_batch = 8; // whole-mesh workgroups (N <= 48) for species solves from this batch size on (FEMFCT_SINGLE_PATCH_BATCH; 0 = off)
    bool form_groups = true;    // FEMFCT_FORM_GROUPS: independent quadrature forms of a time step share one launch (forms.h: FormGroup)
    bool geom_rot = true;       // FEMFCT_GEOM_ROT: bandwidth-regime step kernels derive the rotation operator from (ix, iy)
    double rot_om = 0.0;        // angular velocity of the last femfct_assemble_rotation
    bool rot_om_set = false;
    int* d_rot_check = nullptr;
    bool last_rot_geom = false; // femfct_rotation_derived
    bool geom_mass = true;      // structured mesh: Chebyshev on M from the cell geometry instead of the stored matrix
    bool t4_dpp = true;         // 64-patch kernels: register-resident strips + DPP lane shifts (else LDS image)
    int t4_k = 8;               // sweeps per 64-patch launch (FEMFCT_T4_K: measurement knob, 1..8)
    int t4_snake = 1;           // walking Jacobi launches alternate their direction (FEMFCT_T4_SNAKE)
    bool t4_pair = true;        // upwind rows as one value per opposing pair: k_strip_jacobi_pair_walk, two workgroups per CU (FEMFCT_T4_PAIR)
    int pair_stagger = 0;       // FEMFCT_PAIR_STAGGER_US * 100: ticks of the 100 MHz clock the second half of the pair walkers waits
    // what the most recent bandwidth-regime launches were (femfct_launch_info): Jacobi kernel (0 none, 1 one workgroup per
    // patch, 2 k_strip4_jacobi_walk, 3 k_strip_jacobi_pair_walk), its walkers, interior Chebyshev patches per side, halo depth
    int32_t last_launch[4] = {0, 0, 0, 0};
    unsigned long long* d_pair_trace = nullptr;   // FEMFCT_PAIR_TRACE=<file>: phase timestamps of the pair walkers, dumped at destroy
    int pair_prio = 0, pair_split = 50;   // FEMFCT_PAIR_PRIO / FEMFCT_PAIR_SPLIT: balance between the two workgroups of a CU (k_strip_jacobi_pair_walk)
    int pair_shape = 5;         // FEMFCT_PAIR_SHAPE: 5 = 6 rows x 8 waves (the product's), 3 = 8 x 8, 4 = 7 x 8, 6 = 12 x 4 (measurement)
    bool pair_rows = false;     // set by femfct_run_sweep for the sweep in progress: its kind has only shown upwind rows so far
    std::set<int> kind_fullrows;    // sweep kinds that raised FEMFCT_FLAG_ROW_PAIRS: full-row kernels from then on
    // one workgroup = one trajectory (kernels_mesh.hip): the whole step of a small mesh (N <= 42) in one launch
    bool mesh_step = true;          // FEMFCT_MESH_STEP
    int mesh_step_min_batch = 1;    // FEMFCT_MESH_STEP_BATCH: trajectories per launch from which it replaces the tile path
    int mesh_step_min_batch_large = 64;   // FEMFCT_MESH_STEP_BATCH_LARGE: the same for N = 81 (3 x 3 blocks)
    bool mesh_step_attr[6] = {false, false, false, false, false, false};
    double* d_zero = nullptr;             // n zeros: the 'no rhs' vector of the one-workgroup step
    unsigned long long* d_mesh_trace = nullptr;   // FEMFCT_TUNING builds only
    int defer_check = 1;        // two-launch tile solves: residual test reconstructed after the solve (FEMFCT_DEFER_CHECK)
    int t4_int = 1;             // Chebyshev on the mesh's mass matrix: interior patches by the two-workgroups-per-CU kernel (FEMFCT_T4_INT)
    int t4_walk = 1;            // 64-patch Jacobi: persistent workgroups walk down columns of patches, shared rows carried in LDS (FEMFCT_T4_WALK)
    int num_cus = 256;          // compute units of the device (one 1024-thread walker each)
    int t4_xcd = 0;             // 64-patch kernels: x-neighbouring patches under the same XCD's L2 (FEMFCT_T4_XCD)
    int t4_stagger = 600 | (7 << 24);   // 64-patch Jacobi launches of >= 4 rounds: first-round stagger, ticks of 10 ns | pattern << 24
                                        // (FEMFCT_T4_STAGGER_US / _PAT; 6 us, second half of every XCD; 0 = off)
    int tile4_mode = 1;         // 64x64-patch tiles: 0 off, 1 automatic (n*batch large), 2 always
    bool fuse_end = true;       // log + level advance done by the last workgroup of the step's final kernel
    bool fuse_build = true;     // low-order operator construction inside the first tile-Jacobi launch (small grids)
    int wg_slots = 256;         // workgroups that run at once without queueing: bound for the deep-halo / fused-tail choices
    bool deep_halo = true;      // Jacobi halos 11..13 while every tile gets its own CU (fewer launches)
    bool fuse_flux = true;      // last Chebyshev iterations + flux + limiter in one tile launch (small grids)
    bool fuse_dudt = true;      // du/dt rhs + first Chebyshev iterations in one tile launch (small grids)
    bool exact_iters = false;   // last fused launch logs per-sweep residuals (exact sweep count; measured 10 % slower)
    int32_t bandwidth = 0;      // max |col - row| of the pattern
    int32_t strip_k = 0;        // 0: automatic
    bool implicit_cols = false; // structured mesh in vertex order: neighbour index = row + const offset

    // workspace (sized for ws_batch systems)
    int32_t ws_batch = 0;
    double *d_L = nullptr, *d_D = nullptr, *d_F = nullptr;           // ELL [B*W*n]
    double *d_b = nullptr, *d_xa = nullptr, *d_xb = nullptr;          // [B*n]
    double *d_du = nullptr, *d_y0 = nullptr, *d_y1 = nullptr, *d_y2 = nullptr, *d_rdu = nullptr;
    double *d_rp = nullptr, *d_rm = nullptr;                          // R+ / R- [B*n]
    double *d_part = nullptr;                                         // block partials [B*4*MAX_PARTIALS]
    StepCtl* d_ctl = nullptr;                                         // [B]
    double* d_partk = nullptr;       // [B][16][MAX_PARTIALS] per-sweep residual partials of the last fused launch
    double* d_bigpart = nullptr;     // [B * bigpart_count] residual partials of fused launches on large grids
    unsigned long long* d_Lmask = nullptr;   // one zero word, then [B][n] bytes: bit s-1 of node i: l_(i,s) != 0 (written by k_build_low for the 64-patch Jacobi kernels)
    bool half_d = true;              // bandwidth regime: D stored once per edge (FEMFCT_HALF_D)
    bool inline_ops = true;          // solid-body sweeps in the bandwidth regime derive A inside the step kernels (FEMFCT_INLINE_OPS)
    bool l_mask = true;              // skip the exactly-zero off-diagonals of L when loading Jacobi patches (FEMFCT_LMASK)
    int64_t bigpart_count = 0;
    // host staging for the *_host convenience calls
    double *d_hA = nullptr, *d_hN = nullptr, *d_hrhs = nullptr, *d_hu = nullptr, *d_hout = nullptr, *d_hcsr = nullptr;

    // hipGraph cache of captured kernel sequences, keyed by the bit patterns of every
    // captured argument (pointers, scalars, batch, sweep budget)
    typedef std::vector<uint64_t> GraphKey;
    std::map<GraphKey, hipGraphExec_t> graphs;

    // trajectory workspace
    int32_t tr_batch = 0, tr_steps = 0;
    double* d_trAall = nullptr;     // pre-assembled per-level operators of a sweep [members][Nt][W*n]
    size_t trAall_count = 0;
    bool preassemble = true;
    double preassemble_max_bytes = 16.0 * 1024 * 1024 * 1024;
    double *d_trA = nullptr, *d_trN = nullptr, *d_trRhs = nullptr;  // per-step operators [B*W*n], [B*n]
    int32_t* d_level = nullptr;                                     // [2]: current level, step ordinal
    StepCtl* d_log = nullptr;                                       // [tr_steps * tr_batch]
    unsigned* d_ticket = nullptr;                                   // last-block ticket of the fused step end
    // graph-relative time levels: step r of a captured graph of R steps addresses level[0] + r * delta (+ its offsets) and
    // logs at ordinal level[1] + r; only the graph's last step moves the device counters (by R * delta and R).  Set by
    // femfct_run_graph_reps around every enqueue; outside it (0, 0, true, 1) reproduces one self-contained step.
    int level_bias = 0, ord_bias = 0, rep_total = 1;
    bool rep_last = true;
    int end_req_delta = 0;          // != 0: the next step's last kernel also logs + advances the level
    bool end_req_krylov = false, end_fused = false;
    std::vector<StepCtl> h_log;                                     // last trajectory's log
    int32_t log_steps = 0, log_batch = 0;

    // optional per-kernel HIP-event timing (femfct_set_profiling): event pairs recorded on the
    // ctx stream around every launch of the step sequence, resolved by femfct_profile_report
    bool prof_on = false;
    struct ProfRec { int cls; hipEvent_t a, b; };
    std::vector<ProfRec> prof;

    // Krylov workspace / settings (kernels_krylov.hip)
    int32_t kry_batch = 0;
    double* d_kry = nullptr;        // 9 vectors [9][kry_batch*n]
    double* d_kry_part = nullptr;
    void* d_kry_ctl = nullptr;      // KrylovCtl[kry_batch]
    void* d_kry_ctl2 = nullptr;     // KrylovCtl[kry_batch] of the FCT step's own low-order solve (solver = BICGSTAB)
    double kry_tol = 1e-13;
    int kry_max_iters = 2000;
    int kry_budget = 40;            // adaptive
    void* d_klog = nullptr;         // KrylovCtl[tr_steps * tr_batch]
    int species_solver = 0;         // 0: Chebyshev where the tile plan applies (fallback BiCGStab), 1: BiCGStab
    std::set<int> kind_cheb_off;    // sweep kinds whose Chebyshev iteration failed to contract
    double* d_chs_om = nullptr;     // [kry_batch][chs_om_cap] omega tables
    double* d_chs_scale = nullptr;  // [kry_batch] (lmin+lmax)/2
    int chs_om_cap = 0;
    std::vector<char> h_klog;
    // extra trajectory operators
    double *d_trMat = nullptr, *d_trBase = nullptr, *d_trBase2 = nullptr;  // [B*W*n], [W*n], [W*n]
    double *d_trRhs2 = nullptr, *d_trTmp = nullptr;                        // [B*n]
    double* d_wscale = nullptr;     // per-level factors s(t_k) of a separable time-dependent wind [wscale_count]
    size_t wscale_count = 0;

    // scratch for reductions (kernels_pgd.hip)
    double* d_scratch = nullptr;
    size_t scratch_count = 0;
};

// error helpers ---------------------------------------------------------------
int femfct_fail(femfct_ctx* ctx, int code, const char* fmt, ...);
#define HIP_TRY(ctx, expr)                                                          \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess)                                                       \
            return femfct_fail((ctx), FEMFCT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                               hipGetErrorString(_e), __FILE__, __LINE__);          \
    } while (0)
#define ARG_TRY(ctx, cond, msg)                                                     \
    do {                                                                            \
        if (!(cond)) return femfct_fail((ctx), FEMFCT_ERR_INVALID, "%s", msg);      \
    } while (0)

// First statement of every extern "C" entry point that allocates, copies or launches: HIP's current device is per
// thread, so a context created on device d must select d on whichever thread calls it (two contexts on different
// GPUs in one process, or a context used from a thread other than its creator's).
#define FEMFCT_ENTER(ctx)                                                           \
    do {                                                                            \
        if (!(ctx)) return FEMFCT_ERR_INVALID;                                      \
        HIP_TRY((ctx), hipSetDevice((ctx)->device));                                \
    } while (0)

int femfct_ensure_workspace(femfct_ctx* ctx, int32_t batch);
int femfct_ensure_krylov_ws(femfct_ctx* ctx, int32_t batch);
void femfct_drop_graphs(femfct_ctx* ctx);
void femfct_release_pattern(femfct_ctx* ctx);
int femfct_round_budget(const femfct_ctx* ctx, int b);

// graph-key helpers
static inline uint64_t key_bits(const void* p) { return (uint64_t)(uintptr_t)p; }
static inline uint64_t key_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline uint64_t key_bits(int64_t v) { return (uint64_t)v; }
static inline uint64_t key_bits(int32_t v) { return (uint64_t)(int64_t)v; }
// Run `enqueue` (a callable that launches kernels on ctx->stream) through a cached hipGraph.
template <class F>
int femfct_run_graph(femfct_ctx* ctx, const femfct_ctx::GraphKey& key, F&& enqueue);

// strip-fused kernels (kernels_strip.hip)
struct StripPlan { int K, R, bw, rpt, S; };
bool femfct_strip_plan(const femfct_ctx* ctx, StripPlan* pl);
int femfct_strip_init(femfct_ctx* ctx);
int femfct_enqueue_strip_jacobi(femfct_ctx* ctx, const StripPlan& pl, const double* L, const double* b, double* xa,
                                double* xb, int launch, int g_build, int32_t batch);
int femfct_enqueue_strip_cheb(femfct_ctx* ctx, const StripPlan& pl, const double* b, const double* in_mid,
                              const double* in_old, double* y_out, int k_first, int k_last, const double* omegas,
                              double md_scale, double* bufA0, double* bufA1, double* bufB0, double* bufB1,
                              int32_t batch);
struct TilePlan { int tiles, K, H; };
bool femfct_tile_plan(const femfct_ctx* ctx, TilePlan* pl, bool need_partials = true, int budget = 0, int batch = 1);
int femfct_enqueue_tile_jacobi(femfct_ctx* ctx, const TilePlan& pl, const double* L, const double* b, double* xa,
                               double* xb, int launch, int g_build, int32_t batch, bool last, int bn_launch = 0, int defer = 0);
int femfct_enqueue_tile_build_jacobi(femfct_ctx* ctx, const TilePlan& pl, struct MatRef A, const double* Nm, int32_t nshared,
                                     struct VecRef rhs, int64_t rhs_bstride, struct VecRef u_n, int64_t u_bstride, double dt,
                                     int32_t batch);
bool femfct_tile_big(const femfct_ctx* ctx, const TilePlan& pl);   // more workgroups than in-kernel partials
int femfct_enqueue_tile_cheb(femfct_ctx* ctx, const TilePlan& pl, const double* b, const double* in_mid,
                             const double* in_old, double* y_out, int k_first, int k_last, const double* omegas,
                             double md_scale, double* bufA0, double* bufA1, double* bufB0, double* bufB1, int32_t batch,
                             const struct ChebIO* io = nullptr);
int femfct_enqueue_tile_flux_limit(femfct_ctx* ctx, const double* D, const double* ulow, const double* du, double dt,
                                   struct VecRef out, int64_t out_bstride, int32_t batch, bool* fuse_end_io, int half_d = 0);
int femfct_enqueue_tile_dudt_cheb(femfct_ctx* ctx, struct MatRef A, struct VecRef rhs, int64_t rhs_bstride, double* ulow,
                                  int budget_units, int part_count, int iters_per_unit, int exact_k, int iters,
                                  const double* omegas, double md_scale, int32_t batch, int* tail_first = nullptr);
bool femfct_cheb_flux_fusable(const femfct_ctx* ctx, int32_t batch);
bool femfct_rotation_is_geometric(femfct_ctx* ctx, const double* Arot, double* om_out);   // kernels_asm.hip
bool femfct_geom_mass(const femfct_ctx* ctx);   // M may be derived from the cell geometry instead of loaded
int femfct_enqueue_tile_cheb_flux_limit(femfct_ctx* ctx, const double* b, const double* in_mid, const double* in_old,
                                        int k_first, int k_last, const double* omegas, double md_scale, const double* D,
                                        const double* ulow, double dt, struct VecRef out, int64_t out_bstride, int32_t batch,
                                        bool fuse_end);
int femfct_tile4_init(femfct_ctx* ctx);
bool femfct_tile4_wanted(const femfct_ctx* ctx, int32_t batch);
int femfct_tile4_tiles(const femfct_ctx* ctx, int H = 8);
int femfct_tile4_walkers(const femfct_ctx* ctx, int H, int32_t batch, bool pair);   // 0: one workgroup per patch
bool femfct_jacobi_pair_wanted(const femfct_ctx* ctx, int H, int32_t batch, bool have_lmask, bool assume_upwind_rows = false);
int femfct_tile4_halo(const femfct_ctx* ctx, int sweeps);
bool femfct_single_patch(const femfct_ctx* ctx, int32_t batch);
// launches and sweeps per launch the low-order solve of femfct_enqueue_step_mat will use for a budget
bool femfct_jacobi_plan(const femfct_ctx* ctx, int budget, int batch, int* K, int* launches);
int femfct_enqueue_tile4_jacobi(femfct_ctx* ctx, const double* L, const double* b, double* xa, double* xb, int launch,
                                int g_build, int32_t batch, int H = 8, int K = 8, int check_every = 0,
                                const uint8_t* lmask = nullptr);
int femfct_enqueue_tile4_cheb(femfct_ctx* ctx, const double* b, const double* in_mid, const double* in_old, double* y_out,
                              int k_first, int k_last, const double* omegas, double md_scale, double* bufA0, double* bufA1,
                              double* bufB0, double* bufB1, int32_t batch,
                              const struct ChebIO* io = nullptr);
// one workgroup per trajectory (kernels_mesh.hip)
bool femfct_mesh_step_wanted(const femfct_ctx* ctx, int32_t batch, bool have_nm = false);
int femfct_enqueue_mesh_step(femfct_ctx* ctx, struct MatRef A, const double* Nm, int32_t nshared, struct VecRef rhs,
                             int64_t rhs_bstride, struct VecRef u_n, int64_t u_bstride, double dt, struct VecRef u_out,
                             int64_t out_bstride, int32_t batch, int32_t budget, bool fuse_end);
// number of sweeps one fused launch performs (1 when neither tiles nor strips apply)
int femfct_fused_k(const femfct_ctx* ctx);
// sweep-budget policy (sweeps to enqueue for the next step sequence)
int femfct_next_budget(const femfct_ctx* ctx, int worst_iters, bool coarse = false);
int femfct_grow_budget(const femfct_ctx* ctx, int budget);

// kernel classes for profiling
enum { KC_BUILD_LOW = 0, KC_JACOBI, KC_DUDT_RHS, KC_CHEB, KC_FLUX, KC_LIMIT, KC_ASSEMBLE, KC_OTHER, KC_COUNT };
void femfct_prof_begin(femfct_ctx* ctx, int cls);
void femfct_prof_end(femfct_ctx* ctx);

// kernel launchers (kernels_step.hip) -----------------------------------------
struct LaunchGeom { dim3 grid; dim3 block; };
LaunchGeom femfct_geom(const femfct_ctx* ctx, int32_t batch);

int femfct_enqueue_step(femfct_ctx* ctx, const double* A, const double* N, int32_t nshared,
                        const double* rhs, const double* u_n, double dt, double* u_out,
                        int32_t batch, int32_t budget);

bool femfct_profiler_attached();   // ctx.hip: a rocprofiler-sdk tool is resident in this process

template <class F>
int femfct_run_graph(femfct_ctx* ctx, const femfct_ctx::GraphKey& key, F&& enqueue) {
    if (!ctx->use_graphs || ctx->prof_on || ctx->graphs_blocked) return enqueue();
    auto it = ctx->graphs.find(key);
    if (it == ctx->graphs.end()) {
        if (!ctx->profiler_graphs_ok && femfct_profiler_attached()) {      // a tool attached since femfct_create
            ctx->graphs_blocked = true;
            return enqueue();
        }
        if (ctx->graphs.size() > 64) femfct_drop_graphs(ctx);
        ++ctx->graph_captures;
        hipGraph_t graph = nullptr;
        HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        int rc = enqueue();
        hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        if (rc != FEMFCT_OK) { if (graph) hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) return femfct_fail(ctx, FEMFCT_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (e != hipSuccess) return femfct_fail(ctx, FEMFCT_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        it = ctx->graphs.emplace(key, exec).first;
    }
    HIP_TRY(ctx, hipGraphLaunch(it->second, ctx->stream));
    return FEMFCT_OK;
}
