/*
 * femfct.h -- C ABI of libfemfct.so, the MI355X (gfx950) FEM-FCT forward/adjoint
 * time-stepping core.
 *
 * The reference (KarolinaBenkova/FEM-FCT-PDECO) has no FFI layer: its boundary
 * is the Python call signature of helpers.py.  Each entry point below names the
 * reference interface (file:line under /root/reference) it replaces; the Python
 * mirror of those signatures lives in fem-fct-pdeco_amd/ and binds this header
 * through ctypes.
 *
 * Conventions
 *   - plain C, no C++ types or exceptions cross the ABI; every function returns
 *     an int status (FEMFCT_OK = 0) unless documented otherwise;
 *   - one femfct_ctx = one GPU + one HIP stream; a ctx is not thread-safe,
 *     distinct ctxs are independent (one process per GPU);
 *   - the caller owns host buffers; device buffers are either owned by the ctx
 *     or allocated by the caller (femfct_malloc, or any hipMalloc'd pointer such
 *     as torch.Tensor.data_ptr());
 *   - all arithmetic is IEEE float64; indices are int32;
 *   - "dev" pointers are device pointers, "host" pointers are host pointers;
 *   - all work is enqueued on the ctx stream; functions taking host output
 *     buffers synchronise before returning, pure-device functions do not.
 *
 * Data layout (device)
 *   vector        double[n]                      (DoF order chosen at pattern set-up)
 *   trajectory    double[(Nt+1)*n], level k at [k*n,(k+1)*n)   (helpers.py:563-564)
 *   ELL matrix    double[W*n], slot-major: entry (row i, slot s) at [s*n+i];
 *                 slot 0 is the diagonal; unused slots have column i and value 0
 *   batch         B independent systems: vector b at [b*n], ELL b at [b*W*n],
 *                 trajectory b at [b*(Nt+1)*n]
 */
#ifndef FEMFCT_H
#define FEMFCT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEMFCT_ABI_VERSION 5   /* 5: femfct_build_id, FEMFCT_REGIME_MESH; 4: femfct_patch_walkers; 3: femfct_kernel_regime, femfct_lowop_nonzero_fraction, femfct_chebsi_md, femfct_schnak_*_tw; 2: femfct_schnak_adjoint(alltime), species solver / PDECO / source-term entry points */

typedef struct femfct_ctx femfct_ctx;

/* status codes */
#define FEMFCT_OK                 0
#define FEMFCT_ERR_INVALID        1   /* bad argument / call order (ValueError in the Python mirror) */
#define FEMFCT_ERR_HIP            2   /* HIP runtime error, text in femfct_last_error */
#define FEMFCT_ERR_NOT_CONVERGED  3   /* an iterative solve missed its tolerance */
#define FEMFCT_ERR_NOMEM          4

/* femfct_step_info.flags */
#define FEMFCT_FLAG_MMATRIX_ROWSUM 1  /* some row sum of the low-order matrix <= 0:
                                         the "3: False" diagnostic of helpers.py:1796-1799 */
#define FEMFCT_FLAG_SOLVER_BUDGET  2  /* sweep/iteration budget exhausted before tolerance */
#define FEMFCT_FLAG_COARSE_ITERS   4  /* solver_iters is an upper bound (whole fused launches), not the exact count */
#define FEMFCT_FLAG_CHEBYSHEV      8  /* species solve done by the Chebyshev iteration; solver_iters is the count that
                                         meets tolerance/10 at its asymptotic rate (the next sweep's budget) */
#define FEMFCT_FLAG_ROW_PAIRS      16 /* internal to a trajectory sweep: the pair-compact Jacobi launch met a row with both
                                         entries of an opposing stencil pair; the sweep is repeated with full rows, so
                                         a caller never sees this flag after a successful call */

/* DoF numbering of the structured mesh */
#define FEMFCT_ORDER_VERTEX 0         /* iy*N+ix (dolfin vertex order) */
#define FEMFCT_ORDER_FENICS 1         /* FEniCS CG1 dof order: rank of (ix-iy, iy) */

/* low-order solver selection */
#define FEMFCT_SOLVER_JACOBI   0      /* Jacobi sweeps, device-side convergence test (default; fused multi-sweep kernels) */
#define FEMFCT_SOLVER_BICGSTAB 1      /* Jacobi-preconditioned BiCGStab: for operators far outside the scheme's
                                         dt restriction, where Jacobi would need hundreds of sweeps */

typedef struct femfct_step_info {
    int32_t flags;          /* FEMFCT_FLAG_* */
    int32_t solver_iters;   /* sweeps / iterations used by the low-order solve */
    double  solver_resid;   /* ||b - L x||_inf / ||b||_inf of the iterate before the last sweep */
    double  min_rowsum;     /* min_i sum_j L_ij (helpers.py:1796) */
} femfct_step_info;

/* ------------------------------------------------------------------ context */
int         femfct_abi_version(void);
/* 16 hex digits: sha256 over the library's source files (names + contents) as they were when THIS binary was compiled.
 * Measurement records (bench.py, profiles/traffic.json) are stamped with it.  No reference counterpart. */
const char* femfct_build_id(void);
int         femfct_create(femfct_ctx** ctx, int device_id);
int         femfct_destroy(femfct_ctx* ctx);
const char* femfct_last_error(const femfct_ctx* ctx);
int         femfct_synchronize(femfct_ctx* ctx);
void*       femfct_stream(femfct_ctx* ctx);                 /* hipStream_t */
int         femfct_set_solver(femfct_ctx* ctx, int solver, double rel_tol, int max_iters);
int         femfct_set_graphs(femfct_ctx* ctx, int enable); /* hipGraph replay of the step sequence (default on) */
/* 1 while sweeps are replayed as hipGraphs.  0 after femfct_set_graphs(0), during per-class profiling, and while a
 * rocprofiler-sdk tool (rocprofv3) is attached to the process: its HSA queue interceptor in ROCm 7.2.0 reads a graph
 * launch's packet batch past the end of the queue ring (host SIGSEGV); FEMFCT_PROFILER_GRAPHS=1 overrides.  No reference
 * counterpart (diagnostic). */
/* Which bandwidth-regime kernels the most recent step / sweep enqueued: out[0] Jacobi launch (0 none or another regime,
 * 1 one workgroup per patch, 2 k_strip4_jacobi_walk, 3 k_strip_jacobi_pair_walk), out[1] its walkers, out[2] interior patches
 * per side of the split Chebyshev launch (0: not split), out[3] halo depth.  No reference counterpart (diagnostic). */
int         femfct_launch_info(const femfct_ctx* ctx, int32_t* out4_host);
int         femfct_graph_replay_active(const femfct_ctx* ctx, int* active_host);
/* 1 when the most recent femfct_solidbody_forward / _adjoint evaluated the rotation operator from the node positions
 * (large meshes, operator from femfct_assemble_rotation), 0 when it loaded the stored array.  Diagnostic. */
int         femfct_rotation_derived(const femfct_ctx* ctx, int* derived_host);
/* multi-sweep fusion of the Jacobi / Chebyshev kernels: row strips (any banded pattern) and 2-D tiles
 * (structured mesh in vertex order); both default on, results agree with the one-sweep kernels to the
 * solver tolerance.  For tests and tuning. */
int         femfct_set_fusion(femfct_ctx* ctx, int strips, int tiles);
/* kernel family a step with `batch` members uses on the registered pattern (diagnostic; -1 without a pattern) */
#define FEMFCT_REGIME_ROWS    0   /* one-sweep row kernels (any ELL pattern) */
#define FEMFCT_REGIME_STRIPS  1   /* multi-sweep row strips (banded patterns) */
#define FEMFCT_REGIME_TILE32  2   /* 32 x 32-patch tiles, latency regime (structured mesh, vertex order) */
#define FEMFCT_REGIME_PATCH64 3   /* 64 x 64-patch register/DPP kernels, bandwidth regime (n * batch >= 90 000) */
#define FEMFCT_REGIME_MESH    4   /* one workgroup per trajectory, whole step in one launch (structured mesh, vertex order,
                                     N <= 42 nodes per side, from FEMFCT_MESH_STEP_BATCH trajectories per launch on) */
int         femfct_kernel_regime(const femfct_ctx* ctx, int32_t batch);
/* bandwidth regime: persistent workgroups per batch member of the Jacobi / Chebyshev launches for a solve of `sweeps`
 * sweeps (each walks over its share of the 64 x 64 patches; the Jacobi walkers carry the rows two vertically adjacent
 * patches share in LDS); 0 = one workgroup per patch (fewer than two patches per compute unit, other regimes,
 * FEMFCT_T4_WALK=0).  This is the count of the 1024-thread walkers; the pair-compact Jacobi launch of an upwind
 * operator (512-thread walkers, two per compute unit) uses up to twice as many -- femfct_launch_info reports the launch
 * that actually ran.  Diagnostic. */
int         femfct_patch_walkers(const femfct_ctx* ctx, int32_t batch, int32_t sweeps);
/* share of the off-diagonal entries of the most recent low-order operator L = M_L + dt (A - D + N) that are non-zero
 * (an upwind stencil: about one half for pure convection).  In the bandwidth regime the Jacobi launches neither store
 * nor load the vanishing ones; 1.0 when that shortcut is not in use.  Diagnostic; synchronises. */
int         femfct_lowop_nonzero_fraction(femfct_ctx* ctx, double* fraction_host);

/* Per-kernel timing with HIP events recorded on the ctx stream around every kernel of the step
 * sequence (used by bench.py for the roofline figures).  While enabled, launches are eager.
 * Classes: 0 build_low, 1 jacobi sweep, 2 dudt_rhs, 3 chebyshev step, 4 flux, 5 limit, 6 assembly, 7 other. */
int femfct_set_profiling(femfct_ctx* ctx, int enable);
int femfct_profile_report(femfct_ctx* ctx, double* total_ms_host, int32_t* launches_host, int32_t n_classes);

/* device memory helpers (so that a ctypes-only host needs no other GPU library) */
int femfct_malloc(femfct_ctx* ctx, void** dev_ptr, size_t bytes);
int femfct_free(femfct_ctx* ctx, void* dev_ptr);
int femfct_memcpy_h2d(femfct_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int femfct_memcpy_d2h(femfct_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int femfct_memcpy_d2d(femfct_ctx* ctx, void* dev_dst, const void* dev_src, size_t bytes);
int femfct_memset0(femfct_ctx* ctx, void* dev_ptr, size_t bytes);

/* ----------------------------------------------------------------- pattern
 * Replaces find_node_neighbours (helpers.py:271-307) + the implicit CSR pattern
 * of assemble_sparse (helpers.py:87-104).  The pattern must be structurally
 * symmetric and contain the diagonal. */
int femfct_set_pattern_csr(femfct_ctx* ctx, int32_t n, const int32_t* indptr_host,
                           const int32_t* indices_host);
/* RectangleMesh(Point(a1,a1),Point(a2,a2),n_cells,n_cells) + FunctionSpace(mesh,'CG',1)
 * (advection_solidbody_FCT_PDECO_finaltime.py:63-64); also assembles M, M_lumped and
 * the stiffness matrix Ad on the device (helpers.py:553-555). */
int femfct_set_mesh_square(femfct_ctx* ctx, double a1, double a2, int32_t n_cells, int32_t order);
int32_t femfct_n(const femfct_ctx* ctx);
int32_t femfct_ell_width(const femfct_ctx* ctx);
/* ELL structure back to the host (tests, INTEGRATION): cols[W*n] */
int femfct_get_ell_cols(femfct_ctx* ctx, int32_t* cols_host);

/* CSR values on the registered pattern (host) <-> ELL values (device) */
int femfct_csr_to_ell(femfct_ctx* ctx, const double* csr_vals_host, double* ell_dev);
int femfct_ell_to_csr(femfct_ctx* ctx, const double* ell_dev, double* csr_vals_host);
/* mass matrix M (CSR values on the pattern) and diag of row_lump(M) (helpers.py:309-328)
 * for a generic pattern; not needed after femfct_set_mesh_square */
int femfct_set_mass(femfct_ctx* ctx, const double* M_csr_vals_host, const double* ml_host);
/* device pointers to the ctx-owned constant operators (ELL / vector) */
const double* femfct_mass_ell(const femfct_ctx* ctx);
const double* femfct_stiffness_ell(const femfct_ctx* ctx);
const double* femfct_lumped_mass(const femfct_ctx* ctx);

/* ----------------------------------------------------------- step operator */
/* FCT_alg_ref(A, rhs, u_n, dt, nodes, M, M_lumped, dof_neighbors, non_flux_mat)
 * helpers.py:1715-1872.  All pointers are device pointers; N_ell and rhs may be
 * NULL (= zero).  batch >= 1 independent systems (see layout above); N_shared != 0
 * means one N_ell is shared by the whole batch.  Asynchronous on the ctx stream. */
int femfct_fct_step(femfct_ctx* ctx, const double* A_ell, const double* N_ell, int32_t N_shared,
                    const double* rhs, const double* u_n, double dt, double* u_out, int32_t batch);
/* diagnostics of the most recent femfct_fct_step (synchronises) */
int femfct_last_step_info(femfct_ctx* ctx, femfct_step_info* info_host, int32_t batch);

/* host-buffer convenience form of the same call (CSR values on the pattern;
 * uploads, runs, downloads, synchronises) */
int femfct_fct_step_host(femfct_ctx* ctx, const double* A_csr_vals, const double* N_csr_vals,
                         const double* rhs, const double* u_n, double dt, double* u_out,
                         femfct_step_info* info);

/* ChebSI(vec, M, Md, cheb_iter, lmin, lmax)  helpers.py:143-185 (M = registered mass matrix,
 * Md = its diagonal). */
int femfct_chebsi(femfct_ctx* ctx, const double* b_dev, double* y_dev, int32_t cheb_iter,
                  double lmin, double lmax, int32_t batch);
/* The same with a caller-supplied preconditioner diagonal Md != diag(M) (third argument of ChebSI, helpers.py:143,
 * 181-182: z = r / (Md * (lmin+lmax)/2)); one system per call (batch must be 1), one-sweep row kernels. */
int femfct_chebsi_md(femfct_ctx* ctx, const double* b_dev, double* y_dev, const double* md_dev, int32_t cheb_iter,
                     double lmin, double lmax, int32_t batch);
/* artificial_diffusion_mat(mat)  helpers.py:206-242: D_ell from K_ell (diagonal included) */
int femfct_artificial_diffusion(femfct_ctx* ctx, const double* K_ell, double* D_ell, int32_t batch);
/* y = alpha * Mat * x + beta * y  with an ELL matrix on the registered pattern */
int femfct_spmv(femfct_ctx* ctx, const double* mat_ell, const double* x_dev, double alpha,
                double beta, double* y_dev, int32_t batch);

/* ------------------------------------------------ structured-mesh assembly
 * Device replacements of the dolfin assembly calls on the hot path (helpers.py:87-141). */

/* coordinates of the 6 quadrature points of every triangle, layout
 * [((cy*n_cells+cx)*2 + type)*6 + q], type 0 = (v0,v1,v3), 1 = (v0,v2,v3); host arrays of
 * n_cells*n_cells*2*6 doubles.  The host evaluates its wind (a dolfin Expression in the
 * reference, helpers.py:506-508, 876-878) at these points. */
int femfct_mesh_quad_points(femfct_ctx* ctx, double* xq_host, double* yq_host);
/* scale * assemble_sparse(dot(wind, grad(v))*u*dx)  (helpers.py:581,933,1015;
 * advection_solidbody_FCT_PDECO_finaltime.py:122); wind_q_host[.. *2 + {0,1}] at the points above. */
int femfct_assemble_convection(femfct_ctx* ctx, const double* wind_q_host, double scale, double* A_ell_dev);
/* The same form for the rigid rotation wind = omega * (-x[1], x[0]) (advection_solidbody_FCT_PDECO_finaltime.py:91-93,122
 * with omega = 1/om) in closed form: the wind is linear, so int_K (w.grad v) u dx = grad v . |K|/12 (w_0 + w_1 + w_2 + w_k)
 * -- the integral the quadrature above evaluates, equal to it up to rounding.  An operator assembled here is recognised
 * (bit for bit) by femfct_solidbody_forward / _adjoint, whose large-mesh step kernels then evaluate its rows from the node
 * positions instead of loading them (FEMFCT_GEOM_ROT=0: always load). */
int femfct_assemble_rotation(femfct_ctx* ctx, double omega, double* A_ell_dev);
/* rhs_dk = -(beta*M*c + assemble(p*dot(drift, grad(u))*v*dx)) for `levels` consecutive time levels
 * (advection_solidbody_FCT_PDECO_finaltime.py:228-236); feed to femfct_chebsi(batch=levels). */
int femfct_drift_gradient_rhs(femfct_ctx* ctx, const double* c_dev, const double* u_dev, const double* p_dev,
                              double beta, double bx, double by, double* out_dev, int32_t levels);

/* ------------------------------------------------------ trajectory sweeps
 * Device-resident time loops.  Trajectories are (num_steps+1)*n doubles, level-major;
 * batch member b of a trajectory argument starts at b*(num_steps+1)*n.  c_shared != 0: one
 * control trajectory for the whole batch.  Each call synchronises once at its end (it reads
 * the per-step solver log) and returns FEMFCT_ERR_NOT_CONVERGED if a low-order solve missed
 * the tolerance even with max_iters sweeps. */

/* state sweep of the drift-control problem, u level 0 = initial condition (in place):
 * advection_solidbody_FCT_PDECO_finaltime.py:175-193  (A_u = -eps*Ad + rot_scale*Arot + Adrift1 + Adrift2,
 * control at level n+1, FCT_alg(A_u,...) == FCT_alg_ref(-A_u,...)) */
int femfct_solidbody_forward(femfct_ctx* ctx, const double* Arot_ell, const double* c_traj, int32_t c_shared,
                             double* u_traj, int32_t num_steps, double dt, double eps, double rot_scale,
                             double bx, double by, int32_t batch);
/* the same sweep with a source term: rhs_{n+1} = assemble(src_{n+1}*v*dx) with src a trajectory, e.g. g + c of the
 * linear source-control problems (advection_FCT_PDECO_alltime_exact.py:249-253, A_u = A - eps*Ad: pass the
 * convection matrix as Arot_ell, rot_scale = 1 and a zero control).  src_traj NULL: no source. */
int femfct_solidbody_forward_src(femfct_ctx* ctx, const double* Arot_ell, const double* c_traj, int32_t c_shared,
                                 const double* src_traj, double* u_traj, int32_t num_steps, double dt, double eps,
                                 double rot_scale, double bx, double by, int32_t batch);
/* adjoint sweep: advection_solidbody_FCT_PDECO_finaltime.py:200-221 (alltime = 0: uhat is n doubles
 * per batch member, p(T) = uhat - u(T), zero rhs) and advection_solidbody_FCT_PDECO_alltime.py:232-259
 * (alltime = 1: uhat is a trajectory, p(T) = 0, rhs = assemble((uhat_n - u_n)*v*dx)) */
int femfct_solidbody_adjoint(femfct_ctx* ctx, const double* Arot_ell, const double* c_traj, int32_t c_shared,
                             const double* u_traj, const double* uhat, double* p_traj, int32_t num_steps,
                             double dt, double eps, double rot_scale, double bx, double by, int32_t alltime,
                             int32_t batch);
/* solver diagnostics of the most recent sweep: info_host[step*batch + b] */
int femfct_traj_info(femfct_ctx* ctx, femfct_step_info* info_host, int32_t num_steps, int32_t batch);

/* ------------------------------------------------------ optimisation layer
 * Reductions/updates of the projected-gradient loop kept on the device (only scalars
 * cross PCIe).  M is the registered mass matrix; results are written to host memory
 * (these calls synchronise).  batch member b of a trajectory starts at b*(num_steps+1)*n. */

/* L2_norm_sq_Q(phi, num_steps, dt, M)  helpers.py:330-360, phi = a - b (b may be NULL) */
int femfct_l2_norm_sq_Q(femfct_ctx* ctx, const double* a_dev, const double* b_dev, int32_t num_steps,
                        double dt, double* out_host, int32_t batch);
/* L2_norm_sq_Omega(phi, M)  helpers.py:362-381, phi = a - b, one level of n doubles per batch member */
int femfct_l2_norm_sq_Omega(femfct_ctx* ctx, const double* a_dev, const double* b_dev, double* out_host,
                            int32_t batch);
/* cost_functional(var1, var1_target, projected_control, num_steps, dt, M, beta, optim, var2, var2_target)
 * helpers.py:383-441; finaltime = 0: optim == "alltime" (targets are trajectories),
 * finaltime = 1: optim == "finaltime" (targets are n doubles per batch member). */
int femfct_cost_functional(femfct_ctx* ctx, const double* var1, const double* var1_target,
                           const double* control, int32_t control_shared, int32_t num_steps, double dt,
                           double beta, int32_t finaltime, const double* var2, const double* var2_target,
                           double* J_host, int32_t batch);
/* update_control: out = clip(c + s*d, c_lower, c_upper)  helpers.py:1666-1667 (out may alias c) */
int femfct_project_control(femfct_ctx* ctx, const double* c_dev, double s, const double* d_dev,
                           double c_lower, double c_upper, double* out_dev, int64_t count);

/* descent direction of the pointwise-gradient problems, d = -(beta*c - t) with t = x*y/divisor (y given)
 * or t = scale*x (y NULL): nonlinear_FCT_PDECO_refactored.py:148, Schnak_FCT_PDECO_refactored.py:167,
 * chemotaxis_FCT_PDECO_AT_refactored.py:158 (same floating-point operation order) */
int femfct_descent_pointwise(femfct_ctx* ctx, int64_t count, double beta, const double* c_dev, double scale,
                             const double* x_dev, const double* y_dev, double divisor, double* out_dev);

/* ------------------------------------------------ non-FCT species and the three PDE systems */

/* out = in^T on the registered pattern: assemble_sparse(dot(wind,grad(u))*w*dx) (helpers.py:681) is the
 * transpose of assemble_sparse(dot(wind,grad(w))*u*dx) (helpers.py:581) */
int femfct_ell_transpose(femfct_ctx* ctx, const double* in_ell, double* out_ell);
/* out = alpha*a + beta*b over count doubles (b may be NULL): Du*Ad - omega1*A etc. (helpers.py:583) */
int femfct_axpby(femfct_ctx* ctx, int64_t count, double alpha, const double* a_dev, double beta,
                 const double* b_dev, double* out_dev);
/* tolerance / iteration cap of the iterative solver used for the non-FCT implicit solves */
int femfct_set_krylov(femfct_ctx* ctx, double rel_tol, int32_t max_iters);
/* solver of those solves inside the trajectory sweeps: FEMFCT_SPECIES_AUTO = tile-fused Chebyshev
 * iteration on the structured vertex-order mesh (falls back to BiCGStab when it does not contract),
 * FEMFCT_SPECIES_BICGSTAB = always BiCGStab.  Both meet the same residual tolerance. */
#define FEMFCT_SPECIES_AUTO 0
#define FEMFCT_SPECIES_BICGSTAB 1
int femfct_set_species_solver(femfct_ctx* ctx, int32_t mode);
/* spsolve(Mat, b) (helpers.py:596,686,1342,1538): Jacobi-preconditioned BiCGStab from the initial guess
 * x0; mat_shared != 0: one matrix for the whole batch.  Synchronises. */
int femfct_bicgstab(femfct_ctx* ctx, const double* mat_ell, int32_t mat_shared, const double* b_dev,
                    const double* x0_dev, double* x_dev, int32_t batch, femfct_step_info* info_host);

/* solve_nonlinear_equation (helpers.py:881-966).  Aw_ell = assemble_sparse(dot(wind,grad(v))*u*dx);
 * c_level = the control level the reference uses for the whole sweep (level 1, helpers.py:950-951; or
 * the constant control_fun), n doubles per batch member; u_traj level 0 = initial condition. */
int femfct_nonlinear_forward(femfct_ctx* ctx, const double* Aw_ell, const double* c_level, double* u_traj,
                             int32_t num_steps, double dt, double eps, int32_t batch);
/* solve_adjoint_nonlinear_equation (helpers.py:968-1038); uhat_T: n doubles per batch member */
int femfct_nonlinear_adjoint(femfct_ctx* ctx, const double* Aw_ell, const double* u_traj,
                             const double* uhat_T, double* p_traj, int32_t num_steps, double dt, double eps,
                             int32_t batch);
/* solve_schnak_system (helpers.py:511-597); par = {Du, Dv, c_b, gamma, omega1, omega2} */
int femfct_schnak_forward(femfct_ctx* ctx, const double* Aw_ell, const double* c_level, double* u_traj,
                          double* v_traj, int32_t num_steps, double dt, const double* par, double rescaling,
                          int32_t batch);
/* solve_adjoint_schnak_system (helpers.py:599-698); AwT_ell = femfct_ell_transpose(Aw_ell).
 * alltime = 0: the reference's final-time problem (uhat_T, vhat_T: n doubles per batch member).
 * alltime = 1: all-time misfit (targets are trajectories, p(T) = q(T) = 0, misfit loads in both equations as
 * in the inline loop of Schnak_FCT_PDECO_alltime.py:204-284; helpers.py has no such variant). */
int femfct_schnak_adjoint(femfct_ctx* ctx, const double* AwT_ell, const double* u_traj, const double* v_traj,
                          const double* uhat_T, const double* vhat_T, double* p_traj, double* q_traj,
                          int32_t num_steps, double dt, const double* par, int32_t alltime, int32_t batch);
/* The two Schnakenberg sweeps with a separable time-dependent wind w(x, t) = s(t) w0(x) -- the set-up of the script
 * BASELINE config 3 names, Schnak_FCT_PDECO_alltime.py:55,174-175,228-230 (rotation * sin(2 pi t), convection matrix
 * re-assembled every step; helpers.py:565-566, 664, 679 set wind.t the same way).  Aw_ell / AwT_ell are the matrices
 * of w0; wind_scale_host[k] = s(t_k), k = 0..num_steps (host array, copied).  The step to level n+1 (forward) uses
 * s(t_{n+1}); the step to level n (adjoint) uses s(t_n).  wind_scale_host == NULL: stationary wind (= the calls above). */
int femfct_schnak_forward_tw(femfct_ctx* ctx, const double* Aw_ell, const double* wind_scale_host, const double* c_level,
                             double* u_traj, double* v_traj, int32_t num_steps, double dt, const double* par,
                             double rescaling, int32_t batch);
int femfct_schnak_adjoint_tw(femfct_ctx* ctx, const double* AwT_ell, const double* wind_scale_host, const double* u_traj,
                             const double* v_traj, const double* uhat_T, const double* vhat_T, double* p_traj,
                             double* q_traj, int32_t num_steps, double dt, const double* par, int32_t alltime,
                             int32_t batch);
/* solve_chtxs_system (helpers.py:1250-1385, non-generation mode); par = {delta, Dm, Df, chi, eta} */
int femfct_chtxs_forward(femfct_ctx* ctx, const double* c_level, double* u_traj, double* v_traj,
                         int32_t num_steps, double dt, const double* par, double rescaling, int32_t batch);
/* solve_adjoint_chtxs_system (helpers.py:1387-1581); alltime = 0: optim "finaltime" (uhat/vhat n doubles
 * per member), 1: optim "alltime" (uhat/vhat trajectories; raw nodal misfits as in helpers.py:1506-1507,
 * 1533-1534; level num_steps of p/q is taken as passed) */
int femfct_chtxs_adjoint(femfct_ctx* ctx, const double* u_traj, const double* v_traj, const double* uhat,
                         const double* vhat, double* p_traj, double* q_traj, const double* c_traj,
                         int32_t num_steps, double dt, const double* par, double rescaling, int32_t alltime,
                         int32_t batch);
/* BiCGStab diagnostics of the most recent sweep that used it: info_host[step*batch + b] */
int femfct_traj_krylov_info(femfct_ctx* ctx, femfct_step_info* info_host, int32_t num_steps, int32_t batch);

#ifdef __cplusplus
}
#endif
#endif /* FEMFCT_H */
