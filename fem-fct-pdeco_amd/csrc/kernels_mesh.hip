// One workgroup = one trajectory: the whole FEM-FCT step of a small mesh (N <= 42 nodes per side: the 41 x 41 meshes of
// configs 3 and 4; N = 81: config C2, large batches) in ONE launch.
//   FCT_alg_ref                 /root/reference/helpers.py:1715-1872
//   artificial_diffusion_mat    /root/reference/helpers.py:206-242   (build phase)
//   spsolve(L, b)               /root/reference/helpers.py:1782      (solve phase, stops by itself)
//   ChebSI                      /root/reference/helpers.py:143-185   (20 iterations on the mesh's own mass matrix)
//   Zalesak limiter             /root/reference/helpers.py:1818-1870 (flux phase)
//
// Why: on a 41 x 41 mesh a step is ~50 sweeps over 1681 nodes.  The tile path runs them as four to seven dependent
// launches; with B trajectories per launch (the Armijo trials of helpers.py:1583-1713, the beta copies of
// advection_solidbody_FCT_PDECO_alltime.py:43-74) those launches reach 2-27 % of the HBM roofline.  Here B trajectories
// are B workgroups that never wait for each other: the mesh lives in one CU -- matrix rows in registers, the iterate in
// registers + an LDS image for the neighbours -- and __syncthreads() is the only synchronisation.  HBM traffic per step
// and trajectory: A once (2 x 2 blocks without a non-flux matrix keep its rows in registers from the build to du/dt; the
// other variants read it a second time there), the forward half of D out and in, u_n / rhs in, u_{n+1} out.
//
// Layout.  A thread owns a 2 x 2 block of nodes (21 x 21 = 441 threads at 41^2).  In-block neighbours are the thread's
// own registers; values on the block's rim are published in an LDS image (pad columns and pad rows around the mesh:
// nodes outside it read 0; on the 41 x 41 config mesh the columns are stored de-interleaved, even ones then odd ones, so
// that a wave's accesses are unit-stride -- see xo) and the 2 BX + 2 BY + 2 rim values of the neighbouring blocks are
// read back after the barrier.  Two images alternate, so a sweep needs one barrier.  Blocks that do not touch the mesh
// boundary are dealt to the first waves, the boundary ring to the last ones: the interior waves run code without
// existence masks, with constant mass-matrix weights.
//
// Every edge quantity that is symmetric or antisymmetric to the bit is computed ONCE, by the node that sees the edge
// in a forward slot (E, NE, N), and handed to the other end through LDS: d_ij = max(0, a_ij, a_ji), the raw flux
// f_ij = -f_ji, the limited flux alpha_ij f_ij.  So only the forward half of D exists (in HBM scratch between build
// and flux phase) and the limiter reads three neighbours' R+- instead of six.
//
// Low-order solve: x <- D^-1 (b - O x) with the rows scaled by 1 / l_ii, Gauss-Seidel inside a thread's block (the
// in-block neighbours' new values are at hand), Jacobi across blocks; it stops when max_i l_ii |x_i' - x_i| <= tol ||b||
// (the residual test of the tile path).
//
// 81 x 81 (config C2): the same kernel with 3 x 3 blocks on 27 x 27 = 729 of 768 threads ("LEAN": the scaled right-hand
// side and 1 / l_ii live in the third LDS image during the solve, the stopping test uses max_j l_jj, the spare rim rows
// of the build are parked in image 1).  It holds all 54 off-diagonal coefficients of a block in registers and still
// needs 496 B of scratch per thread; per step it costs 116 us alone and 151 / 234 us for 64 / 256 trajectories (tiles:
// 46 / 159 / 550), so the host picks it from 64 trajectories per launch on.  A variant with one value per opposing pair
// of L (half the registers) was slower (135-150 us) -- DESIGN.md section 8, tools/mesh_pair_experiment.hip.
#include "femfct_internal.h"
#include "device_utils.h"
#include "forms.h"
#include "step_end.h"
#include "stencil.h"

#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <type_traits>

struct MeshStepArgs {
    int n, N;               // nodes, nodes per side
    int TX, TY;             // blocks per side
    int nint, nint_pad;     // interior blocks, rounded up to whole waves
    double h, dt;
    MatRef A;               // flux matrix of this step, ELL slot-major [7][n]
    const double* Nm;       // non-flux matrix (helpers.py:1775: L += dt N) or null
    int64_t nm_bs;          // its batch stride (0: shared)
    VecRef rhs; int64_t rhs_bs;
    VecRef u_n; int64_t u_bs;
    VecRef out; int64_t out_bs;
    const double* ml;       // lumped mass [n]
    const double* Md;       // diagonal of the consistent mass matrix [n]
    double* Dh;             // scratch [batch][7][n]: slots 1..3 hold the forward half of D
    StepCtl* ctl;
    double rel_tol;
    int max_sweeps;
    int dbg_sweeps, dbg_cheb;   // FEMFCT_TUNING builds (FEMFCT_MESH_DBG=sweeps,cheb): forced counts, results void; else 0, 20
    double om[20];          // Chebyshev weights (helpers.py:170-179)
    EndArgs e;
    unsigned long long* trace;   // FEMFCT_TUNING builds: phase timestamps of workgroup 0 (100 MHz), else null
};

namespace {

constexpr int ms_dx(int s) { return s == 1 || s == 2 ? 1 : (s == 4 || s == 5 ? -1 : 0); }
constexpr int ms_dy(int s) { return s == 2 || s == 3 ? 1 : (s == 5 || s == 6 ? -1 : 0); }

#define MS_UNROLL _Pragma("unroll")
// Scheduling fence between the nodes of a thread in the once-per-step phases (left alone, the scheduler interleaves
// all nodes' reciprocal / limiter chains and the live values pile up) and between the block rows of a sweep.
#define MS_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef FEMFCT_TUNING
#define MS_STAMP(i) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MS_STAMP(i) do { } while (0)
#endif

// keep ? v : +0.0 with v made opaque first: hipcc otherwise sinks the load (or the whole chain) that produced v into a
// branch of its own around the select -- nine little basic blocks per phase instead of straight-line code
__device__ __forceinline__ double zsel(bool keep, double v) {
    asm volatile("" : "+v"(v));
    return keep ? v : 0.0;
}

// LDS image of a nodal field: rows -1 .. N (zero pad rows), columns -1 .. N.
// DEINT (2 x 2 blocks on the config mesh, N a compile-time constant): the columns are stored DE-INTERLEAVED -- column x at (u >> 1) + (u & 1) * half, u = x + 1, half = (N + 3) / 2 --
// so that the 64 threads of a wave, whose blocks are two columns apart, touch consecutive doubles with every access
// (interleaved, their 16-byte stride uses every other LDS bank pair: two-way conflicts on each of the ~14 reads and
// writes a sweep makes).  Relative to the slot of column ix0 - 1 the thread's columns ix0 - 1 .. ix0 + 2 are at
// 0, half, 1, half + 1 (immediates).  The other instantiations keep the plain row, pitch N + 1 with one shared pad
// column: 3 x 3 blocks are conflict-free as they are (stride 24 bytes), and with N at run time `half` would cost the
// 2 x 2 variant with a non-flux matrix the registers it does not have.
template <bool DEINT>
__host__ __device__ constexpr int mesh_img_doubles(int NMAX) {
    return DEINT ? (NMAX + 2) * (2 * ((NMAX + 3) / 2)) : (NMAX + 2) * (NMAX + 1) + 1;
}
template <bool DEINT>
__device__ __forceinline__ int xo(int cc, int half) {
    return DEINT ? ((cc + 1) >> 1) + (((cc + 1) & 1) ? half : 0) : cc;
}

// value of `val` at node (r + DY, c + DX): the thread's own register inside the block, else the image
template <int BX, int BY, int DX, int DY>
__device__ __forceinline__ double nb_get(const double (&val)[BY][BX], double* const (&rowp)[BY + 2], int img, int r, int c, int half) {
    const int rr = r + DY, cc = c + DX;
    if (rr >= 0 && rr < BY && cc >= 0 && cc < BX) return val[rr][cc];
    return half ? rowp[rr + 1][img + xo<true>(cc, half)] : rowp[rr + 1][img + cc];      // (half: 0 = plain rows; a constant at every call)
}

// publish `val` at the nodes that a neighbouring block reads with nb_get<DX, DY>
// (stores are unconditional: a node outside the mesh -- N not a multiple of the block size -- writes to a trash word
// instead, chosen by an address select; a guarded store would be a branch of its own in the instruction stream)
template <int BX, int BY, int DX, int DY, class Dst>
__device__ __forceinline__ void publish(const double (&val)[BY][BX], int img, Dst dst) {
    MS_UNROLL for (int r = 0; r < BY; ++r) {
        MS_UNROLL for (int c = 0; c < BX; ++c) {
            const int rr = r - DY, cc = c - DX;
            if (rr >= 0 && rr < BY && cc >= 0 && cc < BX) continue;
            *dst(r, c, img) = val[r][c];
        }
    }
}

// publish the block's rim (every node another block can see)
template <int BX, int BY, class Dst>
__device__ __forceinline__ void publish_rim(const double (&val)[BY][BX], int img, Dst dst) {
    MS_UNROLL for (int r = 0; r < BY; ++r) {
        MS_UNROLL for (int c = 0; c < BX; ++c) {
            if (r > 0 && r < BY - 1 && c > 0 && c < BX - 1) continue;
            *dst(r, c, img) = val[r][c];
        }
    }
}

// (volatile: hipcc otherwise pairs neighbouring reads into ds_read2_b64, which gfx950 serves at half the rate of two
// ds_read_b64 -- MI355X_MICROARCH.md, LDS table: 8 LDS cycles against 2 + 2)
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ double lds_read(const double* p) { return *(volatile lds_double*)p; }
template <int BX, int BY>
__device__ __forceinline__ void gather_halo(double (&v)[BY + 2][BX + 2], double* const (&rowp)[BY + 2], int img, int half) {
    auto X = [&](int cc) { return half ? xo<true>(cc, half) : cc; };
    MS_UNROLL for (int c = -1; c < BX; ++c) v[0][c + 1] = lds_read(&rowp[0][img + X(c)]);
    MS_UNROLL for (int c = 0; c <= BX; ++c) v[BY + 1][c + 1] = lds_read(&rowp[BY + 1][img + X(c)]);
    MS_UNROLL for (int r = 0; r < BY; ++r) {
        v[r + 1][0] = lds_read(&rowp[r + 1][img + X(-1)]);
        v[r + 1][BX + 1] = lds_read(&rowp[r + 1][img + X(BX)]);
    }
}

// neighbour (rr, cc) of a node of the block: the thread's own array inside the block, the gathered halo outside
template <int BX, int BY>
__device__ __forceinline__ double nbv(const double (&own)[BY][BX], const double (&v)[BY + 2][BX + 2], int rr, int cc) {
    if (rr >= 0 && rr < BY && cc >= 0 && cc < BX) return own[rr][cc];
    return v[rr + 1][cc + 1];
}

// 1 / v to about an ulp: v_rcp_f64 + two Newton steps (the IEEE division sequence costs three times as much, and the
// fixed part of a step holds five of them per node)
__device__ __forceinline__ double frcp(double v) {
    double r = __builtin_amdgcn_rcp(v);
    r = fma(r, fma(-v, r, 1.0), r);
    r = fma(r, fma(-v, r, 1.0), r);
    return r;
}

// a value every lane holds alike, moved to scalar registers (the reductions' results live across the whole step)
__device__ __forceinline__ double uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// four reductions (max, min, max, max) behind one barrier pair; every thread gets the results
template <int NW>
__device__ __forceinline__ void reduce4(double& a, double& b, double& c, double& d, double* red) {
    a = wave_reduce(a, OpMax());
    b = wave_reduce(b, OpMin());
    c = wave_reduce(c, OpMax());
    d = wave_reduce(d, OpMax());
    const int wid = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    __syncthreads();
    if (lane == 0) { red[wid] = a; red[NW + wid] = b; red[2 * NW + wid] = c; red[3 * NW + wid] = d; }
    __syncthreads();
    double ra = 0.0, rb = INFINITY, rc = 0.0, rd = 0.0;
    MS_UNROLL for (int w = 0; w < NW; ++w) {
        ra = fmax(ra, red[w]); rb = fmin(rb, red[NW + w]); rc = fmax(rc, red[2 * NW + w]); rd = fmax(rd, red[3 * NW + w]);
    }
    a = uniform(ra); b = uniform(rb); c = uniform(rc); d = uniform(rd);
}

template <int BX, int BY, int NMAX, int NT, int NFIX, bool HASNM, bool INTERIOR>
__device__ __forceinline__ void mesh_step_body(const MeshStepArgs& a, double* lds, double* red, int* flg, int bx, int by,
                                               int bz, int ord) {
    constexpr bool DEINT = BX == 2 && NFIX != 0;      // de-interleaved image columns (see xo)
    constexpr int IMG = mesh_img_doubles<DEINT>(NMAX);      // doubles per image
    constexpr int NW = NT / WAVE;
    // NFIX: the config mesh (41 nodes per side) gets its own instantiation: every LDS / HBM row offset is then an immediate
    const int N = NFIX ? NFIX : a.N, n = NFIX ? NFIX * NFIX : a.n;
    constexpr int half = DEINT ? (NFIX + 3) / 2 : 0;
    const int P = DEINT ? 2 * half : N + 1;
    const int ix0 = bx * BX, iy0 = by * BY;
    const double dt = a.dt;
    double* rowp[BY + 2];      // rowp[r + 1][img + xo(c)]: node (r, c) of the block; DEINT: based at column ix0 - 1
    MS_UNROLL for (int r = -1; r <= BY; ++r) rowp[r + 1] = DEINT ? lds + (iy0 + r + 1) * P + bx : lds + 1 + (iy0 + r + 1) * P + ix0;
    bool colok[BX], rowok[BY];
    MS_UNROLL for (int c = 0; c < BX; ++c) colok[c] = INTERIOR || (ix0 + c < N);
    MS_UNROLL for (int r = 0; r < BY; ++r) rowok[r] = INTERIOR || (iy0 + r < N);
    // Threads beyond the blocks of their kind repeat block (1, 1) resp. (0, 0): the same inputs through the same code give
    // the same bits, so their (unconditional) stores are harmless duplicates -- as long as no address is read back after
    // being overwritten with something else (the LEAN parking below keeps to that).
    // ALLV: every node of every block lies in the mesh (N a multiple of the block size)
    constexpr bool ALLV = INTERIOR || (NFIX != 0 && NFIX % BX == 0 && NFIX % BY == 0);
    auto valid = [&](int r, int c) { return ALLV ? true : (rowok[r] && colok[c]); };
    auto vz = [&](int r, int c, double val) { return ALLV ? val : zsel(valid(r, c), val); };      // 0 outside the mesh
    int gi0 = iy0 * N + ix0;
    auto gidx = [&](int r, int c) { return ALLV ? gi0 + r * N + c : (valid(r, c) ? gi0 + r * N + c : 0); };   // safe to load from
    double* const trash = reinterpret_cast<double*>(flg + 4);
    auto dst = [&](int r, int c, int img) -> double* {
        double* p = &rowp[r + 1][img + xo<DEINT>(c, half)];
        return ALLV ? p : (valid(r, c) ? p : trash);
    };
    // interior nodes: M_L = h^2, m_ii = h^2 / 2, m_ij = h^2 / 12 (six triangles around the node)
    const double hh = a.h * a.h;

    const double* Ab = mat_ptr(a.A, bz);
    // HASNM: with a non-flux matrix (its own instantiation: a run-time test inside the unrolled node loops would be a
    // branch per block row, and the values loaded before it are spilled across it)
    const double* Nm = HASNM ? a.Nm + bz * a.nm_bs : nullptr;
    const double* un = vec_ptr(a.u_n) + bz * a.u_bs;
    const double* rhs = vec_ptr(a.rhs) + bz * a.rhs_bs;      // (never null: the launcher passes a vector of zeros for 'no rhs')
    double* out = const_cast<double*>(vec_ptr(a.out)) + bz * a.out_bs;
    double* Dh = a.Dh + (int64_t)bz * 7 * n;      // (slot 0 is unused: Dh[0] takes the stores of nodes outside the mesh)

    // ------------------------------------------------------------------ build: D, L = M_L + dt (A - D + N), b
    // LEAN (3 x 3 blocks, 81 x 81 nodes: three waves per SIMD, 168 registers for nine nodes): only the six scaled
    // off-diagonals and the iterate stay in registers through the solve; b / l_ii lives in image 2 (idle during the solve),
    // and the residual test uses max_j l_jj for every node (an upper bound of the residual: it can only ask for a sweep
    // more).  With 2 x 2 blocks everything fits and the test is the exact one.
    constexpr bool LEAN = BX * BY > 4;
    double x[BY][BX], bp[BY][BX], lc[6][BY][BX], ldv[BY][BX];
    double bmax = 0.0, rsmin = INFINITY, ldmax = 0.0;
    // KEEP (2 x 2 blocks without a non-flux matrix: 146 of the 256 registers a 512-thread workgroup may have): every HBM
    // operand of the step -- the seven entries of each row of A, u_n, rhs, and on the boundary ring M_L and diag(M) -- is
    // requested ONCE, here, ahead of the first barrier, and stays in registers: the pair loop, the du/dt phase and the
    // limiter then wait for no memory round trip (three, two and one of them before).
    constexpr bool KEEP = !HASNM && BX * BY <= 4;
    double ak[7][BY][BX], uk[BY][BX], rk[BY][BX], mlk[BY][BX], mdk[BY][BX];
    if (KEEP) {
        MS_UNROLL for (int r = 0; r < BY; ++r)
            MS_UNROLL for (int c = 0; c < BX; ++c) {
                const int i = gidx(r, c);
                MS_UNROLL for (int s = 0; s < 7; ++s) ak[s][r][c] = (Ab + (int64_t)s * n)[i];
                uk[r][c] = un[i];
                rk[r][c] = rhs[i];
                mlk[r][c] = hh; mdk[r][c] = 0.5 * hh;
                if (!INTERIOR) { mlk[r][c] = a.ml[i]; mdk[r][c] = a.Md[i]; }
            }
    }
    double nrow[BY][BX];          // HASNM: row sums of the non-flux matrix (for the row-sum diagnostic, with du/dt)
    MS_UNROLL for (int r = 0; r < BY; ++r)
        MS_UNROLL for (int c = 0; c < BX; ++c) nrow[r][c] = 0.0;
    {
        double dsum[BY][BX];
        MS_UNROLL for (int r = 0; r < BY; ++r)
            MS_UNROLL for (int c = 0; c < BX; ++c) dsum[r][c] = 0.0;
        MS_UNROLL for (int p = 0; p < 3; ++p) {
            const int s = p + 1, o = p + 4;
            double as[BY][BX], ao[BY][BX], ds[BY][BX];
            MS_UNROLL for (int r = 0; r < BY; ++r)
                MS_UNROLL for (int c = 0; c < BX; ++c) {
                    // (uniform base + one 32-bit offset per thread: the slot and the node are immediates)
                    const double t1 = KEEP ? ak[s][r][c] : (Ab + (int64_t)s * n)[gidx(r, c)],
                                 t2 = KEEP ? ak[o][r][c] : (Ab + (int64_t)o * n)[gidx(r, c)];
                    as[r][c] = vz(r, c, t1);
                    ao[r][c] = vz(r, c, t2);
                }
            if (p == 0) {
                // (the first loads are in flight while the images are cleared: nodes outside the mesh read 0 from them)
                for (int kz = threadIdx.x; kz < 3 * IMG; kz += NT) lds[kz] = 0.0;
                if (threadIdx.x < 4) flg[threadIdx.x] = 0;         // three residual flags
                __syncthreads();
                MS_STAMP(1);
            }
            // a_ji of the forward edge: the neighbour's entry in the opposite slot
            if (p == 0) publish<BX, BY, 1, 0>(ao, 0, dst);
            if (p == 1) publish<BX, BY, 1, 1>(ao, 0, dst);
            if (p == 2) publish<BX, BY, 0, 1>(ao, 0, dst);
            __syncthreads();
            MS_UNROLL for (int r = 0; r < BY; ++r)
                MS_UNROLL for (int c = 0; c < BX; ++c) {
                    const double at = p == 0 ? nb_get<BX, BY, 1, 0>(ao, rowp, 0, r, c, half)
                                    : p == 1 ? nb_get<BX, BY, 1, 1>(ao, rowp, 0, r, c, half)
                                             : nb_get<BX, BY, 0, 1>(ao, rowp, 0, r, c, half);
                    ds[r][c] = fmax(0.0, fmax(as[r][c], at));            // d_ij = max(0, a_ij, a_ji), once per edge
                    (Dh + (ALLV || valid(r, c) ? (int64_t)s * n : 0))[gidx(r, c)] = ds[r][c];
                    dsum[r][c] += ds[r][c];
                    lc[p][r][c] = dt * (as[r][c] - ds[r][c]);            // (before the barrier: a_ij and d_ij need not live across it)
                }
            // ... and back to the other end of the edge, which sees it in the backward slot
            if (p == 0) publish<BX, BY, -1, 0>(ds, IMG, dst);
            if (p == 1) publish<BX, BY, -1, -1>(ds, IMG, dst);
            if (p == 2) publish<BX, BY, 0, -1>(ds, IMG, dst);
            __syncthreads();
            MS_UNROLL for (int r = 0; r < BY; ++r)
                MS_UNROLL for (int c = 0; c < BX; ++c) {
                    const double dbk = p == 0 ? nb_get<BX, BY, -1, 0>(ds, rowp, IMG, r, c, half)
                                     : p == 1 ? nb_get<BX, BY, -1, -1>(ds, rowp, IMG, r, c, half)
                                              : nb_get<BX, BY, 0, -1>(ds, rowp, IMG, r, c, half);
                    dsum[r][c] += dbk;
                    double lb = dt * (ao[r][c] - dbk);
                    if (HASNM) {
                        const double n1 = (Nm + (int64_t)s * n)[gidx(r, c)], n2 = (Nm + (int64_t)o * n)[gidx(r, c)];
                        lc[p][r][c] += vz(r, c, dt * n1);
                        lb += vz(r, c, dt * n2);
                        nrow[r][c] += n1 + n2;
                    }
                    lc[3 + p][r][c] = lb;
                }
        }
        MS_STAMP(2);
        if (LEAN) __syncthreads();      // (image 1 still holds the last pair's d_ij for the neighbours)
        if (LEAN) {        // (the sums wait in image 1 -- idle until the first sweep -- while the block rows are scaled one at a time;
                           //  not in image 2, where b / l_ii goes: no address is ever read back after a different value went there)
            MS_UNROLL for (int r = 0; r < BY; ++r)
                MS_UNROLL for (int c = 0; c < BX; ++c) *dst(r, c, IMG) = dsum[r][c];
        }
        // (a block row's loads first, then its nodes one by one)
        MS_UNROLL for (int r = 0; r < BY; ++r) {
            double a0v[BX], uv[BX], rb[BX], mlv0[BX], nmv[BX], dsv[BX];
            MS_UNROLL for (int c = 0; c < BX; ++c) {
                const int i = gidx(r, c);
                a0v[c] = KEEP ? ak[0][r][c] : Ab[i]; uv[c] = KEEP ? uk[r][c] : un[i];
                rb[c] = 0.0; nmv[c] = 0.0; mlv0[c] = hh;
                if (!INTERIOR) mlv0[c] = KEEP ? mlk[r][c] : a.ml[i];
                dsv[c] = LEAN ? lds_read(&rowp[r + 1][IMG + xo<DEINT>(c, half)]) : dsum[r][c];
            }
            MS_UNROLL for (int c = 0; c < BX; ++c) rb[c] = KEEP ? rk[r][c] : rhs[gidx(r, c)];
            if (HASNM) {
                MS_UNROLL for (int c = 0; c < BX; ++c) nmv[c] = Nm[gidx(r, c)];
            }
            MS_FENCE();
            MS_UNROLL for (int c = 0; c < BX; ++c) {
                const bool vv = valid(r, c);
                const double mli = INTERIOR ? hh : (ALLV ? mlv0[c] : zsel(vv, mlv0[c] - 1.0) + 1.0);
                const double a0 = vz(r, c, a0v[c]), ui = vz(r, c, uv[c]);
                const double ld = mli + dt * (a0 + dsv[c]) + vz(r, c, dt * nmv[c]);
                if (HASNM) nrow[r][c] += nmv[c];
                const double bi = mli * ui + vz(r, c, dt * rb[c]);
                bmax = fmax(bmax, fabs(bi));
                const double rinv = frcp(ld);
                MS_UNROLL for (int q = 0; q < 6; ++q) lc[q][r][c] *= rinv;
                if (LEAN) {
                    *dst(r, c, 2 * IMG) = bi * rinv;
                    ldmax = fmax(ldmax, ld);
                } else {
                    bp[r][c] = bi * rinv;
                    ldv[r][c] = ld;
                }
                x[r][c] = ui;
                MS_FENCE();
            }
        }
    }
    {
        double t3 = 0.0;
        reduce4<NW>(bmax, rsmin, ldmax, t3, red);      // (rsmin: still the identity; the row sums come with du/dt)
    }
    // Residual test as on the tile path: l_ii |x_i' - x_i| = |b_i - (L x)_i| for a Jacobi update (in-block Gauss-Seidel
    // updates make it the residual with the block's earlier nodes already renewed), against tol ||b||_inf.
    // LEAN: |x_i' - x_i| against tol ||b|| / max_j l_jj.
    const double tolb = uniform(LEAN ? a.rel_tol * bmax / ldmax : a.rel_tol * bmax);
    MS_STAMP(3);
    int iters = 0, sflags = 0;
    double resid = 0.0;
    {
        // -------------------------------------------------------------- low-order solve
        publish_rim<BX, BY>(x, 0, dst);
        __syncthreads();
        double v[BY + 2][BX + 2];         // halo of the current iterate (its rim is in image `cur`)
        double rmx = 0.0;
        // Inside the block a node takes the values its in-block neighbours have just got (block Gauss-Seidel, never slower
        // than Jacobi for an M-matrix: the iterate is updated in place); successive sweeps run through the block in
        // opposite orders so that every wind direction is followed by one of them.  Across blocks it stays Jacobi: nothing
        // crosses a barrier early.
        // The halo is read block row by block row, just ahead of the row that needs it (a fence after every row keeps the
        // scheduler from pulling all the reads to the top).  Stage for row r, rows ascending: its W value, the E value of
        // row r + 1 (NE of the last column), with the first row the whole bottom halo row and its own E value, with the
        // last row the top halo row.  Rows descending: mirrored.
        auto stage = [&](const int img_in, const int r, auto revtag) {
            constexpr bool rev = decltype(revtag)::value;
            auto ld = [&](int hr, int hc) { v[hr][hc] = lds_read(&rowp[hr][img_in + xo<DEINT>(hc - 1, half)]); };    // v[hr][hc] = node (hr - 1, hc - 1)
            if (!rev) {
                ld(r + 1, 0);
                if (r + 2 <= BY) ld(r + 2, BX + 1);
                if (r == 0) {
                    MS_UNROLL for (int hc = 0; hc <= BX; ++hc) ld(0, hc);
                    ld(1, BX + 1);
                }
                if (r == BY - 1) {
                    MS_UNROLL for (int hc = 1; hc <= BX + 1; ++hc) ld(BY + 1, hc);
                }
            } else {
                ld(r + 1, BX + 1);
                if (r >= 1) ld(r, 0);
                if (r == BY - 1) {
                    MS_UNROLL for (int hc = 1; hc <= BX + 1; ++hc) ld(BY + 1, hc);
                    ld(BY, 0);
                }
                if (r == 0) {
                    MS_UNROLL for (int hc = 0; hc <= BX; ++hc) ld(0, hc);
                }
            }
        };
        // Sweep k raises flag k % 3 when its input missed the tolerance; the flag is looked at behind the first halo reads
        // of the next sweep, so the test has no LDS round trip of its own.
        // returns 0: swept; 1: the previous sweep's input had met the tolerance (x is the solution); 2: out of budget
        auto sweep = [&](const int img_in, const int img_out, const int k, auto revtag) -> int {
            constexpr bool rev = decltype(revtag)::value;
            stage(img_in, rev ? BY - 1 : 0, revtag);
            if (k > 0) {
                const bool done = (*(volatile int*)&flg[(k - 1) % 3] == 0) && k >= a.dbg_sweeps;
                if (done) return 1;
            }
            if (k >= a.max_sweeps) return 2;
            rmx = 0.0;
            MS_UNROLL for (int t = 0; t < BY; ++t) {
                const int r = rev ? BY - 1 - t : t;
                if (t > 0) stage(img_in, r, revtag);
                double bpv[BX];            // LEAN: the row's b / l_ii, requested with its halo values
                MS_UNROLL for (int c = 0; c < BX; ++c) bpv[c] = LEAN ? lds_read(&rowp[r + 1][2 * IMG + xo<DEINT>(c, half)]) : bp[r][c];
                MS_UNROLL for (int u = 0; u < BX; ++u) {
                    const int c = rev ? BX - 1 - u : u;
                    auto nb = [&](int dy, int dx) -> double { return nbv<BX, BY>(x, v, r + dy, c + dx); };
                    double acc = bpv[c];
                    // (E, W), (NE, SW), (N, S): the accumulation order of the tile kernels
                    acc = fma(-lc[0][r][c], nb(0, 1), acc);
                    acc = fma(-lc[3][r][c], nb(0, -1), acc);
                    acc = fma(-lc[1][r][c], nb(1, 1), acc);
                    acc = fma(-lc[4][r][c], nb(-1, -1), acc);
                    acc = fma(-lc[2][r][c], nb(1, 0), acc);
                    acc = fma(-lc[5][r][c], nb(-1, 0), acc);
                    acc = vz(r, c, acc);
                    rmx = fmax(rmx, LEAN ? fabs(acc - x[r][c]) : fabs(acc - x[r][c]) * ldv[r][c]);
                    x[r][c] = acc;
                }
                MS_FENCE();
            }
            publish_rim<BX, BY>(x, img_out, dst);
            const bool viol = rmx > tolb;
            if (__any(viol) && (threadIdx.x % WAVE) == 0) flg[k % 3] = 1;
            if (threadIdx.x == 0) flg[(k + 1) % 3] = 0;
            __syncthreads();
            return 0;
        };
        int k = 0, cur = 0, st = 0;
        for (;;) {
            st = sweep(0, IMG, k, std::false_type{});
            if (st) { cur = 0; break; }
            ++k;
            st = sweep(IMG, 0, k, std::true_type{});
            if (st) { cur = 1; break; }
            ++k;
        }
        iters = k;
        // (the node index goes through an opaque move: the HBM addresses of the phases below are then formed here, not
        //  ahead of the sweep loop, where they would sit in a dozen registers the sweeps need)
        asm volatile("" : "+v"(gi0));
        MS_STAMP(4);
        if (st != 1) sflags |= FEMFCT_FLAG_SOLVER_BUDGET;

        // -------------------------------------------------------------- du/dt: r = rhs - A u_L, then ChebSI on M
        // (x = u_L, its rim in image `cur`: the halo is read row by row as in a sweep; u_L's rim is kept in image 2 for the
        //  flux phase.)
        double zb[BY][BX], ym[BY][BX], yo[BY][BX];
        double rw[BY][BX];                // boundary ring: 1 / (2.5 ntri) of the node (mass-matrix weights cnt_ij / (2.5 ntri))
        bool ex[BY][BX], ey[BY][BX];      // ... and whether its E / W resp. N / S edges lie on the mesh boundary (cnt = 1)
        {
            MS_UNROLL for (int r = 0; r < BY; ++r) {
                // one block row at a time: its 7 x BX matrix entries are requested together, then consumed
                stage(cur ? IMG : 0, r, std::false_type{});
                double ar[7][BX], rd[BX], mdv[BX], nsum[BX];
                MS_UNROLL for (int c = 0; c < BX; ++c) {
                    const int i = gidx(r, c);
                    MS_UNROLL for (int s = 0; s < 7; ++s) ar[s][c] = KEEP ? ak[s][r][c] : (Ab + (int64_t)s * n)[i];
                    rd[c] = 0.0; mdv[c] = 0.5 * hh; nsum[c] = 0.0;
                    if (!INTERIOR) mdv[c] = KEEP ? mdk[r][c] : a.Md[i];
                }
                if (HASNM) {
                    MS_UNROLL for (int c = 0; c < BX; ++c) nsum[c] = nrow[r][c];
                }
                MS_UNROLL for (int c = 0; c < BX; ++c) rd[c] = KEEP ? rk[r][c] : rhs[gidx(r, c)];
                MS_FENCE();
                MS_UNROLL for (int c = 0; c < BX; ++c) {
                    double acc = ar[0][c] * x[r][c], asum = ar[0][c];
                    MS_UNROLL for (int s = 1; s < 7; ++s) {
                        acc = fma(ar[s][c], nbv<BX, BY>(x, v, r + ms_dy(s), c + ms_dx(s)), acc);
                        asum += ar[s][c];
                    }
                    // row sum of L = M_L + dt (A - D + N): D's rows sum to zero, so it is M_L + dt sum_j (a_ij + n_ij)
                    // (helpers.py:1796-1809 sums the dense matrix; only the sign and the minimum are used)
                    {
                        const double mlr = INTERIOR ? hh : 2.0 * mdv[c];      // M_L = 2 m_ii on this mesh
                        const double rs = mlr + dt * (asum + nsum[c]);
                        rsmin = valid(r, c) ? fmin(rsmin, rs) : rsmin;
                    }
                    double rr = -acc + rd[c];
                    if (INTERIOR) {
                        zb[r][c] = rr * (1.6 / hh);                  // 1 / (1.25 m_ii), m_ii = h^2 / 2
                    } else {
                        rr = vz(r, c, rr);
                        zb[r][c] = rr * frcp(1.25 * mdv[c]);
                        // m_ij / (1.25 m_ii) = cnt_ij / (2.5 ntri): cnt = 2 for an edge inside the mesh, 1 on its boundary
                        const int gxn = ix0 + c, gyn = iy0 + r, nc = N - 1;
                        const int c00 = (gxn < nc && gyn < nc), c10 = (gxn > 0 && gyn < nc), c01 = (gxn < nc && gyn > 0),
                                  c11 = (gxn > 0 && gyn > 0);
                        const int ntri = 2 * c00 + c10 + c01 + 2 * c11;
                        rw[r][c] = zsel(valid(r, c) && ntri > 0, frcp(2.5 * (double)max(ntri, 1)));
                        ex[r][c] = (gyn == 0 || gyn == nc);
                        ey[r][c] = (gxn == 0 || gxn == nc);
                    }
                    ym[r][c] = a.om[0] * zb[r][c];
                    yo[r][c] = 0.0;
                }
                MS_FENCE();
            }
        }
        {
            double r0 = rmx, r2 = 0.0, r3 = 0.0;
            reduce4<NW>(r0, rsmin, r2, r3, red);
            resid = bmax > 0.0 ? (LEAN ? r0 * ldmax : r0) / bmax : 0.0;      // (LEAN: the bound the test went by)
            if (!(rsmin > 0.0)) sflags |= FEMFCT_FLAG_MMATRIX_ROWSUM;
        }
        publish_rim<BX, BY>(x, 2 * IMG, dst);
        MS_STAMP(5);
        int q = cur ^ 1;          // image the next iterate's rim goes to
        // y_k in ym, y_(k-1) in yo; an iteration overwrites y_(k-1) with y_(k+1) and the two arrays swap roles
        auto cheb = [&](double (&ycur)[BY][BX], double (&yold)[BY][BX], const int it) {
            publish_rim<BX, BY>(ycur, q ? IMG : 0, dst);
            __syncthreads();
            gather_halo<BX, BY>(v, rowp, q ? IMG : 0, half);
            const double om = a.om[(it - 1) % 20];
            MS_UNROLL for (int r = 0; r < BY; ++r)
                MS_UNROLL for (int c = 0; c < BX; ++c) {
                    const double sx = nbv<BX, BY>(ycur, v, r, c + 1) + nbv<BX, BY>(ycur, v, r, c - 1),
                                 sd = nbv<BX, BY>(ycur, v, r + 1, c + 1) + nbv<BX, BY>(ycur, v, r - 1, c - 1),
                                 sy = nbv<BX, BY>(ycur, v, r + 1, c) + nbv<BX, BY>(ycur, v, r - 1, c);
                    // z = (b - M y) / (1.25 m_ii) = zb - 0.8 y - sum_j w_ij y_j;  y' = om (z + y - y_old) + y_old
                    double z;
                    if (INTERIOR) {
                        z = fma(-2.0 / 15.0, (sx + sd) + sy, zb[r][c]);
                    } else {
                        const double t = ((ex[r][c] ? sx : sx + sx) + (sd + sd)) + (ey[r][c] ? sy : sy + sy);
                        z = fma(-rw[r][c], t, zb[r][c]);
                    }
                    yold[r][c] = fma(om, fma(0.2, ycur[r][c], z) - yold[r][c], yold[r][c]);
                }
            q ^= 1;
        };
        int it = 2;
        for (; it + 1 <= a.dbg_cheb; it += 2) {
            cheb(ym, yo, it);
            cheb(yo, ym, it + 1);
        }
        if (it <= a.dbg_cheb) {          // an even number of iterations in all: the result ends up in yo
            cheb(ym, yo, it);
            MS_UNROLL for (int r = 0; r < BY; ++r)
                MS_UNROLL for (int c = 0; c < BX; ++c) ym[r][c] = yo[r][c];
        }
        MS_STAMP(6);
        // ym = du/dt.  Flux phase; images: 2 = u_L rim, q = du/dt rim (published now)
        publish_rim<BX, BY>(ym, q ? IMG : 0, dst);
        __syncthreads();

        // -------------------------------------------------------------- fluxes, Zalesak limiter, update
        double ff[3][BY][BX], qp[BY][BX], qm[BY][BX];
        {
            // forward fluxes first (u_L, du/dt of the E, NE, N neighbours), then the bounds (all six u_L): the register
            // budget of 3 waves per SIMD holds one of the two halo sets at a time
            gather_halo<BX, BY>(v, rowp, 2 * IMG, half);
            {
                const double mq = (0.5 * hh) / 12.0;
                const int qi = q ? IMG : 0;
                MS_UNROLL for (int r = 0; r < BY; ++r)
                    MS_UNROLL for (int c = 0; c < BX; ++c) {
                        const int i = gidx(r, c);
                        int pc = 0;
                        if (!INTERIOR) pc = mass_edge_counts(ix0 + c, iy0 + r, N - 1);
                        MS_UNROLL for (int s = 1; s <= 3; ++s) {
                            const double duj = s == 1 ? nb_get<BX, BY, 1, 0>(ym, rowp, qi, r, c, half)
                                             : s == 2 ? nb_get<BX, BY, 1, 1>(ym, rowp, qi, r, c, half)
                                                      : nb_get<BX, BY, 0, 1>(ym, rowp, qi, r, c, half);
                            const double mij = INTERIOR ? 2.0 * mq : (double)((pc >> (2 * (s - 1))) & 3) * mq;
                            const double dij = vz(r, c, (Dh + (int64_t)s * n)[i]);
                            const double fs = mij * (ym[r][c] - duj) + dij * (x[r][c] - nbv<BX, BY>(x, v, r + ms_dy(s), c + ms_dx(s)));
                            ff[s - 1][r][c] = vz(r, c, fs);
                        }
                    }
            }
            MS_UNROLL for (int r = 0; r < BY; ++r)
                MS_UNROLL for (int c = 0; c < BX; ++c) {
                    const double ui = x[r][c];
                    double umax = ui, umin = ui;
                    int pc = 0;
                    if (!INTERIOR) pc = valid(r, c) ? mass_edge_counts(ix0 + c, iy0 + r, N - 1) : 0;
                    MS_UNROLL for (int s = 1; s < 7; ++s) {
                        const double uj = nbv<BX, BY>(x, v, r + ms_dy(s), c + ms_dx(s));
                        const bool exists = INTERIOR || ((pc >> (2 * (s - 1))) & 3) != 0;
                        umax = exists ? fmax(umax, uj) : umax;
                        umin = exists ? fmin(umin, uj) : umin;
                    }
                    qp[r][c] = umax - ui;
                    qm[r][c] = umin - ui;
                }
        }
        MS_STAMP(7);
        __syncthreads();       // every rim of u_L and du/dt has been read: the three images take the forward fluxes
        publish<BX, BY, -1, 0>(ff[0], 0, dst);
        publish<BX, BY, -1, -1>(ff[1], IMG, dst);
        publish<BX, BY, 0, -1>(ff[2], 2 * IMG, dst);
        __syncthreads();
        double rp[BY][BX], rm[BY][BX], rml[BY][BX];      // R+, R-, dt / M_L
        MS_UNROLL for (int r = 0; r < BY; ++r)
            MS_UNROLL for (int c = 0; c < BX; ++c) rml[r][c] = INTERIOR ? dt / hh : (KEEP ? mlk[r][c] : a.ml[gidx(r, c)]);
        if (!INTERIOR) {
            MS_FENCE();
            MS_UNROLL for (int r = 0; r < BY; ++r)
                MS_UNROLL for (int c = 0; c < BX; ++c) { rml[r][c] = dt * frcp(rml[r][c]); MS_FENCE(); }
        }
        MS_UNROLL for (int r = 0; r < BY; ++r)
            MS_UNROLL for (int c = 0; c < BX; ++c) {
                // backward slots: f_ij = -f_ji, the bits the neighbour computed
                const double fw = -nb_get<BX, BY, -1, 0>(ff[0], rowp, 0, r, c, half);
                const double fsw = -nb_get<BX, BY, -1, -1>(ff[1], rowp, IMG, r, c, half);
                const double fs = -nb_get<BX, BY, 0, -1>(ff[2], rowp, 2 * IMG, r, c, half);
                double pp = 0.0, pm = 0.0;
                MS_UNROLL for (int s = 0; s < 3; ++s) { pp += fmax(ff[s][r][c], 0.0); pm += fmin(ff[s][r][c], 0.0); }
                pp += fmax(fw, 0.0); pm += fmin(fw, 0.0);
                pp += fmax(fsw, 0.0); pm += fmin(fsw, 0.0);
                pp += fmax(fs, 0.0); pm += fmin(fs, 0.0);
                // R+- = min(1, M_L Q+- / (dt P+-))
                rp[r][c] = (pp != 0.0) ? fmin(1.0, qp[r][c] * frcp(rml[r][c] * pp)) : 1.0;
                rm[r][c] = (pm != 0.0) ? fmin(1.0, qm[r][c] * frcp(rml[r][c] * pm)) : 1.0;
                MS_FENCE();
            }
        __syncthreads();
        publish_rim<BX, BY>(rp, 0, dst);
        publish_rim<BX, BY>(rm, IMG, dst);
        __syncthreads();
        MS_STAMP(8);
        double fbar[BY][BX];
        MS_UNROLL for (int r = 0; r < BY; ++r)
            MS_UNROLL for (int c = 0; c < BX; ++c) {
                double acc = 0.0;
                MS_UNROLL for (int s = 0; s < 3; ++s) {
                    const double rpj = s == 0 ? nb_get<BX, BY, 1, 0>(rp, rowp, 0, r, c, half)
                                     : s == 1 ? nb_get<BX, BY, 1, 1>(rp, rowp, 0, r, c, half)
                                              : nb_get<BX, BY, 0, 1>(rp, rowp, 0, r, c, half);
                    const double rmj = s == 0 ? nb_get<BX, BY, 1, 0>(rm, rowp, IMG, r, c, half)
                                     : s == 1 ? nb_get<BX, BY, 1, 1>(rm, rowp, IMG, r, c, half)
                                              : nb_get<BX, BY, 0, 1>(rm, rowp, IMG, r, c, half);
                    const double f = ff[s][r][c];
                    const double al = (f > 0.0) ? fmin(rp[r][c], rmj) : fmin(rm[r][c], rpj);
                    ff[s][r][c] = al * f;                 // limited flux of the forward edge
                    acc += ff[s][r][c];
                }
                fbar[r][c] = acc;
                MS_FENCE();
            }
        __syncthreads();
        publish<BX, BY, -1, 0>(ff[0], 0, dst);
        publish<BX, BY, -1, -1>(ff[1], IMG, dst);
        publish<BX, BY, 0, -1>(ff[2], 2 * IMG, dst);
        __syncthreads();
        MS_UNROLL for (int r = 0; r < BY; ++r)
            MS_UNROLL for (int c = 0; c < BX; ++c) {
                double acc = fbar[r][c];
                acc -= nb_get<BX, BY, -1, 0>(ff[0], rowp, 0, r, c, half);
                acc -= nb_get<BX, BY, -1, -1>(ff[1], rowp, IMG, r, c, half);
                acc -= nb_get<BX, BY, 0, -1>(ff[2], rowp, 2 * IMG, r, c, half);
                (ALLV || valid(r, c) ? out : Dh)[gidx(r, c)] = fma(rml[r][c], acc, x[r][c]);
            }
    }

    MS_STAMP(9);
#ifdef FEMFCT_TUNING
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[11] = __builtin_readcyclecounter();
#endif
    // ------------------------------------------------------------------ solver record, log
    if (threadIdx.x == 0) {
        StepCtl rec;
        rec.flags = sflags; rec.iters = iters; rec.done = 1; rec.parity = 0;
        rec.resid = resid; rec.bnorm = bmax; rec.min_rowsum = rsmin; rec.rs[0] = rec.rs[1] = 0.0; rec.pad = 0.0;
        a.ctl[bz] = rec;
        if (a.e.level) a.e.log[(int64_t)ord * a.e.batch + bz] = rec;
    }
    if (a.e.level && a.e.kctl) {
        const uint32_t* ks = reinterpret_cast<const uint32_t*>(a.e.kctl + bz);
        uint32_t* kd = reinterpret_cast<uint32_t*>(a.e.klog + (int64_t)ord * a.e.batch + bz);
        if (threadIdx.x < 16) kd[threadIdx.x] = ks[threadIdx.x];
    }
}

template <int BX, int BY, int NMAX, int NT, int NFIX, bool HASNM>
__global__ void __launch_bounds__(NT) k_mesh_step(MeshStepArgs a) {
    extern __shared__ double lds[];
    constexpr int IMG = mesh_img_doubles<(BX == 2 && NFIX != 0)>(NMAX);
    double* red = lds + 3 * IMG;                         // 4 * NT / 64 doubles
    int* flg = reinterpret_cast<int*>(red + 4 * (NT / WAVE));   // 4 ints, then one trash double
    const int bz = blockIdx.x, tid = threadIdx.x;
    // time level and log ordinal are read before anything else (the last workgroup of the graph's last step moves them)
    const int ord = a.e.level ? a.e.level[1] + a.e.ord_off : 0;
    MS_STAMP(0);
    // interior blocks first (whole waves), the boundary ring behind them
    // (spare threads repeat block (1, 1) of the interior resp. block (0, 0) of the ring)
    int bx = 1, by = 1;
    const int TXi = a.TX - 2;
    if (tid < a.nint) {
        by = tid / TXi; bx = tid - by * TXi + 1; by += 1;
    } else if (tid >= a.nint_pad) {
        const int kk = tid - a.nint_pad, nb = 2 * a.TX + 2 * (a.TY - 2);
        bx = 0; by = 0;
        if (kk < a.TX) { bx = kk; }
        else if (kk < 2 * a.TX) { bx = kk - a.TX; by = a.TY - 1; }
        else if (kk < nb) { const int k2 = kk - 2 * a.TX; by = 1 + (k2 >> 1); bx = (k2 & 1) ? a.TX - 1 : 0; }
    }
    if (tid < a.nint_pad) mesh_step_body<BX, BY, NMAX, NT, NFIX, HASNM, true>(a, lds, red, flg, bx, by, bz, ord);
    else mesh_step_body<BX, BY, NMAX, NT, NFIX, HASNM, false>(a, lds, red, flg, bx, by, bz, ord);
    // the graph's last step moves the time level and the step ordinal, once every workgroup has logged
    if (a.e.level && a.e.ord_adv != 0) {
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            if (atomicAdd(a.e.ticket, 1u) == gridDim.x - 1) {
                a.e.level[0] += a.e.delta;
                a.e.level[1] = (ord - a.e.ord_off) + a.e.ord_adv;
                *a.e.ticket = 0u;
            }
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- host side
static bool mesh_step_fits(const femfct_ctx* ctx, bool have_nm) {
    if (!ctx->mesh_step || !ctx->use_strips || !ctx->use_tiles) return false;      // (femfct_set_fusion(0, 0): one-sweep kernels)
    if (!ctx->structured || !ctx->implicit_cols || ctx->W != 7 || !ctx->mass_is_mesh) return false;
    if (ctx->solver != FEMFCT_SOLVER_JACOBI || ctx->N < 5) return false;
    if (ctx->N <= 42) return true;
    // 3 x 3 blocks: the 81 x 81 meshes of config C2 with N as a compile-time constant, no non-flux matrix.  (The variant
    // with N at run time needs 820 B of scratch per thread and is 3-4x slower than the tile path at 61 x 61: tuning builds only.)
#ifdef FEMFCT_TUNING
    return ctx->N <= 81 && !have_nm;
#else
    return ctx->N == 81 && !have_nm;
#endif
}

// N <= 42: 2 x 2 blocks, from mesh_step_min_batch trajectories per launch on (1: always).  N = 81: 3 x 3 blocks, from
// mesh_step_min_batch_large on -- below that the tile path's four launches, spread over many CUs, are faster
// (r04, us per step of the batch, one-workgroup | tiles: B=1 116 | 46, B=64 151 | 159, B=256 234 | 550).
bool femfct_mesh_step_wanted(const femfct_ctx* ctx, int32_t batch, bool have_nm) {
    if (!mesh_step_fits(ctx, have_nm)) return false;
    return batch >= (ctx->N <= 42 ? ctx->mesh_step_min_batch : ctx->mesh_step_min_batch_large);
}

int femfct_enqueue_mesh_step(femfct_ctx* ctx, MatRef A, const double* Nm, int32_t nshared, VecRef rhs, int64_t rhs_bstride,
                             VecRef u_n, int64_t u_bstride, double dt, VecRef u_out, int64_t out_bstride, int32_t batch,
                             int32_t budget, bool fuse_end) {
    if (!mesh_step_fits(ctx, Nm != nullptr)) return femfct_fail(ctx, FEMFCT_ERR_INVALID, "one-workgroup step outside its regime");
    if (!rhs.base) {          // 'no rhs' = a vector of zeros (a null test in the kernel would be a branch per block row)
        if (!ctx->d_zero) return femfct_fail(ctx, FEMFCT_ERR_INVALID, "one-workgroup step: workspace not set up");
        rhs = make_ref(ctx->d_zero);
        rhs_bstride = 0;
    }
    MeshStepArgs a{};
    const int B = ctx->N <= 42 ? 2 : 3;
    a.n = ctx->n; a.N = ctx->N;
    a.TX = a.TY = (ctx->N + B - 1) / B;
    a.nint = (a.TX - 2) * (a.TY - 2);
    a.nint_pad = (a.nint + WAVE - 1) / WAVE * WAVE;
    a.h = ctx->h; a.dt = dt;
    a.A = A; a.Nm = Nm; a.nm_bs = nshared ? 0 : (int64_t)ctx->W * ctx->n;
    a.rhs = rhs; a.rhs_bs = rhs_bstride; a.u_n = u_n; a.u_bs = u_bstride; a.out = u_out; a.out_bs = out_bstride;
    a.ml = ctx->d_ml; a.Md = ctx->d_M; a.Dh = ctx->d_D; a.ctl = ctx->d_ctl;
    a.rel_tol = ctx->rel_tol;
    // the solve stops by itself: the budget only bounds a diverging iteration
    a.max_sweeps = std::min(ctx->max_iters, std::max(budget, 8));
    a.dbg_sweeps = 0; a.dbg_cheb = 20;
#ifdef FEMFCT_TUNING
    if (const char* e = getenv("FEMFCT_MESH_DBG")) {
        if (sscanf(e, "%d,%d", &a.dbg_sweeps, &a.dbg_cheb) == 2) a.max_sweeps = std::max(a.max_sweeps, a.dbg_sweeps);
        else { a.dbg_sweeps = 0; a.dbg_cheb = 20; }
    }
#endif
    {
        const double lmin = 0.5, lmax = 2.0, rho = (lmax - lmin) / (lmax + lmin);
        double w = 0.0;
        for (int k = 1; k <= 20; ++k) {          // helpers.py:170-179
            if (k == 2) w = 1.0 / (1.0 - rho * rho / 2.0);
            else w = 1.0 / (1.0 - (w * rho * rho) / 4.0);
            a.om[k - 1] = w;
        }
    }
    a.e = EndArgs{};
    a.trace = ctx->d_mesh_trace;       // (FEMFCT_TUNING builds with FEMFCT_MESH_TRACE set; else null)
    if (fuse_end) {
        a.e.level = ctx->d_level; a.e.delta = ctx->rep_last ? ctx->end_req_delta * ctx->rep_total : 0;
        a.e.ord_adv = ctx->rep_last ? ctx->rep_total : 0; a.e.ord_off = ctx->ord_bias; a.e.ctl = ctx->d_ctl; a.e.log = ctx->d_log;
        a.e.kctl = ctx->end_req_krylov ? (const KrylovCtl*)ctx->d_kry_ctl : nullptr; a.e.klog = (KrylovCtl*)ctx->d_klog;
        a.e.batch = batch; a.e.ticket = ctx->d_ticket;
    }
    const int nthreads = a.nint_pad + 2 * a.TX + 2 * (a.TY - 2);
    femfct_prof_begin(ctx, KC_JACOBI);
#define MS_LAUNCH(B, NMAX_, NT_, NFIX_, NM_, slot)                                                                       \
    do {                                                                                                                 \
        constexpr size_t lds = (size_t)(3 * mesh_img_doubles<(B == 2 && NFIX_ != 0)>(NMAX_) + 4 * (NT_ / WAVE)) * 8 + 24;                \
        if (nthreads > NT_) return femfct_fail(ctx, FEMFCT_ERR_INVALID, "one-workgroup step: mesh does not fit");        \
        if (!ctx->mesh_step_attr[slot]) {                                                                                \
            HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_mesh_step<B, B, NMAX_, NT_, NFIX_, NM_>,                          \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
            ctx->mesh_step_attr[slot] = true;                                                                            \
        }                                                                                                                \
        hipLaunchKernelGGL((k_mesh_step<B, B, NMAX_, NT_, NFIX_, NM_>), dim3(batch), dim3(NT_), lds, ctx->stream, a);         \
    } while (0)
    if (ctx->N == 41 && Nm) MS_LAUNCH(2, 42, 512, 41, true, 0);
    else if (ctx->N == 41) MS_LAUNCH(2, 42, 512, 41, false, 1);
    else if (ctx->N <= 42 && Nm) MS_LAUNCH(2, 42, 512, 0, true, 2);
    else if (ctx->N <= 42) MS_LAUNCH(2, 42, 512, 0, false, 3);
    else if (ctx->N == 81) MS_LAUNCH(3, 81, 768, 81, false, 4);
#ifdef FEMFCT_TUNING
    else MS_LAUNCH(3, 81, 768, 0, false, 5);
#else
    else return femfct_fail(ctx, FEMFCT_ERR_INVALID, "one-workgroup step: no variant for this mesh");
#endif
#undef MS_LAUNCH
    femfct_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return femfct_fail(ctx, FEMFCT_ERR_HIP, "one-workgroup step launch failed: %s", hipGetErrorString(e));
    return FEMFCT_OK;
}

#ifdef FEMFCT_TUNING
// phase timestamps (100 MHz ticks) of workgroup 0 of the most recent one-workgroup step
extern "C" int femfct_mesh_trace(femfct_ctx* ctx, unsigned long long* out16) {
    FEMFCT_ENTER(ctx);
    if (!ctx->d_mesh_trace) return FEMFCT_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out16, ctx->d_mesh_trace, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return FEMFCT_OK;
}
#endif
