// Device-side helpers shared by the gfx950 kernels: row partitioning (XCD-aware),
// wave64/block reductions, level-indirected vector references.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

// A vector that may live inside a time-major trajectory whose current level is
// held in a device-side counter (so that one captured hipGraph serves every
// time step): address = base + (*level + level_off) * stride.
struct VecRef {
    const double* base;
    const int32_t* level;   // may be null (level = 0)
    int64_t stride;
    int32_t level_off;
};

__device__ __forceinline__ const double* vec_ptr(const VecRef& r) {
    if (r.base == nullptr) return nullptr;
    int64_t lv = (r.level ? (int64_t)(*r.level) : 0) + r.level_off;
    return r.base + lv * r.stride;
}

// A per-step matrix (ELL values) that may be one of a pre-assembled sequence indexed by the device-side
// time level: address = base + (*level + level_off) * lstride + batch_member * bstride.
struct MatRef {
    const double* base;
    const int32_t* level;   // null: a single matrix
    int64_t lstride;
    int32_t level_off;
    int64_t bstride;        // batch stride (0: shared by the batch)
};

__device__ __forceinline__ const double* mat_ptr(const MatRef& r, int bz) {
    int64_t lv = r.level ? (int64_t)(*r.level) + r.level_off : 0;
    return r.base + lv * r.lstride + (int64_t)bz * r.bstride;
}

// Optional overrides of the tile Chebyshev kernels (species solves inside trajectory sweeps): another
// matrix than the mass matrix (per batch member), a level-indirected start iterate / destination.
struct ChebIO {
    const double* mat;      // null: the registered mass matrix
    int64_t mat_bs;         // batch stride of mat (0: shared)
    VecRef mid_ref;         // base null: use the plain pointer argument
    int64_t mid_bs;
    VecRef out_ref;
    int64_t out_bs;
    const double* om_dev;   // null: the by-value omega table; else omega_(k0+k) = om_dev[bz*om_bs + k0 + k]
    int32_t om_bs, k0;
    const double* scale_dev;   // null: md_scale argument; else per batch member
};

static inline VecRef make_ref(const double* p) { return VecRef{p, nullptr, 0, 0}; }
static inline VecRef make_ref(const double* p, const int32_t* level, int64_t stride, int32_t off) {
    return VecRef{p, level, stride, off};
}

// ---------------------------------------------------------------------------
// Row partitioning.  gridDim.x blocks each own one contiguous chunk of rows.  xcd_remap makes
// consecutive chunks land on the same XCD (blocks are dealt round-robin over the 8 XCDs, each with
// a private L2); it is bijective for any grid size and only affects speed.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int xcd_remap(int b, int G) {
    int q = G >> 3, r = G & 7;
    int xcd = b & 7, idx = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

struct RowRange { int begin, end; };

// Measured on MI355X (tools/bw_probe.hip, 7 value streams + 6 gathers, n = 2049^2): the plain
// block -> chunk map streams 4-15 % faster than the XCD-contiguous remap (8 XCDs x 11 streams far
// apart cost more DRAM locality than the shared x-halo lines save in L2), so the remap is off.
#ifndef FEMFCT_XCD_REMAP
#define FEMFCT_XCD_REMAP 0
#endif

__device__ __forceinline__ RowRange block_rows(int n) {
    int G = gridDim.x;
    int lb = FEMFCT_XCD_REMAP ? xcd_remap(blockIdx.x, G) : (int)blockIdx.x;
    int chunk = (n + G - 1) / G;
    // keep chunks a multiple of the wave size so that wave loads stay aligned
    chunk = (chunk + WAVE - 1) & ~(WAVE - 1);
    int b = lb * chunk;
    int e = b + chunk;
    if (b > n) b = n;
    if (e > n) e = n;
    return RowRange{b, e};
}

// ---------------------------------------------------------------------------
// Reductions (wave64 shuffles, then LDS across the block's waves).  All of them
// return the result in every thread.  smem must hold >= 32 doubles.
// ---------------------------------------------------------------------------
struct OpMax { __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); } };
struct OpMin { __device__ __forceinline__ double operator()(double a, double b) const { return fmin(a, b); } };
struct OpSum { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };

template <class Op>
__device__ __forceinline__ double wave_reduce(double v, Op op) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, WAVE));
    return v;
}

// max / min over the wave with DPP lane moves instead of ds_bpermute shuffles: the shuffle version is six dependent LDS
// round trips (~700 cycles per reduction -- a microsecond for the three reductions that end k_tile_build_jacobi); this
// one is six register moves + ops (~100 cycles).  Order of combination does not matter for max / min (exact), so the
// result is the same bits; sums keep the shuffle tree (their rounding depends on the order).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {      // lanes outside ROW_MASK (or without a source) keep their own v
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <class Op>
__device__ __forceinline__ double wave_reduce_dpp(double v, Op op) {
    v = op(v, dpp_take<0xB1, 0xf>(v));     // quad_perm [1,0,3,2]
    v = op(v, dpp_take<0x4E, 0xf>(v));     // quad_perm [2,3,0,1]
    v = op(v, dpp_take<0x141, 0xf>(v));    // row_half_mirror
    v = op(v, dpp_take<0x140, 0xf>(v));    // row_mirror: every lane of a 16-lane row holds the row's result
    v = op(v, dpp_take<0x142, 0xa>(v));    // row_bcast:15 into rows 1, 3
    v = op(v, dpp_take<0x143, 0xc>(v));    // row_bcast:31 into rows 2, 3: lane 63 holds the wave's result
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_reduce(double v, OpMax op) { return wave_reduce_dpp(v, op); }
__device__ __forceinline__ double wave_reduce(double v, OpMin op) { return wave_reduce_dpp(v, op); }

// the same tree for sums (lanes without a source add 0): a fixed order, so still deterministic
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take0(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_reduce(double v, OpSum) {
    v += dpp_take0<0xB1, 0xf>(v);
    v += dpp_take0<0x4E, 0xf>(v);
    v += dpp_take0<0x141, 0xf>(v);
    v += dpp_take0<0x140, 0xf>(v);
    v += dpp_take0<0x142, 0xa>(v);
    v += dpp_take0<0x143, 0xc>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

template <class Op>
__device__ __forceinline__ double block_reduce(double v, Op op, double identity, double* smem) {
    v = wave_reduce(v, op);
    int nw = (blockDim.x + WAVE - 1) / WAVE;
    if (nw == 1) return v;
    int wid = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    __syncthreads();  // smem reuse across successive calls
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    double r = identity;
    for (int w = 0; w < nw; ++w) r = op(r, smem[w]);  // fixed order: deterministic
    return r;
}

// Three reductions (max, max, min) with one barrier pair instead of three; smem must hold >= 96 doubles.
__device__ __forceinline__ void block_reduce_max_max_min(double& a, double& b, double& c, double* smem) {
    a = wave_reduce(a, OpMax());
    b = wave_reduce(b, OpMax());
    c = wave_reduce(c, OpMin());
    const int nw = (blockDim.x + WAVE - 1) / WAVE;
    if (nw == 1) return;
    const int wid = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    __syncthreads();  // smem reuse across successive calls
    if (lane == 0) { smem[wid] = a; smem[32 + wid] = b; smem[64 + wid] = c; }
    __syncthreads();
    double ra = 0.0, rb = 0.0, rc = INFINITY;
    for (int w = 0; w < nw; ++w) {   // fixed order: deterministic
        ra = fmax(ra, smem[w]); rb = fmax(rb, smem[32 + w]); rc = fmin(rc, smem[64 + w]);
    }
    a = ra; b = rb; c = rc;
}

// Every block reduces the same `count` per-block partials in the same order, so
// all blocks obtain a bitwise identical value without any atomics.
template <class Op>
__device__ __forceinline__ double reduce_partials(const double* part, int count, Op op,
                                                  double identity, double* smem) {
    double v = identity;
    for (int k = threadIdx.x; k < count; k += blockDim.x) v = op(v, part[k]);
    return block_reduce(v, op, identity, smem);
}
