// End of a time step folded into the step's last kernel, and the geometric mass-matrix counts: shared by the
// tile kernels (kernels_strip.hip) and the one-workgroup-per-trajectory step kernel (kernels_mesh.hip).
#pragma once

#include "femfct_internal.h"
#include "device_utils.h"
#include "forms.h"

// optional tail of the step, done by the last workgroup to finish (ticket): copy the solver control
// blocks to the per-step log and move the time-level counter -- saves the separate k_step_end launch
struct EndArgs {
    int32_t* level;          // null: nothing to do
    int delta;               // what the LAST step of a graph adds to the time level (R * step); 0 for the other steps
    int ord_adv;             // what it adds to the step ordinal (R); 0: not the last step -- log only
    int ord_off;             // this step's position in its graph: it logs at ordinal level[1] + ord_off
    const StepCtl* ctl;
    StepCtl* log;
    const KrylovCtl* kctl;   // may be null
    KrylovCtl* klog;
    int batch;
    unsigned* ticket;
};

// End of a time step folded into the step's last kernel.  The device counters (time level, step ordinal) move once per
// captured graph (femfct_run_graph_reps): every step but the last only has its solver records copied into the
// trajectory log, by workgroup (0,0,0) at the START of the kernel (the records are final by then) -- no atomic, no
// extra barrier, nothing at the kernel's end.  The last step keeps the ticket: the workgroup that draws the last one
// logs and moves the counters for all R steps (every workgroup has resolved its level-dependent addresses before it
// draws a ticket, so moving the level is safe).  Measured at C2: the per-step ticket cost 6 of 37 us.
// (the records are 64 bytes each: copied word by word, one load and one store per thread -- a struct copy per thread
// costs the limiter, which runs at its 64-VGPR limit, spills in its load phase)
static_assert(sizeof(StepCtl) == 64 && sizeof(KrylovCtl) == 64, "step records are copied as 16 words");
__device__ __forceinline__ void step_log_copy(const EndArgs& e, int ord) {
    const int nw = e.batch * 16;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(e.ctl);
    uint32_t* d = reinterpret_cast<uint32_t*>(e.log + (int64_t)ord * e.batch);
    for (int t = threadIdx.x; t < nw; t += blockDim.x) d[t] = s[t];
    if (e.kctl) {
        const uint32_t* ks = reinterpret_cast<const uint32_t*>(e.kctl);
        uint32_t* kd = reinterpret_cast<uint32_t*>(e.klog + (int64_t)ord * e.batch);
        for (int t = threadIdx.x; t < nw; t += blockDim.x) kd[t] = ks[t];
    }
}

__device__ __forceinline__ void step_log_early(const EndArgs& e) {
    if (!e.level || e.ord_adv != 0) return;
    if (blockIdx.x | blockIdx.y | blockIdx.z) return;
    step_log_copy(e, e.level[1] + e.ord_off);
}

__device__ __forceinline__ void step_end_by_last_workgroup(const EndArgs& e) {
    if (!e.level || e.ord_adv == 0) return;
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        is_last = (atomicAdd(e.ticket, 1u) == total - 1);
    }
    __syncthreads();
    if (is_last) {
        const int ord0 = e.level[1];
        step_log_copy(e, ord0 + e.ord_off);
        __syncthreads();
        if (threadIdx.x == 0) {
            e.level[0] += e.delta;
            e.level[1] = ord0 + e.ord_adv;
            *e.ticket = 0u;
        }
    }
}

// Off-diagonal entries of the mesh's own P1 mass matrix from the cell geometry (right-diagonal mesh): m_ij = cnt_ij |K| / 12
// with cnt_ij in {0, 1, 2} triangles on the edge -- the bits k_mesh_constants stores (cnt * (area / 12.0) is exact for
// cnt <= 2), so a limiter that derives them instead of loading six doubles per row computes the identical step.
// Packed as six 2-bit counts, slots E, NE, N, W, SW, S (the order of k_strip4_cheb_mass).
__device__ __forceinline__ int mass_edge_counts(int gx, int gy, int nc) {
    const int c00 = (gx < nc && gy < nc), c10 = (gx > 0 && gy < nc), c01 = (gx < nc && gy > 0), c11 = (gx > 0 && gy > 0);
    return (c00 + c01) | ((2 * c00) << 2) | ((c00 + c10) << 4) | ((c10 + c11) << 6) | ((2 * c11) << 8) | ((c11 + c01) << 10);
}

