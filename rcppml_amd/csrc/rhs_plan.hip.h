// rhs_plan.hip.h -- the plan object behind rcppml_hip_rhs_plan_create / rcppml_hip_rhs_planned: either the round-4 window
// plan (kind 1: kernels_rhs_win.hip.h, ops_rhs_win.hip) or the round-2/3 slab plan (kind 0: kernels_rhs_tiled.hip.h).
#pragma once
#include "common.hip.h"
#include "kernels_rhs_tiled.hip.h"
#include "kernels_rhs_win.hip.h"

struct rcppml_rhs_plan {
    int kind = 0;                // 0 = slab plan (two 64 KiB tiles, S slots per (column, tile)), 1 = window plan (ring of four 32 KiB tiles)
    int dtype = 0, k = 0, device = 0;
    rk::RhsTiledGeom G{};
    rk::RhsWinGeom WG{};
    void* svals = nullptr;       // kind 0: slot values / kind 1: the whole slot stream
    uint16_t* soffs = nullptr;
    int* ovptr = nullptr;
    int* ovrow = nullptr;
    void* ovval = nullptr;
    void* Bp = nullptr;          // P > 1: per-partition partial outputs
    unsigned* dest = nullptr;    // kind 1, created without values: where the value of nonzero e goes (rcppml_hip_rhs_plan_set_values)
    const int* colptr = nullptr; // the caller's CSC (not owned): the tail columns go through the gather kernel
    const int* rowidx = nullptr;
    const void* vals = nullptr;
    int64_t ovnnz = 0, nnz = 0, nslots = 0;
    double ov_fraction = 0.0, fill = 0.0, stream_bytes = 0.0;
    // Ownership is decided ONCE per plan: either every buffer comes from the creating context's per-fit arena (freed with it,
    // nothing to free here) or every buffer is hipMalloc'ed and freed by the destructor -- never a mixture.
    bool in_arena = false;
    void* block = nullptr;       // kind 1: the one allocation all buffers of the plan are carved from (hipMalloc'ed iff !in_arena)
    ~rcppml_rhs_plan() {
        if (in_arena) return;
        if (block) { (void)hipFree(block); return; }
        for (void* p : {svals, (void*)soffs, (void*)ovptr, (void*)ovrow, ovval, Bp})
            if (p) (void)hipFree(p);
    }
};

// ops_rhs_win.hip
rcppml_rhs_plan* rcppml_rw_build_f32(rcppml_hip_ctx* c, const int* colptr, const int* rowidx, const float* vals, int64_t ncols, int64_t nrows, int k, int partitions, int rate_code);
rcppml_rhs_plan* rcppml_rw_build_f64(rcppml_hip_ctx* c, const int* colptr, const int* rowidx, const double* vals, int64_t ncols, int64_t nrows, int k, int partitions, int rate_code);
void rcppml_rw_set_values_f32(rcppml_hip_ctx* c, rcppml_rhs_plan* pl, const float* vals);
void rcppml_rw_set_values_f64(rcppml_hip_ctx* c, rcppml_rhs_plan* pl, const double* vals);
void rcppml_rw_run_f32(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const float* F, float* B);
void rcppml_rw_run_f64(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const double* F, double* B);
