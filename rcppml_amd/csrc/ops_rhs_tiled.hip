// ops_rhs_tiled.hip -- planned (LDS row-tiled) form of the sparse right-hand side: plan = the tile-partitioned slot copy
// of one CSC matrix for one rank and precision, built once per fit on the device; rcppml_hip_rhs_planned runs it.
// Reference semantics: primitives/cpu/rhs.hpp:52-70, fused_nnls.hpp:109-114 (same numbers as rcppml_hip_rhs up to
// summation order).
#include <memory>
#include <mutex>
#include "rhs_tiled_launch.hip.h"
#include "scan.hip.h"

using namespace rk;

namespace {

// plan buffers: ownership is decided by the FIRST buffer -- from the fit's arena (then the plan owns nothing) or hipMalloc (then
// the destructor frees every one of them) -- and never mixed: a later buffer that does not fit the arena fails the plan (the
// caller then runs without one) instead of leaving earlier hipMalloc'ed buffers behind
template <class P>
void plan_alloc(rcppml_hip_ctx* c, rcppml_rhs_plan* pl, P** out, size_t bytes) {
    const size_t b = bytes < 16 ? 16 : bytes;
    const bool first = !pl->svals && !pl->soffs && !pl->ovptr && !pl->ovrow && !pl->ovval && !pl->Bp;
    void* q = nullptr;
    if (first || pl->in_arena) {
        q = c->arena_take(b);
        if (q) pl->in_arena = true;
        else if (pl->in_arena) throw std::runtime_error("rhs_plan: arena exhausted half-way through a plan");
    }
    if (!q) HIPCHK(hipMalloc(&q, b));
    *out = static_cast<P*>(q);
}

template <class T>
void run_plan(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const T* F, T* B) {
    const RhsTiledGeom& G = pl->G;
    const bool ov = pl->ovnnz > 0;
    const int64_t nt = G.ncols_tiled;
    if (nt < G.ncols) {          // tail columns (those that do not fill a whole round of workgroups): a workgroup per column
        const unsigned grid = (unsigned)(G.ncols - nt);
        if (G.rowb == 256)
            hipLaunchKernelGGL((rhs_tail_kernel<T, 1, 4>), dim3(grid), dim3(256), 0, c->stream, pl->colptr, pl->rowidx, (const T*)pl->vals,
                               nt, G.ncols, F, pl->k, B);
        else if (G.rowb == 1024)
            hipLaunchKernelGGL((rhs_tail_kernel<T, 4, 2>), dim3(grid), dim3(256), 0, c->stream, pl->colptr, pl->rowidx, (const T*)pl->vals,
                               nt, G.ncols, F, pl->k, B);
        else
            hipLaunchKernelGGL((rhs_tail_kernel<T, 2, 2>), dim3(grid), dim3(256), 0, c->stream, pl->colptr, pl->rowidx, (const T*)pl->vals,
                               nt, G.ncols, F, pl->k, B);
        HIPCHK(hipGetLastError());
    }
    if (ov) {          // the spilled nonzeros first: their sums seed the accumulators
        const unsigned grid = (unsigned)((nt + 15) / 16);
        if (G.rowb == 256)
            hipLaunchKernelGGL((rhs_tiled_spill_kernel<T, 1, 4>), dim3(grid), dim3(256), 0, c->stream, (const int*)pl->ovptr,
                               (const int*)pl->ovrow, (const T*)pl->ovval, nt, F, pl->k, B);
        else if (G.rowb == 1024)
            hipLaunchKernelGGL((rhs_tiled_spill_kernel<T, 4, 2>), dim3(grid), dim3(256), 0, c->stream, (const int*)pl->ovptr,
                               (const int*)pl->ovrow, (const T*)pl->ovval, nt, F, pl->k, B);
        else
            hipLaunchKernelGGL((rhs_tiled_spill_kernel<T, 2, 4>), dim3(grid), dim3(256), 0, c->stream, (const int*)pl->ovptr,
                               (const int*)pl->ovrow, (const T*)pl->ovval, nt, F, pl->k, B);
        HIPCHK(hipGetLastError());
    }
    const T* Binit = (G.P == 1 && ov) ? B : nullptr;
    T* Bout = G.P == 1 ? B : (T*)pl->Bp;
    if constexpr (std::is_same<T, float>::value) rcppml_rt_launch_f32(c, pl, F, Binit, Bout);
    else rcppml_rt_launch_f64(c, pl, F, Binit, Bout);
    if (G.P > 1) {
        const int64_t n4 = nt * pl->k / RtVec<T>::N;
        hipLaunchKernelGGL(rhs_tiled_reduce_kernel<T>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, c->stream,
                           (const T*)pl->Bp, G.P, n4, ov ? 1 : 0, B);
        HIPCHK(hipGetLastError());
    }
}

template <class T>
rcppml_rhs_plan* build_plan(rcppml_hip_ctx* c, int dtype, const int* colptr, const int* rowidx, const T* vals, int64_t ncols_in,
                            int64_t nrows, int k, int partitions, int force_S) {
    int64_t ncols = ncols_in;
    const int rowb = k * (int)sizeof(T);
    if (rowb != 256 && rowb != 512 && !(rowb == 1024 && sizeof(T) == 8)) return nullptr;   // rows of one, two (or, fp64 k = 128, four) 256-byte slices
    if (ncols <= 0 || nrows <= 0) return nullptr;
    if (nrows * (int64_t)rowb < 16) return nullptr;
    int nnz_i = 0;
    HIPCHK(hipMemcpyAsync(&nnz_i, colptr + ncols, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nnz_i <= 0) return nullptr;
    std::unique_ptr<rcppml_rhs_plan> pl(new rcppml_rhs_plan());
    pl->dtype = dtype; pl->k = k; pl->device = c->device; pl->nnz = nnz_i;
    RhsTiledGeom& G = pl->G;
    G.ncols = ncols; G.nrows = nrows; G.rowb = rowb;
    const int R = RT_SLAB_BYTES / rowb;
    G.rshift = 0;
    while ((1 << G.rshift) < R) ++G.rshift;
    G.ntiles = (int)((nrows + R - 1) / R);
    // partitions: one per XCD when the factor is far beyond an XCD's 4 MiB L2, else a single one
    int P = partitions;
    if (P <= 0) P = (nrows * (int64_t)rowb > (12ll << 20)) ? 8 : 1;
    if (P > G.ntiles) P = G.ntiles;
    G.P = P;
    // segment-length histogram -> S
    DevTmp dh(c, (RT_MAX_HIST + 1) * sizeof(unsigned long long));
    HIPCHK(hipMemsetAsync(dh.p, 0, (RT_MAX_HIST + 1) * sizeof(unsigned long long), c->stream));
    unsigned long long* dhist = (unsigned long long*)dh.p;
    int* dflag = (int*)(dhist + RT_MAX_HIST);
    const unsigned gcol = (unsigned)((ncols + 3) / 4);
    hipLaunchKernelGGL(rhs_tiled_hist_kernel, dim3(gcol), dim3(256), 0, c->stream, colptr, rowidx, ncols, G.rshift, dhist, dflag);
    HIPCHK(hipGetLastError());
    unsigned long long hh[RT_MAX_HIST + 1];
    HIPCHK(hipMemcpyAsync(hh, dh.p, sizeof(hh), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((int)(hh[RT_MAX_HIST] & 0xffffffffull) != 0) return nullptr;             // rows not sorted inside a column
    const double nseg = (double)ncols * (double)G.ntiles;
    const int cand[6] = {2, 3, 4, 5, 6, 8};
    int S = 0;
    double best = 0, ovf_est = 0;
    auto spilled = [&](int s) { double o = 0; for (int b = s + 1; b < RT_MAX_HIST; ++b) o += (double)hh[b] * (b - s); return o; };
    for (int ci = 0; ci < 6; ++ci) {
        const int s = cand[ci];
        const double ovf = spilled(s);
        const double cost = nseg * s + 6.0 * ovf;          // a spilled nonzero costs about six slot steps in the gather kernel
        if (S == 0 || cost < best) { S = s; best = cost; ovf_est = ovf; }
    }
    if (force_S > 0) {
        bool okS = false;
        for (int ci = 0; ci < 6; ++ci) okS |= cand[ci] == force_S;
        if (!okS) throw std::runtime_error("rhs_plan: slot count must be one of 2,3,4,5,6,8");
        S = force_S;
        ovf_est = spilled(S);
    }
    G.S = S;
    G.dbg = 0;
#ifdef RCPPML_EXPERIMENTS
    { const char* e = getenv("RCPPML_RT_DBG"); G.dbg = e ? atoi(e) : 0; }
#endif
    // Workgroup shape (NW waves x NR rounds, 4 NR NW columns per workgroup) and how many columns go through the tiles.
    // Model (cycles per SIMD and tile, fitted to rocprofv3 on C2): waves per SIMD x (fixed 250 + 24 per step and 256-byte
    // slice of the row); workgroups run in rounds of num_cu / P.  Columns that would only fill part of a last round are cheaper
    // in the workgroup-per-column gather kernel (25 ps per nonzero and slice on the whole chip: 10 us for C2's 1 696 tail
    // columns) than as a round of their own -- unless there are many of them: fp64 at k = 64 on C2's W side left 4 640 of
    // 20 000 columns (0.23 ms of gathers) to that kernel while the slices were not in the model; it now takes two rounds.
    const int NV = rowb / 256;
    const int64_t conc = std::max<int64_t>(1, c->num_cu / P);                 // workgroups of one partition resident at once
    const double tiles_pp = (double)G.ntiles / P;
    const double nnz_per_col = (double)pl->nnz / (double)ncols;
    double best_t = -1;
    int bestNW = 0, bestnr = 0;
    int64_t best_tiled = 0;
    for (int NW = 16; NW >= 12; NW -= 4)
        for (int nr = 2; nr <= 20; ++nr) {
            if (!rt_launch::shape_ok(NV, S, NW, nr, (int)sizeof(T))) continue;
            const int64_t cap = 4ll * nr * NW;
            const double t_wg = tiles_pp * (NW / 4) * (250.0 + 24.0 * NV * nr * S) / 2400.0 + 15.0;       // microseconds
            const int64_t full_rounds = ncols / (cap * conc);
            for (int opt = 0; opt < 2; ++opt) {
                int64_t tiled;
                double t;
                if (opt == 0) {                       // every column through the tiles
                    tiled = ncols;
                    const int64_t ncb = (ncols + cap - 1) / cap;
                    t = (double)((ncb + conc - 1) / conc) * t_wg;
                } else {                              // whole rounds only, the rest to the gather kernel
                    if (full_rounds < 1) continue;
                    tiled = full_rounds * cap * conc;
                    t = (double)full_rounds * t_wg + 5.0 + (double)(ncols - tiled) * nnz_per_col * 25e-6 * NV;
                }
                if (best_t < 0 || t < best_t) { best_t = t; bestNW = NW; bestnr = nr; best_tiled = tiled; }
            }
        }
    if (best_t < 0) return nullptr;
    if (force_S <= 0) {
        // Is the input dense enough for fixed slots at all?  The slot stream holds ncols x ntiles x S slots whatever nnz is: a
        // hypersparse matrix (200 000 x 200 000 with 1.1 M nonzeros: 313 M slots, 1.9 GB, ~1 ms per product against ~14 us for
        // the gather kernel) must not get a plan.  Decline -- the caller then uses the gather kernel -- when the predicted fill
        // is below a quarter, when the slot stream would exceed four times the CSC itself or half of the free device memory,
        // or when the cost model says the gather kernel (12.5 ps per nonzero and 256-byte slice on the whole chip) is faster.
        const int64_t cap0 = 4ll * bestnr * bestNW;
        const double nslots_est = (double)((best_tiled + cap0 - 1) / cap0) * G.ntiles * bestNW * (double)(bestnr * S) * 4.0;
        const double fill_est = ((double)pl->nnz - ovf_est) * ((double)best_tiled / (double)ncols) / std::max(nslots_est, 1.0);
        const double stream_bytes = nslots_est * (sizeof(T) + 2.0);
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        // (small plans -- the unit tests' matrices -- are harmless either way and are left alone)
        const bool big = stream_bytes > (double)(32u << 20);
        if (stream_bytes > 0.5 * (double)free_b ||
            (big && (fill_est < 0.25 || stream_bytes > 4.0 * (double)pl->nnz * (sizeof(T) + 4.0) || best_t > (double)pl->nnz * 12.5e-6 * NV + 5.0)))
            return nullptr;
    }
#ifdef RCPPML_EXPERIMENTS
    { const char* e1 = getenv("RCPPML_RT_NW"); const char* e2 = getenv("RCPPML_RT_NR");
      if (e1 && e2 && rt_launch::shape_ok(NV, S, atoi(e1), atoi(e2), (int)sizeof(T))) { bestNW = atoi(e1); bestnr = atoi(e2); best_tiled = ncols; } }
#endif
    const int64_t bestcap = 4ll * bestnr * bestNW;
    G.NW = bestNW; G.nr = bestnr;
    G.ncols_tiled = best_tiled;
    G.ncb = (int)((best_tiled + bestcap - 1) / bestcap);
    pl->colptr = colptr; pl->rowidx = rowidx; pl->vals = vals;
    const int64_t ncols_all = ncols;
    ncols = best_tiled;                                   // from here on: the tiled columns only
    const unsigned gcol2 = (unsigned)((ncols + 3) / 4);

    // overflow column pointers
    DevTmp cnt(c, ((size_t)ncols + 1) * sizeof(int));
    HIPCHK(hipMemsetAsync(cnt.p, 0, ((size_t)ncols + 1) * sizeof(int), c->stream));
    hipLaunchKernelGGL(rhs_tiled_ovcount_kernel, dim3(gcol2), dim3(256), 0, c->stream, colptr, rowidx, ncols, G.rshift, S, (int*)cnt.p);
    HIPCHK(hipGetLastError());
    plan_alloc(c, pl.get(), &pl->ovptr, ((size_t)ncols + 1) * sizeof(int));
    {
        exclusive_scan_i32(c, (const int*)cnt.p, pl->ovptr, ncols + 1);
        int ovn = 0;
        HIPCHK(hipMemcpyAsync(&ovn, pl->ovptr + ncols, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        pl->ovnnz = ovn;
    }
    (void)ncols_all;
    pl->ov_fraction = (double)pl->ovnnz / (double)pl->nnz;
    if (force_S <= 0 && pl->ov_fraction > 0.35) return nullptr;   // too irregular for fixed slots: the gather kernel is the better tool
    pl->nslots = (int64_t)G.ncb * G.ntiles * G.NW * (int64_t)(G.nr * S) * 4;
    pl->fill = (double)(pl->nnz - pl->ovnnz) / (double)pl->nslots;
    const size_t nalloc = (size_t)pl->nslots + 64;
    plan_alloc(c, pl.get(), &pl->svals, nalloc * sizeof(T));
    plan_alloc(c, pl.get(), &pl->soffs, nalloc * sizeof(uint16_t));
    HIPCHK(hipMemsetAsync(pl->svals, 0, nalloc * sizeof(T), c->stream));
    HIPCHK(hipMemsetAsync(pl->soffs, 0, nalloc * sizeof(uint16_t), c->stream));
    plan_alloc(c, pl.get(), &pl->ovrow, (size_t)std::max<int64_t>(pl->ovnnz, 1) * sizeof(int));
    plan_alloc(c, pl.get(), &pl->ovval, (size_t)std::max<int64_t>(pl->ovnnz, 1) * sizeof(T));
    hipLaunchKernelGGL(rhs_tiled_fill_kernel<T>, dim3(gcol2), dim3(256), 0, c->stream, colptr, rowidx, vals, G, (T*)pl->svals,
                       pl->soffs, (const int*)pl->ovptr, pl->ovrow, (T*)pl->ovval);
    HIPCHK(hipGetLastError());
    if (G.P > 1) plan_alloc(c, pl.get(), &pl->Bp, (size_t)G.P * (size_t)ncols * (size_t)k * sizeof(T));     // ncols = tiled columns
    HIPCHK(hipStreamSynchronize(c->stream));          // temporaries die here
    return pl.release();
}

}  // namespace

void rcppml_rt_launch_f32(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const float* F, const float* Binit, float* Bout) {
    rt_launch::launch_tiled_any<float>(c, pl, F, Binit, Bout);
}

extern "C" int rcppml_hip_rhs_plan_create(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx,
                                          const void* values, int64_t ncols, int64_t nrows, int k, int partitions, int slots,
                                          rcppml_rhs_plan** out) {
    try {
        if (!out) throw std::runtime_error("rhs_plan_create: null output");
        *out = nullptr;
        HIPCHK(hipSetDevice(c->device));
        if (ncols > 0x7ffffff0ll || nrows > 0x7ffffff0ll) return 0;
        // slots: 0 = choose (window plan first, slab plan if the window planner declines); 1 = slab plan, its own choice of S;
        // 2..8 = slab plan with S slots per (column, tile); >= 100 = window plan with (slots - 100) / 4 slots per column and phase
        if (slots == 0 || slots >= 100) {
            const int rate_code = slots >= 100 ? slots - 100 : 0;
            if (dtype == RCPPML_F32) *out = rcppml_rw_build_f32(c, col_ptr, row_idx, (const float*)values, ncols, nrows, k, partitions, rate_code);
            else *out = rcppml_rw_build_f64(c, col_ptr, row_idx, (const double*)values, ncols, nrows, k, partitions, rate_code);
            if (*out || slots >= 100) return 0;
        }
        const int S = slots == 1 ? 0 : slots;
        if (dtype == RCPPML_F32) *out = build_plan<float>(c, dtype, col_ptr, row_idx, (const float*)values, ncols, nrows, k, partitions, S);
        else *out = build_plan<double>(c, dtype, col_ptr, row_idx, (const double*)values, ncols, nrows, k, partitions, S);
        return 0;
    }
    RCPPML_CATCH_RET
}

// The plan in two steps (window plans only): everything that needs only the INDEX arrays now, the values later -- the plugin
// builds both plans while the values are still crossing PCIe.  *out = NULL when the window planner declines (the caller then
// uses rcppml_hip_rhs_plan_create once the values are there).
extern "C" int rcppml_hip_rhs_plan_create_indices(rcppml_hip_ctx* c, int dtype, const int* col_ptr, const int* row_idx, int64_t ncols,
                                                  int64_t nrows, int k, int partitions, int slots, rcppml_rhs_plan** out) {
    try {
        if (!out) throw std::runtime_error("rhs_plan_create_indices: null output");
        *out = nullptr;
        HIPCHK(hipSetDevice(c->device));
        if (ncols > 0x7ffffff0ll || nrows > 0x7ffffff0ll) return 0;
        if (slots != 0 && slots < 100) return 0;                 // slab plans are built in one step
        const int rate_code = slots >= 100 ? slots - 100 : 0;
        if (dtype == RCPPML_F32) *out = rcppml_rw_build_f32(c, col_ptr, row_idx, nullptr, ncols, nrows, k, partitions, rate_code);
        else *out = rcppml_rw_build_f64(c, col_ptr, row_idx, nullptr, ncols, nrows, k, partitions, rate_code);
        return 0;
    }
    RCPPML_CATCH_RET
}
extern "C" int rcppml_hip_rhs_plan_set_values(rcppml_hip_ctx* c, rcppml_rhs_plan* plan, const void* values) {
    try {
        if (!plan || !values) throw std::runtime_error("rhs_plan_set_values: null argument");
        if (plan->device != c->device) throw std::runtime_error("rhs_plan_set_values: plan belongs to another device");
        HIPCHK(hipSetDevice(c->device));
        if (plan->dtype == RCPPML_F32) rcppml_rw_set_values_f32(c, plan, (const float*)values);
        else rcppml_rw_set_values_f64(c, plan, (const double*)values);
        return 0;
    }
    RCPPML_CATCH_RET
}

extern "C" void rcppml_hip_rhs_plan_destroy(rcppml_rhs_plan* plan) { delete plan; }

extern "C" int rcppml_hip_rhs_plan_kind(const rcppml_rhs_plan* pl) { return pl ? pl->kind : -1; }
extern "C" int rcppml_hip_rhs_plan_info(const rcppml_rhs_plan* pl, double* out10 /* 11 doubles */) {
    if (!pl || !out10) return 1;
    if (pl->kind == 1) {
        const RhsWinGeom& W = pl->WG;
        out10[0] = W.P; out10[1] = W.NW; out10[2] = W.nr; out10[3] = W.clo + W.nhi / 4.0; out10[4] = W.ncb; out10[5] = W.ntiles;
        out10[6] = (double)pl->nslots; out10[7] = (double)pl->ovnnz; out10[8] = pl->fill; out10[9] = pl->stream_bytes;
        out10[10] = (double)W.ncols;
        return 0;
    }
    const RhsTiledGeom& G = pl->G;
    out10[0] = G.P; out10[1] = G.NW; out10[2] = G.nr; out10[3] = G.S; out10[4] = G.ncb; out10[5] = G.ntiles;
    out10[6] = (double)pl->nslots; out10[7] = (double)pl->ovnnz; out10[8] = pl->fill;
    out10[10] = (double)G.ncols_tiled;
    out10[9] = (double)pl->nslots * ((pl->dtype == RCPPML_F32 ? 4 : 8) + 2);      // bytes of the slot stream
    return 0;
}

extern "C" int rcppml_hip_rhs_planned(rcppml_hip_ctx* c, const rcppml_rhs_plan* plan, const void* F, void* B) {
    try {
        if (!plan) throw std::runtime_error("rhs_planned: null plan");
        if (plan->device != c->device) throw std::runtime_error("rhs_planned: plan belongs to another device");
        if (plan->dest && !plan->vals) throw std::runtime_error("rhs_planned: the plan has no values yet (rcppml_hip_rhs_plan_set_values)");
        if (reinterpret_cast<uintptr_t>(F) % 16 || reinterpret_cast<uintptr_t>(B) % 16)
            throw std::runtime_error("rhs_planned: F and B must be 16-byte aligned");
        HIPCHK(hipSetDevice(c->device));
        if (plan->kind == 1) {
            if (plan->dtype == RCPPML_F32) rcppml_rw_run_f32(c, plan, (const float*)F, (float*)B);
            else rcppml_rw_run_f64(c, plan, (const double*)F, (double*)B);
        } else if (plan->dtype == RCPPML_F32) run_plan<float>(c, plan, (const float*)F, (float*)B);
        else run_plan<double>(c, plan, (const double*)F, (double*)B);
        return 0;
    }
    RCPPML_CATCH_RET
}

