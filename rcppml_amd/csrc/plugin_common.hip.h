// plugin_common.hip.h -- helpers shared by the plugin's translation units (plugin.hip: the single-device ALS loop and the
// entry points; plugin_multi.hip: the multi-device loop): device buffers, context guard, the parameter carrier of a fit
// and the host<->device conversions of the boundary's double buffers.
#pragma once
#include <memory>
#include <thread>
#include "common.hip.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace rcppml_plugin {

#define OPCHK(expr)                                                                  \
    do {                                                                             \
        if ((expr) != 0) throw std::runtime_error(std::string(#expr) + ": " + rcppml_err()); \
    } while (0)

// the context whose per-fit arena (common.hip.h) serves this thread's DevBuf allocations; set by CtxGuard::reserve
inline rcppml_hip_ctx*& devbuf_arena_ctx() {
    static thread_local rcppml_hip_ctx* c = nullptr;
    return c;
}
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() {}
    explicit DevBuf(size_t b) { alloc(b); }
    void alloc(size_t b) {
        release();
        bytes = b < 16 ? 16 : b;
        rcppml_hip_ctx* ac = devbuf_arena_ctx();
        p = ac ? ac->arena_take(bytes) : nullptr;
        if (p) { owned = false; return; }
        HIPCHK(hipMalloc(&p, bytes));
    }
    bool owned = true;
    void borrow(const void* ptr) { release(); p = const_cast<void*>(ptr); owned = false; }   // caller-owned device memory
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; bytes = 0; owned = true; }
    ~DevBuf() { release(); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    template <class U> U* as() const { return static_cast<U*>(p); }
};

struct CtxGuard {
    rcppml_hip_ctx* c = nullptr;
    hipStream_t s = nullptr;
    explicit CtxGuard(int device) {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        if (rcppml_hip_ctx_create(&c, device, s) != 0) {
            (void)hipStreamDestroy(s);
            throw std::runtime_error("ctx_create: " + rcppml_err());
        }
    }
    // a second stream for PCIe copies that should run beside device work of the fit's stream (created on first use)
    hipStream_t s2 = nullptr;
    hipStream_t second_stream() {
        if (!s2) HIPCHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        return s2;
    }
    // One device allocation for the whole fit (see rcppml_hip_ctx::arena).  Best effort: if it cannot be had, every buffer
    // falls back to its own hipMalloc.
    void reserve(size_t bytes) {
        // never more than 60 % of what is free: near device capacity the arena must not starve the allocations that do not go
        // through it (context scratch, plans that fall back to hipMalloc); what does not fit falls back to hipMalloc per buffer
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b / 10 * 6) bytes = free_b / 10 * 6;
        void* q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
        c->arena = static_cast<char*>(q); c->arena_cap = bytes; c->arena_off = 0;
        devbuf_arena_ctx() = c;
    }
    ~CtxGuard() {
        if (devbuf_arena_ctx() == c) devbuf_arena_ctx() = nullptr;
        void* arena = c ? c->arena : nullptr;
        if (c) { c->arena = nullptr; rcppml_hip_ctx_destroy(c); }
        if (arena) (void)hipFree(arena);
        if (s2) (void)hipStreamDestroy(s2);
        if (s) (void)hipStreamDestroy(s);
    }
};

inline int env_device() {
    const char* e = getenv("RCPPML_GPU_DEVICE");
    return e ? atoi(e) : 0;
}

// sides with at least this many columns run their CD solve in sweep-sorted column order (previous iteration's counts):
// from one 16-column wavefront per SIMD the three small ordering kernels pay (C2's 20 000-column W side: 0.350 -> 0.313 ms)
constexpr int64_t kOrderMinColumns = 16384;

struct FitParams {
    int m, n, k;
    int64_t nnz;
    const int* col_ptr; const int* row_idx; const double* values;
    double *W, *H, *d;            // in/out (host, double)
    int max_iter; double tol;
    double L1_H, L1_W, L2_H, L2_W, ub_H, ub_W;
    double L21_H = 0, L21_W = 0, angular_H = 0, angular_W = 0;
    // graph Laplacians (host CSC, dim x dim): graph_H over the columns of H (dim = n), graph_W over the columns of W_T (dim = m)
    const int* gH_p = nullptr; const int* gH_i = nullptr; const double* gH_x = nullptr; int gH_nnz = 0; double gH_lambda = 0;
    const int* gW_p = nullptr; const int* gW_i = nullptr; const double* gW_x = nullptr; int gW_nnz = 0; double gW_lambda = 0;
    int symmetric = 0;                       // A ~ W diag(d) W^T (A square): only W is solved, H = W_T
    const double* dense = nullptr;           // dense input (column-major m x n): the unfused standard path of fit_cpu.hpp
    int csc_on_device = 0;                   // col_ptr / row_idx / values are DEVICE pointers (zero-copy entry)
    int device = -1;                         // >= 0: run on this device (zero-copy: the device that owns the CSC); -1: RCPPML_GPU_DEVICE
    int projective = 0;                      // H = (diag(d) W_T) A instead of the NNLS half-update (variant_helpers.hpp:308-325)
    // target regularisation (variant_helpers.hpp:107-146): host matrices k x n / k x m (k leading), NULL = none
    const double* target_H = nullptr; double target_lambda_H = 0;
    const double* target_W = nullptr; double target_lambda_W = 0;
    int cd_maxit; double cd_tol;
    int verbose, patience, nonneg_W, nonneg_H, norm_type, solver_mode;
    const int* mask_p; const int* mask_i;   // NULL = no mask
    int sort_model;
    double* loss_history;                    // may be NULL
    int loss_type = 0;                       // 0 = MSE, 4 = GP, 5 = NB, 6 = Gamma, 7 = inverse Gaussian, 8 = Tweedie
    double tweedie_power = 1.5;
    double robust_delta = 0;                 // > 0: Huber modifier on Pearson residuals (all losses -> IRLS path)
    int irls_max_iter = 5; double irls_tol = 1e-4;
    int dispersion_mode = 2;                 // 0 none, 1 global, 2 per-row
    double nb_size_init = 10, nb_size_max = 1e6, nb_size_min = 0.01;
    double gp_theta_init = 0.1, gp_theta_max = 5.0;                       // core/config.hpp:169-172
    double gamma_phi_init = 1.0, gamma_phi_max = 1e4, gamma_phi_min = 1e-6;   // core/config.hpp:201-207
    double* out_theta = nullptr; int out_theta_len = 0;
    // outputs
    int out_iter = 0, out_converged = 0; double out_loss = 0, out_tol = 0;
};

template <class T> struct DT;
template <> struct DT<float> { static constexpr int id = RCPPML_F32; };
template <> struct DT<double> { static constexpr int id = RCPPML_F64; };

// The boundary hands over double buffers (bridge_nmf.hpp:310-342).  They are uploaded as they are and cast on the
// device (no host-side temporaries, no host loops over nnz).
template <class T>
inline void upload_cast(rcppml_hip_ctx* c, const double* src, size_t n, DevBuf& dst, hipStream_t s) {
    dst.alloc(n * sizeof(T));
    if constexpr (std::is_same<T, double>::value) {
        HIPCHK(hipMemcpyAsync(dst.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    } else {
        DevBuf stage(n * sizeof(double));
        HIPCHK(hipMemcpyAsync(stage.p, src, n * sizeof(double), hipMemcpyHostToDevice, s));
        if (s != c->stream) {
            // copy on a side stream (it overlaps whatever the fit's stream is running), cast behind that work on the fit's stream;
            // with an arena the staging copy outlives this call, so nothing waits for the cast here
            HIPCHK(hipStreamSynchronize(s));
            OPCHK(rcppml_hip_cast(c, RCPPML_F64, stage.p, RCPPML_F32, dst.p, (int64_t)n));
            if (stage.owned) HIPCHK(hipStreamSynchronize(c->stream));
            return;
        }
        OPCHK(rcppml_hip_cast(c, RCPPML_F64, stage.p, RCPPML_F32, dst.p, (int64_t)n));
        HIPCHK(hipStreamSynchronize(s));
    }
}
// Host -> device copies of double buffers on a helper thread and a side stream, so that the calling thread can go on
// building things on the fit's stream (pageable copies block the thread that issues them): add() allocates (calling
// thread: the arena is per thread), start() launches the copies, finish() joins and enqueues the precision casts.
template <class T>
struct AsyncUpload {
    struct Item { const double* src; size_t n; DevBuf* dst; DevBuf stage; };
    std::vector<std::unique_ptr<Item>> items;
    std::thread th;
    std::string err;
    void add(const double* src, size_t n, DevBuf& dst) {
        std::unique_ptr<Item> it(new Item{src, n, &dst, DevBuf()});
        dst.alloc(n * sizeof(T));
        if (!std::is_same<T, double>::value) it->stage.alloc(n * sizeof(double));
        items.push_back(std::move(it));
    }
    void start(int device, hipStream_t side) {
        th = std::thread([this, device, side] {
            try {
                HIPCHK(hipSetDevice(device));
                for (auto& it : items)
                    HIPCHK(hipMemcpyAsync(std::is_same<T, double>::value ? it->dst->p : it->stage.p, it->src, it->n * sizeof(double),
                                          hipMemcpyHostToDevice, side));
                HIPCHK(hipStreamSynchronize(side));
            } catch (const std::exception& e) { err = e.what(); }
        });
    }
    void finish(rcppml_hip_ctx* c) {
        if (th.joinable()) th.join();
        if (!err.empty()) throw std::runtime_error(err);
        if (!std::is_same<T, double>::value) {
            bool owned = false;
            for (auto& it : items) {
                OPCHK(rcppml_hip_cast(c, RCPPML_F64, it->stage.p, RCPPML_F32, it->dst->p, (int64_t)it->n));
                owned |= it->stage.owned;
            }
            if (owned) HIPCHK(hipStreamSynchronize(c->stream));      // hipMalloc'ed staging copies are freed with `items`
        }
        items.clear();
    }
    ~AsyncUpload() { if (th.joinable()) th.join(); }
};
template <class T>
inline void download_cast(rcppml_hip_ctx* c, const DevBuf& src, size_t n, double* dst, hipStream_t s) {
    if constexpr (std::is_same<T, double>::value) {
        HIPCHK(hipMemcpyAsync(dst, src.p, n * sizeof(T), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    } else {
        DevBuf stage(n * sizeof(double));
        OPCHK(rcppml_hip_cast(c, RCPPML_F32, src.p, RCPPML_F64, stage.p, (int64_t)n));
        HIPCHK(hipMemcpyAsync(dst, stage.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
}
inline void upload_ints(const int* src, size_t n, DevBuf& dst, hipStream_t s) {
    dst.alloc(n * sizeof(int));
    HIPCHK(hipMemcpyAsync(dst.p, src, n * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
}

// The LDS row-tiled plan is an optimisation: a plan that cannot be built (out of device memory, unsupported shape) is "no
// plan" -- the products then run on the gather kernel -- never a failed fit.
inline void plan_or_none(int rc, rcppml_rhs_plan*& plan) {
    if (rc != 0) {
        plan = nullptr;
        if (const char* v = getenv("RCPPML_GPU_VERBOSE"); v && atoi(v) >= 1)
            fprintf(stderr, "[rcppml_gpu] row-tiled rhs plan dropped (%s): this side runs the gather kernel\n", rcppml_err().c_str());
        rcppml_err().clear();
        (void)hipGetLastError();          // a failed hipMalloc leaves a sticky-until-read error behind
    }
}

}  // namespace rcppml_plugin
using namespace rcppml_plugin;

// Multi-device fit (plugin_multi.hip): plain sparse MSE fits sharded over `ndev` devices of this process, RCCL between
// them.  Returns false when the configuration is not one it handles (the caller then runs the single-device loop).
bool rcppml_fit_multi(FitParams& P, int precision, int ndev);
