// common.hip.h -- context, error plumbing and scratch memory shared by ops.hip and plugin.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/rcppml_gpu.h"

// thread-local last error (never throw across the C ABI: reference src/gpu_bridge_nmf.cu:206-209)
inline std::string& rcppml_err() {
    static thread_local std::string e;
    return e;
}

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) +  \
                                     " (" __FILE__ ":" + std::to_string(__LINE__) + ")");        \
    } while (0)

#define RCPPML_CATCH_RET                                      \
    catch (const std::exception& e) {                         \
        rcppml_err() = e.what();                              \
        return 1;                                             \
    }                                                         \
    catch (...) {                                             \
        rcppml_err() = "unknown error";                       \
        return 1;                                             \
    }

// Experiment switches (kernel variants kept for probes) exist only in -DRCPPML_EXPERIMENTS builds; the shipping library
// never reads them.
#ifdef RCPPML_EXPERIMENTS
inline bool exp_flag(const char* name, const char* value) { const char* e = getenv(name); return e && !strcmp(e, value); }
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline bool exp_flag(const char*, const char*) { return false; }
inline const char* exp_env(const char*) { return nullptr; }
#endif

struct rcppml_hip_ctx;
enum { WS_GRAM = 0, WS_GPAD, WS_CHOL, WS_RED, WS_RED2, WS_ORDER, WS_IRLS, WS_MFMA, WS_FEAT, WS_GRAPH, WS_COUNT };

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: remember the largest size
// requested per device (one static instance per kernel instantiation) instead of a process-wide "done" flag.
struct DynSmemOnce {
    size_t set_bytes[32] = {};
    std::mutex mu;                  // fits may run concurrently from several host threads (rcppml_err is thread_local)
    void ensure(const void* fn, size_t smem, int device) {
        const int d = device & 31;
        std::lock_guard<std::mutex> lk(mu);
        if (smem > 48 * 1024 && smem > set_bytes[d]) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                // e.g. the rank-65..128 IRLS / CV / mask kernels (68 KB fp32, 136 KB fp64 per workgroup) on a 64 KB-LDS device
                throw std::runtime_error("this kernel needs " + std::to_string(smem) + " bytes of LDS per workgroup, more than device " +
                                         std::to_string(device) + " grants (" + hipGetErrorString(e) +
                                         "): the configuration (rank above 64?) needs gfx950's 160 KiB of LDS per CU");
            }
            set_bytes[d] = smem;
        }
    }
};

struct rcppml_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cu = 256;
    // user mask of a cross-validation fit (rcppml_hip_ctx_set_cv_mask): pattern CSC of the mask [0] and of its transpose [1], device
    // pointers owned by the caller; the CV ops read the one that matches the CSC they are handed (their `transposed` flag)
    const int* cv_mask_p[2] = {nullptr, nullptr};
    const int* cv_mask_i[2] = {nullptr, nullptr};
    struct Buf { void* ptr = nullptr; size_t bytes = 0; };
    Buf bufs[WS_COUNT];
    // device counters read by rcppml_hip_ctx_stats: [0] column-sweeps executed by the CD kernels, [1] columns solved
    //   [2] slot-sweeps executed by the persistent CD kernel (idle and correction sweeps included), [3] coordinate steps of
    //   that kernel in which no column of the wave moved (only counted when opt_cd_count is set)
    unsigned long long* stats = nullptr;
    // rcppml_hip_ctx_set_option
    int opt_cd_count = 0, opt_lmf_lg = 0, opt_lmf_wps = 0, opt_cd_no_lmf = 0, opt_irls_cpw = 0, opt_small_give_up = 0;
    // Per-fit arena (plugin entries): ONE hipMalloc / hipFree for everything a fit allocates instead of ~60 pairs -- hipMalloc
    // costs tens of microseconds and every hipFree synchronises the device; together they were 5-6 ms of a 20 ms one-iteration
    // call.  Pure bump allocation, nothing is handed back before the fit ends.  arena_take() returns nullptr when there is no
    // arena (the harness's contexts) or it is full: callers then fall back to hipMalloc.
    char* arena = nullptr;
    size_t arena_cap = 0, arena_off = 0;
    void* arena_take(size_t bytes) {
        const size_t a = (arena_off + 255) & ~(size_t)255;
        if (!arena || a + bytes > arena_cap) return nullptr;
        arena_off = a + bytes;
        return arena + a;
    }
    // Grow-only scratch.  Growth frees the old block with hipFree, which synchronises the device,
    // so no in-flight kernel can still be using it.
    void* scratch(int slot, size_t bytes) {
        Buf& b = bufs[slot];
        if (bytes > b.bytes) {
            if (b.ptr) HIPCHK(hipFree(b.ptr));
            b.ptr = nullptr; b.bytes = 0;
            size_t want = bytes < 4096 ? 4096 : bytes;
            HIPCHK(hipMalloc(&b.ptr, want));
            b.bytes = want;
        }
        return b.ptr;
    }
};

// Device temporary of a setup routine: from the context's arena when there is one, else hipMalloc / hipFree.
struct DevTmp {
    void* p = nullptr;
    bool owned = false;
    DevTmp(rcppml_hip_ctx* c, size_t bytes) {
        if (bytes < 16) bytes = 16;
        p = c ? c->arena_take(bytes) : nullptr;
        if (!p) { HIPCHK(hipMalloc(&p, bytes)); owned = true; }
    }
    ~DevTmp() { if (p && owned) (void)hipFree(p); }
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
};
