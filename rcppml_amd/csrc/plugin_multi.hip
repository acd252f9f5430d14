// ============================================================================
// plugin_multi.hip -- the ALS loop of the plugin over several devices of ONE process (RCPPML_GPU_DEVICES=n), so that the R
// caller's 73-pointer call shards without any change on its side.  The reference is single-GPU (core/resources.hpp:
// 205-210, config.max_gpus is never read): this is the north star's design, SURVEY.md 8(e), the same scheme as the
// harness loop in rcppml_amd/als.py:
//   * columns of A and H in contiguous, nnz-balanced shards, one per device; W_T replicated;
//   * H half-update: no communication; its row scaling needs the k row sums over all shards: one all-reduce of k values;
//   * W half-update: every device forms H_loc H_loc^T and H_loc A_loc^T into ONE buffer [G_p | B_p] -> one all-reduce
//     (RCCL over xGMI; 5.1 MB fp32 at C2) -> every device holds the full Gram and right-hand side and solves the m columns
//     of W redundantly (m << n): identical inputs and deterministic kernels give bitwise identical W_T on every device,
//     which the loop relies on (no broadcast) and tests/test_gpu_plugin_multi.py asserts;
//   * loss: Gram trick from replicated quantities, read from device 0.
// N devices differ from one device only by the summation order of the two reduced buffers.
// Scope: the plain sparse MSE fit (CD or Cholesky, L1/L2/bounds, any norm).  Masks, IRLS losses, graph / L21 / angular
// terms, dense input, projective and symmetric fits need whole rows or other global state on the W side: the caller keeps
// the single-device loop for them (rcppml_fit_multi returns false).
// RCCL is loaded at run time (dlopen) -- the single-device plugin has no dependency on it.
// RCPPML_GPU_DEVICES_SHARE=1 maps every shard onto device 0 and replaces the collectives by a local sum kernel: the
// sharded loop can then be exercised on a one-GPU box (RCCL refuses two ranks on one device).
// RCPPML_GPU_DEVICES_FORCE=1 lets RCPPML_GPU_DEVICES=1 through this loop with a ONE-rank RCCL communicator whose collectives
// are really issued (a one-rank sum is the identity): dlopen, the seven dlsym's, ncclCommInitAll and the grouped
// ncclAllReduce / ncclAllGather then execute on a one-GPU box (tests/test_gpu_plugin_multi.py).
// ============================================================================
#include "plugin_common.hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <memory>

namespace {

template <class T>
__global__ void mg_diag_add_kernel(T* __restrict__ G, int k, T v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) G[(size_t)i * k + i] += v;
}
// G(f, g) /= d_f d_g: the scaling D^-1 applied to a Gram that was formed (and all-reduced) from the unscaled factor
template <class T>
__global__ void mg_scale_gram_kernel(T* __restrict__ G, int k, const T* __restrict__ d) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < k * k) G[e] = G[e] / d[e % k] / d[e / k];
}
// out[r][i] = sum_r in[r][i] for every replica r (shared-device stand-in of the all-reduce; fixed order)
template <class T>
__global__ void mg_local_sum_kernel(T* const* __restrict__ bufs, int n, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T s = bufs[0][i];
    for (int r = 1; r < n; ++r) s += bufs[r][i];
    for (int r = 0; r < n; ++r) bufs[r][i] = s;
}

// 64-bit checksum of a buffer's bit patterns (position-weighted, integer sum: independent of the order the blocks finish in):
// the replicas of W_T must agree BIT FOR BIT on every device -- the replicated W solve relies on it (no broadcast)
template <class T>
__global__ void mg_checksum_kernel(const T* __restrict__ x, size_t count, unsigned long long* __restrict__ out) {
    unsigned long long h = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long b;
        if constexpr (sizeof(T) == 4) b = (unsigned long long)__float_as_uint((float)x[i]);
        else b = (unsigned long long)__double_as_longlong((double)x[i]);
        h += b * (2ull * (unsigned long long)i + 1ull);
    }
    for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

struct Rccl {
    void* h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void load() {
        if (h) return;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) throw std::runtime_error(std::string("RCPPML_GPU_DEVICES: cannot load RCCL: ") + dlerror());
        auto sym = [&](const char* s) {
            void* p = dlsym(h, s);
            if (!p) throw std::runtime_error(std::string("RCCL symbol missing: ") + s);
            return p;
        };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    }
    void chk(ncclResult_t r, const char* what) {
        if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error"));
    }
};
Rccl& rccl() { static Rccl r; return r; }

// all-reduce(sum) of one buffer per device; `shared`: all shards live on one device (test mode)
struct Exchange {
    int n = 0;
    bool shared = false;
    bool always = false;               // issue the collectives for n == 1 as well (RCPPML_GPU_DEVICES_FORCE)
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<int> devs;
    DevBuf ptrs;                       // shared mode: device array of the n buffer pointers
    std::vector<hipEvent_t> ev, ev2;
    ~Exchange() {
        for (auto c : comms) if (c) (void)rccl().CommDestroy(c);
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
        for (auto e : ev2) if (e) (void)hipEventDestroy(e);
    }
    void init(const std::vector<int>& devices, const std::vector<hipStream_t>& st, bool share, bool force) {
        n = (int)devices.size(); devs = devices; streams = st; shared = share; always = force;
        if (shared) {
            HIPCHK(hipSetDevice(devs[0]));
            ptrs.alloc((size_t)n * sizeof(void*));
            ev.resize(n + 1, nullptr);
            for (auto& e : ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev2.resize(n, nullptr);
            for (auto& e : ev2) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        } else {
            rccl().load();
            comms.assign(n, nullptr);
            rccl().chk(rccl().CommInitAll(comms.data(), n, devs.data()), "ncclCommInitAll");
        }
    }
    template <class T>
    void all_reduce(const std::vector<void*>& bufs, size_t count) {
        if (n == 1 && !always) return;
        if (shared) {
            // every stream's work on its buffer -> stream 0 sums -> every stream waits for the sum
            HIPCHK(hipSetDevice(devs[0]));
            for (int r = 1; r < n; ++r) { HIPCHK(hipEventRecord(ev[r], streams[r])); HIPCHK(hipStreamWaitEvent(streams[0], ev[r], 0)); }
            HIPCHK(hipMemcpyAsync(ptrs.p, bufs.data(), (size_t)n * sizeof(void*), hipMemcpyHostToDevice, streams[0]));
            hipLaunchKernelGGL(mg_local_sum_kernel<T>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, streams[0],
                               (T* const*)ptrs.p, n, count);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(ev[n], streams[0]));
            for (int r = 1; r < n; ++r) HIPCHK(hipStreamWaitEvent(streams[r], ev[n], 0));
            HIPCHK(hipStreamSynchronize(streams[0]));       // `bufs.data()` (host vector) was the copy source
            return;
        }
        const ncclDataType_t ty = std::is_same<T, float>::value ? ncclFloat : ncclDouble;
        rccl().chk(rccl().GroupStart(), "ncclGroupStart");
        for (int r = 0; r < n; ++r)
            rccl().chk(rccl().AllReduce(bufs[r], bufs[r], count, ty, ncclSum, comms[r], streams[r]), "ncclAllReduce");
        rccl().chk(rccl().GroupEnd(), "ncclGroupEnd");
    }
    // In-place all-gather of equal blocks: block r (count elements at bufs[q] + r * count) of device r goes to every device q.
    template <class T>
    void all_gather(const std::vector<void*>& bufs, size_t count) {
        if (n == 1 && !always) return;
        if (shared) {
            // every stream has written its own block -> each stream copies the other blocks once their owners are done
            for (int r = 0; r < n; ++r) { HIPCHK(hipSetDevice(devs[r])); HIPCHK(hipEventRecord(ev[r], streams[r])); }
            for (int q = 0; q < n; ++q) {
                HIPCHK(hipSetDevice(devs[q]));
                for (int r = 0; r < n; ++r) {
                    if (r == q) continue;
                    HIPCHK(hipStreamWaitEvent(streams[q], ev[r], 0));
                    HIPCHK(hipMemcpyAsync((T*)bufs[q] + (size_t)r * count, (const T*)bufs[r] + (size_t)r * count, count * sizeof(T),
                                          hipMemcpyDeviceToDevice, streams[q]));
                }
            }
            // a collective ends for everybody at once: no stream may go on (and scale its W in place) while another one is still
            // copying its block out of it
            for (int q = 0; q < n; ++q) { HIPCHK(hipSetDevice(devs[q])); HIPCHK(hipEventRecord(ev2[q], streams[q])); }
            for (int r = 0; r < n; ++r) {
                HIPCHK(hipSetDevice(devs[r]));
                for (int q = 0; q < n; ++q) if (q != r) HIPCHK(hipStreamWaitEvent(streams[r], ev2[q], 0));
            }
            return;
        }
        const ncclDataType_t ty = std::is_same<T, float>::value ? ncclFloat : ncclDouble;
        rccl().chk(rccl().GroupStart(), "ncclGroupStart");
        for (int r = 0; r < n; ++r)
            rccl().chk(rccl().AllGather((const T*)bufs[r] + (size_t)r * count, bufs[r], count, ty, comms[r], streams[r]), "ncclAllGather");
        rccl().chk(rccl().GroupEnd(), "ncclGroupEnd");
    }
};

template <class T>
struct Shard {
    int dev = 0;
    std::unique_ptr<CtxGuard> g;
    int c0 = 0, n_loc = 0;
    int64_t nnz_loc = 0;
    DevBuf Ap, Ai, Ax, Tp, Ti, Tx;
    DevBuf W, H, d, Bh, xbuf, G, Gs, Gwt, sums, tr, loss, swH, ordH, swW, ordW;
    rcppml_rhs_plan* planA = nullptr;
    rcppml_rhs_plan* planT = nullptr;
    ~Shard() {
        if (g) (void)hipSetDevice(dev);
        rcppml_hip_rhs_plan_destroy(planA);
        rcppml_hip_rhs_plan_destroy(planT);
    }
};

template <class T>
void fit_multi(FitParams& P, const std::vector<int>& devices, bool shared, bool force) {
    constexpr int dt = DT<T>::id;
    const int m = P.m, n = P.n, k = P.k, nd = (int)devices.size();
    const double eps = 1e-15;
    // ---- contiguous column shards balanced by nonzeros (als.partition_columns_by_nnz)
    std::vector<int> cut(nd + 1, 0);
    cut[nd] = n;
    {
        int j = 0;
        for (int r = 1; r < nd; ++r) {
            const int64_t target = P.nnz * r / nd;
            while (j < n && P.col_ptr[j] < target) ++j;
            cut[r] = std::max(cut[r - 1], std::min(j, n));
        }
    }
    // W half-update: "replicated" (default) = every device solves all m columns of W_T from the all-reduced (G, B);
    // RCPPML_GPU_W_SOLVE=block = device r solves rows [r rows_per, (r+1) rows_per) and ONE all-gather replicates them (the harness's
    // default, als.AlsConfig.w_solve).  Same numbers either way; which is faster over xGMI is for the first multi-GPU run to
    // decide -- unmeasured on hardware so far.
    const char* ws_env = getenv("RCPPML_GPU_W_SOLVE");
    if (ws_env && strcmp(ws_env, "block") && strcmp(ws_env, "replicated") && getenv("RCPPML_GPU_VERBOSE"))
        fprintf(stderr, "[rcppml_gpu] RCPPML_GPU_W_SOLVE=%s is neither 'block' nor 'replicated': using 'replicated'\n", ws_env);
    const bool w_block = (nd > 1 || force) && ws_env && !strcmp(ws_env, "block");
    const int rows_per = w_block ? (((m + nd - 1) / nd + 3) / 4) * 4 : m;         // multiples of 4 rows: 16-byte aligned blocks for any k
    const size_t w_elems = w_block ? (size_t)rows_per * nd * k : (size_t)k * m;
    std::vector<std::unique_ptr<Shard<T>>> S(nd);
    std::vector<hipStream_t> streams(nd);
    for (int r = 0; r < nd; ++r) {
        S[r].reset(new Shard<T>());
        Shard<T>& s = *S[r];
        s.dev = devices[r];
        s.g.reset(new CtxGuard(s.dev));
        streams[r] = s.g->s;
        rcppml_hip_ctx* c = s.g->c;
        s.c0 = cut[r]; s.n_loc = cut[r + 1] - cut[r];
        const int e0 = P.col_ptr[s.c0], e1 = P.col_ptr[cut[r + 1]];
        s.nnz_loc = e1 - e0;
        const size_t nz = (size_t)std::max<int64_t>(s.nnz_loc, 1);
        std::vector<int> p((size_t)s.n_loc + 1);
        for (int j = 0; j <= s.n_loc; ++j) p[j] = P.col_ptr[s.c0 + j] - e0;
        upload_ints(p.data(), p.size(), s.Ap, s.g->s);
        // (an empty shard -- e0 == nnz, or c0 == n when the last column alone holds more than nnz / nd entries -- uploads a dummy
        // element instead of reading one past the caller's arrays)
        const int zero_i = 0; const double zero_d = 0.0;
        upload_ints(s.nnz_loc > 0 ? P.row_idx + e0 : &zero_i, nz, s.Ai, s.g->s);
        upload_cast<T>(c, s.nnz_loc > 0 ? P.values + e0 : &zero_d, nz, s.Ax, s.g->s);
        s.Tp.alloc(((size_t)m + 1) * sizeof(int));
        s.Ti.alloc(nz * sizeof(int));
        s.Tx.alloc(nz * sizeof(T));
        OPCHK(rcppml_hip_transpose_csc(c, dt, m, s.n_loc, s.Ap.template as<int>(), s.Ai.template as<int>(), s.Ax.p, s.Tp.template as<int>(), s.Ti.template as<int>(), s.Tx.p));
        if (w_block) {           // padded so that every device owns a whole block; the pad rows stay zero
            DevBuf w0;
            upload_cast<T>(c, P.W, (size_t)k * m, w0, s.g->s);
            s.W.alloc(w_elems * sizeof(T));
            HIPCHK(hipMemsetAsync(s.W.p, 0, w_elems * sizeof(T), c->stream));
            HIPCHK(hipMemcpyAsync(s.W.p, w0.p, (size_t)k * m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        } else
            upload_cast<T>(c, P.W, (size_t)k * m, s.W, s.g->s);
        if (s.n_loc > 0) upload_cast<T>(c, P.H + (size_t)k * s.c0, (size_t)k * s.n_loc, s.H, s.g->s);
        else { std::vector<double> hz(k, 0.0); upload_cast<T>(c, hz.data(), (size_t)k, s.H, s.g->s); }
        s.d.alloc((size_t)k * sizeof(T));
        {
            std::vector<T> ones(k, T(1));
            HIPCHK(hipMemcpyAsync(s.d.p, ones.data(), k * sizeof(T), hipMemcpyHostToDevice, s.g->s));
            HIPCHK(hipStreamSynchronize(s.g->s));
        }
        s.Bh.alloc((size_t)k * std::max(s.n_loc, 1) * sizeof(T));
        s.xbuf.alloc(((size_t)k * k + (size_t)k * m + (size_t)k) * sizeof(T));           // [G_p | B_p | row sums of H]: ONE all-reduce
        s.G.alloc((size_t)k * k * sizeof(T)); s.Gs.alloc((size_t)k * k * sizeof(T)); s.Gwt.alloc((size_t)k * k * sizeof(T));
        s.sums.alloc((size_t)k * sizeof(T));
        s.tr.alloc(sizeof(double)); s.loss.alloc(4 * sizeof(double));
        s.swH.alloc((size_t)std::max(s.n_loc, 1) * sizeof(int)); s.ordH.alloc((size_t)std::max(s.n_loc, 1) * sizeof(int));
        s.swW.alloc((size_t)m * sizeof(int)); s.ordW.alloc((size_t)m * sizeof(int));
        if (s.nnz_loc >= (1 << 20)) {
            plan_or_none(rcppml_hip_rhs_plan_create(c, dt, s.Ap.template as<int>(), s.Ai.template as<int>(), s.Ax.p, s.n_loc, m, k, 0, 0, &s.planA), s.planA);
            plan_or_none(rcppml_hip_rhs_plan_create(c, dt, s.Tp.template as<int>(), s.Ti.template as<int>(), s.Tx.p, m, s.n_loc, k, 0, 0, &s.planT), s.planT);
        }
    }
    Exchange X;
    X.init(devices, streams, shared, force);

    // ---- ||A||^2 over all shards
    double trAtA = 0;
    for (int r = 0; r < nd; ++r) {
        Shard<T>& s = *S[r];
        OPCHK(rcppml_hip_sumsq(s.g->c, dt, s.Ax.p, s.nnz_loc, s.tr.template as<double>()));
        double v = 0;
        HIPCHK(hipMemcpyAsync(&v, s.tr.p, sizeof(double), hipMemcpyDeviceToHost, s.g->s));
        HIPCHK(hipStreamSynchronize(s.g->s));
        trAtA += v;
    }
    for (int r = 0; r < nd; ++r) {
        HIPCHK(hipSetDevice(S[r]->dev));
        HIPCHK(hipMemcpyAsync(S[r]->tr.p, &trAtA, sizeof(double), hipMemcpyHostToDevice, S[r]->g->s));
        HIPCHK(hipStreamSynchronize(S[r]->g->s));
    }
    const bool use_order = P.solver_mode == 0 && P.cd_tol > 0;
    std::vector<void*> bufs(nd);
    auto T_ptr = [](DevBuf& b, size_t off) { return static_cast<void*>(static_cast<T*>(b.p) + off); };

    double prev_loss = std::is_same<T, float>::value ? (double)std::numeric_limits<float>::max() : std::numeric_limits<double>::max();
    int patience_counter = 0, iterations = 0;
    bool converged = false;
    double final_tol = 0, train_loss = 0, last_loss = 0;
    double* hloss = nullptr;
    HIPCHK(hipSetDevice(S[0]->dev));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&hloss), 4 * sizeof(double)));
    struct HostFree { double* p; ~HostFree() { (void)hipHostFree(p); } } hf{hloss};

    // RCPPML_GPU_VERBOSE >= 2 (the ABI's verbose >= 2): per-shard HIP-event times of the five phases of an iteration and the
    // payloads of the collectives, to stderr after the fit -- the full-size input of bench.py's collectives_model.  (With
    // RCPPML_GPU_DEVICES_SHARE the shards' streams share one device: their phases overlap and each is slower than on its own GPU.)
    constexpr int NPH = 6;                                    // H solve | partials [G_p, B_p, row sums] | all-reduce (incl. waiting for the slowest shard) | scaling + W solve (+ all-gather) | W scaling + loss | (whole iteration)
    const bool timing = P.verbose >= 2;
    std::vector<std::vector<hipEvent_t>> tev(nd);
    std::vector<std::vector<double>> tms(nd, std::vector<double>(NPH, 0.0));
    struct EvFree { std::vector<std::vector<hipEvent_t>>& v; ~EvFree() { for (auto& a : v) for (auto e : a) if (e) (void)hipEventDestroy(e); } } evfree{tev};
    if (timing)
        for (int r = 0; r < nd; ++r) {
            HIPCHK(hipSetDevice(S[r]->dev));
            tev[r].assign(NPH + 2, nullptr);
            for (auto& e : tev[r]) HIPCHK(hipEventCreate(&e));
        }
    auto mark = [&](int r, int slot) { if (timing) HIPCHK(hipEventRecord(tev[r][slot], S[r]->g->s)); };

    for (int iter = 0; iter < P.max_iter; ++iter) {
        const int warm = iter > 0 ? 1 : 0;
        // ================= H half-update on every shard (fit_cpu.hpp:486-645): no communication
        for (int r = 0; r < nd; ++r) {
            Shard<T>& s = *S[r];
            rcppml_hip_ctx* c = s.g->c;
            HIPCHK(hipSetDevice(s.dev));
            mark(r, 0);
            if (s.n_loc == 0) { mark(r, 1); mark(r, 2); }
            if (s.n_loc == 0) { HIPCHK(hipSetDevice(s.dev)); HIPCHK(hipMemsetAsync(s.xbuf.p, 0, s.xbuf.bytes, s.g->s)); continue; }
            OPCHK(rcppml_hip_gram(c, dt, s.W.p, k, m, eps, P.L2_H, s.G.p));
            if (s.planA) OPCHK(rcppml_hip_rhs_planned(c, s.planA, s.W.p, s.Bh.p));
            else OPCHK(rcppml_hip_rhs(c, dt, s.Ap.template as<int>(), s.Ai.template as<int>(), s.Ax.p, s.n_loc, s.W.p, k, s.Bh.p));
            if (P.solver_mode == 0) {
                const bool ord = use_order && iter > 0 && s.n_loc >= kOrderMinColumns;
                if (ord) OPCHK(rcppml_hip_order_columns(c, s.swH.template as<int>(), s.n_loc, s.ordH.template as<int>()));
                OPCHK(rcppml_hip_solve_cd(c, dt, s.G.p, s.Bh.p, s.H.p, k, s.n_loc, P.L1_H > 0 ? P.L1_H : 0.0, warm, 0, 0.0, 0.0,
                                          P.nonneg_H, P.cd_maxit, P.cd_tol, 0.0, P.ub_H, RCPPML_CD_AUTO,
                                          use_order ? s.swH.template as<int>() : nullptr, ord ? s.ordH.template as<int>() : nullptr));
            } else {
                OPCHK(rcppml_hip_solve_chol(c, dt, s.G.p, s.Bh.p, s.H.p, k, s.n_loc, P.L1_H > 0 ? P.L1_H : 0.0, P.nonneg_H, P.ub_H));
            }
            mark(r, 1);
            // ================= W half-update (fit_cpu.hpp:711-893).  H H^T, H A^T and the row norms of H are all sums over ALL
            // columns: every shard forms its partials from the UNSCALED H into one buffer [G_p | B_p | row sums] ...
            OPCHK(rcppml_hip_row_norms(c, dt, s.H.p, k, s.n_loc, P.norm_type, T_ptr(s.xbuf, (size_t)k * k + (size_t)k * m)));   // L1: sum |h|; L2: sum h^2
            OPCHK(rcppml_hip_gram(c, dt, s.H.p, k, s.n_loc, 0.0, 0.0, s.xbuf.p));               // eps after the sum
            if (s.planT) OPCHK(rcppml_hip_rhs_planned(c, s.planT, s.H.p, T_ptr(s.xbuf, (size_t)k * k)));
            else OPCHK(rcppml_hip_rhs(c, dt, s.Tp.template as<int>(), s.Ti.template as<int>(), s.Tx.p, m, s.H.p, k, T_ptr(s.xbuf, (size_t)k * k)));
            mark(r, 2);
        }
        // ... ONE all-reduce per iteration ...
        for (int r = 0; r < nd; ++r) bufs[r] = S[r]->xbuf.p;
        X.template all_reduce<T>(bufs, (size_t)k * k + (size_t)k * m + (size_t)k);
        for (int r = 0; r < nd; ++r) {
            Shard<T>& s = *S[r];
            rcppml_hip_ctx* c = s.g->c;
            HIPCHK(hipSetDevice(s.dev));
            hipStream_t st = s.g->s;
            mark(r, 3);
            // ... and the scaling D = diag(d) (variant_helpers.hpp:286-305) is applied after the sum: H_loc <- D^-1 H_loc,
            // B = D^-1 B_raw, G = D^-1 G_raw D^-1 -- the reference's "normalise, then multiply" up to rounding
            void* gsums = T_ptr(s.xbuf, (size_t)k * k + (size_t)k * m);
            OPCHK(rcppml_hip_apply_scaling(c, dt, s.H.p, k, s.n_loc, P.norm_type, gsums, s.d.p));      // also d from the global sums
            OPCHK(rcppml_hip_apply_scaling(c, dt, T_ptr(s.xbuf, (size_t)k * k), k, m, P.norm_type, gsums, s.sums.p));
            hipLaunchKernelGGL(mg_scale_gram_kernel<T>, dim3((k * k + 255) / 256), dim3(256), 0, st, (T*)s.xbuf.p, k, (const T*)s.d.p);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(mg_diag_add_kernel<T>, dim3((k + 63) / 64), dim3(64), 0, st, (T*)s.xbuf.p, k, (T)eps);   // gram.hpp:50-52
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(s.Gs.p, s.xbuf.p, (size_t)k * k * sizeof(T), hipMemcpyDeviceToDevice, st));           // G_saved (:719-722)
            HIPCHK(hipMemcpyAsync(s.G.p, s.xbuf.p, (size_t)k * k * sizeof(T), hipMemcpyDeviceToDevice, st));
            if (P.L2_W > 0) {
                hipLaunchKernelGGL(mg_diag_add_kernel<T>, dim3((k + 63) / 64), dim3(64), 0, st, (T*)s.G.p, k, (T)P.L2_W);   // :738
                HIPCHK(hipGetLastError());
            }
            const int lo = w_block ? std::min(m, r * rows_per) : 0, hi = w_block ? std::min(m, lo + rows_per) : m;
            const int mb = hi - lo;                                      // the columns of W_T this device solves
            if (mb > 0) {
                void* Bb = T_ptr(s.xbuf, (size_t)k * k + (size_t)k * lo);
                void* Wb = (T*)s.W.p + (size_t)k * lo;
                if (P.solver_mode == 0) {
                    const bool ord = use_order && iter > 0 && mb >= kOrderMinColumns;
                    int* sw = use_order ? s.swW.template as<int>() + lo : nullptr;
                    int* od = ord ? s.ordW.template as<int>() + lo : nullptr;
                    if (ord) OPCHK(rcppml_hip_order_columns(c, sw, mb, od));
                    OPCHK(rcppml_hip_solve_cd(c, dt, s.G.p, Bb, Wb, k, mb, P.L1_W > 0 ? P.L1_W : 0.0, warm, 0, 0.0, 0.0, P.nonneg_W,
                                              P.cd_maxit, P.cd_tol, 0.0, P.ub_W, RCPPML_CD_AUTO, sw, od));
                } else
                    OPCHK(rcppml_hip_solve_chol(c, dt, s.G.p, Bb, Wb, k, mb, P.L1_W > 0 ? P.L1_W : 0.0, P.nonneg_W, P.ub_W));
            }
        }
        if (w_block) {
            for (int r = 0; r < nd; ++r) bufs[r] = S[r]->W.p;
            X.template all_gather<T>(bufs, (size_t)rows_per * k);
        }
        for (int r = 0; r < nd; ++r) {
            Shard<T>& s = *S[r];
            rcppml_hip_ctx* c = s.g->c;
            HIPCHK(hipSetDevice(s.dev));
            void* Bw = T_ptr(s.xbuf, (size_t)k * k);
            mark(r, 4);
            OPCHK(rcppml_hip_row_norms(c, dt, s.W.p, k, m, P.norm_type, s.sums.p));
            OPCHK(rcppml_hip_apply_scaling(c, dt, s.W.p, k, m, P.norm_type, s.sums.p, s.d.p));
            // ---- loss (fit_cpu.hpp:1729-1753) from replicated quantities: identical on every device
            OPCHK(rcppml_hip_gram(c, dt, s.W.p, k, m, eps, 0.0, s.Gwt.p));
            OPCHK(rcppml_hip_loss_mse(c, dt, s.tr.template as<double>(), s.d.p, s.W.p, Bw, k, m, s.Gwt.p, s.Gs.p, s.loss.template as<double>()));
            mark(r, 5);
        }
        HIPCHK(hipSetDevice(S[0]->dev));
        HIPCHK(hipMemcpyAsync(hloss, S[0]->loss.p, 4 * sizeof(double), hipMemcpyDeviceToHost, S[0]->g->s));
        for (int r = 0; r < nd; ++r) { HIPCHK(hipSetDevice(S[r]->dev)); HIPCHK(hipStreamSynchronize(S[r]->g->s)); }
        if (timing)
            for (int r = 0; r < nd; ++r) {
                // slots: 0 start, 1 after the H solve, 2 after the partials, 3 after the all-reduce, 4 after the W solve (+ all-gather), 5 end
                const int a[NPH] = {0, 1, 2, 3, 4, 0}, b[NPH] = {1, 2, 3, 4, 5, 5};
                for (int ph = 0; ph < NPH; ++ph) {
                    float ms = 0;
                    if (hipEventElapsedTime(&ms, tev[r][a[ph]], tev[r][b[ph]]) == hipSuccess) tms[r][ph] += ms; else (void)hipGetLastError();
                }
            }
        double loss_val = hloss[0];
        if (std::is_same<T, float>::value) loss_val = static_cast<double>(static_cast<float>(loss_val));
        last_loss = loss_val;
        if (P.loss_history) P.loss_history[iter] = loss_val;
        bool loss_converged = false;
        double rel = 0;
        if (iter > 0) {                                                                 // fit_cpu.hpp:1769-1775
            rel = std::fabs(prev_loss - loss_val) / (std::fabs(prev_loss) + 1e-15);
            final_tol = rel;
            if (rel < P.tol) loss_converged = true;
        }
        prev_loss = loss_val;
        if (P.verbose) fprintf(stderr, "[rcppml_gpu x%d] iter %d loss %.9g rel %.3g\n", nd, iter + 1, loss_val, rel);
        if (iter > 0) {                                                                 // :1797-1809
            if (loss_converged) {
                if (++patience_counter >= P.patience) { converged = true; train_loss = prev_loss; iterations = iter + 1; break; }
            } else patience_counter = 0;
        }
        iterations = iter + 1;
    }
    if (!converged) train_loss = last_loss;
    if (timing && iterations > 0) {
        const size_t sv = sizeof(T);
        fprintf(stderr, "[rcppml_gpu x%d] %s per iteration: all-reduce [G_p | B_p | row sums] %zu bytes%s; W solve %s; %d iteration(s)\n", nd,
                shared ? "shared-device stand-in" : "RCCL", ((size_t)k * k + (size_t)k * m + (size_t)k) * sv,
                w_block ? (std::string(", all-gather of W_T ") + std::to_string((size_t)rows_per * nd * k * sv) + " bytes").c_str() : "",
                w_block ? "block" : "replicated", iterations);
        for (int r = 0; r < nd; ++r)
            fprintf(stderr, "[rcppml_gpu x%d] shard %d: columns [%d, %d) nnz %lld  ms/iteration: H solve %.3f | partials %.3f | all-reduce %.3f | scaling + W solve %.3f | W scaling + loss %.3f | whole %.3f\n",
                    nd, r, S[r]->c0, S[r]->c0 + S[r]->n_loc, (long long)S[r]->nnz_loc, tms[r][0] / iterations, tms[r][1] / iterations,
                    tms[r][2] / iterations, tms[r][3] / iterations, tms[r][4] / iterations, tms[r][5] / iterations);
    }
    // ---- the replicas of W_T must be bitwise equal (replicated solve: identical inputs, deterministic kernels; block solve: the
    // all-gather): checked on every fit -- a divergence would otherwise be silent, H of the other shards being solved against
    // another W_T than the one that is returned
    {
        std::vector<unsigned long long> sums(nd, 0);
        for (int r = 0; r < nd; ++r) {
            HIPCHK(hipSetDevice(S[r]->dev));
            DevBuf acc(sizeof(unsigned long long));
            HIPCHK(hipMemsetAsync(acc.p, 0, sizeof(unsigned long long), S[r]->g->s));
            hipLaunchKernelGGL(mg_checksum_kernel<T>, dim3(256), dim3(256), 0, S[r]->g->s, (const T*)S[r]->W.p, (size_t)k * m, (unsigned long long*)acc.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(&sums[r], acc.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, S[r]->g->s));
            HIPCHK(hipStreamSynchronize(S[r]->g->s));
        }
        for (int r = 1; r < nd; ++r)
            if (sums[r] != sums[0]) throw std::runtime_error("multi-device fit: the W_T replicas of devices 0 and " + std::to_string(r) + " differ");
        if (timing) fprintf(stderr, "[rcppml_gpu x%d] W_T replicas bitwise equal (checksum %016llx)\n", nd, sums[0]);
    }

    // ---- download (W, d from device 0; every shard's H) and sort by descending d (core/result.hpp:169-188)
    download_cast<T>(S[0]->g->c, S[0]->W, (size_t)k * m, P.W, S[0]->g->s);
    download_cast<T>(S[0]->g->c, S[0]->d, (size_t)k, P.d, S[0]->g->s);
    for (int r = 0; r < nd; ++r)
        if (S[r]->n_loc > 0) download_cast<T>(S[r]->g->c, S[r]->H, (size_t)k * S[r]->n_loc, P.H + (size_t)k * S[r]->c0, S[r]->g->s);
    if (P.sort_model) {
        std::vector<int> idx(k);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return P.d[a] > P.d[b]; });
        std::vector<double> tmp(k);
        for (int j = 0; j < m; ++j) { double* w = P.W + (size_t)j * k; for (int i = 0; i < k; ++i) tmp[i] = w[idx[i]]; std::copy(tmp.begin(), tmp.end(), w); }
        for (int j = 0; j < n; ++j) { double* h = P.H + (size_t)j * k; for (int i = 0; i < k; ++i) tmp[i] = h[idx[i]]; std::copy(tmp.begin(), tmp.end(), h); }
        for (int i = 0; i < k; ++i) tmp[i] = P.d[idx[i]];
        std::copy(tmp.begin(), tmp.end(), P.d);
    }
    P.out_iter = iterations; P.out_converged = converged ? 1 : 0; P.out_loss = train_loss; P.out_tol = final_tol;
}

}  // namespace

bool rcppml_fit_multi(FitParams& P, int precision, int ndev) {
    // what the sharded loop covers (see the header of this file)
    if (P.dense || P.csc_on_device || P.mask_p || P.loss_type != 0 || P.robust_delta > 0 || P.projective || P.symmetric) return false;
    if (P.L21_H > 0 || P.L21_W > 0 || P.angular_H > 0 || P.angular_W > 0) return false;
    if ((P.target_H && P.target_lambda_H != 0) || (P.target_W && P.target_lambda_W != 0)) return false;
    if ((P.gH_p && P.gH_nnz > 0 && P.gH_lambda > 0) || (P.gW_p && P.gW_nnz > 0 && P.gW_lambda > 0)) return false;
    const char* fo = getenv("RCPPML_GPU_DEVICES_FORCE");
    const bool force = fo && atoi(fo) != 0;
    if (ndev < (force ? 1 : 2) || P.n < ndev) return false;
    const char* sh = getenv("RCPPML_GPU_DEVICES_SHARE");
    const bool shared = sh && atoi(sh) != 0;
    int have = 0;
    HIPCHK(hipGetDeviceCount(&have));
    std::vector<int> devices(ndev);
    if (shared) {
        for (int r = 0; r < ndev; ++r) devices[r] = env_device();
    } else {
        if (have < ndev) throw std::runtime_error("RCPPML_GPU_DEVICES=" + std::to_string(ndev) + " but only " + std::to_string(have) +
                                                  " device(s) visible (RCPPML_GPU_DEVICES_SHARE=1 maps the shards onto one device for testing)");
        for (int r = 0; r < ndev; ++r) devices[r] = r;
    }
    if (precision == RCPPML_F64) fit_multi<double>(P, devices, shared, force);
    else fit_multi<float>(P, devices, shared, force);
    return true;
}
