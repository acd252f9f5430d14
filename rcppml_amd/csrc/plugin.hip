// ============================================================================
// plugin.hip -- the plugin boundary of RcppML_gpu.so (include/rcppml_gpu.h, layer 1):
// host-pointer entry points bound by the unmodified R package, and the ALS loop that drives the
// device-level ops.  Replaces reference src/gpu_bridge_nmf.cu:34-210,460-624 and
// src/gpu_bridge_cluster.cu:24-46 (boundary) and inst/include/FactorNet/nmf/fit_gpu.cuh (loop);
// the loop follows the CPU semantics of inst/include/FactorNet/nmf/fit_cpu.hpp:444-1855 because
// CPU nmf() is the parity target.
// ============================================================================
#include <chrono>
#include "plugin_common.hip.h"

extern "C" const char* rcppml_gpu_last_error(void) { return rcppml_err().c_str(); }

namespace {


// PROJ_ADV (nmf/variant_helpers.hpp:112-146) on a k x k Gram held in `G` (host, double, symmetric):
//   G -= |lambda| * (trace G / trace TG) * TG;  eigenvalues below 1e-8 are raised to 1e-8 (G = V max(L, eps) V^T, only if
//   one was).  The eigen-decomposition is a cyclic Jacobi iteration: the clipped matrix is a function of G alone, so any
//   convergent symmetric eigensolver reproduces the reference's SelfAdjointEigenSolver result up to rounding.
void proj_adv_host(std::vector<double>& G, const std::vector<double>& TG, int k, double abs_lambda) {
    double trG = 0, trT = 0;
    for (int i = 0; i < k; ++i) { trG += G[(size_t)i * k + i]; trT += TG[(size_t)i * k + i]; }
    const double scale = trT > 1e-10 ? trG / trT : 0.0;
    for (size_t e = 0; e < (size_t)k * k; ++e) G[e] -= abs_lambda * scale * TG[e];
    std::vector<double> A = G, V((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i) V[(size_t)i * k + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int p = 0; p < k; ++p) for (int q = p + 1; q < k; ++q) off += A[(size_t)q * k + p] * A[(size_t)q * k + p];
        if (off < 1e-300) break;
        for (int p = 0; p < k; ++p)
            for (int q = p + 1; q < k; ++q) {
                const double apq = A[(size_t)q * k + p];
                if (std::fabs(apq) < 1e-300) continue;
                const double app = A[(size_t)p * k + p], aqq = A[(size_t)q * k + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int r = 0; r < k; ++r) {                       // columns p, q
                    const double arp = A[(size_t)p * k + r], arq = A[(size_t)q * k + r];
                    A[(size_t)p * k + r] = cs * arp - sn * arq;
                    A[(size_t)q * k + r] = sn * arp + cs * arq;
                }
                for (int r = 0; r < k; ++r) {                       // rows p, q
                    const double apr = A[(size_t)r * k + p], aqr = A[(size_t)r * k + q];
                    A[(size_t)r * k + p] = cs * apr - sn * aqr;
                    A[(size_t)r * k + q] = sn * apr + cs * aqr;
                }
                for (int r = 0; r < k; ++r) {
                    const double vrp = V[(size_t)p * k + r], vrq = V[(size_t)q * k + r];
                    V[(size_t)p * k + r] = cs * vrp - sn * vrq;
                    V[(size_t)q * k + r] = sn * vrp + cs * vrq;
                }
            }
    }
    bool clipped = false;
    std::vector<double> ev(k);
    for (int i = 0; i < k; ++i) { ev[i] = A[(size_t)i * k + i]; if (ev[i] < 1e-8) { ev[i] = 1e-8; clipped = true; } }
    if (!clipped) return;
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
            double s = 0;
            for (int e = 0; e < k; ++e) s += V[(size_t)e * k + i] * ev[e] * V[(size_t)e * k + j];      // V stored column e = eigenvector e
            G[(size_t)j * k + i] = s;
        }
}

// out(:, j) = in(idx, j): the factors (rows of the k x ncols arrays) in the order of descending d
template <class T>
__global__ void permute_rows_kernel(const T* __restrict__ in, int k, int64_t ncols, const int* __restrict__ idx, T* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)k * ncols) return;
    const int64_t j = e / k;
    const int i = (int)(e - j * k);
    out[e] = in[j * k + idx[i]];
}

// index arrays of the CSC that stores every entry of a dense m x n matrix (column j = rows 0 .. m-1): the dense array itself is
// its value array.  Used for dense input under a distribution loss, whose per-element passes are written over a CSC.
__global__ void full_csc_index_kernel(int m, int64_t n, int* __restrict__ col_ptr, int* __restrict__ row_idx) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= n) col_ptr[e] = (int)(e * m);
    if (e < (int64_t)m * n) row_idx[e] = (int)(e % m);
}

// ----------------------------------------------------------------------------
// The ALS loop (MSE; fused-path semantics of fit_cpu.hpp, or the explicit-mask path).
// ----------------------------------------------------------------------------
// RCPPML_GPU_NO_SMALL set to anything but "" / "0": small plain fits stay on the multi-launch loop
inline bool env_no_small() {
    const char* e = getenv("RCPPML_GPU_NO_SMALL");
    return e && *e && strcmp(e, "0") != 0;
}

template <class T>
void fit(FitParams& P) {
    constexpr int dt = DT<T>::id;
    const int m = P.m, n = P.n, k = P.k;
    // RCPPML_GPU_VERBOSE >= 2 (the ABI's verbose >= 2): wall time of the one-time phases around the iterations, to stderr
    const auto tstart = std::chrono::steady_clock::now();
    auto tlast = tstart;
    auto phase = [&](const char* what, hipStream_t st) {
        if (P.verbose < 2) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rcppml_gpu setup] %-28s %8.3f ms (at %8.3f)\n", what, std::chrono::duration<double, std::milli>(now - tlast).count(),
                std::chrono::duration<double, std::milli>(now - tstart).count());
        tlast = now;
    };
    CtxGuard g(P.device >= 0 ? P.device : env_device());
    rcppml_hip_ctx* c = g.c;
    hipStream_t s = g.s;
    if (!P.dense && !P.csc_on_device) {
        // everything this fit allocates, generously: the CSC twice (+ the double staging copy and the sort's temporaries),
        // the factors, right-hand sides and their staging copies, the two row-tiled plans
        // (the window plans take ~12 bytes per nonzero and side + the partial slabs of their row partitions: inside the 96 nnz)
        const size_t est = (size_t)96 * (size_t)std::max<int64_t>(P.nnz, 1) + (size_t)96 * (size_t)k * ((size_t)m + n) + ((size_t)256 << 20);
        g.reserve(est);
    }
    phase("context + stream + arena", s);

    // ---- upload A, build and upload A^T (one-time setup, fit_cpu.hpp:237-254)
    DevBuf dAp, dAi, dAx, dTp, dTi, dTx;
    struct PlanGuard { rcppml_rhs_plan* p = nullptr; ~PlanGuard() { rcppml_hip_rhs_plan_destroy(p); } } planA, planT;   // row-tiled rhs plans (see below)
    const bool dense = P.dense != nullptr;
    // the IRLS half-updates (every loss but plain MSE) gather the factor rows per nonzero themselves and never call the rhs; only
    // the projective H update does under such a loss.  No plans then: they were 6 ms of an NB fit's set-up (C5, fp64)
    const bool plans_useful = !(P.loss_type != 0 || P.robust_delta > 0) || P.projective;
    bool csc_transposed = false;
    if (dense) {                    // dense input: A itself (m x n, column-major) in the compute precision; no CSC, no transpose
        upload_cast<T>(c, P.dense, (size_t)m * n, dAx, s);
        if (P.loss_type != 0 || P.robust_delta > 0) {
            // ... except under a distribution loss (nnls_batch_irls_dense, fit_cpu.hpp:607-614, :855-863; explicit_loss_dense;
            // the dense branches of the dispersion updates): every entry is weighted, zeros included -- which is what the
            // per-nonzero IRLS / dispersion / likelihood passes compute on the CSC that stores every entry.  Its value array is the
            // dense array; the transpose's comes from the stable device sort like any other.  (The reference's dense column solve
            // forms G_w = sum_i w_i f_i f_i^T from nothing where the sparse one adds sum_i (w_i - 1) f_i f_i^T to G_base, and its
            // dispersion sums floor every prediction at 1e-10: the same numbers up to rounding, oracle dense_input.)
            const int64_t tot = (int64_t)m * n;
            dAp.alloc(((size_t)n + 1) * sizeof(int));
            dAi.alloc((size_t)tot * sizeof(int));
            hipLaunchKernelGGL(full_csc_index_kernel, dim3((unsigned)((tot + 256) / 256)), dim3(256), 0, s, m, (int64_t)n, dAp.as<int>(), dAi.as<int>());
            HIPCHK(hipGetLastError());
            dTp.alloc(((size_t)m + 1) * sizeof(int));
            dTi.alloc((size_t)tot * sizeof(int));
            dTx.alloc((size_t)tot * sizeof(T));
            OPCHK(rcppml_hip_transpose_csc(c, dt, m, n, dAp.as<int>(), dAi.as<int>(), dAx.p, dTp.as<int>(), dTi.as<int>(), dTx.p));
        }
    } else if (P.csc_on_device) {          // zero-copy: the CSC already lives in device memory (values double)
        dAp.borrow(P.col_ptr);
        dAi.borrow(P.row_idx);
        if constexpr (std::is_same<T, double>::value) dAx.borrow(P.values);
        else {
            dAx.alloc((size_t)std::max<int64_t>(P.nnz, 1) * sizeof(T));
            OPCHK(rcppml_hip_cast(c, RCPPML_F64, P.values, RCPPML_F32, dAx.p, P.nnz));
        }
    } else if (c->arena) {
        // Host CSC, overlapped setup: the row indices go first and the device starts sorting them (the transpose's expensive
        // part needs nothing else) while the VALUES -- two thirds of the bytes -- are still crossing PCIe.  Pageable copies
        // block the host, not the device, so the sort runs under the value upload; everything is ordered on the one stream.
        upload_ints(P.col_ptr, (size_t)n + 1, dAp, s);
        upload_ints(P.row_idx, (size_t)P.nnz, dAi, s);
        phase("  upload col_ptr + row_idx", s);
        dTp.alloc(((size_t)m + 1) * sizeof(int));
        dTi.alloc((size_t)std::max<int64_t>(P.nnz, 1) * sizeof(int));
        dTx.alloc((size_t)std::max<int64_t>(P.nnz, 1) * sizeof(T));
        DevBuf dpos((size_t)std::max<int64_t>(P.nnz, 1) * sizeof(int));
        OPCHK(rcppml_hip_transpose_csc_sort(c, m, n, P.nnz, dAp.as<int>(), dAi.as<int>(), dTp.as<int>(), dpos.as<int>()));      // asynchronous (arena)
        OPCHK(rcppml_hip_transpose_csc_gather(c, dt, n, P.nnz, dAp.as<int>(), dpos.as<int>(), nullptr, dTi.as<int>(), nullptr));   // column indices of A^T
        if (P.verbose >= 2) phase("  transpose: sort + column indices (values in flight)", s);
        hipStream_t s2 = g.second_stream();
        dAx.alloc((size_t)std::max<int64_t>(P.nnz, 1) * sizeof(T));
        DevBuf stage;
        if constexpr (!std::is_same<T, double>::value) stage.alloc((size_t)std::max<int64_t>(P.nnz, 1) * sizeof(double));
        // the values cross PCIe on a helper thread (a pageable copy blocks its caller); this thread meanwhile builds the INDEX
        // halves of the two row-tiled plans (schedule, offsets, overflow lists: everything but the values) behind the sort
        std::exception_ptr up_err;
        std::thread up([&] {
            try {
                HIPCHK(hipSetDevice(c->device));
                HIPCHK(hipMemcpyAsync(std::is_same<T, double>::value ? dAx.p : stage.p, P.values, (size_t)P.nnz * sizeof(double),
                                      hipMemcpyHostToDevice, s2));
                HIPCHK(hipStreamSynchronize(s2));
            } catch (...) { up_err = std::current_exception(); }
        });
        try {
            if (P.nnz >= (1 << 20) && !P.mask_p && plans_useful) {
                plan_or_none(rcppml_hip_rhs_plan_create_indices(c, dt, dAp.as<int>(), dAi.as<int>(), n, m, k, 0, 0, &planA.p), planA.p);
                plan_or_none(rcppml_hip_rhs_plan_create_indices(c, dt, dTp.as<int>(), dTi.as<int>(), m, n, k, 0, 0, &planT.p), planT.p);
            }
        } catch (...) { up.join(); throw; }
        if (P.verbose >= 2) phase("  plan index halves", s);
        up.join();
        if (up_err) std::rethrow_exception(up_err);
        if (P.verbose >= 2) phase("  values across PCIe (joined)", s);
        if constexpr (!std::is_same<T, double>::value) OPCHK(rcppml_hip_cast(c, RCPPML_F64, stage.p, RCPPML_F32, dAx.p, P.nnz));
        OPCHK(rcppml_hip_transpose_csc_gather(c, dt, n, P.nnz, dAp.as<int>(), dpos.as<int>(), dAx.p, nullptr, dTx.p));             // its values
        if (planA.p) OPCHK(rcppml_hip_rhs_plan_set_values(c, planA.p, dAx.p));
        if (planT.p) OPCHK(rcppml_hip_rhs_plan_set_values(c, planT.p, dTx.p));
        csc_transposed = true;
    } else {
        upload_ints(P.col_ptr, (size_t)n + 1, dAp, s);
        upload_ints(P.row_idx, (size_t)P.nnz, dAi, s);
        upload_cast<T>(c, P.values, (size_t)P.nnz, dAx, s);
    }
    phase("upload CSC", s);
    // A^T on the device (stable sort by row index): rcppml_hip_transpose_csc
    if (!dense && !csc_transposed) {
    dTp.alloc(((size_t)m + 1) * sizeof(int));
    dTi.alloc((size_t)P.nnz * sizeof(int));
    dTx.alloc((size_t)P.nnz * sizeof(T));
    OPCHK(rcppml_hip_transpose_csc(c, dt, m, n, dAp.as<int>(), dAi.as<int>(), dAx.p, dTp.as<int>(), dTi.as<int>(), dTx.p));
    }
    phase("transpose", s);
    const bool has_mask = P.mask_p != nullptr;
    DevBuf dMp, dMi, dMTp, dMTi;
    if (has_mask) {
        const int mnnz = P.mask_p[n];
        upload_ints(P.mask_p, (size_t)n + 1, dMp, s);
        upload_ints(P.mask_i, (size_t)std::max(mnnz, 1), dMi, s);
        dMTp.alloc(((size_t)m + 1) * sizeof(int));
        dMTi.alloc((size_t)std::max(mnnz, 1) * sizeof(int));
        OPCHK(rcppml_hip_transpose_csc(c, dt, m, n, dMp.as<int>(), dMi.as<int>(), nullptr, dMTp.as<int>(), dMTi.as<int>(), nullptr));
    }

    // ---- graph Laplacians (features/graph_reg.hpp), uploaded once
    DevBuf dGHp, dGHi, dGHx, dGWp, dGWi, dGWx;
    const bool graph_H = P.gH_p && P.gH_nnz > 0 && P.gH_lambda > 0, graph_W = P.gW_p && P.gW_nnz > 0 && P.gW_lambda > 0;
    if (graph_H) {
        upload_ints(P.gH_p, (size_t)n + 1, dGHp, s); upload_ints(P.gH_i, (size_t)P.gH_nnz, dGHi, s);
        upload_cast<T>(c, P.gH_x, (size_t)P.gH_nnz, dGHx, s);
    }
    if (graph_W) {
        upload_ints(P.gW_p, (size_t)m + 1, dGWp, s); upload_ints(P.gW_i, (size_t)P.gW_nnz, dGWi, s);
        upload_cast<T>(c, P.gW_x, (size_t)P.gW_nnz, dGWx, s);
    }
    // ---- factors (their PCIe copies run on the second stream: under the transpose's gather when the setup is overlapped)
    DevBuf dW, dH, dd;
    AsyncUpload<T> factors;               // joined just before the first iteration: the copies run under trAtA and the plans
    factors.add(P.W, (size_t)k * m, dW);
    factors.add(P.H, (size_t)k * n, dH);
    factors.start(c->device, g.second_stream());
    dd.alloc((size_t)k * sizeof(T));
    {
        std::vector<T> ones(k, T(1));                       // fit_cpu.hpp:198 d = 1
        HIPCHK(hipMemcpyAsync(dd.p, ones.data(), k * sizeof(T), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    DevBuf dBh((size_t)k * n * sizeof(T)), dBw((size_t)k * m * sizeof(T));
    DevBuf dWd(P.projective ? (size_t)k * m * sizeof(T) : 16);
    DevBuf dG((size_t)k * k * sizeof(T)), dGs((size_t)k * k * sizeof(T)), dGwt((size_t)k * k * sizeof(T));
    DevBuf dsums((size_t)k * sizeof(T));
    DevBuf dtr(sizeof(double)), dloss(4 * sizeof(double));
    double* hloss = nullptr;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&hloss), 4 * sizeof(double)));
    struct HostFree { double* p; ~HostFree() { (void)hipHostFree(p); } } hf{hloss};

    OPCHK(rcppml_hip_sumsq(c, dt, dAx.p, dense ? (int64_t)m * n : P.nnz, dtr.as<double>()));       // trAtA, primitives.hpp:100-115
    // CD work order: columns sorted by the sweeps of the previous iteration (results are order-independent)
    DevBuf dswH((size_t)n * sizeof(int)), dswW((size_t)m * sizeof(int)), dordH((size_t)n * sizeof(int)), dordW((size_t)m * sizeof(int));
    const bool use_order = P.solver_mode == 0 && !has_mask && P.loss_type == 0 && P.cd_tol > 0 && !exp_env("RCPPML_GPU_NO_ORDER");
    // the work order of a side's NEXT solve is ranked inside the launches of its scaling pass (rcppml_hip_scale_order); these say
    // whether dordH / dordW already hold the ranking of the sweep counts in dswH / dswW
    bool ordH_fresh = false, ordW_fresh = false;

    const bool is_pow = P.loss_type >= 6;                               // phi_vec = 1 for dispersion none (fit_cpu.hpp:336-347)
    const bool is_gp = P.loss_type == 4 || is_pow || P.loss_type == 0;  // theta_vec = Zero(m) (:297-304); "no theta in the solve"
    const bool is_nb = P.loss_type == 5 || (is_gp && (P.loss_type != 0 || P.robust_delta > 0));   // "is_irls": requires_irls()
    // dispersion = "per_col" (DispersionMode::PER_COL, fit_cpu.hpp:300-301, :319-320, :341-342): one value per COLUMN of A.  Every
    // device op below is written per row of the matrix whose CSC(transpose) it is handed; PER_COL hands it CSC(A) itself with the
    // two factors swapped (mu_ij = (h_j * d) . w_i, the same product with d applied to the other factor)
    const bool per_col = P.dispersion_mode == 3;
    const size_t dlen = per_col ? (size_t)n : (size_t)m;
    DevBuf dtheta;
    if (is_nb) {                                                        // fit_cpu.hpp:316-328
        // GP theta: gp_theta_init or 0 (:297-307); phi: gamma_phi_init or 1 (:337-347); robust MSE: unused
        const double t0 = is_pow ? (P.dispersion_mode == 0 ? 1.0 : P.gamma_phi_init)
                        : P.loss_type == 4 ? (P.dispersion_mode == 0 ? 0.0 : P.gp_theta_init)
                        : is_gp ? 0.0 : (P.dispersion_mode == 0 ? P.nb_size_max : P.nb_size_init);
        std::vector<T> th(dlen, static_cast<T>(t0));
        dtheta.alloc(dlen * sizeof(T));
        HIPCHK(hipMemcpyAsync(dtheta.p, th.data(), dlen * sizeof(T), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    const double eps = 1e-15;
    double prev_loss = std::numeric_limits<double>::max();
    if (std::is_same<T, float>::value) prev_loss = std::numeric_limits<float>::max();
    int patience_counter = 0;
    int iterations = 0;
    bool converged = false;
    double final_tol = 0, train_loss = 0, last_loss = 0;

    // All device work of one ALS iteration, enqueued on the fit's stream (ends with the loss terms in dloss).
    // B = F * A (columns of A) / B = F * A^T (rows of A): CSC gather kernels, or GEMMs for a dense A
    // Large sparse inputs: the LDS row-tiled form (kernels_rhs_tiled.hip.h), planned once per fit for each side; NULL
    // plans (small input, rank or layout the kernel is not compiled for) keep the gather kernel.
    if (!dense && P.nnz >= (1 << 20) && plans_useful) {             // (the overlapped set-up above has usually built them already)
        if (!planA.p) plan_or_none(rcppml_hip_rhs_plan_create(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, n, m, k, 0, 0, &planA.p), planA.p);
        if (!planT.p) plan_or_none(rcppml_hip_rhs_plan_create(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, n, k, 0, 0, &planT.p), planT.p);
    }
    factors.finish(c);
    phase("buffers + trAtA + plans + factors", s);
    // ---- target regularisation (variant_helpers.hpp:107-146): standard (unfused) path, as in the reference (fit_cpu.hpp:430-433)
    const bool tgtH = P.target_H && P.target_lambda_H != 0, tgtW = P.target_W && P.target_lambda_W != 0;
    const bool unfused = dense || tgtH || tgtW;
    DevBuf dTH, dTW, dBt;
    std::vector<double> TG_H, TG_W;                 // PROJ_ADV: target target^T / ncols (nmf/fit.hpp:259-271), host copy
    if (tgtH || tgtW) dBt.alloc((size_t)k * std::max(m, n) * sizeof(T));
    auto target_setup = [&](const double* Th, int ncols, double lambda, DevBuf& dT, std::vector<double>& TG) {
        upload_cast<T>(c, Th, (size_t)k * ncols, dT, s);
        if (lambda < 0) {
            DevBuf g((size_t)k * k * sizeof(T));
            OPCHK(rcppml_hip_gram(c, dt, dT.p, k, ncols, 0.0, 0.0, g.p));
            TG.resize((size_t)k * k);
            download_cast<T>(c, g, (size_t)k * k, TG.data(), s);
            for (auto& v : TG) v /= (double)ncols;
        }
    };
    if (tgtH) target_setup(P.target_H, n, P.target_lambda_H, dTH, TG_H);
    if (tgtW) target_setup(P.target_W, m, P.target_lambda_W, dTW, TG_W);
    // applies the target terms to (G, B) after L1/L2, graph and L21 (apply_features order); returns the B to solve with
    auto apply_target = [&](double lambda, DevBuf& dT, const std::vector<double>& TG, void* Gd, void* Bd, int ncols) -> void* {
        if (lambda > 0) {
            OPCHK(rcppml_hip_add_diag(c, dt, Gd, k, lambda));
            OPCHK(rcppml_hip_axpy(c, dt, Bd, dT.p, lambda, (int64_t)k * ncols, dBt.p));
            return dBt.p;
        }
        std::vector<double> Gh((size_t)k * k);
        DevBuf gview; gview.borrow(Gd);
        download_cast<T>(c, gview, (size_t)k * k, Gh.data(), s);
        proj_adv_host(Gh, TG, k, -lambda);
        DevBuf up;
        upload_cast<T>(c, Gh.data(), (size_t)k * k, up, s);
        HIPCHK(hipMemcpyAsync(Gd, up.p, (size_t)k * k * sizeof(T), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        return Bd;
    };
    auto rhs_fwd = [&](const void* F, void* B) {
        if (dense) OPCHK(rcppml_hip_rhs_dense(c, dt, dAx.p, m, n, 0, F, k, B));
        else if (planA.p) OPCHK(rcppml_hip_rhs_planned(c, planA.p, F, B));
        else OPCHK(rcppml_hip_rhs(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, n, F, k, B));
    };
    auto rhs_bwd = [&](const void* F, void* B) {
        if (dense) OPCHK(rcppml_hip_rhs_dense(c, dt, dAx.p, m, n, 1, F, k, B));
        else if (planT.p) OPCHK(rcppml_hip_rhs_planned(c, planT.p, F, B));
        else OPCHK(rcppml_hip_rhs(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, F, k, B));
    };
    auto enqueue_iteration = [&](int iter) {
        const int warm = iter > 0 ? 1 : 0;
        // dense input runs the reference's STANDARD path (fit_cpu.hpp:540-631, :774-881): nnls_batch starts from zero at
        // iteration 0 and from the residual-corrected previous solution afterwards; the sparse fused path starts
        // iteration 0 from the initial factor without correction
        const int zinit = (unfused && !warm) ? 1 : 0;
        // ================= H half-update (fit_cpu.hpp:486-645)
        if (P.symmetric) {
            // :474-477 SYMMETRIC_SKIP: H is not updated and not scaled; it is set to W_T after the W update
        } else if (P.projective) {                                                             // :462-472
            OPCHK(rcppml_hip_mul_rows(c, dt, dW.p, k, m, dd.p, dWd.p));
            rhs_fwd(dWd.p, dH.p);
        } else if (has_mask) {                                                                 // :560-564: the mask branch comes FIRST -- with a mask the
            // half-updates are the masked MSE solves whatever the loss (requires_irls() is only asked in the else-branch)
            OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, eps, 0.0, dG.p));                 // :562 unmodified G
            OPCHK(rcppml_hip_solve_masked(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, dMp.as<int>(), dMi.as<int>(),
                                          n, dW.p, dG.p, dH.p, k, P.L1_H, P.L2_H, P.nonneg_H, P.cd_maxit, P.cd_tol,
                                          P.solver_mode, warm));
            if (P.ub_H > 0) OPCHK(rcppml_hip_clip_upper(c, dt, dH.p, (int64_t)k * n, P.ub_H));      // :636-637
        } else if (is_nb) {                                                                    // :565-606 (G: eps only)
            OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, eps, 0.0, dG.p));
            OPCHK(rcppml_hip_solve_irls(c, dt, P.loss_type, dAp.as<int>(), dAi.as<int>(), dAx.p, n, dW.p, dG.p, dH.p, k, P.L1_H,
                                        P.L2_H, P.nonneg_H, P.cd_maxit, P.irls_max_iter, P.irls_tol,
                                        (is_gp || per_col) ? nullptr : dtheta.p, (!is_gp && per_col) ? dtheta.p : nullptr,   // :577-583
                                        P.tweedie_power, P.robust_delta));
            if (P.ub_H > 0) OPCHK(rcppml_hip_clip_upper(c, dt, dH.p, (int64_t)k * n, P.ub_H));      // :636-637
        } else {
            // W_T^T W_T + eps I is what the previous iteration's loss formed from this very W_T (dGwt: same kernel, same
            // input, bitwise the same matrix): reused when nothing is added to it -- one Gram less per iteration
            const bool reuse_gwt = iter > 0 && !is_nb && P.L2_H == 0 && !graph_H && P.L21_H == 0 && !tgtH;
            void* const Gh = reuse_gwt ? dGwt.p : dG.p;
            if (!reuse_gwt) OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, eps, P.L2_H, dG.p));              // :491,506
            if (graph_H) OPCHK(rcppml_hip_apply_graph_reg(c, dt, dG.p, dGHp.as<int>(), dGHi.as<int>(), dGHx.p, dH.p, k, n, P.gH_lambda));   // :508-509
            if (P.L21_H > 0) OPCHK(rcppml_hip_apply_l21(c, dt, dG.p, dH.p, k, n, P.L21_H));   // :509-510 (current H)
            rhs_fwd(dW.p, dBh.p);
            void* Bh_use = tgtH ? apply_target(P.target_lambda_H, dTH, TG_H, dG.p, dBh.p, n) : dBh.p;
            if (P.solver_mode == 0) {                                                   // :516-524
                const bool ord = use_order && iter > 0 && n >= kOrderMinColumns;
                if (ord && !ordH_fresh) OPCHK(rcppml_hip_order_columns(c, dswH.as<int>(), n, dordH.as<int>()));
                OPCHK(rcppml_hip_solve_cd(c, dt, Gh, Bh_use, dH.p, k, n, P.L1_H > 0 ? P.L1_H : 0.0, warm, zinit, 0.0, 0.0,
                                          P.nonneg_H, P.cd_maxit, P.cd_tol, 0.0, P.ub_H, RCPPML_CD_AUTO,
                                          use_order ? dswH.as<int>() : nullptr, ord ? dordH.as<int>() : nullptr));
                ordH_fresh = false;
            }
            else                                                                        // :527-534
                OPCHK(rcppml_hip_solve_chol(c, dt, Gh, Bh_use, dH.p, k, n, P.L1_H > 0 ? P.L1_H : 0.0, P.nonneg_H, P.ub_H));
        }
        if (P.angular_H > 0 && !P.projective && !P.symmetric) OPCHK(rcppml_hip_angular_posthoc(c, dt, dH.p, k, n, P.angular_H));   // :638-639 (standard branch only)
        // the branch whose CD solves leave sweep counts behind (dswH / dswW) and run in sweep-sorted work order
        const bool std_cd = use_order && !P.symmetric && !P.projective && !has_mask && !is_nb;
        // the W half-update's standard branch forms gram(H) + eps -> dGs right after this scaling (:715-722): one call does both
        // (rcppml_hip_tail_scale_gram; fp32 k = 64 scales inside the Gram's partial-tile kernel)
        const bool w_std = !P.symmetric && !has_mask && !is_nb;
        if (!P.symmetric) {
            const bool rank = std_cd && n >= kOrderMinColumns;
            if (w_std)
                OPCHK(rcppml_hip_tail_scale_gram(c, dt, dH.p, k, n, P.norm_type, dsums.p, dd.p,   // :645 extract_scaling (+ the next H solve's work order), :715-722 G_w_saved
                                                 rank ? dswH.as<int>() : nullptr, rank ? dordH.as<int>() : nullptr, eps, 0.0, dGs.p));
            else
                OPCHK(rcppml_hip_scale_order(c, dt, dH.p, k, n, P.norm_type, dsums.p, dd.p,       // :645 extract_scaling
                                             rank ? dswH.as<int>() : nullptr, rank ? dordH.as<int>() : nullptr));
            if (rank) ordH_fresh = true;
        }

        // ================= W half-update (fit_cpu.hpp:711-893)
        if (P.symmetric) {
            // :659-704  Gram = W_T W_T^T, RHS = W_T A (forward product: A = A^T); both saved for the loss before the
            // features; nnls_batch semantics: zero start at iteration 0, residual-corrected warm start afterwards
            OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, eps, 0.0, dGs.p));                 // :664, :669-673
            if (P.L2_W > 0) OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, eps, P.L2_W, dG.p));
            else HIPCHK(hipMemcpyAsync(dG.p, dGs.p, (size_t)k * k * sizeof(T), hipMemcpyDeviceToDevice, s));
            if (graph_W) OPCHK(rcppml_hip_apply_graph_reg(c, dt, dG.p, dGWp.as<int>(), dGWi.as<int>(), dGWx.p, dW.p, k, m, P.gW_lambda));
            if (P.L21_W > 0) OPCHK(rcppml_hip_apply_l21(c, dt, dG.p, dW.p, k, m, P.L21_W));
            rhs_fwd(dW.p, dBw.p);                                                       // :665
            if (P.solver_mode == 0)
                OPCHK(rcppml_hip_solve_cd(c, dt, dG.p, dBw.p, dW.p, k, m, P.L1_W > 0 ? P.L1_W : 0.0, warm, warm ? 0 : 1, 0.0, 0.0,
                                          P.nonneg_W, P.cd_maxit, P.cd_tol, 0.0, P.ub_W, RCPPML_CD_AUTO, nullptr, nullptr));   // :684-692
            else
                OPCHK(rcppml_hip_solve_chol(c, dt, dG.p, dBw.p, dW.p, k, m, P.L1_W > 0 ? P.L1_W : 0.0, P.nonneg_W, P.ub_W));   // :679-683
        } else if (has_mask) {                                                          // :799-803 (mask before requires_irls())
            OPCHK(rcppml_hip_gram(c, dt, dH.p, k, n, eps, 0.0, dG.p));
            OPCHK(rcppml_hip_solve_masked(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, dMTp.as<int>(), dMTi.as<int>(),
                                          m, dH.p, dG.p, dW.p, k, P.L1_W, P.L2_W, P.nonneg_W, P.cd_maxit, P.cd_tol,
                                          P.solver_mode, warm));
            if (P.ub_W > 0) OPCHK(rcppml_hip_clip_upper(c, dt, dW.p, (int64_t)k * m, P.ub_W));      // :884-885
        } else if (is_nb) {                                                             // :811-852 theta_per_col = r of the row
            OPCHK(rcppml_hip_gram(c, dt, dH.p, k, n, eps, 0.0, dG.p));
            OPCHK(rcppml_hip_solve_irls(c, dt, P.loss_type, dTp.as<int>(), dTi.as<int>(), dTx.p, m, dH.p, dG.p, dW.p, k, P.L1_W,
                                        P.L2_W, P.nonneg_W, P.cd_maxit, P.irls_max_iter, P.irls_tol,
                                        (!is_gp && per_col) ? dtheta.p : nullptr,            // :820-830 PER_COL: the ROW of A^T
                                        (is_gp || per_col) ? nullptr : dtheta.p, P.tweedie_power, P.robust_delta));
            if (P.ub_W > 0) OPCHK(rcppml_hip_clip_upper(c, dt, dW.p, (int64_t)k * m, P.ub_W));      // :884-885
        } else {
            // :715-722 G_w_saved = gram(H) + eps: formed with the scaling of H above (w_std)
            // the solve's Gram: G_saved itself when nothing is added to it (no copy: one kernel boundary less per iteration)
            const bool gw_modified = P.L2_W > 0 || graph_W || P.L21_W > 0 || tgtW;
            void* const Gw = gw_modified ? dG.p : dGs.p;
            if (P.L2_W > 0) OPCHK(rcppml_hip_gram(c, dt, dH.p, k, n, eps, P.L2_W, dG.p)); // :738
            else if (gw_modified) HIPCHK(hipMemcpyAsync(dG.p, dGs.p, (size_t)k * k * sizeof(T), hipMemcpyDeviceToDevice, s));
            if (graph_W) OPCHK(rcppml_hip_apply_graph_reg(c, dt, dG.p, dGWp.as<int>(), dGWi.as<int>(), dGWx.p, dW.p, k, m, P.gW_lambda));   // :740-741
            if (P.L21_W > 0) OPCHK(rcppml_hip_apply_l21(c, dt, dG.p, dW.p, k, m, P.L21_W));   // :741-745 (current W_T)
            rhs_bwd(dH.p, dBw.p);
            void* Bw_use = tgtW ? apply_target(P.target_lambda_W, dTW, TG_W, dG.p, dBw.p, m) : dBw.p;   // the loss keeps the raw B_w (:786-789)
            if (P.solver_mode == 0) {
                const bool ord = use_order && iter > 0 && m >= kOrderMinColumns;
                if (ord && !ordW_fresh) OPCHK(rcppml_hip_order_columns(c, dswW.as<int>(), m, dordW.as<int>()));
                OPCHK(rcppml_hip_solve_cd(c, dt, Gw, Bw_use, dW.p, k, m, P.L1_W > 0 ? P.L1_W : 0.0, warm, zinit, 0.0, 0.0,
                                          P.nonneg_W, P.cd_maxit, P.cd_tol, 0.0, P.ub_W, RCPPML_CD_AUTO,
                                          use_order ? dswW.as<int>() : nullptr, ord ? dordW.as<int>() : nullptr));
                ordW_fresh = false;
            }
            else
                OPCHK(rcppml_hip_solve_chol(c, dt, Gw, Bw_use, dW.p, k, m, P.L1_W > 0 ? P.L1_W : 0.0, P.nonneg_W, P.ub_W));
        }
        if (P.angular_W > 0) OPCHK(rcppml_hip_angular_posthoc(c, dt, dW.p, k, m, P.angular_W));   // :886-887
        // the Gram-trick loss (:1729-1753) needs gram(W_T) of the scaled W_T right after this scaling: one call does both
        const bool nb_fused = P.loss_type == 5 && P.dispersion_mode == 2 && !(P.robust_delta > 0) && !has_mask;
        const bool mse_loss = !nb_fused && !has_mask && !is_nb;
        {
            const bool rank = std_cd && m >= kOrderMinColumns;
            if (mse_loss)
                OPCHK(rcppml_hip_tail_scale_gram_loss(c, dt, dW.p, k, m, P.norm_type, dsums.p, dd.p,   // :893 (+ the next W solve's work order), :1734-1753
                                                      rank ? dswW.as<int>() : nullptr, rank ? dordW.as<int>() : nullptr, eps, dtr.as<double>(),
                                                      dBw.p, dGs.p, dGwt.p, dloss.as<double>()));
            else
                OPCHK(rcppml_hip_scale_order(c, dt, dW.p, k, m, P.norm_type, dsums.p, dd.p,       // :893
                                             rank ? dswW.as<int>() : nullptr, rank ? dordW.as<int>() : nullptr));
            if (rank) ordW_fresh = true;
        }
        if (P.symmetric) HIPCHK(hipMemcpyAsync(dH.p, dW.p, (size_t)k * m * sizeof(T), hipMemcpyDeviceToDevice, s));   // :704 H = W_T

        // ================= NB size update (fit_cpu.hpp:1094-1265), then loss (fit_cpu.hpp:1684-1753)
        // (PER_ROW sizes without the robust modifier: both in one pass over A^T, the predictions of the size update reused by the loss)
        if (nb_fused) {
            OPCHK(rcppml_hip_nb_size_update_loss(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, P.nnz, dW.p, dd.p, dH.p, n, k,
                                                 P.nb_size_min, P.nb_size_max, dtheta.p, dloss.as<double>()));
        } else if (is_nb && !is_gp && per_col) {                                    // :1103-1162
            OPCHK(rcppml_hip_nb_size_update(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, n, dH.p, dd.p, dW.p, m, k,
                                            P.nb_size_min, P.nb_size_max, dtheta.p));
        } else if (is_nb && !is_gp && P.dispersion_mode != 0) {
            OPCHK(rcppml_hip_nb_size_update(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, dW.p, dd.p, dH.p, n, k,
                                            P.nb_size_min, P.nb_size_max, dtheta.p));
            if (P.dispersion_mode == 1) OPCHK(rcppml_hip_vec_global(c, dt, 1, dtheta.p, m));   // GLOBAL: median (nth_element at m/2), :1257-1262
        }
        // GP theta (fit_cpu.hpp:914-1008) / Gamma, inverse-Gaussian, Tweedie phi (:1561-1670), PER_ROW or GLOBAL
        if ((P.loss_type == 4 || is_pow) && per_col)                                 // :1009-1083, :1570-1611
            OPCHK(rcppml_hip_dispersion_update(c, dt, P.loss_type, 2, dAp.as<int>(), dAi.as<int>(), dAx.p, n, P.nnz,
                                               dH.p, dd.p, dW.p, m, k, P.tweedie_power, P.gamma_phi_min,
                                               P.loss_type == 4 ? P.gp_theta_max : P.gamma_phi_max, dtheta.p));
        else if ((P.loss_type == 4 || is_pow) && P.dispersion_mode != 0)
            OPCHK(rcppml_hip_dispersion_update(c, dt, P.loss_type, P.dispersion_mode, dTp.as<int>(), dTi.as<int>(), dTx.p, m, P.nnz,
                                               dW.p, dd.p, dH.p, n, k, P.tweedie_power, P.gamma_phi_min,
                                               P.loss_type == 4 ? P.gp_theta_max : P.gamma_phi_max, dtheta.p));
        if (nb_fused) {
            // loss already in dloss
        } else if (has_mask) {
            // :1685-1690 masked_loss comes before the explicit loss of a distribution: compute_loss(a, pred, config.loss) per
            // unmasked nonzero with its default theta = 0 and no robust modifier (masked_nnls.hpp:277) -- the fitted dispersion
            // is updated above and returned, but does not enter this number
            OPCHK(rcppml_hip_loss_masked(c, dt, P.loss_type, P.tweedie_power, dAp.as<int>(), dAi.as<int>(), dAx.p, dMp.as<int>(), dMi.as<int>(), n,
                                         dW.p, dd.p, dH.p, k, dloss.as<double>()));
        } else if (is_nb && per_col) {      // explicit_loss.hpp:59-71 theta_is_per_col: over CSC(A^T), whose nonzeros' "row" is the column j of A
            OPCHK(rcppml_hip_irls_loss(c, dt, P.loss_type, dTp.as<int>(), dTi.as<int>(), dTx.p, m, dH.p, dd.p, dW.p, dtheta.p, k, P.tweedie_power, P.robust_delta, dloss.as<double>()));
        } else if (is_nb) {
            OPCHK(rcppml_hip_irls_loss(c, dt, P.loss_type, dAp.as<int>(), dAi.as<int>(), dAx.p, n, dW.p, dd.p, dH.p, dtheta.p, k, P.tweedie_power, P.robust_delta, dloss.as<double>()));
        } else {
            // :1734-1753 Gram of the scaled W_T and the Gram-trick loss: formed with the scaling of W_T above (mse_loss).  B_w (raw RHS of
            // the W update) is exactly the h_at of loss_cross_term_sparse_via_At (fused_nnls.hpp:305-362): no third O(nnz k) pass.
        }
    };

    // From the third iteration on the launch sequence is identical every time (same kernels, same arguments, scratch
    // sizes settled), so it is captured once into a hipGraph and replayed: small inputs (hawaiibirds: ~27 launches for
    // < 50 us of GPU work) are bound by the host's launch rate, not by the kernels.  Plain MSE path only; any capture
    // failure falls back to eager launches.  RCPPML_GPU_NO_GRAPH=1 disables.
    const bool graph_ok = !is_nb && !has_mask && !unfused && !getenv("RCPPML_GPU_NO_GRAPH");   // (dense input: a few large launches, eager)
    struct GraphHolder {
        hipGraph_t g = nullptr; hipGraphExec_t e = nullptr; bool failed = false;
        ~GraphHolder() { if (e) (void)hipGraphExecDestroy(e); if (g) (void)hipGraphDestroy(g); }
    } gh;

    // ---- small plain fits: the whole loop as ONE persistent kernel on one XCD (kernels_small.hip.h): no launch per phase, no loss
    // round trip per iteration; the convergence rule runs on the device.  RCPPML_GPU_NO_SMALL=1 keeps the multi-launch loop.
    bool small_done = false;
    if (!dense && !has_mask && !is_nb && !unfused && !graph_H && !graph_W && P.L21_H == 0 && P.L21_W == 0 && P.angular_H == 0 &&
        P.angular_W == 0 && !P.projective && !P.symmetric && P.max_iter >= 1 && !env_no_small() &&
        rcppml_hip_als_small_eligible(m, n, P.nnz, k)) {
        DevBuf dres(8 * sizeof(double)), dhist((size_t)P.max_iter * sizeof(double));
        // (RCPPML_GPU_SMALL_GIVE_UP_TEST=1, test switch like RCPPML_GPU_DEVICES_FORCE: the kernel's first barrier gives up, so that the restart below runs)
        if (getenv("RCPPML_GPU_SMALL_GIVE_UP_TEST")) OPCHK(rcppml_hip_ctx_set_option(c, RCPPML_OPT_SMALL_GIVE_UP, 1));
        OPCHK(rcppml_hip_als_small_fit(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, dTp.as<int>(), dTi.as<int>(), dTx.p, m, n, P.nnz, k, dW.p, dH.p, dd.p,
                                       dtr.as<double>(), P.L1_H, P.L1_W, P.L2_H, P.L2_W, P.ub_H, P.ub_W, P.nonneg_H, P.nonneg_W, P.norm_type,
                                       P.solver_mode, P.cd_maxit, P.cd_tol, P.max_iter, P.tol, P.patience, 0, dhist.as<double>(), dres.as<double>()));
        double res[8] = {0};
        HIPCHK(hipMemcpyAsync(res, dres.p, sizeof(res), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (res[4] == 1.0) {
            iterations = (int)res[0]; converged = res[1] != 0.0; train_loss = res[2]; final_tol = res[3];
            if (P.loss_history || P.verbose) {
                std::vector<double> hist((size_t)iterations);
                HIPCHK(hipMemcpyAsync(hist.data(), dhist.p, hist.size() * sizeof(double), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                if (P.loss_history) std::copy(hist.begin(), hist.end(), P.loss_history);
                if (P.verbose) for (int i = 0; i < iterations; ++i) fprintf(stderr, "[rcppml_gpu] iter %d loss %.9g (one-kernel fit)\n", i + 1, hist[i]);
            }
            small_done = true;
        } else {
            // the kernel's workgroups did not all arrive on one XCD (its barrier gave up, nothing hung): start again on the multi-launch loop
            if (P.verbose) fprintf(stderr, "[rcppml_gpu] one-kernel fit gave up at its barrier; running the multi-launch loop\n");
            AsyncUpload<T> again;
            again.add(P.W, (size_t)k * m, dW);
            again.add(P.H, (size_t)k * n, dH);
            again.start(c->device, g.second_stream());
            again.finish(c);
            std::vector<T> ones(k, T(1));
            HIPCHK(hipMemcpyAsync(dd.p, ones.data(), k * sizeof(T), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        }
    }

    for (int iter = 0; iter < P.max_iter && !small_done; ++iter) {
        bool launched = false;
        if (graph_ok && !gh.failed && iter >= 2) {
            if (!gh.e) {
                if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    bool body_ok = true;
                    try { enqueue_iteration(iter); } catch (...) { body_ok = false; }
                    const hipError_t ec = hipStreamEndCapture(s, &gh.g);
                    if (!body_ok || ec != hipSuccess || gh.g == nullptr ||
                        hipGraphInstantiate(&gh.e, gh.g, nullptr, nullptr, 0) != hipSuccess) {
                        gh.failed = true; gh.e = nullptr;
                        (void)hipGetLastError();
                    }
                } else {
                    gh.failed = true;
                    (void)hipGetLastError();
                }
            }
            if (gh.e) { HIPCHK(hipGraphLaunch(gh.e, s)); launched = true; }
        }
        if (!launched) enqueue_iteration(iter);
        HIPCHK(hipMemcpyAsync(hloss, dloss.p, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        double loss_val = hloss[0];
        if (std::is_same<T, float>::value) loss_val = static_cast<double>(static_cast<float>(loss_val));
        last_loss = loss_val;
        if (P.loss_history) P.loss_history[iter] = loss_val;

        bool loss_converged = false;
        double rel = 0;
        if (iter > 0) {                                                                 // :1769-1775
            rel = std::fabs(prev_loss - loss_val) / (std::fabs(prev_loss) + 1e-15);
            final_tol = rel;
            if (rel < P.tol) loss_converged = true;
        }
        prev_loss = loss_val;
        if (P.verbose) fprintf(stderr, "[rcppml_gpu] iter %d loss %.9g rel %.3g\n", iter + 1, loss_val, rel);
        if (iter > 0) {                                                                 // :1797-1809
            if (loss_converged) {
                if (++patience_counter >= P.patience) {
                    converged = true; train_loss = prev_loss; iterations = iter + 1;
                    break;
                }
            } else patience_counter = 0;
        }
        iterations = iter + 1;
    }
    if (!converged && !small_done) train_loss = last_loss;

    if (is_nb && P.out_theta) {
        DevBuf& th = dtheta;
        download_cast<T>(c, th, dlen, P.out_theta, s);
        P.out_theta_len = (int)dlen;
    }
    phase("iterations", s);
    // ---- sort by descending d (core/result.hpp:169-188) on the device, download
    download_cast<T>(c, dd, (size_t)k, P.d, s);
    if (P.sort_model) {
        std::vector<int> idx(k);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return P.d[a] > P.d[b]; });
        std::vector<double> tmp(k);
        for (int i = 0; i < k; ++i) tmp[i] = P.d[idx[i]];
        std::copy(tmp.begin(), tmp.end(), P.d);
        DevBuf didx((size_t)k * sizeof(int));
        HIPCHK(hipMemcpyAsync(didx.p, idx.data(), (size_t)k * sizeof(int), hipMemcpyHostToDevice, s));
        // the right-hand-side buffers are free now: W_T and H leave through them with their rows (factors) permuted
        hipLaunchKernelGGL(permute_rows_kernel<T>, dim3((unsigned)(((size_t)k * m + 255) / 256)), dim3(256), 0, s, (const T*)dW.p, k, (int64_t)m, didx.as<int>(), (T*)dBw.p);
        hipLaunchKernelGGL(permute_rows_kernel<T>, dim3((unsigned)(((size_t)k * n + 255) / 256)), dim3(256), 0, s, (const T*)dH.p, k, (int64_t)n, didx.as<int>(), (T*)dBh.p);
        HIPCHK(hipGetLastError());
        download_cast<T>(c, dBw, (size_t)k * m, P.W, s);
        download_cast<T>(c, dBh, (size_t)k * n, P.H, s);
    } else {
        download_cast<T>(c, dW, (size_t)k * m, P.W, s);
        download_cast<T>(c, dH, (size_t)k * n, P.H, s);
    }
    phase("sort + download", s);
    P.out_iter = iterations; P.out_converged = converged ? 1 : 0; P.out_loss = train_loss; P.out_tol = final_tol;
}

// Shared body of the three NMF entry points
void nmf_entry(RCPPML_NMF_UNIFIED_ARGS, const int* mask_p, const int* mask_i, double cd_tol, int sort_model,
               int precision, double* loss_history, const double* target_H = nullptr, double target_lambda_H = 0,
               const double* target_W = nullptr, double target_lambda_W = 0, int theta_capacity = 0) {
    try {
        rcppml_err().clear();
        *out_status = -1;
        *out_theta_len = 0;
        // out_theta: the reference bridge hands over m doubles (gpu/bridge_nmf.hpp:284) and its entries carry no capacity; the
        // build-defined entries read the capacity from *out_theta_len ON INPUT (theta_capacity; <= 0 = the bridge's m).
        // dispersion = "per_col" writes n values: refused unless the caller's buffer is known to hold them
        const bool theta_holds_n = theta_capacity >= *n;
        (void)seed; (void)loss_every; (void)huber_delta;
        (void)gp_theta_min; (void)guide_H_labels_flat; (void)guide_H_ns;
        (void)guide_H_lambdas; (void)guide_H_ncs;
        // Reject what is not implemented so the caller falls back to CPU (SURVEY.md 8b "Semantics")
        if (*loss_type != 0 && (*loss_type < 4 || *loss_type > 8))
            throw std::runtime_error("loss_type must be MSE (0), GP (4), NB (5), Gamma (6), inverse Gaussian (7) or Tweedie (8) for this plugin build");
        if (*loss_type == 4 || *loss_type >= 6) {
            // GP with theta = 0 (Poisson / KL NMF) and the power-variance family with phi = 1: IRLS half-updates with
            // parameter-free weights; theta / phi (dispersion global or per row) only enter the GP likelihood and the output
            if (*gp_dispersion_mode == 3 && !theta_holds_n) throw std::runtime_error("dispersion='per_col' returns n values: the bridge's out_theta holds m (gpu/bridge_nmf.hpp:284); use rcppml_gpu_nmf_ex with *out_theta_len = capacity >= n on input");
            if (*gp_dispersion_mode < 0 || *gp_dispersion_mode > 3) throw std::runtime_error("bad dispersion mode");
            if (*k > 128) throw std::runtime_error("IRLS losses: k must be <= 128");
            if (*solver_mode != 0) throw std::runtime_error("IRLS losses require the CD solver");
        }
        if (*loss_type == 5) {
            if (*k > 128) throw std::runtime_error("NB loss: k must be <= 128");
            if (*gp_dispersion_mode == 3 && !theta_holds_n) throw std::runtime_error("dispersion='per_col' returns n values: the bridge's out_theta holds m (gpu/bridge_nmf.hpp:284); use rcppml_gpu_nmf_ex with *out_theta_len = capacity >= n on input");
            if (*solver_mode != 0) throw std::runtime_error("NB loss requires the CD solver");      // core/config.hpp:447-452
        }
        if (*robust_delta > 0 && *gp_dispersion_mode == 3 && !theta_holds_n)       // (robust MSE returns a theta vector too: zeros, n of them under per_col)
            throw std::runtime_error("dispersion='per_col' returns n values: the bridge's out_theta holds m (gpu/bridge_nmf.hpp:284); use rcppml_gpu_nmf_ex with *out_theta_len = capacity >= n on input");
        if (*robust_delta > 0) {   // Huber on Pearson residuals: every loss (MSE included) goes through the IRLS path
            if (*k > 128) throw std::runtime_error("robust loss: k must be <= 128");
            if (*solver_mode != 0) throw std::runtime_error("robust loss requires the CD solver");
            if (*L21_H != 0 || *L21_W != 0 || *ortho_H != 0 || *ortho_W != 0) throw std::runtime_error("robust loss with L21 / angular: not supported");
        }
        if ((*L21_H != 0 || *L21_W != 0 || *ortho_H != 0 || *ortho_W != 0) && (*loss_type != 0 || mask_p))
            throw std::runtime_error("L21 / angular penalties are implemented for the MSE path without explicit mask");
        if ((*ortho_H != 0 || *ortho_W != 0) && *k > 128) throw std::runtime_error("angular penalty: k must be <= 128");
        if (*L21_H < 0 || *L21_W < 0 || *ortho_H < 0 || *ortho_W < 0) throw std::runtime_error("negative L21 / angular penalty");
        if (*graph_W_nnz > 0 || *graph_H_nnz > 0) {
            if (*loss_type != 0 || *robust_delta > 0 || mask_p) throw std::runtime_error("graph regularisation: MSE path without explicit mask only");
            if (*k > 128) throw std::runtime_error("graph regularisation: k must be <= 128");
            if ((*graph_H_nnz > 0 && *graph_H_dim != *n) || (*graph_W_nnz > 0 && *graph_W_dim != *m)) throw std::runtime_error("graph Laplacian dimension mismatch");
        }
        if (*guide_H_count > 0) throw std::runtime_error("classifier guides not supported");
        if (*symmetric != 0) {
            if (*loss_type != 0 || *robust_delta > 0 || mask_p || *projective != 0) throw std::runtime_error("symmetric NMF: plain MSE path only");
            if (*m != *n) throw std::runtime_error("symmetric NMF needs a square matrix");
        }
        if (*projective != 0 && (*loss_type != 0 || *robust_delta > 0 || mask_p)) throw std::runtime_error("projective NMF: MSE path without explicit mask only");
        if (*solver_mode != 0 && *solver_mode != 1) throw std::runtime_error("solver_mode must be 0 (CD) or 1 (Cholesky+clip)");
        if (*k < 1 || *k > 256) throw std::runtime_error("k must be in [1,256]");
        if (*m < 1 || *n < 1) throw std::runtime_error("empty matrix");
        if (*norm_type < 0 || *norm_type > 2) throw std::runtime_error("bad norm_type");
        if (col_ptr[*n] != *nnz) throw std::runtime_error("col_ptr[n] != nnz");
        FitParams P;
        P.m = *m; P.n = *n; P.k = *k; P.nnz = *nnz;
        P.col_ptr = col_ptr; P.row_idx = row_idx; P.values = values;
        P.W = W; P.H = H; P.d = d;
        P.max_iter = *max_iter; P.tol = *tol;
        P.L1_H = *L1_H; P.L1_W = *L1_W; P.L2_H = *L2_H; P.L2_W = *L2_W; P.ub_H = *ub_H; P.ub_W = *ub_W;
        P.L21_H = *L21_H; P.L21_W = *L21_W; P.angular_H = *ortho_H; P.angular_W = *ortho_W;
        P.cd_maxit = *cd_maxit > 0 ? *cd_maxit : 10;          // src/RcppFunctions_nmf.cpp:22-95
        P.cd_tol = cd_tol > 0 ? cd_tol : 1e-8;
        P.verbose = *verbose; P.patience = *patience; P.nonneg_W = *nonneg_W; P.nonneg_H = *nonneg_H;
        P.norm_type = *norm_type; P.solver_mode = *solver_mode;
        P.mask_p = mask_p; P.mask_i = mask_i;
        P.sort_model = sort_model; P.loss_history = loss_history;
        P.loss_type = *loss_type; P.irls_max_iter = *irls_max_iter; P.irls_tol = *irls_tol;
        P.dispersion_mode = *gp_dispersion_mode; P.nb_size_init = *nb_size_init; P.nb_size_max = *nb_size_max;
        P.gp_theta_init = *gp_theta_init; P.gp_theta_max = *gp_theta_max; P.gamma_phi_init = *gamma_phi_init;
        P.gamma_phi_max = *gamma_phi_max; P.gamma_phi_min = *gamma_phi_min;
        P.nb_size_min = *nb_size_min; P.out_theta = out_theta; P.tweedie_power = *tweedie_power; P.robust_delta = *robust_delta; P.projective = *projective != 0 ? 1 : 0; P.symmetric = *symmetric != 0 ? 1 : 0;
        P.gH_p = graph_H_p; P.gH_i = graph_H_i; P.gH_x = graph_H_x; P.gH_nnz = *graph_H_nnz; P.gH_lambda = *graph_H_lambda;
        P.gW_p = graph_W_p; P.gW_i = graph_W_i; P.gW_x = graph_W_x; P.gW_nnz = *graph_W_nnz; P.gW_lambda = *graph_W_lambda;
        P.target_H = target_H; P.target_lambda_H = target_H ? target_lambda_H : 0;
        P.target_W = target_W; P.target_lambda_W = target_W ? target_lambda_W : 0;
        if ((P.target_H && P.target_lambda_H != 0) || (P.target_W && P.target_lambda_W != 0)) {
            if (*loss_type != 0 || *robust_delta > 0 || mask_p || *projective != 0 || *symmetric != 0)
                throw std::runtime_error("target regularisation: plain MSE fits only");
        }
        // RCPPML_GPU_DEVICES=n (build-defined: the reference bridge never transmits config.max_gpus, core/config.hpp:86):
        // plain sparse MSE fits shard their columns over n devices of this process (plugin_multi.hip); everything else,
        // and n <= 1, runs the single-device loop
        int ndev = 1;
        if (const char* e = getenv("RCPPML_GPU_DEVICES")) ndev = atoi(e);
        const char* force_env = getenv("RCPPML_GPU_DEVICES_FORCE");      // test switch: RCPPML_GPU_DEVICES=1 through the sharded loop
        const bool force_multi = ndev == 1 && getenv("RCPPML_GPU_DEVICES") && force_env && atoi(force_env) != 0;
        if ((ndev > 1 || force_multi) && rcppml_fit_multi(P, precision, ndev)) {
            // done
        } else if (precision == RCPPML_F64) fit<double>(P);
        else fit<float>(P);
        *out_iter = P.out_iter; *out_converged = P.out_converged; *out_loss = P.out_loss; *out_tol = P.out_tol;
        *out_theta_len = P.out_theta_len;
        *out_status = 0;
    } catch (const std::exception& e) {
        rcppml_err() = e.what();
        if (getenv("RCPPML_GPU_VERBOSE")) fprintf(stderr, "[rcppml_gpu] NMF error: %s\n", e.what());
        *out_status = -1;
    } catch (...) {
        rcppml_err() = "unknown error";
        *out_status = -1;
    }
}

#define RCPPML_NMF_DENSE_PASS                                                                               \
    A_data, m, n, k, W, H, d, max_iter, tol, L1_H, L1_W, L2_H, L2_W, L21_H, L21_W, ortho_H, ortho_W, ub_H, ub_W, cd_maxit,    \
        verbose, seed, loss_every, patience, nonneg_W, nonneg_H, loss_type, huber_delta, irls_max_iter, irls_tol, norm_type,   \
        gp_dispersion_mode, gp_theta_init, gp_theta_max, gp_theta_min, nb_size_init, nb_size_max, nb_size_min, robust_delta,   \
        tweedie_power, projective, symmetric, solver_mode, out_theta, out_theta_len, out_iter, out_converged, out_loss,        \
        out_status, out_tol
#define RCPPML_NMF_UNIFIED_PASS                                                                              \
    col_ptr, row_idx, values, m, n, nnz, k, W, H, d, max_iter, tol, L1_H, L1_W, L2_H, L2_W, L21_H, L21_W,    \
        ortho_H, ortho_W, ub_H, ub_W, cd_maxit, verbose, seed, loss_every, patience, nonneg_W, nonneg_H,     \
        loss_type, huber_delta, irls_max_iter, irls_tol, norm_type, projective, symmetric, solver_mode,      \
        graph_W_p, graph_W_i, graph_W_x, graph_W_dim, graph_W_nnz, graph_W_lambda, graph_H_p, graph_H_i,     \
        graph_H_x, graph_H_dim, graph_H_nnz, graph_H_lambda, gp_dispersion_mode, gp_theta_init,              \
        gp_theta_max, gp_theta_min, nb_size_init, nb_size_max, nb_size_min, gamma_phi_init, gamma_phi_max,   \
        gamma_phi_min, robust_delta, tweedie_power, out_theta, out_theta_len, guide_H_labels_flat,           \
        guide_H_ns, guide_H_lambdas, guide_H_ncs, guide_H_count, out_iter, out_converged, out_loss,          \
        out_status, out_tol

int env_sort() {
    const char* e = getenv("RCPPML_GPU_SORT");
    return (e && !strcmp(e, "0")) ? 0 : 1;
}

}  // namespace

// ----------------------------------------------------------------------------
// exported entry points
// ----------------------------------------------------------------------------
extern "C" void rcppml_gpu_detect(int* num_gpus, double* total_mem_mb, double* free_mem_mb, int* max_gpus,
                                  int* out_status) {
    *num_gpus = 0;
    *out_status = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return; }
    const int cap = (max_gpus && *max_gpus > 0) ? *max_gpus : 8;
    int cnt = 0;
    for (int dv = 0; dv < ndev && cnt < cap; ++dv) {
        size_t fr = 0, tot = 0;
        if (hipSetDevice(dv) != hipSuccess) continue;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) continue;
        total_mem_mb[cnt] = static_cast<double>(tot) / (1024.0 * 1024.0);
        free_mem_mb[cnt] = static_cast<double>(fr) / (1024.0 * 1024.0);
        ++cnt;
    }
    *num_gpus = cnt;
}

#define RCPPML_NMF_CV_PASS                                                                                  \
    col_ptr, row_idx, values, m, n, nnz, k, W, H, d, max_iter, tol, L1_H, L1_W, L2_H, L2_W, cd_maxit, verbose,      \
        seed_only_used_for_cv_seed_fallback, holdout_frac, cv_seed, mask_zeros, nonneg_W, nonneg_H, norm_type,      \
        loss_type, huber_delta, irls_max_iter, irls_tol, graph_W_p, graph_W_i, graph_W_x, graph_W_dim, graph_W_nnz, \
        graph_W_lambda, graph_H_p, graph_H_i, graph_H_x, graph_H_dim, graph_H_nnz, graph_H_lambda, projective,      \
        symmetric, solver_mode, out_iter, out_converged, out_train_loss, out_test_loss, out_best_test,              \
        out_best_iter, out_status

// ----------------------------------------------------------------------------
// Cross-validation fit (reference nmf/fit_cv.hpp:123-1667, MSE / sparse / standard updates / no user mask):
// speckled holdout mask, per-column Gram correction in both half-updates, held-out error + Gram-trick train loss every
// iteration, early stopping on the test loss (cv_patience = NMF_PATIENCE = 5: the bridge does not transmit it) and the
// relative-change convergence test on the test loss.  On return H carries d ("absorb d into H", :1636-1638) and d is
// returned as well, as the reference packages its result.
// ----------------------------------------------------------------------------
namespace {
int env_sort();
struct CvParams {
    int m, n, k; int64_t nnz;
    const int* col_ptr; const int* row_idx; const double* values;
    double *W, *H, *d;
    int max_iter; double tol;
    double L1_H, L1_W, L2_H, L2_W;
    int cd_maxit, verbose, nonneg_W, nonneg_H, norm_type, solver_mode;
    double holdout_fraction; unsigned long long cv_seed; int mask_zeros; int cv_patience; int sort_model;
    double* train_history = nullptr; double* test_history = nullptr;     // optional, max_iter entries each
    const int* gH_p = nullptr; const int* gH_i = nullptr; const double* gH_x = nullptr; int gH_nnz = 0; double gH_lambda = 0;
    const int* gW_p = nullptr; const int* gW_i = nullptr; const double* gW_x = nullptr; int gW_nnz = 0; double gW_lambda = 0;
    // IRLS losses (cv_detail.hpp:101-292).  The CV boundary carries loss_type, irls_max_iter and irls_tol only (bridge_nmf.hpp:77-99):
    // dispersion mode, GP theta bounds, Tweedie power and robust_delta take the reference's config defaults (core/config.hpp:151-172,
    // math/loss.hpp:109-115) unless the build-defined entry overrides them
    int loss_type = 0, irls_max_iter = 5; double irls_tol = 1e-4;
    int dispersion_mode = 2; double gp_theta_init = 0.1, gp_theta_max = 5.0, tweedie_power = 1.5, robust_delta = 0.0;
    double* out_theta = nullptr;          // m doubles (GP theta at exit), may be NULL
    // user mask (NMFConfig::mask under nmf_fit_cv, fit_cv.hpp:327-331): pattern CSC, m x n; build-defined entry rcppml_gpu_nmf_cv_masked_ex
    const int* mask_p = nullptr; const int* mask_i = nullptr;
    int out_iter = 0, out_converged = 0, out_best_iter = 0; double out_train = 0, out_test = 0, out_best_test = 0;
};

template <class T>
void fit_cv(CvParams& P) {
    constexpr int dt = DT<T>::id;
    const int m = P.m, n = P.n, k = P.k;
    CtxGuard g(env_device());
    rcppml_hip_ctx* c = g.c;
    hipStream_t s = g.s;
    DevBuf dAp, dAi, dAx, dTp, dTi, dTx;
    upload_ints(P.col_ptr, (size_t)n + 1, dAp, s);
    upload_ints(P.row_idx, (size_t)std::max<int64_t>(P.nnz, 1), dAi, s);
    upload_cast<T>(c, P.values, (size_t)std::max<int64_t>(P.nnz, 1), dAx, s);
    dTp.alloc(((size_t)m + 1) * sizeof(int));
    dTi.alloc((size_t)std::max<int64_t>(P.nnz, 1) * sizeof(int));
    dTx.alloc((size_t)std::max<int64_t>(P.nnz, 1) * sizeof(T));
    OPCHK(rcppml_hip_transpose_csc(c, dt, m, n, dAp.as<int>(), dAi.as<int>(), dAx.p, dTp.as<int>(), dTi.as<int>(), dTx.p));
    // user mask: its pattern and the pattern of its transpose stay on the device for the fit; the CV ops find them through the
    // context (rcppml_hip_ctx_set_cv_mask), cleared again on every way out
    const bool has_mask = P.mask_p != nullptr;
    DevBuf dMp, dMi, dMTp, dMTi;
    struct CvMaskGuard { rcppml_hip_ctx* c; bool on = false; ~CvMaskGuard() { if (on) (void)rcppml_hip_ctx_set_cv_mask(c, nullptr, nullptr, nullptr, nullptr); } } mask_guard{c};
    if (has_mask) {
        const int mnnz = P.mask_p[n];
        upload_ints(P.mask_p, (size_t)n + 1, dMp, s);
        upload_ints(P.mask_i, (size_t)std::max(mnnz, 1), dMi, s);
        dMTp.alloc(((size_t)m + 1) * sizeof(int));
        dMTi.alloc((size_t)std::max(mnnz, 1) * sizeof(int));
        OPCHK(rcppml_hip_transpose_csc(c, dt, m, n, dMp.as<int>(), dMi.as<int>(), nullptr, dMTp.as<int>(), dMTi.as<int>(), nullptr));
        OPCHK(rcppml_hip_ctx_set_cv_mask(c, dMp.as<int>(), dMi.as<int>(), dMTp.as<int>(), dMTi.as<int>()));
        mask_guard.on = true;
    }
    DevBuf dW, dH, dd;
    upload_cast<T>(c, P.W, (size_t)k * m, dW, s);
    upload_cast<T>(c, P.H, (size_t)k * n, dH, s);
    dd.alloc((size_t)k * sizeof(T));
    {
        std::vector<T> ones(k, T(1));
        HIPCHK(hipMemcpyAsync(dd.p, ones.data(), k * sizeof(T), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    DevBuf dBw((size_t)k * m * sizeof(T)), dG((size_t)k * k * sizeof(T)), dGs((size_t)k * k * sizeof(T)), dGwt((size_t)k * k * sizeof(T));
    DevBuf dsums((size_t)k * sizeof(T)), dtr(sizeof(double)), dloss(4 * sizeof(double)), dtest(2 * sizeof(double));
    double* hbuf = nullptr;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&hbuf), 6 * sizeof(double)));
    struct HostFree { double* p; ~HostFree() { (void)hipHostFree(p); } } hf{hbuf};
    OPCHK(rcppml_hip_sumsq(c, dt, dAx.p, P.nnz, dtr.as<double>()));
    const double eps = 1e-15;
    double best_test = std::numeric_limits<double>::max(), prev_conv = std::numeric_limits<double>::max();
    if (std::is_same<T, float>::value) { best_test = std::numeric_limits<float>::max(); prev_conv = best_test; }
    int best_iter = 0, patience_count = 0, iterations = 0;
    bool converged = false;
    double train_loss = 0, test_loss = 0, final_tol = 0;
    auto as_scalar = [](double v) { return std::is_same<T, float>::value ? static_cast<double>(static_cast<float>(v)) : v; };

    // graph Laplacians (apply_cv_features, variant_helpers.hpp:174-189), uploaded once
    DevBuf dGHp, dGHi, dGHx, dGWp, dGWi, dGWx;
    const bool graph_H = P.gH_p && P.gH_nnz > 0 && P.gH_lambda > 0, graph_W = P.gW_p && P.gW_nnz > 0 && P.gW_lambda > 0;
    if (graph_H) {
        upload_ints(P.gH_p, (size_t)n + 1, dGHp, s); upload_ints(P.gH_i, (size_t)P.gH_nnz, dGHi, s);
        upload_cast<T>(c, P.gH_x, (size_t)P.gH_nnz, dGHx, s);
    }
    if (graph_W) {
        upload_ints(P.gW_p, (size_t)m + 1, dGWp, s); upload_ints(P.gW_i, (size_t)P.gW_nnz, dGWi, s);
        upload_cast<T>(c, P.gW_x, (size_t)P.gW_nnz, dGWx, s);
    }
    const bool irls = P.loss_type != 0 || P.robust_delta > 0;        // LossConfig::requires_irls()
    const bool is_gp = P.loss_type == 4;
    DevBuf dtheta, dGadd;
    if (irls) {
        dGadd.alloc((size_t)k * k * sizeof(T));
        if (is_gp) {                                                  // fit_cv.hpp:195-202
            std::vector<T> th((size_t)m, static_cast<T>(P.dispersion_mode != 0 ? P.gp_theta_init : 0.0));
            dtheta.alloc((size_t)m * sizeof(T));
            HIPCHK(hipMemcpyAsync(dtheta.p, th.data(), (size_t)m * sizeof(T), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        }
    }
    // the additive CV features of one side (apply_cv_features on a zero matrix: L2 on the diagonal, graph term), added to every
    // column's weighted Gram by the IRLS solve
    auto features_only = [&](double l2, bool graph, DevBuf& gp, DevBuf& gi, DevBuf& gx, void* factor, int64_t len, double lambda) {
        HIPCHK(hipMemsetAsync(dGadd.p, 0, (size_t)k * k * sizeof(T), s));
        if (l2 > 0) OPCHK(rcppml_hip_add_diag(c, dt, dGadd.p, k, l2));
        if (graph) OPCHK(rcppml_hip_apply_graph_reg(c, dt, dGadd.p, gp.as<int>(), gi.as<int>(), gx.p, factor, k, len, lambda));
    };
    for (int iter = 0; iter < P.max_iter; ++iter) {
        if (irls) {
            // ---- IRLS path (fit_cv.hpp:446-456, :670-689): per-column weighted Grams over the training entries
            features_only(P.L2_H, graph_H, dGHp, dGHi, dGHx, dH.p, n, P.gH_lambda);
            OPCHK(rcppml_hip_solve_cv_irls(c, dt, P.loss_type, dAp.as<int>(), dAi.as<int>(), dAx.p, n, m, dW.p, dGadd.p, dH.p, k,
                                           P.holdout_fraction, P.cv_seed, P.mask_zeros, 0, P.L1_H, P.nonneg_H, P.cd_maxit, P.solver_mode,
                                           P.irls_max_iter, P.irls_tol, P.tweedie_power, P.robust_delta));
            OPCHK(rcppml_hip_row_norms(c, dt, dH.p, k, n, P.norm_type, dsums.p));
            OPCHK(rcppml_hip_apply_scaling(c, dt, dH.p, k, n, P.norm_type, dsums.p, dd.p));
            features_only(P.L2_W, graph_W, dGWp, dGWi, dGWx, dW.p, m, P.gW_lambda);
            OPCHK(rcppml_hip_solve_cv_irls(c, dt, P.loss_type, dTp.as<int>(), dTi.as<int>(), dTx.p, m, n, dH.p, dGadd.p, dW.p, k,
                                           P.holdout_fraction, P.cv_seed, P.mask_zeros, 1, P.L1_W, P.nonneg_W, P.cd_maxit, P.solver_mode,
                                           P.irls_max_iter, P.irls_tol, P.tweedie_power, P.robust_delta));
            OPCHK(rcppml_hip_row_norms(c, dt, dW.p, k, m, P.norm_type, dsums.p));
            OPCHK(rcppml_hip_apply_scaling(c, dt, dW.p, k, m, P.norm_type, dsums.p, dd.p));
            // ---- GP theta over the training entries (:866-961).  (The NB size and the Gamma-family phi the reference also
            // estimates here reach neither the CV weights, nor the CV losses, nor anything this boundary returns.)
            if (is_gp && P.dispersion_mode != 0)
                // (the user mask is NOT handed to this update: the reference's MM update skips held-out entries only -- `mask.is_holdout`,
                // nmf/fit_cv.hpp:866-961 -- and sums over user-masked ones like any other training entry; checked for ADVICE r5)
                OPCHK(rcppml_hip_cv_gp_theta_update(c, dt, P.dispersion_mode, dTp.as<int>(), dTi.as<int>(), dTx.p, m, P.nnz, dW.p, dd.p, dH.p,
                                                    n, k, P.holdout_fraction, P.cv_seed, P.gp_theta_max, dtheta.p));
            // ---- per-element losses (:1377-1443, :1546-1549)
            OPCHK(rcppml_hip_cv_irls_loss(c, dt, P.loss_type, dAp.as<int>(), dAi.as<int>(), dAx.p, n, m, dW.p, dd.p, dH.p,
                                          is_gp ? dtheta.p : nullptr, k, P.holdout_fraction, P.cv_seed, P.mask_zeros, P.tweedie_power,
                                          dloss.as<double>()));
            HIPCHK(hipMemcpyAsync(hbuf, dloss.p, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            train_loss = hbuf[1] > 0 ? as_scalar(as_scalar(hbuf[0]) / hbuf[1]) : 0.0;
            test_loss = hbuf[3] > 0 ? as_scalar(as_scalar(hbuf[2]) / hbuf[3]) : 0.0;
            if (P.train_history) P.train_history[iter] = train_loss;
            if (P.test_history) P.test_history[iter] = test_loss;
            double rel = 0;
            if (iter > 0) rel = std::fabs(prev_conv - test_loss) / (std::fabs(prev_conv) + 1e-15);
            if (test_loss < best_test) { best_test = test_loss; best_iter = iter; patience_count = 0; }
            else ++patience_count;
            if (P.verbose) fprintf(stderr, "[rcppml_gpu cv] iter %d train %.9g test %.9g best %.9g\n", iter + 1, train_loss, test_loss, best_test);
            iterations = iter + 1;
            if (P.cv_patience > 0 && patience_count >= P.cv_patience) { converged = false; break; }
            if (iter > 0) {
                final_tol = rel;
                if (rel < P.tol) { converged = true; break; }
            }
            prev_conv = test_loss;
            continue;
        }
        // ---- H half-update (:408-550): G = gram(W) (eps) + 1e-15 (:410) + L2
        OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, 2 * eps, P.L2_H, dG.p));
        if (graph_H) OPCHK(rcppml_hip_apply_graph_reg(c, dt, dG.p, dGHp.as<int>(), dGHi.as<int>(), dGHx.p, dH.p, k, n, P.gH_lambda));   // :416-417
        OPCHK(rcppml_hip_solve_cv(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, n, m, dW.p, dG.p, dH.p, k, P.holdout_fraction,
                                  P.cv_seed, P.mask_zeros, 0, P.L1_H, P.nonneg_H, P.cd_maxit, P.solver_mode));
        OPCHK(rcppml_hip_row_norms(c, dt, dH.p, k, n, P.norm_type, dsums.p));
        OPCHK(rcppml_hip_apply_scaling(c, dt, dH.p, k, n, P.norm_type, dsums.p, dd.p));
        // ---- W half-update (:555-860): G_H_saved = gram(H) (eps); G = G_H_saved + 1e-15 + L2
        OPCHK(rcppml_hip_gram(c, dt, dH.p, k, n, eps, 0.0, dGs.p));
        OPCHK(rcppml_hip_gram(c, dt, dH.p, k, n, 2 * eps, P.L2_W, dG.p));
        if (graph_W) OPCHK(rcppml_hip_apply_graph_reg(c, dt, dG.p, dGWp.as<int>(), dGWi.as<int>(), dGWx.p, dW.p, k, m, P.gW_lambda));   // :580-581
        if (!has_mask) OPCHK(rcppml_hip_rhs(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, dH.p, k, dBw.p));      // B_W_full: train + test
        OPCHK(rcppml_hip_solve_cv(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, n, dH.p, dG.p, dW.p, k, P.holdout_fraction,
                                  P.cv_seed, P.mask_zeros, 1, P.L1_W, P.nonneg_W, P.cd_maxit, P.solver_mode));
        OPCHK(rcppml_hip_row_norms(c, dt, dW.p, k, m, P.norm_type, dsums.p));
        OPCHK(rcppml_hip_apply_scaling(c, dt, dW.p, k, m, P.norm_type, dsums.p, dd.p));
        if (has_mask) {
            // ---- losses under a user mask (:1377-1443, "requires_irls() || use_mask"): explicit squared errors, train and test, over the
            // entries that are not user-masked -- the Gram trick would count the masked ones
            OPCHK(rcppml_hip_cv_irls_loss(c, dt, 0, dAp.as<int>(), dAi.as<int>(), dAx.p, n, m, dW.p, dd.p, dH.p, nullptr, k, P.holdout_fraction,
                                          P.cv_seed, P.mask_zeros, P.tweedie_power, dloss.as<double>()));
            HIPCHK(hipMemcpyAsync(hbuf, dloss.p, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            train_loss = hbuf[1] > 0 ? as_scalar(as_scalar(hbuf[0]) / hbuf[1]) : 0.0;
            test_loss = hbuf[3] > 0 ? as_scalar(as_scalar(hbuf[2]) / hbuf[3]) : 0.0;
        } else {
        // ---- losses (:1345-1550): held-out squared error; total squared error by the Gram trick with B_W_full
        OPCHK(rcppml_hip_cv_test_error(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, n, m, dW.p, dd.p, dH.p, k, P.holdout_fraction,
                                       P.cv_seed, P.mask_zeros, dtest.as<double>()));
        OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, eps, 0.0, dGwt.p));
        OPCHK(rcppml_hip_loss_mse(c, dt, dtr.as<double>(), dd.p, dW.p, dBw.p, k, m, dGwt.p, dGs.p, dloss.as<double>()));
        HIPCHK(hipMemcpyAsync(hbuf, dloss.p, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(hbuf + 4, dtest.p, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const double total_sq = std::max(as_scalar(hbuf[0]), 0.0);
        const double test_sq = as_scalar(hbuf[4]);
        const int64_t n_test = (int64_t)hbuf[5];
        const double train_sq = std::max(as_scalar(total_sq - test_sq), 0.0);
        const int64_t total_entries = P.mask_zeros ? P.nnz : (int64_t)m * n;
        const int64_t n_train = total_entries - n_test;
        train_loss = n_train > 0 ? as_scalar(train_sq / (double)n_train) : 0.0;
        test_loss = n_test > 0 ? as_scalar(test_sq / (double)n_test) : 0.0;
        }
        if (P.train_history) P.train_history[iter] = train_loss;
        if (P.test_history) P.test_history[iter] = test_loss;
        double rel = 0;
        if (iter > 0) rel = std::fabs(prev_conv - test_loss) / (std::fabs(prev_conv) + 1e-15);
        if (test_loss < best_test) { best_test = test_loss; best_iter = iter; patience_count = 0; }
        else ++patience_count;
        if (P.verbose) fprintf(stderr, "[rcppml_gpu cv] iter %d train %.9g test %.9g best %.9g\n", iter + 1, train_loss, test_loss, best_test);
        iterations = iter + 1;
        if (P.cv_patience > 0 && patience_count >= P.cv_patience) { converged = false; break; }
        if (iter > 0) {
            final_tol = rel;
            if (rel < P.tol) { converged = true; break; }
        }
        prev_conv = test_loss;
    }
    (void)final_tol;
    if (is_gp && P.out_theta) download_cast<T>(c, dtheta, (size_t)m, P.out_theta, s);      // result.theta (:1647)
    download_cast<T>(c, dW, (size_t)k * m, P.W, s);
    download_cast<T>(c, dH, (size_t)k * n, P.H, s);
    download_cast<T>(c, dd, (size_t)k, P.d, s);
    for (int j = 0; j < n; ++j) {                                // absorb d into H (:1636-1638), in the Scalar type
        double* h = P.H + (size_t)j * k;
        for (int i = 0; i < k; ++i) h[i] = static_cast<double>(static_cast<T>(h[i]) * static_cast<T>(P.d[i]));
    }
    if (P.sort_model) {                                          // core/result.hpp:169-188
        std::vector<int> idx(k);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return P.d[a] > P.d[b]; });
        std::vector<double> tmp(k);
        for (int j = 0; j < m; ++j) { double* w = P.W + (size_t)j * k; for (int i = 0; i < k; ++i) tmp[i] = w[idx[i]]; std::copy(tmp.begin(), tmp.end(), w); }
        for (int j = 0; j < n; ++j) { double* h = P.H + (size_t)j * k; for (int i = 0; i < k; ++i) tmp[i] = h[idx[i]]; std::copy(tmp.begin(), tmp.end(), h); }
        for (int i = 0; i < k; ++i) tmp[i] = P.d[idx[i]];
        std::copy(tmp.begin(), tmp.end(), P.d);
    }
    P.out_iter = iterations; P.out_converged = converged ? 1 : 0; P.out_train = train_loss; P.out_test = test_loss;
    P.out_best_test = best_test; P.out_best_iter = best_iter;
}

// what the reference's CV boundary does not carry (build-defined entry rcppml_gpu_nmf_cv_irls_ex)
struct CvExtra { int dispersion_mode; double gp_theta_init, gp_theta_max, tweedie_power, robust_delta; double* out_theta; };
void nmf_cv_entry(RCPPML_NMF_CV_ARGS, int sort_model, int precision, int cv_patience, double* train_history, double* test_history,
                  const CvExtra* cv_extra = nullptr, const int* mask_p = nullptr, const int* mask_i = nullptr, int mask_nnz = 0) {
    try {
        rcppml_err().clear();
        *out_status = -1;
        (void)seed_only_used_for_cv_seed_fallback; (void)huber_delta;
        // LossType (math/loss.hpp:36-47): 0 MSE; 4 GP, 5 NB, 6 Gamma, 7 inverse Gaussian, 8 Tweedie run the IRLS CV path; the legacy
        // MAE / Huber / KL losses (1-3) are handed back, as in the non-CV entries
        if (!(*loss_type == 0 || (*loss_type >= 4 && *loss_type <= 8))) throw std::runtime_error("CV: loss_type must be 0 (MSE) or 4..8 (GP, NB, Gamma, inverse Gaussian, Tweedie)");
        if ((*graph_H_nnz > 0 && *graph_H_dim != *n) || (*graph_W_nnz > 0 && *graph_W_dim != *m)) throw std::runtime_error("CV: graph Laplacian dimension mismatch");
        if (*projective != 0 || *symmetric != 0) throw std::runtime_error("CV: projective/symmetric NMF not supported");
        if (*solver_mode != 0 && *solver_mode != 1) throw std::runtime_error("CV: solver_mode must be 0 (CD) or 1 (Cholesky+clip)");
        if (*k < 1 || *k > 128) throw std::runtime_error("CV: k must be in [1,128]");
        if (*m < 1 || *n < 1) throw std::runtime_error("empty matrix");
        if (*norm_type < 0 || *norm_type > 2) throw std::runtime_error("bad norm_type");
        if (!(*holdout_frac > 0 && *holdout_frac < 1)) throw std::runtime_error("CV: holdout fraction must be in (0, 1)");
        if (col_ptr[*n] != *nnz) throw std::runtime_error("col_ptr[n] != nnz");
        CvParams P;
        P.m = *m; P.n = *n; P.k = *k; P.nnz = *nnz;
        P.col_ptr = col_ptr; P.row_idx = row_idx; P.values = values;
        P.W = W; P.H = H; P.d = d;
        P.max_iter = *max_iter; P.tol = *tol;
        P.L1_H = *L1_H; P.L1_W = *L1_W; P.L2_H = *L2_H; P.L2_W = *L2_W;
        P.cd_maxit = *cd_maxit > 0 ? *cd_maxit : 10;
        P.verbose = *verbose; P.nonneg_W = *nonneg_W; P.nonneg_H = *nonneg_H; P.norm_type = *norm_type;
        P.solver_mode = *solver_mode;
        P.holdout_fraction = *holdout_frac;
        // config.effective_cv_seed(): cv_seed unless 0, then seed (core/config.hpp:415-418)
        P.cv_seed = *cv_seed != 0 ? (unsigned long long)(unsigned)*cv_seed : (unsigned long long)(unsigned)*seed_only_used_for_cv_seed_fallback;
        P.mask_zeros = *mask_zeros != 0 ? 1 : 0;
        P.cv_patience = cv_patience; P.sort_model = sort_model;
        P.train_history = train_history; P.test_history = test_history;
        P.loss_type = *loss_type; P.irls_max_iter = *irls_max_iter > 0 ? *irls_max_iter : 5; P.irls_tol = *irls_tol;
        if (cv_extra) {
            P.dispersion_mode = cv_extra->dispersion_mode; P.gp_theta_init = cv_extra->gp_theta_init; P.gp_theta_max = cv_extra->gp_theta_max;
            P.tweedie_power = cv_extra->tweedie_power; P.robust_delta = cv_extra->robust_delta; P.out_theta = cv_extra->out_theta;
            if (P.dispersion_mode < 0 || P.dispersion_mode > 2) throw std::runtime_error("CV: dispersion_mode must be 0 (none), 1 (global) or 2 (per row)");
        }
        P.gH_p = graph_H_p; P.gH_i = graph_H_i; P.gH_x = graph_H_x; P.gH_nnz = *graph_H_nnz; P.gH_lambda = *graph_H_lambda;
        P.gW_p = graph_W_p; P.gW_i = graph_W_i; P.gW_x = graph_W_x; P.gW_nnz = *graph_W_nnz; P.gW_lambda = *graph_W_lambda;
        if (mask_p && mask_nnz > 0) {
            if (mask_p[*n] != mask_nnz) throw std::runtime_error("CV: mask_p[n] != mask_nnz");
            for (int j = 0; j < *n; ++j)
                for (int t = mask_p[j]; t < mask_p[j + 1]; ++t)
                    if (mask_i[t] < 0 || mask_i[t] >= *m || (t > mask_p[j] && mask_i[t] <= mask_i[t - 1])) throw std::runtime_error("CV: mask rows must be ascending inside a column and inside the matrix");
            P.mask_p = mask_p; P.mask_i = mask_i;
        }
        if (precision == RCPPML_F64) fit_cv<double>(P); else fit_cv<float>(P);
        *out_iter = P.out_iter; *out_converged = P.out_converged; *out_train_loss = P.out_train; *out_test_loss = P.out_test;
        *out_best_test = P.out_best_test; *out_best_iter = P.out_best_iter;
        *out_status = 0;
    } catch (const std::exception& e) {
        rcppml_err() = e.what();
        if (getenv("RCPPML_GPU_VERBOSE")) fprintf(stderr, "[rcppml_gpu cv] %s\n", e.what());
        *out_status = -1;
    } catch (...) {
        rcppml_err() = "unknown error";
        *out_status = -1;
    }
}
int env_cv_patience() { const char* e = getenv("RCPPML_GPU_CV_PATIENCE"); return e ? atoi(e) : 5; }   // NMF_PATIENCE
}  // namespace

extern "C" void rcppml_gpu_nmf_cv_unified_float(RCPPML_NMF_CV_ARGS) {
    const char* e = getenv("RCPPML_GPU_PRECISION");
    nmf_cv_entry(RCPPML_NMF_CV_PASS, env_sort(), (e && !strcmp(e, "fp64")) ? RCPPML_F64 : RCPPML_F32, env_cv_patience(), nullptr, nullptr);
}
extern "C" void rcppml_gpu_nmf_cv_unified_double(RCPPML_NMF_CV_ARGS) {
    const char* e = getenv("RCPPML_GPU_PRECISION");
    nmf_cv_entry(RCPPML_NMF_CV_PASS, env_sort(), (e && !strcmp(e, "fp32")) ? RCPPML_F32 : RCPPML_F64, env_cv_patience(), nullptr, nullptr);
}
// build-defined: + sort flag, precision, patience, loss histories (max_iter entries each, may be NULL)
extern "C" void rcppml_gpu_nmf_cv_ex(RCPPML_NMF_CV_ARGS, int* sort_model, int* precision, int* cv_patience, double* train_history,
                                     double* test_history) {
    nmf_cv_entry(RCPPML_NMF_CV_PASS, *sort_model, *precision, *cv_patience, train_history, test_history);
}

// build-defined: the same + what the reference's CV boundary has no slot for: dispersion mode (0 none / 1 global / 2 per row), GP theta
// init / max, Tweedie variance power, robust_delta, and the GP theta vector at exit (out_theta: m doubles, may be NULL)
extern "C" void rcppml_gpu_nmf_cv_irls_ex(RCPPML_NMF_CV_ARGS, int* sort_model, int* precision, int* cv_patience, double* train_history,
                                          double* test_history, int* dispersion_mode, double* gp_theta_init, double* gp_theta_max,
                                          double* tweedie_power, double* robust_delta, double* out_theta) {
    CvExtra ex{*dispersion_mode, *gp_theta_init, *gp_theta_max, *tweedie_power, *robust_delta, out_theta};
    nmf_cv_entry(RCPPML_NMF_CV_PASS, *sort_model, *precision, *cv_patience, train_history, test_history, &ex);
}

// build-defined: the same + a user mask (pattern CSC of the m x n mask; NULL / 0 = none) -- the reference's nmf_fit_cv honours
// NMFConfig::mask (nmf/fit_cv.hpp:327-331) but its CV boundary has no slot for it (gpu/bridge_nmf.hpp:77-99)
extern "C" void rcppml_gpu_nmf_cv_masked_ex(RCPPML_NMF_CV_ARGS, int* sort_model, int* precision, int* cv_patience, double* train_history,
                                            double* test_history, int* dispersion_mode, double* gp_theta_init, double* gp_theta_max,
                                            double* tweedie_power, double* robust_delta, double* out_theta, const int* mask_p,
                                            const int* mask_i, int* mask_nnz) {
    CvExtra ex{*dispersion_mode, *gp_theta_init, *gp_theta_max, *tweedie_power, *robust_delta, out_theta};
    nmf_cv_entry(RCPPML_NMF_CV_PASS, *sort_model, *precision, *cv_patience, train_history, test_history, &ex, mask_p, mask_i, mask_nnz ? *mask_nnz : 0);
}

// Zero-copy entry (reference src/gpu_bridge_nmf.cu:879-967, R/sp_gpu.R): the CSC arrays are DEVICE pointers whose
// addresses arrive as doubles (R has no int64); values are double; W (k x m), H (k x n), d are host buffers.
extern "C" void rcppml_gpu_nmf_zerocopy_double(double* d_col_ptr_addr, double* d_row_idx_addr, double* d_values_addr, int* m, int* n,
                                               double* nnz_d, int* k, double* W, double* H, double* d, int* max_iter, double* tol,
                                               double* L1_H, double* L1_W, double* L2_H, double* L2_W, double* L21_H, double* L21_W,
                                               double* ortho_H, double* ortho_W, double* ub_H, double* ub_W, int* cd_maxit,
                                               int* verbose, int* seed, int* loss_every, int* patience, int* nonneg_W,
                                               int* nonneg_H, int* loss_type, double* huber_delta, int* irls_max_iter,
                                               double* irls_tol, int* norm_type, int* out_iter, int* out_converged,
                                               double* out_loss, int* out_status, double* out_tol) {
    try {
        rcppml_err().clear();
        *out_status = -1;
        (void)seed; (void)loss_every; (void)huber_delta;
        auto to_ptr = [](double addr) { return reinterpret_cast<void*>(static_cast<uintptr_t>(addr)); };
        // the entry carries loss_type, irls_max_iter and irls_tol and nothing else of the IRLS configuration: the reference builds
        // its NMFConfig from them and leaves the rest at the defaults (src/gpu_bridge_nmf.cu:908-934: dispersion PER_ROW,
        // core/config.hpp:166-207) -- the same fit as the unified entry called with those defaults; theta is not returned
        if (*loss_type != 0 && (*loss_type < 4 || *loss_type > 8))
            throw std::runtime_error("loss_type must be MSE (0), GP (4), NB (5), Gamma (6), inverse Gaussian (7) or Tweedie (8) for this plugin build");
        if (*loss_type != 0 && *k > 128) throw std::runtime_error("IRLS losses: k must be <= 128");
        if (*loss_type != 0 && (*L21_H != 0 || *L21_W != 0 || *ortho_H != 0 || *ortho_W != 0))
            throw std::runtime_error("L21 / angular penalties are implemented for the MSE path");
        if (*k < 1 || *k > 256) throw std::runtime_error("k must be in [1,256]");
        if ((*ortho_H != 0 || *ortho_W != 0) && *k > 128) throw std::runtime_error("angular penalty: k must be <= 128");
        if (*m < 1 || *n < 1) throw std::runtime_error("empty matrix");
        if (*norm_type < 0 || *norm_type > 2) throw std::runtime_error("bad norm_type");
        FitParams P;
        P.m = *m; P.n = *n; P.k = *k; P.nnz = static_cast<int64_t>(*nnz_d);
        P.csc_on_device = 1;
        {   // run where the CSC lives (rcppml_sp_read_gpu allocates on the device it was given, R/sp_gpu.R), not on RCPPML_GPU_DEVICE
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, P.col_ptr) == hipSuccess) P.device = at.device;
            else (void)hipGetLastError();
        }
        P.col_ptr = static_cast<const int*>(to_ptr(*d_col_ptr_addr));
        P.row_idx = static_cast<const int*>(to_ptr(*d_row_idx_addr));
        P.values = static_cast<const double*>(to_ptr(*d_values_addr));
        P.W = W; P.H = H; P.d = d;
        P.max_iter = *max_iter; P.tol = *tol;
        P.L1_H = *L1_H; P.L1_W = *L1_W; P.L2_H = *L2_H; P.L2_W = *L2_W; P.ub_H = *ub_H; P.ub_W = *ub_W;
        P.L21_H = *L21_H; P.L21_W = *L21_W; P.angular_H = *ortho_H; P.angular_W = *ortho_W;
        P.cd_maxit = *cd_maxit > 0 ? *cd_maxit : 10;
        P.cd_tol = 1e-8;
        P.verbose = *verbose; P.patience = *patience; P.nonneg_W = *nonneg_W; P.nonneg_H = *nonneg_H;
        P.norm_type = *norm_type;
        P.solver_mode = 0;                                   // the entry has no solver argument: CD (config default)
        P.mask_p = nullptr; P.mask_i = nullptr; P.sort_model = env_sort(); P.loss_history = nullptr;
        P.loss_type = *loss_type; P.irls_max_iter = *irls_max_iter; P.irls_tol = *irls_tol;
        P.dispersion_mode = 2; P.nb_size_init = 10.0; P.nb_size_max = 1e6; P.nb_size_min = 0.01;      // core/config.hpp:166-207
        P.gp_theta_init = 0.1; P.gp_theta_max = 5.0; P.gamma_phi_init = 1.0; P.gamma_phi_max = 1e4; P.gamma_phi_min = 1e-6;
        P.tweedie_power = 1.5; P.robust_delta = 0.0; P.out_theta = nullptr;
        const char* e = getenv("RCPPML_GPU_PRECISION");
        if (e && !strcmp(e, "fp32")) fit<float>(P); else fit<double>(P);
        *out_iter = P.out_iter; *out_converged = P.out_converged; *out_loss = P.out_loss; *out_tol = P.out_tol;
        *out_status = 0;
    } catch (const std::exception& e) {
        rcppml_err() = e.what();
        if (getenv("RCPPML_GPU_VERBOSE")) fprintf(stderr, "[rcppml_gpu] zero-copy NMF error: %s\n", e.what());
        *out_status = -1;
    } catch (...) {
        rcppml_err() = "unknown error";
        *out_status = -1;
    }
}

// Dense-input NMF (reference bridge: gpu/bridge_nmf.hpp:101-126, 537-690; CUDA side src/gpu_bridge_nmf.cu).  A is an m x n
// column-major double array on the host; the fit is the reference's STANDARD (unfused) path with GEMM right-hand sides.
namespace {
void nmf_dense_entry(RCPPML_NMF_DENSE_ARGS, int precision) {
    try {
        rcppml_err().clear();
        *out_status = -1;
        (void)seed; (void)loss_every; (void)huber_delta; (void)gp_theta_min;
        if (out_theta_len) *out_theta_len = 0;
        const bool irls = *loss_type != 0 || *robust_delta > 0;
        if (*loss_type != 0 && (*loss_type < 4 || *loss_type > 8))
            throw std::runtime_error("loss_type must be MSE (0), GP (4), NB (5), Gamma (6), inverse Gaussian (7) or Tweedie (8) for this plugin build");
        if (irls) {          // the same conditions as on the sparse entries (nmf_entry above)
            if (*gp_dispersion_mode < 0 || *gp_dispersion_mode > 3) throw std::runtime_error("bad dispersion mode");
            if (*k > 128) throw std::runtime_error("IRLS losses: k must be <= 128");
            if (*solver_mode != 0) throw std::runtime_error("IRLS losses require the CD solver");
            if (*L21_H != 0 || *L21_W != 0 || *ortho_H != 0 || *ortho_W != 0) throw std::runtime_error("L21 / angular penalties are implemented for the MSE path");
            if (*projective != 0 || *symmetric != 0) throw std::runtime_error("projective / symmetric NMF: plain MSE path only");
            if ((int64_t)*m * *n > (int64_t)0x7fffffff) throw std::runtime_error("dense input with a distribution loss: m * n must fit 31 bits");
        }
        if (*projective != 0 && *symmetric != 0) throw std::runtime_error("projective and symmetric cannot both be true");
        if (*symmetric != 0 && *m != *n) throw std::runtime_error("symmetric NMF needs a square matrix");
        if (*solver_mode != 0 && *solver_mode != 1) throw std::runtime_error("solver_mode must be 0 (CD) or 1 (Cholesky+clip)");
        if (*k < 1 || *k > 128) throw std::runtime_error("k must be in [1,128]");
        if ((*ortho_H != 0 || *ortho_W != 0) && *k > 128) throw std::runtime_error("angular penalty: k must be <= 128");
        if (*L21_H < 0 || *L21_W < 0 || *ortho_H < 0 || *ortho_W < 0) throw std::runtime_error("negative L21 / angular penalty");
        if (*m < 1 || *n < 1) throw std::runtime_error("empty matrix");
        if (*norm_type < 0 || *norm_type > 2) throw std::runtime_error("bad norm_type");
        FitParams P;
        P.m = *m; P.n = *n; P.k = *k; P.nnz = (int64_t)*m * *n;
        P.dense = A_data;
        P.col_ptr = nullptr; P.row_idx = nullptr; P.values = nullptr;
        P.W = W; P.H = H; P.d = d;
        P.max_iter = *max_iter; P.tol = *tol;
        P.L1_H = *L1_H; P.L1_W = *L1_W; P.L2_H = *L2_H; P.L2_W = *L2_W; P.ub_H = *ub_H; P.ub_W = *ub_W;
        P.L21_H = *L21_H; P.L21_W = *L21_W; P.angular_H = *ortho_H; P.angular_W = *ortho_W;
        P.cd_maxit = *cd_maxit > 0 ? *cd_maxit : 10;
        P.cd_tol = 1e-8;
        P.verbose = *verbose; P.patience = *patience; P.nonneg_W = *nonneg_W; P.nonneg_H = *nonneg_H;
        P.norm_type = *norm_type; P.solver_mode = *solver_mode;
        P.projective = *projective != 0 ? 1 : 0; P.symmetric = *symmetric != 0 ? 1 : 0;
        P.mask_p = nullptr; P.mask_i = nullptr; P.sort_model = env_sort(); P.loss_history = nullptr;
        // the distribution losses (src/gpu_bridge_nmf.cu:685-700: the 50 pointers carry no gamma_phi_* -> NMFConfig defaults, core/config.hpp)
        P.loss_type = *loss_type; P.irls_max_iter = *irls_max_iter; P.irls_tol = *irls_tol;
        P.dispersion_mode = *gp_dispersion_mode; P.nb_size_init = *nb_size_init; P.nb_size_max = *nb_size_max; P.nb_size_min = *nb_size_min;
        P.gp_theta_init = *gp_theta_init; P.gp_theta_max = *gp_theta_max;
        P.gamma_phi_init = 1.0; P.gamma_phi_max = 1e4; P.gamma_phi_min = 1e-6;
        P.tweedie_power = *tweedie_power; P.robust_delta = *robust_delta; P.out_theta = irls ? out_theta : nullptr;    // max(m, n) doubles (gpu/bridge_nmf.hpp:622)
        if (precision == RCPPML_F64) fit<double>(P); else fit<float>(P);
        if (out_theta_len) *out_theta_len = P.out_theta_len;
        *out_iter = P.out_iter; *out_converged = P.out_converged; *out_loss = P.out_loss; *out_tol = P.out_tol;
        *out_status = 0;
    } catch (const std::exception& e) {
        rcppml_err() = e.what();
        if (getenv("RCPPML_GPU_VERBOSE")) fprintf(stderr, "[rcppml_gpu] dense NMF error: %s\n", e.what());
        *out_status = -1;
    } catch (...) {
        rcppml_err() = "unknown error";
        *out_status = -1;
    }
}
}  // namespace
extern "C" void rcppml_gpu_nmf_dense_unified_float(RCPPML_NMF_DENSE_ARGS) {
    const char* e = getenv("RCPPML_GPU_PRECISION");
    nmf_dense_entry(RCPPML_NMF_DENSE_PASS, (e && !strcmp(e, "fp64")) ? RCPPML_F64 : RCPPML_F32);
}
extern "C" void rcppml_gpu_nmf_dense_unified_double(RCPPML_NMF_DENSE_ARGS) {
    const char* e = getenv("RCPPML_GPU_PRECISION");
    nmf_dense_entry(RCPPML_NMF_DENSE_PASS, (e && !strcmp(e, "fp32")) ? RCPPML_F32 : RCPPML_F64);
}

extern "C" void rcppml_gpu_nmf_unified_float(RCPPML_NMF_UNIFIED_ARGS) {
    const char* e = getenv("RCPPML_GPU_PRECISION");
    const int prec = (e && !strcmp(e, "fp64")) ? RCPPML_F64 : RCPPML_F32;
    nmf_entry(RCPPML_NMF_UNIFIED_PASS, nullptr, nullptr, 1e-8, env_sort(), prec, nullptr);
}
extern "C" void rcppml_gpu_nmf_unified_double(RCPPML_NMF_UNIFIED_ARGS) {
    const char* e = getenv("RCPPML_GPU_PRECISION");
    const int prec = (e && !strcmp(e, "fp32")) ? RCPPML_F32 : RCPPML_F64;
    nmf_entry(RCPPML_NMF_UNIFIED_PASS, nullptr, nullptr, 1e-8, env_sort(), prec, nullptr);
}
extern "C" void rcppml_gpu_nmf_ex(RCPPML_NMF_UNIFIED_ARGS, const int* mask_p, const int* mask_i, int* mask_nnz,
                                  double* cd_tol, int* sort_model, int* precision, double* loss_history) {
    const bool use_mask = mask_p && mask_nnz && *mask_nnz > 0;
    // (build-defined contract of this entry: *out_theta_len is, ON INPUT, the capacity of out_theta in doubles -- dispersion = "per_col"
    // returns n values and is refused with status -1 when fewer fit; <= 0 means the reference bridge's m doubles)
    const int theta_cap = out_theta_len ? *out_theta_len : 0;
    nmf_entry(RCPPML_NMF_UNIFIED_PASS, use_mask ? mask_p : nullptr, use_mask ? mask_i : nullptr, *cd_tol, *sort_model,
              *precision, loss_history, nullptr, 0.0, nullptr, 0.0, theta_cap);
}

extern "C" void rcppml_gpu_nmf_target(RCPPML_NMF_UNIFIED_ARGS, const int* mask_p, const int* mask_i, int* mask_nnz,
                                      double* cd_tol, int* sort_model, int* precision, double* loss_history,
                                      const double* target_H, double* target_lambda_H, const double* target_W,
                                      double* target_lambda_W) {
    const bool use_mask = mask_p && mask_nnz && *mask_nnz > 0;
    const int theta_cap = out_theta_len ? *out_theta_len : 0;          // capacity of out_theta on input, as in rcppml_gpu_nmf_ex
    nmf_entry(RCPPML_NMF_UNIFIED_PASS, use_mask ? mask_p : nullptr, use_mask ? mask_i : nullptr, *cd_tol, *sort_model,
              *precision, loss_history, target_H, target_lambda_H ? *target_lambda_H : 0.0, target_W,
              target_lambda_W ? *target_lambda_W : 0.0, theta_cap);
}

// nnls()/predict() projection in fp64 (src/RcppFunctions_utils.cpp:313-366)
extern "C" void rcppml_gpu_nnls_double(const int* col_ptr, const int* row_idx, const double* values, int* m, int* n,
                                       int* nnz, int* k, const double* w_T, double* h, int* cd_maxit, double* cd_tol,
                                       double* L1, double* L2, double* ub, int* nonneg, int* warm, int* out_status) {
    try {
        rcppml_err().clear();
        *out_status = -1;
        if (*k < 1 || *k > 256) throw std::runtime_error("k must be in [1,256]");
        CtxGuard g(env_device());
        hipStream_t s = g.s;
        DevBuf dAp, dAi, dAx, dW, dH;
        upload_ints(col_ptr, (size_t)*n + 1, dAp, s);
        upload_ints(row_idx, (size_t)std::max(*nnz, 1), dAi, s);
        upload_cast<double>(g.c, values, (size_t)std::max(*nnz, 1), dAx, s);
        upload_cast<double>(g.c, w_T, (size_t)*k * *m, dW, s);
        upload_cast<double>(g.c, h, (size_t)*k * *n, dH, s);
        DevBuf dG((size_t)*k * *k * 8), dB((size_t)*k * *n * 8);
        // gram adds eps; c_nnls adds eps a second time (:327), then L2
        OPCHK(rcppml_hip_gram(g.c, RCPPML_F64, dW.p, *k, *m, 2e-15, *L2 > 0 ? *L2 : 0.0, dG.p));
        OPCHK(rcppml_hip_rhs(g.c, RCPPML_F64, dAp.as<int>(), dAi.as<int>(), dAx.p, *n, dW.p, *k, dB.p));
        // warm: B -= G h, CD with default cd_tol = 0 (:349-356); cold: X = 0, CD(cd_tol)
        OPCHK(rcppml_hip_solve_cd(g.c, RCPPML_F64, dG.p, dB.p, dH.p, *k, *n, 0.0, *warm ? 1 : 0, *warm ? 0 : 1, *L1, 0.0,
                                  *nonneg, *cd_maxit, *warm ? 0.0 : *cd_tol, *ub, 0.0, RCPPML_CD_AUTO, nullptr, nullptr));
        download_cast<double>(g.c, dH, (size_t)*k * *n, h, s);
        *out_status = 0;
    } catch (const std::exception& e) {
        rcppml_err() = e.what();
        *out_status = -1;
    } catch (...) { rcppml_err() = "unknown error"; *out_status = -1; }
}

// evaluate() in fp64 without densifying W H (src/RcppFunctions_utils.cpp:95-163 semantics: MEAN)
extern "C" void rcppml_gpu_evaluate_mse_double(const int* col_ptr, const int* row_idx, const double* values, int* m,
                                               int* n, int* nnz, int* k, const double* W_T, const double* d,
                                               const double* H, int* mask_zeros, double* out_loss, int* out_status) {
    try {
        rcppml_err().clear();
        *out_status = -1;
        CtxGuard g(env_device());
        hipStream_t s = g.s;
        DevBuf dAp, dAi, dAx, dW, dH, dd;
        upload_ints(col_ptr, (size_t)*n + 1, dAp, s);
        upload_ints(row_idx, (size_t)std::max(*nnz, 1), dAi, s);
        upload_cast<double>(g.c, values, (size_t)std::max(*nnz, 1), dAx, s);
        upload_cast<double>(g.c, W_T, (size_t)*k * *m, dW, s);
        upload_cast<double>(g.c, H, (size_t)*k * *n, dH, s);
        upload_cast<double>(g.c, d, (size_t)*k, dd, s);
        DevBuf dout(4 * sizeof(double));
        OPCHK(rcppml_hip_loss_nonzeros(g.c, RCPPML_F64, dAp.as<int>(), dAi.as<int>(), dAx.p, nullptr, nullptr, *n, dW.p,
                                       dd.p, dH.p, *k, dout.as<double>()));
        double nz[2];
        HIPCHK(hipMemcpyAsync(nz, dout.p, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (*mask_zeros) {
            *out_loss = *nnz > 0 ? nz[0] / static_cast<double>(*nnz) : 0.0;
        } else {
            // sum_all (a-p)^2 = sum_nz[(a-p)^2 - p^2] + sum_all p^2,  sum_all p^2 = <G_Wd, G_H> (Gram trick)
            DevBuf dGw((size_t)*k * *k * 8), dGh((size_t)*k * *k * 8);
            OPCHK(rcppml_hip_gram(g.c, RCPPML_F64, dW.p, *k, *m, 0.0, 0.0, dGw.p));
            OPCHK(rcppml_hip_gram(g.c, RCPPML_F64, dH.p, *k, *n, 0.0, 0.0, dGh.p));
            std::vector<double> gw((size_t)*k * *k), gh((size_t)*k * *k);
            HIPCHK(hipMemcpyAsync(gw.data(), dGw.p, gw.size() * 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(gh.data(), dGh.p, gh.size() * 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            double allp2 = 0;
            for (int a = 0; a < *k; ++a)
                for (int b = 0; b < *k; ++b) allp2 += d[a] * d[b] * gw[(size_t)b * *k + a] * gh[(size_t)b * *k + a];
            const double total = (nz[0] - nz[1]) + allp2;
            *out_loss = total / (static_cast<double>(*m) * static_cast<double>(*n));
        }
        *out_status = 0;
    } catch (const std::exception& e) {
        rcppml_err() = e.what();
        *out_status = -1;
    } catch (...) { rcppml_err() = "unknown error"; *out_status = -1; }
}

// ----------------------------------------------------------------------------
// Per-phase profile of the batch-CD ALS iteration: the reference's `rcppml_gpu_nmf_profile_double`
// (src/gpu_bridge_utils.cu:14-36 the eleven slots, :48-57 the fifteen pointers, :131-142 the untimed warm-up iteration,
// :145-285 the loop: an event pair around every phase, one synchronisation per iteration, `rel < tol` on the loss from the
// second timed iteration on).  Same protocol here -- SplitMix64(seed) fills W_T then H (one stream, :96-99 = initialize_factors),
// d = 1, fixed `cd_maxit` sweeps per solve (no tolerance stop), cold solves start from max(B, 0) and later ones from the previous
// factor (gpu/batch_nnls.cuh:66-75), L1 row scaling -- with this library's kernels in the slots:
//   [0] gram_H  [1] rhs_H (planned, LDS row tiles)  [2] nnls_H  [3] norm_H  [4] gram_W
//   [5] rhs_W by the plan-free gather kernel on CSC(A^T) (the reference's slot times its atomicAdd form as the baseline; there is
//       no atomic form in this build -- the gather kernel is what runs when no plan exists)
//   [6] rhs_W planned (the one the solve uses)  [7] nnls_W  [8] norm_W  [9] loss (Gram(W) + Gram-trick terms)  [10] iteration
// The loss is the exact ||A - W diag(d) H||^2 of the factors the iteration ends with (fit_cpu.hpp:1710-1753); the reference's
// profiler pairs the new W with the right-hand side of the old one (:248-262) -- its value only feeds the stopping rule.
namespace {
template <class T>
__global__ void clip_copy_kernel(const T* __restrict__ in, int64_t n, T* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = in[e] > T(0) ? in[e] : T(0);
}
// uniform<double>() of SplitMix64 (rng/rng.hpp:100-104): next() / 2^64, state advanced by the golden gamma
struct SplitMixStream {
    uint64_t s;
    explicit SplitMixStream(uint64_t seed) : s(seed == 0 ? 12345ULL : seed) {}
    double uniform() {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        return static_cast<double>(z) / 18446744073709551616.0;
    }
};
}  // namespace

extern "C" void rcppml_gpu_nmf_profile_double(const int* col_ptr, const int* row_idx, const double* values, int* m_ptr, int* n_ptr,
                                              int* nnz_ptr, int* k_ptr, int* max_iter_ptr, double* tol_ptr, int* cd_maxit_ptr,
                                              int* seed_ptr, double* out_phase_ms_total, double* out_phase_ms_per_iter,
                                              int* out_n_iters, int* out_status) {
    constexpr int NP = 11;
    for (int p = 0; p < NP; ++p) out_phase_ms_total[p] = out_phase_ms_per_iter[p] = 0.0;
    *out_n_iters = 0;
    *out_status = -1;
    hipEvent_t ev0[NP] = {}, ev1[NP] = {};
    try {
        rcppml_err().clear();
        const int m = *m_ptr, n = *n_ptr, nnz = *nnz_ptr, k = *k_ptr, max_iter = *max_iter_ptr, cd_maxit = *cd_maxit_ptr;
        const double tol = *tol_ptr;
        if (k < 1 || k > 256) throw std::runtime_error("k must be in [1,256]");
        if (m < 1 || n < 1 || nnz < 0 || col_ptr[n] != nnz) throw std::runtime_error("profile: inconsistent CSC");
        constexpr int dt = RCPPML_F64;
        CtxGuard g(env_device());
        rcppml_hip_ctx* c = g.c;
        hipStream_t s = g.s;
        DevBuf dAp, dAi, dAx, dTp, dTi, dTx, dW, dH, dd;
        upload_ints(col_ptr, (size_t)n + 1, dAp, s);
        upload_ints(row_idx, (size_t)std::max(nnz, 1), dAi, s);
        upload_cast<double>(c, values, (size_t)std::max(nnz, 1), dAx, s);
        dTp.alloc(((size_t)m + 1) * sizeof(int));
        dTi.alloc((size_t)std::max(nnz, 1) * sizeof(int));
        dTx.alloc((size_t)std::max(nnz, 1) * sizeof(double));
        OPCHK(rcppml_hip_transpose_csc(c, dt, m, n, dAp.as<int>(), dAi.as<int>(), dAx.p, dTp.as<int>(), dTi.as<int>(), dTx.p));
        struct PlanGuard { rcppml_rhs_plan* p = nullptr; ~PlanGuard() { rcppml_hip_rhs_plan_destroy(p); } } planA, planT;
        if (nnz >= (1 << 20)) {
            plan_or_none(rcppml_hip_rhs_plan_create(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, n, m, k, 0, 0, &planA.p), planA.p);
            plan_or_none(rcppml_hip_rhs_plan_create(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, n, k, 0, 0, &planT.p), planT.p);
        }
        {
            SplitMixStream rng(static_cast<uint32_t>(*seed_ptr));
            std::vector<double> w((size_t)k * m), h((size_t)k * n), ones(k, 1.0);
            for (auto& v : w) v = rng.uniform();
            for (auto& v : h) v = rng.uniform();
            upload_cast<double>(c, w.data(), w.size(), dW, s);
            upload_cast<double>(c, h.data(), h.size(), dH, s);
            upload_cast<double>(c, ones.data(), ones.size(), dd, s);
        }
        DevBuf dBh((size_t)k * n * 8), dBw((size_t)k * m * 8), dBbase((size_t)k * m * 8);
        DevBuf dG((size_t)k * k * 8), dGs((size_t)k * k * 8), dGwt((size_t)k * k * 8), dsums((size_t)k * 8);
        DevBuf dtr(sizeof(double)), dloss(4 * sizeof(double));
        OPCHK(rcppml_hip_sumsq(c, dt, dAx.p, nnz, dtr.as<double>()));
        for (int p = 0; p < NP; ++p) { HIPCHK(hipEventCreate(&ev0[p])); HIPCHK(hipEventCreate(&ev1[p])); }
        auto rhs_fwd = [&]() {
            if (planA.p) OPCHK(rcppml_hip_rhs_planned(c, planA.p, dW.p, dBh.p));
            else OPCHK(rcppml_hip_rhs(c, dt, dAp.as<int>(), dAi.as<int>(), dAx.p, n, dW.p, k, dBh.p));
        };
        auto rhs_bwd = [&]() {
            if (planT.p) OPCHK(rcppml_hip_rhs_planned(c, planT.p, dH.p, dBw.p));
            else OPCHK(rcppml_hip_rhs(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, dH.p, k, dBw.p));
        };
        auto solve = [&](DevBuf& B, DevBuf& X, int64_t ncols, bool warm) {
            if (!warm) {
                const int64_t tot = (int64_t)k * ncols;
                hipLaunchKernelGGL(clip_copy_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, B.as<double>(), tot, X.as<double>());
                HIPCHK(hipGetLastError());
            }
            // the previous (or clipped) solution as the start: b -= G x, then cd_maxit sweeps, no tolerance stop
            OPCHK(rcppml_hip_solve_cd(c, dt, dG.p, B.p, X.p, k, ncols, 0.0, 1, 0, 0.0, 0.0, 1, cd_maxit, 0.0, 0.0, 0.0, RCPPML_CD_AUTO, nullptr, nullptr));
        };
        auto scale = [&](DevBuf& X, int64_t ncols) {
            OPCHK(rcppml_hip_row_norms(c, dt, X.p, k, ncols, 0, dsums.p));
            OPCHK(rcppml_hip_apply_scaling(c, dt, X.p, k, ncols, 0, dsums.p, dd.p));
        };
        auto timed = [&](int p, auto&& body) {
            HIPCHK(hipEventRecord(ev0[p], s));
            body();
            HIPCHK(hipEventRecord(ev1[p], s));
        };
        // warm-up iteration (:131-142): cold solves, no timing
        OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, 1e-15, 0.0, dG.p));
        rhs_fwd();
        solve(dBh, dH, n, false);
        scale(dH, n);
        OPCHK(rcppml_hip_gram(c, dt, dH.p, k, n, 1e-15, 0.0, dG.p));
        rhs_bwd();
        solve(dBw, dW, m, false);
        scale(dW, m);
        HIPCHK(hipStreamSynchronize(s));
        double prev_loss = std::numeric_limits<double>::max();
        bool converged = false;
        int iters = 0;
        double hl[4];
        for (int iter = 0; iter < max_iter && !converged; ++iter) {
            HIPCHK(hipEventRecord(ev0[10], s));
            timed(0, [&] { OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, 1e-15, 0.0, dG.p)); });
            timed(1, [&] { rhs_fwd(); });
            timed(2, [&] { solve(dBh, dH, n, iter > 0); });
            timed(3, [&] { scale(dH, n); });
            timed(4, [&] { OPCHK(rcppml_hip_gram(c, dt, dH.p, k, n, 1e-15, 0.0, dG.p)); });
            timed(5, [&] { OPCHK(rcppml_hip_rhs(c, dt, dTp.as<int>(), dTi.as<int>(), dTx.p, m, dH.p, k, dBbase.p)); });
            timed(6, [&] { rhs_bwd(); });
            HIPCHK(hipMemcpyAsync(dGs.p, dG.p, (size_t)k * k * 8, hipMemcpyDeviceToDevice, s));           // Gram(H) for the loss
            timed(7, [&] { solve(dBw, dW, m, iter > 0); });
            timed(8, [&] { scale(dW, m); });
            timed(9, [&] {
                OPCHK(rcppml_hip_gram(c, dt, dW.p, k, m, 1e-15, 0.0, dGwt.p));
                OPCHK(rcppml_hip_loss_mse(c, dt, dtr.as<double>(), dd.p, dW.p, dBw.p, k, m, dGwt.p, dGs.p, dloss.as<double>()));
            });
            HIPCHK(hipEventRecord(ev1[10], s));
            HIPCHK(hipMemcpyAsync(hl, dloss.p, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            for (int p = 0; p < NP; ++p) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, ev0[p], ev1[p]));
                out_phase_ms_total[p] += static_cast<double>(ms);
            }
            ++iters;
            const double loss_val = hl[0];
            if (iter > 0 && prev_loss > 0) {
                const double rel = std::fabs(prev_loss - loss_val) / (std::fabs(prev_loss) + 1e-15);
                if (rel < tol) converged = true;
            }
            prev_loss = loss_val;
        }
        *out_n_iters = iters;
        if (iters > 0) for (int p = 0; p < NP; ++p) out_phase_ms_per_iter[p] = out_phase_ms_total[p] / iters;
        *out_status = 0;
    } catch (const std::exception& e) {
        rcppml_err() = e.what();
        *out_status = -1;
    } catch (...) { rcppml_err() = "unknown error"; *out_status = -1; }
    for (int p = 0; p < NP; ++p) { if (ev0[p]) (void)hipEventDestroy(ev0[p]); if (ev1[p]) (void)hipEventDestroy(ev1[p]); }
}
