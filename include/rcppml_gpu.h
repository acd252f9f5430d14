/* ============================================================================
 * rcppml_gpu.h -- C ABI of RcppML_gpu.so, the MI355X (gfx950) backend for RcppML's
 * alternating-NNLS NMF hot path.
 *
 * Two layers, both plain C (pointers and sizes only; no torch / Rcpp / Eigen types):
 *
 *  (1) PLUGIN BOUNDARY -- host-pointer entry points the unmodified R package binds:
 *      R `.C()` through `.gpu_call` (reference R/gpu_backend.R:24-27) and C++
 *      `dlsym(RTLD_DEFAULT, name)` from RcppML.so (reference
 *      inst/include/FactorNet/gpu/loader.hpp:44-48).  Signatures are byte-for-byte the
 *      reference plugin's (src/gpu_bridge_cluster.cu:24-46, src/gpu_bridge_nmf.cu:34-80,
 *      type inst/include/FactorNet/gpu/bridge_nmf.hpp:39-75).  Every scalar is a pointer
 *      (R `.C` convention).  Nothing throws across the boundary: failures set
 *      *out_status = -1 (reference src/gpu_bridge_nmf.cu:206-209), which makes the caller
 *      fall back to its CPU path (inst/include/FactorNet/nmf/fit.hpp:125-133).
 *
 *  (2) DEVICE-LEVEL OPS -- the kernels of the path on caller-owned DEVICE memory and a
 *      caller-supplied HIP stream.  This is what the host harness (the Python modules under rcppml_amd/, one
 *      process per GPU under torch.distributed/RCCL) and bench.py drive; PyTorch only
 *      provides the allocations, the stream and the collectives.
 *
 * All dense matrices are column-major with the factor rank k as leading dimension:
 *   W_T : k x m   (reference "W" in/out array of the ABI, W[f + i*k])
 *   H   : k x n
 *   G   : k x k
 * Sparse input is CSC with int32 indices (dgCMatrix / Eigen::SparseMatrix<_,ColMajor,int>).
 * ==========================================================================*/
#ifndef RCPPML_GPU_H
#define RCPPML_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCPPML_GPU_API __attribute__((visibility("default")))

/* --------------------------------------------------------------------------
 * (1) Plugin boundary
 * ------------------------------------------------------------------------*/

/* Replaces reference src/gpu_bridge_cluster.cu:24-46 (callers: R/gpu_backend.R:101-106,
 * gpu/loader.hpp:73-92 -- arrays of length *max_gpus, normally 8).
 * out_status = 0 and num_gpus > 0  <=>  backend available. */
RCPPML_GPU_API void rcppml_gpu_detect(int* num_gpus, double* total_mem_mb, double* free_mem_mb,
                                      int* max_gpus, int* out_status);

/* Replaces reference src/gpu_bridge_nmf.cu:460-624 (`rcppml_gpu_nmf_unified_float`, the symbol
 * gpu/bridge_nmf.hpp:187 resolves) and :34-210 (`_double`).  73 pointer arguments, order per
 * SURVEY.md Appendix B.  `_float` computes in fp32 (like the reference; set env
 * RCPPML_GPU_PRECISION=fp64 to force fp64), `_double` computes in fp64.  Arrays at the ABI are
 * always double.  W (k x m), H (k x n), d (k) are in/out.  Unlike the reference plugin this one
 * sorts factors by descending d before returning (what CPU nmf() returns) unless env
 * RCPPML_GPU_SORT=0.  Implemented: losses MSE / NB / GP(KL) / Gamma / inverse Gaussian / Tweedie with
 * dispersion none, global or per-row and the robust (Huber) modifier; L1, L2, L21, angular, graph
 * Laplacians, upper bounds, nonneg, projective and symmetric NMF (MSE path); CD and Cholesky+clip;
 * env RCPPML_GPU_DEVICES=n shards plain MSE fits over n devices (plugin_multi.hip).  Not implemented
 * -- REJECTED with *out_status = -1 so the caller falls back to CPU rather than silently dropping
 * them: classifier guides, dispersion = per_col (n values: the bridge's out_theta holds m, gpu/bridge_nmf.hpp:284 --
 * rcppml_gpu_nmf_ex takes a capacity), zero-inflated losses, k > 256 (k > 128 for IRLS losses,
 * explicit masks, angular and graph penalties).  Target regularisation has no slot in these 73 arguments: see
 * rcppml_gpu_nmf_target below. */
#define RCPPML_NMF_UNIFIED_ARGS                                                                    \
    const int* col_ptr, const int* row_idx, const double* values, int* m, int* n, int* nnz, int* k, \
        double* W, double* H, double* d, int* max_iter, double* tol, double* L1_H, double* L1_W,   \
        double* L2_H, double* L2_W, double* L21_H, double* L21_W, double* ortho_H, double* ortho_W, \
        double* ub_H, double* ub_W, int* cd_maxit, int* verbose, int* seed, int* loss_every,       \
        int* patience, int* nonneg_W, int* nonneg_H, int* loss_type, double* huber_delta,          \
        int* irls_max_iter, double* irls_tol, int* norm_type, int* projective, int* symmetric,     \
        int* solver_mode, const int* graph_W_p, const int* graph_W_i, const double* graph_W_x,     \
        int* graph_W_dim, int* graph_W_nnz, double* graph_W_lambda, const int* graph_H_p,          \
        const int* graph_H_i, const double* graph_H_x, int* graph_H_dim, int* graph_H_nnz,         \
        double* graph_H_lambda, int* gp_dispersion_mode, double* gp_theta_init,                    \
        double* gp_theta_max, double* gp_theta_min, double* nb_size_init, double* nb_size_max,     \
        double* nb_size_min, double* gamma_phi_init, double* gamma_phi_max, double* gamma_phi_min, \
        double* robust_delta, double* tweedie_power, double* out_theta, int* out_theta_len,        \
        const int* guide_H_labels_flat, const int* guide_H_ns, const double* guide_H_lambdas,      \
        const int* guide_H_ncs, int* guide_H_count, int* out_iter, int* out_converged,             \
        double* out_loss, int* out_status, double* out_tol

/* StreamPress / SparsePress v2 `.spz` reader.  Replaces reference `rcppml_sp_read_gpu` /
 * `rcppml_sp_free_gpu` (src/sp_gpu_bridge.cu:41-123, :132-155; bound by R/sp_gpu.R through .C()): reads the file,
 * entropy-decodes it ON THE DEVICE (one wavefront per rANS stream) and returns device pointers to int32 col_ptr (n+1),
 * int32 row_idx (nnz) and double values (nnz) -- the arrays rcppml_gpu_nmf_zerocopy_double takes -- with the addresses
 * stored in doubles (R has no 64-bit integer).  out_status: 0 ok, 1 cannot open, 2 short read, 3 file too small,
 * 4 not a v2 file, 5 decode error.  Row-sorted files: the stored row permutation is applied as the reference decoder applies it.
 * The caller releases the three arrays with rcppml_sp_free_gpu, which also zeroes the addresses. */
RCPPML_GPU_API void rcppml_sp_read_gpu(const char** path_ptr, int* device_id, double* out_col_ptr_addr,
                                       double* out_row_idx_addr, double* out_values_addr, int* out_m, int* out_n,
                                       double* out_nnz, int* out_status);
RCPPML_GPU_API void rcppml_sp_free_gpu(double* col_ptr_addr, double* row_idx_addr, double* values_addr, int* out_status);
/* Zero-copy NMF on a device-resident CSC.  Replaces reference `rcppml_gpu_nmf_zerocopy_double`
 * (src/gpu_bridge_nmf.cu:879-967, called from R/sp_gpu.R through .C()): col_ptr (int32), row_idx (int32) and values
 * (double) are DEVICE pointers whose addresses are passed as doubles (R has no 64-bit integer); W (k x m), H (k x n), d are
 * host buffers, in/out.  MSE loss, CD solver (the entry carries no solver argument), L1 / L2 / L21 / angular / bounds.
 * No upload of A: only the transpose, the factors and the per-iteration loss touch PCIe. */
RCPPML_GPU_API void rcppml_gpu_nmf_zerocopy_double(double* d_col_ptr_addr, double* d_row_idx_addr, double* d_values_addr,
                                                   int* m, int* n, double* nnz_d, int* k, double* W, double* H, double* d,
                                                   int* max_iter, double* tol, double* L1_H, double* L1_W, double* L2_H,
                                                   double* L2_W, double* L21_H, double* L21_W, double* ortho_H,
                                                   double* ortho_W, double* ub_H, double* ub_W, int* cd_maxit, int* verbose,
                                                   int* seed, int* loss_every, int* patience, int* nonneg_W, int* nonneg_H,
                                                   int* loss_type, double* huber_delta, int* irls_max_iter, double* irls_tol,
                                                   int* norm_type, int* out_iter, int* out_converged, double* out_loss,
                                                   int* out_status, double* out_tol);

/* Cross-validation NMF.  Replaces reference `rcppml_gpu_nmf_cv_unified_float` (type
 * inst/include/FactorNet/gpu/bridge_nmf.hpp:77-99, resolved and called at :407-497 by bridge_nmf_cv_sparse): 51 pointer
 * arguments, W (k x m) and H (k x n) are initialised by the caller, d = 1.  Implemented: MSE loss and loss_type 4..8 (GP, NB,
 * Gamma, inverse Gaussian, Tweedie: per-column weighted Grams, nmf/fit_cv.hpp:446-456, 670-689; dispersion settings = the
 * reference's config defaults, the boundary has no slot for them -- rcppml_gpu_nmf_cv_irls_ex below carries them), CD and
 * Cholesky+clip, L1 / L2, both mask_zeros settings, k <= 128; anything else sets out_status = -1 (CPU fallback).  The held-out set is
 * the speckled mask of rcppml_hip_solve_cv with cv_seed, or seed when cv_seed = 0 (core/config.hpp:415-418).  Early
 * stopping: cv_patience = NMF_PATIENCE = 5 (not transmitted by the bridge; RCPPML_GPU_CV_PATIENCE overrides).  On
 * return H carries d and d is returned too (nmf/fit_cv.hpp:1636-1647). */
#define RCPPML_NMF_CV_ARGS                                                                         \
    const int *col_ptr, const int *row_idx, const double *values, int *m, int *n, int *nnz, int *k, \
        double *W, double *H, double *d, int *max_iter, double *tol, double *L1_H, double *L1_W,    \
        double *L2_H, double *L2_W, int *cd_maxit, int *verbose,                                    \
        int *seed_only_used_for_cv_seed_fallback, double *holdout_frac, int *cv_seed,               \
        int *mask_zeros, int *nonneg_W, int *nonneg_H, int *norm_type, int *loss_type,              \
        double *huber_delta, int *irls_max_iter, double *irls_tol, const int *graph_W_p,            \
        const int *graph_W_i, const double *graph_W_x, int *graph_W_dim, int *graph_W_nnz,          \
        double *graph_W_lambda, const int *graph_H_p, const int *graph_H_i,                         \
        const double *graph_H_x, int *graph_H_dim, int *graph_H_nnz, double *graph_H_lambda,        \
        int *projective, int *symmetric, int *solver_mode, int *out_iter, int *out_converged,       \
        double *out_train_loss, double *out_test_loss, double *out_best_test, int *out_best_iter,   \
        int *out_status
RCPPML_GPU_API void rcppml_gpu_nmf_cv_unified_float(RCPPML_NMF_CV_ARGS);
RCPPML_GPU_API void rcppml_gpu_nmf_cv_unified_double(RCPPML_NMF_CV_ARGS);
/* Build-defined: + sort flag, precision (RCPPML_F32/F64), cv_patience, train / test loss histories (may be NULL). */
RCPPML_GPU_API void rcppml_gpu_nmf_cv_ex(RCPPML_NMF_CV_ARGS, int* sort_model, int* precision, int* cv_patience,
                                         double* train_history, double* test_history);
/* CV with the IRLS losses (loss_type 4 GP, 5 NB, 6 Gamma, 7 inverse Gaussian, 8 Tweedie; reference nmf/fit_cv.hpp:446-456, :670-689,
 * :866-961, :1377-1443, nmf/cv_detail.hpp:101-292) runs through ALL the CV entries above.  The reference boundary carries loss_type,
 * irls_max_iter and irls_tol only, so its entries take the reference's config defaults for the rest (per-row GP dispersion with
 * theta_init 0.1 / theta_max 5, Tweedie power 1.5, no robust modifier: core/config.hpp:151-172, math/loss.hpp:109-115); this
 * build-defined entry passes them, and returns GP's theta (out_theta: m doubles, may be NULL). */
RCPPML_GPU_API void rcppml_gpu_nmf_cv_irls_ex(RCPPML_NMF_CV_ARGS, int* sort_model, int* precision, int* cv_patience,
                                              double* train_history, double* test_history, int* dispersion_mode,
                                              double* gp_theta_init, double* gp_theta_max, double* tweedie_power,
                                              double* robust_delta, double* out_theta);

/* Build-defined: rcppml_gpu_nmf_cv_irls_ex + a user mask -- pattern CSC of the m x n mask (mask_p: n + 1, mask_i: *mask_nnz rows, ascending
 * inside a column; NULL / 0 = no mask).  The reference's nmf_fit_cv honours NMFConfig::mask (nmf/fit_cv.hpp:327-331, :491-501, :779-790;
 * nmf/cv_detail.hpp:433-505): a masked entry is in no training sum, its row joins the rows taken out of the column's Gram, and both
 * losses skip it -- computed explicitly per element for MSE too (:1377-1443).  Its CV boundary has no slot for the mask
 * (gpu/bridge_nmf.hpp:77-99), hence this entry. */
RCPPML_GPU_API void rcppml_gpu_nmf_cv_masked_ex(RCPPML_NMF_CV_ARGS, int* sort_model, int* precision, int* cv_patience,
                                                double* train_history, double* test_history, int* dispersion_mode,
                                                double* gp_theta_init, double* gp_theta_max, double* tweedie_power,
                                                double* robust_delta, double* out_theta, const int* mask_p, const int* mask_i,
                                                int* mask_nnz);

/* Dense-input NMF.  Replaces reference `rcppml_gpu_nmf_dense_unified_float` (resolved by
 * inst/include/FactorNet/gpu/bridge_nmf.hpp:544-545; 50 pointers, typedef :101-126): A_data is the m x n column-major
 * matrix as doubles on the host; W (k x m), H (k x n), d in/out as in the sparse entry.  Semantics: the reference's
 * STANDARD path (separate RHS -> features -> nnls_batch / cholesky_clip_batch, nmf/fit_cpu.hpp:540-631, :774-881) --
 * zero start at iteration 0, residual-corrected warm start afterwards -- MSE loss; L1, L2, L21, angular, bounds,
 * projective, symmetric.  loss_type 4..8 and robust_delta > 0 (round 5): the dense IRLS half-updates
 * (nnls_batch_irls_dense, nmf/fit_cpu.hpp:607-614, :855-863 -- EVERY entry weighted, zeros included, each batch from zero),
 * the dense branches of the dispersion updates and explicit_loss_dense; CD solver, k <= 128, m * n < 2^31, no L21 / angular /
 * projective / symmetric; gamma_phi_* take the config defaults (no slot); out_theta holds max(m, n) doubles
 * (gpu/bridge_nmf.hpp:622), *out_theta_len = m, or n under dispersion mode 3.  `_double` is build-defined (fp64 compute). */
#define RCPPML_NMF_DENSE_ARGS                                                                       \
    const double* A_data, int* m, int* n, int* k, double* W, double* H, double* d, int* max_iter,   \
        double* tol, double* L1_H, double* L1_W, double* L2_H, double* L2_W, double* L21_H,         \
        double* L21_W, double* ortho_H, double* ortho_W, double* ub_H, double* ub_W, int* cd_maxit, \
        int* verbose, int* seed, int* loss_every, int* patience, int* nonneg_W, int* nonneg_H,      \
        int* loss_type, double* huber_delta, int* irls_max_iter, double* irls_tol, int* norm_type,  \
        int* gp_dispersion_mode, double* gp_theta_init, double* gp_theta_max, double* gp_theta_min, \
        double* nb_size_init, double* nb_size_max, double* nb_size_min, double* robust_delta,       \
        double* tweedie_power, int* projective, int* symmetric, int* solver_mode, double* out_theta,\
        int* out_theta_len, int* out_iter, int* out_converged, double* out_loss, int* out_status,   \
        double* out_tol
RCPPML_GPU_API void rcppml_gpu_nmf_dense_unified_float(RCPPML_NMF_DENSE_ARGS);
RCPPML_GPU_API void rcppml_gpu_nmf_dense_unified_double(RCPPML_NMF_DENSE_ARGS);

RCPPML_GPU_API void rcppml_gpu_nmf_unified_float(RCPPML_NMF_UNIFIED_ARGS);
RCPPML_GPU_API void rcppml_gpu_nmf_unified_double(RCPPML_NMF_UNIFIED_ARGS);

/* Build-defined extras (not in the reference ABI).  Same 73 arguments plus what the
 * reference bridge does not transmit (gpu/bridge_nmf.hpp:180-346 drops config.mask, cd_tol,
 * sort_model): an explicit mask in CSC (nonzero = masked; nmf/masked_nnls.hpp), cd_tol, sort flag,
 * compute precision (0 = fp32, 1 = fp64) and an optional per-iteration loss history buffer
 * (length >= *max_iter, may be NULL).
 * out_theta CAPACITY: in this entry and in rcppml_gpu_nmf_target, *out_theta_len is read ON INPUT as the number of doubles
 * out_theta can hold (<= 0: m, the reference bridge's buffer, gpu/bridge_nmf.hpp:284); on return it is the number written.
 * dispersion mode 3 (per_col) writes n values and is refused with *out_status = -1 when the capacity is smaller than n. */
RCPPML_GPU_API void rcppml_gpu_nmf_ex(RCPPML_NMF_UNIFIED_ARGS, const int* mask_p, const int* mask_i,
                                      int* mask_nnz, double* cd_tol, int* sort_model,
                                      int* precision, double* loss_history);

/* rcppml_gpu_nmf_ex plus TARGET regularisation (nmf/variant_helpers.hpp:107-146, not carried by the
 * 73-pointer ABI): target_H (k x n) / target_W (k x m) column-major with k leading, NULL = none, and their lambdas.
 *   lambda > 0  enrichment: G.diagonal() += lambda, B += lambda * target;
 *   lambda < 0  PROJ_ADV:   G -= |lambda| * (trace G / trace TG) * TG with TG = target target^T / ncols (nmf/fit.hpp:259-271),
 *               then the eigenvalues of G below 1e-8 are raised to 1e-8 (B untouched).
 * As in the reference (fit_cpu.hpp:430-433) a fit with a target runs the STANDARD path: B materialised, features on
 * (G, B), nnls_batch from zero at iteration 0 and residual-corrected afterwards.  Plain MSE fits only. */
RCPPML_GPU_API void rcppml_gpu_nmf_target(RCPPML_NMF_UNIFIED_ARGS, const int* mask_p, const int* mask_i,
                                          int* mask_nnz, double* cd_tol, int* sort_model, int* precision,
                                          double* loss_history, const double* target_H, double* target_lambda_H,
                                          const double* target_W, double* target_lambda_W);

/* fp64 projection h = NNLS(w^T w, w^T A): GPU entry for R nnls()/predict(), which have no GPU
 * hook in the reference (src/RcppFunctions_utils.cpp:313-366 c_nnls, :23-52 Rcpp_predict).
 * w_T: k x m, A: m x n CSC (host pointers), h: k x n in/out (warm start if *warm != 0). */
RCPPML_GPU_API void rcppml_gpu_nnls_double(const int* col_ptr, const int* row_idx,
                                           const double* values, int* m, int* n, int* nnz, int* k,
                                           const double* w_T, double* h, int* cd_maxit,
                                           double* cd_tol, double* L1, double* L2, double* ub,
                                           int* nonneg, int* warm, int* out_status);

/* fp64 evaluate(): mean squared error of W diag(d) H against A over all m*n entries or (mask_zeros)
 * over nonzeros only, WITHOUT densifying W H (reference src/RcppFunctions_utils.cpp:95-163 builds
 * the dense m x n product).  W_T: k x m. */
RCPPML_GPU_API void rcppml_gpu_evaluate_mse_double(const int* col_ptr, const int* row_idx,
                                                   const double* values, int* m, int* n, int* nnz,
                                                   int* k, const double* W_T, const double* d,
                                                   const double* H, int* mask_zeros,
                                                   double* out_loss, int* out_status);

/* Per-phase profile of the batch-CD ALS iteration -- the reference's `rcppml_gpu_nmf_profile_double`
 * (src/gpu_bridge_utils.cu:48-57: the same fifteen pointers, in order; :14-36 the slots).  fp64; W_T and H from
 * SplitMix64(*seed) (one stream, W_T first: initialize_factors), d = 1; one untimed warm-up iteration, then up to *max_iter
 * iterations with a HIP-event pair around every phase and `|prev - loss| / (|prev| + 1e-15) < *tol` from the second timed
 * iteration on; every solve runs *cd_maxit sweeps (no tolerance stop), cold solves start from max(B, 0).
 * out_phase_ms_total[11] / out_phase_ms_per_iter[11] (total / *out_n_iters), slots:
 *   0 gram_H   1 rhs_H   2 nnls_H   3 norm_H   4 gram_W   5 rhs_W by the plan-free gather kernel (the reference's slot holds
 *   its atomicAdd baseline; it is not used by the solve here either)   6 rhs_W planned   7 nnls_W   8 norm_W   9 loss   10 iteration.
 * *out_status 0 / -1 (rcppml_gpu_last_error). */
RCPPML_GPU_API void rcppml_gpu_nmf_profile_double(const int* col_ptr, const int* row_idx, const double* values,
                                                  int* m, int* n, int* nnz, int* k, int* max_iter, double* tol,
                                                  int* cd_maxit, int* seed, double* out_phase_ms_total,
                                                  double* out_phase_ms_per_iter, int* out_n_iters, int* out_status);

/* Last error text of the calling thread ("" if none). */
RCPPML_GPU_API const char* rcppml_gpu_last_error(void);

/* --------------------------------------------------------------------------
 * (2) Device-level ops.  dtype: 0 = fp32, 1 = fp64.  All pointers are DEVICE pointers unless
 * named host_*.  Every op is enqueued on the context's stream and returns 0 on success
 * (non-zero: see rcppml_gpu_last_error).  No op synchronises the stream.
 * ------------------------------------------------------------------------*/
typedef struct rcppml_hip_ctx rcppml_hip_ctx;

enum { RCPPML_F32 = 0, RCPPML_F64 = 1 };
/* CD kernel variants (rcppml_hip_solve_cd `variant`) */
enum { RCPPML_CD_AUTO = 0 /* = GROUP unless RCPPML_GPU_CD_VARIANT says otherwise */,
       RCPPML_CD_LANE = 1 /* one lane per column, residual+iterate in registers, G through the scalar cache (SGPRs) */,
       RCPPML_CD_WAVE = 2 /* one wavefront per column, active-coordinate ballot skipping, G in LDS */,
       RCPPML_CD_GROUP = 5 /* 1, 2 or 4 adjacent lanes per column (DPP broadcasts), G from LDS */,
       RCPPML_CD_MFMA16 = 7 /* k <= 64, fp32 or fp64: 16 columns per wave, four coordinates per v_mfma_*_16x16x4 (the fp64 default) */,
       RCPPML_CD_MFMA = 6 /* fp32, k <= 128: residual tiles in MFMA accumulators, two coordinates per v_mfma_f32_32x32x2_f32 */,
       RCPPML_CD_LMF = 8 /* fp32, k <= 64, non-negativity only: lane = column, rank-1 updates as v_mfma_f32_4x4x1_16B blocks, persistent
                            waves with column refill (the fp32 default for NMF half-updates) */ };

/* stream: a hipStream_t (NULL = the device's null stream).  The context owns scratch memory only. */
RCPPML_GPU_API int rcppml_hip_ctx_create(rcppml_hip_ctx** out, int device, void* stream);
RCPPML_GPU_API void rcppml_hip_ctx_destroy(rcppml_hip_ctx* ctx);
RCPPML_GPU_API int rcppml_hip_ctx_sync(rcppml_hip_ctx* ctx);
/* Work counters since creation / the last reset (synchronises the stream): out4[0] = column-sweeps executed by the
 * coordinate-descent kernels (sum over solved columns of the sweeps cd_nnls_col_fixed ran, nnls_batch.hpp:127-131;
 * x 2 k_pad^2 = the flops of the residual updates), out4[1] = columns solved, out4[2] = slot-sweeps the persistent LMF
 * kernel executed (column slots x sweeps of their wave: idle slots and warm-start correction sweeps included, so
 * 1 - out4[0] / out4[2] is its idle fraction), out4[3] = coordinate steps of that kernel in which NO column of the wave
 * moved (only with RCPPML_OPT_CD_COUNT_NOOP; a step is one coordinate of one wave-sweep).  Counted by the GROUP, MFMA and
 * LMF kernels (what RCPPML_CD_AUTO dispatches to). */
RCPPML_GPU_API int rcppml_hip_ctx_stats(rcppml_hip_ctx* ctx, int reset, unsigned long long* out4);
/* IRLS work counters (only while RCPPML_OPT_CD_COUNT_NOOP is set; two atomics per column): out2[0] = IRLS passes summed over
 * the columns rcppml_hip_solve_irls solved (nnls_batch_irls.hpp:480-560: each pass rebuilds the weighted Gram and solves),
 * out2[1] = the same weighted by the column's nonzeros = rank-1 updates f f^T of the weighted Grams (x 2 k_pad^2 = flops). */
RCPPML_GPU_API int rcppml_hip_ctx_irls_stats(rcppml_hip_ctx* ctx, int reset, unsigned long long* out2);
/* out1[0] = CD sweeps executed inside the IRLS half-updates (per column and pass, until the column's own fixed point); counted with the above */
RCPPML_GPU_API int rcppml_hip_ctx_irls_sweep_stats(rcppml_hip_ctx* ctx, int reset, unsigned long long* out1);
/* Per-(column, coordinate) step counters of the lane = column CD kernel (only while RCPPML_OPT_CD_COUNT_NOOP is set): out2[0] =
 * coordinate steps whose update is exactly 0 -- the steps the reference skips with `continue` (primitives/cpu/nnls_batch.hpp:
 * 102,106,109) and a dense sweep still executes --, out2[1] = all coordinate steps of live columns (warm-start correction
 * sweeps and padded coordinates excluded).  bench.py: roofline.useful_frac = frac x (1 - out2[0] / out2[1]). */
RCPPML_GPU_API int rcppml_hip_ctx_cd_step_stats(rcppml_hip_ctx* ctx, int reset, unsigned long long* out2);
/* Tuning / diagnostic switches of a context (0 = default behaviour for all of them). */
enum { RCPPML_OPT_CD_COUNT_NOOP = 1 /* LMF kernel counts all-zero coordinate steps into stats[3], IRLS kernels count passes (slower) */,
       RCPPML_OPT_CD_LMF_LANE_GROUPS = 2 /* 1, 2 or 4 lane groups per column instead of the size heuristic */,
       RCPPML_OPT_CD_LMF_WAVES_PER_SIMD = 3 /* resident persistent waves per SIMD instead of the heuristic */,
       RCPPML_OPT_CD_NO_LMF = 4 /* RCPPML_CD_AUTO falls back to the 32- / 16-column MFMA kernels */,
       RCPPML_OPT_IRLS_COLUMNS_PER_WAVE = 5, /* fp32 k <= 32 IRLS half-update: 0 = by the number of columns, 1 or 4 columns per wavefront */
       RCPPML_OPT_SMALL_GIVE_UP = 6 /* test switch: the one-kernel fit's first barrier gives up at once (its abort flag is preset), as if
                                       its workgroups had not all arrived on one XCD -- exercises the caller's restart on the multi-launch ops */ };
RCPPML_GPU_API int rcppml_hip_ctx_set_option(rcppml_hip_ctx* ctx, int option, int value);

/* One-time setup of a fit, on the device.
 * rcppml_hip_transpose_csc: CSC of A^T from the CSC of A (rows x cols; all pointers device memory; t_col_ptr rows+1 ints,
 *   t_row_idx / t_values nnz entries; values / t_values may be NULL for a pattern-only matrix such as a mask).  Replaces the
 *   reference's host-side `At = A.transpose()` (nmf/fit_cpu.hpp:251-253): a stable sort by row index, so the column
 *   indices inside each row of A come out ascending, as Eigen produces them.  Synchronises the stream.
 * rcppml_hip_cast: elementwise precision cast between device buffers (the plugin boundary hands over doubles,
 *   gpu/bridge_nmf.hpp:310-342; the fp32 entry computes in fp32). */
RCPPML_GPU_API int rcppml_hip_transpose_csc(rcppml_hip_ctx* ctx, int dtype, int rows, int cols, const int* col_ptr,
                                            const int* row_idx, const void* values, int* t_col_ptr, int* t_row_idx,
                                            void* t_values);
/* The same transpose in two asynchronous steps (no stream synchronisation when the context serves temporaries from a per-fit
 * arena): _sort needs the index arrays only -- it makes the row pointers of A^T and the nonzero positions sorted by row
 * (sorted_pos: nnz ints; a stable counting sort over column chunks, own kernels) -- so the plugin runs it while the VALUES are
 * still crossing PCIe; _gather then fills t_row_idx / t_values (either
 * may be NULL: indices before the values have arrived, values after). */
RCPPML_GPU_API int rcppml_hip_transpose_csc_sort(rcppml_hip_ctx* ctx, int rows, int cols, int64_t nnz, const int* col_ptr,
                                                 const int* row_idx, int* t_col_ptr, int* sorted_pos);
RCPPML_GPU_API int rcppml_hip_transpose_csc_gather(rcppml_hip_ctx* ctx, int dtype, int cols, int64_t nnz, const int* col_ptr,
                                                   const int* sorted_pos, const void* values, int* t_row_idx, void* t_values);
RCPPML_GPU_API int rcppml_hip_cast(rcppml_hip_ctx* ctx, int dtype_src, const void* src, int dtype_dst, void* dst, int64_t n);

/* G = F F^T (+ eps on the diagonal, then + l2) -- reference primitives/cpu/gram.hpp:37-67 and
 * nmf/fit_cpu.hpp:506,738.  F: k x r.  MFMA kernel (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64),
 * split over r with a deterministic two-pass reduction. */
RCPPML_GPU_API int rcppml_hip_gram(rcppml_hip_ctx* ctx, int dtype, const void* F, int k, int64_t r,
                                   double eps, double l2, void* G);

/* B(:,j) = sum_{i in nz(j)} A(i,j) F(:,i) -- reference primitives/cpu/rhs.hpp:52-70 and the RHS
 * step of primitives/cpu/fused_nnls.hpp:109-114.  One wavefront per output column. */
RCPPML_GPU_API int rcppml_hip_rhs(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr,
                                  const int* row_idx, const void* values, int64_t ncols,
                                  const void* F, int k, void* B);

/* Planned form of the same product for large inputs (build-defined; the reference has one CPU loop, rhs.hpp:52-70).
 * A plan is a device copy of ONE CSC matrix for one rank k and precision, laid out for a kernel that stages the rows of F through
 * LDS and keeps the output columns in registers.  Two plan kinds (rcppml_hip_rhs_plan_kind says which the planner took):
 *   window plan (kind 1, kernels_rhs_win.hip.h -- the default): a ring of four 32 KiB row tiles, nonzeros scheduled over a sliding
 *     window of three tiles at a fractional slot rate; what the window cannot place (the "overflow", a few per cent) is added by
 *     the finishing pass that also sums the row partitions; no tail launch.  Declined (-> slab plan) for slot rates above 6 per
 *     phase, rows not sorted inside a column, hypersparse inputs, or more than 25 % overflow.
 *   slab plan (kind 0, kernels_rhs_tiled.hip.h -- the r2/r3 form, the fallback): 64 KiB tiles, S slots per (column, tile), the
 *     rest spilled to a gather kernel (refused above 35 % spill); columns that would only part-fill a last round of workgroups go
 *     to the gather kernel.
 *   plan_create: nrows = rows of the sparse matrix (= number of k-vectors in F).  partitions: 0 = automatic.
 *     slots: 0 = window plan, then slab plan, then no plan; 1 = slab plan with S chosen from the data; 2..8 = slab plan with S
 *     slots; >= 100 = window plan at (slots - 100) / 4 slots per column and phase (e.g. 107 = 1.75).
 *     *out_plan stays NULL (return 0) when no plan kind is eligible (k * sizeof(T) not 256 or 512 bytes -- or 1024 in fp64 --,
 *     unsorted rows, too irregular): call rcppml_hip_rhs then.
 *   rhs_planned: B = F * A(:, j) for all columns, same numbers as rcppml_hip_rhs up to summation order; deterministic.
 *     The plan keeps the col_ptr / row_idx / values POINTERS: the CSC must outlive the plan.
 *   plan_info (11 doubles): {P, waves per workgroup, rounds per wave, slots (slab: S; window: clo + nhi / 4 = the slot rate),
 *     workgroups per partition, tiles, slot count, spilled / overflow nonzeros, slot fill fraction, slot stream bytes, columns
 *     handled by the tiled kernel}.
 *   plan_set_values works only on plans made by plan_create_indices (it needs their per-nonzero destination table). */
typedef struct rcppml_rhs_plan rcppml_rhs_plan;
RCPPML_GPU_API int rcppml_hip_rhs_plan_create(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr, const int* row_idx,
                                              const void* values, int64_t ncols, int64_t nrows, int k, int partitions,
                                              int slots, rcppml_rhs_plan** out_plan);
/* The same plan in two steps: _create_indices does everything that needs only the index arrays (the schedule, the offsets, the
 * overflow lists, and for every nonzero where its value will go), _set_values scatters the values with one coalesced pass and
 * may be called again when the values of the same pattern change.  The plugin builds both plans while the values are still
 * crossing PCIe.  *out_plan = NULL when the window planner declines the input (use rcppml_hip_rhs_plan_create then). */
/* 1 = window plan, 0 = slab plan (see above), -1 = NULL. */
RCPPML_GPU_API int rcppml_hip_rhs_plan_kind(const rcppml_rhs_plan* plan);
RCPPML_GPU_API int rcppml_hip_rhs_plan_create_indices(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr, const int* row_idx, int64_t ncols,
                                                      int64_t nrows, int k, int partitions, int slots, rcppml_rhs_plan** out_plan);
RCPPML_GPU_API int rcppml_hip_rhs_plan_set_values(rcppml_hip_ctx* ctx, rcppml_rhs_plan* plan, const void* values);
RCPPML_GPU_API void rcppml_hip_rhs_plan_destroy(rcppml_rhs_plan* plan);
RCPPML_GPU_API int rcppml_hip_rhs_plan_info(const rcppml_rhs_plan* plan, double* out11);
RCPPML_GPU_API int rcppml_hip_rhs_planned(rcppml_hip_ctx* ctx, const rcppml_rhs_plan* plan, const void* F, void* B);

/* Per-column CD NNLS -- reference primitives/cpu/nnls_batch.hpp:70-132 (cd_nnls_col_fixed) with the
 * prologues of fused_nnls.hpp:116-123 / nnls_batch.hpp:167-174:
 *   b = B(:,j); if (l1_pre>0) b -= l1_pre; x = zero_init ? 0 : X(:,j); if (warm) b -= G x;
 *   CD(G, b, x, l1_cd, l2_cd, nonneg, maxit, ub_cd, tol); if (ub_post>0) x = min(x, ub_post).
 * B is NOT modified (the residual lives in registers).  G: k x k.
 * sweeps_out (device, ncols ints, may be NULL): sweeps executed per column = the value cd_nnls_col_fixed
 * returns (nnls_batch.hpp:127-131).
 * col_order (device, ncols ints, may be NULL): work order -- slot s of the launch solves column col_order[s]
 * (see rcppml_hip_order_columns).  Columns are independent: any permutation gives identical results. */
RCPPML_GPU_API int rcppml_hip_solve_cd(rcppml_hip_ctx* ctx, int dtype, const void* G, const void* B,
                                       void* X, int k, int64_t ncols, double l1_pre, int warm,
                                       int zero_init, double l1_cd, double l2_cd, int nonneg,
                                       int maxit, double tol, double ub_cd, double ub_post,
                                       int variant, int* sweeps_out, const int* col_order);

/* order = columns sorted by DESCENDING sweeps (counting sort on the device).  Feeding the sweep counts of the
 * previous ALS iteration groups columns that converge together into the same wavefront. */
RCPPML_GPU_API int rcppml_hip_order_columns(rcppml_hip_ctx* ctx, const int* sweeps, int64_t ncols, int* order);

/* Cholesky solve + clip -- reference primitives/cpu/fused_nnls.hpp:185-219:
 *   L = chol(G) once; x = L^-T L^-1 (B(:,j) - l1_pre); clip >= 0 (nonneg); clip <= ub_post. */
RCPPML_GPU_API int rcppml_hip_solve_chol(rcppml_hip_ctx* ctx, int dtype, const void* G,
                                         const void* B, void* X, int k, int64_t ncols,
                                         double l1_pre, int nonneg, double ub_post);

/* Row norms of X (k x c): out[i] = sum_j |X_ij| (norm_type 0), sum_j X_ij^2 (norm_type 1) or sum_j X_ij (norm_type 3);
 * no epsilon, no sqrt -- the distributed harness all-reduces these partial sums. */
RCPPML_GPU_API int rcppml_hip_row_norms(rcppml_hip_ctx* ctx, int dtype, const void* X, int k,
                                        int64_t ncols, int norm_type, void* out);
/* d = (norm_type==1 ? sqrt(s) : s) + 1e-15; X(i,:) /= d_i -- reference
 * nmf/variant_helpers.hpp:286-305.  `sums` as produced by rcppml_hip_row_norms. */
RCPPML_GPU_API int rcppml_hip_apply_scaling(rcppml_hip_ctx* ctx, int dtype, void* X, int k,
                                            int64_t ncols, int norm_type, const void* sums, void* d);

/* extract_scaling (rcppml_hip_row_norms + rcppml_hip_apply_scaling: the same row sums, d and X, bit for bit) and -- when `sweeps` /
 * `order` are not NULL -- rcppml_hip_order_columns(sweeps, ncols, order) for the NEXT solve of this side inside the same three
 * launches instead of five (independent kernels share a launch: the histogram beside the row sums, the scatter beside the scaling).  Reference:
 * nmf/variant_helpers.hpp:286-305 (the work order has no reference counterpart).  `sums`: k values of scratch (the raw row sums). */
RCPPML_GPU_API int rcppml_hip_scale_order(rcppml_hip_ctx* ctx, int dtype, void* X, int k, int64_t ncols, int norm_type,
                                          void* sums, void* d, const int* sweeps, int* order);

/* sum of squares of a length-len vector in fp64 -> out[0] (double, device) -- trace_AtA,
 * reference primitives/primitives.hpp:100-115. */
RCPPML_GPU_API int rcppml_hip_sumsq(rcppml_hip_ctx* ctx, int dtype, const void* x, int64_t len,
                                    double* out);

/* MSE loss by the Gram trick -- reference nmf/fit_cpu.hpp:1710-1753:
 *   cross = sum_{l,i} d_i W_T(i,l) B_w(i,l); recon = sum_ij d_i d_j G_wt(i,j) G_saved(i,j);
 *   out[0] = trAtA[0] - 2 cross + recon; out[1] = cross; out[2] = recon   (double, device). */
RCPPML_GPU_API int rcppml_hip_loss_mse(rcppml_hip_ctx* ctx, int dtype, const double* trAtA,
                                       const void* d, const void* W_T, const void* B_w, int k,
                                       int64_t m, const void* G_wt, const void* G_saved,
                                       double* out);

/* G_wt = W_T W_T^T + eps I (rcppml_hip_gram with l2 = 0) and then rcppml_hip_loss_mse with it, in three launches instead of four (the
 * cross-term partials share the launch of the Gram's final sum);
 * the same G_wt and out[0..2] bit for bit -- reference nmf/fit_cpu.hpp:1729-1753. */
RCPPML_GPU_API int rcppml_hip_gram_loss_mse(rcppml_hip_ctx* ctx, int dtype, const void* W_T, int k, int64_t m, double eps,
                                            const double* trAtA, const void* d, const void* B_w, const void* G_saved,
                                            void* G_wt, double* out);

/* The tail of a half-update in one call.  rcppml_hip_tail_scale_gram = rcppml_hip_scale_order(X, ...) then rcppml_hip_gram(X, eps, l2)
 * -> G (the H side: extract_scaling of H, then its Gram for the W update; nmf/fit_cpu.hpp:645, :715-722);
 * rcppml_hip_tail_scale_gram_loss = rcppml_hip_scale_order(W_T, ...) then rcppml_hip_gram_loss_mse (the W side: extract_scaling of W_T,
 * its Gram and the loss; :893, :1729-1753).  Same results as the separate calls bit for bit.  fp32 with k = 64 runs the scaling pass
 * INSIDE the Gram's partial-tile kernel (the lane that loads an element divides, stores and multiplies it): four and five launches
 * instead of seven and nine; every other shape takes the two calls above. */
RCPPML_GPU_API int rcppml_hip_tail_scale_gram(rcppml_hip_ctx* ctx, int dtype, void* X, int k, int64_t ncols, int norm_type, void* sums,
                                              void* d, const int* sweeps, int* order, double eps, double l2, void* G);
RCPPML_GPU_API int rcppml_hip_tail_scale_gram_loss(rcppml_hip_ctx* ctx, int dtype, void* W_T, int k, int64_t m, int norm_type, void* sums,
                                                   void* d, const int* sweeps, int* order, double eps, const double* trAtA,
                                                   const void* B_w, const void* G_saved, void* G_wt, double* out);

/* The WHOLE plain sparse MSE fit of a SMALL matrix as one persistent kernel on one XCD (kernels_small.hip.h; round 6): nmf_fit<CPU>
 * (nmf/fit_cpu.hpp:444-1855) with fused right-hand side + solve per column (primitives/cpu/fused_nnls.hpp:70-134 CD, :185-219
 * Cholesky + clip), L1 / L2 / upper bounds / non-negativity, row scaling (nmf/variant_helpers.hpp:286-305), Gram-trick loss every
 * iteration (fit_cpu.hpp:1729-1753) and the convergence rule (:1769-1809) evaluated on the device -- no launch per phase, no host
 * round trip per iteration.  The iteration's grid-wide dependencies are barriers inside the kernel among 32 workgroups that all
 * sit on XCD 0 (one L2: a relaxed L2 atomic + an L1 invalidate, 0.8 us; profiles/r06_grid_barrier.txt).
 * rcppml_hip_als_small_eligible: where the kernel PAYS and the plugin takes it -- k <= 16, m + n <= 3072, nnz <= 2^17 (measured:
 * profiles/r06_small_threshold.txt; hawaiibirds is, movielens at k = 32 is not); rcppml_hip_als_small_fit itself accepts k <= 32, nnz <= 2^20, m + n <= 65536.
 * CSC(A) and CSC(A^T) with sorted rows; W (k x m), H (k x n) in / out, d (k) out; trAtA = sum a^2 (device, rcppml_hip_sumsq);
 * iter0: iterations already run by earlier calls on these factors (0 = a fit from its start: iteration 0 then solves without the
 * warm-start correction; > 0 continues a fit -- bench.py's warm-up / timed split); loss_history: max_iter doubles (device, may be NULL); result8 (device): [0] iterations [1] converged [2] train loss [3] last relative
 * change [4] 1 = done, anything else = the kernel gave up at a barrier (its workgroups did not all land on one XCD): W / H are then
 * partly overwritten and the caller must restart on the multi-launch ops; [5..7] workgroup 0's 100 MHz clock ticks inside the launch:
 * fused half-updates, waiting at barriers, whole kernel (bench.py --config c1 `inside_the_kernel`). */
RCPPML_GPU_API int rcppml_hip_als_small_eligible(int m, int n, int64_t nnz, int k);
RCPPML_GPU_API int rcppml_hip_als_small_fit(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr, const int* row_idx, const void* values,
                                            const int* t_col_ptr, const int* t_row_idx, const void* t_values, int m, int n, int64_t nnz,
                                            int k, void* W, void* H, void* d, const double* trAtA, double L1_H, double L1_W, double L2_H,
                                            double L2_W, double ub_H, double ub_W, int nonneg_H, int nonneg_W, int norm_type,
                                            int solver_mode, int cd_maxit, double cd_tol, int max_iter, double tol, int patience,
                                            int iter0, double* loss_history, double* result8);

/* Explicit-mask per-column NNLS -- reference nmf/masked_nnls.hpp:96-154 / 177-242.
 * A and mask share shape (rows x ncols, CSC; mask nonzero = masked; mask values not needed). */
RCPPML_GPU_API int rcppml_hip_solve_masked(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr,
                                           const int* row_idx, const void* values,
                                           const int* mask_p, const int* mask_i, int64_t ncols,
                                           const void* F, const void* G_full, void* X, int k,
                                           double l1, double l2, int nonneg, int cd_maxit,
                                           double cd_tol, int solver_mode, int warm);
/* Over unmasked NONZEROS, with p = sum_f d_f W_T(f,i) H(f,j):  out[0] = sum (a - p)^2 (reference
 * nmf/masked_nnls.hpp:250-282), out[1] = sum p^2.  mask_p may be NULL (no mask: all nonzeros).
 * out: 2 doubles, device. */
RCPPML_GPU_API int rcppml_hip_loss_nonzeros(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr,
                                            const int* row_idx, const void* values,
                                            const int* mask_p, const int* mask_i, int64_t ncols,
                                            const void* W_T, const void* d, const void* H, int k,
                                            double* out);
/* The same pass with the per-element term of the configured loss: out[0] = sum over the unmasked nonzeros of
 * compute_loss(a, p, loss) (math/loss.hpp:512-536) -- the loss of a fit that has BOTH an explicit mask and a distribution loss
 * (fit_cpu.hpp:1685-1690 -> nmf/masked_nnls.hpp:250-282, :277).  As in the reference the term is evaluated with theta = 0 (masked_loss
 * passes no dispersion) and without the robust modifier.  loss_type 0 (MSE) or 4..8; out: 2 doubles, device (out[1] = sum p^2). */
RCPPML_GPU_API int rcppml_hip_loss_masked(rcppml_hip_ctx* ctx, int dtype, int loss_type, double tweedie_power,
                                          const int* col_ptr, const int* row_idx, const void* values,
                                          const int* mask_p, const int* mask_i, int64_t ncols,
                                          const void* W_T, const void* d, const void* H, int k, double* out);

/* Cross-validation (reference nmf/fit_cv.hpp, nmf/cv_detail.hpp, nmf/speckled_cv.hpp).  The held-out
 * set is the lazy speckled mask  SplitMix64::hash(seed, i, j) < UINT64_MAX / floor(1 / holdout_fraction)  with
 * seed = (uint32) cv_seed (0 -> 12345) and (i, j) the coordinates in A (rng/rng.hpp:129-170); no mask matrix exists.
 * rcppml_hip_solve_cv: one CV half-update over the columns of the CSC handed in (A for H, A^T with transposed = 1 for
 *   W; nrows = its row count): b = sum over the column's TRAIN nonzeros, G_local = G - sum over its TEST rows f f^T
 *   (mask_zeros = 1: held-out nonzeros only; 0: every held-out row, zeros included), then Cholesky+clip (solver_mode 1,
 *   L1 subtracted from b) or CD (solver_mode 0: L1 inside, cd_maxit sweeps, no tolerance) started from the current X
 *   without a warm-start correction -- fit_cv.hpp:420-478,591-830.  k <= 128 (above 64: kernels_wide.hip.h).
 * rcppml_hip_cv_test_error: out2[0] = sum of squared errors, out2[1] = count over the held-out entries of A
 *   (prediction W diag(d) H; fit_cv.hpp:1444-1494). */
RCPPML_GPU_API int rcppml_hip_solve_cv(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr, const int* row_idx,
                                       const void* values, int64_t ncols, int nrows, const void* F, const void* G, void* X,
                                       int k, double holdout_fraction, unsigned long long cv_seed, int mask_zeros,
                                       int transposed, double l1, int nonneg, int cd_maxit, int solver_mode);
/* The user mask of a CV fit (see rcppml_gpu_nmf_cv_masked_ex): device pointers to the pattern CSC of the mask (rows x cols of A) and of
 * its transpose; they must stay valid until the mask is cleared (all four NULL).  While set, rcppml_hip_solve_cv,
 * rcppml_hip_solve_cv_irls and rcppml_hip_cv_irls_loss exclude its entries as the reference does (rcppml_hip_cv_test_error does not: with
 * a mask both losses come from rcppml_hip_cv_irls_loss, loss_type 0). */
RCPPML_GPU_API int rcppml_hip_ctx_set_cv_mask(rcppml_hip_ctx* ctx, const int* mask_p, const int* mask_i, const int* maskT_p,
                                              const int* maskT_i);
RCPPML_GPU_API int rcppml_hip_cv_test_error(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr, const int* row_idx,
                                            const void* values, int64_t ncols, int nrows, const void* W_T, const void* d,
                                            const void* H, int k, double holdout_fraction, unsigned long long cv_seed,
                                            int mask_zeros, double* out2);

/* Graph (Laplacian) regularisation -- reference features/graph_reg.hpp:38-50 at its fused-path place
 * (nmf/fit_cpu.hpp:508-509,741-742):  G += lambda * (X L) X^T  for the CURRENT factor X (k x ncols) and a sparse
 * ncols x ncols Laplacian L in CSC (device).  X L is formed by the SpMM kernel, the k x k product by a blocked reduction
 * with a fixed summation order.  k <= 128. */
RCPPML_GPU_API int rcppml_hip_apply_graph_reg(rcppml_hip_ctx* ctx, int dtype, void* G, const int* lap_p, const int* lap_i,
                                              const void* lap_x, const void* X, int k, int64_t ncols, double lambda);

/* Y = diag(d) X for a k x ncols factor (variant_helpers.hpp:265-272 apply_scaling): W diag(d) of the projective H
 * update  H = (diag(d) W_T) A  (nmf/fit_cpu.hpp:462-472, variant_helpers.hpp:308-325). */
RCPPML_GPU_API int rcppml_hip_mul_rows(rcppml_hip_ctx* ctx, int dtype, const void* X, int k, int64_t ncols, const void* d, void* Y);

/* Y = X + alpha * T elementwise (n entries) and G(i,i) += v: the two device-side pieces of target regularisation
 * (nmf/variant_helpers.hpp:107-111: G.diagonal() += lambda; B += lambda * target). */
RCPPML_GPU_API int rcppml_hip_axpy(rcppml_hip_ctx* ctx, int dtype, const void* X, const void* T, double alpha, int64_t n, void* Y);
RCPPML_GPU_API int rcppml_hip_add_diag(rcppml_hip_ctx* ctx, int dtype, void* G, int k, double v);
/* X = min(X, ub) over n entries (features/bounds.hpp apply_upper_bound; nmf/fit_cpu.hpp:636-637, 884-885). */
RCPPML_GPU_API int rcppml_hip_clip_upper(rcppml_hip_ctx* ctx, int dtype, void* X, int64_t n, double ub);

/* k x k feature layer, fused-path placement of nmf/fit_cpu.hpp:505-511,636-639 / :738-745,884-887:
 * rcppml_hip_apply_l21: G(i,i) += lambda / ||X.row(i)||_2 for rows with norm > 1e-10 (features/L21.hpp:38-51); X is the
 *   CURRENT factor (k x ncols), applied to the Gram before the solve.
 * rcppml_hip_angular_posthoc: X <- max(0, X - lambda * diag(norms) * offdiag(Xh Xh^T) Xh), Xh = rows of X scaled to unit
 *   norm (features/angular.hpp:67-103); applied after the solve and the upper bound, before scaling.  k <= 128. */
RCPPML_GPU_API int rcppml_hip_apply_l21(rcppml_hip_ctx* ctx, int dtype, void* G, const void* X, int k, int64_t ncols,
                                        double lambda);
RCPPML_GPU_API int rcppml_hip_angular_posthoc(rcppml_hip_ctx* ctx, int dtype, void* X, int k, int64_t ncols, double lambda);

/* NB (negative-binomial) IRLS half-update -- reference primitives/cpu/nnls_batch_irls.hpp:465-520,202-329 with the NB
 * weight of math/loss.hpp:248-256: X = 0; per column up to irls_max_iter passes of
 *   w_i = min(r/(mu(r+mu)), 1e6) at the column's nonzeros (mu = F(:,i).x, in fp64), G_w = G_base + F_nz diag(w-1) F_nz^T
 *   (+ l2 on the diagonal), b_w = F_nz (w o a), b = b_w - G_w x, CD(G_w, b, x, L1 inside, cd_maxit sweeps, no tolerance),
 *   stop when max |dx|/(|x_old|+1e-12) < irls_tol.
 * theta_row: NB size indexed by the nonzero's row (H side) or theta_col: indexed by the column (W side over A^T);
 * exactly one is non-NULL.  k <= 128 (above 64: one wave per column with two features per lane, kernels_wide.hip.h).
 * The tolerance-free sweeps end early at the iterate's floating-point fixed point (a whole sweep that changes no coordinate). */
RCPPML_GPU_API int rcppml_hip_solve_irls_nb(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr, const int* row_idx,
                                            const void* values, int64_t ncols, const void* F, const void* G_base,
                                            void* X, int k, double l1, double l2, int nonneg, int cd_maxit,
                                            int irls_max_iter, double irls_tol, const void* theta_row,
                                            const void* theta_col);
/* Generic forms: loss_type = the reference's LossType (math/loss.hpp:36-47), implemented: 5 = NB, 4 = GP, 6 = Gamma,
 * 7 = inverse Gaussian, 8 = Tweedie (loss_param = the variance power p; ignored otherwise).  The GP loss updates W and H
 * with the KL weight 1/max(mu, 1e-4) (nmf/fit_cpu.hpp:568-574: "GP strategy: use KL weights for W/H updates") and
 * evaluates the GP likelihood (math/loss.hpp:382-398) with theta_row (zeros for dispersion = "none": Poisson /
 * KL-divergence NMF); 6-8 use the power-variance weight min(1/mu^p, 1e6) (:270-278) and the deviance terms (:439-505).
 * theta pointers are read by NB only (pass NULL otherwise).  The plugin implements 4 and 6-8 for dispersion = "none".
 * robust_delta > 0 multiplies every weight by the Huber modifier of the Pearson residual (math/loss.hpp:294-303,
 * nnls_batch_irls.hpp:95-120) and makes the loss the Huber rho of that residual (:549-607); with it loss_type 0 (MSE,
 * distribution weight 1) is accepted too -- the reference routes robust MSE through the IRLS path. */
RCPPML_GPU_API int rcppml_hip_solve_irls(rcppml_hip_ctx* ctx, int dtype, int loss_type, const int* col_ptr,
                                         const int* row_idx, const void* values, int64_t ncols, const void* F,
                                         const void* G_base, void* X, int k, double l1, double l2, int nonneg,
                                         int cd_maxit, int irls_max_iter, double irls_tol, const void* theta_row,
                                         const void* theta_col, double loss_param, double robust_delta);
RCPPML_GPU_API int rcppml_hip_irls_loss(rcppml_hip_ctx* ctx, int dtype, int loss_type, const int* col_ptr,
                                        const int* row_idx, const void* values, int64_t ncols, const void* W_T,
                                        const void* d, const void* H, const void* theta_row, int k, double loss_param,
                                        double robust_delta, double* out);
/* Dense-input right-hand sides (reference primitives::rhs<CPU> on a dense A and detail::rhs_transpose,
 * nmf/fit_cpu.hpp:547-549 / :783): transposed = 0: B (k x n) = F (k x m) A;  1: B (k x m) = F (k x n) A^T.
 * A column-major m x n on the device; F, B column-major with leading dimension k, k <= 128.  Hand-written skinny MFMA GEMMs
 * (fp32 32x32x2 tiles, fp64 16x16x4 tiles) that stream A once; no BLAS library. */
RCPPML_GPU_API int rcppml_hip_rhs_dense(rcppml_hip_ctx* ctx, int dtype, const void* A, int64_t m, int64_t n,
                                        int transposed, const void* F, int k, void* B);
/* Same decoder on a byte buffer in host memory, into caller-allocated device arrays (layer 2; what the parity tests
 * call).  Format restated from streampress/sparsepress_v2.hpp:897-1090, codec/rans.hpp:128-247, codec/varint.hpp:43-52.
 * rcppml_hip_spz_info parses the 128-byte header only (host).  Both return the out_status codes above. */
RCPPML_GPU_API int rcppml_hip_spz_info(const void* file_bytes, uint64_t size, int* m, int* n, int64_t* nnz, int* value_type);
RCPPML_GPU_API int rcppml_hip_spz_decode(rcppml_hip_ctx* ctx, const void* file_bytes, uint64_t size, int* d_col_ptr,
                                         int* d_row_idx, double* d_values);

/* Dispersion estimators of the other IRLS losses, per ROW of A (takes CSC(A^T), nnz = its nonzero count):
 *   loss_type 4        GP theta by the auxiliary-function (MM) update, five inner passes -- reference
 *                      nmf/fit_cpu.hpp:914-1008 (PER_ROW / GLOBAL, sparse branch); hi = gp_theta_max, lo unused;
 *   loss_type 6 / 7 / 8  Gamma / inverse-Gaussian / Tweedie phi by the Pearson method of moments over the positive
 *                      nonzeros, clamped to [lo, hi] -- nmf/fit_cpu.hpp:1561-1670; power = Tweedie variance power.
 * mode 2 = PER_ROW, 1 = GLOBAL (GP: mean of the per-row values; phi: the median sorted[m/2]).  theta (m) is updated in
 * place (rows without usable nonzeros keep their value). */
RCPPML_GPU_API int rcppml_hip_dispersion_update(rcppml_hip_ctx* ctx, int dtype, int loss_type, int mode,
                                                const int* t_col_ptr, const int* t_row_idx, const void* t_values,
                                                int64_t m, int64_t nnz, const void* W_T, const void* d, const void* H,
                                                int64_t n, int k, double power, double lo, double hi, void* theta);
/* GLOBAL dispersion: x[0..m) <- mean(x) (stat 0; fit_cpu.hpp:1005-1008) or <- sorted(x)[m/2] (stat 1: the reference's
 * nth_element at m/2; NB fit_cpu.hpp:1257-1262, phi :1664-1669). */
RCPPML_GPU_API int rcppml_hip_vec_global(rcppml_hip_ctx* ctx, int dtype, int stat, void* x, int64_t m);
/* NB size (r) per ROW of A by the method of moments -- reference nmf/fit_cpu.hpp:1094-1265 (PER_ROW branch, sparse):
 * r_i = clamp(S mu^2 / (S (y-mu)^2 - S mu), r_min, r_max), else r_max.  Takes CSC(A^T); W_T k x m, H k x n. */
RCPPML_GPU_API int rcppml_hip_nb_size_update(rcppml_hip_ctx* ctx, int dtype, const int* t_col_ptr,
                                             const int* t_row_idx, const void* t_values, int64_t m, const void* W_T,
                                             const void* d, const void* H, int64_t n, int k, double r_min,
                                             double r_max, void* nb_size);
/* The two calls above and below in one pass over CSC(A^T) (PER_ROW dispersion, no robust modifier): nb_size updated as by
 * rcppml_hip_nb_size_update, then out[0] = the NB negative log-likelihood with the UPDATED sizes, as rcppml_hip_nb_loss returns it
 * (the order of the reference's fit loop, nmf/fit_cpu.hpp:1094-1265 then :1684-1753 / explicit_loss.hpp:53-77).  The predictions of
 * the first pass are parked (nnz Scalars of context scratch) and the likelihood terms are evaluated from them: one gather of
 * factor rows per nonzero instead of two, lgamma(r_i) once per row.  nnz = t_col_ptr[m]. */
RCPPML_GPU_API int rcppml_hip_nb_size_update_loss(rcppml_hip_ctx* ctx, int dtype, const int* t_col_ptr,
                                                  const int* t_row_idx, const void* t_values, int64_t m, int64_t nnz,
                                                  const void* W_T, const void* d, const void* H, int64_t n, int k,
                                                  double r_min, double r_max, void* nb_size, double* out);
/* out[0] = NB negative log-likelihood over the NONZEROS of A with per-row size -- reference nmf/explicit_loss.hpp:53-77,
 * math/loss.hpp:415-426. */
RCPPML_GPU_API int rcppml_hip_nb_loss(rcppml_hip_ctx* ctx, int dtype, const int* col_ptr, const int* row_idx,
                                      const void* values, int64_t ncols, const void* W_T, const void* d, const void* H,
                                      const void* theta_row, int k, double* out);

/* Cross-validation with the IRLS losses -- reference nmf/cv_detail.hpp:101-292 (irls_solve_col_cv / irls_solve_row_cv), callers
 * nmf/fit_cv.hpp:446-456, :670-689.  Per column of the CSC (H side: A, F = W_T; W side: A^T with transposed = 1, F = H; the speckled
 * mask SplitMix64::hash(seed, i, j) < UINT64_MAX / floor(1 / fraction) is always asked in the coordinates of A), up to irls_max_iter
 * passes of:  G_w = sum over the TRAINING entries of w f f^T + G_add + 1e-15 I,  b_w = sum (w a) f,  x <- CD(G_w, b_w, x; L1
 * inside, cd_maxit sweeps, no tolerance) or Cholesky + clip, started from the current column WITHOUT a residual correction of
 * b_w, until max |dx| / (|x_old| + 1e-12) < irls_tol.  Training entries: the nonzeros that are not held out (mask_zeros = 1) or
 * every row that is not held out, zeros included (0).  The weight is the reference's compute_irls_weight(residual, predicted,
 * loss) with its DEFAULT observed = 0 and theta = 0: NB / Gamma / inverse Gaussian / Tweedie distribution weights at theta 0,
 * GP = irls_weight_gp(0, mu, 0, blend 1) (math/loss.hpp:197-229; not the KL weight of the non-CV fit), times the Huber
 * modifier when robust_delta > 0.  G_add (k x k, may be NULL): the additive CV features (L2, graph, L21).  k <= 128. */
RCPPML_GPU_API int rcppml_hip_solve_cv_irls(rcppml_hip_ctx* ctx, int dtype, int loss_type, const int* col_ptr, const int* row_idx,
                                            const void* values, int64_t ncols, int nrows, const void* F, const void* G_add, void* X,
                                            int k, double holdout_fraction, unsigned long long cv_seed, int mask_zeros,
                                            int transposed, double l1, int nonneg, int cd_maxit, int solver_mode, int irls_max_iter,
                                            double irls_tol, double loss_param, double robust_delta);
/* out4 = {train sum, n_train, test sum, n_test} of compute_loss(value, prediction, loss, theta) (math/loss.hpp:511-535; loss_type 0
 * = squared error) over the entries of A -- the nonzeros (mask_zeros = 1) or every entry -- split by the speckled mask; theta_row
 * (per row, may be NULL) is read by GP only.  Reference nmf/fit_cv.hpp:1377-1443. */
RCPPML_GPU_API int rcppml_hip_cv_irls_loss(rcppml_hip_ctx* ctx, int dtype, int loss_type, const int* col_ptr, const int* row_idx,
                                           const void* values, int64_t ncols, int nrows, const void* W_T, const void* d,
                                           const void* H, const void* theta_row, int k, double holdout_fraction,
                                           unsigned long long cv_seed, int mask_zeros, double loss_param, double* out4);
/* GP theta per row by the MM update (five inner passes) over the TRAINING entries only -- reference nmf/fit_cv.hpp:866-961: held-out
 * nonzeros are skipped and the held-out pairs' predictions (zeros included) are taken out of sum_s.  Takes CSC(A^T) (nnz = its
 * nonzero count).  mode 2 = PER_ROW, 1 = GLOBAL (mean).  holdout_fraction 0: the non-CV update. */
RCPPML_GPU_API int rcppml_hip_cv_gp_theta_update(rcppml_hip_ctx* ctx, int dtype, int mode, const int* t_col_ptr, const int* t_row_idx,
                                                 const void* t_values, int64_t m, int64_t nnz, const void* W_T, const void* d,
                                                 const void* H, int64_t n, int k, double holdout_fraction,
                                                 unsigned long long cv_seed, double theta_max, void* theta);

#ifdef __cplusplus
}
#endif
#endif /* RCPPML_GPU_H */
