// ============================================================================
// oracle/nmf_oracle.hpp -- CPU restatement of RcppML's alternating-NNLS NMF path
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may load the library built from
// this directory, and only as the checker / the timed CPU baseline.  The product
// path (rcppml_amd/, RcppML_gpu.so) never links, imports or calls it.
//
// PARITY STATUS: pinned bit-for-bit to the reference itself for SplitMix64 / factor
// initialisation (rng/rng.hpp) and the NB IRLS weight / NB NLL (math/loss.hpp), whose
// headers are self-contained and are compiled from /root/reference into oracle/_ref
// (make ref; tests/test_oracle_ref.py, tests/golden/ref_vectors.npz).
// "parity unpinned" for everything else except the known answers the
// reference's own tests hold (tests/cpp/test_nnls.cpp:65-84 2x2 NNLS,
// tests/cpp/test_rng.cpp:36-39 seed-0 remap, tests/cpp/test_nmf.cpp:14-27
// reconstruct()==6, tests/cpp/test_gram.cpp:52-66 gram == H*H^T) and the
// published SplitMix64 vectors.  The reference (FactorNet headers) needs Eigen >= 3.4
// and Rcpp; neither exists in this image and there is no network, so the reference
// cannot be compiled or run here.  Every function below
// cites the reference file:line it restates (paths relative to
// /root/reference/inst/include/FactorNet unless noted).
//
// Eigen is the one third-party dependency on the path (RcppEigen, version unpinned
// in DESCRIPTION; GUIDE says >= 3.4).  Its published algorithms are restated:
// selfadjointView::rankUpdate -> plain lower-triangle accumulation, LLT -> the
// standard (unblocked) lower Cholesky, llt.solve -> forward + back substitution.
// Summation order inside Eigen's kernels is not reproducible, so parity against
// the real reference could only ever be to tolerance, never bitwise.
// ============================================================================
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace oracle {

// ---------------------------------------------------------------------------
// rng/rng.hpp:60-104,194-201  SplitMix64 (seed 0 -> 12345), uniform<T>, fill_uniform
// ---------------------------------------------------------------------------
struct SplitMix64 {
    uint64_t state;
    explicit SplitMix64(uint64_t seed = 12345) : state(seed == 0 ? 12345ULL : seed) {}
    uint64_t next() {
        state += 0x9e3779b97f4a7c15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    template <class T> T uniform() { return static_cast<T>(next()) / static_cast<T>(UINT64_MAX); }
    template <class T> void fill_uniform(T* data, int rows, int cols) {
        for (int j = 0; j < cols; ++j)
            for (int i = 0; i < rows; ++i) data[(size_t)j * rows + i] = uniform<T>();
    }
};

// CSC view (int32 indices, as the reference: Eigen::SparseMatrix<Scalar,ColMajor,int>)
template <class S> struct Csc {
    int rows = 0, cols = 0;
    const int* p = nullptr;   // cols+1
    const int* i = nullptr;   // nnz
    const S* x = nullptr;     // nnz
};
template <class S> struct CscOwned {
    int rows = 0, cols = 0;
    std::vector<int> p, i;
    std::vector<S> x;
    Csc<S> view() const { return Csc<S>{rows, cols, p.data(), i.data(), x.data()}; }
};

// nmf/fit_cpu.hpp:251-253  At = A.transpose() (CSC of A^T, rows sorted within a column)
template <class S> CscOwned<S> transpose_csc(const Csc<S>& A) {
    CscOwned<S> T;
    T.rows = A.cols; T.cols = A.rows;
    const int nnz = A.p[A.cols];
    T.p.assign((size_t)A.rows + 1, 0);
    T.i.resize(nnz); T.x.resize(nnz);
    for (int t = 0; t < nnz; ++t) T.p[(size_t)A.i[t] + 1]++;
    for (int r = 0; r < A.rows; ++r) T.p[r + 1] += T.p[r];
    std::vector<int> cur(T.p.begin(), T.p.end() - 1);
    for (int j = 0; j < A.cols; ++j)
        for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
            const int dst = cur[A.i[t]]++;
            T.i[dst] = j; T.x[dst] = A.x[t];
        }
    return T;
}

inline int eff_threads(int threads) {
#ifdef _OPENMP
    return threads > 0 ? threads : omp_get_max_threads();
#else
    (void)threads; return 1;
#endif
}

// ---------------------------------------------------------------------------
// primitives/cpu/gram.hpp:37-67   G = F F^T (lower via rankUpdate, mirrored) + 1e-15 I
// F is k x r column-major.
// ---------------------------------------------------------------------------
template <class S> void gram(const S* F, int k, int r, S* G) {
    std::fill(G, G + (size_t)k * k, S(0));
    for (int c = 0; c < r; ++c) {
        const S* f = F + (size_t)c * k;
        for (int b = 0; b < k; ++b) {
            const S fb = f[b];
            S* g = G + (size_t)b * k;       // column b, rows a >= b (lower triangle)
            for (int a = b; a < k; ++a) g[a] += f[a] * fb;
        }
    }
    for (int b = 0; b < k; ++b)
        for (int a = b + 1; a < k; ++a) G[(size_t)a * k + b] = G[(size_t)b * k + a];
    for (int a = 0; a < k; ++a) G[(size_t)a * k + a] += static_cast<S>(1e-15);  // tiny_num
}

// primitives/primitives.hpp:100-115   trace_AtA (Scalar accumulate, storage order)
template <class S> S trace_AtA(const Csc<S>& A) {
    S total = 0;
    const int nnz = A.p[A.cols];
    for (int t = 0; t < nnz; ++t) total += A.x[t] * A.x[t];
    return total;
}

// primitives/cpu/rhs.hpp:52-70   B(:,j) = sum_{i in nz(j)} A(i,j) F(:,i)
template <class S> void rhs(const Csc<S>& A, const S* F, int k, S* B, int threads) {
    const int nt = eff_threads(threads); (void)nt;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt)
    for (int j = 0; j < A.cols; ++j) {
        S* b = B + (size_t)j * k;
        for (int f = 0; f < k; ++f) b[f] = 0;
        for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
            const S a = A.x[t];
            const S* fc = F + (size_t)A.i[t] * k;
            for (int f = 0; f < k; ++f) b[f] += a * fc[f];
        }
    }
}

// ---------------------------------------------------------------------------
// primitives/cpu/nnls_batch.hpp:70-132   cd_nnls_col_fixed
// Gauss-Seidel CD on the residual form; b and x updated in place.
// ---------------------------------------------------------------------------
template <class S>
int cd_nnls_col_fixed(const S* G, S* b, S* x, int k, S L1, S L2, bool nonneg, int maxit,
                      S upper_bound, S cd_tol) {
    const bool has_upper = upper_bound > 0;
    const bool check = cd_tol > 0;
    const S inv_k = S(1) / static_cast<S>(k);
    for (int it = 0; it < maxit; ++it) {
        S tol_sum = 0;
        for (int i = 0; i < k; ++i) {
            const S g = G[(size_t)i * k + i];
            if (g <= S(0)) continue;
            S diff = b[i] / g;
            if (L1 != 0) diff -= L1;
            if (L2 != 0) diff += L2 * x[i];
            const S nv = x[i] + diff;
            S ad;
            if (nonneg && nv < S(0)) {
                ad = -x[i];
                if (ad == S(0)) continue;
                x[i] = S(0);
            } else if (has_upper && nv > upper_bound) {
                ad = upper_bound - x[i];
                if (ad == S(0)) continue;
                x[i] = upper_bound;
            } else {
                if (diff == S(0)) continue;
                ad = diff;
                x[i] = nv;
            }
            if (check) {
                const S aa = ad >= 0 ? ad : -ad;
                tol_sum += aa / (std::abs(x[i]) + static_cast<S>(1e-15));  // CD_ABS_TOL
            }
            const S* gc = G + (size_t)i * k;
            for (int r = 0; r < k; ++r) b[r] -= gc[r] * ad;
        }
        if (check && tol_sum * inv_k < cd_tol) return it + 1;
    }
    return maxit;
}

// primitives/cpu/nnls_batch.hpp:150-225   nnls_batch (B modified in place)
template <class S>
void nnls_batch(const S* G, S* B, S* X, int k, int n, int cd_maxit, S cd_tol, S L1, S L2,
                bool nonneg, int threads, S upper_bound, bool warm_start) {
    const int nt = eff_threads(threads); (void)nt;
    if (warm_start) {
#pragma omp parallel for schedule(static) num_threads(nt)
        for (int j = 0; j < n; ++j) {
            S* b = B + (size_t)j * k; const S* x = X + (size_t)j * k;
            for (int c = 0; c < k; ++c) {
                const S xc = x[c]; const S* gc = G + (size_t)c * k;
                for (int r = 0; r < k; ++r) b[r] -= gc[r] * xc;
            }
        }
    } else {
        std::fill(X, X + (size_t)k * n, S(0));
    }
#pragma omp parallel for schedule(dynamic) num_threads(nt)
    for (int j = 0; j < n; ++j)
        cd_nnls_col_fixed(G, B + (size_t)j * k, X + (size_t)j * k, k, L1, L2, nonneg, cd_maxit,
                          upper_bound, cd_tol);
}

// ---------------------------------------------------------------------------
// Eigen::LLT restated: standard lower Cholesky G = L L^T, then solve.
// Returns false if a non-positive pivot is met (Eigen would report NumericalIssue).
// ---------------------------------------------------------------------------
template <class S> bool llt_factor(const S* G, int k, S* L) {
    std::fill(L, L + (size_t)k * k, S(0));
    bool ok = true;
    for (int j = 0; j < k; ++j) {
        S s = G[(size_t)j * k + j];
        for (int p = 0; p < j; ++p) s -= L[(size_t)p * k + j] * L[(size_t)p * k + j];
        if (!(s > S(0))) { ok = false; s = std::abs(s) + std::numeric_limits<S>::min(); }
        const S ljj = std::sqrt(s);
        L[(size_t)j * k + j] = ljj;
        for (int i = j + 1; i < k; ++i) {
            S t = G[(size_t)j * k + i];
            for (int p = 0; p < j; ++p) t -= L[(size_t)p * k + i] * L[(size_t)p * k + j];
            L[(size_t)j * k + i] = t / ljj;   // L(i,j), column-major
        }
    }
    return ok;
}
template <class S> void llt_solve(const S* L, int k, S* x) {  // x := (L L^T)^{-1} x
    for (int i = 0; i < k; ++i) {
        S t = x[i];
        for (int p = 0; p < i; ++p) t -= L[(size_t)p * k + i] * x[p];
        x[i] = t / L[(size_t)i * k + i];
    }
    for (int i = k - 1; i >= 0; --i) {
        S t = x[i];
        for (int p = i + 1; p < k; ++p) t -= L[(size_t)i * k + p] * x[p];
        x[i] = t / L[(size_t)i * k + i];
    }
}

// primitives/cpu/fused_nnls.hpp:70-134   fused_rhs_nnls_sparse
template <class S>
void fused_rhs_nnls_sparse(const Csc<S>& A, const S* F, const S* G, S* X, int k, int cd_maxit,
                           S cd_tol, S L1, bool nonneg, int threads, bool warm_start,
                           S upper_bound) {
    const int nt = eff_threads(threads); (void)nt;
    const bool has_L1 = L1 > S(0);
#pragma omp parallel num_threads(nt)
    {
        std::vector<S> bv(k);
        S* b = bv.data();
#pragma omp for schedule(dynamic)
        for (int j = 0; j < A.cols; ++j) {
            for (int f = 0; f < k; ++f) b[f] = 0;
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                const S a = A.x[t]; const S* fc = F + (size_t)A.i[t] * k;
                for (int f = 0; f < k; ++f) b[f] += a * fc[f];
            }
            if (has_L1) for (int f = 0; f < k; ++f) b[f] -= L1;
            S* x = X + (size_t)j * k;
            if (warm_start) {
                for (int c = 0; c < k; ++c) {
                    const S xc = x[c]; const S* gc = G + (size_t)c * k;
                    for (int r = 0; r < k; ++r) b[r] -= gc[r] * xc;
                }
            }
            cd_nnls_col_fixed(G, b, x, k, S(0), S(0), nonneg, cd_maxit, upper_bound, cd_tol);
        }
    }
}

// primitives/cpu/fused_nnls.hpp:155-221   fused_rhs_cholesky_sparse (solve + clip, no warm start)
template <class S>
void fused_rhs_cholesky_sparse(const Csc<S>& A, const S* F, const S* G, S* X, int k, S L1,
                               bool nonneg, int threads, S upper_bound) {
    const int nt = eff_threads(threads); (void)nt;
    std::vector<S> L((size_t)k * k);
    llt_factor(G, k, L.data());
    const bool has_L1 = L1 > S(0), do_upper = upper_bound > S(0);
#pragma omp parallel num_threads(nt)
    {
        std::vector<S> bv(k);
        S* b = bv.data();
#pragma omp for schedule(dynamic)
        for (int j = 0; j < A.cols; ++j) {
            for (int f = 0; f < k; ++f) b[f] = 0;
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                const S a = A.x[t]; const S* fc = F + (size_t)A.i[t] * k;
                for (int f = 0; f < k; ++f) b[f] += a * fc[f];
            }
            if (has_L1) for (int f = 0; f < k; ++f) b[f] -= L1;
            llt_solve(L.data(), k, b);
            S* x = X + (size_t)j * k;
            for (int f = 0; f < k; ++f) {
                S v = b[f];
                if (nonneg) v = std::max(v, S(0));
                if (do_upper) v = std::min(v, upper_bound);
                x[f] = v;
            }
        }
    }
}

// primitives/cpu/cholesky_clip.hpp:128-164   cholesky_clip_batch (dense-B variant)
template <class S>
void cholesky_clip_batch(const S* G, const S* B, S* X, int k, int n, bool nonneg, int threads) {
    const int nt = eff_threads(threads); (void)nt;
    std::vector<S> L((size_t)k * k);
    llt_factor(G, k, L.data());
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int j = 0; j < n; ++j) {
        S* x = X + (size_t)j * k;
        std::memcpy(x, B + (size_t)j * k, sizeof(S) * k);
        llt_solve(L.data(), k, x);
        if (nonneg) for (int f = 0; f < k; ++f) x[f] = std::max(x[f], S(0));
    }
}

// features/graph_reg.hpp:38-50   apply_graph_reg:  G += lambda * (X L) X^T,  L sparse (len x len, CSC), X k x len
template <class S> void apply_graph_reg(S* G, const Csc<S>& L, const S* X, int k, S lambda) {
    if (lambda <= 0) return;
    const int64_t len = L.cols;
    std::vector<S> FL((size_t)k * len, S(0));
    for (int64_t j = 0; j < len; ++j)
        for (int t = L.p[j]; t < L.p[j + 1]; ++t) {
            const S l = L.x[t]; const S* xi = X + (size_t)L.i[t] * k; S* fj = FL.data() + (size_t)j * k;
            for (int f = 0; f < k; ++f) fj[f] += l * xi[f];
        }
    for (int b = 0; b < k; ++b)
        for (int a = 0; a < k; ++a) {
            S acc = 0;
            for (int64_t j = 0; j < len; ++j) acc += FL[a + j * k] * X[b + j * k];
            G[(size_t)b * k + a] += lambda * acc;
        }
}

// features/L21.hpp:38-51   apply_L21: G(i,i) += lambda / ||factor.row(i)||_2 for rows with norm > 1e-10.
// X: k x len column-major (X[f + j*k]).
template <class S> void apply_L21(S* G, const S* X, int k, int64_t len, S lambda) {
    if (lambda <= 0) return;
    for (int i = 0; i < k; ++i) {
        S ss = 0;
        for (int64_t j = 0; j < len; ++j) ss += X[i + j * k] * X[i + j * k];
        const S row_norm = std::sqrt(ss);
        if (row_norm > static_cast<S>(1e-10)) G[(size_t)i * k + i] += lambda / row_norm;
    }
}

// features/angular.hpp:67-103   apply_angular_posthoc: Factor -= lambda * diag(norms) * (offdiag(F_hat F_hat^T) F_hat),
// F_hat = rows scaled to unit norm (rows with norm <= 1e-15 stay as they are), then clamp at 0.
template <class S> void apply_angular_posthoc(S* X, int k, int64_t len, S lambda) {
    if (lambda <= 0) return;
    std::vector<S> norms(k), Fh((size_t)k * len), cosm((size_t)k * k, S(0));
    for (int i = 0; i < k; ++i) {
        S ss = 0;
        for (int64_t j = 0; j < len; ++j) ss += X[i + j * k] * X[i + j * k];
        norms[i] = std::sqrt(ss);
    }
    for (int64_t j = 0; j < len; ++j)
        for (int i = 0; i < k; ++i) Fh[i + j * k] = norms[i] > S(1e-15) ? X[i + j * k] / norms[i] : X[i + j * k];
    for (int64_t j = 0; j < len; ++j)                      // rankUpdate: lower triangle, then mirrored
        for (int a = 0; a < k; ++a)
            for (int b = 0; b <= a; ++b) cosm[(size_t)b * k + a] += Fh[a + j * k] * Fh[b + j * k];
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < a; ++b) cosm[(size_t)a * k + b] = cosm[(size_t)b * k + a];
    for (int a = 0; a < k; ++a) cosm[(size_t)a * k + a] = 0;
    std::vector<S> g(k);
    for (int64_t j = 0; j < len; ++j) {
        for (int a = 0; a < k; ++a) {
            S acc = 0;
            for (int b = 0; b < k; ++b) acc += cosm[(size_t)b * k + a] * Fh[b + j * k];
            g[a] = acc * norms[a];
        }
        for (int a = 0; a < k; ++a) {
            const S v = X[a + j * k] - lambda * g[a];
            X[a + j * k] = v > S(0) ? v : S(0);
        }
    }
}

// features/bounds.hpp:38-42   apply_upper_bound
template <class S> void apply_upper_bound(S* X, size_t len, S ub) {
    for (size_t t = 0; t < len; ++t) X[t] = std::min(X[t], ub);
}

// nmf/variant_helpers.hpp:286-305   extract_scaling   (norm_type: 0=L1, 1=L2, 2=None)
template <class S> void extract_scaling(S* X, int k, int c, S* d, int norm_type) {
    if (norm_type == 2) { for (int i = 0; i < k; ++i) d[i] = 1; return; }
    for (int i = 0; i < k; ++i) d[i] = 0;
    if (norm_type == 0) {
        for (int j = 0; j < c; ++j) { const S* x = X + (size_t)j * k; for (int i = 0; i < k; ++i) d[i] += std::abs(x[i]); }
    } else {
        for (int j = 0; j < c; ++j) { const S* x = X + (size_t)j * k; for (int i = 0; i < k; ++i) d[i] += x[i] * x[i]; }
        for (int i = 0; i < k; ++i) d[i] = std::sqrt(d[i]);
    }
    for (int i = 0; i < k; ++i) d[i] += static_cast<S>(1e-15);
    for (int j = 0; j < c; ++j) { S* x = X + (size_t)j * k; for (int i = 0; i < k; ++i) x[i] /= d[i]; }
}

// primitives/cpu/fused_nnls.hpp:305-362   loss_cross_term_sparse_via_At
template <class S>
S loss_cross_term_sparse_via_At(const Csc<S>& At, const S* W_T, const S* H, const S* d, int k,
                                int threads) {
    const int nt = eff_threads(threads); (void)nt;
    const int m = At.cols;
    S cross = 0;
#pragma omp parallel num_threads(nt) reduction(+ : cross)
    {
        std::vector<S> h_at(k), u(k);
#pragma omp for schedule(dynamic)
        for (int l = 0; l < m; ++l) {
            for (int f = 0; f < k; ++f) h_at[f] = 0;
            for (int t = At.p[l]; t < At.p[l + 1]; ++t) {
                const S a = At.x[t]; const S* hc = H + (size_t)At.i[t] * k;
                for (int f = 0; f < k; ++f) h_at[f] += a * hc[f];
            }
            const S* w = W_T + (size_t)l * k;
            S dot = 0;
            for (int f = 0; f < k; ++f) dot += (w[f] * d[f]) * h_at[f];
            cross += dot;
        }
    }
    return cross;
}

// ---------------------------------------------------------------------------
// Fit configuration and result carriers (core/config.hpp:54-454 defaults,
// core/result.hpp:71-189).  Only the fields the hot path reads.
// ---------------------------------------------------------------------------
template <class S> struct FitConfig {
    int k = 2;
    int max_iter = 100;            // config.hpp max_iter
    S tol = S(1e-4);
    S L1_H = 0, L1_W = 0, L2_H = 0, L2_W = 0, ub_H = 0, ub_W = 0;
    S L21_H = 0, L21_W = 0, angular_H = 0, angular_W = 0;   // features/L21.hpp, features/angular.hpp (post-hoc form)
    int cd_maxit = 100;
    S cd_tol = S(1e-8);
    int patience = 5;
    bool nonneg_W = true, nonneg_H = true;
    int norm_type = 0;             // L1
    int solver_mode = 0;           // 0 = CD, 1 = Cholesky + clip
    int loss_type = 0;             // 0 = MSE, 5 = NB
    int irls_max_iter = 5;
    S irls_tol = S(1e-4);
    int dispersion_mode = 2;       // PER_ROW
    S nb_size_init = 10, nb_size_max = S(1e6), nb_size_min = S(0.01);
    S gp_theta_init = S(0.1), gp_theta_max = S(5.0);                        // core/config.hpp:169-172
    S gamma_phi_init = S(1.0), gamma_phi_max = S(1e4), gamma_phi_min = S(1e-6);   // core/config.hpp:201-207
    bool has_graph_H = false, has_graph_W = false;   // FactorConfig::graph (Laplacians), graph_lambda
    Csc<S> graph_H, graph_W; S graph_H_lambda = 0, graph_W_lambda = 0;
    bool projective = false;
    bool unfused = false;          // dense input: the STANDARD (separate RHS -> features -> nnls_batch) path of fit_cpu.hpp:540-640, :774-882
    bool dense_input = false;      // A stores EVERY entry (oracle.py dense_as_csc) and the fit takes the reference's DENSE branches where they differ from the
                                   // sparse ones: irls_nnls_col_dense (nnls_batch_irls.hpp:376-450), fit_cpu.hpp:953-968, :1041-1053, :1137-1148, :1226-1238
    bool symmetric = false;        // A ~ W diag(d) W^T, A square: only W is solved, H = W_T (fit_cpu.hpp:659-704)                  // NMFConfig::projective: H = diag(d) W_T A instead of an NNLS solve
    S robust_delta = 0;                       // LossConfig::robust_delta (math/loss.hpp:89-96): > 0 -> Huber on Pearson residuals
    S tweedie_power = S(1.5);                 // LossConfig::power_param (math/loss.hpp:99-104), loss_type 8 only
    // target regularisation (core/factor_config.hpp:80-102; nmf/variant_helpers.hpp:107-146): k x n / k x m, nullptr = none
    const S* target_H = nullptr; S target_lambda_H = 0;
    const S* target_W = nullptr; S target_lambda_W = 0;
    bool sort_model = true;
    int threads = 0;
    // explicit mask (nonzero = masked), 0 cols => absent  (core/config.hpp:411-413)
    Csc<S> mask;
    bool has_mask = false;
};
// nmf/variant_helpers.hpp:107-146 (target part of apply_features).  target_gram = T T^T / ncols (nmf/fit.hpp:259-271).
template <class S> void proj_adv_gram(S* G, const S* target_gram, int k, S abs_lambda);
template <class S> void apply_target(S* G, S* B, const S* target, const S* target_gram, int k, int64_t ncols, S lambda);
template <class S> struct FitResult {
    int iterations = 0;
    bool converged = false;
    S train_loss = 0, final_tol = 0;
    std::vector<S> loss_history;
    std::vector<S> theta;
};

// core/result.hpp:169-188  sort by descending d (std::sort, ties unspecified; we use stable)
template <class S> void sort_by_d(S* W_T, S* H, S* d, int k, int m, int n) {
    std::vector<int> idx(k);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] > d[b]; });
    std::vector<S> tmp(k);
    for (int j = 0; j < m; ++j) { S* w = W_T + (size_t)j * k; for (int i = 0; i < k; ++i) tmp[i] = w[idx[i]]; std::memcpy(w, tmp.data(), sizeof(S) * k); }
    for (int j = 0; j < n; ++j) { S* h = H + (size_t)j * k; for (int i = 0; i < k; ++i) tmp[i] = h[idx[i]]; std::memcpy(h, tmp.data(), sizeof(S) * k); }
    for (int i = 0; i < k; ++i) tmp[i] = d[idx[i]];
    std::memcpy(d, tmp.data(), sizeof(S) * k);
}

// Implemented in nmf_oracle.cpp
template <class S>
FitResult<S> nmf_fit(const Csc<S>& A, const FitConfig<S>& cfg, S* W_T /*k x m in/out*/,
                     S* H /*k x n in/out*/, S* d /*k out*/);

}  // namespace oracle
