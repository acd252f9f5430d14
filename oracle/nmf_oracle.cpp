// ============================================================================
// oracle/nmf_oracle.cpp -- CPU restatement of RcppML's ALS-NNLS NMF loop + C surface
//
// TEST INFRASTRUCTURE ONLY (see nmf_oracle.hpp header).  "parity unpinned" except for
// the known answers listed there.  Built by oracle/Makefile into oracle/liboracle.so.
//
// Restates (paths relative to /root/reference/inst/include/FactorNet):
//   nmf/fit_cpu.hpp:171-254        setup (W_T, H, d, trAtA, CSC(A^T))
//   nmf/fit_cpu.hpp:444-894        ALS loop, H then W half-update
//   nmf/fit_cpu.hpp:1674-1855      loss, convergence/patience, packaging, sort
//   nmf/masked_nnls.hpp:45-282     explicit-mask per-column NNLS and loss
//   primitives/cpu/nnls_batch_irls.hpp:202-329,465-520  per-column IRLS (NB weights)
//   nmf/fit_cpu.hpp:1094-1265      NB size (r) method-of-moments update
//   nmf/explicit_loss.hpp:53-77, math/loss.hpp:248-256,415-426   NB weight and NLL
//   src/RcppFunctions_utils.cpp:23-52,95-163,313-366  predict / evaluate / c_nnls (fp64)
// ============================================================================
#include "nmf_oracle.hpp"

namespace oracle {

// math/loss.hpp:248-256  irls_weight_nb (computed in double, eps = tiny_num = 1e-15)
template <class S> static inline S irls_weight_nb(S predicted, S nb_size) {
    double mu = std::max(static_cast<double>(predicted), static_cast<double>(static_cast<S>(1e-15)));
    double r = std::max(static_cast<double>(nb_size), 1e-10);
    double w = r / (mu * (r + mu));
    w = std::min(w, 1e6);
    return static_cast<S>(w);
}
// math/loss.hpp:415-426  loss_contribution_nb
template <class S> static inline S loss_contribution_nb(S observed, S predicted, S nb_size) {
    double y = static_cast<double>(observed);
    double mu = std::max(static_cast<double>(predicted), 1e-10);
    double r = std::max(static_cast<double>(nb_size), 1e-10);
    double nll = -std::lgamma(y + r) + std::lgamma(r) - r * std::log(r / (r + mu)) -
                 y * std::log(mu / (r + mu));
    return static_cast<S>(nll);
}

// math/loss.hpp:176-179  irls_weight_kl (eps = 1e-4, Scalar arithmetic) -- the weight the GP loss uses for W/H
// (fit_cpu.hpp:568-574: "GP strategy: use KL weights for W/H updates")
template <class S> static inline S irls_weight_kl(S predicted) {
    return static_cast<S>(1) / std::max(predicted, static_cast<S>(1e-4));
}
// math/loss.hpp:382-398  loss_contribution_gp (fp64 inside)
// math/loss.hpp:197-229  irls_weight_gp (observed Fisher information, geometric blend with the KL weight, fp64 inside).
// The non-CV fit never calls it (fit_cpu.hpp:568-574 switches the GP updates to KL weights); the CV path does, through
// compute_irls_weight's defaults: observed = 0, theta = 0, blend = LossConfig::gp_blend = 1 (cv_detail.hpp:154).
template <class S> static inline S irls_weight_gp(S observed, S predicted, S theta, S blend = S(1)) {
    const double s = std::max(static_cast<double>(predicted), static_cast<double>(static_cast<S>(1e-15)));
    const double y = static_cast<double>(observed), th = static_cast<double>(theta), bl = static_cast<double>(blend);
    const double eff_blend = bl * std::min(s, 1.0);
    double w_gp = 1.0 / (s * s);
    if (y >= 1.0) {
        double denom = s + th * y;
        denom = std::max(denom, static_cast<double>(static_cast<S>(1e-15)));
        w_gp += (y - 1.0) / (denom * denom);
    }
    if (eff_blend < 0.999) {
        const double log_w_kl = -std::log(s);
        const double log_w_gp = std::log(std::max(w_gp, 1e-300));
        double weight = std::exp((1.0 - eff_blend) * log_w_kl + eff_blend * log_w_gp);
        weight = std::min(weight, 1e6);
        return static_cast<S>(weight);
    }
    w_gp = std::min(w_gp, 1e6);
    return static_cast<S>(w_gp);
}
template <class S> static inline S loss_contribution_gp(S observed, S predicted, S theta) {
    double s = std::max(static_cast<double>(predicted), 1e-10);
    double y = static_cast<double>(observed);
    double th = static_cast<double>(theta);
    double one_plus_theta = 1.0 + th;
    double loss = -std::log(s / one_plus_theta);
    if (y >= 1.0) {
        double inner = (s + th * y) / one_plus_theta;
        inner = std::max(inner, 1e-10);
        loss -= (y - 1.0) * std::log(inner);
    }
    loss += (s + th * y) / one_plus_theta;
    return static_cast<S>(loss);
}
// math/loss.hpp:270-278  irls_weight_power: V(mu) = mu^p (Gamma p = 2, inverse Gaussian p = 3, Tweedie p free)
template <class S> static inline S irls_weight_power(S predicted, S power) {
    double mu = std::max(static_cast<double>(predicted), static_cast<double>(static_cast<S>(1e-15)));
    double w = 1.0 / std::pow(mu, static_cast<double>(power));
    w = std::min(w, 1e6);
    return static_cast<S>(w);
}
// math/loss.hpp:439-505  deviance contributions
template <class S> static inline S loss_contribution_gamma(S observed, S predicted) {
    double y = std::max(static_cast<double>(observed), 1e-10);
    double mu = std::max(static_cast<double>(predicted), 1e-10);
    return static_cast<S>(2.0 * (-std::log(y / mu) + (y - mu) / mu));
}
template <class S> static inline S loss_contribution_invgauss(S observed, S predicted) {
    double y = std::max(static_cast<double>(observed), 1e-10);
    double mu = std::max(static_cast<double>(predicted), 1e-10);
    double diff = y - mu;
    return static_cast<S>(diff * diff / (mu * mu * y));
}
template <class S> static inline S loss_contribution_tweedie(S observed, S predicted, S p) {
    double y = std::max(static_cast<double>(observed), 1e-10);
    double mu = std::max(static_cast<double>(predicted), 1e-10);
    double pp = static_cast<double>(p);
    if (std::abs(pp - 1.0) < 1e-6) return static_cast<S>(2.0 * (y * std::log(y / mu) - (y - mu)));
    if (std::abs(pp - 2.0) < 1e-6) return static_cast<S>(2.0 * (-std::log(y / mu) + (y - mu) / mu));
    double omp = 1.0 - pp, tmp = 2.0 - pp;
    double term1 = std::pow(y, tmp) / (omp * tmp);
    double term2 = y * std::pow(mu, omp) / omp;
    double term3 = std::pow(mu, tmp) / tmp;
    return static_cast<S>(2.0 * (term1 - term2 + term3));
}
// nnls_batch_irls.hpp:57-83 distribution_weight: 5 = NB, 4 = GP (-> KL weight), 6 = Gamma, 7 = inverse Gaussian,
// 8 = Tweedie(power); math/loss.hpp:511-535 compute_loss
template <class S> static inline S irls_weight(int loss_type, S predicted, S theta, S power = S(1.5)) {
    switch (loss_type) {
        case 4: return irls_weight_kl(predicted);
        case 6: return irls_weight_power(predicted, S(2));
        case 7: return irls_weight_power(predicted, S(3));
        case 8: return irls_weight_power(predicted, power);
        default: return irls_weight_nb(predicted, theta);
    }
}
template <class S> static inline S loss_contribution(int loss_type, S observed, S predicted, S theta, S power = S(1.5)) {
    switch (loss_type) {
        case 4: return loss_contribution_gp(observed, predicted, theta);
        case 6: return loss_contribution_gamma(observed, predicted);
        case 7: return loss_contribution_invgauss(observed, predicted);
        case 8: return loss_contribution_tweedie(observed, predicted, power);
        default: return loss_contribution_nb(observed, predicted, theta);
    }
}

// math/loss.hpp:294-303  robust_huber_modifier (eps = tiny_num = 1e-15)
template <class S> static inline S robust_huber_modifier(S pearson_residual, S delta) {
    const S abs_r = std::abs(pearson_residual);
    if (abs_r <= delta) return static_cast<S>(1);
    return delta / (abs_r + static_cast<S>(1e-15));
}
// nnls_batch_irls.hpp:95-120  compute_irls_weight: distribution weight x optional Huber modifier on the Pearson residual
// (loss_type 0 = MSE: distribution weight 1, only reached with robust_delta > 0)
template <class S> static inline S compute_irls_weight(int loss_type, S residual, S predicted, S theta, S power, S robust_delta) {
    const S w_dist = loss_type == 0 ? S(1) : irls_weight(loss_type, predicted, theta, power);
    if (robust_delta > S(0)) {
        const S sd_inv = std::sqrt(std::max(w_dist, static_cast<S>(1e-15)));
        return w_dist * robust_huber_modifier(residual * sd_inv, robust_delta);
    }
    return w_dist;
}
// math/loss.hpp:549-607  compute_robust_loss: Huber rho of the Pearson residual (variance function per distribution)
template <class S> static inline S compute_robust_loss(int loss_type, S observed, S predicted, S theta, S power, S robust_delta) {
    if (robust_delta <= S(0)) return loss_type == 0 ? (observed - predicted) * (observed - predicted)
                                                    : loss_contribution(loss_type, observed, predicted, theta, power);
    const S mu = std::max(predicted, static_cast<S>(1e-10));
    const S residual = observed - mu;
    S var_mu;
    switch (loss_type) {
        case 4: case 3: var_mu = mu; break;
        case 5: { const S r = std::max(theta, static_cast<S>(1e-10)); var_mu = mu + mu * mu / r; break; }
        case 6: var_mu = mu * mu; break;
        case 7: var_mu = mu * mu * mu; break;
        case 8: var_mu = static_cast<S>(std::pow(static_cast<double>(mu), static_cast<double>(power))); break;
        default: var_mu = S(1); break;
    }
    const S sd = std::sqrt(std::max(var_mu, static_cast<S>(1e-20)));
    const S pearson_r = residual / sd;
    const S abs_pr = std::abs(pearson_r);
    if (abs_pr <= robust_delta) return static_cast<S>(0.5) * pearson_r * pearson_r;
    return robust_delta * abs_pr - static_cast<S>(0.5) * robust_delta * robust_delta;
}

// nmf/masked_nnls.hpp:96-154 (H side) / :177-242 (W side): same routine, data = A or A^T
template <class S>
static void masked_nnls(const Csc<S>& A, const S* F, const S* G_full, S* X, const Csc<S>& mask,
                        int k, S L1, S L2, bool nonneg, int cd_maxit, S cd_tol, int solver_mode,
                        int threads, bool warm_start) {
    const int nt = eff_threads(threads); (void)nt;
    const int nrow = A.rows;
#pragma omp parallel num_threads(nt)
    {
        std::vector<S> b(k), x(k), Gl((size_t)k * k), L((size_t)k * k);
        std::vector<char> is_masked(nrow, 0);
        std::vector<int> mrows;
#pragma omp for schedule(dynamic, 64)
        for (int j = 0; j < A.cols; ++j) {
            mrows.clear();
            for (int t = mask.p[j]; t < mask.p[j + 1]; ++t)
                if (mask.x[t] != 0) mrows.push_back(mask.i[t]);
            for (int f = 0; f < k; ++f) b[f] = 0;
            for (int r : mrows) is_masked[r] = 1;
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                if (is_masked[A.i[t]]) continue;
                const S a = A.x[t]; const S* fc = F + (size_t)A.i[t] * k;
                for (int f = 0; f < k; ++f) b[f] += a * fc[f];
            }
            for (int r : mrows) is_masked[r] = 0;
            std::memcpy(Gl.data(), G_full, sizeof(S) * k * k);
            for (int r : mrows) {
                const S* fr = F + (size_t)r * k;
                for (int c = 0; c < k; ++c) {
                    const S fc = fr[c];
                    S* g = Gl.data() + (size_t)c * k;
                    for (int a2 = 0; a2 < k; ++a2) g[a2] -= fr[a2] * fc;
                }
            }
            for (int i = 0; i < k; ++i) { b[i] -= L1; Gl[(size_t)i * k + i] += L2; }
            S* xj = X + (size_t)j * k;
            for (int i = 0; i < k; ++i) x[i] = warm_start ? xj[i] : S(0);
            if (solver_mode == 1) {
                // cholesky_clip_col(G_local, b, x, k, 0, 0, nonneg, ...)  cholesky_clip.hpp:64-106
                llt_factor(Gl.data(), k, L.data());
                for (int i = 0; i < k; ++i) x[i] = b[i];
                llt_solve(L.data(), k, x.data());
                if (nonneg) for (int i = 0; i < k; ++i) x[i] = std::max(x[i], S(0));
            } else {
                cd_nnls_col_fixed(Gl.data(), b.data(), x.data(), k, S(0), S(0), nonneg, cd_maxit,
                                  S(0), cd_tol);
            }
            for (int i = 0; i < k; ++i) xj[i] = x[i];
        }
    }
}

// ---------------------------------------------------------------------------
// Cross-validation path (nmf/fit_cv.hpp, nmf/cv_detail.hpp, nmf/speckled_cv.hpp), MSE, sparse A, no user mask.
// ---------------------------------------------------------------------------
// rng/rng.hpp:129-170  SplitMix64::hash / is_holdout; speckled_cv.hpp:57-83 LazySpeckledMask (no subsampling)
struct SpeckledMask {
    uint64_t seed, inv_prob, threshold;
    bool mask_zeros;
    SpeckledMask(double holdout_fraction, uint64_t cv_seed, bool mz) {
        seed = static_cast<uint32_t>(cv_seed) == 0 ? 12345ULL : static_cast<uint32_t>(cv_seed);
        inv_prob = holdout_fraction > 0 ? static_cast<uint64_t>(1.0 / holdout_fraction) : 0;
        threshold = inv_prob ? UINT64_MAX / inv_prob : 0;
        mask_zeros = mz;
    }
    static uint64_t hash(uint64_t seed, uint32_t i, uint32_t j) {
        uint64_t h = seed + static_cast<uint64_t>(i) * 0x9e3779b97f4a7c15ULL + static_cast<uint64_t>(j) * 0x6c62272e07bb0142ULL;
        h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ULL;
        h = (h ^ (h >> 27)) * 0x94d049bb133111ebULL;
        return h ^ (h >> 31);
    }
    // (i, j) always in the coordinates of A, also on the W side
    bool is_holdout(int i, int j) const { return inv_prob != 0 && hash(seed, (uint32_t)i, (uint32_t)j) < threshold; }
};

// One half-update of the CV path: fit_cv.hpp:420-478 (H side: data = A, columns j, F = W_T) / :591-830 (W side:
// data = A^T, "columns" = rows i of A, F = H; `transposed` swaps the arguments of the mask).
// Per column: b = train right-hand side (cv_detail.hpp:304-352 / :355-405), G_local = G - sum_test f f^T (:66-85),
// then cholesky_clip_col(G_local, b, x, L1) or cd_nnls_col_fixed(G_local, b, x, L1 inside, cd_maxit sweeps, no
// tolerance) started from the current column WITHOUT a warm-start correction of b -- as the reference does.
// umask (optional): the user mask in the orientation of D (pattern; nmf/cv_detail.hpp:433-468 adjust_rhs_for_user_mask_h / _w): a
// masked row that is not already a test row leaves b (if its entry is a nonzero) and joins the rows of the Gram correction
template <class S>
static void cv_half_update(const Csc<S>& D, const S* F, const S* G_full, S* X, int k, const SpeckledMask& mask,
                           bool transposed, S L1, bool nonneg, int cd_maxit, int solver_mode, int threads, const Csc<S>* umask = nullptr) {
    const int nt = eff_threads(threads); (void)nt;
    const int nrow = D.rows;
#pragma omp parallel num_threads(nt)
    {
        std::vector<S> b(k), x(k), Gl((size_t)k * k), L((size_t)k * k);
        std::vector<int> test;
#pragma omp for schedule(dynamic, 64)
        for (int j = 0; j < D.cols; ++j) {
            test.clear();
            for (int f = 0; f < k; ++f) b[f] = 0;
            auto held = [&](int r) { return transposed ? mask.is_holdout(j, r) : mask.is_holdout(r, j); };
            if (mask.mask_zeros) {
                for (int t = D.p[j]; t < D.p[j + 1]; ++t) {
                    const int r = D.i[t];
                    if (held(r)) test.push_back(r);
                    else { const S a = D.x[t]; const S* fc = F + (size_t)r * k; for (int f = 0; f < k; ++f) b[f] += a * fc[f]; }
                }
            } else {
                int t = D.p[j];
                const int te = D.p[j + 1];
                for (int r = 0; r < nrow; ++r) {
                    S val = 0;
                    if (t < te && D.i[t] == r) { val = D.x[t]; ++t; }
                    if (held(r)) test.push_back(r);
                    else if (val != S(0)) { const S* fc = F + (size_t)r * k; for (int f = 0; f < k; ++f) b[f] += val * fc[f]; }
                }
            }
            if (umask) {
                const size_t ntest = test.size();                 // `test` is ascending (both loops above visit rows in order)
                for (int t = umask->p[j]; t < umask->p[j + 1]; ++t) {
                    const int r = umask->i[t];
                    if (std::binary_search(test.begin(), test.begin() + ntest, r)) continue;
                    const int* lo = std::lower_bound(D.i + D.p[j], D.i + D.p[j + 1], r);
                    const S a = (lo != D.i + D.p[j + 1] && *lo == r) ? D.x[lo - D.i] : S(0);
                    if (a != S(0)) { const S* fc = F + (size_t)r * k; for (int f = 0; f < k; ++f) b[f] -= a * fc[f]; }
                    test.push_back(r);
                }
                std::sort(test.begin(), test.end());
            }
            std::memcpy(Gl.data(), G_full, sizeof(S) * k * k);
            for (int r : test) {
                const S* fr = F + (size_t)r * k;
                for (int c = 0; c < k; ++c) {
                    const S fc = fr[c];
                    S* g = Gl.data() + (size_t)c * k;
                    for (int a2 = 0; a2 < k; ++a2) g[a2] -= fr[a2] * fc;
                }
            }
            S* xj = X + (size_t)j * k;
            for (int i = 0; i < k; ++i) x[i] = xj[i];
            if (solver_mode == 1) {                      // cholesky_clip.hpp:64-106
                if (L1 > 0) for (int i = 0; i < k; ++i) b[i] -= L1;
                llt_factor(Gl.data(), k, L.data());
                for (int i = 0; i < k; ++i) x[i] = b[i];
                llt_solve(L.data(), k, x.data());
                if (nonneg) for (int i = 0; i < k; ++i) x[i] = std::max(x[i], S(0));
            } else {
                cd_nnls_col_fixed(Gl.data(), b.data(), x.data(), k, L1, S(0), nonneg, cd_maxit, S(0), S(0));
            }
            for (int i = 0; i < k; ++i) xj[i] = x[i];
        }
    }
}

// fit_cv.hpp:1444-1494  squared error and count over the held-out entries (zeros included unless mask_zeros)
template <class S>
static void cv_test_error(const Csc<S>& A, const S* W_Td, const S* H, int k, const SpeckledMask& mask, int threads,
                          S* sq_err, int64_t* n_test) {
    const int nt = eff_threads(threads); (void)nt;
    S total = 0;
    int64_t cnt = 0;
#pragma omp parallel for reduction(+ : total, cnt) num_threads(nt) schedule(dynamic, 64)
    for (int j = 0; j < A.cols; ++j) {
        const S* h = H + (size_t)j * k;
        if (mask.mask_zeros) {
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                if (!mask.is_holdout(A.i[t], j)) continue;
                const S* w = W_Td + (size_t)A.i[t] * k;
                S pred = 0;
                for (int f = 0; f < k; ++f) pred += w[f] * h[f];
                const S diff = A.x[t] - pred;
                total += diff * diff; ++cnt;
            }
        } else {
            int t = A.p[j];
            const int te = A.p[j + 1];
            for (int i = 0; i < A.rows; ++i) {
                S actual = 0;
                const bool nz = t < te && A.i[t] == i;
                if (nz) { actual = A.x[t]; ++t; }
                if (!mask.is_holdout(i, j)) continue;
                const S* w = W_Td + (size_t)i * k;
                S pred = 0;
                for (int f = 0; f < k; ++f) pred += w[f] * h[f];
                const S diff = actual - pred;
                total += diff * diff; ++cnt;
            }
        }
    }
    *sq_err = total; *n_test = cnt;
}

template <class S> struct CvResult {
    int iterations = 0, best_iter = 0; bool converged = false;
    S train_loss = 0, test_loss = 0, best_test_loss = 0, final_tol = 0;
    std::vector<S> train_hist, test_hist;
    std::vector<S> theta;                 // GP: theta_vec at exit (result.theta, fit_cv.hpp:1647)
};

template <class S>
static void gp_theta_update(const Csc<S>& A, const S* W_T, const S* H, const S* d, int k, const FitConfig<S>& cfg,
                            std::vector<S>& theta, const SpeckledMask* cv_mask = nullptr);

// The IRLS weight as the CV solves evaluate it: compute_irls_weight(residual, predicted, loss) with the DEFAULT observed = 0 and
// theta = 0 (cv_detail.hpp:154, :255) -- the dispersion estimates never reach the CV weights, and GP takes irls_weight_gp
// (not the KL weight of the non-CV fit).
template <class S> static inline S cv_irls_weight(const FitConfig<S>& cfg, S residual, S predicted) {
    const int lt = cfg.loss_type;
    S w_dist;
    if (lt == 0) w_dist = S(1);
    else if (lt == 4) w_dist = irls_weight_gp(S(0), predicted, S(0), S(1));
    else w_dist = irls_weight(lt, predicted, S(0), cfg.tweedie_power);
    if (cfg.robust_delta > S(0)) {
        const S sd_inv = std::sqrt(std::max(w_dist, static_cast<S>(1e-15)));
        return w_dist * robust_huber_modifier(residual * sd_inv, cfg.robust_delta);
    }
    return w_dist;
}

// nmf/cv_detail.hpp:101-186 irls_solve_col_cv (H side: D = A, F = W_T) / :200-292 irls_solve_row_cv (W side: D = A^T, F = H,
// transposed = the mask is asked in the coordinates of A).  Per column: the TRAIN entries (mask_zeros: the nonzeros that are not
// held out; otherwise every row that is not held out, zeros included); up to irls_max_iter passes of
//   G_w = sum_train w f f^T (+ the CV features: L2, graph, L21 -- all additive, passed in as G_add) + 1e-15 I,
//   b_w = sum_train (w a) f,  x <- cholesky_clip_col / cd_nnls_col_fixed(G_w, b_w, x, L1 inside, cd_maxit sweeps, no tolerance)
// started from the current column with b_w NOT residual-corrected for it (as in the MSE CV solve), until the largest relative
// change of x drops below irls_tol.
template <class S>
static void cv_irls_half_update(const Csc<S>& D, const S* F, S* X, int k, const SpeckledMask& mask, bool transposed, S L1, bool nonneg,
                                const FitConfig<S>& cfg, const S* G_add, int threads, const Csc<S>* umask = nullptr) {
    const int nt = eff_threads(threads); (void)nt;
    const int nrow = D.rows;
#pragma omp parallel num_threads(nt)
    {
        std::vector<S> bw(k), x(k), xo(k), Gw((size_t)k * k), L((size_t)k * k);
        std::vector<int> trow;
        std::vector<S> tval;
#pragma omp for schedule(dynamic, 16)
        for (int j = 0; j < D.cols; ++j) {
            trow.clear(); tval.clear();
            // user-masked rows are excluded rows like the test rows (fit_cv.hpp:491-501: gram_rows = test U masked -> is_test)
            auto held = [&](int r) {
                if (umask && std::binary_search(umask->i + umask->p[j], umask->i + umask->p[j + 1], r)) return true;
                return transposed ? mask.is_holdout(j, r) : mask.is_holdout(r, j);
            };
            if (mask.mask_zeros) {
                for (int t = D.p[j]; t < D.p[j + 1]; ++t)
                    if (!held(D.i[t])) { trow.push_back(D.i[t]); tval.push_back(D.x[t]); }
            } else {
                int t = D.p[j];
                const int te = D.p[j + 1];
                for (int r = 0; r < nrow; ++r) {
                    S val = 0;
                    if (t < te && D.i[t] == r) { val = D.x[t]; ++t; }
                    if (!held(r)) { trow.push_back(r); tval.push_back(val); }
                }
            }
            S* xj = X + (size_t)j * k;
            for (int i = 0; i < k; ++i) x[i] = xj[i];
            for (int it = 0; it < cfg.irls_max_iter; ++it) {
                std::fill(Gw.begin(), Gw.end(), S(0));
                std::fill(bw.begin(), bw.end(), S(0));
                for (size_t e = 0; e < trow.size(); ++e) {
                    const S* f = F + (size_t)trow[e] * k;
                    S predicted = 0;
                    for (int a = 0; a < k; ++a) predicted += f[a] * x[a];
                    const S residual = tval[e] - predicted;
                    const S w = cv_irls_weight(cfg, residual, predicted);
                    const S wv = w * tval[e];
                    for (int a = 0; a < k; ++a) {
                        bw[a] += f[a] * wv;
                        for (int bb = a; bb < k; ++bb) Gw[(size_t)bb * k + a] += w * f[a] * f[bb];      // G_w(a, bb), a <= bb
                    }
                }
                for (int a = 0; a < k; ++a)
                    for (int bb = a + 1; bb < k; ++bb) Gw[(size_t)a * k + bb] = Gw[(size_t)bb * k + a];   // G_w(bb, a) = G_w(a, bb)
                if (G_add) for (size_t e = 0; e < (size_t)k * k; ++e) Gw[e] += G_add[e];
                for (int a = 0; a < k; ++a) Gw[(size_t)a * k + a] += static_cast<S>(1e-15);
                xo = x;
                if (cfg.solver_mode == 1) {
                    if (L1 > 0) for (int i = 0; i < k; ++i) bw[i] -= L1;
                    llt_factor(Gw.data(), k, L.data());
                    for (int i = 0; i < k; ++i) x[i] = bw[i];
                    llt_solve(L.data(), k, x.data());
                    if (nonneg) for (int i = 0; i < k; ++i) x[i] = std::max(x[i], S(0));
                } else {
                    cd_nnls_col_fixed(Gw.data(), bw.data(), x.data(), k, L1, S(0), nonneg, cfg.cd_maxit, S(0), S(0));
                }
                S max_change = 0;
                for (int i = 0; i < k; ++i) {
                    const S rel = std::abs(x[i] - xo[i]) / (std::abs(xo[i]) + static_cast<S>(1e-12));
                    if (rel > max_change) max_change = rel;
                }
                if (max_change < cfg.irls_tol) break;
            }
            for (int i = 0; i < k; ++i) xj[i] = x[i];
        }
    }
}

// fit_cv.hpp:1377-1443  per-element loss of the non-MSE CV fit: compute_loss(value, prediction, loss, theta) summed separately over
// the training and the held-out entries (mask_zeros: nonzeros only; otherwise every entry); theta = GP's theta_vec, 0 otherwise.
template <class S>
static void cv_explicit_loss(const Csc<S>& A, const S* W_Td, const S* H, int k, const SpeckledMask& mask, const FitConfig<S>& cfg,
                             const S* theta, int threads, S* train_sum, int64_t* n_train, S* test_sum, int64_t* n_test,
                             const Csc<S>* umask = nullptr) {
    const int nt = eff_threads(threads); (void)nt;
    S tr = 0, te = 0;
    int64_t ntr = 0, nte = 0;
#pragma omp parallel for reduction(+ : tr, te, ntr, nte) num_threads(nt) schedule(dynamic, 16)
    for (int j = 0; j < A.cols; ++j) {
        const S* h = H + (size_t)j * k;
        auto term = [&](int i, S actual) {
            if (umask && std::binary_search(umask->i + umask->p[j], umask->i + umask->p[j + 1], i)) return;   // :1391, :1407, :1427
            const S* w = W_Td + (size_t)i * k;
            S pred = 0;
            for (int f = 0; f < k; ++f) pred += w[f] * h[f];
            const S th = cfg.loss_type == 4 ? theta[i] : S(0);
            const S lv = cfg.loss_type == 0 ? (actual - pred) * (actual - pred) : loss_contribution(cfg.loss_type, actual, pred, th, cfg.tweedie_power);
            if (mask.is_holdout(i, j)) { te += lv; ++nte; } else { tr += lv; ++ntr; }
        };
        if (mask.mask_zeros) {
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) term(A.i[t], A.x[t]);
        } else {
            int t = A.p[j];
            const int tend = A.p[j + 1];
            for (int i = 0; i < A.rows; ++i) {
                S actual = 0;
                if (t < tend && A.i[t] == i) { actual = A.x[t]; ++t; }
                term(i, actual);
            }
        }
    }
    *train_sum = tr; *n_train = ntr; *test_sum = te; *n_test = nte;
}

// nmf/fit_cv.hpp:123-1667  nmf_fit_cv, sparse / standard updates / no user mask / no zero inflation; MSE, or (cfg.loss_type in
// 4..8, or robust_delta > 0) the IRLS path: per-column weighted Grams, GP theta over the training entries, per-element losses.
// W_T (k x m) and H (k x n) hold the initial factors on entry; on exit W_T is normalised, H carries d ("absorb d
// into H", :1636-1638) and d is ALSO returned, exactly as the reference packages it.
template <class S>
static CvResult<S> nmf_fit_cv(const Csc<S>& A, const FitConfig<S>& cfg, double holdout_fraction, uint64_t cv_seed,
                              bool mask_zeros, int cv_patience, S* W_T, S* H, S* d) {
    const int m = A.rows, n = A.cols, k = cfg.k;
    for (int i = 0; i < k; ++i) d[i] = S(1);
    const SpeckledMask mask(holdout_fraction, cv_seed, mask_zeros);
    CscOwned<S> At_own = transpose_csc(A);
    const Csc<S> At = At_own.view();
    const int threads = eff_threads(cfg.threads);
    S trAtA = 0;
    for (int t = 0; t < A.p[n]; ++t) trAtA += A.x[t] * A.x[t];
    std::vector<S> G((size_t)k * k), G_H_saved((size_t)k * k), G_W_new((size_t)k * k), B_W_full((size_t)k * m), Wd((size_t)k * m);
    CvResult<S> res;
    S best_test = std::numeric_limits<S>::max(), prev_conv = std::numeric_limits<S>::max();
    int best_iter = 0, patience_count = 0;
    const bool irls = cfg.loss_type != 0 || cfg.robust_delta > S(0);                       // LossConfig::requires_irls()
    const bool is_gp = cfg.loss_type == 4;
    // user mask (:327-331): excluded rows of every half-update, skipped by the losses -- which are then the explicit per-element
    // ones for MSE too (:1377-1379 "requires_irls() || use_mask")
    const Csc<S>* um = cfg.has_mask ? &cfg.mask : nullptr;
    CscOwned<S> umT_own;
    Csc<S> umT_view{};
    if (um) { umT_own = transpose_csc(cfg.mask); umT_view = umT_own.view(); }
    const Csc<S>* umT = um ? &umT_view : nullptr;
    std::vector<S> theta_vec;                                                               // :195-202
    if (is_gp) theta_vec.assign(m, (cfg.dispersion_mode == 2 || cfg.dispersion_mode == 1) ? cfg.gp_theta_init : S(0));
    std::vector<S> G_add((size_t)k * k);
    for (int iter = 0; iter < cfg.max_iter; ++iter) {
        if (irls) {
            // ---- H update, IRLS (:446-456): the CV features enter every column's weighted Gram; they are additive, so they are
            // formed once on a zero matrix
            std::fill(G_add.begin(), G_add.end(), S(0));
            if (cfg.L2_H > 0) for (int i = 0; i < k; ++i) G_add[(size_t)i * k + i] += cfg.L2_H;
            if (cfg.has_graph_H) apply_graph_reg(G_add.data(), cfg.graph_H, H, k, cfg.graph_H_lambda);
            apply_L21(G_add.data(), H, k, (int64_t)n, cfg.L21_H);
            cv_irls_half_update(A, W_T, H, k, mask, false, cfg.L1_H, cfg.nonneg_H, cfg, G_add.data(), threads, um);
            if (cfg.ub_H > 0) apply_upper_bound(H, (size_t)k * n, cfg.ub_H);
            apply_angular_posthoc(H, k, (int64_t)n, cfg.angular_H);
            extract_scaling(H, k, n, d, cfg.norm_type);
            // ---- W update, IRLS (:670-689)
            std::fill(G_add.begin(), G_add.end(), S(0));
            if (cfg.L2_W > 0) for (int i = 0; i < k; ++i) G_add[(size_t)i * k + i] += cfg.L2_W;
            if (cfg.has_graph_W) apply_graph_reg(G_add.data(), cfg.graph_W, W_T, k, cfg.graph_W_lambda);
            apply_L21(G_add.data(), W_T, k, (int64_t)m, cfg.L21_W);
            cv_irls_half_update(At, H, W_T, k, mask, true, cfg.L1_W, cfg.nonneg_W, cfg, G_add.data(), threads, umT);
            if (cfg.ub_W > 0) apply_upper_bound(W_T, (size_t)k * m, cfg.ub_W);
            apply_angular_posthoc(W_T, k, (int64_t)m, cfg.angular_W);
            extract_scaling(W_T, k, m, d, cfg.norm_type);
            // ---- GP theta over the training entries (:866-961); the NB size and the Gamma-family phi are estimated by the reference
            // too (:966-1155) but reach neither the CV weights nor the CV losses nor the plugin's outputs
            if (is_gp && cfg.dispersion_mode != 0) gp_theta_update(A, W_T, H, d, k, cfg, theta_vec, &mask);
            // ---- losses (:1377-1443, :1546-1549)
            for (int i = 0; i < m; ++i) for (int f = 0; f < k; ++f) Wd[(size_t)i * k + f] = W_T[(size_t)i * k + f] * d[f];
            S tr = 0, te = 0; int64_t ntr = 0, nte = 0;
            cv_explicit_loss(A, Wd.data(), H, k, mask, cfg, theta_vec.data(), threads, &tr, &ntr, &te, &nte, um);
            const S train_loss = ntr > 0 ? tr / static_cast<S>(ntr) : S(0);
            const S test_loss = nte > 0 ? te / static_cast<S>(nte) : S(0);
            res.train_hist.push_back(train_loss); res.test_hist.push_back(test_loss);
            res.train_loss = train_loss; res.test_loss = test_loss;
            S rel = 0;
            if (iter > 0) rel = std::abs(prev_conv - test_loss) / (std::abs(prev_conv) + static_cast<S>(1e-15));
            if (test_loss < best_test) { best_test = test_loss; best_iter = iter; patience_count = 0; }
            else ++patience_count;
            if (cv_patience > 0 && patience_count >= cv_patience) { res.iterations = iter + 1; res.converged = false; break; }
            if (iter > 0) {
                res.final_tol = rel;
                if (rel < cfg.tol) { res.iterations = iter + 1; res.converged = true; break; }
            }
            prev_conv = test_loss;
            res.iterations = iter + 1;
            continue;
        }
        // ---- H update (:408-535)
        gram(W_T, k, m, G.data());
        for (int i = 0; i < k; ++i) G[(size_t)i * k + i] += static_cast<S>(1e-15);        // :410 (on top of gram's own eps)
        if (cfg.L2_H > 0) for (int i = 0; i < k; ++i) G[(size_t)i * k + i] += cfg.L2_H;   // apply_cv_features (variant_helpers.hpp:174-189)
        if (cfg.has_graph_H) apply_graph_reg(G.data(), cfg.graph_H, H, k, cfg.graph_H_lambda);
        apply_L21(G.data(), H, k, (int64_t)n, cfg.L21_H);
        cv_half_update(A, W_T, G.data(), H, k, mask, false, cfg.L1_H, cfg.nonneg_H, cfg.cd_maxit, cfg.solver_mode, threads, um);
        if (cfg.ub_H > 0) apply_upper_bound(H, (size_t)k * n, cfg.ub_H);
        apply_angular_posthoc(H, k, (int64_t)n, cfg.angular_H);
        extract_scaling(H, k, n, d, cfg.norm_type);                                        // :538-550
        // ---- W update (:555-860)
        gram(H, k, n, G.data());
        G_H_saved = G;                                                                     // :574-576 (with gram's eps)
        for (int i = 0; i < k; ++i) G[(size_t)i * k + i] += static_cast<S>(1e-15);        // :578
        if (cfg.L2_W > 0) for (int i = 0; i < k; ++i) G[(size_t)i * k + i] += cfg.L2_W;
        if (cfg.has_graph_W) apply_graph_reg(G.data(), cfg.graph_W, W_T, k, cfg.graph_W_lambda);
        apply_L21(G.data(), W_T, k, (int64_t)m, cfg.L21_W);
        // B_W_full.col(i) = sum_j A(i,j) H(:,j) over ALL nonzeros (train + test), :617-655
        for (int i = 0; i < m; ++i) {
            S* bw = B_W_full.data() + (size_t)i * k;
            for (int f = 0; f < k; ++f) bw[f] = 0;
            for (int t = At.p[i]; t < At.p[i + 1]; ++t) {
                const S a = At.x[t]; const S* hc = H + (size_t)At.i[t] * k;
                for (int f = 0; f < k; ++f) bw[f] += a * hc[f];
            }
        }
        cv_half_update(At, H, G.data(), W_T, k, mask, true, cfg.L1_W, cfg.nonneg_W, cfg.cd_maxit, cfg.solver_mode, threads, umT);
        if (cfg.ub_W > 0) apply_upper_bound(W_T, (size_t)k * m, cfg.ub_W);
        apply_angular_posthoc(W_T, k, (int64_t)m, cfg.angular_W);
        extract_scaling(W_T, k, m, d, cfg.norm_type);                                      // :848-859
        // ---- loss (:1345-1550): every iteration (cv_patience > 0 / history)
        for (int i = 0; i < m; ++i) for (int f = 0; f < k; ++f) Wd[(size_t)i * k + f] = W_T[(size_t)i * k + f] * d[f];
        if (um) {          // :1377-1443 with loss MSE: explicit train and test sums over the entries that are not user-masked
            S tr = 0, te = 0; int64_t ntr = 0, nte = 0;
            cv_explicit_loss(A, Wd.data(), H, k, mask, cfg, theta_vec.data(), threads, &tr, &ntr, &te, &nte, um);
            const S train_loss = ntr > 0 ? tr / static_cast<S>(ntr) : S(0);
            const S test_loss = nte > 0 ? te / static_cast<S>(nte) : S(0);
            res.train_hist.push_back(train_loss); res.test_hist.push_back(test_loss);
            res.train_loss = train_loss; res.test_loss = test_loss;
            S rel = 0;
            if (iter > 0) rel = std::abs(prev_conv - test_loss) / (std::abs(prev_conv) + static_cast<S>(1e-15));
            if (test_loss < best_test) { best_test = test_loss; best_iter = iter; patience_count = 0; }
            else ++patience_count;
            if (cv_patience > 0 && patience_count >= cv_patience) { res.iterations = iter + 1; res.converged = false; break; }
            if (iter > 0) {
                res.final_tol = rel;
                if (rel < cfg.tol) { res.iterations = iter + 1; res.converged = true; break; }
            }
            prev_conv = test_loss;
            res.iterations = iter + 1;
            continue;
        }
        S test_sq = 0; int64_t n_test = 0;
        cv_test_error(A, Wd.data(), H, k, mask, threads, &test_sq, &n_test);
        S cross = 0;
        for (int r = 0; r < k; ++r) {
            S dot = 0;
            for (int i = 0; i < m; ++i) dot += W_T[(size_t)i * k + r] * B_W_full[(size_t)i * k + r];
            cross += d[r] * dot;
        }
        gram(W_T, k, m, G_W_new.data());
        S recon = 0;
        for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) recon += d[r] * d[c] * G_W_new[(size_t)c * k + r] * G_H_saved[(size_t)c * k + r];
        const S total_sq = std::max(trAtA - S(2) * cross + recon, S(0));
        const S train_sq = std::max(total_sq - test_sq, S(0));
        const int64_t total_entries = mask_zeros ? (int64_t)A.p[n] : (int64_t)m * n;
        const int64_t n_train = total_entries - n_test;
        const S train_loss = n_train > 0 ? train_sq / static_cast<S>(n_train) : S(0);
        const S test_loss = n_test > 0 ? test_sq / static_cast<S>(n_test) : S(0);
        res.train_hist.push_back(train_loss); res.test_hist.push_back(test_loss);
        res.train_loss = train_loss; res.test_loss = test_loss;
        S rel = 0;
        if (iter > 0) rel = std::abs(prev_conv - test_loss) / (std::abs(prev_conv) + static_cast<S>(1e-15));
        if (test_loss < best_test) { best_test = test_loss; best_iter = iter; patience_count = 0; }
        else ++patience_count;
        if (cv_patience > 0 && patience_count >= cv_patience) { res.iterations = iter + 1; res.converged = false; break; }
        if (iter > 0) {
            res.final_tol = rel;
            if (rel < cfg.tol) { res.iterations = iter + 1; res.converged = true; break; }
        }
        prev_conv = test_loss;
        res.iterations = iter + 1;
    }
    res.best_test_loss = best_test; res.best_iter = best_iter;
    res.theta = theta_vec;
    for (int j = 0; j < n; ++j) for (int f = 0; f < k; ++f) H[(size_t)j * k + f] *= d[f];   // :1636-1638
    return res;
}

// nmf/masked_nnls.hpp:250-282   masked_loss (sparse A: unmasked NONZEROS only).  Per element compute_loss(a, pred, loss_config)
// (:277) -- math/loss.hpp:512-536 with its DEFAULT theta = 0: under a mask the reference evaluates a GP / NB term without the fitted
// dispersion and never applies the robust modifier (compute_loss, not compute_robust_loss); restated as it is.
template <class S>
static S masked_loss(const Csc<S>& A, const S* W_Td, const S* H, const Csc<S>& mask, int k,
                     int threads, int loss_type = 0, S power = S(1.5)) {
    const int nt = eff_threads(threads); (void)nt;
    S total = 0;
#pragma omp parallel num_threads(nt) reduction(+ : total)
    {
        std::vector<char> is_masked(A.rows, 0);
#pragma omp for schedule(dynamic, 64)
        for (int j = 0; j < A.cols; ++j) {
            for (int t = mask.p[j]; t < mask.p[j + 1]; ++t)
                if (mask.x[t] != 0) is_masked[mask.i[t]] = 1;
            const S* h = H + (size_t)j * k;
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                const int i = A.i[t];
                if (is_masked[i]) continue;
                const S* w = W_Td + (size_t)i * k;
                S pred = 0;
                for (int f = 0; f < k; ++f) pred += w[f] * h[f];
                if (loss_type == 0) {
                    const S diff = A.x[t] - pred;
                    total += diff * diff;
                } else total += loss_contribution(loss_type, A.x[t], pred, S(0), power);
            }
            for (int t = mask.p[j]; t < mask.p[j + 1]; ++t) is_masked[mask.i[t]] = 0;
        }
    }
    return total;
}

// primitives/cpu/nnls_batch_irls.hpp:465-520 + :202-329   NB-weighted per-column IRLS
// theta_row: indexed by the ROW of `A` (nonzero's row); theta_col: indexed by the column.
template <class S>
static void nnls_batch_irls_sparse_nb(const Csc<S>& A, const S* F, const S* G_base, S* X, int k,
                                      S L1, S L2, bool nonneg, int cd_maxit, int irls_max_iter,
                                      S irls_tol, int threads, const S* theta_row,
                                      const S* theta_col, int loss_type = 5, S power = S(1.5), S robust = S(0), bool dense = false) {
    // dense: irls_nnls_col_dense (nnls_batch_irls.hpp:376-450; batch :525-555, H.setZero() there too) -- every row of the column is
    // stored and weighted, and G_w = sum_i w_i f_i f_i^T is formed from nothing (no G_base, hence no eps on its diagonal)
    const int nt = eff_threads(threads); (void)nt;
    std::fill(X, X + (size_t)k * A.cols, S(0));   // H.setZero(): no warm start across ALS iters
#pragma omp parallel num_threads(nt)
    {
        std::vector<S> Gw((size_t)k * k), bw(k), bc(k), xold(k);
#pragma omp for schedule(dynamic)
        for (int j = 0; j < A.cols; ++j) {
            S* x = X + (size_t)j * k;
            for (int it = 0; it < irls_max_iter; ++it) {
                if (dense) std::fill(Gw.begin(), Gw.end(), S(0));
                else std::memcpy(Gw.data(), G_base, sizeof(S) * k * k);
                for (int f = 0; f < k; ++f) bw[f] = 0;
                for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                    const int row = A.i[t];
                    const S* fr = F + (size_t)row * k;
                    S recon = 0;
                    for (int f = 0; f < k; ++f) recon += fr[f] * x[f];
                    const S th = theta_col ? theta_col[j] : (theta_row ? theta_row[row] : S(0));
                    const S w = compute_irls_weight(loss_type, A.x[t] - recon, recon, th, power, robust);
                    const S dw = dense ? w : w - S(1);
                    const S wv = w * A.x[t];
                    // G_w += dw * f f^T  (reference: W_nnz_scaled * W_block^T)
                    for (int c = 0; c < k; ++c) {
                        const S fc = fr[c];
                        S* g = Gw.data() + (size_t)c * k;
                        for (int a2 = 0; a2 < k; ++a2) g[a2] += (fr[a2] * dw) * fc;
                    }
                    for (int f = 0; f < k; ++f) bw[f] += fr[f] * wv;
                }
                if (L2 > 0) for (int i = 0; i < k; ++i) Gw[(size_t)i * k + i] += L2;
                for (int i = 0; i < k; ++i) { xold[i] = x[i]; bc[i] = bw[i]; }
                for (int c = 0; c < k; ++c) {
                    const S xc = xold[c]; const S* gc = Gw.data() + (size_t)c * k;
                    for (int r = 0; r < k; ++r) bc[r] -= gc[r] * xc;
                }
                cd_nnls_col_fixed(Gw.data(), bc.data(), x, k, L1, S(0), nonneg, cd_maxit, S(0), S(0));
                S max_change = 0;
                for (int i = 0; i < k; ++i) {
                    const S rel = std::abs(x[i] - xold[i]) / (std::abs(xold[i]) + static_cast<S>(1e-12));
                    if (rel > max_change) max_change = rel;
                }
                if (max_change < irls_tol) break;
            }
        }
    }
}

// nmf/fit_cpu.hpp:1094-1265   NB size update (PER_ROW / GLOBAL / PER_COL), sparse branch
template <class S>
static void nb_size_update(const Csc<S>& A, const S* W_T, const S* H, const S* d, int k,
                           const FitConfig<S>& cfg, std::vector<S>& nb_size) {
    const int m = A.rows, n = A.cols;
    std::vector<S> Wd((size_t)k * m);
    for (int i = 0; i < m; ++i) for (int f = 0; f < k; ++f) Wd[(size_t)i * k + f] = W_T[(size_t)i * k + f] * d[f];
    const double r_min = static_cast<double>(cfg.nb_size_min), r_max = static_cast<double>(cfg.nb_size_max);
    if (cfg.dense_input) {
        // dense branches (:1137-1148 PER_COL, :1226-1238 PER_ROW / GLOBAL): EVERY entry with its prediction floored at 1e-10, no
        // Gram-trick totals; then the same moment estimator and clamps (:1151-1160, :1241-1262)
        const bool pc = cfg.dispersion_mode == 3;
        const int len = pc ? n : m;
        std::vector<double> a_mu2(len, 0.0), a_exc(len, 0.0);
        for (int j = 0; j < n; ++j) {
            const S* h = H + (size_t)j * k;
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                const S* w = Wd.data() + (size_t)A.i[t] * k;
                S dot = 0;
                for (int f = 0; f < k; ++f) dot += w[f] * h[f];
                const double y = static_cast<double>(A.x[t]);
                const double mu = std::max(static_cast<double>(dot), 1e-10);
                const double resid = y - mu;
                const int o = pc ? j : A.i[t];
                a_mu2[o] += mu * mu;
                a_exc[o] += resid * resid - mu;
            }
        }
        for (int o = 0; o < len; ++o) {
            if (a_exc[o] > 1e-10 && a_mu2[o] > 1e-10) {
                double r_new = a_mu2[o] / a_exc[o];
                r_new = std::max(r_min, std::min(r_new, r_max));
                if (std::isfinite(r_new)) nb_size[o] = static_cast<S>(r_new);
            } else {
                nb_size[o] = static_cast<S>(r_max);
            }
        }
        if (cfg.dispersion_mode == 1) {
            std::vector<S> r_vals(nb_size.begin(), nb_size.begin() + m);
            std::nth_element(r_vals.begin(), r_vals.begin() + m / 2, r_vals.end());
            const S med = r_vals[m / 2];
            std::fill(nb_size.begin(), nb_size.end(), med);
        }
        return;
    }
    if (cfg.dispersion_mode == 3) {
        // :1103-1162 PER_COL: per column j the nonzero sums, then the zeros' share from the totals over ALL rows, which the
        // reference forms by a direct loop over the m rows (Scalar dot, summed in double, NOT clamped)
        for (int j = 0; j < n; ++j) {
            const S* h = H + (size_t)j * k;
            double s_mu2 = 0.0, s_exc = 0.0, nz_mu = 0.0, nz_mu2 = 0.0;
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                const S* w = Wd.data() + (size_t)A.i[t] * k;
                S dot = 0;
                for (int f = 0; f < k; ++f) dot += w[f] * h[f];
                const double y = static_cast<double>(A.x[t]);
                const double mu = std::max(static_cast<double>(dot), 1e-10);
                const double resid = y - mu;
                s_mu2 += mu * mu;                                   // :1110
                s_exc += resid * resid - mu;                        // :1111
                nz_mu += mu; nz_mu2 += mu * mu;                     // :1127-1131 (second pass over the same nonzeros)
            }
            double tot_mu = 0.0, tot_mu2 = 0.0;                     // :1117-1122
            for (int i = 0; i < m; ++i) {
                const S* w = Wd.data() + (size_t)i * k;
                S dot = 0;
                for (int f = 0; f < k; ++f) dot += w[f] * h[f];
                const double mu = static_cast<double>(dot);
                tot_mu += mu; tot_mu2 += mu * mu;
            }
            s_mu2 += (tot_mu2 - nz_mu2);                            // :1134
            s_exc += (tot_mu2 - nz_mu2) - (tot_mu - nz_mu);         // :1135
            if (s_exc > 1e-10 && s_mu2 > 1e-10) {                   // :1151-1160
                double r_new = s_mu2 / s_exc;
                r_new = std::max(r_min, std::min(r_new, r_max));
                if (std::isfinite(r_new)) nb_size[j] = static_cast<S>(r_new);
            } else {
                nb_size[j] = static_cast<S>(r_max);
            }
        }
        return;
    }
    std::vector<double> s_mu2(m, 0.0), s_res2(m, 0.0);
    for (int j = 0; j < n; ++j) {
        const S* h = H + (size_t)j * k;
        for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
            const int i = A.i[t];
            const S* w = Wd.data() + (size_t)i * k;
            S dot = 0;
            for (int f = 0; f < k; ++f) dot += w[f] * h[f];
            const double y = static_cast<double>(A.x[t]);
            const double mu = std::max(static_cast<double>(dot), 1e-10);
            const double resid = y - mu;
            s_mu2[i] += mu * mu;
            s_res2[i] += resid * resid;
        }
    }
    std::vector<S> h_rs(k, S(0));
    for (int j = 0; j < n; ++j) { const S* h = H + (size_t)j * k; for (int f = 0; f < k; ++f) h_rs[f] += h[f]; }
    std::vector<S> GH((size_t)k * k);
    gram(H, k, n, GH.data());
    for (int i = 0; i < m; ++i) {
        const S* w = Wd.data() + (size_t)i * k;
        S tm = 0;
        for (int f = 0; f < k; ++f) tm += w[f] * h_rs[f];
        const double total_mu = static_cast<double>(tm);
        double total_mu_sq = 0.0;
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b)
                total_mu_sq += static_cast<double>(w[a]) * static_cast<double>(GH[(size_t)b * k + a]) * static_cast<double>(w[b]);
        const double total_resid_sq = s_res2[i] + (total_mu_sq - s_mu2[i]);
        const double excess = total_resid_sq - total_mu;
        if (excess > 1e-10 && total_mu_sq > 1e-10) {
            double r_new = total_mu_sq / excess;
            r_new = std::max(r_min, std::min(r_new, r_max));
            if (std::isfinite(r_new)) nb_size[i] = static_cast<S>(r_new);
        } else {
            nb_size[i] = static_cast<S>(r_max);
        }
    }
    if (cfg.dispersion_mode == 1) {  // GLOBAL: median via nth_element at m/2
        std::vector<S> r_vals(nb_size.begin(), nb_size.begin() + m);
        std::nth_element(r_vals.begin(), r_vals.begin() + m / 2, r_vals.end());
        const S med = r_vals[m / 2];
        std::fill(nb_size.begin(), nb_size.end(), med);
    }
}

// nmf/fit_cpu.hpp:914-1008   GP theta: auxiliary-function (MM) update, PER_ROW / GLOBAL, sparse branch.  Five inner MM
// passes over the cached (row, y, s) of the nonzeros; everything in fp64 except s (a Scalar dot) and theta (stored Scalar).
template <class S>
static void gp_theta_update(const Csc<S>& A, const S* W_T, const S* H, const S* d, int k, const FitConfig<S>& cfg,
                            std::vector<S>& theta, const SpeckledMask* cv_mask) {
    // cv_mask: nmf/fit_cv.hpp:866-961 -- the same update over the TRAINING entries only: held-out nonzeros are skipped and the
    // held-out pairs' predictions (zeros included) leave sum_s
    const int m = A.rows, n = A.cols;
    std::vector<S> Wd((size_t)k * m);
    for (int i = 0; i < m; ++i) for (int f = 0; f < k; ++f) Wd[(size_t)i * k + f] = W_T[(size_t)i * k + f] * d[f];
    if (cfg.dispersion_mode == 3 && !cv_mask) {
        // :1009-1083 PER_COL: the same MM update with every accumulator indexed by the column; sum_s of column j is the direct
        // sum over all m rows of the Scalar dot (:1021-1026)
        std::vector<double> sy(n, 0.0), ss(n, 0.0);
        std::vector<int> nn(n, 0);
        struct NzC { int col; double y, s; };
        std::vector<NzC> cc; cc.reserve((size_t)A.p[n]);
        for (int j = 0; j < n; ++j) {
            const S* h = H + (size_t)j * k;
            double tot = 0;
            for (int i = 0; i < m; ++i) {
                const S* w = Wd.data() + (size_t)i * k;
                S dot = 0; for (int f = 0; f < k; ++f) dot += w[f] * h[f];
                tot += static_cast<double>(dot);
            }
            ss[j] = cfg.dense_input ? 0.0 : tot;                         // dense (:1041-1053): sum_s_col(j) += s, the floored prediction
            for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
                const S* w = Wd.data() + (size_t)A.i[t] * k;
                S dot = 0; for (int f = 0; f < k; ++f) dot += w[f] * h[f];
                const double y = static_cast<double>(A.x[t]);
                const double sv = std::max(static_cast<double>(dot), 1e-10);
                sy[j] += y;
                if (cfg.dense_input) ss[j] += sv;
                if (y >= 1.0) nn[j]++;
                cc.push_back({j, y, sv});
            }
        }
        const double capc = static_cast<double>(cfg.gp_theta_max);
        std::vector<double> al(n), ga(n);
        for (int mm = 0; mm < 5; ++mm) {
            std::fill(al.begin(), al.end(), 0.0); std::fill(ga.begin(), ga.end(), 0.0);
            for (const NzC& z : cc)
                if (z.y >= 1.0) {
                    const double th = static_cast<double>(theta[z.col]);
                    const double denom = std::max(z.s + th * z.y, 1e-10);
                    const double eta1 = z.s / denom;
                    al[z.col] += (z.y - 1.0) * eta1;
                    ga[z.col] += (z.y - 1.0) * (1.0 - eta1);
                }
            for (int j = 0; j < n; ++j) {
                const double a = al[j] + static_cast<double>(nn[j]);
                const double b = (sy[j] - ss[j]) - ga[j] + a;
                if (a > 1e-15) {
                    const double disc = b * b + 4.0 * a * ga[j];
                    if (disc > 0.0 && std::isfinite(disc)) {
                        const double nt = (-b + std::sqrt(disc)) / (2.0 * a);
                        if (std::isfinite(nt) && nt >= 0.0) theta[j] = static_cast<S>(std::min(nt, capc));
                    }
                }
            }
        }
        return;
    }
    std::vector<double> sum_y(m, 0.0), sum_s(m, 0.0);
    std::vector<int> n_nz(m, 0);
    std::vector<S> h_rs(k, S(0));                                                                    // :935
    for (int j = 0; j < n; ++j) { const S* h = H + (size_t)j * k; for (int f = 0; f < k; ++f) h_rs[f] += h[f]; }
    for (int i = 0; i < m; ++i) {                                                                    // :936-938
        S t = 0; const S* w = Wd.data() + (size_t)i * k;
        for (int f = 0; f < k; ++f) t += w[f] * h_rs[f];
        sum_s[i] = cfg.dense_input ? 0.0 : static_cast<double>(t);    // dense (:953-968): sum_s_d(i) += s per entry, floored
    }
    if (cv_mask)                                                                                     // fit_cv.hpp:886-893
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < m; ++i)
                if (cv_mask->is_holdout(i, j)) {
                    S t = 0; const S* w = Wd.data() + (size_t)i * k; const S* h = H + (size_t)j * k;
                    for (int f = 0; f < k; ++f) t += w[f] * h[f];
                    sum_s[i] -= static_cast<double>(t);
                }
    struct Nz { int row; double y, s; };
    std::vector<Nz> cache; cache.reserve((size_t)A.p[n]);
    for (int j = 0; j < n; ++j) {                                                                    // :941-952
        const S* h = H + (size_t)j * k;
        for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
            const int i = A.i[t]; const S* w = Wd.data() + (size_t)i * k;
            if (cv_mask && cv_mask->is_holdout(i, j)) continue;                                       // fit_cv.hpp:899
            S dot = 0; for (int f = 0; f < k; ++f) dot += w[f] * h[f];
            const double y = static_cast<double>(A.x[t]);
            const double sv = std::max(static_cast<double>(dot), 1e-10);
            sum_y[i] += y;
            if (cfg.dense_input) sum_s[i] += sv;
            if (y >= 1.0) n_nz[i]++;
            cache.push_back({i, y, sv});
        }
    }
    const double cap = static_cast<double>(cfg.gp_theta_max);
    std::vector<double> alpha(m), gamma(m);
    for (int mm = 0; mm < 5; ++mm) {                                                                 // :972-1001
        std::fill(alpha.begin(), alpha.end(), 0.0); std::fill(gamma.begin(), gamma.end(), 0.0);
        for (const Nz& z : cache)
            if (z.y >= 1.0) {
                const double th = static_cast<double>(theta[z.row]);
                const double denom = std::max(z.s + th * z.y, 1e-10);
                const double eta1 = z.s / denom;
                alpha[z.row] += (z.y - 1.0) * eta1;
                gamma[z.row] += (z.y - 1.0) * (1.0 - eta1);
            }
        for (int i = 0; i < m; ++i) {
            const double a = alpha[i] + static_cast<double>(n_nz[i]);
            const double b = (sum_y[i] - sum_s[i]) - gamma[i] + a;
            if (a > 1e-15) {
                const double disc = b * b + 4.0 * a * gamma[i];
                if (disc > 0.0 && std::isfinite(disc)) {
                    const double nt = (-b + std::sqrt(disc)) / (2.0 * a);
                    if (std::isfinite(nt) && nt >= 0.0) theta[i] = static_cast<S>(std::min(nt, cap));
                }
            }
        }
    }
    if (cfg.dispersion_mode == 1) {                                                                  // :1005-1008 GLOBAL: mean
        S acc = 0; for (int i = 0; i < m; ++i) acc += theta[i];
        std::fill(theta.begin(), theta.end(), acc / static_cast<S>(m));
    }
}

// nmf/fit_cpu.hpp:1561-1670   Gamma / inverse Gaussian / Tweedie dispersion phi: Pearson method of moments over the
// positive nonzeros of each row, PER_ROW / GLOBAL (median), sparse branch.  Diagnostic: the W/H updates do not use it.
template <class S>
static void phi_update(const Csc<S>& A, const S* W_T, const S* H, const S* d, int k, const FitConfig<S>& cfg,
                       std::vector<S>& phi) {
    const int m = A.rows, n = A.cols;
    std::vector<S> Wd((size_t)k * m);
    for (int i = 0; i < m; ++i) for (int f = 0; f < k; ++f) Wd[(size_t)i * k + f] = W_T[(size_t)i * k + f] * d[f];
    const double var_power = cfg.loss_type == 8 ? static_cast<double>(cfg.tweedie_power) : (cfg.loss_type == 6 ? 2.0 : 3.0);
    const double phi_min = static_cast<double>(cfg.gamma_phi_min), phi_max = static_cast<double>(cfg.gamma_phi_max);
    const bool per_col = cfg.dispersion_mode == 3;                 // :1570-1611: the same sums indexed by the column
    const int len = per_col ? n : m;
    std::vector<double> sum_p(len, 0.0);
    std::vector<int> cnt(len, 0);
    for (int j = 0; j < n; ++j) {
        const S* h = H + (size_t)j * k;
        for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
            const int i = A.i[t];
            const double y = static_cast<double>(A.x[t]);
            if (y <= 0.0) continue;
            const S* w = Wd.data() + (size_t)i * k;
            S dot = 0; for (int f = 0; f < k; ++f) dot += w[f] * h[f];
            const double mu = std::max(static_cast<double>(dot), 1e-10);
            const double resid = y - mu;
            const double v_mu = std::pow(mu, var_power);
            sum_p[per_col ? j : i] += (resid * resid) / std::max(v_mu, 1e-20);
            cnt[per_col ? j : i]++;
        }
    }
    for (int i = 0; i < len; ++i)
        if (cnt[i] > 0) {
            double pn = sum_p[i] / static_cast<double>(cnt[i]);
            pn = std::max(phi_min, std::min(pn, phi_max));
            if (std::isfinite(pn)) phi[i] = static_cast<S>(pn);
        }
    if (cfg.dispersion_mode == 1) {                                                                  // :1664-1669 GLOBAL: median
        std::vector<S> v(phi.begin(), phi.begin() + m);
        std::nth_element(v.begin(), v.begin() + m / 2, v.end());
        std::fill(phi.begin(), phi.end(), v[m / 2]);
    }
}

// nmf/explicit_loss.hpp:53-77   NB NLL over NONZEROS only (per-row theta)
template <class S>
static S explicit_loss_sparse_nb(const Csc<S>& A, const S* W_Td, const S* H, int k,
                                 const S* theta_row, int threads, int loss_type = 5, S power = S(1.5), S robust = S(0),
                                 bool theta_is_per_col = false) {                                      // explicit_loss.hpp:59, :70-71
    const int nt = eff_threads(threads); (void)nt;
    S total = 0;
#pragma omp parallel for reduction(+ : total) num_threads(nt) schedule(dynamic, 64)
    for (int j = 0; j < A.cols; ++j) {
        const S* h = H + (size_t)j * k;
        for (int t = A.p[j]; t < A.p[j + 1]; ++t) {
            const S* w = W_Td + (size_t)A.i[t] * k;
            S pred = 0;
            for (int f = 0; f < k; ++f) pred += w[f] * h[f];
            total += compute_robust_loss(loss_type, A.x[t], pred, theta_row ? theta_row[theta_is_per_col ? j : A.i[t]] : S(0), power, robust);
        }
    }
    return total;
}

// ---------------------------------------------------------------------------
// nmf/fit_cpu.hpp:171-1855   nmf_fit<CPU>, standard (non-projective, non-symmetric) variant
// W_T (k x m) and H (k x n) hold the initial factors on entry (the harness builds them:
// fit_cpu.hpp:195-207 / nmf_init.hpp:166-182) and the result on exit (W_T not transposed).
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Target regularisation -- nmf/variant_helpers.hpp:107-146 (the `fc.target` block of apply_features).
// PROJ_ADV uses Eigen::SelfAdjointEigenSolver in the reference (third-party, absent here): its published algorithm is a
// symmetric eigen-decomposition G = V L V^T; the clipped matrix V max(L, eps) V^T is a function of G alone, so the cyclic
// Jacobi iteration below reproduces it to rounding (pinned against numpy.linalg.eigh in tests/test_oracle.py).
// ---------------------------------------------------------------------------
template <class S> void proj_adv_gram(S* G, const S* TG, int k, S abs_lambda) {
    S trG = 0, trT = 0;
    for (int i = 0; i < k; ++i) { trG += G[(size_t)i * k + i]; trT += TG[(size_t)i * k + i]; }          // :123-124
    const S scale = trT > static_cast<S>(1e-10) ? trG / trT : static_cast<S>(0);                      // :125-126
    for (size_t e = 0; e < (size_t)k * k; ++e) G[e] -= abs_lambda * scale * TG[e];                    // :127
    // eigen-decomposition in double (Jacobi rotations on the upper triangle, eigenvectors accumulated in V)
    std::vector<double> A((size_t)k * k), V((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) A[(size_t)j * k + i] = 0.5 * ((double)G[(size_t)j * k + i] + (double)G[(size_t)i * k + j]);
    for (int i = 0; i < k; ++i) V[(size_t)i * k + i] = 1.0;
    for (int sweep = 0; sweep < 200; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < k; ++i) { diag += A[(size_t)i * k + i] * A[(size_t)i * k + i]; for (int j = i + 1; j < k; ++j) off += A[(size_t)j * k + i] * A[(size_t)j * k + i]; }
        if (off <= 1e-32 * (diag + off) || off == 0) break;
        for (int p = 0; p + 1 < k; ++p)
            for (int q = p + 1; q < k; ++q) {
                const double apq = A[(size_t)q * k + p];
                if (apq == 0.0) continue;
                const double tau = (A[(size_t)q * k + q] - A[(size_t)p * k + p]) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::abs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = t * c;
                for (int r = 0; r < k; ++r) {          // A <- A J (columns p, q)
                    const double x = A[(size_t)p * k + r], y = A[(size_t)q * k + r];
                    A[(size_t)p * k + r] = c * x - sn * y; A[(size_t)q * k + r] = sn * x + c * y;
                }
                for (int r = 0; r < k; ++r) {          // A <- J^T A (rows p, q)
                    const double x = A[(size_t)r * k + p], y = A[(size_t)r * k + q];
                    A[(size_t)r * k + p] = c * x - sn * y; A[(size_t)r * k + q] = sn * x + c * y;
                }
                for (int r = 0; r < k; ++r) {          // V <- V J
                    const double x = V[(size_t)p * k + r], y = V[(size_t)q * k + r];
                    V[(size_t)p * k + r] = c * x - sn * y; V[(size_t)q * k + r] = sn * x + c * y;
                }
            }
    }
    constexpr double eps = 1e-8;                                                                       // :133
    bool has_neg = false;
    std::vector<double> ev(k);
    for (int i = 0; i < k; ++i) { ev[i] = A[(size_t)i * k + i]; if (ev[i] < eps) { ev[i] = eps; has_neg = true; } }   // :135-140
    if (!has_neg) return;                                                                              // :141
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
            double s = 0;
            for (int e = 0; e < k; ++e) s += V[(size_t)e * k + i] * ev[e] * V[(size_t)e * k + j];
            G[(size_t)j * k + i] = static_cast<S>(s);                                                  // :142-143
        }
}
template <class S> void apply_target(S* G, S* B, const S* target, const S* target_gram, int k, int64_t ncols, S lambda) {
    if (!target || lambda == 0) return;                                                                // :105
    if (lambda > 0) {                                                                                  // :106-110 enrichment
        for (int i = 0; i < k; ++i) G[(size_t)i * k + i] += lambda;
        for (int64_t e = 0; e < (int64_t)k * ncols; ++e) B[e] += lambda * target[e];
    } else {
        proj_adv_gram(G, target_gram, k, static_cast<S>(-lambda));                                     // :111-145 (B untouched)
    }
}
template void proj_adv_gram<float>(float*, const float*, int, float);
template void proj_adv_gram<double>(double*, const double*, int, double);
template void apply_target<float>(float*, float*, const float*, const float*, int, int64_t, float);
template void apply_target<double>(double*, double*, const double*, const double*, int, int64_t, double);


template <class S>
FitResult<S> nmf_fit(const Csc<S>& A, const FitConfig<S>& cfg, S* W_T, S* H, S* d) {
    const int m = A.rows, n = A.cols, k = cfg.k;
    FitResult<S> res;
    for (int i = 0; i < k; ++i) d[i] = 1;

    const S trAtA = trace_AtA(A);                 // fit_cpu.hpp:224
    CscOwned<S> At_own = transpose_csc(A);        // fit_cpu.hpp:251-253
    const Csc<S> At = At_own.view();
    const int threads = eff_threads(cfg.threads);

    CscOwned<S> maskT_own;
    if (cfg.has_mask) maskT_own = transpose_csc(cfg.mask);   // fit_cpu.hpp:276-280
    const bool is_nb = cfg.loss_type == 5;
    const bool is_gp = cfg.loss_type == 4;                    // theta: gp_theta_init (PER_ROW/GLOBAL) or 0 (NONE) (fit_cpu.hpp:297-307)
    const bool is_pow = cfg.loss_type == 6 || cfg.loss_type == 7 || cfg.loss_type == 8;   // phi: gamma_phi_init or 1 (:337-347)
    const bool irls = is_nb || is_gp || is_pow || cfg.robust_delta > 0;   // requires_irls() (math/loss.hpp:106-108)
    std::vector<S> nb_size;
    const bool per_col = cfg.dispersion_mode == 3;            // DispersionMode::PER_COL: one value per COLUMN of A (:300-301, :319-320, :341-342)
    const size_t dlen = per_col ? (size_t)n : (size_t)m;
    if (is_nb)                                                // fit_cpu.hpp:316-328 (PER_ROW/GLOBAL/NONE/PER_COL)
        nb_size.assign(dlen, cfg.dispersion_mode == 0 ? cfg.nb_size_max : cfg.nb_size_init);
    if (is_gp) nb_size.assign(dlen, cfg.dispersion_mode == 0 ? S(0) : cfg.gp_theta_init);
    if (is_pow) nb_size.assign(dlen, cfg.dispersion_mode == 0 ? S(1) : cfg.gamma_phi_init);
    if (cfg.loss_type == 0) nb_size.assign(m, S(0));          // robust MSE: no dispersion; the IRLS of GP / power losses gets no theta

    std::vector<S> G((size_t)k * k), G_saved((size_t)k * k), G_wt((size_t)k * k);
    // target regularisation: nmf/fit.hpp:259-271 pre-computes T T^T / ncols for PROJ_ADV; a target forces the standard path
    // (fit_cpu.hpp:430-433 can_fuse = ... && !config.has_target())
    const bool tgt_H = cfg.target_H && cfg.target_lambda_H != 0, tgt_W = cfg.target_W && cfg.target_lambda_W != 0;
    const bool unfused = cfg.unfused || tgt_H || tgt_W;
    std::vector<S> TG_H, TG_W;
    auto target_gram = [&](const S* Tm, int64_t ncols, std::vector<S>& out) {
        out.assign((size_t)k * k, S(0));
        for (int64_t j = 0; j < ncols; ++j)
            for (int a = 0; a < k; ++a)
                for (int b = 0; b < k; ++b) out[(size_t)b * k + a] += Tm[(size_t)j * k + a] * Tm[(size_t)j * k + b];
        for (auto& v : out) v /= static_cast<S>(ncols);
    };
    if (tgt_H && cfg.target_lambda_H < 0) target_gram(cfg.target_H, n, TG_H);
    if (tgt_W && cfg.target_lambda_W < 0) target_gram(cfg.target_W, m, TG_W);
    S prev_loss = std::numeric_limits<S>::max();
    int patience_counter = 0;

    for (int iter = 0; iter < cfg.max_iter; ++iter) {
        // ------------------------------------------------ H half-update
        std::vector<S> B_sym;                                             // symmetric: B_w_saved (:671)
        if (cfg.symmetric) {
            // :474-477  SYMMETRIC_SKIP: no H update, no H scaling
        } else if (cfg.projective) {
            // variant_helpers.hpp:308-325  projective_h_update: H = (diag(d) W_T) A, then H -> d (fit_cpu.hpp:462-472)
            std::vector<S> Wd((size_t)k * m);
            for (int i = 0; i < m; ++i) for (int f = 0; f < k; ++f) Wd[(size_t)i * k + f] = W_T[(size_t)i * k + f] * d[f];
            rhs(A, Wd.data(), k, H, threads);
            extract_scaling(H, k, n, d, cfg.norm_type);
        } else {
        gram(W_T, k, m, G.data());                                       // :491
        if (cfg.has_mask) {
            // :560-564 G rebuilt unmodified (eps only); L1/L2 per column inside
            masked_nnls(A, W_T, G.data(), H, cfg.mask, k, cfg.L1_H, cfg.L2_H, cfg.nonneg_H,
                        cfg.cd_maxit, cfg.cd_tol, cfg.solver_mode, threads, iter > 0);
        } else if (irls) {
            // :565-606 gram recomputed (eps only); L1 inside CD, L2 on G_w
            nnls_batch_irls_sparse_nb(A, W_T, G.data(), H, k, cfg.L1_H, cfg.L2_H, cfg.nonneg_H,
                                      cfg.cd_maxit, cfg.irls_max_iter, cfg.irls_tol, cfg.threads,
                                      is_nb && !per_col ? nb_size.data() : (const S*)nullptr,          // :577-583: PER_COL -> theta_per_col
                                      is_nb && per_col ? nb_size.data() : (const S*)nullptr, cfg.loss_type, cfg.tweedie_power, cfg.robust_delta, cfg.dense_input);
        } else {
            if (cfg.L2_H > 0) for (int i = 0; i < k; ++i) G[(size_t)i * k + i] += cfg.L2_H;   // :506
            if (cfg.has_graph_H) apply_graph_reg(G.data(), cfg.graph_H, H, k, cfg.graph_H_lambda);      // :508-509
            apply_L21(G.data(), H, k, (int64_t)n, cfg.L21_H);                                  // :509-510 (current H)
            if (unfused) {
                // :540-631 STANDARD PATH (dense data / target): B = W_T A, L1 on B (sparsity.hpp:47), target, nnls_batch / cholesky_clip_batch
                std::vector<S> B((size_t)k * n);
                rhs(A, W_T, k, B.data(), threads);
                if (cfg.L1_H > 0) for (auto& b : B) b -= cfg.L1_H;
                if (tgt_H) apply_target(G.data(), B.data(), cfg.target_H, TG_H.data(), k, (int64_t)n, cfg.target_lambda_H);   // variant_helpers.hpp:105-146
                if (cfg.solver_mode == 1) cholesky_clip_batch(G.data(), B.data(), H, k, n, cfg.nonneg_H, threads);
                else nnls_batch(G.data(), B.data(), H, k, n, cfg.cd_maxit, cfg.cd_tol, S(0), S(0), cfg.nonneg_H, threads, S(0), iter > 0);
            } else
            if (cfg.solver_mode == 0)
                fused_rhs_nnls_sparse(A, W_T, G.data(), H, k, cfg.cd_maxit, cfg.cd_tol, cfg.L1_H,
                                      cfg.nonneg_H, threads, iter > 0, S(0));                // :516-524
            else
                fused_rhs_cholesky_sparse(A, W_T, G.data(), H, k, cfg.L1_H, cfg.nonneg_H, threads, S(0));
        }
        if (cfg.ub_H > 0) apply_upper_bound(H, (size_t)k * n, cfg.ub_H);  // :636-637
        apply_angular_posthoc(H, k, (int64_t)n, cfg.angular_H);              // :638-639
        extract_scaling(H, k, n, d, cfg.norm_type);                       // :645
        }   // standard H update

        // ------------------------------------------------ W half-update
        const bool saved_for_loss = !cfg.has_mask && !irls;
        if (cfg.symmetric) {
            // :659-704  Gram = W_T W_T^T, RHS = W_T A (A = A^T); the saved pair is the one of the OLD W_T
            gram(W_T, k, m, G.data());
            B_sym.assign((size_t)k * n, S(0));
            rhs(A, W_T, k, B_sym.data(), threads);
            G_saved = G;                                                  // :669-673
            std::vector<S> B = B_sym;
            if (cfg.L2_W > 0) for (int i = 0; i < k; ++i) G[(size_t)i * k + i] += cfg.L2_W;   // apply_w_features: sparsity.hpp:46-47
            if (cfg.L1_W > 0) for (auto& b : B) b -= cfg.L1_W;
            if (cfg.has_graph_W) apply_graph_reg(G.data(), cfg.graph_W, W_T, k, cfg.graph_W_lambda);
            apply_L21(G.data(), W_T, k, (int64_t)m, cfg.L21_W);
            if (cfg.solver_mode == 1) cholesky_clip_batch(G.data(), B.data(), W_T, k, m, cfg.nonneg_W, threads);   // :679-683
            else nnls_batch(G.data(), B.data(), W_T, k, m, cfg.cd_maxit, cfg.cd_tol, S(0), S(0), cfg.nonneg_W, threads, S(0), iter > 0);   // :684-692
        } else {
        gram(H, k, n, G.data());                                          // :715
        if (saved_for_loss) G_saved = G;                                  // :719-722
        if (cfg.has_mask) {
            masked_nnls(At, H, G.data(), W_T, maskT_own.view(), k, cfg.L1_W, cfg.L2_W, cfg.nonneg_W,
                        cfg.cd_maxit, cfg.cd_tol, cfg.solver_mode, threads, iter > 0);   // :799-810
        } else if (irls) {
            // :811-852: theta_per_col = nb_size (row of A == column of A^T)
            nnls_batch_irls_sparse_nb(At, H, G.data(), W_T, k, cfg.L1_W, cfg.L2_W, cfg.nonneg_W,
                                      cfg.cd_maxit, cfg.irls_max_iter, cfg.irls_tol, cfg.threads,
                                      is_nb && per_col ? nb_size.data() : (const S*)nullptr,           // :820-830: PER_COL -> row of A^T
                                      is_nb && !per_col ? nb_size.data() : (const S*)nullptr, cfg.loss_type, cfg.tweedie_power, cfg.robust_delta, cfg.dense_input);
        } else {
            if (cfg.L2_W > 0) for (int i = 0; i < k; ++i) G[(size_t)i * k + i] += cfg.L2_W;   // :738
            if (cfg.has_graph_W) apply_graph_reg(G.data(), cfg.graph_W, W_T, k, cfg.graph_W_lambda);    // :740-741
            apply_L21(G.data(), W_T, k, (int64_t)m, cfg.L21_W);                                // :741-745 (current W_T)
            if (unfused) {
                // :774-881 STANDARD PATH: B = H A^T saved BEFORE the features (:786-789), then as on the H side
                B_sym.assign((size_t)k * m, S(0));
                rhs(At, H, k, B_sym.data(), threads);
                std::vector<S> B = B_sym;
                if (cfg.L1_W > 0) for (auto& b : B) b -= cfg.L1_W;
                if (tgt_W) apply_target(G.data(), B.data(), cfg.target_W, TG_W.data(), k, (int64_t)m, cfg.target_lambda_W);
                if (cfg.solver_mode == 1) cholesky_clip_batch(G.data(), B.data(), W_T, k, m, cfg.nonneg_W, threads);
                else nnls_batch(G.data(), B.data(), W_T, k, m, cfg.cd_maxit, cfg.cd_tol, S(0), S(0), cfg.nonneg_W, threads, S(0), iter > 0);
            } else
            if (cfg.solver_mode == 0)
                fused_rhs_nnls_sparse(At, H, G.data(), W_T, k, cfg.cd_maxit, cfg.cd_tol, cfg.L1_W,
                                      cfg.nonneg_W, threads, iter > 0, S(0));                // :748-757
            else
                fused_rhs_cholesky_sparse(At, H, G.data(), W_T, k, cfg.L1_W, cfg.nonneg_W, threads, S(0));
        }
        }   // standard / projective W update
        if (cfg.ub_W > 0) apply_upper_bound(W_T, (size_t)k * m, cfg.ub_W);  // :884-885 (symmetric: :694-695)
        apply_angular_posthoc(W_T, k, (int64_t)m, cfg.angular_W);              // :886-887 (:696-697)
        extract_scaling(W_T, k, m, d, cfg.norm_type);                       // :893 (:700)
        if (cfg.symmetric) std::memcpy(H, W_T, sizeof(S) * (size_t)k * m);  // :704 symmetric_enforce_h

        // ------------------------------------------------ NB dispersion (:1094-1265)
        if (is_gp && cfg.dispersion_mode != 0) gp_theta_update(A, W_T, H, d, k, cfg, nb_size);        // :914-1008
        if (is_nb && cfg.dispersion_mode != 0) nb_size_update(A, W_T, H, d, k, cfg, nb_size);
        if (is_pow && cfg.dispersion_mode != 0) phi_update(A, W_T, H, d, k, cfg, nb_size);            // :1561-1670

        // ------------------------------------------------ loss (:1684-1767)
        S loss_val;
        if (cfg.has_mask || irls) {
            std::vector<S> Wd((size_t)k * m);
            for (int i = 0; i < m; ++i) for (int f = 0; f < k; ++f) Wd[(size_t)i * k + f] = W_T[(size_t)i * k + f] * d[f];
            loss_val = cfg.has_mask ? masked_loss(A, Wd.data(), H, cfg.mask, k, threads, cfg.loss_type, cfg.tweedie_power)
                                    : explicit_loss_sparse_nb(A, Wd.data(), H, k, nb_size.data(), cfg.threads > 0 ? cfg.threads : 1, cfg.loss_type, cfg.tweedie_power, cfg.robust_delta, per_col);
        } else {
            gram(W_T, k, m, G_wt.data());                                                   // :1734-1735
            S cross;
            if (cfg.symmetric || unfused) {                                             // :1717-1720 (B_w_saved materialised)
                cross = 0;
                for (int f = 0; f < k; ++f) {
                    S rowdot = 0;
                    for (int j = 0; j < m; ++j) rowdot += W_T[(size_t)j * k + f] * B_sym[(size_t)j * k + f];
                    cross += d[f] * rowdot;
                }
            } else cross = loss_cross_term_sparse_via_At(At, W_T, H, d, k, threads);        // :1740-1741
            S recon = 0;
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j)
                    recon += d[i] * d[j] * G_wt[(size_t)j * k + i] * G_saved[(size_t)j * k + i];   // :1748-1751
            loss_val = trAtA - static_cast<S>(2) * cross + recon;                            // :1753
        }
        res.loss_history.push_back(loss_val);
        bool loss_converged = false;
        if (iter > 0) {
            const S rel = std::abs(prev_loss - loss_val) / (std::abs(prev_loss) + static_cast<S>(1e-15));
            res.final_tol = rel;
            if (rel < cfg.tol) loss_converged = true;
        }
        prev_loss = loss_val;
        if (iter > 0) {                                                   // :1797-1809
            if (loss_converged) {
                if (++patience_counter >= cfg.patience) {
                    res.converged = true; res.train_loss = prev_loss; res.iterations = iter + 1;
                    break;
                }
            } else patience_counter = 0;
        }
        res.iterations = iter + 1;
    }
    if (res.train_loss == 0 && !res.loss_history.empty()) res.train_loss = res.loss_history.back();
    if (irls) res.theta = nb_size;      // NB sizes / GP theta (zeros) / phi (ones)
    if (cfg.sort_model) sort_by_d(W_T, H, d, k, m, n);                    // :1847
    return res;
}

template FitResult<float> nmf_fit<float>(const Csc<float>&, const FitConfig<float>&, float*, float*, float*);
template FitResult<double> nmf_fit<double>(const Csc<double>&, const FitConfig<double>&, double*, double*, double*);

}  // namespace oracle

// ============================================================================
// C surface (ctypes).  *_f32 / *_f64 pairs.  All matrices column-major, k leading.
// ============================================================================
using namespace oracle;

#define ORACLE_API extern "C" __attribute__((visibility("default")))

ORACLE_API uint64_t oracle_splitmix_next(uint64_t* state) {
    SplitMix64 r(1); r.state = *state; const uint64_t v = r.next(); *state = r.state; return v;
}
// Test infrastructure for the PRODUCT's fp64 coordinate sweeps (rcppml_amd/csrc/kernels.hip.h cd_quotient): they form b / g as
// q0 = b ginv, q = q0 + (b - q0 g) ginv with ginv = 1 / g rounded once per column.  Counts the operand pairs (random sign, mantissa and
// an exponent spread of 2^-40 .. 2^40) on which that differs from the division operator the reference writes (nnls_batch.hpp:100).
// Compiled with -ffp-contract=off: the two fmas are the explicit ones.
ORACLE_API long long oracle_corrected_quotient_mismatches(uint64_t seed, long long n, double* first_b, double* first_g) {
    SplitMix64 r(seed);
    long long bad = 0;
    for (long long t = 0; t < n; ++t) {
        const uint64_t u = r.next(), v = r.next();
        const double g = std::ldexp(1.0 + (double)(u >> 12) * 0x1p-52, (int)(u & 63) - 32);                   // > 0
        double b = std::ldexp(1.0 + (double)(v >> 12) * 0x1p-52, (int)(v & 63) - 32 + (int)((u >> 6) & 15) - 8);
        if (v & 64) b = -b;
        const double ginv = 1.0 / g;
        const double q0 = b * ginv;
        const double q = std::fma(std::fma(-q0, g, b), ginv, q0);
        if (q != b / g) { if (!bad) { *first_b = b; *first_g = g; } ++bad; }
    }
    return bad;
}
ORACLE_API uint64_t oracle_splitmix_init_state(uint64_t seed) { return SplitMix64(seed).state; }
ORACLE_API void oracle_fill_uniform_f64(uint64_t seed, double* out, int rows, int cols) { SplitMix64 r(seed); r.fill_uniform(out, rows, cols); }
ORACLE_API void oracle_fill_uniform_f32(uint64_t seed, float* out, int rows, int cols) { SplitMix64 r(seed); r.fill_uniform(out, rows, cols); }
// nmf_init.hpp:166-182  one stream fills W_T (k x m) then continues into H (k x n)
ORACLE_API void oracle_init_factors_f64(uint32_t seed, int k, int m, int n, double* W_T, double* H) { SplitMix64 r(seed); r.fill_uniform(W_T, k, m); r.fill_uniform(H, k, n); }
ORACLE_API void oracle_init_factors_f32(uint32_t seed, int k, int m, int n, float* W_T, float* H) { SplitMix64 r(seed); r.fill_uniform(W_T, k, m); r.fill_uniform(H, k, n); }

template <class S> static Csc<S> mk(int rows, int cols, const int* p, const int* i, const S* x) { return Csc<S>{rows, cols, p, i, x}; }

#define DEFINE_PRIMS(SUF, S)                                                                              \
    ORACLE_API void oracle_gram_##SUF(const S* F, int k, int r, S* G) { gram(F, k, r, G); }                \
    ORACLE_API void oracle_rhs_##SUF(int rows, int cols, const int* p, const int* i, const S* x,          \
                                     const S* F, int k, S* B, int threads) {                              \
        rhs(mk(rows, cols, p, i, x), F, k, B, threads);                                                   \
    }                                                                                                     \
    ORACLE_API int oracle_cd_col_##SUF(const S* G, S* b, S* x, int k, S L1, S L2, int nonneg, int maxit,  \
                                       S ub, S tol) {                                                     \
        return cd_nnls_col_fixed(G, b, x, k, L1, L2, nonneg != 0, maxit, ub, tol);                        \
    }                                                                                                     \
    ORACLE_API void oracle_nnls_batch_##SUF(const S* G, S* B, S* X, int k, int n, int maxit, S tol,       \
                                            S L1, S L2, int nonneg, int threads, S ub, int warm) {        \
        nnls_batch(G, B, X, k, n, maxit, tol, L1, L2, nonneg != 0, threads, ub, warm != 0);               \
    }                                                                                                     \
    ORACLE_API void oracle_fused_cd_##SUF(int rows, int cols, const int* p, const int* i, const S* x,     \
                                          const S* F, const S* G, S* X, int k, int maxit, S tol, S L1,    \
                                          int nonneg, int threads, int warm, S ub) {                      \
        fused_rhs_nnls_sparse(mk(rows, cols, p, i, x), F, G, X, k, maxit, tol, L1, nonneg != 0, threads,  \
                              warm != 0, ub);                                                             \
    }                                                                                                     \
    ORACLE_API void oracle_fused_chol_##SUF(int rows, int cols, const int* p, const int* i, const S* x,   \
                                            const S* F, const S* G, S* X, int k, S L1, int nonneg,        \
                                            int threads, S ub) {                                          \
        fused_rhs_cholesky_sparse(mk(rows, cols, p, i, x), F, G, X, k, L1, nonneg != 0, threads, ub);     \
    }                                                                                                     \
    ORACLE_API void oracle_chol_clip_batch_##SUF(const S* G, const S* B, S* X, int k, int n, int nonneg,  \
                                                 int threads) {                                           \
        cholesky_clip_batch(G, B, X, k, n, nonneg != 0, threads);                                         \
    }                                                                                                     \
    ORACLE_API int oracle_llt_##SUF(const S* G, int k, S* L) { return llt_factor(G, k, L) ? 1 : 0; }      \
    ORACLE_API void oracle_extract_scaling_##SUF(S* X, int k, int c, S* d, int norm_type) {               \
        extract_scaling(X, k, c, d, norm_type);                                                           \
    }                                                                                                     \
    ORACLE_API S oracle_trace_AtA_##SUF(int rows, int cols, const int* p, const int* i, const S* x) {     \
        return trace_AtA(mk(rows, cols, p, i, x));                                                        \
    }                                                                                                     \
    ORACLE_API S oracle_loss_cross_##SUF(int rows, int cols, const int* p, const int* i, const S* x,      \
                                         const S* W_T, const S* H, const S* d, int k, int threads) {      \
        return loss_cross_term_sparse_via_At(mk(rows, cols, p, i, x), W_T, H, d, k, threads);             \
    }                                                                                                     \
    ORACLE_API void oracle_transpose_csc_##SUF(int rows, int cols, const int* p, const int* i,            \
                                               const S* x, int* tp, int* ti, S* tx) {                     \
        CscOwned<S> T = transpose_csc(mk(rows, cols, p, i, x));                                           \
        std::memcpy(tp, T.p.data(), sizeof(int) * T.p.size());                                            \
        std::memcpy(ti, T.i.data(), sizeof(int) * T.i.size());                                            \
        std::memcpy(tx, T.x.data(), sizeof(S) * T.x.size());                                              \
    }                                                                                                     \
    /* Full fit.  mask_p == NULL => no mask.  loss_hist: >= max_iter entries or NULL.  theta: m or NULL */ \
    ORACLE_API void oracle_nmf_fit_##SUF(                                                                  \
        int m, int n, const int* p, const int* i, const S* x, int k, S* W_T, S* H, S* d, int max_iter,    \
        S tol, S L1_H, S L1_W, S L2_H, S L2_W, S ub_H, S ub_W, int cd_maxit, S cd_tol, int patience,      \
        int nonneg_W, int nonneg_H, int norm_type, int solver_mode, int loss_type, int irls_max_iter,     \
        S irls_tol, int dispersion_mode, S nb_size_init, S nb_size_max, S nb_size_min, int sort_model,    \
        int threads, const int* mask_p, const int* mask_i, const S* mask_x, int* out_iter,                \
        int* out_converged, S* out_loss, S* out_tol, S* loss_hist, S* out_theta, S tweedie_power,         \
        S L21_H, S L21_W, S angular_H, S angular_W, S robust_delta, int projective,                       \
        const int* gH_p, const int* gH_i, const S* gH_x, S gH_lambda, const int* gW_p, const int* gW_i,   \
        const S* gW_x, S gW_lambda, S gp_theta_init, S gp_theta_max, S gamma_phi_init, S gamma_phi_max,   \
        S gamma_phi_min, int symmetric, int unfused, const S* target_H, S target_lambda_H,                \
        const S* target_W, S target_lambda_W) {                                                           \
        FitConfig<S> c;                                                                                   \
        c.k = k; c.max_iter = max_iter; c.tol = tol; c.L1_H = L1_H; c.L1_W = L1_W; c.L2_H = L2_H;         \
        c.L2_W = L2_W; c.ub_H = ub_H; c.ub_W = ub_W; c.cd_maxit = cd_maxit; c.cd_tol = cd_tol;            \
        c.patience = patience; c.nonneg_W = nonneg_W != 0; c.nonneg_H = nonneg_H != 0;                    \
        c.norm_type = norm_type; c.solver_mode = solver_mode; c.loss_type = loss_type;                    \
        c.irls_max_iter = irls_max_iter; c.irls_tol = irls_tol; c.dispersion_mode = dispersion_mode;      \
        c.nb_size_init = nb_size_init; c.nb_size_max = nb_size_max; c.nb_size_min = nb_size_min;          \
        c.sort_model = sort_model != 0; c.threads = threads; c.tweedie_power = tweedie_power;             \
        c.L21_H = L21_H; c.L21_W = L21_W; c.angular_H = angular_H; c.angular_W = angular_W; c.robust_delta = robust_delta; c.projective = projective != 0; \
        c.gp_theta_init = gp_theta_init; c.gp_theta_max = gp_theta_max; c.gamma_phi_init = gamma_phi_init;  \
        c.gamma_phi_max = gamma_phi_max; c.gamma_phi_min = gamma_phi_min; c.symmetric = symmetric != 0;   \
        c.unfused = unfused != 0;                                                                         \
        c.dense_input = unfused == 2;                                                                         \
        c.target_H = target_H; c.target_lambda_H = target_lambda_H; c.target_W = target_W; c.target_lambda_W = target_lambda_W; \
        if (gH_p) { c.has_graph_H = true; c.graph_H = mk(n, n, gH_p, gH_i, gH_x); c.graph_H_lambda = gH_lambda; } \
        if (gW_p) { c.has_graph_W = true; c.graph_W = mk(m, m, gW_p, gW_i, gW_x); c.graph_W_lambda = gW_lambda; } \
        if (mask_p) { c.has_mask = true; c.mask = mk(m, n, mask_p, mask_i, mask_x); }                     \
        FitResult<S> r = nmf_fit(mk(m, n, p, i, x), c, W_T, H, d);                                        \
        *out_iter = r.iterations; *out_converged = r.converged ? 1 : 0; *out_loss = r.train_loss;         \
        *out_tol = r.final_tol;                                                                           \
        if (loss_hist) for (size_t t = 0; t < r.loss_history.size(); ++t) loss_hist[t] = r.loss_history[t]; \
        if (out_theta) for (size_t t = 0; t < r.theta.size(); ++t) out_theta[t] = r.theta[t];             \
    }

ORACLE_API void oracle_proj_adv_f64(double* G, const double* TG, int k, double abs_lambda) { proj_adv_gram<double>(G, TG, k, abs_lambda); }

DEFINE_PRIMS(f32, float)
DEFINE_PRIMS(f64, double)


// per-element NB pieces (pinned against the reference's math/loss.hpp in tests/test_oracle_ref.py)
ORACLE_API double oracle_irls_weight_nb_f64(double p, double r) { return irls_weight_nb<double>(p, r); }
ORACLE_API float oracle_irls_weight_nb_f32(float p, float r) { return irls_weight_nb<float>(p, r); }
ORACLE_API double oracle_loss_nb_f64(double y, double p, double r) { return loss_contribution_nb<double>(y, p, r); }
ORACLE_API float oracle_loss_nb_f32(float y, float p, float r) { return loss_contribution_nb<float>(y, p, r); }

ORACLE_API double oracle_robust_modifier_f64(double r, double delta) { return robust_huber_modifier<double>(r, delta); }
ORACLE_API float oracle_robust_modifier_f32(float r, float delta) { return robust_huber_modifier<float>(r, delta); }
ORACLE_API double oracle_robust_loss_f64(int lt, double y, double p, double th, double pw, double delta) { return compute_robust_loss<double>(lt, y, p, th, pw, delta); }
ORACLE_API float oracle_robust_loss_f32(int lt, float y, float p, float th, float pw, float delta) { return compute_robust_loss<float>(lt, y, p, th, pw, delta); }
ORACLE_API double oracle_irls_weight_power_f64(double p, double pw) { return irls_weight_power<double>(p, pw); }
ORACLE_API float oracle_irls_weight_power_f32(float p, float pw) { return irls_weight_power<float>(p, pw); }
ORACLE_API double oracle_loss_dev_f64(int loss_type, double y, double p, double pw) { return loss_contribution<double>(loss_type, y, p, 0.0, pw); }
ORACLE_API float oracle_loss_dev_f32(int loss_type, float y, float p, float pw) { return loss_contribution<float>(loss_type, y, p, 0.f, pw); }
ORACLE_API double oracle_irls_weight_kl_f64(double p) { return irls_weight_kl<double>(p); }
ORACLE_API float oracle_irls_weight_kl_f32(float p) { return irls_weight_kl<float>(p); }
ORACLE_API double oracle_loss_gp_f64(double y, double p, double th) { return loss_contribution_gp<double>(y, p, th); }
ORACLE_API float oracle_loss_gp_f32(float y, float p, float th) { return loss_contribution_gp<float>(y, p, th); }

// NB-IRLS primitives exposed for kernel-level parity tests
#define DEFINE_NB(SUF, S)                                                                                    \
    ORACLE_API void oracle_irls_nb_##SUF(int rows, int cols, const int* p, const int* i, const S* x, const S* F, \
                                         const S* G, S* X, int k, S L1, S L2, int nonneg, int cd_maxit,           \
                                         int irls_max_iter, S irls_tol, int threads, const S* theta_row,          \
                                         const S* theta_col) {                                                    \
        nnls_batch_irls_sparse_nb(mk(rows, cols, p, i, x), F, G, X, k, L1, L2, nonneg != 0, cd_maxit,             \
                                  irls_max_iter, irls_tol, threads, theta_row, theta_col);                        \
    }                                                                                                             \
    ORACLE_API void oracle_nb_size_update_##SUF(int m, int n, const int* p, const int* i, const S* x,             \
                                                const S* W_T, const S* H, const S* d, int k, int dispersion_mode, \
                                                S r_min, S r_max, S* nb_size) {                                   \
        FitConfig<S> c; c.k = k; c.dense_input = dispersion_mode >= 16;   /* + 16: A stores every entry, dense branches */ \
        dispersion_mode &= 15; c.dispersion_mode = dispersion_mode; c.nb_size_min = r_min; c.nb_size_max = r_max;   \
        const size_t len = dispersion_mode == 3 ? (size_t)n : (size_t)m;   /* PER_COL: n values */                \
        std::vector<S> v(nb_size, nb_size + len);                                                                 \
        nb_size_update(mk(m, n, p, i, x), W_T, H, d, k, c, v);                                                    \
        std::memcpy(nb_size, v.data(), sizeof(S) * len);                                                          \
    }                                                                                                             \
    /* loss_type 4: GP theta (hi = cap); 6 / 7 / 8: Pearson phi in [lo, hi]; mode 1 = GLOBAL, 2 = PER_ROW */       \
    ORACLE_API void oracle_dispersion_update_##SUF(int loss_type, int m, int n, const int* p, const int* i,       \
                                                   const S* x, const S* W_T, const S* H, const S* d, int k,       \
                                                   int dispersion_mode, S power, S lo, S hi, S* theta) {          \
        FitConfig<S> c; c.k = k; c.dense_input = dispersion_mode >= 16; dispersion_mode &= 15;                    \
        c.dispersion_mode = dispersion_mode; c.loss_type = loss_type;                                             \
        c.tweedie_power = power; c.gp_theta_max = hi; c.gamma_phi_min = lo; c.gamma_phi_max = hi;                 \
        const size_t len = dispersion_mode == 3 ? (size_t)n : (size_t)m;   /* PER_COL: n values */                \
        std::vector<S> v(theta, theta + len);                                                                     \
        if (loss_type == 4) gp_theta_update(mk(m, n, p, i, x), W_T, H, d, k, c, v);                               \
        else phi_update(mk(m, n, p, i, x), W_T, H, d, k, c, v);                                                   \
        std::memcpy(theta, v.data(), sizeof(S) * len);                                                            \
    }                                                                                                             \
    ORACLE_API S oracle_nb_loss_##SUF(int m, int n, const int* p, const int* i, const S* x, const S* W_T,         \
                                      const S* d, const S* H, int k, const S* theta_row) {                        \
        std::vector<S> Wd((size_t)k * m);                                                                         \
        for (int r = 0; r < m; ++r) for (int f = 0; f < k; ++f) Wd[(size_t)r * k + f] = W_T[(size_t)r * k + f] * d[f]; \
        return explicit_loss_sparse_nb(mk(m, n, p, i, x), Wd.data(), H, k, theta_row, 1);                         \
    }                                                                                                             \
    /* generic forms: loss_type 5 = NB, 4 = GP (KL weights for the half-updates, GP likelihood for the loss) */   \
    ORACLE_API void oracle_irls_##SUF(int loss_type, int rows, int cols, const int* p, const int* i, const S* x,  \
                                      const S* F, const S* G, S* X, int k, S L1, S L2, int nonneg, int cd_maxit,  \
                                      int irls_max_iter, S irls_tol, int threads, const S* theta_row,             \
                                      const S* theta_col, S power, S robust) {                                    \
        /* loss_type + 16: every row stored, the dense column solve (irls_nnls_col_dense) */                      \
        nnls_batch_irls_sparse_nb(mk(rows, cols, p, i, x), F, G, X, k, L1, L2, nonneg != 0, cd_maxit,             \
                                  irls_max_iter, irls_tol, threads, theta_row, theta_col, loss_type & 15, power, robust, loss_type >= 16); \
    }                                                                                                             \
    ORACLE_API S oracle_irls_loss_##SUF(int loss_type, int m, int n, const int* p, const int* i, const S* x,      \
                                        const S* W_T, const S* d, const S* H, int k, const S* theta_row, S power, S robust) { \
        std::vector<S> Wd((size_t)k * m);                                                                         \
        for (int r = 0; r < m; ++r) for (int f = 0; f < k; ++f) Wd[(size_t)r * k + f] = W_T[(size_t)r * k + f] * d[f]; \
        return explicit_loss_sparse_nb(mk(m, n, p, i, x), Wd.data(), H, k, theta_row, 1, loss_type, power, robust); \
    }
DEFINE_NB(f32, float)
DEFINE_NB(f64, double)

// Cross-validation pieces (kernel-level parity) and the CV fit
ORACLE_API uint64_t oracle_cv_hash(uint64_t seed, uint32_t i, uint32_t j) { return SpeckledMask::hash(seed, i, j); }
ORACLE_API int oracle_cv_is_holdout(double frac, uint64_t cv_seed, int i, int j) { return SpeckledMask(frac, cv_seed, false).is_holdout(i, j) ? 1 : 0; }
#define DEFINE_CV(SUF, S)                                                                                         \
    ORACLE_API void oracle_cv_half_update_##SUF(int rows, int cols, const int* p, const int* i, const S* x, const S* F, \
                                                const S* G, S* X, int k, double frac, uint64_t cv_seed, int mask_zeros, \
                                                int transposed, S L1, int nonneg, int cd_maxit, int solver_mode,   \
                                                int threads) {                                                     \
        cv_half_update(mk(rows, cols, p, i, x), F, G, X, k, SpeckledMask(frac, cv_seed, mask_zeros != 0),          \
                       transposed != 0, L1, nonneg != 0, cd_maxit, solver_mode, threads);                          \
    }                                                                                                              \
    ORACLE_API void oracle_cv_test_error_##SUF(int m, int n, const int* p, const int* i, const S* x, const S* W_T, \
                                               const S* d, const S* H, int k, double frac, uint64_t cv_seed,       \
                                               int mask_zeros, S* sq_err, int64_t* n_test) {                       \
        std::vector<S> Wd((size_t)k * m);                                                                          \
        for (int r = 0; r < m; ++r) for (int f = 0; f < k; ++f) Wd[(size_t)r * k + f] = W_T[(size_t)r * k + f] * d[f]; \
        cv_test_error(mk(m, n, p, i, x), Wd.data(), H, k, SpeckledMask(frac, cv_seed, mask_zeros != 0), 1, sq_err, n_test); \
    }                                                                                                              \
    ORACLE_API void oracle_cv_irls_half_update_##SUF(int rows, int cols, const int* p, const int* i, const S* x, const S* F, \
                                                     const S* G_add, S* X, int k, double frac, uint64_t cv_seed, int mask_zeros, \
                                                     int transposed, S L1, int nonneg, int cd_maxit, int solver_mode, int loss_type, \
                                                     int irls_max_iter, S irls_tol, S power, S robust, int threads) { \
        FitConfig<S> c;                                                                                            \
        c.k = k; c.cd_maxit = cd_maxit; c.solver_mode = solver_mode; c.loss_type = loss_type; c.irls_max_iter = irls_max_iter; \
        c.irls_tol = irls_tol; c.tweedie_power = power; c.robust_delta = robust;                                   \
        cv_irls_half_update(mk(rows, cols, p, i, x), F, X, k, SpeckledMask(frac, cv_seed, mask_zeros != 0), transposed != 0, L1, \
                            nonneg != 0, c, G_add, threads);                                                       \
    }                                                                                                              \
    ORACLE_API void oracle_cv_explicit_loss_##SUF(int m, int n, const int* p, const int* i, const S* x, const S* W_T, const S* d, \
                                                  const S* H, int k, double frac, uint64_t cv_seed, int mask_zeros, int loss_type, \
                                                  S power, const S* theta, S* out_train, int64_t* n_train, S* out_test, int64_t* n_test) { \
        std::vector<S> Wd((size_t)k * m);                                                                          \
        for (int r = 0; r < m; ++r) for (int f = 0; f < k; ++f) Wd[(size_t)r * k + f] = W_T[(size_t)r * k + f] * d[f]; \
        FitConfig<S> c; c.k = k; c.loss_type = loss_type; c.tweedie_power = power;                                 \
        cv_explicit_loss(mk(m, n, p, i, x), Wd.data(), H, k, SpeckledMask(frac, cv_seed, mask_zeros != 0), c, theta, 1, out_train, \
                         n_train, out_test, n_test);                                                               \
    }                                                                                                              \
    ORACLE_API void oracle_cv_gp_theta_update_##SUF(int m, int n, const int* p, const int* i, const S* x, const S* W_T, const S* d, \
                                                    const S* H, int k, double frac, uint64_t cv_seed, int mode, S theta_max, S* theta) { \
        FitConfig<S> c; c.k = k; c.dispersion_mode = mode; c.gp_theta_max = theta_max;                             \
        std::vector<S> th(theta, theta + m);                                                                       \
        const SpeckledMask mk_(frac, cv_seed, false);                                                              \
        gp_theta_update(mk(m, n, p, i, x), W_T, H, d, k, c, th, frac > 0 ? &mk_ : nullptr);                        \
        std::copy(th.begin(), th.end(), theta);                                                                    \
    }                                                                                                              \
    ORACLE_API S oracle_irls_weight_gp_##SUF(S observed, S predicted, S theta, S blend) { return irls_weight_gp(observed, predicted, theta, blend); } \
    ORACLE_API void oracle_nmf_fit_cv_irls_##SUF(int m, int n, const int* p, const int* i, const S* x, int k, S* W_T, S* H, \
                                            S* d, int max_iter, S tol, S L1_H, S L1_W, S L2_H, S L2_W, int cd_maxit, \
                                            int nonneg_W, int nonneg_H, int norm_type, int solver_mode, double frac, \
                                            uint64_t cv_seed, int mask_zeros, int cv_patience, int threads,         \
                                            int loss_type, int irls_max_iter, S irls_tol, int dispersion_mode, S gp_theta_init, \
                                            S gp_theta_max, S power, S robust,                                      \
                                            int* out_iter, int* out_converged, S* out_train, S* out_test,           \
                                            S* out_best_test, int* out_best_iter, S* train_hist, S* test_hist, S* theta_out) { \
        FitConfig<S> c;                                                                                            \
        c.k = k; c.max_iter = max_iter; c.tol = tol; c.L1_H = L1_H; c.L1_W = L1_W; c.L2_H = L2_H; c.L2_W = L2_W;    \
        c.cd_maxit = cd_maxit; c.nonneg_W = nonneg_W != 0; c.nonneg_H = nonneg_H != 0; c.norm_type = norm_type;     \
        c.solver_mode = solver_mode; c.threads = threads; c.loss_type = loss_type; c.irls_max_iter = irls_max_iter; \
        c.irls_tol = irls_tol; c.dispersion_mode = dispersion_mode; c.gp_theta_init = gp_theta_init; c.gp_theta_max = gp_theta_max; \
        c.tweedie_power = power; c.robust_delta = robust;                                                          \
        CvResult<S> r = nmf_fit_cv(mk(m, n, p, i, x), c, frac, cv_seed, mask_zeros != 0, cv_patience, W_T, H, d);   \
        *out_iter = r.iterations; *out_converged = r.converged ? 1 : 0; *out_train = r.train_loss;                 \
        *out_test = r.test_loss; *out_best_test = r.best_test_loss; *out_best_iter = r.best_iter;                   \
        for (size_t t = 0; t < r.train_hist.size(); ++t) { if (train_hist) train_hist[t] = r.train_hist[t]; if (test_hist) test_hist[t] = r.test_hist[t]; } \
        if (theta_out) for (size_t t = 0; t < r.theta.size(); ++t) theta_out[t] = r.theta[t];                       \
    }                                                                                                              \
    /* the same with a user mask (pattern CSC, m x n): MSE or IRLS by loss_type / robust */                         \
    ORACLE_API void oracle_nmf_fit_cv_masked_##SUF(int m, int n, const int* p, const int* i, const S* x, int k, S* W_T, S* H, \
                                            S* d, int max_iter, S tol, S L1_H, S L1_W, S L2_H, S L2_W, int cd_maxit, \
                                            int nonneg_W, int nonneg_H, int norm_type, int solver_mode, double frac, \
                                            uint64_t cv_seed, int mask_zeros, int cv_patience, int threads,         \
                                            int loss_type, int irls_max_iter, S irls_tol, int dispersion_mode, S gp_theta_init, \
                                            S gp_theta_max, S power, S robust, const int* mask_p, const int* mask_i, const S* mask_x, \
                                            int* out_iter, int* out_converged, S* out_train, S* out_test,           \
                                            S* out_best_test, int* out_best_iter, S* train_hist, S* test_hist, S* theta_out) { \
        FitConfig<S> c;                                                                                            \
        c.k = k; c.max_iter = max_iter; c.tol = tol; c.L1_H = L1_H; c.L1_W = L1_W; c.L2_H = L2_H; c.L2_W = L2_W;    \
        c.cd_maxit = cd_maxit; c.nonneg_W = nonneg_W != 0; c.nonneg_H = nonneg_H != 0; c.norm_type = norm_type;     \
        c.solver_mode = solver_mode; c.threads = threads; c.loss_type = loss_type; c.irls_max_iter = irls_max_iter; \
        c.irls_tol = irls_tol; c.dispersion_mode = dispersion_mode; c.gp_theta_init = gp_theta_init; c.gp_theta_max = gp_theta_max; \
        c.tweedie_power = power; c.robust_delta = robust;                                                          \
        if (mask_p) { c.has_mask = true; c.mask = mk(m, n, mask_p, mask_i, mask_x); }                              \
        CvResult<S> r = nmf_fit_cv(mk(m, n, p, i, x), c, frac, cv_seed, mask_zeros != 0, cv_patience, W_T, H, d);   \
        *out_iter = r.iterations; *out_converged = r.converged ? 1 : 0; *out_train = r.train_loss;                 \
        *out_test = r.test_loss; *out_best_test = r.best_test_loss; *out_best_iter = r.best_iter;                   \
        for (size_t t = 0; t < r.train_hist.size(); ++t) { if (train_hist) train_hist[t] = r.train_hist[t]; if (test_hist) test_hist[t] = r.test_hist[t]; } \
        if (theta_out) for (size_t t = 0; t < r.theta.size(); ++t) theta_out[t] = r.theta[t];                       \
    }                                                                                                              \
    ORACLE_API void oracle_nmf_fit_cv_##SUF(int m, int n, const int* p, const int* i, const S* x, int k, S* W_T, S* H, \
                                            S* d, int max_iter, S tol, S L1_H, S L1_W, S L2_H, S L2_W, int cd_maxit, \
                                            int nonneg_W, int nonneg_H, int norm_type, int solver_mode, double frac, \
                                            uint64_t cv_seed, int mask_zeros, int cv_patience, int threads,         \
                                            int* out_iter, int* out_converged, S* out_train, S* out_test,           \
                                            S* out_best_test, int* out_best_iter, S* train_hist, S* test_hist,     \
                                            const int* gH_p, const int* gH_i, const S* gH_x, S gH_lambda,           \
                                            const int* gW_p, const int* gW_i, const S* gW_x, S gW_lambda) {         \
        FitConfig<S> c;                                                                                            \
        c.k = k; c.max_iter = max_iter; c.tol = tol; c.L1_H = L1_H; c.L1_W = L1_W; c.L2_H = L2_H; c.L2_W = L2_W;    \
        c.cd_maxit = cd_maxit; c.nonneg_W = nonneg_W != 0; c.nonneg_H = nonneg_H != 0; c.norm_type = norm_type;     \
        c.solver_mode = solver_mode; c.threads = threads;                                                          \
        if (gH_p) { c.has_graph_H = true; c.graph_H = mk(n, n, gH_p, gH_i, gH_x); c.graph_H_lambda = gH_lambda; }  \
        if (gW_p) { c.has_graph_W = true; c.graph_W = mk(m, m, gW_p, gW_i, gW_x); c.graph_W_lambda = gW_lambda; }  \
        CvResult<S> r = nmf_fit_cv(mk(m, n, p, i, x), c, frac, cv_seed, mask_zeros != 0, cv_patience, W_T, H, d);   \
        *out_iter = r.iterations; *out_converged = r.converged ? 1 : 0; *out_train = r.train_loss;                 \
        *out_test = r.test_loss; *out_best_test = r.best_test_loss; *out_best_iter = r.best_iter;                   \
        for (size_t t = 0; t < r.train_hist.size(); ++t) { if (train_hist) train_hist[t] = r.train_hist[t]; if (test_hist) test_hist[t] = r.test_hist[t]; } \
    }
DEFINE_CV(f32, float)
DEFINE_CV(f64, double)

// ---------------------------------------------------------------------------
// src/RcppFunctions_utils.cpp:313-366  c_nnls (fp64): h = NNLS(w^T w, w^T A)
//   w_T: k x m (already transposed), A: m x n CSC.  G gets eps TWICE (gram + :327).
//   warm != 0: h holds the warm start; B -= G h; CD with cd_tol = 0 (default arg).
// ---------------------------------------------------------------------------
ORACLE_API void oracle_c_nnls(const double* w_T, int k, int m, int n, const int* p, const int* i,
                              const double* x, double* h, int cd_maxit, double cd_tol, double L1,
                              double L2, double ub, int nonneg, int threads, int warm) {
    std::vector<double> G((size_t)k * k), B((size_t)k * n);
    gram(w_T, k, m, G.data());
    for (int a = 0; a < k; ++a) G[(size_t)a * k + a] += 1e-15;
    if (L2 > 0) for (int a = 0; a < k; ++a) G[(size_t)a * k + a] += L2;
    rhs(mk(m, n, p, i, x), w_T, k, B.data(), threads);
    if (warm) nnls_batch(G.data(), B.data(), h, k, n, cd_maxit, 0.0, L1, 0.0, nonneg != 0, threads, ub, true);
    else      nnls_batch(G.data(), B.data(), h, k, n, cd_maxit, cd_tol, L1, 0.0, nonneg != 0, threads, ub, false);
}
// src/RcppFunctions_utils.cpp:23-52  Rcpp_predict: same with cd_maxit=100, cd_tol=1e-8, nonneg=true
ORACLE_API void oracle_predict(const double* w_T, int k, int m, int n, const int* p, const int* i,
                               const double* x, double* h, double L1, double L2, int threads, double ub) {
    oracle_c_nnls(w_T, k, m, n, p, i, x, h, 100, 1e-8, L1, L2, ub, 1, threads, 0);
}
// src/RcppFunctions_utils.cpp:95-163  Rcpp_evaluate_loss (MSE): MEAN squared error of
// W diag(d) H over all m*n entries, or over nonzeros only if mask_zeros.  W: m x k col-major.
ORACLE_API double oracle_evaluate_mse(const double* W, const double* d, const double* H, int k, int m,
                                      int n, const int* p, const int* i, const double* x, int mask_zeros) {
    double total = 0; long long count = 0;
    std::vector<double> col(m);
    for (int j = 0; j < n; ++j) {
        const double* h = H + (size_t)j * k;
        for (int r = 0; r < m; ++r) {
            double s = 0;
            for (int f = 0; f < k; ++f) s += W[(size_t)f * m + r] * d[f] * h[f];
            col[r] = s;
        }
        if (mask_zeros) {
            for (int t = p[j]; t < p[j + 1]; ++t) { const double df = x[t] - col[i[t]]; total += df * df; ++count; }
        } else {
            int t = p[j];
            for (int r = 0; r < m; ++r) {
                double obs = 0;
                if (t < p[j + 1] && i[t] == r) obs = x[t++];
                const double df = obs - col[r]; total += df * df; ++count;
            }
        }
    }
    return count > 0 ? total / (double)count : 0.0;
}
ORACLE_API int oracle_num_threads() { return eff_threads(0); }
