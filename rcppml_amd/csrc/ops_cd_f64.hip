// ops_cd_f64.hip -- fp64 instantiation of the CD solve kernels
#include "solve_cd_impl.hip.h"
void rcppml_solve_cd_f64(rcppml_hip_ctx* c, const double* G, const double* B, double* X, int k, int64_t ncols, double l1_pre,
 int warm, int zero_init, double l1_cd, double l2_cd, int nonneg, int maxit, double tol, double ub_cd, double ub_post, int variant, int* sweeps, const int* order) {
    solve_cd_impl<double>(c, G, B, X, k, ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, variant, sweeps, order);
}
