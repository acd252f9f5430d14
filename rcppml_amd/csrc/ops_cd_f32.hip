// ops_cd_f32.hip -- fp32 instantiation of the CD solve kernels
#include "solve_cd_impl.hip.h"
void rcppml_solve_cd_f32(rcppml_hip_ctx* c, const float* G, const float* B, float* X, int k, int64_t ncols, float l1_pre,
 int warm, int zero_init, float l1_cd, float l2_cd, int nonneg, int maxit, float tol, float ub_cd, float ub_post, int variant, int* sweeps, const int* order) {
    solve_cd_impl<float>(c, G, B, X, k, ncols, l1_pre, warm, zero_init, l1_cd, l2_cd, nonneg, maxit, tol, ub_cd, ub_post, variant, sweeps, order);
}
