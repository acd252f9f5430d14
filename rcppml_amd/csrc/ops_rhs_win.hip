// ops_rhs_win.hip -- planner and finishing kernels of the window form of the sparse right-hand side, both precisions
// (rhs_win_impl.hip.h); the tile-loop kernels are instantiated in ops_rhs_win_{f32,f64}_nv*.hip.
#include "rhs_win_impl.hip.h"
#include "rhs_win_finish.hip.h"

rcppml_rhs_plan* rcppml_rw_build_f32(rcppml_hip_ctx* c, const int* colptr, const int* rowidx, const float* vals, int64_t ncols,
                                     int64_t nrows, int k, int partitions, int rate_code) {
    return rw_launch::build_plan<float>(c, RCPPML_F32, colptr, rowidx, vals, ncols, nrows, k, partitions, rate_code);
}
void rcppml_rw_run_f32(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const float* F, float* B) { rw_launch::run_plan<float>(c, pl, F, B); }
rcppml_rhs_plan* rcppml_rw_build_f64(rcppml_hip_ctx* c, const int* colptr, const int* rowidx, const double* vals, int64_t ncols,
                                     int64_t nrows, int k, int partitions, int rate_code) {
    return rw_launch::build_plan<double>(c, RCPPML_F64, colptr, rowidx, vals, ncols, nrows, k, partitions, rate_code);
}
void rcppml_rw_run_f64(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const double* F, double* B) { rw_launch::run_plan<double>(c, pl, F, B); }
void rcppml_rw_set_values_f32(rcppml_hip_ctx* c, rcppml_rhs_plan* pl, const float* vals) { rw_launch::set_values<float>(c, pl, vals); }
void rcppml_rw_set_values_f64(rcppml_hip_ctx* c, rcppml_rhs_plan* pl, const double* vals) { rw_launch::set_values<double>(c, pl, vals); }
