// ops_rhs_win_f64_nv2.hip -- tile-loop kernels of the window rhs for double, rows of 512 bytes (own unit: parallel build)
#include "rhs_win_impl.hip.h"
void rcppml_rw_launch_f64_nv2(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const double* F, double* Bout) {
    rw_launch::launch_clo<double, 2>(c, pl, F, Bout);
}
