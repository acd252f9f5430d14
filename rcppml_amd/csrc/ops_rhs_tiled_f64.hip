// ops_rhs_tiled_f64.hip -- fp64 instantiations of the LDS row-tiled right-hand-side kernel (see ops_rhs_tiled.hip)
#include "rhs_tiled_launch.hip.h"

void rcppml_rt_launch_f64(rcppml_hip_ctx* c, const rcppml_rhs_plan* pl, const double* F, const double* Binit, double* Bout) {
    rt_launch::launch_tiled_any<double>(c, pl, F, Binit, Bout);
}
