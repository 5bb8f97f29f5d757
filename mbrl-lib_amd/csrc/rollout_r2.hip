// rollout_r2.hip -- rollout_kernel with R = 2 row tiles (16 rows each) per workgroup; see rollout.hpp.
#define HIPETS_R 2
#define HIPETS_LAUNCH_FN launch_rollout_r2
#include "rollout_inst.inc"
