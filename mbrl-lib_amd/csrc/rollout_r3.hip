// rollout_r3.hip -- rollout_kernel with R = 3 row tiles (16 rows each) per workgroup; see rollout.hpp.
#define HIPETS_R 3
#define HIPETS_LAUNCH_FN launch_rollout_r3
#include "rollout_inst.inc"
