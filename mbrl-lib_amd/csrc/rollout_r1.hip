// rollout_r1.hip -- rollout_kernel with R = 1 row tiles (16 rows each) per workgroup; see rollout.hpp.
#define HIPETS_R 1
#define HIPETS_LAUNCH_FN launch_rollout_r1
#include "rollout_inst.inc"
