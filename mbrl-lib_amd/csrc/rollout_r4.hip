// rollout_r4.hip -- rollout_kernel with R = 4 row tiles (16 rows each) per workgroup; see rollout.hpp.  (Seven instances: one unit.)
#define HIPETS_R 4
#define HIPETS_PART 0
#define HIPETS_LAUNCH_FN launch_rollout_r4
#include "rollout_inst.inc"
