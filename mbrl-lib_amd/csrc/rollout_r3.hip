// rollout_r3.hip -- rollout_kernel with R = 3 row tiles (16 rows each) per workgroup; see rollout.hpp.  This unit: the launcher and every
// instance but the FAST-mode shape-specialised ones (rollout_r3_fast.hip; rollout_inst.inc HIPETS_PART).
#define HIPETS_R 3
#define HIPETS_PART 1
#define HIPETS_LAUNCH_FN launch_rollout_r3
#define HIPETS_LAUNCH_FAST_FN launch_rollout_r3_fast
#include "rollout_inst.inc"
