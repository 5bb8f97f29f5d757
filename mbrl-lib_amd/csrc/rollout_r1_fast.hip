// rollout_r1_fast.hip -- the FAST-mode shape-specialised instances of rollout_kernel with R = 1 row tiles per workgroup (the other
// half of rollout_r1.hip: the two compile in parallel; rollout_inst.inc HIPETS_PART).
#define HIPETS_R 1
#define HIPETS_PART 2
#define HIPETS_LAUNCH_FN launch_rollout_r1
#define HIPETS_LAUNCH_FAST_FN launch_rollout_r1_fast
#include "rollout_inst.inc"
