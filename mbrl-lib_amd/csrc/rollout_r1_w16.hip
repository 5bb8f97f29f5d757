// rollout_r1_w16.hip -- the small-batch variant of the rollout kernel: R = 1 row tile per workgroup, 16 waves (1024 threads), one
// column tile of a hidden layer per wave.  Compiled into hipets::w16 (common.hpp); see launch.hpp.
#define HIPETS_WAVES 16
#define HIPETS_NS w16
#define HIPETS_R 1
#define HIPETS_LAUNCH_FN launch_rollout_r1_w16
#define HIPETS_OPAQUE_ARGS 1
#define HIPETS_LEAN_ONLY 1
#include "rollout_inst.inc"
