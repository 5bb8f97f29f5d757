// rollout_r3_fast.hip -- rollout_kernel with R = 3 row tiles (16 rows each) per workgroup (rollout.hpp): the FAST-mode shape-specialised instances.
// One of the four translation units of this R (rollout_inst.inc HIPETS_PART): they compile in parallel.
#define HIPETS_R 3
#define HIPETS_PART 2
#include "rollout_inst.inc"
