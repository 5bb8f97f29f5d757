// rollout_r1_gen.hip -- rollout_kernel with R = 1 row tiles (16 rows each) per workgroup (rollout.hpp): the fully generic instance (activation read at run time).
// One of the four translation units of this R (rollout_inst.inc HIPETS_PART): they compile in parallel.
#define HIPETS_R 1
#define HIPETS_PART 3
#include "rollout_inst.inc"
