// rollout_r2.hip -- rollout_kernel with R = 2 row tiles (16 rows each) per workgroup (rollout.hpp): the launcher, the reference-semantics shape-specialised instances and the hidden-static instance.
// One of the four translation units of this R (rollout_inst.inc HIPETS_PART): they compile in parallel.
#define HIPETS_R 2
#define HIPETS_PART 1
#include "rollout_inst.inc"
