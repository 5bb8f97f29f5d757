// planet_w16.hip -- the PlaNet latent rollout kernel with 16 waves per workgroup (small populations: pop 1000 x 1 particle is 63
// workgroups); compiled into hipets::w16, see launch.hpp.
#define HIPETS_WAVES 16
#define HIPETS_NS w16
#define HIPETS_OPAQUE_ARGS 1
#include "planet.hip"
