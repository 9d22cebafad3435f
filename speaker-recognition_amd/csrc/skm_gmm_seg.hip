// Explicit instantiations only: see skm_gmm_kernels.h.
#include "skm_gmm_kernels.h"

SKM_SEG_TEAM_INSTANCES_A(SKM_DEFINE_SEG_TEAM)
