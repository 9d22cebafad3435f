// Explicit instantiations only: see skm_gmm_kernels.h.
#include "skm_gmm_kernels.h"

SKM_ESTEP_TEAM_INSTANCES(SKM_DEFINE_ESTEP_TEAM)
