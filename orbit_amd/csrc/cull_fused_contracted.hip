// cull_fused_contracted.hip — cull_fused.hip compiled a second time with ORBIT_CONTRACT = 1 (orbit_device.h madd: every OpDot /
// OpMatrixTimesVector / OpMatrixTimesMatrix / Length / Distance of the cull shaders as an fma chain): the kernels behind
// OrbitCaps.arith_profile = ORBIT_ARITH_CONTRACTED.  The externally visible launchers get a suffix; the canonical ones hand
// a launch whose parameter block says `arith` over to them (kernels.h).  The product's default build is cull_fused.hip itself,
// untouched by this file.
#define ORBIT_CONTRACT 1
#define launch_cull_fused_views launch_cull_fused_views_contracted
#include "cull_fused.hip"
