// meshlet_eval_contracted.hip — meshlet_eval.hip compiled a second time with ORBIT_CONTRACT = 1 (orbit_device.h madd: every OpDot /
// OpMatrixTimesVector / OpMatrixTimesMatrix / Length / Distance of the cull shaders as an fma chain): the kernels behind
// OrbitCaps.arith_profile = ORBIT_ARITH_CONTRACTED.  The externally visible launchers get a suffix; the canonical ones hand
// a launch whose parameter block says `arith` over to them (kernels.h).  The product's default build is meshlet_eval.hip itself,
// untouched by this file.
#define ORBIT_CONTRACT 1
#define launch_meshlet_eval launch_meshlet_eval_contracted
#define launch_meshlet_eval_views launch_meshlet_eval_views_contracted
#define launch_shard_cull launch_shard_cull_contracted
#include "meshlet_eval.hip"
