// entity_cull_contracted.hip — entity_cull.hip compiled a second time with ORBIT_CONTRACT = 1 (orbit_device.h madd: every OpDot /
// OpMatrixTimesVector / OpMatrixTimesMatrix / Length / Distance of the cull shaders as an fma chain): the kernels behind
// OrbitCaps.arith_profile = ORBIT_ARITH_CONTRACTED.  The externally visible launchers get a suffix; the canonical ones hand
// a launch whose parameter block says `arith` over to them (kernels.h).  The product's default build is entity_cull.hip itself,
// untouched by this file.
#define ORBIT_CONTRACT 1
#define launch_entity_cull launch_entity_cull_contracted
#define launch_entity_cull_views launch_entity_cull_views_contracted
#define launch_split_records launch_split_records_contracted
#define launch_split_records_views launch_split_records_views_contracted
#include "entity_cull.hip"
