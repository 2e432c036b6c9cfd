// jh_engine_v5_cap64.hip -- the leap kernel (jh_engine_v5.hip) instantiated a second time with a contact capacity of 64 per rollout: the 16 contacts above the LDS pool
// live in a per-rollout row of global memory and the rare copy of the solver runs four slots per lane.  A separate translation unit (compiled in parallel with the first)
// because the larger copy costs the common path 2.8 %: the headline model keeps the 48-contact build, the models whose SHIPPED workloads need more
// -- leap_cube_down (the cube caged under the palm) and caltech_leap_cube: 2e-4 .. 4e-4 contacts dropped per rollout-step at 48 -- select this one
// (jh_model_set_contact_capacity).
#define JH_V5_NSBIG 4
#define JH_V5_NAME(f) f##_cap64
#include "jh_engine_v5.hip"
