// placeholder until the articulated-body engine lands
#include "jh_internal.h"
int jh_engine_rollout_cost(const jh_model*, const float*, const float*, const float*, int, const float*, const float*, const float*, const float*, int, int, int, int, int, float*, float*, hipStream_t) { jh_set_error("engine not built"); return JH_ERR_UNSUPPORTED; }
int jh_engine_materialize(const jh_model*, const float*, int, const float*, int, int, float*, float*, hipStream_t) { jh_set_error("engine not built"); return JH_ERR_UNSUPPORTED; }
int jh_engine_reward(const jh_model*, const float*, const float*, const float*, const float*, int, int, int, float*, hipStream_t) { jh_set_error("engine not built"); return JH_ERR_UNSUPPORTED; }
