// jh_xcheck.hip -- TEST BUILD ONLY (libjudo_amd_xcheck.so): hands the cross-check kernel generations to the product library.
//
// Generation 1 (jh_engine.hip: one lane per rollout, model-generic, the independent second GPU implementation of the articulated step) and generation 2
// (jh_engine_v2.hip / jh_engine_v3.hip: the cooperative kernels of rounds 1 and 2, one wave per SIMD) are what the parity tests compare the shipped kernels
// with; the product library does not contain them.  Loading this library and calling jh_xcheck_register() installs their launchers
// through jh_register_xcheck (include/judo_amd_xcheck.h).
#include "jh_internal.h"

static int xc_cost(const jh_model* m, int gen, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W, const float* lohi,
                   const float* tp, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (gen == 2 && m->kind == JH_TASK_FR3_PICK) return jh_engine3_rollout_cost(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, phase, N, n_offset, H, K, costs, knots_out, st);
  if (gen == 2 && m->kind == JH_TASK_LEAP_CUBE) return jh_engine2_rollout_cost(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs, knots_out, st);
  return jh_engine_rollout_cost(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, phase, N, n_offset, H, K, costs, knots_out, st);
}

static int xc_materialize(const jh_model* m, int gen, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (gen == 2 && m->kind == JH_TASK_FR3_PICK) return jh_engine3_materialize(m, x0, x0_batched, controls, N, H, states, sensors, st);
  if (gen == 2 && m->kind == JH_TASK_LEAP_CUBE) return jh_engine2_materialize(m, x0, x0_batched, controls, N, H, states, sensors, st);
  return jh_engine_materialize(m, x0, x0_batched, controls, N, H, states, sensors, st);
}

extern "C" int jh_xcheck_register(void) {
  const jh_xcheck_launchers t = {xc_cost, xc_materialize, jh_engine_max_knots};
  return jh_register_xcheck(&t);
}
