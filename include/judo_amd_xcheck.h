/*
 * judo_amd_xcheck.h -- TEST-BUILD interface of libjudo_amd.so, not part of the drop-in boundary (include/judo_amd.h).
 *
 * The parity suite cross-checks the shipped kernels (generation 3) against two older, independently written kernel generations.  Those live in
 * tests/libjudo_amd_xcheck.so, which is built next to the tests, is never loaded by judo_amd/ and hands its launchers to the product library through the
 * two entry points below.  A deployment neither ships that library nor calls these functions; without it jh_model_set_kernel(m, 1 | 2) returns
 * JH_ERR_UNSUPPORTED and every model runs its generation-3 kernel.
 */
#ifndef JUDO_AMD_XCHECK_H
#define JUDO_AMD_XCHECK_H

#include "judo_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Articulated-body engine kernel generation for this model: 3 (default, the only one in this library) = cooperative kernel on a register diet, two waves per
 * SIMD (leap_cube: hand self-collision, jh_engine_v5.hip; fr3_pick: matrix-free contact Jacobian, jh_engine_v6.hip); 2 = the cooperative kernels of round 1 / 2,
 * one wave per SIMD; 1 = one lane per rollout, an independent second implementation -- both only after jh_register_xcheck (test builds).  "leap_cube" is the model family: leap_cube, leap_cube_down
 * and caltech_leap_cube (the last one only on generation 3: its sensor layout and static-geometry groups exist there alone). */
int jh_model_set_kernel(jh_model* m, int generation);

/* Kernel generations 1 and 2 are cross-check implementations for the parity tests and are NOT part of this library: they live in the test-only
 * libjudo_amd_xcheck.so (built next to the tests), which hands its launchers to this library when it is loaded.  Until then jh_model_set_kernel(m, 1 | 2)
 * returns JH_ERR_UNSUPPORTED.  `generation` is passed through to the launchers; `max_knots` is the LDS staging limit of the one-lane kernel. */
typedef struct jh_xcheck_launchers {
  int (*rollout_cost)(const jh_model* m, int generation, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                      const float* ctrl_lo_hi, const float* task_params, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, void* stream);
  int (*rollout_materialize)(const jh_model* m, int generation, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                             void* stream);
  int (*max_knots)(const jh_model* m, int H);
} jh_xcheck_launchers;
int jh_register_xcheck(const jh_xcheck_launchers* launchers);

#ifdef __cplusplus
}
#endif
#endif
