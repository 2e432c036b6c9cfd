// jh_internal.h -- shared definitions of libjudo_amd.so (gfx950 only; no other backend exists).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "judo_amd.h"
#include "judo_amd_xcheck.h"  // (the test-build hooks: the cross-check kernel generations register through them)

#define JH_BLOB_MAGIC 0x314D484Au /* "JHM1" */
#define JH_BLOB_VERSION 1u
#define JH_NSTATS 512

// Host-side blob header produced by judo_amd/models.py::pack_model (little-endian, 64 bytes):
struct jh_blob_header {
  uint32_t magic, version, kind, nq, nv, nu, ns, ntaskparam, nfloat, nint;
  uint32_t reserved[6];
};

struct jh_model {
  int device, kind, nq, nv, nu, ns, ntaskparam;
  size_t nf, ni;
  float* d_f;  // device copy of the float section
  int* d_i;    // device copy of the int section
  int kernel_gen;  // articulated engine kernel: 3 = cooperative 16-lanes-per-rollout, two waves per SIMD (leap_cube: v5; fr3_pick: v6, matrix-free contact Jacobian), 2 = cooperative, one wave per SIMD (leap_cube: v2, fr3_pick: v3), 1 = one lane per rollout
  int contact_capacity;  // leap_cube generation 3: 48 (all in LDS, jh_engine_v5.hip) or 64 (jh_engine_v5_cap64.hip); jh_model_set_contact_capacity
  int self_collision;  // leap_cube on jh_engine_v5.hip: model the hand's own contacts (finger-finger, finger-palm) as MuJoCo does; 0 = the cube's contacts only
  mutable int ovf_fallbacks = 0;  // launches that ran without their overflow rows (jh_launch_scratch); updated with __atomic builtins: a planner thread may launch while another polls jh_model_stats
  int* d_stats;  // JH_NSTATS diagnostic counters: [0..3] contact-cap overflows, Newton iteration-cap hits, Newton iterations, steps; [20..21] wave-level iterations, steps; the rest: diagnostic builds
  std::vector<float> h_f;
  std::vector<int> h_i;
};

void jh_set_error(const char* fmt, ...);

// Per-launch scratch of the cooperative kernels (the contacts above the LDS pool, one row per rollout): a stream-ordered allocation from the default pool of the MODEL's
// device (not of whatever device is current in the calling thread).  nullptr when the pool refuses: the kernel then holds what its LDS pool holds, the drops are
// counted, and the launch is counted in `ovf_fallbacks` (jh_model_stats out[6]) so that the lower capacity does not go unnoticed.
inline float* jh_launch_scratch(const jh_model* m, size_t bytes, hipStream_t st) {
  hipMemPool_t pool = nullptr; void* p = nullptr;
  if (hipDeviceGetDefaultMemPool(&pool, m->device) == hipSuccess && pool && hipMallocFromPoolAsync(&p, bytes, pool, st) == hipSuccess) return (float*)p;
  (void)hipGetLastError();
  if (__atomic_fetch_add(&m->ovf_fallbacks, 1, __ATOMIC_RELAXED) == 0)  // the first fallback of a model is also logged: the lower capacity must not depend on somebody polling the counter
    fprintf(stderr, "judo_amd: no stream-ordered scratch for the contacts above the LDS pool (%zu bytes): this launch runs with the LDS capacity alone; see jh_model_stats out[6]\n", bytes);
  return nullptr;
}

#define JH_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess) {                                                              \
      jh_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return JH_ERR_HIP;                                                                  \
    }                                                                                     \
  } while (0)

// end of a cooperative launcher: the launch's error is read FIRST, the stream-ordered overflow block is freed whether the launch succeeded or not, then the error is reported
inline int jh_launch_done(float* ovf, hipStream_t st) {
  const hipError_t e = hipGetLastError();
  const hipError_t f = ovf ? hipFreeAsync(ovf, st) : hipSuccess;
  if (e != hipSuccess) { jh_set_error("kernel launch failed: %s", hipGetErrorString(e)); return JH_ERR_HIP; }
  if (f != hipSuccess) { jh_set_error("hipFreeAsync failed: %s", hipGetErrorString(f)); return JH_ERR_HIP; }
  return JH_OK;
}

#define JH_REQUIRE(cond, ...)    \
  do {                           \
    if (!(cond)) {               \
      jh_set_error(__VA_ARGS__); \
      return JH_ERR_INVALID;     \
    }                            \
  } while (0)

// ---- float-section layouts of the two closed-form models (written by judo_amd/models.py) -------------------
// cartpole (judo/models/xml/cartpole.xml): 2 DoF, contacts disabled, Euler + implicit joint damping
enum {
  CP_DT = 0, CP_G, CP_MCART, CP_MPOLE, CP_L, CP_IPOLE, CP_DAMP_X, CP_DAMP_TH, CP_KP, CP_KV, CP_CTRL_LO, CP_CTRL_HI, CP_CTRL_LIMITED,
  CP_FRC_LO, CP_FRC_HI, CP_FRC_LIMITED, CP_X_LO, CP_X_HI, CP_X_LIMITED, CP_LIM_K, CP_LIM_B, CP_SOLIMP0, CP_SOLIMP1, CP_SOLIMP2,
  CP_SOLIMP3, CP_SOLIMP4, CP_INVW_X, CP_TIP, CP_NPARAM
};
// cylinder_push (judo/models/xml/cylinder_push.xml): 4 slide DoF, one circle-circle contact, Euler + implicit damping
enum {
  CY_DT = 0, CY_MP, CY_MC, CY_DAMP_P, CY_DAMP_C, CY_KP, CY_KV, CY_CTRL_LO, CY_CTRL_HI, CY_CTRL_LIMITED, CY_FRC_LO, CY_FRC_HI,
  CY_FRC_LIMITED, CY_RSUM, CY_CON_K, CY_CON_B, CY_SOLIMP0, CY_SOLIMP1, CY_SOLIMP2, CY_SOLIMP3, CY_SOLIMP4, CY_TRAN, CY_MU, CY_SITE_Z,
  CY_MARGIN, CY_NPARAM
};

// ---- launchers implemented per translation unit ------------------------------------------------------------
int jh_simple_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                           const float* W, const float* lohi, const float* tp, int N, int n_offset, int H, int K, float* costs,
                           float* knots_out, float* trace, hipStream_t st);
// Latency mode of the cooperative kernels (rows of `lanes` lanes, `rpw` rows per wave): a launch too small to give every SIMD a wave lets 1 << shift rows of a wave
// compute the same rollout -- the copies run the same arithmetic, only the first writes -- so that a wave no longer waits for the slowest of `rpw` different Newton
// solves in every step.  Returns the largest shift (<= log2 rpw) that still leaves every wave of the launch a SIMD of its own; JUDO_AMD_LATENCY_SHIFT=0..2 overrides.
int jh_latency_shift(int N, int rpw);
int jh_simple_max_knots(const jh_model* m, int H);  // largest fused K at horizon H (LDS staging budget of the launcher)
int jh_engine_max_knots(const jh_model* m, int H);
int jh_simple_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states,
                          float* sensors, hipStream_t st);
int jh_simple_reward(const jh_model* m, const float* states, const float* controls, const float* tp, int N, int H, float* rewards, hipStream_t st);

int jh_engine_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                           const float* W, const float* lohi, const float* tp, int phase, int N, int n_offset, int H, int K, float* costs,
                           float* knots_out, hipStream_t st);
int jh_engine_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states,
                          float* sensors, hipStream_t st);
int jh_engine_reward(const jh_model* m, const float* states, const float* sensors, const float* controls, const float* tp, int phase, int N,
                     int H, float* rewards, hipStream_t st);

int jh_engine2_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                            const float* W, const float* lohi, const float* tp, int N, int n_offset, int H, int K, float* costs, float* knots_out,
                            hipStream_t st);
int jh_engine2_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                           hipStream_t st);

// jh_engine_v5.hip: the leap_cube cooperative kernel on a register diet (several waves per SIMD)
int jh_engine5_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                            const float* lohi, const float* tp, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, hipStream_t st);
int jh_engine5_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                           hipStream_t st);
int jh_engine5_rollout_cost_cap64(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                                  const float* lohi, const float* tp, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, hipStream_t st);
int jh_engine5_materialize_cap64(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                                 hipStream_t st);

// jh_engine_v3.hip: cooperative kernel for fr3_pick (serial arm with a two-finger fork + free box, pyramidal cones)
bool jh_model_is_fr3(const jh_model* m);
int jh_engine3_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                            const float* lohi, const float* tp, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, hipStream_t st);
int jh_engine3_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                           hipStream_t st);

// jh_engine_v6.hip: fr3_pick, matrix-free contact Jacobian (kernel generation 3 of the fr3 model)
int jh_engine6_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                            const float* lohi, const float* tp, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, hipStream_t st);
int jh_engine6_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                           hipStream_t st);

// ---- device helpers ------------------------------------------------------------------------------------------
// MuJoCo's impedance curve d(x) (solimp = dmin, dmax, width, midpoint, power), x = |pos - margin| / width.
__device__ __forceinline__ float jh_impedance(float s0, float s1, float s2, float s3, float s4, float dist) {
  if (s0 == s1 || s2 <= 1e-15f) return 0.5f * (s0 + s1);
  float x = fabsf(dist / s2);
  if (x >= 1.f) return s1;
  if (x <= 0.f) return s0;
  float y;
  if (s4 == 1.f) y = x;
  else if (x <= s3) y = __powf(x, s4) / __powf(s3, s4 - 1.f);
  else y = 1.f - __powf(1.f - x, s4) / __powf(1.f - s3, s4 - 1.f);
  return s0 + y * (s1 - s0);
}

__device__ __forceinline__ float jh_clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }

// jh_policy.hip: one policy step with explicit row strides (used by the policy rollout loop in jh_engine_v4.hip)
struct jh_policy;
int jh_policy_step_strided(const jh_policy* p, const float* states, int ld, int nq, int base_qpos, int base_qvel, int leg_qpos, int leg_qvel, const float* command, int ldc,
                           float* policy_out, float* control, float* scratch, int N, hipStream_t st);
