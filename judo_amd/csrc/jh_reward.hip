// jh_reward.hip -- Task.reward of the articulated tasks on materialised device arrays (judo/tasks/leap_cube.py:63-88, fr3_pick.py:225-311): one lane per
// rollout walks its H rows; the per-step terms are the ones the fused kernels accumulate (jh_engine_common.h).
#include "jh_engine_common.h"

using namespace jh_eng;

namespace {
constexpr int kBlock = 64;

__global__ __launch_bounds__(kBlock) void k_leap_reward(const float* __restrict__ states, const float* __restrict__ tp, int N, int H, int nx,
                                                        float* __restrict__ rewards) {
  __shared__ float sTp[9];
  if (threadIdx.x < 9) sTp[threadIdx.x] = tp[threadIdx.x];
  __syncthreads();
  const int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int h = 0; h < H; h++) {
    float q[7];
    for (int i = 0; i < 7; i++) q[i] = states[((size_t)n * H + h) * nx + i];
    acc += leap_step_cost(sTp, q);
  }
  rewards[n] = -acc / (float)H;
}

__global__ __launch_bounds__(kBlock) void k_fr3_reward(const float* __restrict__ states, const float* __restrict__ sensors, const float* __restrict__ tp,
                                                       int phase, int N, int H, float* __restrict__ rewards) {
  __shared__ float sTp[22];
  if (threadIdx.x < 22) sTp[threadIdx.x] = tp[threadIdx.x];
  __syncthreads();
  const int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int h = 0; h < H; h++) {
    float x[31], y[14];
    for (int i = 0; i < 31; i++) x[i] = states[((size_t)n * H + h) * 31 + i];
    for (int i = 0; i < 14; i++) y[i] = sensors[((size_t)n * H + h) * 14 + i];
    acc += fr3_step_cost(sTp, phase, x, x + 16, 15, y, H > 1 ? 1.f - (float)h / (float)(H - 1) : 1.f);
  }
  rewards[n] = -acc;
}
}  // namespace

int jh_engine_reward(const jh_model* m, const float* states, const float* sensors, const float* controls, const float* tp, int phase, int N, int H,
                     float* rewards, hipStream_t st) {
  (void)controls;
  int grid = (N + kBlock - 1) / kBlock;
  if (m->kind == JH_TASK_LEAP_CUBE) hipLaunchKernelGGL(k_leap_reward, dim3(grid), dim3(kBlock), 0, st, states, tp, N, H, m->nq + m->nv, rewards);
  else if (m->kind == JH_TASK_FR3_PICK) {
    JH_REQUIRE(sensors != nullptr, "task_reward: fr3_pick needs the sensor array");
    JH_REQUIRE(phase >= 0 && phase <= 3, "task_reward: fr3_pick phase must be 0..3 (got %d)", phase);
    hipLaunchKernelGGL(k_fr3_reward, dim3(grid), dim3(kBlock), 0, st, states, sensors, tp, phase, N, H, rewards);
  } else { jh_set_error("task_reward: unknown articulated task"); return JH_ERR_UNSUPPORTED; }
  JH_HIP(hipGetLastError());
  return JH_OK;
}
