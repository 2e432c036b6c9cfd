// jh_policy.hip -- the policy half of the Spot policy rollout (mujoco_extensions/system/system_class.cpp:125-238), batched over rollouts:
//   obs_row       System::setObservation: 84-d observation per rollout from its state, its 25-d command and its previous policy output
//   small_layer   the actor 84 -> 512 -> 256 -> 128 -> 12 (Gemm + Elu, spot_locomotion.onnx): exact-f32 MFMA tiles
//                 (v_mfma_f32_32x32x2_f32; same arithmetic as an fmaf chain, at the f32 vector rate but one VGPR per operand)
//   control_row   System::policyInference's mapping of the 12 actions to the 19 joint targets, arm pass-through, leg override
//   k_policy_step all of it in one launch
// This is the one dense contraction on the path: a batch of N rollouts is an (N x 84) x (84 x 512) ... GEMM chain,
// 0.42 MFLOP per rollout and control step.  The physics substeps between two policy steps are jh_engine_v4.hip (k_tree_v4); jh_policy_rollout there
// alternates the two for a whole rollout.
#include "jh_internal.h"
#include <cstdlib>
#include <vector>

namespace {

constexpr int OBS = 84, H0 = 512, H1 = 256, H2 = 128, ACT = 12, NJ = 19, NCMD = 25;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PolicyTables {  // system_class.cpp:103-123
  int m2o[NJ];         // mujoco_to_orbit.indices(): (P v)[idx[i]] = v[i]
  int o2m_legs[ACT];   // orbit_to_mujoco_legs.indices()
  float default_pos[NJ];
};

__device__ __forceinline__ void rot_vec_quat(float* r, const float* v, const float* q) {  // mju_rotVecQuat
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  r[0] = (1 - 2 * (y * y + z * z)) * v[0] + 2 * (x * y - w * z) * v[1] + 2 * (x * z + w * y) * v[2];
  r[1] = 2 * (x * y + w * z) * v[0] + (1 - 2 * (x * x + z * z)) * v[1] + 2 * (y * z - w * x) * v[2];
  r[2] = 2 * (x * z - w * y) * v[0] + 2 * (y * z + w * x) * v[1] + (1 - 2 * (x * x + y * y)) * v[2];
}

struct ObsLayout { int ld, nq, base_qpos, base_qvel, leg_qpos, leg_qvel, ldc; };

// one thread per rollout; the row is written as 84 consecutive floats
__device__ __forceinline__ void obs_row(const PolicyTables& T, const float* __restrict__ states, const ObsLayout& Y, const float* __restrict__ command,
                                        const float* __restrict__ prev_out, int n, float* __restrict__ obs, int orow = -1 /* row of `obs` to write; default: n */) {
  const int ld = Y.ld, nq = Y.nq, base_qpos = Y.base_qpos, base_qvel = Y.base_qvel, leg_qpos = Y.leg_qpos, leg_qvel = Y.leg_qvel, ldc = Y.ldc;
  const float* qpos = states + (size_t)n * ld; const float* qvel = qpos + nq;
  const float* cmd = command + (size_t)n * ldc;
  float* o = obs + (size_t)(orow < 0 ? n : orow) * OBS;
  const float inv[4] = {qpos[base_qpos + 3], -qpos[base_qpos + 4], -qpos[base_qpos + 5], -qpos[base_qpos + 6]};
  const float lv[3] = {qvel[base_qvel], qvel[base_qvel + 1], qvel[base_qvel + 2]}, g0[3] = {0.f, 0.f, -1.f};
  float lin[3], grav[3];
  rot_vec_quat(lin, lv, inv); rot_vec_quat(grav, g0, inv);
  for (int i = 0; i < 3; i++) { o[i] = lin[i]; o[3 + i] = qvel[base_qvel + 3 + i]; o[6 + i] = grav[i]; o[9 + i] = cmd[i]; o[31 + i] = cmd[22 + i]; }
  for (int i = 0; i < 7; i++) o[12 + i] = cmd[3 + i];
  for (int i = 0; i < 12; i++) { o[19 + i] = cmd[10 + i]; o[72 + i] = prev_out[(size_t)n * ACT + i]; }
  for (int i = 0; i < NJ; i++) { o[34 + T.m2o[i]] = qpos[leg_qpos + i] - T.default_pos[i]; o[53 + T.m2o[i]] = qvel[leg_qvel + i]; }
}
constexpr int SM = 32, BN = 128, BK = 32, LDT = BK + 4;  // row stride 36 floats: 16-byte rows, and eight lanes' float4 of eight consecutive rows cover the 32 banks

__device__ __forceinline__ void control_row(const PolicyTables& T, const float* __restrict__ obs, const float* __restrict__ actions, int n,
                                            float* __restrict__ policy_out, float* __restrict__ control) {
  const float* o = obs + (size_t)n * OBS; const float* a = actions + (size_t)n * ACT;
  float c[NJ];
  for (int i = 0; i < ACT; i++) { const float ai = a[i]; policy_out[(size_t)n * ACT + i] = ai; c[T.o2m_legs[i]] = 0.2f * ai; }
  for (int i = 0; i < ACT; i++) c[i] += T.default_pos[i];
  for (int i = 0; i < 7; i++) c[12 + i] = o[12 + i];
  bool done = false;  // the first leg with a non-zero command overrides its three joint targets (if / else-if chain in the reference)
  for (int leg = 0; leg < 4; leg++) {
    const float x = o[19 + 3 * leg], y = o[20 + 3 * leg], z = o[21 + 3 * leg];
    if (!done && x * x + y * y + z * z > 0.f) { c[3 * leg] = x; c[3 * leg + 1] = y; c[3 * leg + 2] = z; done = true; }
  }
  for (int i = 0; i < NJ; i++) control[(size_t)n * NJ + i] = c[i];
}
// ---- the whole policy step in ONE launch.  A workgroup (4 waves) takes 32 rollouts through the observation, the four layers and the control mapping.
// C (32 x Nout) = act(A (32 x K) W^T + b), W (Nout x K) row-major (the ONNX Gemm layout with transB = 1): a layer's column tiles (32 x 128, one 32 x 32 MFMA
// accumulator per class of K chunks and wave) are walked one after the other, K in chunks of 32 through two LDS buffers (one barrier per chunk) with two more chunks in flight
// from global memory in registers; the activations go through the global scratch (L2-resident).  Operand map of v_mfma_f32_32x32x2_f32: lane l supplies
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; result register v of lane l is C[row = (v & 3) + 8 (v >> 2) + 4 (l >> 5)][col = l & 31].
// History: one launch per layer with 128 x 128 tiles (2 x 2 accumulators per wave) took 510 us at 65 536 rollouts; this kernel with one accumulator 380 us; with the four
// accumulators that let the per-layer launches below reproduce its sums 428 us (two waves per SIMD instead of three), with the two-buffer pipeline 414 us = 59 TFLOP/s, 38 % of
// the f32 MFMA peak: 2 048 workgroups each read all 830 KB of weights, 5.6 TB/s out of the L2.
struct ActorWeights { const float* w[4]; const float* b[4]; };

template <bool ELU>
__device__ __forceinline__ void small_layer(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias, int M, int K, int Nout,
                                            float* __restrict__ C, int m0, float (*sA)[SM * LDT], float (*sW)[BN * LDT]) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
  const int NC = (K + BK - 1) / BK;  // chunks of 32 along K
  for (int n0 = 0; n0 < Nout; n0 += BN) {
    f32x16 acc[4];  // one per class of K chunks (chunk c -> acc[c & 3]): the summation order the per-layer launches below can reproduce with the classes on four waves
#pragma unroll
    for (int u = 0; u < 4; u++)
      for (int v = 0; v < 16; v++) acc[u][v] = 0.f;
    // software pipeline: chunk c is contracted out of LDS buffer c & 1 while chunk c + 1 moves from registers to the other buffer and chunks c + 2, c + 3 are in flight from
    // global memory (two register stages): one barrier per chunk
    f32x4 ra[2], rw[2][4];
    auto fetch = [&](int c, f32x4& fa, f32x4* fw) __attribute__((always_inline)) {
      const int k = c * BK + (tid & 7) * 4;
      {  // A: 32 rows x 8 float4 = one per thread
        const int r = tid >> 3;
        fa = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m0 + r < M && k < K) fa = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r) * K + k);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {  // W: 128 rows x 8 float4 = four per thread
        const int r = (tid + 256 * q) >> 3;
        fw[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n0 + r < Nout && k < K) fw[q] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + r) * K + k);
      }
    };
    auto stage = [&](int buf, const f32x4& fa, const f32x4* fw) __attribute__((always_inline)) {
      *reinterpret_cast<f32x4*>(sA[buf] + (tid >> 3) * LDT + (tid & 7) * 4) = fa;
#pragma unroll
      for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4*>(sW[buf] + ((tid + 256 * q) >> 3) * LDT + (tid & 7) * 4) = fw[q];
    };
    fetch(0, ra[0], rw[0]);
    if (1 < NC) fetch(1, ra[1], rw[1]);
    stage(0, ra[0], rw[0]);
    if (2 < NC) fetch(2, ra[0], rw[0]);
    __syncthreads();
    for (int c0 = 0; c0 < NC; c0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u;
        if (c < NC) {
          // MFMA step t of a chunk contracts k = t (lanes 0..31) and k = 16 + t (lanes 32..63): a lane's 16 operands are 64 contiguous bytes of its row
          const f32x4* pa = reinterpret_cast<const f32x4*>(sA[u & 1] + (l & 31) * LDT + 16 * (l >> 5));
          const f32x4* pw = reinterpret_cast<const f32x4*>(sW[u & 1] + (32 * wave + (l & 31)) * LDT + 16 * (l >> 5));
          f32x4 oa[4], ow[4];
#pragma unroll
          for (int q = 0; q < 4; q++) { oa[q] = pa[q]; ow[q] = pw[q]; }
          if (c + 1 < NC) {
            stage((u + 1) & 1, ra[(u + 1) & 1], rw[(u + 1) & 1]);
            if (c + 3 < NC) fetch(c + 3, ra[(u + 1) & 1], rw[(u + 1) & 1]);
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[q].x, ow[q].x, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[q].y, ow[q].y, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[q].z, ow[q].z, acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[q].w, ow[q].w, acc[u], 0, 0, 0);
          }
          __syncthreads();
        }
      }
    }
    const int col = n0 + 32 * wave + (l & 31);
    if (col < Nout) {
      const float b = bias[col];
#pragma unroll
      for (int v = 0; v < 16; v++) {
        const int row = m0 + (v & 3) + 8 * (v >> 2) + 4 * (l >> 5);
        if (row < M) { float x = ((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) + b; if (ELU) x = x > 0.f ? x : expm1f(x); C[(size_t)row * Nout + col] = x; }
      }
    }
  }
  __syncthreads();  // the layer's activations are visible to the whole workgroup before the next layer reads them
}

__global__ __launch_bounds__(256) void k_policy_step(PolicyTables T, ActorWeights Wt, const float* __restrict__ states, ObsLayout Y, const float* __restrict__ command,
                                                      float* policy_out, int N, float* obs, float* h0, float* h1, float* h2, float* act, float* __restrict__ control) {
  __shared__ __attribute__((aligned(16))) float sA[2][SM * LDT], sW[2][BN * LDT];
  const int m0 = blockIdx.x * SM;
  if ((int)threadIdx.x < SM && m0 + (int)threadIdx.x < N) obs_row(T, states, Y, command, policy_out, m0 + threadIdx.x, obs);
  __syncthreads();
  small_layer<true>(obs, Wt.w[0], Wt.b[0], N, OBS, H0, h0, m0, sA, sW);
  small_layer<true>(h0, Wt.w[1], Wt.b[1], N, H0, H1, h1, m0, sA, sW);
  small_layer<true>(h1, Wt.w[2], Wt.b[2], N, H1, H2, h2, m0, sA, sW);
  small_layer<false>(h2, Wt.w[3], Wt.b[3], N, H2, ACT, act, m0, sA, sW);
  if ((int)threadIdx.x < SM && m0 + (int)threadIdx.x < N) control_row(T, obs, act, m0 + threadIdx.x, policy_out, control);
}


// ---- a few dozen rollouts (the reference ships 24: `judo/tasks/spot/spot_base.py`, num_rollouts): k_policy_step above is ONE workgroup walking 56 weight chunks one
// after the other, 68 us of which 57 k cycles are dependent MFMA chains on four SIMDs.  From ROW_N (below) to SMALL_N rollouts the step is a chain of four launches instead, a layer each,
// its column tiles (32 rollouts x 32 outputs) spread over workgroups and the four classes of K chunks over the tile's four waves: lane (i = l & 31, h = l >> 5) of wave w loads
// floats [16 h, 16 h + 16) of the chunks w, w + 4, ... of the input row i and of the weight row j = i straight into registers (all loads in flight at once: no LDS staging, no
// barrier in the chain), issues 16 MFMA steps per chunk, and the four partial tiles are added as k_policy_step adds its four accumulators: the two paths give the same bits, so a
// rollout's result does not depend on the size of the batch it is in (tests/test_gpu_spot.py::test_policy_rollout_is_reproducible_and_batch_independent).
constexpr int SMALL_N = 2048;  // largest batch that takes the per-layer launches (measured: 34 / 37 / 51 / 80 / 134 us at 512 / 1 024 / 2 048 / 4 096 / 8 192 rollouts against 75 / 76 / 76 / 80 / 84 of k_policy_step)

template <int K, int NOUT, bool ELU, bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void k_policy_layer(PolicyTables T, const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ states, ObsLayout Y,
                                                      const float* __restrict__ command, float* policy_out, int N, float* obs, const float* A, float* __restrict__ C,
                                                      float* __restrict__ control) {
  static_assert(K % 4 == 0, "rows are read as float4");
  constexpr int NCHUNK = (K + BK - 1) / BK, NCW = (NCHUNK + 3) / 4;  // chunks of 32 along K; chunks per wave
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float part[3][16][64];
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, j = l & 31, h = l >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  __shared__ __attribute__((aligned(16))) float sobs[FIRST ? 32 * OBS : 4];
  if (FIRST) {  // every column tile's workgroup builds the observation rows of its row block in its own LDS and contracts out of that copy; the global rows (which the
                // LAST layer's control mapping reads, launches later) are written by the first column tile alone: no workgroup reads what another one writes
    if (tid < 32 && m0 + tid < N) obs_row(T, states, Y, command, policy_out, m0 + tid, sobs, tid);
    __syncthreads();
    if (blockIdx.x == 0)
      for (int e = tid; e < 32 * OBS; e += 256)
        if (m0 + e / OBS < N) obs[(size_t)m0 * OBS + e] = sobs[e];
  }
  const bool arow = m0 + j < N, wrow = n0 + j < NOUT;
  const float* pa = FIRST ? sobs + j * K : A + (size_t)(m0 + j) * K;
  const float* pw = W + (size_t)(n0 + j) * K;
  f32x4 a[NCW][4], b[NCW][4];
#pragma unroll
  for (int m = 0; m < NCW; m++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = BK * (wave + 4 * m) + 16 * h + 4 * q;
      a[m][q] = f32x4{0.f, 0.f, 0.f, 0.f}; b[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (arow && k < K) a[m][q] = *reinterpret_cast<const f32x4*>(pa + k);
      if (wrow && k < K) b[m][q] = *reinterpret_cast<const f32x4*>(pw + k);
    }
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; v++) acc[v] = 0.f;
#pragma unroll
  for (int m = 0; m < NCW; m++)
    if (BK * (wave + 4 * m) < K) {  // (k_policy_step skips a chunk past K too: the same MFMA steps in both)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q].x, b[m][q].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q].y, b[m][q].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q].z, b[m][q].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][q].w, b[m][q].w, acc, 0, 0, 0);
      }
    }
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < 16; v++) part[wave - 1][v][l] = acc[v];
  }
  __syncthreads();
  if (wave == 0) {
    const int col = n0 + j;
    const float bv = col < NOUT ? bias[col] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; v++) {
      const int row = m0 + (v & 3) + 8 * (v >> 2) + 4 * h;
      float x = ((acc[v] + part[0][v][l]) + (part[1][v][l] + part[2][v][l])) + bv;
      if (ELU) x = x > 0.f ? x : expm1f(x);
      if (row < N && col < NOUT) C[(size_t)row * NOUT + col] = x;
    }
  }
  if (LAST) {  // NOUT <= 32: one column tile, so this workgroup holds every action of its rows
    __syncthreads();
    if (tid < 32 && m0 + tid < N) control_row(T, obs, C, m0 + tid, policy_out, control);
  }
}


// ---- a few dozen to a few hundred rollouts (24 in the shipped Spot tasks): even four launches are mostly launch latency (4 x ~4 us of the 29).  Up to ROW_N rollouts the step is ONE launch of
// one workgroup PER ROLLOUT (16 waves) on the vector ALUs: a thread owns one or two class sums of one output of a layer and evaluates them in k_policy_step's order -- chunk c of
// 32 goes to class c & 3, inside a chunk the MFMA step t contracts k = t and then k = 16 + t, one fused multiply-add each, which is what v_mfma_f32_32x32x2_f32 does per output --
// so the bits are those of the other two paths (the batch-independence test crosses all three).  The weights are read from a transposed copy ([k][output]: a wave's loads are
// contiguous), the layer's input is broadcast out of LDS, the class sums of an output meet in LDS and are added in the same order.
constexpr int ROW_N = 512;  // 16-17 us up to 256 rollouts (a workgroup per CU), 29 us at 512; the per-layer launches: 29-34 us

template <int K, int NOUT, int NCLS, int U>  // a thread's share of output o: NCLS of the 4 classes from class cls0 on, summed in class order; U chunks per class in flight
__device__ __forceinline__ float row_dot(const float* __restrict__ Wt, const float* x /* LDS, zero-padded to a multiple of 32 */, int o, int cls0) {
  constexpr int NC = (K + BK - 1) / BK;
  float acc[NCLS];
#pragma unroll
  for (int a = 0; a < NCLS; a++) acc[a] = 0.f;
#pragma unroll 1
  for (int c0 = 0; c0 < NC; c0 += 4 * U) {
    float w[U][NCLS][BK];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int a = 0; a < NCLS; a++) {
        const int c = c0 + 4 * u + cls0 + a;
#pragma unroll
        for (int q = 0; q < BK; q++) { const int k = c * BK + q; w[u][a][q] = (c < NC && k < K) ? Wt[(size_t)k * NOUT + o] : 0.f; }
      }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int t = 0; t < BK / 2; t++)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int a = 0; a < NCLS; a++) {
            const int c = c0 + 4 * u + cls0 + a;
            if (c < NC) acc[a] = __builtin_fmaf(x[c * BK + 16 * h + t], w[u][a][16 * h + t], acc[a]);
          }
  }
  float r = acc[0];
  if (NCLS == 2) r = acc[0] + acc[1];
  if (NCLS == 4) r = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  return r;
}
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

constexpr int ROW_T = 1024;  // threads of a rollout's workgroup: (output, class) pairs of the widest layers
__global__ __launch_bounds__(ROW_T) void k_policy_row(PolicyTables T, ActorWeights Wt /* transposed */, const float* __restrict__ states, ObsLayout Y, const float* __restrict__ command,
                                                      float* policy_out, int N, float* obs, float* act, float* __restrict__ control) {
  __shared__ float x0[96], x1[H0], x2[H1], x3[H2], part[ROW_T];
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) obs_row(T, states, Y, command, policy_out, n, obs);
  __syncthreads();
  if (tid < 96) x0[tid] = tid < OBS ? obs[(size_t)n * OBS + tid] : 0.f;
  __syncthreads();
  part[tid] = row_dot<OBS, H0, 2, 1>(Wt.w[0], x0, tid & (H0 - 1), 2 * (tid >> 9));  // 512 outputs x 2 pairs of classes
  __syncthreads();
  if (tid < H0) x1[tid] = elu1((part[tid] + part[H0 + tid]) + Wt.b[0][tid]);
  __syncthreads();
  part[tid] = row_dot<H0, H1, 1, 2>(Wt.w[1], x1, tid & (H1 - 1), tid >> 8);  // 256 outputs x 4 classes
  __syncthreads();
  if (tid < H1) x2[tid] = elu1(((part[tid] + part[H1 + tid]) + (part[2 * H1 + tid] + part[3 * H1 + tid])) + Wt.b[1][tid]);
  __syncthreads();
  if (tid < 4 * H2) part[tid] = row_dot<H1, H2, 1, 2>(Wt.w[2], x2, tid & (H2 - 1), tid >> 7);  // 128 outputs x 4 classes
  __syncthreads();
  if (tid < H2) x3[tid] = elu1(((part[tid] + part[H2 + tid]) + (part[2 * H2 + tid] + part[3 * H2 + tid])) + Wt.b[2][tid]);
  __syncthreads();
  if (tid < 4 * ACT) part[tid] = row_dot<H2, ACT, 1, 1>(Wt.w[3], x3, tid >> 2, tid & 3);  // 12 outputs x 4 classes
  __syncthreads();
  if (tid < ACT) act[(size_t)n * ACT + tid] = ((part[4 * tid] + part[4 * tid + 1]) + (part[4 * tid + 2] + part[4 * tid + 3])) + Wt.b[3][tid];
  __syncthreads();
  if (tid == 0) control_row(T, obs, act, n, policy_out, control);
}

}  // namespace

struct jh_policy { float* d_w[4]; float* d_wt[4]; float* d_b[4]; PolicyTables tab; };  // d_wt: the weights transposed ([in][out]) for k_policy_row

extern "C" int jh_policy_create(const float* const* weights /* W0..W3, (out,in) row-major */, const float* const* biases, jh_policy** out) {
  JH_REQUIRE(weights && biases && out, "policy_create: null pointer");
  const int dims[5] = {OBS, H0, H1, H2, ACT};
  jh_policy* p = new jh_policy();
  for (int i = 0; i < 4; i++) {
    JH_REQUIRE(weights[i] && biases[i], "policy_create: null layer %d", i);
    JH_HIP(hipMalloc(&p->d_w[i], sizeof(float) * dims[i] * dims[i + 1]));
    JH_HIP(hipMalloc(&p->d_b[i], sizeof(float) * dims[i + 1]));
    JH_HIP(hipMemcpy(p->d_w[i], weights[i], sizeof(float) * dims[i] * dims[i + 1], hipMemcpyHostToDevice));
    std::vector<float> tr((size_t)dims[i] * dims[i + 1]);
    for (int o = 0; o < dims[i + 1]; o++) for (int k = 0; k < dims[i]; k++) tr[(size_t)k * dims[i + 1] + o] = weights[i][(size_t)o * dims[i] + k];
    JH_HIP(hipMalloc(&p->d_wt[i], sizeof(float) * tr.size()));
    JH_HIP(hipMemcpy(p->d_wt[i], tr.data(), sizeof(float) * tr.size(), hipMemcpyHostToDevice));
    JH_HIP(hipMemcpy(p->d_b[i], biases[i], sizeof(float) * dims[i + 1], hipMemcpyHostToDevice));
  }
  const int m2o[NJ] = {1, 6, 11, 2, 7, 12, 3, 8, 13, 4, 9, 14, 0, 5, 10, 15, 16, 17, 18};
  const int o2m[ACT] = {0, 3, 6, 9, 1, 4, 7, 10, 2, 5, 8, 11};
  const float dpos[NJ] = {0.12f, 0.5f, -1.f, -0.12f, 0.5f, -1.f, 0.12f, 0.5f, -1.f, -0.12f, 0.5f, -1.f, 0.f, -0.9f, 1.8f, 0.f, -0.9f, 0.f, -1.54f};
  for (int i = 0; i < NJ; i++) { p->tab.m2o[i] = m2o[i]; p->tab.default_pos[i] = dpos[i]; }
  for (int i = 0; i < ACT; i++) p->tab.o2m_legs[i] = o2m[i];
  *out = p;
  return JH_OK;
}

extern "C" void jh_policy_destroy(jh_policy* p) {
  if (!p) return;
  for (int i = 0; i < 4; i++) { (void)hipFree(p->d_w[i]); (void)hipFree(p->d_wt[i]); (void)hipFree(p->d_b[i]); }
  delete p;
}

extern "C" size_t jh_policy_scratch_floats(int N) { return (size_t)(N > 0 ? N : 0) * (OBS + H0 + H1 + H2 + ACT); }

// One policy step with row strides for the states (ld; 0 = one state broadcast to every rollout) and the commands (ldc): the form the rollout loop uses.
int jh_policy_step_strided(const jh_policy* p, const float* states, int ld, int nq, int base_qpos, int base_qvel, int leg_qpos, int leg_qvel, const float* command, int ldc,
                           float* policy_out, float* control, float* scratch, int N, hipStream_t st) {
  float* obs = scratch; float* h0 = obs + (size_t)N * OBS; float* h1 = h0 + (size_t)N * H0; float* h2 = h1 + (size_t)N * H1; float* act = h2 + (size_t)N * H2;
  const ObsLayout Y = {ld, nq, base_qpos, base_qvel, leg_qpos, leg_qvel, ldc};
  ActorWeights Wt; for (int i = 0; i < 4; i++) { Wt.w[i] = p->d_w[i]; Wt.b[i] = p->d_b[i]; }
  static const int layers_max = [] { const char* e = getenv("JUDO_AMD_POLICY_LAYERS_MAX"); return e && e[0] ? atoi(e) : SMALL_N; }();  // diagnostic override of the switch-over
  static const int rows_max = [] { const char* e = getenv("JUDO_AMD_POLICY_ROWS_MAX"); return e && e[0] ? atoi(e) : ROW_N; }();
  if (N <= rows_max) {
    ActorWeights Wr; for (int i = 0; i < 4; i++) { Wr.w[i] = p->d_wt[i]; Wr.b[i] = p->d_b[i]; }
    hipLaunchKernelGGL(k_policy_row, dim3(N), dim3(ROW_T), 0, st, p->tab, Wr, states, Y, command, policy_out, N, obs, act, control);
  } else if (N <= layers_max) {
    const int rb = (N + 31) / 32;
    hipLaunchKernelGGL((k_policy_layer<OBS, H0, true, true, false>), dim3(H0 / 32, rb), dim3(256), 0, st, p->tab, Wt.w[0], Wt.b[0], states, Y, command, policy_out, N, obs, obs, h0, control);
    hipLaunchKernelGGL((k_policy_layer<H0, H1, true, false, false>), dim3(H1 / 32, rb), dim3(256), 0, st, p->tab, Wt.w[1], Wt.b[1], states, Y, command, policy_out, N, obs, h0, h1, control);
    hipLaunchKernelGGL((k_policy_layer<H1, H2, true, false, false>), dim3(H2 / 32, rb), dim3(256), 0, st, p->tab, Wt.w[2], Wt.b[2], states, Y, command, policy_out, N, obs, h1, h2, control);
    hipLaunchKernelGGL((k_policy_layer<H2, ACT, false, false, true>), dim3(1, rb), dim3(256), 0, st, p->tab, Wt.w[3], Wt.b[3], states, Y, command, policy_out, N, obs, h2, act, control);
  } else {
    hipLaunchKernelGGL(k_policy_step, dim3((N + SM - 1) / SM), dim3(256), 0, st, p->tab, Wt, states, Y, command, policy_out, N, obs, h0, h1, h2, act, control);
  }
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_policy_step(const jh_policy* p, const float* states, int ld, int nq, int base_qpos, int base_qvel, int leg_qpos, int leg_qvel,
                              const float* command, float* policy_out, float* control, float* scratch, int N, void* stream) {
  JH_REQUIRE(p && states && command && policy_out && control && scratch, "policy_step: null pointer");
  JH_REQUIRE(N > 0, "policy_step: need at least one rollout");
  JH_REQUIRE(nq > 0 && ld >= nq + leg_qvel + NJ && base_qpos >= 0 && base_qpos + 7 <= nq && leg_qpos >= 0 && leg_qpos + NJ <= nq && base_qvel >= 0 && leg_qvel >= 0,
             "policy_step: state layout (ld=%d nq=%d base %d/%d joints %d/%d) does not hold a free base and 19 joints", ld, nq, base_qpos, base_qvel, leg_qpos, leg_qvel);
  return jh_policy_step_strided(p, states, ld, nq, base_qpos, base_qvel, leg_qpos, leg_qvel, command, NCMD, policy_out, control, scratch, N, (hipStream_t)stream);
}
