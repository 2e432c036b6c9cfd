// jh_update.hip -- sample / update kernels of the three optimizers (gfx950, wave64 reductions).
//
//   MPPI  judo/optimizers/mppi.py:76-82   beta = min c; w = exp(-(c-beta)/lambda); nominal = sum w*knots / sum w
//   CEM   judo/optimizers/cem.py:88-92    k best by reward -> mean, clipped population std
//   PS    judo/optimizers/ps.py:64-65     argmax reward
//
// Shard-local kernels emit a small record; a merge kernel combines G records after the all-gather (G = 1 on one GPU).
// The exponential weighting is a two-level reduction: wave64 butterflies (`__shfl_xor`, lowered to DPP/permute) then LDS
// across the four waves of a workgroup, one partial record per workgroup, finished by a single-workgroup merge with a
// log-sum-exp rescale so that no global minimum pass is needed first.
#include "jh_internal.h"

namespace {

constexpr int kUB = 256;  // threads per workgroup (4 waves)

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// knot (idx = k*nu+u) of local rollout n: explicit (N,K,nu) array, or recomputed clip(nominal + sigma*noise)
struct KnotSrc {
  const float* knots_nku; const float* nominal; const float* noise; const float* sigma; const float* lohi;
  int ldn, n_offset, KU, nu;
  __device__ __forceinline__ float get(int n, int idx) const {
    if (knots_nku) return knots_nku[(size_t)n * KU + idx];
    float v = nominal[idx];
    if (n_offset + n != 0) v = fmaf(sigma[idx], noise[(size_t)idx * ldn + n], v);
    if (lohi) { int u = idx % nu; v = jh_clampf(v, lohi[u], lohi[nu + u]); }
    return v;
  }
};

__global__ __launch_bounds__(kUB) void k_sample_knots(KnotSrc src, int N, float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * kUB + threadIdx.x;
  if (i >= (size_t)N * src.KU) return;
  int n = (int)(i / src.KU), idx = (int)(i % src.KU);
  out[i] = src.get(n, idx);
}

// Candidate control trajectories for the materialise path (judo/controller/controller.py:239-249: candidate splines evaluated
// at the rollout times): controls[n,h,u] = sum_k W[h,k] * knot(n,k,u).  A wave owns 64 rollouts: their K*nu knots go to LDS
// (lane-fastest noise reads, odd row stride), then each rollout's contiguous H*nu floats are written with lane-consecutive addresses.
__global__ __launch_bounds__(64) void k_spline_controls(KnotSrc src, const float* __restrict__ W, int N, int H, int K, float* __restrict__ out) {
  extern __shared__ float sm[];
  const int KU = src.KU, nu = src.nu, SK = KU | 1;
  float* sW = sm;            // H*K
  float* sK = sm + H * K;    // 64 * SK
  const int lane = threadIdx.x, n0 = blockIdx.x * 64, nvalid = min(64, N - n0);
  for (int i = lane; i < H * K; i += 64) sW[i] = W[i];
  if (lane < nvalid) for (int idx = 0; idx < KU; idx++) sK[lane * SK + idx] = src.get(n0 + lane, idx);
  __syncthreads();
  const int row = H * nu;
  for (int f = lane; f < nvalid * row; f += 64) {
    const int r = f / row, i = f - r * row, h = i / nu, u = i - h * nu;
    float v = 0.f;
    for (int k = 0; k < K; k++) v = fmaf(sW[h * K + k], sK[r * SK + k * nu + u], v);
    out[(size_t)(n0 + r) * row + i] = v;
  }
}

// Moments of the candidate knots per actuator for the running action normaliser (judo/utils/normalization.py:176-200 on
// `candidate_knots`): out[u] += sum_{n,k} (x - center[u]), out[nu+u] += sum_{n,k} (x - center[u])^2.  One wave per 64 rollouts, wave
// butterflies, one atomic per (block, actuator, moment).
__global__ __launch_bounds__(64) void k_knot_moments(KnotSrc src, const float* __restrict__ center, int N, int K, float* __restrict__ out) {
  const int n = blockIdx.x * 64 + threadIdx.x, nu = src.nu;
  const bool live = n < N;
  for (int u = 0; u < nu; u++) {
    float s1 = 0.f, s2 = 0.f;
    if (live) for (int k = 0; k < K; k++) { float d = src.get(n, k * nu + u) - center[u]; s1 += d; s2 = fmaf(d, d, s2); }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (threadIdx.x == 0) { atomicAdd(out + u, s1); atomicAdd(out + nu + u, s2); }
  }
}

// ---------------------------------------------------------------- MPPI
// (the bodies of the update kernels are device functions: the one-launch tail k_update_tail below runs the same arithmetic in the same order -- bit-identical results)
__device__ __forceinline__ void mppi_block_body(const float* __restrict__ costs, const KnotSrc& src, int N, float inv_lambda, float* __restrict__ scratch, float* sred,
                                                float (*sV)[JH_MAX_KNOT_DIM]) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = blockIdx.x * kUB + tid;
  const bool live = n < N;
  float c = live ? costs[n] : INFINITY;
  if (!(c == c)) c = INFINITY;  // a NaN cost (diverged rollout) gets zero weight
  float m = wave_min(c);
  if (lane == 0) sred[wave] = m;
  __syncthreads();
  float beta = fminf(fminf(sred[0], sred[1]), fminf(sred[2], sred[3]));
  __syncthreads();
  float w = (live && c < INFINITY) ? __expf(-(c - beta) * inv_lambda) : 0.f;
  float s = wave_sum(w);
  if (lane == 0) sred[wave] = s;
  const int nc = live ? n : 0;
  for (int idx = 0; idx < src.KU; idx++) {
    float v = wave_sum(w * src.get(nc, idx));
    if (lane == 0) sV[wave][idx] = v;
  }
  __syncthreads();
  float* rec = scratch + (size_t)blockIdx.x * (2 + src.KU);
  if (tid == 0) { rec[0] = beta; rec[1] = sred[0] + sred[1] + sred[2] + sred[3]; }
  for (int idx = tid; idx < src.KU; idx += kUB) rec[2 + idx] = sV[0][idx] + sV[1][idx] + sV[2][idx] + sV[3][idx];
}
__global__ __launch_bounds__(kUB) void k_mppi_block(const float* __restrict__ costs, KnotSrc src, int N, float inv_lambda, float* __restrict__ scratch) {
  __shared__ float sred[4];
  __shared__ float sV[4][JH_MAX_KNOT_DIM];
  mppi_block_body(costs, src, N, inv_lambda, scratch, sred, sV);
}

// merge nrec records [beta, S, V...] -> one record, or (finalize) the nominal knots V/S
// (`stride`: floats from one record to the next -- 2 + KU for packed records, the length of a rank's whole record when the update record is followed by the trace records)
__device__ __forceinline__ void mppi_merge_body(const float* __restrict__ recs, int nrec, int KU, float inv_lambda, int finalize, float* __restrict__ out, float* sred, float& sS,
                                                int stride = 0) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (stride == 0) stride = 2 + KU;
  float m = INFINITY;
  for (int r = tid; r < nrec; r += kUB) m = fminf(m, recs[(size_t)r * stride]);
  m = wave_min(m);
  if (lane == 0) sred[wave] = m;
  __syncthreads();
  float beta = fminf(fminf(sred[0], sred[1]), fminf(sred[2], sred[3]));
  __syncthreads();
  float s = 0.f;
  for (int r = tid; r < nrec; r += kUB) s += __expf(-(recs[(size_t)r * stride] - beta) * inv_lambda) * recs[(size_t)r * stride + 1];
  s = wave_sum(s);
  if (lane == 0) sred[wave] = s;
  __syncthreads();
  if (tid == 0) sS = sred[0] + sred[1] + sred[2] + sred[3];
  __syncthreads();
  for (int idx = tid; idx < KU; idx += kUB) {
    float v = 0.f;
    for (int r = 0; r < nrec; r++) v += __expf(-(recs[(size_t)r * stride] - beta) * inv_lambda) * recs[(size_t)r * stride + 2 + idx];
    if (finalize) out[idx] = v / sS; else out[2 + idx] = v;
  }
  if (!finalize && tid == 0) { out[0] = beta; out[1] = sS; }
}
__global__ __launch_bounds__(kUB) void k_mppi_merge(const float* __restrict__ recs, int nrec, int KU, float inv_lambda, int finalize, float* __restrict__ out) {
  __shared__ float sred[4];
  __shared__ float sS;
  mppi_merge_body(recs, nrec, KU, inv_lambda, finalize, out, sred, sS);
}

// ---------------------------------------------------------------- top-k (CEM elites, PS argmax)
struct Cand { float c; int i; };
__device__ __forceinline__ bool better(const Cand& a, const Cand& b, int tie_high) {
  if (a.c != b.c) return a.c < b.c;
  if (a.i < 0 || b.i < 0) return a.i >= 0;
  return tie_high ? a.i > b.i : a.i < b.i;
}
__device__ __forceinline__ Cand wave_best(Cand v, int tie_high) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Cand w{__shfl_xor(v.c, o, 64), __shfl_xor(v.i, o, 64)};
    if (better(w, v, tie_high)) v = w;
  }
  return v;
}

__device__ __forceinline__ void topk_block_body(const float* __restrict__ costs, int N, int n_offset, int k, int tie_high, float* __restrict__ scratch, Cand* sred) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = blockIdx.x * kUB + tid;
  Cand mine{n < N ? costs[n] : INFINITY, n < N ? n_offset + n : -1};
  if (mine.c != mine.c) mine.c = INFINITY;  // NaN costs never win
  for (int e = 0; e < k; e++) {
    Cand b = wave_best(mine, tie_high);
    if (lane == 0) sred[wave] = b;
    __syncthreads();
    Cand best = sred[0];
    for (int w = 1; w < 4; w++) if (better(sred[w], best, tie_high)) best = sred[w];
    __syncthreads();
    if (tid == 0) { scratch[((size_t)blockIdx.x * k + e) * 2] = best.c; scratch[((size_t)blockIdx.x * k + e) * 2 + 1] = __int_as_float(best.i); }
    if (best.i == mine.i) { mine.c = INFINITY; mine.i = -1; }
  }
}
__global__ __launch_bounds__(kUB) void k_topk_block(const float* __restrict__ costs, int N, int n_offset, int k, int tie_high, float* __restrict__ scratch) {
  __shared__ Cand sred[4];
  topk_block_body(costs, N, n_offset, k, tie_high, scratch, sred);
}

// one workgroup: choose k best of ncand (cost, global index) pairs; emit records [cost, index, knots...]
__device__ __forceinline__ void topk_choose(const float* __restrict__ cand, int ncand, int k, int tie_high, Cand* sred, Cand* chosen) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int e = 0; e < k; e++) {
    Cand mine{INFINITY, -1};
    for (int r = tid; r < ncand; r += kUB) {
      Cand c{cand[(size_t)r * 2], __float_as_int(cand[(size_t)r * 2 + 1])};
      bool taken = false;
      for (int q = 0; q < e; q++) taken |= (chosen[q].i == c.i);
      if (!taken && c.i >= 0 && better(c, mine, tie_high)) mine = c;
    }
    Cand b = wave_best(mine, tie_high);
    if (lane == 0) sred[wave] = b;
    __syncthreads();
    if (tid == 0) {
      Cand best = sred[0];
      for (int w = 1; w < 4; w++) if (better(sred[w], best, tie_high)) best = sred[w];
      chosen[e] = best;
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void topk_records(const Cand* chosen, int k, const KnotSrc& src, int n_offset, float* __restrict__ rec) {
  const int tid = threadIdx.x;
  const int stride = 2 + src.KU;
  for (int e = 0; e < k; e++) {
    if (tid == 0) { rec[(size_t)e * stride] = chosen[e].c; rec[(size_t)e * stride + 1] = __int_as_float(chosen[e].i); }
    int nl = chosen[e].i - n_offset;
    for (int idx = tid; idx < src.KU; idx += kUB) rec[(size_t)e * stride + 2 + idx] = chosen[e].i >= 0 ? src.get(nl, idx) : 0.f;
  }
}
__global__ __launch_bounds__(kUB) void k_topk_select(const float* __restrict__ cand, int ncand, int k, int tie_high, KnotSrc src, int n_offset, float* __restrict__ rec) {
  __shared__ Cand sred[4];
  __shared__ Cand chosen[JH_MAX_ELITES];
  topk_choose(cand, ncand, k, tie_high, sred, chosen);
  topk_records(chosen, k, src, n_offset, rec);
}

// one workgroup: G*k records -> k elites -> mean / clipped population std
// (`per_rank`, `rank_stride`: record r sits at (r / per_rank) * rank_stride + (r % per_rank) * (2 + KU) -- the all-gathered per-rank records of the sharded plan step, where a
// rank's elite records are followed by its trace records; per_rank = 0: packed, r * (2 + KU))
__device__ __forceinline__ void elite_merge_body(const float* __restrict__ recs, int nrec, int k, int KU, int tie_high, float smin, float smax, float* __restrict__ nominal_out,
                                                 float* __restrict__ sigma_out, int* chosen, int per_rank = 0, int rank_stride = 0) {
  const int tid = threadIdx.x, stride = 2 + KU;
  auto at = [&](int r) -> size_t { return per_rank > 0 ? (size_t)(r / per_rank) * rank_stride + (size_t)(r % per_rank) * stride : (size_t)r * stride; };
  if (tid == 0) {
    for (int e = 0; e < k; e++) {
      int bi = -1; Cand best{INFINITY, -1};
      for (int r = 0; r < nrec; r++) {
        Cand c{recs[at(r)], __float_as_int(recs[at(r) + 1])};
        bool taken = false;
        for (int q = 0; q < e; q++) taken |= (chosen[q] == r);
        if (!taken && c.i >= 0 && (bi < 0 || better(c, best, tie_high))) { best = c; bi = r; }
      }
      chosen[e] = bi;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < KU; idx += kUB) {
    float mean = 0.f; int cnt = 0;
    for (int e = 0; e < k; e++) if (chosen[e] >= 0) { mean += recs[at(chosen[e]) + 2 + idx]; cnt++; }
    mean /= (float)(cnt > 0 ? cnt : 1);
    float var = 0.f;
    for (int e = 0; e < k; e++) if (chosen[e] >= 0) { float d = recs[at(chosen[e]) + 2 + idx] - mean; var += d * d; }
    var /= (float)(cnt > 0 ? cnt : 1);
    nominal_out[idx] = mean;
    if (sigma_out) sigma_out[idx] = jh_clampf(sqrtf(var), smin, smax);
  }
}
__global__ __launch_bounds__(kUB) void k_elite_merge(const float* __restrict__ recs, int nrec, int k, int KU, int tie_high, float smin, float smax,
                                                    float* __restrict__ nominal_out, float* __restrict__ sigma_out) {
  __shared__ int chosen[JH_MAX_ELITES];
  elite_merge_body(recs, nrec, k, KU, tie_high, smin, smax, nominal_out, sigma_out, chosen);
}

// ---------------------------------------------------------------- the whole update of a one-GPU plan step in ONE launch
// Controller.update_action's tail (judo/controller/controller.py:288-299: update_nominal_knots, then update_traces) used to be seven launches -- block partials, two
// merges, block top-k, select, trace gather, plus two downloads -- around rollout kernels of 50 us (cartpole, cylinder_push): the plan step was bound by the launch
// chain, not by any kernel.  Here every workgroup writes its partial records (MPPI weights / the update's elite candidates / the trace elites' candidates), takes a
// ticket, and the workgroup that draws the last one merges them: nominal (and CEM sigma) plus the trace elites' records [cost, index, trace row] land in ONE
// output block, one download.  Same device functions as the separate kernels above, same order of operations: bit-identical nominal, sigma and records.
struct TailArgs {
  const float* costs; KnotSrc src; int N, n_offset;
  int mode;            // 0: MPPI, 1: elites (CEM, PS)
  float inv_lambda;    // MPPI
  int k, tie_high;     // the update's elites
  int E;               // trace elites (0: none); ties: the higher global index first
  const float* trace; int row, colmajor;  // trace buffer of the fused rollout kernel
  float* scratch;      // ticket counter (4 floats) | nb * (2 + KU) | nb * k * 2 | nb * E * 2 | k * (2 + KU)
  float* nominal_out; float* sigma_out; float* trace_out;  // trace_out: E x (2 + row)
  float* rec_out;      // non-null: the SHARD form -- instead of nominal / sigma the last workgroup writes this rank's record for the all-gather (jh_update_shard):
                       // MPPI [beta, S, V(KU)] or k x [cost, index, knots(KU)], then the E trace records; jh_shard_merge finishes the update on every rank
};
__global__ __launch_bounds__(kUB) void k_update_tail(TailArgs a) {
  __shared__ float sred[4];
  __shared__ float sS;
  __shared__ float sV[4][JH_MAX_KNOT_DIM];
  __shared__ Cand cred[4];
  __shared__ Cand chosen[JH_MAX_ELITES];
  __shared__ int ichosen[JH_MAX_ELITES];
  __shared__ int s_last;
  const int tid = threadIdx.x, nb = gridDim.x, KU = a.src.KU;
  unsigned* counter = reinterpret_cast<unsigned*>(a.scratch);  // (a fixed place: the record layout behind it depends on the arguments; zero before the first launch)
  float* s_mppi = a.scratch + 4;
  float* s_topA = s_mppi + (size_t)nb * (2 + KU);
  float* s_topB = s_topA + (size_t)nb * a.k * 2;
  float* s_rec = s_topB + (size_t)nb * a.E * 2;
  if (a.mode == 0) mppi_block_body(a.costs, a.src, a.N, a.inv_lambda, s_mppi, sred, sV);
  else topk_block_body(a.costs, a.N, a.n_offset, a.k, a.tie_high, s_topA, cred);
  if (a.E > 0) { __syncthreads(); topk_block_body(a.costs, a.N, a.n_offset, a.E, 1, s_topB, cred); }
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = (atomicAdd(counter, 1u) == (unsigned)(nb - 1));
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (tid == 0) *counter = 0u;  // (the next launch on this stream finds it reset)
  if (a.rec_out) {  // shard form: the record itself (what jh_mppi_partial / jh_topk_partial write), merged across the ranks by jh_shard_merge
    if (a.mode == 0) mppi_merge_body(s_mppi, nb, KU, a.inv_lambda, 0, a.rec_out, sred, sS);
    else { topk_choose(s_topA, nb * a.k, a.k, a.tie_high, cred, chosen); topk_records(chosen, a.k, a.src, a.n_offset, a.rec_out); }
  } else if (a.mode == 0) mppi_merge_body(s_mppi, nb, KU, a.inv_lambda, 1, a.nominal_out, sred, sS);
  else {
    topk_choose(s_topA, nb * a.k, a.k, a.tie_high, cred, chosen);
    topk_records(chosen, a.k, a.src, a.n_offset, s_rec);
    __threadfence_block();
    __syncthreads();
    elite_merge_body(s_rec, a.k, a.k, KU, a.tie_high, 0.f, INFINITY, a.nominal_out, a.sigma_out, ichosen);
  }
  if (a.E > 0) {
    __syncthreads();
    topk_choose(s_topB, nb * a.E, a.E, 1, cred, chosen);
    for (int e = 0; e < a.E; e++) {
      const float cost = chosen[e].c; const int gi = chosen[e].i, li = gi - a.n_offset;
      float* o = a.trace_out + (size_t)e * (2 + a.row);
      const bool ok = gi >= 0 && li >= 0 && li < a.N && cost < 3.0e38f;
      if (tid == 0) { o[0] = ok ? cost : __int_as_float(0x7f800000); o[1] = __int_as_float(ok ? gi : -1); }
      for (int i = tid; i < a.row; i += kUB) o[2 + i] = ok ? (a.colmajor ? a.trace[(size_t)i * a.N + li] : a.trace[(size_t)li * a.row + i]) : 0.f;
    }
  }
}

int check_dims(int N, int K, int nu) {
  JH_REQUIRE(N > 0 && K > 0 && nu > 0, "N, K, nu must be positive (N=%d K=%d nu=%d)", N, K, nu);
  JH_REQUIRE(K * nu <= JH_MAX_KNOT_DIM, "K*nu = %d exceeds JH_MAX_KNOT_DIM = %d", K * nu, JH_MAX_KNOT_DIM);
  return JH_OK;
}

}  // namespace

extern "C" size_t jh_update_scratch_floats(int N, int K, int nu) {
  size_t nb = (size_t)(N + kUB - 1) / kUB;
  size_t per = (size_t)(2 + K * nu);
  size_t tk = 2 * (size_t)JH_MAX_ELITES;
  return nb * (per > tk ? per : tk) + 16;
}

// ------------------------------------------------------------------------------------------------ optimizer noise: counter-based normals
// np.random.randn(N - 1, K, nu) of the reference's optimizers (judo/optimizers/mppi.py:52) in the kernels' (K*nu, N) layout.  Element (row, n) of draw number
// `draw` under `seed` is a pure function of those four numbers -- Philox4x32-10 keyed by the seed, counter = (block of 4 global rollout indices, row, draw) --
// so a rank that owns rollouts [n_offset, n_offset + n_local) generates exactly its columns and the plan does not depend on how the rollouts are sharded.
// One thread: one Philox block = 4 consecutive rollouts of one row (two Box-Muller pairs), 16 contiguous bytes.
namespace {
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(kUB) void k_noise_normal(uint32_t seed_lo, uint32_t seed_hi, uint32_t draw, int rows, int n_offset, int n_local, float* __restrict__ out, int ldn) {
  const int blocks_per_row = (n_offset + n_local + 3) / 4 - n_offset / 4;  // Philox blocks that touch this shard's columns
  const size_t t = (size_t)blockIdx.x * kUB + threadIdx.x;
  if (t >= (size_t)rows * blocks_per_row) return;
  const int row = (int)(t / blocks_per_row), blk = n_offset / 4 + (int)(t % blocks_per_row);
  uint32_t x[4];
  philox4x32_10((uint32_t)blk, (uint32_t)row, draw, 0u, seed_lo, seed_hi, x);
  float z[4];
#pragma unroll
  for (int h = 0; h < 2; h++) {  // Box-Muller on (u1, u2) in (0, 1): u = (x + 0.5) * 2^-32
    const float u1 = ((float)(x[2 * h] >> 8) + 0.5f) * 5.9604644775390625e-8f, u2 = ((float)(x[2 * h + 1] >> 8) + 0.5f) * 5.9604644775390625e-8f;  // 24 bits each
    const float rad = sqrtf(-2.f * logf(u1));
    float sn, cs; sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
  }
  const int n0 = 4 * blk - n_offset;  // local index of the block's first rollout (may be negative / run past the shard at the edges)
  float* o = out + (size_t)row * ldn + n0;
  if (n0 >= 0 && n0 + 3 < n_local && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) *reinterpret_cast<float4*>(o) = make_float4(z[0], z[1], z[2], z[3]);
  else
    for (int k = 0; k < 4; k++) if (n0 + k >= 0 && n0 + k < n_local) o[k] = z[k];
}
}  // namespace

extern "C" int jh_noise_normal(unsigned long long seed, unsigned int draw, int rows, int n_offset, int n_local, float* out, int ldn, void* stream) {
  JH_REQUIRE(out != nullptr, "noise_normal: null pointer");
  JH_REQUIRE(rows > 0 && n_local > 0 && n_offset >= 0 && ldn >= n_local, "noise_normal: rows, n_local must be positive, n_offset >= 0, ldn >= n_local (rows=%d n_local=%d n_offset=%d ldn=%d)", rows, n_local, n_offset, ldn);
  const size_t blocks_per_row = (size_t)((n_offset + n_local + 3) / 4 - n_offset / 4), total = (size_t)rows * blocks_per_row;
  hipLaunchKernelGGL(k_noise_normal, dim3((unsigned)((total + kUB - 1) / kUB)), dim3(kUB), 0, (hipStream_t)stream, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)draw, rows, n_offset,
                     n_local, out, ldn);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_sample_knots(const float* nominal, const float* noise, int ldn, const float* sigma, const float* lohi, int N, int n_offset, int K,
                               int nu, float* knots_nku, void* stream) {
  if (int e = check_dims(N, K, nu)) return e;
  JH_REQUIRE(nominal && noise && sigma && knots_nku, "sample_knots: null pointer");
  JH_REQUIRE(ldn >= N, "sample_knots: ldn (%d) < N (%d)", ldn, N);
  KnotSrc src{nullptr, nominal, noise, sigma, lohi, ldn, n_offset, K * nu, nu};
  size_t total = (size_t)N * K * nu;
  hipLaunchKernelGGL(k_sample_knots, dim3((unsigned)((total + kUB - 1) / kUB)), dim3(kUB), 0, (hipStream_t)stream, src, N, knots_nku);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_spline_controls(const float* W, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma,
                                 const float* lohi, int N, int n_offset, int H, int K, int nu, float* controls, void* stream) {
  if (int e = check_dims(N, K, nu)) return e;
  JH_REQUIRE(W && controls, "spline_controls: null pointer");
  JH_REQUIRE(knots_nku || (nominal && noise && sigma), "spline_controls: need either knots_nku or nominal+noise+sigma");
  JH_REQUIRE(knots_nku || ldn >= N, "spline_controls: ldn (%d) < N (%d)", ldn, N);
  JH_REQUIRE(H > 0, "spline_controls: H must be positive");
  size_t lds = sizeof(float) * ((size_t)H * K + 64 * (size_t)((K * nu) | 1));
  JH_REQUIRE(lds <= 64 * 1024, "spline_controls: H*K too large for the LDS staging (%zu bytes)", lds);
  KnotSrc src{knots_nku, nominal, noise, sigma, lohi, ldn, n_offset, K * nu, nu};
  hipLaunchKernelGGL(k_spline_controls, dim3((N + 63) / 64), dim3(64), lds, (hipStream_t)stream, src, W, N, H, K, controls);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_knot_moments(const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* lohi,
                               const float* center, int N, int n_offset, int K, int nu, float* out, void* stream) {
  if (int e = check_dims(N, K, nu)) return e;
  JH_REQUIRE(center && out, "knot_moments: null pointer");
  JH_REQUIRE(knots_nku || (nominal && noise && sigma), "knot_moments: need either knots_nku or nominal+noise+sigma");
  JH_REQUIRE(knots_nku || ldn >= N, "knot_moments: ldn (%d) < N (%d)", ldn, N);
  KnotSrc src{knots_nku, nominal, noise, sigma, lohi, ldn, n_offset, K * nu, nu};
  hipStream_t st = (hipStream_t)stream;
  JH_HIP(hipMemsetAsync(out, 0, sizeof(float) * 2 * nu, st));
  hipLaunchKernelGGL(k_knot_moments, dim3((N + 63) / 64), dim3(64), 0, st, src, center, N, K, out);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_mppi_partial(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma,
                               const float* lohi, int N, int n_offset, int K, int nu, float lambda, float* scratch, float* rec, void* stream) {
  if (int e = check_dims(N, K, nu)) return e;
  JH_REQUIRE(costs && scratch && rec, "mppi_partial: null pointer");
  JH_REQUIRE(knots_nku || (nominal && noise && sigma), "mppi_partial: need either knots_nku or nominal+noise+sigma");
  JH_REQUIRE(knots_nku || ldn >= N, "mppi_partial: ldn (%d) < N (%d)", ldn, N);
  JH_REQUIRE(lambda > 0.f, "mppi_partial: temperature must be positive");
  KnotSrc src{knots_nku, nominal, noise, sigma, lohi, ldn, n_offset, K * nu, nu};
  int nb = (N + kUB - 1) / kUB;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mppi_block, dim3(nb), dim3(kUB), 0, st, costs, src, N, 1.f / lambda, scratch);
  hipLaunchKernelGGL(k_mppi_merge, dim3(1), dim3(kUB), 0, st, scratch, nb, K * nu, 1.f / lambda, 0, rec);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_mppi_merge(const float* recs, int G, int K, int nu, float lambda, float* nominal_out, void* stream) {
  if (int e = check_dims(1, K, nu)) return e;
  JH_REQUIRE(recs && nominal_out && G > 0, "mppi_merge: bad arguments");
  hipLaunchKernelGGL(k_mppi_merge, dim3(1), dim3(kUB), 0, (hipStream_t)stream, recs, G, K * nu, 1.f / lambda, 1, nominal_out);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_topk_partial(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma,
                               const float* lohi, int N, int n_offset, int K, int nu, int k, int tie_high, float* scratch, float* rec, void* stream) {
  if (int e = check_dims(N, K, nu)) return e;
  JH_REQUIRE(k >= 1 && k <= JH_MAX_ELITES, "topk_partial: k = %d outside [1, %d]", k, JH_MAX_ELITES);
  JH_REQUIRE(costs && scratch && rec, "topk_partial: null pointer");
  JH_REQUIRE(knots_nku || (nominal && noise && sigma), "topk_partial: need either knots_nku or nominal+noise+sigma");
  KnotSrc src{knots_nku, nominal, noise, sigma, lohi, ldn, n_offset, K * nu, nu};
  int nb = (N + kUB - 1) / kUB;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_topk_block, dim3(nb), dim3(kUB), 0, st, costs, N, n_offset, k, tie_high, scratch);
  hipLaunchKernelGGL(k_topk_select, dim3(1), dim3(kUB), 0, st, scratch, nb * k, k, tie_high, src, n_offset, rec);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

// rows of the trace buffer for the elites of a jh_topk_partial record set: rec_out[e] = [cost, global index (bits), trace row]
__global__ void k_trace_gather(const float* __restrict__ rec_in, int k, int stride_in, int n_offset, int n_local, const float* __restrict__ trace, int row, int colmajor,
                               float* __restrict__ rec_out) {
  const int e = blockIdx.x;
  const float cost = rec_in[(size_t)e * stride_in]; const int gi = __float_as_int(rec_in[(size_t)e * stride_in + 1]);
  float* o = rec_out + (size_t)e * (2 + row);
  const int li = gi - n_offset;
  const bool ok = gi >= 0 && li >= 0 && li < n_local && cost < 3.0e38f;
  if (threadIdx.x == 0) { o[0] = ok ? cost : __int_as_float(0x7f800000); o[1] = __int_as_float(ok ? gi : -1); }
  for (int i = threadIdx.x; i < row; i += blockDim.x) o[2 + i] = ok ? (colmajor ? trace[(size_t)i * n_local + li] : trace[(size_t)li * row + i]) : 0.f;
}

extern "C" int jh_trace_gather(const float* rec_in, int k, int stride_in, int n_offset, int n_local, const float* trace, int row_floats, int colmajor, float* rec_out,
                               void* stream) {
  JH_REQUIRE(rec_in && trace && rec_out && k >= 1 && k <= JH_MAX_ELITES && stride_in >= 2 && row_floats >= 1 && n_local >= 1, "trace_gather: bad arguments");
  hipLaunchKernelGGL(k_trace_gather, dim3(k), dim3(256), 0, (hipStream_t)stream, rec_in, k, stride_in, n_offset, n_local, trace, row_floats, colmajor, rec_out);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" size_t jh_update_fused_scratch_floats(int N, int K, int nu) {
  const size_t nb = (size_t)(N + kUB - 1) / kUB, KU = (size_t)K * nu;
  return nb * (2 + KU) + nb * 4 * JH_MAX_ELITES + (size_t)JH_MAX_ELITES * (2 + KU) + 16;
}

extern "C" int jh_update_fused(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* lohi, int N,
                               int n_offset, int K, int nu, int mode, float lambda, int k, int tie_high, int E, const float* trace, int row_floats, int colmajor,
                               float* scratch, float* nominal_out, float* sigma_out, float* trace_out, void* stream) {
  if (int e = check_dims(N, K, nu)) return e;
  JH_REQUIRE(costs && scratch && nominal_out, "update_fused: null pointer");
  JH_REQUIRE(knots_nku || (nominal && noise && sigma), "update_fused: need either knots_nku or nominal+noise+sigma");
  JH_REQUIRE(knots_nku || ldn >= N, "update_fused: ldn (%d) < N (%d)", ldn, N);
  JH_REQUIRE(mode == 0 || mode == 1, "update_fused: mode must be 0 (MPPI) or 1 (elites)");
  JH_REQUIRE(mode != 0 || lambda > 0.f, "update_fused: temperature must be positive");
  JH_REQUIRE(mode != 1 || (k >= 1 && k <= JH_MAX_ELITES), "update_fused: k = %d outside [1, %d]", k, JH_MAX_ELITES);
  JH_REQUIRE(E >= 0 && E <= JH_MAX_ELITES && (E == 0 || (trace && trace_out && row_floats >= 1)), "update_fused: bad trace arguments (E=%d)", E);
  TailArgs a;
  a.costs = costs; a.src = KnotSrc{knots_nku, nominal, noise, sigma, lohi, ldn, n_offset, K * nu, nu}; a.N = N; a.n_offset = n_offset;
  a.mode = mode; a.inv_lambda = mode == 0 ? 1.f / lambda : 0.f; a.k = mode == 1 ? k : 0; a.tie_high = tie_high; a.E = E;
  a.trace = trace; a.row = row_floats; a.colmajor = colmajor; a.scratch = scratch; a.nominal_out = nominal_out; a.sigma_out = sigma_out; a.trace_out = trace_out; a.rec_out = nullptr;
  const int nb = (N + kUB - 1) / kUB;
  hipLaunchKernelGGL(k_update_tail, dim3(nb), dim3(kUB), 0, (hipStream_t)stream, a);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

// ------------------------------------------------------------------------------------------------ the sharded plan step: launch -> all-gather -> merge
// A rank's record (jh_update_shard, written by the last workgroup of k_update_tail): [update record | E trace records].  After ONE all-gather every rank runs
// k_shard_merge on the G records: the update (log-sum-exp merge / global elites) and the E best of the G * E trace records, into the same output block the one-GPU
// tail writes -- nominal | sigma | E x [cost, index, trace row] -- so that what follows on the host is the same code for one rank and for eight.
namespace {
__global__ __launch_bounds__(kUB) void k_shard_merge(const float* __restrict__ recs, int G, int L, int mode, int KU, float inv_lambda, int k, int tie_high, int E, int row,
                                                     float* __restrict__ nominal_out, float* __restrict__ sigma_out, float* __restrict__ trace_out) {
  __shared__ float sred[4];
  __shared__ float sS;
  __shared__ int ichosen[JH_MAX_ELITES];
  __shared__ Cand cred[4];
  __shared__ Cand chosen[JH_MAX_ELITES];
  __shared__ float spair[2 * 64 * JH_MAX_ELITES];  // (cost, index) of the G * E trace candidates (G <= 64)
  const int tid = threadIdx.x;
  const int urec = mode == 0 ? 2 + KU : k * (2 + KU);
  if (mode == 0) mppi_merge_body(recs, G, KU, inv_lambda, 1, nominal_out, sred, sS, L);
  else elite_merge_body(recs, G * k, k, KU, tie_high, 0.f, INFINITY, nominal_out, sigma_out, ichosen, k, L);
  if (E <= 0) return;
  __syncthreads();
  for (int r = tid; r < G * E; r += kUB) {
    const float* t = recs + (size_t)(r / E) * L + urec + (size_t)(r % E) * (2 + row);
    spair[2 * r] = t[0]; spair[2 * r + 1] = t[1];
  }
  __syncthreads();
  topk_choose(spair, G * E, E, 1, cred, chosen);
  for (int e = 0; e < E; e++) {
    float* o = trace_out + (size_t)e * (2 + row);
    const int gi = chosen[e].i;
    int src = -1;
    for (int r = 0; r < G * E; r++) if (gi >= 0 && __float_as_int(spair[2 * r + 1]) == gi) { src = r; break; }  // (global rollout indices are unique)
    const bool ok = src >= 0 && chosen[e].c < 3.0e38f;
    const float* t = ok ? recs + (size_t)(src / E) * L + urec + (size_t)(src % E) * (2 + row) : nullptr;
    if (tid == 0) { o[0] = ok ? chosen[e].c : __int_as_float(0x7f800000); o[1] = __int_as_float(ok ? gi : -1); }
    for (int i = tid; i < row; i += kUB) o[2 + i] = ok ? t[2 + i] : 0.f;
  }
}
}  // namespace

extern "C" size_t jh_shard_record_floats(int K, int nu, int mode, int k, int E, int row_floats) {
  const size_t KU = (size_t)K * nu;
  return (mode == 0 ? 2 + KU : (size_t)k * (2 + KU)) + (size_t)E * (2 + (size_t)row_floats);
}

extern "C" int jh_update_shard(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* lohi, int N,
                               int n_offset, int K, int nu, int mode, float lambda, int k, int tie_high, int E, const float* trace, int row_floats, int colmajor,
                               float* scratch, float* rec_out, void* stream) {
  if (int e = check_dims(N, K, nu)) return e;
  JH_REQUIRE(costs && scratch && rec_out, "update_shard: null pointer");
  JH_REQUIRE(knots_nku || (nominal && noise && sigma), "update_shard: need either knots_nku or nominal+noise+sigma");
  JH_REQUIRE(knots_nku || ldn >= N, "update_shard: ldn (%d) < N (%d)", ldn, N);
  JH_REQUIRE(mode == 0 || mode == 1, "update_shard: mode must be 0 (MPPI) or 1 (elites)");
  JH_REQUIRE(mode != 0 || lambda > 0.f, "update_shard: temperature must be positive");
  JH_REQUIRE(mode != 1 || (k >= 1 && k <= JH_MAX_ELITES), "update_shard: k = %d outside [1, %d]", k, JH_MAX_ELITES);
  JH_REQUIRE(E >= 0 && E <= JH_MAX_ELITES && (E == 0 || (trace && row_floats >= 1)), "update_shard: bad trace arguments (E=%d)", E);
  TailArgs a;
  a.costs = costs; a.src = KnotSrc{knots_nku, nominal, noise, sigma, lohi, ldn, n_offset, K * nu, nu}; a.N = N; a.n_offset = n_offset;
  a.mode = mode; a.inv_lambda = mode == 0 ? 1.f / lambda : 0.f; a.k = mode == 1 ? k : 0; a.tie_high = tie_high; a.E = E;
  a.trace = trace; a.row = row_floats; a.colmajor = colmajor; a.scratch = scratch; a.nominal_out = nullptr; a.sigma_out = nullptr;
  a.rec_out = rec_out; a.trace_out = rec_out + (mode == 0 ? 2 + K * nu : k * (2 + K * nu));
  const int nb = (N + kUB - 1) / kUB;
  hipLaunchKernelGGL(k_update_tail, dim3(nb), dim3(kUB), 0, (hipStream_t)stream, a);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_shard_merge(const float* recs, int G, int K, int nu, int mode, float lambda, int k, int tie_high, int E, int row_floats, float* nominal_out,
                              float* sigma_out, float* trace_out, void* stream) {
  if (int e = check_dims(1, K, nu)) return e;
  JH_REQUIRE(recs && nominal_out && G >= 1 && G <= 64, "shard_merge: bad arguments (G=%d)", G);
  JH_REQUIRE(mode == 0 || mode == 1, "shard_merge: mode must be 0 (MPPI) or 1 (elites)");
  JH_REQUIRE(mode != 0 || lambda > 0.f, "shard_merge: temperature must be positive");
  JH_REQUIRE(mode != 1 || (k >= 1 && k <= JH_MAX_ELITES), "shard_merge: k = %d outside [1, %d]", k, JH_MAX_ELITES);
  JH_REQUIRE(E >= 0 && E <= JH_MAX_ELITES && (E == 0 || (trace_out && row_floats >= 1)), "shard_merge: bad trace arguments (E=%d)", E);
  const int L = (int)jh_shard_record_floats(K, nu, mode, k, E, row_floats);
  hipLaunchKernelGGL(k_shard_merge, dim3(1), dim3(kUB), 0, (hipStream_t)stream, recs, G, L, mode, K * nu, mode == 0 ? 1.f / lambda : 0.f, mode == 1 ? k : 0, tie_high, E, row_floats,
                     nominal_out, sigma_out, trace_out);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_elite_merge(const float* recs, int G, int k, int K, int nu, int tie_high, float smin, float smax, float* nominal_out, float* sigma_out,
                              void* stream) {
  if (int e = check_dims(1, K, nu)) return e;
  JH_REQUIRE(recs && nominal_out && G > 0 && k >= 1 && k <= JH_MAX_ELITES, "elite_merge: bad arguments");
  hipLaunchKernelGGL(k_elite_merge, dim3(1), dim3(kUB), 0, (hipStream_t)stream, recs, G * k, k, K * nu, tie_high, smin, smax, nominal_out, sigma_out);
  JH_HIP(hipGetLastError());
  return JH_OK;
}
