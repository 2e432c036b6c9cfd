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
#include "jh_update_dev.h"

namespace {

using namespace jh_upd;

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

__global__ __launch_bounds__(kUB) void k_mppi_block(const float* __restrict__ costs, KnotSrc src, int N, float inv_lambda, float* __restrict__ scratch) {
  __shared__ float sred[4];
  __shared__ float sV[4][JH_MAX_KNOT_DIM];
  mppi_block_body(costs, src, N, inv_lambda, scratch, sred, sV);
}

__global__ __launch_bounds__(kUB) void k_mppi_merge(const float* __restrict__ recs, int nrec, int KU, float inv_lambda, int finalize, float* __restrict__ out) {
  __shared__ float sred[4];
  __shared__ float sS;
  mppi_merge_body(recs, nrec, KU, inv_lambda, finalize, out, sred, sS);
}

__global__ __launch_bounds__(kUB) void k_topk_block(const float* __restrict__ costs, int N, int n_offset, int k, int tie_high, float* __restrict__ scratch) {
  __shared__ Cand sred[4];
  topk_block_body(costs, N, n_offset, k, tie_high, scratch, sred);
}

__global__ __launch_bounds__(kUB) void k_topk_select(const float* __restrict__ cand, int ncand, int k, int tie_high, KnotSrc src, int n_offset, float* __restrict__ rec) {
  __shared__ Cand sred[4];
  __shared__ Cand chosen[JH_MAX_ELITES];
  topk_choose(cand, ncand, k, tie_high, sred, chosen);
  topk_records(chosen, k, src, n_offset, rec);
}

__global__ __launch_bounds__(kUB) void k_elite_merge(const float* __restrict__ recs, int nrec, int k, int KU, int tie_high, float smin, float smax,
                                                    float* __restrict__ nominal_out, float* __restrict__ sigma_out) {
  __shared__ int chosen[JH_MAX_ELITES];
  elite_merge_body(recs, nrec, k, KU, tie_high, smin, smax, nominal_out, sigma_out, chosen);
}

__global__ __launch_bounds__(kUB) void k_update_tail(TailArgs a) { update_tail_body(a); }

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

int jh_update_tail_args(const char* who, const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* lohi, int N,
                        int n_offset, int K, int nu, int mode, float lambda, int k, int tie_high, int E, const float* trace, int row_floats, int colmajor, float* scratch,
                        float* nominal_out, float* sigma_out, float* trace_out, float* rec_out, TailArgs* ap) {
  if (int e = check_dims(N, K, nu)) return e;
  JH_REQUIRE(costs && scratch && (nominal_out || rec_out), "%s: null pointer", who);
  JH_REQUIRE(knots_nku || (nominal && noise && sigma), "%s: need either knots_nku or nominal+noise+sigma", who);
  JH_REQUIRE(knots_nku || ldn >= N, "%s: ldn (%d) < N (%d)", who, ldn, N);
  JH_REQUIRE(mode == 0 || mode == 1, "%s: mode must be 0 (MPPI) or 1 (elites)", who);
  JH_REQUIRE(mode != 0 || lambda > 0.f, "%s: temperature must be positive", who);
  JH_REQUIRE(mode != 1 || (k >= 1 && k <= JH_MAX_ELITES), "%s: k = %d outside [1, %d]", who, k, JH_MAX_ELITES);
  JH_REQUIRE(E >= 0 && E <= JH_MAX_ELITES && (E == 0 || (trace && (trace_out || rec_out) && row_floats >= 1)), "%s: bad trace arguments (E=%d)", who, E);
  TailArgs& a = *ap;
  a.costs = costs; a.src = KnotSrc{knots_nku, nominal, noise, sigma, lohi, ldn, n_offset, K * nu, nu}; a.N = N; a.n_offset = n_offset;
  a.mode = mode; a.inv_lambda = mode == 0 ? 1.f / lambda : 0.f; a.k = mode == 1 ? k : 0; a.tie_high = tie_high; a.E = E;
  a.trace = trace; a.row = row_floats; a.colmajor = colmajor; a.scratch = scratch; a.done_flag = nullptr; a.done_value = 0u;
  if (rec_out) {  // the shard form: this rank's record [update record | E trace records]
    a.nominal_out = nullptr; a.sigma_out = nullptr; a.rec_out = rec_out; a.trace_out = rec_out + (mode == 0 ? 2 + K * nu : k * (2 + K * nu));
  } else { a.nominal_out = nominal_out; a.sigma_out = sigma_out; a.trace_out = trace_out; a.rec_out = nullptr; }
  return JH_OK;
}

int jh_update_tail_launch(const TailArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_update_tail, dim3((a.N + kUB - 1) / kUB), dim3(kUB), 0, st, a);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

extern "C" int jh_update_fused(const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* lohi, int N,
                               int n_offset, int K, int nu, int mode, float lambda, int k, int tie_high, int E, const float* trace, int row_floats, int colmajor,
                               float* scratch, float* nominal_out, float* sigma_out, float* trace_out, void* stream) {
  JH_REQUIRE(nominal_out, "update_fused: null pointer");
  TailArgs a;
  if (int e = jh_update_tail_args("update_fused", costs, knots_nku, nominal, noise, ldn, sigma, lohi, N, n_offset, K, nu, mode, lambda, k, tie_high, E, trace, row_floats, colmajor, scratch,
                                  nominal_out, sigma_out, trace_out, nullptr, &a)) return e;
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
  JH_REQUIRE(rec_out, "update_shard: null pointer");
  TailArgs a;
  if (int e = jh_update_tail_args("update_shard", costs, knots_nku, nominal, noise, ldn, sigma, lohi, N, n_offset, K, nu, mode, lambda, k, tie_high, E, trace, row_floats, colmajor, scratch,
                                  nullptr, nullptr, nullptr, rec_out, &a)) return e;
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
