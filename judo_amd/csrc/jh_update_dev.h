// jh_update_dev.h -- device functions of the update kernels (jh_update.hip) in a header, so that a rollout kernel can run the update's tail in its own launch
// (jh_simple.hip: the closed-form models' plan step as ONE launch).  Reference: judo/optimizers/mppi.py:76-82, cem.py:88-92, ps.py:64-65, judo/controller/controller.py:288-299.
#pragma once
#include "jh_internal.h"

namespace jh_upd {

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
// ---------------------------------------------------------------- top-k (CEM elites, PS argmax)
struct Cand { float c; int i; };
__device__ __forceinline__ bool better(const Cand& a, const Cand& b, int tie_high) {
  if (a.c != b.c) return a.c < b.c;
  if (a.i < 0 || b.i < 0) return a.i >= 0;
  return tie_high ? a.i > b.i : a.i < b.i;
}
// Best candidate of the wave, in every lane.  Round 6: the six exchange levels stay in the VALU -- four DPP row modifiers (quad xor 1, quad xor 2, half mirror, mirror), then
// gfx950's v_permlane16_swap / v_permlane32_swap (with both operands the same register they leave (row0, row0, row2, row2) | (row1, row1, row3, row3), resp. the two halves) --
// instead of six `__shfl_xor` = ds_bpermute trips through the LDS crossbar of ~100 cycles each: the arg-min rounds of the update's tail run on waves that have their SIMD
// to themselves.  `better` is a strict total order on distinct candidates, so every lane ends with the same winner whatever the pairing.
template <int CTRL>
__device__ __forceinline__ Cand dpp_cand(const Cand& v) {
  return Cand{__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.c), CTRL, 0xF, 0xF, true)), __builtin_amdgcn_update_dpp(0, v.i, CTRL, 0xF, 0xF, true)};
}
__device__ __forceinline__ Cand wave_best(Cand v, int tie_high) {
  { const Cand w = dpp_cand<0xB1>(v); if (better(w, v, tie_high)) v = w; }
  { const Cand w = dpp_cand<0x4E>(v); if (better(w, v, tie_high)) v = w; }
  { const Cand w = dpp_cand<0x141>(v); if (better(w, v, tie_high)) v = w; }
  { const Cand w = dpp_cand<0x140>(v); if (better(w, v, tie_high)) v = w; }
  {
    const auto rc = __builtin_amdgcn_permlane16_swap(__float_as_uint(v.c), __float_as_uint(v.c), false, false);
    const auto ri = __builtin_amdgcn_permlane16_swap((unsigned)v.i, (unsigned)v.i, false, false);
    const Cand a{__uint_as_float(rc[0]), (int)ri[0]}, b{__uint_as_float(rc[1]), (int)ri[1]};
    v = better(b, a, tie_high) ? b : a;
  }
  {
    const auto rc = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.c), __float_as_uint(v.c), false, false);
    const auto ri = __builtin_amdgcn_permlane32_swap((unsigned)v.i, (unsigned)v.i, false, false);
    const Cand a{__uint_as_float(rc[0]), (int)ri[0]}, b{__uint_as_float(rc[1]), (int)ri[1]};
    v = better(b, a, tie_high) ? b : a;
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
// one workgroup: choose k best of ncand (cost, global index) pairs; emit records [cost, index, knots...]
__device__ __forceinline__ void topk_choose(const float* __restrict__ cand, int ncand, int k, int tie_high, Cand* sred, Cand* chosen) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int OWN = 4;  // candidates a thread keeps in registers: up to OWN * kUB = 1 024 of them are read from memory ONCE (round 6) instead of once per round -- the
                          // candidates were written by other workgroups, on other XCDs: every read is a trip to memory, and the rounds are a dependent chain
  if (ncand <= OWN * kUB) {
    Cand own[OWN];
#pragma unroll
    for (int j = 0; j < OWN; j++) {
      const int r = tid + j * kUB;
      own[j] = r < ncand ? Cand{cand[(size_t)r * 2], __float_as_int(cand[(size_t)r * 2 + 1])} : Cand{INFINITY, -1};
    }
    for (int e = 0; e < k; e++) {
      Cand mine{INFINITY, -1};
#pragma unroll
      for (int j = 0; j < OWN; j++) if (own[j].i >= 0 && better(own[j], mine, tie_high)) mine = own[j];
      Cand b = wave_best(mine, tie_high);
      if (lane == 0) sred[wave] = b;
      __syncthreads();
      Cand best = sred[0];
      for (int w = 1; w < 4; w++) if (better(sred[w], best, tie_high)) best = sred[w];
      __syncthreads();
      if (tid == 0) chosen[e] = best;
#pragma unroll
      for (int j = 0; j < OWN; j++) if (best.i >= 0 && own[j].i == best.i) own[j].i = -1;  // (global rollout indices are unique)
    }
    __syncthreads();  // `chosen` is read by every thread behind this call
    return;
  }
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
  unsigned* done_flag; unsigned done_value;  // non-null: a word in device-visible pinned host memory that receives done_value once everything above is written (the host polls it: jh_download_end)
  float* rec_out;      // non-null: the SHARD form -- instead of nominal / sigma the last workgroup writes this rank's record for the all-gather (jh_update_shard):
                       // MPPI [beta, S, V(KU)] or k x [cost, index, knots(KU)], then the E trace records; jh_shard_merge finishes the update on every rank
};
// (the body of k_update_tail: also the tail of the closed-form rollout kernels' one-launch plan step, jh_simple.hip -- a workgroup of kUB threads whose thread t holds local rollout blockIdx.x * kUB + t)
#ifdef JH_TAIL_TICKS  // diagnostic builds (tools/diag/tail_ticks.py): 100 MHz wall-clock stamps of the last workgroup's way through the tail
__device__ long long g_tail_ticks[16];
#define TAIL_TICK(i) do { if (threadIdx.x == 0) g_tail_ticks[i] = wall_clock64(); } while (0)
#else
#define TAIL_TICK(i)
#endif
__device__ __forceinline__ void update_tail_body(const TailArgs& a) {
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
#ifdef JH_TAIL_TICKS
  const long long t_in = wall_clock64();
#endif
  if (a.mode == 0) mppi_block_body(a.costs, a.src, a.N, a.inv_lambda, s_mppi, sred, sV);
  else topk_block_body(a.costs, a.N, a.n_offset, a.k, a.tie_high, s_topA, cred);
#ifdef JH_TAIL_TICKS
  const long long t_b1 = wall_clock64();
#endif
  if (a.E > 0) { __syncthreads(); topk_block_body(a.costs, a.N, a.n_offset, a.E, 1, s_topB, cred); }
#ifdef JH_TAIL_TICKS
  const long long t_b2 = wall_clock64();
#endif
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = (atomicAdd(counter, 1u) == (unsigned)(nb - 1));
  __syncthreads();
  if (!s_last) return;
#ifdef JH_TAIL_TICKS
  if (tid == 0) { g_tail_ticks[0] = t_in; g_tail_ticks[1] = t_b1; g_tail_ticks[2] = t_b2; }
#endif
  TAIL_TICK(3);
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
  TAIL_TICK(4);
  if (a.E > 0) {
    __syncthreads();
    topk_choose(s_topB, nb * a.E, a.E, 1, cred, chosen);
    TAIL_TICK(5);
    for (int e = 0; e < a.E; e++) {
      const float cost = chosen[e].c; const int gi = chosen[e].i, li = gi - a.n_offset;
      float* o = a.trace_out + (size_t)e * (2 + a.row);
      const bool ok = gi >= 0 && li >= 0 && li < a.N && cost < 3.0e38f;
      if (tid == 0) { o[0] = ok ? cost : __int_as_float(0x7f800000); o[1] = __int_as_float(ok ? gi : -1); }
      for (int i = tid; i < a.row; i += kUB) o[2 + i] = ok ? (a.colmajor ? a.trace[(size_t)i * a.N + li] : a.trace[(size_t)li * a.row + i]) : 0.f;
    }
  }
  TAIL_TICK(6);
  if (a.done_flag) {  // completion flag for the polling host: every thread's stores to the (host) output block first, system scope
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.done_flag, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  TAIL_TICK(7);
}

}  // namespace jh_upd

// jh_update.hip: the argument checks of jh_update_fused / jh_update_shard and the launch record of the tail (rec_out non-null: the shard form)
int jh_update_tail_args(const char* who, const float* costs, const float* knots_nku, const float* nominal, const float* noise, int ldn, const float* sigma, const float* lohi, int N,
                        int n_offset, int K, int nu, int mode, float lambda, int k, int tie_high, int E, const float* trace, int row_floats, int colmajor, float* scratch,
                        float* nominal_out, float* sigma_out, float* trace_out, float* rec_out, jh_upd::TailArgs* a);
// jh_simple.hip: rollout + cost + update tail of a closed-form model in one launch
int jh_update_tail_launch(const jh_upd::TailArgs& a, hipStream_t st);  // k_update_tail on its own (the articulated models' plan step)
bool jh_simple_plan_step_fits(const jh_model* m, int H, int K);
int jh_simple_plan_step(const jh_model* m, const float* x0, const float* W, const float* tp, int H, int K, const jh_upd::TailArgs& a, hipStream_t st);
