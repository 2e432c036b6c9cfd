// jh_simple.hip -- closed-form rollout kernels for the two low-DoF tasks (gfx950).
//
//   cartpole       judo/models/xml/cartpole.xml:1-45       2 DoF (slide x, hinge y), no contact, cart joint limit,
//                                                          position servo kp=100 with +-10 N force clamp
//   cylinder_push  judo/models/xml/cylinder_push.xml:1-45  4 slide DoF, one circle-circle contact (pusher <-> cart),
//                                                          frictionless (mu clamped to 1e-5), pyramidal cone
//
// One lane owns one rollout (wave64, one wave per workgroup so that small N still spreads over the CUs).  The H x K
// spline matrix, nominal knots, per-knot sigma, control bounds, x0, the model constants and the cost weights are
// staged in LDS once per workgroup; each lane's clipped knots live in LDS (lane-fastest, bank-conflict free);
// noise is read coalesced from the (K,nu,N) layout exactly once; the running cost is accumulated inside the
// integration loop and only one float per rollout is written back.
//
// Physics = MuJoCo's Euler step with implicit joint damping, restated in closed form for these two models
// (derivation in DESIGN.md section 4): semi-implicit update  v += h*a ; q += h*v  with
// a = (M + h*diag(damping))^-1 (qfrc_smooth + qfrc_constraint); soft constraints by solref/solimp.
#include <cstdint>
#include "jh_internal.h"
#include "jh_update_dev.h"

namespace {

constexpr int kBlock = 64;

struct SplineCtx {
  const float* W;      // LDS  H*K
  const float* knots;  // LDS  (K*nu) x workgroup size, lane fastest
  int K;
};

template <int NU, int BS = kBlock>
__device__ __forceinline__ void spline_controls(const SplineCtx& s, int h, int lane, float* u) {
#pragma unroll
  for (int j = 0; j < NU; j++) u[j] = 0.f;
  for (int k = 0; k < s.K; k++) {
    float w = s.W[h * s.K + k];
#pragma unroll
    for (int j = 0; j < NU; j++) u[j] = fmaf(w, s.knots[(k * NU + j) * BS + lane], u[j]);
  }
}

__device__ __forceinline__ float impedance_pow(const float* si, float dist) {
  float s0 = si[0], s1 = si[1], s2 = si[2], s3 = si[3], s4 = si[4];
  if (s0 == s1 || s2 <= 1e-15f) return 0.5f * (s0 + s1);
  float x = fabsf(dist / s2);
  if (x >= 1.f) return s1;
  if (x <= 0.f) return s0;
  float y;
  if (s4 == 1.f) y = x;
  else if (s4 == 2.f) y = (x <= s3) ? x * x / s3 : 1.f - (1.f - x) * (1.f - x) / (1.f - s3);
  else if (x <= s3) y = powf(x, s4) / powf(s3, s4 - 1.f);
  else y = 1.f - powf(1.f - x, s4) / powf(1.f - s3, s4 - 1.f);
  return s0 + y * (s1 - s0);
}

// ------------------------------------------------------------------------------------------------ cartpole
struct Cartpole {
  static constexpr int NX = 4, NU = 1, NS = 6, NP = CP_NPARAM, NTP = 6;
  float x, th, xd, thd;
  float sn, cs;  // sin / cos of the pole angle, carried: one sincos per step serves the step's dynamics, the cost after it and the sensors
  __device__ void load(const float* x0) { x = x0[0]; th = x0[1]; xd = x0[2]; thd = x0[3]; sincosf(th, &sn, &cs); }
  __device__ void store(float* o) const { o[0] = x; o[1] = th; o[2] = xd; o[3] = thd; }
  // framepos of the sites trace_cart (cart origin) and trace_pole (pole tip), cartpole.xml:27,31,41-42
  __device__ void sensors(const float* P, float* s) const {
    s[0] = x; s[1] = 0.f; s[2] = 0.f; s[3] = x + P[CP_TIP] * sn; s[4] = 0.f; s[5] = P[CP_TIP] * cs;
  }
  __device__ void step(const float* P, const float* u) {
    const float h = P[CP_DT], mp = P[CP_MPOLE], l = P[CP_L];
    // joint-space inertia and its inverse
    float m11 = P[CP_MCART] + mp, m12 = mp * l * cs, m22 = P[CP_IPOLE] + mp * l * l;
    // bias = Coriolis/centrifugal + gravity; passive = -damping*v; actuator = clamp(kp*(clamp(u) - x) - kv*xd)
    float c = u[0];
    if (P[CP_CTRL_LIMITED] != 0.f) c = jh_clampf(c, P[CP_CTRL_LO], P[CP_CTRL_HI]);
    float fa = P[CP_KP] * (c - x) - P[CP_KV] * xd;
    if (P[CP_FRC_LIMITED] != 0.f) fa = jh_clampf(fa, P[CP_FRC_LO], P[CP_FRC_HI]);
    float f1 = -P[CP_DAMP_X] * xd + mp * l * sn * thd * thd + fa;
    float f2 = -P[CP_DAMP_TH] * thd + mp * P[CP_G] * l * sn;
    // cart joint limit: one-sided soft constraint, solved in closed form (a single row)
    float fc = 0.f;
    if (P[CP_X_LIMITED] != 0.f) {
      float dlo = x - P[CP_X_LO], dhi = P[CP_X_HI] - x;
      float dist = fminf(dlo, dhi);
      if (dist < 0.f) {
        float J = dlo < dhi ? 1.f : -1.f;
        float det = m11 * m22 - m12 * m12;
        float a0x = (m22 * f1 - m12 * f2) / det;  // unconstrained cart acceleration
        float A = m22 / det;                      // J Minv J'
        float imp = impedance_pow(P + CP_SOLIMP0, dist);
        float R = fmaxf(1e-15f, (1.f - imp) / imp * P[CP_INVW_X]);
        float aref = -P[CP_LIM_B] * (J * xd) - P[CP_LIM_K] * imp * dist;
        float lam = fmaxf(0.f, (aref - J * a0x) / (A + R));
        fc = J * lam;
      }
    }
    // Euler with implicit damping: (M + h*D) a = qfrc_smooth + qfrc_constraint
    float a11 = m11 + h * P[CP_DAMP_X], a22 = m22 + h * P[CP_DAMP_TH];
    float r1 = f1 + fc, r2 = f2;
    float idet = __builtin_amdgcn_rcpf(a11 * a22 - m12 * m12);  // (v_rcp_f32, 1 ulp: the step is a chain of dependent instructions, a full division is ten of them)
    float ax = (a22 * r1 - m12 * r2) * idet, ath = (a11 * r2 - m12 * r1) * idet;
    xd = fmaf(h, ax, xd); thd = fmaf(h, ath, thd);
    x = fmaf(h, xd, x); th = fmaf(h, thd, th);
    sincosf(th, &sn, &cs);
  }
  // running cost of judo/tasks/cartpole.py:61-78 (w = w_vertical,w_centered,w_velocity,w_control,p_vertical,p_centered)
  __device__ float cost(const float* w, const float* u) const {
    float cv = cs - 1.f;
    return w[0] * (__builtin_amdgcn_sqrtf(cv * cv + w[4] * w[4]) - w[4]) + w[1] * (__builtin_amdgcn_sqrtf(x * x + w[5] * w[5]) - w[5]) +  // (v_sqrt_f32, 1 ulp: a correctly rounded root is a dozen dependent instructions)
           w[2] * 0.5f * (xd * xd + thd * thd) + w[3] * 0.5f * u[0] * u[0];
  }
  __device__ static float finish(float acc, int /*H*/) { return acc; }
};

// ------------------------------------------------------------------------------------------------ cylinder_push
struct CylinderPush {
  static constexpr int NX = 8, NU = 2, NS = 6, NP = CY_NPARAM, NTP = 6;
  float px, py, cx, cy, pvx, pvy, cvx, cvy;
  __device__ void load(const float* x0) { px = x0[0]; py = x0[1]; cx = x0[2]; cy = x0[3]; pvx = x0[4]; pvy = x0[5]; cvx = x0[6]; cvy = x0[7]; }
  __device__ void store(float* o) const { o[0] = px; o[1] = py; o[2] = cx; o[3] = cy; o[4] = pvx; o[5] = pvy; o[6] = cvx; o[7] = cvy; }
  __device__ void sensors(const float* P, float* s) const { s[0] = px; s[1] = py; s[2] = P[CY_SITE_Z]; s[3] = cx; s[4] = cy; s[5] = P[CY_SITE_Z]; }
  __device__ void step(const float* P, const float* u) {
    const float h = P[CY_DT];
    float c0 = u[0], c1 = u[1];
    if (P[CY_CTRL_LIMITED] != 0.f) { c0 = jh_clampf(c0, P[CY_CTRL_LO], P[CY_CTRL_HI]); c1 = jh_clampf(c1, P[CY_CTRL_LO], P[CY_CTRL_HI]); }
    float fx = P[CY_KP] * (c0 - px) - P[CY_KV] * pvx, fy = P[CY_KP] * (c1 - py) - P[CY_KV] * pvy;
    if (P[CY_FRC_LIMITED] != 0.f) { fx = jh_clampf(fx, P[CY_FRC_LO], P[CY_FRC_HI]); fy = jh_clampf(fy, P[CY_FRC_LO], P[CY_FRC_HI]); }
    float mp = P[CY_MP], mc = P[CY_MC];
    float sp0 = -P[CY_DAMP_P] * pvx + fx, sp1 = -P[CY_DAMP_P] * pvy + fy;  // qfrc_smooth (no gravity / Coriolis in the plane)
    float sc0 = -P[CY_DAMP_C] * cvx, sc1 = -P[CY_DAMP_C] * cvy;
    // circle-circle contact; normal from the pusher (geom 1) to the cart (geom 2)
    float dx = cx - px, dy = cy - py, dn = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);  // (v_sqrt_f32, 1 ulp)
    float dist = dn - P[CY_RSUM];
    float q0 = 0.f, q1 = 0.f;  // constraint force on the cart (= -force on the pusher)
    // (round 6: v_rcp_f32 (1 ulp) and reciprocals of the two masses where the step had nine correctly rounded divisions -- ten dependent instructions each in a step that is one
    // dependent chain; the same choice as cartpole's determinant)
    const float imp_ = __builtin_amdgcn_rcpf(mp), imc_ = __builtin_amdgcn_rcpf(mc);
    if (dist < P[CY_MARGIN] && dn > 1e-12f) {
      const float idn = __builtin_amdgcn_rcpf(dn);
      float nx = dx * idn, ny = dy * idn;
      float imp = impedance_pow(P + CY_SOLIMP0, dist - P[CY_MARGIN]);
      float mu = P[CY_MU];
      // pyramidal cone with mu -> 1e-5: the 2*(condim-1) edge rows coincide with the normal row up to O(mu); their common
      // regulariser is Rpy = 2 mu^2 R_n, so the summed normal force obeys (A + Rpy/4) F = aref - J a0
      float Rn = fmaxf(1e-15f, (1.f - imp) * __builtin_amdgcn_rcpf(imp) * P[CY_TRAN] * (1.f + mu * mu));
      float Rpy = fmaxf(1e-15f, 2.f * mu * mu * Rn);
      float vn = (cvx - pvx) * nx + (cvy - pvy) * ny;
      float aref = -P[CY_CON_B] * vn - P[CY_CON_K] * imp * (dist - P[CY_MARGIN]);
      float a0n = (sc0 * imc_ - sp0 * imp_) * nx + (sc1 * imc_ - sp1 * imp_) * ny;
      float A = imp_ + imc_;
      float F = fmaxf(0.f, (aref - a0n) * __builtin_amdgcn_rcpf(A + 0.25f * Rpy));
      q0 = F * nx; q1 = F * ny;
    }
    float ip = __builtin_amdgcn_rcpf(mp + h * P[CY_DAMP_P]), ic = __builtin_amdgcn_rcpf(mc + h * P[CY_DAMP_C]);
    pvx = fmaf(h, (sp0 - q0) * ip, pvx); pvy = fmaf(h, (sp1 - q1) * ip, pvy);
    cvx = fmaf(h, (sc0 + q0) * ic, cvx); cvy = fmaf(h, (sc1 + q1) * ic, cvy);
    px = fmaf(h, pvx, px); py = fmaf(h, pvy, py); cx = fmaf(h, cvx, cx); cy = fmaf(h, cvy, cy);
  }
  // judo/tasks/cylinder_push.py:65-93 (w = w_pusher_proximity, w_pusher_velocity, w_cart_position, offset, goal_x, goal_y)
  __device__ float cost(const float* w, const float* /*u*/) const {
    float gx = w[4] - cx, gy = w[5] - cy, gn = __builtin_amdgcn_sqrtf(gx * gx + gy * gy);
    const float ign = __builtin_amdgcn_rcpf(gn);
    float tx = cx - w[3] * gx * ign, ty = cy - w[3] * gy * ign;  // no epsilon guard, as in the reference (gn = 0: inf * 0 = NaN, as 0 / 0)
    float ex = px - tx, ey = py - ty;
    return w[0] * 0.5f * (ex * ex + ey * ey) + w[1] * 0.5f * (pvx * pvx + pvy * pvy) + w[2] * 0.5f * (gx * gx + gy * gy);
  }
  __device__ static float finish(float acc, int /*H*/) { return acc; }
};

// ------------------------------------------------------------------------------------------------ kernels
template <class T, int BS>
__device__ __forceinline__ void rollout_cost_body(const float* __restrict__ P, const float* __restrict__ x0,
                                                         const float* __restrict__ nominal, const float* __restrict__ noise, int ldn,
                                                         const float* __restrict__ sigma, const float* __restrict__ W,
                                                         const float* __restrict__ lohi, const float* __restrict__ tp, int N, int n_offset,
                                                         int H, int K, float* __restrict__ costs, float* __restrict__ knots_out, float* __restrict__ trace) {
  extern __shared__ float lds[];
  const int KU = K * T::NU;
  float* sW = lds;                 // H*K
  float* sKn = sW + H * K;         // KU * BS
  float* sP = sKn + KU * BS;   // NP
  float* sTp = sP + T::NP;         // NTP
  float* sX0 = sTp + T::NTP;       // NX
  const int lane = threadIdx.x;
  for (int i = lane; i < H * K; i += BS) sW[i] = W[i];
  for (int i = lane; i < T::NP; i += BS) sP[i] = P[i];
  for (int i = lane; i < T::NTP; i += BS) sTp[i] = tp[i];
  for (int i = lane; i < T::NX; i += BS) sX0[i] = x0[i];
  const int n = blockIdx.x * BS + lane;
  const bool live = n < N;
  const int nc = live ? n : N - 1;
  // sample + clip: knot = clip(nominal + sigma * noise); global sample 0 keeps the nominal (also clipped, controller.py:253)
  for (int i = 0; i < KU; i++) {
    float v = nominal[i];
    if (n_offset + nc != 0) v = fmaf(sigma[i], noise[(size_t)i * ldn + nc], v);
    int u = i % T::NU;
    v = jh_clampf(v, lohi[u], lohi[T::NU + u]);
    sKn[i * BS + lane] = v;
    if (knots_out && live) knots_out[(size_t)i * ldn + n] = v;
  }
  __syncthreads();
  T s; s.load(sX0);
  SplineCtx sp{sW, sKn, K};
  float acc = 0.f;
  for (int h = 0; h < H; h++) {
    float u[T::NU];
    spline_controls<T::NU, BS>(sp, h, lane, u);
    if (trace) {  // jh_rollout_cost_traced: the sensors of this forward pass (all of them are trace sensors in these two models), column-major: element (n, i) at [i * N + n]
      float y[T::NS]; s.sensors(sP, y);
      if (live) for (int k = 0; k < T::NS; k++) trace[(size_t)(h * T::NS + k) * N + n] = y[k];
    }
    s.step(sP, u);
    acc += s.cost(sTp, u);
  }
  if (live) costs[n] = T::finish(acc, H);
}

template <class T>
__global__ __launch_bounds__(kBlock) void k_rollout_cost(const float* __restrict__ P, const float* __restrict__ x0,
                                                         const float* __restrict__ nominal, const float* __restrict__ noise, int ldn,
                                                         const float* __restrict__ sigma, const float* __restrict__ W,
                                                         const float* __restrict__ lohi, const float* __restrict__ tp, int N, int n_offset,
                                                         int H, int K, float* __restrict__ costs, float* __restrict__ knots_out, float* __restrict__ trace) {
  rollout_cost_body<T, kBlock>(P, x0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs, knots_out, trace);
}

// The plan step of a closed-form model as ONE launch (round 6): the rollout + cost body above in workgroups of the update's size (thread t of workgroup b holds local
// rollout b * kUB + t, the layout the update's block stage assumes), then the update's tail in the same launch -- every workgroup's partial records, a ticket, the last
// workgroup merges (jh_update_dev.h; Controller.update_action's body, judo/controller/controller.py:246-299).  A rollout's arithmetic does not depend on the workgroup
// size, the tail is the code of k_update_tail: the same nominal, sigma and trace records as the two launches, one launch latency (and the idle gap in front of it) less.
template <class T>
__global__ __launch_bounds__(jh_upd::kUB) void k_plan_step(const float* __restrict__ P, const float* __restrict__ x0, const float* __restrict__ W, const float* __restrict__ tp, int H, int K,
                                                           jh_upd::TailArgs a) {
  rollout_cost_body<T, jh_upd::kUB>(P, x0, a.src.nominal, a.src.noise, a.src.ldn, a.src.sigma, W, a.src.lohi, tp, a.N, a.n_offset, H, K, const_cast<float*>(a.costs), nullptr,
                                    const_cast<float*>(a.trace));
  __syncthreads();  // (a thread reads back the cost it wrote; the trace rows are read by the last workgroup only, behind the tail's own fence and ticket)
  jh_upd::update_tail_body(a);
}
#ifdef JH_TAIL_TICKS
__global__ void k_tail_ticks_start() { if (threadIdx.x == 0) jh_upd::g_tail_ticks[8] = wall_clock64(); }
#endif

// Drop-in RolloutBackend.rollout: this path IS bound by HBM (it streams 4*H*(nx+ns+nu) bytes per rollout), so the row-major
// (N,H,.) arrays the interface prescribes are moved in tiles: a wave owns 64 rollouts, stages TS time steps of controls /
// states / sensors in LDS (row stride padded to an odd number of words: conflict-free for the per-lane accesses) and moves each
// rollout's TS*width contiguous floats with lane-consecutive addresses.  Full tiles (64 live rollouts, TS steps, 16-byte aligned
// rows) move as float4 with compile-time index arithmetic; ragged tiles take the scalar path.
#ifndef JH_MAT_TS
#define JH_MAT_TS 8  // time steps per LDS tile (measured on MI355X, 1M rollouts: 2 -> 1.3/2.1, 4 -> 2.5/2.9, 8 -> 5.0/5.0, 16 -> 3.8/2.6 TB/s)
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
template <int CH, int STRIDE, bool TO_LDS>
__device__ __forceinline__ void tile_move_full(float* __restrict__ g, size_t pitch, float* __restrict__ sm, int lane) {
  static_assert(CH % 4 == 0, "vector tile path needs a multiple of 4 floats per rollout chunk");
  constexpr int V = CH / 4;
#pragma unroll
  for (int f0 = 0; f0 < kBlock * V; f0 += kBlock) {
    const int f = f0 + lane, r = f / V, i = f - r * V;
    v4f* gp = reinterpret_cast<v4f*>(g + (size_t)r * pitch) + i;
    float* sp = sm + r * STRIDE + 4 * i;
    if (TO_LDS) { v4f v = *gp; sp[0] = v.x; sp[1] = v.y; sp[2] = v.z; sp[3] = v.w; }
    else {
      v4f v = {sp[0], sp[1], sp[2], sp[3]};
      __builtin_nontemporal_store(v, gp);  // written once, never re-read by this kernel (+35 % over a cached store at TS=8)
    }
  }
}

template <class T, int TS>
__global__ __launch_bounds__(kBlock) void k_materialize(const float* __restrict__ P, const float* __restrict__ x0, int x0_batched,
                                                        const float* __restrict__ controls, int N, int H, float* __restrict__ states,
                                                        float* __restrict__ sensors, int vec_ok) {
  constexpr int SU = (TS * T::NU) | 1, SX = (TS * T::NX) | 1, SY = (TS * T::NS) | 1;
  constexpr bool kVec = (TS * T::NU) % 4 == 0 && (TS * T::NX) % 4 == 0 && (TS * T::NS) % 4 == 0;
  __shared__ float sP[T::NP];
  __shared__ float sU[kBlock * SU], sXo[kBlock * SX], sYo[kBlock * SY];
  const int lane = threadIdx.x;
  for (int i = lane; i < T::NP; i += kBlock) sP[i] = P[i];
  const int n0 = blockIdx.x * kBlock, n = n0 + lane;
  const int nvalid = min(kBlock, N - n0);
  float xi[T::NX];
#pragma unroll
  for (int i = 0; i < T::NX; i++) xi[i] = x0[(x0_batched ? (size_t)min(n, N - 1) * T::NX : 0) + i];
  T s; s.load(xi);
  __syncthreads();
  for (int h0 = 0; h0 < H; h0 += TS) {
    const int ts = min(TS, H - h0);
    const bool full = kVec && vec_ok && nvalid == kBlock && ts == TS;  // block-uniform
    // controls tile: rollout r's ts*NU floats are contiguous in global memory
    if constexpr (kVec) {
      if (full) tile_move_full<TS * T::NU, SU, true>(const_cast<float*>(controls) + ((size_t)n0 * H + h0) * T::NU, (size_t)H * T::NU, sU, lane);
    }
    if (!full) for (int f = lane; f < nvalid * ts * T::NU; f += kBlock) { int r = f / (ts * T::NU), i = f - r * (ts * T::NU); sU[r * SU + i] = controls[((size_t)(n0 + r) * H + h0) * T::NU + i]; }
    __syncthreads();
    if (n < N) {
      for (int t = 0; t < ts; t++) {
        float u[T::NU], o[T::NX], y[T::NS];
#pragma unroll
        for (int j = 0; j < T::NU; j++) u[j] = sU[lane * SU + t * T::NU + j];
        s.sensors(sP, y);  // sensor values belong to the forward pass at the start of the step
        s.step(sP, u);
        s.store(o);
#pragma unroll
        for (int i = 0; i < T::NX; i++) sXo[lane * SX + t * T::NX + i] = o[i];
#pragma unroll
        for (int i = 0; i < T::NS; i++) sYo[lane * SY + t * T::NS + i] = y[i];
      }
    }
    __syncthreads();
    bool done = false;
    if constexpr (kVec) {
      if (full) {
        if (states) tile_move_full<TS * T::NX, SX, false>(states + ((size_t)n0 * H + h0) * T::NX, (size_t)H * T::NX, sXo, lane);
        if (sensors) tile_move_full<TS * T::NS, SY, false>(sensors + ((size_t)n0 * H + h0) * T::NS, (size_t)H * T::NS, sYo, lane);
        done = true;
      }
    }
    if (!done) {
      if (states) for (int f = lane; f < nvalid * ts * T::NX; f += kBlock) { int r = f / (ts * T::NX), i = f - r * (ts * T::NX); states[((size_t)(n0 + r) * H + h0) * T::NX + i] = sXo[r * SX + i]; }
      if (sensors) for (int f = lane; f < nvalid * ts * T::NS; f += kBlock) { int r = f / (ts * T::NS), i = f - r * (ts * T::NS); sensors[((size_t)(n0 + r) * H + h0) * T::NS + i] = sYo[r * SY + i]; }
    }
    __syncthreads();
  }
}

template <class T>
__global__ __launch_bounds__(kBlock) void k_reward(const float* __restrict__ states, const float* __restrict__ controls,
                                                   const float* __restrict__ tp, int N, int H, float* __restrict__ rewards) {
  __shared__ float sTp[T::NTP];
  for (int i = threadIdx.x; i < T::NTP; i += kBlock) sTp[i] = tp[i];
  __syncthreads();
  const int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int h = 0; h < H; h++) {
    float xi[T::NX], u[T::NU];
#pragma unroll
    for (int i = 0; i < T::NX; i++) xi[i] = states[((size_t)n * H + h) * T::NX + i];
#pragma unroll
    for (int j = 0; j < T::NU; j++) u[j] = controls ? controls[((size_t)n * H + h) * T::NU + j] : 0.f;
    T s; s.load(xi);
    acc += s.cost(sTp, u);
  }
  rewards[n] = -T::finish(acc, H);
}

template <class T>
int launch_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                const float* lohi, const float* tp, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, hipStream_t st) {
  size_t lds = sizeof(float) * ((size_t)H * K + (size_t)K * T::NU * kBlock + T::NP + T::NTP + T::NX);
  JH_REQUIRE(lds <= 64 * 1024, "rollout_cost: H*K too large for the LDS staging (%zu bytes)", lds);
  int grid = (N + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_rollout_cost<T>, dim3(grid), dim3(kBlock), lds, st, m->d_f, x0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K,
                     costs, knots_out, trace);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

template <class T>
int launch_plan_step(const jh_model* m, const float* x0, const float* W, const float* tp, int H, int K, const jh_upd::TailArgs& a, hipStream_t st) {
  constexpr int BS = jh_upd::kUB;
  size_t lds = sizeof(float) * ((size_t)H * K + (size_t)K * T::NU * BS + T::NP + T::NTP + T::NX);
  JH_REQUIRE(lds <= 48 * 1024, "plan_step: H*K too large for the LDS staging of the one-launch plan step (%zu bytes)", lds);
  int grid = (a.N + BS - 1) / BS;
  hipLaunchKernelGGL(k_plan_step<T>, dim3(grid), dim3(BS), lds, st, m->d_f, x0, W, tp, H, K, a);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

}  // namespace

// can the plan step of this model run as one launch?  (closed-form models; the knots of a workgroup of 256 rollouts must fit the LDS staging next to the update's own 9 KB)
bool jh_simple_plan_step_fits(const jh_model* m, int H, int K) {
  if (m->kind != JH_TASK_CARTPOLE && m->kind != JH_TASK_CYLINDER_PUSH) return false;
  const int nu = m->kind == JH_TASK_CARTPOLE ? Cartpole::NU : CylinderPush::NU;
  const int np = m->kind == JH_TASK_CARTPOLE ? Cartpole::NP + Cartpole::NTP + Cartpole::NX : CylinderPush::NP + CylinderPush::NTP + CylinderPush::NX;
  return sizeof(float) * ((size_t)H * K + (size_t)K * nu * jh_upd::kUB + np) <= 48 * 1024;
}

#ifdef JH_TAIL_TICKS
extern "C" int jh_debug_tail_ticks(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(jh_upd::g_tail_ticks), 16 * sizeof(long long)) == hipSuccess ? 0 : -2; }
#endif
int jh_simple_plan_step(const jh_model* m, const float* x0, const float* W, const float* tp, int H, int K, const jh_upd::TailArgs& a, hipStream_t st) {
  if (m->kind == JH_TASK_CARTPOLE) return launch_plan_step<Cartpole>(m, x0, W, tp, H, K, a, st);
  return launch_plan_step<CylinderPush>(m, x0, W, tp, H, K, a, st);
}

// largest K for which launch_cost's LDS staging (W, the lanes' knots, model / task constants, x0) fits the 64 KiB it may ask for
template <class T>
static int max_knots(int H) { return (int)((16 * 1024 - T::NP - T::NTP - T::NX) / ((size_t)H + (size_t)T::NU * kBlock)); }
int jh_simple_max_knots(const jh_model* m, int H) { return m->kind == JH_TASK_CARTPOLE ? max_knots<Cartpole>(H) : max_knots<CylinderPush>(H); }

int jh_simple_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                           const float* W, const float* lohi, const float* tp, int N, int n_offset, int H, int K, float* costs,
                           float* knots_out, float* trace, hipStream_t st) {
  if (m->kind == JH_TASK_CARTPOLE) return launch_cost<Cartpole>(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs, knots_out, trace, st);
  return launch_cost<CylinderPush>(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs, knots_out, trace, st);
}

int jh_simple_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states,
                          float* sensors, hipStream_t st) {
  int grid = (N + kBlock - 1) / kBlock;
  // float4 tile moves need 16-byte aligned rows: aligned base pointers and H a multiple of the tile length
  const int vec_ok = (H % JH_MAT_TS == 0) && ((reinterpret_cast<uintptr_t>(controls) | reinterpret_cast<uintptr_t>(states) | reinterpret_cast<uintptr_t>(sensors)) & 15) == 0;
  if (m->kind == JH_TASK_CARTPOLE)
    hipLaunchKernelGGL((k_materialize<Cartpole, JH_MAT_TS>), dim3(grid), dim3(kBlock), 0, st, m->d_f, x0, x0_batched, controls, N, H, states, sensors, vec_ok);
  else
    hipLaunchKernelGGL((k_materialize<CylinderPush, JH_MAT_TS>), dim3(grid), dim3(kBlock), 0, st, m->d_f, x0, x0_batched, controls, N, H, states, sensors, vec_ok);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

int jh_simple_reward(const jh_model* m, const float* states, const float* controls, const float* tp, int N, int H, float* rewards, hipStream_t st) {
  int grid = (N + kBlock - 1) / kBlock;
  if (m->kind == JH_TASK_CARTPOLE) hipLaunchKernelGGL(k_reward<Cartpole>, dim3(grid), dim3(kBlock), 0, st, states, controls, tp, N, H, rewards);
  else hipLaunchKernelGGL(k_reward<CylinderPush>, dim3(grid), dim3(kBlock), 0, st, states, controls, tp, N, H, rewards);
  JH_HIP(hipGetLastError());
  return JH_OK;
}
