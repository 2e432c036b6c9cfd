// jh_engine_v2.hip -- cooperative articulated-body rollout kernel for leap_cube (gfx950): 16 lanes per rollout.
//
// Why: the one-lane-per-rollout kernel (jh_engine.hip) keeps ~7.5 KB of per-lane working set in scratch and is bound by
// that traffic (profiles/r01_v1_*: 670 GB of HBM traffic per launch for 17 MB of algorithmic bytes); at 65 536 rollouts it
// also gives the chip only one wave per SIMD.  Here a rollout is spread over one DPP row of 16 lanes (4 rollouts per
// wave64), which maps the model exactly: lane (c,s) owns link s of finger chain c, its joint, its actuator, its dof
// constraint rows and a share of the collision geoms; the cube state is replicated in registers.  Per-rollout shared
// state (body poses, the 22-vectors of the solver, the arrow Hessian, the contact pool) lives in LDS; contacts are
// balanced over the lanes (<= 2 per lane, Jacobian columns in registers); cross-lane sums are DPP/permute butterflies
// inside the row, contact contributions to the gradient and Hessian are LDS float atomics.  Nothing spills except the
// box-box clipping polygon.  Workgroup = one wave, so `__syncthreads()` is a free intra-wave LDS fence and rollouts
// in different waves never wait for each other.
//
// The arithmetic is the same restatement of MuJoCo's step as jh_engine.hip / oracle/jo_engine.c;
// fp32 throughout (an fp32 Hessian was measured to need the same number of Newton iterations as fp64).
#include "jh_coop.h"

using namespace jh_eng;
using namespace jh_coop;

namespace {

constexpr int G = 16, RPW = 4, WAVE = 64;
constexpr int NCH = 4, NLK = 4;
#ifndef JH_V2_LSMAX
#define JH_V2_LSMAX 16  // line-search evaluation cap (MuJoCo: ls_iterations 50); measured on fixed plan inputs: 8 -> 96.5 ms, 10 -> 87.6, 12 -> 82.9, 16 -> 82.5, 24 -> 82.8
#endif
#if !defined(JH_V2_KEEPW) && !defined(JH_V2_RECOMPUTEW)
#define JH_V2_KEEPW 1     // keep the cone weights of the gradient pass for the Hessian pass (12 registers) instead of evaluating the cones twice: -1.5 % on fixed plan inputs
#endif
#ifndef JH_V2_LSBRACKET
#define JH_V2_LSBRACKET 0.f
#endif
#ifndef JH_V2_NSLOT
#define JH_V2_NSLOT 2
#endif
constexpr int NSLOT = JH_V2_NSLOT;
constexpr int NCP = 16 * NSLOT;  // contact pool per rollout = NSLOT slots per lane
constexpr int MAXHIT = 32;  // broad-phase survivors per rollout
constexpr int POOL_F = 10;  // pos3, normal3, dist, mu, body, tran
constexpr int MAXG = 80, MAXLG = 8;  // collision geoms / broad-phase list length per lane staged in LDS
constexpr int NV = 22, NQ = 23, NU = 16, NS = 31, NX = 45, NMB = 17;

struct __attribute__((aligned(16))) RS {  // per-rollout shared state in LDS
  float xpos[NMB][3], xR[NMB][9], axw[NMB][3];
  float qv[NV], g[NV], p[NV];
  float rhs6[6];
  int hits[MAXHIT];
  union {  // the contact pool is dead once every lane has loaded its slots; the Newton Hessian then reuses its storage
    float pool[NCP][POOL_F];
    struct { float Hcc[21], Hbb[NCH][10], Hcb[NCH][24]; };  // Hcb[c][j*6+q]: chain column j, cube row q
  };
  int ncon, nhit;
};

// ------------------------------------------------------------------------------------------------ collision into the LDS pool
struct PoolCtx { RS* S; int* overflow; };

__device__ __forceinline__ void push_contact(const PoolCtx& pc, const float* pos, const float* n, float dist, int body, float mu, float tran) {
  int i = atomicAdd(&pc.S->ncon, 1);
  if (i >= NCP) { if (pc.overflow) atomicAdd(pc.overflow, 1); return; }
  float* e = pc.S->pool[i];
  e[0] = pos[0]; e[1] = pos[1]; e[2] = pos[2]; e[3] = n[0]; e[4] = n[1]; e[5] = n[2]; e[6] = dist; e[7] = mu; e[8] = __int_as_float(body); e[9] = tran;
}

struct LeapSink {  // contact sink of the narrow phase: cube (geom 1) against one hand geom
  PoolCtx pc; int body; float mu, tran;
  __device__ __forceinline__ void push(const float* pos, const float* n, float dist) { push_contact(pc, pos, n, dist, body, mu, tran); }
};

// ------------------------------------------------------------------------------------------------ per-lane contact slot
struct Slot {
  bool valid;
  int chain;         // finger chain of geom 2's body, -1 = static
  float fr[9];       // contact frame rows (normal, t1, t2)
  float Jr[3][3];    // cube rotational columns (negated), Jr[k][row]
  float Jb[NLK][3];  // chain columns, Jb[j][row] (0 beyond the body's depth)
  float aref[3], D[3], Dm, mu, fri;  // Dm = D0 / (mu^2 (1 + mu^2)): middle-zone weight of the elliptic cone
  float jar[3], jp[3];
};

// J x for one slot: xc = cube part (6, registers), chain part read from the LDS vector `vec`
__device__ __forceinline__ void slot_Jx(const Slot& s, const float* xc, const float* vec, float* out) {
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float v = -(s.fr[3 * r] * xc[0] + s.fr[3 * r + 1] * xc[1] + s.fr[3 * r + 2] * xc[2]) + s.Jr[0][r] * xc[3] + s.Jr[1][r] * xc[4] + s.Jr[2][r] * xc[5];
    out[r] = v;
  }
  if (s.chain >= 0) {
    const float* xb = vec + 6 + 4 * s.chain;
#pragma unroll
    for (int j = 0; j < NLK; j++) { float xj = xb[j]; out[0] += s.Jb[j][0] * xj; out[1] += s.Jb[j][1] * xj; out[2] += s.Jb[j][2] * xj; }
  }
}

// cost / derivative / curvature of this lane's rows at jar + al*jp (contacts in both slots + own dof friction/limit rows)
struct DofRows { float fl, fD, fR, faref, lims, laref, lD, jf, jl, pf, pl; };  // fR = 1/fD

__device__ __forceinline__ void lane_rows_eval(const Slot* sl, const DofRows& dr, float al, bool with_dir, float* cost, float* d1, float* d2) {
  float cs = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
  for (int k = 0; k < NSLOT; k++) {
    if (!sl[k].valid) continue;
    float jar[3], f[3], W[6];
    const float* jp = sl[k].jp;
    for (int r = 0; r < 3; r++) jar[r] = sl[k].jar[r] + (with_dir ? al * jp[r] : 0.f);
    cs += cone_eval(jar, sl[k].D, sl[k].Dm, sl[k].mu, sl[k].fri, f, W);
    if (with_dir) {
      g1 -= f[0] * jp[0] + f[1] * jp[1] + f[2] * jp[2];
      g2 += W[0] * jp[0] * jp[0] + W[2] * jp[1] * jp[1] + W[5] * jp[2] * jp[2] + 2.f * (W[1] * jp[0] * jp[1] + W[3] * jp[0] * jp[2] + W[4] * jp[1] * jp[2]);
    }
  }
  if (dr.fl > 0.f) {
    float D = dr.fD, R = dr.fR, jp = dr.pf, x = dr.jf + (with_dir ? al * jp : 0.f), fl = dr.fl;
    if (x <= -R * fl) { cs += -0.5f * R * fl * fl - fl * x; g1 -= fl * jp; }
    else if (x >= R * fl) { cs += -0.5f * R * fl * fl + fl * x; g1 += fl * jp; }
    else { cs += 0.5f * D * x * x; g1 += D * x * jp; g2 += D * jp * jp; }
  }
  if (dr.lims != 0.f) {
    float jp = dr.pl, x = dr.jl + (with_dir ? al * jp : 0.f);
    if (x < 0.f) { cs += 0.5f * dr.lD * x * x; g1 += dr.lD * x * jp; g2 += dr.lD * jp * jp; }
  }
  *cost = cs; *d1 = g1; *d2 = g2;
}

// slope and curvature of the lane's rows along the search direction at step al (line search)
__device__ __forceinline__ void lane_rows_dir(const Slot* sl, const DofRows& dr, float al, float* d1, float* d2) {
  float g1 = 0.f, g2 = 0.f;
#pragma unroll
  for (int k = 0; k < NSLOT; k++) {
    if (!sl[k].valid) continue;
    const float* jp = sl[k].jp;
    float jar[3] = {fmaf(al, jp[0], sl[k].jar[0]), fmaf(al, jp[1], sl[k].jar[1]), fmaf(al, jp[2], sl[k].jar[2])};
    cone_dir(jar, jp, sl[k].D, sl[k].Dm, sl[k].mu, sl[k].fri, &g1, &g2);
  }
  if (dr.fl > 0.f) {
    float D = dr.fD, jp = dr.pf, x = fmaf(al, jp, dr.jf), fl = dr.fl, lim = dr.fR * fl;
    if (x <= -lim) g1 -= fl * jp;
    else if (x >= lim) g1 += fl * jp;
    else { g1 += D * x * jp; g2 += D * jp * jp; }
  }
  if (dr.lims != 0.f) {
    float jp = dr.pl, x = fmaf(al, jp, dr.jl);
    if (x < 0.f) { g1 += dr.lD * x * jp; g2 += dr.lD * jp * jp; }
  }
  *d1 = g1; *d2 = g2;
}

// 4x4 Cholesky (packed lower) + triangular solves on registers; the diagonal is kept as its reciprocal (one v_rsq per
// pivot, no divisions in the solves)
__device__ __forceinline__ void chol4(float* L, float* inv) {
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float s = L[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[tri(i, k)] * L[tri(j, k)];
      if (i == j) { float r = __frsqrt_rn(fmaxf(s, 1e-30f)); inv[i] = r; L[tri(i, i)] = s * r; }
      else L[tri(i, j)] = s * inv[j];
    }
}
__device__ __forceinline__ void fwd4(const float* L, const float* inv, float* x) {
#pragma unroll
  for (int i = 0; i < 4; i++) { float s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k];
    x[i] = s * inv[i]; }
}
__device__ __forceinline__ void bwd4(const float* L, const float* inv, float* x) {
#pragma unroll
  for (int i = 3; i >= 0; i--) { float s = x[i];
#pragma unroll
    for (int k = i + 1; k < 4; k++) s -= L[tri(k, i)] * x[k];
    x[i] = s * inv[i]; }
}

struct LaneConst {  // per-lane model constants (own joint / actuator / dof rows), loaded once
  float damp, kvd, kp, kv, clo, chi, clim, fl, fB, fD, invw, limited, lo, hi, lK, lB, si[5];
};

// ------------------------------------------------------------------------------------------------ the kernel
#ifndef JH_V2_WAVES_PER_EU
#define JH_V2_WAVES_PER_EU 1
#endif
template <bool MATERIALIZE>
__global__ __launch_bounds__(WAVE, JH_V2_WAVES_PER_EU) void k_leap_v2(const float* __restrict__ gF, const int* __restrict__ gI, const float* __restrict__ x0, int x0_batched,
                                                   const float* __restrict__ nominal, const float* __restrict__ noise, int ldn,
                                                   const float* __restrict__ sigma, const float* __restrict__ W, const float* __restrict__ lohi,
                                                   const float* __restrict__ tp, int N, int n_offset, int H, int K, float* __restrict__ costs,
                                                   float* __restrict__ knots_out, const float* __restrict__ controls, float* __restrict__ states,
                                                   float* __restrict__ sensors, int* __restrict__ stats) {
  __shared__ RS sRS[RPW];
  __shared__ float sBody[16 * BODY_F];  // records of the 16 finger links
  __shared__ float sTp[16];
  __shared__ float sGeomF[MAXG * GEOM_F];
  __shared__ int sGeomI[MAXG * GEOM_I];
  __shared__ int sLaneG[16 * MAXLG];
  const int lane = threadIdx.x, l = lane & 15, r = lane >> 4, c = l >> 2, s = l & 3, cb = lane & ~3;
  RS& S = sRS[r];
  const int nmI = gI[0], nblkI = gI[1], nuI = gI[4], ngI = gI[5], nsiteI = gI[6];
  const int oBodyF = HEADER_F, oDofF = oBodyF + nmI * BODY_F, oActF = oDofF + gI[2] * DOF_F, oGeomF = oActF + nuI * ACT_F, oSiteF = oGeomF + ngI * GEOM_F;
  const int oGeomI = HEADER_I + nmI * BODY_I + nblkI * BLOCK_I + nuI * ACT_I, oSiteI = oGeomI + ngI * GEOM_I;
  const int oLane = gI[11], lgm = gI[12];
  for (int i = lane; i < 16 * BODY_F; i += WAVE) sBody[i] = gF[oBodyF + BODY_F + i];
  for (int i = lane; i < ngI * GEOM_F; i += WAVE) sGeomF[i] = gF[oGeomF + i];
  for (int i = lane; i < ngI * GEOM_I; i += WAVE) sGeomI[i] = gI[oGeomI + i];
  for (int i = lane; i < 16 * lgm; i += WAVE) sLaneG[i] = gI[oLane + i];
  if (!MATERIALIZE && lane < 9) sTp[lane] = tp[lane];
  const int n = blockIdx.x * RPW + r;  // rollout handled by this row of 16 lanes
  const bool live = n < N;
  const int nc = live ? n : N - 1;
  // ---- per-lane constants
  LaneConst lc;
  {
    const float* df = gF + oDofF + (6 + l) * DOF_F; const float* af = gF + oActF + l * ACT_F;
    lc.damp = df[DF_DAMP]; lc.kvd = df[DF_KV]; lc.fl = df[DF_FL]; lc.fB = df[DF_FB]; lc.fD = df[DF_FD]; lc.invw = df[DF_INVW];
    lc.limited = df[DF_LIMITED]; lc.lo = df[DF_LO]; lc.hi = df[DF_HI]; lc.lK = df[DF_LK]; lc.lB = df[DF_LB];
    for (int k = 0; k < 5; k++) lc.si[k] = df[DF_SOLIMP + k];
    lc.kp = af[AF_KP]; lc.kv = af[AF_KV]; lc.clim = af[AF_CLIM]; lc.clo = af[AF_CLO]; lc.chi = af[AF_CHI];
  }
  const float h = gF[HF_DT], impratio = gF[HF_IMPRATIO], tol = gF[HF_TOL], lstol = gF[HF_LSTOL]; const int cap = (int)gF[HF_MAXITER];
  const float grav[3] = {gF[HF_GRAV], gF[HF_GRAV + 1], gF[HF_GRAV + 2]};
  const float cmass = gF[HF_CMASS], cI[3] = {gF[HF_CINERTIA], gF[HF_CINERTIA + 1], gF[HF_CINERTIA + 2]};
  const float chs[3] = {gF[HF_CSIZE], gF[HF_CSIZE + 1], gF[HF_CSIZE + 2]}, crb = gF[HF_CRBOUND], ctran = gF[HF_CTRAN];
  const float cK = gF[HF_CK], cB = gF[HF_CB];
  float csi[5]; for (int k = 0; k < 5; k++) csi[k] = gF[HF_SOLIMP + k];
  // ---- state: own joint + replicated cube
  float q, qd, qws = 0.f, qc[7], vc[6], wsc[6] = {0, 0, 0, 0, 0, 0};
  {
    const float* xi = x0 + ((MATERIALIZE && x0_batched) ? (size_t)nc * NX : 0);
    for (int k = 0; k < 7; k++) qc[k] = xi[k];
    for (int k = 0; k < 6; k++) vc[k] = xi[NQ + k];
    q = xi[7 + l]; qd = xi[NQ + 6 + l];
  }
  // ---- own actuator's spline knots (fused mode): clip(nominal + sigma*noise); global sample 0 keeps the nominal
  float kn[8];
  if (!MATERIALIZE) {
    for (int k = 0; k < 8; k++) kn[k] = 0.f;
    for (int k = 0; k < K && k < 8; k++) {
      int i = k * NU + l;
      float v = nominal[i];
      if (n_offset + nc != 0) v = fmaf(sigma[i], noise[(size_t)i * ldn + nc], v);
      v = jh_clampf(v, lohi[l], lohi[NU + l]);
      kn[k] = v;
#ifndef JH_V2_ITERDUMP
      if (knots_out && live) knots_out[(size_t)i * ldn + n] = v;
#endif
    }
  }
  int n_overflow = 0, n_iters = 0, n_maxed = 0;
  float acc = 0.f;
  // JH_V2_ABLATE (tools/diag/time_ablate.py): repeat one phase gI[21] times (gI[20] selects it) to measure its share of the step;
  // the repeated work is idempotent, V2_OPAQUE keeps the compiler from hoisting it out of the repeat loop
#ifdef JH_V2_ABLATE
  const int ab_phase = gI[20], ab_reps = gI[21] > 0 ? gI[21] : 1;
#define V2_REPEAT(phase) for (int rep__ = 0, nrep__ = (ab_phase == (phase) ? ab_reps : 1); rep__ < nrep__; rep__++)
#define V2_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define V2_REPEAT(phase)
#define V2_OPAQUE(x)
#endif
#ifdef JH_ENGINE_PROFILE
  long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = clock64();
#define V2_TICK(slot) { long long t__ = clock64(); cyc[slot] += t__ - t0; t0 = t__; }
#else
#define V2_TICK(slot)
#endif
  __syncthreads();

  for (int hh = 0; hh < H; hh++) {
    // ================================================================ controls
    float u;
    if (MATERIALIZE) u = controls[((size_t)nc * H + hh) * NU + l];
    else { u = 0.f; for (int k = 0; k < K && k < 8; k++) u = fmaf(W[hh * K + k], kn[k], u); }
    // ================================================================ kinematics (each lane walks its chain up to its own link)
    float ax[NLK][3], og[NLK][3], Rown[9], pown[3], Rc[9];
    float Mc[10], fs_own, a0_own, fsc[6], a0c[6];
    V2_REPEAT(5) {
    V2_OPAQUE(q); V2_OPAQUE(qd);
    {
      float nn = rsqrtf(qc[3] * qc[3] + qc[4] * qc[4] + qc[5] * qc[5] + qc[6] * qc[6]);
      qc[3] *= nn; qc[4] *= nn; qc[5] *= nn; qc[6] *= nn;
      quat2mat(Rc, qc + 3);
      float P[3] = {0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      float sn_own, cs_own; sincosf(q, &sn_own, &cs_own);  // each lane evaluates its own joint once; the chain-mates fetch it with a quad broadcast
#pragma unroll
      for (int j = 0; j < NLK; j++) {
        const float* bf = sBody + (4 * c + j) * BODY_F;
        float P2[3], R0[9];
        if (j == 0) { for (int k = 0; k < 3; k++) P2[k] = bf[BF_LPOS + k]; for (int k = 0; k < 9; k++) R0[k] = bf[BF_LR + k]; }
        else { mulMV(P2, R, bf + BF_LPOS); for (int k = 0; k < 3; k++) P2[k] += P[k]; mulMM(R0, R, bf + BF_LR); }
        const float* al = bf + BF_AXIS;
        mulMV(ax[j], R0, al);
        for (int k = 0; k < 3; k++) { og[j][k] = P2[k]; P[k] = P2[k]; }
        const float sn = quad_get(sn_own, j), cs = quad_get(cs_own, j), t = 1.f - cs, x = al[0], y = al[1], z = al[2];
        float Rq[9] = {t * x * x + cs, t * x * y - sn * z, t * x * z + sn * y, t * x * y + sn * z, t * y * y + cs, t * y * z - sn * x, t * x * z - sn * y, t * y * z + sn * x, t * z * z + cs};
        mulMM(R, R0, Rq);
        if (j == s) { for (int k = 0; k < 3; k++) pown[k] = P2[k]; for (int k = 0; k < 9; k++) Rown[k] = R[k]; }
      }
      for (int k = 0; k < 3; k++) { S.xpos[1 + l][k] = pown[k]; S.axw[1 + l][k] = ax[s][k]; }
      for (int k = 0; k < 9; k++) S.xR[1 + l][k] = Rown[k];
      if (l == 0) { for (int k = 0; k < 3; k++) S.xpos[0][k] = qc[k]; for (int k = 0; k < 9; k++) S.xR[0][k] = Rc[k]; S.ncon = 0; S.nhit = 0; }
      S.qv[6 + l] = qd;
      if (l < 6) S.qv[l] = vc[l];
    }
    // sensors of this forward pass (materialise mode): 16 joint positions, then 5 site positions
    if (MATERIALIZE && sensors) {
      float* y = sensors + ((size_t)nc * H + hh) * NS;
      if (live) y[l] = q;
      __syncthreads();
      if (live && l < nsiteI && l < 5) {
        int b = gI[oSiteI + l]; float p3[3]; mulMV(p3, S.xR[b], gF + oSiteF + l * SITE_F);
        for (int k = 0; k < 3; k++) y[16 + 3 * l + k] = p3[k] + S.xpos[b][k];
      }
    }
    V2_TICK(0)
    // ================================================================ chain dynamics: inertia block, bias, smooth force
    {
      const float* bf = sBody + (4 * c + s) * BODY_F;
      float Rk[9], rr[3], cs3[3]; mulMM(Rk, Rown, bf + BF_IR); mulMV(rr, Rown, bf + BF_IPOS);
      for (int k = 0; k < 3; k++) cs3[k] = pown[k] + rr[k];
      const float mass = bf[BF_MASS]; const float* di = bf + BF_INERTIA;
      // velocities / bias accelerations of the own link (walk down the chain; parent of link 0 is static)
      float wv[3] = {0, 0, 0}, al[3] = {0, 0, 0}, ao[3] = {-grav[0], -grav[1], -grav[2]};
#pragma unroll
      for (int j = 0; j < NLK; j++) {
        float qdj = quad_get(qd, j);
        if (j <= s) {
          if (j > 0) {
            float d[3] = {og[j][0] - og[j - 1][0], og[j][1] - og[j - 1][1], og[j][2] - og[j - 1][2]}, t1[3], t2[3], t3[3];
            cross3(t1, wv, d); cross3(t2, wv, t1); cross3(t3, al, d);
            for (int k = 0; k < 3; k++) ao[k] += t3[k] + t2[k];
          }
          float wxa[3]; cross3(wxa, wv, ax[j]);
          for (int k = 0; k < 3; k++) { al[k] += wxa[k] * qdj; wv[k] += ax[j][k] * qdj; }
        }
      }
      float t1[3], t2[3], t3[3], ac[3];
      cross3(t1, wv, rr); cross3(t2, wv, t1); cross3(t3, al, rr);
      for (int k = 0; k < 3; k++) ac[k] = ao[k] + t3[k] + t2[k];
      float Iw[3], Ia[3], gy[3]; inertia_mul(Iw, Rk, di, wv); inertia_mul(Ia, Rk, di, al); cross3(gy, wv, Iw);
      float Fk[3] = {mass * ac[0], mass * ac[1], mass * ac[2]}, Nk[3] = {Ia[0] + gy[0], Ia[1] + gy[1], Ia[2] + gy[2]};
      // contributions of the own link to the chain's bias vector and inertia block
      float bias[NLK];
      for (int k = 0; k < 10; k++) Mc[k] = 0.f;
#pragma unroll
      for (int i = 0; i < NLK; i++) {
        bias[i] = 0.f;
        if (i <= s) {
          float ri[3] = {cs3[0] - og[i][0], cs3[1] - og[i][1], cs3[2] - og[i][2]}, rxF[3], Jvi[3], tB[3];
          cross3(rxF, ri, Fk);
          bias[i] = ax[i][0] * (Nk[0] + rxF[0]) + ax[i][1] * (Nk[1] + rxF[1]) + ax[i][2] * (Nk[2] + rxF[2]);
          cross3(Jvi, ax[i], ri); inertia_mul(tB, Rk, di, ax[i]);
#pragma unroll
          for (int j = 0; j <= i; j++) {
            float rj[3] = {cs3[0] - og[j][0], cs3[1] - og[j][1], cs3[2] - og[j][2]}, Jvj[3]; cross3(Jvj, ax[j], rj);
            Mc[tri(i, j)] = mass * dot3(Jvi, Jvj) + dot3(tB, ax[j]);
          }
        }
      }
      for (int k = 0; k < 10; k++) Mc[k] = csum(Mc[k]);
      float bown = 0.f;
#pragma unroll
      for (int i = 0; i < NLK; i++) { float b = csum(bias[i]); if (i == s) bown = b; }
      // position servo on the own joint
      float cc = u; if (lc.clim != 0.f) cc = jh_clampf(cc, lc.clo, lc.chi);
      fs_own = -lc.damp * qd - bown + lc.kp * (cc - q) - lc.kv * qd;
      float x4[4], L[10];
#pragma unroll
      for (int j = 0; j < NLK; j++) x4[j] = quad_get(fs_own, j);
      for (int k = 0; k < 10; k++) L[k] = Mc[k];
      float inv4[4]; chol4(L, inv4); fwd4(L, inv4, x4); bwd4(L, inv4, x4);
      a0_own = x4[0];
#pragma unroll
      for (int j = 1; j < NLK; j++) if (j == s) a0_own = x4[j];
      // free cube: M = diag(m,m,m,I)
      float Icw[3] = {cI[0] * vc[3], cI[1] * vc[4], cI[2] * vc[5]}, gc[3]; cross3(gc, vc + 3, Icw);
      for (int k = 0; k < 3; k++) { fsc[k] = cmass * grav[k]; a0c[k] = grav[k]; fsc[3 + k] = -gc[k]; a0c[3 + k] = -gc[k] / cI[k]; }
    }
    }  // V2_REPEAT(5)
    __syncthreads();
    V2_TICK(1)
    // ================================================================ collision: broad phase on the lane's geoms, balanced narrow phase
    V2_REPEAT(4) {
#ifdef JH_V2_ABLATE
      if (rep__ > 0) { __syncthreads(); if (l == 0) S.ncon = 0; __syncthreads(); }
      V2_OPAQUE(qc[0]);
#endif
      int nh = 0;
      for (int i = 0; i < lgm; i++) {
        int gid = sLaneG[l * lgm + i];
        bool hit = false;
        if (gid >= 0) {
          const float* gf = sGeomF + gid * GEOM_F;
          float gp[3];
          if (sGeomI[gid * GEOM_I] < 0) { gp[0] = gf[GF_POS]; gp[1] = gf[GF_POS + 1]; gp[2] = gf[GF_POS + 2]; }
          else { mulMV(gp, Rown, gf + GF_POS); gp[0] += pown[0]; gp[1] += pown[1]; gp[2] += pown[2]; }
          float dc[3] = {gp[0] - qc[0], gp[1] - qc[1], gp[2] - qc[2]}, rs = gf[GF_RBOUND] + crb;
          hit = dot3(dc, dc) <= rs * rs;
          if (hit) {  // second filter: the geom's bounding sphere against the cube's box, and the cube's bounding sphere against the geom's box
            float cl[3]; mulMTV(cl, Rc, dc);
            hit = fabsf(cl[0]) <= chs[0] + gf[GF_RBOUND] && fabsf(cl[1]) <= chs[1] + gf[GF_RBOUND] && fabsf(cl[2]) <= chs[2] + gf[GF_RBOUND];
            if (hit && sGeomI[gid * GEOM_I + 1] == GBOX) {
              float gR[9], gl[3];
              if (sGeomI[gid * GEOM_I] < 0) { for (int k = 0; k < 9; k++) gR[k] = gf[GF_R + k]; } else mulMM(gR, Rown, gf + GF_R);
              mulMTV(gl, gR, dc);
              hit = fabsf(gl[0]) <= gf[GF_SIZE] + crb && fabsf(gl[1]) <= gf[GF_SIZE + 1] + crb && fabsf(gl[2]) <= gf[GF_SIZE + 2] + crb;
            }
          }
        }
        unsigned m16 = (unsigned)((__ballot(hit) >> (16 * r)) & 0xFFFFull);
        int pos = nh + __popc(m16 & ((1u << l) - 1u));
        if (hit && pos < MAXHIT) S.hits[pos] = gid;
        nh += __popc(m16);
      }
      // (Survivors beyond the list are dropped without being counted, as in round 1.  Counting them here -- one global atomic -- changes this kernel's results on
      // leap_cube_down from the second step on although the arithmetic is untouched: suspected code-generation issue of the 487-register build (SGPR spills under
      // divergent control flow).  Generation 2 is kept exactly as it was validated; generation 3 counts and holds 64.)
      nh = nh < MAXHIT ? nh : MAXHIT;
#ifdef JH_ENGINE_PROFILE
#ifndef JH_V2_LSHIST
      if (l == 0 && live && stats) atomicAdd(stats + 48 + (nh < 15 ? nh : 15), 1);
#endif
#endif
      __syncthreads();
      PoolCtx pc{&S, stats};
      for (int base = 0; __any(base < nh); base += G) {
        int idx = base + l;
        if (idx < nh) {
          int gid = S.hits[idx];
          const float* gf = sGeomF + gid * GEOM_F; int body = sGeomI[gid * GEOM_I], gtype = sGeomI[gid * GEOM_I + 1];
          float gp[3], gR[9];
          if (body < 0) { for (int k = 0; k < 3; k++) gp[k] = gf[GF_POS + k]; for (int k = 0; k < 9; k++) gR[k] = gf[GF_R + k]; }
          else {
            float bR[9]; for (int k = 0; k < 9; k++) bR[k] = S.xR[body][k];
            mulMV(gp, bR, gf + GF_POS); for (int k = 0; k < 3; k++) gp[k] += S.xpos[body][k];
            mulMM(gR, bR, gf + GF_R);
          }
          float tran = ctran + gf[GF_TRAN];
          LeapSink sk{pc, body, gf[GF_MU], tran};
          if (gtype == GBOX) collide_box_box(sk, qc, Rc, chs, gp, gR, gf + GF_SIZE);
          else collide_box_sphere(sk, qc, Rc, chs, gp, gf[GF_SIZE]);
        }
      }
    }
    __syncthreads();
    V2_TICK(2)
    // ================================================================ constraint rows: <= 2 contacts per lane + the own dof's friction-loss / limit rows
    const int ncon = S.ncon < NCP ? S.ncon : NCP;
    Slot sl[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
      int idx = l + 16 * k;
      sl[k].valid = idx < ncon;
      sl[k].chain = -1;
      if (sl[k].valid) {
        const float* e = S.pool[idx];
        float pos[3] = {e[0], e[1], e[2]};
        sl[k].fr[0] = e[3]; sl[k].fr[1] = e[4]; sl[k].fr[2] = e[5];
        make_frame(sl[k].fr);
        float dist = e[6], mu = e[7], tran = e[9]; int body = __float_as_int(e[8]);
        float rc3[3] = {pos[0] - qc[0], pos[1] - qc[1], pos[2] - qc[2]};
        for (int a3 = 0; a3 < 3; a3++) { float axc[3], c3[3]; col3(axc, Rc, a3); cross3(c3, axc, rc3); for (int rw = 0; rw < 3; rw++) sl[k].Jr[a3][rw] = -dot3(sl[k].fr + 3 * rw, c3); }
        for (int j = 0; j < NLK; j++) sl[k].Jb[j][0] = sl[k].Jb[j][1] = sl[k].Jb[j][2] = 0.f;
        if (body > 0) {
          int ch = (body - 1) >> 2, dep = (body - 1) & 3;
          sl[k].chain = ch;
          for (int j = 0; j < NLK; j++) if (j <= dep) {
            int bj = 1 + 4 * ch + j;
            float rb[3] = {pos[0] - S.xpos[bj][0], pos[1] - S.xpos[bj][1], pos[2] - S.xpos[bj][2]}, aj[3] = {S.axw[bj][0], S.axw[bj][1], S.axw[bj][2]}, c3[3];
            cross3(c3, aj, rb);
            for (int rw = 0; rw < 3; rw++) sl[k].Jb[j][rw] = dot3(sl[k].fr + 3 * rw, c3);
          }
        }
        float imp = impedance(csi, dist);
        float R0 = fmaxf(1e-15f, (1.f - imp) / imp * tran), R1 = R0 / fmaxf(1e-15f, impratio);
        sl[k].D[0] = 1.f / R0; sl[k].D[1] = 1.f / R1; sl[k].D[2] = 1.f / R1;
        sl[k].fri = mu; sl[k].mu = mu * sqrtf(R1 / R0);
        { float m2 = sl[k].mu * sl[k].mu; sl[k].Dm = sl[k].D[0] / (m2 * (1.f + m2)); }
        float vel[3]; slot_Jx(sl[k], vc, S.qv, vel);
        sl[k].aref[0] = -cB * vel[0] - cK * imp * dist; sl[k].aref[1] = -cB * vel[1]; sl[k].aref[2] = -cB * vel[2];
      }
    }
    DofRows dr;
    dr.fl = lc.fl; dr.fD = lc.fD; dr.fR = lc.fD > 0.f ? 1.f / lc.fD : 0.f; dr.faref = -lc.fB * qd; dr.lims = 0.f; dr.laref = 0.f; dr.lD = 0.f; dr.jf = dr.jl = dr.pf = dr.pl = 0.f;
    if (lc.limited != 0.f) {
      float dlo = q - lc.lo, dhi = lc.hi - q, dist = fminf(dlo, dhi);
      if (dist < 0.f) {
        float sg = dlo < dhi ? 1.f : -1.f, imp = impedance(lc.si, dist), R = fmaxf(1e-15f, (1.f - imp) / imp * lc.invw);
        dr.lims = sg; dr.lD = 1.f / R; dr.laref = -lc.lB * (sg * qd) - lc.lK * imp * dist;
      }
    }
    V2_TICK(3)
    // ================================================================ Newton solver (rows distributed over the 16 lanes)
    float a_own, ac[6];
    float Mdiag_own = Mc[0];
#pragma unroll
    for (int j = 1; j < NLK; j++) if (j == s) Mdiag_own = Mc[tri(j, j)];
    float Mrow[NLK];  // row s of the chain inertia block
#pragma unroll
    for (int j = 0; j < NLK; j++) { float v = 0.f;
#pragma unroll
      for (int i = 0; i < NLK; i++) if (i == s) v = Mc[i >= j ? tri(i, j) : tri(j, i)];
      Mrow[j] = v; }
    const float snorm = gsum(fs_own * fs_own / Mdiag_own + (l < 3 ? fsc[l] * fsc[l] / cmass : (l < 6 ? fsc[l] * fsc[l] / cI[l - 3] : 0.f)));
    bool anyslot = false;
    for (int k = 0; k < NSLOT; k++) anyslot |= sl[k].valid;
    const bool has_rows = gor((int)(anyslot || dr.fl > 0.f || dr.lims != 0.f)) != 0;
    int iters_this = 0;
    if (!has_rows) { a_own = a0_own; for (int k = 0; k < 6; k++) ac[k] = a0c[k]; }
    else {
      // ---- warm start: the better of last step's acceleration and the unconstrained one
      float cost_ws, cost_0;
      {
        S.p[6 + l] = qws; if (l < 6) S.p[l] = wsc[l];
        __syncthreads();
        float cs, d1, d2, jx[3];
        for (int k = 0; k < NSLOT; k++) if (sl[k].valid) { slot_Jx(sl[k], wsc, S.p, jx); for (int rw = 0; rw < 3; rw++) sl[k].jar[rw] = jx[rw] - sl[k].aref[rw]; }
        dr.jf = qws - dr.faref; dr.jl = dr.lims * qws - dr.laref;
        lane_rows_eval(sl, dr, 0.f, false, &cs, &d1, &d2);
        float dws = qws - a0_own, md = 0.f;
#pragma unroll
        for (int j = 0; j < NLK; j++) md += Mrow[j] * (quad_get(qws, j) - quad_get(a0_own, j));
        cs += 0.5f * dws * md;
        if (l < 6) { float dcw = wsc[l] - a0c[l]; cs += 0.5f * dcw * dcw * (l < 3 ? cmass : cI[l < 3 ? 0 : l - 3]); }
        cost_ws = gsum(cs);
        __syncthreads();
        S.p[6 + l] = a0_own; if (l < 6) S.p[l] = a0c[l];
        __syncthreads();
        float jar0[NSLOT][3];
        for (int k = 0; k < NSLOT; k++) if (sl[k].valid) { slot_Jx(sl[k], a0c, S.p, jx); for (int rw = 0; rw < 3; rw++) { jar0[k][rw] = sl[k].jar[rw]; sl[k].jar[rw] = jx[rw] - sl[k].aref[rw]; } }
        float jf_ws = dr.jf, jl_ws = dr.jl;
        dr.jf = a0_own - dr.faref; dr.jl = dr.lims * a0_own - dr.laref;
        lane_rows_eval(sl, dr, 0.f, false, &cs, &d1, &d2);
        cost_0 = gsum(cs);
        const bool use_ws = cost_ws < cost_0;
        if (use_ws) {
          a_own = qws; for (int k = 0; k < 6; k++) ac[k] = wsc[k];
          for (int k = 0; k < NSLOT; k++) if (sl[k].valid) for (int rw = 0; rw < 3; rw++) sl[k].jar[rw] = jar0[k][rw];
          dr.jf = jf_ws; dr.jl = jl_ws;
        } else { a_own = a0_own; for (int k = 0; k < 6; k++) ac[k] = a0c[k]; }
        __syncthreads();
      }
      bool act = true;
      const float mck = (l < 3 ? cmass : (l < 6 ? cI[l < 3 ? 0 : l - 3] : 0.f)), imck = l < 6 ? 1.f / mck : 0.f, iMd = 1.f / Mdiag_own;
      V2_TICK(3)
      for (int it = 0; it < cap && __any(act); it++) {
#ifdef JH_V2_ABLATE
        for (int rep__ = 0, nrep__ = (ab_phase == 1 || ab_phase == 2) ? ab_reps : 1; rep__ < nrep__; rep__++) {
        for (int k = 0; k < NSLOT; k++) { V2_OPAQUE(sl[k].jar[0]); V2_OPAQUE(sl[k].jar[1]); V2_OPAQUE(sl[k].jar[2]); }
        V2_OPAQUE(a_own);
#endif
        // ---- (1) gradient.  Owner lanes: M (a - a0) rows + dof-row forces; contacts: -J'f (chain part: LDS float atomics, cube part: row sums)
        float da_own = a_own - a0_own, g_own = 0.f, hd = 0.f, dac[NLK];
#pragma unroll
        for (int j = 0; j < NLK; j++) { dac[j] = quad_get(da_own, j); g_own += Mrow[j] * dac[j]; }
        if (dr.fl > 0.f) {
          float D = dr.fD, x = dr.jf, fl = dr.fl, lim = dr.fR * fl;
          if (x <= -lim) g_own -= fl; else if (x >= lim) g_own += fl; else { g_own += D * x; hd += D; }
        }
        if (dr.lims != 0.f && dr.jl < 0.f) { g_own += dr.lims * dr.lD * dr.jl; hd += dr.lD; }
        if (act) S.g[6 + l] = g_own;
        __syncthreads();
        float gcp[6] = {0, 0, 0, 0, 0, 0};
#ifdef JH_V2_KEEPW
        float Wm[NSLOT][6];
#endif
        if (act) {
#pragma unroll
          for (int k = 0; k < NSLOT; k++) if (sl[k].valid) {
            float f[3];
#ifdef JH_V2_KEEPW
            cone_eval(sl[k].jar, sl[k].D, sl[k].Dm, sl[k].mu, sl[k].fri, f, Wm[k]);
#else
            float Wtmp[6]; cone_eval(sl[k].jar, sl[k].D, sl[k].Dm, sl[k].mu, sl[k].fri, f, Wtmp);
#endif
            const Slot& t = sl[k];
            if (f[0] == 0.f && f[1] == 0.f && f[2] == 0.f) continue;  // separated contact: no force, nothing to add (the masked lanes issue no LDS atomics: -0.9 %)
            for (int q3 = 0; q3 < 3; q3++) {  // cube columns: translation q3 -> -fr[row][q3]; rotation -> Jr[q3][row]
              gcp[q3] += t.fr[q3] * f[0] + t.fr[3 + q3] * f[1] + t.fr[6 + q3] * f[2];
              gcp[3 + q3] -= t.Jr[q3][0] * f[0] + t.Jr[q3][1] * f[1] + t.Jr[q3][2] * f[2];
            }
            if (t.chain >= 0) for (int j = 0; j < NLK; j++) atomicAdd(&S.g[6 + 4 * t.chain + j], -(t.Jb[j][0] * f[0] + t.Jb[j][1] * f[1] + t.Jb[j][2] * f[2]));
          }
        }
        float gcl = 0.f;
#pragma unroll
        for (int q6 = 0; q6 < 6; q6++) { float v = gsum(gcp[q6]); if (q6 == l) gcl = v; }
        float dcl = 0.f;  // own cube dof's (a - a0); lanes 6..15 carry zeros (mck = 0)
#pragma unroll
        for (int k = 0; k < 6; k++) if (k == l) dcl = ac[k] - a0c[k];
        gcl = fmaf(mck, dcl, gcl);
        __syncthreads();
        V2_TICK(4)
        // ---- (2) convergence on the scaled gradient; the wave leaves the loop before any Hessian work once all its rollouts are done
        g_own = S.g[6 + l];
        float gn = gsum(g_own * g_own * iMd + gcl * gcl * imck);
        if (act && gn <= tol * tol * fmaxf(snorm, 1e-12f)) act = false;
        if (!__any(act)) break;
        if (act) iters_this++;
        // ---- (3) Hessian: M + dof rows on the chain diagonals, J'WJ of the contacts into the arrow blocks
        if (act) {
#pragma unroll
          for (int j = 0; j < NLK; j++) if (j <= s) S.Hbb[c][tri(s, j)] = Mrow[j] + (j == s ? hd : 0.f);
          for (int k = 0; k < 6; k++) S.Hcb[c][s * 6 + k] = 0.f;
        }
        __syncthreads();
        float hcp[21];
        for (int e = 0; e < 21; e++) hcp[e] = 0.f;
        if (act) {
#pragma unroll
          for (int k = 0; k < NSLOT; k++) if (sl[k].valid) {
            const Slot& t = sl[k];
#ifdef JH_V2_KEEPW
            const float* Wk = Wm[k];
#else
            float Wk[6], ftmp[3]; cone_eval(t.jar, t.D, t.Dm, t.mu, t.fri, ftmp, Wk);  // recomputed: cheaper than 12 registers live across the convergence test
#endif
            if (!(Wk[0] == 0.f && Wk[2] == 0.f && Wk[5] == 0.f)) {
              // column by column (one 3-vector W J_v live at a time keeps the register pressure of this block low)
              float Jc[6][3];
              for (int q3 = 0; q3 < 3; q3++) { Jc[q3][0] = -t.fr[q3]; Jc[q3][1] = -t.fr[3 + q3]; Jc[q3][2] = -t.fr[6 + q3]; Jc[3 + q3][0] = t.Jr[q3][0]; Jc[3 + q3][1] = t.Jr[q3][1]; Jc[3 + q3][2] = t.Jr[q3][2]; }
#pragma unroll
              for (int v6 = 0; v6 < 6; v6++) {
                const float* j3 = Jc[v6];
                const float G0 = Wk[0] * j3[0] + Wk[1] * j3[1] + Wk[3] * j3[2], G1 = Wk[1] * j3[0] + Wk[2] * j3[1] + Wk[4] * j3[2], G2 = Wk[3] * j3[0] + Wk[4] * j3[1] + Wk[5] * j3[2];
#pragma unroll
                for (int u6 = v6; u6 < 6; u6++) hcp[tri(u6, v6)] += Jc[u6][0] * G0 + Jc[u6][1] * G1 + Jc[u6][2] * G2;
              }
              if (t.chain >= 0) {
#pragma unroll
                for (int u4 = 0; u4 < NLK; u4++) {
                  const float* j3 = t.Jb[u4];
                  const float G0 = Wk[0] * j3[0] + Wk[1] * j3[1] + Wk[3] * j3[2], G1 = Wk[1] * j3[0] + Wk[2] * j3[1] + Wk[4] * j3[2], G2 = Wk[3] * j3[0] + Wk[4] * j3[1] + Wk[5] * j3[2];
#pragma unroll
                  for (int v4 = 0; v4 <= u4; v4++) atomicAdd(&S.Hbb[t.chain][tri(u4, v4)], t.Jb[v4][0] * G0 + t.Jb[v4][1] * G1 + t.Jb[v4][2] * G2);
#pragma unroll
                  for (int q6 = 0; q6 < 6; q6++) atomicAdd(&S.Hcb[t.chain][u4 * 6 + q6], Jc[q6][0] * G0 + Jc[q6][1] * G1 + Jc[q6][2] * G2);
                }
              }
            }
          }
        }
        {  // every lane ends up with the full cube Hessian block; lanes write their share to LDS
          float hsel0 = 0.f, hsel1 = 0.f;
#pragma unroll
          for (int e = 0; e < 21; e++) { float v = gsum(hcp[e]); if (e == l) hsel0 = v; if (e == l + 16) hsel1 = v; }
          if (act) {
            bool d0 = (l == 0 || l == 2 || l == 5 || l == 9 || l == 14), d1 = (l + 16 == 20);  // packed indices of the diagonal
            int k0 = l == 0 ? 0 : (l == 2 ? 1 : (l == 5 ? 2 : (l == 9 ? 3 : 4)));
            S.Hcc[l] = hsel0 + (d0 ? (k0 < 3 ? cmass : cI[k0 - 3]) : 0.f);
            if (l + 16 < 21) S.Hcc[l + 16] = hsel1 + (d1 ? cI[2] : 0.f);
          }
        }
        __syncthreads();
#ifdef JH_V2_ABLATE
        if (ab_phase == 1 && rep__ + 1 < nrep__) continue;
#endif
        // ---- (4) arrow factorisation: chain blocks first (each chain's 4 lanes redundantly), 6x6 Schur complement on the cube
        float L[10], Linv[4], Y[6][NLK], zb[NLK], xc6[6], pc4[NLK];
        {
          for (int k = 0; k < 10; k++) L[k] = S.Hbb[c][k];
          chol4(L, Linv);
          for (int q6 = 0; q6 < 6; q6++) { for (int j = 0; j < NLK; j++) Y[q6][j] = S.Hcb[c][j * 6 + q6]; fwd4(L, Linv, Y[q6]); }
          for (int j = 0; j < NLK; j++) zb[j] = -S.g[6 + 4 * c + j];
          fwd4(L, Linv, zb);
          if (act && l < 6) S.rhs6[l] = -gcl;
        }
        __syncthreads();
        if (act && s == 0) {
          for (int q6 = 0; q6 < 6; q6++) {
            for (int r6 = 0; r6 <= q6; r6++) { float v = 0.f; for (int j = 0; j < NLK; j++) v += Y[q6][j] * Y[r6][j]; atomicAdd(&S.Hcc[tri(q6, r6)], -v); }
            float v = 0.f; for (int j = 0; j < NLK; j++) v += Y[q6][j] * zb[j];
            atomicAdd(&S.rhs6[q6], -v);
          }
        }
        __syncthreads();
        {
          float Lc[21];
          for (int k = 0; k < 21; k++) Lc[k] = S.Hcc[k];
          for (int k = 0; k < 6; k++) xc6[k] = S.rhs6[k];
          float ci[6];
#pragma unroll
          for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
              float sv = Lc[tri(i, j)];
#pragma unroll
              for (int k = 0; k < j; k++) sv -= Lc[tri(i, k)] * Lc[tri(j, k)];
              if (i == j) { float rr = __frsqrt_rn(fmaxf(sv, 1e-30f)); ci[i] = rr; Lc[tri(i, i)] = sv * rr; }
              else Lc[tri(i, j)] = sv * ci[j];
            }
#pragma unroll
          for (int i = 0; i < 6; i++) { float sv = xc6[i];
#pragma unroll
            for (int k = 0; k < i; k++) sv -= Lc[tri(i, k)] * xc6[k];
            xc6[i] = sv * ci[i]; }
#pragma unroll
          for (int i = 5; i >= 0; i--) { float sv = xc6[i];
#pragma unroll
            for (int k = i + 1; k < 6; k++) sv -= Lc[tri(k, i)] * xc6[k];
            xc6[i] = sv * ci[i]; }
          for (int j = 0; j < NLK; j++) { float v = zb[j]; for (int q6 = 0; q6 < 6; q6++) v -= Y[q6][j] * xc6[q6]; pc4[j] = v; }
          bwd4(L, Linv, pc4);
        }
        float p_own = pc4[0];
#pragma unroll
        for (int j = 1; j < NLK; j++) if (j == s) p_own = pc4[j];
        if (act) { S.p[6 + l] = p_own; if (l < 6) S.p[l] = xc6[l]; }
        __syncthreads();
        V2_TICK(5)
#ifdef JH_V2_ABLATE
        if (ab_phase == 2 && rep__ + 1 < nrep__) continue;
#endif
        // ---- (5) exact line search along p
        float Mp_own = 0.f;
#pragma unroll
        for (int j = 0; j < NLK; j++) Mp_own += Mrow[j] * pc4[j];
        float xcl = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) if (k == l) xcl = xc6[k];
        float pMp = gsum(p_own * Mp_own + mck * xcl * xcl);
        float pMd = gsum(Mp_own * da_own + mck * xcl * dcl);
        float gp = gsum(g_own * p_own + gcl * xcl);
        if (act && !(gp < 0.f)) act = false;
        for (int k = 0; k < NSLOT; k++) if (sl[k].valid) slot_Jx(sl[k], xc6, S.p, sl[k].jp);
        dr.pf = p_own; dr.pl = dr.lims * p_own;
        float lo, hi, alpha, dlo, dhi; int side; bool lsact;
#if defined(JH_ENGINE_PROFILE) && defined(JH_V2_LSHIST)
        int ls_evals = 0;
#endif
        V2_REPEAT(3) {
        lo = 0.f; hi = -1.f; alpha = 1.f; dlo = gp; dhi = 0.f; side = 0; lsact = act;
        V2_OPAQUE(alpha);
        for (int ls = 0; ls < JH_V2_LSMAX && __any(lsact); ls++) {
          float d1, d2;
#if defined(JH_ENGINE_PROFILE) && defined(JH_V2_LSHIST)
          if (lsact) ls_evals++;
#endif
          lane_rows_dir(sl, dr, alpha, &d1, &d2);
          d1 = gsum(d1) + pMd + alpha * pMp; d2 = gsum(d2) + pMp;
          if (lsact) {
            if (fabsf(d1) <= lstol * fabsf(gp)) lsact = false;
            else {
              // bracket [lo, hi] with the slopes at both ends; Newton step from the current point, else Illinois secant, else bisection
              if (d1 < 0.f) { lo = alpha; dlo = d1; if (side < 0) dhi *= 0.5f; side = -1; } else { hi = alpha; dhi = d1; if (side > 0) dlo *= 0.5f; side = 1; }
              float nx = alpha - d1 * __frcp_rn(d2);
              if (hi < 0.f) { if (nx <= lo) nx = 2.f * alpha; }
              else if (nx <= lo || nx >= hi) {
#ifdef JH_V2_LSSECANT
                nx = (lo * dhi - hi * dlo) * __frcp_rn(dhi - dlo);
                if (!(nx > lo && nx < hi)) nx = 0.5f * (lo + hi);
#else
                nx = 0.5f * (lo + hi);
#endif
              }
              if (hi > 0.f && hi - lo <= JH_V2_LSBRACKET * hi) lsact = false;  // step length known to the bracket tolerance
              else alpha = nx;
            }
          }
        }
        }  // V2_REPEAT(3)
#if defined(JH_ENGINE_PROFILE) && defined(JH_V2_LSHIST)
        if (l == 0 && live && stats && act) atomicAdd(stats + 48 + (ls_evals < 15 ? ls_evals : 15), 1);
#endif
        // ---- (6) step
        if (act) {
          a_own += alpha * p_own; for (int k = 0; k < 6; k++) ac[k] += alpha * xc6[k];
          for (int k = 0; k < NSLOT; k++) if (sl[k].valid) for (int rw = 0; rw < 3; rw++) sl[k].jar[rw] += alpha * sl[k].jp[rw];
          dr.jf += alpha * dr.pf; dr.jl += alpha * dr.pl;
          if (-gp * alpha <= tol * tol * fmaxf(snorm, 1e-12f)) act = false;
        }
        __syncthreads();
        V2_TICK(6)
#ifdef JH_V2_ABLATE
        }
#endif
      }
      if (l == 0) { n_iters += iters_this; n_maxed += (iters_this >= cap); }
#ifdef JH_V2_ITERDUMP  // diagnostics: Newton iterations of every rollout and step, in the buffer of the candidate knots ((K nu) x N floats >= H x N)
      if (l == 0 && live && knots_out && hh < 64) knots_out[(size_t)hh * ldn + n] = (float)iters_this;
#endif
#ifdef JH_ENGINE_PROFILE
      if (l == 0 && live && stats) atomicAdd(stats + 24 + (iters_this < 23 ? iters_this : 23), 1);
#endif
    }
    // ================================================================ implicitfast integration: (M + h diag(d + kv)) qacc = fs + M (a - a0)
    {
      float da_own = a_own - a0_own, rhs_own = fs_own, x4[NLK], L[10];
#pragma unroll
      for (int j = 0; j < NLK; j++) rhs_own += Mrow[j] * quad_get(da_own, j);
#pragma unroll
      for (int j = 0; j < NLK; j++) x4[j] = quad_get(rhs_own, j);
      for (int k = 0; k < 10; k++) L[k] = Mc[k];
#pragma unroll
      for (int j = 0; j < NLK; j++) L[tri(j, j)] += h * quad_get(lc.damp + lc.kvd, j);
      float inv4[4]; chol4(L, inv4); fwd4(L, inv4, x4); bwd4(L, inv4, x4);
      float qacc = x4[0];
#pragma unroll
      for (int j = 1; j < NLK; j++) if (j == s) qacc = x4[j];
      qd = fmaf(h, qacc, qd); q = fmaf(h, qd, q); qws = a_own;
      for (int k = 0; k < 3; k++) {
        float al = (fsc[k] + cmass * (ac[k] - a0c[k])) / cmass, aw = (fsc[3 + k] + cI[k] * (ac[3 + k] - a0c[3 + k])) / cI[k];
        vc[k] = fmaf(h, al, vc[k]); vc[3 + k] = fmaf(h, aw, vc[3 + k]); wsc[k] = ac[k]; wsc[3 + k] = ac[3 + k];
      }
      for (int k = 0; k < 3; k++) qc[k] = fmaf(h, vc[k], qc[k]);
      float wn = sqrtf(vc[3] * vc[3] + vc[4] * vc[4] + vc[5] * vc[5]), ang = wn * h;
      if (ang > 0.f) {
        float sn, cs; sincosf(0.5f * ang, &sn, &cs); float kk = sn / wn;
        float dq[4] = {cs, vc[3] * kk, vc[4] * kk, vc[5] * kk}, *qq = qc + 3;
        float r0 = qq[0] * dq[0] - qq[1] * dq[1] - qq[2] * dq[2] - qq[3] * dq[3];
        float r1 = qq[0] * dq[1] + qq[1] * dq[0] + qq[2] * dq[3] - qq[3] * dq[2];
        float r2 = qq[0] * dq[2] - qq[1] * dq[3] + qq[2] * dq[0] + qq[3] * dq[1];
        float r3 = qq[0] * dq[3] + qq[1] * dq[2] - qq[2] * dq[1] + qq[3] * dq[0];
        qq[0] = r0; qq[1] = r1; qq[2] = r2; qq[3] = r3;
      }
      float nn = rsqrtf(qc[3] * qc[3] + qc[4] * qc[4] + qc[5] * qc[5] + qc[6] * qc[6]);
      qc[3] *= nn; qc[4] *= nn; qc[5] *= nn; qc[6] *= nn;
    }
    if (MATERIALIZE) {
      if (states && live) {
        float* o = states + ((size_t)nc * H + hh) * NX;
        o[7 + l] = q; o[NQ + 6 + l] = qd;
        if (l < 7) o[l] = qc[l];
        if (l < 6) o[NQ + l] = vc[l];
      }
    } else acc += leap_step_cost(sTp, qc);
    __syncthreads();
    V2_TICK(7)
  }
#ifdef JH_ENGINE_PROFILE
  if (lane == 0 && stats) for (int k = 0; k < 8; k++) atomicAdd((unsigned long long*)(stats + 4) + k, (unsigned long long)cyc[k]);
#endif
  if (!MATERIALIZE && live && l == 0) costs[n] = acc / (float)H;
  if (stats && live && l == 0) { if (n_maxed) atomicAdd(stats + 1, n_maxed); atomicAdd(stats + 2, n_iters); atomicAdd(stats + 3, H); }
  (void)n_overflow;
}

bool model_is_leap(const jh_model* m) { return m->kind == JH_TASK_LEAP_CUBE && m->nq == 23 && m->nv == 22 && m->nu == 16 && m->ns == 31 && m->h_i.size() > 13 && m->h_i[0] == 17 && m->h_i[1] == 4 && m->h_i[11] > 0 && m->h_i[5] <= MAXG && m->h_i[12] <= MAXLG; }

}  // namespace

int jh_engine2_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                            const float* lohi, const float* tp, int N, int n_offset, int H, int K, float* costs, float* knots_out, hipStream_t st) {
  if (!model_is_leap(m)) { jh_set_error("rollout_cost: the cooperative engine kernel is instantiated for leap_cube only"); return JH_ERR_UNSUPPORTED; }
  JH_REQUIRE(K <= 8, "rollout_cost: the cooperative leap kernel keeps at most 8 knots per actuator in registers (K=%d)", K);
  int grid = (N + RPW - 1) / RPW;
  hipLaunchKernelGGL(k_leap_v2<false>, dim3(grid), dim3(WAVE), 0, st, m->d_f, m->d_i, x0, 0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs,
                     knots_out, (const float*)nullptr, (float*)nullptr, (float*)nullptr, m->d_stats);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

int jh_engine2_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                           hipStream_t st) {
  if (!model_is_leap(m)) { jh_set_error("rollout_materialize: the cooperative engine kernel is instantiated for leap_cube only"); return JH_ERR_UNSUPPORTED; }
  int grid = (N + RPW - 1) / RPW;
  hipLaunchKernelGGL(k_leap_v2<true>, dim3(grid), dim3(WAVE), 0, st, m->d_f, m->d_i, x0, x0_batched, (const float*)nullptr, (const float*)nullptr, 0,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, N, 0, H, 0, (float*)nullptr, (float*)nullptr,
                     controls, states, sensors, m->d_stats);
  JH_HIP(hipGetLastError());
  return JH_OK;
}
