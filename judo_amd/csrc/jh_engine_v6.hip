// jh_engine_v6.hip -- fr3_pick cooperative kernel, second generation (gfx950): the step of jh_engine_v3.hip with a matrix-free contact Jacobian.
//
// jh_engine_v3.hip keeps the contact Jacobian (32 contacts x 3 x 15 floats) and every contact's force / weight block in LDS: 60 % of its 39 KB per wave, which
// together with its 500 registers pins it to one wave per SIMD, stalled four cycles out of five on dependent chains with nothing to switch to.  Here a contact
// slot keeps frame, point and the two body codes in registers of its owner lane and the Jacobian columns are recomputed from the joint axes / anchors in LDS
// wherever they are used (J x, J'f, J'WJ), as jh_engine_v5.hip does for the hand: a dof column of a contact is the world vector
//   cube linear dof q: +-e_q      cube angular dof q (body frame): +-(R e_q) x (pos - cube)      hinge j above the contact's link: +-axis_j x (pos - anchor_j)
//   finger slide: +-axis_f
// and J'WJ = col_x' (Fr' W Fr) col_y goes into a dof-space Hessian in LDS with float atomics, whose rows the lanes then factorise as before.
// Same lane roles as jh_engine_v3.hip (lane l < 6: free-body dof l; 6..14: arm dof l - 6; 15: right-hand side), same solver, same results to summation order.
#include <cstddef>
#include <type_traits>

#include "jh_coop.h"

using namespace jh_eng;
using namespace jh_coop;

#include <type_traits>
#include <utility>

namespace {
template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }  // f(integral_constant<int, 0>) ... f(<N - 1>): compile-time indices for DPP controls


constexpr int G = 16, RPW = 4, WAVE = 64;
constexpr int NA = 9, NCHAIN = 7, NVT = 15, NQ = 16, NU = 8, NS = 14, NX = 31, NMB = 10;
#ifndef JH_V6_NSBIG
#define JH_V6_NSBIG 6  // 96 general contacts per rollout (32 in LDS, 64 in the global row); 4 and 6 cost the common path the same (nothing: 11.14 / 11.17 ms on recorded inputs)
#endif
#ifndef JH_V6_BIGPROB
#define JH_V6_BIGPROB 0.999999  // how sure the compiler may be that a wave-step stays on the two-slot copy of the solver (block frequencies steer the placement of register spills)
#endif
// General contacts: NCP in the LDS pool (two slots per lane, the common case); a rollout with more -- a gripper pressed flat onto the table stacks 4-point manifolds of ten pad
// boxes: 40-60 contacts, the reference's SHIPPED 1 s horizon visits such states all the time -- writes the rest to a row of global memory, and its wave runs a second copy
// of the constraint rows + Newton solver with NSBIG slots per lane (the leap kernel's recipe, jh_engine_v5.hip `solve_step`).
constexpr int NCP = 32, NSL = NCP / G, NSBIG = JH_V6_NSBIG, NOVF = (NSBIG - NSL) * G, MAXHIT = 64, RAW_F = 8, JW = NVT * 3;  // MAXHIT: broad-phase survivors (candidate pairs) per rollout and step, 16 bits each
constexpr int MAXDT = 32;  // box pairs behind the distance sensors
#ifndef JH_V6_NFS
#define JH_V6_NFS 6
#endif
constexpr int FF_F = 5;
constexpr int NFS = JH_V6_NFS, NFF = NFS * G;  // finger-finger contacts: kept in registers of their owner lanes (6 per lane = 96 per rollout), never in the LDS Jacobian
#ifndef JH_V6_NOISE
#define JH_V6_NOISE 1.2e-7f  // one fp32 ulp (2^-23), relative: the resolution of the iterate
#endif
#ifndef JH_V6_OPAQUE
#define JH_V6_OPAQUE 1
#endif
#define OPAQUE6(x) asm volatile("" : "+v"(x))
// Measured and taken out of the source in round 4 (profiles/r04_leap_experiments.txt; the git history has it as JH_V6_RIGHTLOOK): a right-looking row Cholesky (one LDS write per
// lane and step, independent updates) with a column-oriented backward solve: 9.54 against 9.60 ms, within the noise -- the factorisation (19 % of the kernel,
// tools/diag/profile_fr3_phases.py) is bound by its fifteen LDS exchanges, not by the dependent chains of the left-looking form; with per-lane masks on the updates it was 13 %
// slower.  A two-column block form (eight exchanges instead of fifteen, both columns of L formed redundantly by every row): correct and 2.8 % slower (8.72 against
// 8.48 ms) -- the exchanges do not bound the factorisation either; what is left is its 15-step dependent structure on one wave.
// Round 5 (the git history has the LDS row form as JH_V6_DPPCHOL=0): that conclusion was wrong about the cause.  The factorisation in registers with DPP row broadcasts (step (4) of
// the Newton iteration: no LDS, no barrier, the same subtraction order and so the same bits) took the kernel from 8.26 to 7.68 ms on the recorded inputs: what the LDS
// variants had in common was the publish / wait / read-back round trip per pivot, whichever way the updates were arranged around it.
#ifndef JH_V6_NS1
#define JH_V6_NS1 1  // among the wave-steps without a finger-finger contact, those with at most 16 general contacts per rollout take a one-slot copy: 8.98 -> 8.79 ms
#endif
#ifndef JH_V6_FFSPLIT
#define JH_V6_FFSPLIT 1  // wave-steps without a finger-finger contact take a copy of rows + solver without the six finger-finger slots per lane (48 registers): 9.59 -> 9.37 ms, and 8.99 ms
                         // with -ffp-contract=on (jh_engine_v6.flags), under which the two copies also round alike (the leap kernel's note on JH_V5_HCSPLIT)
#endif
#ifndef JH_V6_LSCAP
#define JH_V6_LSCAP 12  // line-search evaluations per Newton iteration
#endif
#ifndef JH_V6_WPE
#define JH_V6_WPE 2   // waves per SIMD the register allocation aims at (19.9 KB of LDS per one-wave workgroup: eight workgroups per CU)
#endif
constexpr int LF = 8, RF = 9;       // moving-body indices of the two fingers (arm dofs 7 / 8 = lanes 13 / 14)

struct __attribute__((aligned(16))) RS6 {  // per-rollout shared state
  float xpos[NMB][3], xR[NMB][9], axw[NMB][3];  // moving bodies: 0 = free box, 1..9 = arm links / fingers
  float q[16], qd[16];                          // arm joint positions / velocities by arm index
  float sn[16], cs[16];                         // sin / cos of the hinge angles (each evaluated once, by its owner lane)
  float M[NA][NA];
  float vec[3][16];
  float pad16_;                                 // (the union below starts on a 16-byte boundary: its rows move as ds_read / ds_write_b128)
  union {                                       // the raw contact pool is dead once the slots are loaded; the Newton matrices then reuse its storage
    float raw[NCP][RAW_F];                      // pos3, normal3, dist, pair
    float H[16][16];                            // dof-space Hessian, row r = lane r's row, entries j <= r meaningful (round 6: full rows -- the pool's 256 floats hold them -- so
                                                // that a lane stores and fetches its row with four 16-byte accesses instead of fifteen exec-masked dwords each way)
  };
  float ffraw[NFF][FF_F];                       // finger-finger contacts between narrow phase and slots: normal3, dist, pair
  float g[16];                                  // gradient (own rows + contact forces by float atomics)
  float cq[8], cv[8];                           // the free body's position / quaternion and velocity: ONE copy per rollout here instead of one per lane in registers
  float y[16];                                  // sensordata of the forward pass
  float kn[NU][8];                              // spline knots per actuator (kept out of the register file: they are read once per step)
  unsigned short hits[MAXHIT];
  int ncon, nhit, nff;
};

static_assert(offsetof(RS6, H) % 16 == 0 && sizeof(RS6) % 16 == 0, "RS6: the Hessian rows are moved 16 bytes at a time");

struct Sink6 {  // contact sink of the narrow phase
  RS6* S; int* overflow; int pair; bool ff; float* ovf;  // ovf: this rollout's row of the global overflow pool (NOVF x RAW_F floats), or null
  __device__ __forceinline__ void push(const float* pos, const float* n, float dist) {
    float* e;
    if (ff) {  // finger against finger: the point itself is not needed (both sides slide along one line: no lever arm enters)
      int i = atomicAdd(&S->nff, 1);
      if (i >= NFF) { if (overflow) atomicAdd(overflow, 1); return; }
      float* o = S->ffraw[i];
      o[0] = n[0]; o[1] = n[1]; o[2] = n[2]; o[3] = dist; o[4] = __int_as_float(pair);
      return;
    } else {
      int i = atomicAdd(&S->ncon, 1);
      if (i < NCP) e = S->raw[i];
      else if (ovf && i < NCP + NOVF) e = ovf + (i - NCP) * RAW_F;
      else { if (overflow) atomicAdd(overflow, 1); return; }
    }
    e[0] = pos[0]; e[1] = pos[1]; e[2] = pos[2]; e[3] = n[0]; e[4] = n[1]; e[5] = n[2]; e[6] = dist; e[7] = __int_as_float(pair);
  }
};

// general contact slot (owner lane's registers): sides sa / sb = body of geom 1 / geom 2 (-1 static, 0 the free box, 1..7 arm links, 8 / 9 the fingers; -2 = empty
// slot), contact frame, point, pyramid constants, aref and the running jar = J a - aref, jp = J p
struct Slot6 { int sa, sb; float fr[9], pos[3], D, mu, aref[3], jar[3], jp[3]; };
// finger-finger contact: the only non-zero Jacobian columns are the two finger slides, and because the two slide axes are antiparallel (checked at create:
// jh_model_is_fr3) those two columns are EQUAL -- side A on one finger, side B on the other: s13 a13 = s14 a14.  The contact therefore sees the arm only through
// the scalar a13 + a14 (the closing acceleration of the gripper): jar = Jf (a13 + a14) - aref and jp = Jf (p13 + p14) are recomputed where they are used, and
// a slot is 8 registers instead of 17.  That is what lets 6 slots per lane -- 96 pad-against-pad contacts per rollout, more than the 82 a closed empty gripper
// produces (fr3_components/fr3.xml:84-116) -- fit where 3 did.  D = 0 marks an empty slot (every term of a pyramid row carries D).
struct SlotF { float D, mu, aref[3], Jf[3]; };
struct DofRows6 { float fl, fD, fR, faref, lims, laref, lD, jf, jl, pf, pl; };

// four one-sided rows x_k = jar_n +- mu jar_t1, jar_n +- mu jar_t2: slope and curvature along jp
__device__ __forceinline__ void pyramid_dir(const float* jar, const float* jp, float D, float mu, float* d1, float* d2) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float sg = (k & 1) ? -mu : mu;
    const float x = jar[0] + sg * (k < 2 ? jar[1] : jar[2]), xp = jp[0] + sg * (k < 2 ? jp[1] : jp[2]);
    if (x < 0.f) { *d1 += D * x * xp; *d2 += D * xp * xp; }
  }
}

// A general contact has at most one side on the free box and at most one on the arm (arm against arm is the finger-finger case, in slots of its own): `csign` = +1 / -1 / 0 for
// the box on side B / side A / neither, `ab` = the arm-side body (0: none) and `asign` its side's sign.
struct Sides6 { float csign, asign; int ab; };
__device__ __forceinline__ Sides6 slot_sides(int sa, int sb) {
  Sides6 z;
  z.csign = (sb == 0 ? 1.f : 0.f) - (sa == 0 ? 1.f : 0.f);
  z.ab = sa >= 1 ? sa : (sb >= 1 ? sb : 0);
  z.asign = sb >= 1 ? 1.f : -1.f;
  return z;
}
// the world columns of the arm's eight dofs for a point carried by the arm: hinge j -> axis_j x (pos - anchor_j), entry 7 -> the slide axis of finger `fb` (8 or 9).  ALL of
// them, whatever the depth of the link (round 6: the per-joint form `if (j < b) { load; cross; }` put every joint's LDS loads under an exec mask of their own -- nine regions and
// as many dependent round trips per call, 528 branches in the kernel's listing; the columns a shallower link does not use are cheaper than the regions).
__device__ __forceinline__ void arm_cols(const RS6& S, const float* pos, int fb, float (*c3)[3]) {
#pragma unroll
  for (int j = 0; j < NCHAIN; j++) {
    const float rb[3] = {pos[0] - S.xpos[1 + j][0], pos[1] - S.xpos[1 + j][1], pos[2] - S.xpos[1 + j][2]};
    cross3(c3[j], S.axw[1 + j], rb);
  }
  c3[NCHAIN][0] = S.axw[fb][0]; c3[NCHAIN][1] = S.axw[fb][1]; c3[NCHAIN][2] = S.axw[fb][2];
}
// J x of one slot (contact frame): relative point velocity, side B minus side A, for the generalised velocity (xc = free-body part: linear world, angular body frame; xa = arm
// part by arm index, LDS or registers).  The box's part is computed whatever the slot holds and scaled by csign (an empty slot: sides -2, everything zero); the arm's under one test.
template <class XA>
__device__ __forceinline__ void slot_Jx(const Slot6& t, const RS6& S, const float* xc, const XA& xa, float* out) {
  const Sides6 z = slot_sides(t.sa, t.sb);
  float w[3];
  {
    float wa[3], rc[3] = {t.pos[0] - S.xpos[0][0], t.pos[1] - S.xpos[0][1], t.pos[2] - S.xpos[0][2]}, wx[3];
    mulMV(wa, S.xR[0], xc + 3); cross3(wx, wa, rc);
    w[0] = z.csign * (xc[0] + wx[0]); w[1] = z.csign * (xc[1] + wx[1]); w[2] = z.csign * (xc[2] + wx[2]);
  }
  if (z.ab >= 1) {
    const int fb = z.ab > NCHAIN ? z.ab : LF;
    float c3[NCHAIN + 1][3]; arm_cols(S, t.pos, fb, c3);
#pragma unroll
    for (int j = 0; j < NCHAIN; j++) {
      const float xj = j < z.ab ? z.asign * xa[j] : 0.f;
      w[0] = fmaf(c3[j][0], xj, w[0]); w[1] = fmaf(c3[j][1], xj, w[1]); w[2] = fmaf(c3[j][2], xj, w[2]);
    }
    const float xs7 = xa[NCHAIN], xs8 = xa[NCHAIN + 1];
    const float xs = z.ab > NCHAIN ? z.asign * (z.ab == LF ? xs7 : xs8) : 0.f;
    w[0] = fmaf(c3[NCHAIN][0], xs, w[0]); w[1] = fmaf(c3[NCHAIN][1], xs, w[1]); w[2] = fmaf(c3[NCHAIN][2], xs, w[2]);
  }
  out[0] = dot3(t.fr, w); out[1] = dot3(t.fr + 3, w); out[2] = dot3(t.fr + 6, w);
}
// -J'F of one slot (F = world force on side B): the arm's rows as float atomics into the gradient, the free box's six entries into gcp
// (the free box's six entries: every contact of the rollout that touches the box lands on the same six addresses, and same-address LDS atomics serialise -- they are summed
// over the rollout's lanes instead: 9.77 -> 9.57 ms on recorded inputs; the leap kernel's cube block taught this: jh_engine_v5.hip, the note on the Newton iteration's formulation)
__device__ __forceinline__ void slot_force(RS6& S, const Slot6& t, const float* Fw, float* gcp) {
  const Sides6 z = slot_sides(t.sa, t.sb);
  {
    float rc[3] = {t.pos[0] - S.xpos[0][0], t.pos[1] - S.xpos[0][1], t.pos[2] - S.xpos[0][2]}, tq[3], tb[3];
    cross3(tq, rc, Fw); mulMTV(tb, S.xR[0], tq);
    gcp[0] -= z.csign * Fw[0]; gcp[1] -= z.csign * Fw[1]; gcp[2] -= z.csign * Fw[2]; gcp[3] -= z.csign * tb[0]; gcp[4] -= z.csign * tb[1]; gcp[5] -= z.csign * tb[2];
  }
  if (z.ab >= 1) {
    const int fb = z.ab > NCHAIN ? z.ab : LF;
    float c3[NCHAIN + 1][3]; arm_cols(S, t.pos, fb, c3);
    float v[NCHAIN + 1];
#pragma unroll
    for (int j = 0; j <= NCHAIN; j++) v[j] = -z.asign * dot3(c3[j], Fw);
#pragma unroll
    for (int j = 0; j < NCHAIN; j++) if (j < z.ab) atomicAdd(&S.g[6 + j], v[j]);
    if (z.ab > NCHAIN) atomicAdd(&S.g[6 + z.ab - 1], v[NCHAIN]);
  }
}
// J'WJ of one slot into the dof-space Hessian.  Every dof column of the contact is a world 3-vector col_x with J[w][x] = fr_w . col_x, so the contribution
// to H[x][y] is col_x' A col_y with A = Fr' W Fr (symmetric 3 x 3).  A general contact has at most one side on the free box and at most one on the arm
// (arm against arm is the finger-finger case, handled in its own slots): up to 6 + 8 columns.
__device__ __forceinline__ void slot_assemble(RS6& S, const Slot6& t, const float* Wk, float* hcp) {
  float A[6];
  {
    float T0[3], T1[3], T2[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      T0[q] = Wk[0] * t.fr[q] + Wk[1] * t.fr[3 + q] + Wk[3] * t.fr[6 + q];
      T1[q] = Wk[1] * t.fr[q] + Wk[2] * t.fr[3 + q] + Wk[4] * t.fr[6 + q];
      T2[q] = Wk[3] * t.fr[q] + Wk[4] * t.fr[3 + q] + Wk[5] * t.fr[6 + q];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) A[tri(i, j)] = t.fr[i] * T0[j] + t.fr[3 + i] * T1[j] + t.fr[6 + i] * T2[j];
  }
  auto Amul = [&](const float* v, float* y) __attribute__((always_inline)) {
    y[0] = A[0] * v[0] + A[1] * v[1] + A[3] * v[2]; y[1] = A[1] * v[0] + A[2] * v[1] + A[4] * v[2]; y[2] = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  };
  const Sides6 z = slot_sides(t.sa, t.sb);
  const bool cube = z.csign != 0.f;
  const int ab = z.ab;  // the arm-side body, 0 = none
  // the signs of the two sides cancel in every product of two columns of the same body and give -1 between the free box and the arm
  float c3[3][3];
  if (cube) {
    const float rc[3] = {t.pos[0] - S.xpos[0][0], t.pos[1] - S.xpos[0][1], t.pos[2] - S.xpos[0][2]};
#pragma unroll
    for (int q = 0; q < 3; q++) { float ea[3]; col3(ea, S.xR[0], q); cross3(c3[q], ea, rc); }
#pragma unroll
    for (int e = 0; e < 6; e++) hcp[e] += A[e];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      float z3[3]; Amul(c3[q], z3);
#pragma unroll
      for (int r2 = 0; r2 < 3; r2++) hcp[tri(3 + q, r2)] += z3[r2];
#pragma unroll
      for (int r2 = 0; r2 <= q; r2++) hcp[tri(3 + q, 3 + r2)] += dot3(c3[r2], z3);
    }
  }
  if (ab >= 1) {
    // (round 6: the arm's columns once per contact -- arm_cols -- instead of once per ENTRY: the two rolled loops over (u, v) recomputed column v inside every pair, each behind
    // its own loads, and tested every pair; a valid column u has all columns v <= u valid, so one test per column decides its whole row)
    const int fb = ab > NCHAIN ? ab : LF, dsl = 6 + ab - 1;
    float cu[NCHAIN + 1][3]; arm_cols(S, t.pos, fb, cu);
#pragma unroll
    for (int u = 0; u <= NCHAIN; u++) {  // arm columns 0..6 = hinges, 7 = the finger's slide
      if (!(u < NCHAIN ? u < ab : ab > NCHAIN)) continue;
      float y[3]; Amul(cu[u], y);
      const int du = u < NCHAIN ? 6 + u : dsl;
#pragma unroll
      for (int v = 0; v <= u; v++) atomicAdd(&S.H[du][v < NCHAIN ? 6 + v : dsl], dot3(cu[v], y));
      if (cube) {
#pragma unroll
        for (int q = 0; q < 3; q++) atomicAdd(&S.H[du][q], -y[q]);
#pragma unroll
        for (int q = 0; q < 3; q++) atomicAdd(&S.H[du][3 + q], -dot3(c3[q], y));
      }
    }
  }
}

template <int NS, int NF>
__device__ __forceinline__ float lane_rows_cost(const Slot6* sl, const SlotF* sf, float sff, const DofRows6& dr, bool eq_lane, float eD, float ejar) {
  float cs = 0.f;
#pragma unroll
  for (int k = 0; k < NF; k++) if (sf[k].D > 0.f) {  // sff = a13 + a14 of the point the cost is taken at
    const float jar[3] = {fmaf(sf[k].Jf[0], sff, -sf[k].aref[0]), fmaf(sf[k].Jf[1], sff, -sf[k].aref[1]), fmaf(sf[k].Jf[2], sff, -sf[k].aref[2])};
    float f[3], W[6]; cs += pyramid_eval(jar, sf[k].D, sf[k].mu, f, W);
  }
#pragma unroll
  for (int k = 0; k < NS; k++) if (sl[k].sa > -2) { float f[3], W[6]; cs += pyramid_eval(sl[k].jar, sl[k].D, sl[k].mu, f, W); }
  if (dr.fl > 0.f) {
    float x = dr.jf, fl = dr.fl, lim = dr.fR * fl;
    if (x <= -lim) cs += -0.5f * dr.fR * fl * fl - fl * x; else if (x >= lim) cs += -0.5f * dr.fR * fl * fl + fl * x; else cs += 0.5f * dr.fD * x * x;
  }
  if (dr.lims != 0.f && dr.jl < 0.f) cs += 0.5f * dr.lD * dr.jl * dr.jl;
  if (eq_lane) cs += 0.5f * eD * ejar * ejar;
  return cs;
}

template <int NS, int NF>
__device__ __forceinline__ void lane_rows_dir(const Slot6* sl, const SlotF* sf, float sff, float spf, const DofRows6& dr, bool eq_lane, float eD, float ejar, float ejp, float al, float* d1, float* d2) {
  float g1 = 0.f, g2 = 0.f;
  const float sal = fmaf(al, spf, sff);  // a13 + a14 at the trial point
#pragma unroll
  for (int k = 0; k < NF; k++) if (sf[k].D > 0.f) {
    const float jp[3] = {sf[k].Jf[0] * spf, sf[k].Jf[1] * spf, sf[k].Jf[2] * spf};
    const float jar[3] = {fmaf(sf[k].Jf[0], sal, -sf[k].aref[0]), fmaf(sf[k].Jf[1], sal, -sf[k].aref[1]), fmaf(sf[k].Jf[2], sal, -sf[k].aref[2])};
    pyramid_dir(jar, jp, sf[k].D, sf[k].mu, &g1, &g2);
  }
#pragma unroll
  for (int k = 0; k < NS; k++) if (sl[k].sa > -2) {
    const float* jp = sl[k].jp;
    float jar[3] = {fmaf(al, jp[0], sl[k].jar[0]), fmaf(al, jp[1], sl[k].jar[1]), fmaf(al, jp[2], sl[k].jar[2])};
    pyramid_dir(jar, jp, sl[k].D, sl[k].mu, &g1, &g2);
  }
  if (dr.fl > 0.f) {
    float jp = dr.pf, x = fmaf(al, jp, dr.jf), fl = dr.fl, lim = dr.fR * fl;
    if (x <= -lim) g1 -= fl * jp; else if (x >= lim) g1 += fl * jp; else { g1 += dr.fD * x * jp; g2 += dr.fD * jp * jp; }
  }
  if (dr.lims != 0.f) { float jp = dr.pl, x = fmaf(al, jp, dr.jl); if (x < 0.f) { g1 += dr.lD * x * jp; g2 += dr.lD * jp * jp; } }
  if (eq_lane) { float x = fmaf(al, ejp, ejar); g1 += eD * x * ejp; g2 += eD * ejp * ejp; }
  *d1 = g1; *d2 = g2;
}

// dense Cholesky + solve of an n x n system held in registers (packed lower L, reciprocal diagonal), redundantly per lane
template <int N_>
__device__ __forceinline__ void chol_solve(float* L, float* x) {
  float inv[N_];
#pragma unroll
  for (int i = 0; i < N_; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float s = L[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[tri(i, k)] * L[tri(j, k)];
      if (i == j) { float rr = __frsqrt_rn(fmaxf(s, 1e-30f)); inv[i] = rr; L[tri(i, i)] = s * rr; }
      else L[tri(i, j)] = s * inv[j];
    }
#pragma unroll
  for (int i = 0; i < N_; i++) { float s = x[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k];
    x[i] = s * inv[i]; }
#pragma unroll
  for (int i = N_ - 1; i >= 0; i--) { float s = x[i];
#pragma unroll
    for (int k = i + 1; k < N_; k++) s -= L[tri(k, i)] * x[k];
    x[i] = s * inv[i]; }
}

// The arm's 9 x 9 system with its rows spread over the arm's lanes (lane 6 + a holds row a, entries b <= a; lane 15 the right-hand side as a tenth row): right-looking
// Cholesky and both triangular solves with DPP row broadcasts, every entry receiving its subtractions in the order of chol_solve<9> (the same bits); every lane gets x.
// Replaces the factorisation every lane did redundantly on its own copy of the 45 entries.
__device__ __forceinline__ void arm_chol_solve(float* R, float* x, int l) {
  static_for<NA>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const float rinv = __frsqrt_rn(fmaxf(row_bcast<6 + k>(R[k]), 1e-30f));
    const float lk = R[k] * rinv;
    R[k] = l == 6 + k ? rinv : lk;
    static_for<NA - 1 - k>([&](auto jc) {
      constexpr int j = k + 1 + decltype(jc)::value;
      R[j] = __builtin_fmaf(-lk, row_bcast<6 + j>(lk), R[j]);
    });
  });
  static_for<NA>([&](auto kc) {
    constexpr int k = NA - 1 - decltype(kc)::value;
    float s = row_bcast<15>(R[k]);
    static_for<NA - 1 - k>([&](auto jc) {
      constexpr int j = k + 1 + decltype(jc)::value;
      s = __builtin_fmaf(-row_bcast<6 + j>(R[k]), x[j], s);
    });
    x[k] = s * row_bcast<6 + k>(R[k]);
  });
}

__device__ __forceinline__ void rodrigues(float* Rq, const float* al, float sn, float cs) {
  float t = 1.f - cs, x = al[0], y = al[1], z = al[2];
  Rq[0] = t * x * x + cs; Rq[1] = t * x * y - sn * z; Rq[2] = t * x * z + sn * y;
  Rq[3] = t * x * y + sn * z; Rq[4] = t * y * y + cs; Rq[5] = t * y * z - sn * x;
  Rq[6] = t * x * z - sn * y; Rq[7] = t * y * z + sn * x; Rq[8] = t * z * z + cs;
}

// world pose of collision geom g (all fr3 collision geoms are boxes); body -1 = static (pose stored in world coordinates)
__device__ __forceinline__ void geom_pose3(const RS6& S, const float* gf, int body, float* gp, float* gR, bool want_R) {
  if (body < 0) { for (int k = 0; k < 3; k++) gp[k] = gf[GF_POS + k]; if (want_R) for (int k = 0; k < 9; k++) gR[k] = gf[GF_R + k]; }
  else {
    float bR[9]; for (int k = 0; k < 9; k++) bR[k] = S.xR[body][k];
    mulMV(gp, bR, gf + GF_POS); for (int k = 0; k < 3; k++) gp[k] += S.xpos[body][k];
    if (want_R) mulMM(gR, bR, gf + GF_R);
  }
}

template <bool MATERIALIZE>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(JH_V6_WPE, JH_V6_WPE))) void k_fr3_v6(const float* __restrict__ gF, const int* __restrict__ gI, const float* __restrict__ x0, int x0_batched,
                                                    const float* __restrict__ nominal, const float* __restrict__ noise, int ldn,
                                                    const float* __restrict__ sigma, const float* __restrict__ W, const float* __restrict__ lohi,
                                                    const float* __restrict__ tp, int phase, int N, int n_offset, int H, int K,
                                                    float* __restrict__ costs, float* __restrict__ knots_out, const float* __restrict__ controls,
                                                    float* __restrict__ states, float* __restrict__ sensors, int* __restrict__ stats, int dshift, float* __restrict__ trace, float* __restrict__ ovf_all) {
  __shared__ RS6 sRS[RPW];
  __shared__ int sDT[MAXDT][3];  // distance-sensor tasks: sensordata address, geom a, geom b
  __shared__ int sNDT, sDadr[8], sDall[8];  // sDadr: sensordata address of distance sensor s, or -1 when this mode does not evaluate it; sDall: its address either way
  __shared__ float sTp[24];
  const int lane = threadIdx.x, l = lane & 15, r = lane >> 4;
  RS6& S = sRS[r];
  EngineModel m; m.init(gF, gI);
  const bool isarm = l >= 6 && l < 15, iscube = l < 6, hasdof = l < 15;
  const int ai = isarm ? l - 6 : 0;        // arm dof index 0..8 (7, 8 = finger slides)
  const bool isfinger = isarm && ai >= NCHAIN;
  // (latency mode, jh_internal.h: 1 << dshift rows of the wave compute the same rollout, the first of them writes)
  const int n = (blockIdx.x << (2 - dshift)) + (r >> dshift);
  const bool live = n < N && (r & ((1 << dshift) - 1)) == 0;
  const int nc = n < N ? n : N - 1;
  if (lane == 0) {  // flatten the distance sensors into (address, geom, geom) tasks
    int nt = 0;
    for (int s = 0; s < m.NGS; s++) {
      const int* si = gI + m.oSensG + s * 4;
      if (si[0] != 4) continue;
      // fused mode: the sensor array is never written out, and FR3Pick.reward (judo/tasks/fr3_pick.py:225-311, fr3_step_cost) reads the finger-table distances (2, 3) and, in
      // the PLACE phase only, the object-table distance (4); the finger-object distances (0, 1) are not part of the cost: 10 of the 21 box pairs, and with them the second
      // round of 15-axis separations per step.  The drop-in (materialise) mode evaluates all of them.
      // Which ones and how is decided by fr3_cost_reads_distance / fr3_cost_reads_sign_only, next to the cost itself (jh_engine_common.h).
      if (si[1] < 8) sDall[si[1]] = si[3];
      if (!MATERIALIZE && !fr3_cost_reads_distance(si[3], phase)) { if (si[1] < 8) sDadr[si[1]] = -1; continue; }
      if (si[1] < 8) sDadr[si[1]] = si[3];
      const int* di = gI + m.oDistI + si[1] * 4;
      for (int a = 0; a < di[1]; a++) for (int b = 0; b < di[3]; b++) if (nt < MAXDT) { sDT[nt][0] = si[3]; sDT[nt][1] = gI[m.oGlist + di[0] + a]; sDT[nt][2] = gI[m.oGlist + di[2] + b]; nt++; }
    }
    sNDT = nt;
  }
  if (!MATERIALIZE && lane < 22) sTp[lane] = tp[lane];
  // ---- per-lane constants: own dof rows, own actuator
  // (the 25 constants of the own dof / actuator are NOT held in registers across the step loop: they are read from the model image where they are used --
  // a few cached vector loads per step -- through an index the compiler cannot see through, see the top of the step loop)
  int dofc = hasdof ? l : 14;
  const float en = hasdof ? 1.f : 0.f;
  const bool hasact = l >= 6 && l < 6 + NU;
  const float h = gF[HF_DT], impratio = gF[HF_IMPRATIO], tol = gF[HF_TOL], lstol = gF[HF_LSTOL]; const int cap = (int)gF[HF_MAXITER];
  const float grav[3] = {gF[HF_GRAV], gF[HF_GRAV + 1], gF[HF_GRAV + 2]};
  const float cmass = gF[HF_CMASS], cI[3] = {gF[HF_CINERTIA], gF[HF_CINERTIA + 1], gF[HF_CINERTIA + 2]};
  // joint equality (finger coupling), rows live in lanes 13 / 14 (dofs ei0 / ei1)
  const bool has_eq = m.NEQ > 0;
  const float* ef = gF + m.oEqF;
  const float e_a0 = has_eq ? ef[EF_A0] : 0.f, e_a1 = has_eq ? ef[EF_A1] : 0.f, e_K = ef[EF_K], e_B = ef[EF_B], e_invw = ef[EF_INVW];
  float e_si[5]; for (int k = 0; k < 5; k++) e_si[k] = ef[EF_SOLIMP + k];
  const bool eq_lane = has_eq && l == 13;  // the lane that accounts for the row's cost / line-search terms
  // ---- state: replicated free body + own joint
  float q = 0.f, qd = 0.f, qws = 0.f;
  {
    const float* xi = x0 + ((MATERIALIZE && x0_batched) ? (size_t)nc * NX : 0);
    if (l < 7) S.cq[l] = xi[l];
    if (l < 6) S.cv[l] = xi[NQ + l];
    if (isarm) { q = xi[7 + ai]; qd = xi[NQ + 6 + ai]; }
  }
  // ---- own actuator's spline knots (fused mode)
  if (!MATERIALIZE) {
    if (hasact) {
      const int u = l - 6;
      for (int k = 0; k < K && k < 8; k++) {
        int i = k * NU + u;
        float v = nominal[i];
        if (n_offset + nc != 0) v = fmaf(sigma[i], noise[(size_t)i * ldn + nc], v);
        v = jh_clampf(v, lohi[u], lohi[NU + u]);
        S.kn[u][k] = v;
        if (knots_out && live) knots_out[(size_t)i * ldn + n] = v;
      }
    }
  }
  int n_iters = 0, n_maxed = 0;
#ifdef JH_V6_EXITSTATS
  int n_x[5] = {0, 0, 0, 0, 0};  // solver exits: gradient / not a descent direction / expected decrease / iteration cap / gradient at its fp32 rounding floor
#endif
  float acc = 0.f;
  __syncthreads();
  const int ndt = sNDT;
  // the distance sensors this mode does not evaluate hold their cutoff ("nothing within range"), not whatever the LDS held: fr3_step_cost receives all of y[]
  if (l < m.NDIST && l < 8 && sDadr[l] < 0) S.y[sDall[l]] = gF[m.oDistF + l];

#ifdef JH_V6_PHASES
  long long ph_t = __builtin_readcyclecounter(), ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PH6(i) { const long long ph_n = __builtin_readcyclecounter(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; }
#else
#define PH6(i)
#endif
  for (int hh = 0; hh < H; hh++) {
    OPAQUE6(dofc);
    const float* df = gF + m.oDofF + dofc * DOF_F;
    const float* af = gF + m.oActF + (hasact ? dofc - 6 : 0) * ACT_F;
    // ================================================================ controls
    float u = 0.f;
    if (hasact) {
      if (MATERIALIZE) u = controls[((size_t)nc * H + hh) * NU + (l - 6)];
      else for (int k = 0; k < K && k < 8; k++) u = fmaf(W[hh * K + k], S.kn[l - 6][k], u);
    }
    if (isarm) { S.q[ai] = q; S.qd[ai] = qd; float sn_, cs_; sincosf(q, &sn_, &cs_); S.sn[ai] = sn_; S.cs[ai] = cs_; }
    if (l == 0) { S.ncon = 0; S.nhit = 0; S.nff = 0; }
    __syncthreads();
    // ================================================================ kinematics: every lane walks the 7-hinge chain (uniform records -> scalar loads)
    float Rown[9], pown[3], axown[3];  // joint axes / anchors of the whole chain go to LDS (S.axw, S.xpos): the dynamics below reads them from there
    {
      float qc[7], Rc[9];
      for (int k = 0; k < 7; k++) qc[k] = S.cq[k];
      float nn = rsqrtf(qc[3] * qc[3] + qc[4] * qc[4] + qc[5] * qc[5] + qc[6] * qc[6]);
      qc[3] *= nn; qc[4] *= nn; qc[5] *= nn; qc[6] *= nn;
      quat2mat(Rc, qc + 3);
      float P[3] = {0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      for (int k = 0; k < 9; k++) Rown[k] = R[k];
      for (int k = 0; k < 3; k++) { pown[k] = 0.f; axown[k] = 0.f; }
#pragma unroll
      for (int j = 0; j < NCHAIN; j++) {
        const float* bf = gF + m.oBodyF + (1 + j) * BODY_F;
        float P2[3], R0[9];
        if (j == 0) { for (int k = 0; k < 3; k++) P2[k] = bf[BF_LPOS + k]; for (int k = 0; k < 9; k++) R0[k] = bf[BF_LR + k]; }
        else { mulMV(P2, R, bf + BF_LPOS); for (int k = 0; k < 3; k++) P2[k] += P[k]; mulMM(R0, R, bf + BF_LR); }
        float axj[3]; mulMV(axj, R0, bf + BF_AXIS);
        for (int k = 0; k < 3; k++) P[k] = P2[k];
        float Rq[9]; rodrigues(Rq, bf + BF_AXIS, S.sn[j], S.cs[j]);
        mulMM(R, R0, Rq);
        if (j == ai) { for (int k = 0; k < 3; k++) { pown[k] = P2[k]; axown[k] = axj[k]; } for (int k = 0; k < 9; k++) Rown[k] = R[k]; }
      }
      {  // finger slide hanging off link 7 (lanes that are not a finger compute finger 7 and ignore it)
        const int fi = isfinger ? ai : NCHAIN;
        const float* bf = gF + m.oBodyF + (1 + fi) * BODY_F;
        float P2[3], R0[9], lp[3] = {bf[BF_LPOS], bf[BF_LPOS + 1], bf[BF_LPOS + 2]}, lr[9], la[3] = {bf[BF_AXIS], bf[BF_AXIS + 1], bf[BF_AXIS + 2]};
        for (int k = 0; k < 9; k++) lr[k] = bf[BF_LR + k];
        mulMV(P2, R, lp); mulMM(R0, R, lr);
        float ax7[3]; mulMV(ax7, R0, la);
        const float qf = S.q[fi];
        for (int k = 0; k < 3; k++) P2[k] += P[k] + ax7[k] * qf;
        if (isfinger) { for (int k = 0; k < 3; k++) { pown[k] = P2[k]; axown[k] = ax7[k]; } for (int k = 0; k < 9; k++) Rown[k] = R0[k]; }
      }
      if (isarm) {
        for (int k = 0; k < 3; k++) { S.xpos[1 + ai][k] = pown[k]; S.axw[1 + ai][k] = axown[k]; }
        for (int k = 0; k < 9; k++) S.xR[1 + ai][k] = Rown[k];
      }
      if (l == 0) { for (int k = 0; k < 3; k++) S.xpos[0][k] = qc[k]; for (int k = 0; k < 9; k++) S.xR[0][k] = Rc[k]; }
    }
    __syncthreads();
    PH6(0)
    // ================================================================ sensors of this forward pass
    {
      if (l < m.NGS) {
        const int* si = gI + m.oSensG + l * 4; const int st = si[0], obj = si[1], adr = si[3];
        if (st == 0 || st == 5) {
          int b = gI[m.oFrameI + obj]; const float* ff = gF + m.oFrameF + obj * FRAME_F;
          float bR[9]; for (int k = 0; k < 9; k++) bR[k] = S.xR[b][k];
          if (st == 0) { float lp[3] = {ff[0], ff[1], ff[2]}, p3[3]; mulMV(p3, bR, lp); for (int k = 0; k < 3; k++) S.y[adr + k] = p3[k] + S.xpos[b][k]; }
          else { float zl[3] = {ff[3 + 2], ff[3 + 5], ff[3 + 8]}, z[3]; mulMV(z, bR, zl); for (int k = 0; k < 3; k++) S.y[adr + k] = z[k]; }
        } else if (st == 1) { for (int k = 0; k < 3; k++) S.y[adr + k] = S.xpos[obj][k]; }
        else if (st == 3) { for (int k = 0; k < 3; k++) S.y[adr + k] = S.xR[obj][3 * k + 2]; }
      }
      float dmin[2] = {3.0e38f, 3.0e38f}; int dadr[2] = {-1, -1};
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int t = l + 16 * k;
        if (t < ndt) {
          const int ga = sDT[t][1], gb = sDT[t][2];
          const float* fa = gF + m.oAGF + ga * GEOM_F; const float* fb = gF + m.oAGF + gb * GEOM_F;
          float pa[3], Ra[9], pb[3], Rb[9], ha[3] = {fa[GF_SIZE], fa[GF_SIZE + 1], fa[GF_SIZE + 2]}, hb[3] = {fb[GF_SIZE], fb[GF_SIZE + 1], fb[GF_SIZE + 2]};
          geom_pose3(S, fa, gI[m.oAGI + ga * GEOM_I], pa, Ra, true); geom_pose3(S, fb, gI[m.oAGI + gb * GEOM_I], pb, Rb, true);
          dadr[k] = sDT[t][0];
          // (fused mode: the cost asks of the finger-table sensors only whether they read <= 0 -- the sign, decided at the first separating axis)
          if (!MATERIALIZE && fr3_cost_reads_sign_only(dadr[k])) dmin[k] = box_box_touching(pa, Ra, ha, pb, Rb, hb) ? -1.f : 1.f;
          else dmin[k] = box_box_distance(pa, Ra, ha, pb, Rb, hb);
        }
      }
      for (int s = 0; s < m.NDIST; s++) {  // per sensor: minimum over its box pairs, clipped at the cutoff
        const int adr = sDadr[s < 8 ? s : 7];
        if (adr < 0) continue;  // (not evaluated in this mode: above)
        float v = fminf(dadr[0] == adr ? dmin[0] : 3.0e38f, dadr[1] == adr ? dmin[1] : 3.0e38f);
        v = fminf(gmin(v), gF[m.oDistF + s]);
        if (l == 0 && adr >= 0) S.y[adr] = v;
      }
      __syncthreads();
      if (MATERIALIZE && sensors && live && l < NS) sensors[((size_t)nc * H + hh) * NS + l] = S.y[l];
      if (!MATERIALIZE && trace && live && l < 6) trace[((size_t)n * H + hh) * 6 + l] = S.y[8 + l];  // trace_object, trace_grasp_site of every rollout (jh_rollout_cost_traced)
    }
    PH6(1)
    // ================================================================ arm dynamics: 9x9 inertia and bias from per-link contributions
    float a0_own, fs_own, Md_own, fsc[6], a0c[6];  // (the own row of the arm inertia is read from S.M where it is used: nine registers less to carry)
    {
      // own link (lanes that own no arm dof contribute a massless link 1)
      const float* bo = gF + m.oBodyF + (1 + ai) * BODY_F;
      const float mass = isarm ? bo[BF_MASS] : 0.f;
      float di[3] = {bo[BF_INERTIA] * (isarm ? 1.f : 0.f), bo[BF_INERTIA + 1] * (isarm ? 1.f : 0.f), bo[BF_INERTIA + 2] * (isarm ? 1.f : 0.f)};
      float lip[3] = {bo[BF_IPOS], bo[BF_IPOS + 1], bo[BF_IPOS + 2]}, lir[9]; for (int k = 0; k < 9; k++) lir[k] = bo[BF_IR + k];
      float Rk[9], rr[3], com[3]; mulMM(Rk, Rown, lir); mulMV(rr, Rown, lip);
      for (int k = 0; k < 3; k++) com[k] = pown[k] + rr[k];
      const int depth = isfinger ? NCHAIN - 1 : ai;  // deepest chain hinge above (or at) the own link
      // velocity-product accelerations down the chain (gravity as base acceleration)
      float wv[3] = {0, 0, 0}, al[3] = {0, 0, 0}, ao[3] = {-grav[0], -grav[1], -grav[2]};
#pragma unroll
      for (int j = 0; j < NCHAIN; j++) {
        const float qdj = S.qd[j];
        if (j <= depth) {
          if (j > 0) {
            float d[3] = {S.xpos[1 + j][0] - S.xpos[j][0], S.xpos[1 + j][1] - S.xpos[j][1], S.xpos[1 + j][2] - S.xpos[j][2]}, t1[3], t2[3], t3[3];
            cross3(t1, wv, d); cross3(t2, wv, t1); cross3(t3, al, d);
            for (int k = 0; k < 3; k++) ao[k] += t3[k] + t2[k];
          }
          const float axj[3] = {S.axw[1 + j][0], S.axw[1 + j][1], S.axw[1 + j][2]};
          float wxa[3]; cross3(wxa, wv, axj);
          for (int k = 0; k < 3; k++) { al[k] += wxa[k] * qdj; wv[k] += axj[k] * qdj; }
        }
      }
      if (isfinger) {  // slide joint: the origin moves with the parent, plus the Coriolis term of the sliding rate
        float d[3] = {pown[0] - S.xpos[NCHAIN][0], pown[1] - S.xpos[NCHAIN][1], pown[2] - S.xpos[NCHAIN][2]}, t1[3], t2[3], t3[3], wxa[3];
        cross3(t1, wv, d); cross3(t2, wv, t1); cross3(t3, al, d); cross3(wxa, wv, axown);
        for (int k = 0; k < 3; k++) ao[k] += t3[k] + t2[k] + 2.f * wxa[k] * qd;
      }
      float t1[3], t2[3], t3[3], ac[3];
      cross3(t1, wv, rr); cross3(t2, wv, t1); cross3(t3, al, rr);
      for (int k = 0; k < 3; k++) ac[k] = ao[k] + t3[k] + t2[k];
      float Iw[3], Ia[3], gy[3]; inertia_mul(Iw, Rk, di, wv); inertia_mul(Ia, Rk, di, al); cross3(gy, wv, Iw);
      float Fk[3] = {mass * ac[0], mass * ac[1], mass * ac[2]}, Nk[3] = {Ia[0] + gy[0], Ia[1] + gy[1], Ia[2] + gy[2]};
      // projections on the own link's ancestors: entries 0..6 = chain hinges, entry 7 = own finger slide
      float Jv[8][3], bias[8], Mc[28], Mf[8];
#pragma unroll
      for (int e = 0; e < NCHAIN; e++) {
        const bool val = e <= depth;
        const float axe[3] = {S.axw[1 + e][0], S.axw[1 + e][1], S.axw[1 + e][2]};
        float ri[3] = {com[0] - S.xpos[1 + e][0], com[1] - S.xpos[1 + e][1], com[2] - S.xpos[1 + e][2]}, rxF[3];
        cross3(Jv[e], axe, ri); cross3(rxF, ri, Fk);
        bias[e] = val ? axe[0] * (Nk[0] + rxF[0]) + axe[1] * (Nk[1] + rxF[1]) + axe[2] * (Nk[2] + rxF[2]) : 0.f;
        if (!val) Jv[e][0] = Jv[e][1] = Jv[e][2] = 0.f;
      }
      for (int k = 0; k < 3; k++) Jv[7][k] = isfinger ? axown[k] : 0.f;
      bias[7] = isfinger ? dot3(axown, Fk) : 0.f;
#pragma unroll
      for (int a = 0; a < NCHAIN; a++) {
        const float axa[3] = {S.axw[1 + a][0], S.axw[1 + a][1], S.axw[1 + a][2]};
        float tB[3]; inertia_mul(tB, Rk, di, axa);
        const bool va = a <= depth;
#pragma unroll
        for (int b = 0; b <= a; b++) Mc[tri(a, b)] = va ? mass * dot3(Jv[a], Jv[b]) + tB[0] * S.axw[1 + b][0] + tB[1] * S.axw[1 + b][1] + tB[2] * S.axw[1 + b][2] : 0.f;
      }
#pragma unroll
      for (int b = 0; b < NCHAIN; b++) Mf[b] = mass * dot3(Jv[7], Jv[b]);  // slide row: translational coupling only
      Mf[7] = mass * dot3(Jv[7], Jv[7]);
      // sums over the links: chain block (all lanes), finger rows (only the finger's own link contributes)
      for (int k = 0; k < 28; k++) Mc[k] = gsum(Mc[k]);
      float M7[8], M8[8];
#pragma unroll
      for (int b = 0; b < 8; b++) { M7[b] = gsum(ai == 7 && isarm ? Mf[b] : 0.f); M8[b] = gsum(ai == 8 && isarm ? Mf[b] : 0.f); }
      float bown = isfinger ? bias[7] : 0.f;
#pragma unroll
      for (int e = 0; e < NCHAIN; e++) { float b = gsum(bias[e]); if (e == ai && isarm) bown = b; }
      // full 9x9 (packed lower) with armature on the diagonal, shared through LDS so that each lane can fetch its row
      float Lm[45];
#pragma unroll
      for (int a = 0; a < NCHAIN; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) Lm[tri(a, b)] = Mc[tri(a, b)];
#pragma unroll
      for (int b = 0; b < NCHAIN; b++) { Lm[tri(7, b)] = M7[b]; Lm[tri(8, b)] = M8[b]; }
      Lm[tri(7, 7)] = M7[7]; Lm[tri(8, 7)] = 0.f; Lm[tri(8, 8)] = M8[7];
#pragma unroll
      for (int a = 0; a < NA; a++) Lm[tri(a, a)] += gF[m.oDofF + (6 + a) * DOF_F + DF_ARM];
      if (l == 0) {
#pragma unroll
        for (int a = 0; a < NA; a++)
#pragma unroll
          for (int b = 0; b <= a; b++) { S.M[a][b] = Lm[tri(a, b)]; S.M[b][a] = Lm[tri(a, b)]; }
      }
      // position servo, joint-level actuator force clamp
      float cc = u; if (af[AF_CLIM] != 0.f) cc = jh_clampf(cc, af[AF_CLO], af[AF_CHI]);
      float fa = hasact ? af[AF_KP] * (cc - q) - af[AF_KV] * qd : 0.f;
      if (df[DF_FRCLIM] * en != 0.f) fa = jh_clampf(fa, df[DF_FRCLO], df[DF_FRCHI]);
      fs_own = -df[DF_DAMP] * en * qd - bown + fa;
      __syncthreads();
      float x9[NA];
      {
        float R9[NA];
        static_for<NA>([&](auto bc) { constexpr int b = decltype(bc)::value; const float fb = row_bcast<6 + b>(fs_own); R9[b] = l == 15 ? fb : S.M[ai][b]; });
        arm_chol_solve(R9, x9, l);
      }
      a0_own = 0.f;
#pragma unroll
      for (int a = 0; a < NA; a++) if (a == ai) a0_own = x9[a];
      Md_own = S.M[ai][ai];
      // free body: M = diag(m, m, m, I)
      const float wc[3] = {S.cv[3], S.cv[4], S.cv[5]};
      float Icw[3] = {cI[0] * wc[0], cI[1] * wc[1], cI[2] * wc[2]}, gc[3]; cross3(gc, wc, Icw);
      for (int k = 0; k < 3; k++) { fsc[k] = cmass * grav[k]; a0c[k] = grav[k]; fsc[3 + k] = -gc[k]; a0c[3 + k] = -gc[k] / cI[k]; }
      if (iscube) {
#pragma unroll
        for (int k = 0; k < 6; k++) if (k == l) { fs_own = fsc[k]; a0_own = a0c[k]; }
        Md_own = l < 3 ? cmass : cI[l < 3 ? 0 : l - 3];
      }
      if (!hasdof) { fs_own = 0.f; a0_own = 0.f; Md_own = 1.f; }
    }
    __syncthreads();
    PH6(2)
    // ================================================================ collision: the candidate pairs (78) over the lanes, balanced narrow phase
    {
      int nh = 0;
      // every geom's world centre and bounding radius go to LDS once (the storage of the raw contact pool: its last reader was the previous step's slot loading, its next
      // writer is the narrow phase below), so a pair's first test is two LDS reads instead of a chain of dependent global loads (pair -> geom -> body -> pose) per pair
      float (*gcen)[4] = reinterpret_cast<float (*)[4]>(&S.raw[0][0]);
      for (int g = l; g < m.NAG; g += G) {
        const float* f = gF + m.oAGF + g * GEOM_F;
        float pc[3]; geom_pose3(S, f, gI[m.oAGI + g * GEOM_I], pc, nullptr, false);
        gcen[g][0] = pc[0]; gcen[g][1] = pc[1]; gcen[g][2] = pc[2]; gcen[g][3] = f[GF_RBOUND];
      }
      __syncthreads();
      for (int base = 0; base < m.NPAIR; base += G) {
        const int p = base + l;
        bool hit = false;
        if (p < m.NPAIR) {
          const int g1 = gI[m.oPairI + 2 * p], g2 = gI[m.oPairI + 2 * p + 1];
          const float* f1 = gF + m.oAGF + g1 * GEOM_F; const float* f2 = gF + m.oAGF + g2 * GEOM_F;
          const float dc[3] = {gcen[g2][0] - gcen[g1][0], gcen[g2][1] - gcen[g1][1], gcen[g2][2] - gcen[g1][2]}, rb2 = gcen[g2][3], rs = gcen[g1][3] + rb2;
          hit = dot3(dc, dc) <= rs * rs;
          if (hit) {  // second level: the bounding sphere of geom 2 against the BOX geom 1 itself (the table's bounding sphere alone contains the whole scene: without
                      // this its twenty pairs reach the narrow phase in every step).  Conservative, so the contacts do not change.
            float R1[9], dl[3];
            const int b1 = gI[m.oAGI + g1 * GEOM_I];
            if (b1 < 0) { for (int k = 0; k < 9; k++) R1[k] = f1[GF_R + k]; } else mulMM(R1, S.xR[b1], f1 + GF_R);
            mulMTV(dl, R1, dc);
            const float ex = fmaxf(fabsf(dl[0]) - f1[GF_SIZE], 0.f), ey = fmaxf(fabsf(dl[1]) - f1[GF_SIZE + 1], 0.f), ez = fmaxf(fabsf(dl[2]) - f1[GF_SIZE + 2], 0.f);
            hit = ex * ex + ey * ey + ez * ez <= rb2 * rb2;
          }
        }
        unsigned m16 = (unsigned)((__ballot(hit) >> (16 * r)) & 0xFFFFull);
        int pos = nh + __popc(m16 & ((1u << l) - 1u));
        if (hit && pos < MAXHIT) S.hits[pos] = (unsigned short)p;
        nh += __popc(m16);
      }
      if (nh > MAXHIT) { if (l == 0 && live && stats) atomicAdd(stats, nh - MAXHIT); nh = MAXHIT; }  // candidate pairs beyond the list: counted with the dropped contacts
      __syncthreads();
#ifdef JH_V6_PHASE_BROAD
      PH6(6)  // (diagnostic: the broad phase lands in the 'tail' slot)
#endif
      for (int base = 0; __any(base < nh); base += G) {
        const int idx = base + l;
        if (idx < nh) {
          const int p = S.hits[idx];
          const int g1 = gI[m.oPairI + 2 * p], g2 = gI[m.oPairI + 2 * p + 1];
          const float* f1 = gF + m.oAGF + g1 * GEOM_F; const float* f2 = gF + m.oAGF + g2 * GEOM_F;
          float p1[3], R1[9], p2[3], R2[9], h1[3] = {f1[GF_SIZE], f1[GF_SIZE + 1], f1[GF_SIZE + 2]}, h2[3] = {f2[GF_SIZE], f2[GF_SIZE + 1], f2[GF_SIZE + 2]};
          geom_pose3(S, f1, gI[m.oAGI + g1 * GEOM_I], p1, R1, true); geom_pose3(S, f2, gI[m.oAGI + g2 * GEOM_I], p2, R2, true);
          const int sbA = gI[m.oAGI + g1 * GEOM_I], sbB = gI[m.oAGI + g2 * GEOM_I];
          Sink6 sk{&S, stats, p, sbA >= 1 && sbB >= 1, ovf_all ? ovf_all + (size_t)nc * (NOVF * RAW_F) : nullptr};  // (copies of a rollout in latency mode write the same values to the same row)
          if (gI[m.oAGI + g2 * GEOM_I + 1] == GCAPSULE) collide_box_capsule(sk, p1, R1, h1, p2, R2, h2[0], h2[1]);  // (a pair's capsule is its second geom: jh_model_is_fr3)
          else collide_box_box(sk, p1, R1, h1, p2, R2, h2);
        }
      }
    }
    __syncthreads();
    PH6(3)
    // ================================================================ constraint rows
    const int nff = S.nff < NFF ? S.nff : NFF;
    __syncthreads();
    // The constraint rows and the Newton solver exist once per slot count (jh_engine_v5.hip does the same): a wave in which some rollout has more general contacts than the
    // LDS pool holds runs the copy with NSBIG slots per lane, whose upper slots come from the rollout's row of the global overflow pool; every other wave the copy with NSL.
    float a_own; int iters_this = 0;
    auto solve_step = [&](auto NS_, auto NF_) __attribute__((always_inline)) {
    constexpr int NS = decltype(NS_)::value;
    constexpr int NF = decltype(NF_)::value;  // finger-finger slots per lane: NFS, or 0 in the copy for wave-steps without a finger-finger contact (48 registers less)
    SlotF sf[NF > 0 ? NF : 1];
#pragma unroll
    for (int k = 0; k < NF; k++) {  // finger-finger contacts go straight into registers of their owner lane
      const int c = l + 16 * k;
      sf[k].D = 0.f; sf[k].mu = 0.f;
      for (int w = 0; w < 3; w++) sf[k].aref[w] = sf[k].Jf[w] = 0.f;
      if (c < nff) {
        const float* e = S.ffraw[c];
        float fr[9]; fr[0] = e[0]; fr[1] = e[1]; fr[2] = e[2];
        make_frame(fr);
        const float dist = e[3]; const int p = __float_as_int(e[4]);
        const int g1 = gI[m.oPairI + 2 * p], g2 = gI[m.oPairI + 2 * p + 1];
        const int bA = gI[m.oAGI + g1 * GEOM_I], bB = gI[m.oAGI + g2 * GEOM_I];
        const float s13 = (bB == LF ? 1.f : 0.f) - (bA == LF ? 1.f : 0.f);
        const float a13[3] = {S.axw[LF][0], S.axw[LF][1], S.axw[LF][2]};
        for (int w = 0; w < 3; w++) sf[k].Jf[w] = s13 * dot3(fr + 3 * w, a13);  // = s14 * (frame . a14): the two slide axes are antiparallel
        const float* f1 = gF + m.oAGF + g1 * GEOM_F; const float* f2 = gF + m.oAGF + g2 * GEOM_F;
        const float* q1 = gF + m.oGPF + g1 * GP_F; const float* q2 = gF + m.oGPF + g2 * GP_F;
        const float mu = fmaxf(f1[GF_MU], f2[GF_MU]), tran = f1[GF_TRAN] + f2[GF_TRAN];
        float si[5]; for (int w = 0; w < 5; w++) si[w] = 0.5f * (q1[2 + w] + q2[2 + w]);
        const float tc = fmaxf(0.5f * (q1[0] + q2[0]), 2.f * h), dr_ = 0.5f * (q1[1] + q2[1]);
        const float cK = 1.f / fmaxf(1e-15f, si[1] * si[1] * tc * tc * dr_ * dr_), cB = 2.f / fmaxf(1e-15f, si[1] * tc);
        const float imp = impedance(si, dist);
        const float R0 = fmaxf(1e-15f, (1.f - imp) / imp * tran * (1.f + mu * mu));
        const float Rpy = fmaxf(1e-15f, 2.f * (mu * mu / fmaxf(1e-15f, impratio)) * R0);
        sf[k].D = 1.f / Rpy; sf[k].mu = mu;
        const float vff = S.qd[LF - 1] + S.qd[RF - 1];
        for (int w = 0; w < 3; w++) { const float vel = sf[k].Jf[w] * vff; sf[k].aref[w] = -cB * vel - (w == 0 ? cK * imp * dist : 0.f); }
      }
    }
    const int ncon = S.ncon < 16 * NS ? S.ncon : 16 * NS;
    Slot6 sl[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {  // owner lanes: frame, point, sides, per-pair solver parameters, reference acceleration
      const int c = l + 16 * k;
      sl[k].sa = -2; sl[k].sb = -2; sl[k].D = 0.f; sl[k].mu = 0.f;
      for (int w = 0; w < 9; w++) sl[k].fr[w] = 0.f;
      for (int w = 0; w < 3; w++) sl[k].pos[w] = sl[k].aref[w] = sl[k].jar[w] = sl[k].jp[w] = 0.f;
      if (c < ncon) {
        const float* e = c < NCP ? S.raw[c] : ovf_all + (size_t)nc * (NOVF * RAW_F) + (c - NCP) * RAW_F;
        sl[k].pos[0] = e[0]; sl[k].pos[1] = e[1]; sl[k].pos[2] = e[2];
        sl[k].fr[0] = e[3]; sl[k].fr[1] = e[4]; sl[k].fr[2] = e[5];
        make_frame(sl[k].fr);
        const float dist = e[6]; const int p = __float_as_int(e[7]);
        const int g1 = gI[m.oPairI + 2 * p], g2 = gI[m.oPairI + 2 * p + 1];
        sl[k].sa = gI[m.oAGI + g1 * GEOM_I]; sl[k].sb = gI[m.oAGI + g2 * GEOM_I];
        const float* f1 = gF + m.oAGF + g1 * GEOM_F; const float* f2 = gF + m.oAGF + g2 * GEOM_F;
        const float* q1 = gF + m.oGPF + g1 * GP_F; const float* q2 = gF + m.oGPF + g2 * GP_F;
        const float mu = fmaxf(f1[GF_MU], f2[GF_MU]), tran = f1[GF_TRAN] + f2[GF_TRAN];
        float si[5]; for (int w = 0; w < 5; w++) si[w] = 0.5f * (q1[2 + w] + q2[2 + w]);
        const float tc = fmaxf(0.5f * (q1[0] + q2[0]), 2.f * h), dr_ = 0.5f * (q1[1] + q2[1]);
        const float cK = 1.f / fmaxf(1e-15f, si[1] * si[1] * tc * tc * dr_ * dr_), cB = 2.f / fmaxf(1e-15f, si[1] * tc);
        const float imp = impedance(si, dist);
        const float R0 = fmaxf(1e-15f, (1.f - imp) / imp * tran * (1.f + mu * mu));
        const float Rpy = fmaxf(1e-15f, 2.f * (mu * mu / fmaxf(1e-15f, impratio)) * R0);
        sl[k].D = 1.f / Rpy; sl[k].mu = mu;
        float vel[3]; slot_Jx(sl[k], S, S.cv, S.qd, vel);
        sl[k].aref[0] = -cB * vel[0] - cK * imp * dist; sl[k].aref[1] = -cB * vel[1]; sl[k].aref[2] = -cB * vel[2];
      }
    }
    __syncthreads();  // (every lane has taken its contacts out of the raw pool: the Newton matrices may now overwrite it)
    DofRows6 dr;
    dr.fl = df[DF_FL] * en; dr.fD = df[DF_FD]; dr.fR = dr.fD > 0.f ? 1.f / dr.fD : 0.f; dr.faref = -df[DF_FB] * qd; dr.lims = 0.f; dr.laref = 0.f; dr.lD = 0.f; dr.jf = dr.jl = dr.pf = dr.pl = 0.f;
    if (df[DF_LIMITED] * en != 0.f) {
      float dlo = q - df[DF_LO], dhi = df[DF_HI] - q, dist = fminf(dlo, dhi);
      if (dist < 0.f) {
        float c_si[5]; for (int k = 0; k < 5; k++) c_si[k] = df[DF_SOLIMP + k];
        float sg = dlo < dhi ? 1.f : -1.f, imp = impedance(c_si, dist), R = fmaxf(1e-15f, (1.f - imp) / imp * df[DF_INVW]);
        dr.lims = sg; dr.lD = 1.f / R; dr.laref = -df[DF_LB] * (sg * qd) - df[DF_LK] * imp * dist;
      }
    }
    // joint equality (q13 - a0 - a1 q14 = 0): both finger lanes hold the row; quad 3 = lanes 12..15
    float eD = 0.f, earef = 0.f, ejar = 0.f, ejp = 0.f;
    if (has_eq) {
      const float q13 = quad_get(q, 1), q14 = quad_get(q, 2), v13 = quad_get(qd, 1), v14 = quad_get(qd, 2);
      const float pos = q13 - e_a0 - e_a1 * q14, vel = v13 - e_a1 * v14;
      const float imp = impedance(e_si, pos), R = fmaxf(1e-15f, (1.f - imp) / imp * e_invw);
      eD = 1.f / R; earef = -e_B * vel - e_K * imp * pos;
    }
    PH6(4)
    // ================================================================ Newton solver
    float sff = 0.f;  // sff = a13 + a14 of the current iterate: all a finger-finger contact sees of it
    const float iMd = 1.f / Md_own;
    const float snorm = gsum(hasdof ? fs_own * fs_own * iMd : 0.f);
    // does any general contact of the wave touch the arm?  Only those scatter into the arm's gradient rows and Hessian rows (LDS float atomics); a wave-step whose contacts all
    // lie between the free box and the static geometry keeps both in registers (the box's entries are row sums) and skips the publish / barrier / read-back of either.
    bool armc = false;
#pragma unroll
    for (int k = 0; k < NS; k++) armc = armc || (sl[k].sa > -2 && (sl[k].sa >= 1 || sl[k].sb >= 1));
    const bool arm_any = __any(armc);
    iters_this = 0;
    {
      // ---- warm start: the better of last step's acceleration and the unconstrained one
      {
        if (hasdof) { S.vec[0][l] = qws; S.vec[1][l] = a0_own; S.vec[2][l] = qws - a0_own; }
        __syncthreads();
        float xc[6], jar_ws[NS][3];
        for (int k = 0; k < 6; k++) xc[k] = S.vec[0][k];
        for (int k = 0; k < NS; k++) if (sl[k].sa > -2) { float jx[3]; slot_Jx(sl[k], S, xc, S.vec[0] + 6, jx); for (int w = 0; w < 3; w++) sl[k].jar[w] = jx[w] - sl[k].aref[w]; }
        const float sff_ws = S.vec[0][13] + S.vec[0][14], sff_0 = S.vec[1][13] + S.vec[1][14];
        dr.jf = qws - dr.faref; dr.jl = dr.lims * qws - dr.laref;
        ejar = has_eq ? quad_get(qws, 1) - e_a1 * quad_get(qws, 2) - earef : 0.f;
        float mdw = 0.f;
        if (isarm) { for (int a = 0; a < NA; a++) mdw += S.M[ai][a] * S.vec[2][6 + a]; } else if (iscube) mdw = Md_own * (qws - a0_own);
        const float cost_ws = gsum(lane_rows_cost<NS, NF>(sl, sf, sff_ws, dr, eq_lane, eD, ejar) + (hasdof ? 0.5f * (qws - a0_own) * mdw : 0.f));
        for (int k = 0; k < NS; k++) for (int w = 0; w < 3; w++) jar_ws[k][w] = sl[k].jar[w];
        const float jf_ws = dr.jf, jl_ws = dr.jl, ej_ws = ejar;
        for (int k = 0; k < 6; k++) xc[k] = S.vec[1][k];
        for (int k = 0; k < NS; k++) if (sl[k].sa > -2) { float jx[3]; slot_Jx(sl[k], S, xc, S.vec[1] + 6, jx); for (int w = 0; w < 3; w++) sl[k].jar[w] = jx[w] - sl[k].aref[w]; }
        dr.jf = a0_own - dr.faref; dr.jl = dr.lims * a0_own - dr.laref;
        ejar = has_eq ? quad_get(a0_own, 1) - e_a1 * quad_get(a0_own, 2) - earef : 0.f;
        const float cost_0 = gsum(lane_rows_cost<NS, NF>(sl, sf, sff_0, dr, eq_lane, eD, ejar));
        if (cost_ws < cost_0) {
          a_own = qws;
          for (int k = 0; k < NS; k++) for (int w = 0; w < 3; w++) sl[k].jar[w] = jar_ws[k][w];
          dr.jf = jf_ws; dr.jl = jl_ws; ejar = ej_ws; sff = sff_ws;
        } else { a_own = a0_own; sff = sff_0; }
        __syncthreads();
      }
      bool has_rows_l = dr.fl > 0.f || dr.lims != 0.f || has_eq;
      for (int k = 0; k < NS; k++) has_rows_l |= sl[k].sa > -2;
      for (int k = 0; k < NF; k++) has_rows_l |= sf[k].D > 0.f;
      bool act = gor((int)has_rows_l) != 0;
      if (!act) { a_own = a0_own; sff = 0.f; }
      float hdiag = 0.f;  // own diagonal entry of the last assembled Hessian
      for (int it = 0; it < cap && __any(act); it++) {
        // Everything a contact slot needs in an iteration is invariant over the iterations, so the compiler would compute it once before the loop (lever arms,
        // Jacobian columns, pyramid constants of every slot), run out of registers and reload all of it from scratch memory in every iteration; with the
        // slots made opaque it recomputes from the 25 / 8 numbers of a slot instead (jh_engine_v5.hip found the same)
#if JH_V6_OPAQUE
#pragma unroll
        for (int k = 0; k < NS; k++) {
          OPAQUE6(sl[k].sa); OPAQUE6(sl[k].sb);
          for (int w = 0; w < 3; w++) OPAQUE6(sl[k].pos[w]);
          for (int w = 0; w < 9; w++) OPAQUE6(sl[k].fr[w]);
          OPAQUE6(sl[k].D); OPAQUE6(sl[k].mu);
        }
#pragma unroll
        for (int k = 0; k < NF; k++) { OPAQUE6(sf[k].D); OPAQUE6(sf[k].mu); for (int w = 0; w < 3; w++) { OPAQUE6(sf[k].Jf[w]); OPAQUE6(sf[k].aref[w]); } }
#endif
        // ---- (1) gradient row: M (a - a0) + dof rows + equality - J' f
        const float da_own = a_own - a0_own;
        if (hasdof) S.vec[0][l] = da_own;
        __syncthreads();
        float g_own = 0.f, hd = 0.f;
        if (isarm) { for (int a = 0; a < NA; a++) g_own += S.M[ai][a] * S.vec[0][6 + a]; } else if (iscube) g_own = Md_own * da_own;
        if (dr.fl > 0.f) {
          float x = dr.jf, fl = dr.fl, lim = dr.fR * fl;
          if (x <= -lim) g_own -= fl; else if (x >= lim) g_own += fl; else { g_own += dr.fD * x; hd += dr.fD; }
        }
        if (dr.lims != 0.f && dr.jl < 0.f) { g_own += dr.lims * dr.lD * dr.jl; hd += dr.lD; }
        if (has_eq) { if (l == 13) g_own += eD * ejar; if (l == 14) g_own -= e_a1 * eD * ejar; }
        float ffg = 0.f, ffh = 0.f;  // finger-finger contacts: -Jf'f and Jf'W Jf, the same number on both finger dofs and on their coupling
#pragma unroll
        for (int k = 0; k < NF; k++) if (sf[k].D > 0.f) {
          const float* jf = sf[k].Jf;
          const float jar[3] = {fmaf(jf[0], sff, -sf[k].aref[0]), fmaf(jf[1], sff, -sf[k].aref[1]), fmaf(jf[2], sff, -sf[k].aref[2])};
          float f[3], Wm[6]; pyramid_eval(jar, sf[k].D, sf[k].mu, f, Wm);
          ffg -= jf[0] * f[0] + jf[1] * f[1] + jf[2] * f[2];
          const float G0 = Wm[0] * jf[0] + Wm[1] * jf[1] + Wm[3] * jf[2], G1 = Wm[1] * jf[0] + Wm[2] * jf[1] + Wm[4] * jf[2], G2 = Wm[3] * jf[0] + Wm[4] * jf[1] + Wm[5] * jf[2];
          ffh += jf[0] * G0 + jf[1] * G1 + jf[2] * G2;
        }
        if (__any(nff > 0)) {
          ffg = gsum(ffg); ffh = gsum(ffh);
          if (l == 13 || l == 14) g_own += ffg;
        }
        // the general contacts' -J'f: the owner lane of a contact adds its force to the rows of the dofs it acts on (float atomics, matrix-free columns)
        if (arm_any) {
          if (hasdof) S.g[l] = g_own;
          __syncthreads();
        }
        float gcp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (act) {
#pragma unroll
          for (int k = 0; k < NS; k++) if (sl[k].sa > -2) {
            const Slot6& t = sl[k];
            float f[3], Wm[6]; pyramid_eval(t.jar, t.D, t.mu, f, Wm);
            if (f[0] == 0.f && f[1] == 0.f && f[2] == 0.f) continue;  // separated contact
            const float Fw[3] = {t.fr[0] * f[0] + t.fr[3] * f[1] + t.fr[6] * f[2], t.fr[1] * f[0] + t.fr[4] * f[1] + t.fr[7] * f[2], t.fr[2] * f[0] + t.fr[5] * f[1] + t.fr[8] * f[2]};
            slot_force(S, t, Fw, gcp);
          }
        }
        if (arm_any) {
          __syncthreads();
          if (hasdof) g_own = S.g[l];
        }
#pragma unroll
        for (int q = 0; q < 6; q++) { const float v = gsum(gcp[q]); if (l == q) g_own += v; }
        // ---- (2) convergence on the scaled gradient; leave before any Hessian work once every rollout of the wave is done
        // fp32 floor of the gradient: one ulp of the iterate moves row l of the gradient by H_ll * eps * |a_l|.  Stiff rows (sum D J'J ~ 1e4..1e5 on a finger
        // of 0.2 kg against an acceleration of a few hundred m/s^2: a closing gripper whose pad stacks meet at 1 m/s) put that far above tol * |smooth force|;
        // the iterate then sits on the fp32 number nearest to the minimiser, the gradient test never passes and the solve would run to the iteration cap
        // hopping in the soft directions on the rounding noise of the stiff ones.  H_ll is the diagonal of the last assembled Hessian (0 before the first).
        // Row by row: only what a row's gradient exceeds its own floor by counts, so the soft rows still have to meet the tolerance themselves.
        const float gfl = JH_V6_NOISE * hdiag * a_own;
        const float gn = gsum(hasdof ? fmaxf(g_own * g_own - gfl * gfl, 0.f) * iMd : 0.f);
        const float gtol = tol * tol * fmaxf(snorm, 1e-12f);
#ifdef JH_V6_EXITSTATS
        if (act && gn <= gtol) n_x[gsum(hasdof ? g_own * g_own * iMd : 0.f) <= gtol ? 0 : 4]++;
#endif
        if (act && gn <= gtol) act = false;
        if (!__any(act)) break;
        if (act) iters_this++;
        // ---- (3) Hessian row r (columns 0..r): M + dof rows + equality + sum_c J_c[:,r]' W_c J_c[:,0..r]; lane 15 holds -g
        PH6(5)
        float Hrow[16];
#pragma unroll
        for (int j = 0; j < 16; j++) Hrow[j] = 0.f;
        if (isarm) {
#pragma unroll
          for (int a = 0; a < NA; a++) Hrow[6 + a] = S.M[ai][a];
        }
#pragma unroll
        for (int j = 0; j < NVT; j++) if (j == l) Hrow[j] = (iscube ? Md_own : Hrow[j]) + hd;
        if (has_eq) {
          if (l == 13) Hrow[13] += eD;
          if (l == 14) { Hrow[14] += e_a1 * e_a1 * eD; Hrow[13] -= e_a1 * eD; }
        }
        if (l == 13) Hrow[13] += ffh;
        if (l == 14) { Hrow[13] += ffh; Hrow[14] += ffh; }
        // the rows go to LDS, the general contacts add J'WJ there (float atomics, matrix-free columns), the lanes take their rows back
        if (arm_any) {
          {  // (whole rows, whatever the lane's role and the rollout's state: entries beyond the diagonal and the rows of lanes without a dof are never read)
            float4* hr = reinterpret_cast<float4*>(S.H[l]);
#pragma unroll
            for (int j4 = 0; j4 < 4; j4++) hr[j4] = make_float4(Hrow[4 * j4], Hrow[4 * j4 + 1], Hrow[4 * j4 + 2], Hrow[4 * j4 + 3]);
          }
          __syncthreads();
        }
        float hcp[21];
#pragma unroll
        for (int e = 0; e < 21; e++) hcp[e] = 0.f;
        if (act) {
#pragma unroll
          for (int k = 0; k < NS; k++) if (sl[k].sa > -2) {
            float f[3], Wm[6]; pyramid_eval(sl[k].jar, sl[k].D, sl[k].mu, f, Wm);
            if (Wm[0] == 0.f && Wm[2] == 0.f && Wm[5] == 0.f) continue;  // no active pyramid row
            slot_assemble(S, sl[k], Wm, hcp);
          }
        }
        if (arm_any) {
          __syncthreads();
          {
            const float4* hr = reinterpret_cast<const float4*>(S.H[l]);
#pragma unroll
            for (int j4 = 0; j4 < 4; j4++) {
              const float4 v = hr[j4]; const bool on = hasdof && act;
              Hrow[4 * j4] = on ? v.x : Hrow[4 * j4]; Hrow[4 * j4 + 1] = on ? v.y : Hrow[4 * j4 + 1]; Hrow[4 * j4 + 2] = on ? v.z : Hrow[4 * j4 + 2]; Hrow[4 * j4 + 3] = on ? v.w : Hrow[4 * j4 + 3];
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 6; q++)
#pragma unroll
          for (int r2 = 0; r2 <= q; r2++) { const float v = gsum(hcp[tri(q, r2)]); if (l == q && act) Hrow[r2] += v; }
        PH6(7)  // (the Hessian assembly alone; the rest of the solve stays in slot 5)
#pragma unroll
        for (int j = 0; j < NVT; j++) if (j == l) hdiag = Hrow[j];
        static_for<NVT>([&](auto jc) {  // the right-hand side -g as a sixteenth row in lane 15 (dof j is lane j's)
          constexpr int j = decltype(jc)::value;
          const float gj = row_bcast<j>(-g_own);
          if (l == 15) Hrow[j] = gj;
        });
        // ---- (4) Cholesky in registers: lane r holds row r (lane 15: the right-hand side as a sixteenth row); at step k every lane takes lane k's diagonal and the
        // column-k entries of the rows its own trailing entries meet with row broadcasts (DPP row_newbcast: an operand modifier, no LDS, no barrier).  Entry (r, j) receives
        // its subtractions in the order k = 0, 1, ... of the left-looking row form this replaces (fifteen publish-to-LDS / barrier / read-back rounds): the same bits.
        // Entries above the diagonal are never read.
        static_for<NVT>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const float rinv = __frsqrt_rn(fmaxf(row_bcast<k>(Hrow[k]), 1e-30f));
          const float lk = Hrow[k] * rinv;
          Hrow[k] = l == k ? rinv : lk;  // (lane k keeps 1 / L_kk in its diagonal slot: the backward solve takes it from there, no register array of its own)
          static_for<NVT - 1 - k>([&](auto jc) {
            constexpr int j = k + 1 + decltype(jc)::value;
            Hrow[j] = __builtin_fmaf(-lk, row_bcast<j>(lk), Hrow[j]);
          });
        });
        // ---- (5) backward solve, redundantly: every lane gets the whole direction p (row 15 = y, column k of L across the lanes)
        float p[NVT];
        static_for<NVT>([&](auto kc) {
          constexpr int k = NVT - 1 - decltype(kc)::value;
          float s = row_bcast<15>(Hrow[k]);
          static_for<NVT - 1 - k>([&](auto jc) {
            constexpr int j = k + 1 + decltype(jc)::value;
            s = __builtin_fmaf(-row_bcast<j>(Hrow[k]), p[j], s);
          });
          p[k] = s * row_bcast<k>(Hrow[k]);
        });
        float p_own = 0.f;
#pragma unroll
        for (int j = 0; j < NVT; j++) if (j == l) p_own = p[j];
        // ---- (6) exact line search along p
        PH6(8)  // (row Cholesky + backward solve)
        float Mp_own = 0.f;
        if (isarm) {
#pragma unroll
          for (int a = 0; a < NA; a++) Mp_own += S.M[ai][a] * p[6 + a];
        } else if (iscube) Mp_own = Md_own * p_own;
        const float pMp = gsum(p_own * Mp_own), pMd = gsum(Mp_own * da_own), gp = gsum(g_own * p_own);
#ifdef JH_V6_EXITSTATS
        if (act && !(gp < 0.f)) n_x[1]++;
#endif
        if (act && !(gp < 0.f)) act = false;
#pragma unroll
        for (int k = 0; k < NS; k++) if (sl[k].sa > -2) slot_Jx(sl[k], S, p, p + 6, sl[k].jp);
        const float spf = p[13] + p[14];
        dr.pf = p_own; dr.pl = dr.lims * p_own;
        ejp = has_eq ? p[13] - e_a1 * p[14] : 0.f;
        float lo = 0.f, hi = -1.f, alpha = 1.f; bool lsact = act;
        for (int ls = 0; ls < JH_V6_LSCAP && __any(lsact); ls++) {
          float d1, d2;
          lane_rows_dir<NS, NF>(sl, sf, sff, spf, dr, eq_lane, eD, ejar, ejp, alpha, &d1, &d2);
          d1 = gsum(d1) + pMd + alpha * pMp; d2 = gsum(d2) + pMp;
          if (lsact) {
            if (fabsf(d1) <= lstol * fabsf(gp)) lsact = false;
            else {
              if (d1 < 0.f) lo = alpha; else hi = alpha;
              float nx = alpha - d1 * __builtin_amdgcn_rcpf(d2);  // (1 ulp: the correctly rounded division is ten instructions per slope evaluation)
              if (hi < 0.f) { if (nx <= lo) nx = 2.f * alpha; }
              else if (nx <= lo || nx >= hi) nx = 0.5f * (lo + hi);
              alpha = nx;
            }
          }
        }
        PH6(9)  // (line search: set-up and slope evaluations)
        // ---- (7) step
#ifdef JH_V6_TRACE
        if (lane == 0 && it >= 6 && it < 22) printf("step %d it %d gn %.3e gtol %.3e gp %.3e alpha %.5g sff %.6f spf %.3e lo %.4g hi %.4g\n", hh, it, gn, gtol, gp, alpha, sff, spf, lo, hi);
#endif
        if (act) {
          a_own += alpha * p_own;
          for (int k = 0; k < NS; k++) for (int w = 0; w < 3; w++) sl[k].jar[w] += alpha * sl[k].jp[w];
          sff = fmaf(alpha, spf, sff);
          dr.jf += alpha * dr.pf; dr.jl += alpha * dr.pl; ejar += alpha * ejp;
#ifdef JH_V6_EXITSTATS
          if (-gp * alpha <= tol * tol * fmaxf(snorm, 1e-12f)) n_x[2]++;
#endif
          if (-gp * alpha <= tol * tol * fmaxf(snorm, 1e-12f)) act = false;
        }
        __syncthreads();
      }
      if (l == 0) { n_iters += iters_this; n_maxed += (iters_this >= cap); }
#ifdef JH_V6_EXITSTATS
      if (act) {
        n_x[3]++;
        if (stats && l == 0 && atomicCAS(stats + 39, 0, 1) == 0) { stats[40] = n_offset + n; stats[41] = hh; }  // the first solve that ran into the iteration cap: which rollout, which step
      }
#endif
    }
    };
    if (__builtin_expect_with_probability(ovf_all != nullptr && __any(S.ncon > NCP), 0, JH_V6_BIGPROB)) {
      __threadfence();  // the overflow rows were written with plain global stores by other lanes of this wave
      solve_step(std::integral_constant<int, NSBIG>{}, std::integral_constant<int, NFS>{});
    }
#if JH_V6_FFSPLIT
#if JH_V6_NS1
    else if (!__any(S.nff > 0) && !__any(S.ncon > G)) solve_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});  // (and at most 16 general contacts per rollout)
#endif
    else if (!__any(S.nff > 0)) solve_step(std::integral_constant<int, NSL>{}, std::integral_constant<int, 0>{});
#endif
    else solve_step(std::integral_constant<int, NSL>{}, std::integral_constant<int, NFS>{});
    PH6(5)
    // ================================================================ implicitfast integration: (M + h diag(d + kv)) qacc = fs + M (a - a0)
    float qc[7], vc[6];  // the free body's state after the step (registers from here to the end of the step only)
    {
      __syncthreads();
      // (the own joint's position / velocity and the free body's state come back from LDS: registers held them only up to the kinematics)
      if (isarm) { q = S.q[ai]; qd = S.qd[ai]; }
      const float da_own = a_own - a0_own;
      if (hasdof) S.vec[0][l] = da_own;
      __syncthreads();
      float rhs_own = fs_own;
      if (isarm) { for (int a = 0; a < NA; a++) rhs_own += S.M[ai][a] * S.vec[0][6 + a]; } else if (iscube) rhs_own += Md_own * da_own;
      if (hasdof) S.vec[1][l] = rhs_own;
      __syncthreads();
      float x9[NA];
      {
        float R9[NA];
        const float dh = h * (gF[m.oDofF + (6 + ai) * DOF_F + DF_DAMP] + gF[m.oDofF + (6 + ai) * DOF_F + DF_KV]);
        static_for<NA>([&](auto bc) {
          constexpr int b = decltype(bc)::value;
          const float rb = row_bcast<6 + b>(rhs_own);
          float v = S.M[ai][b];
          if (b == ai) v += dh;
          R9[b] = l == 15 ? rb : v;
        });
        arm_chol_solve(R9, x9, l);
      }
      float qacc = 0.f;
#pragma unroll
      for (int a = 0; a < NA; a++) if (a == ai) qacc = x9[a];
      if (isarm) { qd = fmaf(h, qacc, qd); q = fmaf(h, qd, q); }
      qws = a_own;
      for (int k = 0; k < 7; k++) qc[k] = S.cq[k];
      for (int k = 0; k < 6; k++) vc[k] = S.cv[k];
      for (int k = 0; k < 3; k++) {  // free body: every lane integrates the rollout's copy of the state from the published right-hand side
        const float al = S.vec[1][k] / cmass, aw = S.vec[1][3 + k] / cI[k];
        vc[k] = fmaf(h, al, vc[k]); vc[3 + k] = fmaf(h, aw, vc[3 + k]);
      }
      for (int k = 0; k < 3; k++) qc[k] = fmaf(h, vc[k], qc[k]);
      const float wn = sqrtf(vc[3] * vc[3] + vc[4] * vc[4] + vc[5] * vc[5]), ang = wn * h;
      if (ang > 0.f) {
        float sn, cs; sincosf(0.5f * ang, &sn, &cs); const float kk = sn / wn;
        float dq[4] = {cs, vc[3] * kk, vc[4] * kk, vc[5] * kk}, *qq = qc + 3;
        float r0 = qq[0] * dq[0] - qq[1] * dq[1] - qq[2] * dq[2] - qq[3] * dq[3];
        float r1 = qq[0] * dq[1] + qq[1] * dq[0] + qq[2] * dq[3] - qq[3] * dq[2];
        float r2 = qq[0] * dq[2] - qq[1] * dq[3] + qq[2] * dq[0] + qq[3] * dq[1];
        float r3 = qq[0] * dq[3] + qq[1] * dq[2] - qq[2] * dq[1] + qq[3] * dq[0];
        qq[0] = r0; qq[1] = r1; qq[2] = r2; qq[3] = r3;
      }
      const float nn = rsqrtf(qc[3] * qc[3] + qc[4] * qc[4] + qc[5] * qc[5] + qc[6] * qc[6]);
      qc[3] *= nn; qc[4] *= nn; qc[5] *= nn; qc[6] *= nn;
      __syncthreads();  // (every lane has read the old copy)
      if (l < 7) { float v = qc[0]; for (int k = 1; k < 7; k++) if (k == l) v = qc[k]; S.cq[l] = v; }
      if (l < 6) { float v = vc[0]; for (int k = 1; k < 6; k++) if (k == l) v = vc[k]; S.cv[l] = v; }
    }
    if (MATERIALIZE) {
      if (states && live) {
        float* o = states + ((size_t)nc * H + hh) * NX;
        if (isarm) { o[7 + ai] = q; o[NQ + 6 + ai] = qd; }
      }
      __syncthreads();
      if (states && live) {
        float* o = states + ((size_t)nc * H + hh) * NX;
        if (l < 7) o[l] = S.cq[l];
        if (l < 6) o[NQ + l] = S.cv[l];
      }
    } else {
      // running cost: the state after the step with the sensors of the forward pass that produced it (judo/tasks/fr3_pick.py:225-311)
      __syncthreads();
      if (isarm) { S.vec[0][ai] = q; S.vec[1][ai] = qd; }
      __syncthreads();
      float qpos[NQ], qvel[NVT], y[NS];
      for (int k = 0; k < 7; k++) qpos[k] = qc[k];
      for (int k = 0; k < 6; k++) qvel[k] = vc[k];
      for (int a = 0; a < NA; a++) { qpos[7 + a] = S.vec[0][a]; qvel[6 + a] = S.vec[1][a]; }
      for (int k = 0; k < NS; k++) y[k] = S.y[k];
      acc += fr3_step_cost(sTp, phase, qpos, qvel, NVT, y, H > 1 ? 1.f - (float)hh / (float)(H - 1) : 1.f);
      __syncthreads();
    }
  }
  if (!MATERIALIZE && live && l == 0) costs[n] = acc;
#ifdef JH_V6_EXITSTATS
  if (stats && live && l == 0) for (int k = 0; k < 5; k++) atomicAdd(stats + 24 + k, n_x[k]);
#ifdef JH_V6_PHASES
  PH6(6)
  if (stats && lane == 0) for (int k = 0; k < 10; k++) atomicAdd((unsigned long long*)(stats + 4) + k, (unsigned long long)ph_acc[k]);
#endif
#endif
  if (stats && live && l == 0) { if (n_maxed) atomicAdd(stats + 1, n_maxed); atomicAdd(stats + 2, n_iters); atomicAdd(stats + 3, H); }
}

}  // namespace

bool jh_model_is_fr3(const jh_model* m) {
  if (!(m->kind == JH_TASK_FR3_PICK && m->nq == NQ && m->nv == NVT && m->nu == NU && m->ns == NS && m->h_i.size() > 24 && m->h_i[0] == NMB && m->h_i[9] == 0)) return false;
  const int gi = m->h_i[13];
  // 7 hinges in a chain welded to the world, two slides on the last link; boxes only; at most one joint equality on the fingers
  for (int b = 1; b <= 9; b++) {
    const int* bi = m->h_i.data() + jh_eng::HEADER_I + b * jh_eng::BODY_I;
    const int par = bi[0], jt = bi[1], dof = bi[2];
    if (dof != 5 + b) return false;
    if (b <= 7) { if (jt != jh_eng::JHINGE || par != (b == 1 ? -1 : b - 1)) return false; }
    else if (jt != jh_eng::JSLIDE || par != 7) return false;
  }
  const int nag = m->h_i[gi], npair = m->h_i[gi + 1], neq = m->h_i[gi + 2], ngs = m->h_i[gi + 5];
  if (neq > 1 || ngs > G || m->h_i[gi + 4] > 8) return false;
  if (nag > NCP * RAW_F / 4) return false;  // the geoms' centres live in the raw contact pool's storage during the broad phase (k_fr3_v6)
  for (int s = 0; s < ngs; s++) if (m->h_i[gi + 8 + nag * jh_eng::GEOM_I + npair * 2 + neq * jh_eng::EQ_I + m->h_i[gi + 3] + m->h_i[gi + 4] * 4 + s * 4] == 2) return false;  // no jointpos sensors
  {  // the fused mode's shortcut (fr3_cost_reads_distance, jh_engine_common.h) names the distance sensors by their sensordata address: they must be the model's
     // first FR3_NDIST sensordata entries, distance sensor d at address d (fr3_pick.xml: finger-object x 2, finger-table x 2, object-table)
    if (m->h_i[gi + 4] != jh_eng::FR3_NDIST) return false;
    for (int s = 0; s < ngs; s++) {
      const int* si = m->h_i.data() + gi + 8 + nag * jh_eng::GEOM_I + npair * 2 + neq * jh_eng::EQ_I + m->h_i[gi + 3] + m->h_i[gi + 4] * 4 + s * 4;
      if (si[0] == 4 && si[3] != si[1]) return false;
    }
  }
  { int nt = 0; const int* di = m->h_i.data() + gi + 8 + nag * jh_eng::GEOM_I + npair * 2 + neq * jh_eng::EQ_I + m->h_i[gi + 3]; for (int s = 0; s < m->h_i[gi + 4]; s++) nt += di[4 * s + 1] * di[4 * s + 3]; if (nt > MAXDT) return false; }
  for (int g = 0; g < nag; g++) { const int tp = m->h_i[gi + 8 + g * jh_eng::GEOM_I + 1]; if (tp != jh_eng::GBOX && tp != jh_eng::GCAPSULE) return false; }  // boxes; capsules = arm-link stand-ins
  for (int p = 0; p < npair; p++) {  // pairs between two articulated bodies are kept as finger-finger contacts: nothing else qualifies
    const int* pi = m->h_i.data() + gi + 8 + nag * jh_eng::GEOM_I + 2 * p;
    const int b1 = m->h_i[gi + 8 + pi[0] * jh_eng::GEOM_I], b2 = m->h_i[gi + 8 + pi[1] * jh_eng::GEOM_I];
    if (b1 >= 1 && b2 >= 1 && !((b1 == LF && b2 == RF) || (b1 == RF && b2 == LF))) return false;
    // a capsule is always the second geom of its pair and meets a box of the static geometry or of the free body only
    if (m->h_i[gi + 8 + pi[0] * jh_eng::GEOM_I + 1] != jh_eng::GBOX) return false;
    if (m->h_i[gi + 8 + pi[1] * jh_eng::GEOM_I + 1] == jh_eng::GCAPSULE && b1 >= 1) return false;
  }
  {  // the two finger slides must be antiparallel in the frame of their common parent: the finger-finger slots rely on it (struct SlotF)
    float w[2][3];
    for (int f = 0; f < 2; f++) {
      const float* bf = m->h_f.data() + jh_eng::HEADER_F + (LF + f) * jh_eng::BODY_F;
      for (int i = 0; i < 3; i++) w[f][i] = bf[jh_eng::BF_LR + 3 * i] * bf[jh_eng::BF_AXIS] + bf[jh_eng::BF_LR + 3 * i + 1] * bf[jh_eng::BF_AXIS + 1] + bf[jh_eng::BF_LR + 3 * i + 2] * bf[jh_eng::BF_AXIS + 2];
    }
    for (int i = 0; i < 3; i++) if (fabsf(w[0][i] + w[1][i]) > 1e-6f) return false;
  }
  if (neq == 1) { const int* ei = m->h_i.data() + gi + 8 + nag * jh_eng::GEOM_I + npair * 2; if (ei[0] != 13 || ei[1] != 14) return false; }
  for (int u = 0; u < NU; u++) if (m->h_i[jh_eng::HEADER_I + NMB * jh_eng::BODY_I + m->h_i[1] * jh_eng::BLOCK_I + u * jh_eng::ACT_I] != 6 + u) return false;
  return true;
}

int jh_engine6_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                            const float* lohi, const float* tp, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, hipStream_t st) {
  if (!jh_model_is_fr3(m)) { jh_set_error("rollout_cost: the cooperative arm kernel (matrix-free generation) is instantiated for fr3_pick only"); return JH_ERR_UNSUPPORTED; }
  JH_REQUIRE(K <= 8, "rollout_cost: the cooperative arm kernel keeps at most 8 knots per actuator in registers (K=%d)", K);
  const int dshift = jh_latency_shift(N, RPW), per_wave = RPW >> dshift;
  int grid = (N + per_wave - 1) / per_wave;
  // one overflow row per rollout for the general contacts above the LDS pool: stream-ordered allocation (the pool hands the same block back launch after launch),
  // no state on the model handle
  float* ovf = nullptr;
  ovf = jh_launch_scratch(m, (size_t)N * NOVF * RAW_F * sizeof(float), st);  // (nullptr: the LDS capacity alone, drops and the fallback counted)
  hipLaunchKernelGGL(k_fr3_v6<false>, dim3(grid), dim3(WAVE), 0, st, m->d_f, m->d_i, x0, 0, nominal, noise, ldn, sigma, W, lohi, tp, phase, N, n_offset, H, K,
                     costs, knots_out, (const float*)nullptr, (float*)nullptr, (float*)nullptr, m->d_stats, dshift, trace, ovf);
  return jh_launch_done(ovf, st);
}

int jh_engine6_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                           hipStream_t st) {
  if (!jh_model_is_fr3(m)) { jh_set_error("rollout_materialize: the cooperative arm kernel is instantiated for fr3_pick only"); return JH_ERR_UNSUPPORTED; }
  const int dshift = jh_latency_shift(N, RPW), per_wave = RPW >> dshift;
  int grid = (N + per_wave - 1) / per_wave;
  // one overflow row per rollout for the general contacts above the LDS pool: stream-ordered allocation (the pool hands the same block back launch after launch),
  // no state on the model handle
  float* ovf = nullptr;
  ovf = jh_launch_scratch(m, (size_t)N * NOVF * RAW_F * sizeof(float), st);  // (nullptr: the LDS capacity alone, drops and the fallback counted)
  hipLaunchKernelGGL(k_fr3_v6<true>, dim3(grid), dim3(WAVE), 0, st, m->d_f, m->d_i, x0, x0_batched, (const float*)nullptr, (const float*)nullptr, 0,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, N, 0, H, 0, (float*)nullptr,
                     (float*)nullptr, controls, states, sensors, m->d_stats, dshift, (float*)nullptr, ovf);
  return jh_launch_done(ovf, st);
}
