// jh_engine_v4.hip -- cooperative kernel for a floating-base robot on a ground plane (Spot, judo/models/xml/spot_primitive/robot.xml):
// the physics substeps of the policy rollout (mujoco_extensions/system/system_class.cpp:277-331 calls mj_step `physics_substeps` times per
// command row with the control of the policy step held).  32 lanes (two DPP rows) per rollout, 2 rollouts per wave64.
//
//   lane l < 6    free-base dof l (world-frame translation, body-frame rotation: MuJoCo's free-joint convention)
//   lane 6..24    joint dof l-6 (4 legs x 3 + arm x 7): joint state, servo, friction-loss / limit rows, its link's inertia and geoms
//   lane 25       carries the right-hand side as an extra matrix row through the factorisations
//   every lane    one contact slot (capacity 32 per rollout)
//
// Dynamics in spatial-vector form about the base origin (an inertial point at this instant; keeps fp32 magnitudes small however far the
// robot has walked): each dof lane publishes its spatial axis, each body lane its spatial inertia (10 numbers) and bias wrench; composite inertias and
// subtree wrenches are suffix sums along the chains plus one wave reduction for the base (fixed order, no atomics); M rows are `S_j . (Ic_i S_i)` over
// the ancestors (mj_crb / mj_rne restated; oracle/jo_engine.c crb(), rne_bias()).  The solves of a step -- M^-1 f, the Newton systems, (M + h D)^-1 --
// all run through tree_cholesky_solve: chains eliminated before the base (no fill-in), the small diagonal blocks factorised redundantly in registers,
// three LDS exchanges per solve.  Contacts: sphere / capsule / box against the plane (mjc_PlaneSphere / PlaneCapsule / PlaneBox), pyramidal cones,
// parameters mixed on the host; the contact Jacobian is compact (base + the contact's chain).  Sensors (site positions / frame axes) in the last step.
#include "jh_coop.h"

#include <vector>

using namespace jh_eng;
using namespace jh_coop;

namespace {

constexpr int G = 32, RPW = 2, WAVE = 64;
constexpr int NJ = 19, NVT = 25, NQ = 26, NX = 51, NB = 20, MAXD = 7;
constexpr int NCP = 32, RAW_F = 8, NR = NVT + 1;  // NR: row registers
// Contact Jacobian row (contact frame x dofs), compact: a ground contact moves with the base and ONE chain.  [3 b + r] base dof b, [JC + 3 m + r] position m of
// the contact's chain (zero beyond the owner link), rows padded to float4s.
constexpr int JC = 20, JW = 44;
constexpr int TH_F = 32, TH_I = 16, TD_F = 56, TD_I = 4, TG_F = 28, TG_I = 2, TS_F = 16, TS_I = 4;  // judo_amd/tree_model.py
// header floats
enum { TF_DT = 0, TF_IMPRATIO, TF_TOL, TF_MAXITER, TF_LSTOL, TF_GRAV, TF_PLANE_P = 8, TF_PLANE_N = 11, TF_BMASS = 14, TF_BIPOS = 15, TF_BIR = 18, TF_BINERTIA = 27 };
// joint floats
enum { JF_LPOS = 0, JF_LR = 3, JF_AXIS = 12, JF_MASS = 15, JF_IPOS = 16, JF_IR = 19, JF_INERTIA = 28, JF_DAMP = 31, JF_ARM, JF_FL, JF_FB, JF_FD, JF_INVW, JF_LIMITED, JF_LO, JF_HI,
       JF_LK, JF_LB, JF_SOLIMP = 42, JF_KP = 47, JF_KV, JF_CLIM, JF_CLO, JF_CHI, JF_FLIM, JF_FLO, JF_FHI };
// geom floats
enum { GF4_SIZE = 0, GF4_POS = 3, GF4_R = 6, GF4_MU = 15, GF4_K, GF4_B, GF4_SOLIMP = 18, GF4_TRAN = 23 };

// Chain layout the kernel is instantiated for: four legs of three hinges and one arm of seven, contiguous in the dof order (checked by
// jh_tree_create).  With the chains eliminated before the base, the Cholesky factor of the inertia and of every Newton Hessian has no fill-in:
// chain blocks (3x3, 7x7), base-chain coupling, base block -- contacts with the ground couple the base with ONE chain, friction loss and limits
// are diagonal -- so the chains factorise side by side (7 levels) and only the 6 base pivots are sequential (mj_factorM's sparsity, cooperative).
constexpr int NCH = 5;
__host__ __device__ constexpr int CS(int c) { return c < 4 ? 3 * c : 12; }
__host__ __device__ constexpr int CL(int c) { return c < 4 ? 3 : 7; }
constexpr int SF_MAX = 2048, SI_MAX = 160, LBW = 28;

struct __attribute__((aligned(16))) RS4 {  // per-rollout shared state; positions are relative to the base origin
  float vec[3][G];                        // broadcast vectors (read as float4)
  float Lb[7][LBW];                       // base rows (6 = the rhs) pushed through the chain factors: [0..18] chain columns
  float LbT[NJ][8];                       // the same, transposed: per joint the six base-row entries and the rhs entry
  float Hc[NJ][8];                        // rows of the chain blocks: [m] = entry at chain position m <= own
  float Sb[7][8];                         // reduced base system: rows 0..5 lower triangle, row 6 the rhs
  float Tl[NJ][12];                       // this step's joint transforms: body offset (3), body frame x joint rotation (9)
  float Ib[NB][12];                       // body spatial inertias about the base origin: mass, m c (3), rotational part (xx, xy, xz, yy, yz, zz)
  float frc[NB][8];                       // body bias wrenches
  float xpos[NB][3], xR[NB][9];           // body 0 = base, 1 + k = link of joint k
  float Sax[NVT][6];                      // spatial axes (angular, linear)
  float qd[G];
  float raw[NCP][RAW_F];                  // pos3 (relative), dist, geom | chain, tangent hint 3
  float fW[NCP][12];                      // contact frame (9) while the rows are built; then force [0..2] and the 3x3 weight [4..9] of the current Newton iterate
  union { float M[NVT][NVT]; float J[NCP][JW]; };  // the inertia lives in LDS only until every lane has its row in registers (25*25 < 32*44)
  int ncon;
};

// Sum over the 32 lanes (two DPP rows) of a rollout, bit-identical in every lane.  The row exchange is gfx950's v_permlane16_swap: with both
// operands = v it leaves (row0, row0, row2, row2) in one and (row1, row1, row3, row3) in the other -- a VALU op, no trip through the LDS crossbar.
__device__ __forceinline__ float gsum32(float v) {
  v = gsum(v);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int gor32(int v) {
  v = gor(v);
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  return (int)(r[0] | r[1]);
}

__device__ __forceinline__ void rodrigues4(float* Rq, const float* al, float q) {
  float sn, cs; sincosf(q, &sn, &cs); const float t = 1.f - cs, x = al[0], y = al[1], z = al[2];
  Rq[0] = t * x * x + cs; Rq[1] = t * x * y - sn * z; Rq[2] = t * x * z + sn * y;
  Rq[3] = t * x * y + sn * z; Rq[4] = t * y * y + cs; Rq[5] = t * y * z - sn * x;
  Rq[6] = t * x * z - sn * y; Rq[7] = t * y * z + sn * x; Rq[8] = t * z * z + cs;
}
// spatial inertia (mass, mc, rotational part about the reference point) applied to a motion vector (w, v)
__device__ __forceinline__ void inertia6_mul(float* f, const float* I10, const float* s) {
  const float m = I10[0]; const float* mc = I10 + 1; const float* R = I10 + 4;  // R: xx xy xz yy yz zz
  const float* w = s; const float* v = s + 3;
  float cxv[3], cxw[3]; cross3(cxv, mc, v); cross3(cxw, mc, w);
  f[0] = R[0] * w[0] + R[1] * w[1] + R[2] * w[2] + cxv[0];
  f[1] = R[1] * w[0] + R[3] * w[1] + R[4] * w[2] + cxv[1];
  f[2] = R[2] * w[0] + R[4] * w[1] + R[5] * w[2] + cxv[2];
  f[3] = m * v[0] - cxw[0]; f[4] = m * v[1] - cxw[1]; f[5] = m * v[2] - cxw[2];
}
__device__ __forceinline__ void crossm6(float* r, const float* v, const float* s) {  // motion cross product v x s
  float a[3], b[3], c[3]; cross3(a, v, s); cross3(b, v, s + 3); cross3(c, v + 3, s);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
__device__ __forceinline__ void crossf6(float* r, const float* v, const float* f) {  // force cross product v x* f
  float a[3], b[3], c[3]; cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
__device__ __forceinline__ float dot6(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }

struct Slot4 { bool valid; float D, mu, aref[3], jar[3], jp[3]; };
struct DofRows4 { float fl, fD, fR, faref, lims, laref, lD, jf, jl, pf, pl; };

__device__ __forceinline__ void pyramid_dir4(const float* jar, const float* jp, float D, float mu, float* d1, float* d2) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float sg = (k & 1) ? -mu : mu;
    const float x = jar[0] + sg * (k < 2 ? jar[1] : jar[2]), xp = jp[0] + sg * (k < 2 ? jp[1] : jp[2]);
    if (x < 0.f) { *d1 += D * x * xp; *d2 += D * xp * xp; }
  }
}
__device__ __forceinline__ float lane_cost4(const Slot4& sl, const DofRows4& dr) {
  float cs = 0.f;
  if (sl.valid) { float f[3], W[6]; cs += pyramid_eval(sl.jar, sl.D, sl.mu, f, W); }
  if (dr.fl > 0.f) {
    const float x = dr.jf, fl = dr.fl, lim = dr.fR * fl;
    if (x <= -lim) cs += -0.5f * dr.fR * fl * fl - fl * x; else if (x >= lim) cs += -0.5f * dr.fR * fl * fl + fl * x; else cs += 0.5f * dr.fD * x * x;
  }
  if (dr.lims != 0.f && dr.jl < 0.f) cs += 0.5f * dr.lD * dr.jl * dr.jl;
  return cs;
}
__device__ __forceinline__ void lane_dir4(const Slot4& sl, const DofRows4& dr, float al, float* d1, float* d2) {
  float g1 = 0.f, g2 = 0.f;
  if (sl.valid) {
    float jar[3] = {fmaf(al, sl.jp[0], sl.jar[0]), fmaf(al, sl.jp[1], sl.jar[1]), fmaf(al, sl.jp[2], sl.jar[2])};
    pyramid_dir4(jar, sl.jp, sl.D, sl.mu, &g1, &g2);
  }
  if (dr.fl > 0.f) {
    const float jp = dr.pf, x = fmaf(al, jp, dr.jf), fl = dr.fl, lim = dr.fR * fl;
    if (x <= -lim) g1 -= fl * jp; else if (x >= lim) g1 += fl * jp; else { g1 += dr.fD * x * jp; g2 += dr.fD * jp * jp; }
  }
  if (dr.lims != 0.f) { const float jp = dr.pl, x = fmaf(al, jp, dr.jl); if (x < 0.f) { g1 += dr.lD * x * jp; g2 += dr.lD * jp * jp; } }
  *d1 = g1; *d2 = g2;
}

#ifdef JH_V4_PHASES
#define PH_DECL long long ph_t = __builtin_readcyclecounter(), ph_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long* pa = ph_acc;
#define PH(i) { const long long ph_n = __builtin_readcyclecounter(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; }
#define PH_FLUSH if (stats && lane == 0) for (int i = 0; i < 16; i++) atomicAdd((unsigned long long*)(stats + 8) + i, (unsigned long long)ph_acc[i]);
#define PHS(i) { const long long ph_n = __builtin_readcyclecounter(); pa[i] += ph_n - ph_s; ph_s = ph_n; }
#define PHS_DECL long long ph_s = __builtin_readcyclecounter();
#define PA_ARG , pa
#define PA_PARAM , long long* pa
#else
#define PHS(i)
#define PHS_DECL
#define PA_ARG
#define PA_PARAM
#define PH_DECL
#define PH(i)
#define PH_FLUSH
#endif

// who a lane is in the factorisation
struct Role {
  bool isjoint; int cdepth, cstart, clen, cid;  // chain lanes: position in the chain, first joint of the chain, chain length, chain index
  int bl;                                       // base lanes 0..5, the right-hand-side lane 6, everybody else -1
};

// Row layout (26 registers), the same for every lane: row[j] = A[i][joint j] (j < 19), row[19 + m] = A[i][base m].  A joint lane only ever uses
// the entries of its own chain up to itself, a base lane b its 19 joint columns and base columns m <= b; lane 6 of the base group (the rhs lane)
// carries the right-hand side in the same layout.  One layout = one instruction stream for the accumulation of J' W J.
__device__ __forceinline__ float chain_entry(const float* row, int cid, int m) {  // row[CS(cid) + m] without a run-time register index
  float v = m < CL(4) ? row[CS(4) + m] : 0.f;
  if (m < 3) {  // (pinned: otherwise the selects fold back into one load from a computed address, i.e. the row moves to scratch memory)
    float c0 = row[CS(0) + m], c1 = row[CS(1) + m], c2 = row[CS(2) + m], c3 = row[CS(3) + m];
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    v = cid == 0 ? c0 : (cid == 1 ? c1 : (cid == 2 ? c2 : (cid == 3 ? c3 : v)));
  }
  return v;
}
__device__ __forceinline__ void load_row(float* row, const float* fullrow, const Role& R, int k, const float* rhs /*LDS, dof order*/, float diag_add) {
  if (R.bl == 6) {
#pragma unroll
    for (int j = 0; j < NJ; j++) row[j] = rhs[6 + j];
#pragma unroll
    for (int m = 0; m < 6; m++) row[NJ + m] = rhs[m];
  } else {
#pragma unroll
    for (int j = 0; j < NJ; j++) row[j] = fullrow[6 + j] + ((R.isjoint && j == k) ? diag_add : 0.f);
#pragma unroll
    for (int m = 0; m < 6; m++) row[NJ + m] = fullrow[m] + (m == R.bl ? diag_add : 0.f);
  }
  row[NVT] = 0.f;
}

// Solve A x = rhs for the tree-structured matrix with three LDS exchanges instead of one per pivot: a lane on its own SIMD pays every LDS
// round trip and barrier in full, so the small dense blocks are factorised redundantly in registers by every lane that needs them.
//   1. chain lanes publish their rows of the chain blocks; every joint lane factorises its own chain's block (legs padded to 7 x 7 with the
//      identity), the base lanes and the rhs lane all five, and push their 19 chain-column entries through those factors;
//   2. base lanes publish those entries, take the Schur complement against the earlier base rows and publish the reduced 6 x 6 system + rhs;
//   3. every lane factorises the reduced system and solves it (base solution in every lane); every joint lane back-substitutes its own chain.
// Returns the lane's own entry of x; xb = the six base entries.
__device__ __forceinline__ float tree_cholesky_solve(float* row, RS4& S, const Role& R, int k, float* xb PA_PARAM) {
  PHS_DECL
  if (R.isjoint) {
#pragma unroll
    for (int m = 0; m < 7; m++) { const float v = chain_entry(row, R.cid, m); if (m <= R.cdepth) S.Hc[k][m] = v; }
  }
  __syncthreads();
  const int fcs = R.isjoint ? R.cstart : CS(4), flen = R.isjoint ? R.clen : 7;
  float L[28];
#pragma unroll
  for (int pp = 0; pp < 7; pp++) {
    const float* hr = S.Hc[fcs + pp];
#pragma unroll
    for (int m = 0; m <= pp; m++) L[tri4(pp, m)] = pp < flen ? hr[m] : (m == pp ? 1.f : 0.f);
  }
  chol_packed<7>(L);
  PHS(10)
  if (R.bl >= 0) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float Lg[6];
#pragma unroll
      for (int pp = 0; pp < 3; pp++) {
#pragma unroll
        for (int m = 0; m <= pp; m++) Lg[tri4(pp, m)] = S.Hc[CS(c) + pp][m];
      }
      chol_packed<3>(Lg);
      fwd_packed<3>(row + CS(c), Lg);
    }
    fwd_packed<7>(row + CS(4), L);
#pragma unroll
    for (int j = 0; j < NJ; j++) { S.Lb[R.bl][j] = row[j]; S.LbT[j][R.bl] = row[j]; }
#pragma unroll
    for (int m = 0; m < 6; m++) S.Lb[R.bl][NJ + m] = row[NJ + m];
  }
  __syncthreads();
  {  // Schur complement: the 21 + 6 pairs (b' >= b) of base rows / rhs row, one 19-term dot product per lane
    const int pl = threadIdx.x & 31;
    const int bp = pl >= 21 ? 6 : (pl >= 15 ? 5 : (pl >= 10 ? 4 : (pl >= 6 ? 3 : (pl >= 3 ? 2 : (pl >= 1 ? 1 : 0)))));
    const int bq = pl >= 21 ? pl - 21 : pl - bp * (bp + 1) / 2;
    if (pl < 27) {
      const float* ra = S.Lb[bp]; const float* rb = S.Lb[bq];
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; j++) d = fmaf(ra[j], rb[j], d);
      S.Sb[bp][bq] = ra[NJ + bq] - d;
    }
  }
  __syncthreads();
  PHS(11)
  {
    float B[21], yb[6];
#pragma unroll
    for (int pp = 0; pp < 6; pp++) {
#pragma unroll
      for (int m = 0; m <= pp; m++) B[tri4(pp, m)] = S.Sb[pp][m];
      yb[pp] = S.Sb[6][pp];
    }
    chol_packed<6>(B);
    fwd_packed<6>(yb, B);
#pragma unroll
    for (int m = 5; m >= 0; m--) {
      float sx = yb[m];
#pragma unroll
      for (int q = m + 1; q < 6; q++) sx -= B[tri4(q, m)] * xb[q];
      xb[m] = sx * B[tri4(m, m)];
    }
  }
  PHS(12)
  float x_own = 0.f;
#pragma unroll
  for (int b = 0; b < 6; b++) x_own = R.bl == b ? xb[b] : x_own;
  if (R.isjoint) {
    float xc[7];
#pragma unroll
    for (int m = 0; m < 7; m++) {  // rhs of the chain's triangular system: y minus the base part
      const float* t = S.LbT[R.cstart + m];
      float z = t[6];
#pragma unroll
      for (int b = 0; b < 6; b++) z -= t[b] * xb[b];
      xc[m] = m < R.clen ? z : 0.f;
    }
#pragma unroll
    for (int m = 6; m >= 0; m--) {
      float sx = xc[m];
#pragma unroll
      for (int q = m + 1; q < 7; q++) sx -= L[tri4(q, m)] * xc[q];
      xc[m] = sx * L[tri4(m, m)];
    }
#pragma unroll
    for (int tt = 0; tt < 7; tt++) x_own = R.cdepth == tt ? xc[tt] : x_own;
  }
  __syncthreads();
  PHS(13)
  return x_own;
}

// contact-frame 3-vector J_c v for a dof vector v in LDS (dof order); cs = first joint of the contact's chain
__device__ __forceinline__ void jac_mul(float* o, const float* Jc, const float* v, int cs) {
  o[0] = o[1] = o[2] = 0.f;
#pragma unroll
  for (int b = 0; b < 6; b++) { const float w = v[b]; o[0] = fmaf(Jc[3 * b], w, o[0]); o[1] = fmaf(Jc[3 * b + 1], w, o[1]); o[2] = fmaf(Jc[3 * b + 2], w, o[2]); }
  const float* vc = v + 6 + cs;
#pragma unroll
  for (int m = 0; m < 7; m++) { const float w = vc[m]; o[0] = fmaf(Jc[JC + 3 * m], w, o[0]); o[1] = fmaf(Jc[JC + 3 * m + 1], w, o[1]); o[2] = fmaf(Jc[JC + 3 * m + 2], w, o[2]); }
}

// dot of a register row (dof order) with a broadcast LDS vector
__device__ __forceinline__ float dot_row(const float* Mrow, const float* v) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NVT; j++) s = fmaf(Mrow[j], v[j], s);
  return s;
}


__global__ __launch_bounds__(WAVE, 1) void k_tree_v4(const float* __restrict__ gF, const int* __restrict__ gI, int nF, int nI, const float* state_in, int ld_in,
                                                    const float* __restrict__ ctrl, float* __restrict__ warm, int N, int substeps, float* state_out, int ld_out,
                                                    float* __restrict__ sensors_out, int ld_sens, int* __restrict__ stats, int dshift) {
  __shared__ RS4 sRS[RPW];
  __shared__ __attribute__((aligned(16))) float sF[SF_MAX];  // the model image, shared by the rollouts of the wave
  __shared__ int sI[SI_MAX];
  const int lane = threadIdx.x, l = lane & 31, r = lane >> 5;
  RS4& S = sRS[r];
  for (int i = lane; i < nF && i < SF_MAX; i += WAVE) sF[i] = gF[i];  // (the sensor records at the end of the image are read from global memory, once per launch)
  for (int i = lane; i < nI && i < SI_MAX; i += WAVE) sI[i] = gI[i];
  __syncthreads();
  // (latency mode, jh_internal.h: with dshift = 1 both rows of the wave compute the same rollout and the first writes -- the shipped 24-rollout plans)
  const int n = (blockIdx.x << (1 - dshift)) + (r >> dshift);
  const bool live = n < N && (r & ((1 << dshift) - 1)) == 0;
  const int nc = n < N ? n : N - 1;
  const int nj = NJ, ng = sI[1];
  const bool isbase = l < 6, isjoint = l >= 6 && l < 6 + nj, hasdof = isbase || isjoint, isbody = l == 0 || isjoint;
  const int k = isjoint ? l - 6 : 0;                   // own joint
  const int bidx = isjoint ? 1 + k : 0;                // own body
  const float* jf = sF + TH_F + k * TD_F;
  const int* ji = sI + TH_I + k * TD_I;
  Role R;
  R.isjoint = isjoint; R.cstart = isjoint ? ji[1] : 0; R.cdepth = isjoint ? ji[2] : -1;
  R.cid = R.cstart < 12 ? R.cstart / 3 : 4; R.clen = R.cid < 4 ? 3 : 7;
  R.bl = l < 6 ? l : (l == NVT ? 6 : -1);
  const int cstart = R.cstart, cdepth = R.cdepth;
  const int oGF = TH_F + nj * TD_F, oGI = TH_I + nj * TD_I;
  const float h = sF[TF_DT], impratio = sF[TF_IMPRATIO], tol = sF[TF_TOL], lstol = sF[TF_LSTOL]; const int cap = (int)sF[TF_MAXITER];
  const float grav[3] = {sF[TF_GRAV], sF[TF_GRAV + 1], sF[TF_GRAV + 2]};
  const float pln[3] = {sF[TF_PLANE_N], sF[TF_PLANE_N + 1], sF[TF_PLANE_N + 2]}, plp[3] = {sF[TF_PLANE_P], sF[TF_PLANE_P + 1], sF[TF_PLANE_P + 2]};
  const float en = isjoint ? 1.f : 0.f;
  const float c_damp = jf[JF_DAMP] * en, c_arm = jf[JF_ARM] * en, c_fl = jf[JF_FL] * en, c_fB = jf[JF_FB], c_fD = jf[JF_FD], c_invw = jf[JF_INVW];
  const float c_limited = jf[JF_LIMITED] * en, c_lo = jf[JF_LO], c_hi = jf[JF_HI], c_lK = jf[JF_LK], c_lB = jf[JF_LB];
  float c_si[5]; for (int i = 0; i < 5; i++) c_si[i] = jf[JF_SOLIMP + i];
  const bool hasact = isjoint && ji[3] != 0;
  const float c_kp = hasact ? jf[JF_KP] : 0.f, c_kv = hasact ? jf[JF_KV] : 0.f, c_clim = jf[JF_CLIM], c_clo = jf[JF_CLO], c_chi = jf[JF_CHI];
  const float c_flim = hasact ? jf[JF_FLIM] : 0.f, c_flo = jf[JF_FLO], c_fhi = jf[JF_FHI];
  // ---- state: replicated base + own joint
  float qb[7], vb[6], q = 0.f, qd = 0.f, qws = 0.f;
  {
    const float* xi = state_in + (size_t)nc * ld_in;  // ld_in = 0: one state for every rollout
    for (int i = 0; i < 7; i++) qb[i] = xi[i];
    for (int i = 0; i < 6; i++) vb[i] = xi[NQ + i];
    if (isjoint) { q = xi[7 + k]; qd = xi[NQ + 6 + k]; }
    if (hasdof && warm) qws = warm[(size_t)nc * NVT + l];
  }
  const float u = hasact ? ctrl[(size_t)nc * NJ + k] : 0.f;
  int n_iters = 0, n_maxed = 0;
  PH_DECL

  for (int step = 0; step < substeps; step++) {
    // ================================================================ joint transforms (one sincos per joint), velocities
    float Rq[9];
    {
      rodrigues4(Rq, jf + JF_AXIS, q);
      if (isjoint) {
        float Tl[9]; mulMM(Tl, jf + JF_LR, Rq);
        float* o = S.Tl[k];
        o[0] = jf[JF_LPOS]; o[1] = jf[JF_LPOS + 1]; o[2] = jf[JF_LPOS + 2];
        for (int i = 0; i < 9; i++) o[3 + i] = Tl[i];
      }
      float myqd = qd;
#pragma unroll
      for (int i = 0; i < 6; i++) if (i == l) myqd = vb[i];
      if (hasdof) S.qd[l] = myqd;
      if (l == 0) S.ncon = 0;
    }
    __syncthreads();
    // ================================================================ kinematics (positions relative to the base origin)
    float Rb[9], Rown[9], pown[3] = {0.f, 0.f, 0.f}, Sown[6] = {0, 0, 0, 0, 0, 0};
    {
      const float nn = rsqrtf(qb[3] * qb[3] + qb[4] * qb[4] + qb[5] * qb[5] + qb[6] * qb[6]);
      qb[3] *= nn; qb[4] *= nn; qb[5] *= nn; qb[6] *= nn;
      quat2mat(Rb, qb + 3);
      float P[3] = {0.f, 0.f, 0.f}, Rm[9];
      for (int i = 0; i < 9; i++) { Rm[i] = Rb[i]; Rown[i] = Rb[i]; }
#pragma unroll
      for (int t = 0; t < MAXD - 1; t++) {
        if (t < cdepth) {  // ancestors in the own chain
          const float* T = S.Tl[cstart + t];
          float lp[3] = {T[0], T[1], T[2]}, Tl[9], P2[3], R2[9];
          for (int i = 0; i < 9; i++) Tl[i] = T[3 + i];
          mulMV(P2, Rm, lp); mulMM(R2, Rm, Tl);
          for (int i = 0; i < 3; i++) P[i] += P2[i];
          for (int i = 0; i < 9; i++) Rm[i] = R2[i];
        }
      }
      if (isjoint) {
        float lp[3] = {jf[JF_LPOS], jf[JF_LPOS + 1], jf[JF_LPOS + 2]}, la[3] = {jf[JF_AXIS], jf[JF_AXIS + 1], jf[JF_AXIS + 2]}, P2[3], R0[9], axw[3];
        mulMV(P2, Rm, lp); for (int i = 0; i < 3; i++) pown[i] = P[i] + P2[i];
        mulMM(R0, Rm, jf + JF_LR);
        mulMV(axw, R0, la);
        mulMM(Rown, R0, Rq);
        float lin[3]; cross3(lin, pown, axw);  // velocity of the reference point under unit joint rate: anchor x axis
        for (int i = 0; i < 3; i++) { Sown[i] = axw[i]; Sown[3 + i] = lin[i]; }
      } else if (l < 3) Sown[3 + l] = 1.f;
      else if (l < 6) { Sown[0] = Rb[l - 3]; Sown[1] = Rb[3 + l - 3]; Sown[2] = Rb[6 + l - 3]; }
      if (hasdof) for (int i = 0; i < 6; i++) S.Sax[l][i] = Sown[i];
      if (isbody) { for (int i = 0; i < 3; i++) S.xpos[bidx][i] = pown[i]; for (int i = 0; i < 9; i++) S.xR[bidx][i] = Rown[i]; }
    }
    PH(0)
    // ================================================================ body spatial inertia; composite inertias = suffix sums along the chains
    float Ib[10], Ic[10];
    {
      const float* bm = isjoint ? jf + JF_MASS : sF + TF_BMASS;  // mass, ipos(3), iR(9), inertia(3) in both records
      const float mass = isbody ? bm[0] : 0.f;
      float lip[3] = {bm[1], bm[2], bm[3]}, lir[9], di[3] = {bm[13], bm[14], bm[15]};
      for (int i = 0; i < 9; i++) lir[i] = bm[4 + i];
      float Rk[9], c[3]; mulMM(Rk, Rown, lir); mulMV(c, Rown, lip);
      for (int i = 0; i < 3; i++) c[i] += pown[i];
      const float cc = dot3(c, c);
      Ib[0] = mass; Ib[1] = mass * c[0]; Ib[2] = mass * c[1]; Ib[3] = mass * c[2];
      const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
      for (int e = 0; e < 6; e++) {
        float v = 0.f;
        for (int t = 0; t < 3; t++) v += Rk[ia[e] * 3 + t] * di[t] * Rk[ib[e] * 3 + t];
        Ib[4 + e] = isbody ? v + mass * ((ia[e] == ib[e] ? cc : 0.f) - c[ia[e]] * c[ib[e]]) : 0.f;
      }
      if (isbody) for (int i = 0; i < 10; i++) S.Ib[bidx][i] = Ib[i];
    }
    __syncthreads();
    // ================================================================ sensors of the step that produces the returned state (mjData.sensordata after mj_step holds
    // the values of that step's forward pass, i.e. of the state before its integration): site positions / frame axes, one sensor per lane
    if (sensors_out && step == substeps - 1 && live && l < sI[4]) {
      const int oSF = oGF + ng * TG_F, oSI = oGI + ng * TG_I;
      const float* sf = gF + oSF + l * TS_F; const int* si = gI + oSI + l * TS_I;
      const int kind = si[0], owner = si[1], adr = si[2], hasref = si[3];
      float o[3] = {sf[0], sf[1], sf[2]};
      if (owner > -2) {
        const int b = owner < 0 ? 0 : 1 + owner;
        float Rs[9]; for (int i = 0; i < 9; i++) Rs[i] = S.xR[b][i];
        if (kind == 0) { float w[3]; mulMV(w, Rs, o); for (int i = 0; i < 3; i++) o[i] = w[i] + S.xpos[b][i] + qb[i]; }
        else col3(o, Rs, kind - 1);
      }
      if (hasref) {  // position in the frame of a world-fixed reference site: R_ref' (p - p_ref)
        const float dv[3] = {o[0] - sf[3], o[1] - sf[4], o[2] - sf[5]};
        for (int i = 0; i < 3; i++) o[i] = sf[6 + i] * dv[0] + sf[9 + i] * dv[1] + sf[12 + i] * dv[2];
      }
      float* out = sensors_out + (size_t)n * ld_sens + adr;
      out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
    }
    {
      float tot[10];  // whole robot: a reduction over the body lanes (two DPP rows), no LDS
#pragma unroll
      for (int i = 0; i < 10; i++) { tot[i] = gsum32(Ib[i]); Ic[i] = isbase ? tot[i] : 0.f; }
      if (isjoint) {
#pragma unroll
        for (int t = MAXD - 1; t >= 0; t--)
          if (t >= cdepth && t < R.clen) { const float* o = S.Ib[1 + cstart + t]; for (int i = 0; i < 10; i++) Ic[i] += o[i]; }
      }
    }
    // ================================================================ inertia rows (mj_crb) and bias forces (mj_rne with gravity as base acceleration)
    float Mrow[NVT];
#pragma unroll
    for (int j = 0; j < NVT; j++) Mrow[j] = 0.f;
    if (hasdof) {
      float f[6];
      inertia6_mul(f, Ic, Sown);
#pragma unroll
      for (int j = 0; j < 6; j++) { float sj[6]; for (int i = 0; i < 6; i++) sj[i] = S.Sax[j][i]; Mrow[j] = dot6(f, sj); }   // base columns (base rows: all six by symmetry)
      if (isjoint) {
#pragma unroll
        for (int t = 0; t < MAXD; t++) {
          float v = 0.f;
          if (t <= cdepth) { float sj[6]; for (int i = 0; i < 6; i++) sj[i] = S.Sax[6 + cstart + t][i]; v = dot6(f, sj) + (t == cdepth ? c_arm : 0.f); }
          // scatter to column 6 + cstart + t without a run-time register index
#pragma unroll
          for (int c = 0; c < NCH; c++) if (t < CL(c)) Mrow[6 + CS(c) + t] = (R.cid == c && t <= cdepth) ? v : Mrow[6 + CS(c) + t];
        }
      }
    }
    if (hasdof) {  // publish the lower triangle; the descendants' columns and the base rows' joint columns come back mirrored
#pragma unroll
      for (int j = 0; j < NVT; j++) S.M[l][j] = Mrow[j];
    }
    __syncthreads();
    if (hasdof) {
      if (isbase) {
#pragma unroll
        for (int j = 6; j < NVT; j++) Mrow[j] = S.M[j][l];
      } else {
#pragma unroll
        for (int t = 0; t < MAXD; t++) {
          float v = 0.f;
          if (t > cdepth && t < R.clen) v = S.M[6 + cstart + t][l];
#pragma unroll
          for (int c = 0; c < NCH; c++) if (t < CL(c)) Mrow[6 + CS(c) + t] = (R.cid == c && t > cdepth) ? v : Mrow[6 + CS(c) + t];
        }
      }
    }
    float bias_own = 0.f;
    {
      float frc[6] = {0, 0, 0, 0, 0, 0};
      if (isbody) {
        float vel[6] = {0, 0, 0, 0, 0, 0}, acc[6] = {0, 0, 0, -grav[0], -grav[1], -grav[2]};
        for (int j = 0; j < 3; j++) vel[3 + j] += S.qd[j];  // translational axes are world-fixed unit vectors
        float Sd[3][6], sr[3][6];
        for (int j = 0; j < 3; j++) { for (int i = 0; i < 6; i++) sr[j][i] = S.Sax[3 + j][i]; crossm6(Sd[j], vel, sr[j]); }
        for (int j = 0; j < 3; j++) { const float w = S.qd[3 + j]; for (int i = 0; i < 6; i++) { acc[i] += Sd[j][i] * w; vel[i] += sr[j][i] * w; } }
#pragma unroll
        for (int t = 0; t < MAXD; t++) if (t <= cdepth) {
          const int j = 6 + cstart + t; float sj[6], sd[6]; for (int i = 0; i < 6; i++) sj[i] = S.Sax[j][i];
          crossm6(sd, vel, sj); const float w = S.qd[j];
          for (int i = 0; i < 6; i++) { acc[i] += sd[i] * w; vel[i] += sj[i] * w; }
        }
        float Ia[6], Iv[6], vIv[6];
        inertia6_mul(Ia, Ib, acc); inertia6_mul(Iv, Ib, vel); crossf6(vIv, vel, Iv);
        for (int i = 0; i < 6; i++) { frc[i] = Ia[i] + vIv[i]; S.frc[bidx][i] = frc[i]; }
      }
      __syncthreads();
      {  // wrench of the subtree the dof carries: the chain tail for a joint, every body for the base
        float Fs[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { const float tot = gsum32(frc[i]); Fs[i] = isbase ? tot : 0.f; }
        if (isjoint) {
#pragma unroll
          for (int t = MAXD - 1; t >= 0; t--) if (t >= cdepth && t < R.clen) { const float* o = S.frc[1 + cstart + t]; for (int i = 0; i < 6; i++) Fs[i] += o[i]; }
        }
        if (hasdof) bias_own = dot6(Sown, Fs);
      }
    }
    PH(1)
    // ================================================================ smooth force, unconstrained acceleration
    float fs_own = 0.f, a0_own = 0.f, Md_own = 1.f, kv_eff = 0.f;
    {
      float fa = 0.f;
      if (hasact) {
        float cc = u; if (c_clim != 0.f) cc = jh_clampf(cc, c_clo, c_chi);
        fa = c_kp * (cc - q) - c_kv * qd;
        kv_eff = c_kv;
        if (c_flim != 0.f) { if (fa <= c_flo || fa >= c_fhi) kv_eff = 0.f; fa = jh_clampf(fa, c_flo, c_fhi); }  // a saturated servo has no velocity derivative (implicitfast)
      }
      if (hasdof) fs_own = -c_damp * qd - bias_own + fa;
#pragma unroll
      for (int j = 0; j < NVT; j++) if (j == l) Md_own = Mrow[j];
      if (hasdof) S.vec[0][l] = fs_own;
      __syncthreads();
      float row[NR], xb[6];
      load_row(row, Mrow, R, k, S.vec[0], 0.f);
      a0_own = tree_cholesky_solve(row, S, R, k, xb PA_ARG);
    }
    PH(2)
    // ================================================================ collision: every robot geom against the plane
    if (l < ng) {
      const float* gf = sF + oGF + l * TG_F; const int owner = sI[oGI + l * TG_I], gtype = sI[oGI + l * TG_I + 1];
      const int gb = owner < 0 ? 0 : 1 + owner;
      const int och = owner < 0 ? 0 : 1 + (sI[TH_I + owner * TD_I + 1] < 12 ? sI[TH_I + owner * TD_I + 1] / 3 : 4);  // 0: base geom, 1 + chain otherwise
      float bR[9], gp[3], lp[3] = {gf[GF4_POS], gf[GF4_POS + 1], gf[GF4_POS + 2]};
      for (int i = 0; i < 9; i++) bR[i] = S.xR[gb][i];
      mulMV(gp, bR, lp); for (int i = 0; i < 3; i++) gp[i] += S.xpos[gb][i];
      const float pr[3] = {plp[0] - qb[0], plp[1] - qb[1], plp[2] - qb[2]};  // plane point relative to the base origin
      auto push = [&](const float* pos, float dist, const float* tng) __attribute__((always_inline)) {
        const int i = atomicAdd(&S.ncon, 1);
        if (i >= NCP) { if (stats) atomicAdd(stats, 1); return; }
        float* e = S.raw[i];
        e[0] = pos[0]; e[1] = pos[1]; e[2] = pos[2]; e[3] = dist; e[4] = __int_as_float(l | (och << 8)); e[5] = tng[0]; e[6] = tng[1]; e[7] = tng[2];
      };
      const float zero3[3] = {0.f, 0.f, 0.f};
      float lr[9], gR[9];
      for (int i = 0; i < 9; i++) lr[i] = gf[GF4_R + i];
      mulMM(gR, bR, lr);
      if (gtype == 2 || gtype == 3) {  // sphere, or the two end spheres of a capsule (+ end first)
        float axis[3] = {0.f, 0.f, 0.f};
        const float rad = gf[GF4_SIZE], half = gtype == 3 ? gf[GF4_SIZE + 1] : 0.f;
        if (gtype == 3) col3(axis, gR, 2);
        for (int e = 0; e < (gtype == 3 ? 2 : 1); e++) {
          const float sg = e == 0 ? 1.f : -1.f;
          const float c[3] = {gp[0] + sg * half * axis[0], gp[1] + sg * half * axis[1], gp[2] + sg * half * axis[2]};
          const float dif[3] = {c[0] - pr[0], c[1] - pr[1], c[2] - pr[2]};
          const float dist = dot3(dif, pln) - rad;
          if (dist <= 0.f) { const float pos[3] = {c[0] - pln[0] * (rad + 0.5f * dist), c[1] - pln[1] * (rad + 0.5f * dist), c[2] - pln[2] * (rad + 0.5f * dist)}; push(pos, dist, gtype == 3 ? axis : zero3); }
        }
      } else {  // box: corners in MuJoCo's order, at most 4 contacts
        const float hs[3] = {gf[GF4_SIZE], gf[GF4_SIZE + 1], gf[GF4_SIZE + 2]};
        const float dif[3] = {gp[0] - pr[0], gp[1] - pr[1], gp[2] - pr[2]};
        const float dist0 = dot3(dif, pln);
        int cnt = 0;
        for (int i = 0; i < 8; i++) {
          const float vl[3] = {(i & 1) ? hs[0] : -hs[0], (i & 2) ? hs[1] : -hs[1], (i & 4) ? hs[2] : -hs[2]};
          float vec3[3]; mulMV(vec3, gR, vl);
          const float d = dist0 + dot3(pln, vec3);
          if (d <= 0.f && cnt < 4) { const float pos[3] = {gp[0] + vec3[0] - pln[0] * 0.5f * d, gp[1] + vec3[1] - pln[1] * 0.5f * d, gp[2] + vec3[2] - pln[2] * 0.5f * d}; push(pos, d, zero3); cnt++; }
        }
      }
    }
    __syncthreads();
    // ================================================================ constraint rows
    const int ncon = S.ncon < NCP ? S.ncon : NCP;
    Slot4 sl;
    sl.valid = l < ncon; sl.D = 0.f; sl.mu = 0.f;
    int my_cs = 0;  // first joint of the own contact's chain
    if (sl.valid) { const int och = __float_as_int(S.raw[l][4]) >> 8; my_cs = och == 0 ? 0 : (och <= 4 ? 3 * (och - 1) : CS(4)); }
    for (int w = 0; w < 3; w++) sl.aref[w] = sl.jar[w] = sl.jp[w] = 0.f;
    if (sl.valid) {  // contact frame: plane normal (from the plane, geom 1, to the robot geom), second axis along a capsule's axis
      float fr[9] = {pln[0], pln[1], pln[2], 0, 0, 0, 0, 0, 0};
      make_frame(fr);
      const float* e = S.raw[l];
      float y[3] = {e[5], e[6], e[7]};
      const float dp = dot3(fr, y); y[0] -= fr[0] * dp; y[1] -= fr[1] * dp; y[2] -= fr[2] * dp;
      const float nn = sqrtf(dot3(y, y));
      if (nn > 0.5e-3f) { for (int i = 0; i < 3; i++) fr[3 + i] = y[i] / nn; cross3(fr + 6, fr, fr + 3); }
      for (int w = 0; w < 9; w++) S.fW[l][w] = fr[w];
    }
    __syncthreads();
    for (int c = 0; c < ncon; c++) {  // Jacobian: every dof lane its own column (robot side only: the plane is static); the spare lanes clear the padding
      const float* e = S.raw[c];
      const int och = __float_as_int(e[4]) >> 8, odepth = och == 0 ? -1 : sI[TH_I + sI[oGI + (__float_as_int(e[4]) & 255) * TG_I] * TD_I + 2];
      if (isbase || (isjoint && och == 1 + R.cid)) {
        const float pos[3] = {e[0], e[1], e[2]};
        float col[3] = {0.f, 0.f, 0.f};
        if (isbase || cdepth <= odepth) {
          float v[3]; cross3(v, Sown, pos); for (int i = 0; i < 3; i++) v[i] += Sown[3 + i];
          const float* fr = S.fW[c];
          col[0] = dot3(fr, v); col[1] = dot3(fr + 3, v); col[2] = dot3(fr + 6, v);
        }
        float* o = S.J[c] + (isbase ? 3 * l : JC + 3 * cdepth);
        o[0] = col[0]; o[1] = col[1]; o[2] = col[2];
      } else if (l >= NVT) {  // positions the contact's chain does not have (legs: 3..6; base geoms: all)
        const int m = l - NVT, len = och == 0 ? 0 : (och < 5 ? 3 : 7);
        if (m >= len) { float* o = S.J[c] + JC + 3 * m; o[0] = o[1] = o[2] = 0.f; }
      }
    }
    __syncthreads();
    if (sl.valid) {
      const float* e = S.raw[l]; const float dist = e[3]; const int gid = __float_as_int(e[4]) & 255;
      const float* gf = sF + oGF + gid * TG_F;
      float si[5]; for (int w = 0; w < 5; w++) si[w] = gf[GF4_SOLIMP + w];
      const float mu = gf[GF4_MU], imp = impedance(si, dist);
      const float R0 = fmaxf(1e-15f, (1.f - imp) / imp * gf[GF4_TRAN] * (1.f + mu * mu));
      const float Rpy = fmaxf(1e-15f, 2.f * (mu * mu / fmaxf(1e-15f, impratio)) * R0);
      sl.D = 1.f / Rpy; sl.mu = mu;
      float vel[3];
      jac_mul(vel, S.J[l], S.qd, my_cs);
      sl.aref[0] = -gf[GF4_B] * vel[0] - gf[GF4_K] * imp * dist; sl.aref[1] = -gf[GF4_B] * vel[1]; sl.aref[2] = -gf[GF4_B] * vel[2];
    }
    DofRows4 dr;
    dr.fl = c_fl; dr.fD = c_fD; dr.fR = c_fD > 0.f ? 1.f / c_fD : 0.f; dr.faref = -c_fB * qd; dr.lims = 0.f; dr.laref = 0.f; dr.lD = 0.f; dr.jf = dr.jl = dr.pf = dr.pl = 0.f;
    if (c_limited != 0.f) {
      const float dlo = q - c_lo, dhi = c_hi - q, dist = fminf(dlo, dhi);
      if (dist < 0.f) {
        const float sg = dlo < dhi ? 1.f : -1.f, imp = impedance(c_si, dist), Rr = fmaxf(1e-15f, (1.f - imp) / imp * c_invw);
        dr.lims = sg; dr.lD = 1.f / Rr; dr.laref = -c_lB * (sg * qd) - c_lK * imp * dist;
      }
    }
    PH(3)
    // ================================================================ Newton solver (tree-structured Hessian, one row per lane)
    float a_own = a0_own;
    unsigned involved = 0;  // bit c: contact c moves with this lane's dof (base lanes: every contact; joint lanes: contacts on their chain)
    if (hasdof) for (int c = 0; c < ncon; c++) involved |= (isbase || (__float_as_int(S.raw[c][4]) >> 8) == 1 + R.cid) ? (1u << c) : 0u;
    const int own_col = isbase ? 3 * l : JC + 3 * (cdepth < 0 ? 0 : cdepth);  // own column in a compact Jacobian row
    const float iMd = 1.f / Md_own;
    const float snorm = gsum32(hasdof ? fs_own * fs_own * iMd : 0.f);
    int iters_this = 0;
    {
      // ---- warm start: the better of last step's acceleration and the unconstrained one
      if (hasdof) { S.vec[0][l] = qws; S.vec[1][l] = a0_own; S.vec[2][l] = qws - a0_own; }
      __syncthreads();
      float jar_ws[3] = {0.f, 0.f, 0.f};
      if (sl.valid) { float o[3]; jac_mul(o, S.J[l], S.vec[0], my_cs); for (int w = 0; w < 3; w++) sl.jar[w] = o[w] - sl.aref[w]; }
      dr.jf = qws - dr.faref; dr.jl = dr.lims * qws - dr.laref;
      const float mdw = dot_row(Mrow, S.vec[2]);
      const float cost_ws = gsum32(lane_cost4(sl, dr) + (hasdof ? 0.5f * (qws - a0_own) * mdw : 0.f));
      for (int w = 0; w < 3; w++) jar_ws[w] = sl.jar[w];
      const float jf_ws = dr.jf, jl_ws = dr.jl;
      if (sl.valid) { float o[3]; jac_mul(o, S.J[l], S.vec[1], my_cs); for (int w = 0; w < 3; w++) sl.jar[w] = o[w] - sl.aref[w]; }
      dr.jf = a0_own - dr.faref; dr.jl = dr.lims * a0_own - dr.laref;
      const float cost_0 = gsum32(lane_cost4(sl, dr));
      if (cost_ws < cost_0) { a_own = qws; for (int w = 0; w < 3; w++) sl.jar[w] = jar_ws[w]; dr.jf = jf_ws; dr.jl = jl_ws; }
      __syncthreads();
      bool act = gor32((int)(sl.valid || dr.fl > 0.f || dr.lims != 0.f)) != 0;
      if (!act) a_own = a0_own;
      for (int it = 0; it < cap && __any(act); it++) {
        PH(4)
        // ---- (1) gradient row
        const float da_own = a_own - a0_own;
        if (hasdof) S.vec[0][l] = da_own;
        if (sl.valid) { float f[3], Wm[6]; pyramid_eval(sl.jar, sl.D, sl.mu, f, Wm); float* o = S.fW[l]; o[0] = f[0]; o[1] = f[1]; o[2] = f[2]; for (int w = 0; w < 6; w++) o[4 + w] = Wm[w]; }
        __syncthreads();
        float g_own = dot_row(Mrow, S.vec[0]), hd = 0.f;
        if (dr.fl > 0.f) {
          const float x = dr.jf, fl = dr.fl, lim = dr.fR * fl;
          if (x <= -lim) g_own -= fl; else if (x >= lim) g_own += fl; else { g_own += dr.fD * x; hd += dr.fD; }
        }
        if (dr.lims != 0.f && dr.jl < 0.f) { g_own += dr.lims * dr.lD * dr.jl; hd += dr.lD; }
        for (int c = 0; c < ncon; c++) {  // branch-free: a lane the contact does not move reads some other column and multiplies it by zero
          const float* jc = S.J[c] + own_col; const float* fc = S.fW[c];
          const float on = (involved >> c) & 1u ? 1.f : 0.f;
          g_own -= on * (jc[0] * fc[0] + jc[1] * fc[1] + jc[2] * fc[2]);
        }
        // ---- (2) convergence; leave before any Hessian work once both rollouts of the wave are done
        const float gn = gsum32(hasdof ? g_own * g_own * iMd : 0.f);
        if (act && gn <= tol * tol * fmaxf(snorm, 1e-12f)) act = false;
        if (!__any(act)) break;
        if (act) iters_this++;
        PH(5)
        // ---- (3) Hessian row in the factorisation layout; the rhs lane takes -g
        if (hasdof) S.vec[1][l] = -g_own;
        __syncthreads();
        float row[NR];
        load_row(row, Mrow, R, k, S.vec[1], hd);
        for (int c = 0; c < ncon; c++) {  // branch-free accumulation of J' W J: the row of the compact Jacobian is the same for every lane
          const float* fc = S.fW[c]; const float* Jc = S.J[c];
          const float on = (involved >> c) & 1u ? 1.f : 0.f;
          const float j0 = on * Jc[own_col], j1 = on * Jc[own_col + 1], j2 = on * Jc[own_col + 2];
          const float G0 = fc[4] * j0 + fc[5] * j1 + fc[7] * j2, G1 = fc[5] * j0 + fc[6] * j1 + fc[8] * j2, G2 = fc[7] * j0 + fc[8] * j1 + fc[9] * j2;
          float dch[7], dba[6];
#pragma unroll
          for (int m = 0; m < 7; m++) dch[m] = Jc[JC + 3 * m] * G0 + Jc[JC + 3 * m + 1] * G1 + Jc[JC + 3 * m + 2] * G2;
#pragma unroll
          for (int m = 0; m < 6; m++) dba[m] = Jc[3 * m] * G0 + Jc[3 * m + 1] * G1 + Jc[3 * m + 2] * G2;
          const int och = __float_as_int(S.raw[c][4]) >> 8;  // 0: the geom sits on the base (its chain part is zero), else 1 + chain
#pragma unroll
          for (int ch = 0; ch < NCH; ch++) {
            const float sel = och == 1 + ch ? 1.f : 0.f;
#pragma unroll
            for (int m = 0; m < CL(ch); m++) row[CS(ch) + m] = fmaf(sel, dch[m], row[CS(ch) + m]);
          }
#pragma unroll
          for (int m = 0; m < 6; m++) row[NJ + m] += dba[m];
        }
        PH(6)
        // ---- (4) factorise and solve; the direction goes back through LDS
        float xb[6];
        const float p_own = tree_cholesky_solve(row, S, R, k, xb PA_ARG);
        if (hasdof) S.vec[2][l] = p_own;
        __syncthreads();
        PH(7)
        // ---- (5) exact line search
        const float Mp_own = dot_row(Mrow, S.vec[2]);
        const float pMp = gsum32(hasdof ? p_own * Mp_own : 0.f), pMd = gsum32(hasdof ? Mp_own * da_own : 0.f), gp = gsum32(hasdof ? g_own * p_own : 0.f);
        if (act && !(gp < 0.f)) act = false;
        if (sl.valid) jac_mul(sl.jp, S.J[l], S.vec[2], my_cs);
        dr.pf = p_own; dr.pl = dr.lims * p_own;
        float lo = 0.f, hi = -1.f, alpha = 1.f; bool lsact = act;
        for (int ls = 0; ls < 12 && __any(lsact); ls++) {
          float d1, d2;
          lane_dir4(sl, dr, alpha, &d1, &d2);
          d1 = gsum32(d1) + pMd + alpha * pMp; d2 = gsum32(d2) + pMp;
          if (lsact) {
            if (fabsf(d1) <= lstol * fabsf(gp)) lsact = false;
            else {
              if (d1 < 0.f) lo = alpha; else hi = alpha;
              float nx = alpha - d1 * __frcp_rn(d2);
              if (hi < 0.f) { if (nx <= lo) nx = 2.f * alpha; }
              else if (nx <= lo || nx >= hi) nx = 0.5f * (lo + hi);
              alpha = nx;
            }
          }
        }
        PH(8)
        // ---- (6) step
        if (act) {
          a_own += alpha * p_own;
          for (int w = 0; w < 3; w++) sl.jar[w] += alpha * sl.jp[w];
          dr.jf += alpha * dr.pf; dr.jl += alpha * dr.pl;
          if (-gp * alpha <= tol * tol * fmaxf(snorm, 1e-12f)) act = false;
        }
        __syncthreads();
      }
      if (l == 0) { n_iters += iters_this; n_maxed += (iters_this >= cap); }
    }
    PH(4)
    // ================================================================ implicitfast integration: (M + h diag(d + kv)) qacc = fs + M (a - a0)
    {
      __syncthreads();
      const float da_own = a_own - a0_own;
      if (hasdof) S.vec[0][l] = da_own;
      __syncthreads();
      const float rhs_own = fs_own + dot_row(Mrow, S.vec[0]);
      if (hasdof) S.vec[1][l] = rhs_own;
      __syncthreads();
      float row[NR], x[6];
      load_row(row, Mrow, R, k, S.vec[1], h * (c_damp + kv_eff));
      const float qacc = tree_cholesky_solve(row, S, R, k, x PA_ARG);
      if (isjoint) { qd = fmaf(h, qacc, qd); q = fmaf(h, qd, q); }
      qws = a_own;
      for (int i = 0; i < 6; i++) vb[i] = fmaf(h, x[i], vb[i]);  // free base: every lane integrates the replicated state
      for (int i = 0; i < 3; i++) qb[i] = fmaf(h, vb[i], qb[i]);
      const float wn = sqrtf(vb[3] * vb[3] + vb[4] * vb[4] + vb[5] * vb[5]), ang = wn * h;
      if (ang > 0.f) {
        float sn, cs; sincosf(0.5f * ang, &sn, &cs); const float kk = sn / wn;
        float dq[4] = {cs, vb[3] * kk, vb[4] * kk, vb[5] * kk}, *qq = qb + 3;
        const float r0 = qq[0] * dq[0] - qq[1] * dq[1] - qq[2] * dq[2] - qq[3] * dq[3];
        const float r1 = qq[0] * dq[1] + qq[1] * dq[0] + qq[2] * dq[3] - qq[3] * dq[2];
        const float r2 = qq[0] * dq[2] - qq[1] * dq[3] + qq[2] * dq[0] + qq[3] * dq[1];
        const float r3 = qq[0] * dq[3] + qq[1] * dq[2] - qq[2] * dq[1] + qq[3] * dq[0];
        qq[0] = r0; qq[1] = r1; qq[2] = r2; qq[3] = r3;
      }
      const float nn = rsqrtf(qb[3] * qb[3] + qb[4] * qb[4] + qb[5] * qb[5] + qb[6] * qb[6]);
      qb[3] *= nn; qb[4] *= nn; qb[5] *= nn; qb[6] *= nn;
    }
    __syncthreads();
    PH(9)
  }
  PH_FLUSH
  if (live) {
    float* o = state_out + (size_t)n * ld_out;
    if (isjoint) { o[7 + k] = q; o[NQ + 6 + k] = qd; }
    if (l < 7) o[l] = qb[l];
    if (l < 6) o[NQ + l] = vb[l];
    if (hasdof && warm) warm[(size_t)n * NVT + l] = qws;
    if (stats && l == 0) { if (n_maxed) atomicAdd(stats + 1, n_maxed); atomicAdd(stats + 2, n_iters); atomicAdd(stats + 3, substeps); }
  }
}

}  // namespace

struct jh_tree { float* d_f; int* d_i; int* d_stats; int nj, ng, nq, nv, nf, ni, ns; std::vector<hipEvent_t> events; };

extern "C" int jh_tree_create(const void* blob, size_t nbytes, jh_tree** out) {
  JH_REQUIRE(blob && out && nbytes >= 16, "tree_create: null or short blob");
  const unsigned* hd = (const unsigned*)blob;
  JH_REQUIRE(hd[0] == 0x34564A54u, "tree_create: bad magic");
  const size_t nf = hd[1], ni = hd[2];
  JH_REQUIRE(nbytes == 16 + 4 * (nf + ni), "tree_create: blob size mismatch");
  const float* f = (const float*)(hd + 4); const int* ii = (const int*)(f + nf);
  JH_REQUIRE(ii[0] == NJ && ii[2] == NQ && ii[3] == NVT && ii[1] <= G - 1, "tree_create: the kernel is instantiated for a free base + 19 hinges (got %d joints, nq %d, nv %d, %d geoms)", ii[0], ii[2], ii[3], ii[1]);
  JH_REQUIRE((size_t)(TH_F + ii[0] * TD_F + ii[1] * TG_F) <= (size_t)SF_MAX && (size_t)(TH_I + ii[0] * TD_I + ii[1] * TG_I) <= (size_t)SI_MAX,
             "tree_create: model image too large for the kernel's LDS copy");
  JH_REQUIRE(ii[4] >= 0 && ii[4] <= G && nf == (size_t)(TH_F + ii[0] * TD_F + ii[1] * TG_F + ii[4] * TS_F) && ni == (size_t)(TH_I + ii[0] * TD_I + ii[1] * TG_I + ii[4] * TS_I),
             "tree_create: image sizes do not match the counts in its header (or more than 32 sensors)");
  for (int c = 0; c < NCH; c++)
    for (int m = 0; m < CL(c); m++) {
      const int* jr = ii + TH_I + (CS(c) + m) * TD_I;
      JH_REQUIRE(jr[1] == CS(c) && jr[2] == m, "tree_create: the kernel is instantiated for four 3-joint chains followed by one 7-joint chain (joint %d: chain start %d, depth %d)", CS(c) + m, jr[1], jr[2]);
    }
  jh_tree* t = new jh_tree();
  t->nf = (int)nf; t->ni = (int)ni; t->ns = ii[5];
  t->nj = ii[0]; t->ng = ii[1]; t->nq = ii[2]; t->nv = ii[3];
  JH_HIP(hipMalloc(&t->d_f, 4 * nf)); JH_HIP(hipMalloc(&t->d_i, 4 * ni)); JH_HIP(hipMalloc(&t->d_stats, 64 * sizeof(int)));
  JH_HIP(hipMemcpy(t->d_f, f, 4 * nf, hipMemcpyHostToDevice)); JH_HIP(hipMemcpy(t->d_i, ii, 4 * ni, hipMemcpyHostToDevice));
  JH_HIP(hipMemset(t->d_stats, 0, 64 * sizeof(int)));
  *out = t;
  return JH_OK;
}

extern "C" void jh_tree_destroy(jh_tree* t) {
  if (!t) return;
  (void)hipFree(t->d_f); (void)hipFree(t->d_i); (void)hipFree(t->d_stats);
  for (hipEvent_t e : t->events) (void)hipEventDestroy(e);
  delete t;
}

extern "C" int jh_tree_stats(jh_tree* t, int* out4, int reset) {
  JH_REQUIRE(t && out4, "tree_stats: null pointer");
  JH_HIP(hipDeviceSynchronize());
  JH_HIP(hipMemcpy(out4, t->d_stats, 4 * sizeof(int), hipMemcpyDeviceToHost));
#ifdef JH_V4_PHASES
  {
    unsigned long long ph[16]; double tot = 0;
    JH_HIP(hipMemcpy(ph, t->d_stats + 8, sizeof(ph), hipMemcpyDeviceToHost));
    for (int i = 0; i < 10; i++) tot += (double)ph[i];
    const char* nm[10] = {"kinematics", "inertia+bias", "a0 solve", "collision+rows", "warm/step", "gradient", "hessian", "factor", "linesearch", "integrate"};
    for (int i = 0; i < 10; i++) fprintf(stderr, "  phase %-15s %6.2f%%  %.0f cycles/step/wave\n", nm[i], 100.0 * ph[i] / (tot > 0 ? tot : 1), out4[3] ? 2.0 * ph[i] / out4[3] : 0.0);
    const char* sn[4] = {"chain blocks", "base rows+schur", "reduced system", "chain backsub"};
    for (int i = 0; i < 4; i++) fprintf(stderr, "    solve: %-15s %.0f cycles/step/wave\n", sn[i], out4[3] ? 2.0 * ph[10 + i] / out4[3] : 0.0);
  }
#endif
  if (reset) JH_HIP(hipMemset(t->d_stats, 0, 64 * sizeof(int)));
  return JH_OK;
}

extern "C" int jh_tree_dims(const jh_tree* t, int* out4) {
  JH_REQUIRE(t && out4, "tree_dims: null pointer");
  out4[0] = t->nq; out4[1] = t->nv; out4[2] = t->nj; out4[3] = t->ns;
  return JH_OK;
}

extern "C" int jh_tree_substeps(const jh_tree* t, const float* state_in, const float* ctrl, float* warmstart, int N, int substeps, float* state_out, float* sensors_out,
                                void* stream) {
  JH_REQUIRE(t && state_in && ctrl && state_out, "tree_substeps: null pointer");
  JH_REQUIRE(N > 0 && substeps > 0, "tree_substeps: need at least one rollout and one step");
  const int dshift = jh_latency_shift(N, RPW), per_wave = RPW >> dshift;
  hipLaunchKernelGGL(k_tree_v4, dim3((N + per_wave - 1) / per_wave), dim3(WAVE), 0, (hipStream_t)stream, t->d_f, t->d_i, t->nf, t->ni, state_in, NX, ctrl, warmstart, N, substeps, state_out, NX,
                     sensors_out, t->ns, t->d_stats, dshift);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

namespace {
__global__ void k_fill_after_cutoff(float* rows, int W, int N, int T, int done) {  // rows done..T-1 of every rollout repeat row done-1 (zeros when nothing was computed)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)(T - done) * W;
  if (i >= (size_t)N * per) return;
  const size_t n = i / per, r = i % per, tt = done + r / W, c = r % W;
  rows[(n * T + tt) * W + c] = done > 0 ? rows[(n * T + done - 1) * W + c] : 0.f;
}
}  // namespace

extern "C" size_t jh_policy_rollout_scratch_floats(int N) { return jh_policy_scratch_floats(N) + (size_t)(N > 0 ? N : 0) * NJ; }

extern "C" int jh_policy_rollout(const jh_policy* p, jh_tree* t, const float* x0, int x0_batched, const float* commands, float* policy_out, float* warmstart, int reset_warmstart,
                                 int N, int T, int substeps, double cutoff_seconds, float* states, float* sensors, float* scratch, int* steps_done, void* stream) {
  JH_REQUIRE(p && t && x0 && commands && policy_out && states && scratch, "policy_rollout: null pointer");
  JH_REQUIRE(N > 0 && T > 0 && substeps > 0, "policy_rollout: need at least one rollout, one command row and one substep");
  JH_REQUIRE(!reset_warmstart || warmstart, "policy_rollout: reset_warmstart needs a warmstart buffer");
  hipStream_t st = (hipStream_t)stream;
  float* control = scratch + jh_policy_scratch_floats(N);
  const bool deadline = cutoff_seconds >= 0.0;
  if (deadline) {
    while ((int)t->events.size() < T + 1) { hipEvent_t e; JH_HIP(hipEventCreate(&e)); t->events.push_back(e); }
    JH_HIP(hipEventRecord(t->events[0], st));
  }
  int done = T;
  for (int i = 0; i < T; i++) {
    if (deadline) {  // System::rollout checks its clock before every command row; here: device time, read two control steps back so the queue never drains
      float ms = 0.f;
      if (i >= 2) { JH_HIP(hipEventSynchronize(t->events[i - 1])); JH_HIP(hipEventElapsedTime(&ms, t->events[0], t->events[i - 1])); }
      if (!((double)ms * 1e-3 < cutoff_seconds)) { done = i; break; }
    }
    const float* xin = i == 0 ? x0 : states + (size_t)(i - 1) * NX;
    const int ld = i == 0 ? (x0_batched ? NX : 0) : T * NX;
    const int rc = jh_policy_step_strided(p, xin, ld, NQ, 0, 0, 7, 6, commands + (size_t)i * 25, T * 25, policy_out, control, scratch, N, st);
    if (rc != JH_OK) return rc;
    if (reset_warmstart) JH_HIP(hipMemsetAsync(warmstart, 0, (size_t)N * NVT * sizeof(float), st));
    const int dshift = jh_latency_shift(N, RPW), per_wave = RPW >> dshift;
    hipLaunchKernelGGL(k_tree_v4, dim3((N + per_wave - 1) / per_wave), dim3(WAVE), 0, st, t->d_f, t->d_i, t->nf, t->ni, xin, ld, control, warmstart, N, substeps, states + (size_t)i * NX, T * NX,
                       sensors ? sensors + (size_t)i * t->ns : nullptr, T * t->ns, t->d_stats, dshift);
    if (deadline) JH_HIP(hipEventRecord(t->events[i + 1], st));
  }
  JH_HIP(hipGetLastError());
  if (done < T) {
    const size_t tot = (size_t)N * (T - done) * NX;
    hipLaunchKernelGGL(k_fill_after_cutoff, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, states, NX, N, T, done);
    if (sensors && t->ns > 0) {
      const size_t tots = (size_t)N * (T - done) * t->ns;
      hipLaunchKernelGGL(k_fill_after_cutoff, dim3((unsigned)((tots + 255) / 256)), dim3(256), 0, st, sensors, t->ns, N, T, done);
    }
    JH_HIP(hipGetLastError());
  }
  if (steps_done) *steps_done = done;
  return JH_OK;
}
