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
//
// Round 5: the robot against ITSELF (judo/models/xml/spot_primitive/contact.xml:4-14 excludes 11 body pairs; MuJoCo's static filters leave 287 geom pairs:
// capsule-capsule, box-capsule, sphere-capsule, box-sphere, sphere-sphere, box-box), template parameter SELF.  A pair between two links of one chain, or between a
// link and the base, still moves with the base and ONE chain: the compact row holds the difference of the two sides' columns (the base columns cancel).  A pair
// between two different chains (leg against leg, arm against leg) needs a second chain block (J2, at most NX2 such contacts per rollout and step) and couples the two
// chains in the Hessian: the fill-in-free tree factorisation does not apply, and the rollouts that have such a contact in a step take a dense 25 x 25 Cholesky
// (dense_cholesky_solve: one row per lane in registers, pivot rows broadcast through LDS) for that step's Newton systems.
#include <cstddef>
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
enum { GF4_SIZE = 0, GF4_POS = 3, GF4_R = 6, GF4_MU = 15, GF4_K, GF4_B, GF4_SOLIMP = 18, GF4_TRAN = 23, GF4_RBOUND = 24, GF4_MUOWN = 25 };
// MU, K, B, SOLIMP: mixed with the plane (a geom's plane contacts); robot-robot pairs: max of the two MUOWN, the sum of the two TRAN, and K / B / SOLIMP as stored (tree_model.py
// checks that every robot geom carries the same solref / solimp / priority, so the mixed values are the stored ones)

// Chain layout the kernel is instantiated for: four legs of three hinges and one arm of seven, contiguous in the dof order (checked by
// jh_tree_create).  With the chains eliminated before the base, the Cholesky factor of the inertia and of every Newton Hessian has no fill-in:
// chain blocks (3x3, 7x7), base-chain coupling, base block -- contacts with the ground couple the base with ONE chain, friction loss and limits
// are diagonal -- so the chains factorise side by side (7 levels) and only the 6 base pivots are sequential (mj_factorM's sparsity, cooperative).
constexpr int NCH = 5;
__host__ __device__ constexpr int CS(int c) { return c < 4 ? 3 * c : 12; }
__host__ __device__ constexpr int CL(int c) { return c < 4 ? 3 : 7; }
constexpr int SF_MAX = 2048, SI_MAX = 160, LBW = 28;
constexpr int NX2 = 8;       // contacts between two different chains a rollout can hold per step (their second chain block lives in RS4::J2)
constexpr int MAXHIT4 = 64;  // robot-robot geom pairs that survive the broad phase, per rollout and step
constexpr int MAXPP = 9;     // robot-robot geom pairs per lane (32 lanes: 288 pairs; Spot has 287): the lane's share of the pair list is read once per launch and kept in registers

template <bool SELF>
struct __attribute__((aligned(16))) RS4T {  // per-rollout shared state; positions are relative to the base origin
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
  union {
    float M[NVT][NVT]; float J[NCP][JW];  // the inertia lives in LDS only until every lane has its row in registers (25*25 < 32*44)
    struct { float gc[G][4]; float bx[G][12]; int hits[MAXHIT4]; } col;  // robot-robot broad phase (between the inertia rows and the Jacobian): geom centres + bounding radii, world poses of the box geoms (rotation 9, centre 3), surviving pairs
  };
  float J2[SELF ? NX2 : 1][24];              // second chain block (7 x 3, padded) of the contacts between two chains
  int cinfo[SELF ? NCP : 1];                 // per contact: chain / depth of both sides, cross index (decoded by cinfo_* below)
  int xc[SELF ? NX2 : 1];                    // contact index of cross contact x
  int ncon, nx2;
};
using RS4 = RS4T<false>;
static_assert(offsetof(RS4T<true>, Hc) % 16 == 0 && offsetof(RS4T<false>, Hc) % 16 == 0 && sizeof(RS4T<true>) % 16 == 0 && sizeof(RS4T<false>) % 16 == 0, "Hc rows are moved as 16-byte vectors");
// cinfo: bits 0-2 side B's chain (0 = the base body, 1 + chain otherwise), 3-5 its depth in the chain, 6-8 / 9-11 the same for side A, 12 side A is a robot geom (a
// robot-robot contact; else the plane), 13-16 cross index + 1 (0: both sides move with the base and at most one chain), 17 dead (a cross contact above NX2: dropped)
__device__ __forceinline__ int ci_chB(int ci) { return ci & 7; }
__device__ __forceinline__ int ci_depB(int ci) { return (ci >> 3) & 7; }
__device__ __forceinline__ int ci_chA(int ci) { return (ci >> 6) & 7; }
__device__ __forceinline__ int ci_depA(int ci) { return (ci >> 9) & 7; }
__device__ __forceinline__ bool ci_self(int ci) { return (ci >> 12) & 1; }
__device__ __forceinline__ int ci_x(int ci) { return (ci >> 13) & 15; }
__device__ __forceinline__ int ci_P(int ci) { return ci_chB(ci) > 0 ? ci_chB(ci) : (ci_self(ci) ? ci_chA(ci) : 0); }                                   // primary chain (1 + id), 0 = none
__device__ __forceinline__ int ci_Q(int ci) { return (ci_self(ci) && ci_chA(ci) > 0 && ci_chB(ci) > 0 && ci_chA(ci) != ci_chB(ci)) ? ci_chA(ci) : 0; }  // second chain of a cross contact

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
    float n1 = *d1 + D * x * xp, n2 = *d2 + D * xp * xp; asm volatile("" : "+v"(n1), "+v"(n2));  // (both sums computed, then selected -- the same fused multiply-adds: the branch form was four exec-masked regions per evaluation of the line search)
    *d1 = x < 0.f ? n1 : *d1; *d2 = x < 0.f ? n2 : *d2;
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
  // Straight-line code (round 6; the leap kernel's lane_rows_dir): a lane without a contact has D = 0 and a zero jar / jp -- its terms are zero -- and the dof rows are selects
  // on the same expressions as the branches they replace (same bits).
  float g1 = 0.f, g2 = 0.f;
  {
    float jar[3] = {fmaf(al, sl.jp[0], sl.jar[0]), fmaf(al, sl.jp[1], sl.jar[1]), fmaf(al, sl.jp[2], sl.jar[2])};
    float c1 = 0.f, c2 = 0.f; pyramid_dir4(jar, sl.jp, sl.D, sl.mu, &c1, &c2);
    g1 = sl.valid ? c1 : 0.f; g2 = sl.valid ? c2 : 0.f;
  }
  {
    const float jp = dr.pf, x = fmaf(al, jp, dr.jf), fl = dr.fl, lim = dr.fR * fl;
    const bool has = fl > 0.f, lo_ = x <= -lim, hi_ = x >= lim, mid = has & !lo_ & !hi_;
    float ta = g1 - fl * jp, tb = g1 + fl * jp, tm = g1 + dr.fD * x * jp, t2 = g2 + dr.fD * jp * jp; asm volatile("" : "+v"(ta), "+v"(tb), "+v"(tm), "+v"(t2));
    g1 = has ? (lo_ ? ta : (hi_ ? tb : tm)) : g1;
    g2 = mid ? t2 : g2;
  }
  {
    const float jp = dr.pl, x = fmaf(al, jp, dr.jl);
    const bool on = (dr.lims != 0.f) & (x < 0.f);
    float t1 = g1 + dr.lD * x * jp, t2 = g2 + dr.lD * jp * jp; asm volatile("" : "+v"(t1), "+v"(t2));
    g1 = on ? t1 : g1; g2 = on ? t2 : g2;
  }
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
    // (four single selects: the nested ternary came out as exec-masked regions -- fifteen per solve)
    float r = cid == 3 ? c3 : v; r = cid == 2 ? c2 : r; r = cid == 1 ? c1 : r; r = cid == 0 ? c0 : r;
    asm volatile("" : "+v"(r));
    v = r;
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
template <class RS>
__device__ __forceinline__ float tree_cholesky_solve(float* row, RS& S, const Role& R, int k, float* xb PA_PARAM) {
  PHS_DECL
  if (R.isjoint) {
    // (round 6: the whole 8-float row as two 16-byte stores.  Entries beyond the lane's own chain position are never read -- a reader takes [m] for m <= the row's position --
    // and `if (m <= R.cdepth)` in front of each of seven stores was seven exec-masked regions on a wave that has its SIMD to itself.)
    float v[8];
#pragma unroll
    for (int m = 0; m < 7; m++) v[m] = chain_entry(row, R.cid, m);
    float4* o = reinterpret_cast<float4*>(S.Hc[k]);
    o[0] = make_float4(v[0], v[1], v[2], v[3]); o[1] = make_float4(v[4], v[5], v[6], 0.f);
  }
  __syncthreads();
  const int fcs = R.isjoint ? R.cstart : CS(4), flen = R.isjoint ? R.clen : 7;
  float L[28];
#pragma unroll
  for (int pp = 0; pp < 7; pp++) {
    // (every row is LOADED whatever the lane's chain length -- S.Hc[fcs + pp] is a valid row for every pp: fcs + 6 <= 18 -- and the identity padding of a leg's 3 x 3 block is a
    // select: a load in one arm of the ternary was an exec-masked region per entry, 28 of them)
    const float* hr = S.Hc[fcs + pp];
    float hv[7];
    {
      const float4 h0 = *reinterpret_cast<const float4*>(hr), h1 = *reinterpret_cast<const float4*>(hr + 4);
      hv[0] = h0.x; hv[1] = h0.y; hv[2] = h0.z; hv[3] = h0.w; hv[4] = h1.x; hv[5] = h1.y; hv[6] = h1.z;
    }
#pragma unroll
    for (int m = 0; m <= pp; m++) L[tri4(pp, m)] = pp < flen ? hv[m] : (m == pp ? 1.f : 0.f);
  }
  chol_packed<7>(L);
  PHS(10)
  // (round 6) A leg's lanes have just factorised the leg's 3 x 3 block (the leading block of their padded 7 x 7): the lane at chain position p puts row p of the FACTOR back
  // where row p of the block was, and the base lanes read the four factors instead of factorising the four blocks again -- 4 x chol_packed<3> per base lane and call, on a
  // phase (base rows + Schur) that only seven lanes of the wave work in.  Row p of chol_packed<3> and of chol_packed<7> are the same expressions on the same numbers.
  if (R.isjoint && R.cid < 4) {
    const int cd = R.cdepth;
    const float f0 = cd == 0 ? L[0] : (cd == 1 ? L[1] : L[3]), f1 = cd == 1 ? L[2] : L[4], f2 = L[5];
    float* o = S.Hc[k]; o[0] = f0; o[1] = f1; o[2] = f2;  // (entries beyond the own position are never read)
  }
  __syncthreads();
  if (R.bl >= 0) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float Lg[6];
#pragma unroll
      for (int pp = 0; pp < 3; pp++) {
#pragma unroll
        for (int m = 0; m <= pp; m++) Lg[tri4(pp, m)] = S.Hc[CS(c) + pp][m];
      }
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

// The same system when contacts couple two chains (robot self-collision: leg against leg, arm against leg): the matrix is no longer tree-structured, so it is
// factorised densely.  Every dof lane holds its full symmetric row (26 registers, the layout above; the accumulation of J' W J fills every block a contact touches),
// the right-hand-side lane its vector in the same layout.  Right-looking Cholesky in the column order of that layout (joints, then base): the owner of column p scales
// its row by 1 / sqrt(pivot) and publishes it (26 floats, two alternating LDS vectors: one barrier per pivot), every later row -- and the right-hand side, which so
// becomes y = L^-1 b -- subtracts its multiple.  The owner keeps the scaled row: it IS column p of L, which the back substitution L' x = y needs row-wise in
// exactly that lane.  Columns already eliminated are left alone (a later row's entries there are dead; the right-hand side's hold the finished y_j).  ~25 x (26 multiply-adds + 7 broadcast reads) forward and as many
// backward: three to four times the tree solve, paid only by the wave-steps in which a rollout has such a contact.
template <class RS>
__device__ __forceinline__ float dense_cholesky_solve(float* row, RS& S, const Role& R, int k, float* xb) {
  const int l = threadIdx.x & 31;
  const bool isbase = l < 6, isrhs = R.bl == 6;
  const int mycol = R.isjoint ? k : (isbase ? NJ + l : 99);  // own column in the row layout (99: not a dof lane)
  float myr = 1.f;
#pragma unroll
  for (int p = 0; p < NVT; p++) {
    float* u = S.vec[p & 1];
    if (mycol == p) {
      const float r = __frsqrt_rn(fmaxf(row[p], 1e-30f));
      myr = r;
#pragma unroll
      for (int j = p; j < NVT; j++) { row[j] *= r; u[j] = row[j]; }
      u[NVT] = r;
    }
    __syncthreads();
    if ((mycol > p && mycol < 99) || isrhs) {
      const float m = row[p] * u[NVT];
#pragma unroll
      for (int j = p; j < NVT; j++) row[j] = fmaf(-m, u[j], row[j]);  // columns j < p are finished: a later row's entries there are dead, the right-hand side's hold y_j
      if (isrhs) row[p] = m;  // y_p
    }
  }
  __syncthreads();
  float* xv = S.vec[2];  // y, overwritten from the last column backwards by x
  if (isrhs) {
#pragma unroll
    for (int j = 0; j < NVT; j++) xv[j] = row[j];
  }
  __syncthreads();
#pragma unroll
  for (int p = NVT - 1; p >= 0; p--) {
    if (mycol == p) {
      float sx = xv[p];
#pragma unroll
      for (int j = p + 1; j < NVT; j++) sx = fmaf(-row[j], xv[j], sx);
      xv[p] = sx * myr;
    }
    __syncthreads();
  }
  float x_own = 0.f;
#pragma unroll
  for (int b = 0; b < 6; b++) xb[b] = xv[NJ + b];
  if (mycol < NVT) x_own = xv[mycol];
  __syncthreads();
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

// the second chain block of a contact between two chains (7 x 3 in RS4::J2), added to o
__device__ __forceinline__ void jac_mul2(float* o, const float* J2x, const float* v, int csq) {
  const float* vc = v + 6 + csq;
#pragma unroll
  for (int m = 0; m < 7; m++) { const float w = vc[m]; o[0] = fmaf(J2x[3 * m], w, o[0]); o[1] = fmaf(J2x[3 * m + 1], w, o[1]); o[2] = fmaf(J2x[3 * m + 2], w, o[2]); }
}

// dot of a register row (dof order) with a broadcast LDS vector
__device__ __forceinline__ float dot_row(const float* Mrow, const float* v) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NVT; j++) s = fmaf(Mrow[j], v[j], s);
  return s;
}


template <bool SELF>
__global__ __launch_bounds__(WAVE, 1) void k_tree_v4(const float* __restrict__ gF, const int* __restrict__ gI, int nF, int nI, const float* state_in, int ld_in,
                                                    const float* __restrict__ ctrl, float* __restrict__ warm, int N, int substeps, float* state_out, int ld_out,
                                                    float* __restrict__ sensors_out, int ld_sens, int* __restrict__ stats, int dshift) {
  using RS = RS4T<SELF>;
  __shared__ RS sRS[RPW];
  __shared__ __attribute__((aligned(16))) float sF[SF_MAX];  // the model image, shared by the rollouts of the wave
  __shared__ int sI[SI_MAX];
  const int lane = threadIdx.x, l = lane & 31, r = lane >> 5;
  RS& S = sRS[r];
  for (int i = lane; i < nF && i < SF_MAX; i += WAVE) sF[i] = gF[i];  // (the sensor records at the end of the image are read from global memory, once per launch)
  for (int i = lane; i < nI && i < SI_MAX; i += WAVE) sI[i] = gI[i];
  __syncthreads();
  // (latency mode, jh_internal.h: with dshift = 1 both rows of the wave compute the same rollout and the first writes -- the shipped 24-rollout plans)
  const int n = (blockIdx.x << (1 - dshift)) + (r >> dshift);
  const bool live = n < N && (r & ((1 << dshift) - 1)) == 0;
  const int nc = n < N ? n : N - 1;
  const int nj = NJ, ng = sI[1];
  const int npair = SELF ? sI[6] : 0, oPair = sI[7];  // robot-robot geom pairs (g1 | g2 << 8, g1 < g2), read from the global image
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
  int pkr[MAXPP];  // this lane's robot-robot pairs (pair 32 i + l), -1 beyond the list: nine L2 round trips per LAUNCH instead of nine per step
#pragma unroll
  for (int i = 0; i < MAXPP; i++) pkr[i] = (SELF && 32 * i + l < npair) ? gI[oPair + 32 * i + l] : -1;
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
      if (l == 0) { S.ncon = 0; S.nx2 = 0; }
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
      float bR[9], gp[3], lp[3] = {gf[GF4_POS], gf[GF4_POS + 1], gf[GF4_POS + 2]};
      for (int i = 0; i < 9; i++) bR[i] = S.xR[gb][i];
      mulMV(gp, bR, lp); for (int i = 0; i < 3; i++) gp[i] += S.xpos[gb][i];
      const float pr[3] = {plp[0] - qb[0], plp[1] - qb[1], plp[2] - qb[2]};  // plane point relative to the base origin
      auto push = [&](const float* pos, float dist, const float* tng) __attribute__((always_inline)) {
        const int i = atomicAdd(&S.ncon, 1);
        if (i >= NCP) { if (stats) atomicAdd(stats, 1); return; }
        float* e = S.raw[i];
        e[0] = pos[0]; e[1] = pos[1]; e[2] = pos[2]; e[3] = dist; e[4] = __int_as_float(l); e[5] = tng[0]; e[6] = tng[1]; e[7] = tng[2];  // [4]: geom B | (geom A + 1) << 8, A = the plane: 0
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
    if constexpr (SELF) {
      PH(3)
      // ================================================================ the robot against itself: bounding spheres of the 287 pairs, survivors one per lane
      auto geom_pose = [&](int g, float* gp, float* gR) __attribute__((always_inline)) {
        const float* gf = sF + oGF + g * TG_F; const int owner = sI[oGI + g * TG_I]; const int gb = owner < 0 ? 0 : 1 + owner;
        float bR[9]; for (int i = 0; i < 9; i++) bR[i] = S.xR[gb][i];
        mulMV(gp, bR, gf + GF4_POS); for (int i = 0; i < 3; i++) gp[i] += S.xpos[gb][i];
        mulMM(gR, bR, gf + GF4_R);
      };
      if (l < ng) {
        float gp[3], gR[9]; geom_pose(l, gp, gR);
        S.col.gc[l][0] = gp[0]; S.col.gc[l][1] = gp[1]; S.col.gc[l][2] = gp[2]; S.col.gc[l][3] = sF[oGF + l * TG_F + GF4_RBOUND];
        if (sI[oGI + l * TG_I + 1] == 6) { for (int i = 0; i < 9; i++) S.col.bx[l][i] = gR[i]; }  // (a box's pose once per step, not once per pair that names it)
        if (sI[oGI + l * TG_I + 1] == 3) { S.col.bx[l][0] = gR[2]; S.col.bx[l][1] = gR[5]; S.col.bx[l][2] = gR[8]; }  // a capsule's axis
      }
      __syncthreads();
      int nh = 0;
#pragma unroll
      for (int ip = 0; ip < MAXPP; ip++) {
        if (32 * ip >= npair) break;
        bool hit = false; const int pk = pkr[ip];
        if (pk >= 0) {
          const int g1 = pk & 255, g2 = pk >> 8;
          const float* c1 = S.col.gc[g1]; const float* c2 = S.col.gc[g2];
          const float d[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]}, rs = c1[3] + c2[3];
          hit = dot3(d, d) <= rs * rs;  // (mj_collideGeoms' bounding-sphere filter; conservative, like the box cull below: neither changes a contact)
          // a box against the other geom's bounding sphere: the body box's own sphere (0.44 m) holds most of the robot
#pragma unroll
          for (int sd = 0; sd < 2; sd++) {
            const int gb_ = sd == 0 ? g1 : g2; const float* co = sd == 0 ? c2 : c1;
            if (hit && sI[oGI + gb_ * TG_I + 1] == 6) {
              const float* bp = S.col.gc[gb_]; const float* bRm = S.col.bx[gb_];
              const float dw[3] = {co[0] - bp[0], co[1] - bp[1], co[2] - bp[2]}; float dl[3]; mulMTV(dl, bRm, dw);
              const float* hs = sF + oGF + gb_ * TG_F + GF4_SIZE;
              // a capsule as its axis segment's extent along the box axes (an outer bound of the segment) plus its radius, anything else as its bounding sphere
              const int go_ = sd == 0 ? g2 : g1; const bool cap = sI[oGI + go_ * TG_I + 1] == 3;
              float al[3] = {0.f, 0.f, 0.f}, rr = co[3];
              if (cap) { mulMTV(al, bRm, S.col.bx[go_]); const float* so = sF + oGF + go_ * TG_F + GF4_SIZE; rr = so[0]; for (int i = 0; i < 3; i++) al[i] = fabsf(al[i]) * so[1]; }
              const float ex = fmaxf(fabsf(dl[0]) - hs[0] - al[0], 0.f), ey = fmaxf(fabsf(dl[1]) - hs[1] - al[1], 0.f), ez = fmaxf(fabsf(dl[2]) - hs[2] - al[2], 0.f);
              hit = ex * ex + ey * ey + ez * ez <= rr * rr;
            }
          }
        }
        const unsigned m32 = (unsigned)(__ballot(hit) >> (32 * r));
        const int pos = nh + __popc(m32 & ((1u << l) - 1u));
        if (hit && pos < MAXHIT4) S.col.hits[pos] = pk;
        nh += __popc(m32);
      }
      if (nh > MAXHIT4) { if (l == 0 && live && stats) atomicAdd(stats, nh - MAXHIT4); nh = MAXHIT4; }  // (candidate pairs lost: counted with the dropped contacts)
      __syncthreads();
      PH(14)
      // narrow phase, as restated in oracle/jo_engine.c collide_geoms; the normal points from geom 1 to geom 2 of the pair.  MuJoCo's own primitives: capsule-capsule
      // (mjc_CapsuleCapsule), sphere-capsule, sphere-sphere (mjraw_SphereSphere, incl. its (1,0,0) normal for coincident centres), box-sphere.  NOT MuJoCo's: box-capsule (75 of
      // the 287 pairs) and box-box (9) go through the build's own jh_coop.h routines -- collide_box_capsule gives up to THREE contacts (both end spheres and the interior
      // closest point) where mjc_CapsuleBox gives at most two, box-box its own face manifold: stated deviations, shared with the oracle, and therefore
      // invisible to the parity tests -- no MuJoCo contact set is recorded anywhere in the reference to hold them to.  MuJoCo orders a pair by geom TYPE (sphere < capsule <
      // box) and points the normal from the lower type; the oracle does that since round 6, this kernel keeps the model's order: mju_makeFrame(-n) = (-n, y, -z) for
      // (n, y, z) and the pyramid is symmetric in its tangents, so the constraint set is the same either way (the parity test sees only a different row order).
      struct SelfSink {
        RS* S; int* stats; int pk; bool flip;
        __device__ __forceinline__ void push(const float* pos, const float* n, float dist) {
          const int i = atomicAdd(&S->ncon, 1);
          if (i >= NCP) { if (stats) atomicAdd(stats, 1); return; }
          float* e = S->raw[i];
          e[0] = pos[0]; e[1] = pos[1]; e[2] = pos[2]; e[3] = dist; e[4] = __int_as_float((pk >> 8) | (((pk & 255) + 1) << 8));
          e[5] = flip ? -n[0] : n[0]; e[6] = flip ? -n[1] : n[1]; e[7] = flip ? -n[2] : n[2];
        }
      };
      for (int base = 0; __any(base < nh); base += G) {
        const int idx = base + l;
        if (idx < nh) {
          const int pk = S.col.hits[idx], g1 = pk & 255, g2 = pk >> 8;
          const int t1 = sI[oGI + g1 * TG_I + 1], t2 = sI[oGI + g2 * TG_I + 1];
          const float* f1 = sF + oGF + g1 * TG_F; const float* f2 = sF + oGF + g2 * TG_F;
          float p1[3], R1[9], p2[3], R2[9]; geom_pose(g1, p1, R1); geom_pose(g2, p2, R2);
          SelfSink sk{&S, live ? stats : nullptr, pk, false};
          auto sphere_sphere = [&](const float* ca, float ra, const float* cb, float rb) __attribute__((always_inline)) {
            const float d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]}; const float ln = sqrtf(dot3(d, d)), dist = ln - ra - rb;
            if (dist > 0.f) return;  // (mjraw_SphereSphere keeps dist == margin)
            const bool same = ln < 1e-12f;  // coincident centres: mju_normalize3 returns (1, 0, 0)
            const float n3[3] = {same ? 1.f : d[0] / ln, same ? 0.f : d[1] / ln, same ? 0.f : d[2] / ln}, mm = ra + 0.5f * dist;
            const float ps[3] = {ca[0] + mm * n3[0], ca[1] + mm * n3[1], ca[2] + mm * n3[2]};
            sk.push(ps, n3, dist);
          };
          auto sphere_capsule = [&](const float* ps, float rs_, const float* pc, const float* Rc, const float* sc) __attribute__((always_inline)) {  // normal: sphere -> capsule
            float a[3]; col3(a, Rc, 2);
            const float d[3] = {ps[0] - pc[0], ps[1] - pc[1], ps[2] - pc[2]};
            const float x = jh_clampf(dot3(a, d), -sc[1], sc[1]);
            const float v[3] = {pc[0] + a[0] * x, pc[1] + a[1] * x, pc[2] + a[2] * x};
            sphere_sphere(ps, rs_, v, sc[0]);
          };
          if (t1 == 6 && t2 == 6) collide_box_box(sk, p1, R1, f1 + GF4_SIZE, p2, R2, f2 + GF4_SIZE);
          else if (t1 == 6 && t2 == 2) collide_box_sphere(sk, p1, R1, f1 + GF4_SIZE, p2, f2[GF4_SIZE]);
          else if (t1 == 2 && t2 == 6) { sk.flip = true; collide_box_sphere(sk, p2, R2, f2 + GF4_SIZE, p1, f1[GF4_SIZE]); }
          else if (t1 == 6 && t2 == 3) collide_box_capsule(sk, p1, R1, f1 + GF4_SIZE, p2, R2, f2[GF4_SIZE], f2[GF4_SIZE + 1]);
          else if (t1 == 3 && t2 == 6) { sk.flip = true; collide_box_capsule(sk, p2, R2, f2 + GF4_SIZE, p1, R1, f1[GF4_SIZE], f1[GF4_SIZE + 1]); }
          else if (t1 == 2 && t2 == 2) sphere_sphere(p1, f1[GF4_SIZE], p2, f2[GF4_SIZE]);
          else if (t1 == 2 && t2 == 3) sphere_capsule(p1, f1[GF4_SIZE], p2, R2, f2 + GF4_SIZE);
          else if (t1 == 3 && t2 == 2) { sk.flip = true; sphere_capsule(p2, f2[GF4_SIZE], p1, R1, f1 + GF4_SIZE); }
          else if (t1 == 3 && t2 == 3) {  // mjc_CapsuleCapsule: closest points of the two axis segments, then sphere against sphere
            float a1[3], a2[3]; col3(a1, R1, 2); col3(a2, R2, 2);
            const float r1_ = f1[GF4_SIZE], r2_ = f2[GF4_SIZE];
            for (int i = 0; i < 3; i++) { a1[i] *= f1[GF4_SIZE + 1]; a2[i] *= f2[GF4_SIZE + 1]; }
            const float dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
            const float ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
            const float det = ma * mc - mb * mb;
            if (fabsf(det) >= 1e-6f * ma * mc) {  // (MuJoCo: |det| >= mjMINVAL in fp64; in fp32 the determinant of two axes less than ~1e-3 rad apart is rounding noise)
              float x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
              if (x1 > 1.f) { x1 = 1.f; x2 = (v - mb) / mc; } else if (x1 < -1.f) { x1 = -1.f; x2 = (v + mb) / mc; }
              if (x2 > 1.f) { x2 = 1.f; x1 = jh_clampf((u - mb) / ma, -1.f, 1.f); }
              else if (x2 < -1.f) { x2 = -1.f; x1 = jh_clampf((u + mb) / ma, -1.f, 1.f); }
              const float v1[3] = {p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, p1[2] + a1[2] * x1}, v2[3] = {p2[0] + a2[0] * x2, p2[1] + a2[1] * x2, p2[2] + a2[2] * x2};
              sphere_sphere(v1, r1_, v2, r2_);
            } else {  // parallel axes: the ends of capsule 1 against segment 2, then the ends of capsule 2 against segment 1, at most two contacts
              int nfound = 0;
              for (int e = 0; e < 4; e++) {
                float x1, x2;
                if (e < 2) { x1 = e == 0 ? 1.f : -1.f; x2 = jh_clampf((v - x1 * mb) / mc, -1.f, 1.f); }
                else { x2 = e == 2 ? 1.f : -1.f; x1 = jh_clampf((u - x2 * mb) / ma, -1.f, 1.f); }
                const float v1[3] = {p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, p1[2] + a1[2] * x1}, v2[3] = {p2[0] + a2[0] * x2, p2[1] + a2[1] * x2, p2[2] + a2[2] * x2};
                const float d[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]}; const float ln = sqrtf(dot3(d, d));
                if (nfound < 2 && ln - r1_ - r2_ < 0.f && ln >= 1e-12f) { sphere_sphere(v1, r1_, v2, r2_); nfound++; }
              }
            }
          }
        }
      }
      PH(15)
    }
    __syncthreads();
    // ================================================================ constraint rows
    const int ncon = S.ncon < NCP ? S.ncon : NCP;
    Slot4 sl;
    sl.valid = l < ncon; sl.D = 0.f; sl.mu = 0.f;
    for (int w = 0; w < 3; w++) sl.aref[w] = sl.jar[w] = sl.jp[w] = 0.f;
    // sides of the own contact: B = the robot geom of a plane contact / geom 2 of a robot-robot pair, A = the plane / geom 1
    int my_ci = 0;
    {
      bool cross = false;
      if (sl.valid) {
        const int pk = __float_as_int(S.raw[l][4]), gB = pk & 255, gA1 = (pk >> 8) & 255;
        auto side = [&](int g, int& ch, int& dep) __attribute__((always_inline)) {
          const int owner = sI[oGI + g * TG_I];
          if (owner < 0) { ch = 0; dep = 0; } else { const int cs_ = sI[TH_I + owner * TD_I + 1]; ch = 1 + (cs_ < 12 ? cs_ / 3 : 4); dep = sI[TH_I + owner * TD_I + 2]; }
        };
        int chB, depB, chA = 0, depA = 0; side(gB, chB, depB);
        const bool selfc = SELF && gA1 != 0;
        if (selfc) side(gA1 - 1, chA, depA);
        my_ci = chB | (depB << 3) | (chA << 6) | (depA << 9) | ((selfc ? 1 : 0) << 12);
        cross = ci_Q(my_ci) != 0;
      }
      if constexpr (SELF) {
        const unsigned xm = (unsigned)(__ballot(cross) >> (32 * r));
        if (cross) {
          const int x = __popc(xm & ((1u << l) - 1u));
          if (x < NX2) { my_ci |= (x + 1) << 13; S.xc[x] = l; }
          else { my_ci |= 1 << 17; sl.valid = false; if (live && stats) atomicAdd(stats, 1); }  // (above the capacity for contacts between two chains: dropped and counted)
        }
        if (l == 0) S.nx2 = min((int)__popc(xm), NX2);
        if (l < ncon) S.cinfo[l] = my_ci;
      }
    }
    const int my_P = ci_P(my_ci), my_Q = ci_Q(my_ci);
    const int my_cs = my_P == 0 ? 0 : (my_P <= 4 ? 3 * (my_P - 1) : CS(4));   // first joint of the own contact's (primary) chain
    const int my_csq = my_Q == 0 ? 0 : (my_Q <= 4 ? 3 * (my_Q - 1) : CS(4));  // ... of its second chain (contacts between two chains)
    const int my_x = ci_x(my_ci) > 0 ? ci_x(my_ci) - 1 : 0;
    if (l < ncon) {  // contact frame.  Plane contacts: plane normal (from the plane, geom 1, to the robot geom), second axis along a capsule's axis; robot-robot: the narrow phase's normal
      const float* e = S.raw[l];
      float fr[9] = {pln[0], pln[1], pln[2], 0, 0, 0, 0, 0, 0};
      if (SELF && ci_self(my_ci)) { fr[0] = e[5]; fr[1] = e[6]; fr[2] = e[7]; }
      make_frame(fr);
      if (!(SELF && ci_self(my_ci))) {
        float y[3] = {e[5], e[6], e[7]};
        const float dp = dot3(fr, y); y[0] -= fr[0] * dp; y[1] -= fr[1] * dp; y[2] -= fr[2] * dp;
        const float nn = sqrtf(dot3(y, y));
        if (nn > 0.5e-3f) { for (int i = 0; i < 3; i++) fr[3 + i] = y[i] / nn; cross3(fr + 6, fr, fr + 3); }
      }
      for (int w = 0; w < 9; w++) S.fW[l][w] = fr[w];
    }
    __syncthreads();
    const int nx2 = SELF ? S.nx2 : 0;
    for (int c = 0; c < ncon; c++) {  // Jacobian: every dof lane its own column, side B minus side A (the plane is static); the spare lanes clear the padding
      const float* e = S.raw[c];
      int ci;
      if constexpr (SELF) ci = S.cinfo[c];
      else { const int owner = sI[oGI + (__float_as_int(e[4]) & 255) * TG_I]; ci = owner < 0 ? 0 : ((1 + (sI[TH_I + owner * TD_I + 1] < 12 ? sI[TH_I + owner * TD_I + 1] / 3 : 4)) | (sI[TH_I + owner * TD_I + 2] << 3)); }
      const int P = ci_P(ci), Q = SELF ? ci_Q(ci) : 0, xq = ci_x(ci) > 0 ? ci_x(ci) - 1 : 0;
      const bool dead = SELF && ((ci >> 17) & 1);
      const bool inP = isjoint && P == 1 + R.cid, inQ = SELF && isjoint && Q == 1 + R.cid && !dead;
      if (isbase || inP || inQ) {
        float sgn = 0.f;
        if (isbase) sgn = ci_self(ci) ? 0.f : 1.f;  // (both sides of a robot-robot contact ride on the base: its columns cancel)
        else {
          if (ci_chB(ci) == 1 + R.cid && cdepth <= ci_depB(ci)) sgn += 1.f;
          if (ci_self(ci) && ci_chA(ci) == 1 + R.cid && cdepth <= ci_depA(ci)) sgn -= 1.f;
        }
        const float pos[3] = {e[0], e[1], e[2]};
        float v[3]; cross3(v, Sown, pos); for (int i = 0; i < 3; i++) v[i] += Sown[3 + i];
        const float* fr = S.fW[c];
        const float col[3] = {sgn * dot3(fr, v), sgn * dot3(fr + 3, v), sgn * dot3(fr + 6, v)};
        float* o = isbase ? S.J[c] + 3 * l : (inP ? S.J[c] + JC + 3 * cdepth : S.J2[xq] + 3 * cdepth);
        o[0] = col[0]; o[1] = col[1]; o[2] = col[2];
      } else if (l >= NVT) {  // positions the contact's chain(s) do not have (legs: 3..6; a contact on the base alone: all)
        const int m = l - NVT, len = P == 0 ? 0 : (P < 5 ? 3 : 7);
        if (m >= len) { float* o = S.J[c] + JC + 3 * m; o[0] = o[1] = o[2] = 0.f; }
        if (SELF && Q != 0 && !dead && m >= (Q < 5 ? 3 : 7)) { float* o = S.J2[xq] + 3 * m; o[0] = o[1] = o[2] = 0.f; }
      }
    }
    __syncthreads();
    if (sl.valid) {
      const float* e = S.raw[l]; const float dist = e[3]; const int gid = __float_as_int(e[4]) & 255;
      const float* gf = sF + oGF + gid * TG_F;
      float si[5]; for (int w = 0; w < 5; w++) si[w] = gf[GF4_SOLIMP + w];
      float mu = gf[GF4_MU], tran = gf[GF4_TRAN];
      if (SELF && ci_self(my_ci)) {  // robot against robot: mj_contactParam with equal priorities -- the larger friction; the two bodies' inverse weights
        const float* ga = sF + oGF + (((__float_as_int(e[4]) >> 8) & 255) - 1) * TG_F;
        mu = fmaxf(gf[GF4_MUOWN], ga[GF4_MUOWN]); tran = gf[GF4_TRAN] + ga[GF4_TRAN];
      }
      const float imp = impedance(si, dist);
      const float R0 = fmaxf(1e-15f, (1.f - imp) / imp * tran * (1.f + mu * mu));
      const float Rpy = fmaxf(1e-15f, 2.f * (mu * mu / fmaxf(1e-15f, impratio)) * R0);
      sl.D = 1.f / Rpy; sl.mu = mu;
      float vel[3];
      jac_mul(vel, S.J[l], S.qd, my_cs);
      if (SELF && my_Q != 0) jac_mul2(vel, S.J2[my_x], S.qd, my_csq);
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
    if (hasdof) for (int c = 0; c < ncon; c++) {
      int P;
      if constexpr (SELF) P = ci_P(S.cinfo[c]);
      else { const int owner = sI[oGI + (__float_as_int(S.raw[c][4]) & 255) * TG_I]; P = owner < 0 ? 0 : 1 + (sI[TH_I + owner * TD_I + 1] < 12 ? sI[TH_I + owner * TD_I + 1] / 3 : 4); }
      involved |= (isbase || P == 1 + R.cid) ? (1u << c) : 0u;
    }
    // contacts between two chains (SELF): which of them have this lane's chain as their second chain, and the wave-uniform question whether a rollout has any
    unsigned involved2 = 0;
    if constexpr (SELF) { if (isjoint) for (int x = 0; x < nx2; x++) involved2 |= ci_Q(S.cinfo[S.xc[x]]) == 1 + R.cid ? (1u << x) : 0u; }
    const bool dense_step = SELF && __any(nx2 > 0);
    const int cd3 = 3 * (cdepth < 0 ? 0 : cdepth);
    const int own_col = isbase ? 3 * l : JC + 3 * (cdepth < 0 ? 0 : cdepth);  // own column in a compact Jacobian row
    const float iMd = 1.f / Md_own;
    const float snorm = gsum32(hasdof ? fs_own * fs_own * iMd : 0.f);
    int iters_this = 0;
    {
      // ---- warm start: the better of last step's acceleration and the unconstrained one
      if (hasdof) { S.vec[0][l] = qws; S.vec[1][l] = a0_own; S.vec[2][l] = qws - a0_own; }
      __syncthreads();
      float jar_ws[3] = {0.f, 0.f, 0.f};
      if (sl.valid) { float o[3]; jac_mul(o, S.J[l], S.vec[0], my_cs); if (SELF && my_Q != 0) jac_mul2(o, S.J2[my_x], S.vec[0], my_csq); for (int w = 0; w < 3; w++) sl.jar[w] = o[w] - sl.aref[w]; }
      dr.jf = qws - dr.faref; dr.jl = dr.lims * qws - dr.laref;
      const float mdw = dot_row(Mrow, S.vec[2]);
      const float cost_ws = gsum32(lane_cost4(sl, dr) + (hasdof ? 0.5f * (qws - a0_own) * mdw : 0.f));
      for (int w = 0; w < 3; w++) jar_ws[w] = sl.jar[w];
      const float jf_ws = dr.jf, jl_ws = dr.jl;
      if (sl.valid) { float o[3]; jac_mul(o, S.J[l], S.vec[1], my_cs); if (SELF && my_Q != 0) jac_mul2(o, S.J2[my_x], S.vec[1], my_csq); for (int w = 0; w < 3; w++) sl.jar[w] = o[w] - sl.aref[w]; }
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
        if (l < ncon) {  // (a dropped contact -- above the capacity for contacts between two chains -- keeps its row with zero force and weight)
          float f[3] = {0.f, 0.f, 0.f}, Wm[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (sl.valid) pyramid_eval(sl.jar, sl.D, sl.mu, f, Wm);
          float* o = S.fW[l]; o[0] = f[0]; o[1] = f[1]; o[2] = f[2]; for (int w = 0; w < 6; w++) o[4 + w] = Wm[w];
        }
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
        if constexpr (SELF) {
          for (int x = 0; x < nx2; x++) {  // second chain block of the contacts between two chains
            const float* jc = S.J2[x] + cd3; const float* fc = S.fW[S.xc[x]];
            const float on = (involved2 >> x) & 1u ? 1.f : 0.f;
            g_own -= on * (jc[0] * fc[0] + jc[1] * fc[1] + jc[2] * fc[2]);
          }
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
          int och;  // 0: both sides sit on the base (the chain part is zero), else 1 + the contact's (primary) chain
          if constexpr (SELF) och = ci_P(S.cinfo[c]);
          else { const int owner = sI[oGI + (__float_as_int(S.raw[c][4]) & 255) * TG_I]; och = owner < 0 ? 0 : 1 + (sI[TH_I + owner * TD_I + 1] < 12 ? sI[TH_I + owner * TD_I + 1] / 3 : 4); }
#pragma unroll
          for (int ch = 0; ch < NCH; ch++) {
            const float sel = och == 1 + ch ? 1.f : 0.f;
#pragma unroll
            for (int m = 0; m < CL(ch); m++) row[CS(ch) + m] = fmaf(sel, dch[m], row[CS(ch) + m]);
          }
#pragma unroll
          for (int m = 0; m < 6; m++) row[NJ + m] += dba[m];
        }
        if constexpr (SELF) {
          // contacts between two chains P and Q: a lane of P (or of the base) has its own column in J and took the P block and the base block above -- it still needs the Q
          // block; a lane of Q has its own column in J2 and takes all three.  Same J' W J, the row of every dof the contact moves complete and symmetric.
          for (int x = 0; x < nx2; x++) {
            const int c = S.xc[x], ci = S.cinfo[c], P = ci_P(ci), Q = ci_Q(ci);
            const float* fc = S.fW[c]; const float* Jc = S.J[c]; const float* Jq = S.J2[x];
            const float onP = (involved >> c) & 1u ? 1.f : 0.f, onQ = (involved2 >> x) & 1u ? 1.f : 0.f;
            const float j0 = onP * Jc[own_col] + onQ * Jq[cd3], j1 = onP * Jc[own_col + 1] + onQ * Jq[cd3 + 1], j2 = onP * Jc[own_col + 2] + onQ * Jq[cd3 + 2];
            const float G0 = fc[4] * j0 + fc[5] * j1 + fc[7] * j2, G1 = fc[5] * j0 + fc[6] * j1 + fc[8] * j2, G2 = fc[7] * j0 + fc[8] * j1 + fc[9] * j2;
            float dq[7], dch[7], dba[6];
#pragma unroll
            for (int m = 0; m < 7; m++) { dq[m] = Jq[3 * m] * G0 + Jq[3 * m + 1] * G1 + Jq[3 * m + 2] * G2; dch[m] = onQ * (Jc[JC + 3 * m] * G0 + Jc[JC + 3 * m + 1] * G1 + Jc[JC + 3 * m + 2] * G2); }
#pragma unroll
            for (int m = 0; m < 6; m++) dba[m] = onQ * (Jc[3 * m] * G0 + Jc[3 * m + 1] * G1 + Jc[3 * m + 2] * G2);
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
              const float selq = Q == 1 + ch ? 1.f : 0.f, selp = P == 1 + ch ? 1.f : 0.f;
#pragma unroll
              for (int m = 0; m < CL(ch); m++) row[CS(ch) + m] = fmaf(selq, dq[m], fmaf(selp, dch[m], row[CS(ch) + m]));
            }
#pragma unroll
            for (int m = 0; m < 6; m++) row[NJ + m] += dba[m];
          }
        }
        PH(6)
        // ---- (4) factorise and solve; the direction goes back through LDS
        float xb[6];
        // (a rollout with a contact between two chains has no tree-structured Hessian: the wave then factorises densely -- valid for its other rollout too)
        const float p_own = dense_step ? dense_cholesky_solve(row, S, R, k, xb) : tree_cholesky_solve(row, S, R, k, xb PA_ARG);
        if (hasdof) S.vec[2][l] = p_own;
        __syncthreads();
        PH(7)
        // ---- (5) exact line search
        const float Mp_own = dot_row(Mrow, S.vec[2]);
        const float pMp = gsum32(hasdof ? p_own * Mp_own : 0.f), pMd = gsum32(hasdof ? Mp_own * da_own : 0.f), gp = gsum32(hasdof ? g_own * p_own : 0.f);
        if (act && !(gp < 0.f)) act = false;
        if (sl.valid) { jac_mul(sl.jp, S.J[l], S.vec[2], my_cs); if (SELF && my_Q != 0) jac_mul2(sl.jp, S.J2[my_x], S.vec[2], my_csq); }
        dr.pf = p_own; dr.pl = dr.lims * p_own;
        float lo = 0.f, hi = -1.f, alpha = 1.f; bool lsact = act;
        for (int ls = 0; ls < 12 && __any(lsact); ls++) {
          float d1, d2;
          lane_dir4(sl, dr, alpha, &d1, &d2);
          d1 = gsum32(d1) + pMd + alpha * pMp; d2 = gsum32(d2) + pMp;
          {  // safeguarded Newton step on the slope, as selects (the leap kernel's form: the same arithmetic as the nested branches it replaces)
            const bool upd = lsact & !(fabsf(d1) <= lstol * fabsf(gp)), neg = d1 < 0.f;
            lo = (upd & neg) ? alpha : lo; hi = (upd & !neg) ? alpha : hi;
            float nx = alpha - d1 * __frcp_rn(d2);
            const bool out_lo = nx <= lo, out_hi = nx >= hi;
            float dbl = 2.f * alpha, mid_ = 0.5f * (lo + hi); asm volatile("" : "+v"(dbl), "+v"(mid_));
            nx = hi < 0.f ? (out_lo ? dbl : nx) : ((out_lo | out_hi) ? mid_ : nx);
            alpha = upd ? nx : alpha; lsact = upd;
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

struct jh_tree { float* d_f; int* d_i; int* d_stats; int nj, ng, nq, nv, nf, ni, ns, npair, self_collision; std::vector<hipEvent_t> events; };

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
  JH_REQUIRE(ii[4] >= 0 && ii[4] <= G && ii[6] >= 0 && nf == (size_t)(TH_F + ii[0] * TD_F + ii[1] * TG_F + ii[4] * TS_F) &&
             ni == (size_t)(TH_I + ii[0] * TD_I + ii[1] * TG_I + ii[4] * TS_I + ii[6]) && (ii[6] == 0 || ii[7] == TH_I + ii[0] * TD_I + ii[1] * TG_I + ii[4] * TS_I),
             "tree_create: image sizes do not match the counts in its header (or more than 32 sensors)");
  JH_REQUIRE(ii[6] <= MAXPP * G, "tree_create: %d robot-robot geom pairs, the kernel holds %d", ii[6], MAXPP * G);
  for (int p = 0; p < ii[6]; p++) {  // robot-robot geom pairs: g1 | g2 << 8 with g1 < g2 < number of geoms
    const int pk = ii[ii[7] + p], g1 = pk & 255, g2 = pk >> 8;
    JH_REQUIRE(g1 < g2 && g2 < ii[1], "tree_create: bad geom pair %d (%d, %d)", p, g1, g2);
  }
  for (int c = 0; c < NCH; c++)
    for (int m = 0; m < CL(c); m++) {
      const int* jr = ii + TH_I + (CS(c) + m) * TD_I;
      JH_REQUIRE(jr[1] == CS(c) && jr[2] == m, "tree_create: the kernel is instantiated for four 3-joint chains followed by one 7-joint chain (joint %d: chain start %d, depth %d)", CS(c) + m, jr[1], jr[2]);
    }
  jh_tree* t = new jh_tree();
  t->nf = (int)nf; t->ni = (int)ni; t->ns = ii[5];
  t->nj = ii[0]; t->ng = ii[1]; t->nq = ii[2]; t->nv = ii[3];
  t->npair = ii[6]; t->self_collision = ii[6] > 0 ? 1 : 0;  // the robot collides with itself when the image lists pairs, as MuJoCo does (jh_tree_set_self_collision)
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
    tot += (double)ph[14] + (double)ph[15];
    const char* nm[10] = {"kinematics", "inertia+bias", "a0 solve", "collision+rows", "warm/step", "gradient", "hessian", "factor", "linesearch", "integrate"};
    for (int i = 0; i < 10; i++) fprintf(stderr, "  phase %-15s %6.2f%%  %.0f cycles/step/wave\n", nm[i], 100.0 * ph[i] / (tot > 0 ? tot : 1), out4[3] ? 2.0 * ph[i] / out4[3] : 0.0);
    const char* sn[4] = {"chain blocks", "base rows+schur", "reduced system", "chain backsub"};
    for (int i = 0; i < 4; i++) fprintf(stderr, "    solve: %-15s %.0f cycles/step/wave\n", sn[i], out4[3] ? 2.0 * ph[10 + i] / out4[3] : 0.0);
    fprintf(stderr, "    robot-robot broad phase %.0f, narrow phase %.0f cycles/step/wave (inside collision+rows)\n", out4[3] ? 2.0 * ph[14] / out4[3] : 0.0, out4[3] ? 2.0 * ph[15] / out4[3] : 0.0);
  }
#endif
  if (reset) JH_HIP(hipMemset(t->d_stats, 0, 64 * sizeof(int)));
  return JH_OK;
}

extern "C" int jh_tree_set_self_collision(jh_tree* t, int on) {
  JH_REQUIRE(t, "tree_set_self_collision: null pointer");
  JH_REQUIRE(!on || t->npair > 0, "tree_set_self_collision: the model image lists no robot-robot geom pairs");
  t->self_collision = on ? 1 : 0;
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
  if (t->self_collision)
    hipLaunchKernelGGL(k_tree_v4<true>, dim3((N + per_wave - 1) / per_wave), dim3(WAVE), 0, (hipStream_t)stream, t->d_f, t->d_i, t->nf, t->ni, state_in, NX, ctrl, warmstart, N, substeps, state_out, NX,
                       sensors_out, t->ns, t->d_stats, dshift);
  else
    hipLaunchKernelGGL(k_tree_v4<false>, dim3((N + per_wave - 1) / per_wave), dim3(WAVE), 0, (hipStream_t)stream, t->d_f, t->d_i, t->nf, t->ni, state_in, NX, ctrl, warmstart, N, substeps, state_out, NX,
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
    if (t->self_collision)
      hipLaunchKernelGGL(k_tree_v4<true>, dim3((N + per_wave - 1) / per_wave), dim3(WAVE), 0, st, t->d_f, t->d_i, t->nf, t->ni, xin, ld, control, warmstart, N, substeps, states + (size_t)i * NX, T * NX,
                         sensors ? sensors + (size_t)i * t->ns : nullptr, T * t->ns, t->d_stats, dshift);
    else
      hipLaunchKernelGGL(k_tree_v4<false>, dim3((N + per_wave - 1) / per_wave), dim3(WAVE), 0, st, t->d_f, t->d_i, t->nf, t->ni, xin, ld, control, warmstart, N, substeps, states + (size_t)i * NX, T * NX,
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
