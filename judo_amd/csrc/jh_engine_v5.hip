// jh_engine_v5.hip -- leap_cube rollout kernel, second cooperative generation (gfx950): the arithmetic of jh_engine_v2.hip on a register
// diet, so that more than one wave fits on a SIMD.
//
// jh_engine_v2.hip runs one wave per SIMD (487 registers) and is stalled ~75 % of the time on dependent VALU / LDS / DPP chains with
// nothing to switch to.  Same decomposition here -- 16 lanes (one DPP row) per rollout, lane (c,s) owns link s of finger chain c, its
// joint, its actuator, its dof rows and <= 2 contacts -- but everything a lane does not need in every instruction lives in LDS:
//   * per-lane model constants, the lane's spline knots, the warm start, the chain inertia blocks;
//   * a contact slot keeps frame (9), lever arm (3), cone constants (5), aref / jar / jp (9): the finger-side Jacobian columns are
//     recomputed from the joint axes / anchors in LDS wherever they are used (J x, J'f, J'WJ), the cube-side rotational columns from the
//     lever arm and the cube rotation;
//   * every contact contribution to the gradient and to the arrow Hessian is an LDS float atomic (the cube block as well: no 21-entry
//     register partials and no 21 row sums per iteration); row sums remain only for the scalars of the convergence test and line search;
//   * the cube-chain coupling Y = L^-1 Hcb is spread over the chain's four lanes (<= 2 of the 6 columns per lane) instead of being held
//     whole by every lane; the Schur-complement dot products are shared the same way.
// The solver itself (primal Newton, exact line search, warm start, termination) is unchanged: results agree with jh_engine_v2.hip to
// summation order, and the parity suite (tests/test_gpu_leap.py) runs against both.
#include <cstddef>
#include <type_traits>

#include "jh_coop.h"

using namespace jh_eng;
using namespace jh_coop;

namespace {

constexpr int G = 16, RPW = 4, WAVE = 64;
constexpr int NCH = 4, NLK = 4;
#ifndef JH_V5_LSMAX
#define JH_V5_LSMAX 16
#endif
#ifndef JH_V5_LSKINK
#define JH_V5_LSKINK 0  // 1: where the plain safeguarded Newton search is in trouble the line search tries the step lengths at which the slope jumps (below).  Round 4, recorded
                        // inputs (profiles/r04_leap_experiments.txt): it does what the CPU prototype promised -- 3.65 instead of 4.68 slope evaluations per Newton iteration of a
                        // wave, the 9..16-evaluation searches gone (15 % -> 3 % of the wave's searches) -- and the kernel is 2.1 % SLOWER (82.45 against 80.75 ms): a slope
                        // evaluation is ~1 % of an iteration's issue slots, the candidates' registers and the extra block cost more than a fifth of them.  Off.
#endif
#ifndef JH_V5_LSSHRINK
#define JH_V5_LSSHRINK 0.9f
#endif
#ifndef JH_V5_LSREV
#define JH_V5_LSREV 0.03f
#endif
// ---- The Newton iteration's formulation (round 4).  The alternatives below were A/B-measured on the recorded plan inputs within one GPU box each
// (profiles/r04_leap_experiments.txt) and then taken out of the source (the git history has them as JH_V5_* switches):
//  * gradient and Hessian of an iterate from ONE pass over the contacts (joint columns, world force and cone weights computed once, one fence less) instead of a gradient pass,
//    the convergence test, and a Hessian pass that re-used the cone weights: 69.1 -> 68.2 ms, same iterates bit for bit;
//  * the cube block of J'WJ (21 entries) and the cube part of -J'f (6) as sums over the rollout's lanes instead of LDS float atomics: every cube contact of a rollout adds to
//    the same addresses and same-address atomics serialise: 68.2 -> 64.5 ms.  (With packed-fp32 code, rounds 2-3, six atomics for the gradient were 1 % faster than six row
//    sums; without the SLP vectorizer the row sums are 1.7 % faster.)  Cost probes on the atomics that remain -- the instruction issued twice, the second adding zero -- put the
//    24 Hcb atomics of a contact at 3.4 % and the 10 Hbb ones at 1.4 % of the kernel;
//  * the 27 sums as two 16-value reduce-scatters (jh_coop.h row_scatter16: lane l receives entry l; 45 instructions per 16 sums instead of 80): 61.85 -> 61.25 ms;
//  * the Schur complement Hcc - sum over chains of Y'Y never goes back to the LDS: the four chains' terms are summed with qsum4_same, whose result is bit-identical in every
//    lane, and each lane subtracts them from the assembled block with 27 quad broadcasts: 64.6 -> 63.4 ms.  Earlier forms: `S.Hcc[..] -= da` (every read-modify-write its own
//    LDS round trip, fourteen in a row), LDS atomics by the first chain's lanes (79.4 -> 78.4 ms), by every chain's lanes (+9.7 %).  With qsum4's association, which differs
//    from chain to chain, the lanes of a rollout solved four slightly different 6 x 6 systems: +0.3 % iterations and 2 000 rollout-steps at the iteration cap;
//  * J'WJ from world-frame dof columns and A = Fr' W Fr (one symmetric 3 x 3 per contact; the cube's translation columns are unit vectors) instead of frame-space columns
//    times W: 63.3 -> 61.7 ms;
//  * two surviving hand body pairs per level-2 broad-phase pass when both have at most 8 geoms: never applies (the survivors involve the palm's geom groups), +0.8 %.
#ifndef JH_V5_HCSPLIT
#define JH_V5_HCSPLIT 1  // wave-steps without a candidate pair off the cube (two thirds of them on the headline workload) take a copy of the solver compiled without the code for the
                         // hand's own contacts: 61.1 -> 59.4 ms.  The same contacts through the hand-capable copy cost 14 % more (48.5 against 42.5 ms with the hand's broad phase
                         // switched off) for the registers its extra paths hold.  The two copies must give the same bits for a cube contact (a rollout's result may not depend on
                         // its wave-mates: tests/test_gpu_leap.py permutes them): with -ffp-contract=fast they do not -- the backend fuses a product into an add only when the
                         // product has no other use, and the hand paths are such uses -- so this file is built with -ffp-contract=on (fusion within a source expression only:
                         // +0.7 % on its own, jh_engine_v5.flags).
#endif
#ifndef JH_V5_C3CACHE
#define JH_V5_C3CACHE 1  // (round 5; recorded inputs 59.3 -> 58.3 ms, the same iterates bit for bit; profiles/r05_leap_experiments.txt)  1: the joint columns axis_j x (pos - anchor_j) of the FIRST slot's side-B link (12 floats per lane, invariant over the Newton iterations of a step) are computed
                         // once per step and kept in the part of the contact pool's storage the Newton matrices leave free (768 of 820 bytes), three ds_read_b128 per use instead of
                         // 24 loads + 36 multiply-adds, twice per iteration
#endif
#ifndef JH_V5_WORLDROT
#define JH_V5_WORLDROT 1  // (round 6) the cube's three rotational dofs are solved for in WORLD coordinates (w = R w_body) inside the constraint solver: with the cube's isotropic
                          // inertia (a cube: leap_cube, leap_cube_down, caltech_leap_cube; checked at model load) the quadratic term is the same in both frames, and the
                          // rotation columns of a contact become e_q x r -- two non-zeros each -- instead of (R e_q) x r: no read of the cube's rotation matrix, no 3 x 3 product
                          // anywhere in a Newton iteration (gradient torque, Hessian columns, J p of the line search).  The warm start and the integrated acceleration stay in
                          // the body frame (MuJoCo's free-joint convention): two 3 x 3 products per STEP.  Same minimiser, different rounding than the body-frame form.
#endif
#ifndef JH_V5_HCMERGE
#define JH_V5_HCMERGE 1  // (round 6) the hand-capable copy's chain part with one exec-masked region per joint (see the contact pass)
#endif
#ifndef JH_V5_C3PAD
#define JH_V5_C3PAD 278
#endif
#ifndef JH_V5_LSRCP
#define JH_V5_LSRCP 1  // the line search's Newton step divides with v_rcp_f32 (1 ulp) instead of the correctly rounded division sequence (10 instructions per evaluation): -0.2 %
#endif
#ifndef JH_V5_WAVES_PER_EU
#define JH_V5_WAVES_PER_EU 2
#endif
#ifndef JH_V5_WPB
#define JH_V5_WPB 4  // waves per workgroup: they share one LDS copy of the model image and nothing else
#endif
// Waves of a workgroup never exchange data after the image is staged: inside the step loop a "barrier" only has to order one wave's own LDS traffic
// (a wave's LDS instructions execute in issue order), so it is a compiler fence, not an s_barrier -- rollouts in different waves never wait for each other.
// OPAQUE(x): the compiler forgets what it knows about x, so nothing derived from it is hoisted out of the enclosing loop.  Used on the contact slots at the top of
// every Newton iteration: otherwise the slots' Jacobian columns, lever arms and LDS addresses (all invariant over the iterations) are computed once before
// the loop, do not fit in the register file and are spilled and reloaded in every iteration.
#define OPAQUE(x) asm volatile("" : "+v"(x))
#ifndef JH_V5_RSPAD
#define JH_V5_RSPAD 4
#endif
#ifndef JH_V5_LCN
#define JH_V5_LCN 25
#endif
#ifndef JH_V5_BFS
#define JH_V5_BFS 33   // row stride of the LDS copy of the body records (BODY_F = 32 floats each): odd, the four chains' rows in different banks
#endif
#ifndef JH_V5_PAS
#define JH_V5_PAS 9    // row stride of S.pa
#endif
#ifndef JH_V5_PARK
#define JH_V5_PARK 1
#endif
#ifndef JH_V5_OPAQUE_LANE
#define JH_V5_OPAQUE_LANE 1
#endif
#ifndef JH_V5_OPAQUE
#define JH_V5_OPAQUE -1  // -1: per instantiation (2 for both since round 4's -fno-slp-vectorize build: 71.4 against 72.9 ms with 3; rounds 2-3: 3 with the hand's own contacts); 0 = off, 1 = the sides, 2 = + lever arm, 3 = + frame,
                         // 4 = 3 and again before the Hessian assembly and before the line search
#endif
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#ifndef JH_V5_NSLOT
#define JH_V5_NSLOT 2
#endif
#ifndef JH_V5_BIGPROB
#define JH_V5_BIGPROB 0.999999  // how sure the compiler may be that a wave-step stays on the NSLOT copy (block frequencies steer the placement of register spills)
#endif
#ifndef JH_V5_NSBIG
#define JH_V5_NSBIG 3  // 4: 64 contacts, the 16 above the LDS pool in a row of global memory (as jh_engine_v6.hip does): 0 instead of 1.7e-6 contacts dropped per rollout-step on the
                       // recorded headline inputs and 97 % instead of 91 % of the jammed-cube sweep inside the capacity, for +2.8 % on every plan step (81.0 against 78.8 ms): not the default
#endif
#ifndef JH_V5_NS1
#define JH_V5_NS1 0  // 1: a third copy of the constraint rows + Newton solver with ONE slot per lane for the wave-steps in which no rollout of the wave has more than 16 contacts.
                     // Measured in round 4 (recorded inputs; profiles/r04_leap_experiments.txt): 93 % of the wave-steps qualify (5.2 contacts per rollout-step on average, the
                     // wave's maximum 8.2), and the kernel is 1.1 % faster (79.7 against 80.6 ms) -- the second slot's code is skipped by wave-uniform branches already -- while the
                     // bits of a rollout then depend on its wave-mates (the copies contract differently; test_leap_full_size_properties fails).  Off.
#endif
#ifndef JH_V5_NS1PROB
#define JH_V5_NS1PROB 0.5
#endif
constexpr int NSLOT = JH_V5_NSLOT;  // contact slots per lane of the common case: steps with at most 16 * NSLOT contacts in every rollout of the wave
constexpr int NSBIG = JH_V5_NSBIG;  // ... of the rare case (6e-4 of the rollout-steps of the headline workload): the wave runs a second copy of the solver with this many slots
#ifndef JH_V5_NSLDS
#define JH_V5_NSLDS 3
#endif
constexpr int NCP = 16 * (NSBIG < JH_V5_NSLDS ? NSBIG : JH_V5_NSLDS);  // contact pool per rollout in LDS
constexpr int NCAP = 16 * NSBIG;                                          // contacts a rollout can hold: those above the LDS pool live in its row of the global overflow pool
constexpr int NOVF = NCAP - NCP;
#ifndef JH_V5_MAXHIT
#define JH_V5_MAXHIT 64
#endif
constexpr int MAXHIT = JH_V5_MAXHIT;  // broad-phase survivors (candidate geom pairs) per rollout and step; 16 bits each.  More than that are counted as dropped contacts
constexpr int POOL_F = 10;  // pos3, normal3, dist, mu, body, tran
constexpr int MAXG = 80, MAXLG = 8;
constexpr int CUBE = 17;          // contact side codes: 0 = static geometry, 1..16 = finger link (1 + 4*chain + depth), 17 = the cube
constexpr int HITPAIR = 1 << 15;  // broad-phase survivors >= HITPAIR are hand-hand geom pairs (ga << 7 | gb, geom ids < 128), smaller ones are cube-vs-geom
static_assert(MAXG <= 128, "hand geom pairs are packed into 14 bits");
constexpr int MAXBP = 128;         // hand body pairs in the model image (leap_cube 106, caltech_leap_cube 122)
constexpr int MAXBPL = 96;        // hand body pairs whose bounding volumes overlap, per rollout and step (one byte each: pair indices < MAXBP <= 256)
constexpr int NDH = 22 * 23 / 2;  // dense Hessian (packed lower) of a rollout whose contacts couple two finger chains
constexpr int NV = 22, NQ = 23, NU = 16, NS = 31, NS_CALTECH = 23, NX = 45, NMB = 17;
constexpr int NBC = 20;  // hand bodies of the self-collision tables: 0 = static geometry, 1..16 = finger links, 17..19 = further groups of static geometry (engine_model.py)
__device__ __forceinline__ bool static_code(int b) { return b == 0 || b >= NMB; }
#ifndef JH_V5_KNOTS_LDS
#define JH_V5_KNOTS_LDS 0  // 1: the round-1..3 layout (the lane's knots staged in LDS; at most JH_V5_MAXK of them)
#endif
#ifndef JH_V5_MAXK
#define JH_V5_MAXK 8
#endif
constexpr int MAXK = JH_V5_MAXK;

// per-lane model constants staged in LDS (index = lane & 15)
enum { LC_DAMP = 0, LC_KVD, LC_KP, LC_KV, LC_CLO, LC_CHI, LC_CLIM, LC_FL, LC_FB, LC_FD, LC_INVW, LC_LIMITED, LC_LO, LC_HI, LC_LK, LC_LB, LC_SI, LC_IMCK = LC_SI + 5, LC_N = JH_V5_LCN };  // (LC_IMCK: 1 / (the cube's mass or inertia of dof l), lanes 0..5; zero elsewhere)  // (row stride of the LDS table: odd, so that the 16 lanes' rows start in 16 different banks)

#ifdef JH_V5_X_DIET  // occupancy experiments on the cube-only instantiations: the arrays only the hand's own contacts use shrink to stubs
constexpr int RS_NBC = 1, RS_NBPL = 4, RS_NHX = 1, RS_NDH = 4;
#else
constexpr int RS_NBC = NBC, RS_NBPL = MAXBPL, RS_NHX = 6, RS_NDH = NDH;
#endif
struct __attribute__((aligned(16))) RS {  // per-rollout shared state in LDS
  float pa[NMB][JH_V5_PAS];   // body origin (0..2) and joint axis in the world (4..6); odd row stride: lanes reading 16 different bodies hit 16 different banks
  float xR[NMB][9];
  float qv[NV], g[NV], p[NV], ws[NV];
  float acn[6];       // constraint-consistent cube acceleration of this step (every lane integrates the replicated cube state)
  float Mbb[NCH][10];
  float rhs6[6];
  float cmd[4];      // the cube's mass and its three principal inertias: the diagonal of its mass block, next to rhs6 so that the Schur step's reads bring it along
  union {
    struct {
      float bs[RS_NBC][4];   // bounding sphere of the hand bodies (0, 17.. = static geometry, 1..16 = finger links): world centre, radius (the centre is the
                          // bounding box's too; its half sizes and axes come from the model image and the body rotation)
      unsigned short hits[MAXHIT];
      unsigned char bpl[RS_NBPL];
    };
    // the collision arrays are dead from the constraint rows on: the step-level state the Newton loop does not touch is parked here instead of being held in
    // registers (or spilled to scratch memory by the compiler) across the loop.  The joint velocity and the cube's velocity are in qv already.
    struct { float pk_q[G], pk_fs[G], pk_cq[4], pk_acc; };
  };
  union {  // the contact pool is dead once every lane has loaded its slots; the Newton Hessian then reuses its storage
    float pool[NCP][POOL_F];
    struct { float Hcc[21], Hbb[NCH][10], Hcb[NCH][24], Hx[RS_NHX][16]; };  // Hcb[c][j*6+q]: chain column j, cube row q; Hx[pidx(a,b)][ib*4+ia]: block (chain b, chain a)
                                                                         // of a contact-coupled pair of chains a < b (hand self-collision)
    struct { float Hd[RS_NDH], dinv[NV]; };                    // dense path (contacts between two finger chains): packed lower 22 x 22, reciprocal pivots
#if JH_V5_C3CACHE
    struct { float c3pad_[JH_V5_C3PAD]; float c3s[G][12]; };   // behind the Newton matrices (275 floats at most): the first slot's joint columns, per lane; 16-byte aligned rows (static_assert below)
#endif
  };
  int ncon, nhit;
#if JH_V5_RSPAD > 0
  float pad_[JH_V5_RSPAD];  // record size = 16 banks modulo 64: the same field of a wave's four rollouts starts in four disjoint groups of 16 banks
#endif
};

#if JH_V5_C3CACHE
static_assert(offsetof(RS, c3s) % 16 == 0 && JH_V5_C3PAD >= RS_NDH + NV && sizeof(((RS*)nullptr)->c3pad_) + sizeof(((RS*)nullptr)->c3s) <= sizeof(((RS*)nullptr)->pool), "c3 cache: aligned rows behind the Newton matrices, inside the pool's storage");
#endif
struct PoolCtx { RS* S; int* overflow; float* ovf; };  // ovf: this rollout's row of the global overflow pool (NOVF x POOL_F floats), or null

__device__ __forceinline__ void push_contact(const PoolCtx& pc, const float* pos, const float* n, float dist, int body, float mu, float tran) {
  int i = atomicAdd(&pc.S->ncon, 1);
  float* e;
  if (i < NCP) e = pc.S->pool[i];
  else if (NOVF > 0 && pc.ovf && i < NCAP) e = pc.ovf + (i - NCP) * POOL_F;
  else { if (pc.overflow) atomicAdd(pc.overflow, 1); return; }
  e[0] = pos[0]; e[1] = pos[1]; e[2] = pos[2]; e[3] = n[0]; e[4] = n[1]; e[5] = n[2]; e[6] = dist; e[7] = mu; e[8] = __int_as_float(body); e[9] = tran;
}

struct LeapSink {  // contacts of one geom pair: sides packed as A | B << 8; `flip`: the narrow phase ran with the two geoms swapped (normal B -> A)
  PoolCtx pc; int sides; float mu, tran; bool flip;
  __device__ __forceinline__ void push(const float* pos, const float* n, float dist) {
    const float nn[3] = {flip ? -n[0] : n[0], flip ? -n[1] : n[1], flip ? -n[2] : n[2]};
    push_contact(pc, pos, nn, dist, sides, mu, tran);
  }
};

// ------------------------------------------------------------------------------------------------ per-lane contact slot (27 registers)
struct Slot {
  int la, lb;        // the two sides (normal from A to B): la = -1 empty slot; 0 = static geom, 1 + 4*chain + depth = finger link, CUBE
  float fr[9];       // contact frame rows (normal, t1, t2), world
  float rc[3];       // contact point relative to the cube origin, world
  float aref[3], D0, D1, Dm, mu, fri;
  float jar[3], jp[3];
};

// two oriented boxes (centre, rotation, half sizes): false when one of the six face axes separates them (a conservative cull: the nine edge axes are
// left to the narrow phase)
__device__ __forceinline__ bool obb_face_overlap(const float* ca, const float* Ra, const float* ha, const float* cb, const float* Rb, const float* hb) {
  const float d[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
  float C[9];  // C = |Ra' Rb|
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C[3 * i + j] = fabsf(Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]);
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float da = fabsf(d[0] * Ra[i] + d[1] * Ra[3 + i] + d[2] * Ra[6 + i]);
    ok = ok && da <= ha[i] + hb[0] * C[3 * i] + hb[1] * C[3 * i + 1] + hb[2] * C[3 * i + 2];
    const float db = fabsf(d[0] * Rb[i] + d[1] * Rb[3 + i] + d[2] * Rb[6 + i]);
    ok = ok && db <= hb[i] + ha[0] * C[i] + ha[1] * C[3 + i] + ha[2] * C[6 + i];
  }
  return ok;
}

// The three functions below visit the joints above a finger link.  JH_V5_LINKBATCH (round 4): all four joints of the chain are loaded and their columns
// axis_j x (pos - anchor_j) computed UNCONDITIONALLY, the depth of the link only masks what is accumulated.  The per-joint form (`if (j <= dep) { load; compute; }`) put every
// joint's LDS loads under its own exec mask, so each of them was a round trip of its own -- 50 of the 80 `s_waitcnt lgkmcnt(0)` of a Newton iteration sat in these loops;
// the batched form waits once per call and has no branches (the wasted columns of the shallower links are cheaper than the waits).  Same expressions: same bits.
#ifndef JH_V5_LINKBATCH
#define JH_V5_LINKBATCH 1  // (with packed-fp32 code this was 4.5 % SLOWER -- 83.0 against 79.4 ms: the register pairs of the packed operands left no room for 24 joint floats at once;
                           // without the SLP vectorizer it is 1-2.5 % faster: 72.2 against 72.9, and 70.0 against 71.8 ms in the adopted combination)
#endif
__device__ __forceinline__ void link_c3(const RS& S, int ch, const float* pos, float (*c3)[3]) {
#pragma unroll
  for (int j = 0; j < NLK; j++) {
    const float* pj = S.pa[1 + 4 * ch + j];
    const float rb[3] = {pos[0] - pj[0], pos[1] - pj[1], pos[2] - pj[2]};
    cross3(c3[j], pj + 4, rb);
  }
}
// velocity of the point `pos` carried by finger link `code` for the joint-rate vector `vec` (22-vector in LDS), times `sign`, added to w
// (c3c: the lane's cached columns of this link and point, JH_V5_C3CACHE, or null)
__device__ __forceinline__ void load_c3(const float* c3c, float (*c3)[3]) {
  const float4 a = *reinterpret_cast<const float4*>(c3c), b = *reinterpret_cast<const float4*>(c3c + 4), c = *reinterpret_cast<const float4*>(c3c + 8);
  c3[0][0] = a.x; c3[0][1] = a.y; c3[0][2] = a.z; c3[1][0] = a.w; c3[1][1] = b.x; c3[1][2] = b.y; c3[2][0] = b.z; c3[2][1] = b.w; c3[2][2] = c.x; c3[3][0] = c.y; c3[3][1] = c.z; c3[3][2] = c.w;
}
__device__ __forceinline__ void link_vel(const RS& S, int code, const float* pos, const float* vec, float sign, float* w, const float* c3c = nullptr) {
  const int ch = (code - 1) >> 2, dep = (code - 1) & 3;
#if JH_V5_LINKBATCH
  float c3[NLK][3]; if (c3c) load_c3(c3c, c3); else link_c3(S, ch, pos, c3);
#pragma unroll
  for (int j = 0; j < NLK; j++) {
    const float vj = vec[6 + 4 * ch + j];  // (loaded whatever the depth: a valid address, and a conditional load is an exec-mask round trip)
    const float xj = j <= dep ? sign * vj : 0.f;
    w[0] = fmaf(c3[j][0], xj, w[0]); w[1] = fmaf(c3[j][1], xj, w[1]); w[2] = fmaf(c3[j][2], xj, w[2]);
  }
#else
#pragma unroll
  for (int j = 0; j < NLK; j++) if (j <= dep) {
    const float* pj = S.pa[1 + 4 * ch + j];
    const float rb[3] = {pos[0] - pj[0], pos[1] - pj[1], pos[2] - pj[2]}; float c3[3];
    cross3(c3, pj + 4, rb);
    const float xj = sign * vec[6 + 4 * ch + j];
    w[0] = fmaf(c3[0], xj, w[0]); w[1] = fmaf(c3[1], xj, w[1]); w[2] = fmaf(c3[2], xj, w[2]);
  }
#endif
}
// -J'F for the joints of finger link `code` (F = world force on side B; sign = +1 for side B, -1 for side A): LDS float atomics into g
__device__ __forceinline__ void link_force(RS& S, int code, const float* pos, const float* Fw, float sign) {
  const int ch = (code - 1) >> 2, dep = (code - 1) & 3;
#if JH_V5_LINKBATCH
  float c3[NLK][3]; link_c3(S, ch, pos, c3);
  float v[NLK];
#pragma unroll
  for (int j = 0; j < NLK; j++) v[j] = -sign * dot3(c3[j], Fw);
#pragma unroll
  for (int j = 0; j < NLK; j++) if (j <= dep) atomicAdd(&S.g[6 + 4 * ch + j], v[j]);
#else
#pragma unroll
  for (int j = 0; j < NLK; j++) if (j <= dep) {
    const float* pj = S.pa[1 + 4 * ch + j];
    const float rb[3] = {pos[0] - pj[0], pos[1] - pj[1], pos[2] - pj[2]}; float c3[3];
    cross3(c3, pj + 4, rb);
    atomicAdd(&S.g[6 + 4 * ch + j], -sign * dot3(c3, Fw));
  }
#endif
}
// Jacobian columns (contact frame) of the joints of finger link `code`, times `sign`, added to Jb[j] (j = depth in the chain)
__device__ __forceinline__ void link_cols(const RS& S, int code, const float* pos, const float* fr, float sign, float (*Jb)[3]) {
  const int ch = (code - 1) >> 2, dep = (code - 1) & 3;
#if JH_V5_LINKBATCH
  float c3[NLK][3]; link_c3(S, ch, pos, c3);
#pragma unroll
  for (int j = 0; j < NLK; j++) {
    const float sg = j <= dep ? sign : 0.f;
    Jb[j][0] = fmaf(sg, dot3(fr, c3[j]), Jb[j][0]); Jb[j][1] = fmaf(sg, dot3(fr + 3, c3[j]), Jb[j][1]); Jb[j][2] = fmaf(sg, dot3(fr + 6, c3[j]), Jb[j][2]);
  }
#else
#pragma unroll
  for (int j = 0; j < NLK; j++) if (j <= dep) {
    const float* pj = S.pa[1 + 4 * ch + j];
    const float rb[3] = {pos[0] - pj[0], pos[1] - pj[1], pos[2] - pj[2]}; float c3[3];
    cross3(c3, pj + 4, rb);
    Jb[j][0] = fmaf(sign, dot3(fr, c3), Jb[j][0]); Jb[j][1] = fmaf(sign, dot3(fr + 3, c3), Jb[j][1]); Jb[j][2] = fmaf(sign, dot3(fr + 6, c3), Jb[j][2]);
  }
#endif
}

// contact-frame image of the relative point velocity (side B minus side A) for the generalised velocity whose cube part is (xl = linear, world;
// wang = R_cube * angular part, world) and whose finger part is `vec` (22-vector in LDS)
template <bool SELF>
__device__ __forceinline__ void slot_Jx(const Slot& s, const RS& S, const float* qcpos, const float* xl, const float* wang, const float* vec, float* out, const float* c3c = nullptr) {
  float w[3] = {0.f, 0.f, 0.f};
  const float pos[3] = {s.rc[0] + qcpos[0], s.rc[1] + qcpos[1], s.rc[2] + qcpos[2]};
  if (!SELF || s.la == CUBE) { float wx[3]; cross3(wx, wang, s.rc); w[0] = -(xl[0] + wx[0]); w[1] = -(xl[1] + wx[1]); w[2] = -(xl[2] + wx[2]); }
  else if (SELF && s.la > 0) link_vel(S, s.la, pos, vec, -1.f, w);
  if (s.lb > 0) link_vel(S, s.lb, pos, vec, 1.f, w, c3c);
  out[0] = dot3(s.fr, w); out[1] = dot3(s.fr + 3, w); out[2] = dot3(s.fr + 6, w);
}

// slot_Jx for a cube contact of the copy without the hand's own contacts, FIRST slot, without a branch: the finger part is computed whatever the slot holds (an empty slot or a
// contact between the cube and the static geometry reads valid addresses -- link code 0 gives chain -1, entries 2..5 of `vec`) and dropped by a select; an empty slot's frame is
// zero, so its image is zero.  The same fused multiply-adds in the same order as slot_Jx: same bits.
__device__ __forceinline__ void slot_Jx_first(const Slot& s, const float* xl, const float* wang, const float* vec, float* out, const float* c3c) {
  float wx[3]; cross3(wx, wang, s.rc);
  const float w0[3] = {-(xl[0] + wx[0]), -(xl[1] + wx[1]), -(xl[2] + wx[2])};
  const int ch = (s.lb - 1) >> 2, dep = (s.lb - 1) & 3;
  float c3[NLK][3]; load_c3(c3c, c3);
  float w[3] = {w0[0], w0[1], w0[2]};
#pragma unroll
  for (int j = 0; j < NLK; j++) {
    const float vj = vec[6 + 4 * ch + j];
    const float xj = j <= dep ? vj : 0.f;
    w[0] = fmaf(c3[j][0], xj, w[0]); w[1] = fmaf(c3[j][1], xj, w[1]); w[2] = fmaf(c3[j][2], xj, w[2]);
  }
  const bool lk = s.lb > 0;
  w[0] = lk ? w[0] : w0[0]; w[1] = lk ? w[1] : w0[1]; w[2] = lk ? w[2] : w0[2];
  out[0] = dot3(s.fr, w); out[1] = dot3(s.fr + 3, w); out[2] = dot3(s.fr + 6, w);
}

struct DofRows { float fl, fD, fR, faref, lims, laref, lD, jf, jl, pf, pl; };  // fR = 1/fD

__device__ __forceinline__ float cone_cost(const Slot& s) {
  float f[3], W[6];
  const float D[3] = {s.D0, s.D1, s.D1};
  return cone_eval(s.jar, D, s.Dm, s.mu, s.fri, f, W);
}

__device__ __forceinline__ float dof_rows_cost(const DofRows& dr) {
  float cs = 0.f;
  if (dr.fl > 0.f) {
    const float R = dr.fR, x = dr.jf, fl = dr.fl;
    if (x <= -R * fl) cs += -0.5f * R * fl * fl - fl * x;
    else if (x >= R * fl) cs += -0.5f * R * fl * fl + fl * x;
    else cs += 0.5f * dr.fD * x * x;
  }
  if (dr.lims != 0.f && dr.jl < 0.f) cs += 0.5f * dr.lD * dr.jl * dr.jl;
  return cs;
}

// slope and curvature of the lane's rows along the search direction at step al (line search)
// (`used`: bit k set iff some lane of the WAVE has a contact in slot k.  An empty slot is all zeros -- its cone test reads `top`, its terms are zero -- so a slot is evaluated
// by every lane or skipped by the whole wave: no exec-masked region inside the line search's loop.)
template <int NS>
__device__ __forceinline__ void lane_rows_dir(const Slot* sl, const DofRows& dr, float al, float* d1, float* d2, unsigned used) {
  float g1 = 0.f, g2 = 0.f;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    if (k > 0 && !((used >> k) & 1u)) continue;
    const float* jp = sl[k].jp;
    const float jar[3] = {fmaf(al, jp[0], sl[k].jar[0]), fmaf(al, jp[1], sl[k].jar[1]), fmaf(al, jp[2], sl[k].jar[2])};
    const float D[3] = {sl[k].D0, sl[k].D1, sl[k].D1};
    cone_dir(jar, jp, D, sl[k].Dm, sl[k].mu, sl[k].fri, &g1, &g2);
  }
  {  // the own dof's friction-loss and limit rows, as selects (the same fused multiply-adds as the branches they replace: same bits, a dozen exec-mask instructions fewer per evaluation)
    const float D = dr.fD, jp = dr.pf, x = fmaf(al, jp, dr.jf), fl = dr.fl, lim = dr.fR * fl;
    const bool has = fl > 0.f, lo_ = x <= -lim, hi_ = x >= lim, mid = has & !lo_ & !hi_;
    const float t = lo_ ? -fl : (hi_ ? fl : D * x);
    g1 = has ? fmaf(t, jp, g1) : g1;
    g2 = mid ? fmaf(D * jp, jp, g2) : g2;
    const float jl = dr.pl, xl_ = fmaf(al, jl, dr.jl);
    const bool lim_on = (dr.lims != 0.f) & (xl_ < 0.f);
    g1 = lim_on ? fmaf(dr.lD * xl_, jl, g1) : g1;
    g2 = lim_on ? fmaf(dr.lD * jl, jl, g2) : g2;
  }
  *d1 = g1; *d2 = g2;
}

// 4x4 Cholesky (packed lower) + triangular solves on registers; the diagonal is kept as its reciprocal
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
// element j (run-time, wave-nonuniform) of a 4-array held in registers: compare-and-select
__device__ __forceinline__ float sel4(const float* a, int j) { return j == 0 ? a[0] : (j == 1 ? a[1] : (j == 2 ? a[2] : a[3])); }

// the same as two levels of selects with opaque results: the compiler cannot turn the chain into a branch on j and copy what follows into both sides (a quad holds all four j:
// it would run every copy)
__device__ __forceinline__ float sel4o(const float* a, int j) {
  float lo = (j & 1) ? a[1] : a[0], hi = (j & 1) ? a[3] : a[2];
  asm volatile("" : "+v"(lo), "+v"(hi));
  float r = (j & 2) ? hi : lo;
  asm volatile("" : "+v"(r));
  return r;
}

// Elimination order of the four finger chains when contacts couple them (hand self-collision).  `cmask` has bit 4a + b for every coupled pair a < b.
// A coupling graph without a cycle is eliminated leaf first: every chain is eliminated when at most one of its neighbours is left (its parent, which
// receives the Schur update); `level` is the stage at which chain c goes, `maxlev` the last stage of the rollout.  Returns false for a graph with a
// cycle (elimination would create a block between two chains that had none): those rollouts take the dense direction.
__device__ __forceinline__ int pidx(int a, int b) { return a * (7 - a) / 2 + (b - a - 1); }  // index of the chain pair a < b: 01 02 03 12 13 23
__device__ __forceinline__ bool chain_elim_order(int cmask, int c, int& level, int& parent, int& maxlev) {
  int adj[NCH] = {0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < NCH; a++)
#pragma unroll
    for (int b = a + 1; b < NCH; b++) if ((cmask >> (4 * a + b)) & 1) { adj[a] |= 1 << b; adj[b] |= 1 << a; }
  int rem = 0xF; bool ok = true;
  level = 0; parent = -1; maxlev = 0;
#pragma unroll
  for (int st = 0; st < NCH; st++) {
    int elig = 0, pr[NCH] = {-1, -1, -1, -1};
#pragma unroll
    for (int a = 0; a < NCH; a++) if ((rem >> a) & 1) {
      const int nbm = adj[a] & rem, deg = __popc(nbm);
      if (deg == 0) elig |= 1 << a;
      else if (deg == 1) {
        const int nb = __ffs(nbm) - 1;
        const int anb = nb == 0 ? adj[0] : (nb == 1 ? adj[1] : (nb == 2 ? adj[2] : adj[3]));
        if (__popc(anb & rem) > 1 || a < nb) { elig |= 1 << a; pr[a] = nb; }
      }
    }
    if (rem != 0 && elig == 0) ok = false;
    if (elig != 0) maxlev = st;
#pragma unroll
    for (int a = 0; a < NCH; a++) if (((elig >> a) & 1) && a == c) { level = st; parent = pr[a]; }
    rem &= ~elig;
  }
  return ok;
}

// ------------------------------------------------------------------------------------------------ the kernel
template <bool MATERIALIZE, int WPB, bool SELF>
#ifdef JH_V5_NUM_VGPR  // (occupancy experiments: a register budget independent of what the LDS footprint allows)
#define JH_V5_REGATTR __attribute__((amdgpu_num_vgpr(JH_V5_NUM_VGPR)))
#else
#define JH_V5_REGATTR
#endif
__global__ __launch_bounds__(WAVE * WPB, JH_V5_WAVES_PER_EU) JH_V5_REGATTR void k_leap_v5(const float* __restrict__ gF, const int* __restrict__ gI, const float* __restrict__ x0, int x0_batched,
                                                   const float* __restrict__ nominal, const float* __restrict__ noise, int ldn,
                                                   const float* __restrict__ sigma, const float* __restrict__ W, const float* __restrict__ lohi,
                                                   const float* __restrict__ tp, int N, int n_offset, int H, int K, float* __restrict__ costs,
                                                   float* __restrict__ knots_out, const float* __restrict__ controls, float* __restrict__ states,
                                                   float* __restrict__ sensors, int* __restrict__ stats, int dshift, float* __restrict__ trace, float* __restrict__ ovf_all) {
#ifdef JH_V5_X_DYNRS  // (occupancy experiments: the compiler does not see the per-rollout LDS, so the register budget follows JH_V5_WAVES_PER_EU alone)
  extern __shared__ __attribute__((aligned(16))) unsigned char dynRS[];
  RS* sRS = reinterpret_cast<RS*>(dynRS);
#else
  __shared__ RS sRS[RPW * WPB];
#endif
  __shared__ float sBody[16 * JH_V5_BFS];
  __shared__ float sTp[16];
  __shared__ float sGeomF[MAXG * GEOM_F];
  __shared__ int sGeomI[MAXG * GEOM_I];
  __shared__ int sLaneG[16 * MAXLG];
  __shared__ float sLane[16 * LC_N];
  __shared__ int sBP[SELF ? 2 * MAXBP : 4];    // hand body pairs (side A, side B): every geom of A is a candidate against every geom of B
  __shared__ int sBG[SELF ? 2 * NBC : 4];    // per hand body: first collision geom, number of geoms (contiguous in the geom table)
  __shared__ float sBB[SELF ? NBC * 8 : 4];  // per hand body: bounding-box centre (body frame; static geometry: world), bounding radius, half sizes
#if JH_V5_KNOTS_LDS
  __shared__ float sKnAll[MAXK * WAVE * WPB];
#endif
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, r = lane >> 4;
  int l = lane & 15, c = l >> 2, s = l & 3;  // (not const: see the top of the step loop)
  RS& S = sRS[wv * RPW + r];
#if JH_V5_KNOTS_LDS
  float* sKn = sKnAll + wv * (MAXK * WAVE);
#endif
  const int nmI = gI[0], nblkI = gI[1], nuI = gI[4], ngI = gI[5], nsiteI = gI[6], nsI = gI[7], oRef = gI[18];
  const int oBodyF = HEADER_F, oDofF = oBodyF + nmI * BODY_F, oActF = oDofF + gI[2] * DOF_F, oGeomF = oActF + nuI * ACT_F, oSiteF = oGeomF + ngI * GEOM_F;
  const int oGeomI = HEADER_I + nmI * BODY_I + nblkI * BLOCK_I + nuI * ACT_I, oSiteI = oGeomI + ngI * GEOM_I;
  const int oLane = gI[11], lgm = gI[12];
  const int oBP = gI[15], oBS = gI[16], nBP = gI[17], oBG = oBP + 2 * nBP;  // hand self-collision: body pairs, body bounding volumes, per-body geom ranges
  for (int i = tid; i < 16 * BODY_F; i += WAVE * WPB) sBody[(i / BODY_F) * JH_V5_BFS + i % BODY_F] = gF[oBodyF + BODY_F + i];
  for (int i = tid; i < ngI * GEOM_F; i += WAVE * WPB) sGeomF[i] = gF[oGeomF + i];
  for (int i = tid; i < ngI * GEOM_I; i += WAVE * WPB) sGeomI[i] = gI[oGeomI + i];
  for (int i = tid; i < 16 * lgm; i += WAVE * WPB) sLaneG[i] = gI[oLane + i];
  if (!MATERIALIZE && tid < 9) sTp[tid] = tp[tid];
  if constexpr (SELF) {
    for (int i = tid; i < 2 * nBP && i < 2 * MAXBP; i += WAVE * WPB) sBP[i] = gI[oBP + i];
    for (int i = tid; i < 2 * NBC; i += WAVE * WPB) sBG[i] = gI[oBG + i];
    for (int i = tid; i < NBC * 8; i += WAVE * WPB) sBB[i] = gF[oBS + i];
  }
  if (tid < 16) {
    const float* df = gF + oDofF + (6 + l) * DOF_F; const float* af = gF + oActF + l * ACT_F;
    float* lc = sLane + l * LC_N;
    lc[LC_DAMP] = df[DF_DAMP]; lc[LC_KVD] = df[DF_KV]; lc[LC_FL] = df[DF_FL]; lc[LC_FB] = df[DF_FB]; lc[LC_FD] = df[DF_FD]; lc[LC_INVW] = df[DF_INVW];
    lc[LC_LIMITED] = df[DF_LIMITED]; lc[LC_LO] = df[DF_LO]; lc[LC_HI] = df[DF_HI]; lc[LC_LK] = df[DF_LK]; lc[LC_LB] = df[DF_LB];
    for (int k = 0; k < 5; k++) lc[LC_SI + k] = df[DF_SOLIMP + k];
    lc[LC_KP] = af[AF_KP]; lc[LC_KV] = af[AF_KV]; lc[LC_CLIM] = af[AF_CLIM]; lc[LC_CLO] = af[AF_CLO]; lc[LC_CHI] = af[AF_CHI];
    // (the convergence test's scale of the cube's dofs: read from here in every iteration -- kept in a register across the loop it is spilled to scratch memory)
    lc[LC_IMCK] = l < 6 ? 1.f / (l < 3 ? gF[HF_CMASS] : gF[HF_CINERTIA + (l < 3 ? 0 : l - 3)]) : 0.f;
  }
  const float* lc = sLane + l * LC_N;
  // Rollout handled by this row of 16 lanes.  A launch too small to fill the GPU (`dshift` > 0, chosen by the launcher) gives a wave 4 >> dshift rollouts instead of four and
  // lets 1 << dshift rows compute the same one: the copies run the same arithmetic (a rollout's result does not depend on its wave-mates), only the first writes, and the
  // wave no longer waits for the slowest of four different Newton solves in every step -- the latency mode of small shards and of the reference's 32-rollout configurations.
  const int n = ((blockIdx.x * WPB + wv) << (2 - dshift)) + (r >> dshift);
  const bool live = n < N && (r & ((1 << dshift) - 1)) == 0;
  const int nc = n < N ? n : N - 1;
  const float h = gF[HF_DT], impratio = gF[HF_IMPRATIO], tol = gF[HF_TOL], lstol = gF[HF_LSTOL]; const int cap = (int)gF[HF_MAXITER];
  const float grav[3] = {gF[HF_GRAV], gF[HF_GRAV + 1], gF[HF_GRAV + 2]};
  const float cmass = gF[HF_CMASS], cI[3] = {gF[HF_CINERTIA], gF[HF_CINERTIA + 1], gF[HF_CINERTIA + 2]};
  const float chs[3] = {gF[HF_CSIZE], gF[HF_CSIZE + 1], gF[HF_CSIZE + 2]}, crb = gF[HF_CRBOUND], ctran = gF[HF_CTRAN];
  const float cK = gF[HF_CK], cB = gF[HF_CB];
  float csi[5]; for (int k = 0; k < 5; k++) csi[k] = gF[HF_SOLIMP + k];
  // own cube dof (lanes 0..5): inertia and its inverse; zero elsewhere
  const float mck = (l < 3 ? cmass : (l == 3 ? cI[0] : (l == 4 ? cI[1] : (l == 5 ? cI[2] : 0.f))));
  // ---- state: own joint + replicated cube
  float q, qd, qc[7], vc[6];
  {
    const float* xi = x0 + ((MATERIALIZE && x0_batched) ? (size_t)nc * NX : 0);
    for (int k = 0; k < 7; k++) qc[k] = xi[k];
    for (int k = 0; k < 6; k++) vc[k] = xi[NQ + k];
    q = xi[7 + l]; qd = xi[NQ + 6 + l];
  }
  // ---- own actuator's spline knots (fused mode): clip(nominal + sigma*noise) -> LDS; global sample 0 keeps the nominal
  // The knots are NOT kept on chip: 8 knots x 16 actuators are 512 B of LDS per rollout (a sixth of its state), and the step loop needs them once per step -- K
  // L2-resident loads of the noise matrix and of the nominal, issued ahead of the kinematics (`knot_at` below); the LDS goes to the contact pool instead.
  auto knot_at = [&](int k, int ll, int ncc) -> float {  // clip(nominal + sigma * noise) of actuator ll, knot k, rollout ncc (global sample 0 keeps the nominal)
    const int i = k * NU + ll;
    float v = nominal[i];
    if (n_offset + ncc != 0) v = fmaf(sigma[i], noise[(size_t)i * ldn + ncc], v);
    return jh_clampf(v, lohi[ll], lohi[NU + ll]);
  };
  if (!MATERIALIZE) {
#if JH_V5_KNOTS_LDS
    for (int k = 0; k < MAXK; k++) sKn[k * WAVE + lane] = k < K ? knot_at(k, l, nc) : 0.f;
#endif
    if (knots_out && live) for (int k = 0; k < K; k++) knots_out[(size_t)(k * NU + l) * ldn + n] = knot_at(k, l, nc);
  }
  S.ws[6 + l] = 0.f; if (l < 6) S.ws[l] = 0.f;
  if (l < 4) S.cmd[l] = l == 0 ? cmass : (l == 1 ? cI[0] : (l == 2 ? cI[1] : cI[2]));
  int n_iters = 0, n_maxed = 0, n_wave_iters = 0;  // (the last: iterations this wave ran -- per step the maximum over its four rollouts)
  float acc = 0.f;
#ifdef JH_V5_TICKS  // shader-clock totals per phase (diagnostic builds; tools/diag/profile_v5.py): 0 kinematics+dynamics, 1 broad phase, 2 narrow phase, 3 rows+warm start,
                     // 4 gradient, 5 Newton matrix, 6 factorisation+direction, 7 line search+step and integration
  long long cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = clock64();
#define V5_TICK(slot) { long long t__ = clock64(); cyc[slot] += t__ - t0; t0 = t__; }
  // finer split (tools/diag/profile_v5b.py reads stats[384..]): 8 convergence test + Hessian initialisation, 9 Schur complement + 6x6 + back-substitution (the rest of 6 is the chain
  // blocks), 10 line-search set-up (M p, J p), 11 the slope evaluations, 12 step
#else
#define V5_TICK(slot)
#endif
#ifdef JH_V5_COUNT
  int cnt_dense = 0, cnt_it = 0, cnt_l2 = 0, cnt_bp = 0, cnt_hh = 0, cnt_cls[4] = {0, 0, 0, 0};
#endif
  __syncthreads();  // the only workgroup barrier: the model image is staged

  for (int hh = 0; hh < H; hh++) {
#if JH_V5_OPAQUE_LANE
    // the model constants of a lane (sBody, sLane, ...) do not change over the steps: left alone the compiler loads them once before the loop, runs out of
    // registers and reloads them from scratch memory in every step instead of from LDS
    OPAQUE(l); c = l >> 2; s = l & 3;
    const float* lc = sLane + l * LC_N;
#endif
    // ================================================================ controls
    float u;
    if (MATERIALIZE) u = controls[((size_t)nc * H + hh) * NU + l];
    else {
      u = 0.f;
#if JH_V5_KNOTS_LDS
      for (int k = 0; k < K && k < MAXK; k++) u = fmaf(W[hh * K + k], sKn[k * WAVE + lane], u);
#else
      {  // (the rollout index is recomputed from an opaque copy of the lane id: held across the step loop it would cost a register the loop does not have)
        int lo_ = lane; OPAQUE(lo_);
        const int n_ = ((blockIdx.x * WPB + wv) << (2 - dshift)) + ((lo_ >> 4) >> dshift), nc_ = n_ < N ? n_ : N - 1;
        for (int k = 0; k < K; k++) u = fmaf(W[hh * K + k], knot_at(k, l, nc_), u);
      }
#endif
    }
    // ================================================================ kinematics (each lane walks its chain up to its own link)
    float Mrow[NLK], fs_own, a0_own;
    {
      float ax[NLK][3], og[NLK][3], Rown[9], pown[3];
      {
        float nn = rsqrtf(qc[3] * qc[3] + qc[4] * qc[4] + qc[5] * qc[5] + qc[6] * qc[6]);
        qc[3] *= nn; qc[4] *= nn; qc[5] *= nn; qc[6] *= nn;
        float Rc[9]; quat2mat(Rc, qc + 3);
        float P[3] = {0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        float sn_own, cs_own; sincosf(q, &sn_own, &cs_own);
#pragma unroll
        for (int j = 0; j < NLK; j++) {
          const float* bf = sBody + (4 * c + j) * JH_V5_BFS;
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
        for (int k = 0; k < 3; k++) { S.pa[1 + l][k] = pown[k]; S.pa[1 + l][4 + k] = s == 0 ? ax[0][k] : (s == 1 ? ax[1][k] : (s == 2 ? ax[2][k] : ax[3][k])); }
        for (int k = 0; k < 9; k++) S.xR[1 + l][k] = Rown[k];
        if (l == 0) { for (int k = 0; k < 3; k++) S.pa[0][k] = qc[k]; for (int k = 0; k < 9; k++) S.xR[0][k] = Rc[k]; S.ncon = 0; S.nhit = 0; }
        S.qv[6 + l] = qd;
        if (l < 6) S.qv[l] = vc[l];
      }
      // fused mode with a trace buffer (jh_rollout_cost_traced): the five trace sites of this forward pass -- what the materialise mode writes as sensors 16..30 --
      // for EVERY rollout: 60 B per rollout-step, 250 MB per plan step of the headline workload, and `Controller.traces` becomes a gather instead of a re-rollout
#ifndef JH_V5_X_NOTRACE  // (A/B probe, profiles/r05_trace_ab.txt: the kernel without its trace rows -- what re-rolling the E <= 5 elites instead would save in this launch)
      if (!MATERIALIZE && trace && nsI == NS) {
        WSYNC();
        if (live && l < nsiteI && l < 5) {
          int b = gI[oSiteI + l]; float p3[3]; mulMV(p3, S.xR[b], gF + oSiteF + l * SITE_F);
          float* tr = trace + ((size_t)n * H + hh) * 15 + 3 * l;
          for (int k = 0; k < 3; k++) tr[k] = p3[k] + S.pa[b][k];
        }
      }
#endif
      // sensors of this forward pass (materialise mode): 16 joint positions, then 5 site positions
      if (MATERIALIZE && sensors) {
        float* y = sensors + ((size_t)nc * H + hh) * nsI;
        if (live) y[l] = q;
        WSYNC();
        if (nsI == NS) {  // leap_cube / leap_cube_down: five site positions
          if (live && l < nsiteI && l < 5) {
            int b = gI[oSiteI + l]; float p3[3]; mulMV(p3, S.xR[b], gF + oSiteF + l * SITE_F);
            for (int k = 0; k < 3; k++) y[16 + 3 * l + k] = p3[k] + S.pa[b][k];
          }
        } else if (live && l < 7) {  // caltech_leap_cube: cube position in the (world-fixed) grasp-site frame, cube orientation relative to the goal body's
          const float* rf = gF + oRef;  // [p_ref(3), R_ref(9), q_ref(4)]
          if (l < 3) y[16 + l] = rf[3 + l] * (qc[0] - rf[0]) + rf[6 + l] * (qc[1] - rf[1]) + rf[9 + l] * (qc[2] - rf[2]);
          else {
            const float a0 = rf[12], a1 = -rf[13], a2 = -rf[14], a3 = -rf[15], b0 = qc[3], b1 = qc[4], b2 = qc[5], b3 = qc[6];  // conj(q_ref) * q_cube
            const float qo[4] = {a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2, a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1, a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0};
            y[16 + l] = l == 3 ? qo[0] : (l == 4 ? qo[1] : (l == 5 ? qo[2] : qo[3]));
          }
        }
      }
      // ================================================================ chain dynamics: inertia block, bias, smooth force
      {
        const float* bf = sBody + (4 * c + s) * JH_V5_BFS;
        float Rk[9], rr[3], cs3[3]; mulMM(Rk, Rown, bf + BF_IR); mulMV(rr, Rown, bf + BF_IPOS);
        for (int k = 0; k < 3; k++) cs3[k] = pown[k] + rr[k];
        const float mass = bf[BF_MASS]; const float* di = bf + BF_INERTIA;
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
        float t1[3], t2[3], t3[3], ac3[3];
        cross3(t1, wv, rr); cross3(t2, wv, t1); cross3(t3, al, rr);
        for (int k = 0; k < 3; k++) ac3[k] = ao[k] + t3[k] + t2[k];
        float Iw[3], Ia[3], gy[3]; inertia_mul(Iw, Rk, di, wv); inertia_mul(Ia, Rk, di, al); cross3(gy, wv, Iw);
        float Fk[3] = {mass * ac3[0], mass * ac3[1], mass * ac3[2]}, Nk[3] = {Ia[0] + gy[0], Ia[1] + gy[1], Ia[2] + gy[2]};
        float bias[NLK], Mc[10];
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
        float cc = u; if (lc[LC_CLIM] != 0.f) cc = jh_clampf(cc, lc[LC_CLO], lc[LC_CHI]);
        fs_own = -lc[LC_DAMP] * qd - bown + lc[LC_KP] * (cc - q) - lc[LC_KV] * qd;
        float x4[4], L[10];
#pragma unroll
        for (int j = 0; j < NLK; j++) x4[j] = quad_get(fs_own, j);
        for (int k = 0; k < 10; k++) L[k] = Mc[k];
        float inv4[4]; chol4(L, inv4); fwd4(L, inv4, x4); bwd4(L, inv4, x4);
        a0_own = sel4(x4, s);
#pragma unroll
        for (int j = 0; j < NLK; j++) { float v = 0.f;
#pragma unroll
          for (int i = 0; i < NLK; i++) if (i == s) v = Mc[i >= j ? tri(i, j) : tri(j, i)];
          Mrow[j] = v; }
        if (s == 0) for (int k = 0; k < 10; k++) S.Mbb[c][k] = Mc[k];
      }
    }
    // free cube: M = diag(m,m,m,I); own component of the unconstrained acceleration (lanes 0..5)
    float a0c_own = 0.f;
    {
      float Icw[3] = {cI[0] * vc[3], cI[1] * vc[4], cI[2] * vc[5]}, gc[3]; cross3(gc, vc + 3, Icw);
#if JH_V5_WORLDROT
      (void)gc;  // isotropic inertia: w x (I w) = 0 exactly -- the body-frame expression only carries its own rounding noise
      a0c_own = l < 3 ? (l == 0 ? grav[0] : (l == 1 ? grav[1] : grav[2])) : 0.f;
#else
      a0c_own = l < 3 ? (l == 0 ? grav[0] : (l == 1 ? grav[1] : grav[2])) : (l == 3 ? -gc[0] / cI[0] : (l == 4 ? -gc[1] / cI[1] : (l == 5 ? -gc[2] / cI[2] : 0.f)));
#endif
    }
    WSYNC();
    V5_TICK(0)
    // ================================================================ collision: broad phase (cube vs the lane's geoms; hand body pairs), balanced narrow phase
    bool hand_hits = false;  // this rollout has candidate geom pairs of the hand against itself or the static geometry this step (a superset of its contacts off the cube)
    {
      int nh = 0;
      float Rc[9]; for (int k = 0; k < 9; k++) Rc[k] = S.xR[0][k];
      const float* pw = S.pa[1 + l]; const float* Rw = S.xR[1 + l];
      if constexpr (SELF) {  // bounding volume of the own link for the hand's self-collision
        const float* b8 = sBB + 8 * (1 + l); float cw[3]; mulMV(cw, Rw, b8);
        S.bs[1 + l][0] = cw[0] + pw[0]; S.bs[1 + l][1] = cw[1] + pw[1]; S.bs[1 + l][2] = cw[2] + pw[2]; S.bs[1 + l][3] = b8[3];
        if (l < 4) { S.bs[0][l] = sBB[l]; for (int b = NMB; b < NBC; b++) S.bs[b][l] = sBB[8 * b + l]; }
      }
      // (round 6: shaped for latency, the same tests.  The loop was a chain of eight dependent LDS round trips per geom -- its id, its body, its position, ... each
      // behind the test before it -- and nine exec-masked regions.  Now: the lane's geom ids in one batch of loads, a geom's record in one more, the sphere test and the
      // cube-frame box test as straight-line code, one region for the geom's own box.)
      float pwr[3], Rwr[9];
      for (int k = 0; k < 3; k++) pwr[k] = pw[k];
      for (int k = 0; k < 9; k++) Rwr[k] = Rw[k];
      int gids[MAXLG];
#pragma unroll
      for (int i = 0; i < MAXLG; i++) gids[i] = i < lgm ? sLaneG[l * lgm + i] : -1;
#pragma unroll
      for (int i = 0; i < MAXLG; i++) {
        if (i >= lgm) break;
        const int gid = gids[i], g0 = gid < 0 ? 0 : gid;
        const float* gf = sGeomF + g0 * GEOM_F;
        const int gbody = sGeomI[g0 * GEOM_I], gtype = sGeomI[g0 * GEOM_I + 1];
        const float lp[3] = {gf[GF_POS], gf[GF_POS + 1], gf[GF_POS + 2]}, rb = gf[GF_RBOUND];
        float gp[3]; mulMV(gp, Rwr, lp); gp[0] += pwr[0]; gp[1] += pwr[1]; gp[2] += pwr[2];
        if (gbody < 0) { gp[0] = lp[0]; gp[1] = lp[1]; gp[2] = lp[2]; }
        const float dc[3] = {gp[0] - qc[0], gp[1] - qc[1], gp[2] - qc[2]}, rs = rb + crb;
        float cl[3]; mulMTV(cl, Rc, dc);
        bool hit = (gid >= 0) & (dot3(dc, dc) <= rs * rs) & (fabsf(cl[0]) <= chs[0] + rb) & (fabsf(cl[1]) <= chs[1] + rb) & (fabsf(cl[2]) <= chs[2] + rb);
        if (hit && gtype == GBOX) {
          float gR[9], gl[3];
          if (gbody < 0) { for (int k = 0; k < 9; k++) gR[k] = gf[GF_R + k]; } else mulMM(gR, Rwr, gf + GF_R);
          mulMTV(gl, gR, dc);
          hit = fabsf(gl[0]) <= gf[GF_SIZE] + crb && fabsf(gl[1]) <= gf[GF_SIZE + 1] + crb && fabsf(gl[2]) <= gf[GF_SIZE + 2] + crb;
        }
        unsigned m16 = (unsigned)((__ballot(hit) >> (16 * r)) & 0xFFFFull);
        int pos = nh + __popc(m16 & ((1u << l) - 1u));
        if (hit && pos < MAXHIT) S.hits[pos] = (unsigned short)gid;
        nh += __popc(m16);
      }
      WSYNC();
      V5_TICK(13)  // (fine split of the broad phase: 13 = the cube against the lane's geoms, 14 = hand body pairs, 1 = geom level of the surviving pairs)
      if constexpr (SELF) {
#ifndef JH_V5_X_NOL1
      // hand self-collision, level 1: body pairs whose bounding spheres overlap (106 candidate pairs after MuJoCo's static filters, 16 per pass)
      int nbl = 0;
      {
        // (round 6: shaped for latency, the same tests in the same order.  Per pass of 16 pairs the loop was three dependent LDS round trips -- the pair, its two spheres, the
        // boxes' poses -- times seven passes.  Now: every pass's pair in one batch of loads, every pass's sphere test in a second, then the box test of the passes that
        // have a survivor.)
        constexpr int NPASS = MAXBP / G;
        int pab[NPASS]; unsigned sph = 0;
#pragma unroll
        for (int i = 0; i < NPASS; i++) { const int pi = i * G + l, pj = pi < nBP ? pi : 0; pab[i] = sBP[2 * pj] | sBP[2 * pj + 1] << 8; }
#pragma unroll
        for (int i = 0; i < NPASS; i++) {
          const float* sa = S.bs[pab[i] & 0xFF]; const float* sb = S.bs[pab[i] >> 8];
          const float d[3] = {sa[0] - sb[0], sa[1] - sb[1], sa[2] - sb[2]}, rs = sa[3] + sb[3];
          sph |= ((i * G + l < nBP) & (dot3(d, d) <= rs * rs)) ? 1u << i : 0u;
        }
#pragma unroll
        for (int i = 0; i < NPASS; i++) {
          if (i * G >= nBP) break;
          const int pi = i * G + l;
          bool hit = (sph >> i) & 1u;
          if (hit) {  // the two bodies' bounding boxes (static geometry: axis-aligned in the world)
            const int ba = pab[i] & 0xFF, bb = pab[i] >> 8;
            const float* sa = S.bs[ba]; const float* sb = S.bs[bb];
            const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            float Ra[9]; for (int k = 0; k < 9; k++) Ra[k] = static_code(ba) ? I9[k] : S.xR[ba][k];  // (side A is the static one of a pair, if any)
            hit = obb_face_overlap(sa, Ra, sBB + 8 * ba + 4, sb, S.xR[bb], sBB + 8 * bb + 4);
          }
          unsigned m16 = (unsigned)((__ballot(hit) >> (16 * r)) & 0xFFFFull);
          int pos = nbl + __popc(m16 & ((1u << l) - 1u));
          if (hit && pos < MAXBPL) S.bpl[pos] = (unsigned char)pi;
          nbl += __popc(m16);
        }
      }
      if (nbl > MAXBPL) { if (l == 0 && live && stats) atomicAdd(stats, nbl - MAXBPL); nbl = MAXBPL; }  // (counted with the dropped contacts)
#ifdef JH_V5_COUNT
      if (l == 0 && live) cnt_bp += nbl;
#endif
      const int nh_cube = nh;
      WSYNC();
      V5_TICK(14)
      // level 2, per surviving body pair: (a) every geom of either body against the OTHER body's bounding box (one pass: lanes 0..nA-1 take A's geoms,
      // the next nB lanes B's; nA + nB <= 16) -- usually nothing of one side comes near the other and the pair is done; (b) the near geoms of A against
      // the near geoms of B (bounding spheres, then the six face axes of their boxes)
      for (int i = 0; __any(i < nbl); i++) {
#ifdef JH_V5_COUNT
        if (lane == 0) cnt_l2++;
#endif
        int ba = 0, bb = 0, ga0 = 0, na = 0, gb0 = 0, nb = 0;
        if (i < nbl) { const int pi = S.bpl[i]; ba = sBP[2 * pi]; bb = sBP[2 * pi + 1]; ga0 = sBG[2 * ba]; na = sBG[2 * ba + 1]; gb0 = sBG[2 * bb]; nb = sBG[2 * bb + 1]; }
        bool near = false;
        if (l < na + nb) {
          const bool isA = l < na;
          const int g = isA ? ga0 + l : gb0 + (l - na), own = isA ? ba : bb, oth = isA ? bb : ba;
          const float* gf = sGeomF + g * GEOM_F;
          float cw[3];
          if (static_code(own)) { cw[0] = gf[GF_POS]; cw[1] = gf[GF_POS + 1]; cw[2] = gf[GF_POS + 2]; }
          else { mulMV(cw, S.xR[own], gf + GF_POS); cw[0] += S.pa[own][0]; cw[1] += S.pa[own][1]; cw[2] += S.pa[own][2]; }
          const float dw[3] = {cw[0] - S.bs[oth][0], cw[1] - S.bs[oth][1], cw[2] - S.bs[oth][2]};
          float dl[3];
          if (static_code(oth)) { dl[0] = dw[0]; dl[1] = dw[1]; dl[2] = dw[2]; } else mulMTV(dl, S.xR[oth], dw);
          const float* hb = sBB + 8 * oth + 4;
          const float ex = fmaxf(fabsf(dl[0]) - hb[0], 0.f), ey = fmaxf(fabsf(dl[1]) - hb[1], 0.f), ez = fmaxf(fabsf(dl[2]) - hb[2], 0.f);
          near = ex * ex + ey * ey + ez * ez <= gf[GF_RBOUND] * gf[GF_RBOUND];
        }
        const unsigned m16 = (unsigned)((__ballot(near) >> (16 * r)) & 0xFFFFull);
        const unsigned mB = (m16 >> na) & ((1u << nb) - 1u);
        unsigned rem = mB != 0 ? (m16 & ((1u << na) - 1u)) : 0u;
        while (__any(rem != 0)) {
          const int ia = rem != 0 ? __ffs(rem) - 1 : 0;
          bool hit = false;
          const int ga = ga0 + ia, gb = gb0 + l;
          if (rem != 0 && l < nb && ((mB >> l) & 1u)) {
            const float* fa = sGeomF + ga * GEOM_F; const float* fb = sGeomF + gb * GEOM_F;
            float ca[3], cb[3];
            if (static_code(ba)) { ca[0] = fa[GF_POS]; ca[1] = fa[GF_POS + 1]; ca[2] = fa[GF_POS + 2]; }
            else { mulMV(ca, S.xR[ba], fa + GF_POS); ca[0] += S.pa[ba][0]; ca[1] += S.pa[ba][1]; ca[2] += S.pa[ba][2]; }
            mulMV(cb, S.xR[bb], fb + GF_POS); cb[0] += S.pa[bb][0]; cb[1] += S.pa[bb][1]; cb[2] += S.pa[bb][2];
            const float d[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]}, rs = fa[GF_RBOUND] + fb[GF_RBOUND];
            hit = dot3(d, d) <= rs * rs;
            if (hit) {  // the geoms' own boxes (a sphere counts as the cube around it)
              float RA[9], RB[9];
              if (static_code(ba)) { for (int k = 0; k < 9; k++) RA[k] = fa[GF_R + k]; } else mulMM(RA, S.xR[ba], fa + GF_R);
              mulMM(RB, S.xR[bb], fb + GF_R);
              const bool sphA = sGeomI[ga * GEOM_I + 1] != GBOX, sphB = sGeomI[gb * GEOM_I + 1] != GBOX;
              const float hA[3] = {fa[GF_SIZE], sphA ? fa[GF_SIZE] : fa[GF_SIZE + 1], sphA ? fa[GF_SIZE] : fa[GF_SIZE + 2]};
              const float hB[3] = {fb[GF_SIZE], sphB ? fb[GF_SIZE] : fb[GF_SIZE + 1], sphB ? fb[GF_SIZE] : fb[GF_SIZE + 2]};
              hit = obb_face_overlap(ca, RA, hA, cb, RB, hB);
            }
          }
          const unsigned h16 = (unsigned)((__ballot(hit) >> (16 * r)) & 0xFFFFull);
          const int pos = nh + __popc(h16 & ((1u << l) - 1u));
          if (hit && pos < MAXHIT) S.hits[pos] = (unsigned short)(HITPAIR + (ga << 7 | gb));
          nh += __popc(h16);
          rem &= rem - 1u;
        }
      }
#ifdef JH_V5_COUNT
      if (l == 0 && live) cnt_hh += nh - nh_cube;
#endif
      hand_hits = nh > nh_cube;
#endif
      }
      if (nh > MAXHIT) { if (l == 0 && live && stats) atomicAdd(stats, nh - MAXHIT); nh = MAXHIT; }  // (candidate pairs lost: counted with the dropped contacts)
      WSYNC();
      V5_TICK(1)
      // narrow phase: survivor i goes to lane i; side A is the cube or the first geom of a hand pair, side B a hand geom
      // (copies of a rollout in latency mode write the same values to the same row; the padding rows of the last workgroup -- n >= N, they recompute rollout N - 1 out of
      // lock step with the live row -- get no overflow row: their contacts above the LDS pool are dropped, and only live rows count what they drop)
      PoolCtx pc{&S, live ? stats : nullptr, (NOVF > 0 && ovf_all && n < N) ? ovf_all + (size_t)nc * (NOVF * POOL_F) : nullptr};
      for (int base = 0; __any(base < nh); base += G) {
        int idx = base + l;
        if (idx < nh) {
          const int hid = S.hits[idx];
          int ga = -1, gb = hid;
          if (SELF && hid >= HITPAIR) { ga = (hid - HITPAIR) >> 7; gb = (hid - HITPAIR) & 0x7F; }
          // side B
          const float* fb = sGeomF + gb * GEOM_F; const int bodyb = sGeomI[gb * GEOM_I], tb = sGeomI[gb * GEOM_I + 1];
          float pB[3], RB[9];
          if (bodyb < 0) { for (int k = 0; k < 3; k++) pB[k] = fb[GF_POS + k]; for (int k = 0; k < 9; k++) RB[k] = fb[GF_R + k]; }
          else { mulMV(pB, S.xR[bodyb], fb + GF_POS); for (int k = 0; k < 3; k++) pB[k] += S.pa[bodyb][k]; mulMM(RB, S.xR[bodyb], fb + GF_R); }
          // side A
          float pA[3], RA[9], hA[3], mua, trana; int ta, codea;
          if (!SELF || ga < 0) { for (int k = 0; k < 3; k++) { pA[k] = qc[k]; hA[k] = chs[k]; } for (int k = 0; k < 9; k++) RA[k] = Rc[k]; mua = 0.f; trana = ctran; ta = GBOX; codea = CUBE; }
          else {
            const float* fa = sGeomF + ga * GEOM_F; const int bodya = sGeomI[ga * GEOM_I];
            if (bodya < 0) { for (int k = 0; k < 3; k++) pA[k] = fa[GF_POS + k]; for (int k = 0; k < 9; k++) RA[k] = fa[GF_R + k]; }
            else { mulMV(pA, S.xR[bodya], fa + GF_POS); for (int k = 0; k < 3; k++) pA[k] += S.pa[bodya][k]; mulMM(RA, S.xR[bodya], fa + GF_R); }
            for (int k = 0; k < 3; k++) hA[k] = fa[GF_SIZE + k];
            mua = fa[GF_MUOWN]; trana = fa[GF_TRAN]; ta = sGeomI[ga * GEOM_I + 1]; codea = bodya < 0 ? 0 : bodya;
          }
          // contact parameters: friction = the larger of the two geoms' (the cube's is folded into GF_MU), R from the two bodies' inverse weights
          const float mu = (!SELF || ga < 0) ? fb[GF_MU] : fmaxf(mua, fb[GF_MUOWN]);
          LeapSink sk{pc, codea | ((bodyb < 0 ? 0 : bodyb) << 8), mu, trana + fb[GF_TRAN], false};
          if ((!SELF || ta == GBOX) && tb == GBOX) collide_box_box(sk, pA, RA, hA, pB, RB, fb + GF_SIZE);
          else if (!SELF || ta == GBOX) collide_box_sphere(sk, pA, RA, hA, pB, fb[GF_SIZE]);
          else if (tb == GBOX) { sk.flip = true; collide_box_sphere(sk, pB, RB, fb + GF_SIZE, pA, hA[0]); }
          else {  // two spheres (fingertips)
            const float d[3] = {pB[0] - pA[0], pB[1] - pA[1], pB[2] - pA[2]}; const float dn = sqrtf(dot3(d, d)), dist = dn - hA[0] - fb[GF_SIZE];
            if (dist < 0.f && dn > 1e-9f) {
              const float n3[3] = {d[0] / dn, d[1] / dn, d[2] / dn}, m = hA[0] + 0.5f * dist;
              const float pos3[3] = {pA[0] + m * n3[0], pA[1] + m * n3[1], pA[2] + m * n3[2]};
              sk.push(pos3, n3, dist);
            }
          }
        }
      }
    }
    WSYNC();
    V5_TICK(2)
    // ================================================================ constraint rows: <= 2 contacts per lane + the own dof's friction-loss / limit rows
    // The constraint rows and the Newton solver exist once per slot count: a wave in which some rollout has more than 16 * NSLOT contacts this step (jammed cube:
    // several 4-point box-box manifolds at once) runs the copy with NSBIG slots per lane, all others the copy with NSLOT -- the third slot's 27 registers would
    // otherwise be spilled and reloaded inside every iteration of every rollout (measured: +17 % on the headline workload for 6e-4 of its rollout-steps).
    float a_own, ac_own; int iters_this = 0;
#ifdef JH_V5_CENSUS  // diagnostic builds (tools/diag/census_v5.py): histograms in stats[64..]: contacts per rollout-step / the maximum over the wave's rollouts, Newton iterations per
                     // rollout-step / per wave-step, line-search evaluations per Newton iteration of a rollout / of the wave, rollouts still active per Newton iteration of the wave
    const int wave_it0 = n_wave_iters;
    if (stats) {
      int mx = S.ncon; mx = max(mx, __shfl_xor(mx, 16)); mx = max(mx, __shfl_xor(mx, 32));
      if (l == 0 && live) atomicAdd(stats + 64 + min(S.ncon, 63), 1);
      if (lane == 0) atomicAdd(stats + 128 + min(mx, 63), 1);
      // (round 6) what a quad-per-contact assembly would need: cube contacts by the chain of their finger link (static geometry: any quad); passes = the longest quad's list
      int cn[5] = {0, 0, 0, 0, 0};
      const int ncc = S.ncon < NCP ? S.ncon : NCP;
      for (int i = 0; i < ncc; i++) { const int sd = __float_as_int(S.pool[i][8]), lb_ = sd >> 8; cn[(lb_ > 0 && lb_ < NMB) ? (lb_ - 1) >> 2 : 4]++; }
      const int cmx = max(max(cn[0], cn[1]), max(cn[2], cn[3])), tot_ = cn[0] + cn[1] + cn[2] + cn[3] + cn[4];
      int P = max(cmx, (tot_ + 3) >> 2), Pw = P; Pw = max(Pw, __shfl_xor(Pw, 16)); Pw = max(Pw, __shfl_xor(Pw, 32));
      int cw = cmx; cw = max(cw, __shfl_xor(cw, 16)); cw = max(cw, __shfl_xor(cw, 32));
      if (l == 0 && live) { atomicAdd(stats + 328 + min(P, 15), 1); atomicAdd(stats + 360 + min(cmx, 11), 1); }
      if (lane == 0) { atomicAdd(stats + 344 + min(Pw, 15), 1); atomicAdd(stats + 372 + min(cw, 11), 1); }
    }
#endif
    auto solve_step = [&](auto NS_, auto HC_) __attribute__((always_inline)) -> bool {
    constexpr bool HC = decltype(HC_)::value;  // hand contacts (a side that is not the cube) possible in this wave-step: false = the copy without their code (see the dispatch below)
    constexpr int NS = decltype(NS_)::value;
    const int ncap = (NOVF > 0 && (ovf_all == nullptr || n >= N) && 16 * NS > NCP) ? NCP : 16 * NS;  // (no overflow row: what the LDS pool holds)
    const int ncon = S.ncon < ncap ? S.ncon : ncap;
    Slot sl[NS];
    {
      float wv3[3]; mulMV(wv3, S.xR[0], vc + 3);  // world angular velocity of the cube
#pragma unroll
      for (int k = 0; k < NS; k++) {
        int idx = l + 16 * k;
        sl[k].la = -1; sl[k].lb = 0;
        for (int i = 0; i < 9; i++) sl[k].fr[i] = 0.f;
        sl[k].rc[0] = sl[k].rc[1] = sl[k].rc[2] = 0.f;
        sl[k].D0 = sl[k].D1 = sl[k].Dm = sl[k].mu = sl[k].fri = 0.f;  // (an empty slot is all zeros: the line search evaluates it like any other and gets zero, lane_rows_dir)
        for (int i = 0; i < 3; i++) sl[k].aref[i] = sl[k].jar[i] = sl[k].jp[i] = 0.f;
        if (idx < ncon) {
          const float* e = (NOVF == 0 || idx < NCP) ? S.pool[idx] : ovf_all + (size_t)nc * (NOVF * POOL_F) + (idx - NCP) * POOL_F;
          sl[k].rc[0] = e[0] - qc[0]; sl[k].rc[1] = e[1] - qc[1]; sl[k].rc[2] = e[2] - qc[2];
          sl[k].fr[0] = e[3]; sl[k].fr[1] = e[4]; sl[k].fr[2] = e[5];
          make_frame(sl[k].fr);
          float dist = e[6], mu = e[7], tran = e[9]; const int sides = __float_as_int(e[8]);
          sl[k].la = HC ? (sides & 0xFF) : CUBE; sl[k].lb = sides >> 8;
          float imp = impedance(csi, dist);
          float R0 = fmaxf(1e-15f, (1.f - imp) / imp * tran), R1 = R0 / fmaxf(1e-15f, impratio);
          sl[k].D0 = 1.f / R0; sl[k].D1 = 1.f / R1;
          sl[k].fri = mu; sl[k].mu = mu * sqrtf(R1 / R0);
          { float m2 = sl[k].mu * sl[k].mu; sl[k].Dm = sl[k].D0 / (m2 * (1.f + m2)); }
          float vel[3]; slot_Jx<HC>(sl[k], S, qc, vc, wv3, S.qv, vel);
          sl[k].aref[0] = -cB * vel[0] - cK * imp * dist; sl[k].aref[1] = -cB * vel[1]; sl[k].aref[2] = -cB * vel[2];
        }
      }
    }
#if JH_V5_C3CACHE
    // (every lane has loaded its slots: the pool's storage is free from here on -- a wave's LDS instructions execute in order)
    if (sl[0].la >= 0 && sl[0].lb > 0) {
      const float pos0[3] = {sl[0].rc[0] + qc[0], sl[0].rc[1] + qc[1], sl[0].rc[2] + qc[2]};
      float c3[NLK][3]; link_c3(S, (sl[0].lb - 1) >> 2, pos0, c3);
      float4* o = reinterpret_cast<float4*>(S.c3s[l]);
      o[0] = make_float4(c3[0][0], c3[0][1], c3[0][2], c3[1][0]); o[1] = make_float4(c3[1][1], c3[1][2], c3[2][0], c3[2][1]); o[2] = make_float4(c3[2][2], c3[3][0], c3[3][1], c3[3][2]);
    }
#define V5_C3C(k) ((k) == 0 ? (const float*)S.c3s[l] : (const float*)nullptr)
#else
#define V5_C3C(k) ((const float*)nullptr)
#endif
    DofRows dr;
    dr.fl = lc[LC_FL]; dr.fD = lc[LC_FD]; dr.fR = dr.fD > 0.f ? 1.f / dr.fD : 0.f; dr.faref = -lc[LC_FB] * qd; dr.lims = 0.f; dr.laref = 0.f; dr.lD = 0.f; dr.jf = dr.jl = dr.pf = dr.pl = 0.f;
    if (lc[LC_LIMITED] != 0.f) {
      float dlo = q - lc[LC_LO], dhi = lc[LC_HI] - q, dist = fminf(dlo, dhi);
      if (dist < 0.f) {
        float sg = dlo < dhi ? 1.f : -1.f, imp = impedance(lc + LC_SI, dist), R = fmaxf(1e-15f, (1.f - imp) / imp * lc[LC_INVW]);
        dr.lims = sg; dr.lD = 1.f / R; dr.laref = -lc[LC_LB] * (sg * qd) - lc[LC_LK] * imp * dist;
      }
    }
    // Which finger chains are coupled by contacts between their links (hand self-collision): a 4 x 4 bit matrix per rollout, bit 4a + b for chains a < b.
    // Without a cycle in that graph (every coupled step of the headline workload: one pair, mostly middle-ring, or a chain touching both neighbours) the
    // arrow elimination runs in stages, leaf chains first: a chain is eliminated when at most one coupled neighbour is left (its parent `par`, whose
    // blocks take the Schur update and which factorises at a later stage `lvl`); then the cube.  A graph with a cycle takes the dense direction.
    bool anyslot = false; int cmask = 0; unsigned used = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      anyslot |= sl[k].la >= 0;
      used |= __any(sl[k].la >= 0) ? 1u << k : 0u;
      if (HC && sl[k].la > 0 && sl[k].la != CUBE && sl[k].lb > 0 && ((sl[k].la - 1) >> 2) != ((sl[k].lb - 1) >> 2)) cmask |= 1 << (4 * ((sl[k].la - 1) >> 2) + ((sl[k].lb - 1) >> 2));
    }
    int lvl = 0, par = -1, nlev = 0; bool dense_row = false;
    if constexpr (HC) {
      cmask = gor(cmask);
      if (__any(cmask != 0)) dense_row = !chain_elim_order(cmask, c, lvl, par, nlev);
#ifdef JH_V5_COUNT
      if (l == 0 && live) {
        int dmax = 0;
        for (int a = 0; a < NCH; a++) { int d = 0; for (int b = 0; b < NCH; b++) if (b != a) d += (cmask >> (a < b ? 4 * a + b : 4 * b + a)) & 1; dmax = d > dmax ? d : dmax; }
        cnt_cls[cmask == 0 ? 0 : (dense_row ? 3 : (dmax <= 1 ? 1 : 2))]++;
      }
#endif
    }
    if constexpr (HC && NS < NSLOT) { if (__any(dense_row)) return false; }  // (the one-slot copy has no dense direction: the wave takes the NSLOT copy; nothing was written yet)
#if JH_V5_PARK
    S.pk_q[l] = q; S.pk_fs[l] = fs_own;
    if (l == 0) { S.pk_cq[0] = qc[3]; S.pk_cq[1] = qc[4]; S.pk_cq[2] = qc[5]; S.pk_cq[3] = qc[6]; S.pk_acc = acc; }
#endif
    // ================================================================ Newton solver (rows distributed over the 16 lanes)
    const float Mdiag_own = sel4(Mrow, s), iMd = 1.f / Mdiag_own;
    const float fsc_own = mck * a0c_own;
    const float snorm = gsum(fs_own * fs_own * iMd + fsc_own * fsc_own * lc[LC_IMCK]);
    const bool has_rows = gor((int)(anyslot || dr.fl > 0.f || dr.lims != 0.f)) != 0;
    iters_this = 0;
    if (!has_rows) { a_own = a0_own; ac_own = a0c_own; }
    else {
      // ---- warm start: the better of last step's acceleration (S.ws) and the unconstrained one
      {
        const float qws = S.ws[6 + l];
        float xl[3] = {S.ws[0], S.ws[1], S.ws[2]}, xr[3] = {S.ws[3], S.ws[4], S.ws[5]}, wa[3]; mulMV(wa, S.xR[0], xr);  // (S.ws keeps the body-frame acceleration of the last step)
#if JH_V5_WORLDROT
        const float wsc_own = l < 3 ? S.ws[l] : (l == 3 ? wa[0] : (l == 4 ? wa[1] : (l == 5 ? wa[2] : 0.f)));
#else
        const float wsc_own = l < 6 ? S.ws[l] : 0.f;
#endif
        float cs = 0.f, jx[3], jar_ws[NS][3];
#pragma unroll
        for (int k = 0; k < NS; k++) if (sl[k].la >= 0) {
          slot_Jx<HC>(sl[k], S, qc, xl, wa, S.ws, jx, V5_C3C(k));
          for (int rw = 0; rw < 3; rw++) { sl[k].jar[rw] = jx[rw] - sl[k].aref[rw]; jar_ws[k][rw] = sl[k].jar[rw]; }
          cs += cone_cost(sl[k]);
        }
        const float jf_ws = qws - dr.faref, jl_ws = dr.lims * qws - dr.laref;
        dr.jf = jf_ws; dr.jl = jl_ws;
        cs += dof_rows_cost(dr);
        float dws = qws - a0_own, md = 0.f;
#pragma unroll
        for (int j = 0; j < NLK; j++) md += Mrow[j] * (quad_get(qws, j) - quad_get(a0_own, j));
        cs += 0.5f * dws * md;
        { float dcw = wsc_own - a0c_own; cs += 0.5f * dcw * dcw * mck; }
        const float cost_ws = gsum(cs);
        S.p[6 + l] = a0_own; if (l < 6) S.p[l] = a0c_own;
        WSYNC();
        float xl0[3] = {S.p[0], S.p[1], S.p[2]}, xr0[3] = {S.p[3], S.p[4], S.p[5]};
#if JH_V5_WORLDROT
        wa[0] = xr0[0]; wa[1] = xr0[1]; wa[2] = xr0[2];
#else
        mulMV(wa, S.xR[0], xr0);
#endif
        cs = 0.f;
#pragma unroll
        for (int k = 0; k < NS; k++) if (sl[k].la >= 0) {
          slot_Jx<HC>(sl[k], S, qc, xl0, wa, S.p, jx, V5_C3C(k));
          for (int rw = 0; rw < 3; rw++) sl[k].jar[rw] = jx[rw] - sl[k].aref[rw];
          cs += cone_cost(sl[k]);
        }
        dr.jf = a0_own - dr.faref; dr.jl = dr.lims * a0_own - dr.laref;
        cs += dof_rows_cost(dr);
        const float cost_0 = gsum(cs);
        if (cost_ws < cost_0) {
          a_own = qws; ac_own = wsc_own;
#pragma unroll
          for (int k = 0; k < NS; k++) if (sl[k].la >= 0) for (int rw = 0; rw < 3; rw++) sl[k].jar[rw] = jar_ws[k][rw];
          dr.jf = jf_ws; dr.jl = jl_ws;
        } else { a_own = a0_own; ac_own = a0c_own; }
        WSYNC();
      }
      bool act = true;
      V5_TICK(3)
      // The Newton loop exists twice: waves in which some rollout needs the dense direction this step run the copy that contains it, all others a copy
      // without that code (the register needs of the rare path would otherwise make the allocator spill inside every iteration of every rollout)
      constexpr int OPQ = JH_V5_OPAQUE >= 0 ? JH_V5_OPAQUE : 2;
      auto forget_slots = [&]() __attribute__((always_inline)) {
        if constexpr (OPQ > 0) {
#pragma unroll
          for (int k = 0; k < NS; k++) {
            OPAQUE(sl[k].la); OPAQUE(sl[k].lb);
            if constexpr (OPQ > 1) { OPAQUE(sl[k].rc[0]); OPAQUE(sl[k].rc[1]); OPAQUE(sl[k].rc[2]); }
            if constexpr (OPQ > 2) { for (int q9 = 0; q9 < 9; q9++) OPAQUE(sl[k].fr[q9]); }
          }
        }
      };
      auto newton_loop = [&](auto dense_tag) __attribute__((always_inline)) {
      constexpr bool DENSE = decltype(dense_tag)::value;
      for (int it = 0; it < cap && __any(act); it++) {
        // ---- (1) gradient.  Owner lanes: M (a - a0) rows + dof-row forces; contacts: -J'f as LDS float atomics (finger and cube parts)
        forget_slots();
        const float da_own = a_own - a0_own, dcl = ac_own - a0c_own;
        float g_own = 0.f, hd = 0.f;
#pragma unroll
        for (int j = 0; j < NLK; j++) g_own += Mrow[j] * quad_get(da_own, j);
        {  // the own dof's friction-loss and limit rows (selects: see lane_rows_dir)
          const float D = dr.fD, x = dr.jf, fl = dr.fl, lim = dr.fR * fl;
          const bool has = fl > 0.f, lo_ = x <= -lim, hi_ = x >= lim, mid = has & !lo_ & !hi_;
          float ga = g_own - fl, gb = g_own + fl, gm = fmaf(D, x, g_own); asm volatile("" : "+v"(ga), "+v"(gb), "+v"(gm));  // (all three computed: selects, not branches)
          g_own = has ? (lo_ ? ga : (hi_ ? gb : gm)) : g_own;
          hd = mid ? hd + D : hd;
          const bool lon = (dr.lims != 0.f) & (dr.jl < 0.f);
          g_own = lon ? fmaf(dr.lims * dr.lD, dr.jl, g_own) : g_own;
          hd = lon ? hd + dr.lD : hd;
        }
        S.g[6 + l] = g_own;  // (this and the block initialisation below are stored whatever the rollout's state -- nothing of a converged rollout is read again this step, a dense
                             // row's matrix is zeroed after them -- so that the top of an iteration has no exec-masked region in front of the contact pass)
        // One pass over the contacts builds the gradient AND the Hessian of this iterate: the joint columns axis x (pos - anchor), the world force and the cone weights are
        // computed once instead of once per pass.  The pass that finds a rollout converged has then assembled a Hessian nobody reads (one iteration in ten).
        const bool aact0 = act && !(DENSE && dense_row);
        {
          float* hb = &S.Hbb[c][tri(s, 0)];  // row s of the chain's block: entries j < s, then the diagonal -- written once per column index j, the columns beyond the diagonal land on it again
#pragma unroll
          for (int j = 0; j < NLK; j++) hb[j < s ? j : s] = j < s ? Mrow[j] : Mdiag_own + hd;
          for (int k = 0; k < 6; k++) S.Hcb[c][s * 6 + k] = 0.f;
          if (HC) for (int k = 0; k < 6; k++) S.Hx[0][l * 6 + k] = 0.f;
        }
        float hcp[21];  // the cube block J'WJ of this lane's contacts: every cube contact of the rollout lands on the same 21 entries -> row sums instead of 21 conflicting atomics per contact
#pragma unroll
        for (int e = 0; e < 21; e++) hcp[e] = 0.f;
        bool hcany = false;
        WSYNC();
        float gcp[6] = {0, 0, 0, 0, 0, 0};  // cube part of -J'f: every contact of the rollout lands on the same six entries -> row sums, not atomics
        {
#pragma unroll
          for (int k = 0; k < NS; k++) {
            if (k > 0 && !((used >> k) & 1u)) continue;  // (wave-uniform)
            const Slot& t = sl[k];
            float f[3], Wk[6];
            const float D[3] = {t.D0, t.D1, t.D1};
            // separated contact (the cone's top zone: no force, no weights): skipped on the zone test itself, before any of the other zones' arithmetic.  (The other two zones have
            // a positive W[0]: `weights all zero` and `top zone` are the same set of contacts.)  Outside the dense-capable copy `aact0` is `act`, true in here: `on` is a
            // compile-time constant there
            // One exec-masked region per slot: an empty slot is all zeros and reads `top`, a converged rollout is masked by the same test.
            ConeZ cz;
            if (cone_top(t.jar, t.mu, t.fri, cz) | !act) continue;
            cone_below(t.jar, D, t.Dm, t.mu, t.fri, cz, f, Wk);
            bool on = true;
            if constexpr (DENSE) on = aact0;
            const float Fw[3] = {t.fr[0] * f[0] + t.fr[3] * f[1] + t.fr[6] * f[2], t.fr[1] * f[0] + t.fr[4] * f[1] + t.fr[7] * f[2], t.fr[2] * f[0] + t.fr[5] * f[1] + t.fr[8] * f[2]};
            const float pos[3] = {t.rc[0] + qc[0], t.rc[1] + qc[1], t.rc[2] + qc[2]};
            const bool cube = !HC || t.la == CUBE;
            // J'WJ in the world frame: every dof column of the contact is a world 3-vector col_x (J[w][x] = fr_w . col_x), so the entry (x, y) is col_x' A col_y with
            // A = Fr' W Fr, a symmetric 3 x 3 formed once per contact.  The cube's translation columns are -e_q: their block is A itself and their coupling to a column c is
            // -(A c)_q, both free; a third fewer multiply-adds than frame-space columns times W (the fr3 kernel's formulation).
            float A[6];
            if (on) {
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
#if !JH_V5_WORLDROT
            float cq[3][3];  // the cube's rotation columns (body axes x arm), up to the sign
#endif
            if (cube) {
#if JH_V5_WORLDROT
              float tb[3]; cross3(tb, t.rc, Fw);  // torque about the cube's origin, world
#else
              float tq[3], tb[3]; cross3(tq, t.rc, Fw); mulMTV(tb, S.xR[0], tq);
#endif
              gcp[0] += Fw[0]; gcp[1] += Fw[1]; gcp[2] += Fw[2]; gcp[3] += tb[0]; gcp[4] += tb[1]; gcp[5] += tb[2];
              if (on) {
#pragma unroll
                for (int e = 0; e < 6; e++) hcp[e] += A[e];
#if JH_V5_WORLDROT
                // rotation columns e_q x r = (0, -rz, ry), (rz, 0, -rx), (-ry, rx, 0): z_q = A (e_q x r) is a combination of two columns of A, and a dot product with
                // e_p x r has two terms
                const float rx = t.rc[0], ry = t.rc[1], rz = t.rc[2];
                const float Ac[3][3] = {{A[0], A[1], A[3]}, {A[1], A[2], A[4]}, {A[3], A[4], A[5]}};  // column k of A (symmetric)
                float zq[3][3];
#pragma unroll
                for (int i = 0; i < 3; i++) { zq[0][i] = ry * Ac[2][i] - rz * Ac[1][i]; zq[1][i] = rz * Ac[0][i] - rx * Ac[2][i]; zq[2][i] = rx * Ac[1][i] - ry * Ac[0][i]; }
#pragma unroll
                for (int q = 0; q < 3; q++) {
#pragma unroll
                  for (int r2 = 0; r2 < 3; r2++) hcp[tri(3 + q, r2)] += zq[q][r2];
                }
                hcp[tri(3, 3)] += ry * zq[0][2] - rz * zq[0][1];
                hcp[tri(4, 3)] += ry * zq[1][2] - rz * zq[1][1]; hcp[tri(4, 4)] += rz * zq[1][0] - rx * zq[1][2];
                hcp[tri(5, 3)] += ry * zq[2][2] - rz * zq[2][1]; hcp[tri(5, 4)] += rz * zq[2][0] - rx * zq[2][2]; hcp[tri(5, 5)] += rx * zq[2][1] - ry * zq[2][0];
#else
#pragma unroll
                for (int q = 0; q < 3; q++) { float ea[3]; col3(ea, S.xR[0], q); cross3(cq[q], ea, t.rc); }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                  float z[3]; Amul(cq[q], z);
#pragma unroll
                  for (int r2 = 0; r2 < 3; r2++) hcp[tri(3 + q, r2)] += z[r2];
#pragma unroll
                  for (int r2 = 0; r2 <= q; r2++) hcp[tri(3 + q, 3 + r2)] += dot3(cq[r2], z);
                }
#endif
                hcany = true;
              }
            }
            if (t.lb > 0) {
              const int ch = (t.lb - 1) >> 2, dep = (t.lb - 1) & 3;
              const bool linkA = HC && !cube && t.la > 0;
              const int cha = linkA ? (t.la - 1) >> 2 : 0, depa = linkA ? (t.la - 1) & 3 : -1;
              const bool same = linkA && cha == ch;
              float cb[NLK][3]; if (V5_C3C(k)) load_c3(V5_C3C(k), cb); else link_c3(S, ch, pos, cb);
              if constexpr (!HC) {
                // cube contact on a finger link (the copy without the hand's own contacts): joint u4 of the chain carries the contact iff u4 <= dep -- its gradient entry, its row
                // of the chain block and its coupling to the cube under ONE test per joint (the general form below: one for the force, one for the rows, and the columns
                // beyond the link's depth zeroed first -- here they are never read)
#pragma unroll
                for (int u4 = 0; u4 < NLK; u4++) if (u4 <= dep) {
                  atomicAdd(&S.g[6 + 4 * ch + u4], -dot3(cb[u4], Fw));
                  float y[3]; Amul(cb[u4], y);
#pragma unroll
                  for (int v4 = 0; v4 <= u4; v4++) atomicAdd(&S.Hbb[ch][tri(u4, v4)], dot3(cb[v4], y));
#pragma unroll
                  for (int q = 0; q < 3; q++) atomicAdd(&S.Hcb[ch][u4 * 6 + q], -y[q]);
#if JH_V5_WORLDROT
                  atomicAdd(&S.Hcb[ch][u4 * 6 + 3], t.rc[2] * y[1] - t.rc[1] * y[2]);  // -(e_q x r) . y
                  atomicAdd(&S.Hcb[ch][u4 * 6 + 4], t.rc[0] * y[2] - t.rc[2] * y[0]);
                  atomicAdd(&S.Hcb[ch][u4 * 6 + 5], t.rc[1] * y[0] - t.rc[0] * y[1]);
#else
#pragma unroll
                  for (int q = 0; q < 3; q++) atomicAdd(&S.Hcb[ch][u4 * 6 + 3 + q], -dot3(cq[q], y));
#endif
                }
              } else {
#if JH_V5_HCMERGE
              // (round 6, the hand-capable copy as the other one: a joint's gradient entry, its row of the chain block and its coupling to the cube under ONE test per joint; the
              // coupling of a contact that is not the cube's is added as zeros instead of sitting in a nested exec-masked region per joint; the opposite force of a contact with both
              // sides in one chain -- rare -- behind one wave-uniform test.  The same values reach the same addresses in the same order.)
              float fjs[NLK];
#pragma unroll
              for (int u4 = 0; u4 < NLK; u4++) {
                fjs[u4] = dot3(cb[u4], Fw);
                const float sg = (u4 <= dep ? 1.f : 0.f) - ((same && u4 <= depa) ? 1.f : 0.f);
                cb[u4][0] *= sg; cb[u4][1] *= sg; cb[u4][2] *= sg;
                if (u4 <= dep) {
                  atomicAdd(&S.g[6 + 4 * ch + u4], -fjs[u4]);  // side B: -J'f
                  if (on) {
                    float y[3]; Amul(cb[u4], y);
#pragma unroll
                    for (int v4 = 0; v4 <= u4; v4++) atomicAdd(&S.Hbb[ch][tri(u4, v4)], dot3(cb[v4], y));
#if JH_V5_WORLDROT
                    const float hv[6] = {-y[0], -y[1], -y[2], t.rc[2] * y[1] - t.rc[1] * y[2], t.rc[0] * y[2] - t.rc[2] * y[0], t.rc[1] * y[0] - t.rc[0] * y[1]};  // -(e_q x r) . y
#else
                    const float hv[6] = {-y[0], -y[1], -y[2], -dot3(cq[0], y), -dot3(cq[1], y), -dot3(cq[2], y)};
#endif
#if JH_V5_HCMERGE > 1
#pragma unroll
                    for (int q = 0; q < 6; q++) atomicAdd(&S.Hcb[ch][u4 * 6 + q], cube ? hv[q] : 0.f);
#else
                    if (cube) {
#pragma unroll
                      for (int q = 0; q < 6; q++) atomicAdd(&S.Hcb[ch][u4 * 6 + q], hv[q]);
                    }
#endif
                  }
                }
              }
              if (__any(same)) {
#pragma unroll
                for (int j = 0; j < NLK; j++) if (same && j <= depa) atomicAdd(&S.g[6 + 4 * ch + j], fjs[j]);  // side A of the same chain: the opposite force
              }
#else
#pragma unroll
              for (int j = 0; j < NLK; j++) {
                const float fj = dot3(cb[j], Fw);
                if (j <= dep) atomicAdd(&S.g[6 + 4 * ch + j], -fj);       // side B: -J'f
                if (same && j <= depa) atomicAdd(&S.g[6 + 4 * ch + j], fj);  // side A of the same chain: the opposite force
                const float sg = (j <= dep ? 1.f : 0.f) - ((same && j <= depa) ? 1.f : 0.f);
                cb[j][0] *= sg; cb[j][1] *= sg; cb[j][2] *= sg;
              }
              if (on) {
#pragma unroll
                for (int u4 = 0; u4 < NLK; u4++) if (u4 <= dep) {
                  float y[3]; Amul(cb[u4], y);
#pragma unroll
                  for (int v4 = 0; v4 <= u4; v4++) atomicAdd(&S.Hbb[ch][tri(u4, v4)], dot3(cb[v4], y));
                  if (cube) {
#pragma unroll
                    for (int q = 0; q < 3; q++) atomicAdd(&S.Hcb[ch][u4 * 6 + q], -y[q]);
#if JH_V5_WORLDROT
                    atomicAdd(&S.Hcb[ch][u4 * 6 + 3], t.rc[2] * y[1] - t.rc[1] * y[2]);  // -(e_q x r) . y
                    atomicAdd(&S.Hcb[ch][u4 * 6 + 4], t.rc[0] * y[2] - t.rc[2] * y[0]);
                    atomicAdd(&S.Hcb[ch][u4 * 6 + 5], t.rc[1] * y[0] - t.rc[0] * y[1]);
#else
#pragma unroll
                    for (int q = 0; q < 3; q++) atomicAdd(&S.Hcb[ch][u4 * 6 + 3 + q], -dot3(cq[q], y));
#endif
                  }
                }
              }
#endif
              if (linkA && !same) {  // side A sits in another chain: its own block, and the pair's coupling block -Jb'W Ja in Hx (B's chain is always the higher one)
                float ca[NLK][3]; link_c3(S, cha, pos, ca);
#pragma unroll
                for (int j = 0; j < NLK; j++) {
                  if (j <= depa) atomicAdd(&S.g[6 + 4 * cha + j], dot3(ca[j], Fw));
                  const float sg = j <= depa ? 1.f : 0.f;
                  ca[j][0] *= sg; ca[j][1] *= sg; ca[j][2] *= sg;
                }
                if (on) {
#pragma unroll
                  for (int u4 = 0; u4 < NLK; u4++) if (u4 <= depa) {
                    float y[3]; Amul(ca[u4], y);
#pragma unroll
                    for (int v4 = 0; v4 <= u4; v4++) atomicAdd(&S.Hbb[cha][tri(u4, v4)], dot3(ca[v4], y));
#pragma unroll
                    for (int v4 = 0; v4 < NLK; v4++) if (v4 <= dep) atomicAdd(&S.Hx[pidx(cha, ch)][v4 * 4 + u4], -dot3(cb[v4], y));
                  }
                }
              }
              }
            }
          }
        }
        float gcl = 0.f;
        float h0 = 0.f, h1;
        {  // two reduce-scatters: entries 16..20 of the cube block (to lanes 0..4) with the six gradient sums (to lanes 8..13, moved to 0..5 by a rotation), and entries 0..15
          const float v2[16] = {hcp[16], hcp[17], hcp[18], hcp[19], hcp[20], 0.f, 0.f, 0.f, gcp[0], gcp[1], gcp[2], gcp[3], gcp[4], gcp[5], 0.f, 0.f};
          h1 = row_scatter16(v2, l);
          const float gq = dppf<0x128>(h1);
          gcl = l < 6 ? gq : 0.f;
          if (__any(hcany)) h0 = row_scatter16(hcp, l);
        }
        gcl = fmaf(mck, dcl, gcl);
        {
          // (stored whatever the rollout's state: nothing of a converged rollout is read again this step, and a dense row's matrix is zeroed after this.  The cube's own
          // mass diagonal is added where the block is read back: no LDS read -- a round trip behind the atomics above -- between the sums and the convergence test)
          // (addressed from the lane's gradient entry, the address the next statement reads: the hand-capable copy had `&S.Hcc[l]` spilled to scratch memory -- a reload
          // and its wait between the sums and the convergence test)
          constexpr int HCC_G = ((int)offsetof(RS, Hcc) - (int)offsetof(RS, g)) / 4 - 6;
          float* gl = &S.g[6 + l];
          gl[HCC_G] = h0;
          if (l < 5) gl[HCC_G + 16] = h1;
        }
        WSYNC();
        V5_TICK(4)
        // ---- (2) convergence on the scaled gradient; the wave leaves the loop before any Hessian work once all its rollouts are done
        g_own = S.g[6 + l];
        const float gn = gsum(g_own * g_own * iMd + gcl * gcl * lc[LC_IMCK]);
        act = act & !(gn <= tol * tol * fmaxf(snorm, 1e-12f));
        if (!__any(act)) break;
        iters_this += act ? 1 : 0;
        n_wave_iters++;
        // ---- (3) Hessian: M + dof rows on the chain diagonals, cube inertia on Hcc, J'WJ of the contacts as atomics into the arrow blocks.  A rollout with a
        // contact between two finger chains has no arrow structure: its Hessian is assembled densely further down (aact = false here)
        if constexpr (OPQ > 3) forget_slots();
        const bool aact = act && !(DENSE && dense_row);
        V5_TICK(5)
        // ---- (4) arrow factorisation: chain blocks first (each chain's 4 lanes redundantly); the coupling columns Y_q = L^-1 Hcb[:,q] are shared
        // by the chain's lanes (lane s: columns s and s+4); 6x6 Schur complement on the cube, solved by every lane
        float L[10], Linv[4], Ya[NLK], Yb[NLK], zb[NLK], xc6[6], pc4[NLK];
        const bool hasb = s < 2;  // lanes 0,1 of a chain carry a second column (q = 4, 5)
        auto factor_block = [&]() __attribute__((always_inline)) {
          for (int k = 0; k < 10; k++) L[k] = S.Hbb[c][k];
          chol4(L, Linv);
          for (int j = 0; j < NLK; j++) { Ya[j] = S.Hcb[c][j * 6 + s]; const float yb = (&S.Hcb[0][0])[c * 24 + j * 6 + 4 + (s & 1)]; Yb[j] = hasb ? yb : 0.f; }  // (unconditional load, then select)
          fwd4(L, Linv, Ya); fwd4(L, Linv, Yb);
          asm volatile("" : "+v"(Yb[0]), "+v"(Yb[1]), "+v"(Yb[2]), "+v"(Yb[3]));  // (the compiler forgets that they are zero where `hasb` is false: it split everything downstream into a copy per case, and a quad runs both)
          for (int j = 0; j < NLK; j++) zb[j] = -S.g[6 + 4 * c + j];
          fwd4(L, Linv, zb);
        };
        factor_block();
        if (l < 6) S.rhs6[l] = -gcl;
        float Xs[NLK] = {0.f, 0.f, 0.f, 0.f};  // column s of X = L^-1 H(a,P) of a chain a with a parent P
        if constexpr (HC) {
#pragma unroll 1
          for (int st = 1; st < NCH; st++) {
            if (!__any(aact && nlev >= st)) break;
            // chains of stage st - 1 with a parent: their elimination updates the parent's blocks -- Hbb(P) -= X'X, Hcb(P) -= X'Y, g(P) += X'zb (so that the
            // parent's zb becomes L_P^-1 (-g_P - X'zb_a)) -- before the parent factorises at its own stage (atomics: two leaves may share a parent)
            const bool me = aact && lvl == st - 1 && par >= 0;
            if (me) {
              const float* hx = par > c ? S.Hx[pidx(c, par)] : S.Hx[pidx(par, c)];
              for (int j = 0; j < NLK; j++) Xs[j] = par > c ? hx[s * 4 + j] : hx[j * 4 + s];  // H(a,P)[ia = j][iP = s]
              fwd4(L, Linv, Xs);
            }
#pragma unroll
            for (int j = 0; j < NLK; j++) {
              const float v = Xs[0] * quad_get(Xs[0], j) + Xs[1] * quad_get(Xs[1], j) + Xs[2] * quad_get(Xs[2], j) + Xs[3] * quad_get(Xs[3], j);
              if (me && j <= s) atomicAdd(&S.Hbb[par][tri(s, j)], -v);
            }
#pragma unroll
            for (int q6 = 0; q6 < 6; q6++) {
              float v = 0.f;
#pragma unroll
              for (int j = 0; j < NLK; j++) v += Xs[j] * (q6 < 4 ? quad_get(Ya[j], q6) : quad_get(Yb[j], q6 - 4));
              if (me) atomicAdd(&S.Hcb[par][s * 6 + q6], -v);
            }
            if (me) atomicAdd(&S.g[6 + 4 * par + s], Xs[0] * zb[0] + Xs[1] * zb[1] + Xs[2] * zb[2] + Xs[3] * zb[3]);
            WSYNC();
            if (lvl == st) factor_block();  // the updated blocks
          }
        }
        WSYNC();
        V5_TICK(6)
        {
          // Schur complement: Hcc[q][r] -= sum over chains of Y_q . Y_r, rhs6[q] -= sum of Y_q . zb; lane s of a chain owns q in {s, s+4} and fetches Y_r from
          // its chain-mates; the four chains' terms are added with two row rotations and the first chain's lane applies the total
          float dar[6], dbr[6];  // the four chains' terms of row s (dar) and row 4 + s (dbr), column r6, summed over the chains: the same numbers in every chain's lane s
#pragma unroll
          for (int r6 = 0; r6 < 6; r6++) {
            float Yr[NLK];
#pragma unroll
            for (int j = 0; j < NLK; j++) Yr[j] = r6 < 4 ? quad_get(Ya[j], r6) : quad_get(Yb[j], r6 - 4);
            float da = Ya[0] * Yr[0] + Ya[1] * Yr[1] + Ya[2] * Yr[2] + Ya[3] * Yr[3], db = Yb[0] * Yr[0] + Yb[1] * Yr[1] + Yb[2] * Yr[2] + Yb[3] * Yr[3];
            dar[r6] = qsum4_same(da); dbr[r6] = qsum4_same(db);
            continue;
            da = qsum4(da); db = qsum4(db);
            const bool app = aact && c == 0;
            if (app) {
              // (LDS atomics, not `S.Hcc[..] -= da`: the compiler cannot prove the addresses distinct and turns every read-modify-write into its own LDS round trip --
              // ds_read, wait, ds_write -- fourteen of them in a row per iteration; x + (-da) is the same number, and ds_add_f32 is not waited for: 78.4 against 79.4 ms)
              if (r6 <= s) atomicAdd(&S.Hcc[tri(s, r6)], -da);
              if (hasb && r6 <= 4 + s) atomicAdd(&S.Hcc[tri(4 + s, r6)], -db);
            }
          }
          float ra = Ya[0] * zb[0] + Ya[1] * zb[1] + Ya[2] * zb[2] + Ya[3] * zb[3], rb = Yb[0] * zb[0] + Yb[1] * zb[1] + Yb[2] * zb[2] + Yb[3] * zb[3];
          ra = qsum4_same(ra); rb = qsum4_same(rb);
          // the complement stays in registers: every lane needs all of it for the redundant 6 x 6 solve anyway -- 27 quad broadcasts instead of 27 atomics, a fence and
          // the wait for them
          float Lc[21];
#pragma unroll
          for (int q6 = 0; q6 < 6; q6++) {
#pragma unroll
            for (int r6 = 0; r6 <= q6; r6++) Lc[tri(q6, r6)] = (r6 == q6 ? S.Hcc[tri(q6, r6)] + S.cmd[q6 < 3 ? 0 : q6 - 2] : S.Hcc[tri(q6, r6)]) - (q6 < 4 ? quad_get(dar[r6], q6) : quad_get(dbr[r6], q6 - 4));
            xc6[q6] = S.rhs6[q6] - (q6 < 4 ? quad_get(ra, q6) : quad_get(rb, q6 - 4));
          }
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
          // back-substitution through the coupling: pc = L^-T (zb - sum_q Y_q x_q); each lane contributes its columns, summed over the chain
          asm volatile("" : "+v"(xc6[0]), "+v"(xc6[1]), "+v"(xc6[2]), "+v"(xc6[3]), "+v"(xc6[4]), "+v"(xc6[5]));  // (the solution is complete in every lane before anything is picked from it: the picks stay selects)
          float xa = sel4o(xc6, s), xb = s == 0 ? xc6[4] : xc6[5]; asm volatile("" : "+v"(xb));
#pragma unroll
          for (int j = 0; j < NLK; j++) pc4[j] = zb[j] - csum(fmaf(Ya[j], xa, Yb[j] * xb));  // (Yb is zero in the lanes without a second column)
          if constexpr (HC) {
            if (__any(aact && nlev > 0)) {
              // p_a = L_a^-T (zb_a - Y_a x_c - X p_P): parents finish first (last stage first) and publish their part of the direction
#pragma unroll 1
              for (int st = NCH - 1; st >= 0; st--) {
                if (!__any(aact && nlev >= st)) continue;
                const bool me = aact && lvl == st;
                const float pbs = (me && par >= 0) ? S.p[6 + 4 * par + s] : 0.f;
#pragma unroll
                for (int j = 0; j < NLK; j++) pc4[j] -= csum(Xs[j] * pbs);
                if (st > 0) {
                  float pb4[NLK];
                  for (int j = 0; j < NLK; j++) pb4[j] = pc4[j];
                  bwd4(L, Linv, pb4);
                  if (me) S.p[6 + l] = sel4(pb4, s);
                  WSYNC();
                }
              }
            }
          }
          bwd4(L, Linv, pc4);
        }
        asm volatile("" : "+v"(pc4[0]), "+v"(pc4[1]), "+v"(pc4[2]), "+v"(pc4[3]));
        float p_own = sel4o(pc4, s);
        float xcl;  // own entry of the cube's part of the direction (lanes 0..5), zero elsewhere: selects with opaque results (see sel4o)
        { float x03 = sel4o(xc6, l & 3), x45 = (l & 1) ? xc6[5] : xc6[4]; asm volatile("" : "+v"(x45)); xcl = l < 4 ? x03 : (l < 6 ? x45 : 0.f); asm volatile("" : "+v"(xcl)); }
        S.p[6 + l] = p_own; if (l < 6) S.p[l] = xcl;
        WSYNC();
        // ---- (4b) dense path: rollouts with a contact between two finger chains (hand self-collision; rare).  H = M + J'WJ as a packed 22 x 22 matrix in
        // LDS, Cholesky by rows in registers and the two triangular solves with the rollout's 16 lanes (rows l and l + 16)
#ifndef JH_V5_X_NODENSE
        if constexpr (HC && DENSE) {
#ifdef JH_V5_COUNT
        if (lane == 0) { cnt_it++; cnt_dense += __any(act && dense_row) ? 1 : 0; }
#endif
        if (__any(act && dense_row)) {
          const bool dact = act && dense_row;
          if (dact) for (int e = l; e < NDH; e += G) S.Hd[e] = 0.f;
          WSYNC();
          if (dact) {
#pragma unroll
            for (int j = 0; j < NLK; j++) if (j <= s) S.Hd[tri(6 + l, 6 + 4 * c + j)] = Mrow[j] + (j == s ? hd : 0.f);
            if (l < 6) { S.Hd[tri(l, l)] = mck; S.g[l] = gcl; }
          }
          WSYNC();
#pragma unroll
          for (int k = 0; k < NS; k++) {
            const Slot& t = sl[k];
            if (!(dact && t.la >= 0)) continue;
            float f[3], Wk[6];
            const float D[3] = {t.D0, t.D1, t.D1};
            cone_eval(t.jar, D, t.Dm, t.mu, t.fri, f, Wk);
            if (Wk[0] == 0.f && Wk[2] == 0.f && Wk[5] == 0.f) continue;
            const float pos[3] = {t.rc[0] + qc[0], t.rc[1] + qc[1], t.rc[2] + qc[2]};
            // three column blocks: X0 = cube (6 columns at dof 0) or side-A link (4 columns at its chain), X1 = side-B link (4 columns)
            float X0[6][3], X1[NLK][3];
            for (int j = 0; j < 6; j++) X0[j][0] = X0[j][1] = X0[j][2] = 0.f;
            for (int j = 0; j < NLK; j++) X1[j][0] = X1[j][1] = X1[j][2] = 0.f;
            int o0 = 0, n0 = 0, o1 = 0, n1 = 0;
            if (t.la == CUBE) {
              n0 = 6;
              for (int q3 = 0; q3 < 3; q3++) {
                X0[q3][0] = -t.fr[q3]; X0[q3][1] = -t.fr[3 + q3]; X0[q3][2] = -t.fr[6 + q3];
#if JH_V5_WORLDROT
                const float ea[3] = {q3 == 0 ? 1.f : 0.f, q3 == 1 ? 1.f : 0.f, q3 == 2 ? 1.f : 0.f}; float c3[3]; cross3(c3, ea, t.rc);
#else
                float ea[3], c3[3]; col3(ea, S.xR[0], q3); cross3(c3, ea, t.rc);
#endif
                X0[3 + q3][0] = -dot3(t.fr, c3); X0[3 + q3][1] = -dot3(t.fr + 3, c3); X0[3 + q3][2] = -dot3(t.fr + 6, c3);
              }
            } else if (t.la > 0) { o0 = 6 + 4 * ((t.la - 1) >> 2); n0 = 1 + ((t.la - 1) & 3); link_cols(S, t.la, pos, t.fr, -1.f, X0); }
            if (t.lb > 0) {
              o1 = 6 + 4 * ((t.lb - 1) >> 2); n1 = 1 + ((t.lb - 1) & 3);
              if (t.la != CUBE && t.la > 0 && o1 == o0) { link_cols(S, t.lb, pos, t.fr, 1.f, X0); n0 = n0 > n1 ? n0 : n1; n1 = 0; }  // same chain: one block
              else link_cols(S, t.lb, pos, t.fr, 1.f, X1);
            }
            if (n1 > 0 && n0 > 0 && o1 < o0) {  // keep the blocks in dof order (X0 before X1) so that every entry lands in the lower triangle
              for (int j = 0; j < NLK; j++) for (int q = 0; q < 3; q++) { const float tmp = X0[j][q]; X0[j][q] = X1[j][q]; X1[j][q] = tmp; }
              int ti = o0; o0 = o1; o1 = ti; ti = n0; n0 = n1; n1 = ti;
            }
#pragma unroll
            for (int u = 0; u < 6; u++) if (u < n0) {
              const float* j3 = X0[u];
              const float G0 = Wk[0] * j3[0] + Wk[1] * j3[1] + Wk[3] * j3[2], G1 = Wk[1] * j3[0] + Wk[2] * j3[1] + Wk[4] * j3[2], G2 = Wk[3] * j3[0] + Wk[4] * j3[1] + Wk[5] * j3[2];
#pragma unroll
              for (int v = 0; v <= u; v++) atomicAdd(&S.Hd[tri(o0 + u, o0 + v)], X0[v][0] * G0 + X0[v][1] * G1 + X0[v][2] * G2);
#pragma unroll
              for (int v = 0; v < NLK; v++) if (v < n1) atomicAdd(&S.Hd[tri(o1 + v, o0 + u)], X1[v][0] * G0 + X1[v][1] * G1 + X1[v][2] * G2);
            }
#pragma unroll
            for (int u = 0; u < NLK; u++) if (u < n1) {
              const float* j3 = X1[u];
              const float G0 = Wk[0] * j3[0] + Wk[1] * j3[1] + Wk[3] * j3[2], G1 = Wk[1] * j3[0] + Wk[2] * j3[1] + Wk[4] * j3[2], G2 = Wk[3] * j3[0] + Wk[4] * j3[1] + Wk[5] * j3[2];
#pragma unroll
              for (int v = 0; v <= u; v++) atomicAdd(&S.Hd[tri(o1 + u, o1 + v)], X1[v][0] * G0 + X1[v][1] * G1 + X1[v][2] * G2);
            }
          }
          WSYNC();
          // Cholesky by rows: lane l keeps rows l and 16 + l (l < 6) of the factor in registers.  Column k: every lane reads row k's finished entries
          // L[k][0..k-1] from LDS (published as they were finished), forms the pivot redundantly and finishes its own rows' entry of column k --
          // one LDS round trip per column, all reads of a column independent of each other
          const int r1 = 16 + l;
          const bool two = l < 6;
          float h0[16], h1[NV];
#pragma unroll
          for (int j = 0; j < 16; j++) h0[j] = (dact && j <= l) ? S.Hd[tri(l, j)] : 0.f;
#pragma unroll
          for (int j = 0; j < NV; j++) h1[j] = (dact && two && j <= r1) ? S.Hd[tri(r1, j)] : 0.f;
#pragma unroll
          for (int kk = 0; kk < NV; kk++) {
            float dp = 0.f, s0 = kk < 16 ? h0[kk < 16 ? kk : 0] : 0.f, s1 = h1[kk];
            const float hkk = dact ? S.Hd[tri(kk, kk)] : 1.f;
#pragma unroll
            for (int j = 0; j < kk; j++) {
              const float t = dact ? S.Hd[tri(kk, j)] : 0.f;
              dp = fmaf(t, t, dp);
              if (kk < 16) s0 = fmaf(-h0[j < 16 ? j : 0], t, s0);
              s1 = fmaf(-h1[j], t, s1);
            }
            const float rk = __frsqrt_rn(fmaxf(hkk - dp, 1e-30f));
            if (kk < 16) h0[kk < 16 ? kk : 0] = kk <= l ? s0 * rk : 0.f;
            h1[kk] = (two && kk <= r1) ? s1 * rk : 0.f;
            if (dact) {
              if (kk < 16 && l > kk) S.Hd[tri(l, kk)] = h0[kk < 16 ? kk : 0];
              if (two && r1 > kk) S.Hd[tri(r1, kk)] = h1[kk];
              if (l == 0) S.dinv[kk] = rk;
            }
            WSYNC();
          }
          // L y = -g: the owner of row k publishes y_k, every lane takes it out of its own rows' right-hand sides
          float b0 = dact ? -S.g[l] : 0.f, b1 = (dact && two) ? -S.g[r1] : 0.f;
#pragma unroll
          for (int kk = 0; kk < NV; kk++) {
            if (dact && (kk < 16 ? l == kk : r1 == kk)) S.p[kk] = (kk < 16 ? b0 : b1) * S.dinv[kk];
            WSYNC();
            const float yk = dact ? S.p[kk] : 0.f;
            if (kk < 16) b0 -= (l > kk ? h0[kk < 16 ? kk : 0] : 0.f) * yk;
            b1 -= ((two && r1 > kk) ? h1[kk] : 0.f) * yk;
          }
          // L' x = y, in place: the owner of row k finishes x_k and takes it out of the entries above
#pragma unroll
          for (int kk = NV - 1; kk >= 0; kk--) {
            if (dact && (kk < 16 ? l == kk : r1 == kk)) {
              const float xk = S.p[kk] * S.dinv[kk];
              S.p[kk] = xk;
#pragma unroll
              for (int j = 0; j < kk; j++) S.p[j] -= (kk < 16 ? h0[j < 16 ? j : 0] : h1[j]) * xk;
            }
            WSYNC();
          }
          if (dact) {
            p_own = S.p[6 + l]; xcl = l < 6 ? S.p[l] : 0.f;
            for (int k = 0; k < 6; k++) xc6[k] = S.p[k];
            for (int j = 0; j < NLK; j++) pc4[j] = S.p[6 + 4 * c + j];
          }
        }
}
#endif
        V5_TICK(9)
        // ---- (5) exact line search along p
        if constexpr (OPQ > 3) forget_slots();
        float Mp_own = 0.f;
#pragma unroll
        for (int j = 0; j < NLK; j++) Mp_own += Mrow[j] * pc4[j];
        const float pMp = gsum(p_own * Mp_own + mck * xcl * xcl);
        const float pMd = gsum(Mp_own * da_own + mck * xcl * dcl);
        const float gp = gsum(g_own * p_own + gcl * xcl);
        if (act && !(gp < 0.f)) act = false;
        {
#if JH_V5_WORLDROT
          const float* wa = xc6 + 3;
#else
          float wa[3]; mulMV(wa, S.xR[0], xc6 + 3);
#endif
#pragma unroll
          for (int k = 0; k < NS; k++) {
            if (!HC && JH_V5_C3CACHE && k == 0) { slot_Jx_first(sl[0], xc6, wa, S.p, sl[0].jp, V5_C3C(0)); continue; }
            if (k > 0 && !((used >> k) & 1u)) continue;  // (wave-uniform: nobody's slot k holds a contact)
            if (sl[k].la >= 0) slot_Jx<HC>(sl[k], S, qc, xc6, wa, S.p, sl[k].jp, V5_C3C(k));
          }
        }
        dr.pf = p_own; dr.pl = dr.lims * p_own;
#if JH_V5_LSKINK
        // Step lengths at which the slope of the 1-D cost (all but) JUMPS: the zero crossing of the own dof's friction-loss row, and for a contact the point where its
        // tangential part passes closest to zero, if it gets there within JH_V5_LSREV of where it is at 0 or at 1 -- Coulomb friction reverses there, and with a cone as
        // narrow as impratio = 100 makes it that is a jump.  A root of the slope AT such a jump is what the long searches of this workload were looking for (Newton from
        // either side lands beyond the jump, inside the bracket, and the bracket shrinks by parts in a thousand per evaluation: 15 % of the wave's searches took 9 to 16
        // evaluations, half of all its evaluations; CPU prototype: oracle/jo_engine.c::jo_set_ls_experiment, tools/proto/ls_experiment.py).  The search tries the candidate
        // closest to the middle of the bracket whenever a Newton step leaves the bracket or an evaluation leaves more than JH_V5_LSSHRINK of it.
        float kc[NS + 1];
        kc[NS] = (dr.fl > 0.f && dr.pf != 0.f) ? -dr.jf * __frcp_rn(dr.pf) : -1.f;
#pragma unroll
        for (int k = 0; k < NS; k++) {
          kc[k] = -1.f;
          if (sl[k].la >= 0) {
            const float U1 = sl[k].jar[1], U2 = sl[k].jar[2], V1 = sl[k].jp[1], V2 = sl[k].jp[2];  // (both tangential rows carry the same friction coefficient: it drops out)
            const float vv = V1 * V1 + V2 * V2, uv = U1 * V1 + U2 * V2, uu = U1 * U1 + U2 * U2;
            if (vv > 0.f) { const float a = -uv * __frcp_rn(vv); if (uu + a * uv <= JH_V5_LSREV * JH_V5_LSREV * fmaxf(uu, uu + 2.f * uv + vv)) kc[k] = a; }
          }
        }
#endif
        V5_TICK(10)
        float lo = 0.f, hi = -1.f, alpha = 1.f; bool lsact = act;
#ifdef JH_V5_CENSUS
        int cen_ls = 0, cen_lsw = 0; const bool cen_act = act;
        if (stats && lane == 0) atomicAdd(stats + 320 + __popcll(__ballot(act && l == 0)), 1);
#endif
        for (int ls = 0; ls < JH_V5_LSMAX && __any(lsact); ls++) {
          float d1, d2;
#ifdef JH_V5_CENSUS
          cen_ls += lsact; cen_lsw++;
#endif
          lane_rows_dir<NS>(sl, dr, alpha, &d1, &d2, used);
          d1 = gsum(d1) + pMd + alpha * pMp; d2 = gsum(d2) + pMp;
#if JH_V5_LSKINK
          bool trouble = false; float nx = alpha;
          if (lsact) {
            if (fabsf(d1) <= lstol * fabsf(gp)) lsact = false;
            else {
              const float wprev = hi >= 0.f ? hi - lo : -1.f;
              if (d1 < 0.f) lo = alpha; else hi = alpha;
              nx = alpha - d1 * __frcp_rn(d2);
              if (hi < 0.f) { if (nx <= lo) nx = 2.f * alpha; }
              else {
                const bool rejected = nx <= lo || nx >= hi;
                trouble = rejected || (wprev > 0.f && hi - lo > JH_V5_LSSHRINK * wprev);
                if (rejected) nx = 0.5f * (lo + hi);
              }
            }
          }
          if (__any(trouble)) {  // the candidate closest to the middle of the bracket, strictly inside it: |a - mid| with the side in the last mantissa bit, one row minimum
            const float mid = 0.5f * (lo + hi), eps = 1e-6f * hi;
            int key = 0x7f800000;
#pragma unroll
            for (int k = 0; k <= NS; k++) {
              const float a = kc[k], dm = a - mid;
              if (trouble && a > lo + eps && a < hi - eps) key = min(key, (__float_as_int(fabsf(dm)) & ~1) | (dm < 0.f ? 1 : 0));
            }
            key = gmini(trouble ? key : 0x7f800000);
            if (trouble && key != 0x7f800000) { const float mag = __int_as_float(key & ~1); nx = (key & 1) ? mid - mag : mid + mag; }
          }
          if (lsact) alpha = nx;
#else
          {  // safeguarded Newton step on the slope, as selects (round 6: the nested branches were a dozen exec-mask instructions per evaluation; the same arithmetic)
            const bool upd = lsact & !(fabsf(d1) <= lstol * fabsf(gp)), neg = d1 < 0.f;
            lo = (upd & neg) ? alpha : lo; hi = (upd & !neg) ? alpha : hi;
#if JH_V5_LSRCP
            float nx = alpha - d1 * __builtin_amdgcn_rcpf(d2);
#else
            float nx = alpha - d1 * __frcp_rn(d2);
#endif
            const bool out_lo = nx <= lo, out_hi = nx >= hi;
            float dbl = 2.f * alpha, mid_ = 0.5f * (lo + hi); asm volatile("" : "+v"(dbl), "+v"(mid_));  // (both computed: selects, not branches)
            nx = hi < 0.f ? (out_lo ? dbl : nx) : ((out_lo | out_hi) ? mid_ : nx);
            alpha = upd ? nx : alpha; lsact = upd;
          }
#endif
        }
#ifdef JH_V5_CENSUS
        if (stats) { if (l == 0 && live && cen_act) atomicAdd(stats + 256 + min(cen_ls, 31), 1); if (lane == 0) atomicAdd(stats + 288 + min(cen_lsw, 31), 1); }
#endif
        V5_TICK(11)
        // ---- (6) step
        if (act) {
          a_own += alpha * p_own; ac_own += alpha * xcl;
#pragma unroll
          for (int k = 0; k < NS; k++) if (sl[k].la >= 0) for (int rw = 0; rw < 3; rw++) sl[k].jar[rw] += alpha * sl[k].jp[rw];
          dr.jf += alpha * dr.pf; dr.jl += alpha * dr.pl;
          if (-gp * alpha <= tol * tol * fmaxf(snorm, 1e-12f)) act = false;
        }
        WSYNC();
        V5_TICK(12)
      }
      };
      // (the rare slot-count copy instantiates the dense-capable loop only -- it serves rollouts without a dense row as well: three copies of the loop instead of four, and
      // the common one came out 0.9 % faster for it)
      if constexpr (HC && NS > NSLOT) newton_loop(std::true_type{});
      else if constexpr (HC && NS == NSLOT) { if (__any(dense_row)) newton_loop(std::true_type{}); else newton_loop(std::false_type{}); }
      else newton_loop(std::false_type{});
      if (l == 0) { n_iters += iters_this; n_maxed += (iters_this >= cap); }
    }
    return true;
    };
    if (__builtin_expect_with_probability(NSBIG > NSLOT && __any(S.ncon > 16 * NSLOT), 0, JH_V5_BIGPROB)) {
      if (NOVF > 0 && __any(S.ncon > NCP)) __threadfence();  // the overflow rows were written with plain global stores by other lanes of this wave
      solve_step(std::integral_constant<int, NSBIG>{}, std::integral_constant<bool, SELF>{});
    } else {
      bool done = false;
      if constexpr (JH_V5_NS1 == 1 && NSLOT > 1) {
        if (__builtin_expect_with_probability(!__any(S.ncon > 16), 1, JH_V5_NS1PROB)) done = solve_step(std::integral_constant<int, 1>{}, std::integral_constant<bool, SELF>{});
      }
      if constexpr (JH_V5_NS1 == 2 && NSLOT > 1 && SELF) {  // (the one-slot copy only without the hand's code)
        if (!__any(hand_hits) && !__any(S.ncon > 16)) done = solve_step(std::integral_constant<int, 1>{}, std::false_type{});
      }
#if JH_V5_HCSPLIT
      // A wave-step in which no rollout has a candidate pair off the cube has cube contacts only: it takes the copy of the solver compiled without the code for the hand's own
      // contacts (link on side A, chain-coupling blocks, staged elimination, dense direction) -- the same expressions for what remains, fewer live values around them.
      if constexpr (SELF) { if (!done && !__any(hand_hits)) done = solve_step(std::integral_constant<int, NSLOT>{}, std::false_type{}); }
#endif
      if (!done) solve_step(std::integral_constant<int, NSLOT>{}, std::integral_constant<bool, SELF>{});
    }
#ifdef JH_V5_CENSUS
    if (stats) { if (l == 0 && live) atomicAdd(stats + 192 + min(iters_this, 31), 1); if (lane == 0) atomicAdd(stats + 224 + min(n_wave_iters - wave_it0, 31), 1); }
#endif
    // ================================================================ implicitfast integration: (M + h diag(d + kv)) qacc = fs + M (a - a0)
    {
#if JH_V5_WORLDROT
      S.ws[6 + l] = a_own; if (l < 6) { S.acn[l] = ac_own; if (l < 3) S.ws[l] = ac_own; }  // (rotational part: world here; body frame below)
#else
      S.ws[6 + l] = a_own; if (l < 6) { S.ws[l] = ac_own; S.acn[l] = ac_own; }
#endif
#if JH_V5_PARK
      q = S.pk_q[l]; qd = S.qv[6 + l]; fs_own = S.pk_fs[l];  // (own entries: no other lane wrote them)
#endif
      const float da_own = a_own - a0_own;
      float rhs_own = fs_own, x4[NLK], L[10];
#pragma unroll
      for (int j = 0; j < NLK; j++) rhs_own += Mrow[j] * quad_get(da_own, j);
#pragma unroll
      for (int j = 0; j < NLK; j++) x4[j] = quad_get(rhs_own, j);
      for (int k = 0; k < 10; k++) L[k] = S.Mbb[c][k];
      const float hdk = h * (lc[LC_DAMP] + lc[LC_KVD]);
#pragma unroll
      for (int j = 0; j < NLK; j++) L[tri(j, j)] += quad_get(hdk, j);
      float inv4[4]; chol4(L, inv4); fwd4(L, inv4, x4); bwd4(L, inv4, x4);
      const float qacc = sel4(x4, s);
      qd = fmaf(h, qacc, qd); q = fmaf(h, qd, q);
      WSYNC();
#if JH_V5_PARK
      for (int k = 0; k < 6; k++) vc[k] = S.qv[k];
      for (int k = 0; k < 4; k++) qc[3 + k] = S.pk_cq[k];
      if (!MATERIALIZE) acc = S.pk_acc;
#endif
#if JH_V5_WORLDROT
      {  // the constrained rotational acceleration back in the body frame (MuJoCo's free-joint convention): integrated, and kept as the next step's warm start
        const float aw[3] = {S.acn[3], S.acn[4], S.acn[5]}; float ab[3]; mulMTV(ab, S.xR[0], aw);
        for (int k = 0; k < 3; k++) { vc[k] = fmaf(h, S.acn[k], vc[k]); vc[3 + k] = fmaf(h, ab[k], vc[3 + k]); }
        if (l < 3) S.ws[3 + l] = l == 0 ? ab[0] : (l == 1 ? ab[1] : ab[2]);
      }
#else
      for (int k = 0; k < 6; k++) vc[k] = fmaf(h, S.acn[k], vc[k]);  // the cube's inertia is diagonal: its new acceleration is the constrained one itself
#endif
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
    V5_TICK(7)
    if (MATERIALIZE) {
      if (states && live) {
        float* o = states + ((size_t)nc * H + hh) * NX;
        o[7 + l] = q; o[NQ + 6 + l] = qd;
        if (l < 7) o[l] = qc[l];
        if (l < 6) o[NQ + l] = vc[l];
      }
    } else acc += leap_step_cost(sTp, qc);
    WSYNC();
  }
#ifdef JH_V5_COUNT
  if (stats) { if (lane == 0) { atomicAdd(stats + 24, cnt_dense); atomicAdd(stats + 25, cnt_it); atomicAdd(stats + 26, cnt_l2); atomicAdd(stats + 28, H); } if (l == 0 && live) { atomicAdd(stats + 27, cnt_bp); atomicAdd(stats + 29, cnt_hh); for (int k = 0; k < 4; k++) atomicAdd(stats + 30 + k, cnt_cls[k]); } }
#endif
#ifdef JH_V5_TICKS
  if (lane == 0 && stats) { for (int k = 0; k < 8; k++) atomicAdd((unsigned long long*)(stats + 4) + k, (unsigned long long)cyc[k]);
                            for (int k = 0; k < 16; k++) atomicAdd((unsigned long long*)(stats + 384) + k, (unsigned long long)cyc[k]); }
#endif
  if (!MATERIALIZE && live && l == 0) costs[n] = acc / (float)H;
  if (stats && live && l == 0) { if (n_maxed) atomicAdd(stats + 1, n_maxed); atomicAdd(stats + 2, n_iters); atomicAdd(stats + 3, H); }
  if (stats && lane == 0 && live) { atomicAdd(stats + 20, n_wave_iters); atomicAdd(stats + 21, H); }
}

bool model_is_leap(const jh_model* m) {
  return m->kind == JH_TASK_LEAP_CUBE && m->nq == 23 && m->nv == 22 && m->nu == 16 && (m->ns == NS || (m->ns == NS_CALTECH && m->h_i.size() > 18 && m->h_i[18] > 0)) && m->h_i.size() > 17 &&
         m->h_i[0] == 17 && m->h_i[1] == 4 && m->h_i[11] > 0 && m->h_i[5] <= MAXG && m->h_i[12] <= MAXLG && m->h_i[17] <= MAXBP
#if JH_V5_WORLDROT
         && m->h_f.size() > (size_t)HF_CINERTIA + 2 && m->h_f[HF_CINERTIA] == m->h_f[HF_CINERTIA + 1] && m->h_f[HF_CINERTIA] == m->h_f[HF_CINERTIA + 2]  // isotropic cube inertia (JH_V5_WORLDROT)
#endif
         ;
}

}  // namespace

#ifdef JH_V5_X_DYNRS
#define JH_V5_DYNBYTES (sizeof(RS) * RPW * JH_V5_WPB)
#else
#define JH_V5_DYNBYTES 0
#endif
#ifndef JH_V5_NAME
#define JH_V5_NAME(f) f
#endif
int JH_V5_NAME(jh_engine5_rollout_cost)(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W,
                            const float* lohi, const float* tp, int N, int n_offset, int H, int K, float* costs, float* knots_out, float* trace, hipStream_t st) {
  if (!model_is_leap(m)) { jh_set_error("rollout_cost: the cooperative engine kernel is instantiated for leap_cube only"); return JH_ERR_UNSUPPORTED; }
#if JH_V5_KNOTS_LDS
  JH_REQUIRE(K <= MAXK, "rollout_cost: the cooperative leap kernel keeps at most 8 knots per actuator (K=%d)", K);
#endif
  const int dshift = jh_latency_shift(N, RPW); const int per_block = (RPW >> dshift) * JH_V5_WPB;
  int grid = (N + per_block - 1) / per_block;
  float* ovf = nullptr;  // one row per rollout for the contacts above the LDS pool: stream-ordered allocation, no state on the model handle
  if (NOVF > 0) ovf = jh_launch_scratch(m, (size_t)N * NOVF * POOL_F * sizeof(float), st);  // (nullptr: the LDS capacity alone, drops and the fallback counted)
  if (m->self_collision && m->h_i[17] > 0)
    hipLaunchKernelGGL((k_leap_v5<false, JH_V5_WPB, true>), dim3(grid), dim3(WAVE * JH_V5_WPB), JH_V5_DYNBYTES, st, m->d_f, m->d_i, x0, 0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs,
                       knots_out, (const float*)nullptr, (float*)nullptr, (float*)nullptr, m->d_stats, dshift, trace, ovf);
  else
    hipLaunchKernelGGL((k_leap_v5<false, JH_V5_WPB, false>), dim3(grid), dim3(WAVE * JH_V5_WPB), JH_V5_DYNBYTES, st, m->d_f, m->d_i, x0, 0, nominal, noise, ldn, sigma, W, lohi, tp, N, n_offset, H, K, costs,
                       knots_out, (const float*)nullptr, (float*)nullptr, (float*)nullptr, m->d_stats, dshift, trace, ovf);
  return jh_launch_done(ovf, st);
}

int JH_V5_NAME(jh_engine5_materialize)(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                           hipStream_t st) {
  if (!model_is_leap(m)) { jh_set_error("rollout_materialize: the cooperative engine kernel is instantiated for leap_cube only"); return JH_ERR_UNSUPPORTED; }
  const int dshift = jh_latency_shift(N, RPW); const int per_block = (RPW >> dshift) * JH_V5_WPB;
  int grid = (N + per_block - 1) / per_block;
  float* ovf = nullptr;  // one row per rollout for the contacts above the LDS pool: stream-ordered allocation, no state on the model handle
  if (NOVF > 0) ovf = jh_launch_scratch(m, (size_t)N * NOVF * POOL_F * sizeof(float), st);  // (nullptr: the LDS capacity alone, drops and the fallback counted)
  if (m->self_collision && m->h_i[17] > 0)
    hipLaunchKernelGGL((k_leap_v5<true, JH_V5_WPB, true>), dim3(grid), dim3(WAVE * JH_V5_WPB), JH_V5_DYNBYTES, st, m->d_f, m->d_i, x0, x0_batched, (const float*)nullptr, (const float*)nullptr, 0,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, N, 0, H, 0, (float*)nullptr, (float*)nullptr,
                       controls, states, sensors, m->d_stats, dshift, (float*)nullptr, ovf);
  else
    hipLaunchKernelGGL((k_leap_v5<true, JH_V5_WPB, false>), dim3(grid), dim3(WAVE * JH_V5_WPB), JH_V5_DYNBYTES, st, m->d_f, m->d_i, x0, x0_batched, (const float*)nullptr, (const float*)nullptr, 0,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, N, 0, H, 0, (float*)nullptr, (float*)nullptr,
                       controls, states, sensors, m->d_stats, dshift, (float*)nullptr, ovf);
  return jh_launch_done(ovf, st);
}
