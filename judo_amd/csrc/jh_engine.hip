// jh_engine.hip -- generic articulated-body rollout engine, ONE LANE PER ROLLOUT (reference kernel): leap_cube and fr3_pick, gfx950.
// This is the CROSS-CHECK implementation (jh_model_set_kernel(m, 1)), not a production path: leap_cube runs on the cooperative jh_engine_v5.hip (generation 3;
// jh_engine_v2.hip is generation 2), fr3_pick on jh_engine_v3.hip.  It is model-generic -- explicit geom pairs with arbitrary sides, pyramidal or elliptic cones,
// joint equalities, geom-distance sensors -- and leap_cube's contacts here are the cube's only (no hand self-collision).
//
// One lane owns one rollout.  Per step (MuJoCo's pipeline, restated -- see oracle/jo_engine.c for the fp64 checker and
// DESIGN.md section 5 for the derivation):
//   kinematics -> per-chain joint-space inertia (block diagonal: the hand is welded to the world, so
//   M = diag(cube 6x6 diagonal, one small dense block per finger chain)) -> Newton-Euler bias -> position servos ->
//   unconstrained acceleration -> cube-vs-hand collision (box-box SAT + face clipping, box-sphere) -> soft constraints
//   (dof friction loss, joint limits, elliptic-cone contacts, impratio) -> primal Newton with exact line search on an
//   ARROW Hessian (cube block + chain blocks + cube-chain couplings; chains are eliminated first, a 6x6 Schur
//   complement remains) -> implicitfast integration -> task cost accumulated in the same loop.
// The contact Jacobian is never stored: J*x and J'*f are evaluated from the contact point, its frame and the joint
// axes (a handful of cross products).  Model constants are staged in LDS once per workgroup; the lane's clipped spline
// knots live in LDS (lane-fastest); noise is read coalesced once; one float (the cost) is written per rollout.
// Precision: fp32 everywhere except the Newton Hessian (assembly + Cholesky + solve), which is fp64 -- the stiff
// friction rows (impratio 100) against 1e-5 kg m^2 finger inertias give it a condition number ~1e7, and CDNA4's fp64
// VALU rate is half its fp32 rate, so this costs little.
#include "jh_engine_common.h"

using namespace jh_eng;

namespace {

constexpr int kBlock = 64;
// ------------------------------------------------------------------------------------------------ per-lane working set
template <class C>
struct Work {
  float qpos[C::NQ], qvel[C::NV], qws[C::NV];
  float xpos[C::NM][3], xR[C::NM][9], axw[C::NM][3];
  float Mb[C::NBLK][C::TRI];
  float fs[C::NV], a0[C::NV], a[C::NV];
  // constraints
  int ncon, overflow, iters, maxed, cur_bodyA, cur_pair;
#ifdef JH_ENGINE_PROFILE
  long long cyc[8], t0;
#endif
  float cpos[C::NCON][3], cfr[C::NCON][9], caref[C::NCON][3], cD[C::NCON][3], cmu[C::NCON], cfri[C::NCON];
  int cbody[C::NCON], cbodyA[C::NCON], cpair[C::NCON];  // geom 2's body / geom 1's body: -1 static, 0 the free body, >0 articulated
  float faref[C::NV], lims[C::NV], laref[C::NV], lD[C::NV];
  float earef[2], eD[2];  // joint equalities
};

// ------------------------------------------------------------------------------------------------ kinematics
template <class C>
__device__ void kinematics(const EngineModel& m, Work<C>& w) {
  for (int b = 0; b < C::NM; b++) {
    const float* bf = m.F + m.oBodyF + b * BODY_F; const int* bi = m.I + m.oBodyI + b * BODY_I;
    int par = bi[0], jt = bi[1], qa = bi[3];
    if (jt == JFREE) {
      float* q = w.qpos + qa + 3;
      float nn = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      q[0] *= nn; q[1] *= nn; q[2] *= nn; q[3] *= nn;
      w.xpos[b][0] = w.qpos[qa]; w.xpos[b][1] = w.qpos[qa + 1]; w.xpos[b][2] = w.qpos[qa + 2];
      quat2mat(w.xR[b], q);
      w.axw[b][0] = w.axw[b][1] = w.axw[b][2] = 0.f;
    } else {
      float P[3], R0[9];
      if (par < 0) { for (int k = 0; k < 3; k++) P[k] = bf[BF_LPOS + k]; for (int k = 0; k < 9; k++) R0[k] = bf[BF_LR + k]; }
      else { mulMV(P, w.xR[par], bf + BF_LPOS); for (int k = 0; k < 3; k++) P[k] += w.xpos[par][k]; mulMM(R0, w.xR[par], bf + BF_LR); }
      const float* al = bf + BF_AXIS;
      mulMV(w.axw[b], R0, al);
      float q = w.qpos[qa];
      if (jt == JHINGE) {  // Rodrigues about the local axis
        float s, c; sincosf(q, &s, &c); float t = 1.f - c, x = al[0], y = al[1], z = al[2];
        float Rq[9] = {t * x * x + c, t * x * y - s * z, t * x * z + s * y, t * x * y + s * z, t * y * y + c, t * y * z - s * x, t * x * z - s * y, t * y * z + s * x, t * z * z + c};
        mulMM(w.xR[b], R0, Rq);
      } else {
        for (int k = 0; k < 3; k++) P[k] += w.axw[b][k] * q;
        for (int k = 0; k < 9; k++) w.xR[b][k] = R0[k];
      }
      for (int k = 0; k < 3; k++) w.xpos[b][k] = P[k];
    }
  }
}


// ------------------------------------------------------------------------------------------------ block inertia + bias + smooth forces
template <class C>
__device__ void smooth_dynamics(const EngineModel& m, Work<C>& w, const float* ctrl) {
  const float* F = m.F;
  const float grav[3] = {F[HF_GRAV], F[HF_GRAV + 1], F[HF_GRAV + 2]};
  // ---- free body (centre of mass at the origin, principal axes = body axes): M = diag(m,m,m,I1,I2,I3)
  {
    float mass = F[HF_CMASS]; const float* I3 = F + HF_CINERTIA; const float* om = w.qvel + 3;
    float Iw[3] = {I3[0] * om[0], I3[1] * om[1], I3[2] * om[2]}, g[3]; cross3(g, om, Iw);
    for (int k = 0; k < 3; k++) { w.fs[k] = mass * grav[k]; w.a0[k] = grav[k]; w.fs[3 + k] = -g[k]; w.a0[3 + k] = -g[k] / I3[k]; }
  }
  // ---- articulated blocks
  for (int c = 0; c < C::NBLK; c++) {
    const int* blk = m.I + m.oBlockI + c * BLOCK_I;
    const int first = blk[0], nb = blk[1], d0 = blk[2];
    float* M = w.Mb[c];
    for (int i = 0; i < C::TRI; i++) M[i] = 0.f;
    // composite inertia by direct summation over (body, ancestor, ancestor) triples
    for (int k = first; k < first + nb; k++) {
      const float* bf = F + m.oBodyF + k * BODY_F;
      float Rk[9], ck[3]; mulMM(Rk, w.xR[k], bf + BF_IR); mulMV(ck, w.xR[k], bf + BF_IPOS);
      for (int q = 0; q < 3; q++) ck[q] += w.xpos[k][q];
      float mass = bf[BF_MASS];
      for (int i = k; i >= first; i = m.I[m.oBodyI + i * BODY_I]) {
        int jti = m.I[m.oBodyI + i * BODY_I + 1];
        float Jvi[3], tA[3], tB[3] = {0, 0, 0};
        if (jti == JHINGE) { float r[3] = {ck[0] - w.xpos[i][0], ck[1] - w.xpos[i][1], ck[2] - w.xpos[i][2]}; cross3(Jvi, w.axw[i], r); inertia_mul(tB, Rk, bf + BF_INERTIA, w.axw[i]); }
        else { Jvi[0] = w.axw[i][0]; Jvi[1] = w.axw[i][1]; Jvi[2] = w.axw[i][2]; }
        tA[0] = mass * Jvi[0]; tA[1] = mass * Jvi[1]; tA[2] = mass * Jvi[2];
        for (int j = i; j >= first; j = m.I[m.oBodyI + j * BODY_I]) {
          int jtj = m.I[m.oBodyI + j * BODY_I + 1];
          float v;
          if (jtj == JHINGE) { float r[3] = {ck[0] - w.xpos[j][0], ck[1] - w.xpos[j][1], ck[2] - w.xpos[j][2]}, Jvj[3]; cross3(Jvj, w.axw[j], r); v = dot3(tA, Jvj) + dot3(tB, w.axw[j]); }
          else v = dot3(tA, w.axw[j]);
          M[tri(i - first, j - first)] += v;
        }
      }
    }
    for (int l = 0; l < nb; l++) M[tri(l, l)] += F[m.oDofF + (d0 + l) * DOF_F + DF_ARM];
    // ---- recursive Newton-Euler with q'' = 0 (gravity as base acceleration)
    float om[C::BD][3], al[C::BD][3], ao[C::BD][3], ff[C::BD][3], nn[C::BD][3];
    for (int k = first; k < first + nb; k++) {
      const int l = k - first; const int* bi = m.I + m.oBodyI + k * BODY_I; const float* bf = F + m.oBodyF + k * BODY_F;
      int par = bi[0], jt = bi[1]; float qd = w.qvel[bi[2]];
      float wp[3] = {0, 0, 0}, ap[3] = {0, 0, 0}, aop[3] = {-grav[0], -grav[1], -grav[2]}, d[3] = {0, 0, 0};
      if (par >= first) { int lp = par - first; for (int q = 0; q < 3; q++) { wp[q] = om[lp][q]; ap[q] = al[lp][q]; aop[q] = ao[lp][q]; d[q] = w.xpos[k][q] - w.xpos[par][q]; } }
      float t1[3], t2[3]; cross3(t1, wp, d); cross3(t2, wp, t1); cross3(t1, ap, d);
      for (int q = 0; q < 3; q++) ao[l][q] = aop[q] + t1[q] + t2[q];
      float wxa[3]; cross3(wxa, wp, w.axw[k]);
      if (jt == JHINGE) { for (int q = 0; q < 3; q++) { om[l][q] = wp[q] + w.axw[k][q] * qd; al[l][q] = ap[q] + wxa[q] * qd; } }
      else { for (int q = 0; q < 3; q++) { om[l][q] = wp[q]; al[l][q] = ap[q]; ao[l][q] += 2.f * wxa[q] * qd; } }
      float Rk[9], r[3]; mulMM(Rk, w.xR[k], bf + BF_IR); mulMV(r, w.xR[k], bf + BF_IPOS);
      float ac[3]; cross3(t1, om[l], r); cross3(t2, om[l], t1); cross3(t1, al[l], r);
      for (int q = 0; q < 3; q++) ac[q] = ao[l][q] + t1[q] + t2[q];
      float Iw[3], Ia[3], g[3]; inertia_mul(Iw, Rk, bf + BF_INERTIA, om[l]); inertia_mul(Ia, Rk, bf + BF_INERTIA, al[l]); cross3(g, om[l], Iw);
      float Fk[3] = {bf[BF_MASS] * ac[0], bf[BF_MASS] * ac[1], bf[BF_MASS] * ac[2]}, rxF[3]; cross3(rxF, r, Fk);
      for (int q = 0; q < 3; q++) { ff[l][q] = Fk[q]; nn[l][q] = Ia[q] + g[q] + rxF[q]; }
    }
    for (int k = first + nb - 1; k >= first; k--) {
      const int l = k - first; const int* bi = m.I + m.oBodyI + k * BODY_I;
      int par = bi[0], jt = bi[1], dof = bi[2];
      float bias = jt == JHINGE ? dot3(w.axw[k], nn[l]) : dot3(w.axw[k], ff[l]);
      w.fs[dof] = -F[m.oDofF + dof * DOF_F + DF_DAMP] * w.qvel[dof] - bias;
      if (par >= first) {
        int lp = par - first; float d[3] = {w.xpos[k][0] - w.xpos[par][0], w.xpos[k][1] - w.xpos[par][1], w.xpos[k][2] - w.xpos[par][2]}, dxf[3];
        cross3(dxf, d, ff[l]);
        for (int q = 0; q < 3; q++) { ff[lp][q] += ff[l][q]; nn[lp][q] += nn[l][q] + dxf[q]; }
      }
    }
  }
  // ---- position servos: clamp ctrl, force = kp (ctrl - q) - kv qdot; joint-level actuator force clamp
  float fact[C::NV];
  for (int d = 0; d < C::NV; d++) fact[d] = 0.f;
  for (int u = 0; u < C::NU; u++) {
    const float* af = F + m.oActF + u * ACT_F; const int* ai = m.I + m.oActI + u * ACT_I;
    float cc = ctrl[u];
    if (af[AF_CLIM] != 0.f) cc = jh_clampf(cc, af[AF_CLO], af[AF_CHI]);
    fact[ai[0]] += af[AF_KP] * (cc - w.qpos[ai[1]]) - af[AF_KV] * w.qvel[ai[0]];
  }
  for (int d = 6; d < C::NV; d++) {
    const float* df = F + m.oDofF + d * DOF_F;
    float fa = fact[d];
    if (df[DF_FRCLIM] != 0.f) fa = jh_clampf(fa, df[DF_FRCLO], df[DF_FRCHI]);
    w.fs[d] += fa;
  }
  // ---- unconstrained acceleration per block (Cholesky of the small dense block)
  for (int c = 0; c < C::NBLK; c++) {
    const int* blk = m.I + m.oBlockI + c * BLOCK_I; const int nb = blk[1], d0 = blk[2];
    float L[C::TRI];
    for (int i = 0; i < nb; i++)
      for (int j = 0; j <= i; j++) {
        float s = w.Mb[c][tri(i, j)];
        for (int k = 0; k < j; k++) s -= L[tri(i, k)] * L[tri(j, k)];
        L[tri(i, j)] = (i == j) ? sqrtf(fmaxf(s, 1e-30f)) : s / L[tri(j, j)];
      }
    float x[C::BD];
    for (int i = 0; i < nb; i++) { float s = w.fs[d0 + i]; for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k]; x[i] = s / L[tri(i, i)]; }
    for (int i = nb - 1; i >= 0; i--) { float s = x[i]; for (int k = i + 1; k < nb; k++) s -= L[tri(k, i)] * x[k]; x[i] = s / L[tri(i, i)]; }
    for (int i = 0; i < nb; i++) w.a0[d0 + i] = x[i];
  }
}

// ------------------------------------------------------------------------------------------------ collision (cube vs hand geoms)
template <class C>
__device__ __forceinline__ void push_contact(Work<C>& w, const float* pos, const float* n, float dist, int body, float mu, float tran) {
  if (w.ncon >= C::NCON) { w.overflow++; return; }
  int i = w.ncon++;
  w.cbodyA[i] = w.cur_bodyA; w.cpair[i] = w.cur_pair;
  for (int k = 0; k < 3; k++) { w.cpos[i][k] = pos[k]; w.cfr[i][k] = n[k]; }
  make_frame(w.cfr[i]);
  w.caref[i][0] = dist;  // finished in constraint_rows
  w.cD[i][0] = tran;     // diagApprox carried here until then
  w.cbody[i] = body; w.cfri[i] = mu;
}

// box (p1,R1,h1 = the cube, geom 1) against box (p2,R2,h2); normal from box 1 to box 2
template <class C>
__device__ void collide_box_box(Work<C>& w, const float* p1, const float* R1, const float* h1, const float* p2, const float* R2, const float* h2, int body, float mu, float tran) {
  float A[3][3], B[3][3], dv[3], Cm[3][3], AC[3][3], dA[3], dB[3];
  for (int k = 0; k < 3; k++) { col3(A[k], R1, k); col3(B[k], R2, k); dv[k] = p2[k] - p1[k]; }
  for (int i = 0; i < 3; i++) { dA[i] = dot3(dv, A[i]); dB[i] = dot3(dv, B[i]); for (int j = 0; j < 3; j++) { Cm[i][j] = dot3(A[i], B[j]); AC[i][j] = fabsf(Cm[i][j]); } }
  float best = -1e30f; int btype = -1, bi = 0, bj = 0;
  for (int i = 0; i < 3; i++) {
    float s = fabsf(dA[i]) - (h1[i] + h2[0] * AC[i][0] + h2[1] * AC[i][1] + h2[2] * AC[i][2]);
    if (s > 0.f) return;
    if (s > best) { best = s; btype = 0; bi = i; }
  }
  for (int j = 0; j < 3; j++) {
    float s = fabsf(dB[j]) - (h2[j] + h1[0] * AC[0][j] + h1[1] * AC[1][j] + h1[2] * AC[2][j]);
    if (s > 0.f) return;
    if (s > best) { best = s; btype = 1; bj = j; }
  }
  float ebest = -1e30f, eL[3] = {0, 0, 0}; int ei = -1, ej = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float L[3]; cross3(L, A[i], B[j]);
      float l2 = dot3(L, L);
      if (l2 < 1e-12f) continue;
      float il = rsqrtf(l2); L[0] *= il; L[1] *= il; L[2] *= il;
      float ra = 0.f, rb = 0.f;
      for (int k = 0; k < 3; k++) { ra += h1[k] * fabsf(dot3(A[k], L)); rb += h2[k] * fabsf(dot3(B[k], L)); }
      float s = fabsf(dot3(dv, L)) - (ra + rb);
      if (s > 0.f) return;
      if (s > ebest) { ebest = s; ei = i; ej = j; eL[0] = L[0]; eL[1] = L[1]; eL[2] = L[2]; }
    }
  bool use_edge = ei >= 0 && (best < 0.f ? ebest > best / 1.05f + 1e-12f : ebest > best * 1.05f + 1e-12f);
  if (use_edge) {
    float n[3] = {eL[0], eL[1], eL[2]};
    if (dot3(n, dv) < 0.f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    float pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
    for (int k = 0; k < 3; k++) {
      if (k != ei) { float s = (dot3(n, A[k]) > 0.f ? 1.f : -1.f) * h1[k]; pa[0] += A[k][0] * s; pa[1] += A[k][1] * s; pa[2] += A[k][2] * s; }
      if (k != ej) { float s = (dot3(n, B[k]) > 0.f ? -1.f : 1.f) * h2[k]; pb[0] += B[k][0] * s; pb[1] += B[k][1] * s; pb[2] += B[k][2] * s; }
    }
    float wv[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    float b = Cm[ei][ej], dd = dot3(A[ei], wv), e = dot3(B[ej], wv), den = 1.f - b * b;
    float s = den > 1e-12f ? (b * e - dd) / den : 0.f, t = den > 1e-12f ? (e - b * dd) / den : 0.f;
    s = jh_clampf(s, -h1[ei], h1[ei]); t = jh_clampf(t, -h2[ej], h2[ej]);
    float pos[3];
    for (int k = 0; k < 3; k++) pos[k] = 0.5f * ((pa[k] + A[ei][k] * s) + (pb[k] + B[ej][k] * t));
    push_contact(w, pos, n, ebest, body, mu, tran);
    return;
  }
  // face contact: clip the incident face against the reference face
  const float *pr, *pi, *hr, *hi; float (*Ar)[3], (*Ai)[3]; int ri; float n[3];
  if (btype == 0) { pr = p1; pi = p2; hr = h1; hi = h2; Ar = A; Ai = B; ri = bi; float sg = dA[bi] >= 0.f ? 1.f : -1.f; for (int k = 0; k < 3; k++) n[k] = sg * A[bi][k]; }
  else { pr = p2; pi = p1; hr = h2; hi = h1; Ar = B; Ai = A; ri = bj; float sg = dB[bj] >= 0.f ? -1.f : 1.f; for (int k = 0; k < 3; k++) n[k] = sg * B[bj][k]; }
  int mi = 0; float mb = -1.f;
  for (int k = 0; k < 3; k++) { float v = fabsf(dot3(n, Ai[k])); if (v > mb) { mb = v; mi = k; } }
  float sgi = dot3(n, Ai[mi]) > 0.f ? -1.f : 1.f;
  int u = (mi + 1) % 3, v = (mi + 2) % 3;
  float poly[16][3], tmp[16][3]; int np = 4;
  const float su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
  for (int q = 0; q < 4; q++)
    for (int k = 0; k < 3; k++) poly[q][k] = pi[k] + sgi * hi[mi] * Ai[mi][k] + su[q] * hi[u] * Ai[u][k] + sv[q] * hi[v] * Ai[v][k];
  int ra = (ri + 1) % 3, rb = (ri + 2) % 3;
  for (int pl = 0; pl < 4 && np > 0; pl++) {
    const float* ax = Ar[pl < 2 ? ra : rb]; float sg = (pl & 1) ? -1.f : 1.f, lim = hr[pl < 2 ? ra : rb];
    int nq = 0;
    for (int q = 0; q < np; q++) {
      const float* P = poly[q]; const float* Q = poly[(q + 1) % np];
      float dp[3] = {P[0] - pr[0], P[1] - pr[1], P[2] - pr[2]}, dq[3] = {Q[0] - pr[0], Q[1] - pr[1], Q[2] - pr[2]};
      float fp = sg * dot3(dp, ax) - lim, fq = sg * dot3(dq, ax) - lim;
      if (fp <= 0.f) { tmp[nq][0] = P[0]; tmp[nq][1] = P[1]; tmp[nq][2] = P[2]; nq++; }
      if ((fp < 0.f && fq > 0.f) || (fp > 0.f && fq < 0.f)) { float t = fp / (fp - fq); for (int k = 0; k < 3; k++) tmp[nq][k] = P[k] + t * (Q[k] - P[k]); nq++; }
    }
    np = nq;
    for (int q = 0; q < np; q++) { poly[q][0] = tmp[q][0]; poly[q][1] = tmp[q][1]; poly[q][2] = tmp[q][2]; }
  }
  int nc = 0;
  for (int q = 0; q < np && nc < 8; q++) {
    float dx[3] = {poly[q][0] - pr[0], poly[q][1] - pr[1], poly[q][2] - pr[2]};
    float depth = hr[ri] - dot3(dx, n);
    if (-depth >= 0.f) continue;
    float pos[3], nn[3];
    for (int k = 0; k < 3; k++) { pos[k] = poly[q][k] + 0.5f * depth * n[k]; nn[k] = btype == 0 ? n[k] : -n[k]; }
    push_contact(w, pos, nn, -depth, body, mu, tran);
    nc++;
  }
}

template <class C>
__device__ void collide_box_sphere(Work<C>& w, const float* pb, const float* Rb, const float* hb, const float* c, float r, int body, float mu, float tran) {
  float dl[3] = {c[0] - pb[0], c[1] - pb[1], c[2] - pb[2]}, cl[3], q[3]; bool outside = false;
  mulMTV(cl, Rb, dl);
  for (int k = 0; k < 3; k++) { q[k] = cl[k]; if (q[k] > hb[k]) { q[k] = hb[k]; outside = true; } else if (q[k] < -hb[k]) { q[k] = -hb[k]; outside = true; } }
  float nl[3], dist;
  if (outside) {
    float df[3] = {cl[0] - q[0], cl[1] - q[1], cl[2] - q[2]}; float l = sqrtf(dot3(df, df));
    if (l - r >= 0.f) return;
    nl[0] = df[0] / l; nl[1] = df[1] / l; nl[2] = df[2] / l; dist = l - r;
  } else {
    int kb = 0; float mn = 1e30f;
    for (int k = 0; k < 3; k++) { float s = hb[k] - fabsf(cl[k]); if (s < mn) { mn = s; kb = k; } }
    nl[0] = nl[1] = nl[2] = 0.f; nl[kb] = cl[kb] >= 0.f ? 1.f : -1.f;
    q[kb] = nl[kb] * hb[kb]; dist = -mn - r;
  }
  float ql[3] = {q[0] + 0.5f * dist * nl[0], q[1] + 0.5f * dist * nl[1], q[2] + 0.5f * dist * nl[2]}, pos[3], n[3];
  mulMV(pos, Rb, ql); for (int k = 0; k < 3; k++) pos[k] += pb[k];
  mulMV(n, Rb, nl);
  push_contact(w, pos, n, dist, body, mu, tran);
}

template <class C>
__device__ __forceinline__ void geom_pose(const EngineModel& m, const Work<C>& w, int g, float* gp, float* gR, bool want_R) {
  const float* gf = m.F + m.oAGF + g * GEOM_F; int body = m.I[m.oAGI + g * GEOM_I];
  if (body < 0) { for (int k = 0; k < 3; k++) gp[k] = gf[GF_POS + k]; if (want_R) for (int k = 0; k < 9; k++) gR[k] = gf[GF_R + k]; }
  else {
    mulMV(gp, w.xR[body], gf + GF_POS); for (int k = 0; k < 3; k++) gp[k] += w.xpos[body][k];
    if (want_R) mulMM(gR, w.xR[body], gf + GF_R);
  }
}

template <class C>
__device__ void collision(const EngineModel& m, Work<C>& w) {
  w.ncon = 0;
  const float* F = m.F;
  for (int pi = 0; pi < m.NPAIR; pi++) {
    int g1 = m.I[m.oPairI + 2 * pi], g2 = m.I[m.oPairI + 2 * pi + 1];
    const float* f1 = F + m.oAGF + g1 * GEOM_F; const float* f2 = F + m.oAGF + g2 * GEOM_F;
    int t1 = m.I[m.oAGI + g1 * GEOM_I + 1], t2 = m.I[m.oAGI + g2 * GEOM_I + 1];
    float p1[3], p2[3], R1[9], R2[9];
    geom_pose(m, w, g1, p1, R1, false); geom_pose(m, w, g2, p2, R2, false);
    float dc[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, rs = f1[GF_RBOUND] + f2[GF_RBOUND];
    if (dot3(dc, dc) > rs * rs) continue;  // bounding-sphere filter
    float mu = fmaxf(f1[GF_MU], f2[GF_MU]), tran = f1[GF_TRAN] + f2[GF_TRAN];
    int b1 = m.I[m.oAGI + g1 * GEOM_I], b2 = m.I[m.oAGI + g2 * GEOM_I];
    w.cur_pair = pi;
    if (t1 == GBOX && t2 == GBOX) {
      geom_pose(m, w, g1, p1, R1, true); geom_pose(m, w, g2, p2, R2, true);
      w.cur_bodyA = b1;
      collide_box_box(w, p1, R1, f1 + GF_SIZE, p2, R2, f2 + GF_SIZE, b2, mu, tran);
    } else if (t1 == GBOX && t2 == GSPHERE) {
      geom_pose(m, w, g1, p1, R1, true);
      w.cur_bodyA = b1;
      collide_box_sphere(w, p1, R1, f1 + GF_SIZE, p2, f2[GF_SIZE], b2, mu, tran);
    } else if (t1 == GBOX && t2 == GCAPSULE) {  // the capsule's two end spheres and, when it lies inside the segment, the point of its axis closest to the box
      geom_pose(m, w, g1, p1, R1, true); geom_pose(m, w, g2, p2, R2, true);
      w.cur_bodyA = b1;
      float axis[3]; col3(axis, R2, 2);
      const float L = f2[GF_SIZE + 1], tm = capsule_box_closest(p1, R1, f1 + GF_SIZE, p2, axis, L);
      for (int e = 0; e < 3; e++) {
        const float t = e == 0 ? L : (e == 1 ? -L : tm);
        if (e == 2 && tm > L) break;
        const float c[3] = {p2[0] + t * axis[0], p2[1] + t * axis[1], p2[2] + t * axis[2]};
        collide_box_sphere(w, p1, R1, f1 + GF_SIZE, c, f2[GF_SIZE], b2, mu, tran);
      }
    } else if (t1 == GSPHERE && t2 == GBOX) {  // normal must point from geom 1 (sphere) to geom 2 (box): swap the sides of the box-sphere routine
      geom_pose(m, w, g2, p2, R2, true);
      int before = w.ncon;
      w.cur_bodyA = b1;
      collide_box_sphere(w, p2, R2, f2 + GF_SIZE, p1, f1[GF_SIZE], b2, mu, tran);
      for (int i = before; i < w.ncon; i++) { for (int k = 0; k < 9; k++) w.cfr[i][k] = (k < 3) ? -w.cfr[i][k] : w.cfr[i][k]; make_frame(w.cfr[i]); }
    }
  }
}

// ------------------------------------------------------------------------------------------------ matrix-free contact Jacobian
// velocity of the point p (world) fixed to moving body `body` for joint-rate vector x (block dofs only)
template <class C>
__device__ __forceinline__ void body_point_vel(const EngineModel& m, const Work<C>& w, int body, const float* p, const float* x, float* v) {
  v[0] = v[1] = v[2] = 0.f;
  for (int i = body; i > 0; i = m.I[m.oBodyI + i * BODY_I]) {
    const int* bi = m.I + m.oBodyI + i * BODY_I; float xi = x[bi[2]];
    if (bi[1] == JHINGE) { float r[3] = {p[0] - w.xpos[i][0], p[1] - w.xpos[i][1], p[2] - w.xpos[i][2]}, c[3]; cross3(c, w.axw[i], r); v[0] += c[0] * xi; v[1] += c[1] * xi; v[2] += c[2] * xi; }
    else { v[0] += w.axw[i][0] * xi; v[1] += w.axw[i][1] * xi; v[2] += w.axw[i][2] * xi; }
  }
}
template <class C>
__device__ __forceinline__ void body_point_force(const EngineModel& m, const Work<C>& w, int body, const float* p, const float* Fw, float* g) {
  for (int i = body; i > 0; i = m.I[m.oBodyI + i * BODY_I]) {
    const int* bi = m.I + m.oBodyI + i * BODY_I;
    if (bi[1] == JHINGE) { float r[3] = {p[0] - w.xpos[i][0], p[1] - w.xpos[i][1], p[2] - w.xpos[i][2]}, c[3]; cross3(c, r, Fw); g[bi[2]] += dot3(w.axw[i], c); }
    else g[bi[2]] += dot3(w.axw[i], Fw);
  }
}
template <class C>
__device__ __forceinline__ void cube_point_vel(const Work<C>& w, const float* p, const float* x, float* v) {
  float ow[3], r[3] = {p[0] - w.xpos[0][0], p[1] - w.xpos[0][1], p[2] - w.xpos[0][2]}, c[3];
  mulMV(ow, w.xR[0], x + 3); cross3(c, ow, r);
  v[0] = x[0] + c[0]; v[1] = x[1] + c[1]; v[2] = x[2] + c[2];
}
template <class C>
__device__ __forceinline__ void cube_point_force(const Work<C>& w, const float* p, const float* Fw, float* g) {
  float r[3] = {p[0] - w.xpos[0][0], p[1] - w.xpos[0][1], p[2] - w.xpos[0][2]}, c[3], cl[3];
  cross3(c, r, Fw); mulMTV(cl, w.xR[0], c);
  g[0] += Fw[0]; g[1] += Fw[1]; g[2] += Fw[2]; g[3] += cl[0]; g[4] += cl[1]; g[5] += cl[2];
}
// velocity of / generalized force on a world point attached to `body` (-1 static, 0 the free body, >0 articulated)
template <class C>
__device__ __forceinline__ void side_point_vel(const EngineModel& m, const Work<C>& w, int body, const float* p, const float* x, float* v) {
  if (body < 0) { v[0] = v[1] = v[2] = 0.f; }
  else if (body == 0) cube_point_vel(w, p, x, v);
  else body_point_vel(m, w, body, p, x, v);
}
template <class C>
__device__ __forceinline__ void side_point_force(const EngineModel& m, const Work<C>& w, int body, const float* p, const float* Fw, float* g) {
  if (body == 0) cube_point_force(w, p, Fw, g);
  else if (body > 0) body_point_force(m, w, body, p, Fw, g);
}
// contact-frame relative velocity J x = frame * (v_geom2 - v_geom1)
template <class C>
__device__ __forceinline__ void contact_Jx(const EngineModel& m, const Work<C>& w, int i, const float* x, float* out) {
  float vb[3], va[3];
  side_point_vel(m, w, w.cbody[i], w.cpos[i], x, vb);
  side_point_vel(m, w, w.cbodyA[i], w.cpos[i], x, va);
  float d[3] = {vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]};
  out[0] = dot3(w.cfr[i], d); out[1] = dot3(w.cfr[i] + 3, d); out[2] = dot3(w.cfr[i] + 6, d);
}

// ------------------------------------------------------------------------------------------------ constraint rows
template <class C>
__device__ void constraint_rows(const EngineModel& m, Work<C>& w) {
  const float* F = m.F;
  for (int d = 0; d < C::NV; d++) {
    const float* df = F + m.oDofF + d * DOF_F;
    w.faref[d] = -df[DF_FB] * w.qvel[d];
    w.lims[d] = 0.f; w.laref[d] = 0.f; w.lD[d] = 0.f;
    if (df[DF_LIMITED] != 0.f) {
      int qa = d + 1;  // hinge / slide dofs follow the free joint: qpos index = dof index + 1
      float q = w.qpos[qa], dlo = q - df[DF_LO], dhi = df[DF_HI] - q, dist = fminf(dlo, dhi);
      if (dist < 0.f) {
        float s = dlo < dhi ? 1.f : -1.f;
        float imp = impedance(df + DF_SOLIMP, dist);
        float R = fmaxf(1e-15f, (1.f - imp) / imp * df[DF_INVW]);
        w.lims[d] = s; w.lD[d] = 1.f / R;
        w.laref[d] = -df[DF_LB] * (s * w.qvel[d]) - df[DF_LK] * imp * dist;
      }
    }
  }
  const float impratio = F[HF_IMPRATIO];
  for (int i = 0; i < w.ncon; i++) {
    float dist = w.caref[i][0], tran = w.cD[i][0];
    // contact parameters are mixed per pair: solref and solimp averaged (equal solmix), refsafe clamp on the time constant
    const float* q1 = F + m.oGPF + m.I[m.oPairI + 2 * w.cpair[i]] * GP_F; const float* q2 = F + m.oGPF + m.I[m.oPairI + 2 * w.cpair[i] + 1] * GP_F;
    float si[5]; for (int k = 0; k < 5; k++) si[k] = 0.5f * (q1[2 + k] + q2[2 + k]);
    float tc = fmaxf(0.5f * (q1[0] + q2[0]), 2.f * F[HF_DT]), dr = 0.5f * (q1[1] + q2[1]);
    float cK = 1.f / fmaxf(1e-15f, si[1] * si[1] * tc * tc * dr * dr), cB = 2.f / fmaxf(1e-15f, si[1] * tc);
    float imp = impedance(si, dist);
    if (m.cone == 1) {  // elliptic: friction rows R/impratio, regularised cone coefficient
      float R0 = fmaxf(1e-15f, (1.f - imp) / imp * tran), R1 = R0 / fmaxf(1e-15f, impratio);
      w.cD[i][0] = 1.f / R0; w.cD[i][1] = 1.f / R1; w.cD[i][2] = 1.f / R1;
      w.cmu[i] = w.cfri[i] * sqrtf(R1 / R0);
    } else {  // pyramidal: every edge row gets Rpy = 2 mu_reg^2 Rn, Rn from diagApprox = tran (1 + mu^2)
      float mu = w.cfri[i], R0 = fmaxf(1e-15f, (1.f - imp) / imp * tran * (1.f + mu * mu));
      float mr2 = mu * mu / fmaxf(1e-15f, impratio);
      float Rpy = fmaxf(1e-15f, 2.f * mr2 * R0);
      w.cD[i][0] = w.cD[i][1] = w.cD[i][2] = 1.f / Rpy;
      w.cmu[i] = mu;
    }
    float vel[3]; contact_Jx(m, w, i, w.qvel, vel);
    w.caref[i][0] = -cB * vel[0] - cK * imp * dist;
    w.caref[i][1] = -cB * vel[1];
    w.caref[i][2] = -cB * vel[2];
  }
  for (int e = 0; e < m.NEQ && e < 2; e++) {  // joint coupling (q1 - q1_0) - a0 - a1 (q2 - q2_0) = 0, always-active quadratic row
    const int* ei = m.I + m.oEqI + e * EQ_I; const float* ef = F + m.oEqF + e * EQ_F;
    float pos = w.qpos[ei[0] + 1] - ef[EF_A0] - ef[EF_A1] * w.qpos[ei[1] + 1];
    float vel = w.qvel[ei[0]] - ef[EF_A1] * w.qvel[ei[1]];
    float imp = impedance(ef + EF_SOLIMP, pos);
    float R = fmaxf(1e-15f, (1.f - imp) / imp * ef[EF_INVW]);
    w.eD[e] = 1.f / R; w.earef[e] = -ef[EF_B] * vel - ef[EF_K] * imp * pos;
  }
}

// ------------------------------------------------------------------------------------------------ Newton solver (arrow Hessian, fp64 linear algebra)
#ifndef JH_HESS_T
#define JH_HESS_T double
#endif
typedef JH_HESS_T hreal;
template <class C>
struct Hess {
  hreal cc[21];
  hreal bb[C::NBLK][C::TRI];
  hreal cb[C::NBLK][6 * C::BD];
};

// row data of the current iterate: jar (= J a - aref) for contacts / friction-loss / limits
template <class C>
struct Rows {
  float cj[C::NCON][3];
  float fj[C::NV], lj[C::NV], ej[2];
};

template <class C>
__device__ void rows_Jx(const EngineModel& m, const Work<C>& w, const float* x, bool subtract_aref, Rows<C>& r) {
  for (int i = 0; i < w.ncon; i++) {
    contact_Jx(m, w, i, x, r.cj[i]);
    if (subtract_aref) { r.cj[i][0] -= w.caref[i][0]; r.cj[i][1] -= w.caref[i][1]; r.cj[i][2] -= w.caref[i][2]; }
  }
  for (int d = 6; d < C::NV; d++) {
    r.fj[d] = x[d] - (subtract_aref ? w.faref[d] : 0.f);
    r.lj[d] = w.lims[d] * x[d] - (subtract_aref ? w.laref[d] : 0.f);
  }
  for (int e = 0; e < m.NEQ && e < 2; e++) {
    const int* ei = m.I + m.oEqI + e * EQ_I;
    r.ej[e] = x[ei[0]] - m.F[m.oEqF + e * EQ_F + EF_A1] * x[ei[1]] - (subtract_aref ? w.earef[e] : 0.f);
  }
}

// M * x for the block-diagonal inertia
template <class C>
__device__ void mul_M(const EngineModel& m, const Work<C>& w, const float* x, float* y) {
  const float* F = m.F;
  for (int k = 0; k < 3; k++) { y[k] = F[HF_CMASS] * x[k]; y[3 + k] = F[HF_CINERTIA + k] * x[3 + k]; }
  for (int c = 0; c < C::NBLK; c++) {
    const int* blk = m.I + m.oBlockI + c * BLOCK_I; const int nb = blk[1], d0 = blk[2];
    for (int i = 0; i < nb; i++) {
      float s = 0.f;
      for (int j = 0; j < nb; j++) s += w.Mb[c][i >= j ? tri(i, j) : tri(j, i)] * x[d0 + j];
      y[d0 + i] = s;
    }
  }
}

// constraint cost, its derivative along a direction and curvature, evaluated at jar = r0 + alpha * rp
template <class C>
__device__ void rows_eval(const EngineModel& m, const Work<C>& w, const Rows<C>& r0, const Rows<C>* rp, float alpha, float* cost, float* d1, float* d2) {
  const float* F = m.F;
  float cs = 0.f, g1 = 0.f, g2 = 0.f;
  for (int i = 0; i < w.ncon; i++) {
    float jar[3], jp[3] = {0, 0, 0}, f[3], W[6];
    for (int k = 0; k < 3; k++) { jar[k] = r0.cj[i][k]; if (rp) { jp[k] = rp->cj[i][k]; jar[k] += alpha * jp[k]; } }
    cs += contact_eval(m.cone, jar, w.cD[i], w.cmu[i], w.cfri[i], f, W);
    if (rp) {
      g1 -= f[0] * jp[0] + f[1] * jp[1] + f[2] * jp[2];
      g2 += W[0] * jp[0] * jp[0] + W[2] * jp[1] * jp[1] + W[5] * jp[2] * jp[2] + 2.f * (W[1] * jp[0] * jp[1] + W[3] * jp[0] * jp[2] + W[4] * jp[1] * jp[2]);
    }
  }
  for (int d = 6; d < C::NV; d++) {
    const float* df = F + m.oDofF + d * DOF_F;
    float fl = df[DF_FL];
    if (fl > 0.f) {  // dof friction loss: Huber-shaped cost, force saturates at +-frictionloss
      float D = df[DF_FD], R = 1.f / D, jp = rp ? rp->fj[d] : 0.f, x = r0.fj[d] + alpha * jp;
      if (x <= -R * fl) { cs += -0.5f * R * fl * fl - fl * x; g1 -= fl * jp; }
      else if (x >= R * fl) { cs += -0.5f * R * fl * fl + fl * x; g1 += fl * jp; }
      else { cs += 0.5f * D * x * x; g1 += D * x * jp; g2 += D * jp * jp; }
    }
    if (w.lims[d] != 0.f) {  // joint limit: one-sided quadratic
      float jp = rp ? rp->lj[d] : 0.f, x = r0.lj[d] + alpha * jp;
      if (x < 0.f) { cs += 0.5f * w.lD[d] * x * x; g1 += w.lD[d] * x * jp; g2 += w.lD[d] * jp * jp; }
    }
  }
  for (int e = 0; e < m.NEQ && e < 2; e++) {
    float jp = rp ? rp->ej[e] : 0.f, x = r0.ej[e] + alpha * jp;
    cs += 0.5f * w.eD[e] * x * x; g1 += w.eD[e] * x * jp; g2 += w.eD[e] * jp * jp;
  }
  *cost = cs; if (d1) *d1 = g1; if (d2) *d2 = g2;
}

template <class C>
__device__ float total_cost(const EngineModel& m, const Work<C>& w, const float* a, Rows<C>& r) {
  rows_Jx(m, w, a, true, r);
  float cs; rows_eval<C>(m, w, r, nullptr, 0.f, &cs, nullptr, nullptr);
  float da[C::NV], Md[C::NV];
  for (int d = 0; d < C::NV; d++) da[d] = a[d] - w.a0[d];
  mul_M(m, w, da, Md);
  for (int d = 0; d < C::NV; d++) cs += 0.5f * da[d] * Md[d];
  return cs;
}

template <class C>
__device__ int solve_constraints(const EngineModel& m, Work<C>& w, int max_iter, float tol) {
  const float* F = m.F;
  bool any = w.ncon > 0 || m.NEQ > 0;
  for (int d = 6; d < C::NV; d++) any |= (F[m.oDofF + d * DOF_F + DF_FL] > 0.f) || (w.lims[d] != 0.f);
  if (!any) { for (int d = 0; d < C::NV; d++) w.a[d] = w.a0[d]; return 0; }
  Rows<C> r, rp;
  {  // warm start: the better of last step's acceleration and the unconstrained one
    float cw = total_cost(m, w, w.qws, r), c0 = total_cost(m, w, w.a0, r);
    const float* src = (cw < c0) ? w.qws : w.a0;
    for (int d = 0; d < C::NV; d++) w.a[d] = src[d];
  }
  float* a = w.a;
  int it = 0;
  for (; it < max_iter; it++) {
    // ---- gradient g = M (a - a0) - J' f, and per-row Hessian weights
    rows_Jx(m, w, a, true, r);
    float g[C::NV], da[C::NV];
    for (int d = 0; d < C::NV; d++) da[d] = a[d] - w.a0[d];
    mul_M(m, w, da, g);
    Hess<C> H;
    for (int k = 0; k < 21; k++) H.cc[k] = 0.0;
    for (int k = 0; k < 3; k++) { H.cc[tri(k, k)] = F[HF_CMASS]; H.cc[tri(3 + k, 3 + k)] = F[HF_CINERTIA + k]; }
    for (int c = 0; c < C::NBLK; c++) { for (int k = 0; k < C::TRI; k++) H.bb[c][k] = w.Mb[c][k]; for (int k = 0; k < 6 * C::BD; k++) H.cb[c][k] = 0.0; }
    for (int i = 0; i < w.ncon; i++) {
      float f[3], W[6];
      contact_eval(m.cone, r.cj[i], w.cD[i], w.cmu[i], w.cfri[i], f, W);
      // J' f: world force on geom 2's body at the contact point, the opposite on geom 1's body; the gradient gets -J'f
      const float* fr = w.cfr[i];
      float Fw[3] = {fr[0] * f[0] + fr[3] * f[1] + fr[6] * f[2], fr[1] * f[0] + fr[4] * f[1] + fr[7] * f[2], fr[2] * f[0] + fr[5] * f[1] + fr[8] * f[2]};
      float nF[3] = {-Fw[0], -Fw[1], -Fw[2]};
      side_point_force(m, w, w.cbody[i], w.cpos[i], nF, g);
      side_point_force(m, w, w.cbodyA[i], w.cpos[i], Fw, g);
      if (W[0] == 0.f && W[2] == 0.f && W[5] == 0.f) continue;
      // explicit Jacobian columns in contact-frame coordinates: the free body's 6 (if it is a side) and a dense row over the
      // dofs of the ONE articulated block involved (side B adds, side A subtracts)
      float Jc[6][3], Jb[C::BD][3]; int blkid = -1, d0 = 0; bool has_cube = false;
      const float* p = w.cpos[i];
      for (int k = 0; k < 6; k++) Jc[k][0] = Jc[k][1] = Jc[k][2] = 0.f;
      for (int k = 0; k < C::BD; k++) Jb[k][0] = Jb[k][1] = Jb[k][2] = 0.f;
      for (int side = 0; side < 2; side++) {
        const int body = side == 0 ? w.cbodyA[i] : w.cbody[i]; const float sg = side == 0 ? -1.f : 1.f;
        if (body == 0) {
          has_cube = true;
          float rr[3] = {p[0] - w.xpos[0][0], p[1] - w.xpos[0][1], p[2] - w.xpos[0][2]};
          for (int k = 0; k < 3; k++) { Jc[k][0] += sg * fr[k]; Jc[k][1] += sg * fr[3 + k]; Jc[k][2] += sg * fr[6 + k]; }
          for (int k = 0; k < 3; k++) { float ax[3], c3[3]; col3(ax, w.xR[0], k); cross3(c3, ax, rr); Jc[3 + k][0] += sg * dot3(fr, c3); Jc[3 + k][1] += sg * dot3(fr + 3, c3); Jc[3 + k][2] += sg * dot3(fr + 6, c3); }
        } else if (body > 0) {
          blkid = m.I[m.oBodyI + body * BODY_I + 4];
          d0 = m.I[m.oBlockI + blkid * BLOCK_I + 2];
          for (int bq = body; bq > 0; bq = m.I[m.oBodyI + bq * BODY_I]) {
            const int* bi = m.I + m.oBodyI + bq * BODY_I; float c3[3];
            if (bi[1] == JHINGE) { float rb[3] = {p[0] - w.xpos[bq][0], p[1] - w.xpos[bq][1], p[2] - w.xpos[bq][2]}; cross3(c3, w.axw[bq], rb); }
            else { c3[0] = w.axw[bq][0]; c3[1] = w.axw[bq][1]; c3[2] = w.axw[bq][2]; }
            int lq = bi[2] - d0;
            Jb[lq][0] += sg * dot3(fr, c3); Jb[lq][1] += sg * dot3(fr + 3, c3); Jb[lq][2] += sg * dot3(fr + 6, c3);
          }
        }
      }
      // G = W J ; H += J' G
      float Gc[6][3], Gb[C::BD][3];
      for (int k = 0; k < 6; k++) { const float* j = Jc[k]; Gc[k][0] = W[0] * j[0] + W[1] * j[1] + W[3] * j[2]; Gc[k][1] = W[1] * j[0] + W[2] * j[1] + W[4] * j[2]; Gc[k][2] = W[3] * j[0] + W[4] * j[1] + W[5] * j[2]; }
      if (has_cube) for (int u = 0; u < 6; u++) for (int v = 0; v <= u; v++) H.cc[tri(u, v)] += (hreal)(Jc[u][0] * Gc[v][0] + Jc[u][1] * Gc[v][1] + Jc[u][2] * Gc[v][2]);
      if (blkid >= 0) {
        const int nbk = m.I[m.oBlockI + blkid * BLOCK_I + 1];
        for (int k = 0; k < nbk; k++) { const float* j = Jb[k]; Gb[k][0] = W[0] * j[0] + W[1] * j[1] + W[3] * j[2]; Gb[k][1] = W[1] * j[0] + W[2] * j[1] + W[4] * j[2]; Gb[k][2] = W[3] * j[0] + W[4] * j[1] + W[5] * j[2]; }
        for (int u = 0; u < nbk; u++) {
          for (int v = 0; v <= u; v++) H.bb[blkid][tri(u, v)] += (hreal)(Jb[u][0] * Gb[v][0] + Jb[u][1] * Gb[v][1] + Jb[u][2] * Gb[v][2]);
          if (has_cube) for (int q = 0; q < 6; q++) H.cb[blkid][q * C::BD + u] += (hreal)(Jc[q][0] * Gb[u][0] + Jc[q][1] * Gb[u][1] + Jc[q][2] * Gb[u][2]);
        }
      }
    }
    for (int e = 0; e < m.NEQ && e < 2; e++) {  // joint equality: quadratic row over two dofs of one block
      const int* ei = m.I + m.oEqI + e * EQ_I; const float a1 = F[m.oEqF + e * EQ_F + EF_A1];
      float Dx = w.eD[e] * r.ej[e];
      g[ei[0]] += Dx; g[ei[1]] -= a1 * Dx;
      int l1 = ei[3], l2 = ei[4], bk = ei[2];
      H.bb[bk][tri(l1, l1)] += (hreal)w.eD[e]; H.bb[bk][tri(l2, l2)] += (hreal)(a1 * a1 * w.eD[e]);
      H.bb[bk][l1 > l2 ? tri(l1, l2) : tri(l2, l1)] -= (hreal)(a1 * w.eD[e]);
    }
    for (int c = 0; c < C::NBLK; c++) {
      const int* blk = m.I + m.oBlockI + c * BLOCK_I; const int nb = blk[1], d0 = blk[2];
      for (int l = 0; l < nb; l++) {
        int d = d0 + l; const float* df = F + m.oDofF + d * DOF_F; float fl = df[DF_FL];
        if (fl > 0.f) {
          float D = df[DF_FD], R = 1.f / D, x = r.fj[d];
          if (x <= -R * fl) g[d] -= fl; else if (x >= R * fl) g[d] += fl; else { g[d] += D * x; H.bb[c][tri(l, l)] += (hreal)D; }
        }
        if (w.lims[d] != 0.f && r.lj[d] < 0.f) { g[d] += w.lims[d] * w.lD[d] * r.lj[d]; H.bb[c][tri(l, l)] += (hreal)w.lD[d]; }
      }
    }
    // ---- convergence: gradient norm scaled by the inertia diagonal
    float gn = 0.f, sn = 0.f;
    for (int k = 0; k < 3; k++) { gn += g[k] * g[k] / F[HF_CMASS] + g[3 + k] * g[3 + k] / F[HF_CINERTIA + k]; sn += w.fs[k] * w.fs[k] / F[HF_CMASS] + w.fs[3 + k] * w.fs[3 + k] / F[HF_CINERTIA + k]; }
    for (int c = 0; c < C::NBLK; c++) {
      const int* blk = m.I + m.oBlockI + c * BLOCK_I; const int nb = blk[1], d0 = blk[2];
      for (int l = 0; l < nb; l++) { float mi = 1.f / w.Mb[c][tri(l, l)]; gn += g[d0 + l] * g[d0 + l] * mi; sn += w.fs[d0 + l] * w.fs[d0 + l] * mi; }
    }
    if (gn <= tol * tol * fmaxf(sn, 1e-12f)) break;
    // ---- arrow Cholesky: chains first, Schur complement on the cube block
    for (int c = 0; c < C::NBLK; c++) {
      const int nb = m.I[m.oBlockI + c * BLOCK_I + 1];
      hreal* L = H.bb[c];
      for (int i = 0; i < nb; i++)
        for (int j = 0; j <= i; j++) {
          hreal s = L[tri(i, j)];
          for (int k = 0; k < j; k++) s -= L[tri(i, k)] * L[tri(j, k)];
          L[tri(i, j)] = (i == j) ? (hreal)sqrt(fmax((double)s, 1e-30)) : s / L[tri(j, j)];
        }
      for (int q = 0; q < 6; q++) {
        hreal* y = H.cb[c] + q * C::BD;
        for (int i = 0; i < nb; i++) { hreal s = y[i]; for (int k = 0; k < i; k++) s -= L[tri(i, k)] * y[k]; y[i] = s / L[tri(i, i)]; }
      }
      for (int q = 0; q < 6; q++) for (int s2 = 0; s2 <= q; s2++) { hreal acc = 0; for (int i = 0; i < nb; i++) acc += H.cb[c][q * C::BD + i] * H.cb[c][s2 * C::BD + i]; H.cc[tri(q, s2)] -= acc; }
    }
    for (int i = 0; i < 6; i++)
      for (int j = 0; j <= i; j++) {
        hreal s = H.cc[tri(i, j)];
        for (int k = 0; k < j; k++) s -= H.cc[tri(i, k)] * H.cc[tri(j, k)];
        H.cc[tri(i, j)] = (i == j) ? (hreal)sqrt(fmax((double)s, 1e-30)) : s / H.cc[tri(j, j)];
      }
    // ---- p = -H^-1 g
    hreal zc[6], zb[C::NBLK][C::BD];
    for (int q = 0; q < 6; q++) zc[q] = -(hreal)g[q];
    for (int c = 0; c < C::NBLK; c++) {
      const int* blk = m.I + m.oBlockI + c * BLOCK_I; const int nb = blk[1], d0 = blk[2]; const hreal* L = H.bb[c];
      for (int i = 0; i < nb; i++) { hreal s = -(hreal)g[d0 + i]; for (int k = 0; k < i; k++) s -= L[tri(i, k)] * zb[c][k]; zb[c][i] = s / L[tri(i, i)]; }
      for (int q = 0; q < 6; q++) { hreal acc = 0; for (int i = 0; i < nb; i++) acc += H.cb[c][q * C::BD + i] * zb[c][i]; zc[q] -= acc; }
    }
    for (int i = 0; i < 6; i++) { hreal s = zc[i]; for (int k = 0; k < i; k++) s -= H.cc[tri(i, k)] * zc[k]; zc[i] = s / H.cc[tri(i, i)]; }
    for (int i = 5; i >= 0; i--) { hreal s = zc[i]; for (int k = i + 1; k < 6; k++) s -= H.cc[tri(k, i)] * zc[k]; zc[i] = s / H.cc[tri(i, i)]; }
    float p[C::NV];
    for (int q = 0; q < 6; q++) p[q] = (float)zc[q];
    for (int c = 0; c < C::NBLK; c++) {
      const int* blk = m.I + m.oBlockI + c * BLOCK_I; const int nb = blk[1], d0 = blk[2]; const hreal* L = H.bb[c];
      for (int i = 0; i < nb; i++) { hreal acc = 0; for (int q = 0; q < 6; q++) acc += H.cb[c][q * C::BD + i] * zc[q]; zb[c][i] -= acc; }
      for (int i = nb - 1; i >= 0; i--) { hreal s = zb[c][i]; for (int k = i + 1; k < nb; k++) s -= L[tri(k, i)] * zb[c][k]; zb[c][i] = s / L[tri(i, i)]; }
      for (int i = 0; i < nb; i++) p[d0 + i] = (float)zb[c][i];
    }
    // ---- exact line search along p: safeguarded 1-D Newton on phi'(alpha)
    rows_Jx(m, w, p, false, rp);
    float Mp[C::NV]; mul_M(m, w, p, Mp);
    float pMp = 0.f, pMd = 0.f, gp = 0.f;
    for (int d = 0; d < C::NV; d++) { pMp += p[d] * Mp[d]; pMd += Mp[d] * da[d]; gp += g[d] * p[d]; }
    if (!(gp < 0.f)) break;
    const float lstol = F[HF_LSTOL];
    float lo = 0.f, hi = -1.f, al = 1.f;
    for (int ls = 0; ls < 12; ls++) {
      float cs, d1, d2;
      rows_eval<C>(m, w, r, &rp, al, &cs, &d1, &d2);
      d1 += pMd + al * pMp; d2 += pMp;
      if (fabsf(d1) <= lstol * fabsf(gp)) break;
      if (d1 < 0.f) lo = al; else hi = al;
      float nx = al - d1 / d2;
      if (hi < 0.f) { if (nx <= lo) nx = 2.f * al; }
      else if (nx <= lo || nx >= hi) nx = 0.5f * (lo + hi);
      al = nx;
    }
    for (int d = 0; d < C::NV; d++) a[d] += al * p[d];
    if (-gp * al <= tol * tol * fmaxf(sn, 1e-12f)) { it++; break; }  // expected decrease below the noise floor
  }
  return it;
}

// ------------------------------------------------------------------------------------------------ one mj_step
template <class C>
__device__ void engine_step(const EngineModel& m, Work<C>& w, const float* ctrl) {
  const float* F = m.F; const float h = F[HF_DT];
  // forward dynamics was prepared by engine_forward(); integrate (implicitfast): (M + h*diag(damping + kv)) qacc = fs + J'f
  // J'f = M (a - a0) at the solver's fixed point, so the right-hand side is fs + M (a - a0)
  float da[C::NV], rhs[C::NV];
  for (int d = 0; d < C::NV; d++) da[d] = w.a[d] - w.a0[d];
  mul_M(m, w, da, rhs);
  for (int d = 0; d < C::NV; d++) rhs[d] += w.fs[d];
  float qacc[C::NV];
  for (int k = 0; k < 3; k++) { qacc[k] = rhs[k] / F[HF_CMASS]; qacc[3 + k] = rhs[3 + k] / F[HF_CINERTIA + k]; }
  for (int c = 0; c < C::NBLK; c++) {
    const int* blk = m.I + m.oBlockI + c * BLOCK_I; const int nb = blk[1], d0 = blk[2];
    float L[C::TRI];
    for (int i = 0; i < nb; i++)
      for (int j = 0; j <= i; j++) {
        float s = w.Mb[c][tri(i, j)];
        if (i == j) { const float* df = F + m.oDofF + (d0 + i) * DOF_F; s += h * (df[DF_DAMP] + df[DF_KV]); }
        for (int k = 0; k < j; k++) s -= L[tri(i, k)] * L[tri(j, k)];
        L[tri(i, j)] = (i == j) ? sqrtf(fmaxf(s, 1e-30f)) : s / L[tri(j, j)];
      }
    float x[C::BD];
    for (int i = 0; i < nb; i++) { float s = rhs[d0 + i]; for (int k = 0; k < i; k++) s -= L[tri(i, k)] * x[k]; x[i] = s / L[tri(i, i)]; }
    for (int i = nb - 1; i >= 0; i--) { float s = x[i]; for (int k = i + 1; k < nb; k++) s -= L[tri(k, i)] * x[k]; x[i] = s / L[tri(i, i)]; }
    for (int i = 0; i < nb; i++) qacc[d0 + i] = x[i];
  }
  for (int d = 0; d < C::NV; d++) { w.qvel[d] = fmaf(h, qacc[d], w.qvel[d]); w.qws[d] = w.a[d]; }
  // positions with the new velocity; free joint: world-frame translation, body-frame rotation
  for (int k = 0; k < 3; k++) w.qpos[k] = fmaf(h, w.qvel[k], w.qpos[k]);
  {
    float* q = w.qpos + 3; const float* om = w.qvel + 3;
    float wn = sqrtf(dot3(om, om)), ang = wn * h;
    if (ang > 0.f) {
      float s, c; sincosf(0.5f * ang, &s, &c); float k = s / wn;
      float dq[4] = {c, om[0] * k, om[1] * k, om[2] * k};
      float r0 = q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2] - q[3] * dq[3];
      float r1 = q[0] * dq[1] + q[1] * dq[0] + q[2] * dq[3] - q[3] * dq[2];
      float r2 = q[0] * dq[2] - q[1] * dq[3] + q[2] * dq[0] + q[3] * dq[1];
      float r3 = q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1] + q[3] * dq[0];
      q[0] = r0; q[1] = r1; q[2] = r2; q[3] = r3;
    }
    float nn = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] *= nn; q[1] *= nn; q[2] *= nn; q[3] *= nn;
  }
  for (int d = 6; d < C::NV; d++) w.qpos[d + 1] = fmaf(h, w.qvel[d], w.qpos[d + 1]);
  (void)ctrl;
}

#ifdef JH_ENGINE_PROFILE
#define JH_TICK(slot) { long long t__ = clock64(); w.cyc[slot] += t__ - w.t0; w.t0 = t__; }
#else
#define JH_TICK(slot)
#endif
template <class C>
__device__ void engine_forward(const EngineModel& m, Work<C>& w, const float* ctrl) {
  JH_TICK(7)
  kinematics(m, w);
  JH_TICK(0)
  smooth_dynamics(m, w, ctrl);
  JH_TICK(1)
  collision(m, w);
  JH_TICK(2)
  constraint_rows(m, w);
  JH_TICK(3)
  const int cap = (int)m.F[HF_MAXITER];
  int it = solve_constraints(m, w, cap, m.F[HF_TOL]);
  w.iters += it; w.maxed += (it >= cap);
  JH_TICK(4)
}

// signed box-box distance = largest separation over the 15 SAT axes (exact when a face or an edge pair is closest; see the oracle)
__device__ float box_box_distance(const float* p1, const float* R1, const float* h1, const float* p2, const float* R2, const float* h2) {
  float A[3][3], B[3][3], dv[3], best = -1e30f;
  for (int k = 0; k < 3; k++) { col3(A[k], R1, k); col3(B[k], R2, k); dv[k] = p2[k] - p1[k]; }
  for (int i = 0; i < 3; i++) {
    float ra = h1[i], rb = 0.f; for (int k = 0; k < 3; k++) rb += h2[k] * fabsf(dot3(B[k], A[i]));
    float sA = fabsf(dot3(dv, A[i])) - ra - rb; if (sA > best) best = sA;
    ra = 0.f; rb = h2[i]; for (int k = 0; k < 3; k++) ra += h1[k] * fabsf(dot3(A[k], B[i]));
    float sB = fabsf(dot3(dv, B[i])) - ra - rb; if (sB > best) best = sB;
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float L[3]; cross3(L, A[i], B[j]); float l2 = dot3(L, L);
      if (l2 < 1e-12f) continue;
      float il = rsqrtf(l2); L[0] *= il; L[1] *= il; L[2] *= il;
      float ra = 0.f, rb = 0.f; for (int k = 0; k < 3; k++) { ra += h1[k] * fabsf(dot3(A[k], L)); rb += h2[k] * fabsf(dot3(B[k], L)); }
      float sE = fabsf(dot3(dv, L)) - ra - rb; if (sE > best) best = sE;
    }
  return best;
}

// sensors of the forward pass (position stage): framepos of sites/bodies, jointpos, frame z axes, geom distances
template <class C>
__device__ void engine_sensors(const EngineModel& m, const Work<C>& w, float* y) {
  for (int s = 0; s < m.NGS; s++) {
    const int* si = m.I + m.oSensG + s * 4; int tp = si[0], obj = si[1], adr = si[3];
    if (tp == 0 || tp == 5) {  // site frame: position / z axis
      int b = m.I[m.oFrameI + obj]; const float* ff = m.F + m.oFrameF + obj * FRAME_F;
      if (tp == 0) { float p[3]; mulMV(p, w.xR[b], ff); for (int k = 0; k < 3; k++) y[adr + k] = p[k] + w.xpos[b][k]; }
      else { float zl[3] = {ff[3 + 2], ff[3 + 5], ff[3 + 8]}, z[3]; mulMV(z, w.xR[b], zl); for (int k = 0; k < 3; k++) y[adr + k] = z[k]; }
    }
    else if (tp == 1) { for (int k = 0; k < 3; k++) y[adr + k] = w.xpos[obj][k]; }
    else if (tp == 2) y[adr] = w.qpos[obj];
    else if (tp == 3) { float z[3]; col3(z, w.xR[obj], 2); for (int k = 0; k < 3; k++) y[adr + k] = z[k]; }
    else {  // minimum signed distance between the box geoms of two bodies, clipped at the cutoff
      const int* di = m.I + m.oDistI + obj * 4; float best = m.F[m.oDistF + obj];
      for (int a = 0; a < di[1]; a++) {
        int ga = m.I[m.oGlist + di[0] + a]; float pa[3], Ra[9]; geom_pose(m, w, ga, pa, Ra, true);
        for (int bq = 0; bq < di[3]; bq++) {
          int gb = m.I[m.oGlist + di[2] + bq]; float pb[3], Rb[9]; geom_pose(m, w, gb, pb, Rb, true);
          float dd = box_box_distance(pa, Ra, m.F + m.oAGF + ga * GEOM_F + GF_SIZE, pb, Rb, m.F + m.oAGF + gb * GEOM_F + GF_SIZE);
          if (dd < best) best = dd;
        }
      }
      y[adr] = best;
    }
  }
}

// task costs: one struct per task
struct LeapCost {
  static constexpr bool needs_sensors = false;
  __device__ static float step(const float* tp, int, const float* qpos, const float*, const float*, float) { return leap_step_cost(tp, qpos); }
  __device__ static float finish(float acc, int H) { return acc / (float)H; }  // mean over time
};
struct Fr3Cost {
  static constexpr bool needs_sensors = true;
  __device__ static float step(const float* tp, int phase, const float* qpos, const float* qvel, const float* y, float decay) { return fr3_step_cost(tp, phase, qpos, qvel, 15, y, decay); }
  __device__ static float finish(float acc, int) { return acc; }  // sum over time
};

// ------------------------------------------------------------------------------------------------ kernels
template <class C, class TC>
__global__ __launch_bounds__(kBlock) void k_engine_cost(const float* __restrict__ gF, const int* __restrict__ gI, int nF, int nI,
                                                        const float* __restrict__ x0, const float* __restrict__ nominal,
                                                        const float* __restrict__ noise, int ldn, const float* __restrict__ sigma,
                                                        const float* __restrict__ W, const float* __restrict__ lohi, const float* __restrict__ tp,
                                                        int ntp, int phase, int N, int n_offset, int H, int K, float* __restrict__ costs,
                                                        float* __restrict__ knots_out, int* __restrict__ overflow) {
  extern __shared__ float lds[];
  float* sF = lds; int* sI = (int*)(sF + nF);
  float* sW = (float*)(sI + nI); float* sKn = sW + H * K; float* sTp = sKn + K * C::NU * kBlock;
  const int lane = threadIdx.x;
  for (int i = lane; i < nF; i += kBlock) sF[i] = gF[i];
  for (int i = lane; i < nI; i += kBlock) sI[i] = gI[i];
  for (int i = lane; i < H * K; i += kBlock) sW[i] = W[i];
  for (int i = lane; i < ntp; i += kBlock) sTp[i] = tp[i];
  const int n = blockIdx.x * kBlock + lane; const bool live = n < N; const int nc = live ? n : N - 1;
  const int KU = K * C::NU;
  for (int i = 0; i < KU; i++) {
    float v = nominal[i];
    if (n_offset + nc != 0) v = fmaf(sigma[i], noise[(size_t)i * ldn + nc], v);
    int u = i % C::NU;
    v = jh_clampf(v, lohi[u], lohi[C::NU + u]);
    sKn[i * kBlock + lane] = v;
    if (knots_out && live) knots_out[(size_t)i * ldn + n] = v;
  }
  __syncthreads();
  EngineModel m; m.init(sF, sI);
  Work<C> w;
  for (int i = 0; i < C::NQ; i++) w.qpos[i] = x0[i];
  for (int i = 0; i < C::NV; i++) { w.qvel[i] = x0[C::NQ + i]; w.qws[i] = 0.f; }
  w.overflow = 0; w.iters = 0; w.maxed = 0;
#ifdef JH_ENGINE_PROFILE
  for (int k = 0; k < 8; k++) w.cyc[k] = 0;
  w.t0 = clock64();
#endif
  float acc = 0.f;
  for (int h = 0; h < H; h++) {
    float u[C::NU];
    for (int j = 0; j < C::NU; j++) u[j] = 0.f;
    for (int k = 0; k < K; k++) { float wk = sW[h * K + k]; for (int j = 0; j < C::NU; j++) u[j] = fmaf(wk, sKn[(k * C::NU + j) * kBlock + lane], u[j]); }
    engine_forward(m, w, u);
    float y[C::NS];
    if (TC::needs_sensors) engine_sensors(m, w, y);
    engine_step(m, w, u);
    JH_TICK(5)
    acc += TC::step(sTp, phase, w.qpos, w.qvel, y, H > 1 ? 1.f - (float)h / (float)(H - 1) : 1.f);
  }
#ifdef JH_ENGINE_PROFILE
  if (lane == 0 && overflow) for (int k = 0; k < 8; k++) atomicAdd((unsigned long long*)(overflow + 4) + k, (unsigned long long)w.cyc[k]);
#endif
  if (live) {
    costs[n] = TC::finish(acc, H);
    if (overflow) { if (w.overflow) atomicAdd(overflow, w.overflow); if (w.maxed) atomicAdd(overflow + 1, w.maxed); atomicAdd(overflow + 2, w.iters); atomicAdd(overflow + 3, H); }
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void k_engine_materialize(const float* __restrict__ gF, const int* __restrict__ gI, int nF, int nI,
                                                               const float* __restrict__ x0, int x0_batched, const float* __restrict__ controls,
                                                               int N, int H, float* __restrict__ states, float* __restrict__ sensors,
                                                               int* __restrict__ overflow) {
  extern __shared__ float lds[];
  float* sF = lds; int* sI = (int*)(sF + nF);
  for (int i = threadIdx.x; i < nF; i += kBlock) sF[i] = gF[i];
  for (int i = threadIdx.x; i < nI; i += kBlock) sI[i] = gI[i];
  __syncthreads();
  const int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= N) return;
  EngineModel m; m.init(sF, sI);
  Work<C> w;
  const float* xi = x0 + (x0_batched ? (size_t)n * C::NX : 0);
  for (int i = 0; i < C::NQ; i++) w.qpos[i] = xi[i];
  for (int i = 0; i < C::NV; i++) { w.qvel[i] = xi[C::NQ + i]; w.qws[i] = 0.f; }
  w.overflow = 0; w.iters = 0; w.maxed = 0;
  for (int h = 0; h < H; h++) {
    float u[C::NU];
    for (int j = 0; j < C::NU; j++) u[j] = controls[((size_t)n * H + h) * C::NU + j];
    engine_forward(m, w, u);
    if (sensors) { float y[C::NS]; engine_sensors(m, w, y); for (int i = 0; i < C::NS; i++) sensors[((size_t)n * H + h) * C::NS + i] = y[i]; }
    engine_step(m, w, u);
    if (states) {
      float* o = states + ((size_t)n * H + h) * C::NX;
      for (int i = 0; i < C::NQ; i++) o[i] = w.qpos[i];
      for (int i = 0; i < C::NV; i++) o[C::NQ + i] = w.qvel[i];
    }
  }
  if (overflow) { if (w.overflow) atomicAdd(overflow, w.overflow); if (w.maxed) atomicAdd(overflow + 1, w.maxed); atomicAdd(overflow + 2, w.iters); atomicAdd(overflow + 3, H); }
}

bool model_matches_leap(const jh_model* m) { return m->kind == JH_TASK_LEAP_CUBE && m->nq == 23 && m->nv == 22 && m->nu == 16 && m->ns == 31 && m->h_i.size() > 14 && m->h_i[0] == 17 && m->h_i[1] == 4 && m->h_i[13] > 0; }
bool model_matches_fr3(const jh_model* m) { return m->kind == JH_TASK_FR3_PICK && m->nq == 16 && m->nv == 15 && m->nu == 8 && m->ns == 14 && m->h_i.size() > 14 && m->h_i[0] == 10 && m->h_i[1] == 1 && m->h_i[13] > 0; }

template <class C, class TC>
int launch_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma, const float* W, const float* lohi,
                const float* tp, int phase, int N, int n_offset, int H, int K, float* costs, float* knots_out, hipStream_t st) {
  size_t lds = 4 * (m->nf + m->ni + (size_t)H * K + (size_t)K * C::NU * kBlock + JH_MAX_TASK_PARAMS);
  JH_REQUIRE(lds <= 64 * 1024, "rollout_cost: LDS staging needs %zu bytes (> 64 KiB)", lds);
  int grid = (N + kBlock - 1) / kBlock;
  hipLaunchKernelGGL((k_engine_cost<C, TC>), dim3(grid), dim3(kBlock), lds, st, m->d_f, m->d_i, (int)m->nf, (int)m->ni, x0, nominal, noise, ldn, sigma, W, lohi,
                     tp, m->ntaskparam, phase, N, n_offset, H, K, costs, knots_out, m->d_stats);
  JH_HIP(hipGetLastError());
  return JH_OK;
}
template <class C>
int launch_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors, hipStream_t st) {
  size_t lds = 4 * (m->nf + m->ni);
  JH_REQUIRE(lds <= 64 * 1024, "rollout_materialize: LDS staging needs %zu bytes (> 64 KiB)", lds);
  int grid = (N + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_engine_materialize<C>, dim3(grid), dim3(kBlock), lds, st, m->d_f, m->d_i, (int)m->nf, (int)m->ni, x0, x0_batched, controls, N, H, states,
                     sensors, m->d_stats);
  JH_HIP(hipGetLastError());
  return JH_OK;
}

}  // namespace

// largest K for which launch_cost's LDS staging (model image, W, the lanes' knots, task constants) fits the 64 KiB it may ask for
int jh_engine_max_knots(const jh_model* m, int H) {
  const long room = 16 * 1024 - (long)(m->nf + m->ni) - JH_MAX_TASK_PARAMS;
  return room <= 0 ? 0 : (int)(room / ((long)H + (long)m->nu * kBlock));
}

int jh_engine_rollout_cost(const jh_model* m, const float* x0, const float* nominal, const float* noise, int ldn, const float* sigma,
                           const float* W, const float* lohi, const float* tp, int phase, int N, int n_offset, int H, int K, float* costs,
                           float* knots_out, hipStream_t st) {
  if (model_matches_leap(m)) return launch_cost<LeapCfg, LeapCost>(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, phase, N, n_offset, H, K, costs, knots_out, st);
  if (model_matches_fr3(m)) {
    JH_REQUIRE(phase >= 0 && phase <= 3, "rollout_cost: fr3_pick phase must be 0..3 (got %d)", phase);
    return launch_cost<Fr3Cfg, Fr3Cost>(m, x0, nominal, noise, ldn, sigma, W, lohi, tp, phase, N, n_offset, H, K, costs, knots_out, st);
  }
  jh_set_error("rollout_cost: the articulated engine is instantiated for leap_cube and fr3_pick");
  return JH_ERR_UNSUPPORTED;
}

int jh_engine_materialize(const jh_model* m, const float* x0, int x0_batched, const float* controls, int N, int H, float* states, float* sensors,
                          hipStream_t st) {
  if (model_matches_leap(m)) return launch_materialize<LeapCfg>(m, x0, x0_batched, controls, N, H, states, sensors, st);
  if (model_matches_fr3(m)) return launch_materialize<Fr3Cfg>(m, x0, x0_batched, controls, N, H, states, sensors, st);
  jh_set_error("rollout_materialize: the articulated engine is instantiated for leap_cube and fr3_pick");
  return JH_ERR_UNSUPPORTED;
}
