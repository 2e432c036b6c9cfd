/*
 * jo_engine.c -- ORACLE physics step (fp64, CPU).  See jo_engine.h for scope and provenance.
 * Test infrastructure only: never linked into, imported by, or called from the product path.
 *
 * Every block cites the MuJoCo pipeline stage it restates (MuJoCo 3.5 documentation, "Computation"
 * chapter) and the reference call site that reaches it:
 *   judo/utils/mj_rollout_backend.py:84 (mujoco.rollout -> mj_step), tasks/base.py:109 (mj_forward).
 */
#include "jo_engine.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15 /* mjMINVAL */
#define MINIMP 0.0001
#define MAXIMP 0.9999
#define MINMU 1e-5

/* ------------------------------------------------------------------ small math */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
static inline void copy3(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void addscl3(double* r, const double* a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
/* row-major rotation matrix: world = R * local; column k of R is local axis k in world coordinates */
static void quat2mat(double* R, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
static inline void rot(double* r, const double* R, const double* v) {
  double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void rotT(double* r, const double* R, const double* v) {
  double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2], z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* ------------------------------------------------------------------ model builder */
jo_model* jo_model_new(double timestep, int integrator, int cone, double impratio, const double* gravity, int contact_enabled) {
  jo_model* m = (jo_model*)calloc(1, sizeof(jo_model));
  m->dt = timestep; m->integrator = integrator; m->cone = cone; m->impratio = impratio; m->contact_enabled = contact_enabled;
  copy3(m->grav, gravity);
  m->nbody = 1; /* world */
  m->body_parent[0] = -1; m->body_quat[0][0] = 1; m->body_iquat[0][0] = 1;
  m->solver_maxiter = 100; m->solver_tol = 1e-10;
  return m;
}
void jo_model_free(jo_model* m) { free(m); }

int jo_add_body(jo_model* m, int parent, const double* pos, const double* quat, double mass, const double* ipos, const double* iquat, const double* inertia) {
  if (m->nbody >= JO_MAXBODY || parent < 0 || parent >= m->nbody) return -1;
  int b = m->nbody++;
  m->body_parent[b] = parent;
  copy3(m->body_pos[b], pos); memcpy(m->body_quat[b], quat, 4 * sizeof(double)); quat_normalize(m->body_quat[b]);
  m->body_mass[b] = mass; copy3(m->body_ipos[b], ipos); memcpy(m->body_iquat[b], iquat, 4 * sizeof(double)); quat_normalize(m->body_iquat[b]);
  copy3(m->body_inertia[b], inertia);
  m->body_jntadr[b] = -1; m->body_jntnum[b] = 0;
  return b;
}

int jo_add_joint(jo_model* m, int body, int type, const double* pos, const double* axis, double damping, double armature, double frictionloss,
                 int limited, const double* range, double margin, int frclimited, const double* frcrange, const double* solref_limit,
                 const double* solimp_limit, const double* solref_fric, const double* solimp_fric) {
  if (m->njnt >= JO_MAXJNT || body <= 0 || body >= m->nbody) return -1;
  int j = m->njnt++;
  int ndof = type == JO_JNT_FREE ? 6 : 1, nqj = type == JO_JNT_FREE ? 7 : 1;
  if (m->nv + ndof > JO_MAXDOF || m->nq + nqj > JO_MAXQ) return -1;
  /* joints of a body must be added consecutively, bodies in depth-first order (MuJoCo ordering) */
  if (m->body_jntnum[body] == 0) m->body_jntadr[body] = j;
  else if (m->body_jntadr[body] + m->body_jntnum[body] != j) return -2;
  m->body_jntnum[body]++;
  m->jnt_type[j] = type; m->jnt_body[j] = body; m->jnt_qposadr[j] = m->nq; m->jnt_dofadr[j] = m->nv;
  copy3(m->jnt_pos[j], pos); copy3(m->jnt_axis[j], axis);
  m->jnt_limited[j] = limited; m->jnt_range[j][0] = range[0]; m->jnt_range[j][1] = range[1]; m->jnt_margin[j] = margin;
  memcpy(m->jnt_solref[j], solref_limit, 2 * sizeof(double)); memcpy(m->jnt_solimp[j], solimp_limit, 5 * sizeof(double));
  for (int k = 0; k < ndof; k++) {
    int d = m->nv + k;
    m->dof_body[d] = body; m->dof_jnt[d] = j;
    m->dof_damping[d] = damping; m->dof_armature[d] = armature; m->dof_frictionloss[d] = frictionloss;
    m->dof_frclimited[d] = frclimited; m->dof_frcrange[d][0] = frcrange[0]; m->dof_frcrange[d][1] = frcrange[1];
    memcpy(m->dof_solref[d], solref_fric, 2 * sizeof(double)); memcpy(m->dof_solimp[d], solimp_fric, 5 * sizeof(double));
  }
  m->nq += nqj; m->nv += ndof;
  return j;
}

int jo_add_geom(jo_model* m, int body, int type, const double* size, const double* pos, const double* quat, const double* friction,
                const double* solref, const double* solimp, double margin, double gap, int condim) {
  if (m->ngeom >= JO_MAXGEOM) return -1;
  int g = m->ngeom++;
  m->geom_type[g] = type; m->geom_body[g] = body; m->geom_condim[g] = condim;
  copy3(m->geom_size[g], size); copy3(m->geom_pos[g], pos); memcpy(m->geom_quat[g], quat, 4 * sizeof(double)); quat_normalize(m->geom_quat[g]);
  copy3(m->geom_friction[g], friction); memcpy(m->geom_solref[g], solref, 2 * sizeof(double)); memcpy(m->geom_solimp[g], solimp, 5 * sizeof(double));
  m->geom_margin[g] = margin; m->geom_gap[g] = gap; m->geom_priority[g] = 0;
  switch (type) { /* bounding-sphere radius (mjModel.geom_rbound) */
    case JO_GEOM_PLANE: m->geom_rbound[g] = 0; break; /* infinite: the bounding-sphere filter skips planes */
    case JO_GEOM_SPHERE: m->geom_rbound[g] = size[0]; break;
    case JO_GEOM_CAPSULE: m->geom_rbound[g] = size[0] + size[1]; break;
    case JO_GEOM_CYLINDER: m->geom_rbound[g] = sqrt(size[0] * size[0] + size[1] * size[1]); break;
    case JO_GEOM_BOX: m->geom_rbound[g] = norm3(size); break;
    default: return -3;
  }
  return g;
}
int jo_set_geom_priority(jo_model* m, int geom, int priority) { if (geom < 0 || geom >= m->ngeom) return -1; m->geom_priority[geom] = priority; return 0; }
int jo_add_pair(jo_model* m, int g1, int g2) {
  if (m->npair >= JO_MAXPAIR || g1 < 0 || g2 < 0 || g1 >= m->ngeom || g2 >= m->ngeom || g1 == g2) return -1;
  if (g1 > g2) { int t = g1; g1 = g2; g2 = t; } /* MuJoCo orders a pair by geom id: geom1 < geom2 */
  m->pair_g1[m->npair] = g1; m->pair_g2[m->npair] = g2;
  return m->npair++;
}
int jo_add_site(jo_model* m, int body, const double* pos) {
  if (m->nsite >= JO_MAXSITE) return -1;
  m->site_body[m->nsite] = body; copy3(m->site_pos[m->nsite], pos);
  return m->nsite++;
}
int jo_add_actuator(jo_model* m, int joint, double kp, double kv, int ctrllimited, const double* ctrlrange, int forcelimited, const double* forcerange) {
  if (m->nact >= JO_MAXACT || joint < 0 || joint >= m->njnt || m->jnt_type[joint] == JO_JNT_FREE) return -1;
  int a = m->nact++;
  m->act_jnt[a] = joint; m->act_kp[a] = kp; m->act_kv[a] = kv;
  m->act_ctrllimited[a] = ctrllimited; m->act_ctrlrange[a][0] = ctrlrange[0]; m->act_ctrlrange[a][1] = ctrlrange[1];
  m->act_forcelimited[a] = forcelimited; m->act_forcerange[a][0] = forcerange[0]; m->act_forcerange[a][1] = forcerange[1];
  return a;
}
int jo_add_sensor(jo_model* m, int type, int obj, int obj2, double cutoff) {
  if (m->nsensor >= JO_MAXSENSOR) return -1;
  int s = m->nsensor++;
  int dim = (type == JO_SENS_JOINTPOS || type == JO_SENS_DISTANCE) ? 1 : (type == JO_SENS_FRAMEQUAT_BODY ? 4 : 3);
  if (m->nsensordata + dim > JO_MAXSENSORDATA) return -1;
  m->sensor_type[s] = type; m->sensor_obj[s] = obj; m->sensor_obj2[s] = obj2; m->sensor_cutoff[s] = cutoff; m->sensor_adr[s] = m->nsensordata;
  m->nsensordata += dim;
  return s;
}
int jo_add_equality_joint(jo_model* m, int j1, int j2, const double* polycoef, const double* solref, const double* solimp) {
  if (m->neq >= JO_MAXEQ) return -1;
  int e = m->neq++;
  m->eq_j1[e] = j1; m->eq_j2[e] = j2;
  memcpy(m->eq_poly[e], polycoef, 5 * sizeof(double)); memcpy(m->eq_solref[e], solref, 2 * sizeof(double)); memcpy(m->eq_solimp[e], solimp, 5 * sizeof(double));
  return e;
}
int jo_model_dims(const jo_model* m, int* out) {
  out[0] = m->nq; out[1] = m->nv; out[2] = m->nact; out[3] = m->nsensordata; out[4] = m->nbody; out[5] = m->ngeom; out[6] = m->npair;
  return 0;
}
void jo_model_get_invweight0(const jo_model* m, double* dofw, double* bodyw) {
  memcpy(dofw, m->dof_invweight0, m->nv * sizeof(double));
  for (int b = 0; b < m->nbody; b++) { bodyw[2 * b] = m->body_invweight0[b][0]; bodyw[2 * b + 1] = m->body_invweight0[b][1]; }
}
void jo_model_get_qpos0(const jo_model* m, double* q) { memcpy(q, m->qpos0, m->nq * sizeof(double)); }
jo_data* jo_data_new(void) { return (jo_data*)calloc(1, sizeof(jo_data)); }
void jo_data_free(jo_data* d) { free(d); }

/* ------------------------------------------------------------------ kinematics (mj_kinematics) */
static void kinematics(const jo_model* m, jo_data* d) {
  for (int j = 0; j < m->njnt; j++) /* quaternions in qpos are normalised before use */
    if (m->jnt_type[j] == JO_JNT_FREE) quat_normalize(d->qpos + m->jnt_qposadr[j] + 3);
  d->xpos[0][0] = d->xpos[0][1] = d->xpos[0][2] = 0; d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat2mat(d->xmat[0], d->xquat[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    double pos[3], quat[4];
    int nj = m->body_jntnum[b], j0 = m->body_jntadr[b];
    if (nj == 1 && m->jnt_type[j0] == JO_JNT_FREE) { /* free body: pose read straight from qpos */
      const double* q = d->qpos + m->jnt_qposadr[j0];
      copy3(pos, q); memcpy(quat, q + 3, 4 * sizeof(double));
      int da = m->jnt_dofadr[j0];
      double R[9]; quat2mat(R, quat);
      for (int k = 0; k < 3; k++) { /* translations: world axes */
        double* S = d->S[da + k]; S[0] = S[1] = S[2] = 0; S[3] = S[4] = S[5] = 0; S[3 + k] = 1;
      }
      for (int k = 0; k < 3; k++) { /* rotations: body-frame axes (qvel[3:6] is angular velocity in the local frame) */
        double* S = d->S[da + 3 + k]; double ax[3] = {R[k], R[3 + k], R[6 + k]};
        copy3(S, ax); cross3(S + 3, pos, ax);
      }
    } else {
      rot(pos, d->xmat[p], m->body_pos[b]); addscl3(pos, d->xpos[p], 1.0);
      quat_mul(quat, d->xquat[p], m->body_quat[b]);
      for (int jj = 0; jj < nj; jj++) {
        int j = j0 + jj, da = m->jnt_dofadr[j];
        double R[9], axis[3], anchor[3];
        quat2mat(R, quat);
        rot(axis, R, m->jnt_axis[j]);
        rot(anchor, R, m->jnt_pos[j]); addscl3(anchor, pos, 1.0);
        double q = d->qpos[m->jnt_qposadr[j]] - m->qpos0[m->jnt_qposadr[j]];
        double* S = d->S[da];
        if (m->jnt_type[j] == JO_JNT_SLIDE) {
          addscl3(pos, axis, q);
          S[0] = S[1] = S[2] = 0; copy3(S + 3, axis);
        } else { /* hinge: rotate about the anchor */
          double ql[4], qn[4], v[3];
          axisangle2quat(ql, m->jnt_axis[j], q);
          quat_mul(qn, quat, ql); memcpy(quat, qn, sizeof(qn)); quat_normalize(quat);
          quat2mat(R, quat);
          rot(v, R, m->jnt_pos[j]);
          for (int k = 0; k < 3; k++) pos[k] = anchor[k] - v[k];
          copy3(S, axis); cross3(S + 3, anchor, axis);
        }
      }
    }
    quat_normalize(quat);
    copy3(d->xpos[b], pos); memcpy(d->xquat[b], quat, sizeof(quat)); quat2mat(d->xmat[b], quat);
    double iq[4]; quat_mul(iq, quat, m->body_iquat[b]); quat2mat(d->ximat[b], iq);
    rot(d->xipos[b], d->xmat[b], m->body_ipos[b]); addscl3(d->xipos[b], pos, 1.0);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_body[g]; double gq[4];
    rot(d->geom_xpos[g], d->xmat[b], m->geom_pos[g]); addscl3(d->geom_xpos[g], d->xpos[b], 1.0);
    quat_mul(gq, d->xquat[b], m->geom_quat[g]); quat2mat(d->geom_xmat[g], gq);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_body[s];
    rot(d->site_xpos[s], d->xmat[b], m->site_pos[s]); addscl3(d->site_xpos[s], d->xpos[b], 1.0);
  }
}

/* ------------------------------------------------------------------ spatial inertia about the world origin */
/* 6x6 symmetric, (angular, linear) ordering:  [[Ic + m(c.c 1 - c c'), m [c]x], [m [c]x', m 1]] */
static void body_spatial_inertia(const jo_model* m, const jo_data* d, int b, double I[6][6]) {
  memset(I, 0, 36 * sizeof(double));
  double mass = m->body_mass[b];
  const double* R = d->ximat[b]; const double* c = d->xipos[b]; const double* di = m->body_inertia[b];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double v = 0;
      for (int k = 0; k < 3; k++) v += R[i * 3 + k] * di[k] * R[j * 3 + k];
      I[i][j] = v + mass * ((i == j ? dot3(c, c) : 0.0) - c[i] * c[j]);
    }
  double cx[3][3] = {{0, -c[2], c[1]}, {c[2], 0, -c[0]}, {-c[1], c[0], 0}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { I[i][3 + j] = mass * cx[i][j]; I[3 + i][j] = mass * cx[j][i]; }
  for (int i = 0; i < 3; i++) I[3 + i][3 + i] = mass;
}
static void mat6_vec(double* r, double I[6][6], const double* v) {
  for (int i = 0; i < 6; i++) { double s = 0; for (int j = 0; j < 6; j++) s += I[i][j] * v[j]; r[i] = s; }
}

/* ------------------------------------------------------------------ CRB mass matrix (mj_crb) + Cholesky (mj_factorM) */
static int cholesky(int n, double A[JO_MAXDOF][JO_MAXDOF], double L[JO_MAXDOF][JO_MAXDOF]) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i][j];
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      if (i == j) { if (s <= 0) return -1; L[i][i] = sqrt(s); }
      else L[i][j] = s / L[j][j];
    }
  return 0;
}
static void chol_solve(int n, double L[JO_MAXDOF][JO_MAXDOF], double* x /* in: rhs, out: solution */) {
  for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i][k] * x[k]; x[i] = s / L[i][i]; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
}

static void crb(const jo_model* m, jo_data* d) {
  static __thread double Ic[JO_MAXBODY][6][6];
  int nv = m->nv;
  for (int b = 0; b < m->nbody; b++) body_spatial_inertia(m, d, b, Ic[b]);
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parent[b];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Ic[p][i][j] += Ic[b][i][j];
  }
  for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) d->M[i][j] = 0;
  for (int i = 0; i < nv; i++) {
    double f[6]; mat6_vec(f, Ic[m->dof_body[i]], d->S[i]);
    for (int j = i; j >= 0; j = m->dof_parent[j]) {
      double v = 0; for (int k = 0; k < 6; k++) v += f[k] * d->S[j][k];
      d->M[i][j] = d->M[j][i] = v;
    }
    d->M[i][i] += m->dof_armature[i];
  }
  cholesky(nv, d->M, d->L);
}

/* ------------------------------------------------------------------ RNE bias forces (mj_rne, flg_acc=0) incl. gravity */
static void crossm(double* r, const double* v, const double* s) { /* motion cross product v x s */
  double a[3], b[3], c[3];
  cross3(a, v, s); cross3(b, v, s + 3); cross3(c, v + 3, s);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void crossf(double* r, const double* v, const double* f) { /* force cross product v x* f */
  double a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
static void rne_bias(const jo_model* m, jo_data* d) {
  static __thread double vel[JO_MAXBODY][6], acc[JO_MAXBODY][6], frc[JO_MAXBODY][6];
  memset(vel[0], 0, sizeof(vel[0])); memset(acc[0], 0, sizeof(acc[0]));
  acc[0][3] = -m->grav[0]; acc[0][4] = -m->grav[1]; acc[0][5] = -m->grav[2]; /* gravity as base acceleration */
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    memcpy(vel[b], vel[p], sizeof(vel[b])); memcpy(acc[b], acc[p], sizeof(acc[b]));
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == JO_JNT_FREE) {
        /* translational axes are world-fixed (zero derivative); rotational axes use the velocity before
         * the rotation is added (sum over the three is identical to using the full velocity) */
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) vel[b][c] += d->S[da + k][c] * d->qvel[da + k];
        double Sd[3][6];
        for (int k = 0; k < 3; k++) crossm(Sd[k], vel[b], d->S[da + 3 + k]);
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) { acc[b][c] += Sd[k][c] * d->qvel[da + 3 + k]; vel[b][c] += d->S[da + 3 + k][c] * d->qvel[da + 3 + k]; }
      } else {
        double Sd[6]; crossm(Sd, vel[b], d->S[da]);
        for (int c = 0; c < 6; c++) { acc[b][c] += Sd[c] * d->qvel[da]; vel[b][c] += d->S[da][c] * d->qvel[da]; }
      }
    }
    double I[6][6], Ia[6], Iv[6], vIv[6];
    body_spatial_inertia(m, d, b, I);
    mat6_vec(Ia, I, acc[b]); mat6_vec(Iv, I, vel[b]); crossf(vIv, vel[b], Iv);
    for (int c = 0; c < 6; c++) frc[b][c] = Ia[c] + vIv[c];
  }
  memset(frc[0], 0, sizeof(frc[0]));
  for (int b = m->nbody - 1; b > 0; b--) {
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, da = m->jnt_dofadr[j], nd = m->jnt_type[j] == JO_JNT_FREE ? 6 : 1;
      for (int k = 0; k < nd; k++) { double s = 0; for (int c = 0; c < 6; c++) s += d->S[da + k][c] * frc[b][c]; d->qfrc_bias[da + k] = s; }
    }
    int p = m->body_parent[b];
    for (int c = 0; c < 6; c++) frc[p][c] += frc[b][c];
  }
}

/* ------------------------------------------------------------------ collision */
typedef struct { double dist, pos[3], n[3], t[3]; int has_t; } rawcon; /* t: preferred first tangent (mju_makeFrame orthogonalises it) */

/* frame = [normal; t1; t2], tangents as mju_makeFrame builds them */
static void make_frame(double* frame) {
  double* x = frame; double* y = frame + 3; double* z = frame + 6;
  double nn = norm3(x); x[0] /= nn; x[1] /= nn; x[2] /= nn;
  if (x[1] < -0.5 || x[1] > 0.5) { y[0] = 0; y[1] = 0; y[2] = 1; } else { y[0] = 0; y[1] = 1; y[2] = 0; }
  double dp = dot3(x, y); addscl3(y, x, -dp);
  nn = norm3(y); y[0] /= nn; y[1] /= nn; y[2] /= nn;
  cross3(z, x, y);
}

static inline void col(double* a, const double* R, int k) { a[0] = R[k]; a[1] = R[3 + k]; a[2] = R[6 + k]; }

/* Box-box: separating-axis test over the 15 candidate axes; face contact = incident face clipped against
 * the reference face (up to 8 points), edge contact = closest points of the two edges (1 point).
 * Contact position is midway between the surfaces, normal points from box 1 to box 2 (MuJoCo contact convention). */
static int collide_box_box(const double* p1, const double* R1, const double* h1, const double* p2, const double* R2, const double* h2, double margin, rawcon* out) {
  double A[3][3], B[3][3], dv[3], C[3][3], AC[3][3], dA[3], dB[3];
  for (int k = 0; k < 3; k++) { col(A[k], R1, k); col(B[k], R2, k); dv[k] = p2[k] - p1[k]; }
  for (int i = 0; i < 3; i++) { dA[i] = dot3(dv, A[i]); dB[i] = dot3(dv, B[i]); for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], B[j]); AC[i][j] = fabs(C[i][j]); } }
  double best = -1e30; int btype = -1, bi = 0, bj = 0;
  for (int i = 0; i < 3; i++) {
    double s = fabs(dA[i]) - (h1[i] + h2[0] * AC[i][0] + h2[1] * AC[i][1] + h2[2] * AC[i][2]);
    if (s > margin) return 0;
    if (s > best) { best = s; btype = 0; bi = i; }
  }
  for (int j = 0; j < 3; j++) {
    double s = fabs(dB[j]) - (h2[j] + h1[0] * AC[0][j] + h1[1] * AC[1][j] + h1[2] * AC[2][j]);
    if (s > margin) return 0;
    if (s > best) { best = s; btype = 1; bj = j; }
  }
  double ebest = -1e30, eL[3] = {0, 0, 0}; int ei = -1, ej = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double L[3]; cross3(L, A[i], B[j]);
      double l = norm3(L);
      if (l < 1e-6) continue;
      L[0] /= l; L[1] /= l; L[2] /= l;
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += h1[k] * fabs(dot3(A[k], L)); rb += h2[k] * fabs(dot3(B[k], L)); }
      double s = fabs(dot3(dv, L)) - (ra + rb);
      if (s > margin) return 0;
      if (s > ebest) { ebest = s; ei = i; ej = j; copy3(eL, L); }
    }
  /* an edge axis wins only if it is clearly less penetrating than the best face axis */
  int use_edge = ei >= 0 && (best < 0 ? ebest > best / 1.05 + 1e-12 : ebest > best * 1.05 + 1e-12);
  if (use_edge) {
    double n[3]; copy3(n, eL);
    if (dot3(n, dv) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    double pa[3], pb[3]; copy3(pa, p1); copy3(pb, p2);
    for (int k = 0; k < 3; k++) {
      if (k != ei) addscl3(pa, A[k], (dot3(n, A[k]) > 0 ? 1.0 : -1.0) * h1[k]);
      if (k != ej) addscl3(pb, B[k], (dot3(n, B[k]) > 0 ? -1.0 : 1.0) * h2[k]);
    }
    /* closest points of lines pa + s A[ei], pb + t B[ej] */
    double w[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    double b = C[ei][ej], dd = dot3(A[ei], w), e = dot3(B[ej], w), den = 1 - b * b;
    double s = den > 1e-12 ? (b * e - dd) / den : 0.0, t = den > 1e-12 ? (e - b * dd) / den : 0.0;
    if (s > h1[ei]) s = h1[ei]; if (s < -h1[ei]) s = -h1[ei];
    if (t > h2[ej]) t = h2[ej]; if (t < -h2[ej]) t = -h2[ej];
    double ca[3], cb[3]; copy3(ca, pa); addscl3(ca, A[ei], s); copy3(cb, pb); addscl3(cb, B[ej], t);
    out[0].dist = ebest; copy3(out[0].n, n);
    for (int k = 0; k < 3; k++) out[0].pos[k] = 0.5 * (ca[k] + cb[k]);
    return 1;
  }
  /* face contact */
  const double *pr, *pi, *hr, *hi; double (*Ar)[3], (*Ai)[3]; int ri; double n[3];
  if (btype == 0) { pr = p1; pi = p2; hr = h1; hi = h2; Ar = A; Ai = B; ri = bi; double sg = dA[bi] >= 0 ? 1.0 : -1.0; for (int k = 0; k < 3; k++) n[k] = sg * A[bi][k]; }
  else { pr = p2; pi = p1; hr = h2; hi = h1; Ar = B; Ai = A; ri = bj; double sg = dB[bj] >= 0 ? -1.0 : 1.0; for (int k = 0; k < 3; k++) n[k] = sg * B[bj][k]; }
  /* n points from the reference box towards the incident box */
  int mi = 0; double mb = -1;
  for (int k = 0; k < 3; k++) { double v = fabs(dot3(n, Ai[k])); if (v > mb) { mb = v; mi = k; } }
  double sgi = dot3(n, Ai[mi]) > 0 ? -1.0 : 1.0;
  int u = (mi + 1) % 3, v = (mi + 2) % 3;
  double poly[16][3], tmp[16][3]; int np = 4;
  static const double su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
  for (int q = 0; q < 4; q++)
    for (int k = 0; k < 3; k++) poly[q][k] = pi[k] + sgi * hi[mi] * Ai[mi][k] + su[q] * hi[u] * Ai[u][k] + sv[q] * hi[v] * Ai[v][k];
  int ra = (ri + 1) % 3, rb = (ri + 2) % 3;
  for (int pl = 0; pl < 4 && np > 0; pl++) { /* Sutherland-Hodgman against the four side planes of the reference face */
    const double* ax = Ar[pl < 2 ? ra : rb]; double sg = (pl & 1) ? -1.0 : 1.0, lim = hr[pl < 2 ? ra : rb];
    int nn = 0;
    for (int q = 0; q < np; q++) {
      const double* P = poly[q]; const double* Q = poly[(q + 1) % np];
      double dp_[3] = {P[0] - pr[0], P[1] - pr[1], P[2] - pr[2]}, dq_[3] = {Q[0] - pr[0], Q[1] - pr[1], Q[2] - pr[2]};
      double fp = sg * dot3(dp_, ax) - lim, fq = sg * dot3(dq_, ax) - lim;
      if (fp <= 0) { copy3(tmp[nn], P); nn++; }
      if ((fp < 0 && fq > 0) || (fp > 0 && fq < 0)) { double t = fp / (fp - fq); for (int k = 0; k < 3; k++) tmp[nn][k] = P[k] + t * (Q[k] - P[k]); nn++; }
    }
    np = nn; memcpy(poly, tmp, sizeof(double) * 3 * np);
  }
  int nc = 0;
  for (int q = 0; q < np && nc < 8; q++) {
    double dx[3] = {poly[q][0] - pr[0], poly[q][1] - pr[1], poly[q][2] - pr[2]};
    double depth = hr[ri] - dot3(dx, n);
    if (-depth >= margin) continue; /* keep dist < margin */
    out[nc].dist = -depth;
    for (int k = 0; k < 3; k++) { out[nc].pos[k] = poly[q][k] + 0.5 * depth * n[k]; out[nc].n[k] = btype == 0 ? n[k] : -n[k]; }
    nc++;
  }
  return nc;
}

/* sphere (centre c, radius r) against box; normal returned points from the box to the sphere */
static int collide_box_sphere(const double* pb, const double* Rb, const double* hb, const double* c, double r, double margin, rawcon* out) {
  double dl[3] = {c[0] - pb[0], c[1] - pb[1], c[2] - pb[2]}, cl[3], q[3]; int outside = 0;
  rotT(cl, Rb, dl);
  for (int k = 0; k < 3; k++) { q[k] = cl[k]; if (q[k] > hb[k]) { q[k] = hb[k]; outside = 1; } else if (q[k] < -hb[k]) { q[k] = -hb[k]; outside = 1; } }
  double nl[3], dist;
  if (outside) {
    double df[3] = {cl[0] - q[0], cl[1] - q[1], cl[2] - q[2]}; double l = norm3(df);
    if (l - r >= margin) return 0;
    nl[0] = df[0] / l; nl[1] = df[1] / l; nl[2] = df[2] / l; dist = l - r;
  } else { /* centre inside the box: exit through the nearest face */
    int kb = 0; double mn = 1e30;
    for (int k = 0; k < 3; k++) { double s = hb[k] - fabs(cl[k]); if (s < mn) { mn = s; kb = k; } }
    nl[0] = nl[1] = nl[2] = 0; nl[kb] = cl[kb] >= 0 ? 1.0 : -1.0;
    q[kb] = nl[kb] * hb[kb]; dist = -mn - r;
  }
  double ql[3] = {q[0] + 0.5 * dist * nl[0], q[1] + 0.5 * dist * nl[1], q[2] + 0.5 * dist * nl[2]};
  rot(out->pos, Rb, ql); addscl3(out->pos, pb, 1.0); rot(out->n, Rb, nl); out->dist = dist;
  return 1;
}

/* two spheres (MuJoCo's mjc_SphereSphere: normal along the centre line from the first to the second, position midway through the overlap) */
static int collide_sphere_sphere(const double* c1, double r1, const double* c2, double r2, double margin, rawcon* out) {
  double d[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]}; double l = norm3(d), dist = l - r1 - r2;
  if (dist > margin) return 0;                                            /* (mjraw_SphereSphere keeps dist == margin) */
  if (l < 1e-15) { d[0] = 1; d[1] = d[2] = 0; l = 1; }                   /* coincident centres: mju_normalize3 returns (1, 0, 0) below mjMINVAL */
  for (int k = 0; k < 3; k++) { out->n[k] = d[k] / l; out->pos[k] = c1[k] + (r1 + 0.5 * dist) * out->n[k]; }
  out->dist = dist;
  return 1;
}

/* capsule against capsule (MuJoCo's primitive mjc_CapsuleCapsule, `engine_collision_primitive.c`; the pairs of the Spot robot against itself,
 * `judo/models/xml/spot_primitive/contact.xml:4-14` lists the 11 it excludes): the closest points of the two axis SEGMENTS, then sphere against sphere there.
 * Non-parallel axes: the unconstrained minimiser of |c1 + x1 a1 - c2 - x2 a2|^2 (a_i = half length times axis, x_i in [-1, 1]), x1 clamped with x2 re-solved, then x2
 * clamped with x1 re-solved.  Parallel axes (determinant below mjMINVAL): each end of capsule 1 against the nearest point of segment 2, then each end of capsule 2
 * against segment 1, at most two contacts -- a capsule lying along another rests on two points, not one.  size = (radius, half length), axis = local z. */
static int collide_capsule_capsule(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2, double margin, rawcon* out) {
  double a1[3], a2[3]; col(a1, R1, 2); col(a2, R2, 2);
  for (int k = 0; k < 3; k++) { a1[k] *= s1[1]; a2[k] *= s2[1]; }
  const double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
  const double det = ma * mc - mb * mb;
  if (fabs(det) >= MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > 1) { x1 = 1; x2 = (v - mb) / mc; } else if (x1 < -1) { x1 = -1; x2 = (v + mb) / mc; }
    if (x2 > 1) { x2 = 1; x1 = (u - mb) / ma; if (x1 > 1) x1 = 1; else if (x1 < -1) x1 = -1; }
    else if (x2 < -1) { x2 = -1; x1 = (u + mb) / ma; if (x1 > 1) x1 = 1; else if (x1 < -1) x1 = -1; }
    double v1[3], v2[3];
    for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k] * x1; v2[k] = p2[k] + a2[k] * x2; }
    return collide_sphere_sphere(v1, s1[0], v2, s2[0], margin, out);
  }
  int n = 0;
  for (int e = 0; e < 2 && n < 2; e++) {  /* the two ends of capsule 1 against segment 2 */
    const double x1 = e == 0 ? 1.0 : -1.0;
    double x2 = (v - x1 * mb) / mc; if (x2 > 1) x2 = 1; else if (x2 < -1) x2 = -1;
    double v1[3], v2[3];
    for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k] * x1; v2[k] = p2[k] + a2[k] * x2; }
    n += collide_sphere_sphere(v1, s1[0], v2, s2[0], margin, out + n);
  }
  for (int e = 0; e < 2 && n < 2; e++) {  /* the two ends of capsule 2 against segment 1 */
    const double x2 = e == 0 ? 1.0 : -1.0;
    double x1 = (u - x2 * mb) / ma; if (x1 > 1) x1 = 1; else if (x1 < -1) x1 = -1;
    double v1[3], v2[3];
    for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k] * x1; v2[k] = p2[k] + a2[k] * x2; }
    n += collide_sphere_sphere(v1, s1[0], v2, s2[0], margin, out + n);
  }
  return n;
}

/* sphere against capsule (mjc_SphereCapsule): the point of the capsule's axis segment nearest to the centre, then sphere against sphere; normal from the sphere to the capsule */
static int collide_sphere_capsule(const double* ps, double rs, const double* pc, const double* Rc, const double* sc, double margin, rawcon* out) {
  double a[3]; col(a, Rc, 2);
  const double d[3] = {ps[0] - pc[0], ps[1] - pc[1], ps[2] - pc[2]};
  double x = dot3(a, d); if (x > sc[1]) x = sc[1]; else if (x < -sc[1]) x = -sc[1];
  const double v[3] = {pc[0] + a[0] * x, pc[1] + a[1] * x, pc[2] + a[2] * x};
  return collide_sphere_sphere(ps, rs, v, sc[0], margin, out);
}

/* two cylinders with parallel axes whose heights overlap: radial contact (the only cylinder case in the four models,
 * cylinder_push.xml:23,30; MuJoCo itself routes cylinder-cylinder through its general convex collider) */
static int collide_cyl_cyl_parallel(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2, double margin, rawcon* out) {
  double a1[3], a2[3]; col(a1, R1, 2); col(a2, R2, 2);
  if (fabs(dot3(a1, a2)) < 1 - 1e-9) return 0;
  double dv[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double along = dot3(dv, a1);
  if (fabs(along) >= s1[1] + s2[1]) return 0;
  double rad[3] = {dv[0] - along * a1[0], dv[1] - along * a1[1], dv[2] - along * a1[2]};
  double l = norm3(rad);
  if (l < 1e-12) return 0;
  double dist = l - (s1[0] + s2[0]);
  if (dist >= margin) return 0;
  for (int k = 0; k < 3; k++) { out->n[k] = rad[k] / l; }
  for (int k = 0; k < 3; k++) out->pos[k] = p1[k] + out->n[k] * (s1[0] + 0.5 * dist) + 0.5 * along * a1[k];
  out->dist = dist;
  return 1;
}

/* Signed distance between two boxes as the largest separation over the 15 SAT axes: exact whenever the closest
 * features involve a face or an edge pair (all finger-pad / cube / table configurations of fr3_pick), a lower bound for
 * vertex-vertex and vertex-edge pairs; equals -penetration depth when the boxes overlap (what MuJoCo's geom-distance
 * sensor reports, `fr3_components/params_and_default.xml:58-68`). */
static double box_box_distance(const double* p1, const double* R1, const double* h1, const double* p2, const double* R2, const double* h2) {
  double A[3][3], B[3][3], dv[3], best = -1e30;
  for (int k = 0; k < 3; k++) { col(A[k], R1, k); col(B[k], R2, k); dv[k] = p2[k] - p1[k]; }
  for (int i = 0; i < 3; i++) {
    double ra = h1[i], rb = 0; for (int k = 0; k < 3; k++) rb += h2[k] * fabs(dot3(B[k], A[i]));
    double s = fabs(dot3(dv, A[i])) - ra - rb; if (s > best) best = s;
    ra = 0; rb = h2[i]; for (int k = 0; k < 3; k++) ra += h1[k] * fabs(dot3(A[k], B[i]));
    s = fabs(dot3(dv, B[i])) - ra - rb; if (s > best) best = s;
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double L[3]; cross3(L, A[i], B[j]); double l = norm3(L);
      if (l < 1e-6) continue;
      L[0] /= l; L[1] /= l; L[2] /= l;
      double ra = 0, rb = 0; for (int k = 0; k < 3; k++) { ra += h1[k] * fabs(dot3(A[k], L)); rb += h2[k] * fabs(dot3(B[k], L)); }
      double s = fabs(dot3(dv, L)) - ra - rb; if (s > best) best = s;
    }
  return best;
}

/* Box against capsule (radius size[0], half length size[1] along its local z).  The fr3 links' collision meshes are not in the reference repository, so no
 * routine of MuJoCo's could be restated for them anyway (its mjc_CapsuleBox has no closed form either); this is the build's own definition, checked against
 * support-function geometry in tests/test_oracle_independent.py.  The capsule is its axis segment swept by a sphere:
 *   1. both end spheres are tested like box-sphere (a capsule lying flat on a face rests on its two ends);
 *   2. the point of the segment closest to the box -- d(t)^2 = sum_k max(|c_k + t a_k| - h_k, 0)^2 is convex in t, its derivative monotone: bisection -- is
 *      tested the same way when it lies strictly inside the segment (a capsule across an edge of the box touches with its cylinder, not its ends).
 * Up to three contacts, normal from the box to the capsule. */
static int collide_box_capsule(const double* pb, const double* Rb, const double* hb, const double* pc, const double* Rc, const double* size, double margin, rawcon* out) {
  double axis[3]; col(axis, Rc, 2);
  const double r = size[0], L = size[1];
  int n = 0;
  for (int sgn = 1; sgn >= -1; sgn -= 2) {
    double c[3] = {pc[0] + sgn * L * axis[0], pc[1] + sgn * L * axis[1], pc[2] + sgn * L * axis[2]};
    n += collide_box_sphere(pb, Rb, hb, c, r, margin, out + n);
  }
  /* closest point of the segment, in the box frame */
  double d0[3] = {pc[0] - pb[0], pc[1] - pb[1], pc[2] - pb[2]}, c[3], a[3];
  for (int k = 0; k < 3; k++) { c[k] = Rb[k] * d0[0] + Rb[3 + k] * d0[1] + Rb[6 + k] * d0[2]; a[k] = Rb[k] * axis[0] + Rb[3 + k] * axis[1] + Rb[6 + k] * axis[2]; }
  double lo = -L, hi = L, glo = 0, ghi = 0;
  for (int k = 0; k < 3; k++) {
    double sl = c[k] + lo * a[k], sh = c[k] + hi * a[k];
    glo += (sl > hb[k] ? sl - hb[k] : (sl < -hb[k] ? sl + hb[k] : 0)) * a[k];
    ghi += (sh > hb[k] ? sh - hb[k] : (sh < -hb[k] ? sh + hb[k] : 0)) * a[k];
  }
  if (glo >= 0 || ghi <= 0) return n; /* the minimum sits at an end (or the whole segment is equally far / inside): the end spheres have it */
  for (int it = 0; it < 60; it++) {
    double t = 0.5 * (lo + hi), g = 0;
    for (int k = 0; k < 3; k++) { double sk = c[k] + t * a[k]; g += (sk > hb[k] ? sk - hb[k] : (sk < -hb[k] ? sk + hb[k] : 0)) * a[k]; }
    if (g < 0) lo = t; else hi = t;
  }
  double t = 0.5 * (lo + hi);
  if (fabs(t) >= L * (1 - 1e-9)) return n;
  double cm[3] = {pc[0] + t * axis[0], pc[1] + t * axis[1], pc[2] + t * axis[2]};
  n += collide_box_sphere(pb, Rb, hb, cm, r, margin, out + n);
  return n;
}

/* Plane (geom 1) against sphere / capsule / box (engine_collision_primitive.c: mjc_PlaneSphere, mjc_PlaneCapsule, mjc_PlaneBox).
 * The plane's normal is its local z axis; contact normal = plane normal (from geom 1 to geom 2), position midway between the surfaces. */
static int collide_plane_sphere(const double* pp, const double* Rp, const double* c, double r, double margin, rawcon* out) {
  double n[3]; col(n, Rp, 2);
  double dif[3] = {c[0] - pp[0], c[1] - pp[1], c[2] - pp[2]};
  double dist = dot3(dif, n) - r;
  if (dist > margin) return 0;
  out->dist = dist; copy3(out->n, n); out->has_t = 0;
  for (int k = 0; k < 3; k++) out->pos[k] = c[k] - n[k] * (r + 0.5 * dist);
  return 1;
}
static int collide_plane_capsule(const double* pp, const double* Rp, const double* pc, const double* Rc, const double* size, double margin, rawcon* out) {
  double axis[3]; col(axis, Rc, 2);
  int n = 0;
  for (int sgn = 1; sgn >= -1; sgn -= 2) { /* the two end spheres, + end first */
    double c[3] = {pc[0] + sgn * size[1] * axis[0], pc[1] + sgn * size[1] * axis[1], pc[2] + sgn * size[1] * axis[2]};
    int k = collide_plane_sphere(pp, Rp, c, size[0], margin, out + n);
    if (k) { copy3(out[n].t, axis); out[n].has_t = 1; n++; } /* the second frame axis follows the capsule axis */
  }
  return n;
}
static int collide_plane_box(const double* pp, const double* Rp, const double* pb, const double* Rb, const double* h, double margin, rawcon* out) {
  double n[3]; col(n, Rp, 2);
  double dif[3] = {pb[0] - pp[0], pb[1] - pp[1], pb[2] - pp[2]};
  double dist = dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8 && cnt < 4; i++) { /* corners in MuJoCo's order: bit 0 -> x, bit 1 -> y, bit 2 -> z; at most 4 contacts */
    double vl[3] = {(i & 1 ? h[0] : -h[0]), (i & 2 ? h[1] : -h[1]), (i & 4 ? h[2] : -h[2])}, vec[3];
    rot(vec, Rb, vl);
    double ldist = dot3(n, vec);
    if (dist + ldist > margin) continue;
    out[cnt].dist = dist + ldist; copy3(out[cnt].n, n); out[cnt].has_t = 0;
    for (int k = 0; k < 3; k++) out[cnt].pos[k] = pb[k] + vec[k] - n[k] * (0.5 * out[cnt].dist);
    cnt++;
  }
  return cnt;
}

/* ---- general convex pair (MuJoCo's mjc_Convex: every pair without a primitive routine -- box-cylinder, cylinder-cylinder, capsule-cylinder -- goes through its
 * convex collider, since 3.2 the native GJK + EPA pair, `engine_collision_convex.c` / `engine_collision_gjk.c` of the pinned 3.5.0; neither file is in the reference
 * repository).  Restated from the published algorithms: GJK on the Minkowski difference A - B decides overlap and leaves a tetrahedron around the origin, EPA expands it
 * until the face closest to the origin is a supporting plane of A - B within `CCD_TOL` -- that face's normal is the direction of least penetration, its distance the depth.
 * One contact (multiccd is off in every shipped model): normal from A to B, position midway between the two witness points.  Separated shapes give no contact (the
 * shipped geoms carry margin 0).  Support mappings: box, sphere, capsule, cylinder (size = radius, half length along local z). */
#define CCD_TOL 1e-10
#define CCD_MAXV 96
#define CCD_MAXF 192
typedef struct { int type; const double *size, *pos, *R; } cshape;
typedef struct { double w[3], a[3], b[3]; } cvert; /* w = a - b */
static void shape_support(const cshape* s, const double* d, double* out) {
  copy3(out, s->pos);
  double ax[3];
  switch (s->type) {
    case JO_GEOM_SPHERE: { double l = norm3(d); if (l > 0) addscl3(out, d, s->size[0] / l); } break;
    case JO_GEOM_BOX: for (int k = 0; k < 3; k++) { col(ax, s->R, k); addscl3(out, ax, dot3(d, ax) >= 0 ? s->size[k] : -s->size[k]); } break;
    case JO_GEOM_CAPSULE: { col(ax, s->R, 2); addscl3(out, ax, dot3(d, ax) >= 0 ? s->size[1] : -s->size[1]); double l = norm3(d); if (l > 0) addscl3(out, d, s->size[0] / l); } break;
    case JO_GEOM_CYLINDER: {
      col(ax, s->R, 2); double da = dot3(d, ax);
      addscl3(out, ax, da >= 0 ? s->size[1] : -s->size[1]);
      double r[3] = {d[0] - da * ax[0], d[1] - da * ax[1], d[2] - da * ax[2]}; double l = norm3(r);
      if (l > 1e-14 * (fabs(da) + l)) addscl3(out, r, s->size[0] / l);
    } break;
    default: break;
  }
}
static void ccd_support(const cshape* A, const cshape* B, const double* d, cvert* v) {
  double nd[3] = {-d[0], -d[1], -d[2]};
  shape_support(A, d, v->a); shape_support(B, nd, v->b);
  for (int k = 0; k < 3; k++) v->w[k] = v->a[k] - v->b[k];
}
/* GJK, overlap only: returns 1 with four vertices around the origin in sx[0..3], 0 when a separating direction was found */
static int ccd_gjk(const cshape* A, const cshape* B, cvert* sx) {
  double d[3] = {B->pos[0] - A->pos[0], B->pos[1] - A->pos[1], B->pos[2] - A->pos[2]};
  if (norm3(d) < 1e-12) { d[0] = 1; d[1] = 0; d[2] = 0; }
  int n = 0;
  ccd_support(A, B, d, &sx[n++]);
  d[0] = -sx[0].w[0]; d[1] = -sx[0].w[1]; d[2] = -sx[0].w[2];
  for (int it = 0; it < 128; it++) {
    if (norm3(d) < 1e-14) { /* the origin lies on the current simplex: touching or degenerate -- take any direction that grows the simplex */
      double e[3] = {1, 0, 0}; if (n >= 2) { double ab[3] = {sx[1].w[0] - sx[0].w[0], sx[1].w[1] - sx[0].w[1], sx[1].w[2] - sx[0].w[2]}; double t[3] = {0, 1, 0}; cross3(e, ab, t); if (norm3(e) < 1e-12) { t[0] = 0; t[1] = 0; t[2] = 1; cross3(e, ab, t); } }
      copy3(d, e);
    }
    cvert nv; ccd_support(A, B, d, &nv);
    if (dot3(nv.w, d) < 0) return 0; /* the new support point did not pass the origin: separated */
    /* newest vertex first */
    for (int i = n; i > 0; i--) sx[i] = sx[i - 1];
    sx[0] = nv; n++;
    const double *a = sx[0].w, *b = sx[1].w;
    double ao[3] = {-a[0], -a[1], -a[2]}, ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    if (n == 2) {
      if (dot3(ab, ao) > 0) { double t[3]; cross3(t, ab, ao); cross3(d, t, ab); } else { n = 1; copy3(d, ao); }
    } else if (n == 3) {
      const double* c = sx[2].w; double ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, abc[3], t[3];
      cross3(abc, ab, ac);
      cross3(t, abc, ac);
      if (dot3(t, ao) > 0) {
        if (dot3(ac, ao) > 0) { sx[1] = sx[2]; n = 2; double u[3]; cross3(u, ac, ao); cross3(d, u, ac); }
        else if (dot3(ab, ao) > 0) { n = 2; double u[3]; cross3(u, ab, ao); cross3(d, u, ab); }
        else { n = 1; copy3(d, ao); }
      } else {
        cross3(t, ab, abc);
        if (dot3(t, ao) > 0) {
          if (dot3(ab, ao) > 0) { n = 2; double u[3]; cross3(u, ab, ao); cross3(d, u, ab); } else { n = 1; copy3(d, ao); }
        } else if (dot3(abc, ao) > 0) copy3(d, abc);
        else { cvert tmp = sx[1]; sx[1] = sx[2]; sx[2] = tmp; d[0] = -abc[0]; d[1] = -abc[1]; d[2] = -abc[2]; }
      }
    } else { /* tetrahedron a (new), b, c, dd: which face, if any, sees the origin */
      const double *c = sx[2].w, *dd = sx[3].w;
      double ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, ad[3] = {dd[0] - a[0], dd[1] - a[1], dd[2] - a[2]}, abc[3], acd[3], adb[3];
      cross3(abc, ab, ac); cross3(acd, ac, ad); cross3(adb, ad, ab);
      /* orient the three faces through `a` away from the opposite vertex */
      if (dot3(abc, ad) > 0) { abc[0] = -abc[0]; abc[1] = -abc[1]; abc[2] = -abc[2]; }
      if (dot3(acd, ab) > 0) { acd[0] = -acd[0]; acd[1] = -acd[1]; acd[2] = -acd[2]; }
      if (dot3(adb, ac) > 0) { adb[0] = -adb[0]; adb[1] = -adb[1]; adb[2] = -adb[2]; }
      if (dot3(abc, ao) > 0) { n = 3; copy3(d, abc); }                                   /* keep a, b, c */
      else if (dot3(acd, ao) > 0) { sx[1] = sx[2]; sx[2] = sx[3]; n = 3; copy3(d, acd); } /* keep a, c, d */
      else if (dot3(adb, ao) > 0) { sx[2] = sx[1]; sx[1] = sx[3]; n = 3; copy3(d, adb); } /* keep a, d, b */
      else return 1;
      /* (the face case continues with a plain direction: the triangle's own region tests run on the next vertex) */
    }
  }
  return 0;
}
typedef struct { int v[3]; double n[3], dist; int alive; } cface;
static int ccd_make_face(const cvert* V, int i, int j, int k, cface* f) {
  double e1[3], e2[3];
  for (int c = 0; c < 3; c++) { e1[c] = V[j].w[c] - V[i].w[c]; e2[c] = V[k].w[c] - V[i].w[c]; }
  cross3(f->n, e1, e2);
  double l = norm3(f->n);
  if (l < 1e-30) return 0;
  for (int c = 0; c < 3; c++) f->n[c] /= l;
  f->dist = dot3(f->n, V[i].w);
  f->v[0] = i; f->v[1] = j; f->v[2] = k; f->alive = 1;
  if (f->dist < 0) { f->dist = -f->dist; for (int c = 0; c < 3; c++) f->n[c] = -f->n[c]; f->v[1] = k; f->v[2] = j; }
  return 1;
}
static int collide_convex(int t1, const double* s1, const double* p1, const double* R1, int t2, const double* s2, const double* p2, const double* R2, double margin, rawcon* out) {
  (void)margin;
  cshape A = {t1, s1, p1, R1}, B = {t2, s2, p2, R2};
  cvert V[CCD_MAXV]; cface F[CCD_MAXF]; int nv = 0, nf = 0;
  if (!ccd_gjk(&A, &B, V)) return 0;
  nv = 4;
  { /* outward-oriented tetrahedron */
    static const int idx[4][3] = {{0, 1, 2}, {0, 2, 3}, {0, 3, 1}, {1, 3, 2}};
    for (int f = 0; f < 4; f++) {
      if (!ccd_make_face(V, idx[f][0], idx[f][1], idx[f][2], &F[nf])) return 0; /* flat simplex: the shapes touch without volume */
      /* the opposite vertex must lie behind the face */
      int opp = 6 - idx[f][0] - idx[f][1] - idx[f][2];
      if (dot3(F[nf].n, V[opp].w) - F[nf].dist > 1e-12) { for (int c = 0; c < 3; c++) F[nf].n[c] = -F[nf].n[c]; F[nf].dist = -F[nf].dist; int t = F[nf].v[1]; F[nf].v[1] = F[nf].v[2]; F[nf].v[2] = t; }
      nf++;
    }
  }
  int best = -1;
  for (int it = 0; it < 4 * CCD_MAXV; it++) {
    best = -1;
    for (int f = 0; f < nf; f++) if (F[f].alive && (best < 0 || F[f].dist < F[best].dist)) best = f;
    if (best < 0) return 0;
    cvert nw; ccd_support(&A, &B, F[best].n, &nw);
    double grow = dot3(nw.w, F[best].n) - F[best].dist;
    if (grow < CCD_TOL || nv >= CCD_MAXV || nf + 2 * 32 >= CCD_MAXF) break;
    /* remove the faces the new point sees, collect the horizon */
    int edges[256][2], ne = 0;
    for (int f = 0; f < nf; f++) {
      if (!F[f].alive || dot3(F[f].n, nw.w) - F[f].dist <= 1e-14) continue;
      F[f].alive = 0;
      for (int e = 0; e < 3; e++) {
        int a = F[f].v[e], b = F[f].v[(e + 1) % 3], found = -1;
        for (int q = 0; q < ne; q++) if (edges[q][0] == b && edges[q][1] == a) { found = q; break; }
        if (found >= 0) { edges[found][0] = edges[ne - 1][0]; edges[found][1] = edges[ne - 1][1]; ne--; }
        else if (ne < 256) { edges[ne][0] = a; edges[ne][1] = b; ne++; }
      }
    }
    if (ne == 0) break; /* numerically on the surface already */
    V[nv] = nw;
    /* compact the face list */
    int k = 0; for (int f = 0; f < nf; f++) if (F[f].alive) F[k++] = F[f]; nf = k;
    for (int e = 0; e < ne && nf < CCD_MAXF; e++) {
      cface nfc;
      if (!ccd_make_face(V, edges[e][0], edges[e][1], nv, &nfc)) continue;
      /* keep the winding of the horizon edge: the normal must point away from the interior (origin side) */
      F[nf++] = nfc;
    }
    nv++;
  }
  if (best < 0) return 0;
  { /* witness points: barycentric coordinates of the origin's projection onto the closest face */
    const cface* f = &F[best];
    const cvert *a = &V[f->v[0]], *b = &V[f->v[1]], *c = &V[f->v[2]];
    double pr[3] = {f->n[0] * f->dist, f->n[1] * f->dist, f->n[2] * f->dist};
    double v0[3], v1[3], v2[3];
    for (int k = 0; k < 3; k++) { v0[k] = b->w[k] - a->w[k]; v1[k] = c->w[k] - a->w[k]; v2[k] = pr[k] - a->w[k]; }
    double d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1), d20 = dot3(v2, v0), d21 = dot3(v2, v1), den = d00 * d11 - d01 * d01;
    double bv = den > 1e-300 ? (d11 * d20 - d01 * d21) / den : 0, bw = den > 1e-300 ? (d00 * d21 - d01 * d20) / den : 0, bu = 1 - bv - bw;
    double wa[3], wb[3];
    for (int k = 0; k < 3; k++) { wa[k] = bu * a->a[k] + bv * b->a[k] + bw * c->a[k]; wb[k] = bu * a->b[k] + bv * b->b[k] + bw * c->b[k]; }
    /* A - B reaches `dist` along n: B has to move by dist along n to separate, i.e. the normal from A to B is n */
    out->dist = -f->dist;
    for (int k = 0; k < 3; k++) { out->n[k] = f->n[k]; out->pos[k] = 0.5 * (wa[k] + wb[k]); }
    out->has_t = 0;
  }
  return 1;
}

/* sphere against cylinder (MuJoCo's primitive mjc_SphereCylinder, `engine_collision_primitive.c`: side, cap or rim, whichever the sphere's centre is nearest to):
 * the closest point of the solid cylinder to the centre, the normal from the cylinder to the sphere along that line, position midway through the overlap.  A centre
 * inside the cylinder leaves through the nearer of side and cap. */
static int collide_cylinder_sphere(const double* pc, const double* Rc, const double* sc, const double* ps, double rs, double margin, rawcon* out) {
  double ax[3]; col(ax, Rc, 2);
  double v[3] = {ps[0] - pc[0], ps[1] - pc[1], ps[2] - pc[2]};
  double x = dot3(v, ax), rad[3] = {v[0] - x * ax[0], v[1] - x * ax[1], v[2] - x * ax[2]}, rl = norm3(rad);
  double er[3] = {0, 0, 0};
  if (rl > 1e-14) { for (int k = 0; k < 3; k++) er[k] = rad[k] / rl; } else { double t[3] = {1, 0, 0}; if (fabs(ax[0]) > 0.9) { t[0] = 0; t[1] = 1; } cross3(er, ax, t); double l = norm3(er); for (int k = 0; k < 3; k++) er[k] /= l; }
  double r = sc[0], L = sc[1], sx = x >= 0 ? 1.0 : -1.0;
  double n[3], dist, cp[3]; /* closest point on the cylinder */
  if (fabs(x) <= L && rl <= r) { /* centre inside */
    double dside = r - rl, dcap = L - fabs(x);
    if (dside <= dcap) { copy3(n, er); dist = -dside - rs; for (int k = 0; k < 3; k++) cp[k] = pc[k] + x * ax[k] + r * er[k]; }
    else { for (int k = 0; k < 3; k++) { n[k] = sx * ax[k]; cp[k] = pc[k] + sx * L * ax[k] + rl * er[k]; } dist = -dcap - rs; }
  } else {
    double cx = fabs(x) > L ? sx * L : x, cr = rl > r ? r : rl;
    for (int k = 0; k < 3; k++) cp[k] = pc[k] + cx * ax[k] + cr * er[k];
    double dv[3] = {ps[0] - cp[0], ps[1] - cp[1], ps[2] - cp[2]}; double l = norm3(dv);
    if (l < 1e-14) return 0;
    for (int k = 0; k < 3; k++) n[k] = dv[k] / l;
    dist = l - rs;
  }
  if (dist >= margin) return 0;
  out->dist = dist; copy3(out->n, n); out->has_t = 0;
  for (int k = 0; k < 3; k++) out->pos[k] = cp[k] + n[k] * (0.5 * dist);
  return 1;
}

/* One geom pair -> raw contacts (normal from geom 1 to geom 2 after `flip` is applied by the caller).  Returns the number of contacts; -1 = no routine for
 * this pair of types (the shipped models never pair them). */
static int collide_geoms(int t1, const double* s1, const double* p1, const double* R1, int t2, const double* s2, const double* p2, const double* R2, double margin, rawcon* rc, int* flip) {
  int n = -1; *flip = 0;
  for (int i = 0; i < 8; i++) rc[i].has_t = 0;
  if (t1 == JO_GEOM_PLANE || t2 == JO_GEOM_PLANE) {
    int first = t1 == JO_GEOM_PLANE;
    const double *pp = first ? p1 : p2, *Rp = first ? R1 : R2, *po = first ? p2 : p1, *Ro = first ? R2 : R1, *so = first ? s2 : s1;
    int to = first ? t2 : t1;
    *flip = !first;
    if (to == JO_GEOM_SPHERE) n = collide_plane_sphere(pp, Rp, po, so[0], margin, rc);
    else if (to == JO_GEOM_CAPSULE) n = collide_plane_capsule(pp, Rp, po, Ro, so, margin, rc);
    else if (to == JO_GEOM_BOX) n = collide_plane_box(pp, Rp, po, Ro, so, margin, rc);
  } else
  if (t1 == JO_GEOM_BOX && t2 == JO_GEOM_BOX) n = collide_box_box(p1, R1, s1, p2, R2, s2, margin, rc);
  else if (t1 == JO_GEOM_BOX && t2 == JO_GEOM_SPHERE) n = collide_box_sphere(p1, R1, s1, p2, s2[0], margin, rc);
  else if (t1 == JO_GEOM_SPHERE && t2 == JO_GEOM_BOX) { n = collide_box_sphere(p2, R2, s2, p1, s1[0], margin, rc); *flip = 1; }
  else if (t1 == JO_GEOM_SPHERE && t2 == JO_GEOM_SPHERE) n = collide_sphere_sphere(p1, s1[0], p2, s2[0], margin, rc);
  else if (t1 == JO_GEOM_CYLINDER && t2 == JO_GEOM_CYLINDER) {
    double a1[3], a2[3]; col(a1, R1, 2); col(a2, R2, 2);
    /* parallel axes (cylinder_push, where the planar joints keep them parallel for ever): the closed form; anything else: the general convex routine */
    n = fabs(dot3(a1, a2)) >= 1 - 1e-9 ? collide_cyl_cyl_parallel(p1, R1, s1, p2, R2, s2, margin, rc) : collide_convex(t1, s1, p1, R1, t2, s2, p2, R2, margin, rc);
  }
  else if (t1 == JO_GEOM_CYLINDER && t2 == JO_GEOM_SPHERE) n = collide_cylinder_sphere(p1, R1, s1, p2, s2[0], margin, rc);
  else if (t1 == JO_GEOM_SPHERE && t2 == JO_GEOM_CYLINDER) { n = collide_cylinder_sphere(p2, R2, s2, p1, s1[0], margin, rc); *flip = 1; }
  else if ((t1 == JO_GEOM_CYLINDER && (t2 == JO_GEOM_BOX || t2 == JO_GEOM_CAPSULE)) || (t2 == JO_GEOM_CYLINDER && (t1 == JO_GEOM_BOX || t1 == JO_GEOM_CAPSULE)))
    n = collide_convex(t1, s1, p1, R1, t2, s2, p2, R2, margin, rc);
  else if (t1 == JO_GEOM_CAPSULE && t2 == JO_GEOM_CAPSULE) n = collide_capsule_capsule(p1, R1, s1, p2, R2, s2, margin, rc);
  else if (t1 == JO_GEOM_SPHERE && t2 == JO_GEOM_CAPSULE) n = collide_sphere_capsule(p1, s1[0], p2, R2, s2, margin, rc);
  else if (t1 == JO_GEOM_CAPSULE && t2 == JO_GEOM_SPHERE) { n = collide_sphere_capsule(p2, s2[0], p1, R1, s1, margin, rc); *flip = 1; }
  else if (t1 == JO_GEOM_BOX && t2 == JO_GEOM_CAPSULE) n = collide_box_capsule(p1, R1, s1, p2, R2, s2, margin, rc);
  else if (t1 == JO_GEOM_CAPSULE && t2 == JO_GEOM_BOX) { n = collide_box_capsule(p2, R2, s2, p1, R1, s1, margin, rc); *flip = 1; }
  return n;
}

/* test hook (tests/test_oracle_independent.py): one pair of free-standing shapes -> contacts as rows (dist, pos[3], normal[3] from shape 1 to shape 2) */
int jo_collide_shapes(int t1, const double* s1, const double* p1, const double* q1, int t2, const double* s2, const double* p2, const double* q2, double margin, double* out /* 8*7 */) {
  double R1[9], R2[9]; quat2mat(R1, q1); quat2mat(R2, q2);
  rawcon rc[8]; int flip = 0;
  int n = collide_geoms(t1, s1, p1, R1, t2, s2, p2, R2, margin, rc, &flip);
  for (int i = 0; i < n; i++) { out[7 * i] = rc[i].dist; copy3(out + 7 * i + 1, rc[i].pos); for (int k = 0; k < 3; k++) out[7 * i + 4 + k] = flip ? -rc[i].n[k] : rc[i].n[k]; }
  return n;
}

static void collision(const jo_model* m, jo_data* d) {
  d->ncon = 0;
  if (!m->contact_enabled) return;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_g1[p], g2 = m->pair_g2[p];
    /* mj_collideGeoms orders a pair by geom TYPE (the lower type is geom 1), whatever their order in the model: the contact's normal points from that geom to the other.
     * (For the physics the order is immaterial: mju_makeFrame of -n gives (-n, y, -z) for (n, y, z), and elliptic cones and pyramids are symmetric in the tangents.) */
    if (m->geom_type[g1] > m->geom_type[g2]) { int t = g1; g1 = g2; g2 = t; }
    double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    double dc[3] = {d->geom_xpos[g2][0] - d->geom_xpos[g1][0], d->geom_xpos[g2][1] - d->geom_xpos[g1][1], d->geom_xpos[g2][2] - d->geom_xpos[g1][2]};
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    if (t1 != JO_GEOM_PLANE && t2 != JO_GEOM_PLANE && norm3(dc) > m->geom_rbound[g1] + m->geom_rbound[g2] + margin) continue; /* bounding-sphere filter */
    rawcon rc[8]; int flip = 0;
    int n = collide_geoms(t1, m->geom_size[g1], d->geom_xpos[g1], d->geom_xmat[g1], t2, m->geom_size[g2], d->geom_xpos[g2], d->geom_xmat[g2], margin, rc, &flip);
    for (int i = 0; i < n; i++) {  /* n = -1 (no routine for the pair): nothing */
      if (d->ncon >= JO_MAXCON) { d->con_overflow++; break; }
      jo_contact* c = &d->con[d->ncon++];
      c->dist = rc[i].dist; copy3(c->pos, rc[i].pos);
      for (int k = 0; k < 3; k++) c->frame[k] = flip ? -rc[i].n[k] : rc[i].n[k];
      make_frame(c->frame);
      if (rc[i].has_t) { /* mju_makeFrame with a given second axis: orthogonalise against the normal, keep it if it survives */
        double y[3]; copy3(y, rc[i].t); double dp = dot3(c->frame, y); addscl3(y, c->frame, -dp);
        double nn = norm3(y);
        if (nn > 0.5 * 1e-3) { for (int k = 0; k < 3; k++) c->frame[3 + k] = y[k] / nn; cross3(c->frame + 6, c->frame, c->frame + 3); }
      }
      c->g1 = g1; c->g2 = g2;
      /* contact parameter mixing (mj_contactParam, equal priority / solmix): condim max, friction max,
       * solref/solimp average, margin/gap max */
      c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
      double f[3]; for (int k = 0; k < 3; k++) f[k] = fmax(m->geom_friction[g1][k], m->geom_friction[g2][k]);
      int gpri = m->geom_priority[g1] > m->geom_priority[g2] ? g1 : (m->geom_priority[g2] > m->geom_priority[g1] ? g2 : -1);
      if (gpri >= 0) { c->dim = m->geom_condim[gpri]; for (int k = 0; k < 3; k++) f[k] = m->geom_friction[gpri][k]; } /* different priorities: the higher one decides */
      c->friction[0] = c->friction[1] = f[0]; c->friction[2] = f[1]; c->friction[3] = c->friction[4] = f[2];
      for (int k = 0; k < 5; k++) if (c->friction[k] < MINMU) c->friction[k] = MINMU;
      if (m->geom_solref[g1][0] > 0 && m->geom_solref[g2][0] > 0) for (int k = 0; k < 2; k++) c->solref[k] = 0.5 * (m->geom_solref[g1][k] + m->geom_solref[g2][k]);
      else for (int k = 0; k < 2; k++) c->solref[k] = fmin(m->geom_solref[g1][k], m->geom_solref[g2][k]);
      for (int k = 0; k < 5; k++) c->solimp[k] = 0.5 * (m->geom_solimp[g1][k] + m->geom_solimp[g2][k]);
      if (gpri >= 0) { memcpy(c->solref, m->geom_solref[gpri], 2 * sizeof(double)); memcpy(c->solimp, m->geom_solimp[gpri], 5 * sizeof(double)); }
      c->includemargin = margin - fmax(m->geom_gap[g1], m->geom_gap[g2]);
      c->efc_adr = -1; c->mu = c->friction[0];
    }
  }
}

/* ------------------------------------------------------------------ constraint rows (mj_makeConstraint, mj_makeImpedance) */
static void point_jac(const jo_model* m, const jo_data* d, int body, const double* p, double Jp[3][JO_MAXDOF]) {
  for (int k = 0; k < 3; k++) for (int i = 0; i < m->nv; i++) Jp[k][i] = 0;
  if (body <= 0) return;
  /* last dof of the body, then up the chain */
  int b = body, i = -1;
  while (b > 0 && m->body_jntnum[b] == 0) b = m->body_parent[b];
  if (b <= 0) return;
  int jl = m->body_jntadr[b] + m->body_jntnum[b] - 1;
  i = m->jnt_dofadr[jl] + (m->jnt_type[jl] == JO_JNT_FREE ? 5 : 0);
  for (; i >= 0; i = m->dof_parent[i]) {
    double v[3]; cross3(v, d->S[i], p); /* omega x p + v_origin */
    for (int k = 0; k < 3; k++) Jp[k][i] = v[k] + d->S[i][3 + k];
  }
}

static void impedance(const double* solimp_in, double pos, double margin, double* imp) {
  double si[5]; memcpy(si, solimp_in, sizeof(si));
  si[0] = fmin(MAXIMP, fmax(MINIMP, si[0])); si[1] = fmin(MAXIMP, fmax(MINIMP, si[1]));
  si[2] = fmax(0, si[2]); si[3] = fmin(MAXIMP, fmax(MINIMP, si[3])); si[4] = fmax(1, si[4]);
  if (si[0] == si[1] || si[2] <= MINVAL) { *imp = 0.5 * (si[0] + si[1]); return; }
  double x = fabs((pos - margin) / si[2]);
  if (x >= 1) { *imp = si[1]; return; }
  if (x <= 0) { *imp = si[0]; return; }
  double y;
  if (si[4] == 1) y = x;
  else if (x <= si[3]) y = pow(x, si[4]) / pow(si[3], si[4] - 1);
  else y = 1 - pow(1 - x, si[4]) / pow(1 - si[3], si[4] - 1);
  *imp = si[0] + y * (si[1] - si[0]);
}

static int add_row(jo_data* d, int type, int id, double pos, double margin, double frictionloss) {
  if (d->nefc >= JO_MAXEFC) return -1;
  int r = d->nefc++;
  d->efc_type[r] = type; d->efc_id[r] = id; d->efc_pos[r] = pos; d->efc_margin[r] = margin; d->efc_frictionloss[r] = frictionloss;
  return r;
}

static void make_constraint(const jo_model* m, jo_data* d) {
  int nv = m->nv;
  d->nefc = 0;
  /* 1. equality (joint coupling): (q1 - q1_0) - poly(q2 - q2_0) = 0 */
  for (int e = 0; e < m->neq; e++) {
    int j1 = m->eq_j1[e], j2 = m->eq_j2[e];
    double x = d->qpos[m->jnt_qposadr[j2]] - m->qpos0[m->jnt_qposadr[j2]], y = d->qpos[m->jnt_qposadr[j1]] - m->qpos0[m->jnt_qposadr[j1]];
    const double* a = m->eq_poly[e];
    double poly = a[0] + x * (a[1] + x * (a[2] + x * (a[3] + x * a[4]))), dpoly = a[1] + x * (2 * a[2] + x * (3 * a[3] + x * 4 * a[4]));
    int r = add_row(d, JO_EFC_EQUALITY, e, y - poly, 0, 0);
    if (r < 0) return;
    memset(d->efc_J[r], 0, sizeof(double) * nv);
    d->efc_J[r][m->jnt_dofadr[j1]] = 1; d->efc_J[r][m->jnt_dofadr[j2]] = -dpoly;
    d->efc_diagApprox[r] = m->dof_invweight0[m->jnt_dofadr[j1]] + m->dof_invweight0[m->jnt_dofadr[j2]];
  }
  /* 2. dof friction loss */
  for (int i = 0; i < nv; i++)
    if (m->dof_frictionloss[i] > 0) {
      int r = add_row(d, JO_EFC_FRICTION, i, 0, 0, m->dof_frictionloss[i]);
      if (r < 0) return;
      memset(d->efc_J[r], 0, sizeof(double) * nv); d->efc_J[r][i] = 1;
      d->efc_diagApprox[r] = m->dof_invweight0[i];
    }
  /* 3. joint limits (hinge / slide) */
  for (int j = 0; j < m->njnt; j++)
    if (m->jnt_limited[j] && m->jnt_type[j] != JO_JNT_FREE) {
      double q = d->qpos[m->jnt_qposadr[j]];
      for (int side = -1; side <= 1; side += 2) {
        double dist = side * (m->jnt_range[j][(side + 1) / 2] - q);
        if (dist < m->jnt_margin[j]) {
          int r = add_row(d, JO_EFC_LIMIT, j, dist, m->jnt_margin[j], 0);
          if (r < 0) return;
          memset(d->efc_J[r], 0, sizeof(double) * nv); d->efc_J[r][m->jnt_dofadr[j]] = -side;
          d->efc_diagApprox[r] = m->dof_invweight0[m->jnt_dofadr[j]];
        }
      }
    }
  /* 4. contacts */
  static __thread double J1[3][JO_MAXDOF], J2[3][JO_MAXDOF];
  for (int c = 0; c < d->ncon; c++) {
    jo_contact* con = &d->con[c];
    int b1 = m->geom_body[con->g1], b2 = m->geom_body[con->g2];
    point_jac(m, d, b1, con->pos, J1); point_jac(m, d, b2, con->pos, J2);
    double Jf[3][JO_MAXDOF]; /* relative-velocity Jacobian expressed in the contact frame */
    for (int r = 0; r < 3; r++) for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < 3; k++) s += con->frame[3 * r + k] * (J2[k][i] - J1[k][i]); Jf[r][i] = s; }
    double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    if (con->dim == 1) {
      int r = add_row(d, JO_EFC_CONTACT_FRICTIONLESS, c, con->dist, con->includemargin, 0);
      if (r < 0) return;
      con->efc_adr = r; memcpy(d->efc_J[r], Jf[0], sizeof(double) * nv); d->efc_diagApprox[r] = tran;
    } else if (m->cone == JO_CONE_PYRAMIDAL) {
      for (int k = 0; k < con->dim - 1; k++)
        for (int sg = 1; sg >= -1; sg -= 2) {
          int r = add_row(d, JO_EFC_CONTACT_PYRAMIDAL, c, con->dist, con->includemargin, 0);
          if (r < 0) return;
          if (k == 0 && sg == 1) con->efc_adr = r;
          double fr = con->friction[k];
          for (int i = 0; i < nv; i++) d->efc_J[r][i] = Jf[0][i] + sg * fr * Jf[1 + k][i];
          d->efc_diagApprox[r] = tran + fr * fr * tran;
        }
    } else {
      for (int k = 0; k < con->dim; k++) {
        int r = add_row(d, JO_EFC_CONTACT_ELLIPTIC, c, k == 0 ? con->dist : 0.0, k == 0 ? con->includemargin : 0.0, 0);
        if (r < 0) return;
        if (k == 0) con->efc_adr = r;
        memcpy(d->efc_J[r], Jf[k], sizeof(double) * nv);
        d->efc_diagApprox[r] = tran; /* condim 3: both friction rows are translational */
      }
    }
  }
  /* impedance, regulariser, reference acceleration */
  for (int r = 0; r < d->nefc; r++) {
    const double *solref, *solimp; int tp = d->efc_type[r], id = d->efc_id[r];
    switch (tp) {
      case JO_EFC_EQUALITY: solref = m->eq_solref[id]; solimp = m->eq_solimp[id]; break;
      case JO_EFC_FRICTION: solref = m->dof_solref[id]; solimp = m->dof_solimp[id]; break;
      case JO_EFC_LIMIT: solref = m->jnt_solref[id]; solimp = m->jnt_solimp[id]; break;
      default: solref = d->con[id].solref; solimp = d->con[id].solimp; break;
    }
    double imp; impedance(solimp, d->efc_pos[r], d->efc_margin[r], &imp);
    double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
    double K, B;
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2 * m->dt) /* refsafe */, dr = solref[1];
      K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr); B = 2 / fmax(MINVAL, dmax * tc);
    } else { K = -solref[0] / fmax(MINVAL, dmax * dmax); B = -solref[1] / fmax(MINVAL, dmax); }
    int is_friction = tp == JO_EFC_FRICTION || (tp == JO_EFC_CONTACT_ELLIPTIC && r != d->con[id].efc_adr);
    if (is_friction) K = 0;
    d->efc_KBIP[r][0] = K; d->efc_KBIP[r][1] = B; d->efc_KBIP[r][2] = imp; d->efc_KBIP[r][3] = 0;
    d->efc_R[r] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[r] / imp);
  }
  for (int c = 0; c < d->ncon; c++) { /* frictional contacts: friction-row regularisers */
    jo_contact* con = &d->con[c]; int id = con->efc_adr;
    if (id < 0 || con->dim < 2) continue;
    if (m->cone == JO_CONE_PYRAMIDAL) {
      con->mu = con->friction[0] * sqrt(1 / fmax(MINVAL, m->impratio)); /* regularised cone: impratio divides Rpy */
      double Rpy = 2 * con->mu * con->mu * d->efc_R[id];
      for (int r = id; r < id + 2 * (con->dim - 1); r++) d->efc_R[r] = fmax(MINVAL, Rpy);
    } else {
      d->efc_R[id + 1] = d->efc_R[id] / fmax(MINVAL, m->impratio);
      con->mu = con->friction[0] * sqrt(d->efc_R[id + 1] / d->efc_R[id]);
      for (int k = 1; k < con->dim - 1; k++) d->efc_R[id + 1 + k] = d->efc_R[id + 1] * con->friction[0] * con->friction[0] / (con->friction[k] * con->friction[k]);
    }
  }
  for (int r = 0; r < d->nefc; r++) {
    d->efc_D[r] = 1 / d->efc_R[r];
    double v = 0; for (int i = 0; i < nv; i++) v += d->efc_J[r][i] * d->qvel[i];
    d->efc_vel[r] = v;
    d->efc_aref[r] = -d->efc_KBIP[r][1] * v - d->efc_KBIP[r][0] * d->efc_KBIP[r][2] * (d->efc_pos[r] - d->efc_margin[r]);
  }
}

/* ------------------------------------------------------------------ constraint cost s(jar): value, force = -ds/djar, Hessian weights */
/* returns cost; force[] filled; if Hd != NULL: Hd[r] = diagonal quadratic weight of row r (0 if inactive),
 * and for elliptic contacts in the cone zone Hc[c][3][3] holds the dense 3x3 block (flag in cone_zone[c]). */
static double constraint_cost(const jo_model* m, const jo_data* d, const double* jar, double* force, double* Hd, double (*Hc)[9], int* cone_zone) {
  (void)m;
  double cost = 0;
  for (int r = 0; r < d->nefc; r++) {
    int tp = d->efc_type[r]; double D = d->efc_D[r], R = d->efc_R[r], x = jar[r];
    if (Hd && !(tp == JO_EFC_CONTACT_ELLIPTIC && r != d->con[d->efc_id[r]].efc_adr)) Hd[r] = 0; /* elliptic friction rows are written with their normal row */
    switch (tp) {
      case JO_EFC_EQUALITY: force[r] = -D * x; cost += 0.5 * D * x * x; if (Hd) Hd[r] = D; break;
      case JO_EFC_FRICTION: {
        double fl = d->efc_frictionloss[r];
        if (x <= -R * fl) { force[r] = fl; cost += -0.5 * R * fl * fl - fl * x; }
        else if (x >= R * fl) { force[r] = -fl; cost += -0.5 * R * fl * fl + fl * x; }
        else { force[r] = -D * x; cost += 0.5 * D * x * x; if (Hd) Hd[r] = D; }
      } break;
      case JO_EFC_LIMIT: case JO_EFC_CONTACT_FRICTIONLESS: case JO_EFC_CONTACT_PYRAMIDAL:
        if (x < 0) { force[r] = -D * x; cost += 0.5 * D * x * x; if (Hd) Hd[r] = D; } else force[r] = 0;
        break;
      case JO_EFC_CONTACT_ELLIPTIC: {
        int c = d->efc_id[r]; const jo_contact* con = &d->con[c];
        if (r != con->efc_adr) break; /* handled with the normal row */
        int dim = con->dim; double mu = con->mu, U[6], T2 = 0;
        U[0] = jar[r] * mu;
        for (int j = 1; j < dim; j++) { U[j] = jar[r + j] * con->friction[j - 1]; T2 += U[j] * U[j]; }
        double N = U[0], T = sqrt(T2);
        if (cone_zone) cone_zone[c] = 0;
        if (Hd) for (int j = 0; j < dim; j++) Hd[r + j] = 0;
        if (N >= mu * T || (T <= 0 && N >= 0)) { for (int j = 0; j < dim; j++) force[r + j] = 0; } /* top zone */
        else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { /* bottom zone: quadratic in every row */
          for (int j = 0; j < dim; j++) { force[r + j] = -d->efc_D[r + j] * jar[r + j]; cost += 0.5 * d->efc_D[r + j] * jar[r + j] * jar[r + j]; if (Hd) Hd[r + j] = d->efc_D[r + j]; }
        } else { /* middle zone: distance to the cone */
          double Dm = d->efc_D[r] / (mu * mu * (1 + mu * mu)), NT = N - mu * T;
          cost += 0.5 * Dm * NT * NT;
          force[r] = -Dm * NT * mu;
          for (int j = 1; j < dim; j++) force[r + j] = -force[r] / T * U[j] * con->friction[j - 1];
          if (Hc && dim == 3) {
            cone_zone[c] = 1;
            double sc[3] = {mu, con->friction[0], con->friction[1]}, HU[3][3];
            HU[0][0] = Dm;
            for (int j = 1; j < 3; j++) HU[0][j] = HU[j][0] = -Dm * mu * U[j] / T;
            for (int j = 1; j < 3; j++) for (int k = 1; k < 3; k++)
              HU[j][k] = Dm * mu * mu * U[j] * U[k] / (T * T) - Dm * NT * mu * ((j == k ? 1.0 / T : 0.0) - U[j] * U[k] / (T * T * T));
            for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) Hc[c][3 * j + k] = sc[j] * HU[j][k] * sc[k];
          }
        }
      } break;
    }
  }
  return cost;
}

/* total cost + gradient at acceleration a */
static double total_cost(const jo_model* m, jo_data* d, const double* a, double* grad, double* jar, double* Hd, double (*Hc)[9], int* cone_zone) {
  int nv = m->nv;
  for (int r = 0; r < d->nefc; r++) { double s = -d->efc_aref[r]; for (int i = 0; i < nv; i++) s += d->efc_J[r][i] * a[i]; jar[r] = s; }
  double cost = constraint_cost(m, d, jar, d->efc_force, Hd, Hc, cone_zone);
  double Ma[JO_MAXDOF];
  for (int i = 0; i < nv; i++) { double s = 0; for (int j = 0; j < nv; j++) s += d->M[i][j] * (a[j] - d->qacc_smooth[j]); Ma[i] = s; }
  for (int i = 0; i < nv; i++) cost += 0.5 * Ma[i] * (a[i] - d->qacc_smooth[i]);
  if (grad) for (int i = 0; i < nv; i++) { double s = Ma[i]; for (int r = 0; r < d->nefc; r++) s -= d->efc_J[r][i] * d->efc_force[r]; grad[i] = s; }
  return cost;
}

/* ------------------------------------------------------------------ primal Newton solver with exact line search (mj_solNewton) */
/* diagnostics for the tests/tools: Newton iterations per solve, over all threads */
static long g_iter_hist[32];
static int g_trace = 0;
#ifdef JO_EXPERIMENTS
static int g_wsmode = 0;
void jo_set_warmstart_mode(int mode) { g_wsmode = mode; }
#endif
void jo_set_trace(int on) { g_trace = on; }
void jo_solver_histogram(long* out32, int reset) { for (int i = 0; i < 32; i++) { out32[i] = g_iter_hist[i]; if (reset) g_iter_hist[i] = 0; } }
void jo_set_solver(jo_model* m, double tol, int maxiter) { m->solver_tol = tol; m->solver_maxiter = maxiter; }
#ifdef JO_EXPERIMENTS /* the CPU prototypes of round 4 (line search, Hessian reuse, warm start): compiled ONLY into libjudo_oracle_exp.so (`make exp`, loaded by
   tools/proto/ls_experiment.py and tools/diag/oracle_newton_hist.py) -- the library every parity test compares against has none of this state. */
/* line-search experiments (tools/proto/ls_experiment.py; single-threaded runs only): mode 0 = the oracle's search (default: to rounding), mode >= 1 = the KERNELS' search --
   start at 1, stop at |slope| <= lstol |slope(0)|, at most lsmax evaluations, bisection when the Newton step leaves the bracket -- with, for mode 2, a trial at the zero
   crossing of a dof friction-loss row inside the bracket instead of the bisection (the slope jumps there; the kernels' long searches are bisections onto such a jump).
   Every solve appends (-1, number of rows) and per Newton iteration the number of slope evaluations to the log. */
static int g_lsmode = 0, g_lsmax = 16; static double g_lstol = 1e-2, g_lskink = 0.1, g_lsshrink = 0.5;
void jo_set_ls_shrink(double v) { g_lsshrink = v; }
static int g_hreuse = 0; static __thread double g_last_alpha = 0; static long g_hreuse_count = 0;
void jo_set_hessian_reuse(int mode) { g_hreuse = mode; g_hreuse_count = 0; } long jo_hessian_reuse_count(void) { return g_hreuse_count; }
static long g_lstrouble = 0; long jo_ls_trouble(int reset) { long v = g_lstrouble; if (reset) g_lstrouble = 0; return v; }
void jo_set_ls_kink(double v) { g_lskink = v; } static int* g_lslog = NULL; static long g_lslog_n = 0, g_lslog_cap = 0;
void jo_set_ls_experiment(int mode, double lstol, int lsmax, int* log, long cap) { g_lsmode = mode; g_lstol = lstol; g_lsmax = lsmax; g_lslog = log; g_lslog_cap = cap; g_lslog_n = 0; }
long jo_ls_log_size(void) { return g_lslog_n; }
static void lslog(int v) { if (g_lslog && g_lslog_n < g_lslog_cap) g_lslog[g_lslog_n++] = v; }

#endif
static void solve_constraints(const jo_model* m, jo_data* d) {
  int nv = m->nv, ne = d->nefc;
  if (ne == 0) { memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv); memset(d->qfrc_constraint, 0, sizeof(double) * nv); d->solver_iter = 0; return; }
  static __thread double jar[JO_MAXEFC], Hd[JO_MAXEFC], Hc[JO_MAXCON][9], jp[JO_MAXEFC], jar2[JO_MAXEFC], frc2[JO_MAXEFC], Hd2[JO_MAXEFC], Hc2[JO_MAXCON][9];
  static __thread int cz[JO_MAXCON], cz2[JO_MAXCON];
  static __thread double H[JO_MAXDOF][JO_MAXDOF], LH[JO_MAXDOF][JO_MAXDOF];
  double a[JO_MAXDOF], grad[JO_MAXDOF], p[JO_MAXDOF], Mp[JO_MAXDOF];
  /* warm start: the better of the previous acceleration and the unconstrained one (mj_fwdConstraint) */
  double cw = total_cost(m, d, d->qacc_warmstart, NULL, jar, NULL, NULL, NULL);
  double cs = total_cost(m, d, d->qacc_smooth, NULL, jar, NULL, NULL, NULL);
  memcpy(a, cw < cs ? d->qacc_warmstart : d->qacc_smooth, sizeof(double) * nv);
#ifdef JO_EXPERIMENTS
  if (g_wsmode == 1) { /* experiment: keep last step's constraint acceleration on top of the new smooth acceleration */
    double a3[JO_MAXDOF]; for (int i = 0; i < nv; i++) a3[i] = d->qacc_smooth[i] + d->qacc_con_prev[i];
    double c3 = total_cost(m, d, a3, NULL, jar, NULL, NULL, NULL);
    if (c3 < fmin(cw, cs)) memcpy(a, a3, sizeof(double) * nv);
  }
#endif
  double scale = 0; for (int i = 0; i < nv; i++) scale += d->M[i][i]; scale = 1.0 / fmax(MINVAL, scale);
  int it;
  for (it = 0; it < m->solver_maxiter; it++) {
    double cost = total_cost(m, d, a, grad, jar, Hd, Hc, cz);
    double gn = 0; for (int i = 0; i < nv; i++) gn += grad[i] * grad[i]; gn = sqrt(gn);
    d->solver_cost = cost; d->solver_gradnorm = gn;
    if (gn * scale < m->solver_tol) break;
#ifdef JO_EXPERIMENTS
    if (g_hreuse && it > 0 && ((g_hreuse == 1 && (it & 1)) || (g_hreuse == 2 && g_last_alpha > 0.8 && g_last_alpha < 1.25) || (g_hreuse == 3 && (it % 3) != 0))) { g_hreuse_count++; goto have_factor; }  /* (experiment: the previous factor) */
#endif
    for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) H[i][j] = d->M[i][j];
    for (int r = 0; r < ne; r++) if (Hd[r] != 0) for (int i = 0; i < nv; i++) { double ji = d->efc_J[r][i]; if (ji != 0) for (int j = 0; j < nv; j++) H[i][j] += Hd[r] * ji * d->efc_J[r][j]; }
    for (int c = 0; c < d->ncon; c++) if (d->con[c].efc_adr >= 0 && m->cone == JO_CONE_ELLIPTIC && d->con[c].dim == 3 && cz[c]) {
      int r0 = d->con[c].efc_adr;
      for (int u = 0; u < 3; u++) for (int v = 0; v < 3; v++) { double w = Hc[c][3 * u + v]; for (int i = 0; i < nv; i++) { double ji = d->efc_J[r0 + u][i]; if (ji != 0) for (int j = 0; j < nv; j++) H[i][j] += w * ji * d->efc_J[r0 + v][j]; } }
    }
    if (cholesky(nv, H, LH) != 0) break;
#ifdef JO_EXPERIMENTS
    have_factor:
#endif
    for (int i = 0; i < nv; i++) p[i] = -grad[i];
    chol_solve(nv, LH, p);
    /* exact line search: phi(al) = cost(a + al p); safeguarded 1-D Newton on phi' */
    for (int r = 0; r < ne; r++) { double s = 0; for (int i = 0; i < nv; i++) s += d->efc_J[r][i] * p[i]; jp[r] = s; }
    for (int i = 0; i < nv; i++) { double s = 0; for (int j = 0; j < nv; j++) s += d->M[i][j] * p[j]; Mp[i] = s; }
    double pMp = 0, pMd = 0; for (int i = 0; i < nv; i++) { pMp += p[i] * Mp[i]; pMd += Mp[i] * (a[i] - d->qacc_smooth[i]); }
    double lo = 0, hi = -1, al = 1.0, dlo = 0;
    { double g0 = 0; for (int i = 0; i < nv; i++) g0 += grad[i] * p[i]; dlo = g0; if (g0 >= 0) break; }
#ifdef JO_EXPERIMENTS
    if (it == 0) { lslog(-1); lslog(ne); }
    if (g_lsmode >= 1) {
      const double g0 = dlo; double dlo_v = g0, dhi_v = 0; int nev = 0, kink_tries = 0;
      /* candidate step lengths at which the slope (all but) jumps or its curvature jumps: zero crossings of the dof friction-loss rows (mode >= 4); per contact the point where
         the tangential part passes closest to zero, if it gets there close enough for Coulomb friction to reverse (mode >= 5); the crossings of the cone surface N = mu T,
         where a contact switches on or off (mode >= 7) */
      static __thread double cand[4 * JO_MAXEFC]; int ncand = 0;
      if (g_lsmode >= 4) for (int r = 0; r < ne; r++) if (d->efc_type[r] == JO_EFC_FRICTION && jp[r] != 0) { const double a = -jar[r] / jp[r]; if (a > 0) cand[ncand++] = a; }
      if (g_lsmode >= 5) for (int c = 0; c < d->ncon; c++) if (d->con[c].efc_adr >= 0 && m->cone == JO_CONE_ELLIPTIC && d->con[c].dim == 3) {
        const int r0 = d->con[c].efc_adr; const double f1 = d->con[c].friction[0], f2 = d->con[c].friction[1], mu = d->con[c].mu;
        const double U1 = jar[r0 + 1] * f1, U2 = jar[r0 + 2] * f2, V1 = jp[r0 + 1] * f1, V2 = jp[r0 + 2] * f2, vv = V1 * V1 + V2 * V2, uv = U1 * V1 + U2 * V2, uu = U1 * U1 + U2 * U2;
        if (vv > 0) {
          const double a = -uv / vv, tm2 = uu + a * uv;  /* |U + a V|^2 at the minimum */
          if (a > 0 && tm2 <= g_lskink * g_lskink * fmax(uu, uu + 2 * uv + vv)) cand[ncand++] = a;
        }
        if (g_lsmode >= 7) {  /* (N0 + a V0)^2 = mu^2 |U + a V|^2 with N0 + a V0 >= 0 */
          const double N0 = jar[r0] * mu, V0 = jp[r0] * mu, A = V0 * V0 - mu * mu * vv, B = 2 * (N0 * V0 - mu * mu * uv), Cq = N0 * N0 - mu * mu * uu, disc = B * B - 4 * A * Cq;
          if (fabs(A) > 1e-300 && disc >= 0) {
            const double sq = sqrt(disc);
            for (int sgn = -1; sgn <= 1; sgn += 2) { const double a = (-B + sgn * sq) / (2 * A); if (a > 0 && N0 + a * V0 >= 0) cand[ncand++] = a; }
          } else if (fabs(A) <= 1e-300 && B != 0) { const double a = -Cq / B; if (a > 0 && N0 + a * V0 >= 0) cand[ncand++] = a; }
        }
      }
      if (g_lsmode == 8) { double best = 1e300; for (int k = 0; k < ncand; k++) if (cand[k] < 1.0 && cand[k] < best) best = cand[k]; if (best < 1.0) al = best; }  /* (mode 8: the first candidate below 1 first) */
      for (int ls = 0; ls < g_lsmax; ls++) {
        for (int r = 0; r < ne; r++) jar2[r] = jar[r] + al * jp[r];
        constraint_cost(m, d, jar2, frc2, Hd2, Hc2, cz2);
        double d1 = pMd + al * pMp, d2 = pMp; nev++;
        for (int r = 0; r < ne; r++) { d1 -= frc2[r] * jp[r]; d2 += Hd2[r] * jp[r] * jp[r]; }
        for (int c = 0; c < d->ncon; c++) if (d->con[c].efc_adr >= 0 && m->cone == JO_CONE_ELLIPTIC && d->con[c].dim == 3 && cz2[c]) {
          int r0 = d->con[c].efc_adr; for (int u = 0; u < 3; u++) for (int v = 0; v < 3; v++) d2 += Hc2[c][3 * u + v] * jp[r0 + u] * jp[r0 + v];
        }
        if (g_trace == 2) fprintf(stderr, "    ls %2d alpha %.9g d1 %.6e d2 %.6e (g0 %.3e) lo %.9g hi %.9g\n", ls, al, d1, d2, g0, lo, hi);
        if (fabs(d1) <= g_lstol * fabs(g0)) break;
        const double w_before = hi >= 0 ? hi - lo : -1;
        if (d1 < 0) { lo = al; dlo_v = d1; } else { hi = al; dhi_v = d1; }
        double nx = al - d1 / d2;
        if (g_lsmode >= 10 && g_lsmode < 20) {  /* candidates only where the plain search is in trouble: a rejected Newton step, or a bracket that the last evaluation did not shrink to
                                 g_lsshrink of its width */
          const int bracketed = hi >= 0, rejected = bracketed ? (nx <= lo || nx >= hi) : (nx <= lo);
          const int slow = bracketed && w_before > 0 && (hi - lo) > g_lsshrink * w_before;
          if (!bracketed) { if (rejected) nx = 2 * al; }
          else if (rejected || slow) {
            g_lstrouble++;
            const double mid = 0.5 * (lo + hi); double best = 1e300, ak = -1;
            for (int k = 0; k < ncand; k++) if (cand[k] > lo && cand[k] < hi && fabs(cand[k] - lo) > 1e-9 * lo && fabs(hi - cand[k]) > 1e-9 * hi && fabs(cand[k] - mid) < best) { best = fabs(cand[k] - mid); ak = cand[k]; }
            nx = ak > 0 ? ak : (rejected || g_lsmode == 11 ? mid : nx);
          }
          al = nx; continue;
        }
        if (hi < 0) { if (nx <= lo) nx = 2 * al; }
        else if (nx <= lo || nx >= hi) {
          nx = 0.5 * (lo + hi);
          if (g_lsmode == 9) {  /* (mode 9: the first candidate from the point just evaluated towards the other end of the bracket) */
            const double oth = d1 < 0 ? hi : lo; double best = 1e300, ak = -1;
            for (int k = 0; k < ncand; k++) { const double a = cand[k]; const int between = oth > al ? (a > al && a < oth) : (a < al && a > oth);
              if (between && fabs(a - al) > 1e-9 * (fabs(al) + 1e-30) && fabs(a - al) < best) { best = fabs(a - al); ak = a; } }
            if (ak > 0) nx = ak;
          } else
          if (g_lsmode >= 6) {
            double best = 1e300, ak = -1;
            for (int k = 0; k < ncand; k++) if (cand[k] > lo && cand[k] < hi && fabs(cand[k] - nx) < best) { best = fabs(cand[k] - nx); ak = cand[k]; }
            if (ak > 0) nx = ak;
          } else
          if (g_lsmode >= 2 && kink_tries < g_lsmode - 1) {  /* the friction-loss zero crossing inside the bracket that is closest to the secant estimate */
            const double as = lo - dlo_v * (hi - lo) / (dhi_v - dlo_v); double best = 1e300, ak = -1;
            for (int r = 0; r < ne; r++) if (d->efc_type[r] == JO_EFC_FRICTION && jp[r] != 0) {
              const double a = -jar[r] / jp[r];
              if (a > lo && a < hi && fabs(a - as) < best) { best = fabs(a - as); ak = a; }
            }
            if (ak > 0) { nx = ak; kink_tries++; }
          }
        }
        if (g_lsmode >= 4) {  /* clip the step at the first candidate it passes: the slope model behind `nx` does not hold across it */
          double best = 1e300, ak = -1;
          for (int k = 0; k < ncand; k++) {
            const double a = cand[k];
            const int between = nx > al ? (a > al && a < nx) : (a < al && a > nx);
            if (between && (hi < 0 || (a > lo && a < hi)) && fabs(a - al) > 1e-9 * (fabs(al) + 1e-30) && fabs(a - al) < best) { best = fabs(a - al); ak = a; }
          }
          if (ak > 0) nx = ak;
        }
        al = nx;
      }
      lslog(nev);
      if (g_trace == 2) fprintf(stderr, "  == it %d: %d evaluations, ncon %d\n", it, nev, d->ncon);
    } else
#endif
    for (int ls = 0; ls < 60; ls++) {
      for (int r = 0; r < ne; r++) jar2[r] = jar[r] + al * jp[r];
      constraint_cost(m, d, jar2, frc2, Hd2, Hc2, cz2);
      double d1 = pMd + al * pMp, d2 = pMp;
      for (int r = 0; r < ne; r++) { d1 -= frc2[r] * jp[r]; d2 += Hd2[r] * jp[r] * jp[r]; }
      for (int c = 0; c < d->ncon; c++) if (d->con[c].efc_adr >= 0 && m->cone == JO_CONE_ELLIPTIC && d->con[c].dim == 3 && cz2[c]) {
        int r0 = d->con[c].efc_adr; for (int u = 0; u < 3; u++) for (int v = 0; v < 3; v++) d2 += Hc2[c][3 * u + v] * jp[r0 + u] * jp[r0 + v];
      }
      if (fabs(d1) < 1e-14 * (fabs(dlo) + 1e-300) || fabs(d1) < 1e-300) break;
      if (d1 < 0) lo = al; else hi = al;
      double nx = al - d1 / d2;
      if (hi < 0) { if (nx <= lo) nx = 2 * al + 1e-12; }
      else if (nx <= lo || nx >= hi) nx = 0.5 * (lo + hi);
      if (fabs(nx - al) <= 1e-15 * fabs(al)) { al = nx; break; }
      al = nx;
    }
    if (g_trace) {
      int nz[3] = {0, 0, 0}; for (int c = 0; c < d->ncon; c++) if (d->con[c].efc_adr >= 0) nz[cz[c] ? 1 : (d->efc_force[d->con[c].efc_adr] == 0 ? 0 : 2)]++;
      fprintf(stderr, "  it %2d cost %.6e |g|s %.3e alpha %.4g  ncon %d nefc %d zones top(free)/middle/bottom(stick) %d/%d/%d\n", it, cost, gn * scale, al, d->ncon, ne, nz[0], nz[1], nz[2]);
    }
    for (int i = 0; i < nv; i++) a[i] += al * p[i];
#ifdef JO_EXPERIMENTS
    g_last_alpha = al;
#endif
  }
  if (g_trace) fprintf(stderr, "solve done: %d iterations (warm start %s)\n", it, cw < cs ? "used" : "not used");
  d->solver_iter = it;
  __atomic_fetch_add(&g_iter_hist[it < 31 ? it : 31], 1, __ATOMIC_RELAXED);
  total_cost(m, d, a, grad, jar, NULL, NULL, NULL);
  memcpy(d->qacc, a, sizeof(double) * nv);
  for (int i = 0; i < nv; i++) { double s = 0; for (int r = 0; r < ne; r++) s += d->efc_J[r][i] * d->efc_force[r]; d->qfrc_constraint[i] = s; }
}

/* ------------------------------------------------------------------ sensors (position stage) */
static void sensors(const jo_model* m, jo_data* d) {
  for (int s = 0; s < m->nsensor; s++) {
    double* o = d->sensordata + m->sensor_adr[s]; int obj = m->sensor_obj[s];
    switch (m->sensor_type[s]) {
      case JO_SENS_FRAMEPOS_SITE: /* obj2 >= 0: expressed in the frame of that reference site (mj_sensorPos, reftype site): R_ref' (p - p_ref) */
        copy3(o, d->site_xpos[obj]);
        if (m->sensor_obj2[s] >= 0) {
          int rs = m->sensor_obj2[s]; const double* R = d->xmat[m->site_body[rs]];
          double dv[3] = {o[0] - d->site_xpos[rs][0], o[1] - d->site_xpos[rs][1], o[2] - d->site_xpos[rs][2]};
          for (int k = 0; k < 3; k++) o[k] = R[k] * dv[0] + R[3 + k] * dv[1] + R[6 + k] * dv[2];
        }
        break;
      case JO_SENS_FRAMEXAXIS_SITE: col(o, d->xmat[m->site_body[obj]], 0); break;
      case JO_SENS_FRAMEYAXIS_SITE: col(o, d->xmat[m->site_body[obj]], 1); break;
      case JO_SENS_FRAMEZAXIS_SITE: col(o, d->xmat[m->site_body[obj]], 2); break;
      case JO_SENS_FRAMEPOS_BODY: /* obj2 >= 0: in the frame of that reference site, as above */
        copy3(o, d->xpos[obj]);
        if (m->sensor_obj2[s] >= 0) {
          int rs = m->sensor_obj2[s]; const double* R = d->xmat[m->site_body[rs]];
          double dv[3] = {o[0] - d->site_xpos[rs][0], o[1] - d->site_xpos[rs][1], o[2] - d->site_xpos[rs][2]};
          for (int k = 0; k < 3; k++) o[k] = R[k] * dv[0] + R[3 + k] * dv[1] + R[6 + k] * dv[2];
        }
        break;
      case JO_SENS_FRAMEQUAT_BODY: {
        if (m->sensor_obj2[s] >= 0) { const double* qr = d->xquat[m->sensor_obj2[s]]; double qc[4] = {qr[0], -qr[1], -qr[2], -qr[3]}; quat_mul(o, qc, d->xquat[obj]); }
        else memcpy(o, d->xquat[obj], 4 * sizeof(double));
      } break;
      case JO_SENS_JOINTPOS: o[0] = d->qpos[m->jnt_qposadr[obj]]; break;
      case JO_SENS_FRAMEZAXIS_BODY: col(o, d->xmat[obj], 2); break;
      case JO_SENS_DISTANCE: { /* min over the box geoms of body obj and body obj2, clipped at the cutoff */
        double best = m->sensor_cutoff[s]; int b2 = m->sensor_obj2[s];
        for (int g1 = 0; g1 < m->ngeom; g1++) if (m->geom_body[g1] == obj && m->geom_type[g1] == JO_GEOM_BOX)
          for (int g2 = 0; g2 < m->ngeom; g2++) if (m->geom_body[g2] == b2 && m->geom_type[g2] == JO_GEOM_BOX) {
            double dd = box_box_distance(d->geom_xpos[g1], d->geom_xmat[g1], m->geom_size[g1], d->geom_xpos[g2], d->geom_xmat[g2], m->geom_size[g2]);
            if (dd < best) best = dd;
          }
        o[0] = best;
      } break;
    }
  }
}

/* ------------------------------------------------------------------ forward / step */
void jo_forward(const jo_model* m, jo_data* d) {
  int nv = m->nv;
  kinematics(m, d);
  crb(m, d);
  collision(m, d);
  make_constraint(m, d);
  sensors(m, d);
  /* velocity stage: passive + bias */
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
  rne_bias(m, d);
  /* actuation (mj_fwdActuation): position servo, ctrl clamp, force clamp, joint-level actuator force clamp */
  for (int i = 0; i < nv; i++) d->qfrc_actuator[i] = 0;
  for (int a = 0; a < m->nact; a++) {
    int j = m->act_jnt[a]; double c = d->ctrl[a];
    if (m->act_ctrllimited[a]) c = fmin(m->act_ctrlrange[a][1], fmax(m->act_ctrlrange[a][0], c));
    double f = m->act_kp[a] * (c - d->qpos[m->jnt_qposadr[j]]) - m->act_kv[a] * d->qvel[m->jnt_dofadr[j]];
    if (m->act_forcelimited[a]) f = fmin(m->act_forcerange[a][1], fmax(m->act_forcerange[a][0], f));
    d->act_force[a] = f; d->qfrc_actuator[m->jnt_dofadr[j]] += f;
  }
  for (int i = 0; i < nv; i++) if (m->dof_frclimited[i]) d->qfrc_actuator[i] = fmin(m->dof_frcrange[i][1], fmax(m->dof_frcrange[i][0], d->qfrc_actuator[i]));
  for (int i = 0; i < nv; i++) { d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i]; d->qacc_smooth[i] = d->qfrc_smooth[i]; }
  chol_solve(nv, d->L, d->qacc_smooth);
  solve_constraints(m, d);
}

/* mj_integratePos: qpos advanced by qvel over h (free joints: linear velocity in the world frame, angular velocity in the body frame) */
static void integrate_pos(const jo_model* m, jo_data* d, double h) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JO_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
      double w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]}, ang = norm3(w) * h;
      if (ang > 0) {
        double ax[3] = {w[0] / norm3(w), w[1] / norm3(w), w[2] / norm3(w)}, dq[4], qn[4];
        axisangle2quat(dq, ax, ang); quat_mul(qn, d->qpos + qa + 3, dq); memcpy(d->qpos + qa + 3, qn, sizeof(qn));
      }
      quat_normalize(d->qpos + qa + 3);
    } else d->qpos[qa] += h * d->qvel[da];
  }
}

static void integrate(const jo_model* m, jo_data* d) {
  int nv = m->nv; double h = m->dt;
  double extra[JO_MAXDOF]; int any = 0;
  for (int i = 0; i < nv; i++) { extra[i] = m->dof_damping[i]; if (extra[i] > 0) any = 1; }
  if (m->integrator == JO_INT_IMPLICITFAST) {
    /* d(qfrc_actuator)/d(qvel) = -kv on the actuated dof, skipped while the actuator force sits on its forcerange */
    for (int a = 0; a < m->nact; a++) {
      if (m->act_forcelimited[a] && (d->act_force[a] <= m->act_forcerange[a][0] || d->act_force[a] >= m->act_forcerange[a][1])) continue;
      if (m->act_kv[a] != 0) { extra[m->jnt_dofadr[m->act_jnt[a]]] += m->act_kv[a]; any = 1; }
    }
  }
  double qacc[JO_MAXDOF];
  if (any) {
    static __thread double A[JO_MAXDOF][JO_MAXDOF], LA[JO_MAXDOF][JO_MAXDOF];
    for (int i = 0; i < nv; i++) { for (int j = 0; j < nv; j++) A[i][j] = d->M[i][j]; A[i][i] += h * extra[i]; qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    cholesky(nv, A, LA); chol_solve(nv, LA, qacc);
  } else memcpy(qacc, d->qacc, sizeof(double) * nv);
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  integrate_pos(m, d, h); /* mj_integratePos with the updated velocity */
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
  for (int i = 0; i < nv; i++) d->qacc_con_prev[i] = d->qacc[i] - d->qacc_smooth[i];
}

void jo_step(const jo_model* m, jo_data* d) { jo_forward(m, d); integrate(m, d); }

void jo_mass_matrix(const jo_model* m, jo_data* d, double* M_out) {
  kinematics(m, d); crb(m, d);
  for (int i = 0; i < m->nv; i++) for (int j = 0; j < m->nv; j++) M_out[i * m->nv + j] = d->M[i][j];
}

double jo_energy(const jo_model* m, jo_data* d, double* kinetic, double* potential) {
  kinematics(m, d); crb(m, d);
  double ke = 0, pe = 0;
  for (int i = 0; i < m->nv; i++) for (int j = 0; j < m->nv; j++) ke += 0.5 * d->qvel[i] * d->M[i][j] * d->qvel[j];
  for (int b = 1; b < m->nbody; b++) pe -= m->body_mass[b] * dot3(m->grav, d->xipos[b]);
  if (kinetic) *kinetic = ke; if (potential) *potential = pe;
  return ke + pe;
}

int jo_forward_probe(const jo_model* m, const double* qpos, const double* qvel, const double* ctrl, double* qacc, double* qacc_smooth, double* qfrc_bias,
                     double* qfrc_constraint, double* sensordata, int* info, double* contacts, double* stats) {
  jo_data* d = jo_data_new();
  memcpy(d->qpos, qpos, sizeof(double) * m->nq); memcpy(d->qvel, qvel, sizeof(double) * m->nv); if (m->nact) memcpy(d->ctrl, ctrl, sizeof(double) * m->nact);
  jo_forward(m, d);
  if (qacc) memcpy(qacc, d->qacc, sizeof(double) * m->nv);
  if (qacc_smooth) memcpy(qacc_smooth, d->qacc_smooth, sizeof(double) * m->nv);
  if (qfrc_bias) memcpy(qfrc_bias, d->qfrc_bias, sizeof(double) * m->nv);
  if (qfrc_constraint) memcpy(qfrc_constraint, d->qfrc_constraint, sizeof(double) * m->nv);
  if (sensordata) memcpy(sensordata, d->sensordata, sizeof(double) * m->nsensordata);
  if (info) { info[0] = d->ncon; info[1] = d->nefc; info[2] = d->solver_iter; info[3] = d->con_overflow; }
  if (contacts) for (int c = 0; c < d->ncon; c++) { double* o = contacts + 16 * c; o[0] = d->con[c].dist; copy3(o + 1, d->con[c].pos); memcpy(o + 4, d->con[c].frame, 9 * sizeof(double)); o[13] = d->con[c].g1; o[14] = d->con[c].g2; o[15] = d->con[c].friction[0]; }
  if (stats) { stats[0] = d->solver_cost; stats[1] = d->solver_gradnorm; double tr = 0; for (int i = 0; i < m->nv; i++) tr += d->M[i][i]; stats[2] = tr; }
  int n = d->ncon;
  jo_data_free(d);
  return n;
}

/* test hook: kinematics + collision only, over a batch of configurations: how many contacts each candidate pair (in jo_add_pair order) produced in total.
 * (tests: "the pairs the kernel leaves out never touch on this workload" without paying for the solver) */
void jo_pair_contact_counts(const jo_model* m, const double* qpos, int N, long* counts /* npair */) {
  jo_data* d = jo_data_new();
  for (int p = 0; p < m->npair; p++) counts[p] = 0;
  for (int n = 0; n < N; n++) {
    memcpy(d->qpos, qpos + (size_t)n * m->nq, sizeof(double) * m->nq);
    kinematics(m, d);
    collision(m, d);
    for (int c = 0; c < d->ncon; c++) {
      int a = d->con[c].g1, b = d->con[c].g2;
      for (int p = 0; p < m->npair; p++)
        if ((m->pair_g1[p] == a && m->pair_g2[p] == b) || (m->pair_g1[p] == b && m->pair_g2[p] == a)) { counts[p]++; break; }
    }
  }
  jo_data_free(d);
}

/* test hooks: kinematics only.  World pose of a body at qpos; mj_integratePos of qpos by dq (nv) over a unit time. */
void jo_body_pose(const jo_model* m, const double* qpos, int body, double* pos, double* mat) {
  jo_data* d = jo_data_new();
  memcpy(d->qpos, qpos, sizeof(double) * m->nq);
  kinematics(m, d);
  copy3(pos, d->xpos[body]); memcpy(mat, d->xmat[body], 9 * sizeof(double));
  jo_data_free(d);
}
void jo_integrate_pos(const jo_model* m, const double* qpos, const double* dq, double* out) {
  jo_data* d = jo_data_new();
  memcpy(d->qpos, qpos, sizeof(double) * m->nq); memcpy(d->qvel, dq, sizeof(double) * m->nv);
  integrate_pos(m, d, 1.0);
  memcpy(out, d->qpos, sizeof(double) * m->nq);
  jo_data_free(d);
}

/* The assembled constraint problem of one forward pass, for checks that do not share a line with solve_constraints (tests/test_oracle_independent.py builds
 * MuJoCo's documented primal objective from these arrays in numpy and minimises it with a generic solver): M (nv*nv), qacc_smooth (nv), then per row the
 * Jacobian (nefc*nv), aref, R, frictionloss, type, contact id; per contact the first row, dim, regularised mu and the friction coefficients (5).  dims = {nefc, ncon,
 * cone}.  Returns nefc, or -1 when a buffer is too small. */
int jo_export_problem(const jo_model* m, const double* qpos, const double* qvel, const double* ctrl, const double* qacc_warmstart, int max_efc, int max_con,
                      double* M_out, double* qacc_smooth, double* J, double* aref, double* R, double* frictionloss, int* type, int* id,
                      int* con_adr, int* con_dim, double* con_mu, double* con_friction, double* qacc, int* dims) {
  jo_data* d = jo_data_new();
  memcpy(d->qpos, qpos, sizeof(double) * m->nq); memcpy(d->qvel, qvel, sizeof(double) * m->nv); if (m->nact) memcpy(d->ctrl, ctrl, sizeof(double) * m->nact);
  if (qacc_warmstart) memcpy(d->qacc_warmstart, qacc_warmstart, sizeof(double) * m->nv);
  jo_forward(m, d);
  int nv = m->nv, ne = d->nefc, rc = ne;
  dims[0] = ne; dims[1] = d->ncon; dims[2] = m->cone;
  if (ne > max_efc || d->ncon > max_con) rc = -1;
  else {
    for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) M_out[i * nv + j] = d->M[i][j];
    memcpy(qacc_smooth, d->qacc_smooth, sizeof(double) * nv); memcpy(qacc, d->qacc, sizeof(double) * nv);
    for (int r = 0; r < ne; r++) {
      memcpy(J + (size_t)r * nv, d->efc_J[r], sizeof(double) * nv);
      aref[r] = d->efc_aref[r]; R[r] = d->efc_R[r]; frictionloss[r] = d->efc_frictionloss[r]; type[r] = d->efc_type[r]; id[r] = d->efc_id[r];
    }
    for (int c = 0; c < d->ncon; c++) { con_adr[c] = d->con[c].efc_adr; con_dim[c] = d->con[c].dim; con_mu[c] = d->con[c].mu; memcpy(con_friction + 5 * c, d->con[c].friction, 5 * sizeof(double)); }
  }
  jo_data_free(d);
  return rc;
}

/* ------------------------------------------------------------------ finalize: qpos0, dof tree, inverse weights at qpos0 (mjModel "set0") */
int jo_model_finalize(jo_model* m) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], b = m->jnt_body[j];
    if (m->jnt_type[j] == JO_JNT_FREE) { copy3(m->qpos0 + qa, m->body_pos[b]); memcpy(m->qpos0 + qa + 3, m->body_quat[b], 4 * sizeof(double)); }
    else m->qpos0[qa] = 0; /* joint `ref` is 0 in all four models */
  }
  /* dof_parent: previous dof on the kinematic chain */
  for (int i = 0; i < m->nv; i++) {
    int j = m->dof_jnt[i], b = m->dof_body[i];
    if (i > m->jnt_dofadr[j]) { m->dof_parent[i] = i - 1; continue; }                 /* inside a free joint */
    if (j > m->body_jntadr[b]) { int jp = j - 1; m->dof_parent[i] = m->jnt_dofadr[jp] + (m->jnt_type[jp] == JO_JNT_FREE ? 5 : 0); continue; }
    int p = m->body_parent[b]; m->dof_parent[i] = -1;
    while (p > 0) {
      if (m->body_jntnum[p] > 0) { int jp = m->body_jntadr[p] + m->body_jntnum[p] - 1; m->dof_parent[i] = m->jnt_dofadr[jp] + (m->jnt_type[jp] == JO_JNT_FREE ? 5 : 0); break; }
      p = m->body_parent[p];
    }
  }
  jo_data* d = jo_data_new();
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  kinematics(m, d); crb(m, d);
  int nv = m->nv;
  static double Minv[JO_MAXDOF][JO_MAXDOF];
  for (int i = 0; i < nv; i++) { double e[JO_MAXDOF] = {0}; e[i] = 1; chol_solve(nv, d->L, e); for (int j = 0; j < nv; j++) Minv[j][i] = e[j]; }
  for (int i = 0; i < nv; i++) m->dof_invweight0[i] = Minv[i][i];
  for (int j = 0; j < m->njnt; j++) if (m->jnt_type[j] == JO_JNT_FREE) { /* average translational / rotational triplets */
    int da = m->jnt_dofadr[j];
    double t = (Minv[da][da] + Minv[da + 1][da + 1] + Minv[da + 2][da + 2]) / 3, r = (Minv[da + 3][da + 3] + Minv[da + 4][da + 4] + Minv[da + 5][da + 5]) / 3;
    for (int k = 0; k < 3; k++) { m->dof_invweight0[da + k] = t; m->dof_invweight0[da + 3 + k] = r; }
  }
  for (int b = 1; b < m->nbody; b++) { /* tr(J Minv J')/3 at the body's centre of mass, translational and rotational */
    static double Jp[3][JO_MAXDOF], Jr[3][JO_MAXDOF];
    point_jac(m, d, b, d->xipos[b], Jp);
    for (int k = 0; k < 3; k++) for (int i = 0; i < nv; i++) Jr[k][i] = 0;
    { int bb = b; while (bb > 0 && m->body_jntnum[bb] == 0) bb = m->body_parent[bb];
      if (bb > 0) { int jl = m->body_jntadr[bb] + m->body_jntnum[bb] - 1; for (int i = m->jnt_dofadr[jl] + (m->jnt_type[jl] == JO_JNT_FREE ? 5 : 0); i >= 0; i = m->dof_parent[i]) for (int k = 0; k < 3; k++) Jr[k][i] = d->S[i][k]; } }
    double tt = 0, rr = 0;
    for (int k = 0; k < 3; k++) for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) { tt += Jp[k][i] * Minv[i][j] * Jp[k][j]; rr += Jr[k][i] * Minv[i][j] * Jr[k][j]; }
    m->body_invweight0[b][0] = tt / 3; m->body_invweight0[b][1] = rr / 3;
  }
  jo_data_free(d);
  m->finalized = 1;
  return 0;
}

/* ------------------------------------------------------------------ rollouts */
void jo_rollout(const jo_model* m, jo_data* d, const double* x0, const double* controls, int H, double* states, double* sensors_out) {
  int nq = m->nq, nv = m->nv, nu = m->nact, ns = m->nsensordata;
  memcpy(d->qpos, x0, sizeof(double) * nq); memcpy(d->qvel, x0 + nq, sizeof(double) * nv);
  memset(d->qacc_warmstart, 0, sizeof(double) * nv); memset(d->qacc_con_prev, 0, sizeof(double) * nv);
  for (int t = 0; t < H; t++) {
    memcpy(d->ctrl, controls + (size_t)t * nu, sizeof(double) * nu);
    jo_step(m, d);
    if (states) { memcpy(states + (size_t)t * (nq + nv), d->qpos, sizeof(double) * nq); memcpy(states + (size_t)t * (nq + nv) + nq, d->qvel, sizeof(double) * nv); }
    if (sensors_out) memcpy(sensors_out + (size_t)t * ns, d->sensordata, sizeof(double) * ns);
  }
}

typedef struct { const jo_model* m; const double* x0; int x0_batched; const double* controls; int N, H, tid, nthread; double* states; double* sensors; } batch_arg;
static void* batch_worker(void* p) {
  batch_arg* a = (batch_arg*)p; const jo_model* m = a->m;
  jo_data* d = jo_data_new();
  int nx = m->nq + m->nv;
  for (int n = a->tid; n < a->N; n += a->nthread)
    jo_rollout(m, d, a->x0 + (a->x0_batched ? (size_t)n * nx : 0), a->controls + (size_t)n * a->H * m->nact, a->H,
               a->states ? a->states + (size_t)n * a->H * nx : NULL, a->sensors ? a->sensors + (size_t)n * a->H * m->nsensordata : NULL);
  jo_data_free(d);
  return NULL;
}
void jo_rollout_batch(const jo_model* m, const double* x0, int x0_batched, const double* controls, int N, int H, double* states, double* sensors_out, int nthread) {
  if (nthread < 1) nthread = 1; if (nthread > 256) nthread = 256; if (nthread > N) nthread = N;
  pthread_t th[256]; batch_arg args[256];
  for (int t = 0; t < nthread; t++) { args[t] = (batch_arg){m, x0, x0_batched, controls, N, H, t, nthread, states, sensors_out}; pthread_create(&th[t], NULL, batch_worker, &args[t]); }
  for (int t = 0; t < nthread; t++) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------ debugging aid for the tests: cost / gradient / Hessian at a given acceleration */
double jo_debug_cost(const jo_model* m, const double* qpos, const double* qvel, const double* ctrl, const double* a, double* grad, double* Hout) {
  jo_data* d = jo_data_new();
  memcpy(d->qpos, qpos, sizeof(double) * m->nq); memcpy(d->qvel, qvel, sizeof(double) * m->nv); if (m->nact) memcpy(d->ctrl, ctrl, sizeof(double) * m->nact);
  jo_forward(m, d);
  static __thread double jar[JO_MAXEFC], Hd[JO_MAXEFC], Hc[JO_MAXCON][9]; static __thread int cz[JO_MAXCON];
  int nv = m->nv;
  double cost = total_cost(m, d, a, grad, jar, Hd, Hc, cz);
  if (Hout) {
    for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) Hout[i * nv + j] = d->M[i][j];
    for (int r = 0; r < d->nefc; r++) if (Hd[r] != 0) for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) Hout[i * nv + j] += Hd[r] * d->efc_J[r][i] * d->efc_J[r][j];
    for (int c = 0; c < d->ncon; c++) if (d->con[c].efc_adr >= 0 && m->cone == JO_CONE_ELLIPTIC && d->con[c].dim == 3 && cz[c]) {
      int r0 = d->con[c].efc_adr;
      for (int u = 0; u < 3; u++) for (int v = 0; v < 3; v++) for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) Hout[i * nv + j] += Hc[c][3 * u + v] * d->efc_J[r0 + u][i] * d->efc_J[r0 + v][j];
    }
  }
  jo_data_free(d);
  return cost;
}
