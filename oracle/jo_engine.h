/*
 * jo_engine.h -- ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * fp64 CPU restatement of the physics step the reference delegates to MuJoCo 3.5.0
 * (`mj_step`, called N x H times per plan from `mujoco.rollout.Rollout.rollout`,
 * judo/utils/mj_rollout_backend.py:84, and from `System::rollout`,
 * mujoco_extensions/system/system_class.cpp:301-303).  MuJoCo is a third-party wheel that is
 * absent from /root/reference and from this image (pyproject.toml:33 `mujoco>=3.5.0,<3.6`,
 * pixi.lock:96 `mujoco-3.5.0`), so this file restates MuJoCo's *published* algorithm
 * (MuJoCo documentation, "Computation" chapter + XML reference for parameter semantics):
 *
 *   forward:  kinematics -> composite-rigid-body mass matrix (+armature) -> collision ->
 *             constraint rows (equality, dof friction loss, joint limits, contacts; soft
 *             constraints parameterised by solref/solimp, R = (1-d)/d * diagApprox) ->
 *             passive (joint damping) + RNE bias (Coriolis, centrifugal, gravity) ->
 *             position-actuator forces (ctrl clamp, force clamp, joint actuatorfrcrange) ->
 *             unconstrained acceleration -> convex constraint solve (primal Newton with exact
 *             line search on  1/2 (a-a0)' M (a-a0) + s(J a - aref)) ;
 *   integrate: Euler with implicit joint damping, or implicitfast (velocity derivatives of
 *             damping + actuator kv folded into the matrix), semi-implicit position update,
 *             quaternion integration for free joints.
 *
 * PARITY UNPINNED at the MuJoCo boundary: the reference's tests pin no rollout value
 *  and MuJoCo cannot be run here; this engine is pinned only by its own
 * known-answer tests (tests/test_oracle_physics.py) and by agreement with the independently
 * written closed-form cartpole / cylinder_push HIP kernels.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 */
#ifndef JO_ENGINE_H
#define JO_ENGINE_H

#ifdef __cplusplus
extern "C" {
#endif

#define JO_MAXBODY 24
#define JO_MAXJNT 24
#define JO_MAXDOF 32
#define JO_MAXQ 34
#define JO_MAXGEOM 96
#define JO_MAXSITE 16
#define JO_MAXACT 24
#define JO_MAXSENSOR 40
#define JO_MAXSENSORDATA 48
#define JO_MAXPAIR 2048
#define JO_MAXEQ 4
#define JO_MAXCON 160
#define JO_MAXEFC 720

enum { JO_JNT_FREE = 0, JO_JNT_SLIDE = 2, JO_JNT_HINGE = 3 };
enum { JO_GEOM_PLANE = 0, JO_GEOM_SPHERE = 2, JO_GEOM_CAPSULE = 3, JO_GEOM_CYLINDER = 5, JO_GEOM_BOX = 6 };
enum { JO_INT_EULER = 0, JO_INT_IMPLICITFAST = 3 };
enum { JO_CONE_PYRAMIDAL = 0, JO_CONE_ELLIPTIC = 1 };
enum { JO_SENS_FRAMEPOS_SITE = 0, JO_SENS_FRAMEPOS_BODY = 1, JO_SENS_JOINTPOS = 2, JO_SENS_FRAMEZAXIS_BODY = 3, JO_SENS_DISTANCE = 4,
       JO_SENS_FRAMEXAXIS_SITE = 5, JO_SENS_FRAMEYAXIS_SITE = 6, JO_SENS_FRAMEZAXIS_SITE = 7,
       JO_SENS_FRAMEQUAT_BODY = 8 /* orientation of body obj relative to body obj2 (or the world): conj(q_ref) * q_obj, mj_sensorPos mjSENS_FRAMEQUAT */ }; /* site axes: sites carry no rotation of their own here (identity site quat) */
enum { JO_EFC_EQUALITY = 0, JO_EFC_FRICTION = 1, JO_EFC_LIMIT = 2, JO_EFC_CONTACT_FRICTIONLESS = 3, JO_EFC_CONTACT_PYRAMIDAL = 4, JO_EFC_CONTACT_ELLIPTIC = 5 };

typedef struct jo_model {
  /* options */
  double dt, impratio, grav[3];
  int integrator, cone, contact_enabled;
  int nbody, njnt, nq, nv, ngeom, nsite, nact, nsensor, nsensordata, npair, neq;
  /* bodies (index 0 = world) */
  int body_parent[JO_MAXBODY], body_jntadr[JO_MAXBODY], body_jntnum[JO_MAXBODY];
  double body_pos[JO_MAXBODY][3], body_quat[JO_MAXBODY][4], body_mass[JO_MAXBODY];
  double body_ipos[JO_MAXBODY][3], body_iquat[JO_MAXBODY][4], body_inertia[JO_MAXBODY][3];
  double body_invweight0[JO_MAXBODY][2];
  /* joints */
  int jnt_type[JO_MAXJNT], jnt_body[JO_MAXJNT], jnt_qposadr[JO_MAXJNT], jnt_dofadr[JO_MAXJNT], jnt_limited[JO_MAXJNT];
  double jnt_pos[JO_MAXJNT][3], jnt_axis[JO_MAXJNT][3], jnt_range[JO_MAXJNT][2], jnt_margin[JO_MAXJNT];
  double jnt_solref[JO_MAXJNT][2], jnt_solimp[JO_MAXJNT][5];
  /* dofs */
  int dof_body[JO_MAXDOF], dof_jnt[JO_MAXDOF], dof_parent[JO_MAXDOF], dof_frclimited[JO_MAXDOF];
  double dof_damping[JO_MAXDOF], dof_armature[JO_MAXDOF], dof_frictionloss[JO_MAXDOF], dof_invweight0[JO_MAXDOF];
  double dof_solref[JO_MAXDOF][2], dof_solimp[JO_MAXDOF][5], dof_frcrange[JO_MAXDOF][2];
  double qpos0[JO_MAXQ];
  /* geoms */
  int geom_type[JO_MAXGEOM], geom_body[JO_MAXGEOM], geom_condim[JO_MAXGEOM], geom_priority[JO_MAXGEOM];
  double geom_size[JO_MAXGEOM][3], geom_pos[JO_MAXGEOM][3], geom_quat[JO_MAXGEOM][4], geom_friction[JO_MAXGEOM][3];
  double geom_solref[JO_MAXGEOM][2], geom_solimp[JO_MAXGEOM][5], geom_margin[JO_MAXGEOM], geom_gap[JO_MAXGEOM], geom_rbound[JO_MAXGEOM];
  int pair_g1[JO_MAXPAIR], pair_g2[JO_MAXPAIR];
  /* sites */
  int site_body[JO_MAXSITE];
  double site_pos[JO_MAXSITE][3];
  /* actuators: position servos on joints (gain kp, bias -kp q - kv qdot) */
  int act_jnt[JO_MAXACT], act_ctrllimited[JO_MAXACT], act_forcelimited[JO_MAXACT];
  double act_kp[JO_MAXACT], act_kv[JO_MAXACT], act_ctrlrange[JO_MAXACT][2], act_forcerange[JO_MAXACT][2];
  /* sensors */
  int sensor_type[JO_MAXSENSOR], sensor_obj[JO_MAXSENSOR], sensor_obj2[JO_MAXSENSOR], sensor_adr[JO_MAXSENSOR];
  double sensor_cutoff[JO_MAXSENSOR];
  /* joint equalities q2 - q2_0 = poly(q1 - q1_0) */
  int eq_j1[JO_MAXEQ], eq_j2[JO_MAXEQ];
  double eq_poly[JO_MAXEQ][5], eq_solref[JO_MAXEQ][2], eq_solimp[JO_MAXEQ][5];
  /* solver controls (MuJoCo defaults: Newton, tolerance 1e-8; the oracle converges tighter) */
  int solver_maxiter;
  double solver_tol;
  int finalized;
} jo_model;

typedef struct jo_contact {
  double dist, pos[3], frame[9]; /* frame rows: normal (geom1 -> geom2), tangent1, tangent2 */
  double friction[5], solref[2], solimp[5], includemargin, mu;
  int g1, g2, dim, efc_adr;
} jo_contact;

/* per-instance working state (one rollout at a time) */
typedef struct jo_data {
  double qpos[JO_MAXQ], qvel[JO_MAXDOF], ctrl[JO_MAXACT];
  double qacc[JO_MAXDOF], qacc_warmstart[JO_MAXDOF], qacc_smooth[JO_MAXDOF];
  double qacc_con_prev[JO_MAXDOF]; /* diagnostics: last step's constraint acceleration qacc - qacc_smooth (third warm-start candidate, off by default) */
  double qfrc_bias[JO_MAXDOF], qfrc_passive[JO_MAXDOF], qfrc_actuator[JO_MAXDOF], qfrc_smooth[JO_MAXDOF], qfrc_constraint[JO_MAXDOF];
  double act_force[JO_MAXACT];
  double xpos[JO_MAXBODY][3], xquat[JO_MAXBODY][4], xmat[JO_MAXBODY][9], xipos[JO_MAXBODY][3], ximat[JO_MAXBODY][9];
  double geom_xpos[JO_MAXGEOM][3], geom_xmat[JO_MAXGEOM][9], site_xpos[JO_MAXSITE][3];
  double S[JO_MAXDOF][6];          /* motion axis of each dof: (angular, linear-at-world-origin) */
  double M[JO_MAXDOF][JO_MAXDOF];  /* joint-space inertia incl. armature */
  double L[JO_MAXDOF][JO_MAXDOF];  /* Cholesky factor of M */
  double sensordata[JO_MAXSENSORDATA];
  int ncon, nefc, solver_iter;
  jo_contact con[JO_MAXCON];
  int efc_type[JO_MAXEFC], efc_id[JO_MAXEFC];
  double efc_J[JO_MAXEFC][JO_MAXDOF], efc_pos[JO_MAXEFC], efc_margin[JO_MAXEFC], efc_vel[JO_MAXEFC], efc_aref[JO_MAXEFC];
  double efc_R[JO_MAXEFC], efc_D[JO_MAXEFC], efc_frictionloss[JO_MAXEFC], efc_diagApprox[JO_MAXEFC], efc_force[JO_MAXEFC];
  double efc_KBIP[JO_MAXEFC][4];
  double solver_cost, solver_gradnorm;
  int con_overflow;
} jo_data;

/* ---- model builder (called from Python via ctypes; returns the new element's index, <0 on error) */
jo_model* jo_model_new(double timestep, int integrator, int cone, double impratio, const double* gravity, int contact_enabled);
void jo_model_free(jo_model* m);
int jo_add_body(jo_model* m, int parent, const double* pos, const double* quat, double mass, const double* ipos, const double* iquat, const double* inertia);
int jo_add_joint(jo_model* m, int body, int type, const double* pos, const double* axis, double damping, double armature,
                 double frictionloss, int limited, const double* range, double margin, int frclimited, const double* frcrange,
                 const double* solref_limit, const double* solimp_limit, const double* solref_fric, const double* solimp_fric);
int jo_add_geom(jo_model* m, int body, int type, const double* size, const double* pos, const double* quat, const double* friction,
                const double* solref, const double* solimp, double margin, double gap, int condim);
int jo_add_pair(jo_model* m, int g1, int g2);
int jo_add_site(jo_model* m, int body, const double* pos);
int jo_add_actuator(jo_model* m, int joint, double kp, double kv, int ctrllimited, const double* ctrlrange, int forcelimited, const double* forcerange);
int jo_add_sensor(jo_model* m, int type, int obj, int obj2, double cutoff);
int jo_add_equality_joint(jo_model* m, int j1, int j2, const double* polycoef, const double* solref, const double* solimp);
int jo_model_finalize(jo_model* m);
int jo_model_dims(const jo_model* m, int* out /* nq nv nu nsensordata nbody ngeom npair */);
void jo_model_get_invweight0(const jo_model* m, double* dof_invweight0, double* body_invweight0);
void jo_model_get_qpos0(const jo_model* m, double* qpos0);

jo_data* jo_data_new(void);
void jo_data_free(jo_data* d);

/* ---- physics */
void jo_forward(const jo_model* m, jo_data* d);  /* everything up to qacc (mj_forward) */
void jo_step(const jo_model* m, jo_data* d);     /* mj_step: forward + integrate */
void jo_mass_matrix(const jo_model* m, jo_data* d, double* M_out /* nv*nv */);
double jo_energy(const jo_model* m, jo_data* d, double* kinetic, double* potential);
/* one forward pass from (qpos,qvel,ctrl); copies out diagnostic vectors (any pointer may be NULL) */
int jo_forward_probe(const jo_model* m, const double* qpos, const double* qvel, const double* ctrl, double* qacc, double* qacc_smooth,
                     double* qfrc_bias, double* qfrc_constraint, double* sensordata, int* ncon_nefc_iter, double* contacts /* ncon*16 */,
                     double* stats /* cost, gradnorm, trace(M) */);

/* test hooks for checks that use the kinematics / one narrow-phase routine in isolation (tests/test_oracle_independent.py) */
void jo_body_pose(const jo_model* m, const double* qpos, int body, double* pos, double* mat);
void jo_pair_contact_counts(const jo_model* m, const double* qpos, int N, long* counts /* npair: contacts per candidate pair over the batch (kinematics + collision only) */);
void jo_integrate_pos(const jo_model* m, const double* qpos, const double* dq, double* out);
int jo_collide_shapes(int t1, const double* s1, const double* p1, const double* q1, int t2, const double* s2, const double* p2, const double* q2, double margin, double* out /* 8 rows of 7 */);

/* the assembled constraint problem of one forward pass (independent-solver checks in tests/): see jo_engine.c */
int jo_export_problem(const jo_model* m, const double* qpos, const double* qvel, const double* ctrl, const double* qacc_warmstart, int max_efc, int max_con,
                      double* M_out, double* qacc_smooth, double* J, double* aref, double* R, double* frictionloss, int* type, int* id,
                      int* con_adr, int* con_dim, double* con_mu, double* con_friction, double* qacc, int* dims);

/* rollout: x0 (nq+nv), controls (H,nu) -> states (H,nq+nv) after each step, sensors (H,ns) as held by the step
 * that produced the state (computed by the forward pass at the START of that step: MuJoCo semantics,
 * judo/utils/mj_rollout_backend.py:84-88). */
void jo_rollout(const jo_model* m, jo_data* d, const double* x0, const double* controls, int H, double* states, double* sensors);
/* N rollouts on `nthread` host threads (the structure of mujoco.rollout's pool and of threadedRollout,
 * mujoco_extensions/system/system_class.cpp:333-367). x0 is (nq+nv) shared or (N,nq+nv) if x0_batched. */
void jo_rollout_batch(const jo_model* m, const double* x0, int x0_batched, const double* controls, int N, int H, double* states,
                      double* sensors, int nthread);

/* diagnostics (tools/): Newton iterations per solve over all threads; solver tolerance / iteration cap override */
void jo_solver_histogram(long* out32, int reset);
void jo_set_solver(jo_model* m, double tol, int maxiter);
/* contact-parameter priority of a geom (mjModel.geom_priority, default 0): the higher priority side supplies friction / solref / solimp / condim */
int jo_set_geom_priority(jo_model* m, int geom, int priority);
#ifdef JO_EXPERIMENTS /* libjudo_oracle_exp.so only (`make -C oracle exp`): the parity oracle carries none of this state */
void jo_set_ls_experiment(int mode, double lstol, int lsmax, int* log, long cap); /* line-search experiments: see jo_engine.c (0 = off, the default) */
long jo_ls_log_size(void);
void jo_set_warmstart_mode(int mode); /* 0 = MuJoCo (better of previous qacc and qacc_smooth); 1 = also try qacc_smooth + previous constraint acceleration */
void jo_set_hessian_reuse(int mode); long jo_hessian_reuse_count(void); long jo_ls_trouble(int reset); void jo_set_ls_kink(double v); void jo_set_ls_shrink(double v);
#endif

#ifdef __cplusplus
}
#endif
#endif
