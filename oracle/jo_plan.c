/*
 * jo_plan.c -- ORACLE (test infrastructure only): fp64 CPU restatement of the reference's
 * numpy/scipy code on the sample -> clip -> spline -> reward -> update path.  Pinned by the golden
 * vectors in tests/golden/ that tools/gen_golden.py produced from the reference itself.
 *
 * Reference lines restated (relative to /root/reference):
 *   sampling      judo/optimizers/mppi.py:48-59, ps.py:39-50, cem.py:65-74
 *   CEM sigma     judo/optimizers/cem.py:23-27 (init), :44-53 (node-count change), :69-72 (cumulative ramp)
 *   clip          judo/controller/controller.py:253-257
 *   spline        judo/controller/controller.py:382-401 (scipy interp1d zero/linear/cubic, hold-ends),
 *                 :220-221 (time shift), :261-262 (evaluation at t + dt*arange(H))
 *   updates       judo/optimizers/mppi.py:76-82, cem.py:88-92, ps.py:64-65
 *   rewards       judo/tasks/cartpole.py:61-78, cylinder_push.py:65-93, leap_cube.py:63-88
 *                 (+ judo/utils/math_utils.py:6-66,95-104), fr3_pick.py:225-311, cost_functions.py:6-13
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ spline weights */
/* U[h] = sum_k W[h,k] knots[k]: interp1d is linear in the knot values, so the whole evaluation is a (H x K)
 * weight matrix that depends only on (kind, knot times, query times).  Queries outside [t0, t_{K-1}] return the
 * first / last knot (fill_value=(first,last), bounds_error=False). kind: 0 zero, 1 linear, 3 cubic (not-a-knot). */
static void cubic_second_derivs(int K, const double* x, double* A /* K*K, out: inverse-applied basis */, double* Minv_rhs) {
  /* builds the K x K matrix G with m = G y  (m = second derivatives at the knots), not-a-knot end conditions */
  double* T = (double*)calloc((size_t)K * K, sizeof(double));   /* T m = B y */
  double* B = (double*)calloc((size_t)K * K, sizeof(double));
  for (int i = 1; i < K - 1; i++) {
    double h0 = x[i] - x[i - 1], h1 = x[i + 1] - x[i];
    T[i * K + i - 1] = h0; T[i * K + i] = 2 * (h0 + h1); T[i * K + i + 1] = h1;
    B[i * K + i - 1] = 6 / h0; B[i * K + i] = -6 / h0 - 6 / h1; B[i * K + i + 1] = 6 / h1;
  }
  { double h0 = x[1] - x[0], h1 = x[2] - x[1]; T[0] = h1; T[1] = -(h0 + h1); T[2] = h0; }                       /* S''' continuous at x1 */
  { double h0 = x[K - 2] - x[K - 3], h1 = x[K - 1] - x[K - 2]; T[(K - 1) * K + K - 3] = h1; T[(K - 1) * K + K - 2] = -(h0 + h1); T[(K - 1) * K + K - 1] = h0; }
  /* Gauss-Jordan with partial pivoting: G = T^-1 B */
  for (int c = 0; c < K; c++) {
    int piv = c; double mx = fabs(T[c * K + c]);
    for (int r = c + 1; r < K; r++) if (fabs(T[r * K + c]) > mx) { mx = fabs(T[r * K + c]); piv = r; }
    if (piv != c) for (int k = 0; k < K; k++) { double t = T[c * K + k]; T[c * K + k] = T[piv * K + k]; T[piv * K + k] = t; t = B[c * K + k]; B[c * K + k] = B[piv * K + k]; B[piv * K + k] = t; }
    double d = T[c * K + c];
    for (int k = 0; k < K; k++) { T[c * K + k] /= d; B[c * K + k] /= d; }
    for (int r = 0; r < K; r++) if (r != c) { double f = T[r * K + c]; if (f != 0) for (int k = 0; k < K; k++) { T[r * K + k] -= f * T[c * K + k]; B[r * K + k] -= f * B[c * K + k]; } }
  }
  memcpy(A, B, sizeof(double) * K * K);
  (void)Minv_rhs;
  free(T); free(B);
}

int jo_spline_weights(int kind, int K, const double* t, int nq, const double* q, double* W) {
  if (K < 2 || (kind == 3 && K < 4)) return -1;
  double* G = NULL;
  if (kind == 3) { G = (double*)malloc(sizeof(double) * K * K); cubic_second_derivs(K, t, G, NULL); }
  for (int h = 0; h < nq; h++) {
    double* w = W + (size_t)h * K; double x = q[h];
    for (int k = 0; k < K; k++) w[k] = 0;
    if (x < t[0]) { w[0] = 1; continue; }
    if (x > t[K - 1]) { w[K - 1] = 1; continue; }
    int i = 0; /* interval [t_i, t_{i+1}) containing x; x == t_{K-1} belongs to the last interval's right end */
    while (i < K - 2 && x >= t[i + 1]) i++;
    if (kind == 0) { /* previous-knot hold; the value AT knot k is knot k */
      if (x >= t[K - 1]) w[K - 1] = 1; else w[i] = 1;
    } else if (kind == 1) {
      double a = (x - t[i]) / (t[i + 1] - t[i]); w[i] = 1 - a; w[i + 1] = a;
    } else {
      double hh = t[i + 1] - t[i], a = t[i + 1] - x, b = x - t[i];
      double cm0 = a * a * a / (6 * hh) - hh * a / 6, cm1 = b * b * b / (6 * hh) - hh * b / 6;
      for (int k = 0; k < K; k++) w[k] = cm0 * G[i * K + k] + cm1 * G[(i + 1) * K + k];
      w[i] += a / hh; w[i + 1] += b / hh;
    }
  }
  free(G);
  return 0;
}

void jo_spline_eval(const double* W, const double* knots, int N, int H, int K, int nu, double* U) {
  for (int n = 0; n < N; n++)
    for (int h = 0; h < H; h++)
      for (int u = 0; u < nu; u++) {
        double s = 0;
        for (int k = 0; k < K; k++) s += W[h * K + k] * knots[((size_t)n * K + k) * nu + u];
        U[((size_t)n * H + h) * nu + u] = s;
      }
}

/* ------------------------------------------------------------------ sampling */
/* MPPI / PS: sigma_k = noise_ramp * linspace(1/K, 1, K)[k] * sigma if ramp else sigma (mppi.py:52-56, ps.py:43-47) */
void jo_mppi_sigma(double sigma, int use_ramp, double noise_ramp, int K, int nu, double* out) {
  for (int k = 0; k < K; k++) {
    double lin = K > 1 ? 1.0 / K + (1.0 - 1.0 / K) * k / (K - 1) : 1.0 / K;
    double s = use_ramp ? noise_ramp * lin * sigma : sigma;
    for (int u = 0; u < nu; u++) out[k * nu + u] = s;
  }
}
/* CEM: self.sigma = clip(self.sigma * linspace(r/K, r, K)[:,None], smin, smax), applied IN PLACE on every call (cem.py:69-72) */
void jo_cem_sigma_ramp(double* sigma, int use_ramp, double noise_ramp, double smin, double smax, int K, int nu) {
  if (!use_ramp) return;
  for (int k = 0; k < K; k++) {
    double lin = K > 1 ? noise_ramp / K + (noise_ramp - noise_ramp / K) * k / (K - 1) : noise_ramp / K;
    for (int u = 0; u < nu; u++) { double v = sigma[k * nu + u] * lin; sigma[k * nu + u] = v < smin ? smin : (v > smax ? smax : v); }
  }
}
/* CEM node-count change: linear interpolation of sigma over time with linear extrapolation (cem.py:44-53) */
void jo_cem_pre_optimization(const double* sigma_in, int Kold, const double* old_t, int Knew, const double* new_t, int nu, double* sigma_out) {
  for (int k = 0; k < Knew; k++) {
    double x = new_t[k]; int i = 0;
    while (i < Kold - 2 && x >= old_t[i + 1]) i++;
    double a = (x - old_t[i]) / (old_t[i + 1] - old_t[i]);
    for (int u = 0; u < nu; u++) sigma_out[k * nu + u] = sigma_in[i * nu + u] + a * (sigma_in[(i + 1) * nu + u] - sigma_in[i * nu + u]);
  }
}
/* out[0] = nominal; out[1:] = nominal + sigma * noise  (row 0 is the unperturbed nominal) */
void jo_sample_knots(const double* nominal, const double* noise, const double* sigma, int N, int K, int nu, double* out) {
  int KU = K * nu;
  for (int i = 0; i < KU; i++) out[i] = nominal[i];
  for (int n = 1; n < N; n++) for (int i = 0; i < KU; i++) out[(size_t)n * KU + i] = nominal[i] + sigma[i] * noise[(size_t)(n - 1) * KU + i];
}
void jo_clip_knots(double* knots, int N, int K, int nu, const double* lo, const double* hi) {
  for (size_t i = 0; i < (size_t)N * K; i++) for (int u = 0; u < nu; u++) { double v = knots[i * nu + u]; knots[i * nu + u] = v < lo[u] ? lo[u] : (v > hi[u] ? hi[u] : v); }
}

/* ------------------------------------------------------------------ updates */
void jo_mppi_update(const double* knots, const double* rewards, int N, int K, int nu, double temperature, double* out) {
  int KU = K * nu; double beta = 1e300, sum = 0;
  for (int n = 0; n < N; n++) if (-rewards[n] < beta) beta = -rewards[n];
  double* w = (double*)malloc(sizeof(double) * N);
  for (int n = 0; n < N; n++) { w[n] = exp(-(-rewards[n] - beta) / temperature); sum += w[n]; }
  for (int i = 0; i < KU; i++) { double s = 0; for (int n = 0; n < N; n++) s += w[n] / sum * knots[(size_t)n * KU + i]; out[i] = s; }
  free(w);
}
/* elites = flip(argsort(rewards))[:k]: largest rewards first; among exactly equal rewards the higher index first
 * (what a stable ascending sort followed by a flip yields; numpy's introsort leaves ties unspecified) */
void jo_cem_update(const double* knots, const double* rewards, int N, int K, int nu, int k_el, double smin, double smax, double* out, double* sigma_out, int* elite_idx) {
  int KU = K * nu; char* used = (char*)calloc(N, 1);
  for (int e = 0; e < k_el; e++) {
    int best = -1;
    for (int n = 0; n < N; n++) if (!used[n] && (best < 0 || rewards[n] >= rewards[best])) best = n;
    used[best] = 1; elite_idx[e] = best;
  }
  for (int i = 0; i < KU; i++) {
    double mean = 0, var = 0;
    for (int e = 0; e < k_el; e++) mean += knots[(size_t)elite_idx[e] * KU + i];
    mean /= k_el;
    for (int e = 0; e < k_el; e++) { double dv = knots[(size_t)elite_idx[e] * KU + i] - mean; var += dv * dv; }
    var /= k_el; /* population variance (ddof = 0) */
    out[i] = mean; double s = sqrt(var); sigma_out[i] = s < smin ? smin : (s > smax ? smax : s);
  }
  free(used);
}
void jo_ps_update(const double* knots, const double* rewards, int N, int K, int nu, double* out) {
  int best = 0; for (int n = 1; n < N; n++) if (rewards[n] > rewards[best]) best = n; /* first maximum wins (np.argmax) */
  memcpy(out, knots + (size_t)best * K * nu, sizeof(double) * K * nu);
}

/* ------------------------------------------------------------------ rewards */
static double sl1(double z, double p) { return sqrt(z * z + p * p) - p; }
/* w = (w_vertical, w_centered, w_velocity, w_control, p_vertical, p_centered) */
void jo_reward_cartpole(const double* states, const double* controls, int N, int H, const double* w, double* out) {
  for (int n = 0; n < N; n++) {
    double r = 0;
    for (int h = 0; h < H; h++) {
      const double* s = states + ((size_t)n * H + h) * 4; double u = controls[(size_t)n * H + h];
      r -= w[0] * sl1(cos(s[1]) - 1, w[4]) + w[1] * sl1(s[0], w[5]) + w[2] * 0.5 * (s[2] * s[2] + s[3] * s[3]) + w[3] * 0.5 * u * u;
    }
    out[n] = r;
  }
}
/* p = (w_pusher_proximity, w_pusher_velocity, w_cart_position, pusher_goal_offset, goal_x, goal_y); state = [pusher xy, cart xy, pusher v, cart v] */
void jo_reward_cylinder(const double* states, int N, int H, const double* p, double* out) {
  for (int n = 0; n < N; n++) {
    double r = 0;
    for (int h = 0; h < H; h++) {
      const double* s = states + ((size_t)n * H + h) * 8;
      double gx = p[4] - s[2], gy = p[5] - s[3], gn = sqrt(gx * gx + gy * gy);
      double px = s[2] - p[3] * gx / gn, py = s[3] - p[3] * gy / gn; /* no epsilon guard, as in the reference */
      double dx = s[0] - px, dy = s[1] - py;
      r -= p[0] * 0.5 * (dx * dx + dy * dy) + p[1] * 0.5 * (s[4] * s[4] + s[5] * s[5]) + p[2] * 0.5 * (gx * gx + gy * gy);
    }
    out[n] = r;
  }
}
/* p = (w_pos, w_rot, goal_pos[3], goal_quat[4]); only state columns 0:7 are read; MEAN over time */
void jo_reward_leap(const double* states, int N, int H, int nx, const double* p, double* out) {
  const double* gp = p + 2; const double* v = p + 5;
  for (int n = 0; n < N; n++) {
    double pc = 0, rc = 0;
    for (int h = 0; h < H; h++) {
      const double* s = states + ((size_t)n * H + h) * nx;
      double d0 = s[0] - gp[0], d1 = s[1] - gp[1], d2 = s[2] - gp[2];
      pc += d0 * d0 + d1 * d1 + d2 * d2;
      /* diff = conj(u) (x) v with u = state quaternion */
      double u0 = s[3], u1 = -s[4], u2 = -s[5], u3 = -s[6];
      double w = u0 * v[0] - u1 * v[1] - u2 * v[2] - u3 * v[3];
      double x = u0 * v[1] + u1 * v[0] + u2 * v[3] - u3 * v[2];
      double y = u0 * v[2] - u1 * v[3] + u2 * v[0] + u3 * v[1];
      double z = u0 * v[3] + u1 * v[2] - u2 * v[1] + u3 * v[0];
      double sn = sqrt(x * x + y * y + z * z), ax, ay, az;
      if (sn < 1e-6) { ax = 1; ay = 0; az = 0; } else { ax = x / sn; ay = y / sn; az = z / sn; }
      double speed = 2 * atan2(sn, w);
      if (speed > M_PI) speed -= 2 * M_PI;
      rc += (ax * ax + ay * ay + az * az) * speed * speed;
    }
    out[n] = -(p[0] * 0.5 * pc / H + p[1] * 0.5 * rc / H);
  }
}
/* fr3_pick: phase 0 LIFT, 1 MOVE, 2 PLACE, 3 HOMING.
 * p = (w_lift_close, w_lift_height, w_move_goal, w_move_close, w_place_table, w_place_goal, w_upright, w_coll, w_qvel, w_open,
 *      goal_x, goal_y, pick_height, arm_home[9]); sensor addresses sadr = (left_finger_table, right_finger_table, obj_table, grasp_site, ee_z) */
void jo_reward_fr3(const double* states, const double* sensors, int N, int H, int nq, int nv, int ns, int phase, const double* p, const int* sadr, double* out) {
  int nx = nq + nv;
  for (int n = 0; n < N; n++) {
    double r = 0, up = 0, coll = 0, qv = 0, op = 0;
    for (int h = 0; h < H; h++) {
      const double* s = states + ((size_t)n * H + h) * nx; const double* y = sensors + ((size_t)n * H + h) * ns;
      const double* gs = y + sadr[3]; const double* ez = y + sadr[4];
      double gd = (gs[0] - s[0]) * (gs[0] - s[0]) + (gs[1] - s[1]) * (gs[1] - s[1]) + (gs[2] - s[2]) * (gs[2] - s[2]);
      double he = (s[2] - p[12]) * (s[2] - p[12]);
      double og = sqrt((s[0] - p[10]) * (s[0] - p[10]) + (s[1] - p[11]) * (s[1] - p[11]));
      double hd = 0; for (int k = 0; k < 9; k++) hd += (s[7 + k] - p[13 + k]) * (s[7 + k] - p[13 + k]); hd = sqrt(hd);
      if (phase == 0) r -= p[0] * gd + p[1] * he;
      else if (phase == 1) r -= p[2] * og + p[3] * gd;
      else if (phase == 2) r -= p[4] * y[sadr[2]] + p[5] * og;
      else r -= hd;
      up -= sqrt(ez[0] * ez[0] + ez[1] * ez[1] + (ez[2] + 1) * (ez[2] + 1));
      int touching = (y[sadr[0]] <= 0.0) || (y[sadr[1]] <= 0.0);
      coll += 1 - touching;
      double qn = 0; for (int k = 0; k < nv; k++) qn += s[nq + k] * s[nq + k];
      double decay = H > 1 ? 1.0 - (double)h / (H - 1) : 1.0; /* linspace(1, 0, H) */
      qv -= decay * sqrt(qn);
      op -= (s[15] - 0.04) * (s[15] - 0.04);
    }
    out[n] = r + p[6] * up + p[7] * coll + p[8] * qv + p[9] * op;
  }
}
