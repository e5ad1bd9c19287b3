/*
 * orpm_problems.c — CPU ORACLE user callbacks (test infrastructure; PARITY UNPINNED, see orpm.h).
 *
 * Vectorised (all-nodes-at-once) restatements of the reference's example problem
 * classes, operation order kept, plus the three problems BASELINE.json names that the
 * reference does not ship (brachistochrone, minimum-time-to-climb, quadrotor): those are
 * authored here and their model equations are stated in DESIGN.md §Problems.
 * Paths are relative to /root/reference/Lpopc.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#include "orpm.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define COL(a, j, N) ((a) + (size_t)(j) * (N))

/* Armadillo 5.300.4 op_dot::direct_dot_arma for 3-vectors: (a0 b0 + a2 b2) + a1 b1 */
static double dot3(const double* a, const double* b) {
  double v1 = 0.0, v2 = 0.0;
  v1 += a[0] * b[0];
  v2 += a[1] * b[1];
  v1 += a[2] * b[2];
  return v1 + v2;
}
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* ===========================================================================
 * Delta-III launch, example/launch/Launch.cpp.  consts = the CONSTANTS struct (:47-74):
 * [0..8] omega_matrix (column-major 3x3), 9 mu, 10 cd, 11 sa, 12 rho0, 13 H, 14 Re, 15 g0,
 * 16 thrust_srb, 17 thrust_first, 18 thrust_second, 19 ISP_srb, 20 ISP_first, 21 ISP_second
 * ======================================================================== */
enum { LC_MU = 9, LC_CD, LC_SA, LC_RHO0, LC_H, LC_RE, LC_G0, LC_TSRB, LC_TFIRST, LC_TSECOND, LC_ISRB, LC_IFIRST, LC_ISECOND };

/* LaunchFunction::DaeFunction, example/launch/Launch.cpp:660-738 */
static void launch_dae(const orpm_soldae* s, const double* c, double* stateout, double* pathout) {
  int N = s->N;
  for (int k = 0; k < N; k++) {
    double r[3], v[3], u[3], m;
    for (int j = 0; j < 3; j++) {
      r[j] = COL(s->state, j, N)[k];
      v[j] = COL(s->state, 3 + j, N)[k];
      u[j] = COL(s->control, j, N)[k];
    }
    m = COL(s->state, 6, N)[k];
    double rad = sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);     /* :670 */
    double vrel[3];
    for (int cc = 0; cc < 3; cc++) {                                  /* :672-673: r*trans(omega) */
      double ocr = 0.0;
      for (int j = 0; j < 3; j++) ocr += r[j] * c[cc + 3 * j];
      vrel[cc] = v[cc] - ocr;
    }
    double speedrel = sqrt((vrel[0] * vrel[0] + vrel[1] * vrel[1]) + vrel[2] * vrel[2]); /* :674 */
    double altitude = rad - c[LC_RE];
    double ret = -altitude / c[LC_H];
    double rho = exp(ret) * c[LC_RHO0];
    double bc = rho / (m * 2) * (c[LC_SA] * c[LC_CD]);                /* :678 */
    double bcspeed = bc * speedrel;
    double mu3 = (1.0 * c[LC_MU]) / (pow(rad, 3));                    /* :683-684 */
    double T_tot, mdot;
    if (s->phase_num == 1) {                                          /* :688-699 */
      double T_srb = 1.0 * (6 * c[LC_TSRB]);
      double T_first = 1.0 * (c[LC_TFIRST]);
      T_tot = T_srb + T_first;
      double m1dot = 0.0, m2dot = 0.0;
      m1dot -= T_srb / (c[LC_G0] * c[LC_ISRB]);
      m2dot -= T_first / (c[LC_G0] * c[LC_IFIRST]);
      mdot = m1dot + m2dot;
    } else if (s->phase_num == 2) {                                   /* :700-711 */
      double T_srb = 1.0 * (3 * c[LC_TSRB]);
      double T_first = 1.0 * (c[LC_TFIRST]);
      T_tot = T_srb + T_first;
      double m1dot = 0.0, m2dot = 0.0;
      m1dot -= T_srb / (c[LC_G0] * c[LC_ISRB]);
      m2dot -= T_first / (c[LC_G0] * c[LC_IFIRST]);
      mdot = m1dot + m2dot;
    } else if (s->phase_num == 3) {                                   /* :712-717 */
      double T_first = 1.0 * c[LC_TFIRST];
      T_tot = T_first;
      mdot = 0.0;
      mdot -= T_first / (c[LC_G0] * c[LC_IFIRST]);
    } else {                                                          /* :718-724 */
      double T_second = 1.0 * c[LC_TSECOND];
      T_tot = T_second;
      mdot = 0.0;
      mdot -= T_second / (c[LC_G0] * c[LC_ISECOND]);
    }
    COL(pathout, 0, N)[k] = (u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]; /* :726 */
    double Toverm = T_tot / m;
    for (int j = 0; j < 3; j++) {
      double Drag = (bcspeed * (-1.0)) * vrel[j];                     /* :681-682 */
      double grav = (-mu3) * r[j];                                    /* :686 */
      double thrust = Toverm * u[j];
      COL(stateout, j, N)[k] = v[j];                                  /* rdot */
      COL(stateout, 3 + j, N)[k] = (thrust + Drag) + grav;            /* :733 */
    }
    COL(stateout, 6, N)[k] = mdot;
  }
}

/* Launchrv2oe, example/launch/Launch.cpp:592-634 */
static void launch_rv2oe(const double* rv, const double* vv, double mu, double* oe) {
  double K[3] = {0.0, 0.0, 1.0};
  double hv[3], nv[3], ev[3];
  cross3(rv, vv, hv);
  cross3(K, hv, nv);
  double n = sqrt(dot3(nv, nv));
  double h2 = dot3(hv, hv);
  double v2 = dot3(vv, vv);
  double r = sqrt(dot3(rv, rv));
  double s1 = v2 - mu / r, s2 = dot3(rv, vv);
  for (int j = 0; j < 3; j++) ev[j] = rv[j] * s1 - vv[j] * s2;
  for (int j = 0; j < 3; j++) ev[j] *= (1.0 / mu);
  double p = h2 / mu;
  double e = sqrt(dot3(ev, ev));
  double a = p / (1 - e * e);
  double i = acos(hv[2] / sqrt(h2));
  double Om1 = acos(nv[0] / n);
  double eps = 2.220446049250313e-16;
  if (nv[1] < 0 - eps) Om1 = 2 * M_PI - Om1;
  double Om2 = acos(dot3(nv, ev) / n / e);
  if (ev[2] < 0) Om2 = 2 * M_PI - Om2;
  oe[0] = a;
  oe[1] = e;
  oe[2] = i;
  oe[3] = Om1;
  oe[4] = Om2;
}
/* LaunchFunction::EventFunction :744-754 */
static void launch_event(const orpm_solevent* s, const double* c, double* ev) {
  if (s->phase_num == 4) launch_rv2oe(s->terminal_state, s->terminal_state + 3, c[LC_MU], ev);
}
/* LaunchFunction::MayerCost :636-646 / LagrangeCost :652-656 / LinkFunction :760-765 */
static void launch_mayer(const orpm_solcost* s, const double* c, double* mayer) {
  (void)c;
  *mayer = (s->phase_num == 4) ? -s->terminal_state[6] : 0.0;
}
static void zero_lagrange(const orpm_solcost* s, const double* c, double* L) {
  (void)c;
  for (int k = 0; k < s->N; k++) L[k] = 0.0;
}
static void diff_link(const orpm_sollink* s, const double* c, double* lo) {
  (void)c;
  for (int j = 0; j < s->nlink; j++) lo[j] = s->right_state[j] - s->left_state[j];
}

/* ===========================================================================
 * Hypersensitive, example/hypersensitive/HyperSensitive.cpp:74-167
 * ======================================================================== */
static void hyper_dae(const orpm_soldae* s, const double* c, double* so, double* po) {
  (void)c;
  (void)po;
  for (int k = 0; k < s->N; k++) {
    double x = s->state[k], u = s->control[k];
    so[k] = ((-x) * x) * x + u; /* -x%x%x + u, :131 */
  }
}
static void hyper_deriv_dae(const orpm_soldae* s, const double* c, double* ds, double* dp) {
  (void)c;
  (void)dp;
  int N = s->N;
  for (int k = 0; k < N; k++) {
    double x = s->state[k];
    COL(ds, 0, N)[k] = -3 * (x * x); /* :142 */
    COL(ds, 1, N)[k] = 1.0;
    COL(ds, 2, N)[k] = 0.0;
  }
}
static void hyper_mayer(const orpm_solcost* s, const double* c, double* m) {
  (void)s;
  (void)c;
  *m = 0.0;
}
static void hyper_deriv_mayer(const orpm_solcost* s, const double* c, double* d) {
  (void)c;
  for (int j = 0; j < 2 * s->nx + 2; j++) d[j] = 0.0;
}
static void hyper_lagrange(const orpm_solcost* s, const double* c, double* L) {
  (void)c;
  for (int k = 0; k < s->N; k++) {
    double x = s->state[k], u = s->control[k];
    L[k] = 0.5 * (x * x + u * u); /* :107 */
  }
}
static void hyper_deriv_lagrange(const orpm_solcost* s, const double* c, double* d) {
  (void)c;
  int N = s->N;
  for (int k = 0; k < N; k++) {
    COL(d, 0, N)[k] = s->state[k]; /* :119-120 */
    COL(d, 1, N)[k] = s->control[k];
    COL(d, 2, N)[k] = 0.0;
  }
}
static void no_event(const orpm_solevent* s, const double* c, double* e) {
  (void)s;
  (void)c;
  (void)e;
}
static void no_link(const orpm_sollink* s, const double* c, double* l) {
  (void)s;
  (void)c;
  (void)l;
}

/* ===========================================================================
 * Bryson-Denham, example/bryson-denham/BrysonDenham.cpp:100-167
 * ======================================================================== */
static void bd_dae(const orpm_soldae* s, const double* c, double* so, double* po) {
  (void)c;
  (void)po;
  int N = s->N;
  for (int k = 0; k < N; k++) {
    double u = s->control[k];
    COL(so, 0, N)[k] = COL(s->state, 1, N)[k];
    COL(so, 1, N)[k] = u;
    COL(so, 2, N)[k] = 0.5 * (u * u);
  }
}
static void bd_mayer(const orpm_solcost* s, const double* c, double* m) {
  (void)c;
  *m = s->terminal_state[2];
}
static void bd_event(const orpm_solevent* s, const double* c, double* e) {
  (void)c;
  e[0] = s->initial_state[0];
  e[1] = s->initial_state[1];
  e[2] = s->initial_state[2];
  e[3] = s->terminal_state[0];
  e[4] = s->terminal_state[1];
}

/* ===========================================================================
 * Brachistochrone (authored; BASELINE config 1).  nx=3 (x,y,v), nu=1 (theta), ne=5.
 * consts[0] = g.   xdot = v sin(th), ydot = v cos(th), vdot = g cos(th); cost = tf.
 * ======================================================================== */
static void brach_dae(const orpm_soldae* s, const double* c, double* so, double* po) {
  (void)po;
  int N = s->N;
  for (int k = 0; k < N; k++) {
    double v = COL(s->state, 2, N)[k], th = s->control[k];
    double sn = sin(th), cs = cos(th);
    COL(so, 0, N)[k] = v * sn;
    COL(so, 1, N)[k] = v * cs;
    COL(so, 2, N)[k] = c[0] * cs;
  }
}
static void brach_deriv_dae(const orpm_soldae* s, const double* c, double* ds, double* dp) {
  (void)dp;
  int N = s->N, nx = 3;
  size_t cs_ = (size_t)N * nx; /* column stride of the stacked matrix */
  memset(ds, 0, sizeof(double) * cs_ * 5);
  for (int k = 0; k < N; k++) {
    double v = COL(s->state, 2, N)[k], th = s->control[k];
    double sn = sin(th), cs = cos(th);
    /* column 2 (d/dv) */
    ds[(k + 0 * (size_t)N) + 2 * cs_] = sn;
    ds[(k + 1 * (size_t)N) + 2 * cs_] = cs;
    /* column 3 (d/dtheta) */
    ds[(k + 0 * (size_t)N) + 3 * cs_] = v * cs;
    ds[(k + 1 * (size_t)N) + 3 * cs_] = -(v * sn);
    ds[(k + 2 * (size_t)N) + 3 * cs_] = -(c[0] * sn);
  }
}
static void tf_mayer(const orpm_solcost* s, const double* c, double* m) {
  (void)c;
  *m = s->terminal_time;
}
static void tf_deriv_mayer(const orpm_solcost* s, const double* c, double* d) {
  (void)c;
  for (int j = 0; j < 2 * s->nx + 2; j++) d[j] = 0.0;
  d[2 * s->nx + 1] = 1.0;
}
static void zero_deriv_lagrange(const orpm_solcost* s, const double* c, double* d) {
  (void)c;
  memset(d, 0, sizeof(double) * (size_t)s->N * (s->nx + s->nu + 1));
}
static void brach_event(const orpm_solevent* s, const double* c, double* e) {
  (void)c;
  e[0] = s->initial_state[0];
  e[1] = s->initial_state[1];
  e[2] = s->initial_state[2];
  e[3] = s->terminal_state[0];
  e[4] = s->terminal_state[1];
}
static void brach_deriv_event(const orpm_solevent* s, const double* c, double* d) {
  (void)c;
  int ne = 5, nx = s->nx;
  memset(d, 0, sizeof(double) * (size_t)ne * (2 * nx + 2));
  d[0 + (size_t)0 * ne] = 1.0;            /* e0 / x0_0 */
  d[1 + (size_t)1 * ne] = 1.0;            /* e1 / x0_1 */
  d[2 + (size_t)2 * ne] = 1.0;            /* e2 / x0_2 */
  d[3 + (size_t)(nx + 1 + 0) * ne] = 1.0; /* e3 / xf_0 */
  d[4 + (size_t)(nx + 1 + 1) * ne] = 1.0; /* e4 / xf_1 */
}

/* ===========================================================================
 * Minimum time to climb (authored; BASELINE config 2).  nx=4 (h,v,gamma,m), nu=1 (alpha), ne=7.
 * consts: 0 Re, 1 mu, 2 S, 3 g0, 4 Isp, 5 rho0, 6 Hs, 7 a0, 8 a1, 9 Tmax
 * ======================================================================== */
static void climb_dae(const orpm_soldae* s, const double* c, double* so, double* po) {
  (void)po;
  int N = s->N;
  for (int k = 0; k < N; k++) {
    double h = COL(s->state, 0, N)[k], v = COL(s->state, 1, N)[k], gam = COL(s->state, 2, N)[k],
           m = COL(s->state, 3, N)[k], al = s->control[k];
    double r = h + c[0];
    double rho = c[5] * exp(-h / c[6]);
    double as = c[7] - c[8] * h;
    double M = v / as;
    double ch = cosh((M - 1.0) / 0.06);
    double CLa = 3.44 + 1.0 / (ch * ch);
    double CD0 = 0.013 + 0.0144 * (1.0 + tanh((M - 0.98) / 0.06));
    double eta = 0.54 + 0.15 * (1.0 + tanh((M - 0.9) / 0.06));
    double CD = CD0 + eta * CLa * (al * al);
    double CL = CLa * al;
    double q = 0.5 * rho * v * v;
    double D = q * c[2] * CD;
    double Lf = q * c[2] * CL;
    double T = c[9] * pow(rho / c[5], 0.7) * (1.0 + 0.3 * M);
    double sg = sin(gam), cg = cos(gam), sa = sin(al), ca = cos(al);
    COL(so, 0, N)[k] = v * sg;
    COL(so, 1, N)[k] = (T * ca - D) / m - c[1] * sg / (r * r);
    COL(so, 2, N)[k] = (T * sa + Lf) / (m * v) + cg * (v / r - c[1] / (v * (r * r)));
    COL(so, 3, N)[k] = -T / (c[3] * c[4]);
  }
}
static void climb_event(const orpm_solevent* s, const double* c, double* e) {
  (void)c;
  e[0] = s->initial_state[0];
  e[1] = s->initial_state[1];
  e[2] = s->initial_state[2];
  e[3] = s->initial_state[3];
  e[4] = s->terminal_state[0];
  e[5] = s->terminal_state[1];
  e[6] = s->terminal_state[2];
}

/* ===========================================================================
 * Quadrotor (authored; BASELINE config 5).  nx=12 (p, v, euler phi/theta/psi, body rates),
 * nu=4 (rotor thrusts).  consts: 0 mass, 1 g, 2 arm, 3 Ixx, 4 Iyy, 5 Izz, 6 ktau,
 * 7..9 pref, 10 wp, 11 wv, 12 wa, 13 ww, 14 wu
 * ======================================================================== */
static void quad_dae(const orpm_soldae* s, const double* c, double* so, double* po) {
  (void)po;
  int N = s->N;
  for (int k = 0; k < N; k++) {
    double x[12], f[4];
    for (int j = 0; j < 12; j++) x[j] = COL(s->state, j, N)[k];
    for (int j = 0; j < 4; j++) f[j] = COL(s->control, j, N)[k];
    double ph = x[6], th = x[7], ps = x[8], p = x[9], q = x[10], r = x[11];
    double F = ((f[0] + f[1]) + f[2]) + f[3];
    double tx = c[2] * (f[1] - f[3]);
    double ty = c[2] * (f[2] - f[0]);
    double tz = c[6] * (((f[0] - f[1]) + f[2]) - f[3]);
    double sph = sin(ph), cph = cos(ph), sth = sin(th), cth = cos(th), sps = sin(ps), cps = cos(ps);
    double b3x = cph * sth * cps + sph * sps;
    double b3y = cph * sth * sps - sph * cps;
    double b3z = cph * cth;
    double Fm = F / c[0];
    COL(so, 0, N)[k] = x[3];
    COL(so, 1, N)[k] = x[4];
    COL(so, 2, N)[k] = x[5];
    COL(so, 3, N)[k] = Fm * b3x;
    COL(so, 4, N)[k] = Fm * b3y;
    COL(so, 5, N)[k] = Fm * b3z - c[1];
    double w = q * sph + r * cph;
    COL(so, 6, N)[k] = p + w * (sth / cth);
    COL(so, 7, N)[k] = q * cph - r * sph;
    COL(so, 8, N)[k] = w / cth;
    COL(so, 9, N)[k] = (tx - (c[5] - c[4]) * q * r) / c[3];
    COL(so, 10, N)[k] = (ty - (c[3] - c[5]) * p * r) / c[4];
    COL(so, 11, N)[k] = (tz - (c[4] - c[3]) * p * q) / c[5];
  }
}
static void quad_lagrange(const orpm_solcost* s, const double* c, double* L) {
  int N = s->N;
  double hov = c[0] * c[1] / 4.0;
  for (int k = 0; k < N; k++) {
    double x[12], f[4];
    for (int j = 0; j < 12; j++) x[j] = COL(s->state, j, N)[k];
    for (int j = 0; j < 4; j++) f[j] = COL(s->control, j, N)[k];
    double d0 = x[0] - c[7], d1 = x[1] - c[8], d2 = x[2] - c[9];
    double ep = (d0 * d0 + d1 * d1) + d2 * d2;
    double evv = (x[3] * x[3] + x[4] * x[4]) + x[5] * x[5];
    double ea = (x[6] * x[6] + x[7] * x[7]) + x[8] * x[8];
    double ew = (x[9] * x[9] + x[10] * x[10]) + x[11] * x[11];
    double u0 = f[0] - hov, u1 = f[1] - hov, u2 = f[2] - hov, u3 = f[3] - hov;
    double eu = ((u0 * u0 + u1 * u1) + u2 * u2) + u3 * u3;
    L[k] = (((c[10] * ep + c[11] * evv) + c[12] * ea) + c[13] * ew) + c[14] * eu;
  }
}
static void zero_mayer(const orpm_solcost* s, const double* c, double* m) {
  (void)s;
  (void)c;
  *m = 0.0;
}

/* ===========================================================================
 * Minimum-time sled with a design parameter (authored; nq = 1).  x' = v, v' = p u, rest to rest over unit distance,
 * cost tf + c0 p^2; with c0 = 0.5 the optimum is p = 1, cost 2.5.  Analytic derivatives given.  consts: 0 c0.
 * ======================================================================== */
static void sled_dae(const orpm_soldae* s, const double* c, double* so, double* po) {
  (void)c;
  (void)po;
  int N = s->N;
  for (int k = 0; k < N; k++) {
    COL(so, 0, N)[k] = COL(s->state, 1, N)[k];
    COL(so, 1, N)[k] = s->parameter[0] * s->control[k];
  }
}
static void sled_event(const orpm_solevent* s, const double* c, double* e) {
  (void)c;
  e[0] = s->initial_state[0];
  e[1] = s->initial_state[1];
  e[2] = s->terminal_state[0];
  e[3] = s->terminal_state[1];
}
static void sled_mayer(const orpm_solcost* s, const double* c, double* m) {
  *m = s->terminal_time + c[0] * (s->parameter[0] * s->parameter[0]);
}
static void sled_deriv_dae(const orpm_soldae* s, const double* c, double* ds, double* dp) {
  (void)c;
  (void)dp;
  int N = s->N, nx = 2;
  size_t cs_ = (size_t)N * nx;
  memset(ds, 0, sizeof(double) * cs_ * 5); /* columns [x, v, u, t, p] */
  for (int k = 0; k < N; k++) {
    ds[(k + 0 * (size_t)N) + 1 * cs_] = 1.0;             /* f0 / v */
    ds[(k + 1 * (size_t)N) + 2 * cs_] = s->parameter[0]; /* f1 / u */
    ds[(k + 1 * (size_t)N) + 4 * cs_] = s->control[k];   /* f1 / p */
  }
}
static void sled_deriv_event(const orpm_solevent* s, const double* c, double* d) {
  (void)c;
  int ne = 4, nx = s->nx;
  memset(d, 0, sizeof(double) * (size_t)ne * (2 * nx + 2 + 1)); /* [x0(2), t0, xf(2), tf, p] */
  d[0 + (size_t)0 * ne] = 1.0;
  d[1 + (size_t)1 * ne] = 1.0;
  d[2 + (size_t)(nx + 1 + 0) * ne] = 1.0;
  d[3 + (size_t)(nx + 1 + 1) * ne] = 1.0;
}
static void sled_deriv_mayer(const orpm_solcost* s, const double* c, double* d) {
  for (int j = 0; j < 2 * s->nx + 2 + 1; j++) d[j] = 0.0;
  d[2 * s->nx + 1] = 1.0;
  d[2 * s->nx + 2] = 2.0 * c[0] * s->parameter[0];
}
static void sled_deriv_lagrange(const orpm_solcost* s, const double* c, double* d) {
  (void)c;
  memset(d, 0, sizeof(double) * (size_t)s->N * (s->nx + s->nu + 1 + s->nq));
}

/* ===========================================================================
 * Damped oscillator with stiffness / weighting parameters over two linked phases (authored; nq = 2 per phase).
 * consts: 0 damping, 1 weight of p0^2.
 * ======================================================================== */
static void posc_dae(const orpm_soldae* s, const double* c, double* so, double* po) {
  int N = s->N;
  for (int k = 0; k < N; k++) {
    double x1 = COL(s->state, 0, N)[k], x2 = COL(s->state, 1, N)[k], u = s->control[k];
    COL(so, 0, N)[k] = x2;
    COL(so, 1, N)[k] = (-(s->parameter[0] * x1) - c[0] * x2) + u;
    po[k] = x1 + s->parameter[1] * u;
  }
}
static void posc_event(const orpm_solevent* s, const double* c, double* e) {
  (void)c;
  if (s->phase_num == 1) {
    e[0] = s->initial_state[0];
    e[1] = s->initial_state[1];
  } else {
    e[0] = s->terminal_state[0] + s->parameter[1];
  }
}
static void posc_link(const orpm_sollink* s, const double* c, double* lo) {
  (void)c;
  lo[0] = s->left_state[0] - s->right_state[0];
  lo[1] = s->left_state[1] - s->right_state[1];
  lo[2] = s->left_parameter[0] - s->right_parameter[0];
  lo[3] = s->left_parameter[1] - s->right_parameter[1];
}
static void posc_mayer(const orpm_solcost* s, const double* c, double* m) {
  (void)c;
  *m = s->phase_num == 2 ? s->terminal_state[0] * s->terminal_state[0] + s->parameter[0] * s->parameter[1] : 0.0;
}
static void posc_lagrange(const orpm_solcost* s, const double* c, double* L) {
  int N = s->N;
  for (int k = 0; k < N; k++) {
    double x1 = COL(s->state, 0, N)[k], u = s->control[k];
    L[k] = (u * u + s->parameter[1] * (x1 * x1)) + c[1] * (s->parameter[0] * s->parameter[0]);
  }
}

/* ------------------------------------------------------------------------- */
static const orpm_functions F_LAUNCH = {launch_mayer, zero_lagrange, launch_dae, launch_event, diff_link,
                                        NULL, NULL, NULL, NULL, NULL};
static const orpm_functions F_HYPER = {hyper_mayer, hyper_lagrange, hyper_dae, no_event, no_link,
                                       hyper_deriv_mayer, hyper_deriv_lagrange, hyper_deriv_dae, NULL, NULL};
static const orpm_functions F_BD = {bd_mayer, zero_lagrange, bd_dae, bd_event, no_link,
                                    NULL, NULL, NULL, NULL, NULL};
static const orpm_functions F_BRACH = {tf_mayer, zero_lagrange, brach_dae, brach_event, no_link,
                                       tf_deriv_mayer, zero_deriv_lagrange, brach_deriv_dae, brach_deriv_event, NULL};
static const orpm_functions F_CLIMB = {tf_mayer, zero_lagrange, climb_dae, climb_event, no_link,
                                       NULL, NULL, NULL, NULL, NULL};
static const orpm_functions F_QUAD = {zero_mayer, quad_lagrange, quad_dae, no_event, no_link,
                                      NULL, NULL, NULL, NULL, NULL};
static const orpm_functions F_SLED = {sled_mayer, zero_lagrange, sled_dae, sled_event, no_link,
                                      sled_deriv_mayer, sled_deriv_lagrange, sled_deriv_dae, sled_deriv_event, NULL};
static const orpm_functions F_POSC = {posc_mayer, posc_lagrange, posc_dae, posc_event, posc_link,
                                      NULL, NULL, NULL, NULL, NULL};

const orpm_functions* orpm_problem_functions(int id) {
  switch (id) {
    case RPM_PROBLEM_LAUNCH: return &F_LAUNCH;
    case RPM_PROBLEM_HYPERSENSITIVE: return &F_HYPER;
    case RPM_PROBLEM_BRYSON_DENHAM: return &F_BD;
    case RPM_PROBLEM_BRACHISTOCHRONE: return &F_BRACH;
    case RPM_PROBLEM_MIN_TIME_CLIMB: return &F_CLIMB;
    case RPM_PROBLEM_QUADROTOR: return &F_QUAD;
    case RPM_PROBLEM_PARAM_SLED: return &F_SLED;
    case RPM_PROBLEM_PARAM_OSC: return &F_POSC;
  }
  return NULL;
}
