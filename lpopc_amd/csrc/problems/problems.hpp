// problems.hpp — pointwise device functors that stand where the reference's host-side
// FunctionWrapper subclasses stand (Core/LpFunctionWrapper.h:50-69).  One struct per problem:
//
//   dims     NX, NU, NC, NE_MAX, NLINK_MAX, NCONST, HAS_ANALYTIC  [, NQ: static parameters per phase, default 0]
//   dae      f(t,x,u) and path c(t,x,u) at ONE collocation node  (FunctionWrapper::DaeFunction)
//   event    FunctionWrapper::EventFunction      link   FunctionWrapper::LinkFunction
//   mayer    FunctionWrapper::MayerCost          lagrange  FunctionWrapper::LagrangeCost (one node)
//   dae_jac_col / lagrange_grad_col / ...  one column of the user's analytic derivative
//                                          (FunctionWrapper::Deriv*, used with first-derive=analytic)
//
// A functor with NQ > 0 takes the phase's static parameters (SolDae::parameter_ etc., Core/LpFunctionWrapper.h:12-49) as
// one more argument `p` right before the constants: dae(ph,t,x,u,p,c,f,cp), lagrange(ph,t,x,u,p,c), mayer(ph,t0,x0,tf,xf,p,c),
// event(ph,t0,x0,tf,xf,p,c,ev), link(lph,rph,xl,xr,pl,pr,c,nlink,lo); derivative columns are ordered [x.., u.., t, p..]
// (dae / lagrange, Core/LpFiniteDifferenceDerive.cpp:299-317), [x0.., t0, xf.., tf, p..] (event / mayer, :326-409) and
// [xf_left.., p_left.., x0_right.., p_right..] (link, :411-502).  The kernels call through the pf_* helpers below.
//
// `ph` is the 1-based phase number the reference passes as phase_num_; `c` are the problem
// constants (the reference keeps them in globals).  Operation order follows the reference's
// vectorised expressions so that results agree with lpopc's CPU path to rounding of libm.
// Paths below are relative to /root/reference/Lpopc.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include "../../../include/rpm_hip.h"

namespace rpm {

#define RPM_DEV __device__ __forceinline__

// ---- static parameters: NQ of a functor (0 when it does not declare one) and the calls that pass `p` only to functors
//      that take it ----
template <class P, class = void> struct prob_nq { static constexpr int value = 0; };
template <class P> struct prob_nq<P, decltype(void(P::NQ))> { static constexpr int value = P::NQ; };

template <class P, class CP>
RPM_DEV void pf_dae(int ph, double t, const double* x, const double* u, const double* p, CP c, double* f, double* cp) {
  if constexpr (prob_nq<P>::value > 0) P::dae(ph, t, x, u, p, c, f, cp);
  else P::dae(ph, t, x, u, c, f, cp);
}
// Optional, for the finite-difference kernels that walk several perturbation roles of one node in one thread: a functor may
// offer  struct Stage,  stage(ph, t, x, u, c, Stage&)  — the sub-expressions of dae() at the unperturbed point — and
// dae_from(ph, t, x, u, c, const Stage& base, int var, f, p)  = dae() at a point that differs from the base in variable `var`
// ONLY ([x.., u.., t]; -1: the base itself), which recomputes just what depends on that variable.  Same operations on the same
// operands, so the same bits as dae(): of the metric problem's 13 evaluations per node only 4 need the exponential, the
// cube and the first square root again.  (Problems without static parameters.)
template <class P, class = void>
struct has_stage : std::false_type {};
template <class P>
struct has_stage<P, std::void_t<typename P::Stage>> : std::true_type {};
template <class P, class = void>
struct stage_always : std::false_type {};     // P::STAGE_ALWAYS = true: the staged form also pays when the launch stores everything
template <class P>
struct stage_always<P, std::enable_if_t<P::STAGE_ALWAYS>> : std::true_type {};
template <class P, bool = has_stage<P>::value>
struct stage_of { struct type {}; };
template <class P>
struct stage_of<P, true> { using type = typename P::Stage; };

template <class P, class CP>
RPM_DEV void pf_dae_jac_col(int ph, int v, double t, const double* x, const double* u, const double* p, CP c, double* df, double* dc) {
  if constexpr (prob_nq<P>::value > 0) P::dae_jac_col(ph, v, t, x, u, p, c, df, dc);
  else P::dae_jac_col(ph, v, t, x, u, c, df, dc);
}
template <class P>
RPM_DEV double pf_lagrange(int ph, double t, const double* x, const double* u, const double* p, const double* c) {
  if constexpr (prob_nq<P>::value > 0) return P::lagrange(ph, t, x, u, p, c);
  else return P::lagrange(ph, t, x, u, c);
}
template <class P>
RPM_DEV double pf_lagrange_grad_col(int ph, int v, double t, const double* x, const double* u, const double* p, const double* c) {
  if constexpr (prob_nq<P>::value > 0) return P::lagrange_grad_col(ph, v, t, x, u, p, c);
  else return P::lagrange_grad_col(ph, v, t, x, u, c);
}
template <class P>
RPM_DEV double pf_mayer(int ph, double t0, const double* x0, double tf, const double* xf, const double* p, const double* c) {
  if constexpr (prob_nq<P>::value > 0) return P::mayer(ph, t0, x0, tf, xf, p, c);
  else return P::mayer(ph, t0, x0, tf, xf, c);
}
template <class P>
RPM_DEV double pf_mayer_grad_col(int ph, int q, double t0, const double* x0, double tf, const double* xf, const double* p, const double* c) {
  if constexpr (prob_nq<P>::value > 0) return P::mayer_grad_col(ph, q, t0, x0, tf, xf, p, c);
  else return P::mayer_grad_col(ph, q, t0, x0, tf, xf, c);
}
template <class P>
RPM_DEV void pf_event(int ph, double t0, const double* x0, double tf, const double* xf, const double* p, const double* c, double* ev) {
  if constexpr (prob_nq<P>::value > 0) P::event(ph, t0, x0, tf, xf, p, c, ev);
  else P::event(ph, t0, x0, tf, xf, c, ev);
}
template <class P>
RPM_DEV void pf_event_jac_col(int ph, int v, double t0, const double* x0, double tf, const double* xf, const double* p, const double* c, double* de) {
  if constexpr (prob_nq<P>::value > 0) P::event_jac_col(ph, v, t0, x0, tf, xf, p, c, de);
  else P::event_jac_col(ph, v, t0, x0, tf, xf, c, de);
}
template <class P>
RPM_DEV void pf_link(int lph, int rph, const double* xl, const double* xr, const double* pl, const double* pr, const double* c, int nlink, double* lo) {
  if constexpr (prob_nq<P>::value > 0) P::link(lph, rph, xl, xr, pl, pr, c, nlink, lo);
  else P::link(lph, rph, xl, xr, c, nlink, lo);
}
template <class P>
RPM_DEV void pf_link_jac_col(int lph, int rph, int v, const double* xl, const double* xr, const double* pl, const double* pr, const double* c, int nlink, double* dl) {
  if constexpr (prob_nq<P>::value > 0) P::link_jac_col(lph, rph, v, xl, xr, pl, pr, c, nlink, dl);
  else P::link_jac_col(lph, rph, v, xl, xr, c, nlink, dl);
}

// ---------------------------------------------------------------------------------------------
// Delta-III launch vehicle ascent — example/launch/Launch.cpp:636-765
// consts: [0..8] omega_matrix (column-major), 9 mu, 10 cd, 11 sa, 12 rho0, 13 H, 14 Re, 15 g0,
//         16 thrust_srb, 17 thrust_first, 18 thrust_second, 19 ISP_srb, 20 ISP_first, 21 ISP_second
// x^3 rounded once: the square and the product are formed exactly (FMA residuals) and summed, so the result is the
// correctly rounded cube except in near-tie cases.  Stands for std::pow(x, 3.0), which is what the reference's
// arma::pow(rad, 3) calls (example/launch/Launch.cpp:685): it equals glibc's pow in 99.9 % of inputs and is within
// 1 ulp otherwise (tools/ubench/cube_vs_pow.c), at 6 flops instead of a ~180-instruction generic pow.
RPM_DEV double cube_rn(double x) {
  const double p = x * x, ep = fma(x, x, -p);
  const double q = p * x, eq = fma(p, x, -q);
  return q + (eq + ep * x);
}

struct LaunchProblem {
  static constexpr int ID = RPM_PROBLEM_LAUNCH;
  static constexpr int NX = 7, NU = 3, NC = 1, NE_MAX = 5, NLINK_MAX = 7, NCONST = 22;
  static constexpr bool HAS_ANALYTIC = false;

  template <class CP = const double*>
  RPM_DEV static void dae(int ph, double t, const double* x, const double* u, CP c,
                          double* f, double* p) {
    (void)t;
    const double r0 = x[0], r1 = x[1], r2 = x[2], m = x[6];
    const double rad = sqrt((r0 * r0 + r1 * r1) + r2 * r2);            // :670
    // omegacrossr = r * trans(omega_matrix), :672 (generic 3x3 product, zeros included)
    double vrel[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      double ocr = 0.0;
      ocr += r0 * c[cc + 0];
      ocr += r1 * c[cc + 3];
      ocr += r2 * c[cc + 6];
      vrel[cc] = x[3 + cc] - ocr;
    }
    const double speedrel = sqrt((vrel[0] * vrel[0] + vrel[1] * vrel[1]) + vrel[2] * vrel[2]);
    const double altitude = rad - c[14];
    const double rho = exp(-altitude / c[13]) * c[12];                 // :676-677
    const double bc = rho / (m * 2) * (c[11] * c[10]);                 // :678
    const double bcspeed = bc * speedrel;
    const double mu3 = (1.0 * c[9]) / cube_rn(rad);                    // mu / pow(rad, 3), :683-684
    double T_tot, mdot;
    if (ph == 1 || ph == 2) {                                          // :688-711
      const double T_srb = 1.0 * ((ph == 1 ? 6 : 3) * c[16]);
      const double T_first = 1.0 * c[17];
      T_tot = T_srb + T_first;
      double m1dot = 0.0, m2dot = 0.0;
      m1dot -= T_srb / (c[15] * c[19]);
      m2dot -= T_first / (c[15] * c[20]);
      mdot = m1dot + m2dot;
    } else if (ph == 3) {                                              // :712-717
      T_tot = 1.0 * c[17];
      mdot = 0.0;
      mdot -= T_tot / (c[15] * c[20]);
    } else {                                                           // :718-724
      T_tot = 1.0 * c[18];
      mdot = 0.0;
      mdot -= T_tot / (c[15] * c[21]);
    }
    p[0] = (u[0] * u[0] + u[1] * u[1]) + u[2] * u[2];                  // :726
    const double Toverm = T_tot / m;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double drag = (bcspeed * (-1.0)) * vrel[j];                // :681-682
      const double grav = (-mu3) * x[j];                               // :686
      f[j] = x[3 + j];
      f[3 + j] = (Toverm * u[j] + drag) + grav;                        // :733
    }
    f[6] = mdot;
  }

  // The same in stages (has_stage): what depends on the position only, on position and velocity, on the mass.
  struct Stage { double ocr[3], vrel[3], rho, mu3, speedrel, bc, T_tot, mdot; };
  template <class CP = const double*>
  RPM_DEV static void stage_r(const double* x, CP c, Stage& s) {        // position: rad, density, gravity factor, omega x r
    const double r0 = x[0], r1 = x[1], r2 = x[2];
    const double rad = sqrt((r0 * r0 + r1 * r1) + r2 * r2);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      double ocr = 0.0;
      ocr += r0 * c[cc + 0];
      ocr += r1 * c[cc + 3];
      ocr += r2 * c[cc + 6];
      s.ocr[cc] = ocr;
    }
    const double altitude = rad - c[14];
    s.rho = exp(-altitude / c[13]) * c[12];
    s.mu3 = (1.0 * c[9]) / cube_rn(rad);
  }
  RPM_DEV static void stage_v(const double* x, Stage& s) {              // velocity (after stage_r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) s.vrel[cc] = x[3 + cc] - s.ocr[cc];
    s.speedrel = sqrt((s.vrel[0] * s.vrel[0] + s.vrel[1] * s.vrel[1]) + s.vrel[2] * s.vrel[2]);
  }
  template <class CP = const double*>
  RPM_DEV static void stage_m(const double* x, CP c, Stage& s) {        // mass (after stage_r)
    s.bc = s.rho / (x[6] * 2) * (c[11] * c[10]);
  }
  template <class CP = const double*>
  RPM_DEV static void stage_ph(int ph, CP c, Stage& s) {                // the phase's thrust and mass flow
    double T_tot, mdot;
    if (ph == 1 || ph == 2) {
      const double T_srb = 1.0 * ((ph == 1 ? 6 : 3) * c[16]);
      const double T_first = 1.0 * c[17];
      T_tot = T_srb + T_first;
      double m1dot = 0.0, m2dot = 0.0;
      m1dot -= T_srb / (c[15] * c[19]);
      m2dot -= T_first / (c[15] * c[20]);
      mdot = m1dot + m2dot;
    } else if (ph == 3) {
      T_tot = 1.0 * c[17];
      mdot = 0.0;
      mdot -= T_tot / (c[15] * c[20]);
    } else {
      T_tot = 1.0 * c[18];
      mdot = 0.0;
      mdot -= T_tot / (c[15] * c[21]);
    }
    s.T_tot = T_tot;
    s.mdot = mdot;
  }
  template <class CP = const double*>
  RPM_DEV static void stage(int ph, double, const double* x, const double*, CP c, Stage& s) {
    stage_r(x, c, s);
    stage_v(x, s);
    stage_m(x, c, s);
    stage_ph(ph, c, s);
  }
  template <class CP = const double*>
  RPM_DEV static void dae_from(int, double, const double* x, const double* u, CP c, const Stage& base, int var, double* f, double* p) {
    Stage s = base;
    if (var >= 0 && var < 3) { stage_r(x, c, s); stage_v(x, s); stage_m(x, c, s); }
    else if (var >= 3 && var < 6) stage_v(x, s);
    else if (var == 6) stage_m(x, c, s);
    const double bcspeed = s.bc * s.speedrel;
    p[0] = (u[0] * u[0] + u[1] * u[1]) + u[2] * u[2];
    const double Toverm = s.T_tot / x[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double drag = (bcspeed * (-1.0)) * s.vrel[j];
      const double grav = (-s.mu3) * x[j];
      f[j] = x[3 + j];
      f[3 + j] = (Toverm * u[j] + drag) + grav;
    }
    f[6] = s.mdot;
  }
  // ... and at a point that differs from the base in variables va and vb (second differences, rpm_hess_kernel)
  template <class CP = const double*>
  RPM_DEV static void dae_from2(int ph, double t, const double* x, const double* u, CP c, const Stage& base, int va, int vb, double* f, double* p) {
    Stage s = base;
    const bool r = (va >= 0 && va < 3) || (vb >= 0 && vb < 3), v = (va >= 3 && va < 6) || (vb >= 3 && vb < 6), m = va == 6 || vb == 6;
    if (r) { stage_r(x, c, s); stage_v(x, s); stage_m(x, c, s); }
    else {
      if (v) stage_v(x, s);
      if (m) stage_m(x, c, s);
    }
    dae_from(ph, t, x, u, c, s, -1, f, p);
  }

  // Armadillo 5.300.4 dot() on 3-vectors: (a0 b0 + a2 b2) + a1 b1
  RPM_DEV static double dot3(const double* a, const double* b) {
    double v1 = 0.0, v2 = 0.0;
    v1 += a[0] * b[0];
    v2 += a[1] * b[1];
    v1 += a[2] * b[2];
    return v1 + v2;
  }
  // Launchrv2oe, :592-634 (first five elements) — event of phase 4, :744-754
  RPM_DEV static void event(int ph, double t0, const double* x0, double tf, const double* xf,
                            const double* c, double* ev) {
    (void)t0; (void)x0; (void)tf;
    if (ph != 4) return;
    const double mu = c[9];
    const double* rv = xf;
    const double* vv = xf + 3;
    double hv[3], nv[3], e3[3];
    hv[0] = rv[1] * vv[2] - rv[2] * vv[1];
    hv[1] = rv[2] * vv[0] - rv[0] * vv[2];
    hv[2] = rv[0] * vv[1] - rv[1] * vv[0];
    // cross(K,hv), K = (0,0,1)
    nv[0] = 0.0 * hv[2] - 1.0 * hv[1];
    nv[1] = 1.0 * hv[0] - 0.0 * hv[2];
    nv[2] = 0.0 * hv[1] - 0.0 * hv[0];
    const double n = sqrt(dot3(nv, nv));
    const double h2 = dot3(hv, hv);
    const double v2 = dot3(vv, vv);
    const double r = sqrt(dot3(rv, rv));
    const double s1 = v2 - mu / r, s2 = dot3(rv, vv);
#pragma unroll
    for (int j = 0; j < 3; ++j) e3[j] = (rv[j] * s1 - vv[j] * s2) * (1.0 / mu);
    const double p = h2 / mu;
    const double e = sqrt(dot3(e3, e3));
    const double twopi = 2 * 3.14159265358979323846;
    double Om1 = acos(nv[0] / n);
    if (nv[1] < 0 - 2.220446049250313e-16) Om1 = twopi - Om1;
    double Om2 = acos(dot3(nv, e3) / n / e);
    if (e3[2] < 0) Om2 = twopi - Om2;
    ev[0] = p / (1 - e * e);
    ev[1] = e;
    ev[2] = acos(hv[2] / sqrt(h2));
    ev[3] = Om1;
    ev[4] = Om2;
  }
  RPM_DEV static void link(int lph, int rph, const double* xfl, const double* x0r, const double* c,
                           int nlink, double* out) {                   // :760-765
    (void)lph; (void)rph; (void)c;
    for (int j = 0; j < nlink; ++j) out[j] = x0r[j] - xfl[j];
  }
  RPM_DEV static double mayer(int ph, double t0, const double* x0, double tf, const double* xf,
                              const double* c) {                       // :636-646
    (void)t0; (void)x0; (void)tf; (void)c;
    return ph == 4 ? -xf[6] : 0.0;
  }
  RPM_DEV static double lagrange(int ph, double t, const double* x, const double* u, const double* c) {
    (void)ph; (void)t; (void)x; (void)u; (void)c;
    return 0.0;                                                        // :652-656
  }
};

// ---------------------------------------------------------------------------------------------
// Hypersensitive — example/hypersensitive/HyperSensitive.cpp:74-167 (ships analytic derivatives)
struct HypersensitiveProblem {
  static constexpr int ID = RPM_PROBLEM_HYPERSENSITIVE;
  static constexpr int NX = 1, NU = 1, NC = 0, NE_MAX = 0, NLINK_MAX = 0, NCONST = 0;
  static constexpr bool HAS_ANALYTIC = true;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* u, CP, double* f, double*) {
    f[0] = ((-x[0]) * x[0]) * x[0] + u[0];                             // :131
  }
  // column v of [df/dx, df/du, df/dt] (DerivDae :134-151)
  template <class CP = const double*>
  RPM_DEV static void dae_jac_col(int, int v, double, const double* x, const double*, CP,
                                  double* df, double*) {
    df[0] = (v == 0) ? -3 * (x[0] * x[0]) : (v == 1 ? 1.0 : 0.0);
  }
  RPM_DEV static void event(int, double, const double*, double, const double*, const double*, double*) {}
  RPM_DEV static void link(int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer(int, double, const double*, double, const double*, const double*) { return 0.0; }
  RPM_DEV static double lagrange(int, double, const double* x, const double* u, const double*) {
    return 0.5 * (x[0] * x[0] + u[0] * u[0]);                          // :107
  }
  // column v of [dL/dx, dL/du, dL/dt] (DerivLagrange :110-121)
  RPM_DEV static double lagrange_grad_col(int, int v, double, const double* x, const double* u, const double*) {
    return v == 0 ? x[0] : (v == 1 ? u[0] : 0.0);
  }
  // entry v of [dM/dx0.., dM/dt0, dM/dxf.., dM/dtf] (DerivMayer :88-97)
  RPM_DEV static double mayer_grad_col(int, int, double, const double*, double, const double*, const double*) {
    return 0.0;
  }
  RPM_DEV static void event_jac_col(int, int, double, const double*, double, const double*, const double*, double*) {}
  RPM_DEV static void link_jac_col(int, int, int, const double*, const double*, const double*, int, double*) {}
};

// ---------------------------------------------------------------------------------------------
// Bryson-Denham — example/bryson-denham/BrysonDenham.cpp:100-167
struct BrysonDenhamProblem {
  static constexpr int ID = RPM_PROBLEM_BRYSON_DENHAM;
  static constexpr int NX = 3, NU = 1, NC = 0, NE_MAX = 5, NLINK_MAX = 0, NCONST = 0;
  static constexpr bool HAS_ANALYTIC = false;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* u, CP, double* f, double*) {
    f[0] = x[1];
    f[1] = u[0];
    f[2] = 0.5 * (u[0] * u[0]);                                        // :121-123
  }
  RPM_DEV static void event(int, double, const double* x0, double, const double* xf, const double*, double* ev) {
    ev[0] = x0[0]; ev[1] = x0[1]; ev[2] = x0[2]; ev[3] = xf[0]; ev[4] = xf[1];  // :139-153
  }
  RPM_DEV static void link(int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer(int, double, const double*, double, const double* xf, const double*) { return xf[2]; }
  RPM_DEV static double lagrange(int, double, const double*, const double*, const double*) { return 0.0; }
};

// ---------------------------------------------------------------------------------------------
// Brachistochrone (authored, BASELINE config 1; equations in DESIGN.md).  consts[0] = g.
struct BrachistochroneProblem {
  static constexpr int ID = RPM_PROBLEM_BRACHISTOCHRONE;
  static constexpr int NX = 3, NU = 1, NC = 0, NE_MAX = 5, NLINK_MAX = 0, NCONST = 1;
  static constexpr bool HAS_ANALYTIC = true;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* u, CP c, double* f, double*) {
    const double sn = sin(u[0]), cs = cos(u[0]);
    f[0] = x[2] * sn;
    f[1] = x[2] * cs;
    f[2] = c[0] * cs;
  }
  template <class CP = const double*>
  RPM_DEV static void dae_jac_col(int, int v, double, const double* x, const double* u, CP c,
                                  double* df, double*) {
    const double sn = sin(u[0]), cs = cos(u[0]);
    df[0] = df[1] = df[2] = 0.0;
    if (v == 2) { df[0] = sn; df[1] = cs; }
    if (v == 3) { df[0] = x[2] * cs; df[1] = -(x[2] * sn); df[2] = -(c[0] * sn); }
  }
  RPM_DEV static void event(int, double, const double* x0, double, const double* xf, const double*, double* ev) {
    ev[0] = x0[0]; ev[1] = x0[1]; ev[2] = x0[2]; ev[3] = xf[0]; ev[4] = xf[1];
  }
  // column v of d event / d [x0.., t0, xf.., tf]
  RPM_DEV static void event_jac_col(int, int v, double, const double*, double, const double*, const double*, double* de) {
    for (int i = 0; i < 5; ++i) de[i] = 0.0;
    if (v < 3) de[v] = 1.0;
    if (v == 4) de[3] = 1.0;
    if (v == 5) de[4] = 1.0;
  }
  RPM_DEV static void link(int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static void link_jac_col(int, int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer(int, double, const double*, double tf, const double*, const double*) { return tf; }
  RPM_DEV static double mayer_grad_col(int, int v, double, const double*, double, const double*, const double*) {
    return v == 2 * NX + 1 ? 1.0 : 0.0;
  }
  RPM_DEV static double lagrange(int, double, const double*, const double*, const double*) { return 0.0; }
  RPM_DEV static double lagrange_grad_col(int, int, double, const double*, const double*, const double*) { return 0.0; }
};

// ---------------------------------------------------------------------------------------------
// Minimum time to climb (authored, BASELINE config 2; equations in DESIGN.md).
// consts: 0 Re, 1 mu, 2 S, 3 g0, 4 Isp, 5 rho0, 6 Hs, 7 a0, 8 a1, 9 Tmax
struct MinTimeClimbProblem {
  static constexpr int ID = RPM_PROBLEM_MIN_TIME_CLIMB;
  static constexpr int NX = 4, NU = 1, NC = 0, NE_MAX = 7, NLINK_MAX = 0, NCONST = 10;
  static constexpr bool HAS_ANALYTIC = false;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* u, CP c, double* f, double*) {
    const double h = x[0], v = x[1], gam = x[2], m = x[3], al = u[0];
    const double r = h + c[0];
    const double rho = c[5] * exp(-h / c[6]);
    const double as = c[7] - c[8] * h;
    const double M = v / as;
    const double ch = cosh((M - 1.0) / 0.06);
    const double CLa = 3.44 + 1.0 / (ch * ch);
    const double CD0 = 0.013 + 0.0144 * (1.0 + tanh((M - 0.98) / 0.06));
    const double eta = 0.54 + 0.15 * (1.0 + tanh((M - 0.9) / 0.06));
    const double CD = CD0 + eta * CLa * (al * al);
    const double CL = CLa * al;
    const double q = 0.5 * rho * v * v;
    const double D = q * c[2] * CD;
    const double Lf = q * c[2] * CL;
    const double T = c[9] * pow(rho / c[5], 0.7) * (1.0 + 0.3 * M);
    const double sg = sin(gam), cg = cos(gam), sa = sin(al), ca = cos(al);
    f[0] = v * sg;
    f[1] = (T * ca - D) / m - c[1] * sg / (r * r);
    f[2] = (T * sa + Lf) / (m * v) + cg * (v / r - c[1] / (v * (r * r)));
    f[3] = -T / (c[3] * c[4]);
  }
  RPM_DEV static void event(int, double, const double* x0, double, const double* xf, const double*, double* ev) {
    ev[0] = x0[0]; ev[1] = x0[1]; ev[2] = x0[2]; ev[3] = x0[3]; ev[4] = xf[0]; ev[5] = xf[1]; ev[6] = xf[2];
  }
  RPM_DEV static void link(int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer(int, double, const double*, double tf, const double*, const double*) { return tf; }
  RPM_DEV static double lagrange(int, double, const double*, const double*, const double*) { return 0.0; }
};

// ---------------------------------------------------------------------------------------------
// Quadrotor (authored, BASELINE config 5; equations in DESIGN.md).
// consts: 0 mass, 1 g, 2 arm, 3 Ixx, 4 Iyy, 5 Izz, 6 ktau, 7..9 pref, 10 wp, 11 wv, 12 wa, 13 ww, 14 wu
struct QuadrotorProblem {
  static constexpr int ID = RPM_PROBLEM_QUADROTOR;
  static constexpr int NX = 12, NU = 4, NC = 0, NE_MAX = 0, NLINK_MAX = 0, NCONST = 15;
  static constexpr bool HAS_ANALYTIC = false;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* f4, CP c, double* f, double*) {
    const double ph = x[6], th = x[7], ps = x[8], p = x[9], q = x[10], r = x[11];
    const double F = ((f4[0] + f4[1]) + f4[2]) + f4[3];
    const double tx = c[2] * (f4[1] - f4[3]);
    const double ty = c[2] * (f4[2] - f4[0]);
    const double tz = c[6] * (((f4[0] - f4[1]) + f4[2]) - f4[3]);
    double sph, cph, sth, cth, sps, cps;     // (one argument reduction for the pair: sincos)
    sincos(ph, &sph, &cph);
    sincos(th, &sth, &cth);
    sincos(ps, &sps, &cps);
    const double b3x = cph * sth * cps + sph * sps;
    const double b3y = cph * sth * sps - sph * cps;
    const double b3z = cph * cth;
    const double Fm = F / c[0];
    f[0] = x[3];
    f[1] = x[4];
    f[2] = x[5];
    f[3] = Fm * b3x;
    f[4] = Fm * b3y;
    f[5] = Fm * b3z - c[1];
    const double w = q * sph + r * cph;
    f[6] = p + w * (sth / cth);
    f[7] = q * cph - r * sph;
    f[8] = w / cth;
    f[9] = (tx - (c[5] - c[4]) * q * r) / c[3];
    f[10] = (ty - (c[3] - c[5]) * p * r) / c[4];
    f[11] = (tz - (c[4] - c[3]) * p * q) / c[5];
  }
  // In stages (has_stage): the three sine / cosine pairs are all that is worth keeping
  struct Stage { double sph, cph, sth, cth, sps, cps; };
  static constexpr bool STAGE_ALWAYS = true;    // 1024-instance sweep, all stores: 44.2 -> 42.9 us per launch
  template <class CP = const double*>
  RPM_DEV static void stage(int, double, const double* x, const double*, CP, Stage& s) {
    sincos(x[6], &s.sph, &s.cph);
    sincos(x[7], &s.sth, &s.cth);
    sincos(x[8], &s.sps, &s.cps);
  }
  template <class CP = const double*>
  RPM_DEV static void dae_from(int, double, const double* x, const double* f4, CP c, const Stage& base, int var, double* f, double*) {
    Stage s = base;
    if (var == 6) sincos(x[6], &s.sph, &s.cph);
    else if (var == 7) sincos(x[7], &s.sth, &s.cth);
    else if (var == 8) sincos(x[8], &s.sps, &s.cps);
    const double p = x[9], q = x[10], r = x[11];
    const double F = ((f4[0] + f4[1]) + f4[2]) + f4[3];
    const double tx = c[2] * (f4[1] - f4[3]);
    const double ty = c[2] * (f4[2] - f4[0]);
    const double tz = c[6] * (((f4[0] - f4[1]) + f4[2]) - f4[3]);
    const double sph = s.sph, cph = s.cph, sth = s.sth, cth = s.cth, sps = s.sps, cps = s.cps;
    const double b3x = cph * sth * cps + sph * sps;
    const double b3y = cph * sth * sps - sph * cps;
    const double b3z = cph * cth;
    const double Fm = F / c[0];
    f[0] = x[3];
    f[1] = x[4];
    f[2] = x[5];
    f[3] = Fm * b3x;
    f[4] = Fm * b3y;
    f[5] = Fm * b3z - c[1];
    const double w = q * sph + r * cph;
    f[6] = p + w * (sth / cth);
    f[7] = q * cph - r * sph;
    f[8] = w / cth;
    f[9] = (tx - (c[5] - c[4]) * q * r) / c[3];
    f[10] = (ty - (c[3] - c[5]) * p * r) / c[4];
    f[11] = (tz - (c[4] - c[3]) * p * q) / c[5];
  }
  template <class CP = const double*>
  RPM_DEV static void dae_from2(int ph, double t, const double* x, const double* f4, CP c, const Stage& base, int va, int vb, double* f, double* p) {
    Stage s = base;
    if (va == 6 || vb == 6) sincos(x[6], &s.sph, &s.cph);
    if (va == 7 || vb == 7) sincos(x[7], &s.sth, &s.cth);
    if (va == 8 || vb == 8) sincos(x[8], &s.sps, &s.cps);
    dae_from(ph, t, x, f4, c, s, -1, f, p);
  }
  RPM_DEV static void event(int, double, const double*, double, const double*, const double*, double*) {}
  RPM_DEV static void link(int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer(int, double, const double*, double, const double*, const double*) { return 0.0; }
  RPM_DEV static double lagrange(int, double, const double* x, const double* f4, const double* c) {
    const double hov = c[0] * c[1] / 4.0;
    const double d0 = x[0] - c[7], d1 = x[1] - c[8], d2 = x[2] - c[9];
    const double ep = (d0 * d0 + d1 * d1) + d2 * d2;
    const double evv = (x[3] * x[3] + x[4] * x[4]) + x[5] * x[5];
    const double ea = (x[6] * x[6] + x[7] * x[7]) + x[8] * x[8];
    const double ew = (x[9] * x[9] + x[10] * x[10]) + x[11] * x[11];
    const double u0 = f4[0] - hov, u1 = f4[1] - hov, u2 = f4[2] - hov, u3 = f4[3] - hov;
    const double eu = ((u0 * u0 + u1 * u1) + u2 * u2) + u3 * u3;
    return (((c[10] * ep + c[11] * evv) + c[12] * ea) + c[13] * ew) + c[14] * eu;
  }
};

// ---------------------------------------------------------------------------------------------
// Minimum-time sled with a design parameter (authored here; nq = 1).  States (x, v), control u in [-1, 1], parameter p > 0
// scales the available acceleration: x' = v, v' = p u, from rest at 0 to rest at 1; cost tf + c0 p^2.  Bang-bang in u:
// tf = 2 / sqrt(p), so the cost 2 p^(-1/2) + c0 p^2 has its minimum at p = (2 c0)^(-2/5) — with c0 = 0.5: p = 1, cost 2.5
// (tests/test_known_answers.py).  consts: [0] c0.
struct ParamSledProblem {
  static constexpr int ID = RPM_PROBLEM_PARAM_SLED;
  static constexpr int NX = 2, NU = 1, NQ = 1, NC = 0, NE_MAX = 4, NLINK_MAX = 0, NCONST = 1;
  static constexpr bool HAS_ANALYTIC = true;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* u, const double* p, CP, double* f, double*) {
    f[0] = x[1];
    f[1] = p[0] * u[0];
  }
  RPM_DEV static void event(int, double, const double* x0, double, const double* xf, const double*, const double*, double* ev) {
    ev[0] = x0[0];
    ev[1] = x0[1];
    ev[2] = xf[0];
    ev[3] = xf[1];
  }
  RPM_DEV static void link(int, int, const double*, const double*, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer(int, double, const double*, double tf, const double*, const double* p, const double* c) {
    return tf + c[0] * (p[0] * p[0]);
  }
  RPM_DEV static double lagrange(int, double, const double*, const double*, const double*, const double*) { return 0.0; }
  // analytic columns: dae [x, v, u, t, p]; mayer / event [x0(2), t0, xf(2), tf, p]
  template <class CP = const double*>
  RPM_DEV static void dae_jac_col(int, int v, double, const double*, const double* u, const double* p, CP, double* df, double*) {
    df[0] = (v == 1) ? 1.0 : 0.0;
    df[1] = (v == 2) ? p[0] : (v == 4 ? u[0] : 0.0);
  }
  RPM_DEV static void event_jac_col(int, int v, double, const double*, double, const double*, const double*, const double*, double* de) {
    de[0] = (v == 0) ? 1.0 : 0.0;
    de[1] = (v == 1) ? 1.0 : 0.0;
    de[2] = (v == 3) ? 1.0 : 0.0;
    de[3] = (v == 4) ? 1.0 : 0.0;
  }
  RPM_DEV static void link_jac_col(int, int, int, const double*, const double*, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer_grad_col(int, int q, double, const double*, double, const double*, const double* p, const double* c) {
    return q == 5 ? 1.0 : (q == 6 ? 2.0 * c[0] * p[0] : 0.0);
  }
  RPM_DEV static double lagrange_grad_col(int, int, double, const double*, const double*, const double*, const double*) { return 0.0; }
};

// ---------------------------------------------------------------------------------------------
// Damped oscillator with stiffness and weighting parameters over two linked phases (authored here; nq = 2 per phase): every
// callback depends on the parameters — dynamics (stiffness p0), path constraint and running cost (p1), terminal cost and
// events, and the linkage ties the parameters of the two phases together.  consts: [0] damping, [1] weight of p0^2.
struct ParamOscProblem {
  static constexpr int ID = RPM_PROBLEM_PARAM_OSC;
  static constexpr int NX = 2, NU = 1, NQ = 2, NC = 1, NE_MAX = 2, NLINK_MAX = 4, NCONST = 2;
  static constexpr bool HAS_ANALYTIC = false;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* u, const double* p, CP c, double* f, double* cp) {
    f[0] = x[1];
    f[1] = (-(p[0] * x[0]) - c[0] * x[1]) + u[0];
    cp[0] = x[0] + p[1] * u[0];
  }
  RPM_DEV static void event(int ph, double, const double* x0, double, const double* xf, const double* p, const double*, double* ev) {
    if (ph == 1) {
      ev[0] = x0[0];
      ev[1] = x0[1];
    } else {
      ev[0] = xf[0] + p[1];
    }
  }
  RPM_DEV static void link(int, int, const double* xl, const double* xr, const double* pl, const double* pr, const double*, int, double* lo) {
    lo[0] = xl[0] - xr[0];
    lo[1] = xl[1] - xr[1];
    lo[2] = pl[0] - pr[0];
    lo[3] = pl[1] - pr[1];
  }
  RPM_DEV static double mayer(int ph, double, const double*, double, const double* xf, const double* p, const double*) {
    return ph == 2 ? xf[0] * xf[0] + p[0] * p[1] : 0.0;
  }
  RPM_DEV static double lagrange(int, double, const double* x, const double* u, const double* p, const double* c) {
    return (u[0] * u[0] + p[1] * (x[0] * x[0])) + c[1] * (p[0] * p[0]);
  }
};

}  // namespace rpm
