/*
 * orpm_post.c — CPU ORACLE, solution extraction (test infrastructure; PARITY UNPINNED, see orpm.h).
 * Restates Nlp2OpConverter::Nlp2OpControl, Core/Nlp2OPConverter.cpp:13-196 (nq = 0, auto-scale off).
 * Quirks kept: the path multipliers are read from lambda at [N*nx, N*nx + nc*N) WITHOUT the phase's constraint
 * offset (:88, right only for phase 1); SolCost.initial_time_ is never assigned (:120 assigns initial_state_
 * twice) — the problems' Mayer/Lagrange functions do not read it, t_all(0) is passed here.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orpm_internal.h"

void orpm_nlp2op(orpm* o, int ip, const double* x, const double* lambda, double* time, double* state,
                 double* control, double* costate, double* pathmult, double* hamiltonian, double* mayer_cost,
                 double* lagrange_cost) {
  const ophase* p = &o->ph[ip];
  int N = p->N, nx = p->nx, nu = p->nu, nc = p->nc, M = N + 1;
  double t0 = x[p->t0_idx], tf = x[p->tf_idx];
  double* tau_all = NEW(double, M);
  double* t_all = NEW(double, M);
  for (int k = 0; k < N; k++) tau_all[k] = p->points[k];
  tau_all[N] = 1.0;
  for (int k = 0; k < M; k++) t_all[k] = (tf - t0) * (tau_all[k] + 1) / 2 + t0;              /* :49 */
  const double* state_matrix = x + p->state0;                                                 /* (N+1) x nx */
  double* ctl = NEW(double, (size_t)M * (nu > 0 ? nu : 1));                                    /* control_matrixTotal */
  for (int j = 0; j < nu; j++) {                                                              /* :53-64 */
    const double* col = x + p->control0 + (size_t)j * N;
    for (int k = 0; k < N; k++) ctl[k + (size_t)j * M] = col[k];
    ctl[N + (size_t)j * M] = orpm_spline_interp(1.0, tau_all, col, N);
  }
  /* costates, :73-79 */
  const double* lam = lambda + p->con0;
  double* cst = NEW(double, (size_t)M * nx);
  for (int s = 0; s < nx; s++) {
    for (int k = 0; k < N; k++) cst[k + (size_t)s * M] = -((1 / p->weights[k]) * lam[k + (size_t)s * N]);
    double acc = 0.0; /* trans(D.col(N)) * lambda: the last column's entries, ascending row */
    for (int q = 0; q < p->d_nnz; q++)
      if (p->d_j[q] == N) acc += p->d_v[q] * lam[p->d_i[q] + (size_t)s * N];
    cst[N + (size_t)s * M] = -acc;
  }
  /* path multipliers, :81-116 (no phase offset on lambda, see header) */
  double* pm = NEW(double, (size_t)M * (nc > 0 ? nc : 1));
  for (int j = 0; j < nc; j++) {
    double* col = NEW(double, N);
    for (int k = 0; k < N; k++) col[k] = 2 * ((1 / p->weights[k]) * lambda[(size_t)N * nx + (size_t)j * N + k]) / (tf - t0);
    for (int k = 0; k < N; k++) pm[k + (size_t)j * M] = col[k];
    pm[N + (size_t)j * M] = orpm_spline_interp(1.0, tau_all, col, N);
    free(col);
  }
  /* cost and dynamics at the N+1 points, :117-146 */
  double* x0 = NEW(double, nx);
  double* xf = NEW(double, nx);
  for (int s = 0; s < nx; s++) {
    x0[s] = state_matrix[(size_t)s * M];
    xf[s] = state_matrix[(size_t)s * M + N];
  }
  orpm_solcost sc;
  memset(&sc, 0, sizeof(sc));
  sc.phase_num = ip + 1;
  sc.initial_time = t_all[0];
  sc.initial_state = x0;
  sc.terminal_time = t_all[N];
  sc.terminal_state = xf;
  sc.N = M; sc.nx = nx; sc.nu = nu; sc.nq = 0;
  sc.time = t_all; sc.state = state_matrix; sc.control = ctl; sc.parameter = NULL;
  double mayer = 0.0;
  double* L = NEW(double, M);
  o->fun->mayer(&sc, o->consts, &mayer);
  o->fun->lagrange(&sc, o->consts, L);
  double lcost = (tf - t0) * orpm_arma_dot(p->weights, L, N) / 2.0;                            /* :134 */
  orpm_soldae sd;
  sd.phase_num = ip + 1; sd.N = M; sd.nx = nx; sd.nu = nu; sd.nq = 0; sd.nc = nc;
  sd.time = t_all; sd.state = state_matrix; sd.control = ctl; sd.parameter = NULL;
  double* dae = NEW(double, (size_t)M * nx);
  double* path = NEW(double, (size_t)M * (nc > 0 ? nc : 1));
  o->fun->dae(&sd, o->consts, dae, path);
  if (hamiltonian)
    for (int k = 0; k < M; k++) {                                                              /* :146 */
      double sum = 0.0;
      for (int s = 0; s < nx; s++) {
        double term = cst[k + (size_t)s * M] * dae[k + (size_t)s * M];
        sum = (s == 0) ? term : sum + term;
      }
      hamiltonian[k] = L[k] + sum;
    }
  if (time) memcpy(time, t_all, sizeof(double) * M);
  if (state) memcpy(state, state_matrix, sizeof(double) * M * nx);
  if (control && nu > 0) memcpy(control, ctl, sizeof(double) * M * nu);
  if (costate) memcpy(costate, cst, sizeof(double) * M * nx);
  if (pathmult && nc > 0) memcpy(pathmult, pm, sizeof(double) * M * nc);
  if (mayer_cost) *mayer_cost = mayer;
  if (lagrange_cost) *lagrange_cost = lcost;
  free(tau_all); free(t_all); free(ctl); free(cst); free(pm); free(x0); free(xf); free(L); free(dae); free(path);
}
