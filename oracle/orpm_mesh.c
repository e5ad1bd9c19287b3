/*
 * orpm_mesh.c — CPU ORACLE, mesh-error estimate and ph refinement (test infrastructure; PARITY UNPINNED, see orpm.h).
 * Restates (paths relative to /root/reference/Lpopc/src):
 *   SolutionErrorChecker::BarLagrangeInterp / SolutionInterpolation / CheckSolutionDiffError  Core/LpSolutionError.cpp:10-169
 *   RPMGenerator integration / unity matrices (A = inv(D(:,1:)), B(:,0) = 1)                   Core/RPMGenerator.cpp:85-104,200-251
 *   PhMeshRefineAlg::RefineMesh / ModifySegment                                               Core/LpPhMeshRefineAlg.cpp:12-100
 * Armadillo's inv() (LAPACK getrf/getri, or closed forms for n <= 4) is not in the tree; a partial-pivoting LU
 * inverse stands in for it, so the integration matrix agrees with lpopc's to rounding (~cond(D) * 1e-16) only.
 * Quirk kept: the dynamics are evaluated at time (tf-t0)/2*tau + (tf-t0)/2 (t0 is not added, LpSolutionError.cpp:124).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orpm_internal.h"

/* Armadillo arrayops::product: two interleaved running products */
static double arma_product(const double* a, int n) {
  double v1 = 1.0, v2 = 1.0;
  int i, j;
  for (i = 0, j = 1; j < n; i += 2, j += 2) {
    v1 *= a[i];
    v2 *= a[j];
  }
  if (i < n) v1 *= a[i];
  return v1 * v2;
}

/* barycentric interpolation tables of one call: H (Nq x M, column-major, NaN where x hits a data point),
 * row sums S, and fix[r] = index of the coinciding data point or -1.  LpSolutionError.cpp:10-44 */
void orpm_bary_tables(int M, const double* data_x, int Nq, const double* xq, double* H, double* S, int* fix) {
  double* w = NEW(double, M);
  double* col = NEW(double, M);
  for (int j = 0; j < M; j++) {
    for (int i = 0; i < M; i++) col[i] = (data_x[i] - data_x[j]) + (i == j ? 1.0 : 0.0);
    w[j] = 1 / arma_product(col, M);
  }
  for (int r = 0; r < Nq; r++) fix[r] = -1;
  for (int j = 0; j < M; j++)
    for (int r = 0; r < Nq; r++) {
      double xd = xq[r] - data_x[j];
      if (xd == 0) {
        fix[r] = j;
        xd = NAN;
      }
      H[r + (size_t)j * Nq] = w[j] / xd;
    }
  for (int r = 0; r < Nq; r++) {
    double s = H[r];
    for (int j = 1; j < M; j++) s += H[r + (size_t)j * Nq];
    S[r] = s;
  }
  free(w);
  free(col);
}
static void bary_apply(int M, int Nq, const double* H, const double* S, const int* fix, const double* data_y, double* y) {
  for (int r = 0; r < Nq; r++) {
    double acc = 0.0;
    for (int j = 0; j < M; j++) acc += H[r + (size_t)j * Nq] * data_y[j];
    y[r] = fix[r] >= 0 ? data_y[fix[r]] : acc / S[r];
  }
}

/* inverse by LU with partial pivoting (stand-in for arma::inv, see header); A is n x n column-major, overwritten */
void orpm_inverse(int n, const double* Ain, double* inv) {
  double* A = orpm_dupd(Ain, n * n);
  int* piv = NEW(int, n);
  for (int k = 0; k < n; k++) {
    int p = k;
    double mx = fabs(A[k + (size_t)k * n]);
    for (int i = k + 1; i < n; i++)
      if (fabs(A[i + (size_t)k * n]) > mx) { mx = fabs(A[i + (size_t)k * n]); p = i; }
    piv[k] = p;
    if (p != k)
      for (int j = 0; j < n; j++) { double t = A[k + (size_t)j * n]; A[k + (size_t)j * n] = A[p + (size_t)j * n]; A[p + (size_t)j * n] = t; }
    for (int i = k + 1; i < n; i++) {
      A[i + (size_t)k * n] /= A[k + (size_t)k * n];
      double l = A[i + (size_t)k * n];
      for (int j = k + 1; j < n; j++) A[i + (size_t)j * n] -= l * A[k + (size_t)j * n];
    }
  }
  for (int c = 0; c < n; c++) {   /* solve A x = e_c */
    double* b = inv + (size_t)c * n;
    for (int i = 0; i < n; i++) b[i] = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < n; k++)
      if (piv[k] != k) { double t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++) b[i] -= A[i + (size_t)j * n] * b[j];
    for (int i = n - 1; i >= 0; i--) {
      for (int j = i + 1; j < n; j++) b[i] -= A[i + (size_t)j * n] * b[j];
      b[i] /= A[i + (size_t)i * n];
    }
  }
  free(A);
  free(piv);
}

/* CheckSolutionDiffError: relative_error is (sum(N_k+1)+1) x nx, column-major.  Returns the number of rows. */
int orpm_solution_error(orpm* o, int ip, const double* x, double* rel_err) {
  const ophase* p = &o->ph[ip];
  int N = p->N, nx = p->nx, nu = p->nu, nc = p->nc, K = p->K, M1 = N + 1;
  double t0 = x[p->t0_idx], tf = x[p->tf_idx];
  tf = (tf - t0) * (1.0 + 1) / 2 + t0;   /* result->time's last entry, Nlp2OPConverter.cpp:58 */
  const double* state = x + p->state0;     /* result->state: (N+1) x nx */
  const double* control = x + p->control0; /* rows 0..N-1 of result->control */
  int Np = N + K;                          /* nodes of the mesh with one more LGR point per interval */
  int rows = Np + 1;
  double* tau = NEW(double, M1);
  for (int k = 0; k < N; k++) tau[k] = p->points[k];
  tau[N] = 1.0;
  double* temTime = NEW(double, rows);
  double* temState = NEW(double, (size_t)rows * nx);
  double* temControl = NEW(double, (size_t)Np * (nu > 0 ? nu : 1));
  /* SolutionInterpolation, :46-108 */
  int istart = 0, r0 = 0;
  for (int seg = 0; seg < K; seg++) {
    int n = p->nk[seg], ifinish = istart + n;
    double time0 = tau[istart], timef = tau[ifinish];
    double* xi = NEW(double, n + 1);
    double* wi = NEW(double, n + 1);
    orpm_lgr_points(n + 1, xi, wi);
    double* ttem = NEW(double, n + 2);
    for (int q = 0; q < n + 1; q++) ttem[q] = (xi[q] + 1) * (timef - time0) / 2 + time0;
    ttem[n + 1] = p->mesh[seg + 1];
    double* H = NEW(double, (size_t)(n + 2) * (n + 1));
    double* S = NEW(double, n + 2);
    int* fix = NEW(int, n + 2);
    double* yq = NEW(double, n + 2);
    double* yd = NEW(double, n + 1);
    orpm_bary_tables(n + 1, tau + istart, n + 2, ttem, H, S, fix);
    for (int s = 0; s < nx; s++) {
      for (int q = 0; q < n + 1; q++) yd[q] = state[(istart + q) + (size_t)s * M1];
      bary_apply(n + 1, n + 2, H, S, fix, yd, yq);
      for (int q = 0; q < n + 1; q++) temState[(r0 + q) + (size_t)s * rows] = yq[q];
    }
    if (nu > 0) {
      orpm_bary_tables(n, tau + istart, n + 1, ttem, H, S, fix);
      for (int j = 0; j < nu; j++) {
        for (int q = 0; q < n; q++) yd[q] = control[(istart + q) + (size_t)j * N];
        bary_apply(n, n + 1, H, S, fix, yd, yq);
        for (int q = 0; q < n + 1; q++) temControl[(r0 + q) + (size_t)j * Np] = yq[q];
      }
    }
    for (int q = 0; q < n + 1; q++) temTime[r0 + q] = ttem[q];
    free(xi); free(wi); free(ttem); free(H); free(S); free(fix); free(yq); free(yd);
    istart = ifinish;
    r0 += n + 1;
  }
  temTime[Np] = 1.0;
  for (int s = 0; s < nx; s++) temState[Np + (size_t)s * rows] = state[N + (size_t)s * M1];
  /* dynamics on the finer mesh, :121-131 */
  double* tm = NEW(double, Np);
  double* st = NEW(double, (size_t)Np * nx);
  for (int k = 0; k < Np; k++) tm[k] = (tf - t0) / 2 * temTime[k] + (tf - t0) / 2;
  for (int s = 0; s < nx; s++) memcpy(st + (size_t)s * Np, temState + (size_t)s * rows, sizeof(double) * Np);
  orpm_soldae sd;
  sd.phase_num = ip + 1; sd.N = Np; sd.nx = nx; sd.nu = nu; sd.nq = 0; sd.nc = nc;
  sd.time = tm; sd.state = st; sd.control = temControl; sd.parameter = NULL;
  double* dae = NEW(double, (size_t)Np * nx);
  double* path = NEW(double, (size_t)Np * (nc > 0 ? nc : 1));
  o->fun->dae(&sd, o->consts, dae, path);
  for (size_t q = 0; q < (size_t)Np * nx; q++) dae[q] *= (tf - t0) / 2.0;
  /* integration on the (N_k+1)-point mesh: row 0 = temState row 0; row 1+r = X(start of r's interval) + A_K f, :147 */
  double* integ = NEW(double, (size_t)rows * nx);
  for (int s = 0; s < nx; s++) integ[(size_t)s * rows] = temState[(size_t)s * rows];
  r0 = 0;
  for (int seg = 0; seg < K; seg++) {
    int n1 = p->nk[seg] + 1;
    double* xi = NEW(double, n1);
    double* wi = NEW(double, n1);
    orpm_lgr_points(n1, xi, wi);
    double tspan = p->mesh[seg + 1] - p->mesh[seg];
    double* sall = NEW(double, n1 + 1);
    for (int q = 0; q < n1; q++) {
      double v = xi[q] + 1;
      v *= tspan / 2.0;
      v += p->mesh[seg];
      sall[q] = v;
    }
    sall[n1] = p->mesh[seg + 1];
    double* D2 = NEW(double, (size_t)n1 * (n1 + 1));
    orpm_colloc_d(n1 + 1, sall, D2);
    double* A = NEW(double, (size_t)n1 * n1);
    orpm_inverse(n1, D2 + n1, A);   /* inv(D2(:, 1:end)), RPMGenerator.cpp:85 */
    for (int s = 0; s < nx; s++)
      for (int r = 0; r < n1; r++) {
        double acc = 0.0;           /* IntegrationMatrix * daeout: COO loop, ascending column */
        for (int cidx = 0; cidx < n1; cidx++) acc += A[r + (size_t)cidx * n1] * dae[(r0 + cidx) + (size_t)s * Np];
        double unity = 0.0;
        unity += 1.0 * temState[r0 + (size_t)s * rows];   /* UnityMatrix * temState */
        integ[(1 + r0 + r) + (size_t)s * rows] = unity + acc;
      }
    free(xi); free(wi); free(sall); free(D2); free(A);
    r0 += n1;
  }
  for (int s = 0; s < nx; s++) {   /* :148-157 */
    double mx = temState[(size_t)s * rows];
    for (int r = 1; r < rows; r++)
      if (temState[r + (size_t)s * rows] > mx) mx = temState[r + (size_t)s * rows];
    double den = 1 + mx;
    for (int r = 0; r < rows; r++)
      rel_err[r + (size_t)s * rows] = fabs(integ[r + (size_t)s * rows] - temState[r + (size_t)s * rows]) / den;
  }
  free(tau); free(temTime); free(temState); free(temControl); free(tm); free(st); free(dae); free(path); free(integ);
  return rows;
}

/* PhMeshRefineAlg::RefineMesh for one phase.  new_mesh needs room for (sum over intervals of max(1, Bq)) + 1 points.
 * Returns 1 when no interval needed refinement (NoMoreRefine), 0 otherwise. */
int orpm_ph_refine(orpm* o, int ip, const double* x, double tol, int Nmin, int Nmax, double* new_mesh, int* new_nodes,
                   int* new_K, double* max_err_per_interval) {
  const ophase* p = &o->ph[ip];
  int K = p->K, nx = p->nx, rows = p->N + K + 1;
  double* rel = NEW(double, (size_t)rows * nx);
  orpm_solution_error(o, ip, x, rel);
  int no_more = 1, nk = 0, istart = 0;
  new_mesh[0] = -1;
  int nm = 1;
  for (int seg = 0; seg < K; seg++) {
    int n = p->nk[seg], ifinish = istart + n + 1;   /* rows [istart, ifinish] of the (N_k+1)-mesh, inclusive */
    double emax = rel[istart];
    for (int s = 0; s < nx; s++)
      for (int r = istart; r <= ifinish; r++)
        if (rel[r + (size_t)s * rows] > emax) emax = rel[r + (size_t)s * rows];
    if (max_err_per_interval) max_err_per_interval[seg] = emax;
    double m0 = p->mesh[seg], mf = p->mesh[seg + 1];
    if (emax <= tol) {
      new_mesh[nm++] = mf;
      new_nodes[nk++] = n;
    } else {   /* ModifySegment, :78-98 */
      no_more = 0;
      int Pq = (int)(log(emax / tol) / log((double)n));
      int newnodes = n + Pq;
      if (newnodes <= Nmax) {
        new_mesh[nm++] = mf;
        new_nodes[nk++] = newnodes;
      } else {
        int Bq = (int)fmax(ceil((double)newnodes / (double)Nmin), 2.0);
        double delta = (mf - m0) / (double)Bq;   /* linspace(m0, mf, Bq+1): start + i*delta, last = end */
        for (int i = 1; i <= Bq; i++) {
          new_mesh[nm++] = (i == Bq) ? mf : m0 + i * delta;
          new_nodes[nk++] = Nmin;
        }
      }
    }
    istart = ifinish;
  }
  *new_K = nk;
  free(rel);
  return no_more;
}
