/*
 * orpm_core.c — CPU ORACLE (test infrastructure; see orpm.h header: PARITY UNPINNED).
 *
 * Restates, in plain C, the reference's per-mesh set-up and per-iteration NLP
 * callbacks.  Paths in comments are relative to /root/reference/Lpopc/src.
 * Compile with -ffp-contract=off: the reference is plain x86-64 C++ without FMA.
 *
 * Matrices are column-major (Armadillo convention).  The structure deliberately
 * keeps the reference's cost shape (whole-column finite differences through the
 * vectorised user callbacks, COO sparse-times-dense loop, per-call Find() of the
 * off-diagonal differentiation matrix) so that timing it is a fair "port"
 * baseline; its leaks (SURVEY B-1..B-3) are not reproduced.
 */
#include "orpm.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#include "orpm_internal.h"

void* orpm_xcalloc(size_t n, size_t s) {
  void* p = calloc(n ? n : 1, s);
  if (!p) {
    fprintf(stderr, "orpm: out of memory\n");
    abort();
  }
  return p;
}

/* ---------------------------------------------------------------------------
 * Armadillo 5.300.4 arrayops::accumulate (two running sums over even/odd
 * elements, then their sum) — the order used by sum(X) on a column. */
double orpm_arma_accumulate(const double* a, int n) {
  double acc1 = 0.0, acc2 = 0.0;
  int j;
  for (j = 1; j < n; j += 2) {
    acc1 += a[j - 1];
    acc2 += a[j];
  }
  if ((j - 1) < n) acc1 += a[j - 1];
  return acc1 + acc2;
}
/* Armadillo 5.300.4 op_dot::direct_dot_arma (two accumulators) */
double orpm_arma_dot(const double* a, const double* b, int n) {
  double v1 = 0.0, v2 = 0.0;
  int i, j;
  for (i = 0, j = 1; j < n; i += 2, j += 2) {
    v1 += a[i] * b[i];
    v2 += a[j] * b[j];
  }
  if (i < n) v1 += a[i] * b[i];
  return v1 + v2;
}

/* RPMGenerator::GetLGRPointsImp, Core/RPMGenerator.cpp:253-291 */
void orpm_lgr_points(int iniN, double* x, double* w) {
  int N = iniN - 1, N1 = N + 1;
  double eps = DBL_EPSILON;
  double* xold = NEW(double, N1);
  double* P = NEW(double, (size_t)N1 * (N1 + 1));
#define Pc(c) (P + (size_t)(c)*N1)
  for (int k = 0; k <= N; k++) {
    double lin = (double)k; /* linspace(0,N,N+1): start + k*delta with delta == 1 */
    x[k] = -1 * cos(lin * ((2 * M_PI) / (2 * N + 1)));
    xold[k] = 2;
  }
  for (int it = 0; it < 200; it++) {
    double mx = 0;
    for (int k = 0; k < N1; k++) {
      double d = fabs(x[k] - xold[k]);
      if (d > mx) mx = d;
    }
    if (!(mx > eps)) break;
    for (int k = 0; k < N1; k++) xold[k] = x[k];
    for (int r = 0; r < N1; r++) {
      Pc(0)[r] = 1.0;
      Pc(1)[r] = x[r];
    }
    for (int k = 1; k < N1; k++)
      for (int r = 0; r < N1; r++) {
        double ret1 = x[r] * (2 * k + 1) * Pc(k)[r] - (Pc(k - 1)[r] * k);
        Pc(k + 1)[r] = ret1 / (k + 1);
      }
    for (int r = 1; r <= N; r++) {
      double ret2 = (1.0 - xold[r]) / N1;
      double temp = Pc(N1 - 1)[r] + Pc(N1)[r];
      ret2 = ret2 * temp;
      ret2 = xold[r] - (ret2 / (Pc(N1 - 1)[r] - Pc(N1)[r]));
      x[r] = ret2;
    }
  }
  w[0] = 2.0 / (N1 * N1);
  for (int r = 1; r <= N; r++) {
    double ret4 = Pc(N)[r] * N1;
    w[r] = (1 - x[r]) / (ret4 * ret4);
  }
#undef Pc
  free(xold);
  free(P);
}

/* RPMGenerator::CollocD, Core/RPMGenerator.cpp:107-130.
 * x: M points (N_k LGR nodes + right mesh point); D: (M-1) x M, column-major. */
void orpm_colloc_d(int M, const double* x, double* D) {
  double* Yd = NEW(double, (size_t)M * M); /* Ydiff = eye + Y - Y' */
  double* ww = NEW(double, M);
  double* D0 = NEW(double, (size_t)M * M);
  for (int j = 0; j < M; j++)
    for (int i = 0; i < M; i++) Yd[i + (size_t)j * M] = ((i == j ? 1.0 : 0.0) + x[i]) - x[j];
  /* prod(Ydiff,1): out = ones; for col: out[row] *= X(row,col) */
  for (int i = 0; i < M; i++) {
    double p = 1.0;
    for (int j = 0; j < M; j++) p *= Yd[i + (size_t)j * M];
    ww[i] = 1 / p;
  }
  /* D = ww ./ (ww' .* Ydiff) */
  for (int j = 0; j < M; j++)
    for (int i = 0; i < M; i++) D0[i + (size_t)j * M] = ww[i] / (ww[j] * Yd[i + (size_t)j * M]);
  /* D(j,j) = 1 - sum(D)(j)  (column sums, Armadillo accumulate order) */
  for (int j = 0; j < M; j++) {
    double s = orpm_arma_accumulate(D0 + (size_t)j * M, M);
    D0[j + (size_t)j * M] = 1 - s;
  }
  /* D = (-D')(0:M-2, :) */
  for (int b = 0; b < M; b++)
    for (int a = 0; a < M - 1; a++) D[a + (size_t)b * (M - 1)] = -D0[b + (size_t)a * M];
  free(Yd);
  free(ww);
  free(D0);
}

/* RPMGenerator::initialize + CompositeD, Core/RPMGenerator.cpp:43-105,132-181;
 * results land where LpGuessChecker::GetGuess puts them (Core/LpGuessChecker.cpp:116-122). */
static void build_phase_tables(ophase* p) {
  int K = p->K, N = 0, nnzD = 0;
  for (int i = 0; i < K; i++) {
    N += p->nk[i];
    nnzD += p->nk[i] * (p->nk[i] + 1);
  }
  p->N = N;
  p->points = NEW(double, N);
  p->weights = NEW(double, N);
  int* ti = NEW(int, nnzD);
  int* tj = NEW(int, nnzD);
  double* tv = NEW(double, nnzD);   /* D   */
  double* tdv = NEW(double, nnzD);  /* Dd  */
  double* tov = NEW(double, nnzD);  /* Do  */
  int itor = 0, k = 0, rowshift = 0, colshift = 0;
  for (int i = 0; i < K; i++) {
    int nk = p->nk[i];
    double* x = NEW(double, nk);
    double* w = NEW(double, nk);
    orpm_lgr_points(nk, x, w);
    double tspan = p->mesh[i + 1] - p->mesh[i];
    double* sall = NEW(double, nk + 1);
    for (int r = 0; r < nk; r++) {
      double s = x[r] + 1;
      s *= tspan / 2.0;
      s += p->mesh[i];
      sall[r] = s;
      p->points[itor + r] = s;
      double ws = w[r] / 2;
      ws *= tspan;
      p->weights[itor + r] = ws;
    }
    sall[nk] = p->mesh[i + 1];
    double* D2 = NEW(double, (size_t)nk * (nk + 1));
    orpm_colloc_d(nk + 1, sall, D2);
    /* GeneratRowColValue: column-major sweep of the block, SparseMatrix/LpSparseMatrix.cpp:80-102 */
    for (int j = 0; j < nk + 1; j++)
      for (int r = 0; r < nk; r++) {
        double d = D2[r + (size_t)j * nk];
        double dd = (r == j) ? d : 0.0; /* Dd = diagmat(diag(D(:,0:N1-1))) padded, :124-127 */
        ti[k] = r + rowshift;
        tj[k] = j + colshift;
        tv[k] = d;
        tdv[k] = dd;
        tov[k] = d - dd; /* Do = D - Dd */
        k++;
      }
    rowshift += nk;
    colshift += nk;
    itor += nk;
    free(x);
    free(w);
    free(sall);
    free(D2);
  }
  /* dsmatrix::Find drops exact zeros, SparseMatrix/LpSparseMatrix.cpp:240-271 */
  p->d_i = NEW(int, nnzD);
  p->d_j = NEW(int, nnzD);
  p->d_v = NEW(double, nnzD);
  p->diag_v = NEW(double, nnzD);
  p->off_i = NEW(int, nnzD);
  p->off_j = NEW(int, nnzD);
  p->off_v = NEW(double, nnzD);
  p->d_nnz = p->diag_nnz = p->off_nnz = 0;
  for (int q = 0; q < nnzD; q++) {
    if (tv[q] != 0.0) {
      p->d_i[p->d_nnz] = ti[q];
      p->d_j[p->d_nnz] = tj[q];
      p->d_v[p->d_nnz++] = tv[q];
    }
    if (tdv[q] != 0.0) p->diag_v[p->diag_nnz++] = tdv[q];
    if (tov[q] != 0.0) {
      p->off_i[p->off_nnz] = ti[q];
      p->off_j[p->off_nnz] = tj[q];
      p->off_v[p->off_nnz++] = tov[q];
    }
  }
  free(ti);
  free(tj);
  free(tv);
  free(tdv);
  free(tov);
}

/* dsmatrix::operator*(mat&), SparseMatrix/LpSparseMatrix.cpp:127-155 */
static void coo_mul(int nnz, const int* ri, const int* ci, const double* v, int m, const double* X,
                    int xrows, int ncols, double* out) {
  double* temcol = NEW(double, xrows);
  memset(out, 0, sizeof(double) * (size_t)m * ncols);
  for (int icol = 0; icol < ncols; icol++) {
    memcpy(temcol, X + (size_t)icol * xrows, sizeof(double) * xrows);
    for (int k = 0; k < nnz; k++) {
      int j = ci[k], i = ri[k];
      out[i + (size_t)icol * m] += v[k] * temcol[j];
    }
  }
  free(temcol);
}

/* "fair" cost shape: the same product from dense rows.  Row i of the composite D has its non-zeros in one mesh interval,
 * columns ascending; the COO loop above visits a row's entries in exactly that order (blocks are column-major), each time
 * as out += v * x starting from 0, so the sums are bit-identical. */
static void dense_rows_mul(const ophase* p, int m, const double* X, int xrows, int ncols, double* out) {
  for (int icol = 0; icol < ncols; icol++) {
    const double* col = X + (size_t)icol * xrows;
    for (int i = 0; i < m; i++) {
      const double* v = p->fr_vals + p->fr_off[i];
      const double* xx = col + p->fr_col0[i];
      double acc = 0.0;
      for (int q = 0; q < p->fr_len[i]; q++) acc += v[q] * xx[q];
      out[i + (size_t)icol * m] = acc;
    }
  }
}

/* dsmatrix::Find, SparseMatrix/LpSparseMatrix.cpp:240-271 (returns fresh arrays) */
static int coo_find(int nnz, const int* ri, const int* ci, const double* v, double** oi, double** oj,
                    double** ov) {
  int nz = 0;
  for (int i = 0; i < nnz; i++)
    if (v[i] != 0.0) nz++;
  *oi = NEW(double, nz);
  *oj = NEW(double, nz);
  *ov = NEW(double, nz);
  int q = 0;
  for (int i = 0; i < nnz; i++)
    if (v[i] != 0.0) {
      (*oi)[q] = (double)ri[i];
      (*oj)[q] = (double)ci[i];
      (*ov)[q] = v[i];
      q++;
    }
  return nz;
}

/* ---------------------------------------------------------------------------
 * natural cubic spline used for the guess, Core/LpGuessChecker.cpp:208-294 */
static void spline_second_derivative(const double* x, const double* y, int n, double* d2y) {
  double* c = d2y;
  double hi, him1, alphai, li = 0;
  double* mu = NEW(double, n);
  double* z = NEW(double, n);
  mu[0] = 0.0;
  z[0] = 0.0;
  for (int i = 1; i < n - 1; i++) {
    him1 = x[i] - x[i - 1];
    hi = x[i + 1] - x[i];
    alphai = 3.0 / hi * (y[i + 1] - y[i]) - 3.0 / him1 * (y[i] - y[i - 1]);
    li = 2 * (x[i + 1] - x[i - 1]) - him1 * mu[i - 1];
    mu[i] = hi / li;
    z[i] = (alphai - him1 * z[i - 1]) / li;
  }
  c[n - 1] = 0.0;
  for (int j = n - 2; j >= 0; j--) c[j] = z[j] - mu[j] * c[j + 1];
  for (int j = 1; j < n - 1; j++) c[j] = 2 * c[j];
  free(mu);
  free(z);
}
double orpm_spline_interp(double x, const double* xdata, const double* ydata, int n) {
  int kleft = 1, kright = n, k;
  double* d2y = NEW(double, n);
  spline_second_derivative(xdata, ydata, n, d2y);
  while (kright - kleft > 1) {
    k = (int)((kright + kleft) / 2);
    if (xdata[k - 1] > x)
      kright = k;
    else
      kleft = k;
  }
  double h = xdata[kright - 1] - xdata[kleft - 1];
  double A = (xdata[kright - 1] - x) / h;
  double B = (x - xdata[kleft - 1]) / h;
  double C = (pow(A, 3) - A) * (h * h) / 6.0;
  double D = (pow(B, 3) - B) * (h * h) / 6.0;
  double y = A * ydata[kleft - 1] + B * ydata[kright - 1] + C * d2y[kleft - 1] + D * d2y[kright - 1];
  free(d2y);
  return y;
}

/* ------------------------------------------------------------------------- */
static int fail(char* err, int errlen, const char* msg) {
  if (err && errlen > 0) {
    strncpy(err, msg, (size_t)errlen - 1);
    err[errlen - 1] = 0;
  }
  return 0;
}
double* orpm_dupd(const double* s, int n) {
  double* d = NEW(double, n);
  if (n > 0 && s) memcpy(d, s, sizeof(double) * n);
  return d;
}

void* orpm_hess_create(orpm* o);
void orpm_hess_destroy(void* h);

orpm* orpm_create(const rpm_problem_desc* d, char* err, int errlen) {
  if (!d || d->n_phases < 1) {
    fail(err, errlen, "bad description");
    return NULL;
  }
  const orpm_functions* fun = orpm_problem_functions(d->problem_id);
  if (!fun) {
    fail(err, errlen, "unknown problem id");
    return NULL;
  }
  orpm* o = NEW(orpm, 1);
  o->P = d->n_phases;
  o->L = d->n_links;
  o->fun = fun;
  o->nconsts = d->n_consts;
  o->consts = orpm_dupd(d->consts, d->n_consts);
  o->tol = d->fd_tol > 0 ? d->fd_tol : 1e-6;
  o->first_derive = d->first_derive;
  o->hessian_mode = d->hessian_approximation;
  o->ph = NEW(ophase, o->P);
  o->lk = NEW(olink, o->L > 0 ? o->L : 1);

  /* ---- sizes (LpSizeChecker::GetSize, Core/LpSizeChecker.cpp:13-152) + mesh check
   *      (MeshRefiner::SetAndCheckMesh, Core/LpMeshRefiner.cpp:10-62) ---- */
  for (int i = 0; i < o->P; i++) {
    const rpm_phase_desc* pd = &d->phases[i];
    ophase* p = &o->ph[i];
    p->nx = pd->nx;
    p->nu = pd->nu;
    p->nq = pd->nq;
    p->nc = pd->nc;
    p->ne = pd->ne;
    p->K = pd->n_intervals;
    /* Static parameters (nq > 0).  Layout, bounds, guess and the ORDER of every derivative column / Jacobian block follow
     * the reference (LpBoundsChecker.cpp:117-138, LpGuessChecker.cpp:186-189, LpFiniteDifferenceDerive.cpp:282-317,
     * LpNLPWrapper.cpp:763-769,814-820,854-859,461-519,1088-1097).  Its VALUES do not: the reference's parameter path is
     * inconsistent with itself, and this oracle restates the formulas it means —
     *   B-6  dDae/dPath_parameter read derivative column nx+nu+ip, the TIME column for ip = 0 (:618,621): column nx+nu+1+ip here;
     *   B-7  the path rows' parameter block inserts dDae_parameter (:817): dPath_parameter here;
     *   B-8  the parameter blocks' sparsity is a diagonal run indexvector + colstart (:1211,1262): the parameter's ONE column here;
     *   B-9  right_parameter_ = p_left, right-parameter differences divided by the left step, DLink_p_left stored twice
     *        (:203,423, LpFiniteDifferenceDerive.cpp:485,500): the right phase's parameters, each by its own step, here;
     *   B-21 GetObjGrad never fills SolCost.parameter_ and multiplies two column vectors for dCost/dp (:975,1094): the phase's
     *        parameters and the quadrature sum_k w_k (dt/2) dL/dp_j here; dLagrange_time is read from the LAST column (:1025),
     *        the time column nx+nu here.
     * So for nq > 0 this oracle is the specification, not a restatement; the exact Hessian is not defined for nq > 0. */
    if (pd->nq > 0 && d->hessian_approximation == RPM_HESSIAN_EXACT) {
      fail(err, errlen, "hessian-approximation=exact is not defined here for static parameters (nq>0)");
      orpm_destroy(o);
      return NULL;
    }
    if (p->K < 1 || pd->mesh_points[0] != -1 || pd->mesh_points[p->K] != 1) {
      fail(err, errlen, "meshPoints must span -1 to +1");
      orpm_destroy(o);
      return NULL;
    }
    p->mesh = orpm_dupd(pd->mesh_points, p->K + 1);
    p->nk = NEW(int, p->K);
    for (int k = 0; k < p->K; k++) {
      p->nk[k] = pd->nodes_per_interval[k];
      if (p->nk[k] < 2) {
        fail(err, errlen, "nodes per interval must be >= 2");
        orpm_destroy(o);
        return NULL;
      }
    }
    build_phase_tables(p);
    if (p->diag_nnz != p->N) {
      fail(err, errlen, "differentiation matrix has a zero diagonal entry");
      orpm_destroy(o);
      return NULL;
    }
  }
  for (int i = 0; i < o->L; i++) {
    const rpm_link_desc* ld = &d->links[i];
    o->lk[i].left = ld->left_phase - 1; /* Linkage::LeftPhase(), LpOptimalProblem.hpp:264-269 */
    o->lk[i].right = ld->right_phase - 1;
    o->lk[i].nlink = ld->n_links;
    o->lk[i].lmin = orpm_dupd(ld->link_min, ld->n_links);
    o->lk[i].lmax = orpm_dupd(ld->link_max, ld->n_links);
  }

  /* ---- bounds and layout (LpBoundsChecker::GetBounds, Core/LpBoundsChecker.cpp:13-348) ---- */
  int n = 0, mnl = 0;
  for (int i = 0; i < o->P; i++) {
    ophase* p = &o->ph[i];
    p->nvar = p->nx * (p->N + 1) + p->nu * p->N + 2 + p->nq;
    p->ncon = p->nx * p->N + p->nc * p->N + p->ne;
    p->var0 = n;
    p->con0 = mnl;
    p->state0 = n;
    p->control0 = n + p->nx * (p->N + 1);
    p->t0_idx = p->control0 + p->nu * p->N;
    p->tf_idx = p->t0_idx + 1;
    p->param0 = p->tf_idx + 1;
    n += p->nvar;
    mnl += p->ncon;
  }
  int nlinks = 0;
  for (int i = 0; i < o->L; i++) nlinks += o->lk[i].nlink;
  mnl += nlinks;
  o->n = n;
  o->m_nl = mnl; /* conbounds_min.size() */
  o->m = mnl + o->P + o->L;
  o->xl = NEW(double, n);
  o->xu = NEW(double, n);
  o->gl = NEW(double, o->m);
  o->gu = NEW(double, o->m);
  int vi = 0, ci = 0;
  for (int i = 0; i < o->P; i++) {
    const rpm_phase_desc* pd = &d->phases[i];
    ophase* p = &o->ph[i];
    int nodes = p->N;
    for (int j = 0; j < p->nx; j++) { /* :51-86 */
      const double* mn = pd->state_min + 3 * j;
      const double* mx = pd->state_max + 3 * j;
      if (!(mn[0] <= mx[0] && mn[1] <= mx[1] && mn[2] <= mx[2])) {
        fail(err, errlen, "Bounds on State are Inconsistent (i.e. max < min)");
        orpm_destroy(o);
        return NULL;
      }
      o->xl[vi] = mn[0];
      o->xu[vi++] = mx[0];
      o->gl[ci] = 0;
      o->gu[ci++] = 0;
      for (int k = 1; k < nodes; k++) {
        o->xl[vi] = mn[1];
        o->xu[vi++] = mx[1];
        o->gl[ci] = 0;
        o->gu[ci++] = 0;
      }
      o->xl[vi] = mn[2];
      o->xu[vi++] = mx[2];
    }
    for (int j = 0; j < p->nu; j++) { /* :90-110 */
      if (!(pd->control_min[j] <= pd->control_max[j])) {
        fail(err, errlen, "Bounds on Control are Inconsistent (i.e. max < min)");
        orpm_destroy(o);
        return NULL;
      }
      for (int k = 0; k < nodes; k++) {
        o->xl[vi] = pd->control_min[j];
        o->xu[vi++] = pd->control_max[j];
      }
    }
    o->xl[vi] = pd->t0_min; /* :111-116 */
    o->xu[vi++] = pd->t0_max;
    o->xl[vi] = pd->tf_min;
    o->xu[vi++] = pd->tf_max;
    for (int j = 0; j < p->nq; j++) { /* :117-138 */
      if (!(pd->parameter_min[j] <= pd->parameter_max[j])) {
        fail(err, errlen, "Bounds on parameter are Inconsistent (i.e. max < min)");
        orpm_destroy(o);
        return NULL;
      }
      o->xl[vi] = pd->parameter_min[j];
      o->xu[vi++] = pd->parameter_max[j];
    }
    for (int j = 0; j < p->nc; j++) { /* :141-162 */
      if (!(pd->path_min[j] <= pd->path_max[j])) {
        fail(err, errlen, "Bounds on path are Inconsistent (i.e. max < min)");
        orpm_destroy(o);
        return NULL;
      }
      for (int k = 0; k < nodes; k++) {
        o->gl[ci] = pd->path_min[j];
        o->gu[ci++] = pd->path_max[j];
      }
    }
    for (int j = 0; j < p->ne; j++) { /* :164-186 */
      if (!(pd->event_min[j] <= pd->event_max[j])) {
        fail(err, errlen, "Bounds on event are Inconsistent (i.e. max < min)");
        orpm_destroy(o);
        return NULL;
      }
      o->gl[ci] = pd->event_min[j];
      o->gu[ci++] = pd->event_max[j];
    }
  }
  for (int i = 0; i < o->L; i++) /* :228-253 */
    for (int j = 0; j < o->lk[i].nlink; j++) {
      if (!(o->lk[i].lmin[j] <= o->lk[i].lmax[j])) {
        fail(err, errlen, "Bounds on linkage are Inconsistent (i.e. max < min)");
        orpm_destroy(o);
        return NULL;
      }
      o->gl[ci] = o->lk[i].lmin[j];
      o->gu[ci++] = o->lk[i].lmax[j];
    }
  /* linear constraints A_lin, :265-346 */
  o->alin_nnz = 2 * (o->P + o->L);
  o->alin_i = NEW(int, o->alin_nnz);
  o->alin_j = NEW(int, o->alin_nnz);
  o->alin_v = NEW(double, o->alin_nnz);
  o->linmin = NEW(double, o->P + o->L);
  o->linmax = NEW(double, o->P + o->L);
  int a = 0;
  for (int i = 0; i < o->P; i++) {
    const rpm_phase_desc* pd = &d->phases[i];
    o->alin_i[a] = i;
    o->alin_j[a] = o->ph[i].t0_idx;
    o->alin_v[a++] = -1;
    o->alin_i[a] = i;
    o->alin_j[a] = o->ph[i].tf_idx;
    o->alin_v[a++] = 1;
    if (pd->has_duration) {
      if (!(pd->duration_min <= pd->duration_max)) {
        fail(err, errlen, "Bounds on duration are Inconsistent (i.e. max < min)");
        orpm_destroy(o);
        return NULL;
      }
      o->linmin[i] = pd->duration_min;
      o->linmax[i] = pd->duration_max;
    } else {
      o->linmin[i] = 0;
      o->linmax[i] = INFINITY;
    }
  }
  for (int i = 0; i < o->L; i++) {
    o->alin_i[a] = o->P + i;
    o->alin_j[a] = o->ph[o->lk[i].left].tf_idx;
    o->alin_v[a++] = -1;
    o->alin_i[a] = o->P + i;
    o->alin_j[a] = o->ph[o->lk[i].right].t0_idx;
    o->alin_v[a++] = 1;
    o->linmin[o->P + i] = 0;
    o->linmax[o->P + i] = 0;
  }
  for (int i = 0; i < o->P + o->L; i++) { /* LpopcIpopt::get_bounds_info, Core/LpopcIpopt.cpp:71-79 */
    o->gl[o->m_nl + i] = o->linmin[i];
    o->gu[o->m_nl + i] = o->linmax[i];
  }

  /* ---- guess (LpGuessChecker::GetGuess, Core/LpGuessChecker.cpp:11-204) ---- */
  o->guess = NEW(double, n);
  for (int i = 0; i < o->P; i++) {
    const rpm_phase_desc* pd = &d->phases[i];
    ophase* p = &o->ph[i];
    int ng = pd->n_guess;
    if (ng < 2) {
      fail(err, errlen, "Guess must have a least two points");
      orpm_destroy(o);
      return NULL;
    }
    for (int q = 1; q < ng; q++)
      if (pd->time_guess[q] == pd->time_guess[0]) {
        fail(err, errlen, "Guess for time does not contain unique values");
        orpm_destroy(o);
        return NULL;
      }
    double t0G = pd->time_guess[0], tfG = pd->time_guess[ng - 1];
    double* tauG = NEW(double, ng);
    for (int q = 0; q < ng; q++) tauG[q] = 2 * (pd->time_guess[q] - t0G) / (tfG - t0G) - 1;
    double* g = o->guess + p->var0;
    int r = 0;
    for (int j = 0; j < p->nx; j++) {
      for (int k = 0; k < p->N; k++) g[r++] = orpm_spline_interp(p->points[k], tauG, pd->state_guess + (size_t)j * ng, ng);
      g[r++] = orpm_spline_interp(1.0, tauG, pd->state_guess + (size_t)j * ng, ng);
    }
    for (int j = 0; j < p->nu; j++)
      for (int k = 0; k < p->N; k++) g[r++] = orpm_spline_interp(p->points[k], tauG, pd->control_guess + (size_t)j * ng, ng);
    g[r++] = t0G;
    g[r++] = tfG;
    for (int j = 0; j < p->nq; j++) g[r++] = pd->parameter_guess[j]; /* LpGuessChecker.cpp:186-189 */
    free(tauG);
  }

  /* ---- Jacobian counts (NLPWrapper::GetWholeSparsity, Core/LpNLPWrapper.cpp:1332-1374) ---- */
  o->nnz_nl = 0;
  o->nnz_const = 0;
  for (int i = 0; i < o->P; i++) {
    ophase* p = &o->ph[i];
    int ndep = (p->nx + p->nc) * (p->nx + p->nu); /* dependencies.fill(1), :1345 */
    o->nnz_nl += ndep * p->N + 2 * (p->nx + p->nc) * p->N + (p->nx + p->nc) * p->nq * p->N +
                 p->ne * (2 * p->nx + p->nq + 2);
    o->nnz_const += p->off_nnz * p->nx;
  }
  for (int i = 0; i < o->L; i++) {
    /* :1362-1373 queries the LEFT phase twice (SURVEY B-11) */
    ophase* pl = &o->ph[o->lk[i].left];
    if (pl->nq != o->ph[o->lk[i].right].nq || pl->nx != o->ph[o->lk[i].right].nx) {
      fail(err, errlen, "linked phases must have equal nx and nq (SURVEY B-11)");
      orpm_destroy(o);
      return NULL;
    }
    o->nnz_nl += o->lk[i].nlink * (pl->nx + pl->nq + pl->nx + pl->nq);
  }
  o->nnz_lin = o->alin_nnz;
  o->nnz = o->nnz_nl + o->nnz_lin + o->nnz_const;
  o->hess = (o->hessian_mode == RPM_HESSIAN_EXACT) ? orpm_hess_create(o) : NULL;
  return o;
}

void orpm_destroy(orpm* o) {
  if (!o) return;
  if (o->hess) orpm_hess_destroy(o->hess);
  for (int i = 0; i < o->P; i++) {
    ophase* p = &o->ph[i];
    free(p->mesh);
    free(p->nk);
    free(p->points);
    free(p->weights);
    free(p->d_i);
    free(p->d_j);
    free(p->d_v);
    free(p->diag_v);
    free(p->off_i);
    free(p->off_j);
    free(p->off_v);
    free(p->fr_off);
    free(p->fr_col0);
    free(p->fr_len);
    free(p->fr_vals);
    free(p->fair_fv);
  }
  for (int i = 0; i < o->L; i++) {
    free(o->lk[i].lmin);
    free(o->lk[i].lmax);
  }
  free(o->ph);
  free(o->lk);
  free(o->consts);
  free(o->xl);
  free(o->xu);
  free(o->gl);
  free(o->gu);
  free(o->guess);
  free(o->alin_i);
  free(o->alin_j);
  free(o->alin_v);
  free(o->linmin);
  free(o->linmax);
  free(o);
}

int orpm_hess_nnz(void* h);

/* LpopcIpopt::get_nlp_info, Core/LpopcIpopt.cpp:11-24 */
void orpm_get_nlp_info(const orpm* o, int* n, int* m, int* nnz_jac_g, int* nnz_h_lag) {
  if (n) *n = o->n;
  if (m) *m = o->m;
  if (nnz_jac_g) *nnz_jac_g = o->nnz;
  if (nnz_h_lag) *nnz_h_lag = o->hess ? orpm_hess_nnz(o->hess) : 0;
}
/* LpopcIpopt::get_bounds_info, Core/LpopcIpopt.cpp:26-82 */
void orpm_get_bounds_info(const orpm* o, double* x_l, double* x_u, double* g_l, double* g_u) {
  memcpy(x_l, o->xl, sizeof(double) * o->n);
  memcpy(x_u, o->xu, sizeof(double) * o->n);
  memcpy(g_l, o->gl, sizeof(double) * o->m);
  memcpy(g_u, o->gu, sizeof(double) * o->m);
}
/* LpopcIpopt::get_starting_point, Core/LpopcIpopt.cpp:84-104 */
void orpm_get_starting_point(const orpm* o, double* x) { memcpy(x, o->guess, sizeof(double) * o->n); }

void orpm_get_phase_sizes(const orpm* o, int phase, int* n_nodes, int* d_nnz, int* doff_nnz) {
  const ophase* p = &o->ph[phase];
  if (n_nodes) *n_nodes = p->N;
  if (d_nnz) *d_nnz = p->d_nnz;
  if (doff_nnz) *doff_nnz = p->off_nnz;
}
void orpm_get_phase_tables(const orpm* o, int phase, double* points, double* weights, int* d_rows,
                           int* d_cols, double* d_vals, double* diag_vals, int* doff_rows,
                           int* doff_cols, double* doff_vals) {
  const ophase* p = &o->ph[phase];
  if (points) memcpy(points, p->points, sizeof(double) * p->N);
  if (weights) memcpy(weights, p->weights, sizeof(double) * p->N);
  if (d_rows) memcpy(d_rows, p->d_i, sizeof(int) * p->d_nnz);
  if (d_cols) memcpy(d_cols, p->d_j, sizeof(int) * p->d_nnz);
  if (d_vals) memcpy(d_vals, p->d_v, sizeof(double) * p->d_nnz);
  if (diag_vals) memcpy(diag_vals, p->diag_v, sizeof(double) * p->N);
  if (doff_rows) memcpy(doff_rows, p->off_i, sizeof(int) * p->off_nnz);
  if (doff_cols) memcpy(doff_cols, p->off_j, sizeof(int) * p->off_nnz);
  if (doff_vals) memcpy(doff_vals, p->off_v, sizeof(double) * p->off_nnz);
}

/* ===========================================================================
 * slicing shared by every callback (e.g. Core/LpNLPWrapper.cpp:69-96)
 * ======================================================================== */
void orpm_slice_phase(const orpm* o, int i, const double* x, pslice* s) {
  const ophase* p = &o->ph[i];
  int N = p->N;
  s->N = N;
  s->nx = p->nx;
  s->nu = p->nu;
  s->nq = p->nq;
  s->nc = p->nc;
  s->ne = p->ne;
  s->t0 = x[p->t0_idx];
  s->tf = x[p->tf_idx];
  s->tspan = s->tf - s->t0;
  s->t_radau = NEW(double, N);
  for (int k = 0; k < N; k++) s->t_radau[k] = (p->points[k] + 1) * (s->tspan / 2.0) + s->t0;
  s->state_matrix = orpm_dupd(x + p->state0, (N + 1) * p->nx);
  s->state_radau = NEW(double, (size_t)N * p->nx);
  s->x0 = NEW(double, p->nx);
  s->xf = NEW(double, p->nx);
  for (int j = 0; j < p->nx; j++) {
    memcpy(s->state_radau + (size_t)j * N, s->state_matrix + (size_t)j * (N + 1), sizeof(double) * N);
    s->x0[j] = s->state_matrix[(size_t)j * (N + 1)];
    s->xf[j] = s->state_matrix[(size_t)j * (N + 1) + N];
  }
  s->control = orpm_dupd(x + p->control0, N * p->nu);
  s->parameter = orpm_dupd(x + p->param0, p->nq); /* :90-96 */
}
void orpm_free_slice(pslice* s) {
  free(s->t_radau);
  free(s->state_matrix);
  free(s->state_radau);
  free(s->control);
  free(s->x0);
  free(s->xf);
  free(s->parameter);
}
void orpm_mk_soldae(const pslice* s, int phase_num, orpm_soldae* d) {
  d->phase_num = phase_num;
  d->N = s->N;
  d->nx = s->nx;
  d->nu = s->nu;
  d->nq = s->nq;
  d->nc = s->nc;
  d->time = s->t_radau;
  d->state = s->state_radau;
  d->control = s->control;
  d->parameter = s->parameter;
}
void orpm_mk_solcost(const pslice* s, int phase_num, orpm_solcost* c) {
  c->phase_num = phase_num;
  c->initial_time = s->t0;
  c->initial_state = s->x0;
  c->terminal_time = s->tf;
  c->terminal_state = s->xf;
  c->N = s->N;
  c->nx = s->nx;
  c->nu = s->nu;
  c->nq = s->nq;
  c->time = s->t_radau;
  c->state = s->state_radau;
  c->control = s->control;
  c->parameter = s->parameter;
}
void orpm_mk_solevent(const pslice* s, int phase_num, orpm_solevent* e) {
  e->phase_num = phase_num;
  e->initial_time = s->t0;
  e->terminal_time = s->tf;
  e->nx = s->nx;
  e->nq = s->nq;
  e->ne = s->ne;
  e->initial_state = s->x0;
  e->terminal_state = s->xf;
  e->parameter = s->parameter;
}

/* ===========================================================================
 * eval_g: NLPWrapper::GetAllCons / GetConsFun, Core/LpNLPWrapper.cpp:34-229
 * ======================================================================== */
void orpm_eval_g(orpm* o, const double* x, double* g) {
  int row = 0;
  double** x0s = NEW(double*, o->P);
  double** xfs = NEW(double*, o->P);
  for (int i = 0; i < o->P; i++) {
    const ophase* p = &o->ph[i];
    pslice s;
    orpm_slice_phase(o, i, x, &s);
    int N = s.N;
    orpm_soldae sd;
    orpm_mk_soldae(&s, i + 1, &sd);
    double* stateout = NEW(double, (size_t)N * p->nx);
    double* pathout = NEW(double, (size_t)N * (p->nc > 0 ? p->nc : 1));
    o->fun->dae(&sd, o->consts, stateout, pathout);                                /* :110 */
    double* odeleft = NEW(double, (size_t)N * p->nx);
    if (o->fair) dense_rows_mul(p, N, s.state_matrix, N + 1, p->nx, odeleft);
    else coo_mul(p->d_nnz, p->d_i, p->d_j, p->d_v, N, s.state_matrix, N + 1, p->nx, odeleft); /* :111 */
    for (int q = 0; q < N * p->nx; q++) g[row + q] = odeleft[q] - stateout[q] * (s.tspan / 2.0); /* :113,122 */
    row += N * p->nx;
    for (int q = 0; q < N * p->nc; q++) g[row + q] = pathout[q];                   /* :138-164 */
    row += N * p->nc;
    if (p->ne > 0) {                                                               /* :125-136 */
      orpm_solevent se;
      orpm_mk_solevent(&s, i + 1, &se);
      double* ev = NEW(double, p->ne);
      o->fun->event(&se, o->consts, ev);
      for (int q = 0; q < p->ne; q++) g[row + q] = ev[q];
      row += p->ne;
      free(ev);
    }
    x0s[i] = orpm_dupd(s.x0, p->nx);
    xfs[i] = orpm_dupd(s.xf, p->nx);
    free(stateout);
    free(pathout);
    free(odeleft);
    orpm_free_slice(&s);
  }
  for (int ip = 0; ip < o->L; ip++) { /* :180-211 */
    const olink* l = &o->lk[ip];
    orpm_sollink sl;
    sl.left_phase_num = l->left; /* 0-based here, 1-based in GetWholeJacbi (SURVEY B-10) */
    sl.right_phase_num = l->right;
    sl.ipair = ip + 1;
    sl.nxl = o->ph[l->left].nx;
    sl.nxr = o->ph[l->right].nx;
    sl.nql = o->ph[l->left].nq;
    sl.nqr = o->ph[l->right].nq;
    sl.nlink = l->nlink;
    sl.left_state = xfs[l->left];
    sl.right_state = x0s[l->right];
    sl.left_parameter = x + o->ph[l->left].param0;
    sl.right_parameter = x + o->ph[l->right].param0; /* the RIGHT phase's (the reference hands p_left twice, :203 - B-9) */
    double* lo = NEW(double, l->nlink);
    o->fun->link(&sl, o->consts, lo);
    for (int q = 0; q < l->nlink; q++) g[row + q] = lo[q];
    row += l->nlink;
    free(lo);
  }
  /* linearCons = AlinearMatrix * y, :45 */
  coo_mul(o->alin_nnz, o->alin_i, o->alin_j, o->alin_v, o->P + o->L, x, o->n, 1, g + row);
  for (int i = 0; i < o->P; i++) {
    free(x0s[i]);
    free(xfs[i]);
  }
  free(x0s);
  free(xfs);
}

/* ===========================================================================
 * finite differences, Core/LpFiniteDifferenceDerive.cpp
 * ======================================================================== */
/* LpFDderive::DerivDae :194-324.  Output dstate [(N nx) x (nx+nu+1+nq)], dpath [(N nc) x (...)],
 * column order [x.., u.., t, p..], rows output-major then node (:299-317). */
static void fd_deriv_dae(orpm* o, const orpm_soldae* base, double* dstate, double* dpath) {
  int N = base->N, nx = base->nx, nu = base->nu, nc = base->nc, nq = base->nq;
  int nout = nx + nc, ncolD = nx + nu + 1 + nq;
  double tol = o->tol;
  double* daeout = NEW(double, (size_t)N * nx);
  double* pathout = NEW(double, (size_t)N * (nc > 0 ? nc : 1));
  o->fun->dae(base, o->consts, daeout, pathout); /* :206 */
  double* pertTime = NEW(double, N);
  double* tPert = NEW(double, N);
  double* pertState = NEW(double, (size_t)N * nx);
  double* statePert = NEW(double, (size_t)N * nx);
  double* pertControl = NEW(double, (size_t)N * (nu > 0 ? nu : 1));
  double* controlPert = NEW(double, (size_t)N * (nu > 0 ? nu : 1));
  for (int k = 0; k < N; k++) { /* :208-214 */
    pertTime[k] = tol * (1 + fabs(base->time[k]));
    tPert[k] = base->time[k] + pertTime[k];
  }
  for (int q = 0; q < N * nx; q++) {
    pertState[q] = tol * (1 + fabs(base->state[q]));
    statePert[q] = base->state[q] + pertState[q];
  }
  for (int q = 0; q < N * nu; q++) {
    pertControl[q] = tol * (1 + fabs(base->control[q]));
    controlPert[q] = base->control[q] + pertControl[q];
  }
  double* pso = NEW(double, (size_t)N * nx);
  double* ppo = NEW(double, (size_t)N * (nc > 0 ? nc : 1));
  double* work_state = orpm_dupd(base->state, N * nx);
  double* work_control = orpm_dupd(base->control, N * nu);
  orpm_soldae sd = *base;
  /* one column of the stacked result: rows [i*N,(i+1)*N) = d out_i / d var at the N nodes */
#define STORE(col, denom)                                                                          \
  for (int i2 = 0; i2 < nout; i2++)                                                                \
    for (int k = 0; k < N; k++) {                                                                  \
      double pert = (i2 < nx) ? pso[k + (size_t)i2 * N] : ppo[k + (size_t)(i2 - nx) * N];          \
      double b0 = (i2 < nx) ? daeout[k + (size_t)i2 * N] : pathout[k + (size_t)(i2 - nx) * N];     \
      double val = (pert - b0) / (denom)[k];                                                       \
      if (i2 < nx)                                                                                 \
        dstate[(k + (size_t)i2 * N) + (size_t)(col) * ((size_t)N * nx)] = val;                     \
      else                                                                                         \
        dpath[(k + (size_t)(i2 - nx) * N) + (size_t)(col) * ((size_t)N * nc)] = val;               \
    }
  /* time, :225-241 */
  sd.time = tPert;
  o->fun->dae(&sd, o->consts, pso, ppo);
  sd.time = base->time;
  STORE(nx + nu, pertTime);
  /* states, :243-259 */
  sd.state = work_state;
  for (int is = 0; is < nx; is++) {
    memcpy(work_state + (size_t)is * N, statePert + (size_t)is * N, sizeof(double) * N);
    o->fun->dae(&sd, o->consts, pso, ppo);
    STORE(is, pertState + (size_t)is * N);
    memcpy(work_state + (size_t)is * N, base->state + (size_t)is * N, sizeof(double) * N);
  }
  /* controls, :261-278 */
  sd.control = work_control;
  for (int ic = 0; ic < nu; ic++) {
    memcpy(work_control + (size_t)ic * N, controlPert + (size_t)ic * N, sizeof(double) * N);
    o->fun->dae(&sd, o->consts, pso, ppo);
    STORE(nx + ic, pertControl + (size_t)ic * N);
    memcpy(work_control + (size_t)ic * N, base->control + (size_t)ic * N, sizeof(double) * N);
  }
  /* static parameters, :280-297: one whole-vector call per parameter, the same step at every node */
  if (nq > 0) {
    double* work_par = orpm_dupd(base->parameter, nq);
    double* pertPar = NEW(double, N);
    sd.control = base->control;
    sd.parameter = work_par;
    for (int ip = 0; ip < nq; ip++) {
      double hp = tol * (1 + fabs(base->parameter[ip]));
      for (int k = 0; k < N; k++) pertPar[k] = hp;
      work_par[ip] = base->parameter[ip] + hp;
      o->fun->dae(&sd, o->consts, pso, ppo);
      STORE(nx + nu + 1 + ip, pertPar);
      work_par[ip] = base->parameter[ip];
    }
    free(work_par);
    free(pertPar);
  }
#undef STORE
  (void)ncolD;
  free(daeout);
  free(pathout);
  free(pertTime);
  free(tPert);
  free(pertState);
  free(statePert);
  free(pertControl);
  free(controlPert);
  free(pso);
  free(ppo);
  free(work_state);
  free(work_control);
}

/* LpFDderive::DerivEvent :326-409.  Output [ne x (2nx+2+nq)] = [x0.., t0, xf.., tf, p..], column-major. */
static void fd_deriv_event(orpm* o, const orpm_solevent* base, double* d) {
  int nx = base->nx, ne = base->ne, nq = base->nq;
  double tol = o->tol;
  double pert0 = tol * (1 + fabs(base->initial_time));
  double pertf = tol * (1 + fabs(base->terminal_time));
  double* ev = NEW(double, ne);
  double* pe = NEW(double, ne);
  double* x0 = orpm_dupd(base->initial_state, nx);
  double* xf = orpm_dupd(base->terminal_state, nx);
  orpm_solevent se = *base;
  se.initial_state = x0;
  se.terminal_state = xf;
  o->fun->event(&se, o->consts, ev);
  se.initial_time = base->initial_time + pert0; /* :355-358 */
  o->fun->event(&se, o->consts, pe);
  for (int q = 0; q < ne; q++) d[q + (size_t)nx * ne] = (pe[q] - ev[q]) / pert0;
  se.initial_time = base->initial_time;
  se.terminal_time = base->terminal_time + pertf; /* :361-365 */
  o->fun->event(&se, o->consts, pe);
  for (int q = 0; q < ne; q++) d[q + (size_t)(2 * nx + 1) * ne] = (pe[q] - ev[q]) / pertf;
  se.terminal_time = base->terminal_time;
  for (int is = 0; is < nx; is++) { /* :372-383 */
    double px0 = tol * (fabs(base->initial_state[is]) + 1);
    double pxf = tol * (fabs(base->terminal_state[is]) + 1);
    x0[is] = base->initial_state[is] + px0;
    o->fun->event(&se, o->consts, pe);
    for (int q = 0; q < ne; q++) d[q + (size_t)is * ne] = (pe[q] - ev[q]) / (px0 * 1.0);
    x0[is] = base->initial_state[is];
    xf[is] = base->terminal_state[is] + pxf;
    o->fun->event(&se, o->consts, pe);
    for (int q = 0; q < ne; q++) d[q + (size_t)(nx + 1 + is) * ne] = (pe[q] - ev[q]) / (pxf * 1.0);
    xf[is] = base->terminal_state[is];
  }
  if (nq > 0) { /* :385-400 */
    double* wp = orpm_dupd(base->parameter, nq);
    se.parameter = wp;
    for (int ip = 0; ip < nq; ip++) {
      double hp = tol * (1 + fabs(base->parameter[ip]));
      wp[ip] = base->parameter[ip] + hp;
      o->fun->event(&se, o->consts, pe);
      for (int q = 0; q < ne; q++) d[q + (size_t)(2 * nx + 2 + ip) * ne] = (pe[q] - ev[q]) / hp;
      wp[ip] = base->parameter[ip];
    }
    free(wp);
  }
  free(ev);
  free(pe);
  free(x0);
  free(xf);
}

/* LpFDderive::DerivLink :411-502.  Output [nlink x (nxl+nql+nxr+nqr)] = [xf_left.., p_left.., x0_right.., p_right..]. */
static void fd_deriv_link(orpm* o, const orpm_sollink* base, double* d) {
  int nl = base->nlink, nxl = base->nxl, nxr = base->nxr, nql = base->nql, nqr = base->nqr;
  double tol = o->tol;
  double* lo = NEW(double, nl);
  double* pl = NEW(double, nl);
  double* xl = orpm_dupd(base->left_state, nxl);
  double* xr = orpm_dupd(base->right_state, nxr);
  orpm_sollink sl = *base;
  sl.left_state = xl;
  sl.right_state = xr;
  o->fun->link(&sl, o->consts, lo);
  for (int is = 0; is < nxl; is++) {
    double pert = tol * (1 + fabs(base->left_state[is]));
    xl[is] = base->left_state[is] + pert;
    o->fun->link(&sl, o->consts, pl);
    for (int q = 0; q < nl; q++) d[q + (size_t)is * nl] = (pl[q] - lo[q]) / (1.0 * pert);
    xl[is] = base->left_state[is];
  }
  for (int is = 0; is < nxr; is++) {
    double pert = tol * (1 + fabs(base->right_state[is]));
    xr[is] = base->right_state[is] + pert;
    o->fun->link(&sl, o->consts, pl);
    for (int q = 0; q < nl; q++) d[q + (size_t)(nxl + nql + is) * nl] = (pl[q] - lo[q]) / (1.0 * pert);
    xr[is] = base->right_state[is];
  }
  if (nql + nqr > 0) { /* :455-500; each parameter by its own step (SURVEY B-9) */
    double* wl = orpm_dupd(base->left_parameter, nql);
    double* wr = orpm_dupd(base->right_parameter, nqr);
    sl.left_parameter = wl;
    sl.right_parameter = wr;
    for (int ip = 0; ip < nql; ip++) {
      double pert = tol * (1 + fabs(base->left_parameter[ip]));
      wl[ip] = base->left_parameter[ip] + pert;
      o->fun->link(&sl, o->consts, pl);
      for (int q = 0; q < nl; q++) d[q + (size_t)(nxl + ip) * nl] = (pl[q] - lo[q]) / (1.0 * pert);
      wl[ip] = base->left_parameter[ip];
    }
    for (int ip = 0; ip < nqr; ip++) {
      double pert = tol * (1 + fabs(base->right_parameter[ip]));
      wr[ip] = base->right_parameter[ip] + pert;
      o->fun->link(&sl, o->consts, pl);
      for (int q = 0; q < nl; q++) d[q + (size_t)(nxl + nql + nxr + ip) * nl] = (pl[q] - lo[q]) / (1.0 * pert);
      wr[ip] = base->right_parameter[ip];
    }
    free(wl);
    free(wr);
  }
  free(lo);
  free(pl);
  free(xl);
  free(xr);
}

/* LpFDderive::DerivMayer :11-98.  Output [1 x (2nx+2+nq)] = [x0.., t0, xf.., tf, p..]. */
static void fd_deriv_mayer(orpm* o, const orpm_solcost* base, double* d) {
  int nx = base->nx, nq = base->nq;
  double tol = o->tol;
  double pert0 = tol * (1 + fabs(base->initial_time));
  double pertf = tol * (1 + fabs(base->terminal_time));
  double* x0 = orpm_dupd(base->initial_state, nx);
  double* xf = orpm_dupd(base->terminal_state, nx);
  orpm_solcost sc = *base;
  sc.initial_state = x0;
  sc.terminal_state = xf;
  double m0 = 0, mp = 0;
  o->fun->mayer(&sc, o->consts, &m0);
  sc.initial_time = base->initial_time + pert0;
  o->fun->mayer(&sc, o->consts, &mp);
  d[nx] = (mp - m0) / pert0;
  sc.initial_time = base->initial_time;
  sc.terminal_time = base->terminal_time + pertf;
  o->fun->mayer(&sc, o->consts, &mp);
  d[2 * nx + 1] = (mp - m0) / pertf;
  sc.terminal_time = base->terminal_time;
  for (int is = 0; is < nx; is++) {
    double px0 = tol * (1 + fabs(base->initial_state[is]));
    double pxf = tol * (1 + fabs(base->terminal_state[is]));
    x0[is] = base->initial_state[is] + px0;
    o->fun->mayer(&sc, o->consts, &mp);
    d[is] = (mp - m0) / px0;
    x0[is] = base->initial_state[is];
    xf[is] = base->terminal_state[is] + pxf;
    o->fun->mayer(&sc, o->consts, &mp);
    d[nx + 1 + is] = (mp - m0) / pxf;
    xf[is] = base->terminal_state[is];
  }
  if (nq > 0) { /* :74-90 */
    double* wp = orpm_dupd(base->parameter, nq);
    sc.parameter = wp;
    for (int ip = 0; ip < nq; ip++) {
      double hp = tol * (1 + fabs(base->parameter[ip]));
      wp[ip] = base->parameter[ip] + hp;
      o->fun->mayer(&sc, o->consts, &mp);
      d[2 * nx + 2 + ip] = (mp - m0) / hp;
      wp[ip] = base->parameter[ip];
    }
    free(wp);
  }
  free(x0);
  free(xf);
}

/* LpFDderive::DerivLagrange :100-192.  Output [N x (nx+nu+1+nq)] = [x.., u.., t, p..]. */
static void fd_deriv_lagrange(orpm* o, const orpm_solcost* base, double* d) {
  int N = base->N, nx = base->nx, nu = base->nu, nq = base->nq;
  double tol = o->tol;
  double* L0 = NEW(double, N);
  double* Lp = NEW(double, N);
  double* ws = orpm_dupd(base->state, N * nx);
  double* wc = orpm_dupd(base->control, N * nu);
  double* wt = orpm_dupd(base->time, N);
  orpm_solcost sc = *base;
  o->fun->lagrange(&sc, o->consts, L0);
  for (int k = 0; k < N; k++) wt[k] = base->time[k] + tol * (1 + fabs(base->time[k]));
  sc.time = wt;
  o->fun->lagrange(&sc, o->consts, Lp);
  for (int k = 0; k < N; k++) d[k + (size_t)(nx + nu) * N] = (Lp[k] - L0[k]) / (tol * (1 + fabs(base->time[k])));
  sc.time = base->time;
  sc.state = ws;
  for (int is = 0; is < nx; is++) {
    for (int k = 0; k < N; k++) {
      double b = base->state[k + (size_t)is * N];
      ws[k + (size_t)is * N] = b + tol * (1 + fabs(b));
    }
    o->fun->lagrange(&sc, o->consts, Lp);
    for (int k = 0; k < N; k++) {
      double b = base->state[k + (size_t)is * N];
      d[k + (size_t)is * N] = (Lp[k] - L0[k]) / (tol * (1 + fabs(b)));
      ws[k + (size_t)is * N] = b;
    }
  }
  sc.control = wc;
  for (int ic = 0; ic < nu; ic++) {
    for (int k = 0; k < N; k++) {
      double b = base->control[k + (size_t)ic * N];
      wc[k + (size_t)ic * N] = b + tol * (1 + fabs(b));
    }
    o->fun->lagrange(&sc, o->consts, Lp);
    for (int k = 0; k < N; k++) {
      double b = base->control[k + (size_t)ic * N];
      d[k + (size_t)(nx + ic) * N] = (Lp[k] - L0[k]) / (tol * (1 + fabs(b)));
      wc[k + (size_t)ic * N] = b;
    }
  }
  if (nq > 0) { /* :165-182 */
    double* wp = orpm_dupd(base->parameter, nq);
    sc.control = base->control;
    sc.parameter = wp;
    for (int ip = 0; ip < nq; ip++) {
      double hp = tol * (1 + fabs(base->parameter[ip]));
      wp[ip] = base->parameter[ip] + hp;
      o->fun->lagrange(&sc, o->consts, Lp);
      for (int k = 0; k < N; k++) d[k + (size_t)(nx + nu + 1 + ip) * N] = (Lp[k] - L0[k]) / hp;
      wp[ip] = base->parameter[ip];
    }
    free(wp);
  }
  free(L0);
  free(Lp);
  free(ws);
  free(wc);
  free(wt);
}

/* derive_->Deriv*: finite differences or the user's analytic callbacks
 * (Core/LpAnalyticDerive.hpp:24-48) */
void orpm_deriv_dae(orpm* o, const orpm_soldae* sd, double* dstate, double* dpath) {
  if (o->first_derive == RPM_DERIVE_ANALYTIC && o->fun->deriv_dae)
    o->fun->deriv_dae(sd, o->consts, dstate, dpath);
  else
    fd_deriv_dae(o, sd, dstate, dpath);
}
static void deriv_event(orpm* o, const orpm_solevent* se, double* d) {
  if (o->first_derive == RPM_DERIVE_ANALYTIC && o->fun->deriv_event)
    o->fun->deriv_event(se, o->consts, d);
  else
    fd_deriv_event(o, se, d);
}
static void deriv_link(orpm* o, const orpm_sollink* sl, double* d) {
  if (o->first_derive == RPM_DERIVE_ANALYTIC && o->fun->deriv_link)
    o->fun->deriv_link(sl, o->consts, d);
  else
    fd_deriv_link(o, sl, d);
}
static void deriv_mayer(orpm* o, const orpm_solcost* sc, double* d) {
  if (o->first_derive == RPM_DERIVE_ANALYTIC && o->fun->deriv_mayer)
    o->fun->deriv_mayer(sc, o->consts, d);
  else
    fd_deriv_mayer(o, sc, d);
}
void orpm_deriv_lagrange(orpm* o, const orpm_solcost* sc, double* d) {
  if (o->first_derive == RPM_DERIVE_ANALYTIC && o->fun->deriv_lagrange)
    o->fun->deriv_lagrange(sc, o->consts, d);
  else
    fd_deriv_lagrange(o, sc, d);
}

/* ===========================================================================
 * eval_jac_g values: GetConsJacbi / GetWholeJacbi / GetPhaseJacbi,
 * Core/LpNLPWrapper.cpp:230-862
 * ======================================================================== */
static int phase_jac(orpm* o, int iphase, const double* x, double* SV, double* SC) {
  const ophase* p = &o->ph[iphase];
  pslice s;
  orpm_slice_phase(o, iphase, x, &s);
  int N = s.N, nx = p->nx, nu = p->nu, nc = p->nc, ne = p->ne, nq = p->nq;
  int ncolD = nx + nu + 1 + nq;
  double t0 = s.t0, tf = s.tf;
  orpm_soldae sd;
  orpm_mk_soldae(&s, iphase + 1, &sd);
  double* dDaeOut = NEW(double, (size_t)N * nx * ncolD);
  double* dPathOut = NEW(double, (size_t)N * (nc > 0 ? nc : 1) * ncolD);
  orpm_deriv_dae(o, &sd, dDaeOut, dPathOut);                        /* :569 */
  double* daeOut = NEW(double, (size_t)N * nx);
  double* pathOut = NEW(double, (size_t)N * (nc > 0 ? nc : 1));
  o->fun->dae(&sd, o->consts, daeOut, pathOut);                /* :572 */
  /* dDae_state[j].col(i)[k] = dDaeOut[(k + i*N) + j*(N*nx)]  (:586) */
#define DDAE(i, v, k) dDaeOut[((k) + (size_t)(i)*N) + (size_t)(v) * ((size_t)N * nx)]
#define DPATH(i, v, k) dPathOut[((k) + (size_t)(i)*N) + (size_t)(v) * ((size_t)N * nc)]
  double* dEventOut = NULL;
  if (ne > 0) { /* :638-669 */
    orpm_solevent se;
    orpm_mk_solevent(&s, iphase + 1, &se);
    dEventOut = NEW(double, (size_t)ne * (2 * nx + 2 + nq));
    deriv_event(o, &se, dEventOut);
  }
  /* per-call Find of the off-diagonal matrix, :685-687 (the fair shape found it once) */
  double *fi = NULL, *fj = NULL, *fv = NULL;
  int nzoff;
  if (o->fair) {
    nzoff = p->fair_nzoff;
  } else {
    nzoff = coo_find(p->off_nnz, p->off_i, p->off_j, p->off_v, &fi, &fj, &fv);
  }
  const double* fvv = o->fair ? p->fair_fv : fv;
  int sh = 0;
  for (int i = 0; i < nx; i++) { /* :695-770 */
    for (int j = 0; j < nx; j++) {
      if (i == j) {
        for (int k = 0; k < N; k++) SV[sh + k] = p->diag_v[k] - DDAE(i, j, k) * (tf - t0) / 2.0; /* :712 */
        sh += N;
        memcpy(SC + (size_t)i * nzoff, fvv, sizeof(double) * nzoff);                             /* :717-718 */
      } else {
        for (int k = 0; k < N; k++) SV[sh + k] = -(DDAE(i, j, k) * (tf - t0) / 2.0);            /* :725-726 */
        sh += N;
      }
    }
    for (int j = 0; j < nu; j++) { /* :733-743 */
      for (int k = 0; k < N; k++) SV[sh + k] = -(DDAE(i, nx + j, k) * (tf - t0) / 2.0);
      sh += N;
    }
    for (int k = 0; k < N; k++) { /* d/dt0 :748-752 */
      double ret = daeOut[k + (size_t)i * N] * (0.5);
      double ret2 = -(p->points[k] * 0.5) + 0.5;
      ret -= ret2 * (DDAE(i, nx + nu, k) * (tf - t0) / 2.0);
      SV[sh + k] = ret;
    }
    sh += N;
    for (int k = 0; k < N; k++) { /* d/dtf :756-760 (sign as in the reference, SURVEY B-5) */
      double ret = -daeOut[k + (size_t)i * N] * (0.5);
      double ret2 = (p->points[k] * 0.5) + 0.5;
      ret = ret + ret2 * (DDAE(i, nx + nu, k) * (tf - t0) / 2.0);
      SV[sh + k] = ret;
    }
    sh += N;
    for (int j = 0; j < nq; j++) { /* d/dp_j :763-769, from the parameter's own derivative column (B-6) */
      for (int k = 0; k < N; k++) SV[sh + k] = -(DDAE(i, nx + nu + 1 + j, k) * (tf - t0) / 2.0);
      sh += N;
    }
  }
  for (int i = 0; i < nc; i++) { /* :773-820 */
    for (int j = 0; j < nx; j++) {
      for (int k = 0; k < N; k++) SV[sh + k] = DPATH(i, j, k);
      sh += N;
    }
    for (int j = 0; j < nu; j++) {
      for (int k = 0; k < N; k++) SV[sh + k] = DPATH(i, nx + j, k);
      sh += N;
    }
    for (int k = 0; k < N; k++) SV[sh + k] = (-(p->points[k] * 0.5) + 0.5) * DPATH(i, nx + nu, k);
    sh += N;
    for (int k = 0; k < N; k++) SV[sh + k] = ((p->points[k] * 0.5) + 0.5) * DPATH(i, nx + nu, k);
    sh += N;
    for (int j = 0; j < nq; j++) { /* dc/dp_j :814-820, the PATH derivative (B-7) */
      for (int k = 0; k < N; k++) SV[sh + k] = DPATH(i, nx + nu + 1 + j, k);
      sh += N;
    }
  }
  for (int i = 0; i < ne; i++) { /* :833-861 */
    for (int j = 0; j < nx; j++) {
      SV[sh++] = dEventOut[i + (size_t)j * ne];            /* dEvent_x0[j](i) */
      SV[sh++] = dEventOut[i + (size_t)(j + nx + 1) * ne]; /* dEvent_xf[j](i) */
    }
    SV[sh++] = dEventOut[i + (size_t)nx * ne];           /* t0 */
    SV[sh++] = dEventOut[i + (size_t)(2 * nx + 1) * ne]; /* tf */
    for (int j = 0; j < nq; j++) SV[sh++] = dEventOut[i + (size_t)(2 * nx + 2 + j) * ne]; /* :854-859 */
  }
#undef DDAE
#undef DPATH
  free(dDaeOut);
  free(dPathOut);
  free(daeOut);
  free(pathOut);
  free(dEventOut);
  free(fi);
  free(fj);
  free(fv);
  orpm_free_slice(&s);
  return sh;
}

void orpm_set_cost_shape(orpm* o, int fair) {
  o->fair = fair ? 1 : 0;
  if (!o->fair) return;
  for (int ip = 0; ip < o->P; ip++) {
    ophase* p = &o->ph[ip];
    if (p->fr_off) continue;
    int N = p->N;
    p->fr_off = NEW(int, N);
    p->fr_col0 = NEW(int, N);
    p->fr_len = NEW(int, N);
    for (int i = 0; i < N; i++) { p->fr_col0[i] = 1 << 30; p->fr_len[i] = 0; }
    for (int k = 0; k < p->d_nnz; k++) {
      int i = p->d_i[k], j = p->d_j[k];
      if (j < p->fr_col0[i]) p->fr_col0[i] = j;
      p->fr_len[i]++;
    }
    int off = 0;
    for (int i = 0; i < N; i++) { p->fr_off[i] = off; off += p->fr_len[i]; }
    p->fr_vals = NEW(double, off);
    for (int k = 0; k < p->d_nnz; k++) p->fr_vals[p->fr_off[p->d_i[k]] + (p->d_j[k] - p->fr_col0[p->d_i[k]])] = p->d_v[k];
    double *fi, *fj;
    p->fair_nzoff = coo_find(p->off_nnz, p->off_i, p->off_j, p->off_v, &fi, &fj, &p->fair_fv);
    free(fi);
    free(fj);
  }
}

void orpm_eval_jac_g(orpm* o, const double* x, double* values) {
  /* GetWholeJacbi counts with a Find(Doffdiag) per phase, :278-307 */
  for (int i = 0; i < o->P && !o->fair; i++) {
    double *fi, *fj, *fv;
    coo_find(o->ph[i].off_nnz, o->ph[i].off_i, o->ph[i].off_j, o->ph[i].off_v, &fi, &fj, &fv);
    free(fi);
    free(fj);
    free(fv);
  }
  double* NL = values;
  double* LV = values + o->nnz_nl;
  double* CV = values + o->nnz_nl + o->nnz_lin; /* Sjac_V=[NL_V; L_V; C_V], :244-252 */
  int sj = 0, sc = 0;
  for (int i = 0; i < o->P; i++) { /* :329-349 */
    sj += phase_jac(o, i, x, NL + sj, CV + sc);
    sc += o->ph[i].off_nnz * o->ph[i].nx;
  }
  for (int ip = 0; ip < o->L; ip++) { /* :406-522 */
    const olink* l = &o->lk[ip];
    const ophase* pl = &o->ph[l->left];
    const ophase* pr = &o->ph[l->right];
    double* xfl = NEW(double, pl->nx);
    double* x0r = NEW(double, pr->nx);
    for (int j = 0; j < pl->nx; j++) xfl[j] = x[pl->state0 + j * (pl->N + 1) + pl->N];
    for (int j = 0; j < pr->nx; j++) x0r[j] = x[pr->state0 + j * (pr->N + 1)];
    orpm_sollink sl;
    sl.left_phase_num = l->left + 1;
    sl.right_phase_num = l->right + 1;
    sl.ipair = ip + 1;
    sl.nxl = pl->nx;
    sl.nxr = pr->nx;
    sl.nql = pl->nq;
    sl.nqr = pr->nq;
    sl.nlink = l->nlink;
    sl.left_state = xfl;
    sl.right_state = x0r;
    sl.left_parameter = x + pl->param0;
    sl.right_parameter = x + pr->param0;
    int ncl = pl->nx + pl->nq + pr->nx + pr->nq;
    double* dL = NEW(double, (size_t)l->nlink * ncl);
    deriv_link(o, &sl, dL);
    /* column-major walk over DLink_xf_left, DLink_p_left, DLink_x0_Right, DLink_p_right, :461-519 */
    for (int q = 0; q < l->nlink * ncl; q++) NL[sj++] = dL[q];
    free(dL);
    free(xfl);
    free(x0r);
  }
  /* dsmatrix::Find(AlinearMatrix), :242 */
  double *li, *lj, *lv;
  int nl = coo_find(o->alin_nnz, o->alin_i, o->alin_j, o->alin_v, &li, &lj, &lv);
  for (int q = 0; q < nl; q++) LV[q] = lv[q];
  free(li);
  free(lj);
  free(lv);
}

/* ===========================================================================
 * eval_jac_g structure: GetConsSparsity / GetWholeSparsity / GetPhaseSparsity,
 * Core/LpNLPWrapper.cpp:1106-1578
 * ======================================================================== */
void orpm_jac_structure(orpm* o, int* iRow, int* jCol) {
  int sj = 0;
  int sc = o->nnz_nl + o->nnz_lin;
  int rowshift = 0, colshift = 0;
  for (int ip = 0; ip < o->P; ip++) {
    const ophase* p = &o->ph[ip];
    int N = p->N, nx = p->nx, nu = p->nu, nc = p->nc, ne = p->ne, nq = p->nq, disc = N + 1;
#define BLOCK_DIAG(r0, c0)                       \
  for (int k = 0; k < N; k++) {                  \
    iRow[sj] = rowshift + (r0) + k;              \
    jCol[sj++] = colshift + (c0) + k;            \
  }
#define BLOCK_COL(r0, c0)                        \
  for (int k = 0; k < N; k++) {                  \
    iRow[sj] = rowshift + (r0) + k;              \
    jCol[sj++] = colshift + (c0);                \
  }
    for (int i = 0; i < nx; i++) { /* :1150-1214 */
      int rowstart = i * N;
      for (int j = 0; j < nx; j++) {
        BLOCK_DIAG(rowstart, j * disc);
        if (i == j)
          for (int q = 0; q < p->off_nnz; q++) { /* :1164-1166 */
            iRow[sc + i * p->off_nnz + q] = rowshift + p->off_i[q] + rowstart;
            jCol[sc + i * p->off_nnz + q] = colshift + p->off_j[q] + j * disc;
          }
      }
      int cs = nx * disc;
      for (int j = 0; j < nu; j++) BLOCK_DIAG(rowstart, cs + j * N);
      cs += nu * N;
      BLOCK_COL(rowstart, cs);
      BLOCK_COL(rowstart, cs + 1);
      for (int j = 0; j < nq; j++) BLOCK_COL(rowstart, cs + 2 + j); /* every node's row, the parameter's one column (B-8) */
    }
    int rs = nx * N;
    for (int i = 0; i < nc; i++) { /* :1217-1265 */
      int rowstart = rs + i * N;
      for (int j = 0; j < nx; j++) BLOCK_DIAG(rowstart, j * disc);
      int cs = nx * disc;
      for (int j = 0; j < nu; j++) BLOCK_DIAG(rowstart, cs + j * N);
      cs += nu * N;
      BLOCK_COL(rowstart, cs);
      BLOCK_COL(rowstart, cs + 1);
      for (int j = 0; j < nq; j++) BLOCK_COL(rowstart, cs + 2 + j);
    }
    rs = nx * N + nc * N;
    for (int i = 0; i < ne; i++) { /* :1278-1311 */
      int row = rs + i;
      for (int j = 0; j < nx; j++) {
        iRow[sj] = rowshift + row;
        jCol[sj++] = colshift + N * j + j;
        iRow[sj] = rowshift + row;
        jCol[sj++] = colshift + N * (j + 1) + j;
      }
      int cols = nx * (N + 1) + nu * N;
      iRow[sj] = rowshift + row;
      jCol[sj++] = colshift + cols;
      iRow[sj] = rowshift + row;
      jCol[sj++] = colshift + cols + 1;
      for (int j = 0; j < nq; j++) {
        iRow[sj] = rowshift + row;
        jCol[sj++] = colshift + cols + 2 + j;
      }
    }
#undef BLOCK_DIAG
#undef BLOCK_COL
    sc += p->off_nnz * nx;
    rowshift += nx * N + nc * N + ne; /* :1406-1409 */
    colshift += nx * (N + 1) + nu * N + p->nq + 2;
  }
  int linkrow = rowshift;
  for (int ip = 0; ip < o->L; ip++) { /* :1431-1547 */
    const olink* l = &o->lk[ip];
    const ophase* pl = &o->ph[l->left];
    const ophase* pr = &o->ph[l->right];
    for (int jc = 0; jc < pl->nx; jc++)
      for (int ir = 0; ir < l->nlink; ir++) {
        iRow[sj] = ir + linkrow;
        jCol[sj++] = (jc + 1) * pl->N + jc + pl->state0; /* :1488-1489 */
      }
    for (int jc = 0; jc < pl->nq; jc++)
      for (int ir = 0; ir < l->nlink; ir++) {
        iRow[sj] = ir + linkrow;
        jCol[sj++] = pl->param0 + jc; /* :1495-1510 */
      }
    for (int jc = 0; jc < pr->nx; jc++)
      for (int ir = 0; ir < l->nlink; ir++) {
        iRow[sj] = ir + linkrow;
        jCol[sj++] = jc * (pr->N + 1) + pr->state0; /* :1520-1521 */
      }
    for (int jc = 0; jc < pr->nq; jc++)
      for (int ir = 0; ir < l->nlink; ir++) {
        iRow[sj] = ir + linkrow;
        jCol[sj++] = pr->param0 + jc; /* :1527-1543 */
      }
    linkrow += l->nlink;
  }
  /* linear rows, :1559-1570 */
  for (int q = 0; q < o->alin_nnz; q++) {
    iRow[o->nnz_nl + q] = o->alin_i[q] + o->m_nl;
    jCol[o->nnz_nl + q] = o->alin_j[q];
  }
}

/* ===========================================================================
 * eval_f: NLPWrapper::GetObjFun, Core/LpNLPWrapper.cpp:863-939
 * ======================================================================== */
double orpm_eval_f(orpm* o, const double* x) {
  double cost = 0.0;
  for (int i = 0; i < o->P; i++) {
    const ophase* p = &o->ph[i];
    pslice s;
    orpm_slice_phase(o, i, x, &s);
    orpm_solcost sc;
    orpm_mk_solcost(&s, i + 1, &sc);
    double mayer = 0.0;
    double* L = NEW(double, p->N);
    o->fun->mayer(&sc, o->consts, &mayer);
    cost += mayer;
    o->fun->lagrange(&sc, o->consts, L);
    /* trans(Weights)*L*(tspan/2): dot (Armadillo direct_dot order) then the scalar, :931 */
    double integrand = orpm_arma_dot(p->weights, L, p->N) * (s.tspan / 2.0);
    cost += integrand;
    free(L);
    orpm_free_slice(&s);
  }
  return cost;
}

/* ===========================================================================
 * eval_grad_f: NLPWrapper::GetObjGrad, Core/LpNLPWrapper.cpp:940-1104
 * Reference quirks kept bug-for-bug (none bites in the parity domain, where Mayer does
 * not depend on x0 and Lagrange is autonomous):
 *   - Jcost(col0)=dMayer_x0 is overwritten by the Lagrange run that starts at col0 (:1050-1053)
 *   - d/dtf uses `ret3 *= ret2` (an outer product) and keeps only its (0,0) entry (:1085-1087)
 * ======================================================================== */
void orpm_eval_grad_f(orpm* o, const double* x, double* grad_f) {
  memset(grad_f, 0, sizeof(double) * o->n);
  int gs = 0;
  for (int ip = 0; ip < o->P; ip++) {
    const ophase* p = &o->ph[ip];
    pslice s;
    orpm_slice_phase(o, ip, x, &s);
    int N = p->N, nx = p->nx, nu = p->nu, nq = p->nq;
    double tspan = s.tspan;
    orpm_solcost sc;
    orpm_mk_solcost(&s, ip + 1, &sc);
    double* Lout = NEW(double, N);
    double* dM = NEW(double, 2 * nx + 2 + nq);
    double* dL = NEW(double, (size_t)N * (nx + nu + 1 + nq));
    o->fun->lagrange(&sc, o->consts, Lout);  /* :989 */
    deriv_mayer(o, &sc, dM);                 /* :990 */
    orpm_deriv_lagrange(o, &sc, dL);              /* :991 */
    double dMayer_t0 = dM[nx], dMayer_tf = dM[2 * nx + 1];
    const double* dLt = dL + (size_t)(nx + nu) * N; /* the time column (the reference takes the LAST column, :1025: the same for nq = 0) */
    double* J = grad_f + gs;
    for (int j = 0; j < nx; j++) { /* :1045-1055 */
      int col0 = N * j + j, colf = N * (j + 1) + j;
      J[col0] = dM[j];
      for (int k = 0; k < N; k++) J[col0 + k] = (p->weights[k] * tspan / 2.0) * dL[k + (size_t)j * N];
      J[colf] = dM[j + nx + 1];
    }
    int cs = nx * (N + 1);
    for (int j = 0; j < nu; j++) /* :1058-1064 */
      for (int k = 0; k < N; k++)
        J[cs + j * N + k] = (p->weights[k] * tspan / 2.0) * dL[k + (size_t)(nx + j) * N];
    cs += nu * N;
    /* d/dt0, :1069-1078 */
    double* a = NEW(double, N);
    double* r2 = NEW(double, N);
    double* r3 = NEW(double, N);
    for (int k = 0; k < N; k++) {
      a[k] = p->weights[k] * (-0.5);
      r2[k] = (p->weights[k] * (tspan / 2.0)) * dLt[k];
      r3[k] = p->points[k] * (-0.5) + 0.5;
    }
    double ret = orpm_arma_dot(a, Lout, N);
    J[cs] = (orpm_arma_dot(r2, r3, N) + dMayer_t0) + ret;
    /* d/dtf, :1081-1087 */
    for (int k = 0; k < N; k++) a[k] = p->weights[k] * (0.5);
    ret = orpm_arma_dot(a, Lout, N);
    double ret3_00 = (p->points[0] * (0.5) + 0.5) * r2[0];
    J[cs + 1] = dMayer_tf + ret + ret3_00;
    /* d/dp_j, :1088-1097: dMayer/dp_j + sum_k w_k (tspan/2) dL/dp_j (the quadrature the reference means, B-21) */
    for (int j = 0; j < nq; j++) {
      for (int k = 0; k < N; k++) a[k] = p->weights[k] * (tspan / 2.0);
      J[cs + 2 + j] = dM[2 * nx + 2 + j] + orpm_arma_dot(a, dL + (size_t)(nx + nu + 1 + j) * N, N);
    }
    free(a);
    free(r2);
    free(r3);
    free(Lout);
    free(dM);
    free(dL);
    gs += p->nvar;
    orpm_free_slice(&s);
  }
}

