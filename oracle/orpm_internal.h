/* orpm_internal.h — internals shared by orpm_core.c and orpm_hess.c (CPU ORACLE, test infrastructure;
 * PARITY UNPINNED, see orpm.h). */
#ifndef ORPM_INTERNAL_H_
#define ORPM_INTERNAL_H_
#include <stddef.h>

#include "orpm.h"

typedef struct {
  int nx, nu, nq, nc, ne, K, N;
  double* mesh;
  int* nk;
  /* struct ps, Core/LpCalculateData.hpp:35-41 */
  double* points;
  double* weights;
  int d_nnz;
  int *d_i, *d_j;
  double* d_v;
  int diag_nnz;
  double* diag_v;
  int off_nnz;
  int *off_i, *off_j;
  double* off_v;
  /* layout, 0-based absolute indices (phase_indices are 1-based in the reference) */
  int var0, con0, nvar, ncon;
  int state0, control0, t0_idx, tf_idx, param0;
  /* "fair" cost shape only (orpm_set_cost_shape): D as dense rows per node, the Doffdiag value list found once */
  int *fr_off, *fr_col0, *fr_len;
  double* fr_vals;
  int fair_nzoff;
  double* fair_fv;
} ophase;

typedef struct {
  int left, right; /* 0-based */
  int nlink;
  double *lmin, *lmax;
} olink;

struct orpm {
  int P, L;
  ophase* ph;
  olink* lk;
  const orpm_functions* fun;
  int nconsts;
  double* consts;
  double tol;
  int first_derive;
  int hessian_mode;
  int n, m_nl, m, nnz_nl, nnz_lin, nnz_const, nnz;
  double *xl, *xu, *gl, *gu;
  double* guess;
  int alin_nnz;
  int *alin_i, *alin_j;
  double* alin_v;
  double *linmin, *linmax;
  int fair;   /* 0: the reference's cost shape (default); 1: the fair shape, see orpm_set_cost_shape */
  /* Hessian (oracle/orpm_hess.c) */
  void* hess;
};


typedef struct {
  int N, nx, nu, nq, nc, ne;
  double t0, tf, tspan;
  double* t_radau;      /* N */
  double* state_matrix; /* (N+1) x nx */
  double* state_radau;  /* N x nx */
  double* control;      /* N x nu */
  double *x0, *xf;      /* nx */
  double* parameter;    /* nq */
} pslice;


void* orpm_xcalloc(size_t n, size_t s);
#define NEW(T, n) ((T*)orpm_xcalloc((size_t)(n), sizeof(T)))
double* orpm_dupd(const double* s, int n);
double orpm_arma_accumulate(const double* a, int n);
double orpm_arma_dot(const double* a, const double* b, int n);
void orpm_slice_phase(const orpm* o, int i, const double* x, pslice* s);
void orpm_free_slice(pslice* s);
void orpm_mk_soldae(const pslice* s, int phase_num, orpm_soldae* d);
void orpm_mk_solcost(const pslice* s, int phase_num, orpm_solcost* c);
void orpm_mk_solevent(const pslice* s, int phase_num, orpm_solevent* e);
void orpm_deriv_dae(orpm* o, const orpm_soldae* sd, double* dstate, double* dpath);
void orpm_deriv_lagrange(orpm* o, const orpm_solcost* sc, double* d);
double orpm_spline_interp(double x, const double* xdata, const double* ydata, int n);
#endif
