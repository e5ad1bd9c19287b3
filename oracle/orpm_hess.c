/*
 * orpm_hess.c — CPU ORACLE, exact-Hessian mode (test infrastructure; PARITY UNPINNED, see orpm.h).
 *
 * Restates lpopc's "exact" Lagrangian Hessian (hessian-approximation=exact): forward SECOND differences of
 * the user functions, lambda-weighted, lower-triangular COO with intentional duplicates.
 * Reference (paths relative to /root/reference/Lpopc/src):
 *   dependency probe   DeriveDependicieshecker::GetDependiciesForJacobiInEveryPhase  Core/LpDerivDependciesChecker.cpp:10-110
 *   second differences LpHessianCalculator::CalculatePhaseHessian                   Core/LpHessian.cpp:1192-2161
 *                      CalculateLinkHessain                                         :2163-2367
 *   assembly           GetPhaseHessian :12-599, GetLinkHessian :1020-1190, GetHessian :878-1018
 *   structure          GetPhaseHessianSparsity :601-876, GetLinkHessianSparsity :2369-2508, GetHessianSparsity :2510-2600
 * Only nq = 0 (the parity domain).  Quirks kept bug-for-bug: the mixed endpoint denominators use pertxf(istate)
 * where jstate is meant (:1588,1612,1826,1850); link x0R-x0R columns use the LEFT phase's node count (:1150).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orpm_internal.h"

typedef struct {
  int* dep; /* (nx+nc) x (nx+nu), column-major, 0/1 */
  int* H;   /* (nx+nu) x (nx+nu): dep' * dep with unit diagonal */
  int nI, nE;
} hphase;

typedef struct {
  hphase* ph;
  int P;
  int nnz;
} ohess;

int orpm_hess_nnz(void* h) { return h ? ((ohess*)h)->nnz : 0; }

void orpm_hess_destroy(void* hv) {
  ohess* h = (ohess*)hv;
  if (!h) return;
  for (int i = 0; i < h->P; i++) {
    free(h->ph[i].dep);
    free(h->ph[i].H);
  }
  free(h->ph);
  free(h);
}

/* NaN-propagation probe at node 1 of the guess, LpDerivDependciesChecker.cpp:60-93 */
static void probe_dependencies(orpm* o, int ip, int* dep) {
  const ophase* p = &o->ph[ip];
  int nx = p->nx, nu = p->nu, nc = p->nc, nout = nx + nc;
  pslice s;
  orpm_slice_phase(o, ip, o->guess, &s);
  double t = s.t_radau[1];
  double* xs = NEW(double, nx);
  double* us = NEW(double, nu > 0 ? nu : 1);
  for (int j = 0; j < nx; j++) xs[j] = s.state_radau[1 + (size_t)j * s.N];
  for (int j = 0; j < nu; j++) us[j] = s.control[1 + (size_t)j * s.N];
  orpm_soldae sd;
  sd.phase_num = ip + 1;
  sd.N = 1;
  sd.nx = nx;
  sd.nu = nu;
  sd.nq = 0;
  sd.nc = nc;
  sd.time = &t;
  sd.state = xs;
  sd.control = us;
  sd.parameter = NULL;
  double* f = NEW(double, nx);
  double* c = NEW(double, nc > 0 ? nc : 1);
  for (int v = 0; v < nx + nu; v++) {
    double keep = v < nx ? xs[v] : us[v - nx];
    if (v < nx) xs[v] = NAN; else us[v - nx] = NAN;
    o->fun->dae(&sd, o->consts, f, c);
    for (int r = 0; r < nout; r++) {
      double val = r < nx ? f[r] : c[r - nx];
      dep[r + (size_t)v * nout] = isfinite(val) ? 0 : 1;
    }
    if (v < nx) xs[v] = keep; else us[v - nx] = keep;
  }
  free(xs);
  free(us);
  free(f);
  free(c);
  orpm_free_slice(&s);
}

void* orpm_hess_create(orpm* o) {
  ohess* h = NEW(ohess, 1);
  h->ph = NEW(hphase, o->P);
  h->P = o->P;
  h->nnz = 0;
  for (int ip = 0; ip < o->P; ip++) {
    const ophase* p = &o->ph[ip];
    int nx = p->nx, nu = p->nu, nc = p->nc, nv = nx + nu, nout = nx + nc, N = p->N;
    hphase* q = &h->ph[ip];
    q->dep = NEW(int, (size_t)nout * nv);
    q->H = NEW(int, (size_t)nv * nv);
    probe_dependencies(o, ip, q->dep);
    int nnzH = 0;
    for (int a = 0; a < nv; a++)
      for (int b = 0; b < nv; b++) { /* temDependencies = trans(dep)*dep; diag = 1, LpHessian.cpp:899-903 */
        int acc = 0;
        for (int r = 0; r < nout; r++) acc += q->dep[r + (size_t)a * nout] * q->dep[r + (size_t)b * nout];
        if (a == b) acc = 1;
        q->H[a + (size_t)b * nv] = acc;
        if (acc) nnzH++;
      }
    q->nI = N * ((nnzH - nx - nu) / 2 + nx + nu) + 2 * (nx + nu) * N + 3; /* :907-908 with nq = 0 */
    q->nE = (2 * nx) * (2 * nx - 1) / 2 + 2 * nx + 4 * nx + 3;            /* :909-910 */
    h->nnz += q->nI + q->nE;
  }
  for (int i = 0; i < o->L; i++) {
    int nl = o->ph[o->lk[i].left].nx, nr = o->ph[o->lk[i].left].nx; /* left phase twice, :918-921 */
    h->nnz += (nl + nr) * (nl + nr - 1) / 2 + (nl + nr);
  }
  return h;
}

/* ---- structure ------------------------------------------------------------------------------- */
void orpm_hess_structure(orpm* o, int* iRow, int* jCol) {
  ohess* h = (ohess*)o->hess;
  if (!h) return;
  int s = 0;
  for (int ip = 0; ip < o->P; ip++) {
    const ophase* p = &o->ph[ip];
    const hphase* q = &h->ph[ip];
    int nx = p->nx, nu = p->nu, nv = nx + nu, N = p->N, sh = p->var0;
    int sI = s, sE = s + q->nI;
#define HH(a, b) q->H[(a) + (size_t)(b)*nv]
    for (int i = 0; i < nx; i++) { /* :659-690 */
      int rowstart = i * (N + 1);
      for (int j = 0; j <= i; j++) {
        int colstart = j * (N + 1);
        if (HH(i, j))
          for (int k = 0; k < N; k++) { iRow[sI] = sh + rowstart + k; jCol[sI++] = sh + colstart + k; }
        iRow[sE] = sh + rowstart; jCol[sE++] = sh + colstart;
        if (i != j) { iRow[sE] = sh + rowstart; jCol[sE++] = sh + colstart + N; }
        iRow[sE] = sh + rowstart + N; jCol[sE++] = sh + colstart;
        iRow[sE] = sh + rowstart + N; jCol[sE++] = sh + colstart + N;
      }
    }
    int rowshift = nx * (N + 1);
    for (int i = 0; i < nu; i++) { /* :693-722 */
      int rowstart = rowshift + i * N;
      for (int j = 0; j < nx; j++)
        if (HH(i + nx, j))
          for (int k = 0; k < N; k++) { iRow[sI] = sh + rowstart + k; jCol[sI++] = sh + j * (N + 1) + k; }
      for (int j = 0; j <= i; j++)
        if (HH(i + nx, j + nx))
          for (int k = 0; k < N; k++) { iRow[sI] = sh + rowstart + k; jCol[sI++] = sh + rowshift + j * N + k; }
    }
    int trow = nx * (N + 1) + nu * N; /* t0 row, then tf row; :725-800 */
    for (int r = 0; r < 2; r++) {
      int row = trow + r;
      for (int i = 0; i < nx; i++) {
        for (int k = 0; k < N; k++) { iRow[sI] = sh + row; jCol[sI++] = sh + i * (N + 1) + k; }
        iRow[sE] = sh + row; jCol[sE++] = sh + i * (N + 1);
        iRow[sE] = sh + row; jCol[sE++] = sh + i * (N + 1) + N;
      }
      for (int i = 0; i < nu; i++)
        for (int k = 0; k < N; k++) { iRow[sI] = sh + row; jCol[sI++] = sh + nx * (N + 1) + i * N + k; }
      iRow[sI] = sh + row; jCol[sI++] = sh + trow;
      iRow[sE] = sh + row; jCol[sE++] = sh + trow;
      if (r == 1) {
        iRow[sI] = sh + row; jCol[sI++] = sh + trow + 1;
        iRow[sE] = sh + row; jCol[sE++] = sh + trow + 1;
      }
    }
#undef HH
    s += q->nI + q->nE;
  }
  for (int ip = 0; ip < o->L; ip++) { /* GetLinkHessianSparsity, :2369-2508 */
    const ophase* pl = &o->ph[o->lk[ip].left];
    const ophase* pr = &o->ph[o->lk[ip].right];
    for (int i = 0; i < pl->nx; i++)
      for (int j = 0; j <= i; j++) {
        iRow[s] = pl->state0 + (pl->N + 1) * (i + 1) - 1;
        jCol[s++] = pl->state0 + (pl->N + 1) * (j + 1) - 1;
      }
    for (int i = 0; i < pr->nx; i++) {
      int row = pr->state0 + (pr->N + 1) * i;
      for (int j = 0; j < pl->nx; j++) { iRow[s] = row; jCol[s++] = pl->state0 + (pl->N + 1) * (j + 1) - 1; }
      for (int j = 0; j <= i; j++) { iRow[s] = row; jCol[s++] = pr->state0 + (pl->N + 1) * j; } /* nnodesLeft, :2463 */
    }
  }
}

/* ---- second differences of the node functions ---------------------------------------------------- */
typedef struct {
  int N, nx, nu, nc, nv;
  double* F;  /* [N x (nx+nc+1)] : dae, path, lagrange at the evaluation point */
} fpoint;

/* evaluate dae/path/lagrange with variables a and b perturbed (a,b in [0,nv) or -1); a==b adds h twice,
 * exactly `isol.col(a)=xPert.col(a); ijsol.col(b) += pert.col(b)` (LpHessian.cpp:1268-1282) */
static void eval_point(orpm* o, const pslice* s, int phase_num, int a, int b, double* out) {
  int N = s->N, nx = s->nx, nu = s->nu, nc = s->nc;
  double tol = o->tol;
  double* st = orpm_dupd(s->state_radau, N * nx);
  double* ct = orpm_dupd(s->control, N * nu);
  double* tm = orpm_dupd(s->t_radau, N);
  int pv[2] = {a, b};
  for (int w = 0; w < 2; w++) {
    int v = pv[w];
    if (v < 0) continue;
    for (int k = 0; k < N; k++) {
      if (v < nx) st[k + (size_t)v * N] += tol * (1 + fabs(s->state_radau[k + (size_t)v * N]));
      else if (v < nx + nu) ct[k + (size_t)(v - nx) * N] += tol * (1 + fabs(s->control[k + (size_t)(v - nx) * N]));
      else tm[k] += tol * (1 + fabs(s->t_radau[k]));
    }
  }
  orpm_soldae sd;
  sd.phase_num = phase_num; sd.N = N; sd.nx = nx; sd.nu = nu; sd.nq = 0; sd.nc = nc;
  sd.time = tm; sd.state = st; sd.control = ct; sd.parameter = NULL;
  o->fun->dae(&sd, o->consts, out, out + (size_t)N * nx);
  orpm_solcost sc;
  sc.phase_num = phase_num; sc.initial_time = s->t0; sc.initial_state = s->x0; sc.terminal_time = s->tf;
  sc.terminal_state = s->xf; sc.N = N; sc.nx = nx; sc.nu = nu; sc.nq = 0;
  sc.time = tm; sc.state = st; sc.control = ct; sc.parameter = NULL;
  o->fun->lagrange(&sc, o->consts, out + (size_t)N * (nx + nc));
  free(st);
  free(ct);
  free(tm);
}

static double pert_of(const orpm* o, const pslice* s, int v, int k) {
  int N = s->N, nx = s->nx, nu = s->nu;
  if (v < nx) return o->tol * (1 + fabs(s->state_radau[k + (size_t)v * N]));
  if (v < nx + nu) return o->tol * (1 + fabs(s->control[k + (size_t)(v - nx) * N]));
  return o->tol * (1 + fabs(s->t_radau[k]));
}

/* endpoint functions (events, Mayer) with W = [x0.., xf.., t0, tf] variables a, b perturbed */
static void eval_endpoint(orpm* o, const pslice* s, int phase_num, int a, int b, double* ev, double* mayer) {
  int nx = s->nx;
  double tol = o->tol;
  double* x0 = orpm_dupd(s->x0, nx);
  double* xf = orpm_dupd(s->xf, nx);
  double t0 = s->t0, tf = s->tf;
  int pv[2] = {a, b};
  for (int w = 0; w < 2; w++) {
    int v = pv[w];
    if (v < 0) continue;
    if (v < nx) x0[v] += tol * (1 + fabs(s->x0[v]));
    else if (v < 2 * nx) xf[v - nx] += tol * (1 + fabs(s->xf[v - nx]));
    else if (v == 2 * nx) t0 += tol * (1 + fabs(s->t0));
    else tf += tol * (1 + fabs(s->tf));
  }
  if (s->ne > 0) {
    orpm_solevent se;
    se.phase_num = phase_num; se.initial_time = t0; se.terminal_time = tf; se.nx = nx; se.nq = 0; se.ne = s->ne;
    se.initial_state = x0; se.terminal_state = xf; se.parameter = NULL;
    o->fun->event(&se, o->consts, ev);
  }
  orpm_solcost sc;
  memset(&sc, 0, sizeof(sc));
  sc.phase_num = phase_num; sc.initial_time = t0; sc.initial_state = x0; sc.terminal_time = tf; sc.terminal_state = xf;
  sc.N = s->N; sc.nx = nx; sc.nu = s->nu; sc.time = s->t_radau; sc.state = s->state_radau; sc.control = s->control;
  o->fun->mayer(&sc, o->consts, mayer);
  free(x0);
  free(xf);
}
static double wpert(const orpm* o, const pslice* s, int v) {
  int nx = s->nx;
  if (v < nx) return o->tol * (1 + fabs(s->x0[v]));
  if (v < 2 * nx) return o->tol * (1 + fabs(s->xf[v - nx]));
  if (v == 2 * nx) return o->tol * (1 + fabs(s->t0));
  return o->tol * (1 + fabs(s->tf));
}

/* accu(A % B): Armadillo accu_proxy_linear, two interleaved accumulators */
static double accu_prod(const double* a, const double* b, int n) {
  double v1 = 0.0, v2 = 0.0;
  int i, j;
  for (i = 0, j = 1; j < n; i += 2, j += 2) {
    v1 += a[i] * b[i];
    v2 += a[j] * b[j];
  }
  if (i < n) v1 += a[i] * b[i];
  return v1 + v2;
}

void orpm_eval_h(orpm* o, const double* x, double sigma, const double* lambda, double* values) {
  ohess* h = (ohess*)o->hess;
  if (!h) return;
  int s = 0;
  for (int ip = 0; ip < o->P; ip++) {
    const ophase* p = &o->ph[ip];
    const hphase* q = &h->ph[ip];
    int nx = p->nx, nu = p->nu, nc = p->nc, ne = p->ne, nv = nx + nu, NV = nv + 1, N = p->N, nf = nx + nc + 1;
    pslice sl;
    orpm_slice_phase(o, ip, x, &sl);
    double t0 = sl.t0, tf = sl.tf;
    const double* lam_d = lambda + p->con0;              /* diff_lambda(:,s) = lam_d[s*N + k], :85-98 */
    const double* lam_p = lam_d + (size_t)nx * N;
    const double* lam_e = lam_p + (size_t)nc * N;
    size_t fsz = (size_t)N * nf;
    double* F0 = NEW(double, fsz);
    eval_point(o, &sl, ip + 1, -1, -1, F0);
    double** Fa = NEW(double*, NV);
    for (int a = 0; a < NV; a++) {
      Fa[a] = NEW(double, fsz);
      eval_point(o, &sl, ip + 1, a, -1, Fa[a]);
    }
    double* Fab = NEW(double, fsz);
    double* XI = NEW(double, N); /* combined term of one pair */
    /* combination of the second differences of pair (a >= b): ((tf-t0)/2)(sigma w L_ab - sum lam f_ab) + sum mu c_ab */
#define COMBINE(a, b)                                                                                        \
  do {                                                                                                       \
    eval_point(o, &sl, ip + 1, (a), (b), Fab);                                                               \
    for (int k = 0; k < N; k++) {                                                                            \
      double den = pert_of(o, &sl, (a), k) * pert_of(o, &sl, (b), k);                                        \
      double sd_ = 0.0, sp_ = 0.0;                                                                           \
      for (int o_ = 0; o_ < nx; o_++) {                                                                      \
        size_t ix = k + (size_t)o_ * N;                                                                      \
        double hh = (Fab[ix] - Fa[a][ix] - Fa[b][ix] + F0[ix]) / den;                                        \
        double term = lam_d[(size_t)o_ * N + k] * hh;                                                        \
        sd_ = (o_ == 0) ? term : sd_ + term;                                                                 \
      }                                                                                                      \
      for (int o_ = 0; o_ < nc; o_++) {                                                                      \
        size_t ix = k + (size_t)(nx + o_) * N;                                                               \
        double hh = (Fab[ix] - Fa[a][ix] - Fa[b][ix] + F0[ix]) / den;                                        \
        double term = lam_p[(size_t)o_ * N + k] * hh;                                                        \
        sp_ = (o_ == 0) ? term : sp_ + term;                                                                 \
      }                                                                                                      \
      size_t il = k + (size_t)(nx + nc) * N;                                                                 \
      double hL = (Fab[il] - Fa[a][il] - Fa[b][il] + F0[il]) / den;                                          \
      double sL = (sigma * p->weights[k]) * hL;                                                              \
      XI[k] = (tf - t0) / 2.0 * (sL - sd_) + sp_;                                                            \
    }                                                                                                        \
  } while (0)
    double* VI = values + s;
    double* VE = values + s + q->nI;
    int sI = 0, sE = 0;
    /* ---- endpoint second differences (events, Mayer), :1553-1983 ---- */
    int NW = 2 * nx + 2;
    double* E0 = NEW(double, ne > 0 ? ne : 1);
    double* Ea = NEW(double, (size_t)NW * (ne > 0 ? ne : 1));
    double* Eab = NEW(double, ne > 0 ? ne : 1);
    double M0, Mab;
    double* Ma = NEW(double, NW);
    double* hEv = NEW(double, ne > 0 ? ne : 1);
    eval_endpoint(o, &sl, ip + 1, -1, -1, E0, &M0);
    for (int a = 0; a < NW; a++) eval_endpoint(o, &sl, ip + 1, a, -1, Ea + (size_t)a * (ne > 0 ? ne : 1), &Ma[a]);
    /* hE(a,b): first variable a (row), second b; `den` as written in the reference (with its quirks) */
#define ENDPT(a, b, den, dst)                                                                         \
  do {                                                                                                \
    eval_endpoint(o, &sl, ip + 1, (a), (b), Eab, &Mab);                                               \
    double hM = (Mab - Ma[a] - Ma[b] + M0) / (den);                                                   \
    double ls = 0.0;                                                                                  \
    if (ne > 0) {                                                                                     \
      for (int e_ = 0; e_ < ne; e_++)                                                                 \
        hEv[e_] = (Eab[e_] - Ea[(size_t)(a)*ne + e_] - Ea[(size_t)(b)*ne + e_] + E0[e_]) / ((den)*1.0); \
      ls = accu_prod(hEv, lam_e, ne);                                                                 \
    }                                                                                                 \
    (dst) = sigma * hM + ls;                                                                          \
  } while (0)
#define HH(a, b) q->H[(a) + (size_t)(b)*nv]
    /* ---- xx blocks + x0/xf endpoint entries, :409-432 ---- */
    for (int i = 0; i < nx; i++)
      for (int j = 0; j <= i; j++) {
        if (HH(i, j)) {
          COMBINE(i, j);
          memcpy(VI + sI, XI, sizeof(double) * N);
          sI += N;
        }
        double px0i = wpert(o, &sl, i), px0j = wpert(o, &sl, j), pxfi = wpert(o, &sl, nx + i);
        /* ORPM_HESS_CORRECT_DEN=1 (debugging aid of the tests only) uses the mathematically right pertxf(j) */
        double pxfq = getenv("ORPM_HESS_CORRECT_DEN") ? wpert(o, &sl, nx + j) : pxfi;
        ENDPT(i, j, px0i * px0j, VE[sE]); sE++;                         /* x0_i, x0_j */
        if (i != j) { ENDPT(i, nx + j, px0i * pxfq, VE[sE]); sE++; }    /* x0_i, xf_j : pertx0(i)*pertxf(i) (:1588) */
        ENDPT(nx + i, j, pxfi * px0j, VE[sE]); sE++;                    /* xf_i, x0_j */
        ENDPT(nx + i, nx + j, pxfi * pxfq, VE[sE]); sE++;               /* xf_i, xf_j : pertxf(i)*pertxf(i) (:1612) */
      }
    /* ---- ux, uu blocks, :434-462 ---- */
    for (int i = 0; i < nu; i++) {
      for (int j = 0; j < nx; j++)
        if (HH(i + nx, j)) {
          COMBINE(nx + i, j);
          memcpy(VI + sI, XI, sizeof(double) * N);
          sI += N;
        }
      for (int j = 0; j <= i; j++)
        if (HH(i + nx, j + nx)) {
          COMBINE(nx + i, nx + j);
          memcpy(VI + sI, XI, sizeof(double) * N);
          sI += N;
        }
    }
    /* ---- first-derivative pieces of the t0/tf rows, :159-176 ---- */
    int ncolD = nx + nu + 1;
    double* dstate = NEW(double, (size_t)N * nx * ncolD);
    double* dpath = NEW(double, (size_t)N * (nc > 0 ? nc : 1) * ncolD);
    double* dLag = NEW(double, (size_t)N * ncolD);
    orpm_soldae sd;
    orpm_mk_soldae(&sl, ip + 1, &sd);
    orpm_deriv_dae(o, &sd, dstate, dpath);
    orpm_solcost sc;
    orpm_mk_solcost(&sl, ip + 1, &sc);
    orpm_deriv_lagrange(o, &sc, dLag);
    double* D1 = NEW(double, N); /* sum_lambda_plus_ddae_v - sigma_plus_dLagrange_v */
#define FIRST(v)                                                                          \
  for (int k = 0; k < N; k++) {                                                           \
    double sdd = 0.0;                                                                     \
    for (int o_ = 0; o_ < nx; o_++) {                                                     \
      double term = lam_d[(size_t)o_ * N + k] * dstate[(k + (size_t)o_ * N) + (size_t)(v) * ((size_t)N * nx)]; \
      sdd = (o_ == 0) ? term : sdd + term;                                                \
    }                                                                                     \
    D1[k] = sdd - (sigma * p->weights[k]) * dLag[k + (size_t)(v)*N];                      \
  }
    double* rows[2];
    rows[0] = NEW(double, (size_t)N * nv); /* hLI_t0x.., hLI_t0u.. */
    rows[1] = NEW(double, (size_t)N * nv);
    for (int v = 0; v < nv; v++) { /* :177-205 */
      COMBINE(nv, v);
      FIRST(v);
      for (int k = 0; k < N; k++) {
        double talpha = (1 - p->points[k]) / 2.0, tbeta = (1 + p->points[k]) / 2.0;
        rows[0][k + (size_t)v * N] = 0.5 * D1[k] + talpha * XI[k];
        rows[1][k + (size_t)v * N] = -0.5 * D1[k] + tbeta * XI[k];
      }
    }
    /* tt scalars, :206-218 */
    COMBINE(nv, nv);
    FIRST(nv);
    double* ta = NEW(double, N);
    double* tb = NEW(double, N);
    double* w1 = NEW(double, N);
    double* w2 = NEW(double, N);
    double* w3 = NEW(double, N);
    double* w4 = NEW(double, N);
    for (int k = 0; k < N; k++) {
      ta[k] = (1 - p->points[k]) / 2.0;
      tb[k] = (1 + p->points[k]) / 2.0;
      w1[k] = D1[k] + ta[k] * XI[k];
      w2[k] = -D1[k] + tb[k] * XI[k];
      w3[k] = tb[k] - ta[k];
      w4[k] = tb[k] * XI[k];
    }
    double h_t0t0 = orpm_arma_dot(ta, w1, N);
    double h_tftf = orpm_arma_dot(tb, w2, N);
    double h_tft0 = 0.5 * orpm_arma_dot(w3, D1, N) + orpm_arma_dot(ta, w4, N);
    /* ---- t0 row then tf row, :465-540 ---- */
    int T0 = 2 * nx, TF = 2 * nx + 1;
    for (int r = 0; r < 2; r++) {
      int tv = r == 0 ? T0 : TF;
      double pt = wpert(o, &sl, tv);
      for (int i = 0; i < nx; i++) {
        memcpy(VI + sI, rows[r] + (size_t)i * N, sizeof(double) * N);
        sI += N;
        ENDPT(tv, i, pt * wpert(o, &sl, i), VE[sE]); sE++;           /* t, x0_i */
        ENDPT(tv, nx + i, pt * wpert(o, &sl, nx + i), VE[sE]); sE++; /* t, xf_i */
      }
      for (int i = 0; i < nu; i++) {
        memcpy(VI + sI, rows[r] + (size_t)(nx + i) * N, sizeof(double) * N);
        sI += N;
      }
      if (r == 0) {
        VI[sI++] = h_t0t0;
        ENDPT(T0, T0, pt * pt, VE[sE]); sE++;
      } else {
        VI[sI++] = h_tft0;
        ENDPT(TF, T0, pt * wpert(o, &sl, T0), VE[sE]); sE++;
        VI[sI++] = h_tftf;
        ENDPT(TF, TF, pt * pt, VE[sE]); sE++;
      }
    }
#undef HH
#undef COMBINE
#undef FIRST
#undef ENDPT
    free(F0);
    for (int a = 0; a < NV; a++) free(Fa[a]);
    free(Fa);
    free(Fab);
    free(XI);
    free(E0);
    free(Ea);
    free(Eab);
    free(Ma);
    free(hEv);
    free(dstate);
    free(dpath);
    free(dLag);
    free(D1);
    free(rows[0]);
    free(rows[1]);
    free(ta);
    free(tb);
    free(w1);
    free(w2);
    free(w3);
    free(w4);
    orpm_free_slice(&sl);
    s += q->nI + q->nE;
  }
  /* ---- linkages, GetLinkHessian :1020-1190 + CalculateLinkHessain :2163-2367 ---- */
  for (int ip = 0; ip < o->L; ip++) {
    const olink* l = &o->lk[ip];
    const ophase* pl = &o->ph[l->left];
    const ophase* pr = &o->ph[l->right];
    int nxl = pl->nx, nxr = pr->nx, nl = l->nlink, nw = nxl + nxr;
    /* link_lambda: Data_->link_indices[ipair] = constraint_offset + j + 1 with constraint_offset NOT advanced
     * per pair (Core/LpBoundsChecker.cpp:240-244), so every pair reads the multipliers of the FIRST pair's rows */
    int g0 = 0;
    for (int i = 0; i < o->P; i++) g0 += o->ph[i].ncon;
    const double* lam_l = lambda + g0;
    double* base = NEW(double, nw);
    for (int j = 0; j < nxl; j++) base[j] = x[pl->state0 + j * (pl->N + 1) + pl->N];
    for (int j = 0; j < nxr; j++) base[nxl + j] = x[pr->state0 + j * (pr->N + 1)];
    double* L0 = NEW(double, nl);
    double* La = NEW(double, (size_t)nw * nl);
    double* Lab = NEW(double, nl);
    double* hL = NEW(double, nl);
    double* w = NEW(double, nw);
    orpm_sollink sk;
    sk.left_phase_num = l->left + 1; sk.right_phase_num = l->right + 1; sk.ipair = ip + 1;
    sk.nxl = nxl; sk.nxr = nxr; sk.nql = sk.nqr = 0; sk.nlink = nl;
    sk.left_parameter = sk.right_parameter = NULL;
#define LINK_AT(a, b, out)                                                    \
  do {                                                                        \
    memcpy(w, base, sizeof(double) * nw);                                     \
    if ((a) >= 0) w[a] += o->tol * (1 + fabs(base[a]));                      \
    if ((b) >= 0) w[b] += o->tol * (1 + fabs(base[b]));                      \
    sk.left_state = w;                                                        \
    sk.right_state = w + nxl;                                                 \
    o->fun->link(&sk, o->consts, (out));                                      \
  } while (0)
    LINK_AT(-1, -1, L0);
    for (int a = 0; a < nw; a++) LINK_AT(a, -1, La + (size_t)a * nl);
#define LINK_H(a, b, dst)                                                                              \
  do {                                                                                                 \
    LINK_AT((a), (b), Lab);                                                                            \
    double den = (o->tol * (1 + fabs(base[a]))) * (o->tol * (1 + fabs(base[b])));                      \
    for (int q_ = 0; q_ < nl; q_++) hL[q_] = (Lab[q_] - La[(size_t)(a)*nl + q_] - La[(size_t)(b)*nl + q_] + L0[q_]) / den; \
    (dst) = accu_prod(hL, lam_l, nl);                                                                  \
  } while (0)
    for (int i = 0; i < nxl; i++)
      for (int j = 0; j <= i; j++) { LINK_H(i, j, values[s]); s++; }       /* xfL_i, xfL_j */
    for (int i = 0; i < nxr; i++) {
      for (int j = 0; j < nxl; j++) { LINK_H(j, nxl + i, values[s]); s++; } /* hLink_xfL_x0R(j,i): first xfL_j, second x0R_i */
      for (int j = 0; j <= i; j++) { LINK_H(nxl + i, nxl + j, values[s]); s++; }
    }
#undef LINK_AT
#undef LINK_H
    free(base);
    free(L0);
    free(La);
    free(Lab);
    free(hL);
    free(w);
  }
}
