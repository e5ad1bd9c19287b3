/*
 * orpm_hpliu.c — CPU ORACLE, hp-Liu mesh refinement (test infrastructure; PARITY UNPINNED, see orpm.h).
 * Restates LiuHpMeshRefineAlg (paths relative to /root/reference/Lpopc/src/Core):
 *   RefineMesh                              LpLiuHpMeshRefineAlg.cpp:12-260
 *   GetLagrangeInterpCoefficientsImpl / CalculateDi   :282-304, :263-280
 *   Dividing_mesh / Increasing_N / Reducing_N         :321-377, :379-436, :438-481
 *   Merging_mesh                             :483-604  (its verdict is not used by RefineMesh, :197-220: adjacent satisfied
 *                                            segments with equal N are merged unconditionally; restated that way)
 *   CanWeIncreaseN / calculate2nd_derive     :606-681, :683-709
 * As written, bug for bug, including: the second derivative is sampled by interpolating data given on tau in [-1,1] at
 * abscissae of [t0,tf] (:689-699); CanWeIncreaseN's "previous" data are the CURRENT mesh points paired with rows of
 * the PREVIOUS solution's state matrix picked by mesh-point index (:649-660).
 * Where the reference would throw (empty find(), row index past the previous state matrix) or cast a non-finite double
 * to an unsigned integer (undefined), this restatement returns an error code instead.  One deliberate deviation: a phase
 * whose intervals are all satisfied while an EARLIER phase still refines keeps its mesh (the reference leaves a null mesh
 * in its history and dereferences it on the next call, :159).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orpm_internal.h"

enum { NOT_SATISFIED = 0, SATISFIED = 1, REDUCED = 2, MERGED = 3 };

typedef struct {
  int K;
  double* mesh; /* K + 1 */
  int* nodes;   /* K */
  double* e_k;  /* K */
} lmesh;
typedef struct {
  int rows, nx;
  double* v; /* rows x nx, column-major */
} lstate;

struct orpm_hpliu {
  int P, Nmax, mesh_index;
  double tol, R;
  int n_mesh, n_state; /* history lengths */
  lmesh** mesh_hist;   /* [n_mesh][P] */
  lstate** state_hist; /* [n_state][P] */
  lmesh** mp_hist;     /* [n_state][P], only .K and .mesh used (mesh_points_history_) */
};

static lmesh mk_mesh(int K, const double* mesh, const int* nodes) {
  lmesh m;
  m.K = K;
  m.mesh = orpm_dupd(mesh, K + 1);
  m.nodes = NEW(int, K);
  memcpy(m.nodes, nodes, sizeof(int) * K);
  m.e_k = NEW(double, K);
  return m;
}

orpm_hpliu* orpm_hpliu_create(int P, double tol, int Nmax, double R) {
  orpm_hpliu* h = NEW(orpm_hpliu, 1);
  h->P = P; h->tol = tol; h->Nmax = Nmax; h->R = R;
  return h;
}
void orpm_hpliu_destroy(orpm_hpliu* h) {
  if (!h) return;
  for (int c = 0; c < h->n_mesh; c++) {
    for (int p = 0; p < h->P; p++) { free(h->mesh_hist[c][p].mesh); free(h->mesh_hist[c][p].nodes); free(h->mesh_hist[c][p].e_k); }
    free(h->mesh_hist[c]);
  }
  for (int c = 0; c < h->n_state; c++) {
    for (int p = 0; p < h->P; p++) { free(h->state_hist[c][p].v); free(h->mp_hist[c][p].mesh); }
    free(h->state_hist[c]); free(h->mp_hist[c]);
  }
  free(h->mesh_hist); free(h->state_hist); free(h->mp_hist); free(h);
}

/* BarLagrangeInterp, LpSolutionError.cpp:10-44 */
static void bary(int M, const double* dx, const double* dy, int Nq, const double* xq, double* y) {
  double* H = NEW(double, (size_t)Nq * M);
  double* S = NEW(double, Nq);
  int* fix = NEW(int, Nq);
  orpm_bary_tables(M, dx, Nq, xq, H, S, fix);
  for (int r = 0; r < Nq; r++) {
    double acc = 0.0;
    for (int j = 0; j < M; j++) acc += H[r + (size_t)j * Nq] * dy[j];
    y[r] = fix[r] >= 0 ? dy[fix[r]] : acc / S[r];
  }
  free(H); free(S); free(fix);
}

/* power-series coefficients (descending powers) of the N+1 Lagrange basis polynomials on [LGR(N); 1], :282-304 */
void orpm_hpliu_alj(int N, double* alj /* (N+1) x (N+1), column-major */) {
  int M = N + 1;
  double* x = NEW(double, M);
  double* w = NEW(double, N);
  orpm_lgr_points(N, x, w);
  x[N] = 1.0;
  double* T = NEW(double, (size_t)N * N);
  double* t = NEW(double, N);
  double* Di = NEW(double, M);
  double* pw = NEW(double, M);
  double* prod = NEW(double, M);
  for (int i = 0; i < M; i++) {
    int q = 0;
    for (int j = 0; j < M; j++)
      if (j != i) t[q++] = -x[j];
    /* CalculateDi(-temx), :263-280 */
    memset(T, 0, sizeof(double) * (size_t)N * N);
    for (int j = 0; j < N; j++) T[0 + (size_t)j * N] = t[j];
    for (int r = 1; r < N; r++) {
      for (int j = N - 2; j >= 0; j--) T[r + (size_t)j * N] = T[r + (size_t)(j + 1) * N] + T[(r - 1) + (size_t)(j + 1) * N];
      for (int j = 0; j < N; j++) T[r + (size_t)j * N] = T[r + (size_t)j * N] * t[j];
    }
    Di[0] = 1.0;
    for (int r = 0; r < N; r++) {   /* sum(T, 1): first column, then += the others */
      double s = T[r];
      for (int j = 1; j < N; j++) s += T[r + (size_t)j * N];
      Di[1 + r] = s;
    }
    pw[N] = 1.0;
    for (int k = N - 1; k >= 0; k--) pw[k] = pw[k + 1] * x[i];
    for (int k = 0; k < M; k++) prod[k] = pw[k] * Di[k];
    double den = orpm_arma_accumulate(prod, M);
    for (int k = 0; k < M; k++) alj[k + (size_t)i * M] = Di[k] / den;
  }
  free(x); free(w); free(T); free(t); free(Di); free(pw); free(prod);
}

/* Reducing_N, :438-481.  seg: (N+1) x nx rows of the phase's state matrix (leading dimension ld) */
static int reducing_n(const orpm_hpliu* h, int N, const double* seg, int ld, int nx, const double* betai) {
  int M = N + 1;
  double* alj = NEW(double, (size_t)M * M);
  orpm_hpliu_alj(N, alj);
  int best = 0;
  for (int s = 0; s < nx; s++) {
    int first = -1;
    for (int r = 0; r < M && first < 0; r++) {
      double b = 0.0;   /* bil = alj * segment_state */
      for (int k = 0; k < M; k++) b += alj[r + (size_t)k * M] * seg[k + (size_t)s * ld];
      if (b / betai[s] > h->tol) first = r;   /* signed comparison, as written (:472) */
    }
    int maxN = first < 0 ? 1 : M - 1 - first;
    if (maxN > best) best = maxN;
  }
  free(alj);
  return best > 2 ? best : 2;
}

/* calculate2nd_derive, :683-709: |second difference| maxima per state and the abscissa index of each */
static void second_derivative(int n, const double* t, const double* x, int ld, int nx, double* pmax, double* tmax) {
  double t0 = t[0], tf = t[n - 1];
  double* tau = NEW(double, n);
  for (int i = 0; i < n; i++) tau[i] = 2.0 * (t[i] - t0) / (tf - t0) - 1.0;
  double taustep = 2.0 / 500.0;
  double tp[501], xp[501];
  double delta = (tf - t0) / 500.0;
  for (int i = 0; i < 500; i++) tp[i] = t0 + i * delta;
  tp[500] = tf;
  double* col = NEW(double, n);
  for (int s = 0; s < nx; s++) {
    for (int i = 0; i < n; i++) col[i] = x[i + (size_t)s * ld];
    bary(n, tau, col, 501, tp, xp);
    double best = -1.0;
    int bi = 0;
    for (int i = 0; i < 499; i++) {
      double d = (xp[i + 2] - 2 * xp[i + 1]) + xp[i];
      d /= (taustep * taustep);
      d = fabs(d);
      if (d > best) { best = d; bi = i; }   /* NaN never wins, like arma's max */
    }
    pmax[s] = best < 0 ? -INFINITY : best;
    tmax[s] = tp[bi];
  }
  free(tau); free(col);
}

static int last_le(const double* a, int n, double v, int strict) {   /* max(find(a <= v)) or (a < v) */
  int r = -1;
  for (int i = 0; i < n; i++)
    if (strict ? a[i] < v : a[i] <= v) r = i;
  return r;
}
static int first_ge(const double* a, int n, double v, int strict) {  /* min(find(a >= v)) or (a > v) */
  for (int i = 0; i < n; i++)
    if (strict ? a[i] > v : a[i] >= v) return i;
  return -1;
}

/* CanWeIncreaseN, :606-681.  Returns 1/0, or -1 where the reference would throw. */
static int can_increase(const orpm_hpliu* h, int ip, int istart, int n, const double* tau_all, const double* state, int ld, int nx) {
  double* pm = NEW(double, nx);
  double* tm = NEW(double, nx);
  double* pmb = NEW(double, nx);
  double* tmb = NEW(double, nx);
  int rc = -1;
  second_derivative(n + 1, tau_all + istart, state + istart, ld, nx, pm, tm);
  double mintime = tm[0], maxtime = tm[0];
  for (int s = 1; s < nx; s++) { if (tm[s] < mintime) mintime = tm[s]; if (tm[s] > maxtime) maxtime = tm[s]; }
  const lmesh* cur = &h->mesh_hist[h->n_mesh - 1][ip];
  int nm = cur->K + 1, lo, hi;
  if (mintime == maxtime) {
    if (mintime == tau_all[istart]) { lo = last_le(cur->mesh, nm, mintime, 0); hi = first_ge(cur->mesh, nm, maxtime, 1); }
    else if (maxtime == tau_all[istart + n]) { lo = last_le(cur->mesh, nm, mintime, 1); hi = first_ge(cur->mesh, nm, maxtime, 0); }
    else { lo = last_le(cur->mesh, nm, mintime, 1); hi = first_ge(cur->mesh, nm, maxtime, 1); }
  } else {
    lo = last_le(cur->mesh, nm, mintime, 0); hi = first_ge(cur->mesh, nm, maxtime, 0);
  }
  const lmesh* tb = &h->mp_hist[h->n_state - 1][ip];
  const lstate* sb = &h->state_hist[h->n_state - 1][ip];
  if (lo >= 0 && hi >= 0 && hi >= lo && hi <= tb->K && hi < sb->rows) {
    second_derivative(hi - lo + 1, tb->mesh + lo, sb->v + lo, sb->rows, nx, pmb, tmb);
    double mx = -INFINITY;
    for (int s = 0; s < nx; s++) {
      double r = pm[s] / pmb[s];
      if (r > mx) mx = r;
    }
    rc = mx > h->R ? 0 : 1;
  }
  free(pm); free(tm); free(pmb); free(tmb);
  return rc;
}

/* the part Dividing_mesh and Increasing_N share, :323-365 / :381-426: q.  Returns 0 on a throw of the reference. */
static int growth_exponent(const orpm_hpliu* h, int ip, double m0, double mf, int N, double e_k, double* q) {
  const lmesh* b = &h->mesh_hist[h->n_mesh - 2][ip];
  int lo = last_le(b->mesh, b->K + 1, m0, 0), hi = first_ge(b->mesh, b->K + 1, mf, 0);
  if (lo < 0 || hi < 0 || hi - 1 < lo) return 0;
  double hh = mf - m0, hb = b->mesh[hi] - b->mesh[lo];
  int Nb = 0;
  double ekb = b->e_k[lo];
  for (int i = lo; i < hi; i++) { Nb += b->nodes[i]; if (b->e_k[i] > ekb) ekb = b->e_k[i]; }
  double fN = (double)N / (double)Nb, fh = hh / hb, fe = e_k / ekb;
  *q = ceil(log(fe / pow((double)N, 5.0 / 2.0)) / log(fh / fN));
  return 1;
}
static long to_uword(double v) { return (v != v || v < 0 || v > 1e9) ? -1 : (long)v; }   /* -1: the cast is undefined */

/* RefineMesh for all phases of the engine `o` (built on the current mesh) at the solution x.
 * Outputs per phase p: new_K[p], new_mesh + mesh_off[p], new_nodes + nodes_off[p] (offsets filled here, capacity cap
 * entries each).  Returns NoMoreRefine (1/0), or -1 where the reference would throw / hit undefined behaviour. */
int orpm_hpliu_refine(orpm_hpliu* h, orpm* o, const double* x, int cap, double* new_mesh, int* new_nodes, int* mesh_off,
                      int* nodes_off, int* new_K) {
  int P = o->P, no_more = 1, fail = 0;
  if (P != h->P) return -1;
  if (h->mesh_index == 0) {   /* the user's first mesh, :22-33 */
    h->mesh_hist = (lmesh**)realloc(h->mesh_hist, sizeof(lmesh*) * (h->n_mesh + 1));
    h->mesh_hist[h->n_mesh] = NEW(lmesh, P);
    for (int p = 0; p < P; p++) h->mesh_hist[h->n_mesh][p] = mk_mesh(o->ph[p].K, o->ph[p].mesh, o->ph[p].nk);
    h->n_mesh++;
  }
  lmesh* before = h->mesh_hist[h->n_mesh - 1];
  lmesh* out = NEW(lmesh, P);
  lstate* st = NEW(lstate, P);
  lmesh* mp = NEW(lmesh, P);
  int moff = 0, noff = 0;
  for (int ip = 0; ip < P && !fail; ip++) {
    const ophase* p = &o->ph[ip];
    int K = p->K, N = p->N, nx = p->nx, M1 = N + 1, rows = N + K + 1;
    if (before[ip].K != K) { fail = 1; break; }   /* the engine must be built on the mesh this object produced last */
    const double* state = x + p->state0;
    double* rel = NEW(double, (size_t)rows * nx);
    orpm_solution_error(o, ip, x, rel);
    double* tau = NEW(double, M1);
    for (int k = 0; k < N; k++) tau[k] = p->points[k];
    tau[N] = 1.0;
    st[ip].rows = M1; st[ip].nx = nx; st[ip].v = orpm_dupd(state, M1 * nx);
    double* betai = NEW(double, nx);
    for (int s = 0; s < nx; s++) {
      double mx = state[(size_t)s * M1];
      for (int r = 1; r < M1; r++) if (state[r + (size_t)s * M1] > mx) mx = state[r + (size_t)s * M1];
      betai[s] = 1 + mx;
    }
    /* per segment: [m0, mf], one or several node counts */
    int* tag = NEW(int, K);
    int* seg_cnt = NEW(int, K);      /* intervals the segment becomes */
    int* seg_nodes = NEW(int, K);    /* nodes of each of them */
    int istart_e = 0, istart_x = 0;
    for (int seg = 0; seg < K && !fail; seg++) {
      int n = p->nk[seg], ifin = istart_e + n + 1;
      double emax = rel[istart_e];
      for (int s = 0; s < nx; s++)
        for (int r = istart_e; r <= ifin; r++) if (rel[r + (size_t)s * rows] > emax) emax = rel[r + (size_t)s * rows];
      before[ip].e_k[seg] = emax;
      double m0 = p->mesh[seg], mf = p->mesh[seg + 1];
      seg_cnt[seg] = 1;
      if (emax <= h->tol) {
        int need = reducing_n(h, n, state + istart_x, M1, nx, betai);
        seg_nodes[seg] = need;
        if (need == n) tag[seg] = SATISFIED;
        else { tag[seg] = REDUCED; no_more = 0; }
      } else {
        if (h->mesh_index == 0) {
          seg_nodes[seg] = n + 3;   /* second mesh: three more collocation points, :122-131 */
        } else {
          int inc = can_increase(h, ip, istart_x, n, tau, state, M1, nx);
          if (inc < 0) { fail = 1; break; }
          int divide = !inc;
          double q;
          if (inc) {
            if (!growth_exponent(h, ip, m0, mf, n, emax, &q)) { fail = 1; break; }
            long need = to_uword(ceil(n * pow(emax / h->tol, 1.0 / (q - 5.0 / 2.0))));
            if (need < 0 || need > h->Nmax) divide = 1;   /* an undefined cast yields a huge uword on x86-64: > Nmax */
            else seg_nodes[seg] = (int)need;
          }
          if (divide) {
            if (!growth_exponent(h, ip, m0, mf, n, emax, &q)) { fail = 1; break; }
            long H = to_uword(ceil(pow(emax / h->tol, 1 / q)));
            long Hmax = to_uword(ceil(log(emax / h->tol) / log((double)n)));
            if (H < 0 && Hmax < 0) { fail = 1; break; }
            long S = H < 0 ? Hmax : Hmax < 0 ? H : (H < Hmax ? H : Hmax);
            if (S < 2) S = 2;
            seg_cnt[seg] = (int)S;
            seg_nodes[seg] = n;
          }
        }
        tag[seg] = NOT_SATISFIED;
        no_more = 0;
      }
      istart_e = ifin;
      istart_x += n;
    }
    /* merge pass (:164-226) unless everything so far is satisfied; then the new mesh of the phase */
    int nseg = K;
    double* sm0 = NEW(double, K);
    double* smf = NEW(double, K);
    for (int seg = 0; seg < K; seg++) { sm0[seg] = p->mesh[seg]; smf[seg] = p->mesh[seg + 1]; }
    if (!fail && !no_more) {
      int idx = 0;
      for (int seg = 0; seg < K; seg++) {
        if (seg > 0 && tag[idx] != NOT_SATISFIED && tag[idx - 1] != NOT_SATISFIED && seg_nodes[idx] == seg_nodes[idx - 1]) {
          smf[idx - 1] = smf[idx];
          for (int q = idx; q + 1 < nseg; q++) { sm0[q] = sm0[q + 1]; smf[q] = smf[q + 1]; tag[q] = tag[q + 1]; seg_cnt[q] = seg_cnt[q + 1]; seg_nodes[q] = seg_nodes[q + 1]; }
          nseg--;
          tag[idx - 1] = MERGED;
        } else {
          idx++;
        }
      }
    }
    if (!fail) {
      int nk = 0;
      for (int q = 0; q < nseg; q++) nk += seg_cnt[q];
      if (moff + nk + 1 > cap || noff + nk > cap) { fail = 1; }
      else {
        double* om = new_mesh + moff;
        int* on = new_nodes + noff;
        int c = 0;
        om[0] = -1;
        for (int q = 0; q < nseg; q++) {
          double delta = (smf[q] - sm0[q]) / (double)seg_cnt[q];   /* linspace(m0, mf, cnt + 1) */
          for (int i = 1; i <= seg_cnt[q]; i++) {
            om[c + 1] = (i == seg_cnt[q]) ? smf[q] : sm0[q] + i * delta;
            on[c] = seg_nodes[q];
            c++;
          }
        }
        mesh_off[ip] = moff; nodes_off[ip] = noff; new_K[ip] = nk;
        out[ip] = mk_mesh(nk, om, on);
        mp[ip].K = nk; mp[ip].mesh = orpm_dupd(om, nk + 1); mp[ip].nodes = NULL; mp[ip].e_k = NULL;
        moff += nk + 1; noff += nk;
      }
    }
    free(rel); free(tau); free(betai); free(tag); free(seg_cnt); free(seg_nodes); free(sm0); free(smf);
  }
  if (fail) {
    for (int p = 0; p < P; p++) { free(st[p].v); }
    free(out); free(st); free(mp);
    return -1;
  }
  h->mesh_hist = (lmesh**)realloc(h->mesh_hist, sizeof(lmesh*) * (h->n_mesh + 1));
  h->mesh_hist[h->n_mesh++] = out;
  h->state_hist = (lstate**)realloc(h->state_hist, sizeof(lstate*) * (h->n_state + 1));
  h->mp_hist = (lmesh**)realloc(h->mp_hist, sizeof(lmesh*) * (h->n_state + 1));
  h->state_hist[h->n_state] = st;
  h->mp_hist[h->n_state] = mp;
  h->n_state++;
  h->mesh_index++;
  return no_more;
}
