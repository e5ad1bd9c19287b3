"""CPU restatement of the interior-point iteration behind rpm_ipm_* (row f-2) — TEST INFRASTRUCTURE, like everything
under oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

What it restates: the algorithm of Ipopt 3.12.3, the solver the reference calls (NLPSolver::SolveNlp,
Core/LpNLPSolver.cpp:13-53).  Ipopt is a third-party dependency that is absent from /root/reference, so this follows
its published description — A. Waechter, L. T. Biegler, "On the implementation of an interior-point filter line-search
algorithm for large-scale nonlinear programming", Math. Program. 106 (2006) — equation numbers below are the paper's:
  optimality error (5)/(6) with Ipopt's unscaled thresholds beside it (dual_inf_tol 1, constr_viol_tol 1e-4, compl_inf_tol 1e-4),
  barrier update (7), tau (8), primal-dual system (13) with dz from (12), fraction to the
  boundary (15), multiplier reset (16), filter acceptance (18)-(20), filter update (22), alpha_min (23), inertia
  correction Algorithm IC, initial point section 3.6 (bound_push / bound_frac), constants = Ipopt 3.12 defaults.
Restated in round 2 (what the metric problem, Delta-III, turned out to need — tests/experiments/ has the experiments):
  * bound_relax_factor (Ipopt option, default 1e-8): every finite bound of a free variable / inequality row is moved out
    by 1e-8 max(1,|bound|) before anything else.  Delta-III has no strict interior without it (burn rates and phase
    durations are fixed, so each phase's final mass EQUALS its lower bound by the dynamics alone);
  * the second-order correction (paper section 2.4, steps A-5.5 .. A-5.9; max_soc 4, kappa_soc 0.99);
  * the restoration phase (paper section 3.3): min rho |p + n|_1 + zeta/2 |D_R (v - v_R)|^2 s.t. c(v) - p + n = 0,
    p, n >= 0 and the bounds, rho = 1000, zeta = sqrt(mu), solved by the same interior-point iteration (monotone mu, own
    filter) with p, n eliminated from the Newton system — [[zeta D_R^2 + Sigma_v, A^T], [A, -(Sigma_p^-1 + Sigma_n^-1)]],
    the same matrix structure as (13) — starting values (33)/(34), return when the infeasibility is kappa_resto = 0.9
    of where it entered and the original filter accepts the point; then least-squares multipliers (section 3.6,
    lambda = 0 when they exceed 1e3); the same least-squares multipliers (Ipopt's recalc_y) up to 3 times when the line
    search gives up at a FEASIBLE point.  One simplification: the restoration problem's Hessian leaves out the constraint
    curvature sum lambda_j Hess c_j (a Gauss-Newton model of it: no inertia correction is ever needed there).
  * mu_strategy = "adaptive" (what the reference asks Ipopt for, Core/LpNLPSolver.cpp:28), the DEFAULT; "monotone" = rule (7):
    Ipopt's adaptive update (Nocedal, Waechter, Waltz, SIAM J. Optim. 19, 2009) with the LOQO oracle (Ipopt's mu_oracle=loqo:
    mu = sigma * average complementarity, sigma = 0.1 min(0.05 (1 - xi) / xi, 2)^3, xi = smallest / average complementarity)
    and the kkt-error globalisation (free mode while the KKT error — 2-norm-squared of dual / primal infeasibility and
    complementarity, each divided by its length — is below 0.9999 of one of the last 4 accepted values, otherwise the
    monotone rule from mu = 0.8 * average complementarity until it is again); the filter is emptied whenever mu changes.
    Ipopt's default oracle there is the quality function (two more solves and a line search over sigma per iteration): not restated.
    The default because it is what the reference configures and the more robust rule on Delta-III.
Restated in round 3: hessian_approximation = "limited-memory" — what the reference configures by default (Core/LpNLPWrapper.hpp:71,
Core/LpNLPSolver.cpp:27-33), Ipopt's LimMemQuasiNewtonUpdater with its defaults: BFGS, limited_memory_max_history 6, the pair
s = x+ - x, y = grad_x L(x+, lambda+) - grad_x L(x, lambda+) (x only; the slacks' Hessian block is zero), skipped when
s'y <= sqrt(eps) |s| |y| (two skips in a row empty the memory), initial scaling "scalar1" sigma = s'y / s's in [1e-8, 1e8], and the
compact representation B = sigma I - [sigma S  Y] [[sigma S'S, L], [L', -D]]^-1 [sigma S'; Y'] (Byrd, Nocedal, Schnabel 1994) in the
place of the exact Hessian W.  B is positive definite, so the inertia is right without a correction.  The memory is emptied when
the restoration phase returns.  Entries of y at fixed variables are dropped (their rows of the KKT matrix are identity rows).
Ipopt's gradient-based NLP scaling is restated as an option (nlp_scaling_method = "gradient-based", Ipopt's default; "none" here).
Deliberately NOT restated (neither here nor on the device; DESIGN.md "f-2"): the quality-function oracle,
least-squares multipliers at the very first iterate by default (option init_ls_multipliers; lambda_0 = 0 otherwise), watchdog.  One deviation:
the constraint regularisation delta_c = 1e-9 is always on
(Ipopt: only for singular Jacobians), which is what makes the pivot-free LDL^T on the device well defined.
Parity status: UNPINNED by the reference (it holds no solver traces); pinned here by known optima and by scipy.

Dense linear algebra (numpy solve + eigenvalue inertia), one instance at a time; callbacks from oracle.Oracle.
"""
import numpy as np
import scipy.linalg
import scipy.sparse
import scipy.sparse.linalg

INF = 1e19

DEFAULTS = dict(tol=1e-8, mu_init=0.1, kappa_eps=10.0, kappa_mu=0.2, theta_mu=1.5, tau_min=0.99, bound_push=1e-2,
                bound_frac=1e-2, kappa_sigma=1e10, s_max=100.0, gamma_theta=1e-5, gamma_phi=1e-8, eta_phi=1e-8, delta=1.0,
                s_theta=1.1, s_phi=2.3, gamma_alpha=0.05, delta_c=1e-9, delta_w_first=1e-4, delta_w_min=1e-20,
                delta_w_max=1e40, kw_inc_first=100.0, kw_inc=8.0, kw_dec=1.0 / 3.0, max_iter=3000, max_ls=40, resto=1, resto_max=300, kappa_resto=0.9,
                acceptable_tol=1e-6, acceptable_iter=15, linear_solver="dense", bound_relax_factor=1e-8, max_soc=4, kappa_soc=0.99,
                resto_rho=1000.0, mult_reset=1e3,
                mu_strategy="adaptive", mu_max_fact=1e3, adaptive_mu_kkterror_red_iters=4, adaptive_mu_kkterror_red_fact=0.9999,
                adaptive_mu_monotone_init_factor=0.8, max_recalc_y=3,
                init_ls_multipliers=0,                # 1: least-squares multipliers at the very first iterate too (Ipopt's default start)
                nlp_scaling_method="none",            # "gradient-based": Ipopt's default NLP scaling (GradientScaling, nlp_scaling_max_gradient 100): the
                nlp_scaling_max_gradient=100.0,       #   objective and every constraint row are scaled down so that no gradient entry at the starting
                nlp_scaling_min_value=1e-8,           #   point exceeds 100; "none" is this restatement's default
                ic_hot_start=0, ic_hot_min=1e-10,     # 1: Algorithm IC starts at kw_dec * delta_w_last when the previous iteration needed delta_w > 0
                                                      #    and that value is >= ic_hot_min (NOT Ipopt's rule, which always tries 0 first: an experiment)
                hessian_approximation="exact", limited_memory_max_history=6, limited_memory_max_skipping=2,
                limited_memory_init_val_min=1e-8, limited_memory_init_val_max=1e8,
                dual_inf_tol=1.0, constr_viol_tol=1e-4, compl_inf_tol=1e-4,                      # Ipopt's unscaled termination thresholds
                acceptable_dual_inf_tol=1e10, acceptable_constr_viol_tol=1e-2, acceptable_compl_inf_tol=1e-2)


def _n_positive(K):
    """Number of positive eigenvalues by Sylvester's law from a Bunch-Kaufman LDL^T (eigenvalues of K itself would lose
    the 1e-8 constraint regularisation next to barrier terms of 1e+8); -1 when K holds NaN/Inf."""
    if not np.isfinite(K).all():
        return -1
    _, d, _ = scipy.linalg.ldl(K, lower=True)
    pos, i, nk = 0, 0, d.shape[0]
    while i < nk:
        if i + 1 < nk and d[i + 1, i] != 0.0:
            pos += int((np.linalg.eigvalsh(d[i:i + 2, i:i + 2]) > 0).sum())
            i += 2
        else:
            pos += int(d[i, i] > 0)
            i += 1
    return pos


def _push(x, l, u, o):
    """section 3.6: move x inside its bounds by min(kappa_1 max(1,|bound|), kappa_2 (u - l))"""
    lo, up = l > -INF, u < INF
    x = x.copy()
    both = lo & up
    with np.errstate(invalid="ignore"):
        return _push_body(x, l, u, o, lo, up, both)


def _push_body(x, l, u, o, lo, up, both):
    pl = np.where(both, np.minimum(o["bound_push"] * np.maximum(1.0, np.abs(l)), o["bound_frac"] * (u - l)),
                  o["bound_push"] * np.maximum(1.0, np.abs(l)))
    pu = np.where(both, np.minimum(o["bound_push"] * np.maximum(1.0, np.abs(u)), o["bound_frac"] * (u - l)),
                  o["bound_push"] * np.maximum(1.0, np.abs(u)))
    x[lo] = np.maximum(x[lo], (l + pl)[lo])
    x[up] = np.minimum(x[up], (u - pu)[up])
    return x


class _GradientScaled:
    """The NLP seen through Ipopt's gradient-based scaling: f -> sf f, c_i -> sc_i c_i with sf = min(1, gmax / |grad f(x0)|_inf),
    sc_i = min(1, gmax / |grad c_i(x0)|_inf) (columns of fixed variables left out, floor nlp_scaling_min_value), x unscaled."""

    def __init__(self, orc, x0, fixed, gmax, vmin):
        self.o, self.n, self.m = orc, orc.n, orc.m
        self.ji, self.jj = orc.jac_structure()
        g = np.abs(orc.eval_grad_f(x0))[~fixed]
        gr = g.max() if g.size else 0.0
        self.sf = max(gmax / gr, vmin) if gr > gmax else 1.0
        jv = np.abs(orc.eval_jac_g(x0))
        rmax = np.zeros(orc.m)
        keep = ~fixed[self.jj]
        np.maximum.at(rmax, self.ji[keep], jv[keep])
        self.sc = np.where(rmax > gmax, np.maximum(gmax / np.maximum(rmax, 1e-300), vmin), 1.0)

    def bounds(self):
        xl, xu, gl, gu = self.o.bounds()
        return xl, xu, np.where(np.abs(gl) < INF, gl * self.sc, gl), np.where(np.abs(gu) < INF, gu * self.sc, gu)

    def jac_structure(self):
        return self.ji, self.jj

    def hess_structure(self):
        return self.o.hess_structure()

    def eval_f(self, x):
        return self.sf * self.o.eval_f(x)

    def eval_grad_f(self, x):
        return self.sf * self.o.eval_grad_f(x)

    def eval_g(self, x):
        return self.sc * self.o.eval_g(x)

    def eval_jac_g(self, x):
        return self.sc[self.ji] * self.o.eval_jac_g(x)

    def eval_h(self, x, sigma, lam):
        return self.sf * self.o.eval_h(x, sigma, lam * self.sc / self.sf)     # as the device forms it: one scalar factor outside


def solve(orc, x0, x_l=None, x_u=None, **options):
    """-> dict(x, lambda, obj, status, iterations, kkt_error, trace); status codes as rpm_ipm_solve (0 converged, 1 acceptable level, ...)."""
    o = dict(DEFAULTS)
    o.update(options)
    n, m = orc.n, orc.m
    sf_u, sc_u = 1.0, np.ones(m)
    if o["nlp_scaling_method"] == "gradient-based":
        xl0, xu0 = orc.bounds()[:2] if x_l is None else (np.asarray(x_l, float), np.asarray(x_u, float))
        orc = _GradientScaled(orc, np.asarray(x0, float), xl0 == xu0, o["nlp_scaling_max_gradient"], o["nlp_scaling_min_value"])
        sf_u, sc_u = orc.sf, orc.sc
    xl, xu, gl, gu = orc.bounds()
    if x_l is not None:
        xl, xu = np.asarray(x_l, float), np.asarray(x_u, float)
    ji, jj = orc.jac_structure()
    lbfgs = o["hessian_approximation"] == "limited-memory"
    hi, hj = (np.zeros(0, dtype=int), np.zeros(0, dtype=int)) if lbfgs else orc.hess_structure()
    lm = dict(S=[], Y=[], sigma=1.0, skipped=0, prev=None, updates=0, skips=0)
    ineq = np.nonzero(gl != gu)[0]
    ns, nv = ineq.size, n + ineq.size
    row_slack = -np.ones(m, dtype=int)
    row_slack[ineq] = np.arange(ns)
    vl, vu = np.concatenate([xl, gl[ineq]]), np.concatenate([xu, gu[ineq]])
    free = vl != vu
    rel = o["bound_relax_factor"]                      # Ipopt's bound_relax_factor: finite bounds of free unknowns move out
    if rel > 0:
        vl = np.where(free & (vl > -INF), vl - rel * np.maximum(1.0, np.abs(vl)), vl)
        vu = np.where(free & (vu < INF), vu + rel * np.maximum(1.0, np.abs(vu)), vu)
    lo, up = free & (vl > -INF), free & (vu < INF)
    fxi = np.nonzero(~free)[0]

    x = np.where(xl == xu, xl, _push(np.asarray(x0, float), vl[:n], vu[:n], o))
    s = _push(orc.eval_g(x)[ineq], vl[n:], vu[n:], o)
    v = np.concatenate([x, s])
    zL, zU = lo.astype(float), up.astype(float)
    lam = np.zeros(m)
    mu, it, dw_last = o["mu_init"], 0, 0.0
    ic_hot = False                # the last regular iteration's factorisation needed delta_w > 0
    filt, trace = [], []
    theta_max = theta_min = None

    def cons(vv, g):
        c = g - gl
        c[ineq] = g[ineq] - vv[n:]
        return c

    def lnsum(vv):
        return np.log(vv[lo] - vl[lo]).sum() + np.log(vu[up] - vv[up]).sum()

    def jac_dense(jv):
        A = np.zeros((m, nv))
        A[ji, jj] = jv
        A[ineq, n + np.arange(ns)] = -1.0
        return A

    def kkt(diag11, W, A, diag22):
        K = np.zeros((nv + m, nv + m))
        K[:nv, :nv] = np.diag(diag11) if W is None else W + np.diag(diag11)
        K[nv:, :nv] = A
        K[:nv, nv:] = A.T
        K[nv:, nv:] = -np.diag(diag22)
        K[fxi, :] = 0.0                                   # fixed variables stay as identity rows
        K[:, fxi] = 0.0
        K[fxi, fxi] = 1.0
        return K

    def reset16(z, sl_, mu_, present):
        ks = o["kappa_sigma"]
        return np.where(present, np.maximum(np.minimum(z, ks * mu_ / sl_), mu_ / (ks * sl_)), 0.0)

    def ftb(w_, dw_, tau_, sign=-1.0):
        """fraction to the boundary (15) for w + a dw >= (1 - tau) w"""
        k = dw_ < 0
        return float(np.min(-tau_ * w_[k] / dw_[k])) if k.any() else 1.0

    n_resto = n_acc = n_recalc = 0
    free_mode, refs, mu_max = True, [], None          # mu_strategy = adaptive

    def lm_update(x_new, glag_x_new, lam_new):
        """Ipopt's LimMemQuasiNewtonUpdater::UpdateHessian (BFGS): one pair per accepted step."""
        if lm["prev"] is None:
            return
        x_old, grad_old, jv_old = lm["prev"]
        g_old = grad_old.copy()
        np.add.at(g_old, jj, jv_old * lam_new[ji])     # grad_x L(x_old, lambda_new): the SAME multipliers on both sides
        s_, y_ = x_new - x_old, glag_x_new - g_old
        y_ = np.where(free[:n], y_, 0.0)
        sn, yn = np.linalg.norm(s_), np.linalg.norm(y_)
        if sn == 0.0:
            return                                      # lambda changed only (recalc_y): nothing to learn
        sty = float(s_ @ y_)
        if not (sty > np.sqrt(np.finfo(float).eps) * sn * yn):
            lm["skips"] += 1
            lm["skipped"] += 1
            if lm["skipped"] >= o["limited_memory_max_skipping"]:
                lm["S"], lm["Y"], lm["sigma"], lm["skipped"] = [], [], 1.0, 0
            return
        lm["skipped"] = 0
        lm["S"].append(s_)
        lm["Y"].append(y_)
        if len(lm["S"]) > o["limited_memory_max_history"]:
            lm["S"].pop(0)
            lm["Y"].pop(0)
        lm["sigma"] = min(o["limited_memory_init_val_max"], max(o["limited_memory_init_val_min"], sty / float(s_ @ s_)))
        lm["updates"] += 1

    def lm_matrix():
        """B (n x n) from the compact representation."""
        Bm = lm["sigma"] * np.eye(n)
        if lm["S"]:
            Sm, Ym = np.array(lm["S"]).T, np.array(lm["Y"]).T
            SY = Sm.T @ Ym
            Lm, Dm = np.tril(SY, -1), np.diag(np.diag(SY))
            Mm = np.block([[lm["sigma"] * (Sm.T @ Sm), Lm], [Lm.T, -Dm]])
            Q = np.hstack([lm["sigma"] * Sm, Ym])
            Bm = Bm - Q @ np.linalg.solve(Mm, Q.T)
        return Bm

    def ls_multipliers(xr, jR):
        """least-squares multipliers (section 3.6): [[I, A^T], [A, -delta_c]] [w; lambda] = -[grad f - zL + zU; 0]; 0 when they exceed mult_reset"""
        gf = np.zeros(nv)
        gf[:n] = orc.eval_grad_f(xr)
        K = kkt(np.ones(nv), None, jac_dense(jR), o["delta_c"] * np.ones(m))
        sol = np.linalg.solve(K, -np.concatenate([np.where(free, gf - zL + zU, 0.0), np.zeros(m)]))
        return sol[nv:] if np.max(np.abs(sol[nv:])) <= o["mult_reset"] else np.zeros(m)

    def _restore():
        """Ipopt's restoration phase (paper section 3.3) with p, n eliminated from the Newton system; see the module header.
        Returns None (back to the regular iteration) or a status (3 restoration failed, 2 iteration limit)."""
        nonlocal v, lam, zL, zU, it, n_resto
        filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))
        rho = o["resto_rho"]
        vR, th0 = v.copy(), theta
        Dr2 = np.where(free, 1.0 / np.maximum(1.0, np.abs(vR)) ** 2, 0.0)
        mu_r = max(mu, float(np.max(np.abs(c))))
        zeta = np.sqrt(mu_r)
        t_ = (mu_r - rho * c) / (2 * rho)                                                           # (33)
        nn = t_ + np.sqrt(t_ ** 2 + mu_r * c / (2 * rho))
        pp = c + nn                                                                                 # (34)
        zp, zn = mu_r / pp, mu_r / nn
        zL = np.where(lo, np.minimum(rho, zL), 0.0)
        zU = np.where(up, np.minimum(rho, zU), 0.0)
        lam = np.zeros(m)
        rfilt = []
        thr_max = thr_min = None
        mu_min = o["tol"] / 10.0
        for itr in range(o["resto_max"] + 1):
            xr = v[:n]
            gR, jR, fo = orc.eval_g(xr), orc.eval_jac_g(xr), orc.eval_f(xr)
            cR = cons(v, gR)
            th_o, ln_b = np.abs(cR).sum(), lnsum(v)
            if itr > 0 and th_o <= o["kappa_resto"] * th0 and th_o <= theta_max and \
                    not any(th_o >= a_ and fo - mu * ln_b >= b_ for a_, b_ in filt):
                dl, du = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
                zL = reset16(np.minimum(zL, 1e3), dl, mu, lo)
                zU = reset16(np.minimum(zU, 1e3), du, mu, up)
                lam = ls_multipliers(xr, jR)
                n_resto += 1
                return None
            if itr == o["resto_max"]:
                return 3
            if it >= o["max_iter"]:
                return 2
            A = jac_dense(jR)
            rc = cR - pp + nn
            th_r = np.abs(rc).sum()
            if thr_max is None:
                thr_max, thr_min = 1e4 * max(1.0, th_r), 1e-4 * max(1.0, th_r)
            dl, du = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
            dev = Dr2 * (v - vR)
            qd = float(np.sum(dev * (v - vR)))
            glag = A.T @ lam                                                     # without the proximity term zeta D_R^2 (v - v_R)
            dinf = max(np.max(np.abs((zeta * dev + glag - zL + zU)[free])), np.max(np.abs(rho - lam - zp)), np.max(np.abs(rho + lam - zn)))
            prods = np.concatenate([(zL * dl)[lo], (zU * du)[up], zp * pp, zn * nn])
            cmax, cmin = prods.max(), prods.min()
            for _ in range(64):
                emu = max(dinf, np.max(np.abs(rc)), abs(cmax - mu_r), abs(cmin - mu_r))
                if not (emu <= o["kappa_eps"] * mu_r):
                    break
                if mu_r <= mu_min:
                    return 3                                                     # a minimiser of the infeasibility the filter does not take
                mu_r = max(mu_min, min(o["kappa_mu"] * mu_r, mu_r ** o["theta_mu"]))
                rfilt = []
            zeta = np.sqrt(mu_r)
            tau_r = max(o["tau_min"], 1.0 - mu_r)
            sig = np.where(lo, zL / dl, 0.0) + np.where(up, zU / du, 0.0)
            sp, sn = zp / pp, zn / nn
            rp, rn = rho - lam - mu_r / pp, rho + lam - mu_r / nn
            K = kkt(zeta * Dr2 + sig, None, A, 1.0 / sp + 1.0 / sn)
            gb = zeta * dev - np.where(lo, mu_r / dl, 0.0) + np.where(up, mu_r / du, 0.0)     # gradient of the barrier objective in v
            sol = np.linalg.solve(K, np.concatenate([-np.where(free, gb + glag, 0.0), -rc - rp / sp + rn / sn]))
            d, dlam = np.where(free, sol[:nv], 0.0), sol[nv:]
            dp, dn = (dlam - rp) / sp, (-dlam - rn) / sn
            dzp, dzn = mu_r / pp - zp - sp * dp, mu_r / nn - zn - sn * dn
            dzL = np.where(lo, mu_r / dl - zL - zL / dl * d, 0.0)
            dzU = np.where(up, mu_r / du - zU + zU / du * d, 0.0)
            ar = min(1.0, ftb(dl[lo], d[lo], tau_r), ftb(du[up], -d[up], tau_r), ftb(pp, dp, tau_r), ftb(nn, dn, tau_r))
            az = min(1.0, ftb(zL[lo], dzL[lo], tau_r), ftb(zU[up], dzU[up], tau_r), ftb(zp, dzp, tau_r), ftb(zn, dzn, tau_r))
            ln_r = ln_b + np.log(pp).sum() + np.log(nn).sum()
            phi_r = rho * (pp.sum() + nn.sum()) + 0.5 * zeta * qd - mu_r * ln_r
            dphi = float(np.where(free, gb, 0.0) @ d + (rho - mu_r / pp) @ dp + (rho - mu_r / nn) @ dn)
            amin = o["gamma_theta"]
            if dphi < 0:
                amin = min(amin, o["gamma_phi"] * th_r / (-dphi))
                if th_r <= thr_min:
                    amin = min(amin, o["delta"] * th_r ** o["s_theta"] / (-dphi) ** o["s_phi"])
            amin *= o["gamma_alpha"]
            slack = 10.0 * np.finfo(float).eps * abs(phi_r)
            ls, moved, arm = 0, False, False
            while True:
                vt, pt, nt = v + ar * d, pp + ar * dp, nn + ar * dn
                with np.errstate(all="ignore"):
                    tht = np.abs(cons(vt, orc.eval_g(vt[:n])) - pt + nt).sum()
                    phit = rho * (pt.sum() + nt.sum()) + 0.5 * zeta * np.sum(Dr2 * (vt - vR) ** 2) - mu_r * (lnsum(vt) + np.log(pt).sum() + np.log(nt).sum())
                ok = False
                if np.isfinite([tht, phit]).all() and tht <= thr_max and not any(tht >= a_ and phit >= b_ for a_, b_ in rfilt):
                    sw = dphi < 0 and ar * (-dphi) ** o["s_phi"] > o["delta"] * th_r ** o["s_theta"]
                    if th_r <= thr_min and sw:
                        ok = phit - phi_r - o["eta_phi"] * ar * dphi <= slack
                        arm = ok
                    else:
                        ok = tht <= (1 - o["gamma_theta"]) * th_r or phit - (phi_r - o["gamma_phi"] * th_r) <= slack
                if ok:
                    moved = True
                    break
                ar *= 0.5
                ls += 1
                if ar < amin or ls > o["max_ls"]:
                    break
            if not moved:
                return 3
            v = np.where(free, vt, v)
            pp, nn = pt, nt
            lam = lam + ar * dlam
            dl, du = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
            zL, zU = reset16(zL + az * dzL, dl, mu_r, lo), reset16(zU + az * dzU, du, mu_r, up)
            zp, zn = reset16(zp + az * dzp, pp, mu_r, True), reset16(zn + az * dzn, nn, mu_r, True)
            if not arm:
                rfilt.append(((1 - o["gamma_theta"]) * th_r, phi_r - o["gamma_phi"] * th_r))
            trace.append(dict(it=it, f=fo, theta=th_o, mu=mu_r, alpha=ar, alpha_z=az, delta_w=0.0, err0=err0, ls=-1))
            it += 1
        return 3

    status = None
    if o["init_ls_multipliers"] and m:
        lam = ls_multipliers(v[:n], orc.eval_jac_g(v[:n]))
    while True:
        x = v[:n]
        f, grad, g, jv = orc.eval_f(x), orc.eval_grad_f(x), orc.eval_g(x), orc.eval_jac_g(x)
        glag = np.zeros(nv)
        glag[:n] = grad
        np.add.at(glag, jj, jv * lam[ji])
        glag[n:] = -lam[ineq]
        c = cons(v, g)
        if lbfgs:
            lm_update(x.copy(), glag[:n].copy(), lam)
            lm["prev"] = (x.copy(), grad.copy(), jv.copy())
        dinf = np.max(np.abs((glag - zL + zU)[free])) if free.any() else 0.0
        cinf, theta = (np.max(np.abs(c)), np.abs(c).sum()) if m else (0.0, 0.0)
        prods = np.concatenate([zL[lo] * (v[lo] - vl[lo]), zU[up] * (vu[up] - v[up])])
        nzb = prods.size
        sz = zL[lo].sum() + zU[up].sum()
        sd = max(o["s_max"], (np.abs(lam).sum() + sz) / max(1.0, m + nzb)) / o["s_max"]          # (6)
        sc = max(o["s_max"], sz / nzb) / o["s_max"] if nzb else 1.0
        cmax, cmin = (prods.max(), prods.min()) if nzb else (0.0, 1e300)
        err0 = max(dinf / sd, cinf, cmax / sc if nzb else 0.0)                                    # (5), mu = 0
        ln = lnsum(v)
        if not np.isfinite([f, ln, dinf, cinf]).all():
            status = 5
        # Ipopt's secondary thresholds apply to the UNSCALED problem (nlp_scaling_method): gradient and complementarity / sf, rows / sc
        dinf_u, cinf_u, cm_u = dinf / sf_u, (float(np.max(np.abs(c / sc_u))) if m else 0.0), (cmax / sf_u if nzb else 0.0)
        if status is not None:
            pass
        elif err0 <= o["tol"] and dinf_u <= o["dual_inf_tol"] and cinf_u <= o["constr_viol_tol"] and cm_u <= o["compl_inf_tol"]:
            status = 0
        else:
            acc_ok = err0 <= o["acceptable_tol"] and dinf_u <= o["acceptable_dual_inf_tol"] and cinf_u <= o["acceptable_constr_viol_tol"] and \
                cm_u <= o["acceptable_compl_inf_tol"]
            n_acc = n_acc + 1 if acc_ok else 0            # Ipopt: acceptable_tol 1e-6, acceptable_iter 15
            if o["acceptable_iter"] > 0 and n_acc >= o["acceptable_iter"]:
                status = 1
            elif it >= o["max_iter"]:
                status = 2
        if status is not None:
            break
        if it == 0:
            theta_max, theta_min = 1e4 * max(1.0, theta), 1e-4 * max(1.0, theta)
        mu_min = o["tol"] / 10.0
        adaptive = o["mu_strategy"] == "adaptive" and nzb > 0
        oracle_mu = None
        if adaptive:
            avg = float(prods.mean())
            if mu_max is None:
                mu_max = o["mu_max_fact"] * avg
            dres = (glag - zL + zU)[free]
            kkt_err = float(dres @ dres) / max(1, dres.size) + (float(c @ c) / m if m else 0.0) + float(prods @ prods) / nzb
            full = len(refs) >= o["adaptive_mu_kkterror_red_iters"]
            progress = (not full) or any(kkt_err <= o["adaptive_mu_kkterror_red_fact"] * r_ for r_ in refs)
            if progress:
                free_mode = True
                refs.append(kkt_err)
                if len(refs) > o["adaptive_mu_kkterror_red_iters"]:
                    refs.pop(0)
                xi = float(prods.min()) / avg
                sigma_l = 0.1 * min(0.05 * (1.0 - xi) / xi, 2.0) ** 3
                oracle_mu = max(mu_min, min(sigma_l * avg, mu_max))
            elif free_mode:                     # no progress in the free mode: the monotone rule takes over from here
                free_mode = False
                oracle_mu = max(mu_min, min(o["adaptive_mu_monotone_init_factor"] * avg, mu_max))
        if oracle_mu is not None:
            if oracle_mu != mu:
                filt = []
            mu = oracle_mu
        else:
            for _ in range(64):
                comp = max(abs(cmax - mu), abs(cmin - mu)) if nzb else 0.0
                emu = max(dinf / sd, cinf, comp / sc)
                if not (emu <= o["kappa_eps"] * mu) or mu <= mu_min:
                    break
                mu = max(mu_min, min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))                        # (7)
                filt = []
        tau = max(o["tau_min"], 1.0 - mu)                                                         # (8)
        phi = f - mu * ln
        W = np.zeros((nv, nv))
        if lbfgs:
            hv = np.zeros(0)
            W[:n, :n] = lm_matrix()
        else:
            hv = orc.eval_h(x, 1.0, lam)
            np.add.at(W, (hi, hj), hv)
            W = W + np.tril(W, -1).T
        A = jac_dense(jv)
        dl, du = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
        sigma = np.where(lo, zL / dl, 0.0) + np.where(up, zU / du, 0.0)
        rd = glag - np.where(lo, mu / dl, 0.0) + np.where(up, mu / du, 0.0)
        dw = 0.0
        if o["ic_hot_start"] and ic_hot and o["kw_dec"] * dw_last >= o["ic_hot_min"]:
            dw = o["kw_dec"] * dw_last
        Ksp = None
        if o["linear_solver"] == "sparse-lu-no-inertia":
            # timing variant for bench cpu_baseline legs ONLY: sparse LU of the same matrix, inertia taken on trust (valid on
            # problems that need no inertia correction, e.g. the quadrotor sweep; results are then the dense path's to rounding)
            Ws = scipy.sparse.coo_matrix((hv, (hi, hj)), shape=(nv, nv)).tocsr()
            Ws = Ws + scipy.sparse.tril(Ws, -1).T + scipy.sparse.diags(sigma)
            As = scipy.sparse.coo_matrix((np.concatenate([jv, -np.ones(ns)]), (np.concatenate([ji, ineq]), np.concatenate([jj, n + np.arange(ns)]))),
                                         shape=(m, nv)).tocsr()
            keep = scipy.sparse.diags(free.astype(float))
            Ks = scipy.sparse.bmat([[keep @ Ws @ keep + scipy.sparse.diags((~free).astype(float)), keep @ As.T],
                                    [As @ keep, -o["delta_c"] * scipy.sparse.identity(m)]], format="csc")
            Ksp = scipy.sparse.linalg.splu(Ks)
        while Ksp is None:                                                                        # Algorithm IC
            K = kkt(sigma + dw, W, A, o["delta_c"] * np.ones(m))
            if _n_positive(K) == nv:
                if dw > 0:
                    dw_last = dw
                ic_hot = dw > 0
                break
            if dw == 0.0:
                dw = o["delta_w_first"] if dw_last == 0.0 else max(o["delta_w_min"], o["kw_dec"] * dw_last)
            else:
                dw *= o["kw_inc_first"] if dw_last == 0.0 else o["kw_inc"]
            if dw > o["delta_w_max"]:
                status = 4
                break
        if status is not None:
            break
        lu = scipy.linalg.lu_factor(K) if Ksp is None else None

        def newton(c_rhs):
            """the step for constraint right-hand side c_rhs (c itself, or the second-order correction's c_soc)"""
            rhs = -np.concatenate([np.where(free, rd, 0.0), c_rhs])
            sol = scipy.linalg.lu_solve(lu, rhs) if Ksp is None else Ksp.solve(rhs)
            dv_, dlam_ = np.where(free, sol[:nv], 0.0), sol[nv:]
            dzL_ = np.where(lo, mu / dl - zL - zL / dl * dv_, 0.0)                                # (12)
            dzU_ = np.where(up, mu / du - zU + zU / du * dv_, 0.0)
            amax_ = min(1.0, ftb(dl[lo], dv_[lo], tau), ftb(du[up], -dv_[up], tau))               # (15)
            az_ = min(1.0, ftb(zL[lo], dzL_[lo], tau), ftb(zU[up], dzU_[up], tau))
            return dv_, dlam_, dzL_, dzU_, amax_, az_

        dv, dlam, dzL, dzU, amax, az = newton(c)
        gphi = np.concatenate([grad, np.zeros(ns)]) - np.where(lo, mu / dl, 0.0) + np.where(up, mu / du, 0.0)
        dphi = float(np.dot(np.where(free, gphi, 0.0), dv))
        amin = o["gamma_theta"]                                                                   # (23)
        if dphi < 0:
            amin = min(amin, o["gamma_phi"] * theta / (-dphi))
            if theta <= theta_min:
                amin = min(amin, o["delta"] * theta ** o["s_theta"] / (-dphi) ** o["s_phi"])
        amin *= o["gamma_alpha"]
        a, ls, armijo, accepted, n_soc = amax, 0, False, False, 0
        slack = 10.0 * np.finfo(float).eps * abs(phi)

        def acceptable(vt, a_test):
            """filter + switching / Armijo / sufficient-decrease tests (18)-(20) at a trial point -> (ok, armijo, theta_trial, c_trial)"""
            with np.errstate(all="ignore"):
                ft, gt = orc.eval_f(vt[:n]), orc.eval_g(vt[:n])
                ct = cons(vt, gt)
                tht = np.abs(ct).sum() if m else 0.0
                phit = ft - mu * lnsum(vt)
            if not (np.isfinite([ft, tht, phit]).all() and tht <= theta_max):
                return False, False, tht, ct
            if any(tht >= ft_ and phit >= fp_ for ft_, fp_ in filt):
                return False, False, tht, ct
            sw = dphi < 0 and a_test * (-dphi) ** o["s_phi"] > o["delta"] * theta ** o["s_theta"]          # (19)
            if theta <= theta_min and sw:
                ok_ = phit - phi - o["eta_phi"] * a_test * dphi <= slack                                   # (20)
                return ok_, ok_, tht, ct
            return (tht <= (1 - o["gamma_theta"]) * theta or phit - (phi - o["gamma_phi"] * theta) <= slack), False, tht, ct   # (18)

        while True:
            ok, armijo, tht, ct = acceptable(v + a * dv, a)
            if ok:
                accepted = True
                break
            if ls == 0 and o["max_soc"] > 0 and np.isfinite(tht) and tht >= theta:
                # second-order correction, A-5.5 .. A-5.9: the same matrix, c_soc = alpha c_soc + c(trial point)
                csoc, th_old, a_soc = c, theta, a
                for p_soc in range(o["max_soc"]):
                    csoc = a_soc * csoc + ct
                    dvs, dlams, dzLs, dzUs, a_soc, azs = newton(csoc)
                    ok, arm_s, ths, ct = acceptable(v + a_soc * dvs, a)
                    if ok:
                        dv, dlam, dzL, dzU, az, a, armijo = dvs, dlams, dzLs, dzUs, azs, a_soc, arm_s
                        accepted, n_soc = True, p_soc + 1
                        break
                    if not np.isfinite(ths) or ths > o["kappa_soc"] * th_old:
                        break
                    th_old = ths
                if accepted:
                    break
            a *= 0.5
            ls += 1
            if a < amin or ls > o["max_ls"]:
                status = 1 if err0 <= o["acceptable_tol"] else 3
                break
        if not accepted and status == 3 and o["resto"] and theta > o["tol"]:
            status = _restore()
            lm.update(S=[], Y=[], sigma=1.0, skipped=0, prev=None)      # the point moved by another problem's steps
            if status is None:
                continue
        elif not accepted and status == 3 and o["resto"] and n_recalc < o["max_recalc_y"]:
            # the line search gave up at a FEASIBLE point (theta <= tol: nothing for the restoration phase to do) whose
            # multipliers are off — seen at the end of Delta-III solves, the primal variables converged, the dual infeasibility
            # O(10).  Ipopt's recalc_y: least-squares multipliers at this point, then on with the regular iteration.
            lam = ls_multipliers(x, jv)
            n_recalc += 1
            status = None
            continue
        if not accepted:
            break
        v = np.where(free, v + a * dv, v)
        lam = lam + a * dlam
        dl, du = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
        zL, zU = reset16(zL + az * dzL, dl, mu, lo), reset16(zU + az * dzU, du, mu, up)           # (16)
        if not armijo:
            filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))             # (22)
        trace.append(dict(it=it, f=f, theta=theta, mu=mu, alpha=a, alpha_z=az, delta_w=dw, err0=err0, ls=ls, soc=n_soc, dinf=dinf, cinf=cinf, comp=cmax,
                          smin=float(min(dl[lo].min(initial=1e300), du[up].min(initial=1e300)))))
        it += 1
    return dict(x=v[:n].copy(), slack=v[n:].copy(), **{"lambda": lam * sc_u / sf_u}, obj=f / sf_u, status=status, iterations=it, kkt_error=err0, trace=trace, restorations=n_resto,
                multiplier_recalculations=n_recalc, lm_updates=lm["updates"], lm_skips=lm["skips"])
