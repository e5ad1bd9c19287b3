"""CPU restatement of the interior-point iteration behind rpm_ipm_* (row f-2) — TEST INFRASTRUCTURE, like everything
under oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

What it restates: the algorithm of Ipopt 3.12.3, the solver the reference calls (NLPSolver::SolveNlp,
Core/LpNLPSolver.cpp:13-53).  Ipopt is a third-party dependency that is absent from /root/reference, so this follows
its published description — A. Waechter, L. T. Biegler, "On the implementation of an interior-point filter line-search
algorithm for large-scale nonlinear programming", Math. Program. 106 (2006) — equation numbers below are the paper's:
  optimality error (5)/(6), barrier update (7), tau (8), primal-dual system (13) with dz from (12), fraction to the
  boundary (15), multiplier reset (16), filter acceptance (18)-(20), filter update (22), alpha_min (23), inertia
  correction Algorithm IC, initial point section 3.6 (bound_push / bound_frac), constants = Ipopt 3.12 defaults.
Deliberately NOT restated (neither here nor on the device; DESIGN.md "f-2"): Ipopt's restoration phase (its place is
taken by a much simpler Gauss-Newton feasibility restoration, _restore below, built from the same KKT kernels), second-order
correction, adaptive barrier update (the reference sets mu_strategy=adaptive; monotone here), NLP scaling, least-squares
multiplier initialisation (lambda_0 = 0).  One deviation: the constraint regularisation delta_c = 1e-8 is always on
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
                s_theta=1.1, s_phi=2.3, gamma_alpha=0.05, delta_c=1e-8, delta_w_first=1e-4, delta_w_min=1e-20,
                delta_w_max=1e40, kw_inc_first=100.0, kw_inc=8.0, kw_dec=1.0 / 3.0, max_iter=3000, max_ls=40, resto=1, resto_max=60, kappa_resto=0.9, acceptable_tol=1e-6, acceptable_iter=15, linear_solver="dense")


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


def solve(orc, x0, x_l=None, x_u=None, **options):
    """-> dict(x, lambda, obj, status, iterations, kkt_error, trace); status codes as rpm_ipm_solve (0 converged, 1 acceptable level, ...)."""
    o = dict(DEFAULTS)
    o.update(options)
    n, m = orc.n, orc.m
    xl, xu, gl, gu = orc.bounds()
    if x_l is not None:
        xl, xu = np.asarray(x_l, float), np.asarray(x_u, float)
    ji, jj = orc.jac_structure()
    hi, hj = orc.hess_structure()
    ineq = np.nonzero(gl != gu)[0]
    ns, nv = ineq.size, n + ineq.size
    row_slack = -np.ones(m, dtype=int)
    row_slack[ineq] = np.arange(ns)
    vl, vu = np.concatenate([xl, gl[ineq]]), np.concatenate([xu, gu[ineq]])
    free = vl != vu
    lo, up = free & (vl > -INF), free & (vu < INF)

    x = np.where(xl == xu, xl, _push(np.asarray(x0, float), xl, xu, o))
    s = _push(orc.eval_g(x)[ineq], gl[ineq], gu[ineq], o)
    v = np.concatenate([x, s])
    zL, zU = lo.astype(float), up.astype(float)
    lam = np.zeros(m)
    mu, it, dw_last = o["mu_init"], 0, 0.0
    filt, trace = [], []
    theta_max = theta_min = None

    def cons(vv, g):
        c = g - gl
        c[ineq] = g[ineq] - vv[n:]
        return c

    def lnsum(vv):
        return np.log(vv[lo] - vl[lo]).sum() + np.log(vu[up] - vv[up]).sum()

    n_resto = n_acc = 0

    def _restore():
        """Feasibility restoration (NOT Ipopt's l1 restoration NLP; see the module header): damped Gauss-Newton on
        psi(v) = 1/2 |c(v)|^2 + zeta/2 |D_R (v - v_R)|^2 - mu sum ln(bound slacks), zeta = sqrt(mu), D_R = diag(1/max(1,|v_R|)),
        each step from the same KKT matrix with W = zeta D_R^2 + mu/s^2 and -I in the constraint block, until the
        infeasibility has dropped to kappa_resto = 0.9 of where the line search gave up and the point is acceptable to the
        filter.  Returns None (back to the regular iteration, lambda = 0, bound multipliers clipped) or status 3."""
        nonlocal v, lam, zL, zU, it, n_resto
        filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))
        vR, th0 = v.copy(), theta
        Dr2 = 1.0 / np.maximum(1.0, np.abs(vR)) ** 2
        zeta = np.sqrt(mu)
        for itr in range(o["resto_max"] + 1):
            xr = v[:n]
            gR, jR, fR = orc.eval_g(xr), orc.eval_jac_g(xr), orc.eval_f(xr)
            cR = cons(v, gR)
            thR, lnR = np.abs(cR).sum(), lnsum(v)
            if itr > 0 and thR <= o["kappa_resto"] * th0 and thR <= theta_max and \
                    not any(thR >= a_ and fR - mu * lnR >= b_ for a_, b_ in filt):
                lam = np.zeros(m)
                sl_, su_ = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
                ks = o["kappa_sigma"]
                zL = np.where(lo, np.maximum(np.minimum(np.minimum(zL, 1e3), ks * mu / sl_), mu / (ks * sl_)), 0.0)
                zU = np.where(up, np.maximum(np.minimum(np.minimum(zU, 1e3), ks * mu / su_), mu / (ks * su_)), 0.0)
                n_resto += 1
                return None
            if itr == o["resto_max"]:
                return 3
            A = np.zeros((m, nv))
            A[ji, jj] = jR
            A[ineq, n + np.arange(ns)] = -1.0
            sl_, su_ = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
            Md = zeta * Dr2 + np.where(lo, mu / sl_ ** 2, 0.0) + np.where(up, mu / su_ ** 2, 0.0)
            gb = zeta * Dr2 * (v - vR) - np.where(lo, mu / sl_, 0.0) + np.where(up, mu / su_, 0.0)
            K = np.zeros((nv + m, nv + m))
            K[:nv, :nv] = np.diag(Md)
            K[nv:, :nv] = A
            K[:nv, nv:] = A.T
            K[nv:, nv:] = -np.eye(m)
            fxi = np.nonzero(~free)[0]
            K[fxi, :] = 0.0
            K[:, fxi] = 0.0
            K[fxi, fxi] = 1.0
            sol = np.linalg.solve(K, -np.concatenate([np.where(free, gb, 0.0), cR]))
            d, w = np.where(free, sol[:nv], 0.0), sol[nv:]
            ar = 1.0
            k = lo & (d < 0)
            if k.any():
                ar = min(ar, np.min(-tau * sl_[k] / d[k]))
            k = up & (d > 0)
            if k.any():
                ar = min(ar, np.min(tau * su_[k] / d[k]))
            psi = 0.5 * cR @ cR + 0.5 * zeta * np.sum(Dr2 * (v - vR) ** 2) - mu * lnR
            slope = float((w - cR) @ cR + np.where(free, gb, 0.0) @ d)          # (A d)^T c + g_b^T d,  A d = w - c
            moved = False
            for _ in range(o["max_ls"]):
                vt = v + ar * d
                with np.errstate(all="ignore"):
                    ct = cons(vt, orc.eval_g(vt[:n]))
                    psit = 0.5 * ct @ ct + 0.5 * zeta * np.sum(Dr2 * (vt - vR) ** 2) - mu * lnsum(vt)
                if np.isfinite(psit) and psit <= psi + 1e-4 * ar * slope:
                    v = np.where(free, vt, v)
                    moved = True
                    break
                ar *= 0.5
            trace.append(dict(it=it, f=fR, theta=thR, mu=mu, alpha=ar, alpha_z=0.0, delta_w=0.0, err0=err0, ls=-1))
            it += 1
            if not moved:
                return 3
        return 3

    status = None
    while True:
        x = v[:n]
        f, grad, g, jv = orc.eval_f(x), orc.eval_grad_f(x), orc.eval_g(x), orc.eval_jac_g(x)
        glag = np.zeros(nv)
        glag[:n] = grad
        np.add.at(glag, jj, jv * lam[ji])
        glag[n:] = -lam[ineq]
        c = cons(v, g)
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
        elif err0 <= o["tol"]:
            status = 0
        else:
            n_acc = n_acc + 1 if err0 <= o["acceptable_tol"] else 0            # Ipopt: acceptable_tol 1e-6, acceptable_iter 15
            if o["acceptable_iter"] > 0 and n_acc >= o["acceptable_iter"]:
                status = 1
            elif it >= o["max_iter"]:
                status = 2
        if status is not None:
            break
        if it == 0:
            theta_max, theta_min = 1e4 * max(1.0, theta), 1e-4 * max(1.0, theta)
        mu_min = o["tol"] / 10.0
        for _ in range(64):
            comp = max(abs(cmax - mu), abs(cmin - mu)) if nzb else 0.0
            emu = max(dinf / sd, cinf, comp / sc)
            if not (emu <= o["kappa_eps"] * mu) or mu <= mu_min:
                break
            mu = max(mu_min, min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))                        # (7)
            filt = []
        tau = max(o["tau_min"], 1.0 - mu)                                                         # (8)
        phi = f - mu * ln
        hv = orc.eval_h(x, 1.0, lam)
        W = np.zeros((nv, nv))
        np.add.at(W, (hi, hj), hv)
        W = W + np.tril(W, -1).T
        A = np.zeros((m, nv))
        A[ji, jj] = jv
        A[ineq, n + np.arange(ns)] = -1.0
        dl, du = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
        sigma = np.where(lo, zL / dl, 0.0) + np.where(up, zU / du, 0.0)
        rd = glag - np.where(lo, mu / dl, 0.0) + np.where(up, mu / du, 0.0)
        fxi = np.nonzero(~free)[0]
        dw = 0.0
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
            K = np.zeros((nv + m, nv + m))
            K[:nv, :nv] = W + np.diag(sigma + dw)
            K[nv:, :nv] = A
            K[:nv, nv:] = A.T
            K[nv:, nv:] = -o["delta_c"] * np.eye(m)
            K[fxi, :] = 0.0                                   # fixed variables stay as identity rows
            K[:, fxi] = 0.0
            K[fxi, fxi] = 1.0
            if _n_positive(K) == nv:
                if dw > 0:
                    dw_last = dw
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
        rhs = -np.concatenate([np.where(free, rd, 0.0), c])
        sol = np.linalg.solve(K, rhs) if Ksp is None else Ksp.solve(rhs)
        dv, dlam = np.where(free, sol[:nv], 0.0), sol[nv:]
        dzL = np.where(lo, mu / dl - zL - zL / dl * dv, 0.0)                                      # (12)
        dzU = np.where(up, mu / du - zU + zU / du * dv, 0.0)
        amax, az = 1.0, 1.0                                                                       # (15)
        k = lo & (dv < 0)
        if k.any():
            amax = min(amax, np.min(-tau * dl[k] / dv[k]))
        k = up & (dv > 0)
        if k.any():
            amax = min(amax, np.min(tau * du[k] / dv[k]))
        k = lo & (dzL < 0)
        if k.any():
            az = min(az, np.min(-tau * zL[k] / dzL[k]))
        k = up & (dzU < 0)
        if k.any():
            az = min(az, np.min(-tau * zU[k] / dzU[k]))
        gphi = np.concatenate([grad, np.zeros(ns)]) - np.where(lo, mu / dl, 0.0) + np.where(up, mu / du, 0.0)
        dphi = float(np.dot(np.where(free, gphi, 0.0), dv))
        amin = o["gamma_theta"]                                                                   # (23)
        if dphi < 0:
            amin = min(amin, o["gamma_phi"] * theta / (-dphi))
            if theta <= theta_min:
                amin = min(amin, o["delta"] * theta ** o["s_theta"] / (-dphi) ** o["s_phi"])
        amin *= o["gamma_alpha"]
        a, ls, armijo, accepted = amax, 0, False, False
        slack = 10.0 * np.finfo(float).eps * abs(phi)
        while True:
            vt = v + a * dv
            with np.errstate(all="ignore"):
                ft, gt = orc.eval_f(vt[:n]), orc.eval_g(vt[:n])
                tht = np.abs(cons(vt, gt)).sum() if m else 0.0
                phit = ft - mu * lnsum(vt)
            ok = False
            if np.isfinite([ft, tht, phit]).all() and tht <= theta_max:
                if not any(tht >= ft_ and phit >= fp_ for ft_, fp_ in filt):
                    sw = dphi < 0 and a * (-dphi) ** o["s_phi"] > o["delta"] * theta ** o["s_theta"]      # (19)
                    if theta <= theta_min and sw:
                        ok = phit - phi - o["eta_phi"] * a * dphi <= slack                                 # (20)
                        armijo = ok
                    else:
                        ok = tht <= (1 - o["gamma_theta"]) * theta or phit - (phi - o["gamma_phi"] * theta) <= slack   # (18)
            if ok:
                accepted = True
                break
            a *= 0.5
            ls += 1
            if a < amin or ls > o["max_ls"]:
                status = 1 if err0 <= o["acceptable_tol"] else 3
                break
        if not accepted and status == 3 and o["resto"] and theta > o["tol"]:
            status = _restore()
            if status is None:
                continue
        if not accepted:
            break
        v = np.where(free, v + a * dv, v)
        lam = lam + a * dlam
        ks = o["kappa_sigma"]
        dl, du = np.where(lo, v - vl, 1.0), np.where(up, vu - v, 1.0)
        zL = np.where(lo, np.maximum(np.minimum(zL + az * dzL, ks * mu / dl), mu / (ks * dl)), 0.0)    # (16)
        zU = np.where(up, np.maximum(np.minimum(zU + az * dzU, ks * mu / du), mu / (ks * du)), 0.0)
        if not armijo:
            filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))             # (22)
        trace.append(dict(it=it, f=f, theta=theta, mu=mu, alpha=a, alpha_z=az, delta_w=dw, err0=err0, ls=ls))
        it += 1
    return dict(x=v[:n].copy(), slack=v[n:].copy(), **{"lambda": lam.copy()}, obj=f, status=status, iterations=it, kkt_error=err0, trace=trace, restorations=n_resto)
