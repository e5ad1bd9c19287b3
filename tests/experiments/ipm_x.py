"""Sandbox for the interior-point restatement (NOT shipped, NOT the oracle): switches for the pieces of Ipopt that
oracle/ipm_oracle.py leaves out, to find what Delta-III needs.  Dense numpy."""
import numpy as np
import scipy.linalg

from oracle.ipm_oracle import DEFAULTS, INF, _push

X = dict(hess="exact", mu_strategy="monotone", bound_relax=0.0, ls_mult=0, soc=0, resto="gn", mu_oracle="qf", verbose=1,
         reset_filter_on_mu=1, lbfgs_m=6, mu_max_fact=1e3, sigma_max=1e2, sigma_min=1e-6, resto_rho=1e3, resto_iter=200)


def inertia_pos(K):
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


class P:
    """the NLP in slack form: v = [x; s], c(v) = 0, vl <= v <= vu"""

    def __init__(self, orc, o, x_l=None, x_u=None):
        self.orc, self.n, self.m = orc, orc.n, orc.m
        xl, xu, gl, gu = orc.bounds()
        if x_l is not None:
            xl, xu = np.asarray(x_l, float), np.asarray(x_u, float)
        self.ji, self.jj = orc.jac_structure()
        self.hi, self.hj = orc.hess_structure()
        self.ineq = np.nonzero(gl != gu)[0]
        self.ns = self.ineq.size
        self.nv = self.n + self.ns
        self.gl = gl
        vl, vu = np.concatenate([xl, gl[self.ineq]]), np.concatenate([xu, gu[self.ineq]])
        self.free = vl != vu
        if o["bound_relax"] > 0:
            r = o["bound_relax"]
            vl = np.where(self.free & (vl > -INF), vl - r * np.maximum(1.0, np.abs(vl)), vl)
            vu = np.where(self.free & (vu < INF), vu + r * np.maximum(1.0, np.abs(vu)), vu)
        self.vl, self.vu = vl, vu
        self.lo, self.up = self.free & (vl > -INF), self.free & (vu < INF)
        self.xl, self.xu = xl, xu

    def cons(self, v):
        g = self.orc.eval_g(v[:self.n])
        c = g - self.gl
        c[self.ineq] = g[self.ineq] - v[self.n:]
        return c

    def jac(self, v):
        A = np.zeros((self.m, self.nv))
        A[self.ji, self.jj] = self.orc.eval_jac_g(v[:self.n])
        A[self.ineq, self.n + np.arange(self.ns)] = -1.0
        A[:, ~self.free] = 0.0
        return A

    def f(self, v):
        return self.orc.eval_f(v[:self.n])

    def grad(self, v):
        g = np.zeros(self.nv)
        g[:self.n] = self.orc.eval_grad_f(v[:self.n])
        g[~self.free] = 0.0
        return g

    def hess(self, v, lam, sf=1.0):
        hv = self.orc.eval_h(v[:self.n], sf, lam)
        W = np.zeros((self.nv, self.nv))
        np.add.at(W, (self.hi, self.hj), hv)
        W = W + np.tril(W, -1).T
        W[~self.free, :] = 0.0
        W[:, ~self.free] = 0.0
        return W

    def lnsum(self, v):
        return np.log(v[self.lo] - self.vl[self.lo]).sum() + np.log(self.vu[self.up] - v[self.up]).sum()

    def slacks(self, v):
        return np.where(self.lo, v - self.vl, 1.0), np.where(self.up, self.vu - v, 1.0)


def frac_to_bound(p, dl, du, zL, zU, dv, dzL, dzU, tau):
    amax = az = 1.0
    k = p.lo & (dv < 0)
    if k.any():
        amax = min(amax, np.min(-tau * dl[k] / dv[k]))
    k = p.up & (dv > 0)
    if k.any():
        amax = min(amax, np.min(tau * du[k] / dv[k]))
    k = p.lo & (dzL < 0)
    if k.any():
        az = min(az, np.min(-tau * zL[k] / dzL[k]))
    k = p.up & (dzU < 0)
    if k.any():
        az = min(az, np.min(-tau * zU[k] / dzU[k]))
    return amax, az


def solve(orc, x0, **options):
    o = dict(DEFAULTS)
    o.update(X)
    o.update(options)
    p = P(orc, o)
    n, m, nv = p.n, p.m, p.nv
    lo, up, free = p.lo, p.up, p.free
    vl, vu = p.vl, p.vu
    x = np.where(p.xl == p.xu, p.xl, _push(np.asarray(x0, float), np.where(free[:n], vl[:n], p.xl), np.where(free[:n], vu[:n], p.xu), o))
    s = _push(orc.eval_g(x)[p.ineq], vl[n:], vu[n:], o)
    v = np.concatenate([x, s])
    zL, zU = lo.astype(float), up.astype(float)
    lam = np.zeros(m)
    mu = o["mu_init"]
    filt = []
    it = 0
    dw_last = 0.0
    log = print if o["verbose"] else (lambda *a: None)
    nzb = int(lo.sum() + up.sum())
    # quasi-Newton memory
    S_, Y_ = [], []
    B0 = 1.0
    v_prev = glag_prev_parts = None
    fixed_mode = False           # adaptive mu: False = free mode
    ref_vals = []                # for the kkt-error globalisation of the free mode
    status = None
    theta_max = theta_min = None
    n_resto = 0
    mu_max = None
    n_acc = 0

    def kkt_solve(W, A, sigma, dw, dc, rhs_list):
        K = np.zeros((nv + m, nv + m))
        K[:nv, :nv] = W + np.diag(sigma + dw)
        K[nv:, :nv] = A
        K[:nv, nv:] = A.T
        K[nv:, nv:] = -np.diag(dc) if np.ndim(dc) else -dc * np.eye(m)
        fx = np.nonzero(~free)[0]
        K[fx, :] = 0.0
        K[:, fx] = 0.0
        K[fx, fx] = 1.0
        return K

    def bfgs_matrix():
        B = B0 * np.eye(nv)
        for s_, y_ in zip(S_, Y_):
            Bs = B @ s_
            sBs = s_ @ Bs
            sy = s_ @ y_
            th = 1.0 if sy >= 0.2 * sBs else 0.8 * sBs / (sBs - sy)           # Powell damping
            r = th * y_ + (1 - th) * Bs
            B = B - np.outer(Bs, Bs) / sBs + np.outer(r, r) / (s_ @ r)
        B[~free, :] = 0.0
        B[:, ~free] = 0.0
        return B

    def errors(v, lam, zL, zU, mu_):
        grad, A, c = p.grad(v), p.jac(v), p.cons(v)
        glag = grad + A.T @ lam
        dl, du = p.slacks(v)
        dinf = np.max(np.abs((glag - zL + zU)[free]))
        cinf = np.max(np.abs(c))
        prods = np.concatenate([zL[lo] * dl[lo], zU[up] * du[up]])
        sz = zL[lo].sum() + zU[up].sum()
        sd = max(o["s_max"], (np.abs(lam).sum() + sz) / max(1.0, m + nzb)) / o["s_max"]
        sc = max(o["s_max"], sz / nzb) / o["s_max"]
        comp = np.max(np.abs(prods - mu_))
        return max(dinf / sd, cinf, comp / sc), dinf, cinf, prods, sd, sc, grad, A, c, glag

    if o["ls_mult"]:
        grad, A = p.grad(v), p.jac(v)
        K = np.block([[np.eye(nv), A.T], [A, np.zeros((m, m))]])
        try:
            sol = np.linalg.lstsq(K, -np.concatenate([np.where(free, grad - zL + zU, 0.0), np.zeros(m)]), rcond=None)[0]
            l0 = sol[nv:]
            if np.max(np.abs(l0)) <= 1e3:
                lam = l0
            log("ls multipliers: max", np.max(np.abs(l0)))
        except Exception as ex:
            log("ls mult failed", ex)

    while True:
        f = p.f(v)
        err0, dinf, cinf, prods, sd, sc, grad, A, c, glag = errors(v, lam, zL, zU, 0.0)
        theta = np.abs(c).sum()
        avg = prods.mean()
        if not np.isfinite([f, dinf, cinf]).all():
            status = 5
            break
        if err0 <= o["tol"]:
            status = 0
            break
        n_acc = n_acc + 1 if err0 <= o["acceptable_tol"] else 0
        if n_acc >= o["acceptable_iter"]:
            status = 1
            break
        if it >= o["max_iter"]:
            status = 2
            break
        if it == 0:
            theta_max, theta_min = 1e4 * max(1.0, theta), 1e-4 * max(1.0, theta)
            mu_max = o["mu_max_fact"] * avg
        mu_min = min(1e-11, o["tol"] / 10) if o["mu_strategy"] != "monotone" else o["tol"] / 10.0
        dl, du = p.slacks(v)
        sigma = np.where(lo, zL / dl, 0.0) + np.where(up, zU / du, 0.0)
        # quasi-Newton update
        if o["hess"] != "exact":
            if v_prev is not None:
                s_ = v - v_prev
                y_ = (grad + A.T @ lam) - (glag_prev_parts[0] + glag_prev_parts[1].T @ lam)
                s_[~free] = 0
                y_[~free] = 0
                if s_ @ s_ > 0:
                    S_.append(s_)
                    Y_.append(y_)
                    if len(S_) > o["lbfgs_m"]:
                        S_.pop(0)
                        Y_.pop(0)
                    sy = s_ @ y_
                    if sy > 0:
                        B0 = min(1e8, max(1e-8, (y_ @ y_) / sy))
            v_prev, glag_prev_parts = v.copy(), (grad.copy(), A.copy())
            W = bfgs_matrix()
        else:
            W = p.hess(v, lam)

        # ---- barrier parameter
        mu_changed = False
        if o["mu_strategy"] == "monotone" or fixed_mode:
            for _ in range(64):
                emu = errors(v, lam, zL, zU, mu)[0]
                if not (emu <= o["kappa_eps"] * mu) or mu <= mu_min:
                    break
                if fixed_mode:
                    fixed_mode = False        # subproblem solved -> back to the free mode
                    break
                mu = max(mu_min, min(o["kappa_mu"] * mu, mu ** o["theta_mu"]))
                mu_changed = True
        # factorisation with inertia correction (the matrix does not depend on mu)
        dw = 0.0
        while True:
            K = kkt_solve(W, A, sigma, dw, o["delta_c"], None)
            if inertia_pos(K) == nv:
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
        lu = scipy.linalg.lu_factor(K)

        def direction(mu_, with_inf=True):
            rd = (glag if with_inf else 0.0) - np.where(lo, mu_ / dl, 0.0) + np.where(up, mu_ / du, 0.0)
            if not with_inf:
                rd = rd - 0.0 + (-zL + zU) * 0.0
            rhs = -np.concatenate([np.where(free, rd, 0.0), c if with_inf else np.zeros(m)])
            sol = scipy.linalg.lu_solve(lu, rhs)
            dv, dlam = np.where(free, sol[:nv], 0.0), sol[nv:]
            if with_inf:
                dzL = np.where(lo, mu_ / dl - zL - zL / dl * dv, 0.0)
                dzU = np.where(up, mu_ / du - zU + zU / du * dv, 0.0)
            else:
                dzL = np.where(lo, mu_ / dl - zL / dl * dv, 0.0)
                dzU = np.where(up, mu_ / du + zU / du * dv, 0.0)
            return dv, dlam, dzL, dzU

        if o["mu_strategy"] == "adaptive" and not fixed_mode:
            # globalisation (kkt-error variant): sufficient progress relative to the last few accepted iterates
            kerr = dinf + cinf + np.abs(prods).sum() / max(1, nzb) * 0      # 1-norm style measure
            kerr = (np.abs((glag - zL + zU)[free]).sum() + np.abs(c).sum() + prods.sum())
            if ref_vals and not any(kerr <= 0.9999 * r_ for r_ in ref_vals) and len(ref_vals) >= 4:
                fixed_mode = True
                mu = min(mu_max, max(mu_min, 0.8 * avg))
                mu_changed = True
                log("   -> fixed mode, mu = %.2e" % mu)
            else:
                ref_vals.append(kerr)
                if len(ref_vals) > 4:
                    ref_vals.pop(0)
                if o["mu_oracle"] == "loqo":
                    xi = prods.min() / avg
                    sg = 0.1 * min(0.05 * (1 - xi) / xi, 2.0) ** 3
                    mu_new = sg * avg
                else:
                    daff = direction(0.0)
                    dcen = direction(avg, with_inf=False)
                    # residual norms (2-norm squared, scaled by the number of entries)
                    pinf2 = (c @ c) / max(1, m)
                    dvec = (glag - zL + zU)[free]
                    dinf2 = (dvec @ dvec) / max(1, free.sum())

                    def q(sg):
                        dv = daff[0] + sg * dcen[0]
                        dzl = daff[2] + sg * dcen[2]
                        dzu = daff[3] + sg * dcen[3]
                        tau_q = max(o["tau_min"], 1.0 - sg * avg)
                        ap, ad = frac_to_bound(p, dl, du, zL, zU, dv, dzl, dzu, tau_q)
                        pl = (dl + ap * dv)[lo] * (zL + ad * dzl)[lo]
                        pu = (du - ap * dv)[up] * (zU + ad * dzu)[up]
                        cc = np.concatenate([pl, pu])
                        return (1 - ad) ** 2 * dinf2 + (1 - ap) ** 2 * pinf2 + (cc @ cc) / max(1, nzb)

                    smin, smax = max(o["sigma_min"], mu_min / avg), min(o["sigma_max"], mu_max / avg)
                    # golden section in log(sigma)
                    if smin >= smax:
                        sg = smin
                    else:
                        q1 = q(min(1.0, smax))
                        q1m = q(min(1.0, smax) * 0.99)
                        if q1m > q1 and smax > 1.0:
                            a_, b_ = 0.0, np.log(smax)
                        else:
                            a_, b_ = np.log(smin), np.log(min(1.0, smax))
                        gr = 0.5 * (3 - np.sqrt(5))
                        x1, x2 = a_ + gr * (b_ - a_), b_ - gr * (b_ - a_)
                        f1, f2 = q(np.exp(x1)), q(np.exp(x2))
                        for _ in range(12):
                            if f1 < f2:
                                b_, x2, f2 = x2, x1, f1
                                x1 = a_ + gr * (b_ - a_)
                                f1 = q(np.exp(x1))
                            else:
                                a_, x1, f1 = x1, x2, f2
                                x2 = b_ - gr * (b_ - a_)
                                f2 = q(np.exp(x2))
                        sg = np.exp(x1 if f1 < f2 else x2)
                        for cand in (smin, smax):
                            if q(cand) < min(f1, f2):
                                sg = cand
                    mu_new = sg * avg
                mu_new = min(mu_max, max(mu_min, mu_new))
                mu_changed = mu_new != mu
                mu = mu_new
        if mu_changed and o["reset_filter_on_mu"]:
            filt = []
        tau = max(o["tau_min"], 1.0 - mu)
        phi = f - mu * p.lnsum(v)
        dv, dlam, dzL, dzU = direction(mu)
        amax, az = frac_to_bound(p, dl, du, zL, zU, dv, dzL, dzU, tau)
        gphi = grad - np.where(lo, mu / dl, 0.0) + np.where(up, mu / du, 0.0)
        dphi = float(np.where(free, gphi, 0.0) @ dv)
        amin = o["gamma_theta"]
        if dphi < 0:
            amin = min(amin, o["gamma_phi"] * theta / (-dphi))
            if theta <= theta_min:
                amin = min(amin, o["delta"] * theta ** o["s_theta"] / (-dphi) ** o["s_phi"])
        amin *= o["gamma_alpha"]
        a, ls, armijo, accepted = amax, 0, False, False
        slack = 10.0 * np.finfo(float).eps * abs(phi)
        soc_used = 0

        def acceptable(vt, a_):
            nonlocal armijo
            with np.errstate(all="ignore"):
                ft = p.f(vt)
                tht = np.abs(p.cons(vt)).sum()
                phit = ft - mu * p.lnsum(vt)
            if not (np.isfinite([ft, tht, phit]).all() and tht <= theta_max):
                return False, tht
            if any(tht >= ft_ and phit >= fp_ for ft_, fp_ in filt):
                return False, tht
            sw = dphi < 0 and a_ * (-dphi) ** o["s_phi"] > o["delta"] * theta ** o["s_theta"]
            if theta <= theta_min and sw:
                ok = phit - phi - o["eta_phi"] * a_ * dphi <= slack
                armijo = ok
                return ok, tht
            return (tht <= (1 - o["gamma_theta"]) * theta or phit - (phi - o["gamma_phi"] * theta) <= slack), tht

        while True:
            vt = v + a * dv
            ok, tht = acceptable(vt, a)
            if ok:
                accepted = True
                break
            if ls == 0 and o["soc"] and tht >= theta:
                # second-order correction (A-5.5 .. A-5.9)
                csoc, th_old = c.copy(), theta
                asoc = a
                ct = p.cons(vt)
                for ps in range(4):
                    csoc = asoc * csoc + ct
                    rd = glag - np.where(lo, mu / dl, 0.0) + np.where(up, mu / du, 0.0)
                    sol = scipy.linalg.lu_solve(lu, -np.concatenate([np.where(free, rd, 0.0), csoc]))
                    dvs = np.where(free, sol[:nv], 0.0)
                    dzLs = np.where(lo, mu / dl - zL - zL / dl * dvs, 0.0)
                    dzUs = np.where(up, mu / du - zU + zU / du * dvs, 0.0)
                    asoc, azs = frac_to_bound(p, dl, du, zL, zU, dvs, dzLs, dzUs, tau)
                    vts = v + asoc * dvs
                    ok, ths = acceptable(vts, a)
                    if ok:
                        dv, dlam, dzL, dzU, az = dvs, sol[nv:], dzLs, dzUs, azs
                        a = asoc
                        accepted = True
                        soc_used = ps + 1
                        break
                    if ths > 0.99 * th_old:
                        break
                    th_old = ths
                    ct = p.cons(vts)
                if accepted:
                    break
            a *= 0.5
            ls += 1
            if a < amin or ls > o["max_ls"]:
                break
        log("%4d f=%.7f th=%.3e dinf=%.2e mu=%.1e a=%.2e az=%.2e dw=%.1e e0=%.2e ls=%d%s%s" % (
            it, f, theta, dinf, mu, a if accepted else 0.0, az, dw, err0, ls, " soc%d" % soc_used if soc_used else "", " F" if fixed_mode else ""))
        if not accepted:
            if o["resto"] == "l1":
                filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))
                r = resto_l1(p, o, v, zL, zU, mu, theta, filt, theta_max, log)
                if r is None:
                    status = 3
                    break
                v, zL, zU, nit = r
                it += nit
                n_resto += 1
                lam = np.zeros(m)
                # least-squares multipliers after restoration
                grad, A = p.grad(v), p.jac(v)
                Kl = np.block([[np.eye(nv), A.T], [A, -1e-12 * np.eye(m)]])
                try:
                    sol = np.linalg.solve(Kl, -np.concatenate([np.where(free, grad - zL + zU, 0.0), np.zeros(m)]))
                    if o.get("resto_ls_mult", 1) and np.max(np.abs(sol[nv:])) <= 1e3:
                        lam = sol[nv:]
                except Exception:
                    pass
                v_prev = None
                S_, Y_ = [], []
                continue
            status = 3
            break
        v = np.where(free, v + a * dv, v)
        lam = lam + a * dlam
        ks = o["kappa_sigma"]
        dl, du = p.slacks(v)
        zL = np.where(lo, np.maximum(np.minimum(zL + az * dzL, ks * mu / dl), mu / (ks * dl)), 0.0)
        zU = np.where(up, np.maximum(np.minimum(zU + az * dzU, ks * mu / du), mu / (ks * du)), 0.0)
        if not armijo:
            filt.append(((1 - o["gamma_theta"]) * theta, phi - o["gamma_phi"] * theta))
        it += 1
    return dict(x=v[:n].copy(), status=status, iterations=it, obj=p.f(v), kkt_error=err0, restorations=n_resto, lam=lam)


def resto_l1(p, o, vR, zL0, zU0, mu0, theta0, filt, theta_max, log):
    """Ipopt's restoration phase: min rho*sum(pp+nn) + zeta/2 |D_R (v - vR)|^2  s.t. c(v) - pp + nn = 0, pp, nn >= 0,
    bounds on v — solved by a (monotone-mu) interior-point iteration with pp, nn eliminated from the Newton system:
    [[zeta D_R^2 + Sigma_v + W_c, A^T], [A, -(Sigma_p^-1 + Sigma_n^-1)]].  Returns (v, zL, zU, iterations) or None."""
    n, m, nv = p.n, p.m, p.nv
    lo, up, free = p.lo, p.up, p.free
    rho = o["resto_rho"]
    mu = max(mu0, np.max(np.abs(p.cons(vR))))
    zeta = np.sqrt(mu)
    Dr2 = np.where(free, 1.0 / np.maximum(1.0, np.abs(vR)) ** 2, 0.0)
    v = vR.copy()
    c = p.cons(v)
    # initial pp, nn (Ipopt (31),(32) in the implementation paper section 3.3)
    nn = (mu - rho * c) / (2 * rho) + np.sqrt(((mu - rho * c) / (2 * rho)) ** 2 + mu * c / (2 * rho))
    pp = c + nn
    zp, zn = mu / pp, mu / nn
    zL, zU = np.minimum(rho, zL0) * lo, np.minimum(rho, zU0) * up
    zL, zU = np.where(lo, np.maximum(zL, 1e-8), 0.0), np.where(up, np.maximum(zU, 1e-8), 0.0)
    lam = np.zeros(m)
    rfilt = []
    th_max_r = None
    for itr in range(o["resto_iter"]):
        A = p.jac(v)
        c = p.cons(v)
        r_c = c - pp + nn
        dl, du = p.slacks(v)
        fR = rho * (pp.sum() + nn.sum()) + 0.5 * zeta * np.sum(Dr2 * (v - vR) ** 2)
        gradR = zeta * Dr2 * (v - vR)
        glag = gradR + A.T @ lam
        # termination of the restoration: original infeasibility reduced and acceptable to the original filter
        th_orig = np.abs(c).sum()
        if itr > 0 and th_orig <= o["kappa_resto"] * theta0 and th_orig <= theta_max:
            with np.errstate(all="ignore"):
                phi_o = p.f(v) - mu0 * p.lnsum(v)
            if not any(th_orig >= a_ and phi_o >= b_ for a_, b_ in filt):
                log("   resto done after %d its: theta %.3e -> %.3e" % (itr, theta0, th_orig))
                ks = o["kappa_sigma"]
                zL = np.where(lo, np.maximum(np.minimum(np.minimum(zL, 1e3), ks * mu0 / dl), mu0 / (ks * dl)), 0.0)
                zU = np.where(up, np.maximum(np.minimum(np.minimum(zU, 1e3), ks * mu0 / du), mu0 / (ks * du)), 0.0)
                return v, zL, zU, itr
        # optimality of the restoration problem
        dinf = max(np.max(np.abs((glag - zL + zU)[free])), np.max(np.abs(rho - lam - zp)), np.max(np.abs(rho + lam - zn)))
        comp = max(np.max(np.abs(zL * dl - mu)[lo], initial=0), np.max(np.abs(zU * du - mu)[up], initial=0), np.max(np.abs(zp * pp - mu)), np.max(np.abs(zn * nn - mu)))
        emu = max(dinf, np.max(np.abs(r_c)), comp)
        if emu <= 10 * mu:
            if mu <= 1e-9:
                log("   resto: locally infeasible? theta %.3e" % th_orig)
                return None
            mu = max(1e-10, min(0.2 * mu, mu ** 1.5))
            rfilt = []
            zeta = np.sqrt(mu)
            continue
        theta_r = np.abs(r_c).sum()
        if th_max_r is None:
            th_max_r = 1e4 * max(1.0, theta_r)
        sig_v = np.where(lo, zL / dl, 0.0) + np.where(up, zU / du, 0.0)
        sp, sn = zp / pp, zn / nn
        # eliminate pp, nn:  dp = (mu/pp - rho + lam + dlam... ) -> reduced system
        # stationarity p: rho - lam - zp = 0 ; n: rho + lam - zn = 0 ; with z eliminated: rho - lam - mu/pp + sp dp - dlam = 0
        rp = rho - lam - mu / pp
        rn = rho + lam - mu / nn
        # dp = (dlam - rp)/sp ; dn = (-dlam - rn)/sn ; A dv - dp + dn = -r_c
        dcd = 1.0 / sp + 1.0 / sn
        rhs_c = -r_c - rp / sp + rn / sn
        W = np.zeros((nv, nv))
        if o.get("resto_hess", 1) and o["hess"] == "exact":
            W = p.hess(v, lam, 0.0)
        dw = 0.0
        for _ in range(60):
            K = np.zeros((nv + m, nv + m))
            K[:nv, :nv] = W + np.diag(zeta * Dr2 + sig_v + dw)
            K[nv:, :nv] = A
            K[:nv, nv:] = A.T
            K[nv:, nv:] = -np.diag(dcd)
            fx = np.nonzero(~free)[0]
            K[fx, :] = 0.0
            K[:, fx] = 0.0
            K[fx, fx] = 1.0
            if not W.any() or inertia_pos(K) == nv:
                break
            dw = 1e-4 if dw == 0 else dw * 8
        rd = glag - np.where(lo, mu / dl, 0.0) + np.where(up, mu / du, 0.0)
        sol = np.linalg.solve(K, np.concatenate([-np.where(free, rd, 0.0), rhs_c]))
        dv, dlam = np.where(free, sol[:nv], 0.0), sol[nv:]
        dp = (dlam - rp) / sp
        dn = (-dlam - rn) / sn
        dzp = mu / pp - zp - sp * dp
        dzn = mu / nn - zn - sn * dn
        dzL = np.where(lo, mu / dl - zL - zL / dl * dv, 0.0)
        dzU = np.where(up, mu / du - zU + zU / du * dv, 0.0)
        tau = max(0.99, 1 - mu)
        amax, az = frac_to_bound(p, dl, du, zL, zU, dv, dzL, dzU, tau)
        for w_, dw_ in ((pp, dp), (nn, dn)):
            k = dw_ < 0
            if k.any():
                amax = min(amax, np.min(-tau * w_[k] / dw_[k]))
        for w_, dw_ in ((zp, dzp), (zn, dzn)):
            k = dw_ < 0
            if k.any():
                az = min(az, np.min(-tau * w_[k] / dw_[k]))
        phi = fR - mu * (p.lnsum(v) + np.log(pp).sum() + np.log(nn).sum())
        dphi = float(np.where(free, gradR - np.where(lo, mu / dl, 0.0) + np.where(up, mu / du, 0.0), 0.0) @ dv + (rho - mu / pp) @ dp + (rho - mu / nn) @ dn)
        a, okk = amax, False
        for ls in range(40):
            vt, pt, nt = v + a * dv, pp + a * dp, nn + a * dn
            with np.errstate(all="ignore"):
                ct = p.cons(vt)
                tht = np.abs(ct - pt + nt).sum()
                phit = rho * (pt.sum() + nt.sum()) + 0.5 * zeta * np.sum(Dr2 * (vt - vR) ** 2) - mu * (p.lnsum(vt) + np.log(pt).sum() + np.log(nt).sum())
            if np.isfinite([tht, phit]).all() and tht <= th_max_r and not any(tht >= a_ and phit >= b_ for a_, b_ in rfilt):
                sw = dphi < 0 and a * (-dphi) ** 2.3 > theta_r ** 1.1
                if theta_r <= 1e-4 * max(1, theta_r) and sw:
                    ok = phit - phi - 1e-8 * a * dphi <= 1e-14 * abs(phi)
                    arm = ok
                else:
                    ok = tht <= (1 - 1e-5) * theta_r or phit <= phi - 1e-8 * theta_r + 1e-14 * abs(phi)
                    arm = False
                if ok:
                    okk = True
                    break
            a *= 0.5
        log("   r%3d thO=%.3e fR=%.4e thR=%.2e mu=%.1e a=%.2e dw=%.1e emu=%.2e" % (itr, th_orig, fR, theta_r, mu, a if okk else 0, dw, emu))
        if not okk:
            return None
        if not arm:
            rfilt.append(((1 - 1e-5) * theta_r, phi - 1e-8 * theta_r))
        v = np.where(free, vt, v)
        pp, nn = pt, nt
        lam = lam + a * dlam
        ks = o["kappa_sigma"]
        dl, du = p.slacks(v)
        zL = np.where(lo, np.maximum(np.minimum(zL + az * dzL, ks * mu / dl), mu / (ks * dl)), 0.0)
        zU = np.where(up, np.maximum(np.minimum(zU + az * dzU, ks * mu / du), mu / (ks * du)), 0.0)
        zp = np.maximum(np.minimum(zp + az * dzp, ks * mu / pp), mu / (ks * pp))
        zn = np.maximum(np.minimum(zn + az * dzn, ks * mu / nn), mu / (ks * nn))
    return None
