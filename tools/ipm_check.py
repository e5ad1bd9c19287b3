"""Row f-2 on the GPU box: (1) the band + border LDL^T against numpy on random quasi-definite matrices in the solver's
own layout, (2) batched device-resident interior-point solves of small problems with known optima, (3) a quadrotor
sweep.  Run:  python tools/ipm_check.py [instances]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from lpopc_amd.problem import Options


def exact():
    o = Options()
    o.SetStringValue("hessian-approximation", "exact")
    return o


def random_kkt(ipm, B, seed):
    """Random symmetric quasi-definite matrices with the solver's sparsity envelope -> (storage, dense)."""
    info = ipm.info()
    nt, nb_, b, nbd, cs = info["kkt_order"], info["band_order"], info["half_bandwidth"], info["border"], info["storage_doubles"] // info["kkt_order"]
    pos = ipm.permutation()
    nv = ipm._e.n + info["n_slacks"]
    sign = np.ones(nt)
    sign[pos[nv:]] = -1.0
    rng = np.random.RandomState(seed)
    dense = np.zeros((B, nt, nt))
    store = np.zeros((B, nt * cs))
    for bi in range(B):
        A = np.zeros((nt, nt))
        for j in range(nt):
            rows = list(range(j, min(j + b, nb_ - 1) + 1)) if j < nb_ else []
            rows += [i for i in range(nb_, nt) if i >= j]
            for i in rows:
                if i == j:
                    continue
                if sign[i] != sign[j] or rng.rand() < 0.3:
                    A[i, j] = rng.uniform(-1, 1) * (rng.rand() < 0.5)
        A = A + A.T
        # diagonal: dominant inside the (+) and (-) blocks so that they are definite
        for i in range(nt):
            same = np.abs(A[i, sign == sign[i]]).sum()
            A[i, i] = sign[i] * (same + rng.uniform(0.5, 2.0))
        dense[bi] = A
        for j in range(nt):
            for i in range(j, nt):
                if i < nb_:
                    if i - j <= b:
                        store[bi, j * cs + i - j] = A[i, j]
                else:
                    store[bi, j * cs + b + 1 + i - nb_] = A[i, j]
    return store, dense, sign


def check_linear_algebra():
    for name, prob, B in (("brachistochrone 2x6", problems.brachistochrone(2, 6), 3), ("quadrotor 3x4", problems.quadrotor(3, 4), 2),
                          ("launch 2x5", problems.launch(2, 5), 2)):
        eng = NLPEngine(prob, exact(), n_instances=B, device=0)
        eng.set_option("ipm_nested", 0)      # this check fills the band + border storage itself
        ipm = BatchedIPM(eng)
        info = ipm.info()
        store, dense, sign = random_kkt(ipm, B, 7)
        rhs = np.random.RandomState(3).uniform(-1, 1, size=(B, info["kkt_order"]))
        sol, npos, nneg = ipm.debug_solve(store, rhs)
        worst = 0.0
        for bi in range(B):
            ref = np.linalg.solve(dense[bi], rhs[bi])
            worst = max(worst, np.max(np.abs(sol[bi] - ref)) / np.max(np.abs(ref)))
        print("LDL^T %-20s %s  rel err %.2e  inertia (+%d, -%d) expected (+%d, -%d)" % (
            name, info, worst, npos[0], nneg[0], int((sign > 0).sum()), int((sign < 0).sum())), flush=True)
        assert worst < 1e-9 and npos[0] == (sign > 0).sum() and nneg[0] == (sign < 0).sum()
        ipm.close()
        eng.close()


def solve_one(name, prob, B, expect=None, perturb=0.0):
    eng = NLPEngine(prob, exact(), n_instances=B, device=0)
    ipm = BatchedIPM(eng)
    x0 = np.tile(NLPEngine(prob, exact(), device=0).get_starting_point(), (B, 1))
    if perturb:
        rng = np.random.RandomState(1)
        x0 = x0 * (1 + perturb * rng.uniform(-1, 1, size=x0.shape))
    t0 = time.perf_counter()
    r = ipm.solve(x0)
    dt = time.perf_counter() - t0
    print("%-28s B=%d  %s  obj %.9g..%.9g  iters %d..%d  status %s  kkt %.1e  %.3f s  %s" % (
        name, B, ipm.info(), r["obj"].min(), r["obj"].max(), r["iterations"].min(), r["iterations"].max(),
        sorted(set(r["status"].tolist())), r["kkt_error"].max(), dt, ipm.stats()), flush=True)
    if expect is not None:
        assert abs(r["obj"][0] - expect[0]) < expect[1], (r["obj"][0], expect)
    ipm.close()
    eng.close()
    return r


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    check_linear_algebra()
    solve_one("bryson-denham 2x8", problems.bryson_denham(2, 8), 2, (4.0 / (9.0 * (1.0 / 9.0)) if False else None))
    solve_one("brachistochrone 2x10", problems.brachistochrone(2, 10), 2)
    solve_one("quadrotor 2x4", problems.quadrotor(2, 4), 4, perturb=1e-2)
    solve_one("quadrotor 8x8 sweep", problems.quadrotor(8, 8), B, perturb=1e-2)
