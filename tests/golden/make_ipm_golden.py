"""Regression fixtures of row f-2: solutions and traces of the interior-point restatement (oracle/ipm_oracle.py) on
small problems -> tests/golden/ipm/*.npz.  These are outputs of this repository's own CPU restatement (the reference
holds no solver traces; Ipopt is not in its tree): they pin behaviour across rounds, not parity with the reference.
Run from the repo root:  python tests/golden/make_ipm_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from lpopc_amd import problems  # noqa: E402
from lpopc_amd.problem import Options  # noqa: E402

CASES = {
    "bryson_denham_2x8": (lambda: problems.bryson_denham(2, 8), dict(tol=1e-8)),
    # (the monotone barrier rule: with the default, adaptive, this start needs no restoration phase)
    "bryson_denham_default_restoration": (lambda: problems.bryson_denham(), dict(tol=1e-6, mu_strategy="monotone")),
    "hypersensitive_6x10": (lambda: problems.hypersensitive(np.linspace(-1, 1, 7).tolist(), [10] * 6, tf=30.0), dict(tol=1e-8)),
    "quadrotor_2x4": (lambda: problems.quadrotor(2, 4), dict(tol=1e-8)),
    # the metric problem on a small mesh, from lpopc's default guess (example/launch/Launch.cpp:200-457): needs the bound
    # relaxation and, depending on rounding, the restoration phase; optimum = 7529.71 kg of final mass (objective -m_f / m_scale)
    "launch_2x6": (lambda: problems.launch(2, 6), dict(tol=1e-8)),
    "launch_4x8": (lambda: problems.launch(4, 8), dict(tol=1e-8)),       # ~2 minutes of dense numpy
}


def exact():
    o = Options()
    o.SetStringValue("hessian-approximation", "exact")
    return o


if __name__ == "__main__":
    from oracle import ipm_oracle
    from oracle.oracle import Oracle
    os.makedirs(os.path.join(HERE, "ipm"), exist_ok=True)
    for name, (make, opts) in CASES.items():
        o = Oracle(make(), exact())
        x0 = o.starting_point()
        r = ipm_oracle.solve(o, x0, **opts)
        tr = np.array([[t["f"], t["theta"], t["mu"], t["alpha"], t["alpha_z"], t["delta_w"], t["err0"], t["ls"]] for t in r["trace"]])
        np.savez_compressed(os.path.join(HERE, "ipm", name + ".npz"), x0=x0, x=r["x"], lam=r["lambda"], obj=np.array([r["obj"]]),
                            status=np.array([r["status"]]), iterations=np.array([r["iterations"]]),
                            restorations=np.array([r["restorations"]]), trace=tr)
        print(name, r["status"], r["iterations"], r["obj"], r["restorations"])
