"""Generates tests/golden/*.npz.

PROVENANCE: these vectors are produced by THIS repo's CPU oracle (oracle/liborpm.so), not by the
reference: lpopc ships no golden vectors and cannot be built in this image.  They pin
the oracle against accidental change (regression fixtures) and let the GPU tests compare against
committed numbers; they do not pin the oracle to the reference ("parity unpinned").
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lpopc_amd import problems  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

CASES = {
    "brachistochrone_1x10": lambda: problems.brachistochrone(1, 10),
    "bryson_denham_1x20": lambda: problems.bryson_denham(),
    "launch_1x20": lambda: problems.launch(),
    "launch_4x8": lambda: problems.launch(4, 8),
    "hypersensitive_6x5": lambda: problems.hypersensitive([-1, -0.8, -0.3, 0.2, 0.7, 0.9, 1], [5] * 6),
    "climb_4x6": lambda: problems.min_time_climb(4, 6),
    "quadrotor_2x5": lambda: problems.quadrotor(2, 5),
}


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name, make in CASES.items():
        o = Oracle(make())
        xl, xu, gl, gu = o.bounds()
        x = problems.seeded_iterate(o.starting_point(), xl, xu, 17)
        i, j = o.jac_structure()
        np.savez_compressed(os.path.join(here, name + ".npz"), x=x, g=o.eval_g(x), jac_values=o.eval_jac_g(x),
                            jac_i=i, jac_j=j, f=np.array([o.eval_f(x)]), grad_f=o.eval_grad_f(x),
                            x_l=xl, x_u=xu, g_l=gl, g_u=gu, x_start=o.starting_point())
        print(name, o.n, o.m, o.nnz_jac)


if __name__ == "__main__":
    main()
