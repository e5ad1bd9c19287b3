"""The reference's own example programs, with the options their main() sets, end to end on the device:
example/bryson-denham/BrysonDenham.cpp:77-78 (hessian-approximation=exact, max-grid-num=20) and
example/hypersensitive/HyperSensitive.cpp:53-57 (exact, first-derive=analytic, mesh-refine-methods=hp-Liu,
max-grid-num=20, tf = 5000), and example/launch/Launch.cpp (no options; exact Hessian here).  Every NLP is solved by the device interior-point solver (rpm_ipm_*), extraction, error
estimate and refinement run on the GPU too.  python tools/run_reference_examples.py"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import numpy as np

from lpopc_amd import problems
from lpopc_amd.application import LpopcApplication


def run(name, prob, opts, expect):
    app = LpopcApplication(1)
    app.SetOptimalControlProblem(prob)
    for k, v in opts.items():
        if isinstance(v, str):
            app.Options().SetStringValue(k, v)
        elif isinstance(v, int):
            app.Options().SetIntegerValue(k, v)
        else:
            app.Options().SetNumericValue(k, v)
    t0 = time.perf_counter()
    try:
        app.SolveOptimalProblem()
        end = "Optimal Problem Solved"
    except Exception as e:
        end = "stopped: %s" % e
    ph = prob.GetPhase(0)
    print("%s: objective %.9f (expected %.6f), grids %d, intervals %d, nodes %d, %.2f s, %s" % (
        name, app.objective, expect, app.meshrefiner_.CurrentGrid(), len(ph.GetNodesPerInterval()),
        int(np.sum(ph.GetNodesPerInterval())), time.perf_counter() - t0, end), flush=True)
    return app


if __name__ == "__main__":
    from scipy.integrate import quad
    run("bryson-denham", problems.bryson_denham(), {"hessian-approximation": "exact", "max-grid-num": 20}, 4.0)
    V = quad(lambda x: -x ** 3 + np.sqrt(x ** 6 + x ** 2), 0, 1.5)[0]
    W = quad(lambda x: x ** 3 + np.sqrt(x ** 6 + x ** 2), 0, 1.0)[0]
    run("hypersensitive", problems.hypersensitive(), {"hessian-approximation": "exact", "first-derive": "analytic",
                                                      "mesh-refine-methods": "hp-Liu", "max-grid-num": 20}, V + W)
    # example/launch/Launch.cpp sets no option at all (the reference then runs Ipopt's limited-memory Hessian; the device solver
    # needs eval_h): its mesh, its guess, default refinement — the published optimum is 7529.712 kg = -0.0249779 in its mass unit
    run("launch (Delta-III)", problems.launch(), {"hessian-approximation": "exact"}, -7529.712 / 301454.0)
    # (ph refinement hands the 20-node intervals back unchanged — the truncated degree increment, LpPhMeshRefineAlg.cpp:81 — until the
    # grid limit stops the run, as the reference would.)  With hp-Liu refinement the loop ends by itself:
    run("launch (Delta-III), hp-Liu", problems.launch(), {"hessian-approximation": "exact", "mesh-refine-methods": "hp-Liu", "max-grid-num": 8},
        -7529.712 / 301454.0)
