"""End-to-end demo on the GPU box: LpopcApplication mirror (solve -> extract -> estimate -> refine -> ...) on
Bryson-Denham with both refinement methods.  python tools/run_app_demo.py"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
from lpopc_amd import problems
from lpopc_amd.application import LpopcApplication

for method, tol in (("ph", 1e-7), ("hp-Liu", 1e-6)):
    prob = problems.bryson_denham(2, 8)
    app = LpopcApplication(1)
    app.SetOptimalControlProblem(prob)
    app.Options().SetStringValue("mesh-refine-methods", method)
    app.Options().SetNumericValue("desired-relative-error", tol)
    app.Options().SetIntegerValue("max-grid-num", 5)
    app.Options().SetIntegerValue("Nmax", 12)
    try:
        app.SolveOptimalProblem()
    except Exception as e:
        print("stopped:", e)
    print(method, tol, "objective %.8f" % app.objective, "grids", app.meshrefiner_.CurrentGrid(), "nodes", prob.GetPhase(0).GetNodesPerInterval())
