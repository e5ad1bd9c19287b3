import sys, warnings
sys.path.insert(0, "/root/repo")
warnings.filterwarnings("ignore")
from lpopc_amd import problems
from lpopc_amd.application import LpopcApplication
for tol in (1e-5, 1e-6, 1e-7):
    prob = problems.bryson_denham(2, 8)
    app = LpopcApplication(1)
    app.SetOptimalControlProblem(prob)
    app.Options().SetNumericValue("desired-relative-error", tol)
    app.Options().SetIntegerValue("max-grid-num", 8)
    app.Options().SetIntegerValue("Nmax", 12)
    try:
        app.SolveOptimalProblem()
    except Exception as e:
        print("EXC", e)
    print(tol, app.objective, app.meshrefiner_.CurrentGrid(), prob.GetPhase(0).GetNodesPerInterval())
