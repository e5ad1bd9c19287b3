import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
from run_reference_examples import run
from lpopc_amd import problems
K, Nk = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 0)
prob = problems.launch(K, Nk) if K else problems.launch()
run("launch hp-Liu %dx%d" % (K, Nk), prob, {"hessian-approximation": "exact", "mesh-refine-methods": "hp-Liu", "max-grid-num": 8}, -7529.712 / 301454.0)
for i in range(4):
    print("phase", i, prob.GetPhase(i).GetNodesPerInterval())
