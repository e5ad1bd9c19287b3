import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
from run_reference_examples import run
from lpopc_amd import problems
run("launch hp-Liu", problems.launch(), {"hessian-approximation": "exact", "mesh-refine-methods": "hp-Liu", "max-grid-num": 8}, -7529.712 / 301454.0)
