"""Host-pointer (Ipopt-facing) path: PCIe-inclusive pairs/s of every delivery variant, on the small plumbing problem
(fixed cost) and on the metric problem.  Run on the GPU box; prints JSON lines."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpopc_amd import problems                      # noqa: E402
from lpopc_amd.engine import NLPEngine              # noqa: E402
from lpopc_amd.hostbench import time_host_path, time_ipopt_iteration     # noqa: E402

for name, mk, B in (("brachistochrone 1x10", lambda: problems.brachistochrone(1, 10), 1),
                    ("launch 4 x 64 x 16 (metric)", lambda: problems.config("launch"), 1),
                    ("launch 4 x 64 x 16 (metric), 8 iterates per call", lambda: problems.config("launch"), 8),
                    ("quadrotor 8x8, 1024 instances per call", lambda: problems.quadrotor(8, 8), 1024)):
    prob = mk()
    probe = NLPEngine(prob, device=0)
    xl, xu, _, _ = probe.get_bounds_info()
    x0 = probe.get_starting_point()
    probe.close()
    import numpy as np
    xs = [np.concatenate([problems.seeded_iterate(x0, xl, xu, 7 + 100 * i + b) for b in range(B)]) for i in range(4)]
    res = time_host_path(lambda: NLPEngine(prob, n_instances=B, device=0), xs, seconds=0.5)
    rec = {"problem": name, "instances_per_call": B, "variants": res}
    if B == 1:
        rec["ms_per_ipopt_iter_host_pointer"] = time_ipopt_iteration(lambda: NLPEngine(prob, device=0), xs)
        rec["ms_per_ipopt_iter_host_pointer_delta"] = time_ipopt_iteration(lambda: NLPEngine(prob, device=0), xs,
                                                                              options={"pin_host": 1, "delta_values": 1})
    print(json.dumps(rec), flush=True)
