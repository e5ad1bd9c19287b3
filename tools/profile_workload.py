"""One device-resident workload of bench.py alone (for rocprofv3): python3 tools/profile_workload.py <config> <B> [steps]
config: launch | hypersensitive | quadrotor | climb | brachistochrone.  Eager launches (rocprofv3 sees every dispatch),
the same engine options as bench.py (instance_align 16, automatic layout)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse

import bench
from lpopc_amd import problems

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("B", type=int)
ap.add_argument("steps", type=int, nargs="?", default=100)
ap.add_argument("--dx-mode", type=int, default=0)
ap.add_argument("--graph", action="store_true")
a = ap.parse_args()


class Args:
    gpus = 1
    tile_nodes = 0
    dx_mode = a.dx_mode


ctx = bench.Ctx(Args)
bench.MIN_TIMED_S = 0.0
prob = problems.config(a.config)
mode = "uniform" if a.config.startswith("hypersensitive") else "perturb"
res = bench.device_workload(ctx, Args, prob, a.B, 4 * a.B if a.B > 16 else 256, a.steps, 10, False, 3, mode=mode,
                            use_graph=a.graph, dx_mode=a.dx_mode)
res["config"] = a.config
print(json.dumps(res))
