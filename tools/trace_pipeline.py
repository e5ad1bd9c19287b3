"""Timeline of rpm_tile_pl_kernel's first two tiles per workgroup (diagnostic build, perf exploration only).
Run on the GPU box:  python tools/trace_pipeline.py [instances [launch|quadrotor]]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["RPM_HIP_LIB"] = os.path.join(root, "lpopc_amd", "csrc", "librpm_hip_diag.so")
os.environ["RPM_DIAG_TRACE"] = out = os.path.join(root, "gpurun_out", "trace_pl.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
import numpy as np
import torch

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prob = problems.quadrotor(8, 8) if len(sys.argv) > 2 and sys.argv[2] == "quadrotor" else problems.config("launch")
eng = NLPEngine(prob, n_instances=B, device=0)
eng.set_option("pipeline", 1)
xl, xu, _, _ = eng.get_bounds_info()
x0 = eng.get_starting_point()
R = 4
xs = np.stack([problems.seeded_iterate(x0, xl, xu, s) for s in range(R * B)]).reshape(R, B * eng.n)
d_x = torch.from_numpy(xs).cuda()
d_g = torch.empty((R, B * eng.m), dtype=torch.float64, device="cuda")
d_v = torch.empty((R, B * eng.nnz_jac), dtype=torch.float64, device="cuda")
for k in range(12):
    eng.eval_pair_dev(d_x[k % R], d_g[k % R], d_v[k % R])
torch.cuda.synchronize()
assert eng.get_option("pipeline_active") == 1
eng.close()
t = np.fromfile(out, dtype=np.uint64)
G = min(512, t.size // 64)
G = int(os.environ.get("TRACE_HALVES", G))
t = t[:G * 64].reshape(G, 2, 32).astype(np.float64) * 0.01
t00 = t[:, 0, 31].min()


def stat(name, a):
    print("  %-52s mean %6.2f  p10 %6.2f  p90 %6.2f" % (name, a.mean(), np.percentile(a, 10), np.percentile(a, 90)))


stat("kernel start (rel. first)", t[:, 0, 31] - t00)
stat("prologue: first record read (start -> runs known)", t[:, 0, 19] - t[:, 0, 31])
stat("prologue: first tile's loads issued", t[:, 0, 20] - t[:, 0, 19])
stat("prologue: loads issued -> A passed", t[:, 0, 0] - t[:, 0, 20])
for j in (0, 1):
    print("tile", j, " A passed at %.2f" % (t[:, j, 0] - t00).mean())
    for wv in range(4):
        b = wv * 4
        stat("wave %d: first pass (A -> arrive F)" % wv, t[:, j, b + 1] - t[:, j, b])
        stat("wave %d: wait at F" % wv, t[:, j, b + 2] - t[:, j, b + 1])
        stat("wave %d: later passes (F -> end)" % wv, t[:, j, b + 3] - t[:, j, b + 2])
    stat("dma: A -> const stores issued", t[:, j, 17] - t[:, j, 16])
    stat("dma: next tile's loads issued", t[:, j, 18] - t[:, j, 17])
    print("  tile end (slowest wave, abs): mean %.2f" % (t[:, j, [3, 7, 11, 15]].max(axis=1) - t00).mean())
print("kernel end: %.2f" % (t[:, 1, [3, 7, 11, 15]].max() - t00))
print("second pass of wave 1 (role 5: a velocity perturbation, with D.X):")
for j in (0, 1):
    stat("tile %d: LDS reads + D.X" % j, t[:, j, 25] - t[:, j, 24])
    stat("tile %d: perturb + dynamics" % j, t[:, j, 26] - t[:, j, 25])
    stat("tile %d: J, transform, g + J stores" % j, t[:, j, 27] - t[:, j, 26])
