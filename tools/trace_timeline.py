"""Per-workgroup timeline of the role-looped tile kernel (diagnostic build, perf exploration only).
Run on the GPU box:  python tools/trace_timeline.py [instances]
Needs lpopc_amd/csrc/librpm_hip_diag.so (make -C lpopc_amd/csrc librpm_hip_diag.so)."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["RPM_HIP_LIB"] = os.environ.get("RPM_TRACE_LIB", os.path.join(root, "lpopc_amd", "csrc", "librpm_hip_diag.so"))
os.environ["RPM_DIAG_TRACE"] = out = os.path.join(root, "gpurun_out", "trace.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
import numpy as np
import torch

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prob = problems.config("launch")
eng = NLPEngine(prob, n_instances=B, device=0)
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
ntiles = eng.get_option("n_tiles")
eng.close()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 8)
t = t[t[:, 0] > 0]
ts = t[:, :7].astype(np.float64) * 0.01   # 100 MHz -> us
hw = t[:, 7]
tile = ts[:, 6] > 0   # tile workgroups reach point 6
ts = ts[tile]
t00 = ts[:, 0].min()
print("workgroups traced: %d (tiles per instance %d), kernel span %.2f us" % (len(ts), ntiles, ts[:, 6].max() - t00))
names = ["start->loads issued", "loads issued->barrier", "barrier->base dynamics published", "role loop (J stores issued)",
         "const stores issued", "drain (waitcnt 0)"]
for i, nm in enumerate(names):
    d = ts[:, i + 1] - ts[:, i]
    print("  %-34s mean %6.2f  p10 %6.2f  p90 %6.2f us" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
st = ts[:, 0] - t00
print("  start time: p0 %.2f p25 %.2f p50 %.2f p75 %.2f p100 %.2f us" % tuple(np.percentile(st, [0, 25, 50, 75, 100])))
en = ts[:, 6] - t00
print("  end time:   p0 %.2f p25 %.2f p50 %.2f p75 %.2f p100 %.2f us" % tuple(np.percentile(en, [0, 25, 50, 75, 100])))
life = ts[:, 6] - ts[:, 0]
print("  workgroup lifetime mean %.2f us" % life.mean())
xcc = (hw[tile] >> np.uint64(32)) & np.uint64(0xF)
print("  workgroups per XCC:", np.bincount(xcc.astype(int), minlength=8))
# concurrency histogram: number of live tile workgroups over time
grid = np.linspace(0, en.max(), 41)
live = [(np.sum((st <= g) & (en > g))) for g in grid]
print("  live workgroups at 40 time points:", live)
