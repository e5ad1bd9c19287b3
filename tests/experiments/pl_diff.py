"""Experiment: where the pipelined kernel's output differs from the role-looped kernel's (debugging aid).  GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prob = problems.quadrotor(8, 8)
ref = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
pl = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
ref.set_option("pipeline", 0)
pl.set_option("pipeline", 1)
xl, xu, _, _ = ref.get_bounds_info()
x0 = NLPEngine(prob, device=0).get_starting_point()
xs = np.stack([problems.seeded_iterate(x0, xl, xu, 90 + i, "perturb") for i in range(B)])
dx = torch.from_numpy(xs).cuda()
out = []
for eng in (ref, pl):
    dg = torch.full((B, ref.m), np.nan, dtype=torch.float64, device="cuda")
    dv = torch.full((B, ref.nnz_jac), np.nan, dtype=torch.float64, device="cuda")
    eng.eval_pair_dev(dx, dg, dv)
    torch.cuda.synchronize()
    out.append((dg.cpu().numpy(), dv.cpu().numpy()))
print("pipeline", pl.get_option("pipeline_active"), "g equal", np.array_equal(out[0][0], out[1][0]))
a, b = out[0][1], out[1][1]
bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
print("values: %d of %d differ; NaN left: %d" % (len(bad), a.size, int(np.isnan(b).sum())))
if len(bad):
    inst = np.unique(bad[:, 0])
    print("instances", inst[:20], "...")
    pos = bad[bad[:, 0] == inst[0]][:, 1]
    print("first instance: %d entries, positions min %d max %d" % (len(pos), pos.min(), pos.max()))
    blk = np.unique(pos // 64)
    print("64-blocks touched:", len(blk), blk[:40])
    for p in pos[:10]:
        print(p, a[inst[0], p], b[inst[0], p])
ga, gb = out[0][0], out[1][0]
gbad = np.argwhere(ga != gb)
print("g: %d differ" % len(gbad))
if len(gbad):
    i0 = gbad[0, 0]
    rows = gbad[gbad[:, 0] == i0][:, 1]
    print("instance", i0, "rows", rows[:30], "count", len(rows), "of m", ref.m)
    for r in rows[:5]:
        print(r, ga[i0, r], gb[i0, r])
    print("instances with bad g:", np.unique(gbad[:, 0])[:10], len(np.unique(gbad[:, 0])))
if len(bad):
    i0 = inst[0]
    seg = slice(1024, 1024 + 128)
    for other in (i0 - 256, i0 + 256, i0 + 512, i0 - 1, i0 + 1):
        if 0 <= other < B:
            print("pl[%d] t-columns == ref[%d]:" % (i0, other), np.array_equal(b[i0, seg], a[other, seg]), " == pl[%d]:" % other, np.array_equal(b[i0, seg], b[other, seg]))
    print("instances bad:", len(inst), "min", inst.min(), "max", inst.max(), "bad below 256:", int((inst < 256).sum()))
