"""Per kernel AND grid size: launches and mean duration in the trace tools/ipm_sweep_trace.sh wrote (the three substitution passes
of an iteration are one kernel with different grids).  python tools/ipm_sweep_trace_summary.py [tag]"""
import collections
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sorted(glob.glob(os.path.join(root, "gpurun_out/%s_ipm_trace/*/*_kernel_trace.csv" % tag)), key=os.path.getmtime)[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", ""))
    a = agg[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(a[1] for a in agg.values())
print("total kernel time %.1f ms" % (tot * 1e-3))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-60s grid %9s wg %4s  calls %5d  mean %9.1f us  %5.1f %%" % (k[0], k[1], k[2], a[0], a[1] / a[0], 100 * a[1] / tot))
