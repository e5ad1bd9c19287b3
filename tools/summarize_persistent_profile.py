"""profiles/r03_persistent.json + kernel stats from what tools/profile_persistent.sh left under gpurun_out/ (run here afterwards)."""
import collections
import csv
import glob
import json
import os
import shutil

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
stats = newest("gpurun_out/r03pv_stats/*/*_kernel_stats.csv")
shutil.copy(stats, "profiles/r03_persistent_kernel_stats.csv")
row = [r for r in csv.DictReader(open(stats)) if "rpm_tile_pl_kernel" in r["Name"]][0]
pmc = collections.defaultdict(float)
launches = set()
for tag in ("r03pv_sq", "r03pv_sq2"):
    for r in csv.DictReader(open(newest("gpurun_out/%s/*/*_counter_collection.csv" % tag))):
        if "rpm_tile_pl_kernel" in r["Kernel_Name"]:
            pmc[r["Counter_Name"]] += float(r["Counter_Value"])
            if tag == "r03pv_sq":
                launches.add(r["Dispatch_Id"])
n = max(1, len(launches))
avg_us = float(row["AverageNs"]) * 1e-3
Bp = 4624232
out = {"tag": "r03", "command": "python3 bench.py --profile --steps 200 --warmup 20 --persistent (64 iterates per launch, option persistent_values; tools/profile_persistent.sh)",
       "kernel": "rpm_tile_pl_kernel (staged dynamics, STG variant), constant Doffdiag block skipped after the first fill of each resident array",
       "kernel_avg_us": avg_us, "calls": int(row["Calls"]), "algorithmic_bytes_per_pair_Bprime": Bp,
       "hbm_frac_Bprime": Bp * 64 / (avg_us * 1e-6) / 8e12, "pmc_per_launch": {k: v / n for k, v in pmc.items()}}
if "SQ_ACTIVE_INST_VALU" in pmc:
    out["valu_busy_fraction"] = pmc["SQ_ACTIVE_INST_VALU"] / n * 4 / (1024 * avg_us * 1e-6 * 2.1e9)
out["note"] = "SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles (guide); VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel time x ~2.1 GHz)"
json.dump(out, open("profiles/r03_persistent.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel_avg_us", "hbm_frac_Bprime", "valu_busy_fraction") if k in out}))
print({k: round(v) for k, v in out["pmc_per_launch"].items()})
