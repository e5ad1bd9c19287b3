"""Copies the outputs of tools/collect_ipm_profiles.sh from gpurun_out/ into profiles/ (run here, after the gpurun call):
python tools/summarize_ipm_profiles.py [tag]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
shutil.copy(newest("gpurun_out/%s_ipm_sweep_stats/runc/*_kernel_stats.csv" % tag), "profiles/%s_ipm_nd_kernel_stats.csv" % tag)
shutil.copy(newest("gpurun_out/%s_ipm_delta3_stats/runc/*_kernel_stats.csv" % tag), "profiles/%s_ipm_delta3_kernel_stats.csv" % tag)
shutil.copy("gpurun_out/%s_ipm_delta3.json" % tag, "profiles/%s_ipm_delta3.json" % tag)
shutil.copy("gpurun_out/%s_ipm_sweep.json" % tag, "profiles/%s_ipm_sweep_nd1.json" % tag)
agg, launches = collections.defaultdict(float), set()
for r in csv.DictReader(open(newest("gpurun_out/%s_ipm_sweep_mfma/runc/*_counter_collection.csv" % tag))):
    if "kkt_factor" in r["Kernel_Name"]:   # the left-looking kernel and the register-resident one of level 1
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        launches.add(r["Dispatch_Id"])
fetch, l2 = 0.0, set()
for r in csv.DictReader(open(newest("gpurun_out/%s_ipm_sweep_fetch/runc/*_counter_collection.csv" % tag))):
    if "kkt_factor" in r["Kernel_Name"]:
        fetch += float(r["Counter_Value"])
        l2.add(r["Dispatch_Id"])
sts = [r for r in csv.DictReader(open("profiles/%s_ipm_nd_kernel_stats.csv" % tag)) if "kkt_factor" in r["Name"]]
st = sts[0]
tot_s = sum(float(r["TotalDurationNs"]) for r in sts) * 1e-9
sw = json.load(open("profiles/%s_ipm_sweep_nd1.json" % tag))
out = {"workload": sw["workload"], "kernel": " + ".join(r["Name"].split("(")[0] for r in sts), "launches_in_the_profiled_process": len(launches), "factorisations": len(launches) // max(1, len(sts)), "kernel_time_s": tot_s,
       "SQ_VALU_MFMA_BUSY_CYCLES": agg["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_INSTS_VALU_MFMA_MOPS_F64": agg["SQ_INSTS_VALU_MFMA_MOPS_F64"],
       "SQ_BUSY_CYCLES": agg["SQ_BUSY_CYCLES"], "SQ_WAVE_CYCLES": agg["SQ_WAVE_CYCLES"], "SQ_WAVES": agg["SQ_WAVES"],
       "matrix_pipe_busy_fraction": agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * tot_s * 2.1e9),
       "hbm_read_bytes_FETCH_SIZE_x2_x1024": fetch * 1024.0 * 2.0, "fetch_launches": len(l2),
       "hbm_read_bytes_per_factorisation": fetch * 1024.0 * 2.0 / max(1, len(l2) // max(1, len(sts))),
       "note": "separate --pmc passes of tools/bench_ipm.py 1024 0 1 (tools/collect_ipm_profiles.sh); busy fraction = MFMA busy cycles / (1024 SIMDs x kernel time x 2.1 GHz); FETCH_SIZE in KiB, x2 on gfx950 (MI355X_MICROARCH.md)"}
json.dump(out, open("profiles/%s_ipm_mfma.json" % tag, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "kernel_time_s", "matrix_pipe_busy_fraction", "hbm_read_bytes_per_factorisation")}))
print(open("profiles/%s_ipm_delta3.json" % tag).read()[:400])
print({k: sw[k] for k in ("solve_s", "solves_per_s", "factor_ms_per_launch", "substitution_ms_per_launch", "ms_per_batched_iteration")}, sw["roofline"])
