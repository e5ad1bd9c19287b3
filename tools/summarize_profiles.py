"""Copies the rocprofv3 summaries collected by tools/collect_profiles.sh from gpurun_out/ (scratch)
into profiles/ (tracked): the kernel-stats CSV, a PMC summary and profiles/pmc_traffic.json, which
bench.py reports as roofline.traffic."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    return fs[-1] if fs else None


stats = one("%s_stats/*/*kernel_stats.csv" % tag)
shutil.copy(stats, os.path.join(out, "%s_kernel_stats.csv" % tag))
summary = {"tag": tag, "command": "python3 bench.py --profile --steps 200 --warmup 20 (default workload: 64 iterates per launch, 384 resident)"}
for r in csv.DictReader(open(stats)):
    if "rpm_tile" in r["Name"]:   # rpm_tile_kernel or its role-looped layout rpm_tile_rl_kernel
        summary["dominant_kernel"] = r["Name"].split("<")[0].split("::")[-1]
        summary["dominant_kernel_calls"] = int(r["Calls"])
        summary["dominant_kernel_avg_us"] = float(r["AverageNs"]) / 1e3
pmc = {}
for sub in ("fetch", "write", "sq"):
    f = one("%s_%s/*/*counter_collection.csv" % (tag, sub))
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rpm_tile" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc[k] = sum(v) / len(v)
summary["pmc_per_launch"] = pmc
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reads exactly 1/2
    # of the bytes of a WIDE (16 B/lane) streaming read, `global_load` and `... lds` alike, and is uncalibrated for other
    # widths.  rpm_tile_pl_kernel reads its inputs with 16 B/lane direct-to-LDS loads -> the counter is doubled, as the
    # guide prescribes; the older layouts read 8 B/lane -> raw figure (uncalibrated).  WRITE_SIZE reads exact.
    wide = summary.get("dominant_kernel", "") == "rpm_tile_pl_kernel"
    fetch, write = pmc["FETCH_SIZE"] * 1024.0 * (2.0 if wide else 1.0), pmc["WRITE_SIZE"] * 1024.0
    summary["hbm_read_bytes"] = fetch
    summary["hbm_read_correction"] = "x2 (16 B/lane loads, guide §HBM)" if wide else "raw (8 B/lane loads, uncalibrated)"
    summary["hbm_write_bytes"] = write
    json.dump({"tag": tag, "hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
               "note": "FETCH_SIZE " + summary["hbm_read_correction"] + ", WRITE_SIZE exact"},
              open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
bj = os.path.join(ROOT, "gpurun_out", "%s_bench.json" % tag)
if os.path.exists(bj):
    line = [l for l in open(bj) if l.startswith("{")]
    if line:
        summary["bench"] = json.loads(line[-1])
        open(os.path.join(out, "%s_bench.json" % tag), "w").write(line[-1])
json.dump(summary, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench"}, indent=1))


# ---- round 2 extras: D.X on the matrix cores inside the pipelined kernel, configs 4 and 5 -------------------------------
def kernel_avg(pattern):
    f = one(pattern)
    if not f:
        return None, None
    for r in csv.DictReader(open(f)):
        if "rpm_tile" in r["Name"]:
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None, None


def counters(pattern):
    f = one(pattern)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if "rpm_tile" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


us, calls = kernel_avg("%s_mfma_stats/*/*kernel_stats.csv" % tag)
if us:
    shutil.copy(one("%s_mfma_stats/*/*kernel_stats.csv" % tag), os.path.join(out, "%s_mfma_kernel_stats.csv" % tag))
    c = counters("%s_mfma/*/*counter_collection.csv" % tag)
    n_mfma = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / 4.0      # one v_mfma_f64_16x16x4 = 2048 flop = 4 x 512-flop MOPS
    rec = {"tag": tag, "command": "python3 bench.py --profile --steps 200 --warmup 20 --dx-mode 1 (64 iterates per launch)",
           "kernel": "rpm_tile_pl_kernel<..., DXM = true>: D.X of every tile by v_mfma_f64_16x16x4_f64 on the DMA waves",
           "kernel_avg_us": us, "kernel_avg_us_scalar_dx": summary.get("dominant_kernel_avg_us"), "calls": calls, "pmc_per_launch": c,
           "mfma_instructions_per_launch": n_mfma, "mfma_flop_per_launch": n_mfma * 2048.0,
           "mfma_tflops": n_mfma * 2048.0 / (us * 1e-6) / 1e12,
           "fp64_matrix_peak_tflops": 78.6,
           "useful_share": "7 of 16 MFMA columns (nx = 7) and 17 of 20 k-columns per 16-node interval",
           "matrix_pipe_busy_fraction": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * us * 1e-6 * 2.1e9),
           "note": "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x ~2.1 GHz)"}
    json.dump(rec, open(os.path.join(out, "%s_mfma.json" % tag), "w"), indent=1)
    print(json.dumps(rec, indent=1))
for name, label, per_pair in (("quadrotor", "config5_quadrotor_1024", None), ("hypersensitive", "config4_hypersensitive_hp_256", None)):
    us, calls = kernel_avg("%s_%s_stats/*/*kernel_stats.csv" % (tag, name))
    if not us:
        continue
    shutil.copy(one("%s_%s_stats/*/*kernel_stats.csv" % (tag, name)), os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, label)))
    c = counters("%s_%s_sq/*/*counter_collection.csv" % (tag, name))
    c.update(counters("%s_%s_write/*/*counter_collection.csv" % (tag, name)))
    rec = {"tag": tag, "workload": label, "command": "python3 tools/profile_workload.py %s %s 100" % (name, label.split("_")[-1]),
           "kernel_avg_us": us, "calls": calls, "pmc_per_launch": c,
           "hbm_write_bytes": c.get("WRITE_SIZE", 0.0) * 1024.0}
    json.dump(rec, open(os.path.join(out, "%s_%s.json" % (tag, label)), "w"), indent=1)
    print(json.dumps(rec, indent=1))
