#!/bin/bash
# Runs bench.py's main section (64 and 16 iterates per launch) against each experiment build named on the command line
# (lpopc_amd/csrc/librpm_exp_<name>.so) and the default library, interleaved twice.  Perf exploration only.
mkdir -p gpurun_out
for rep in 1 2; do
for v in base "$@"; do
  L=$PWD/lpopc_amd/csrc/librpm_hip.so; [ "$v" != base ] && L=$PWD/lpopc_amd/csrc/librpm_exp_$v.so
  for b in 64 16; do
    RPM_HIP_LIB=$L python bench.py --steps 300 --warmup 30 --only-main --batch $b --iterates 256 > gpurun_out/exp_${v}_${b}_${rep}.json 2>gpurun_out/exp_${v}_${b}_${rep}.err || exit 1
  done
done
done
python - "$@" <<'PY'
import json, sys
for v in ["base"] + sys.argv[1:]:
    for b in (64, 16):
        for rep in (1, 2):
            d = json.loads(open("gpurun_out/exp_%s_%d_%d.json" % (v, b, rep)).read().strip().splitlines()[-1])
            print(v, b, rep, "%.0f pairs/s  %.4f ms/step  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
