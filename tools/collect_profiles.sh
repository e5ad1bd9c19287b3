#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ for the default bench command; run on the GPU box:
#   gpurun -- 'bash tools/collect_profiles.sh r02'
# then `python tools/summarize_profiles.py r02` (here) copies the summaries into profiles/.
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; gpurun refuses --pmc beside trace domains
# other than --kernel-trace).
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="python3 bench.py --profile --steps 200 --warmup 20"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $A > gpurun_out/${TAG}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- $A --no-graph > gpurun_out/${TAG}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- $A --no-graph > gpurun_out/${TAG}_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/${TAG}_sq -- $A --no-graph > gpurun_out/${TAG}_sq.log 2>&1
# D.X on the matrix cores inside the kernel bench.py runs (dx_mode 1): kernel time and MFMA busy cycles at B = 64
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_mfma_stats -- $A --dx-mode 1 > gpurun_out/${TAG}_mfma_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${TAG}_mfma -- $A --no-graph --dx-mode 1 > gpurun_out/${TAG}_mfma.log 2>&1 || true
# config 5 (the 1024-instance quadrotor sweep) and config 4 (hp mesh, 256 iterates): kernel stats + traffic
for W in "quadrotor 1024" "hypersensitive 256"; do
  N=$(echo $W | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_${N}_stats -- python3 tools/profile_workload.py $W 100 > gpurun_out/${TAG}_${N}_stats.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_${N}_write -- python3 tools/profile_workload.py $W 100 > gpurun_out/${TAG}_${N}_write.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/${TAG}_${N}_sq -- python3 tools/profile_workload.py $W 100 > gpurun_out/${TAG}_${N}_sq.log 2>&1
done
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout -k 10 200 python3 bench.py --dx-mode 1 --only-main > gpurun_out/${TAG}_bench_mfma.json 2> gpurun_out/${TAG}_bench_mfma.err
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-400
