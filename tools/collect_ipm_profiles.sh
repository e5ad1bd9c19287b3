#!/bin/bash
# rocprofv3 evidence for row f-2 (the device interior-point solver); run on the GPU box:
#   gpurun -- 'bash tools/collect_ipm_profiles.sh r02'
# Counter passes are separate runs with --kernel-trace only (gpurun refuses --pmc beside other trace domains).
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# the 1024-instance quadrotor sweep (BASELINE config 5), solved on the device
timeout -k 10 200 python3 tools/bench_ipm.py 1024 0 1 > gpurun_out/${TAG}_ipm_sweep.json 2> gpurun_out/${TAG}_ipm_sweep.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_ipm_sweep_stats -- python3 tools/bench_ipm.py 1024 0 1 > gpurun_out/${TAG}_ipm_sweep_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d gpurun_out/${TAG}_ipm_sweep_mfma -- python3 tools/bench_ipm.py 1024 0 1 > gpurun_out/${TAG}_ipm_sweep_mfma.log 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_ipm_sweep_fetch -- python3 tools/bench_ipm.py 1024 0 1 > gpurun_out/${TAG}_ipm_sweep_fetch.log 2>&1 || true
# the metric problem (Delta-III 4 x 64 x 16) solved on the device from lpopc's default guess
timeout -k 10 300 python3 tools/ipm_delta3.py 64 16 3000 -1 > gpurun_out/${TAG}_ipm_delta3.json 2> gpurun_out/${TAG}_ipm_delta3.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_ipm_delta3_stats -- python3 tools/ipm_delta3.py 64 16 3000 -1 > gpurun_out/${TAG}_ipm_delta3_stats.log 2>&1
head -c 600 gpurun_out/${TAG}_ipm_delta3.json
