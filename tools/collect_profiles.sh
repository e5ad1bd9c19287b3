#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ for the default bench command; run on the GPU box:
#   gpurun -- 'bash tools/collect_profiles.sh r01'
# then `python tools/summarize_profiles.py r01` (here) copies the summaries into profiles/.
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="python3 bench.py --profile --steps 200 --warmup 20"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $A > gpurun_out/${TAG}_stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- $A --no-graph > gpurun_out/${TAG}_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- $A --no-graph > gpurun_out/${TAG}_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/${TAG}_sq -- $A --no-graph > gpurun_out/${TAG}_sq.log 2>&1
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-400
