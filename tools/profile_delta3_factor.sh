#!/bin/bash
# SQ / cache counters of the metric problem's device solve (the left-looking factor kernel is the thing looked at); run on the GPU box
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="python3 tools/ipm_delta3.py 64 16 60 -1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/r03d3_sq -- $A > gpurun_out/r03d3_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d gpurun_out/r03d3_sq2 -- $A > gpurun_out/r03d3_sq2.log 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r03d3_tc -- $A > gpurun_out/r03d3_tc.log 2>&1 || true
echo done
