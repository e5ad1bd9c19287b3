#!/bin/bash
# Ipopt's gradient-based NLP scaling on the device solver (option nlp_scaling), Delta-III meshes with and without
set -e
out=gpurun_out/scaling_device.txt
: > $out
for m in "4 8" "8 8" "16 8" "4 16" "16 16" "32 16" "64 16"; do
  for sc in 0 1; do
    echo "mesh $m nlp_scaling $sc" >> $out
    python tools/ipm_delta3.py $m 3000 -1 nlp_scaling=$sc 2>/dev/null | tail -1 >> $out
  done
done
