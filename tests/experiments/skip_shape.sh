set -e
cd "$GRAFT_REPO_ROOT"
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset RPM_HIP_LIB; else export RPM_HIP_LIB=$PWD/lpopc_amd/csrc/librpm_exp_skip$v.so; fi
  timeout -k 10 120 python bench.py --only-main --persistent --steps 100 --warmup 20 > gpurun_out/r03_skip$v.json 2> gpurun_out/r03_skip$v.err || echo "variant $v failed"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03_skip$v.json").read().strip().splitlines()[-1])
    print("variant $v", round(d["value"]), "pairs/s", round(d["roofline"]["avg_launch_us"],1), "us")
except Exception as ex:
    print("variant $v", ex)
PY
done
export RPM_HIP_LIB=$PWD/lpopc_amd/csrc/librpm_exp_skip1.so
python -m pytest tests/test_gpu_parity.py -q -k "persistent" 2>&1 | tail -2
