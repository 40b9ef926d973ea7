#!/bin/bash
# dev: correctness check + the two c3_rgb bench legs in one gpurun call
set -o pipefail
timeout -k 10 300 python tools/dev/bt_check.py > gpurun_out/bt_check.log 2>&1 || { tail -5 gpurun_out/bt_check.log; exit 1; }
grep -c "bad=0" gpurun_out/bt_check.log; grep -v "bad=0" gpurun_out/bt_check.log | grep -v tiles | head
timeout -k 10 200 python bench.py --workload c3_rgb --no-cpu-baseline --no-secondary > gpurun_out/bt_b1.json 2> gpurun_out/bt_b1.err &&
timeout -k 10 200 python bench.py --workload c3_rgb --batch 8 --no-cpu-baseline --no-secondary > gpurun_out/bt_b8.json 2>> gpurun_out/bt_b1.err
python - <<'PY'
import json
for f in ("gpurun_out/bt_b1.json","gpurun_out/bt_b8.json"):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(j["config"]["path"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["iteration_kernels_us"], j["sane"])
PY
