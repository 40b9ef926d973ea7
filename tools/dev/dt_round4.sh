#!/bin/bash
# dev (round 4): k_ibp_dtile after the spill fixes -- parity tests, then the three c3_f4 bench legs
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "frame_fraction or window_golden_phase or full_size_x4 or two_launch" > gpurun_out/dt4_tests.log 2>&1
rc=$?; tail -5 gpurun_out/dt4_tests.log
[ $rc -ne 0 ] && exit $rc
for spec in "c3_f4 1" "c3_f4_float 1" "c3_f4 8"; do
  set -- $spec
  timeout -k 10 300 python3 bench.py --workload $1 --batch $2 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'B=$2', d['config']['path'], 'ms/step', d['ms_per_step'], 'iter us', d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'], d['sane'])" || exit 1
done 2>&1 | tee gpurun_out/dt4_bench.log
