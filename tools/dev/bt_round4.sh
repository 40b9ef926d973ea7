#!/bin/bash
# dev (round 4): the path-B window kernels with both PSF forms -- parity tests, then the four c3_rgb bench legs
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "frame_shift or window_golden_rgb or fused_path or real_crops or 80_iterations_multi or full_frame_paths or full_size_rgb" > gpurun_out/bt4_tests.log 2>&1
rc=$?; tail -15 gpurun_out/bt4_tests.log
[ $rc -ne 0 ] && exit $rc
for wl in c3_rgb c3_rgb_measured; do
  for b in 1 8; do
    timeout -k 10 200 python3 bench.py --workload $wl --batch $b --no-cpu-baseline --no-secondary --steps 4 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', 'B=$b', d['config']['path'], 'ms/step', d['ms_per_step'], 'iter us', d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'], d['sane'])" || exit 1
  done
done 2>&1 | tee gpurun_out/bt4_bench.log
