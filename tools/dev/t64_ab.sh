#!/bin/bash
# dev: float64 tile kernels with T_HR = 32 / 48 / 64 -- parity subset, then the float64 legs that still run on them
for ts in ${VARIANTS:-32 48 64}; do
  cp tools/dev/libs/libsrx_t$ts.so enph459-super-resolution_amd/sr_mi355x/libsrx.so
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "f64 and (fused_path or ibp_c1 or ibp_c2_small or shift_and_add or forward_back or real_crops or multi_tile or ragged)" 2>&1 | tail -1
  for wl in c3_rgb c3_f4; do
  timeout -k 10 200 python3 bench.py --workload $wl --precision f64 --no-cpu-baseline --no-secondary --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('T_HR=$ts $wl', d['config']['path'], 'ms/step', d['ms_per_step'], d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'], d['sane'])" || exit 1
  done
done 2>&1 | tee gpurun_out/t64_ab.log
