#!/bin/bash
# dev: k_ibp_ctile<double> with regions of 32 / 40 / 48 rows (held state) -- parity on the frame golden, then the c3_mono f64 leg
for nr in ${VARIANTS}; do
  cp tools/dev/libs/libsrx_nr$nr.so enph459-super-resolution_amd/sr_mi355x/libsrx.so
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "frame_80_iterations_golden and f64" 2>&1 | tail -1
  timeout -k 10 200 python3 bench.py --workload c3_mono --precision f64 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NR=$nr', d['config']['path'], 'ms/step', d['ms_per_step'], d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'], d['sane'])" || exit 1
done 2>&1 | tee gpurun_out/ct_ab.log
