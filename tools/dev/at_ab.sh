#!/bin/bash
for lib in "$@"; do
    if [ "$lib" = default ]; then unset SRX_LIB; else export SRX_LIB=$PWD/$lib; fi
    timeout -k 10 300 python3 bench.py --workload c3_f4 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['config']['path'], 'ms/step', d['ms_per_step'], 'iter us', d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'], d['sane'])"
done
