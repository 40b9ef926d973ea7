#!/bin/bash
# dev: c3_rgb at B = 1 and B = 8 with several builds on one box: bash tools/dev/bt_ab.sh default path/to/lib.so ...
for lib in "$@"; do
    if [ "$lib" = default ]; then unset SRX_LIB; else export SRX_LIB=$PWD/$lib; fi
    for b in 1 8; do
        timeout -k 10 200 python3 bench.py --workload c3_rgb --batch $b --no-cpu-baseline --no-secondary --steps 4 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'B=$b', d['config']['path'], 'ms/step', d['ms_per_step'], 'iter us', d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'], d['sane'])"
    done
done
