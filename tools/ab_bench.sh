#!/bin/bash
# ab_bench.sh WORKLOAD LIB... -- the same bench leg with several builds of libsrx.so on ONE box (box to box the same build varies by ~6 %):
#   gpurun -- 'bash tools/ab_bench.sh c2 default enph459-super-resolution_amd/build/libsrx_x.so'
WL=$1; shift
for lib in "$@"; do
    if [ "$lib" = default ]; then unset SRX_LIB; else export SRX_LIB=$lib; fi
    for rep in 1 2; do
        python3 bench.py --workload $WL --no-cpu-baseline --no-secondary --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['config']['path'], 'ms/step', d['ms_per_step'], 'iter us', d['roofline']['iteration_kernels_us'], 'frac', d['roofline']['frac'])"
    done
done
